// Host-side file I/O helpers (no device code): report text, the length table, HDF5 chunk decoding.
//
// Text output of the quantify reports: `locus <haplotypes> total [notes]`
// tables with every number in its shortest round-trip form, i.e. exactly what str(numpy.float64) /
// repr(float) print in the reference's writers (emase/EMfactory.py:289-380).  At 120k isoforms + 48k
// genes the four reports hold 1.5 M numbers; formatting them in the interpreter took ~1 s per sample,
// as long as everything else of `gbrs quantify` on the device path together.
#ifdef GBRS_HOST_ONLY
// CPU-only build of this file for the AddressSanitizer / UBSan test (tests/test_hostio_sanitizers.py:
// g++ -x c++ -DGBRS_HOST_ONLY -fsanitize=address,undefined): nothing here touches the device, so the
// HIP runtime headers are not needed; the test driver supplies gbrs::fail.
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/gbrs_hip.h"
namespace gbrs { int fail(int status, const char *fmt, ...); }
#else
#include "common.h"
#endif

#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <string_view>
#include <thread>
#include <unordered_map>

namespace gbrs {

// repr(float): shortest digits that round-trip; fixed notation for 1e-4 <= |x| < 1e16, else d[.ddd]e+XX
// with at least two exponent digits; a ".0" is appended to integral fixed values.  Returns the length.
static int format_repr(double v, char *out) {
    if (std::isnan(v)) { std::memcpy(out, "nan", 3); return 3; }
    if (std::isinf(v)) {
        if (v < 0) { std::memcpy(out, "-inf", 4); return 4; }
        std::memcpy(out, "inf", 3);
        return 3;
    }
    char *p = out;
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (v == 0.0) { std::memcpy(p, "0.0", 3); return (int)(p - out) + 3; }
    char sci[40];
    const auto res = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    // sci = d[.ddd]e[+-]XX[X]
    char digits[24];
    int nd = 0;
    const char *q = sci;
    for (; q < res.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    ++q;                                   // past 'e'
    const bool eneg = *q == '-';
    if (*q == '+' || *q == '-') ++q;
    int e10 = 0;
    for (; q < res.ptr; ++q) e10 = e10 * 10 + (*q - '0');
    if (eneg) e10 = -e10;
    const int decpt = e10 + 1;             // value = 0.d1d2... x 10^decpt
    if (decpt <= -4 || decpt > 16) {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, digits + 1, nd - 1);
            p += nd - 1;
        }
        *p++ = 'e';
        int e = decpt - 1;
        *p++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) { *p++ = (char)('0' + e / 100); e %= 100; *p++ = (char)('0' + e / 10); *p++ = (char)('0' + e % 10); }
        else { *p++ = (char)('0' + e / 10); *p++ = (char)('0' + e % 10); }
    } else if (decpt <= 0) {
        *p++ = '0'; *p++ = '.';
        for (int k = 0; k < -decpt; ++k) *p++ = '0';
        std::memcpy(p, digits, nd);
        p += nd;
    } else if (decpt >= nd) {
        std::memcpy(p, digits, nd);
        p += nd;
        for (int k = nd; k < decpt; ++k) *p++ = '0';
        *p++ = '.'; *p++ = '0';
    } else {
        std::memcpy(p, digits, decpt);
        p += decpt;
        *p++ = '.';
        std::memcpy(p, digits + decpt, nd - decpt);
        p += nd - decpt;
    }
    return (int)(p - out);
}

}  // namespace gbrs

// ---- parallel decode of HDF5 chunks (deflate [+ byte shuffle]) ------------------------------------------
namespace gbrs {

typedef int (*uncompress_fn)(unsigned char *, unsigned long *, const unsigned char *, unsigned long);
typedef void *(*ld_alloc_fn)(void);
typedef void (*ld_free_fn)(void *);
typedef int (*ld_inflate_fn)(void *, const void *, size_t, void *, size_t, size_t *);
// zlib's streaming interface, for raw deflate streams (zip members): z_stream is opaque here except for the
// leading fields the caller sets, so the struct below mirrors zlib.h's layout on LP64
struct ZStream {
    const unsigned char *next_in; unsigned int avail_in; unsigned long total_in;
    unsigned char *next_out; unsigned int avail_out; unsigned long total_out;
    const char *msg; void *state; void *zalloc; void *zfree; void *opaque; int data_type; unsigned long adler; unsigned long reserved;
};
typedef int (*z_init2_fn)(ZStream *, int, const char *, int);
typedef int (*z_inflate_fn)(ZStream *, int);
typedef int (*z_end_fn)(ZStream *);
typedef uint32_t (*ld_crc32_fn)(uint32_t, const void *, size_t);
typedef unsigned long (*z_crc32_fn)(unsigned long, const unsigned char *, unsigned int);

struct Inflaters {
    uncompress_fn z_uncompress = nullptr;
    ld_alloc_fn ld_alloc = nullptr;
    ld_free_fn ld_free = nullptr;
    ld_inflate_fn ld_inflate = nullptr;
    ld_inflate_fn ld_inflate_raw = nullptr;       // libdeflate_deflate_decompress: no zlib wrapper (zip members)
    z_init2_fn z_init2 = nullptr;
    z_inflate_fn z_inflate = nullptr;
    z_end_fn z_end = nullptr;
    ld_crc32_fn ld_crc32 = nullptr;               // libdeflate_crc32: carry-less multiply, several GB/s per core
    z_crc32_fn z_crc32 = nullptr;
    Inflaters() {
        // libdeflate (about three times zlib's inflate speed) when the machine has it, zlib otherwise;
        // both are looked up at run time so the library has no link-time dependency on either
        const char *ld_names[] = {std::getenv("GBRS_LIBDEFLATE"), "libdeflate.so.0", "libdeflate.so", "/opt/conda/lib/libdeflate.so.0"};
        for (const char *n : ld_names) {
            if (!n || !*n) continue;
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                ld_alloc = (ld_alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
                ld_free = (ld_free_fn)dlsym(h, "libdeflate_free_decompressor");
                ld_inflate = (ld_inflate_fn)dlsym(h, "libdeflate_zlib_decompress");
                ld_inflate_raw = (ld_inflate_fn)dlsym(h, "libdeflate_deflate_decompress");
                ld_crc32 = (ld_crc32_fn)dlsym(h, "libdeflate_crc32");
                if (ld_alloc && ld_free && ld_inflate && ld_inflate_raw) break;
                ld_alloc = nullptr; ld_free = nullptr; ld_inflate = nullptr; ld_inflate_raw = nullptr;
            }
        }
        const char *z_names[] = {"libz.so.1", "libz.so", "/opt/conda/lib/libz.so.1"};
        for (const char *n : z_names)
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                z_uncompress = (uncompress_fn)dlsym(h, "uncompress");
                z_init2 = (z_init2_fn)dlsym(h, "inflateInit2_");
                z_inflate = (z_inflate_fn)dlsym(h, "inflate");
                z_end = (z_end_fn)dlsym(h, "inflateEnd");
                z_crc32 = (z_crc32_fn)dlsym(h, "crc32");
                if (z_uncompress) break;
            }
    }
};

static const Inflaters &inflaters() {
    static const Inflaters inf;
    return inf;
}

// one raw deflate stream -> exactly out_bytes bytes; ld = this thread's libdeflate decompressor or null
static bool inflate_raw(const Inflaters &inf, void *ld, const unsigned char *in, size_t in_bytes, unsigned char *out,
                        size_t out_bytes) {
    if (ld && inf.ld_inflate_raw) {
        size_t n = 0;
        if (inf.ld_inflate_raw(ld, in, in_bytes, out, out_bytes, &n) == 0 && n == out_bytes) return true;
    }
    if (inf.z_init2 && inf.z_inflate && inf.z_end && in_bytes <= 0xFFFFFFFFu && out_bytes <= 0xFFFFFFFFu) {
        ZStream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inf.z_init2(&zs, -15, "1.2.11", (int)sizeof(ZStream)) != 0) return false;
        zs.next_in = in; zs.avail_in = (unsigned int)in_bytes;
        zs.next_out = out; zs.avail_out = (unsigned int)out_bytes;
        const int rc = inf.z_inflate(&zs, 4 /* Z_FINISH */);
        const bool ok = rc == 1 /* Z_STREAM_END */ && zs.total_out == out_bytes;
        inf.z_end(&zs);
        return ok;
    }
    return false;
}

// CRC-32 of a zip member's plain bytes (what numpy.load / zipfile check on every access and report as BadZipFile:
// the reference inherits that): libdeflate's when the machine has it, zlib's otherwise, a table walk as the last resort
static uint32_t member_crc32(const Inflaters &inf, const unsigned char *p, size_t n) {
    if (inf.ld_crc32) return inf.ld_crc32(0, p, n);
    if (inf.z_crc32) {
        unsigned long c = 0;
        while (n) {
            const unsigned int part = (unsigned int)std::min<size_t>(n, 1u << 30);
            c = inf.z_crc32(c, p, part);
            p += part;
            n -= part;
        }
        return (uint32_t)c;
    }
    static const std::array<uint32_t, 256> table = [] {
        std::array<uint32_t, 256> t{};
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[i] = c;
        }
        return t;
    }();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const unsigned char *p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

}  // namespace gbrs

extern "C" {

// 1 = libdeflate, 2 = zlib, 0 = neither could be loaded
int gbrs_inflate_backend(void) {
    const gbrs::Inflaters &inf = gbrs::inflaters();
    return inf.ld_inflate ? 1 : (inf.z_uncompress ? 2 : 0);
}

int gbrs_decode_chunks(const char *path, int64_t n_chunks, const uint64_t *file_addr, const uint64_t *stored_bytes,
                       const uint64_t *elem_start, const uint32_t *filter_mask, uint64_t chunk_elems,
                       uint32_t elem_size, uint64_t n_elems, int32_t shuffle_pos, int32_t deflate_pos, void *out,
                       int32_t threads) {
    using gbrs::fail;
    if (!path || !out || n_chunks < 0 || (n_chunks && (!file_addr || !stored_bytes || !elem_start || !filter_mask)) ||
        chunk_elems == 0 || elem_size == 0 || elem_size > 16)
        return fail(GBRS_ERR_INVALID, "bad argument");
    const gbrs::Inflaters &inf = gbrs::inflaters();
    if (deflate_pos >= 0 && !inf.ld_inflate && !inf.z_uncompress)
        return fail(GBRS_ERR_UNSUPPORTED, "neither libdeflate nor zlib could be loaded");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(GBRS_ERR_INVALID, "cannot open %s", path);
    for (int64_t k = 0; k < n_chunks; ++k)
        if (elem_start[k] >= n_elems) { close(fd); return fail(GBRS_ERR_INVALID, "chunk %lld starts outside the dataset", (long long)k); }   // (also: chunks of an empty dataset)
    unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (const char *e = std::getenv("GBRS_IO_THREADS"); threads <= 0 && e && std::atoi(e) > 0) nt = (unsigned)std::atoi(e);
    nt = std::max(1u, std::min({nt, 64u, (unsigned)std::max<int64_t>(n_chunks, 1)}));
    const size_t chunk_bytes = (size_t)chunk_elems * elem_size;
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        std::vector<unsigned char> raw, plain(chunk_bytes);
        void *ld = inf.ld_inflate ? inf.ld_alloc() : nullptr;
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_chunks || failed.load()) break;
            raw.resize(stored_bytes[k]);
            size_t got = 0;
            while (got < raw.size()) {
                const ssize_t r = pread(fd, raw.data() + got, raw.size() - got, (off_t)(file_addr[k] + got));
                if (r <= 0) { failed = 1; break; }
                got += (size_t)r;
            }
            if (failed.load()) break;
            const uint64_t count = std::min<uint64_t>(chunk_elems, n_elems - elem_start[k]);
            unsigned char *dst = (unsigned char *)out + (size_t)elem_start[k] * elem_size;
            // filters were applied in pipeline order when the chunk was written (shuffle, then deflate):
            // undo them back to front; a set bit p in the chunk's mask means filter p was skipped
            const unsigned char *cur = raw.data();
            size_t cur_bytes = raw.size();
            if (deflate_pos >= 0 && !(filter_mask[k] & (1u << deflate_pos))) {
                bool ok = false;
                if (ld) {
                    size_t n = 0;
                    ok = inf.ld_inflate(ld, cur, cur_bytes, plain.data(), chunk_bytes, &n) == 0 && n == chunk_bytes;
                }
                if (!ok && inf.z_uncompress) {
                    unsigned long n = (unsigned long)chunk_bytes;
                    ok = inf.z_uncompress(plain.data(), &n, cur, (unsigned long)cur_bytes) == 0 && n == chunk_bytes;
                }
                if (!ok) { failed = 2; break; }
                cur = plain.data();
                cur_bytes = chunk_bytes;
            }
            if (cur_bytes < chunk_bytes) { failed = 3; break; }
            if (shuffle_pos >= 0 && !(filter_mask[k] & (1u << shuffle_pos))) {
                // byte plane b of the chunk holds byte b of every element
                if (elem_size == 4) {
                    const unsigned char *p0 = cur, *p1 = cur + chunk_elems, *p2 = cur + 2 * chunk_elems, *p3 = cur + 3 * chunk_elems;
                    uint32_t *d = (uint32_t *)dst;
                    for (uint64_t i = 0; i < count; ++i)
                        d[i] = (uint32_t)p0[i] | ((uint32_t)p1[i] << 8) | ((uint32_t)p2[i] << 16) | ((uint32_t)p3[i] << 24);
                } else {
                    for (uint64_t i = 0; i < count; ++i)
                        for (uint32_t b = 0; b < elem_size; ++b) dst[i * elem_size + b] = cur[(size_t)b * chunk_elems + i];
                }
            } else {
                std::memcpy(dst, cur, (size_t)count * elem_size);
            }
        }
        if (ld) inf.ld_free(ld);
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    close(fd);
    if (failed.load() == 1) return fail(GBRS_ERR_INVALID, "short read from %s", path);
    if (failed.load() == 2) return fail(GBRS_ERR_INVALID, "a chunk of %s does not inflate to the chunk size", path);
    if (failed.load() == 3) return fail(GBRS_ERR_INVALID, "a chunk of %s is shorter than the chunk size", path);
    return GBRS_OK;
}

int gbrs_zip_directory(const uint8_t *buf, uint64_t len, uint64_t cap, uint16_t *method, uint64_t *csize,
                       uint64_t *usize, uint64_t *header_off, uint32_t *crc32, char *names, uint64_t names_cap,
                       uint64_t *n_members, uint64_t *names_len) {
    using gbrs::fail;
    using gbrs::rd16; using gbrs::rd32; using gbrs::rd64;
    if (!buf || !n_members || !names_len) return fail(GBRS_ERR_INVALID, "bad argument");
    *n_members = 0;
    *names_len = 0;
    if (len < 22) return fail(GBRS_ERR_INVALID, "not a zip file");
    // end-of-central-directory record: the last 22 bytes + an optional comment of up to 65,535 bytes
    const uint64_t lowest = len > 22 + 65535 ? len - 22 - 65535 : 0;
    uint64_t eocd = len - 22;
    for (;; --eocd) {
        if (rd32(buf + eocd) == 0x06054b50u) break;
        if (eocd == lowest) return fail(GBRS_ERR_INVALID, "no end-of-central-directory record");
    }
    const uint16_t disk = rd16(buf + eocd + 4), cd_disk = rd16(buf + eocd + 6), n_here = rd16(buf + eocd + 8),
                   n_total = rd16(buf + eocd + 10);
    uint64_t cd_off = rd32(buf + eocd + 16), n = n_total;
    if (disk || cd_disk || n_here != n_total) return fail(GBRS_ERR_UNSUPPORTED, "multi-disk zip file");
    if (n_total == 0xFFFFu || cd_off == 0xFFFFFFFFu) {
        // zip64: the locator sits right before the end record and points at the zip64 end record
        if (eocd < 20 || rd32(buf + eocd - 20) != 0x07064b50u) return fail(GBRS_ERR_UNSUPPORTED, "zip64 locator missing");
        const uint64_t z = rd64(buf + eocd - 20 + 8);
        if (z > len || len - z < 56 || rd32(buf + z) != 0x06064b50u) return fail(GBRS_ERR_INVALID, "bad zip64 end record");   // (no wrapping sums on file-supplied offsets)
        n = rd64(buf + z + 32);
        cd_off = rd64(buf + z + 48);
    }
    uint64_t pos = cd_off, nlen_total = 0;
    for (uint64_t k = 0; k < n; ++k) {
        if (pos > len || len - pos < 46 || rd32(buf + pos) != 0x02014b50u) return fail(GBRS_ERR_INVALID, "bad central directory entry %llu", (unsigned long long)k);
        const uint16_t m = rd16(buf + pos + 10), nlen = rd16(buf + pos + 28), xlen = rd16(buf + pos + 30), clen = rd16(buf + pos + 32);
        uint64_t cs = rd32(buf + pos + 20), us = rd32(buf + pos + 24), ho = rd32(buf + pos + 42);
        if (len - pos - 46 < (uint64_t)nlen + xlen + clen) return fail(GBRS_ERR_INVALID, "central directory runs past the file");
        if (cs == 0xFFFFFFFFu || us == 0xFFFFFFFFu || ho == 0xFFFFFFFFu) {
            // zip64 extended information: the 64-bit values of the saturated fields, in this order
            const uint8_t *x = buf + pos + 46 + nlen;
            uint64_t at = 0;
            bool found = false;
            while (at + 4 <= xlen) {
                const uint16_t tag = rd16(x + at), ln = rd16(x + at + 2);
                if (at + 4 + ln > xlen) break;
                if (tag == 1) {
                    uint64_t q = at + 4;
                    const uint64_t end = at + 4 + ln;
                    if (us == 0xFFFFFFFFu && q + 8 <= end) { us = rd64(x + q); q += 8; }
                    if (cs == 0xFFFFFFFFu && q + 8 <= end) { cs = rd64(x + q); q += 8; }
                    if (ho == 0xFFFFFFFFu && q + 8 <= end) { ho = rd64(x + q); q += 8; }
                    found = true;
                    break;
                }
                at += 4 + (uint64_t)ln;
            }
            if (!found) return fail(GBRS_ERR_INVALID, "zip64 sizes missing for entry %llu", (unsigned long long)k);
        }
        if (k < cap) {
            if (!method || !csize || !usize || !header_off) return fail(GBRS_ERR_INVALID, "bad argument");
            method[k] = m; csize[k] = cs; usize[k] = us; header_off[k] = ho;
            if (crc32) crc32[k] = rd32(buf + pos + 16);
            if (names && nlen_total + nlen + 1 <= names_cap) {
                std::memcpy(names + nlen_total, buf + pos + 46, nlen);
                names[nlen_total + nlen] = '\n';
            }
        }
        nlen_total += (uint64_t)nlen + 1;
        pos += 46 + (uint64_t)nlen + xlen + clen;
    }
    *n_members = n;
    *names_len = nlen_total;
    return GBRS_OK;
}

int gbrs_zip_read_members(const uint8_t *buf, uint64_t len, int64_t n, const uint64_t *header_off, const uint16_t *method,
                          const uint64_t *csize, const uint64_t *usize, const uint32_t *crc32, uint8_t *const *out, int32_t threads) {
    using gbrs::fail;
    using gbrs::rd16; using gbrs::rd32;
    if (!buf || n < 0 || (n && (!header_off || !method || !csize || !usize || !out))) return fail(GBRS_ERR_INVALID, "bad argument");
    const gbrs::Inflaters &inf = gbrs::inflaters();
    unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (const char *e = std::getenv("GBRS_IO_THREADS"); threads <= 0 && e && std::atoi(e) > 0) nt = (unsigned)std::atoi(e);
    nt = std::max(1u, std::min({nt, 64u, (unsigned)std::max<int64_t>(n, 1)}));
    // largest members first: a member is one deflate stream, so the biggest one bounds the wall time
    std::vector<int64_t> order((size_t)n);
    for (int64_t k = 0; k < n; ++k) order[(size_t)k] = k;
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return csize[a] > csize[b]; });
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        void *ld = inf.ld_inflate_raw ? inf.ld_alloc() : nullptr;
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n || failed.load()) break;
            const int64_t k = order[(size_t)i];
            const uint64_t ho = header_off[k];
            if (!out[k] || ho > len || len - ho < 30 || rd32(buf + ho) != 0x04034b50u) { failed = 1; break; }
            const uint64_t data = ho + 30 + rd16(buf + ho + 26) + rd16(buf + ho + 28);     // ho <= len - 30: cannot wrap
            if (data > len || csize[k] > len - data) { failed = 1; break; }
            if (method[k] == 0) {
                if (csize[k] != usize[k]) { failed = 2; break; }
                std::memcpy(out[k], buf + data, usize[k]);
            } else if (method[k] == 8) {
                if (!gbrs::inflate_raw(inf, ld, buf + data, csize[k], out[k], usize[k])) { failed = 2; break; }
            } else {
                failed = 3;
                break;
            }
            // checked on the thread that has just produced the bytes (still in its cache)
            if (crc32 && gbrs::member_crc32(inf, out[k], usize[k]) != crc32[k]) { failed = 4; break; }
        }
        if (ld) inf.ld_free(ld);
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (failed.load() == 1) return fail(GBRS_ERR_INVALID, "bad local header in the zip file");
    if (failed.load() == 2) return fail(GBRS_ERR_INVALID, "a member does not inflate to its recorded size");
    if (failed.load() == 3) return fail(GBRS_ERR_UNSUPPORTED, "compression method other than stored / deflate");
    if (failed.load() == 4) return fail(GBRS_ERR_INVALID, "a member fails its CRC-32");
    return GBRS_OK;
}

int gbrs_npz_stack(const uint8_t *buf, uint64_t len, int64_t n, const uint64_t *header_off, const uint16_t *method,
                   const uint64_t *csize, const uint64_t *usize, const uint32_t *crc32, const uint8_t *npy_header,
                   uint64_t npy_header_len, uint64_t item_bytes, uint8_t *out, uint8_t *needs_fallback, int32_t threads) {
    using gbrs::fail;
    using gbrs::rd16; using gbrs::rd32;
    if (!buf || n < 0 || (n && (!header_off || !method || !csize || !usize || !out || !needs_fallback)) || !npy_header)
        return fail(GBRS_ERR_INVALID, "bad argument");
    const gbrs::Inflaters &inf = gbrs::inflaters();
    unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (const char *e = std::getenv("GBRS_IO_THREADS"); threads <= 0 && e && std::atoi(e) > 0) nt = (unsigned)std::atoi(e);
    nt = std::max(1u, std::min({nt, 64u, (unsigned)std::max<int64_t>(n / 128, 1)}));
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    constexpr int64_t GRAIN = 64;
    auto work = [&]() {
        std::vector<unsigned char> plain;
        void *ld = inf.ld_inflate_raw ? inf.ld_alloc() : nullptr;
        for (;;) {
            const int64_t k0 = next.fetch_add(GRAIN);
            if (k0 >= n || failed.load()) break;
            for (int64_t k = k0; k < std::min(n, k0 + GRAIN); ++k) {
                needs_fallback[k] = 1;
                const uint64_t ho = header_off[k];
                if (ho > len || len - ho < 30 || rd32(buf + ho) != 0x04034b50u) { failed = 1; break; }
                const uint64_t data = ho + 30 + rd16(buf + ho + 26) + rd16(buf + ho + 28);
                if (data > len || csize[k] > len - data) { failed = 1; break; }
                if (usize[k] != npy_header_len + item_bytes) continue;      // another shape or dtype: the caller's business
                const unsigned char *img = nullptr;
                if (method[k] == 0) {
                    if (csize[k] != usize[k]) continue;
                    img = buf + data;
                } else if (method[k] == 8) {
                    plain.resize(usize[k]);
                    if (!gbrs::inflate_raw(inf, ld, buf + data, csize[k], plain.data(), usize[k])) continue;   // python decides
                    img = plain.data();
                } else {
                    continue;
                }
                if (std::memcmp(img, npy_header, npy_header_len) != 0) continue;
                if (crc32 && gbrs::member_crc32(inf, img, usize[k]) != crc32[k]) continue;   // the caller's reader reports it by name
                std::memcpy(out + (size_t)k * item_bytes, img + npy_header_len, item_bytes);
                needs_fallback[k] = 0;
            }
        }
        if (ld) inf.ld_free(ld);
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    if (failed.load()) return fail(GBRS_ERR_INVALID, "bad local header in the zip file");
    return GBRS_OK;
}

int gbrs_parse_number_table(const char *text, int64_t text_len, int64_t n_rows, int32_t n_cols, double *out) {
    using gbrs::fail;
    if (!text || text_len < 0 || n_rows < 0 || n_cols <= 0 || (n_rows && !out)) return fail(GBRS_ERR_INVALID, "bad argument");
    const char *p = text, *end = text + text_len;
    for (int64_t r = 0; r < n_rows; ++r) {
        while (p < end && *p != '\t' && *p != '\n') ++p;                 // the row's label
        for (int32_t c = 0; c < n_cols; ++c) {
            if (p >= end || *p != '\t') return 1;
            ++p;
            double v;
            const auto res = std::from_chars(p, end, v);
            if (res.ec != std::errc() || res.ptr == p) return 1;          // inf / nan spellings, blanks, ...: the caller's parser
            out[r * n_cols + c] = v;
            p = res.ptr;
        }
        if (p < end && *p == '\r') ++p;
        if (p < end) {
            if (*p != '\n') return 1;
            ++p;
        } else if (r + 1 < n_rows) {
            return 1;
        }
    }
    return p == end ? GBRS_OK : 1;
}

int gbrs_format_double(double v, char *out32) {
    if (!out32) return gbrs::fail(GBRS_ERR_INVALID, "out is NULL");
    const int n = gbrs::format_repr(v, out32);
    out32[n] = '\0';
    return n;
}

int gbrs_write_locus_table(const char *path, const char *header_line, const double *values, int64_t n_rows,
                           int32_t n_cols, int64_t row_stride, int64_t col_stride, const double *totals,
                           const char *names, const int64_t *name_off, const char *notes,
                           const int64_t *note_off, const int64_t *order) {
    using gbrs::fail;
    if (!path || !header_line || ((!values || !totals) && n_rows > 0) || !names || !name_off || n_rows < 0 ||
        n_cols < 0 || ((notes == nullptr) != (note_off == nullptr)))
        return fail(GBRS_ERR_INVALID, "bad argument");
    for (int64_t k = 0; k < n_rows && order; ++k)
        if (order[k] < 0 || order[k] >= n_rows)
            return fail(GBRS_ERR_INVALID, "row order entry %lld out of range", (long long)order[k]);
    // rows are formatted in contiguous slices on a few threads (std::to_chars costs ~0.2 us a number),
    // then written out in order
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("GBRS_IO_THREADS"); e && std::atoi(e) > 0) nt = (unsigned)std::atoi(e);
    nt = std::max(1u, std::min({nt, 16u, (unsigned)(n_rows / 4096 + 1)}));
    std::vector<std::vector<char>> parts(nt);
    auto work = [&](unsigned t) {
        const int64_t k0 = n_rows * t / nt, k1 = n_rows * (t + 1) / nt;
        std::vector<char> &buf = parts[t];
        buf.reserve((size_t)(k1 - k0) * (16 + ((size_t)n_cols + 1) * 20));
        char num[40];
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t r = order ? order[k] : k;
            buf.insert(buf.end(), names + name_off[r], names + name_off[r + 1]);
            for (int c = 0; c <= n_cols; ++c) {
                buf.push_back('\t');
                const double x = c < n_cols ? values[r * row_stride + c * col_stride] : totals[r];
                const int n = gbrs::format_repr(x, num);
                buf.insert(buf.end(), num, num + n);
            }
            if (notes) {
                buf.push_back('\t');
                buf.insert(buf.end(), notes + note_off[r], notes + note_off[r + 1]);
            }
            buf.push_back('\n');
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    FILE *fh = std::fopen(path, "w");
    if (!fh) return fail(GBRS_ERR_INVALID, "cannot open %s for writing", path);
    bool ok = std::fwrite(header_line, 1, std::strlen(header_line), fh) == std::strlen(header_line);
    for (unsigned t = 0; t < nt && ok; ++t)
        ok = parts[t].empty() || std::fwrite(parts[t].data(), 1, parts[t].size(), fh) == parts[t].size();
    if (std::fclose(fh) != 0 || !ok) return fail(GBRS_ERR_INVALID, "write to %s failed", path);
    return GBRS_OK;
}

// `<locus>_<haplotype> TAB <length>` table -> effective lengths (EMfactory.py:60-94).  Returns 0 when every
// line was plain (one underscore in the key, known names, a number from_chars takes whole), 1 when some
// line needs the interpreter's more permissive parsing / error reporting: the caller then re-reads the
// file with the line-by-line path, which raises what the reference raises.
int gbrs_parse_length_table(const char *text, int64_t text_len, const char *names, const int64_t *name_off,
                            int64_t n_loci, const char *haps, const int64_t *hap_off, int32_t n_haps,
                            double read_length, double *eff_out) {
    using gbrs::fail;
    if (!text || !names || !name_off || !haps || !hap_off || !eff_out || n_loci < 1 || n_haps < 1 || text_len < 0)
        return fail(GBRS_ERR_INVALID, "bad argument");
    std::unordered_map<std::string_view, int64_t> locus_id;
    locus_id.reserve((size_t)n_loci * 2);
    for (int64_t l = 0; l < n_loci; ++l)
        locus_id[std::string_view(names + name_off[l], (size_t)(name_off[l + 1] - name_off[l]))] = l;   // later duplicates win, as dict(zip()) does
    std::vector<std::string_view> hap(n_haps);
    for (int h = 0; h < n_haps; ++h) hap[h] = std::string_view(haps + hap_off[h], (size_t)(hap_off[h + 1] - hap_off[h]));
    const char *p = text, *end = text + text_len;
    std::string_view last_key;
    int64_t last_l = -1;
    while (p < end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *next = eol ? eol + 1 : end;
        const char *le = eol ? eol : end;
        while (le > p && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;      // str.rstrip() on plain lines
        const char *tab = (const char *)std::memchr(p, '\t', (size_t)(le - p));
        if (!tab) return 1;
        const char *num_end = (const char *)std::memchr(tab + 1, '\t', (size_t)(le - tab - 1));
        if (!num_end) num_end = le;
        std::string_view key(p, (size_t)(tab - p));
        int64_t l;
        int h = 0;
        if (n_haps > 1) {
            const size_t us = key.find('_');
            if (us == std::string_view::npos || key.find('_', us + 1) != std::string_view::npos) return 1;
            const std::string_view lk = key.substr(0, us);
            if (lk == last_key) {                               // the haplotypes of a locus are usually adjacent lines
                l = last_l;
            } else {
                const auto it = locus_id.find(lk);
                if (it == locus_id.end()) return 1;
                l = last_l = it->second;
                last_key = lk;
            }
            const std::string_view hn = key.substr(us + 1);
            for (h = n_haps - 1; h >= 0 && hap[h] != hn; --h) {}
            if (h < 0) return 1;
        } else {
            const auto it = locus_id.find(key);
            if (it == locus_id.end()) return 1;
            l = it->second;
        }
        double len = 0.0;
        const auto r = std::from_chars(tab + 1, num_end, len);
        if (r.ec != std::errc() || r.ptr != num_end || num_end == tab + 1) return 1;
        const double e = len - read_length + 1.0;
        eff_out[(size_t)h * n_loci + l] = e > 1.0 ? e : 1.0;                 // max(..., 1.0); nan -> 1.0 differs: send it back
        if (len != len) return 1;
        p = next;
    }
    return 0;
}

// The call table of `gbrs quantify -G` (gbrs/emase_utils.py:262-268: `#` lines that open the file skipped, then
// `<gene> TAB <call>[ TAB ...]` lines; every character of the call names a haplotype).  Per gene: the OR of the
// haplotype bits of all its lines (the mask accumulates), and the call / index of its last line (what the notes
// keep).  Returns 0 when every line was plain, 1 when some line needs the interpreter's own parsing and error
// reporting (unknown gene or letter, a line without a second field, bytes outside printable ASCII, a call longer
// than call_width): the caller then takes the line-by-line path, which raises what the reference raises.
int gbrs_parse_genotype_table(const char *text, int64_t text_len, const char *gene_names, const int64_t *gene_off,
                              int64_t n_genes, const char *haps, const int64_t *hap_off, int32_t n_haps,
                              uint32_t *gene_bits, char *gene_call, int32_t call_width, int32_t *gene_last_line,
                              int64_t *n_lines) {
    using gbrs::fail;
    if (!text || !gene_names || !gene_off || !haps || !hap_off || !gene_bits || !gene_call || !gene_last_line ||
        n_genes < 1 || n_haps < 1 || n_haps > 32 || text_len < 0 || call_width < 1)
        return fail(GBRS_ERR_INVALID, "bad argument");
    int bit_of[128];
    for (int &b : bit_of) b = -1;
    for (int h = 0; h < n_haps; ++h)
        if (hap_off[h + 1] - hap_off[h] == 1 && (unsigned char)haps[hap_off[h]] < 128)
            bit_of[(unsigned char)haps[hap_off[h]]] = h;                     // a later duplicate name wins, as in dict(zip())
    std::unordered_map<std::string_view, int64_t> gene_id;
    gene_id.reserve((size_t)n_genes * 2);
    for (int64_t g = 0; g < n_genes; ++g)
        gene_id[std::string_view(gene_names + gene_off[g], (size_t)(gene_off[g + 1] - gene_off[g]))] = g;
    const char *p = text, *end = text + text_len;
    while (p < end && *p == '#') {                                            // dropwhile(is_comment)
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        p = eol ? eol + 1 : end;
    }
    int64_t line = 0;
    while (p < end) {
        const char *eol = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *next = eol ? eol + 1 : end;
        const char *le = eol ? eol : end;
        for (const char *q = p; q < le; ++q)
            if (((unsigned char)*q < 0x20 && *q != '\t' && *q != '\r') || (unsigned char)*q >= 0x7f) return 1;
        while (le > p && (le[-1] == '\r' || le[-1] == ' ' || le[-1] == '\t')) --le;      // str.rstrip()
        const char *tab = (const char *)std::memchr(p, '\t', (size_t)(le - p));
        if (!tab) return 1;
        const char *call_end = (const char *)std::memchr(tab + 1, '\t', (size_t)(le - tab - 1));
        if (!call_end) call_end = le;
        const auto it = gene_id.find(std::string_view(p, (size_t)(tab - p)));
        if (it == gene_id.end()) return 1;
        const int64_t g = it->second;
        const int64_t clen = call_end - (tab + 1);
        if (clen > call_width) return 1;
        uint32_t bits = 0;
        for (const char *q = tab + 1; q < call_end; ++q) {
            const int b = bit_of[(unsigned char)*q];
            if (b < 0) return 1;
            bits |= 1u << b;
        }
        gene_bits[g] |= bits;
        std::memset(gene_call + g * call_width, 0, (size_t)call_width);
        std::memcpy(gene_call + g * call_width, tab + 1, (size_t)clen);
        gene_last_line[g] = (int32_t)line;
        ++line;
        p = next;
    }
    if (n_lines) *n_lines = line;
    return 0;
}

}  // extern "C"
