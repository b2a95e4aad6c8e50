// Driver for the sanitizer build of gbrs_amd/csrc/hostio.hip (host-side file I/O helpers of libgbrs_hip):
// exercises the number formatter, the report writer, the length-table parser, the chunk decoder and the zip /
// npz helpers,
// including malformed inputs, under AddressSanitizer + UndefinedBehaviorSanitizer.  Exit code 0 = clean.
#include <cinttypes>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "../../include/gbrs_hip.h"

namespace gbrs {
static char g_err[512];
int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return status;
}
}  // namespace gbrs

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); return 1; } \
    } while (0)

static std::string slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    // ---- gbrs_format_double: round trip and a few fixed spellings
    char buf[40];
    std::mt19937_64 rng(7);
    for (int i = 0; i < 200000; ++i) {
        uint64_t bits = rng();
        double v;
        std::memcpy(&v, &bits, 8);
        const int n = gbrs_format_double(v, buf);
        CHECK(n > 0 && n < 32 && (int)std::strlen(buf) == n);
        if (std::isfinite(v)) CHECK(std::strtod(buf, nullptr) == v);
    }
    const struct { double v; const char *s; } fixed[] = {{0.0, "0.0"}, {-0.0, "-0.0"}, {1.0, "1.0"}, {1e16, "1e+16"},
        {1e15, "1000000000000000.0"}, {1e-5, "1e-05"}, {0.0001, "0.0001"}, {123.456, "123.456"}, {5e-324, "5e-324"},
        {1.7976931348623157e308, "1.7976931348623157e+308"}, {std::numeric_limits<double>::infinity(), "inf"}};
    for (const auto &f : fixed) {
        gbrs_format_double(f.v, buf);
        CHECK(std::strcmp(buf, f.s) == 0);
    }
    gbrs_format_double(std::nan(""), buf);
    CHECK(std::strcmp(buf, "nan") == 0);
    CHECK(gbrs_format_double(1.0, nullptr) < 0);

    // ---- gbrs_write_locus_table: both memory orders, notes, an explicit order, bad arguments
    {
        const int H = 3;
        const int64_t n = 5000;
        std::vector<double> hl((size_t)H * n), lh((size_t)H * n), tot(n);
        std::string names, notes;
        std::vector<int64_t> noff(n + 1, 0), toff(n + 1, 0), order(n);
        for (int64_t r = 0; r < n; ++r) {
            tot[r] = 0;
            for (int h = 0; h < H; ++h) {
                const double x = (double)(rng() % 100000) / 7.0;
                hl[(size_t)h * n + r] = x;
                lh[(size_t)r * H + h] = x;
                tot[r] += x;
            }
            names += "T" + std::to_string(r);
            noff[r + 1] = (int64_t)names.size();
            notes += (r % 3 ? "AB" : "None");
            toff[r + 1] = (int64_t)notes.size();
            order[r] = n - 1 - r;
        }
        const std::string p1 = dir + "/t1.tsv", p2 = dir + "/t2.tsv", p3 = dir + "/t3.tsv";
        CHECK(gbrs_write_locus_table(p1.c_str(), "locus\tA\tB\tC\ttotal\n", hl.data(), n, H, 1, n, tot.data(),
                                     names.data(), noff.data(), nullptr, nullptr, nullptr) == 0);
        CHECK(gbrs_write_locus_table(p2.c_str(), "locus\tA\tB\tC\ttotal\n", lh.data(), n, H, H, 1, tot.data(),
                                     names.data(), noff.data(), nullptr, nullptr, nullptr) == 0);
        CHECK(slurp(p1) == slurp(p2) && slurp(p1).size() > 100000);
        CHECK(gbrs_write_locus_table(p3.c_str(), "h\n", hl.data(), n, H, 1, n, tot.data(), names.data(), noff.data(),
                                     notes.data(), toff.data(), order.data()) == 0);
        const std::string t3 = slurp(p3);
        CHECK(t3.rfind("h\nT4999\t", 0) == 0 && t3.find("\tNone\n") != std::string::npos);
        order[7] = n;                                                // out of range
        CHECK(gbrs_write_locus_table(p3.c_str(), "h\n", hl.data(), n, H, 1, n, tot.data(), names.data(), noff.data(),
                                     nullptr, nullptr, order.data()) < 0);
        CHECK(gbrs_write_locus_table((dir + "/no/such/dir/x").c_str(), "h\n", hl.data(), n, H, 1, n, tot.data(),
                                     names.data(), noff.data(), nullptr, nullptr, nullptr) < 0);
        CHECK(gbrs_write_locus_table(p3.c_str(), "h\n", hl.data(), 0, H, 1, 0, tot.data(), names.data(), noff.data(),
                                     nullptr, nullptr, nullptr) == 0);           // empty table: header only
        CHECK(slurp(p3) == "h\n");
    }

    // ---- gbrs_parse_length_table: plain file, fallback triggers, no trailing newline
    {
        const char *names = "T0T1T22";
        const int64_t noff[] = {0, 2, 4, 7};
        const char *haps = "AB";
        const int64_t hoff[] = {0, 1, 2};
        double eff[6];
        for (double &x : eff) x = -1.0;
        const std::string ok = "T0_A\t150\nT0_B\t151\r\nT1_A\t50\nT1_B\t1e3\nT22_A\t100.5\nT22_B\t99";
        CHECK(gbrs_parse_length_table(ok.data(), (int64_t)ok.size(), names, noff, 3, haps, hoff, 2, 100.0, eff) == 0);
        CHECK(eff[0] == 51.0 && eff[3] == 52.0 && eff[1] == 1.0 && eff[4] == 901.0 && eff[2] == 1.5 && eff[5] == 1.0);
        const char *bad[] = {"T9_A\t1\n", "T0_C\t1\n", "T0_A_B\t1\n", "T0_A\n", "T0_A\t 12\n", "T0_A\tnan\n", "T0\t5\n", "\n"};
        for (const char *b : bad)
            CHECK(gbrs_parse_length_table(b, (int64_t)std::strlen(b), names, noff, 3, haps, hoff, 2, 100.0, eff) == 1);
        const char *one = "T1\t300\nT22\t20\n";                      // a single haplotype: plain locus keys
        double e1[3] = {0, 0, 0};
        CHECK(gbrs_parse_length_table(one, (int64_t)std::strlen(one), names, noff, 3, haps, hoff, 1, 100.0, e1) == 0);
        CHECK(e1[1] == 201.0 && e1[2] == 1.0);
        CHECK(gbrs_parse_length_table("", 0, names, noff, 3, haps, hoff, 2, 100.0, eff) == 0);
        CHECK(gbrs_parse_length_table(nullptr, 0, names, noff, 3, haps, hoff, 2, 100.0, eff) < 0);
    }

    // ---- gbrs_decode_chunks: stored (unfiltered) chunks with and without byte shuffle, partial last chunk,
    //      a corrupt "deflate" chunk and a short file
    {
        const uint64_t chunk = 1000, n = 3500;
        std::vector<uint32_t> data(n);
        for (auto &x : data) x = (uint32_t)rng();
        const std::string path = dir + "/chunks.bin";
        std::vector<uint64_t> addr, stored, start;
        std::vector<uint32_t> fmask;
        {
            std::ofstream f(path, std::ios::binary);
            uint64_t pos = 0;
            for (uint64_t s0 = 0; s0 < n; s0 += chunk) {
                std::vector<unsigned char> plane(chunk * 4, 0);
                const uint64_t cnt = std::min(chunk, n - s0);
                for (uint64_t i = 0; i < cnt; ++i)
                    for (int b = 0; b < 4; ++b) plane[(size_t)b * chunk + i] = (unsigned char)(data[s0 + i] >> (8 * b));
                f.write((const char *)plane.data(), (std::streamsize)plane.size());
                addr.push_back(pos); stored.push_back(plane.size()); start.push_back(s0);
                fmask.push_back(2u);                                 // bit 1 = the deflate filter was skipped for this chunk
                pos += plane.size();
            }
        }
        std::vector<uint32_t> out(n, 0);
        CHECK(gbrs_decode_chunks(path.c_str(), (int64_t)addr.size(), addr.data(), stored.data(), start.data(), fmask.data(),
                                 chunk, 4, n, /*shuffle*/ 0, /*deflate*/ 1, out.data(), 3) == 0);
        CHECK(out == data);
        fmask[1] = 0u;                                               // claim chunk 1 is deflated: it is not a zlib stream
        CHECK(gbrs_decode_chunks(path.c_str(), (int64_t)addr.size(), addr.data(), stored.data(), start.data(), fmask.data(),
                                 chunk, 4, n, 0, 1, out.data(), 2) < 0);
        fmask[1] = 2u;
        stored[3] += 4096;                                           // runs past the end of the file
        CHECK(gbrs_decode_chunks(path.c_str(), (int64_t)addr.size(), addr.data(), stored.data(), start.data(), fmask.data(),
                                 chunk, 4, n, 0, 1, out.data(), 2) < 0);
        stored[3] -= 4096;
        start[2] = n + 5;                                            // chunk outside the dataset
        CHECK(gbrs_decode_chunks(path.c_str(), (int64_t)addr.size(), addr.data(), stored.data(), start.data(), fmask.data(),
                                 chunk, 4, n, 0, 1, out.data(), 2) < 0);
        CHECK(gbrs_decode_chunks((dir + "/missing.bin").c_str(), 0, nullptr, nullptr, nullptr, nullptr, chunk, 4, n, 0, 1,
                                 out.data(), 1) < 0);
    }
    // ---- gbrs_parse_number_table: plain tables, every malformed shape is "not plain" (1), never a read past the text
    {
        const std::string t = "g1\t1.5\t0.0\t2e-3\ng2\t-4\t5\t6.25\n";
        double v[6];
        CHECK(gbrs_parse_number_table(t.data(), (int64_t)t.size(), 2, 3, v) == 0);
        CHECK(v[0] == 1.5 && v[2] == 2e-3 && v[3] == -4.0 && v[5] == 6.25);
        const std::string no_nl = t.substr(0, t.size() - 1);
        CHECK(gbrs_parse_number_table(no_nl.data(), (int64_t)no_nl.size(), 2, 3, v) == 0);
        CHECK(gbrs_parse_number_table(t.data(), (int64_t)t.size(), 2, 2, v) == 1);       // too many columns
        CHECK(gbrs_parse_number_table(t.data(), (int64_t)t.size(), 3, 3, v) == 1);       // a row short
        CHECK(gbrs_parse_number_table(t.data(), (int64_t)t.size(), 1, 3, v) == 1);       // a row too many
        const std::string bad = "g1\t1.5\tabc\t2\n";
        CHECK(gbrs_parse_number_table(bad.data(), (int64_t)bad.size(), 1, 3, v) == 1);
        for (size_t cut = 0; cut < t.size(); ++cut) (void)gbrs_parse_number_table(t.data(), (int64_t)cut, 2, 3, v);
        CHECK(gbrs_parse_number_table(nullptr, 0, 1, 1, v) < 0);
    }
    // ---- gbrs_zip_directory / gbrs_npz_stack: a hand-made archive of stored .npy members, then truncations,
    //      random corruption of every region and absurd offsets - nothing may read outside the image
    {
        auto put16 = [](std::vector<unsigned char> &v, uint32_t x) { v.push_back(x & 255); v.push_back((x >> 8) & 255); };
        auto put32 = [&](std::vector<unsigned char> &v, uint32_t x) { put16(v, x & 0xFFFF); put16(v, x >> 16); };
        const int n = 40;
        const std::string npy_header = std::string("\x93NUMPY\x01\x00\x76\x00", 10) + std::string(118, ' ');   // 128 bytes, contents immaterial here
        const uint64_t item = 64;
        std::vector<unsigned char> zip, cd;
        std::vector<uint64_t> offs;
        for (int k = 0; k < n; ++k) {
            char name[32];
            std::snprintf(name, sizeof(name), "gene%04d.npy", k);
            const uint32_t nlen = (uint32_t)std::strlen(name), size = (uint32_t)(npy_header.size() + item);
            offs.push_back(zip.size());
            put32(zip, 0x04034b50u); put16(zip, 20); put16(zip, 0); put16(zip, 0); put16(zip, 0); put16(zip, 0x21);
            put32(zip, 0); put32(zip, size); put32(zip, size); put16(zip, nlen); put16(zip, 0);
            zip.insert(zip.end(), name, name + nlen);
            zip.insert(zip.end(), npy_header.begin(), npy_header.end());
            for (uint64_t b = 0; b < item; ++b) zip.push_back((unsigned char)(k + b));
            put32(cd, 0x02014b50u); put16(cd, 20); put16(cd, 20); put16(cd, 0); put16(cd, 0); put16(cd, 0); put16(cd, 0x21);
            put32(cd, 0); put32(cd, size); put32(cd, size); put16(cd, nlen); put16(cd, 0); put16(cd, 0); put16(cd, 0); put16(cd, 0);
            put32(cd, 0); put32(cd, (uint32_t)offs.back());
            cd.insert(cd.end(), name, name + nlen);
        }
        const uint32_t cd_off = (uint32_t)zip.size(), cd_size = (uint32_t)cd.size();
        zip.insert(zip.end(), cd.begin(), cd.end());
        put32(zip, 0x06054b50u); put16(zip, 0); put16(zip, 0); put16(zip, n); put16(zip, n); put32(zip, cd_size); put32(zip, cd_off); put16(zip, 0);

        std::vector<uint16_t> method(n);
        std::vector<uint64_t> cs(n), us(n), ho(n);
        std::vector<char> names(4096);
        uint64_t count = 0, nbytes = 0;
        CHECK(gbrs_zip_directory(zip.data(), zip.size(), 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &count, &nbytes) == 0);
        CHECK(count == (uint64_t)n && nbytes == (uint64_t)n * 13);
        CHECK(gbrs_zip_directory(zip.data(), zip.size(), n, method.data(), cs.data(), us.data(), ho.data(), nullptr, names.data(), names.size(),
                                 &count, &nbytes) == 0);
        CHECK(std::strncmp(names.data(), "gene0000.npy\ngene0001.npy\n", 26) == 0 && ho[7] == offs[7] && us[7] == 192 && method[7] == 0);
        std::vector<unsigned char> out(n * item, 0), fb(n, 9);
        CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                             (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 3) == 0);
        for (int k = 0; k < n; ++k) CHECK(fb[k] == 0 && out[k * item + 5] == (unsigned char)(k + 5));
        {
            std::vector<std::vector<unsigned char>> imgs(n, std::vector<unsigned char>(192));
            std::vector<uint8_t *> ptrs(n);
            for (int k = 0; k < n; ++k) ptrs[k] = imgs[k].data();
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 3) == 0);
            for (int k = 0; k < n; ++k) CHECK(std::memcmp(imgs[k].data(), npy_header.data(), 128) == 0 && imgs[k][128 + 9] == (unsigned char)(k + 9));
            method[2] = 8;                                            // not a deflate stream
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 2) < 0);
            method[2] = 12;
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 2) < 0);
            method[2] = 0;
            ho[1] = zip.size() - 3;
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 2) < 0);
            ho[1] = offs[1];
        }
        {
            // CRC-32 of the members (what numpy.load checks on every access): a right table passes, a wrong entry sends the
            // member to the caller (stack) / fails the call (read_members)
            auto crc_of = [](const unsigned char *p, size_t nb) {
                uint32_t c = 0xFFFFFFFFu;
                for (size_t i = 0; i < nb; ++i) {
                    c ^= p[i];
                    for (int b = 0; b < 8; ++b) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                }
                return c ^ 0xFFFFFFFFu;
            };
            std::vector<uint32_t> crcs(n);
            std::vector<std::vector<unsigned char>> imgs(n, std::vector<unsigned char>(192));
            std::vector<uint8_t *> ptrs(n);
            for (int k = 0; k < n; ++k) {
                ptrs[k] = imgs[k].data();
                crcs[k] = crc_of(zip.data() + offs[k] + 30 + 12, 192);
            }
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), crcs.data(), ptrs.data(), 3) == 0);
            CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), crcs.data(),
                                 (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 3) == 0);
            for (int k = 0; k < n; ++k) CHECK(fb[k] == 0);
            crcs[11] ^= 0x00100000u;
            CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), crcs.data(), ptrs.data(), 3) < 0);
            CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), crcs.data(),
                                 (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 3) == 0);
            for (int k = 0; k < n; ++k) CHECK(fb[k] == (k == 11 ? 1 : 0));
        }
        std::string other = npy_header;
        other[20] = 'x';                                              // another header: every member goes to the caller
        CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, (const uint8_t *)other.data(),
                             other.size(), item, out.data(), fb.data(), 1) == 0);
        for (int k = 0; k < n; ++k) CHECK(fb[k] == 1);
        method[3] = 8;                                                // claims deflate: the bytes are no deflate stream
        CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                             (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2) == 0);
        CHECK(fb[3] == 1 && fb[4] == 0);
        method[3] = 0;
        ho[5] = zip.size() + 77;                                      // offsets outside the image are errors, not reads
        CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                             (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2) < 0);
        ho[5] = offs[5];
        cs[6] = zip.size();
        CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                             (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2) < 0);
        // every truncation of the tail, and random byte damage anywhere: any status, no out-of-bounds access
        for (size_t cut = 1; cut < 200 && cut < zip.size(); ++cut) {
            std::vector<unsigned char> t(zip.begin(), zip.end() - (long)cut);
            (void)gbrs_zip_directory(t.data(), t.size(), n, method.data(), cs.data(), us.data(), ho.data(), nullptr, names.data(), names.size(),
                                     &count, &nbytes);
        }
        for (int trial = 0; trial < 3000; ++trial) {
            std::vector<unsigned char> t = zip;
            for (int hits = 0; hits < 1 + trial % 4; ++hits) t[rng() % t.size()] = (unsigned char)rng();
            std::vector<uint16_t> m2(n);
            std::vector<uint64_t> c2(n), u2(n), h2(n);
            std::vector<uint32_t> crcs(n);
            if (gbrs_zip_directory(t.data(), t.size(), n, m2.data(), c2.data(), u2.data(), h2.data(), crcs.data(), names.data(), names.size(),
                                   &count, &nbytes) == 0 && count == (uint64_t)n)
                (void)gbrs_npz_stack(t.data(), t.size(), n, h2.data(), m2.data(), c2.data(), u2.data(), (trial & 1) ? crcs.data() : nullptr,
                                     (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2);
        }
        CHECK(gbrs_zip_directory(zip.data(), 10, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &count, &nbytes) < 0);
        // offsets and sizes near 2^64 (a crafted zip64 record): the sums header + size wrap around, the checks must not
        {
            std::vector<std::vector<unsigned char>> imgs(n, std::vector<unsigned char>(192));
            std::vector<uint8_t *> ptrs(n);
            for (int k = 0; k < n; ++k) ptrs[k] = imgs[k].data();
            for (uint64_t huge : {(uint64_t)~0ull, (uint64_t)(~0ull - 29), (uint64_t)(~0ull - 100), (uint64_t)1 << 63}) {
                ho[4] = huge;
                CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 2) < 0);
                CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                                     (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2) < 0);
                ho[4] = offs[4];
                cs[4] = huge;
                CHECK(gbrs_zip_read_members(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr, ptrs.data(), 2) < 0);
                CHECK(gbrs_npz_stack(zip.data(), zip.size(), n, ho.data(), method.data(), cs.data(), us.data(), nullptr,
                                     (const uint8_t *)npy_header.data(), npy_header.size(), item, out.data(), fb.data(), 2) < 0);
                cs[4] = us[4];
            }
            // a zip64 end record whose offset wraps: end-of-central-directory with saturated fields + a locator
            std::vector<unsigned char> z64(zip.begin(), zip.begin() + cd_off + cd_size);
            for (uint64_t where : {(uint64_t)(~0ull - 10), (uint64_t)(~0ull - 55), (uint64_t)zip.size() * 2}) {
                std::vector<unsigned char> t = z64;
                put32(t, 0x07064b50u); put32(t, 0);
                for (int b = 0; b < 8; ++b) t.push_back((unsigned char)(where >> (8 * b)));
                put32(t, 1);
                put32(t, 0x06054b50u); put16(t, 0); put16(t, 0); put16(t, 0xFFFF); put16(t, 0xFFFF); put32(t, 0xFFFFFFFFu);
                put32(t, 0xFFFFFFFFu); put16(t, 0);
                CHECK(gbrs_zip_directory(t.data(), t.size(), n, method.data(), cs.data(), us.data(), ho.data(), nullptr, names.data(),
                                         names.size(), &count, &nbytes) < 0);
            }
        }
    }
    // ---- gbrs_decode_chunks: chunks of an empty dataset are refused (the element count used to underflow)
    {
        const uint64_t addr[1] = {0}, bytes[1] = {16}, start[1] = {0};
        const uint32_t fmask[1] = {0};
        unsigned char sink[64];
        CHECK(gbrs_decode_chunks("/dev/null", 1, addr, bytes, start, fmask, 4, 4, 0, -1, -1, sink, 1) < 0);
    }
    // ---- gbrs_parse_genotype_table: plain lines, repeated genes, lines the caller has to take, truncations
    {
        const std::string genes = "G1G22G3", haps = "ABCD";
        const int64_t goff[4] = {0, 2, 5, 7}, hoff[5] = {0, 1, 2, 3, 4};
        uint32_t bits[3];
        char call[3 * 4];
        int32_t last[3];
        int64_t nl = -1;
        auto reset = [&]() { std::memset(bits, 0, sizeof(bits)); std::memset(call, 0, sizeof(call)); last[0] = last[1] = last[2] = -1; };
        const std::string t = "#Gene_ID\tDiplotype\n#more\nG22\tAB\nG1\tCC\textra\nG22\tDA  \n";
        reset();
        CHECK(gbrs_parse_genotype_table(t.data(), (int64_t)t.size(), genes.data(), goff, 3, haps.data(), hoff, 4, bits, call, 4,
                                        last, &nl) == 0);
        CHECK(nl == 3 && bits[0] == 4u && bits[1] == (1u | 2u | 8u) && bits[2] == 0 && last[0] == 1 && last[1] == 2 && last[2] == -1);
        CHECK(std::strncmp(call + 4, "DA", 4) == 0 && std::strncmp(call, "CC", 4) == 0);
        for (const char *bad : {"G9\tAB\n", "G1\tAE\n", "G1\n", "G1\tABCDA\n", "\n", "G1\tA\xc3\xa9\n", "G1\tAB\nG1 \tAB\n"}) {
            reset();
            CHECK(gbrs_parse_genotype_table(bad, (int64_t)std::strlen(bad), genes.data(), goff, 3, haps.data(), hoff, 4, bits, call,
                                            4, last, &nl) == 1);
        }
        for (size_t cut = 0; cut <= t.size(); ++cut) {
            reset();
            (void)gbrs_parse_genotype_table(t.data(), (int64_t)cut, genes.data(), goff, 3, haps.data(), hoff, 4, bits, call, 4, last, &nl);
        }
        CHECK(gbrs_parse_genotype_table(nullptr, 0, genes.data(), goff, 3, haps.data(), hoff, 4, bits, call, 4, last, &nl) < 0);
    }
    std::printf("hostio sanitizer driver: ok (inflate backend %d)\n", gbrs_inflate_backend());
    return 0;
}
