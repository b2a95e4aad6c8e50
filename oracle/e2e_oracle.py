#!/usr/bin/env python3
"""CPU ORACLE (test infrastructure, not product code): `gbrs quantify` + `gbrs reconstruct` end to end on
one core, file in -> reports out, as the CPU baseline of bench.py's end-to-end measurement.

It restates the reference's two workflows with the numpy oracles of this directory and the same
per-line Python file handling the reference uses, so its stage times stand in for the reference's on a
box where the reference itself cannot run (/root/reference does not travel; PyTables is absent):

  quantify     gbrs/emase_utils.py:180-332  load alignment file + group file [-> the `-G` genotype mask: one line of
               the call table at a time into an (H x L) mask, multiply(axis=2) + eliminate_zeros, :240-273]
               -> EMfactory.prepare (length file parsed line by line, EMfactory.py:60-94) -> run(model 4)
               -> 4 TSV reports written value by value with str() (EMfactory.py:289-380)
  reconstruct  gbrs/gbrs_utils.py:382-609   load avecs / gene order / TPM / tprob -> per gene emission
               (np.load()[gene] per gene, :490) -> forward, backward, posterior, Viterbi per chromosome
               -> genoprobs.npz, genotypes.tsv, genotypes.npz

Differences that favour this baseline over the real reference: the alignment file is the `.npz` mirror
(np.load + zlib) instead of PyTables HDF5, and the unused t2t_mat double loop of EMfactory.prepare
(:48-59, Models 1-3 only) is skipped.

Usage:  python oracle/e2e_oracle.py quantify  ALN.npz GROUPS LENGTHS OUTBASE [GENOTYPES.tsv]
        python oracle/e2e_oracle.py reconstruct GENES.tpm TPROB.npz AVECS.npz GPOS.npz FAI OUTBASE
Prints one JSON line with the stage times.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from oracle.em_oracle import EMOracle  # noqa: E402
from oracle import hmm_oracle  # noqa: E402


def _write_table(path, hnames, names, values, notes=None):
    totals = values.sum(axis=0)
    data = np.vstack((values, totals))
    with open(path, "w") as fh:
        fh.write("locus\t" + "\t".join(hnames) + "\ttotal" + ("\tnotes" if notes is not None else "") + "\n")
        for k in range(len(names)):
            fh.write("\t".join([names[k]] + list(map(str, data[:, k].ravel()))))
            if notes is not None:
                fh.write("\t%s" % notes[names[k]])
            fh.write("\n")


def quantify(aln_file, group_file, length_file, outbase, genotype_file=None, tol=1e-4, max_iters=999):
    t = {}
    t0 = time.perf_counter()
    with np.load(aln_file, allow_pickle=False) as z:
        L, H, R = (int(x) for x in z["shape"])
        indptr = [z[f"indptr{h}"] for h in range(H)]
        indices = [z[f"indices{h}"] for h in range(H)]
        count = z["count"] if "count" in z.files else None
        hname = [str(x) for x in z["hname"]]
        lname = [str(x) for x in z["lname"]]
    lid = dict(zip(lname, range(L)))
    gname, groups = [], []
    with open(group_file) as fh:
        for line in fh:
            item = line.rstrip().split("\t")
            gname.append(item[0])
            groups.append([lid[x] for x in item[1:]])
    t["load"] = time.perf_counter() - t0
    hid = dict(zip(hname, range(H)))
    o = EMOracle(R, L, H, indptr, indices, count)
    kind = "multiway"
    gtcall_g = gtcall_t = None
    if genotype_file is not None:
        # gbrs/emase_utils.py:240-273, line by line as there
        t0 = time.perf_counter()
        kind = "diploid"
        gid = dict(zip(gname, range(len(gname))))
        gtmask = np.zeros((H, L))
        gtcall_g = dict.fromkeys(gname)
        gtcall_t = dict.fromkeys(lname)
        with open(genotype_file) as fh:
            started = False
            for line in fh:
                if not started and line.startswith("#"):
                    continue
                started = True
                item = line.rstrip().split("\t")
                g, gt = item[:2]
                gtcall_g[g] = gt
                hid2set = np.array([hid[c] for c in gt])
                tid2set = np.array(groups[gid[g]])
                gtmask[tuple(np.meshgrid(hid2set, tid2set))] = 1.0
                for tt in tid2set:
                    gtcall_t[lname[tt]] = gt
        o.apply_genotype_mask(gtmask)
        t["mask"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    eff = np.zeros((L, H))
    with open(length_file) as fh:
        for line in fh:
            item = line.rstrip().split("\t")
            locus, hap = item[0].split("_")
            eff[lid[locus], hid[hap]] = max(float(item[1]) - 100 + 1.0, 1.0)
    eff = eff.transpose()
    o.prepare(0.0, eff)
    t["em_setup"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = o.run(tol=tol, max_iters=max_iters)
    t["em_run"] = time.perf_counter() - t0
    t["em_iterations"] = n
    t0 = time.perf_counter()
    theta = o.theta * (1000000.0 / o.theta.sum())
    _write_table(f"{outbase}.{kind}.isoforms.tpm", hname, lname, theta, gtcall_t)
    counts = o.expected_read_counts()
    _write_table(f"{outbase}.{kind}.isoforms.expected_read_counts", hname, lname, counts, gtcall_t)
    # scipy hands the reference this product as the transpose of a C-ordered (G x H) array, and the
    # full .sum() below adds in memory order (EMfactory.py:349-354)
    gene = np.asfortranarray(EMOracle.group_sums(theta, groups))
    gene *= 1000000.0 / gene.sum()
    _write_table(f"{outbase}.{kind}.genes.tpm", hname, gname, gene, gtcall_g)
    _write_table(f"{outbase}.{kind}.genes.expected_read_counts", hname, gname,
                 np.asfortranarray(EMOracle.group_sums(counts, groups)), gtcall_g)
    t["reports"] = time.perf_counter() - t0
    t["rows"] = R
    t["entries"] = int(sum(len(i) for i in indices))
    t["entries_in_em"] = int(sum(len(i) for i in o.indices))
    return t


def reconstruct(expr_file, tprob_file, avec_file, gpos_file, fai_file, outbase):
    t = {}
    t0 = time.perf_counter()
    chroms = [line.split()[0] for line in open(fai_file) if line.strip()]
    avecs_npz = np.load(avec_file)
    gpos = np.load(gpos_file)
    order = {c: [str(g) for g, *_ in gpos[c]] for c in gpos.files}
    expr = {}
    with open(expr_file) as fh:
        haps = fh.readline().rstrip().split("\t")[1:-1]
        for line in fh:
            item = line.rstrip().split("\t")
            expr[item[0]] = np.array(list(map(float, item[1:-1])))
    tprob_npz = np.load(tprob_file)
    tprob = {c: tprob_npz[c] for c in chroms if c in tprob_npz.files}
    t["load"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    # the reference indexes the NpzFile per gene (gbrs_utils.py:490), which re-reads the member each time
    avec_ids = set(avecs_npz.files)
    avecs = {g: avecs_npz[g] for c in tprob for g in order[c] if g in avec_ids}
    res = hmm_oracle.reconstruct_arrays(haps, [c for c in chroms if c in tprob], order, tprob, expr, avecs)
    t["hmm"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    names = [a + b for i, a in enumerate(haps) for b in haps[i:]]
    np.savez_compressed(f"{outbase}.genoprobs.npz", **{c: res[c]["gamma"] for c in res})
    calls = {}
    for c in res:
        for g, s in zip(order[c], res[c]["calls"]):
            if s >= 0:
                calls[g] = names[s]
    with open(f"{outbase}.genotypes.tsv", "w") as fh:
        fh.write("#Gene_ID\tDiplotype\n")
        for g in sorted(calls):
            fh.write(f"{g}\t{calls[g]}\n")
    np.savez_compressed(f"{outbase}.genotypes.npz", **{c: [names[s] for s in res[c]["states"]] for c in res})
    t["save"] = time.perf_counter() - t0
    t["genes"] = int(sum(len(order[c]) for c in tprob))
    return t


if __name__ == "__main__":
    t_start = time.perf_counter()
    cmd = sys.argv[1]
    out = quantify(*sys.argv[2:7]) if cmd == "quantify" else reconstruct(*sys.argv[2:8])
    out["total_in_process"] = time.perf_counter() - t_start
    print(json.dumps(out), flush=True)
