"""`gbrs interpolate` / `gbrs export`: oracle vs reference goldens (CPU), HIP vs goldens (GPU)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, load_golden

FILES = sorted(glob.glob(os.path.join(GOLD, "postproc_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=lambda p: p.split("/")[-1][:-4])
def test_postproc_oracle_matches_reference(path):
    from oracle import postproc_oracle
    g = load_golden(path)
    rows = []
    for c in [str(x) for x in g["chroms"]]:
        out = postproc_oracle.interpolate(g[f"xgene_{c}"], g[f"gamma_{c}"], g[f"grid_{c}"])
        np.testing.assert_array_equal(out, g[f"interp_{c}"])
        rows.append(out.T)
    np.testing.assert_allclose(postproc_oracle.dosage(np.vstack(rows), int(g["num_haps"])), g["dosage"], rtol=1e-13)
    with pytest.raises(ValueError):
        c = str(g["chroms"][0])
        postproc_oracle.interpolate(g[f"xgene_{c}"], g[f"gamma_{c}"], np.array([-1.0, 2.0]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: p.split("/")[-1][:-4])
def test_postproc_hip_files(path, tmp_path, monkeypatch):
    from gbrs_amd import cli
    from gbrs_amd.postproc import interpolate_arrays
    g = load_golden(path)
    chroms = [str(x) for x in g["chroms"]]
    H = int(g["num_haps"])
    strains = [chr(65 + h) for h in range(H)]
    (tmp_path / "ref.fa.fai").write_text("".join(f"{c}\t1000000\t0\t60\t61\n" for c in chroms))
    monkeypatch.setenv("GBRS_DATA", str(tmp_path))
    gpos, gamma = {}, {}
    with open(tmp_path / "grid.txt", "w") as fh:
        fh.write("marker\tchr\tbp\tcM\n")
        for c in chroms:
            arr = np.zeros(len(g[f"xgene_{c}"]), dtype=[("f0", "U24"), ("f1", "f8")])
            arr["f0"] = [f"g{i}" for i in range(len(arr))]
            arr["f1"] = g[f"xgene_{c}"]
            gpos[c], gamma[c] = arr, g[f"gamma_{c}"]
            for x in g[f"grid_{c}"]:
                fh.write(f"m\t{c}\t0\t{repr(float(x))}\n")
            out = interpolate_arrays(g[f"xgene_{c}"], g[f"gamma_{c}"], g[f"grid_{c}"])
            np.testing.assert_allclose(out, g[f"interp_{c}"], rtol=1e-12, atol=1e-300)
    np.savez(tmp_path / "gpos.npz", **gpos)
    np.savez(tmp_path / "genoprobs.npz", **gamma)
    assert cli.main(["interpolate", "-i", str(tmp_path / "genoprobs.npz"), "-g", str(tmp_path / "grid.txt"),
                     "-p", str(tmp_path / "gpos.npz"), "-o", str(tmp_path / "interp.npz")]) == 0
    got = np.load(tmp_path / "interp.npz")
    for c in chroms:
        np.testing.assert_allclose(got[c], g[f"interp_{c}"], rtol=1e-12, atol=1e-300)
    assert cli.main(["export", "-i", str(tmp_path / "interp.npz"), "-s", ",".join(strains[:H // 2]),
                     "-s", ",".join(strains[H // 2:]), "-g", str(tmp_path / "grid.txt"),
                     "-o", str(tmp_path / "export.tsv")]) == 0
    text = open(tmp_path / "export.tsv").read()
    ref = str(g["export_text"])
    assert text.split("\n")[0] == ref.split("\n")[0]
    a = np.loadtxt(tmp_path / "export.tsv", skiprows=1, delimiter="\t")
    b = np.loadtxt(ref.split("\n")[1:-1], delimiter="\t")
    np.testing.assert_allclose(a, b, atol=1.01e-6)          # %.6f text: at most one unit in the last place
    with pytest.raises(ValueError, match="interpolation range"):
        interpolate_arrays(g[f"xgene_{chroms[0]}"], g[f"gamma_{chroms[0]}"], np.array([-1.0, 2.0]))
