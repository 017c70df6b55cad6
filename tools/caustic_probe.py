"""Diagnosis (GPU box): the states the ORACLE's mmlt chains hold while they sit on the caustic's focus (config 5 scene, 64 x 64:
pixel row 39, columns 32-33) are evaluated by the device and by the oracle on identical PSS points -- does the device see
the same path, the same strategy, the same luminance there?
  python tools/caustic_probe.py [--rounds 6]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    pkg, ob = g.load_package(), g.load_oracle()
    ob.build(native=True)
    abi = pkg.abi
    sd = pkg.scenes.caustic_c5(res=64)
    cfg = abi.make_config(technique="mmlt", type="orbital", max_depth=6, fix_emitter_path=1, direct_samples=-1, work_units=4096, sample_count=64, luminance_samples=100000)
    ctx = pkg.Context(cfg, sd)
    rows = []
    for r in range(a.rounds):
        orc = ob.Oracle(abi, cfg, sd, precision=64, native=True)
        orc.seed(9000 + r)
        orc.run(4096 * 150, a.threads)
        c, u = orc.chain_state(27)
        sel = (np.floor(c["y"]) == 39) & ((np.floor(c["x"]) == 32) | (np.floor(c["x"]) == 33)) & (c["luminance"] > 0)
        print("round %d: %d of 4096 oracle chains sit on the focus" % (r, sel.sum()), flush=True)
        for d in range(2, 7):
            m = sel & (c["n_dims"] == d)
            if not m.any():
                continue
            us = u[m, :14]; ue = np.zeros((m.sum(), 14), np.float32); ue[:, :12] = u[m, 14:26]; ud = u[m, 26]
            gg, stg = ctx.eval_paths_mmlt(d, us, ue, ud)
            oo, sto = orc.mmlt_eval(d, us, ue, ud)
            for i in range(m.sum()):
                rows.append((d, int(sto[i, 0]), int(sto[i, 1]), int(stg[i, 0]), int(stg[i, 1]), float(oo["luminance"][i]), float(gg["luminance"][i]), float(c["luminance"][m][i]),
                             float(oo["x"][i]), float(oo["y"][i]), float(gg["x"][i]), float(gg["y"][i]), int(oo["n_rays"][i]), int(gg["n_rays"][i])))
        orc.close()
    rows = np.array(rows)
    print("states:", len(rows))
    lo, lg = rows[:, 5], rows[:, 6]
    print("oracle eval == chain's own luminance:", np.mean(np.abs(lo - rows[:, 7]) <= 1e-6 * rows[:, 7]))
    print("device zero where oracle positive:", np.mean((lg == 0) & (lo > 0)))
    rel = np.abs(lg - lo) / lo
    print("rel diff quantiles 50/90/99:", np.quantile(rel, [0.5, 0.9, 0.99]))
    print("sum device / sum oracle:", lg.sum() / lo.sum())
    for d in range(2, 7):
        for s in range(0, 8):
            m = (rows[:, 0] == d) & (rows[:, 1] == s)
            if m.sum():
                print("depth %d s %d t %d: n %4d  sum ratio %.4f  zero on device %.3f  median rel %.2e  same strategy %.3f  rays equal %.3f" %
                      (d, s, int(rows[m][0, 2]), m.sum(), lg[m].sum() / lo[m].sum(), np.mean(lg[m] == 0), np.median(rel[m]), np.mean((rows[m, 3] == rows[m, 1]) & (rows[m, 4] == rows[m, 2])), np.mean(rows[m, 12] == rows[m, 13])))
    bad = np.argsort(-rel)[:12]
    print("worst:"); print(rows[bad])
    np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "caustic_probe.npy"), rows)


if __name__ == "__main__":
    main()
