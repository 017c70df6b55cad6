"""Diagnosis (GPU box): f(u) of device and oracle on MANY identical uniform PSS points (technique=bdpt: splat-list luminance) --
where does a difference in the bootstrap mean come from? The evaluation-parity tests draw 6000 points; a class of paths that is
one sample in 1e5 and carries a percent of the energy needs a million.
  python tools/eval_probe.py --scene door_c3 --n 1000000"""
import argparse
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="door_c3")
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--max-depth", type=int, default=6)
    ap.add_argument("--direct", type=int, default=1)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--precision", type=int, default=64)
    a = ap.parse_args()
    pkg, ob = g.load_package(), g.load_oracle()
    ob.build(native=True)
    abi = pkg.abi
    sd = pkg.scenes.SCENES[a.scene](res=64)
    cfg = abi.make_config(technique="bdpt", type="orbital", max_depth=a.max_depth, rr_depth=5, direct_samples=-1, work_units=1024, sample_count=1,
                          luminance_samples=1000, no_direct_sampling=0 if a.direct else 1)
    ctx = pkg.Context(cfg, sd)
    rng = np.random.default_rng(5)
    chunk = 100000
    tot_g = tot_o = 0.0
    worst = []
    only_o = only_g = 0.0
    n_only_o = n_only_g = 0
    orcs = [ob.Oracle(abi, cfg, sd, precision=a.precision, native=True) for _ in range(a.threads)]
    LW = np.array([0.212671, 0.715160, 0.072169])
    cls = {"main": [0.0, 0.0], "light": [0.0, 0.0]}                     # [device, oracle] luminance by splat class
    img = {"main": np.zeros((2, 16, 16)), "light": np.zeros((2, 16, 16))}   # the same, by 4 x 4 pixel block of the 64 x 64 film
    by_n = np.zeros((2, 16))                                            # list luminance by number of light-image splats
    for c0 in range(0, a.n, chunk):
        us, ue, ud = (rng.random((chunk, 24), dtype=np.float32) for _ in range(3))
        gl = ctx.eval_lists_bdpt(us, ue, ud) if a.direct else ctx.eval_lists_bdpt(us, ue)
        ol = np.zeros_like(gl)
        parts = np.array_split(np.arange(chunk), a.threads)

        def work(k):
            idx = parts[k]
            ol[idx] = orcs[k].bdpt_eval(us[idx], ue[idx], ud[idx]) if a.direct else orcs[k].bdpt_eval(us[idx], ue[idx])
        th = [threading.Thread(target=work, args=(k,)) for k in range(a.threads)]
        [t.start() for t in th]; [t.join() for t in th]
        lg, lo = gl[:, 0].astype(np.float64), ol[:, 0].astype(np.float64)
        tot_g += lg.sum(); tot_o += lo.sum()
        for side, L in enumerate((gl.astype(np.float64), ol.astype(np.float64))):
            ml = (L[:, 4:7] @ LW) * (L[:, 1] > 0)
            cls["main"][side] += ml.sum()
            bx, by = np.clip((L[:, 2] / 4).astype(int), 0, 15), np.clip((L[:, 3] / 4).astype(int), 0, 15)
            np.add.at(img["main"][side], (by, bx), ml)
            more = L[:, 10:].reshape(len(L), -1, 5)
            valid = np.arange(more.shape[1])[None, :] < L[:, 7:8]
            ll = (more[:, :, 2:5] @ LW) * valid
            cls["light"][side] += ll.sum()
            bx, by = np.clip((more[:, :, 0] / 4).astype(int), 0, 15), np.clip((more[:, :, 1] / 4).astype(int), 0, 15)
            np.add.at(img["light"][side], (by[valid], bx[valid]), ll[valid])
            np.add.at(by_n[side], np.clip(L[:, 7].astype(int), 0, 15), L[:, 0])
        m = (lg == 0) & (lo > 0); only_o += lo[m].sum(); n_only_o += int(m.sum())
        m = (lo == 0) & (lg > 0); only_g += lg[m].sum(); n_only_g += int(m.sum())
        d = lg - lo
        for i in np.argsort(-np.abs(d))[:10]:
            worst.append((float(d[i]), float(lg[i]), float(lo[i]), gl[i, 1:10].tolist(), ol[i, 1:10].tolist(), us[i].tolist(), ue[i].tolist(), ud[i].tolist()))
        print("%d points: sum device %.6g oracle %.6g ratio %.5f | oracle-only %d pts %.4g (%.3f%%), device-only %d pts %.4g (%.3f%%)" %
              (c0 + chunk, tot_g, tot_o, tot_g / tot_o, n_only_o, only_o, 100 * only_o / tot_o, n_only_g, only_g, 100 * only_g / tot_o), flush=True)
    for k in ("main", "light"):
        g_, o_ = cls[k]
        d = img[k][0] - img[k][1]
        print("%s splats: device %.6g oracle %.6g ratio %.5f | 4 x 4 blocks: max |device - oracle| / block mean %.4g, rms %.4g" %
              (k, g_, o_, g_ / max(o_, 1e-30), np.abs(d).max() / max(img[k][1].mean(), 1e-30), np.sqrt((d ** 2).mean()) / max(img[k][1].mean(), 1e-30)))
    print("list luminance by number of light-image splats (device / oracle): " +
          " ".join("%d: %.5f" % (i, by_n[0, i] / by_n[1, i]) for i in range(16) if by_n[1, i] > 0))
    worst.sort(key=lambda w: -abs(w[0]))
    print("largest differences (device - oracle, device, oracle, [hasMain px py r g b nMore nDims nRays] x 2):")
    for w in worst[:15]:
        print("  %+.4g  %.5g %.5g  %s | %s" % (w[0], w[1], w[2], np.round(w[3], 3).tolist(), np.round(w[4], 3).tolist()))
    s = sum(w[0] for w in worst[:200])
    print("the 200 largest differences sum to %.4g = %.3f%% of the oracle total" % (s, 100 * s / tot_o))
    import json
    json.dump([dict(d=w[0], lg=w[1], lo=w[2], g=w[3], o=w[4], us=w[5], ue=w[6], ud=w[7]) for w in worst[:50]], open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "eval_probe_worst.json"), "w"))


if __name__ == "__main__":
    main()
