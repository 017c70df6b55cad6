"""acceptance_map.pfm -> acceptance-map.png (heat = G / (R + G + eps)); see drmlt-mitsuba_amd/heatmap.py.

  python tools/acceptance_heatmap.py -t acceptance_map.pfm -c 0.2 0.8 [-o acceptance-map.png]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Delayed-rejection acceptance map -> false-colour heat map")
    ap.add_argument("-t", "--test", required=True, help="acceptance map (.pfm written by drmlt_render, or .npy H x W x 3)")
    ap.add_argument("-eps", "--epsilon", type=float, default=1e-2)
    ap.add_argument("-c", "--clip", nargs=2, type=float, default=[0.0, 1.0])
    ap.add_argument("-o", "--out", default="acceptance-map.png")
    a = ap.parse_args()
    hm = entry.load_package().heatmap
    import numpy as np
    film = np.load(a.test) if a.test.endswith(".npy") else hm.read_pfm(a.test)
    hm.write_png(a.out, hm.heatmap(film, a.clip, a.epsilon))
    print("wrote %s (%d x %d)" % (a.out, film.shape[1], film.shape[0]))
