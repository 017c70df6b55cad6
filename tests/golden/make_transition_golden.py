"""Generates tests/golden/transition_kat.json by running the REFERENCE's own transition kernels
(src/integrators/drmlt/tools/transition.h compiled in place into oracle/_ref/transition_kat by
`make -C oracle ref`) on a fixed uniform stream. Run in the build container only (the reference
checkout does not travel); the JSON it writes is data: inputs + the reference's outputs.
"""
import json
import math
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_binding as ob  # noqa: E402

SEED = 0xD12A17
N = 256

KERNELS = [
    ("gaussian", 0, 0.1 / 64.0, 0.0, np.linspace(-0.01, 0.01, 41)),
    ("gaussian_wide", 0, 1.0 / 64.0, 0.0, np.linspace(-0.08, 0.08, 41)),
    ("kelemen", 1, 1.0 / 1024.0, 1.0 / 64.0, np.linspace(-0.02, 0.02, 81)),
    ("kelemen_orbital", 1, 1.9 / 1024.0, 1.9 / 64.0, np.linspace(-0.035, 0.035, 81)),
    ("identity", 2, 0.0, 0.0, np.linspace(-1, 1, 5)),
    ("wrapped_cauchy", 3, math.exp(-0.25), 0.0, np.linspace(-math.pi, math.pi, 41)),
]


def run_ref(binary, kind, p0, p1, uniforms, du):
    payload = np.concatenate([uniforms.astype(np.float64), du.astype(np.float64)]).tobytes()
    out = subprocess.run([binary, str(kind), repr(p0), repr(p1), str(N), str(len(du))], input=payload,
                         stdout=subprocess.PIPE, check=True).stdout.decode().split()
    vals = [float(v) for v in out]
    samples = vals[:N]
    rest = vals[N:]
    return samples, rest[0::2], rest[1::2]


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    # the uniform stream the oracle's Random would hand out: U(seed, chain 0, TAG_S1, major 0, idx 0..)
    uniforms = ob.uniforms(SEED, 0, 3, 0, 0, 2 * N)
    golden = {"seed": SEED, "n": N, "uniforms": [float(u) for u in uniforms], "kernels": []}
    for name, kind, p0, p1, du in KERNELS:
        entry = {"name": name, "kind": kind, "p0": p0, "p1": p1, "du": [float(x) for x in du]}
        for prec, binary in ((64, "transition_kat"), (32, "transition_kat_f32")):
            s, pdf, logpdf = run_ref(os.path.join(ROOT, "oracle", "_ref", binary), kind, p0, p1, uniforms, du)
            entry["f%d" % prec] = {"samples": s, "pdf": pdf, "logpdf": [None if math.isinf(x) else x for x in logpdf]}
        golden["kernels"].append(entry)
    path = os.path.join(ROOT, "tests", "golden", "transition_kat.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
