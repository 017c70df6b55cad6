"""SURVEY 8(d) parity protocol (tools/parity_protocol.py) at a reduced size: N independent GPU and oracle renders against
one high-spp path-traced reference. The full-size run (N = 16, 64 x 64, 1024 mutations/pixel, 65 536-spp reference) is kept
in profiles/r02_parity_protocol.json and quoted by bench.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_protocol_items_1_to_3(tmp_path, native_lib):
    out = str(tmp_path / "p.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_protocol.py"), "--res", "32", "--spp", "512", "--n", "8",
                    "--ref-spp", "32768", "--chains", "1024", "--threads", "8", "--out", out], check=True, capture_output=True, timeout=900)
    r = json.load(open(out))
    # (1) unbiased: the error of the mean of n renders falls like 1 / n on both sides (reference noise subtracted)
    s = r["loglog_slope_vs_n_after_subtracting_reference_noise"]
    assert -1.2 < s["gpu"] < -0.8 and -1.2 < s["oracle"] < -0.8, s
    # (2) equal budget: |rMSE_gpu - rMSE_oracle| / rMSE_oracle < 10 %
    assert r["equal_budget_relative_difference"] < 0.10, r["rmse_single_render"]
    # the two means are the same image up to their own noise: far closer to each other than either is to the reference
    assert r["mean_image_gpu_vs_oracle_rmse"] < 0.5 * r["rmse_of_mean_of_n"]["gpu"][-1]
    # (3) only asserted where the oracle meets it too
    b = r["budget_mutations_per_pixel_where_both_meet_1e-3"]
    assert b is None or b <= 8 * 512
