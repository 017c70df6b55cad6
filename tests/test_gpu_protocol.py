"""SURVEY 8(d) parity protocol (tools/parity_protocol.py) at a reduced size: N independent device renders and N independent
oracle renders (different seeds on the two sides), compared with each other (two-sample, reference-free) and with an
independent reference. The full-size runs (N = 16, 64 x 64, 1024 mutations/pixel) for configs 2, 3, 5 and bdpt are kept in
profiles/r03_parity_protocol_*.json and quoted by bench.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,extra", [("c2", ["--ref-spp", "32768"]), ("c5", ["--ref-samples-per-pixel", "1024"])])
def test_protocol(tmp_path, native_lib, config, extra):
    out = str(tmp_path / "p.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "parity_protocol.py"), "--config", config, "--res", "32", "--spp", "512",
                    "--n", "8", "--chains", "1024", "--threads", "8", "--out", out] + extra, check=True, capture_output=True, timeout=900)
    r = json.load(open(out))
    # (1) same expectation: the two means differ by what their own noise explains (a bias the size of ONE render's noise
    # would give a ratio ~ 8 here)
    assert r["two_sample"]["ratio"] < 2.0, r["two_sample"]
    # (2) equal budget: per-render noise within 10 % (+ 3 standard errors of the comparison itself at N = 8)
    assert abs(r["equal_budget_relative_difference"]) < 0.10 + 3 * r["equal_budget_standard_error"], r["noise_single_render_about_own_mean"]
    # (3) the error of the mean of n renders falls like 1 / n on both sides: against the independent reference where that
    # is converged enough to say so (the Cornell box), and against the other side's mean everywhere
    x = r["rmse_of_mean_of_n_vs_other_sides_mean"]
    assert -1.35 < x["slope_gpu"] < -0.7 and -1.35 < x["slope_oracle"] < -0.7, x
    if config == "c2":
        s = r["loglog_slope_vs_n_after_subtracting_reference_noise"]
        assert -1.25 < s["gpu"] < -0.75 and -1.25 < s["oracle"] < -0.75, s
    b = r["budget_mutations_per_pixel_where_both_meet_1e-3"]
    assert b is None or b <= 8 * 512
