"""`python bench.py --gpus N` with NO launcher drives the N devices from one process through drmlt_node_* (VERDICT r03 #1).
A one-GPU box runs exactly that code with two ranks on device 0 (test hook: loopback transport instead of RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--res", "128", "--chains", "4096", "--spp", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def _bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DRMLT_NODE_DEVICES", "DRMLT_TEST_HOOKS")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    return p


def test_two_ranks_from_one_process_without_a_launcher(native_lib):
    p = _bench(["--gpus", "2"] + SMALL, {"DRMLT_TEST_HOOKS": "1", "DRMLT_NODE_DEVICES": "0,0"})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "ONE JSON line on stdout: %r" % lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch_mode"] == "node" and out["rccl_nranks"] == 2 and out["exchanges"] == 1
    assert [r["chains"] for r in out["ranks"]] == [[0, 4096], [4096, 8192]]
    assert [r["film_rows"] for r in out["ranks"]] == [[0, 64], [64, 128]]
    sc = out["selfcheck"]
    per_rank = 2 * 128 * 128 * 64
    assert sc["rank_mutations"] == [per_rank, per_rank] and sc["sum_rank_mutations"] == sc["expected_total"] == 2 * per_rank
    assert sc["mutations_ok"] and sc["luminance_ok"] and abs(sc["image_mean_luminance"] - sc["b"]) <= 1e-3 * sc["b"]
    assert sc["timed_image_equals_this_one"] and all(m > 0 for m in sc["rank_film_mass"])
    # whole-job value: both ranks' mutations over the max-over-ranks time
    assert abs(out["value"] - 2 * per_rank / (out["ms_per_step"] * 2e-3)) <= 1e-6 * out["value"]
    assert out["exchange_ms"] is not None and out["roofline"]["launches"] >= 1


def test_more_ranks_than_devices_is_the_only_refusal(native_lib):
    import torch
    n = torch.cuda.device_count() + 1
    p = _bench(["--gpus", str(n)] + SMALL, {})
    assert p.returncode != 0 and "device(s) visible" in p.stderr and not p.stdout.strip()


def test_single_gpu_line_keeps_its_shape(native_lib):
    p = _bench(["--gpus", "1", "--no-quality"] + SMALL, {})
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["launch_mode"] == "single" and out["rccl_nranks"] is None and out["selfcheck"] is None
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out
