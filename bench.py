#!/usr/bin/env python3
"""Benchmark of the DRMLT hot path on MI355X.

Workload = BASELINE.json configs[1]: Cornell box 512x512, `integrator=drmlt technique=path type=orbital`,
65 536 chains per GPU, sampleCount 256 => one step = one full mutation phase of that render
(512*512*256 = 67 108 864 chain-loop iterations, 1 024 per chain), scene and chain state resident in HBM
before the timed region. Metric = mutations/s (accepted + rejected; one mutation = one first-stage proposal,
reference drmlt_proc.cpp:541).

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, chains partitioned by chain id (no data-path collective); each step ends with the
film exchange of the render it represents: one RCCL all-reduce (sum) of the W*H*3 fp32 film plus one scalar
all-reduce of b (reference: DRMLTProcess::processResult / develop, drmlt_proc.cpp:813-867).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


class _DevArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2,
                                         "strides": None}


def host_threads():
    """Threads for the CPU baseline: the cgroup CPU quota / affinity of this box, not the host's core count."""
    if os.environ.get("BENCH_CPU_THREADS"):
        return max(1, int(os.environ["BENCH_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return min(n, 64)


def _clean(o):
    """Strict JSON: NaN / inf become null."""
    if isinstance(o, float):
        return o if np.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    return o


def lum(img):
    return img @ np.array([0.212671, 0.715160, 0.072169])


def cpu_baseline(pkg, cfg_kw, res, target_seconds):
    """Time the CPU restatement (oracle/, -O3 -march=native) on all host cores on a bounded sample of the same
    workload: same scene/config, fewer chains x fewer mutations."""
    ob = entry.load_oracle()
    ob.build(native=True)
    abi = pkg.abi
    cores = host_threads()
    sd = pkg.scenes.cornell_c2(res)
    chains = 64 * cores
    cfg = abi.make_config(work_units=chains, luminance_samples=20000, **cfg_kw)
    orc = ob.Oracle(abi, cfg, sd, precision=64, native=True)
    orc.seed(0x5EED)
    probe = chains * 2048
    t = time.time()
    orc.run(probe, cores)
    rate = probe / max(time.time() - t, 1e-6)
    per_chain = max(64, int(rate * target_seconds / chains))
    total = chains * per_chain
    t = time.time()
    orc.run(total, cores)
    dt = time.time() - t
    orc.close()
    return {"value": total / dt, "unit": "mutations/s", "cores": cores, "kind": "port",
            "sample": "oracle (fp64 CPU restatement) on %d threads: Cornell 512x512 orbital, %d chains x %d "
                      "mutations = %d mutations in %.1f s" % (cores, chains, per_chain, total, dt),
            "per_core": total / dt / cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--spp", type=int, default=256, help="mutations per pixel per step (sampleCount)")
    ap.add_argument("--type", default="orbital")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    # Libraries loaded below write to the process's stdout on their own (RCCL prints a version banner when its first
    # communicator comes up): keep file descriptor 1 pointed at stderr until the one JSON line is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DRMLT path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 rehearses the RCCL path (process group, zero-copy film tensor, exchange) on one GPU
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    abi = pkg.abi
    cfg_kw = dict(type=args.type, max_depth=8, rr_depth=5, direct_samples=-1, sample_count=args.spp)
    sd = pkg.scenes.cornell_c2(args.res)
    cfg = abi.make_config(work_units=args.chains, luminance_samples=10 * args.chains, **cfg_kw)
    ctx = pkg.Context(cfg, sd, device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    npix = args.res * args.res
    film = torch.as_tensor(_DevArray(ctx.film_device_ptr(), (npix * 3,)), device="cuda")
    b_local = ctx.seed(0x5EED, chain_offset=rank * args.chains)   # bootstrap + seed replay: outside the timed region
    b_t = torch.tensor([b_local], dtype=torch.float64, device="cuda")
    film_sum, b_sum = torch.empty_like(film), torch.empty_like(b_t)
    step_mutations = npix * args.spp                                # per GPU

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ctx.run(step_mutations)
        if use_dist:  # the render's film exchange: sum of the per-GPU films, mean of the per-GPU b estimates
            pkg.exchange.exchange_film(film, b_t, dist, out=film_sum, b_out=b_sum)

    for _ in range(args.warmup):
        step()
        ctx.film_clear()
    ctx.kernel_time(reset=True)
    st0 = ctx.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        et = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())

    st1 = ctx.stats()
    launch_ms, launches = ctx.kernel_time()
    muts = st1.mutations - st0.mutations
    accepted = st1.accepted - st0.accepted
    evals = st1.path_evals - st0.path_evals
    rays = st1.rays - st0.rays
    value = world * args.steps * step_mutations / elapsed

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (k_mutate): algorithmic bytes per SURVEY 8(d) / DESIGN.md
        D = st1.max_dim
        p_acc = accepted / muts
        n_splats = 2.0 + (st1.second_base - st0.second_base) / muts
        bytes_per_mut = (4 * D + 32) * (1.0 + p_acc) + 24.0 * n_splats
        muts_per_launch = muts / max(launches, 1)
        achieved = bytes_per_mut * muts_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic = None
        valu_frac = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                # PMC passes were taken on 1.68e7-mutation launches; traffic is proportional to the mutation count
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_mutation") * muts_per_launch
                # wave-level VALU instructions per mutation (SQ_INSTS_VALU) against 1024 SIMDs issuing one per 4 cycles
                valu_frac = pj["instructions_per_mutation"]["valu"] * muts_per_launch / (launch_ms * 1e-3) / (1024 * 2.4e9 / 4)
            except Exception:
                traffic = None
        # SURVEY 8(d) asks for three separately labelled byte figures: (i) measured HBM traffic, (ii) algorithmic bytes
        # (= roofline.achieved), (iii) bytes REQUESTED per lane including what the scalar cache / LDS serves (every ray
        # tests every 64-byte primitive record of this BVH-less scene) -- (iii) is never used as the roofline figure
        n_records = len(sd.shapes)   # rectangles stay one record each (quad test); no triangle meshes in this scene
        requested = (rays / muts) * n_records * 64.0 + bytes_per_mut
        out = {
            "metric": "mutations/sec (accepted+rejected)", "value": value, "unit": "mutations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cornell box %dx%d, integrator=drmlt technique=path type=%s, %d chains/GPU, "
                                   "sampleCount %d (BASELINE.json configs[1])" % (args.res, args.res, args.type,
                                                                                    args.chains, args.spp),
                       "max_depth": 8, "rr_depth": 5, "p_large": 0.3, "filter": "box",
                       "mutations_per_step_per_gpu": step_mutations, "parallelism": "chains partitioned x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_mutate_v3", "avg_launch_ms": launch_ms, "launches": launches,
                         "algorithmic_bytes_per_mutation": bytes_per_mut,
                         "mutations_per_launch": muts_per_launch,
                         "hbm_measured_gbs": (traffic / (launch_ms * 1e-3) / 1e9) if traffic and launch_ms > 0 else None,
                         "requested_gbs_incl_cache_served": requested * muts_per_launch / (launch_ms * 1e-3) / 1e9
                         if launch_ms > 0 else None,
                         "valu_issue_frac_at_2.4GHz": valu_frac},
            "accepted_mutations_per_s": world * accepted / elapsed,
            "path_evals_per_s": world * evals / elapsed, "rays_per_s": world * rays / elapsed,
            "acceptance": {k: round(v, 5) for k, v in st1.ratios().items()},
        }

    # ---- image quality at the accumulated budget (outside the timed region)
    if not args.no_quality:
        b_mean = b_local
        if use_dist:  # the local films hold `steps` renders each: combine them once and develop the sum
            pkg.exchange.exchange_film(film, b_t, dist, out=film_sum, b_out=b_sum)
            film.copy_(film_sum)
            torch.cuda.synchronize()
            b_mean = float(b_sum.item())
        ctx.set_luminance(b_mean)
        img = ctx.develop()
        if rank == 0:
            ref = ctx.render_pt(2048, seed=4242)
            li, lr = lum(img), lum(ref)
            rmse = float(np.mean((li - lr) ** 2 / (lr ** 2 + 1e-2 * lr.mean() ** 2)))
            out["quality"] = {"rel_mse_vs_pt_2048spp": rmse, "mutations_per_pixel": args.spp * args.steps * world,
                              "b": b_mean, "mean_luminance": float(li.mean())}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pkg, cfg_kw, args.res, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(_clean(out)), flush=True)
        os.dup2(2, 1)  # whatever is printed during teardown goes to stderr as well
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
