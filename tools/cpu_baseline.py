"""CPU baseline matrix of BASELINE.md section 3: the oracle (CPU restatement of the reference's chain loop, -O3 -march=native,
one chain per work unit on a std::thread pool = the reference's work-unit model) in fp64 and fp32, on all host cores and on
one core, for configs C1 / C2 / C3 / C5, with the CPU model string. Run it on the MI355X box's host:

  python tools/cpu_baseline.py [--seconds 6] [--out profiles/r02_cpu_baseline.json] [--md profiles/r02_cpu_baseline.md]

MITSUBA_DIR hook: if $MITSUBA_DIR holds a build of the reference fork
($MITSUBA_DIR/mitsuba or $MITSUBA_DIR/dist/mitsuba), the same scenes are written as Mitsuba XML (+ OBJ meshes) and rendered
by the real binary; mutations/s = W H sampleCount / "Render time" (src/librender/renderjob.cpp:106) and the printed
acceptance statistics are recorded, and the EXR it writes is kept next to the XML: the first reference-held fixtures, the only
route from parity "partial" to "green". Without it the rows say "reference binary unavailable".
"""
import argparse, json, os, re, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

CONFIGS = {
    "C1": dict(scene="cornell_c1", res=256, cfg=dict(algo="pssmlt", technique="path", type="orbital", max_depth=8, rr_depth=5), spp=64,
               mitsuba=dict(integrator="pssmlt", technique="path")),
    "C2": dict(scene="cornell_c2", res=512, cfg=dict(technique="path", type="orbital", max_depth=8, rr_depth=5), spp=256,
               mitsuba=dict(integrator="drmlt", technique="path", type="orbital")),
    "C3": dict(scene="door_c3", res=512, cfg=dict(technique="path", type="green", max_depth=8, rr_depth=5), spp=64,
               mitsuba=dict(integrator="drmlt", technique="path", type="green")),
    "C5": dict(scene="caustic_c5", res=512, cfg=dict(technique="mmlt", type="orbital", max_depth=6, fix_emitter_path=1, acceptance_map=1), spp=64,
               mitsuba=dict(integrator="drmlt", technique="mmlt", type="orbital", fixEmitterPath="true", acceptanceMap="true", maxDepth=6)),
}


def cpu_model():
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
        m = re.search(r"Model name:\s*(.+)", out)
        if m:
            return m.group(1).strip()
    except Exception:
        pass
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def time_oracle(pkg, ob, name, precision, threads, seconds):
    conf = CONFIGS[name]
    abi = pkg.abi
    kw = dict(conf["cfg"])
    if kw.get("algo") == "pssmlt":
        kw["algo"] = abi.ALGO_PSSMLT
    sd = pkg.scenes.SCENES[conf["scene"]](res=conf["res"])
    chains = 64 * threads
    cfg = abi.make_config(work_units=chains, luminance_samples=20000, direct_samples=-1, sample_count=conf["spp"], **kw)
    orc = ob.Oracle(abi, cfg, sd, precision=precision, native=True)
    orc.seed(0x5EED)
    probe = chains * 128
    t = time.time(); orc.run(probe, threads); rate = probe / max(time.time() - t, 1e-6)
    per_chain = max(64, int(rate * seconds / chains))
    total = chains * per_chain
    st0 = orc.stats()
    t = time.time(); orc.run(total, threads); dt = time.time() - t
    st = orc.stats()
    orc.close()
    return {"config": name, "precision": "fp%d" % precision, "threads": threads, "mutations": total, "seconds": dt, "mutations_per_s": total / dt,
            "mutations_per_s_per_core": total / dt / threads, "path_evals_per_s": (st.path_evals - st0.path_evals) / dt,
            "acceptance": {k: (round(v, 4) if v == v else None) for k, v in st.ratios().items()}}


# ------------------------------------------------------------------ Mitsuba XML export (MITSUBA_DIR hook)
def scene_to_xml(pkg, sd, conf, out_dir, name):
    abi = pkg.abi
    os.makedirs(out_dir, exist_ok=True)
    x = ['<?xml version="1.0" encoding="utf-8"?>', '<scene version="0.6.0">', '  <integrator type="$integrator">']
    for k, v in dict(maxDepth=conf["cfg"].get("max_depth", 8), rrDepth=conf["cfg"].get("rr_depth", 5), directSamples=-1).items():
        x.append('    <integer name="%s" value="%d"/>' % (k, v))
    x += ['    <string name="technique" value="$technique"/>', '    <string name="type" value="$type"/>',
          '    <boolean name="fixEmitterPath" value="$fixEmitterPath"/>', '    <boolean name="acceptanceMap" value="$acceptanceMap"/>', '  </integrator>']
    for i, b in enumerate(sd.bsdfs):
        if b.type == abi.BSDF_DIFFUSE:
            x.append('  <bsdf type="diffuse" id="b%d"><spectrum name="reflectance" value="%.7g, %.7g, %.7g"/></bsdf>' % (i, *b.rgb))
        elif b.type == abi.BSDF_DIELECTRIC:
            x.append('  <bsdf type="dielectric" id="b%d"><float name="intIOR" value="%.7g"/><float name="extIOR" value="%.7g"/></bsdf>' % (i, b.p[0], b.p[1]))
        else:
            x.append('  <bsdf type="roughconductor" id="b%d"><string name="distribution" value="%s"/><float name="alpha" value="%.7g"/>'
                     '<float name="extEta" value="1"/><spectrum name="eta" value="%.7g, %.7g, %.7g"/><spectrum name="k" value="%.7g, %.7g, %.7g"/>'
                     '<spectrum name="specularReflectance" value="%.7g, %.7g, %.7g"/></bsdf>' % (i, "ggx" if b.p[7] else "beckmann", b.p[0], *list(b.p)[1:7], *b.rgb))
    n_obj = 0
    for s in sd.shapes:
        em = ""
        if s.emitter >= 0:
            e = sd.emitters[s.emitter]
            em = '<emitter type="area"><spectrum name="radiance" value="%.7g, %.7g, %.7g"/><float name="samplingWeight" value="%.7g"/></emitter>' % (*e.radiance, e.sampling_weight)
        if s.type == abi.SHAPE_RECTANGLE:
            m = list(s.data) + [0, 0, 0, 1]
            x.append('  <shape type="rectangle"><transform name="toWorld"><matrix value="%s"/></transform><ref id="b%d"/>%s</shape>' % (" ".join("%.9g" % v for v in m), s.bsdf, em))
        elif s.type == abi.SHAPE_SPHERE:
            x.append('  <shape type="sphere"><point name="center" x="%.9g" y="%.9g" z="%.9g"/><float name="radius" value="%.9g"/><ref id="b%d"/>%s</shape>' % (*list(s.data)[:4], s.bsdf, em))
        else:
            fn = "%s_tri%d.obj" % (name, n_obj); n_obj += 1
            with open(os.path.join(out_dir, fn), "w") as f:
                for v in range(3):
                    f.write("v %.9g %.9g %.9g\n" % tuple(list(s.data)[3 * v:3 * v + 3]))
                f.write("f 1 2 3\n")
            x.append('  <shape type="obj"><string name="filename" value="%s"/><boolean name="faceNormals" value="true"/><ref id="b%d"/>%s</shape>' % (fn, s.bsdf, em))
    c = sd.camera
    filt = ('<rfilter type="box"><float name="radius" value="%.7g"/></rfilter>' % c.filter_param) if c.filter == abi.FILTER_BOX else \
           ('<rfilter type="gaussian"><float name="stddev" value="%.7g"/></rfilter>' % c.filter_param)
    x += ['  <sensor type="perspective">', '    <transform name="toWorld"><matrix value="%s"/></transform>' % " ".join("%.9g" % v for v in c.to_world),
          '    <float name="fov" value="%.9g"/><string name="fovAxis" value="x"/><float name="nearClip" value="%.7g"/><float name="farClip" value="%.7g"/>' % (c.fov_x_deg, c.near_clip, c.far_clip),
          '    <sampler type="independent"><integer name="sampleCount" value="%d"/></sampler>' % conf["spp"],
          '    <film type="hdrfilm"><integer name="width" value="%d"/><integer name="height" value="%d"/><boolean name="banner" value="false"/>%s</film>' % (c.width, c.height, filt),
          '  </sensor>', '</scene>']
    path = os.path.join(out_dir, name + ".xml")
    open(path, "w").write("\n".join(x) + "\n")
    return path


def run_mitsuba(pkg, mitsuba_dir, name, threads, out_dir):
    exe = next((p for p in (os.path.join(mitsuba_dir, "mitsuba"), os.path.join(mitsuba_dir, "dist", "mitsuba"), os.path.join(mitsuba_dir, "build", "binaries", "mitsuba")) if os.path.exists(p)), None)
    if not exe:
        return {"config": name, "error": "no mitsuba binary under %s" % mitsuba_dir}
    conf = CONFIGS[name]
    sd = pkg.scenes.SCENES[conf["scene"]](res=conf["res"])
    xml = scene_to_xml(pkg, sd, conf, out_dir, name)
    defs = dict(type="orbital", fixEmitterPath="false", acceptanceMap="false")
    defs.update({k: str(v) for k, v in conf["mitsuba"].items() if k != "maxDepth"})
    cmd = [exe, "-p", str(threads), "-o", os.path.join(out_dir, name + ".exr")] + sum((["-D", "%s=%s" % kv] for kv in defs.items()), []) + [xml]
    env = dict(os.environ, LD_LIBRARY_PATH=os.pathsep.join([mitsuba_dir, os.path.join(mitsuba_dir, "dist"), os.environ.get("LD_LIBRARY_PATH", "")]))
    p = subprocess.run(cmd, capture_output=True, text=True, env=env)
    log = p.stdout + p.stderr
    m = re.search(r"Render time: ([0-9.]+)\s*(ms|s|m|h)", log)
    secs = float(m.group(1)) * {"ms": 1e-3, "s": 1, "m": 60, "h": 3600}[m.group(2)] if m else None
    stats = dict(re.findall(r"(Accepted [^:]+|Overall acceptance rate)\s*:\s*([0-9.]+ %)", log))
    muts = conf["res"] * conf["res"] * conf["spp"]
    return {"config": name, "command": " ".join(cmd), "threads": threads, "render_seconds": secs, "mutations_per_s": (muts / secs if secs else None),
            "statistics": stats, "image": os.path.join(out_dir, name + ".exr"), "returncode": p.returncode}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    pkg, ob = g.load_package(), g.load_oracle()
    ob.build(native=True)
    cores = host_threads()
    rows = []
    plan = [("C1", 64), ("C2", 64), ("C2", 32), ("C3", 64), ("C5", 64)]
    for name, prec in plan:
        for threads in (cores, 1):
            r = time_oracle(pkg, ob, name, prec, threads, a.seconds)
            rows.append(r)
            print("%s %s %2d threads: %.3e mutations/s (%.3e per core), %.3e path evaluations/s" % (name, r["precision"], threads, r["mutations_per_s"], r["mutations_per_s_per_core"], r["path_evals_per_s"]), flush=True)
    ref_rows = []
    md = os.environ.get("MITSUBA_DIR")
    if md and os.path.isdir(md):
        for name in ("C1", "C2", "C3"):
            ref_rows.append(run_mitsuba(pkg, md, name, cores, os.path.join("gpurun_out", "mitsuba_ref")))
            print("mitsuba", ref_rows[-1], flush=True)
    gcc = subprocess.run(["g++", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    out = {"cpu": cpu_model(), "host_threads": cores, "compiler": gcc, "flags": "-O3 -march=native", "rows": rows,
           "reference_binary": ref_rows if ref_rows else "unavailable: MITSUBA_DIR is not set on this box (the reference needs Boost / Xerces-C / OpenEXR, SURVEY 8c)"}
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)
    if a.md:
        with open(a.md, "w") as f:
            f.write("CPU: %s, %d threads available; %s %s\n\n" % (out["cpu"], cores, gcc, out["flags"]))
            f.write("| baseline row | config | threads | mutations/s | per core | path evaluations/s | first / overall acceptance |\n|---|---|---|---|---|---|---|\n")
            for r in rows:
                acc = r["acceptance"]
                f.write("| CPU restatement %s | %s | %d | %.3e | %.3e | %.3e | %s / %s |\n" % (r["precision"], r["config"], r["threads"], r["mutations_per_s"],
                        r["mutations_per_s_per_core"], r["path_evals_per_s"], acc.get("first"), acc.get("overall")))
            if ref_rows:
                for r in ref_rows:
                    f.write("| reference `mitsuba` binary | %s | %s | %s | — | — | %s |\n" % (r["config"], r.get("threads"), r.get("mutations_per_s"), r.get("statistics")))
            else:
                f.write("| reference `mitsuba` binary | C1–C3 | — | unavailable (no `MITSUBA_DIR` on the box) | — | — | — |\n")


if __name__ == "__main__":
    main()
