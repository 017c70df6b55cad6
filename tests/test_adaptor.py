"""The Mitsuba plugin adaptor (drmlt-mitsuba_amd/host/mitsuba_adaptor.cpp), compiled against the fake Mitsuba headers of
tests/native/fake_mitsuba and driven the way Mitsuba drives an Integrator plugin: CreateInstance(props) -> preprocess ->
render, cancel() from a second thread. CPU tests link a recording stand-in for the C-ABI (what does the plugin hand
over?); the GPU test links libdrmlt_amd.so and renders through the plugin surface."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "drmlt-mitsuba_amd", "host")


@pytest.fixture(scope="module")
def harness_mock():
    subprocess.run(["make", "-C", HOST, "adaptor_harness_mock"], check=True, capture_output=True)
    return os.path.join(HOST, "adaptor_harness_mock")


def D(**kw):
    out = []
    for k, v in kw.items():
        out += ["-D", "%s=%s" % (k, str(v).lower() if isinstance(v, bool) else v)]
    return out


def run(exe, scene, prefix, *args):
    p = subprocess.run([exe, scene, prefix, *args], capture_output=True, text=True, timeout=300)
    log = [l.rstrip("\n").split("\t", 1) for l in open(prefix + ".log")] if os.path.exists(prefix + ".log") else []
    return p.returncode, [(int(a), b) for a, b in log]


def read_scene(abi, path):
    raw = open(path, "rb").read()
    hdr = struct.unpack_from("<8I", raw, 0)
    assert hdr[0] == 0x4C4D5244 and hdr[1] == abi.ABI_VERSION
    off = 32
    def take(T, n):
        nonlocal off
        arr = (T * n).from_buffer_copy(raw, off)
        off += C.sizeof(T) * n
        return list(arr)
    shapes, bsdfs, emitters = take(abi.Shape, hdr[2]), take(abi.Bsdf, hdr[3]), take(abi.Emitter, hdr[4])
    cam = abi.Camera.from_buffer_copy(raw, off)
    return shapes, bsdfs, emitters, cam


BASE = dict(technique="path", type="orbital", maxDepth=8, directSamples=-1, sampleCount=16, workUnits=1024, seed=1234)


@pytest.mark.parametrize("name", ["cornell_c2", "door_c3", "caustic_c5"])
def test_flattening_reproduces_the_scene_arrays(pkg, harness_mock, tmp_path, name):
    """Rectangle / Sphere / TriMesh shapes, diffuse / dielectric / rough-conductor BSDFs (alpha, eta, k, distribution,
    specularReflectance), area lights, perspective sensor, film and filter come out as scenes.py built them."""
    abi = pkg.abi
    sd = pkg.scenes.SCENES[name](res=48)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, *D(**BASE))
    assert rc == 0, log
    shapes, bsdfs, emitters, cam = read_scene(abi, prefix + ".scene")
    assert len(shapes) == len(sd.shapes) and len(emitters) == len(sd.emitters)
    for a, b in zip(shapes, sd.shapes):
        assert a.type == b.type and a.emitter == b.emitter
        if a.type == abi.SHAPE_SPHERE: assert list(a.data) == pytest.approx(list(b.data), rel=1e-6, abs=1e-7)   # centre / radius come back out of the AABB
        else: assert list(a.data) == list(b.data)
        ba, bb = bsdfs[a.bsdf], sd.bsdfs[b.bsdf]          # the plugin does not share BSDF records between shapes
        assert ba.type == bb.type
        if ba.type == abi.BSDF_DIELECTRIC: assert list(ba.p)[:2] == pytest.approx(list(bb.p)[:2], rel=1e-6)   # rgb is unused for this type
        else: assert list(ba.rgb) == pytest.approx(list(bb.rgb), rel=1e-6) and list(ba.p) == pytest.approx(list(bb.p), rel=1e-6)
    for a, b in zip(emitters, sd.emitters):
        assert (a.type, a.shape, list(a.radiance), a.sampling_weight) == (b.type, b.shape, list(b.radiance), b.sampling_weight)
    assert bytes(cam) == bytes(sd.camera)


def test_area_light_on_a_mesh_becomes_one_emitter_per_triangle(pkg, harness_mock, tmp_path):
    abi = pkg.abi
    sd = pkg.scenes.cornell_c2(32)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, "--meshlight", *D(**BASE))
    assert rc == 0, log
    shapes, bsdfs, emitters, cam = read_scene(abi, prefix + ".scene")
    light = [s for s in sd.shapes if s.emitter >= 0]
    assert len(light) == 1 and len(shapes) == len(sd.shapes) + 1 and len(emitters) == 2
    tris = [s for s in shapes if s.emitter >= 0]
    assert [t.type for t in tris] == [abi.SHAPE_TRIANGLE] * 2 and [t.emitter for t in tris] == [0, 1]
    w0 = sd.emitters[0].sampling_weight
    for e, t in zip(emitters, tris):
        assert shapes[e.shape] is t and list(e.radiance) == list(sd.emitters[0].radiance)
        assert e.sampling_weight == pytest.approx(0.5 * w0, rel=1e-6)         # two halves of a rectangle: equal areas
    # the two triangles tile the rectangle: same corner set as the rectangle's toWorld gives
    m = np.array(list(light[0].data)).reshape(3, 4)
    corners = {tuple(np.round(m[:, :3] @ np.array([u, v, 0.0]) + m[:, 3], 5)) for u in (-1, 1) for v in (-1, 1)}
    got = {tuple(np.round(np.array(list(t.data))[3 * k:3 * k + 3], 5)) for t in tris for k in range(3)}
    assert got == corners


def test_seeding_and_work_unit_rules_are_scene_properties(pkg, harness_mock, tmp_path):
    """The reference's two-stage seeding rule (pathsampler.cpp:901-905) and its work-unit formula (drmlt.cpp:434-444) are
    selectable from scene XML -- properties of the plugin, fields of drmlt_config, nothing in the environment."""
    abi = pkg.abi
    sd = pkg.scenes.cornell_c2(16)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, *D(firstStageSeeding="reference", workUnitsRule="reference", **BASE))
    assert rc == 0, log
    cfg = abi.Config.from_buffer_copy(open(prefix + ".cfg", "rb").read())
    assert (cfg.seed_rule, cfg.work_units_rule) == (abi.SEED_REFERENCE, abi.WORK_UNITS_REFERENCE)
    for bad in (dict(firstStageSeeding="plain"), dict(workUnitsRule="cpu")):
        rc, log = run(harness_mock, scene, prefix, *D(**dict(BASE, **bad)))
        assert rc != 0 and any("Unknown " + list(bad)[0] in t for _, t in log), log


def test_parameters_statistics_progress_and_direct_pass(pkg, harness_mock, tmp_path):
    abi = pkg.abi
    sd = pkg.scenes.cornell_c2(16)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, *D(technique="bdpt", type="mirasym", maxDepth=6, rrDepth=3, pLarge=0.25, sampleCount=8,
                                                  luminanceSamples=5000, acceptanceMap=False, timidAfterLarge=True, useMixture=True,
                                                  sigma=0.02, scaleSecond=0.5, timeout=7, lightImage=False, devices=5, seed=99,
                                                  roundtrip=True))
    assert rc == 0, log
    raw = open(prefix + ".cfg", "rb").read()
    cfg = abi.Config.from_buffer_copy(raw)
    mask = struct.unpack_from("<I", raw, C.sizeof(abi.Config))[0]
    assert (cfg.technique, cfg.type, cfg.max_depth, cfg.rr_depth, cfg.sample_count) == (abi.TECH_BDPT, abi.TYPE_ORBITAL, 6, 3, 8)
    assert cfg.p_large == pytest.approx(0.25) and cfg.sigma == pytest.approx(0.02) and cfg.scale_second == pytest.approx(0.5)
    assert (cfg.timid_after_large, cfg.use_mixture, cfg.timeout_s, cfg.no_light_image, cfg.luminance_samples) == (1, 1, 7, 1, 5000)
    assert cfg.direct_samples == 16 and cfg.no_direct_sampling == 0       # reference defaults: directSamples=16, directSampling=true
    assert cfg.work_units == -1 and cfg.average_luminance == -1.0 and mask == 5 and cfg.struct_size == C.sizeof(abi.Config)
    assert (cfg.seed_rule, cfg.work_units_rule) == (abi.SEED_TARGET, abi.WORK_UNITS_DEVICE)   # the backend's defaults
    assert open(prefix + ".seed").read().strip() == "99"
    assert os.path.getsize(prefix + ".ser") == C.sizeof(abi.Config) + 4 + 1 + 1 + 4 + 1 + 4   # Integrator::serialize round trip payload (mask, twoStage, firstStage, reduction, hasSeed, seed)
    text = [t for _, t in log]
    # the seven StatsCounters of drmlt_proc.cpp:34-49
    for name, pct in (("Accepted 1st-stage mutations", 10), ("Accepted large mutations in the 1st stage", 20),
                      ("Accepted bold mutation in the 1st stage", 30), ("Accepted 2nd-stage mutations :", 40),
                      ("Accepted 2nd-stage mutations after large mutation", 50), ("Accepted 2nd-stage mutations after bold mutation", 60),
                      ("Overall acceptance rate", 70)):
        assert any(t.startswith(name) and t.endswith("%.2f %%" % pct) for t in text), (name, text)
    assert any("Normalization factor computed: 0.125" in t for t in text)
    assert "progress updates: 21" in text[-1]                              # 20 launches + finish()
    assert any("render returned true, 1 refresh signal(s), 16 direct pass sample(s)" in t for t in text)
    img = np.fromfile(prefix + ".img", dtype=np.float32)
    assert img.size == 16 * 16 * 3 and np.all(img == 1.25)                 # develop(direct): MLT image + the host's direct pass


def test_cancel_from_another_thread_makes_render_return_false(pkg, harness_mock, tmp_path):
    sd = pkg.scenes.cornell_c2(16)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, "--cancel-after-ms", "50", *D(**BASE))
    assert rc == 0, log                                                    # no exception: cancellation is not an error
    assert any("render returned false, 0 refresh signal(s)" in t for _, t in log)
    assert not os.path.exists(prefix + ".img")


def test_refusals_match_the_reference_messages(pkg, harness_mock, tmp_path):
    sd = pkg.scenes.cornell_c2(16)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    def fails(msg, *extra, **kw):
        prefix = str(tmp_path / ("r%d" % len(os.listdir(tmp_path))))
        rc, log = run(harness_mock, scene, prefix, *extra, *D(**kw))
        assert rc == 1 and any(lvl >= 400 and msg in t for lvl, t in log), (msg, log)
    fails("Unknown technique type", technique="erpt", type="orbital")
    fails("Unknown implementation type", technique="path", type="nope")
    fails('Property "technique" has not been specified', type="orbital")
    fails("requires the independent sampler", technique="path", type="orbital", maxDepth=8, sampler="ldsampler")
    fails("finite maxDepth", technique="path", type="orbital")             # the C-ABI's message reaches Log(EError)
    door = str(tmp_path / "door.bin")
    pkg.scenes.door_c3(16).save(door)
    prefix = str(tmp_path / "rc")
    # a named material goes through the FileResolver like roughconductor.cpp:176-179; "none" is the perfect mirror
    rc, log = run(harness_mock, door, prefix, "--rc-material", "Cu", "--resolver-prefix", "fake:0.2_0.9_1.1/", *D(**BASE))
    assert rc == 0, log
    _, bsdfs, _, _ = read_scene(pkg.abi, prefix + ".scene")
    rcb = [b for b in bsdfs if b.type == pkg.abi.BSDF_ROUGHCONDUCTOR][0]
    assert list(rcb.p)[1:7] == pytest.approx([0.2, 0.9, 1.1, 0.2, 0.9, 1.1])
    rc, log = run(harness_mock, door, prefix, "--rc-material", "none", *D(**BASE))
    _, bsdfs, _, _ = read_scene(pkg.abi, prefix + ".scene")
    rcb = [b for b in bsdfs if b.type == pkg.abi.BSDF_ROUGHCONDUCTOR][0]
    assert rc == 0 and list(rcb.p)[1:7] == [0, 0, 0, 1, 1, 1]


def test_two_stage_hands_over_the_importance_map(pkg, harness_mock, tmp_path):
    sd = pkg.scenes.cornell_c2(32)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(harness_mock, scene, prefix, *D(twoStage=True, firstStageSizeReduction=4, **BASE))
    assert rc == 0 and os.path.exists(prefix + ".imp") and any("Executing first MLT stage" in t for _, t in log), log


@pytest.mark.gpu
def test_render_through_the_plugin_surface_on_the_gpu(pkg, tmp_path, native_lib):
    """CreateInstance -> preprocess -> render with the real libdrmlt_amd.so: the film Mitsuba would receive equals the
    render of the same configuration through the Python binding of the same C-ABI (same seed => same chains)."""
    subprocess.run(["make", "-C", HOST, "adaptor_harness"], check=True, capture_output=True)
    exe = os.path.join(HOST, "adaptor_harness")
    sd = pkg.scenes.door_c3(32)
    scene = str(tmp_path / "s.bin")
    sd.save(scene)
    prefix = str(tmp_path / "o")
    rc, log = run(exe, scene, prefix, *D(technique="path", type="green", maxDepth=8, directSamples=-1, sampleCount=32,
                                         workUnits=1024, luminanceSamples=20000, seed=4242, device=0))
    assert rc == 0, log
    img = np.fromfile(prefix + ".img", dtype=np.float32).reshape(32, 32, 3)
    cfg = pkg.abi.make_config(technique="path", type="green", max_depth=8, direct_samples=-1, sample_count=32, work_units=1024,
                              luminance_samples=20000)
    node = pkg.Node(cfg, sd, device_mask=1)
    node.seed(4242)
    node.run(32 * 32 * 32)
    np.testing.assert_allclose(img, node.develop(), rtol=2e-4, atol=1e-6)
    assert any("Overall acceptance rate" in t for _, t in log) and any("mutations/s on the device" in t for _, t in log)
    # cancellation on the real library: cancel() lands between kernel launches, render() returns false
    rc, log = run(exe, scene, prefix + "c", "--cancel-after-ms", "30", *D(technique="path", type="green", maxDepth=8, directSamples=-1,
                                                                            sampleCount=200000, workUnits=4096, seed=1))
    assert rc == 0 and any("render returned false" in t for _, t in log), log
