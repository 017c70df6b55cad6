"""C++ host side above the C-ABI (drmlt-mitsuba_amd/host/): parameter surface and error behaviour of the
reference's DRMLT ctor (drmlt.cpp:193-349), and on a GPU the full render() sequence through the CLI."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "drmlt-mitsuba_amd", "host")
CLI = os.path.join(HOST, "drmlt_render")


@pytest.fixture(scope="module")
def cli(native_lib):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return CLI


def run(cli, *args):
    p = subprocess.run([cli] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def D(**kw):
    out = []
    for k, v in kw.items():
        out += ["-D", "%s=%s" % (k, v)]
    return out


def test_defaults_and_required_parameters(cli):
    rc, out, err = run(cli, "--check", *D(technique="path", type="mirasym", maxDepth=8))
    assert rc == 0
    kv = dict(t.split("=") for t in out.split())
    assert kv["type"] == "2" and kv["rrDepth"] == "5" and kv["directSamples"] == "16"
    assert kv["luminanceSamples"] == "100000" and kv["workUnits"] == "-1" and float(kv["pLarge"]) == pytest.approx(0.3)
    assert float(kv["sigma"]) == 1 / 64 and float(kv["scaleSecond"]) == pytest.approx(0.1)
    for missing, needle in ((dict(type="orbital"), '"technique" has not been specified'),
                            (dict(technique="path"), '"type" has not been specified')):
        rc, out, err = run(cli, "--check", *D(**missing))
        assert rc == 1 and needle in err


@pytest.mark.parametrize("kw,needle", [
    (dict(technique="foo", type="orbital"), "Unknown technique type"),
    (dict(technique="path", type="foo"), "Unknown implementation type"),
    (dict(technique="mmlt", type="orbital"), "Impossible to use MMLT with no max depth"),
    (dict(technique="path", type="orbital", fixEmitterPath="true"), "Impossible to use fixEmitterPath without MMLT"),
    (dict(technique="path", type="orbital", scaleSecond="1.5"), "scaleSecond is bigger than the first stage"),
    (dict(technique="path", type="orbital", maxDepth="deep"), "wrong type"),
])
def test_ctor_errors_match_the_reference(cli, kw, needle):
    rc, out, err = run(cli, "--check", *D(**kw))
    assert rc == 1 and needle in err, err


@pytest.mark.gpu
def test_cli_render_matches_python_binding(cli, pkg, tmp_path):
    sd = pkg.scenes.cornell_c2(32)
    scene = str(tmp_path / "c2.bin")
    sd.save(scene)
    out = str(tmp_path / "o.pfm")
    rc, stdout, err = run(cli, scene, "-o", out, *D(technique="path", type="orbital", maxDepth=8, directSamples=-1,
                                                   workUnits=1024, luminanceSamples=20000, sampleCount=64))
    assert rc == 0, err
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"32 32\n" and f.readline() == b"-1.0\n"
        img = np.frombuffer(f.read(), dtype="<f4").reshape(32, 32, 3)[::-1]
    cfg = pkg.abi.make_config(type="orbital", max_depth=8, direct_samples=-1, work_units=1024, luminance_samples=20000,
                              sample_count=64)
    ctx = pkg.Context(cfg, sd)
    b = ctx.seed(0x5EED)
    ctx.run(32 * 32 * 64)
    ref = ctx.develop()
    assert ("b=%.9g" % b) in stdout
    # same seed, same chains: only the order of the float atomics differs between the two runs
    assert np.allclose(img, ref, rtol=1e-3, atol=1e-5)
