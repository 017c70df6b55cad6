"""Cuboid records of the brute-force ray loop, host side (csrc/box_merge.h): faces that bound a parallelepiped -- a `cube`'s six
merged triangle pairs, the walls of a room -- are found from the records' world-space parallelograms; every face keeps its own
(u, v) parametrisation through a three-bit code. Checked on the CPU against the scenes of scenes.py."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "drmlt-mitsuba_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("box") / "box_harness")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", CSRC, "-o", exe, os.path.join(ROOT, "tests", "native", "box_harness.cpp")], check=True)

    def run(quads, min_faces=4):
        text = "".join(("x\n" if q is None else " ".join("%.17g" % v for v in np.concatenate(q)) + "\n") for q in quads)
        return json.loads(subprocess.run([exe, str(min_faces)], input=text, check=True, capture_output=True, text=True).stdout)
    return run


def scene_quads(pkg, sd):
    """The parallelograms drmlt_create sees: rectangles as (corner (-1,-1), full edges), triangle pairs (a,b,c),(a,c,d) that
    form a parallelogram as (a, b - a, d - a); anything else is None."""
    abi = pkg.abi
    out, i = [], 0
    sh = sd.shapes
    while i < len(sh):
        s = sh[i]
        d = np.array(list(s.data), dtype=np.float64)
        if s.type == abi.SHAPE_RECTANGLE:
            m = d.reshape(3, 4)
            eu, ev, c = m[:, 0], m[:, 1], m[:, 3]
            out.append((c - eu - ev, 2 * eu, 2 * ev))
            i += 1
        elif s.type == abi.SHAPE_TRIANGLE and i + 1 < len(sh) and sh[i + 1].type == abi.SHAPE_TRIANGLE and s.emitter < 0:
            a, b, c = d[0:3], d[3:6], d[6:9]
            d2 = np.array(list(sh[i + 1].data), dtype=np.float64)
            if np.array_equal(d2[0:3], a) and np.array_equal(d2[3:6], c) and np.abs(d2[6:9] - (a + c - b)).max() < 1e-6:
                out.append((a, b - a, d2[6:9] - a))
                i += 2
            else:
                out.append(None)
                i += 1
        else:
            out.append(None)
            i += 1
    return out


def check_codes(quads, box):
    """Every face's own (u, v) -> world point equals the cuboid's (p, q) -> world point under the face's code."""
    a, E = np.array(box["a"]), np.array(box["E"])
    rng = np.random.default_rng(0)
    for f in range(6):
        qi = box["face"][f]
        if qi < 0:
            continue
        axis, side = f >> 1, f & 1
        j, k = (1 if axis == 0 else 0), (1 if axis == 2 else 2)
        code = box["code"][f]
        qa, e1, e2 = quads[qi]
        for p, q in rng.random((8, 2)):
            b = np.zeros(3)
            b[axis], b[j], b[k] = side, p, q
            world = a + b @ E
            uu, vv = (q, p) if code & 1 else (p, q)
            u = 1 - uu if code & 2 else uu
            v = 1 - vv if code & 4 else vv
            assert np.abs(qa + u * e1 + v * e2 - world).max() < 1e-6   # (scene coordinates are fp32)


def test_cornell_box_is_a_room_and_two_cubes(pkg, harness):
    quads = scene_quads(pkg, pkg.scenes.cornell_c2(64))
    assert len(quads) == 18 and sum(q is None for q in quads) == 0      # 5 walls, 12 merged pairs, the light
    boxes = harness(quads)
    assert sorted(b["n_faces"] for b in boxes) == [5, 6, 6]              # the room (open towards the camera) and the two boxes
    used = [f for b in boxes for f in b["face"] if f >= 0]
    assert len(used) == len(set(used)) == 17 and 17 not in used          # every wall / box face once; the light stays a flat record
    room = next(b for b in boxes if b["n_faces"] == 5)
    assert sorted(f for f in room["face"] if f >= 0) == [0, 1, 2, 3, 4]
    missing = room["face"].index(-1)
    a, E = np.array(room["a"]), np.array(room["E"])
    centre = a + (np.eye(3)[missing >> 1] * (missing & 1) + 0.5 * (1 - np.eye(3)[missing >> 1])) @ E
    assert np.allclose(centre, (0, 0, 1))                                # the open side faces the camera (z = +1)
    for b in boxes:
        check_codes(quads, b)


@pytest.mark.parametrize("name,want", [("door_c3", [6]), ("caustic_c5", [5]), ("glass_sphere", [5]), ("cornell_c1", [])])
def test_other_scenes(pkg, harness, name, want):
    quads = scene_quads(pkg, pkg.scenes.SCENES[name](res=32))
    boxes = harness(quads)
    assert sorted(b["n_faces"] for b in boxes) == want                   # door: the closed room; the partition's two sides and lights stay flat
    for b in boxes:
        check_codes(quads, b)


def test_sheared_and_rotated_parallelepiped_with_shuffled_faces(harness):
    rng = np.random.default_rng(4)
    a = rng.normal(size=3)
    E = rng.normal(size=(3, 3))                                          # a general parallelepiped: no right angles
    quads = []
    for axis in range(3):
        j, k = (1 if axis == 0 else 0), (1 if axis == 2 else 2)
        for side in (0, 1):
            corner = a + side * E[axis]
            # an arbitrary one of the face's eight parametrisations
            cj, ck, swap = rng.integers(0, 2, 3)
            qa = corner + cj * E[j] + ck * E[k]
            eu, ev = (1 - 2 * cj) * E[j], (1 - 2 * ck) * E[k]
            quads.append((qa, ev, eu) if swap else (qa, eu, ev))
    quads.insert(3, None)
    quads.append((rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)))   # a stray parallelogram
    order = rng.permutation(len(quads))
    quads = [quads[i] for i in order]
    boxes = harness(quads)
    assert len(boxes) == 1 and boxes[0]["n_faces"] == 6
    check_codes(quads, boxes[0])
    # with two faces removed there are still four: a cuboid; with three removed it is not worth a record
    drop = [f for f in boxes[0]["face"]][:2]
    boxes4 = harness([None if i in drop else q for i, q in enumerate(quads)])
    assert len(boxes4) == 1 and boxes4[0]["n_faces"] == 4
    drop = [f for f in boxes[0]["face"]][:3]
    assert harness([None if i in drop else q for i, q in enumerate(quads)]) == []
