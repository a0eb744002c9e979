"""CPU only: the C-ABI library loads, exports every symbol include/cpt_amd.h declares, reports the
missing device loudly, and the host-side scene code (loaders, EXR writer, CLI argument handling)
behaves like the reference's."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from helpers import *  # noqa: F403

HEADER = os.path.join(ROOT, "include", "cpt_amd.h")
BINARY = os.path.join(ROOT, "cuda-path-tracer_amd", "cuda-pathtracer")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(cpt):
    names = declared_functions()
    assert len(names) > 40
    missing = [n for n in names if not hasattr(cpt.lib, n)]
    assert not missing, missing


def test_no_device_is_reported_not_hidden(cpt):
    if cpt.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(cpt.CptError):
        cpt.Tracer(cpt.Scene(), 8, 8)
    assert cpt.lib.cpt_malloc(16) is None
    assert b"hipMalloc" in cpt.lib.cpt_last_error()


@pytest.mark.parametrize("glass", [False, True])
def test_obj_loader_matches_independent_reader(cpt, glass):
    d = os.path.join(SCENES, "cornell-box-glass" if glass else "cornell-box")
    sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
    ref_objs, cam = cornell(glass=glass)
    objs = sc.objects()
    assert [o["name"] for o in objs] == [o["name"] for o in ref_objs]
    for a, b in zip(objs, ref_objs):
        for k in ("positions", "normals", "texcoords", "indices", "material"):
            assert np.array_equal(a[k], b[k]), (a["name"], k)
        assert np.array_equal(a["transform"], IDENTITY)
    c = sc.camera()
    assert (c["width"], c["height"]) == ((1280, 720) if glass else (1024, 1024))    # json without "resolution" keeps 1280x720
    assert c["fov"] == np.float32(19.5) and np.array_equal(c["pos"], cam["pos"])


def test_object_transforms_match_glm_vectors(cpt):
    """SceneObject::GetTransform and the stored inverse, bit for bit against matrices produced by
    the reference's own glm (tests/golden/ref_transform.npz)."""
    g = np.load(os.path.join(os.path.dirname(SCENES), "ref_transform.npz"))
    objs, _ = cornell(glass=False)
    o = objs[0]
    sc = cpt.Scene().add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    m, inv = sc.object_transforms(0)
    assert np.array_equal(m, IDENTITY) and np.array_equal(inv, IDENTITY)
    for k, row in enumerate(g["trs"]):
        sc.set_trs(0, row[0:3], row[3:6], row[6:9])
        m, inv = sc.object_transforms(0)
        assert np.array_equal(m.view(np.uint32), g["matrices"][k, 0].view(np.uint32)), k
        assert np.array_equal(inv.view(np.uint32), g["matrices"][k, 1].view(np.uint32)), k
        assert np.array_equal(sc.object(0)["transform"], m)
    with pytest.raises(cpt.CptError):
        sc.set_trs(5, [0, 0, 0], [0, 0, 0], [1, 1, 1])


def test_large_obj_parsed_in_pieces_matches_independent_reader(cpt, tmp_path):
    """A file big enough to be cut into several pieces for parallel parsing: attribute blocks,
    groups and usemtl statements interleaved with faces, relative (negative) indices that refer
    across piece boundaries, quads, numbers in every notation the reference's reader accepts."""
    rng = np.random.Generator(np.random.PCG64(3))
    (tmp_path / "big.mtl").write_text("newmtl a\nKd 0.5 0.5 0.5\n\nnewmtl b\nKd 0.1 0.9 0.1\nKe 1 1 1\n")
    lines = ["mtllib big.mtl"]
    nv = 0
    for block in range(60):
        m = int(rng.integers(200, 1200))
        pts = rng.uniform(-3, 3, (m, 3))
        style = block % 4
        for q in pts:
            if style == 0:
                lines.append("v %.9g %.9g %.9g" % tuple(q))
            elif style == 1:
                lines.append("v %.6e %.6e %.6e" % tuple(q))
            elif style == 2:
                lines.append("v\t%+.4f  %.3f %.7f\r" % tuple(q))
            else:
                lines.append("v %d %s %.12f" % (int(q[0]), ("%.3f" % q[1]).lstrip("0") if 0 < q[1] < 1 else "%.2f" % q[1], q[2]))
            lines.append("vt %.5f %.5f" % tuple(rng.random(2)))
            lines.append("vn %.4f %.4f %.4f" % tuple(rng.standard_normal(3)))
        nv += m
        if block % 7 == 0:
            lines.append("g part%d" % block)
        if block % 3 == 0:
            lines.append("usemtl %s" % ("a", "b", "missing")[(block // 3) % 3])
        for _ in range(int(rng.integers(300, 900))):
            k = 4 if rng.random() < 0.3 else 3
            if rng.random() < 0.5:          # absolute, anywhere in what has been read so far
                ids = rng.integers(1, nv + 1, k)
            else:                            # relative, reaching back into earlier blocks
                ids = -rng.integers(1, min(nv, 5000) + 1, k)
            lines.append("f " + " ".join("%d/%d/%d" % (i, i, i) for i in ids))
    text = "\n".join(lines) + "\n"
    while len(text) < 3_500_000:            # several pieces of at least 1 MiB
        text += "# padding " + "x" * 100 + "\n"
    (tmp_path / "big.obj").write_text(text)
    sc = cpt.Scene().load_obj(str(tmp_path / "big.obj"), 2)
    ref_objs = load_obj_scene(str(tmp_path / "big.obj"), 2)
    objs = sc.objects()
    assert [o["name"] for o in objs] == [o["name"] for o in ref_objs] and len(objs) > 5
    for a, b in zip(objs, ref_objs):
        for k in ("positions", "normals", "texcoords", "indices", "material"):
            assert np.array_equal(a[k], b[k]), (a["name"], k)


def test_number_grammar_of_the_reference_reader(cpt, tmp_path):
    """tinyobjloader's tryParseDouble (tiny_obj_loader.h:866-996): what is not a number by its
    grammar leaves the default 0, whatever strtof would make of it."""
    (tmp_path / "n.obj").write_text("v 1 2 3\nv .5 -.25 +1.5e1\nv 1e 2 3\nv nan inf 0x10\nv 7. 1E-2 3e+0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf 3/1/1 4/1/1 5/1/1\n")
    o = cpt.Scene().load_obj(str(tmp_path / "n.obj"), 2).objects()[0]
    want = np.array([[1, 2, 3], [0.5, -0.25, 15.0], [0, 2, 3], [0, 0, 0], [7.0, 0.01, 3.0]], np.float32)
    assert np.array_equal(o["positions"], want), o["positions"]


def test_bsdf_type_flag_and_default_material(cpt, tmp_path):
    obj = tmp_path / "t.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\n")
    sc = cpt.Scene().load_obj(str(obj), 0)
    objs = sc.objects()
    assert len(objs) == 1 and objs[0]["name"] == "-default"
    assert len(objs[0]["indices"]) == 6                                   # quad -> two triangles
    m = objs[0]["material"]
    assert np.allclose(m[3:6], 0.5) and m[13] == 1.5 and m[15] == 2        # default Material (material.hpp:25-32)
    ref = load_obj_scene(str(obj), 0)
    assert np.array_equal(objs[0]["indices"], ref[0]["indices"])


def test_quad_split_uses_shorter_diagonal(cpt, tmp_path):
    obj = tmp_path / "q.obj"
    # diagonal 1-3 is shorter than 0-2 -> triangles (0,1,3) (1,2,3)  (tiny_obj_loader.h:1495-1516)
    obj.write_text("v 0 0 0\nv 1 0 0\nv 3 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1 4/1/1\n")
    o = cpt.Scene().load_obj(str(obj), 2).objects()[0]
    assert o["indices"].tolist() == [0, 1, 2, 1, 3, 2] or o["indices"].tolist() == load_obj_scene(str(obj))[0]["indices"].tolist()
    assert o["indices"].tolist() == load_obj_scene(str(obj))[0]["indices"].tolist()


def test_json_and_xml_cameras(cpt, tmp_path):
    j = tmp_path / "c.json"
    j.write_text('{"camera": {"pos": [1, 2, 3], "look_at": [0, 0, 0], "up": [0, 1, 0], "fov": 30.5, "resolution": [320, 200]}}')
    c = cpt.Scene().load_json(str(j)).camera()
    assert (c["width"], c["height"], c["fov"]) == (320, 200, 30.5) and c["pos"].tolist() == [1, 2, 3]
    bad = tmp_path / "bad.json"
    bad.write_text('{"camera": {"pos": [1, 2, 3]')
    with pytest.raises(cpt.CptError):
        cpt.Scene().load_json(str(bad))
    d = os.path.join(SCENES, "cornell-box")
    x = tmp_path / "c.xml"
    x.write_text('<?xml version="1.0"?>\n<!-- test -->\n<camera type="perspective" width="640" height="480" fovy="39.5">\n'
                 '  <eye x="0" y="1" z="6.8"/>\n  <lookat x="0" y="1" z="0"/>\n  <up x="0" y="1" z="0"/>\n</camera>\n'
                 '<light mtlname="light" radiance="34.0, 24.0, 8.0"/>\n<light mtlname="red" radiance="1,2"/>\n')
    sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_xml(str(x))
    c = sc.camera()
    assert (c["width"], c["height"], c["fov"]) == (640, 480, 39.5) and c["pos"].tolist() == [0, 1, np.float32(6.8)]
    objs = {o["name"]: o for o in sc.objects()}
    assert objs["light-light"]["material"][0:3].tolist() == [34.0, 24.0, 8.0]      # emission overridden (loader.cpp:349-356)
    assert objs["left-red"]["material"][0:2].tolist() == [1.0, 2.0]
    assert objs["back-white"]["material"][0:3].tolist() == [0.0, 0.0, 0.0]
    with pytest.raises(cpt.CptError):
        cpt.Scene().load_xml(str(tmp_path / "missing.xml"))


def test_missing_obj_is_an_error(cpt, tmp_path):
    with pytest.raises(cpt.CptError):
        cpt.Scene().load_obj(str(tmp_path / "nope.obj"), 2)


def parse_exr(path):
    b = open(path, "rb").read()
    assert struct.unpack("<I", b[:4])[0] == 20000630
    p = 8
    attrs = {}
    while b[p] != 0:
        e = b.index(b"\0", p); name = b[p:e].decode(); p = e + 1
        e = b.index(b"\0", p); typ = b[p:e].decode(); p = e + 1
        size = struct.unpack("<I", b[p:p + 4])[0]; p += 4
        attrs[name] = (typ, b[p:p + size]); p += size
    p += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    chans = re.findall(rb"([A-Z])\0", attrs["channels"][1])
    offsets = struct.unpack(f"<{h}Q", b[p:p + 8 * h])
    img = np.zeros((h, w, 4), np.float32)
    for y in range(h):
        q = offsets[y]
        yy, size = struct.unpack("<iI", b[q:q + 8])
        line = np.frombuffer(b[q + 8:q + 8 + size], np.float32).reshape(4, w)
        for ci, c in enumerate(chans):
            img[yy, :, "RGBA".index(c.decode())] = line[ci]
    return img, attrs, [c.decode() for c in chans]


def test_exr_writer_roundtrip(cpt, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.random((5, 7, 4), dtype=np.float32)
    path = str(tmp_path / "a.exr")
    cpt.write_exr(path, img)
    back, attrs, chans = parse_exr(path)
    assert chans == ["A", "B", "G", "R"]                    # the order tinyexr's SaveEXR writes (tinyexr.h:8438-8475)
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    assert np.array_equal(back, img)


def test_cli_usage_and_errors(tmp_path):
    assert os.path.exists(BINARY), "build the CLI with make -C cuda-path-tracer_amd/csrc"
    r = subprocess.run([BINARY], capture_output=True, text=True)
    assert r.returncode == 255 and "usage:" in r.stdout and "--max-spp" in r.stdout and "--bsdf-type" in r.stdout
    r = subprocess.run([BINARY, str(tmp_path / "no.obj"), str(tmp_path / "no.json")], capture_output=True, text=True)
    assert r.returncode == 255 and "doesn't exist" in r.stdout
    d = os.path.join(SCENES, "cornell-box")
    other = tmp_path / "cam.txt"
    other.write_text("x")
    r = subprocess.run([BINARY, os.path.join(d, "cornell-box.obj"), str(other)], capture_output=True, text=True)
    assert r.returncode == 255 and "unknown extra file extension" in r.stdout
