"""CPU only: the C-ABI library loads, exports every symbol include/cpt_amd.h declares, reports the
missing device loudly, and the host-side scene code (loaders, EXR writer, CLI argument handling)
behaves like the reference's."""
import os
import re
import struct
import subprocess
import sys

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


def test_binary_scene_copy_round_trip_and_staleness(cpt, tmp_path):
    """load_obj(cache=True): the first load parses and writes <obj>.cptcache, the second reads it and gives the
    same objects bit for bit (names, arrays, materials, maps); a changed .obj, a changed .mtl, another bsdf_type
    or a damaged copy are not served from it."""
    import shutil, time
    from helpers import textured_box_files
    obj, cam, maps = textured_box_files(str(tmp_path))
    plain = cpt.Scene().load_obj(obj, 2)
    first = cpt.Scene().load_obj(obj, 2, cache=True)
    assert not first.from_cache and os.path.exists(obj + ".cptcache")
    second = cpt.Scene().load_obj(obj, 2, cache=True)
    assert second.from_cache

    def same(a, b):
        oa, ob = a.objects(), b.objects()
        assert [o["name"] for o in oa] == [o["name"] for o in ob] and len(oa) > 1
        for x, y in zip(oa, ob):
            for k in ("positions", "normals", "texcoords", "indices", "material", "transform"):
                assert np.array_equal(x[k].view(np.uint32) if x[k].dtype == np.float32 else x[k], y[k].view(np.uint32) if y[k].dtype == np.float32 else y[k]), (x["name"], k)
        for i in range(len(oa)):
            for which in range(3):
                ma, mb = a.object_map(i, which), b.object_map(i, which)
                assert (ma is None) == (mb is None) and (ma is None or np.array_equal(ma, mb))
    same(plain, first)
    same(plain, second)
    assert any(second.object_map(i, 0) is not None for i in range(len(second.objects())))      # maps are there after a cached load

    # another bsdf_type: parsed again, copy rewritten for it
    other = cpt.Scene().load_obj(obj, 0, cache=True)
    assert not other.from_cache
    assert cpt.Scene().load_obj(obj, 0, cache=True).from_cache
    assert not cpt.Scene().load_obj(obj, 2, cache=True).from_cache
    # the material library changes
    mtl = [f for f in os.listdir(tmp_path) if f.endswith(".mtl")][0]
    with open(tmp_path / mtl, "a") as f:
        f.write("\n# touched\n")
    assert not cpt.Scene().load_obj(obj, 2, cache=True).from_cache
    assert cpt.Scene().load_obj(obj, 2, cache=True).from_cache
    # the .obj changes (same size, later time)
    st = os.stat(obj)
    os.utime(obj, ns=(st.st_atime_ns, st.st_mtime_ns + 2_000_000_000))
    assert not cpt.Scene().load_obj(obj, 2, cache=True).from_cache
    # a truncated copy is ignored and replaced
    blob = open(obj + ".cptcache", "rb").read()
    open(obj + ".cptcache", "wb").write(blob[: len(blob) // 2])
    third = cpt.Scene().load_obj(obj, 2, cache=True)
    assert not third.from_cache
    same(plain, third)
    open(obj + ".cptcache", "wb").write(blob + b"x")
    assert not cpt.Scene().load_obj(obj, 2, cache=True).from_cache
    assert cpt.Scene().load_obj(obj, 2, cache=True).from_cache


def test_number_grammar_of_the_reference_reader(cpt, tmp_path):
    """tinyobjloader's tryParseDouble (tiny_obj_loader.h:866-996): what is not a number by its
    grammar leaves the default 0, whatever strtof would make of it."""
    (tmp_path / "n.obj").write_text("v 1 2 3\nv .5 -.25 +1.5e1\nv 1e 2 3\nv nan inf 0x10\nv 7. 1E-2 3e+0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf 3/1/1 4/1/1 5/1/1\n")
    o = cpt.Scene().load_obj(str(tmp_path / "n.obj"), 2).objects()[0]
    want = np.array([[1, 2, 3], [0.5, -0.25, 15.0], [0, 2, 3], [0, 0, 0], [7.0, 0.01, 3.0]], np.float32)
    assert np.array_equal(o["positions"], want), o["positions"]


def _png(path, width, height, color, depth, rows, palette=None, trns=None, filters=None):
    import zlib
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    raw = bytearray()
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color] * depth // 8)
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = filters[y % len(filters)] if filters else 0
        out = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0: pred = 0
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) // 2
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
            out[i] = (x - pred) & 255
        raw += bytes([f]) + out
        prev = row
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, color, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(bytes(raw))
    data += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    open(path, "wb").write(data)


def test_image_readers(cpt, tmp_path):
    """PNG (every colour type, sub-byte depths, all five filters, split IDAT) and PPM/PGM against
    pixel arrays built in the test; the PNG writer round-trips."""
    rng = np.random.Generator(np.random.PCG64(4))
    w, h = 13, 9
    pack = lambda r, g, b, a: (r.astype(np.uint32) | (g.astype(np.uint32) << 8) | (b.astype(np.uint32) << 16) | (a.astype(np.uint32) << 24))
    full = np.full((h, w), 255, np.uint32)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    _png(tmp_path / "rgba.png", w, h, 6, 8, [rgba[y].tobytes() for y in range(h)], filters=[0, 1, 2, 3, 4])
    assert np.array_equal(cpt.load_image(tmp_path / "rgba.png"), pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], rgba[..., 3]))
    _png(tmp_path / "rgb.png", w, h, 2, 8, [rgba[y, :, :3].tobytes() for y in range(h)], filters=[4, 3])
    assert np.array_equal(cpt.load_image(tmp_path / "rgb.png"), pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], full))
    _png(tmp_path / "ga.png", w, h, 4, 8, [rgba[y, :, :2].tobytes() for y in range(h)], filters=[1])
    assert np.array_equal(cpt.load_image(tmp_path / "ga.png"), pack(rgba[..., 0], rgba[..., 0], rgba[..., 0], rgba[..., 1]))
    rgb16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    _png(tmp_path / "rgb16.png", w, h, 2, 16, [rgb16[y].astype(">u2").tobytes() for y in range(h)], filters=[2])
    hi = (rgb16 >> 8).astype(np.uint32)
    assert np.array_equal(cpt.load_image(tmp_path / "rgb16.png"), pack(hi[..., 0], hi[..., 1], hi[..., 2], full))
    idx = rng.integers(0, 4, (h, w), dtype=np.uint8)          # 2-bit palette with transparency
    rows = []
    for y in range(h):
        bits = np.zeros(((w * 2 + 7) // 8) * 4, np.uint8)
        bits[:w] = idx[y]
        rows.append(bytes(int(bits[4 * k] << 6 | bits[4 * k + 1] << 4 | bits[4 * k + 2] << 2 | bits[4 * k + 3]) for k in range(len(bits) // 4)))
    pal = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]], np.uint8)
    _png(tmp_path / "pal.png", w, h, 3, 2, rows, palette=pal.reshape(-1), trns=[255, 128])
    alpha = np.array([255, 128, 255, 255], np.uint32)[idx]
    assert np.array_equal(cpt.load_image(tmp_path / "pal.png"), pack(pal[idx][..., 0], pal[idx][..., 1], pal[idx][..., 2], alpha))
    g4 = rng.integers(0, 16, (h, w), dtype=np.uint8)          # 4-bit grey
    rows = [bytes(int((np.append(g4[y], 0)[2 * k] << 4) | np.append(g4[y], 0)[2 * k + 1]) for k in range((w + 1) // 2)) for y in range(h)]
    _png(tmp_path / "g4.png", w, h, 0, 4, rows)
    g8 = (g4.astype(np.uint32) * 255) // 15
    assert np.array_equal(cpt.load_image(tmp_path / "g4.png"), pack(g8, g8, g8, full))
    write_ppm(tmp_path / "a.ppm", pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], full))
    assert np.array_equal(cpt.load_image(tmp_path / "a.ppm"), pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], full))
    (tmp_path / "t.pgm").write_text("P2\n# comment\n3 2\n15\n0 5 15\n1 2 3\n")
    g = (np.array([[0, 5, 15], [1, 2, 3]], np.uint32) * 255 + 7) // 15
    assert np.array_equal(cpt.load_image(tmp_path / "t.pgm"), pack(g, g, g, np.full((2, 3), 255, np.uint32)))
    cpt.save_png(tmp_path / "round.png", pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], rgba[..., 3]))
    assert np.array_equal(cpt.load_image(tmp_path / "round.png"), pack(rgba[..., 0], rgba[..., 1], rgba[..., 2], rgba[..., 3]))
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 20)
    with pytest.raises(cpt.CptError):
        cpt.load_image(tmp_path / "bad.png")
    with pytest.raises(cpt.CptError):
        cpt.load_image(tmp_path / "missing.png")


def test_material_maps_are_loaded_with_the_scene(cpt, tmp_path):
    obj, cam, maps = textured_box_files(str(tmp_path))
    sc = cpt.Scene().load_obj(obj, 2)
    names = [o["name"] for o in sc.objects()]
    for i, name in enumerate(names):
        mat = name.split("-")[-1]
        for which, key in enumerate(("map_Kd", "map_Ks", "map_Ke")):
            got = sc.object_map(i, which)
            want = maps.get(mat, {}).get(key)
            assert (got is None) == (want is None), (name, key)
            if want is not None:
                assert np.array_equal(got, want), (name, key)
    (tmp_path / "broken.mtl").write_text("newmtl m\nKd 0.25 0.5 0.75\nmap_Kd nothing.png\n")
    (tmp_path / "broken.obj").write_text("mtllib broken.mtl\nusemtl m\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n")
    sc = cpt.Scene().load_obj(str(tmp_path / "broken.obj"), 2)     # a map that cannot be read leaves the constant
    assert sc.object_map(0, 0) is None and np.array_equal(sc.object(0)["material"][3:6], np.float32([0.25, 0.5, 0.75]))


def test_texture_fetch_of_the_oracle():
    """tex2D as the CUDA programming guide describes it for normalised, wrapped, linearly filtered,
    sRGB textures: texel centres, wrap-around neighbours, 8-bit blend weights."""
    L = oracle()
    img = np.array([[0x000000ff, 0x0000ff00], [0x00ff0000, 0x00ffffff]], np.uint32)       # red green / blue white
    out = np.zeros(3, np.float32)
    L.orc_tex2d(2, 2, img, 0.25, 0.25, out)
    assert np.array_equal(out, [1, 0, 0])                       # centre of texel (0, 0)
    L.orc_tex2d(2, 2, img, 0.75, 0.75, out)
    assert np.array_equal(out, [1, 1, 1])
    L.orc_tex2d(2, 2, img, 0.5, 0.25, out)                      # half way between red and green
    assert np.array_equal(out, [0.5, 0.5, 0])
    L.orc_tex2d(2, 2, img, 0.0, 0.25, out)                      # left edge blends with the wrapped right neighbour
    assert np.array_equal(out, [0.5, 0.5, 0])
    L.orc_tex2d(2, 2, img, 1.25, -0.75, out)                    # wrap in both directions
    assert np.array_equal(out, [1, 0, 0])
    grey = np.array([[0xff7f7f7f]], np.uint32)
    L.orc_tex2d(1, 1, grey, 0.3, 0.9, out)
    assert abs(out[0] - 0.2122) < 1e-3 and out[0] == out[1] == out[2]     # sRGB 127 -> linear


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


def test_xml_light_replaces_an_emission_map(cpt, tmp_path):
    """A <light> entry sets the radiance constant and drops the material's emission map, name and image
    (emission_map.reset(), /root/reference/src/scene/loader.cpp:349-356); also when the scene came from the binary copy."""
    from helpers import textured_box_files
    obj, cam, maps = textured_box_files(str(tmp_path))
    x = tmp_path / "room.xml"
    x.write_text('<camera type="perspective" width="64" height="64" fovy="45"><eye x="0" y="1" z="-2.6"/><lookat x="0" y="0.9" z="0"/>'
                 '<up x="0" y="1" z="0"/></camera>\n<light mtlname="lamp" radiance="3.5, 2.5, 1.5"/>\n')
    for cache in (False, True, True):
        sc = cpt.Scene().load_obj(obj, 2, cache=cache)
        lamp = [i for i, o in enumerate(sc.objects()) if o["name"].endswith("-lamp")][0]
        assert sc.object_map(lamp, 2) is not None
        sc.load_xml(str(x))
        assert sc.object_map(lamp, 2) is None
        assert sc.object(lamp)["material"][0:3].tolist() == [3.5, 2.5, 1.5]
        floor = [i for i, o in enumerate(sc.objects()) if o["name"].endswith("-floor")][0]
        assert sc.object_map(floor, 0) is not None                 # other maps stay


def test_damaged_headers_do_not_size_allocations(cpt, tmp_path):
    """A binary scene copy or a PNG whose counts announce far more data than the file holds is refused (and the copy
    re-made) instead of being allocated for."""
    import struct, zlib
    from helpers import textured_box_files
    obj, cam, maps = textured_box_files(str(tmp_path))
    cpt.Scene().load_obj(obj, 2, cache=True)
    blob = bytearray(open(obj + ".cptcache", "rb").read())
    plain = cpt.Scene().load_obj(obj, 2)
    nv = len(plain.object(0)["positions"])
    ni = len(plain.object(0)["indices"])
    needle = struct.pack("<QQ", nv, ni)
    at = bytes(blob).find(needle)
    assert at > 0
    blob[at:at + 16] = struct.pack("<QQ", 0xfffffff0, 3 << 38)      # 4G vertices, 2^39.6 indices: within the format's limits
    open(obj + ".cptcache", "wb").write(bytes(blob))
    sc = cpt.Scene().load_obj(obj, 2, cache=True)
    assert not sc.from_cache and len(sc.objects()) == len(plain.objects())
    assert cpt.Scene().load_obj(obj, 2, cache=True).from_cache

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 60000, 60000, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    (tmp_path / "huge.png").write_bytes(png)
    with pytest.raises(cpt.CptError):
        cpt.load_image(str(tmp_path / "huge.png"))


def test_sample_blocks_of_the_library_match_the_python_partition(cpt):
    sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
    import sharding
    for total, world, first in ((4096, 8, 1), (10, 4, 1), (3, 5, 1), (0, 2, 1), (17, 3, 9), (1, 1, 1)):
        got = [cpt.shard_samples(total, r, world, first) for r in range(world)]
        assert got == sharding.split_samples(total, world, first)
        covered = [k for b, n in got for k in range(b, b + n)]
        assert covered == list(range(first, first + total))
    with pytest.raises(cpt.CptError):
        cpt.shard_samples(8, 2, 2)


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
    for flag in ("--ui", "--max-depth", "--output, -o", "--gpus", "--shards"):
        assert flag in r.stdout
    r = subprocess.run([BINARY, str(tmp_path / "no.obj"), str(tmp_path / "no.json")], capture_output=True, text=True)
    assert r.returncode == 255 and "doesn't exist" in r.stdout
    d = os.path.join(SCENES, "cornell-box")
    other = tmp_path / "cam.txt"
    other.write_text("x")
    r = subprocess.run([BINARY, os.path.join(d, "cornell-box.obj"), str(other)], capture_output=True, text=True)
    assert r.returncode == 255 and "unknown extra file extension" in r.stdout
