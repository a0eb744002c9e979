"""CPU only: pins the oracle (oracle/oracle.cpp) to (a) values observed from the reference during
the survey, (b) golden vectors produced by the reference's own headers (tests/golden/*.npz, made by
tests/golden/make_golden.py) and (c), when oracle/_ref/libref.so is present, the reference code live."""
import json
import os

import numpy as np
import pytest

from helpers import *  # noqa: F403

KNOWN = json.load(open(os.path.join(GOLDEN, "survey_known_answers.json")))


def as_pairs(nodes):
    return [[int(a), -1 if int(b) == 0xFFFFFFFF else int(b)] for a, b in nodes]


@pytest.fixture(scope="module")
def glass():
    objs, cam = cornell(glass=True)
    return objs, cam, OracleScene(objs, cam)


def test_scene_reader_matches_survey(glass):
    objs, cam, _ = glass
    assert [o["name"].split("-")[0] for o in objs] == KNOWN["object_order"]
    assert [len(o["indices"]) // 3 for o in objs] == KNOWN["triangle_counts"]
    assert objs[7]["material"][14] == 0.0 and objs[5]["material"][14] == 1.0     # only `short` is glass (opacity 0)


def test_lbvh_topology_matches_reference_build(glass):
    _, _, osc = glass
    assert as_pairs(osc.accel(-1)[0]) == KNOWN["tlas_nodes"]
    assert as_pairs(osc.accel(5)[0]) == KNOWN["tall_nodes"]
    for wall in (0, 1, 2, 3, 4, 6):
        assert as_pairs(osc.accel(wall)[0]) == KNOWN["wall_nodes"]


def test_keys_are_sorted_and_carry_the_primitive(glass):
    objs, _, osc = glass
    for i in range(len(objs)):
        nodes, _, keys = osc.accel(i)
        n = len(keys)
        assert np.all(keys[1:] > keys[:-1])
        assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), nodes[n - 1:, 0])
        assert sorted(nodes[n - 1:, 0].tolist()) == list(range(n))


def test_flat_axis_gives_1023(glass):
    # every Cornell wall is flat in one axis: 0/0 = NaN and fmin(NaN, 1023) = 1023 (accel_build.cu:31-37)
    L = oracle()
    box = np.array([-1, 0, -1, 1, 1, 0, 1, 1], np.float32)
    merged = np.array([-1, 0, -1, 1, 1, 0, 1, 1], np.float32)
    key = L.orc_morton_key(box, merged, 3)
    code = key >> 32
    ybits = sum(((code >> (3 * b + 1)) & 1) << b for b in range(10))
    assert ybits == 1023 and (key & 0xFFFFFFFF) == 3


def test_primary_misses_match_survey(glass):
    _, _, osc = glass
    inst, prim, t, uv, cnt = osc.first_hit(osc.camera_rays(512, 512, 1))
    assert int((inst == 0xFFFFFFFF).sum()) == KNOWN["primary_misses_512x512_no_contraction"]
    assert cnt[3] <= 32      # the reference's stack[32] is enough here


def test_render_statistics_match_survey(glass):
    _, _, osc = glass
    img, _ = osc.render(64, 64, 1, 17, 5)
    mean = img[..., :3].mean(axis=(0, 1))
    # the survey's run drew some random pairs in the opposite order (g++ vs clang argument order): statistical pin only
    assert np.allclose(mean, KNOWN["mean_rgb_64x64_16spp_depth5"], rtol=0.03), mean
    assert np.all(img[..., 3] == 1.0)


def test_traversal_golden():
    g = np.load(os.path.join(GOLDEN, "ref_traversal.npz"))
    for glass_, tag in ((False, "box"), (True, "glass")):
        objs, cam = cornell(glass=glass_)
        osc = OracleScene(objs, cam)
        rays = np.concatenate([osc.camera_rays(96, 96, 1), osc.camera_rays(96, 96, 5), g[f"{tag}_rand_rays"]])
        inst, prim, t, uv, _ = osc.first_hit(rays)
        assert np.array_equal(inst.astype(np.int64).astype(np.int16), g[f"{tag}_inst"])
        assert np.array_equal(prim.astype(np.int64).astype(np.int16), g[f"{tag}_prim"])
        assert np.array_equal(t.view(np.uint32), g[f"{tag}_t"].view(np.uint32))
        assert np.array_equal(uv, g[f"{tag}_uv"])
        sh = g[f"{tag}_rand_rays"].copy()
        sh[:, 7] = 0.9
        assert np.array_equal(osc.occlude(sh), g[f"{tag}_occluded_tmax0.9"])
        L = oracle()
        q = g[f"{tag}_vertex_q"]
        for mi, mesh in enumerate((5, 3)):
            for s in (0, 1):
                for k in range(64):
                    o9 = np.zeros(9, np.float32)
                    a, b = (q[k, 0] * 0.5, q[k, 1] * 0.5) if s == 0 else (q[k, 0], q[k, 1])
                    L.orc_mesh_vertex(osc.h, mesh, s, k % (len(objs[mesh]["indices"]) // 3), a, b, o9)
                    assert np.array_equal(o9.view(np.uint32), g[f"{tag}_vertex"][mi, s, k].view(np.uint32))
    objs, cam = cornell(glass=False)
    osc = OracleScene(objs, cam)
    assert np.array_equal(osc.camera_rays(24, 24, 1).view(np.uint32), g["camera_rays_24x24"].view(np.uint32))


def test_object_transform_golden():
    """translate/rotate/scale composition and matrix inverse against the reference's glm (vectors
    from make_golden.py), and traversal through transformed instances."""
    g = np.load(os.path.join(GOLDEN, "ref_transform.npz"))
    L = oracle()
    for k, row in enumerate(g["trs"]):
        m, inv = trs_matrix(L, "orc", (row[0:3], row[3:6], row[6:9]))
        assert np.array_equal(m.view(np.uint32), g["matrices"][k, 0].view(np.uint32)), k
        assert np.array_equal(inv.view(np.uint32), g["matrices"][k, 1].view(np.uint32)), k
    objs, cam = cornell_moved(glass=True)
    osc = OracleScene(objs, cam)
    rays = g["moved_rays"]
    assert np.array_equal(osc.camera_rays(96, 96, 1).view(np.uint32), rays[:96 * 96].view(np.uint32))
    inst, prim, t, uv, _ = osc.first_hit(rays)
    assert np.array_equal(inst.astype(np.int64).astype(np.int16), g["moved_inst"])
    assert np.array_equal(prim.astype(np.int64).astype(np.int16), g["moved_prim"])
    assert np.array_equal(t.view(np.uint32), g["moved_t"].view(np.uint32))
    assert np.array_equal(uv, g["moved_uv"])
    sh = rays[96 * 96:].copy()
    sh[:, 7] = 0.9
    assert np.array_equal(osc.occlude(sh), g["moved_occluded"])


def test_sampler_golden():
    g = np.load(os.path.join(GOLDEN, "ref_sampler.npz"))
    L = oracle()
    for i, x in enumerate(g["xs"]):
        for j, y in enumerate(g["ys"]):
            assert L.orc_tea(int(x), int(y)) == g["tea"][i, j]
        o = np.zeros(32, np.float32)
        L.orc_next1d_seq(int(g["tea"][i, 0]), 32, o)
        assert np.array_equal(o, g["next1d"][i])
    assert np.all(g["next1d"] >= 0.0) and np.all(g["next1d"] <= 1.0)


def test_shading_golden():
    g = np.load(os.path.join(GOLDEN, "ref_shading.npz"))
    L = oracle()
    wo, wi, r = g["wo"], g["wi"], g["r"]
    for ty in range(5):
        p = np.ascontiguousarray(g["params"][ty])
        for k in range(len(wo)):
            o9 = np.zeros(9, np.float32)
            L.orc_bsdf_sample(ty, p, wo[k], r[k, 0], r[k, 1], r[k, 2], o9)
            # transcendental functions come from the host libm: allow 4 ulp against vectors made elsewhere
            assert np.allclose(o9, g["sample"][ty, k], rtol=5e-6, atol=1e-7, equal_nan=True), (ty, k)
            o4 = np.zeros(4, np.float32)
            L.orc_bsdf_eval(ty, p, wo[k], wi[k], o4)
            assert np.allclose(o4, g["evalpdf"][ty, k], rtol=5e-6, atol=1e-7, equal_nan=True), (ty, k)
    for k in range(len(wo)):
        o9 = np.zeros(9, np.float32)
        L.orc_frame(wo[k], o9)
        assert np.array_equal(o9.view(np.uint32), g["frames"][k].view(np.uint32))
        assert L.orc_luminance(r[k, 0], r[k, 1], r[k, 2]) == g["lum"][k]
    for k in range(len(g["slab_hit"])):
        assert L.orc_bbox_intersect(g["slab_bmin"][k], g["slab_bmax"][k], g["slab_rays"][k]) == g["slab_hit"][k]


def test_material_to_bsdf_rules():
    L = oracle()
    out = np.zeros(18, np.float32)
    glass_mtl = np.array([0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 324, 1.5, 0.0, 2], np.float32)
    L.orc_material_bsdf(glass_mtl, out)
    assert int(out[0]) == 4 and np.all(out[1:7] == 1.0)              # glass, reflectance := transmittance
    glass_mtl[15] = 3
    L.orc_material_bsdf(glass_mtl, out)
    assert int(out[0]) == 3                                          # microfacet keeps opacity 0 as a parameter
    white = np.array([0, 0, 0, 0.725, 0.71, 0.68, 0, 0, 0, 0, 0, 0, 323.999994, 1.5, 1.0, 2], np.float32)
    L.orc_material_bsdf(white, out)
    n = np.float32(323.999994)
    assert int(out[0]) == 2 and out[10] == n
    assert np.isclose(out[11], (n + 2) / (4 * np.pi * (2 - 2.0 ** (-n / 2))), rtol=1e-6)


def test_oracle_equals_reference_live():
    # looked for when the test runs, not when it is collected: a run that deselects this test never maps the library
    R = ref_lib()
    if R is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    L = oracle()
    sizes = np.zeros(8, np.uint32)
    R.ref_sizes(sizes)
    assert [int(x) for x in sizes[:4]] == [KNOWN["struct_sizes"][k] for k in ("Bbox", "AccelNode", "Ray", "Bsdf")]
    assert int(sizes[5]) == KNOWN["struct_sizes"]["Vertex"] and int(sizes[6]) == KNOWN["struct_sizes"]["TriMesh"]
    # a synthetic two-object scene the fixtures do not cover: heightfield + light, 5k triangles
    objs, cam = synthetic_scene("heightfield", 5000)
    osc = OracleScene(objs, cam)
    h = R.ref_scene_create()
    for i, o in enumerate(objs):
        nodes, boxes, _ = osc.accel(i)
        R.ref_scene_add_mesh(h, o["positions"].reshape(-1), o["normals"].reshape(-1), o["texcoords"].reshape(-1), len(o["positions"]),
                             o["indices"], len(o["indices"]) // 3, nodes.reshape(-1), boxes.reshape(-1))
        R.ref_scene_add_instance(h, i, IDENTITY, IDENTITY)
    nodes, boxes, _ = osc.accel(-1)
    R.ref_scene_set_top(h, nodes.reshape(-1), boxes.reshape(-1))
    rays = osc.camera_rays(128, 128, 3)
    n = len(rays)
    inst = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); uv = np.zeros((n, 2), np.float32)
    R.ref_first_hit(h, rays, n, inst, prim, t.ctypes.data, uv.ctypes.data)
    o_inst, o_prim, o_t, o_uv, _ = osc.first_hit(rays)
    assert np.array_equal(inst, o_inst) and np.array_equal(prim, o_prim)
    assert np.array_equal(t.view(np.uint32), o_t.view(np.uint32)) and np.array_equal(uv, o_uv)
    assert (inst != 0xFFFFFFFF).mean() > 0.3
    R.ref_scene_destroy(h)
