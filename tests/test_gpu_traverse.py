"""GPU: first-hit instance/primitive ids, hit distances and barycentrics of the packed-record
traversal are bit-exact against the oracle (and against vectors produced by the reference's own
traversal code) for identical rays; any-hit queries agree exactly."""
import os

import numpy as np
import pytest

from helpers import *  # noqa: F403

pytestmark = pytest.mark.gpu


def gpu_scene(cpt, objects, cam, width=64, height=64, max_depth=5, node_records=1):
    sc = cpt.Scene()
    for o in objects:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
        if "trs" in o:
            sc.set_trs(sc.num_objects() - 1, *o["trs"])
    sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], width, height)
    tr = cpt.Tracer(sc, width, height, max_depth=max_depth)
    tr.set_node_records(node_records)
    return sc, tr.build_buffers()


def interior_rays(n, seed, lo, hi, tmax=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for k in range(min(n, 96)):
        d[k, k % 3] = 0.0 if k % 4 else -0.0
        if k % 2:
            d[k, (k + 1) % 3] = 0.0
        if not d[k].any():
            d[k, 2] = 1.0
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o
    rays[:, 3] = 1e-4
    rays[:, 4:7] = d
    rays[:, 7] = np.finfo(np.float32).max if tmax is None else tmax
    return rays


def assert_same_hits(tr, osc, rays):
    o_inst, o_prim, o_t, o_uv, cnt = osc.first_hit(rays)
    g_inst, g_prim, g_t, g_uv = tr.first_hit(rays)
    assert np.array_equal(g_inst, o_inst), f"{int((g_inst != o_inst).sum())} instance ids differ"
    assert np.array_equal(g_prim, o_prim), f"{int((g_prim != o_prim).sum())} primitive ids differ"
    assert np.array_equal(g_t.view(np.uint32), o_t.view(np.uint32)), "hit distances differ bitwise"
    assert np.array_equal(g_uv, o_uv), "barycentrics differ"
    return o_inst, cnt


@pytest.mark.parametrize("glass", [False, True])
def test_cornell_camera_and_interior_rays(cpt, glass):
    objs, cam = cornell(glass=glass)
    osc = OracleScene(objs, cam)
    d = os.path.join(SCENES, "cornell-box-glass" if glass else "cornell-box")
    sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
    tr = cpt.Tracer(sc, 64, 64, max_depth=5).build_buffers()
    for spp in (1, 2, 9):
        rays = osc.camera_rays(512, 512, spp)
        g_rays = tr.camera_rays(512, 512, spp)
        assert np.array_equal(g_rays.view(np.uint32), rays.view(np.uint32)), "camera rays differ bitwise"
        inst, _ = assert_same_hits(tr, osc, rays)
        if spp == 1:
            assert int((inst == 0xFFFFFFFF).sum()) == 14       # rays through edge cracks, as the reference shows
    rays = interior_rays(200000, 5, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))
    assert_same_hits(tr, osc, rays)
    # rays that start exactly on surfaces (secondary-ray situation) and short rays
    rays[:50000, 1] = 0.0
    rays[50000:100000, 0] = -1.0
    rays[100000:, 7] = 0.75
    assert_same_hits(tr, osc, rays)
    sh = interior_rays(100000, 6, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32), tmax=0.9)
    assert np.array_equal(tr.occlude(sh), osc.occlude(sh))
    # origins with -0 and +0 coordinates: every instance of this scene has the identity transform, whose product
    # with the origin turns a -0 into +0 or keeps it depending on the other terms (accel.cuh:54-59)
    rays = interior_rays(60000, 7, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))
    for k in range(3):
        rays[k * 10000:(k + 1) * 10000, k] = -0.0
        rays[30000 + k * 10000:30000 + (k + 1) * 10000, k] = 0.0
    rays[5000:10000, 1] = -0.0
    rays[15000:20000, 2] = -0.0
    assert_same_hits(tr, osc, rays)
    sh = rays.copy()
    sh[:, 7] = 0.9
    assert np.array_equal(tr.occlude(sh), osc.occlude(sh))


def test_transformed_instances(cpt):
    """Object transforms (rotation about three axes, non-uniform scale, translation): the top-level
    boxes, TransformRay without renormalisation (accel.cuh:54-59) and hits inside the moved
    instances, against the oracle and against vectors from the reference's own traversal + glm."""
    objs, cam = cornell_moved(glass=True)
    osc = OracleScene(objs, cam)
    sc, tr = gpu_scene(cpt, objs, cam, 64, 64)
    for i, o in enumerate(objs):
        m, inv = sc.object_transforms(i)
        om, oinv = object_transform(o)
        assert np.array_equal(m.view(np.uint32), om.view(np.uint32)) and np.array_equal(inv.view(np.uint32), oinv.view(np.uint32))
    nodes, boxes, keys = tr.accel(-1)
    o_nodes, o_boxes, o_keys = osc.accel(-1)
    assert np.array_equal(nodes, o_nodes) and np.array_equal(keys, o_keys)
    assert np.array_equal(boxes.view(np.uint32), o_boxes.view(np.uint32))
    g = np.load(os.path.join(GOLDEN, "ref_transform.npz"))
    rays = g["moved_rays"]
    inst, prim, t, uv = tr.first_hit(rays)
    assert np.array_equal(inst.astype(np.int64).astype(np.int16), g["moved_inst"])
    assert np.array_equal(prim.astype(np.int64).astype(np.int16), g["moved_prim"])
    assert np.array_equal(t.view(np.uint32), g["moved_t"].view(np.uint32))
    assert np.array_equal(uv, g["moved_uv"])
    sh = rays[96 * 96:].copy()
    sh[:, 7] = 0.9
    assert np.array_equal(tr.occlude(sh), g["moved_occluded"])
    big = interior_rays(300000, 9, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))
    o_inst, _ = assert_same_hits(tr, osc, big)
    assert (o_inst == 5).any() and (o_inst == 7).any()
    assert_same_hits(tr, osc, osc.camera_rays(512, 512, 3))


def test_against_reference_vectors(cpt):
    g = np.load(os.path.join(GOLDEN, "ref_traversal.npz"))
    for glass, tag in ((False, "box"), (True, "glass")):
        d = os.path.join(SCENES, "cornell-box-glass" if glass else "cornell-box")
        sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
        tr = cpt.Tracer(sc, 32, 32).build_buffers()
        rays = np.concatenate([tr.camera_rays(96, 96, 1), tr.camera_rays(96, 96, 5), g[f"{tag}_rand_rays"]])
        inst, prim, t, uv = tr.first_hit(rays)
        assert np.array_equal(inst.astype(np.int64).astype(np.int16), g[f"{tag}_inst"])
        assert np.array_equal(prim.astype(np.int64).astype(np.int16), g[f"{tag}_prim"])
        assert np.array_equal(t.view(np.uint32), g[f"{tag}_t"].view(np.uint32))
        assert np.array_equal(uv, g[f"{tag}_uv"])
        sh = g[f"{tag}_rand_rays"].copy()
        sh[:, 7] = 0.9
        assert np.array_equal(tr.occlude(sh), g[f"{tag}_occluded_tmax0.9"])
    assert np.array_equal(cpt.Tracer(sc, 8, 8).build_buffers().camera_rays(24, 24, 1).view(np.uint32)[:0], np.zeros((0, 8), np.uint32))


@pytest.mark.parametrize("kind,ntris", [("soup", 200000), ("heightfield", 180000), ("soup", 1)])
def test_synthetic_scenes(cpt, kind, ntris):
    objs, cam = synthetic_scene(kind, ntris)
    osc = OracleScene(objs, cam)
    sc, tr = gpu_scene(cpt, objs, cam)
    for i in (-1, 0, 1):
        gn, gb, gk = tr.accel(i)
        on, ob, ok = osc.accel(i)
        assert np.array_equal(gn, on) and np.array_equal(gk, ok) and np.array_equal(gb.view(np.uint32), ob.view(np.uint32))
    rays = osc.camera_rays(384, 384, 3)
    inst, cnt = assert_same_hits(tr, osc, rays)
    if ntris > 1:
        assert (inst != 0xFFFFFFFF).mean() > 0.05
    rays = interior_rays(100000, 8, np.array([0, -0.2, 0], np.float32), np.array([1, 1.5, 1], np.float32))
    assert_same_hits(tr, osc, rays)
    sh = interior_rays(50000, 9, np.array([0, -0.2, 0], np.float32), np.array([1, 1.5, 1], np.float32), tmax=0.5)
    assert np.array_equal(tr.occlude(sh), osc.occlude(sh))


def test_coincident_triangles_report_the_reference_winner(cpt):
    """Every triangle of the mesh sixteen times over: sixteen hits at exactly the same distance on every ray that hits, of
    which the reference keeps the first its visit order meets (strict less-than, trimesh.cuh:35), and long runs of nodes
    with equal boxes in the tree.  The records' slots, whichever cut they hold, must keep that order."""
    objs, cam = synthetic_scene("heightfield", 3000)
    mesh = objs[0]
    mesh["indices"] = np.tile(mesh["indices"].reshape(-1, 3), (16, 1)).reshape(-1).copy()
    osc = OracleScene(objs, cam)
    for records in (1, 0):
        sc, tr = gpu_scene(cpt, objs, cam, node_records=records)
        inst, _ = assert_same_hits(tr, osc, osc.camera_rays(256, 256, 2))
        assert (inst != 0xFFFFFFFF).mean() > 0.3
        assert_same_hits(tr, osc, interior_rays(60000, 21, np.array([0, -0.2, 0], np.float32), np.array([1, 1.5, 1], np.float32)))
        assert tr.stack_overflows() == 0


def test_counters_and_edge_cases(cpt):
    objs, cam = cornell(glass=False)
    osc = OracleScene(objs, cam)
    sc, tr = gpu_scene(cpt, objs, cam)
    tr.reset_counters()
    rays = osc.camera_rays(128, 128, 1)
    _, cnt = assert_same_hits(tr, osc, rays)
    c = tr.counters()
    assert c["rays"] == len(rays) == int(cnt[0])
    assert c["tri_tests"] == int(cnt[2])                 # the same triangles are tested, in the same order
    assert c["paths"] == 0 and c["node_visits"] > 0
    empty = np.zeros((0, 8), np.float32)
    inst, prim, t, uv = tr.first_hit(empty)
    assert len(inst) == 0
    # NaN direction and zero direction: no hit, no hang
    bad = np.zeros((2, 8), np.float32)
    bad[:, 3] = 1e-4
    bad[:, 7] = 1e30
    bad[0, 4:7] = np.nan
    assert_same_hits(tr, osc, bad)


def test_reciprocal_division_is_correctly_rounded(cpt):
    """The slab test divides through RN(1/d) plus two FMA residual steps; over 2^36 operand pairs
    from its admissible range (including all-ones / power-of-two mantissas) it must equal the
    hardware's IEEE division bit for bit."""
    import ctypes as C
    for seed in (1, 2):
        bad = C.c_uint64(123)
        cpt.check(cpt.lib.cpt_selftest_division(1 << 35, seed, C.byref(bad)))
        assert bad.value == 0


def test_scenes_outside_the_fast_division_range(cpt):
    """A box coordinate of 1e-20 (nonzero, below the 2^-50 bound under which the reciprocal
    sequence is proven exact) must switch the floor's bottom-level structure and the top-level
    structure to plain divisions; results still match the oracle bit for bit."""
    objs, cam = cornell(glass=False)
    odd = [dict(o) for o in objs]
    floor = odd[1]["positions"].copy()
    floor[:, 1] = np.float32(1e-20)
    odd[1]["positions"] = floor
    osc = OracleScene(odd, cam)
    sc, tr = gpu_scene(cpt, odd, cam)
    inst, _ = assert_same_hits(tr, osc, osc.camera_rays(384, 384, 1))
    assert (inst == 1).any()                            # the floor is seen
    assert_same_hits(tr, osc, interior_rays(100000, 21, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32)))
    # rays whose origin has a tiny nonzero component take the plain divisions as well
    rays = interior_rays(50000, 22, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))
    rays[:, 1] = np.float32(3e-19)
    sc2, tr2 = gpu_scene(cpt, objs, cam)
    assert_same_hits(tr2, OracleScene(objs, cam), rays)


@pytest.mark.parametrize("scene", ["cornell", "heightfield", "soup"])
def test_fast_traversal_against_exact(cpt, scene):
    """The ordered ("fast") traversal (nearer child first, slab distances by reciprocal multiplication) against the
    exact one on the same rays.  Stated bar: it never reports a farther hit and never misses a ray the exact traversal
    hits; where both report the same triangle, distance and barycentrics are the same bits; a different triangle is
    reported only when it is strictly nearer (coplanar or abutting triangles whose computed distances differ in the
    last place, or a crack between two boxes that the exact slab test falls into), and such rays are counted:
    at most 0.5 % of rays shot from inside the Cornell box (the two blocks stand ON the floor: their bottom faces
    coincide with it), at most 1 in 100,000 elsewhere.  Any-hit answers agree except for those rays."""
    if scene == "cornell":
        objs, cam = cornell(glass=True)
        lo, hi, bound = np.float32([-1, 0, -1]), np.float32([1, 2, 1]), 5e-3
    else:
        objs, cam = synthetic_scene(scene, 200000)
        lo, hi, bound = np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1]), 1e-5
    sc, tr = gpu_scene(cpt, objs, cam)
    for rays, limit in ((tr.camera_rays(768, 768, 2), 1e-5), (interior_rays(400000, 41, lo, hi), bound)):
        tr.set_traversal(0)
        e_inst, e_prim, e_t, e_uv = tr.first_hit(rays)
        tr.reset_counters()
        tr.set_traversal(1)
        f_inst, f_prim, f_t, f_uv = tr.first_hit(rays)
        assert tr.stack_overflows() == 0
        same = (e_inst == f_inst) & (e_prim == f_prim)
        assert np.array_equal(e_t[same].view(np.uint32), f_t[same].view(np.uint32)) and np.array_equal(e_uv[same], f_uv[same])
        assert np.all(f_t[~same] < e_t[~same]), "the fast traversal reported another triangle that is not nearer"
        assert not np.any((f_inst == 0xFFFFFFFF) & (e_inst != 0xFFFFFFFF)), "the fast traversal missed a hit"
        assert (~same).mean() <= limit, (int((~same).sum()), len(rays))
        # distances agree to the last places even where the triangle differs or a crack is closed
        both = (~same) & (e_inst != 0xFFFFFFFF)
        assert np.all(e_t[both] - f_t[both] <= 4e-6 * np.abs(e_t[both]))
        short = rays.copy()
        short[:, 7] = 0.4
        tr.set_traversal(0)
        e_occ = tr.occlude(short)
        tr.set_traversal(1)
        f_occ = tr.occlude(short)
        assert (e_occ != f_occ).mean() <= limit
    # the bounce loop with it: same image up to the paths whose hit changed
    tr.set_integrator(3)
    tr.set_traversal(0)
    tr.reset(); tr.update(8)
    exact = tr.film()
    tr.set_traversal(1)
    tr.reset(); tr.update(8)
    fast = tr.film()
    assert np.all(np.isfinite(fast)) and np.array_equal(fast[..., 3], exact[..., 3])
    changed = np.any(fast[..., :3] != exact[..., :3], axis=-1).mean()
    assert changed <= (0.2 if scene == "cornell" else 1e-3), changed
    assert np.allclose(fast[..., :3].mean(axis=(0, 1)), exact[..., :3].mean(axis=(0, 1)), rtol=0.02)


@pytest.mark.parametrize("scene", ["cornell", "moved", "heightfield", "soup", "two", "one"])
def test_two_level_records_walk_the_same_leaves(cpt, scene):
    """The two-level node records (the grandchildren of a node in one 128-byte record, tested without their parents'
    boxes) against the binary records and against the oracle: first hits, any-hit answers and the number of triangle
    tests are the same, bit for bit, on camera rays, rays from inside the scene (some parallel to an axis, with
    zeros of both signs in the direction), rays that start on box planes and short rays; only the number of node
    records fetched changes (about half)."""
    if scene in ("cornell", "moved"):
        objs, cam = (cornell if scene == "cornell" else cornell_moved)(glass=True)
        lo, hi = np.float32([-1, 0, -1]), np.float32([1, 2, 1])
    else:
        objs, cam = synthetic_scene("soup" if scene in ("two", "one") else scene, {"two": 2, "one": 1}.get(scene, 150000))
        lo, hi = np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1])
    osc = OracleScene(objs, cam)
    _, binary = gpu_scene(cpt, objs, cam, node_records=0)
    _, wide = gpu_scene(cpt, objs, cam, node_records=1)
    rays = np.concatenate([osc.camera_rays(256, 256, 2), interior_rays(150000, 51, lo, hi)])
    on_planes = interior_rays(60000, 52, lo, hi)
    on_planes[:20000, 1] = lo[1]
    on_planes[20000:40000, 0] = hi[0]
    on_planes[40000:, 7] = 0.3
    rays = np.concatenate([rays, on_planes])
    axis = interior_rays(30000, 53, lo, hi)                   # parallel to an axis: two zero components of either sign
    for k in range(len(axis)):
        d = np.zeros(3, np.float32) * (-1.0 if k & 8 else 1.0)
        d[k % 3] = 1.0 if k & 4 else -1.0
        axis[k, 4:7] = d
    rays = np.concatenate([rays, axis])
    binary.reset_counters()
    b = binary.first_hit(rays)
    wide.reset_counters()
    w = wide.first_hit(rays)
    for x, y in zip(b, w):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    cb, cw = binary.counters(), wide.counters()
    assert cb["tri_tests"] == cw["tri_tests"] and cb["rays"] == cw["rays"]
    if scene not in ("two", "one"):
        assert cw["node_visits"] < 0.62 * cb["node_visits"]
    assert wide.stack_overflows() == 0
    assert_same_hits(wide, osc, rays[:200000])
    short = rays.copy()
    short[:, 7] = 0.45
    assert np.array_equal(binary.occlude(short), wide.occlude(short))
    assert np.array_equal(wide.occlude(short[:100000]), osc.occlude(short[:100000]))
    # the ordered traversal of the batch queries walks binary records, which the scene built with two-level records
    # makes when the first such query arrives; exact results are untouched by that
    wide.set_traversal(1)
    f = wide.first_hit(rays[:50000])
    wide.set_traversal(0)
    again = wide.first_hit(rays[:50000])
    for x, y in zip(again, w):
        assert np.array_equal(x.view(np.uint32), y[:50000].view(np.uint32))
    assert (f[2] <= again[2]).all()                           # never a farther hit
    assert cpt.lib.cpt_tracer_set_node_records(wide.h, 2) < 0 and cpt.lib.cpt_tracer_set_node_records(wide.h, -1) < 0


def test_persistent_query_kernel_gives_the_same_first_hits(cpt):
    """cpt_tracer_set_query_kernel(1): lanes take the next ray of the batch when theirs is finished.  Same instance,
    primitive, distance and barycentrics bit for bit as one ray per lane, for a batch that is not a multiple of a
    segment, for one below the size at which the persistent kernel takes over, and against the oracle."""
    objs, cam = synthetic_scene("heightfield", 40000)
    sc, tr = gpu_scene(cpt, objs, cam, 64, 64)
    osc = OracleScene(objs, cam)
    lo, hi = np.float32([0, -0.2, 0]), np.float32([1, 1.2, 1])
    for n in (70001, 65536, 4099):
        rays = interior_rays(n, 11 + n, lo, hi)
        tr.set_query_kernel(0)
        plain = tr.first_hit(rays)
        tr.set_query_kernel(1)
        tr.reset_counters()
        streamed = tr.first_hit(rays)
        assert tr.counters()["rays"] == n
        for a, b in zip(plain, streamed):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        if n == 70001:
            o_inst, o_prim, o_t, o_uv, _ = osc.first_hit(rays)
            assert np.array_equal(streamed[0], o_inst) and np.array_equal(streamed[1], o_prim)
            assert np.array_equal(streamed[2].view(np.uint32), o_t.view(np.uint32)) and np.array_equal(streamed[3], o_uv)
            assert (o_inst != 0xFFFFFFFF).mean() > 0.2
    tr.set_query_kernel(0)
