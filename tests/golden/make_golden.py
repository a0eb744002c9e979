"""Generates the golden vectors under tests/golden/ from the REFERENCE's own header-only device
code compiled for the host (oracle/_ref/libref.so, built by oracle/Makefile from the sources where
they lie under /root/reference).  Run in the build container only:

    make -C oracle && python tests/golden/make_golden.py

Every vector is an input/output pair of a reference function; nothing of the reference's source
text is stored.  The BVH arrays fed to the reference traversal come from the oracle's LBVH build,
whose topology on these scenes is pinned separately by survey_known_answers.json.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import *  # noqa: E402,F403

R = ref_lib()
assert R is not None, "oracle/_ref/libref.so is missing: run `make -C oracle` where /root/reference exists"
L = oracle()


def ref_scene_from(osc, objects):
    """Mirror an OracleScene into the reference structs (BVH arrays copied from the oracle)."""
    h = R.ref_scene_create()
    for i, o in enumerate(objects):
        nodes, boxes, _ = osc.accel(i)
        R.ref_scene_add_mesh(h, o["positions"].reshape(-1), o["normals"].reshape(-1), o["texcoords"].reshape(-1), len(o["positions"]),
                             o["indices"], len(o["indices"]) // 3, nodes.reshape(-1), boxes.reshape(-1))
        m, inv = trs_matrix(R, "ref", o.get("trs", IDENTITY_TRS))     # the reference's own glm builds the matrices it traverses with
        R.ref_scene_add_instance(h, i, m, inv)
    nodes, boxes, _ = osc.accel(-1)
    R.ref_scene_set_top(h, nodes.reshape(-1), boxes.reshape(-1))
    return h


def ref_first_hit(h, rays):
    n = len(rays)
    inst = np.zeros(n, np.uint32); prim = np.zeros(n, np.uint32)
    t = np.zeros(n, np.float32); uv = np.zeros((n, 2), np.float32)
    R.ref_first_hit(h, np.ascontiguousarray(rays), n, inst, prim, t.ctypes.data, uv.ctypes.data)
    return inst, prim, t, uv


def interior_rays(n, seed, lo, hi):
    """Rays starting inside the box [lo,hi] with random directions; a few axis-aligned ones."""
    rng = np.random.Generator(np.random.PCG64(seed))
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    for k in range(min(n, 48)):       # exact zeros in one or two components
        d[k, k % 3] = 0.0
        if k % 2:
            d[k, (k + 1) % 3] = 0.0
        if not d[k].any():
            d[k, 2] = 1.0
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max
    return rays


def main():
    sizes = np.zeros(8, np.uint32)
    R.ref_sizes(sizes)
    assert list(sizes[:4]) == [32, 8, 32, 64], sizes

    out = {}
    # ---- traversal + triangle test on both Cornell scenes -----------------------------------
    for glass in (False, True):
        tag = "glass" if glass else "box"
        objs, cam = cornell(glass=glass)
        osc = OracleScene(objs, cam)
        h = ref_scene_from(osc, objs)
        cam_rays = np.concatenate([osc.camera_rays(96, 96, 1), osc.camera_rays(96, 96, 5)])
        # the reference's own camera must produce the same primary rays as the oracle
        rand = interior_rays(6000, 11 + glass, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))
        rays = np.concatenate([cam_rays, rand])
        inst, prim, t, uv = ref_first_hit(h, rays)
        o_inst, o_prim, o_t, o_uv, _ = osc.first_hit(rays)
        assert np.array_equal(inst, o_inst) and np.array_equal(prim, o_prim), "oracle traversal differs from the reference"
        assert np.array_equal(t.view(np.uint32), o_t.view(np.uint32)) and np.array_equal(uv, o_uv)
        # shadow-style rays: finite tmax towards the light
        sh = rand.copy()
        sh[:, 7] = 0.9
        occ = np.zeros(len(sh), np.uint8)
        R.ref_occlude(h, sh, len(sh), occ)
        assert np.array_equal(occ, osc.occlude(sh)), "oracle any-hit differs from the reference"
        out[f"{tag}_rand_rays"] = rand
        out[f"{tag}_inst"] = inst.astype(np.int64).astype(np.int16)     # 0xffffffff -> -1
        out[f"{tag}_prim"] = prim.astype(np.int64).astype(np.int16)
        out[f"{tag}_t"] = t
        out[f"{tag}_uv"] = uv
        out[f"{tag}_occluded_tmax0.9"] = occ
        # vertex fetch / light sampling on the 12-triangle box and the light quad
        rng = np.random.Generator(np.random.PCG64(5))
        q = rng.random((64, 2)).astype(np.float32)
        verts = np.zeros((2, 2, 64, 9), np.float32)
        for mi, mesh in enumerate((5, 3)):
            for s in (0, 1):
                for k in range(64):
                    o9 = np.zeros(9, np.float32); r9 = np.zeros(9, np.float32)
                    a, b = (q[k, 0] * 0.5, q[k, 1] * 0.5) if s == 0 else (q[k, 0], q[k, 1])
                    prim_id = k % (len(objs[mesh]["indices"]) // 3)
                    R.ref_mesh_vertex(h, mesh, s, prim_id, a, b, r9)
                    L.orc_mesh_vertex(osc.h, mesh, s, prim_id, a, b, o9)
                    assert np.array_equal(o9.view(np.uint32), r9.view(np.uint32)), "oracle vertex fetch differs from the reference"
                    verts[mi, s, k] = r9
        out[f"{tag}_vertex_q"] = q
        out[f"{tag}_vertex"] = verts
        R.ref_scene_destroy(h)
    # camera: reference PinholeCamera vs the oracle's ray generator
    objs, cam = cornell(glass=False)
    osc = OracleScene(objs, cam)
    W = H = 24
    rays = osc.camera_rays(W, H, 1)
    cam18 = np.zeros(18, np.float32)
    L.orc_scene_get_camera(osc.h, cam18)
    ref_rays = np.zeros((W * H, 8), np.float32)
    for py in range(H):
        for px in range(W):
            u = np.float32(np.float32(px + np.float32(0.5)) / np.float32(W))
            v = np.float32(np.float32(py + np.float32(0.5)) / np.float32(H))
            R.ref_camera_ray(cam18, np.float32(W) / np.float32(H), u, v, ref_rays[(H - 1 - py) * W + px])
    assert np.array_equal(rays.view(np.uint32), ref_rays.view(np.uint32)), "oracle camera differs from the reference"
    out["camera18"] = cam18
    out["camera_rays_24x24"] = ref_rays
    np.savez_compressed(os.path.join(HERE, "ref_traversal.npz"), **out)

    # ---- object transforms: glm::translate/rotate/scale/inverse and traversal through them -------
    rng = np.random.Generator(np.random.PCG64(77))
    n_trs = 64
    trs = np.zeros((n_trs, 9), np.float32)
    trs[:, 0:3] = rng.uniform(-2, 2, (n_trs, 3))
    trs[:, 3:6] = rng.uniform(-180, 180, (n_trs, 3))
    trs[:, 6:9] = rng.uniform(0.25, 3.0, (n_trs, 3))
    trs[0] = [0, 0, 0, 0, 0, 0, 1, 1, 1]
    trs[1] = [1, -2, 3, 90, 0, 0, 1, 1, 1]
    trs[2] = [0, 0, 0, 0, 180, 45, 2, 2, 2]
    mats = np.zeros((n_trs, 2, 16), np.float32)
    for k in range(n_trs):
        rm, rinv = trs_matrix(R, "ref", (trs[k, 0:3], trs[k, 3:6], trs[k, 6:9]))
        om, oinv = trs_matrix(L, "orc", (trs[k, 0:3], trs[k, 3:6], trs[k, 6:9]))
        assert np.array_equal(rm.view(np.uint32), om.view(np.uint32)), "oracle object transform differs from glm"
        assert np.array_equal(rinv.view(np.uint32), oinv.view(np.uint32)), "oracle matrix inverse differs from glm"
        mats[k, 0], mats[k, 1] = rm, rinv
    tout = dict(trs=trs, matrices=mats)
    objs, cam = cornell_moved(glass=True)
    osc = OracleScene(objs, cam)
    h = ref_scene_from(osc, objs)
    rays = np.concatenate([osc.camera_rays(96, 96, 1),
                           interior_rays(6000, 21, np.array([-1, 0, -1], np.float32), np.array([1, 2, 1], np.float32))])
    inst, prim, t, uv = ref_first_hit(h, rays)
    o_inst, o_prim, o_t, o_uv, _ = osc.first_hit(rays)
    assert np.array_equal(inst, o_inst) and np.array_equal(prim, o_prim), "oracle traversal of transformed instances differs from the reference"
    assert np.array_equal(t.view(np.uint32), o_t.view(np.uint32)) and np.array_equal(uv, o_uv)
    sh = rays[96 * 96:].copy()
    sh[:, 7] = 0.9
    occ = np.zeros(len(sh), np.uint8)
    R.ref_occlude(h, sh, len(sh), occ)
    assert np.array_equal(occ, osc.occlude(sh))
    assert (inst[:96 * 96] == 5).any() and (inst[:96 * 96] == 7).any(), "camera sees both moved blocks"
    tout.update(moved_rays=rays, moved_inst=inst.astype(np.int64).astype(np.int16), moved_prim=prim.astype(np.int64).astype(np.int16),
                moved_t=t, moved_uv=uv, moved_occluded=occ)
    R.ref_scene_destroy(h)
    np.savez_compressed(os.path.join(HERE, "ref_transform.npz"), **tout)

    # ---- sampler ----------------------------------------------------------------------------
    xs = np.array([0, 1, 2, 255, 256, 65535, 262143, 1048575, 0x7fffffff, 0xffffffff], np.uint32)
    ys = np.array([1, 2, 3, 16, 64, 1024, 4096, 65536], np.uint32)
    tea = np.array([[R.ref_tea(int(x), int(y)) for y in ys] for x in xs], np.uint32)
    assert all(L.orc_tea(int(x), int(y)) == tea[i, j] for i, x in enumerate(xs) for j, y in enumerate(ys))
    seq = np.zeros((len(xs), 32), np.float32)
    for i, x in enumerate(xs):
        s = np.zeros(32, np.float32); o = np.zeros(32, np.float32)
        R.ref_next1d_seq(int(tea[i, 0]), 32, s); L.orc_next1d_seq(int(tea[i, 0]), 32, o)
        assert np.array_equal(s, o)
        seq[i] = s
    # a state whose PCG output rounds up to exactly 1.0f must exist in principle; record the extreme seen
    np.savez_compressed(os.path.join(HERE, "ref_sampler.npz"), xs=xs, ys=ys, tea=tea, next1d=seq)

    # ---- BSDFs, frame, luminance, slab test -------------------------------------------------
    rng = np.random.Generator(np.random.PCG64(2024))
    n = 200
    wo = rng.standard_normal((n, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True); wo = wo.astype(np.float32)
    wi = rng.standard_normal((n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True); wi = wi.astype(np.float32)
    r = rng.random((n, 3)).astype(np.float32)
    params = np.zeros((5, 14), np.float32)
    mats = {0: [0, 0, 0, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0, 0, 0, 40.0, 1.5, 1.0, 0],
            1: [0, 0, 0, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0, 0, 0, 40.0, 1.5, 1.0, 1],
            2: [0, 0, 0, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0, 0, 0, 40.0, 1.5, 1.0, 2],
            3: [0, 0, 0, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.7, 0.8, 0.9, 40.0, 1.5, 0.5, 3],
            4: [0, 0, 0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 0.9, 0.95, 1.0, 324.0, 1.5, 0.0, 2]}
    sample = np.zeros((5, n, 9), np.float32); evalpdf = np.zeros((5, n, 4), np.float32)
    for ty in range(5):
        b18 = np.zeros(18, np.float32)
        L.orc_material_bsdf(np.array(mats[ty], np.float32), b18)
        assert int(b18[0]) == ty
        params[ty] = b18[1:15]
        for k in range(n):
            o9 = np.zeros(9, np.float32); r9 = np.zeros(9, np.float32)
            R.ref_bsdf_sample(ty, params[ty], wo[k], r[k, 0], r[k, 1], r[k, 2], r9)
            L.orc_bsdf_sample(ty, params[ty], wo[k], r[k, 0], r[k, 1], r[k, 2], o9)
            assert np.array_equal(o9.view(np.uint32), r9.view(np.uint32)), ("bsdf sample", ty, k, o9, r9)
            sample[ty, k] = r9
            o4 = np.zeros(4, np.float32); r4 = np.zeros(4, np.float32)
            R.ref_bsdf_eval(ty, params[ty], wo[k], wi[k], r4)
            L.orc_bsdf_eval(ty, params[ty], wo[k], wi[k], o4)
            assert np.array_equal(o4.view(np.uint32), r4.view(np.uint32)), ("bsdf eval", ty, k, o4, r4)
            evalpdf[ty, k] = r4
    frames = np.zeros((n, 9), np.float32)
    lum = np.zeros(n, np.float32)
    for k in range(n):
        o9 = np.zeros(9, np.float32)
        R.ref_frame(wo[k], frames[k]); L.orc_frame(wo[k], o9)
        assert np.array_equal(o9.view(np.uint32), frames[k].view(np.uint32))
        lum[k] = R.ref_luminance(r[k, 0], r[k, 1], r[k, 2])
        assert lum[k] == L.orc_luminance(r[k, 0], r[k, 1], r[k, 2])
    # slab test incl. flat boxes and zero direction components
    nb = 400
    bmin = (rng.random((nb, 3)) * 2 - 1).astype(np.float32)
    bmax = (bmin + rng.random((nb, 3)).astype(np.float32) * (rng.random((nb, 3)) > 0.25)).astype(np.float32)
    srays = interior_rays(nb, 99, np.array([-2, -2, -2], np.float32), np.array([2, 2, 2], np.float32))
    srays[::3, 7] = 1.5
    slab = np.array([R.ref_bbox_intersect(bmin[k], bmax[k], srays[k]) for k in range(nb)], np.uint8)
    assert all(L.orc_bbox_intersect(bmin[k], bmax[k], srays[k]) == slab[k] for k in range(nb))
    np.savez_compressed(os.path.join(HERE, "ref_shading.npz"), wo=wo, wi=wi, r=r, params=params, sample=sample, evalpdf=evalpdf,
                        frames=frames, lum=lum, slab_bmin=bmin, slab_bmax=bmax, slab_rays=srays, slab_hit=slab)
    print("golden vectors written; oracle == reference on every vector")


if __name__ == "__main__":
    main()
