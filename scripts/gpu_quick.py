"""First end-to-end check on a GPU box: LBVH parity, first-hit parity, small render vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import *

cpt = load_cpt()
print("devices", cpt.device_count(), flush=True)

def build_check(name, boxes, merged):
    n = len(boxes)
    L = oracle()
    nn = 2 * n - 1
    o_boxes = np.zeros((nn, 8), np.float32); o_boxes[n - 1:] = boxes
    o_nodes = np.zeros((nn, 2), np.uint32)
    o_keys = np.zeros(n, np.uint64); o_par = np.zeros(nn, np.uint32)
    t0 = time.time()
    L.orc_build_accel(o_nodes.reshape(-1), o_boxes.reshape(-1), merged, n, o_keys.ctypes.data, o_par.ctypes.data)
    t1 = time.time()
    g_nodes, g_boxes, g_keys, g_par, stage = cpt.build_accel(boxes, merged)
    ok_k = np.array_equal(o_keys, g_keys)
    ok_n = np.array_equal(o_nodes, g_nodes)
    ok_p = np.array_equal(o_par[1:], g_par[1:])
    ok_b = np.array_equal(o_boxes.view(np.uint32), g_boxes.view(np.uint32))
    print(f"{name}: n={n} keys={ok_k} nodes={ok_n} parents={ok_p} boxes_bitwise={ok_b} oracle {t1-t0:.2f}s gpu stages ms {stage}", flush=True)
    if not ok_b:
        bad = np.argwhere(o_boxes.view(np.uint32) != g_boxes.view(np.uint32))
        print("  first box mismatches", bad[:5], o_boxes[bad[0][0]], g_boxes[bad[0][0]])
    return ok_k and ok_n and ok_p and ok_b

ok = True
objs, cam = cornell(glass=True)
for o in objs:
    b, m = boxes_of_triangles(o["positions"], o["indices"])
    ok &= build_check(o["name"], b, m)
for n in (1, 2, 3, 7, 1000, 100000, 1000000):
    soup = random_soup(n, seed=n)
    b, m = boxes_of_triangles(soup["positions"], soup["indices"])
    ok &= build_check(f"soup{n}", b, m)
hf = heightfield(300, 300)
b, m = boxes_of_triangles(hf["positions"], hf["indices"])
ok &= build_check("heightfield", b, m)
print("BUILD PARITY", ok, flush=True)

# scene-level
d = os.path.join(SCENES, "cornell-box-glass")
sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
tr = cpt.Tracer(sc, 64, 64, max_depth=5).build_buffers()
osc = OracleScene(objs, cam)
for mesh in (-1, 0, 5, 7):
    gn, gb, gk = tr.accel(mesh)
    on, ob, ok_ = osc.accel(mesh)
    print("accel", mesh, np.array_equal(gn, on), np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), np.array_equal(gk, ok_), flush=True)
rays = osc.camera_rays(512, 512, 1)
grays = tr.camera_rays(512, 512, 1)
print("camera rays bitwise", np.array_equal(rays.view(np.uint32), grays.view(np.uint32)), flush=True)
oi, op, ot, ouv, cnt = osc.first_hit(rays)
gi, gp, gt, guv = tr.first_hit(rays)
print("first hit: inst", np.array_equal(oi, gi), "prim", np.array_equal(op, gp), "t", np.array_equal(ot, gt), "uv", np.array_equal(ouv, guv),
      "misses", int((gi == 0xffffffff).sum()), flush=True)
if not np.array_equal(op, gp):
    bad = np.flatnonzero((op != gp) | (oi != gi)); print("  mismatches", len(bad), bad[:10], oi[bad[:5]], gi[bad[:5]], op[bad[:5]], gp[bad[:5]])
rays2 = osc.camera_rays(512, 512, 7)
oi, op, ot, ouv, cnt = osc.first_hit(rays2); gi, gp, gt, guv = tr.first_hit(rays2)
print("first hit spp7: ", np.array_equal(oi, gi), np.array_equal(op, gp), np.array_equal(ot, gt), flush=True)
print("counters", tr.counters(), "oracle", cnt)

tr.update(16)
img = tr.film()
oimg, ocnt = osc.render(64, 64, 1, 17, 5)
print("gpu mean", img[..., :3].mean(axis=(0, 1)), "oracle mean", oimg[..., :3].mean(axis=(0, 1)))
diff = img[..., :3] - oimg[..., :3]
print("rel L2", np.sqrt((diff ** 2).sum() / (oimg[..., :3] ** 2).sum()), "max abs", np.abs(diff).max(), "frac pixels differing >1e-4",
      (np.abs(diff).max(axis=2) > 1e-4).mean(), flush=True)
print("counters", tr.counters(), "oracle", ocnt)

# timing: cornell glass 1024^2
sc2 = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
tr2 = cpt.Tracer(sc2, 1024, 1024, max_depth=8).build_buffers()
tr2.update(4); tr2.synchronize(); tr2.reset_counters()
t0 = time.time(); tr2.update(64); tr2.synchronize(); t1 = time.time()
c = tr2.counters()
print(f"cornell-glass 1024^2 depth 8: {64*1024*1024/(t1-t0)/1e6:.1f} Mpaths/s, {c['rays']/(t1-t0)/1e6:.1f} Mrays/s", c, flush=True)
