"""Synthetic-scene numbers (the 1M / 10M triangle configurations): LBVH build through the tracer, primary-ray
first-hit rate and path-tracing rate.  usage: python scripts/bench_synth.py <kind> <ntris> [width height spp depth integrators]
integrators: comma list of mode[:wavefront tail], e.g. 0,1,2:-1,2:3; traversal 0 exact (default) or 1 fast; samples per launch (default 32)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_cpt
from helpers import synthetic_scene
cpt = load_cpt()
kind = sys.argv[1] if len(sys.argv) > 1 else "heightfield"
ntris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 8
depth = int(sys.argv[6]) if len(sys.argv) > 6 else 5
modes = sys.argv[7] if len(sys.argv) > 7 else "0"
traversal = int(sys.argv[8]) if len(sys.argv) > 8 else 0
per_launch = int(sys.argv[9]) if len(sys.argv) > 9 else 32
node_records = int(sys.argv[10]) if len(sys.argv) > 10 else 1
t0 = time.time()
objs, cam = synthetic_scene(kind, ntris)
# a camera low over the terrain (inside the cube of the soup) so that nearly every 16:9 ray hits
cam["pos"] = np.array([0.5, 0.3, 0.1], dtype=np.float32)
cam["look_at"] = np.array([0.5, 0.0, 0.45], dtype=np.float32)
cam["fov"] = 40.0
sc = cpt.Scene()
for o in objs:
    sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
sc.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
t1 = time.time()
tr = cpt.Tracer(sc, W, H, max_depth=depth)
tr.set_traversal(traversal)
tr.set_node_records(node_records)
tr.set_samples_per_launch(per_launch)
tr.build_buffers(profile=True)
t2 = time.time()
print(f"{kind} {len(objs[0]['indices'])//3} tris: generate {t1-t0:.1f}s, build_buffers {t2-t1:.2f}s, stages {tr.build_stage_ms()}", flush=True)
d_rays = cpt.DeviceArray((W * H, 8), np.float32)
cpt.check(cpt.lib.cpt_tracer_camera_rays(tr.h, W, H, 1, d_rays.ptr))
n = W * H
d_inst, d_prim, d_t = cpt.DeviceArray((n,), np.uint32), cpt.DeviceArray((n,), np.uint32), cpt.DeviceArray((n,), np.float32)
for rep in range(3):
    tr.synchronize(); tr.reset_counters(); tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        cpt.check(cpt.lib.cpt_tracer_first_hit(tr.h, d_rays.ptr, n, d_inst.ptr, d_prim.ptr, d_t.ptr, None))
    tr.synchronize()
    dt = (time.perf_counter() - t0) / 10
c = tr.counters()
hit = (d_inst.to_host() != 0xFFFFFFFF).mean()
bytes_per = (48 * c["rays"] + 64 * c["node_visits"] + 48 * c["tri_tests"]) / 10
print(f"primary rays {W}x{H}: {dt*1e3:.3f} ms, {n/dt/1e6:.1f} Mrays/s, hit rate {hit:.2f}, nodes/ray {c['node_visits']/c['rays']:.1f}, tris/ray {c['tri_tests']/c['rays']:.2f}, "
      f"algorithmic {bytes_per/dt/1e9:.0f} GB/s", flush=True)
for spec in modes.split(","):
    mode, _, tail = spec.partition(":")
    tr.set_integrator(int(mode))
    if tail:
        tr.set_wavefront_tail(int(tail))
    tr.reset()
    tr.update(spp); tr.synchronize(); tr.reset_counters()      # warm-up with the launch shape of the timed run
    tr.reset()
    ms = tr.update_timed(spp)
    c = tr.counters()
    print(f"path tracing depth {depth} integrator {spec}: {ms/spp:.3f} ms/spp, {W*H*spp/ms/1e3:.1f} Mpaths/s, {c['rays']/ms/1e3:.1f} Mrays/s, rays/path {c['rays']/c['paths']:.2f}, "
          f"nodes/ray {c['node_visits']/c['rays']:.1f}", flush=True)
