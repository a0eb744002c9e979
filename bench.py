#!/usr/bin/env python3
"""Benchmark of the gfx950 path-tracing core.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (the configuration the headline metric is quoted on): cornell-box-glass, 1024x1024, max
depth 8, glass (dielectric) box, 1 sample per pixel per step; the default K = 1024 steps is the full
1024-spp render.  One step is one sample of every pixel; the persistent kernels render
--samples-per-launch of them per launch, leaving the film of one launch per sample bit for bit.
With N > 1 every rank renders its own
block of K sample indices into a sum buffer (weak scaling, no data-path collective) and the timed
region ends with one sum-reduce of the 16.8 MB buffers to rank 0 (RCCL through torch.distributed).

--workload heightfield-10m (etc.) renders a synthetic scene of the named size instead (the 1M / 10M
triangle configurations), --integrator picks the bounce-loop variant; the defaults are the metric's configuration.

Prints ONE JSON line on rank 0.  Scene data comes from tests/golden/scenes (data fixtures);
nothing outside this repository is read.  The oracle (oracle/) is used only for the `cpu_baseline`
leg, never for the measured path.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402  (first: the HIP runtime torch ships must be the one the library binds to)
import torch.distributed as dist  # noqa: E402
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
from conftest import load_cpt  # noqa: E402
import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E, vendor peak


def algorithmic_bytes(c, pixels_per_launch, launches):
    """Bytes the bounce loop has to move per launch, by the traversal/bounce formulas:
    per ray 32 B in + 16 B hit out, 64 B per node record visited, 48 B per triangle record
    tested; per shaded hit 3 x (12 + 12 + 8) B of vertex attributes + 112 B of material;
    16 B read + 16 B written per pixel for the accumulation buffer."""
    total = (48 * c["rays"] + 64 * c["node_visits"] + 48 * c["tri_tests"] + (96 + 112) * c["shades"]
             + 32 * pixels_per_launch * launches)
    return total / max(launches, 1)


def lbvh_timings(cpt, n, seed=12345, repeats=10, sort_variant=0):
    """Median device time of the whole build (Morton -> sort -> nodes -> refit) with buffers
    pre-allocated, after two warm-ups; plus the per-stage split of one extra run (hierarchy = leaf
    gather + topology + boxes, two kernels)."""
    from helpers import random_soup, boxes_of_triangles
    soup = random_soup(n, seed=seed)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    del soup
    d_leaf = cpt.DeviceArray.from_host(boxes)
    d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    b = cpt.lib.cpt_builder_create()
    cpt.check(cpt.lib.cpt_builder_set_sort(b, sort_variant))
    times = []
    for i in range(repeats + 2):
        ms = C.c_float(0.0)
        cpt.check(cpt.lib.cpt_builder_build(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, C.byref(ms), None))
        if i >= 2:
            times.append(ms.value)
    stage = np.zeros(5, np.float32)
    cpt.check(cpt.lib.cpt_builder_build(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, None, stage.ctypes.data))
    cpt.lib.cpt_builder_destroy(b)
    med = float(np.median(times))
    return dict(build_ms=round(med, 4), morton_ms=round(float(stage[0]), 4), sort_ms=round(float(stage[1]), 4),
                hierarchy_ms=round(float(stage[2] + stage[3] + stage[4]), 4),
                hbm_frac=round(312.0 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                sort_hbm_frac=round(72.0 * n / (float(stage[1]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))


def cpu_baseline(width, height, max_depth, glass=True, target_seconds=12.0):
    """The oracle (CPU restatement of the reference's intersection, BVH and shading code) driven
    by a host thread pool over 16x16 tiles, on a bounded number of samples of the same workload."""
    from helpers import cornell, OracleScene
    objs, cam = cornell(glass=glass)
    osc = OracleScene(objs, cam)
    cores = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    osc.render(width, height, 1, 2, max_depth, nthreads=cores)
    one = time.perf_counter() - t0
    spp = int(max(1, min(256, target_seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    _, cnt = osc.render(width, height, 2, 2 + spp, max_depth, nthreads=cores)
    dt = time.perf_counter() - t0
    return dict(value=round(width * height * spp / dt / 1e6, 4), unit="Mpaths/s", cores=cores, kind="port",
                mrays_per_s=round(int(cnt[0]) / dt / 1e6, 4),
                sample=f"{spp} spp of the same {width}x{height} depth-{max_depth} workload ({dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--workload", default="cornell-box-glass",
                    help="cornell-box-glass (the metric's configuration, default) or a synthetic scene: heightfield-1m, heightfield-10m, soup-1m, soup-10m")
    ap.add_argument("--integrator", type=int, default=3,
                    help="3 by scene size (default: 5 for the Cornell box, 4 for the synthetic scenes), 0 persistent megakernel, 1 megakernel, "
                         "2 wavefront, 4 persistent megakernel that shades partly finished waves, 5 persistent megakernel with pooled shadow rays")
    ap.add_argument("--samples-per-launch", type=int, default=128,
                    help="samples (steps) the persistent kernels of integrators 4 and 5 render per launch; the film is bit for bit "
                         "that of one launch per sample (1 = one launch per sample, as the reference launches RenderKernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lbvh", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run the N>1 code path with every rank on GPU 0 and the reduce over gloo (functional rehearsal, not a measurement)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cpt = load_cpt()
    cpt.check(cpt.lib.cpt_set_device(local_rank))
    W, H, K, WU = args.width, args.height, args.steps, args.warmup

    if args.workload == "cornell-box-glass":
        d = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-glass")
        scene = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
        workload = f"cornell-box-glass {W}x{H}, max-depth {args.max_depth}, glass box"
        scene_triangles = 36
    else:
        from helpers import synthetic_scene
        kind, _, size = args.workload.partition("-")
        ntris = {"1m": 1_000_000, "10m": 10_000_000}.get(size)
        if kind not in ("heightfield", "soup") or ntris is None:
            raise SystemExit(f"unknown workload {args.workload}")
        objs, cam = synthetic_scene(kind, ntris)            # seeded generator: every rank builds the same scene
        scene = cpt.Scene()
        for o in objs:
            scene.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
        scene.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
        workload = f"synthetic {kind}, {len(objs[0]['indices']) // 3} triangles + emissive quad, {W}x{H}, max-depth {args.max_depth}"
        scene_triangles = sum(len(o["indices"]) // 3 for o in objs)
    tracer = cpt.Tracer(scene, W, H, max_depth=args.max_depth)
    tracer.set_integrator(args.integrator)
    tracer.set_samples_per_launch(args.samples_per_launch)
    tracer.build_buffers()
    integrator = args.integrator if args.integrator != 3 else (4 if scene_triangles >= 65536 else 5)     # what mode 3 resolves to in the library
    kernel_name = {0: "k_render_regen", 1: "k_render", 2: "k_wf_trace + k_wf_shade + k_wf_finish", 4: "k_render_flow", 5: "k_render_pool"}[integrator]
    integrator_name = {0: "persistent megakernel (lane regeneration)", 1: "megakernel, one lane per pixel", 2: "wavefront (ray queues) + persistent tail",
                       4: "persistent megakernel, traversal state kept across shading",
                       5: "persistent megakernel, finished lanes walk the wave's pooled shadow rays"}[integrator]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kernel_ms = None
    if world == 1:
        # the reference's own path: PathTracer::Update, one sample per call, running mean in the film
        tracer.update(WU)
        tracer.synchronize()
        tracer.reset_counters()
        barrier()
        t0 = time.perf_counter()
        kernel_ms = tracer.update_timed(K)      # K launches bracketed by HIP events on the launch stream
        barrier()
        elapsed = time.perf_counter() - t0
    else:
        scratch = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
        accum = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        tracer.render_samples(1, WU, True, scratch.data_ptr(), W, H)
        tracer.synchronize()
        tracer.reset_counters()
        begin, count = sharding.sample_block(rank, world, K)
        barrier()
        t0 = time.perf_counter()
        tracer.timer_begin()
        tracer.render_samples(begin, count, True, accum.data_ptr(), W, H)
        kernel_ms = tracer.timer_end()          # this rank's K launches, HIP events on the launch stream
        if args.rehearse_on_one_gpu:
            host = accum.cpu()
            sharding.reduce_sum_to_root(host, dst=0)
            accum.copy_(host)
        else:
            sharding.reduce_sum_to_root(accum, dst=0)
        barrier()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            alpha = accum[..., 3]
            if not bool(torch.all(alpha == float(K * world))):
                raise SystemExit("reduced film does not hold K x N samples per pixel")

    counters = tracer.counters()
    if world > 1:
        red_dev = torch.device("cpu") if args.rehearse_on_one_gpu else device
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([counters[k] for k in ("rays", "node_visits", "tri_tests", "paths", "shades")], dtype=torch.int64, device=red_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total = dict(zip(("rays", "node_visits", "tri_tests", "paths", "shades"), [int(x) for x in cnt.tolist()]))
    else:
        total = counters

    if rank == 0:
        paths = W * H * K * world
        result = {
            "metric": "Mpaths/s per node, cornell-box-glass 1024x1024 max-depth 8 (Mrays/s and LBVH build ms @1M/10M alongside)"
                      if args.workload == "cornell-box-glass" else f"Mpaths/s per node, {args.workload}",
            "value": round(paths / elapsed / 1e6, 3),
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": K,
            "warmup": WU,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "reference scene file cornell-box-glass (committed fixture); synthetic soups for the LBVH timings"
                    if args.workload == "cornell-box-glass" else "synthetic scene from a seeded generator (tests/helpers.py); synthetic soups for the LBVH timings",
            "config": {"workload": f"{workload}, 1 spp per step, {K} spp per GPU",
                       "samples_per_launch": args.samples_per_launch if integrator in (4, 5) else 1,
                       "integrator": integrator_name, "sharding": "replicas only" if world == 1 else f"{world} x disjoint sample blocks + 1 sum-reduce"},
            "mrays_per_s": round(total["rays"] / elapsed / 1e6, 3),
            "rays_per_path": round(total["rays"] / max(total["paths"], 1), 4),
        }
        if kernel_ms is not None:
            # rank 0's own launches: K samples in launches of samples_per_launch (the last one may be shorter)
            own = K if world == 1 else sharding.sample_block(rank, world, K)[1]
            per_launch = args.samples_per_launch if integrator in (4, 5) else 1
            launches = -(-own // per_launch)
            per_launch_bytes = algorithmic_bytes(counters, W * H, own) * own / launches
            avg_ms = kernel_ms / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and args.workload == "cornell-box-glass" and (W, H, args.max_depth) == (1024, 1024, 8):
                try:
                    measured = json.load(open(tpath))
                    # counters taken on launches of the same number of samples only
                    same = measured.get("kernel") == kernel_name and measured.get("samples_per_launch", 32) == per_launch
                    traffic = measured.get("hbm_bytes_per_launch") if same else None
                except Exception:
                    traffic = None
            result["roofline"] = {"kernel": kernel_name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                                  "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(per_launch_bytes),
                                  "samples_per_launch": per_launch, "launches": launches}
        if world == 1 and not args.no_lbvh:
            result["lbvh"] = {"1M": lbvh_timings(cpt, 1_000_000), "10M": lbvh_timings(cpt, 10_000_000)}
            # what a kernel that only streams reaches on this device (1 GiB copied, read + written bytes counted),
            # and the 10M build's 312 bytes per primitive against that instead of the nominal peak
            rate = C.c_double(0.0)
            cpt.check(cpt.lib.cpt_stream_copy_rate(1 << 30, 10, C.byref(rate)))
            result["lbvh"]["stream_copy_gb_per_s"] = round(rate.value, 1)
            med = result["lbvh"]["10M"]["build_ms"]
            result["lbvh"]["10M"]["frac_of_stream_copy"] = round(312.0 * 10_000_000 / (med * 1e-3) / 1e9 / rate.value, 4)
        if world == 1 and not args.no_cpu_baseline and args.workload == "cornell-box-glass":
            result["cpu_baseline"] = cpu_baseline(W, H, args.max_depth)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
