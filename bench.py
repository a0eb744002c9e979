#!/usr/bin/env python3
"""Benchmark of the gfx950 path-tracing core.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline workload (the configuration the metric is quoted on): cornell-box-glass, 1024x1024, max depth 8, glass
(dielectric) box, exact traversal.  One step is one sample of every pixel; the default K = 1024 steps is the full
1024-spp render.  The persistent kernels render --samples-per-launch of them per launch, leaving the film of one
launch per sample bit for bit.

N > 1 (one process per GPU): every rank renders its own block of K sample indices as sums (weak scaling, no data-path
collective) and the timed region ends with ONE ncclReduce of the films to rank 0 and the division by the sample
count, both issued by the library (cpt_tracer_render_samples + cpt_tracer_reduce_film, RCCL called from C++); torch is
used only to hand the 128-byte communicator id to the ranks, for the barriers around the timed region and to gather
the ranks' clocks and counters afterwards.

Alongside, in the same JSON line (blocks that do not change `value`):
  primary_rays   1M / 10M-triangle synthetic scenes, camera rays: Mrays/s, records per ray, exact and fast traversal
  scene_10m      the 10M-triangle 3840x2160 configuration: a fixed number of samples split over the N ranks (strong
                 scaling), reduce included; megakernel and wavefront integrators, depth 5 and unbounded
  lbvh           build times at 1M and 10M primitives
  cpu_baseline   the oracle on the host cores: the headline workload (bounded sample), configuration 1 in full and
                 the 1M build + primary rays

Prints ONE JSON line on rank 0.  Scene data comes from tests/golden/scenes (data fixtures) and seeded generators;
nothing outside this repository is read.  The oracle (oracle/) is used only for the `cpu_baseline` legs.
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

HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E, vendor peak
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4    # wave64 VALU instructions per ns: 256 CUs x 4 SIMDs x 2.4 GHz, four cycles each on a 16-lane SIMD
COUNTER_FILE = os.path.join(ROOT, "profiles", "r02_counters.json")
KERNEL_OF = {0: "k_render_regen", 1: "k_render", 2: "k_wf_trace + k_wf_shade + k_wf_finish", 4: "k_render_flow", 5: "k_render_pool"}
INTEGRATOR_NAME = {0: "persistent megakernel (lane regeneration)", 1: "megakernel, one lane per pixel", 2: "wavefront (ray queues) + persistent tail",
                   4: "persistent megakernel, traversal state kept across shading",
                   5: "persistent megakernel, finished lanes walk the wave's pooled shadow rays"}
COUNTER_KEYS = ("rays", "node_visits", "tri_tests", "paths", "shades")


def node_record_bytes(exact):
    """The exact traversal walks two-level records of 128 B, the ordered one binary records of 64 B."""
    return 128 if exact else 64


def traversal_bytes(c, node_bytes):
    """Per ray 32 B in + 16 B hit out, node_bytes per node record visited, 48 B per triangle record tested."""
    return 48 * c["rays"] + node_bytes * c["node_visits"] + 48 * c["tri_tests"]


def bounce_bytes(c, pixel_samples, node_bytes):
    """Traversal bytes + per shaded hit 3 x (12 + 12 + 8) B of vertex attributes and 112 B of material + 16 B read and
    16 B written per pixel and sample for the accumulation buffer."""
    return traversal_bytes(c, node_bytes) + (96 + 112) * c["shades"] + 32 * pixel_samples


def measured_counters(key):
    """Per-sample hardware counter values of one kernel on one workload, from the rocprofv3 --pmc passes kept under
    profiles/ (scripts/profile_round2.sh writes the file); None when there is no entry."""
    try:
        return json.load(open(COUNTER_FILE)).get(key)
    except Exception:
        return None


def roofline_block(kernel, algorithmic_bytes, avg_ms, samples, m):
    """Roofline object of one kernel launch.  Live: the launch duration (HIP events) and the algorithmic bytes (work
    counters of this run).  From the counter passes kept under profiles/ (per sample, scaled to the samples this launch
    held): HBM bytes and VALU instructions.  These traversal kernels read most of their algorithmic bytes from the caches, so
    the binding unit is whichever of the two is busier: VALU issue (a wave64 instruction holds a 16-lane SIMD for four
    cycles) or HBM; without counters only the nominal HBM figure can be given."""
    alg_gbs = algorithmic_bytes / (avg_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "avg_launch_ms": round(avg_ms, 5), "samples_per_launch": round(samples, 2), "algorithmic_bytes_per_launch": int(algorithmic_bytes),
         "algorithmic_gb_per_s": round(alg_gbs, 1), "algorithmic_frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4)}
    if not m:
        r.update(bound="hbm", achieved=round(alg_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_gbs / HBM_PEAK_GBS, 4), traffic=None,
                 note="no counter pass for this launch shape: nominal figure, algorithmic bytes against the HBM peak")
        return r
    traffic = m["hbm_bytes_per_sample"] * samples
    hbm_gbs = traffic / (avg_ms * 1e-3) / 1e9
    ginst = m["valu_insts_per_sample"] * samples / (avg_ms * 1e-3) / 1e9
    if ginst / VALU_PEAK_GINST >= hbm_gbs / HBM_PEAK_GBS:
        r.update(bound="valu", achieved=round(ginst, 1), peak=round(VALU_PEAK_GINST, 1), unit="G wave64 VALU instructions/s", frac=round(ginst / VALU_PEAK_GINST, 4))
    else:
        r.update(bound="hbm", achieved=round(hbm_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(hbm_gbs / HBM_PEAK_GBS, 4))
    r.update(traffic=int(traffic), hbm_gb_per_s=round(hbm_gbs, 1), hbm_frac=round(hbm_gbs / HBM_PEAK_GBS, 4), valu_frac=round(ginst / VALU_PEAK_GINST, 4),
             counters={k: m[k] for k in ("valu_busy", "lanes_active", "l1_hit", "l2_hit", "salu_per_valu", "lds_bank_conflict_frac") if k in m})
    return r


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def lbvh_timings(cpt, n, seed=12345, repeats=10, sort_variant=0):
    """Median device time of the whole build (Morton -> sort -> nodes -> refit) with buffers pre-allocated, after
    two warm-ups; plus the per-stage split of one extra run (hierarchy = leaf gather + topology + boxes)."""
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


# --------------------------------------------------------------------------- CPU legs (oracle = checker, timed as the baseline)
def cpu_render_baseline(width, height, max_depth, glass, bsdf_type, cores, target_seconds=None, spp_exact=None):
    from helpers import cornell, OracleScene
    objs, cam = cornell(glass=glass, bsdf_type=bsdf_type)
    osc = OracleScene(objs, cam)
    if spp_exact is None:
        t0 = time.perf_counter()
        osc.render(width, height, 1, 2, max_depth, nthreads=cores)
        one = time.perf_counter() - t0
        spp = int(max(1, min(256, target_seconds / max(one, 1e-3))))
        first = 2
    else:
        spp, first = spp_exact, 1
    t0 = time.perf_counter()
    _, cnt = osc.render(width, height, first, first + spp, max_depth, nthreads=cores)
    dt = time.perf_counter() - t0
    return dict(value=round(width * height * spp / dt / 1e6, 4), unit="Mpaths/s", cores=cores, kind="port",
                mrays_per_s=round(int(cnt[0]) / dt / 1e6, 4), seconds=round(dt, 2), spp=spp)


def cpu_baselines(W, H, max_depth):
    from helpers import synthetic_scene, OracleScene
    cores = min(16, os.cpu_count() or 1)
    head = cpu_render_baseline(W, H, max_depth, True, 2, cores, target_seconds=10.0)
    head["sample"] = f"{head['spp']} spp of the same {W}x{H} depth-{max_depth} workload ({head['seconds']} s)"
    head["cpu"] = cpu_model()
    others = {}
    c1 = cpu_render_baseline(512, 512, 5, False, 2, cores, spp_exact=64)
    c1["sample"] = "configuration 1 in full: cornell-box 512x512, 64 spp, max-depth 5, blinn-phong"
    others["cornell-box 512x512 64 spp depth 5"] = c1
    objs, cam = synthetic_scene("heightfield", 1_000_000)
    t0 = time.perf_counter()
    osc = OracleScene(objs, cam)                       # uploads + bottom-level and top-level LBVH builds, one thread
    build_s = time.perf_counter() - t0
    osc.L.orc_scene_set_camera(osc.h, np.float32([0.5, 0.3, 0.1]), np.float32([0.5, 0.0, 0.45]), np.float32([0, 1, 0]), 40.0)
    rays = osc.camera_rays(1920, 1080, 1)
    t0 = time.perf_counter()
    _, _, _, _, cnt = osc.first_hit(rays, nthreads=cores)
    dt = time.perf_counter() - t0
    others["heightfield 1M: LBVH build + primary rays 1920x1080"] = dict(
        lbvh_build_ms=round(build_s * 1e3, 1), build_threads=1, mrays_per_s=round(len(rays) / dt / 1e6, 3), cores=cores, kind="port",
        node_visits_per_ray=round(int(cnt[1]) / max(int(cnt[0]), 1), 2),
        sample="configuration 3: scene upload + LBVH build of 1M triangles (single thread), 2,073,600 camera rays on all cores")
    return head, others


# --------------------------------------------------------------------------- synthetic scenes
def synthetic(cpt, kind, ntris, W, H):
    from helpers import synthetic_scene
    objs, cam = synthetic_scene(kind, ntris)            # seeded generator: every rank builds the same scene
    scene = cpt.Scene()
    for o in objs:
        scene.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    scene.set_camera([0.5, 0.3, 0.1], [0.5, 0.0, 0.45], [0, 1, 0], 40.0, W, H)
    return scene, sum(len(o["indices"]) // 3 for o in objs)


def primary_ray_block(cpt, tracer, W, H, label):
    n = W * H
    d_rays = cpt.DeviceArray((n, 8), np.float32)
    cpt.check(cpt.lib.cpt_tracer_camera_rays(tracer.h, W, H, 1, d_rays.ptr))
    d_inst, d_prim, d_t = cpt.DeviceArray((n,), np.uint32), cpt.DeviceArray((n,), np.uint32), cpt.DeviceArray((n,), np.float32)
    out = {"rays": n, "film": [W, H]}
    hits = {}
    for mode, tag in ((0, "exact"), (1, "fast")):
        tracer.set_traversal(mode)
        cpt.check(cpt.lib.cpt_tracer_first_hit(tracer.h, d_rays.ptr, n, d_inst.ptr, d_prim.ptr, d_t.ptr, None))
        tracer.synchronize()
        tracer.reset_counters()
        reps = 10
        tracer.timer_begin()
        for _ in range(reps):
            cpt.check(cpt.lib.cpt_tracer_first_hit(tracer.h, d_rays.ptr, n, d_inst.ptr, d_prim.ptr, d_t.ptr, None))
        ms = tracer.timer_end() / reps
        c = tracer.counters()
        hits[tag] = (d_inst.to_host(), d_prim.to_host(), d_t.to_host())
        block = {"mrays_per_s": round(n / ms / 1e3, 1), "ms": round(ms, 4), "node_records_per_ray": round(c["node_visits"] / c["rays"], 2),
                 "triangles_per_ray": round(c["tri_tests"] / c["rays"], 2),
                 "roofline": roofline_block("k_first_hit", traversal_bytes(c, node_record_bytes(mode == 0)) / reps, ms, 1, measured_counters(f"k_first_hit/{label}/{tag}"))}
        out[tag] = block
    e, f = hits["exact"], hits["fast"]
    out["fast_vs_exact"] = {"id_mismatches": int(((e[0] != f[0]) | (e[1] != f[1])).sum()), "fast_farther": int((f[2] > e[2]).sum()),
                            "fast_nearer": int((f[2] < e[2]).sum()), "speedup": round(out["fast"]["mrays_per_s"] / out["exact"]["mrays_per_s"], 3)}
    tracer.set_traversal(0)
    return out


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
    ap.add_argument("--traversal", choices=("exact", "fast"), default="exact",
                    help="exact (default): the reference's visit order and arithmetic, bit-exact results; fast: ordered traversal")
    ap.add_argument("--samples-per-launch", type=int, default=128,
                    help="samples (steps) the persistent kernels of integrators 4 and 5 render per launch; the film is bit for bit "
                         "that of one launch per sample (1 = one launch per sample, as the reference launches RenderKernel)")
    ap.add_argument("--scene10m-spp", type=int, default=128, help="samples per pixel of each scene_10m variant, split over the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lbvh", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the primary_rays and scene_10m blocks")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="run the N>1 code path with every rank on GPU 0 and the reduce over gloo (functional rehearsal, not a measurement: "
                         "RCCL does not accept two ranks on one device)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world, **({} if rehearse else {"device_id": device}))

    cpt = load_cpt()
    cpt.check(cpt.lib.cpt_set_device(local_rank))
    W, H, K, WU = args.width, args.height, args.steps, args.warmup
    fast = 1 if args.traversal == "fast" else 0

    comm = None
    comm_note = ""
    if world > 1 and not rehearse:
        # the communicator id travels by torch.distributed; everything after that is the library's own RCCL use
        ident = [None]
        if rank == 0:
            try:
                ident = [cpt.Comm.unique_id()]
            except Exception as e:                       # librccl could not be opened: say so and reduce through torch.distributed instead
                ident = [repr(e)]
        dist.broadcast_object_list(ident, src=0)
        ok = torch.tensor([0], dtype=torch.int32, device=device)
        if isinstance(ident[0], bytes):
            try:
                comm = cpt.Comm(ident[0], rank, world)
                ok += 1
            except Exception as e:
                comm_note = repr(e)
        else:
            comm_note = str(ident[0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:                          # all ranks or none
            comm = None
            comm_note = "library communicator unavailable (" + (comm_note or "another rank failed") + "): films reduced by torch.distributed"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(counters):
        if world == 1:
            return dict(counters)
        t = torch.tensor([counters[k] for k in COUNTER_KEYS], dtype=torch.int64, device=torch.device("cpu") if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return dict(zip(COUNTER_KEYS, [int(x) for x in t.tolist()]))

    def sharded_render(tracer, begin, count, finish=True):
        """This rank's block as sums, then the one exchange of the path: reduce to rank 0 (+ division there)."""
        tracer.reset()
        tracer.render_samples(begin, count, True)
        if comm is not None:
            tracer.reduce_film(comm, root=0, finish=finish)
        elif world > 1 and rehearse:                    # rehearsal: same blocks, the sum over gloo on the host
            tracer.synchronize()
            host = torch.from_numpy(tracer.film())
            sharding.reduce_sum_to_root(host, dst=0)
            return host.numpy()
        elif world > 1:                                 # fallback only: the films as torch tensors, summed by torch.distributed
            tracer.synchronize()
            dev = torch.from_numpy(tracer.film()).to(device)
            sharding.reduce_sum_to_root(dev, dst=0)
            return dev.cpu().numpy()
        return None

    # ------------------------------------------------------------------ headline workload
    if args.workload == "cornell-box-glass":
        d = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-glass")
        scene = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
        workload = f"cornell-box-glass {W}x{H}, max-depth {args.max_depth}, glass box"
        scene_triangles = 36
    else:
        kind, _, size = args.workload.partition("-")
        ntris = {"1m": 1_000_000, "10m": 10_000_000}.get(size)
        if kind not in ("heightfield", "soup") or ntris is None:
            raise SystemExit(f"unknown workload {args.workload}")
        scene, scene_triangles = synthetic(cpt, kind, ntris, W, H)
        workload = f"synthetic {kind}, {scene_triangles} triangles (emissive quad included), {W}x{H}, max-depth {args.max_depth}"
    tracer = cpt.Tracer(scene, W, H, max_depth=args.max_depth)
    tracer.set_integrator(args.integrator)
    tracer.set_traversal(fast)
    tracer.set_samples_per_launch(args.samples_per_launch)
    tracer.build_buffers()
    integrator = args.integrator if args.integrator != 3 else (4 if scene_triangles >= 65536 else 5)     # what mode 3 resolves to in the library
    per_launch = args.samples_per_launch if integrator in (4, 5) else 1

    if world == 1:
        # the reference's own path: PathTracer::Update, one sample per call, running mean in the film
        tracer.update(WU)
        tracer.synchronize()
        tracer.reset_counters()
        barrier()
        t0 = time.perf_counter()
        kernel_ms = tracer.update_timed(K)      # K samples bracketed by HIP events on the launch stream
        barrier()
        elapsed = time.perf_counter() - t0
        own = K
    else:
        sharded_render(tracer, 1, WU)
        tracer.synchronize()
        tracer.reset_counters()
        begin, own = sharding.sample_block(rank, world, K)
        barrier()
        t0 = time.perf_counter()
        tracer.timer_begin()
        reduced = sharded_render(tracer, begin, own)
        kernel_ms = tracer.timer_end()          # this rank's launches and its part of the reduce, HIP events on the launch stream
        barrier()
        elapsed = time.perf_counter() - t0
        if rank == 0:
            film = tracer.film() if reduced is None else reduced
            want = 1.0 if reduced is None else float(K * world)
            if not (np.all(film[..., 3] == want) and np.all(np.isfinite(film))):
                raise SystemExit("reduced film does not hold K x N samples per pixel")
    counters = tracer.counters()
    elapsed = all_max(elapsed)
    total = all_sum(counters)

    result = None
    if rank == 0:
        paths = W * H * K * world
        kernel_name = KERNEL_OF[integrator]
        result = {
            "metric": "Mpaths/s per node, cornell-box-glass 1024x1024 max-depth 8 (Mrays/s, 1M/10M-triangle scenes and LBVH build ms alongside)"
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
            "data": "reference scene file cornell-box-glass (committed fixture); synthetic scenes from seeded generators for the other blocks"
                    if args.workload == "cornell-box-glass" else "synthetic scene from a seeded generator (tests/helpers.py)",
            "config": {"workload": f"{workload}, 1 spp per step, {K} spp per GPU", "traversal": args.traversal,
                       "samples_per_launch": per_launch, "integrator": INTEGRATOR_NAME[integrator],
                       "sharding": "replicas only" if world == 1 else f"{world} x disjoint sample blocks, one ncclReduce of the {W * H * 16 / 1e6:.1f} MB films issued by the library"
                       if comm is not None else (f"{world} ranks rehearsed on one GPU, sum over gloo" if rehearse else comm_note)},
            "mrays_per_s": round(total["rays"] / elapsed / 1e6, 3),
            "rays_per_path": round(total["rays"] / max(total["paths"], 1), 4),
        }
        launches = -(-own // per_launch)
        avg_ms = kernel_ms / launches
        samples_in_launch = own / launches                               # what an average timed launch really held
        alg_bytes = bounce_bytes(counters, W * H * own, node_record_bytes(args.traversal == "exact")) / launches
        m = measured_counters(f"{kernel_name}/{args.workload}/{args.traversal}") if (W, H) == (1024, 1024) or args.workload != "cornell-box-glass" else None
        roof = roofline_block(kernel_name, alg_bytes, avg_ms, samples_in_launch, m)
        roof["launches"] = launches
        result["roofline"] = roof
    del tracer, scene

    # ------------------------------------------------------------------ the other configurations, alongside
    if not args.no_extras and args.workload == "cornell-box-glass":
        if world == 1:
            block = {}
            try:
                for label, kind, ntris, w, h in (("heightfield-1m", "heightfield", 1_000_000, 1920, 1080), ("soup-1m", "soup", 1_000_000, 1920, 1080)):
                    sc, _ = synthetic(cpt, kind, ntris, w, h)
                    tr = cpt.Tracer(sc, w, h, max_depth=5).build_buffers()
                    block[label] = primary_ray_block(cpt, tr, w, h, label)
                    del tr, sc
            except Exception as e:                       # the headline number stands on its own
                block["error"] = repr(e)
            result["primary_rays"] = block
        # 10M triangles, 3840x2160: S samples split over the ranks (strong scaling), reduce + division included
        try:
            w, h, S = 3840, 2160, args.scene10m_spp
            sc, ntri = synthetic(cpt, "heightfield", 10_000_000, w, h)
            tr = cpt.Tracer(sc, w, h, max_depth=5)
            tr.set_samples_per_launch(32)
            tr.build_buffers(profile=True)
            stages = tr.build_stage_ms()
            block = {"scene": f"synthetic heightfield, {ntri} triangles (emissive quad included), {w}x{h}", "spp_total": S, "scaling": "strong",
                     "bvh_ms": {"leaf_boxes": round(stages["leaf_boxes"], 4), "build": round(stages["to_traversable"] - stages["leaf_boxes"] - stages["pack_records"], 4),
                                "pack_records": round(stages["pack_records"], 4), "to_traversable": round(stages["to_traversable"], 4),
                                "note": "device time of one build inside build_buffers (first build of the process: scratch allocation excluded, caches cold)"},
                     "sharding": "replicas only" if world == 1 else f"{S} samples split over {world} ranks, one ncclReduce of the {w * h * 16 / 1e6:.1f} MB films"}
            if world == 1:
                block["primary_rays"] = primary_ray_block(cpt, tr, w, h, "heightfield-10m")
            begin, count = cpt.shard_samples(S, rank, world)
            variants = (("depth 5, megakernel", 5, 4, 0), ("depth 5, wavefront", 5, 2, 0), ("unbounded, megakernel", -1, 4, 0),
                        ("unbounded, wavefront", -1, 2, 0), ("depth 5, megakernel, fast traversal", 5, 4, 1))
            for name, depth, mode, trav in variants:
                tr.set_max_depth(depth)
                tr.set_integrator(mode)
                tr.set_traversal(trav)
                sharded_render(tr, 1, 2)
                tr.synchronize()
                tr.reset_counters()
                barrier()
                t0 = time.perf_counter()
                tr.timer_begin()
                sharded_render(tr, begin, count)
                ms_dev = tr.timer_end()
                barrier()
                dt = all_max(time.perf_counter() - t0)
                c_own = tr.counters()
                c = all_sum(c_own)
                if rank == 0:
                    v = {"mpaths_per_s": round(w * h * S / dt / 1e6, 1), "mrays_per_s": round(c["rays"] / dt / 1e6, 1), "ms_per_sample": round(dt / S * 1e3, 4),
                         "rays_per_path": round(c["rays"] / max(c["paths"], 1), 3), "node_records_per_ray": round(c["node_visits"] / max(c["rays"], 1), 2)}
                    kname = KERNEL_OF[mode]
                    launches = max(1, -(-count // 32)) if mode == 4 else 1            # rank 0's own launches against its own bytes
                    m = measured_counters(f"{kname}/heightfield-10m/{'fast' if trav else 'exact'}") if (depth == 5 and mode == 4) else None
                    v["roofline"] = roofline_block(kname, bounce_bytes(c_own, w * h * count, node_record_bytes(trav == 0)) / launches, ms_dev / launches, count / launches, m)
                    block[name] = v
            if rank == 0:
                result["scene_10m"] = block
            del tr, sc
        except Exception as e:
            if rank == 0:
                result["scene_10m"] = {"error": repr(e)}

    if rank == 0:
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
            head, others = cpu_baselines(W, H, args.max_depth)
            result["cpu_baseline"] = head
            result["cpu_baseline_other_configs"] = others
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
