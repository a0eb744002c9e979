#!/usr/bin/env python3
"""Benchmark of the gfx950 path-tracing core.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Headline workload (what `value` is quoted on): the synthetic 10M-triangle scene (tessellated heightfield + one emissive
quad, seeded generator), 3840x2160, max depth 5, exact traversal, persistent megakernel.  One step is one sample of
every pixel on every GPU; the default K = 512 steps is one GPU's share of the 4096-spp render on 8 GPUs.  The
persistent kernel renders --samples-per-launch of them per launch, leaving the film of one launch per sample bit for bit.

N > 1 (one process per GPU): every rank renders its own block of K sample indices as sums (weak scaling: 512 x N
samples per pixel in all, no data-path collective) and the timed region ends with ONE ncclReduce of the 132.7 MB films
to rank 0 and the division by the sample count, both issued by the library (cpt_tracer_render_samples +
cpt_tracer_reduce_film, RCCL called from C++); torch is used only to hand the 128-byte communicator id to the ranks,
for the barriers around the timed region and to gather the ranks' clocks and counters afterwards.  Should the library's
communicator not come up, the line says FALLBACK in config.sharding and `value` is null: such a run is not a
measurement of the RCCL path.

Alongside, in the same JSON line (blocks that do not change `value`):
  variants            the same scene and film with the wavefront (ray-queue) integrator, with unbounded depth (Russian
                      roulette only) for both integrators, and with the ordered traversal; weak scaling like the headline
  primary_rays        camera rays on the 10M scene and on the 1M-triangle heightfield and soup: Mrays/s, records per ray
  cornell_box_glass   cornell-box-glass 1024x1024, 1024 spp, max depth 8 on one GPU
  lbvh                build times at 1M and 10M primitives, the streaming-copy rate and the vector-issue rate of the device
  cpu_baseline        the oracle on the host cores: a bounded sample of the headline workload; the Cornell box at
                      512x512 / 64 spp / depth 5 in full and the 1M build + primary rays in cpu_baseline_other_configs

Every roofline object: `achieved` = algorithmic bytes of one launch / its duration measured in this run with HIP events
on the launch stream, against the 8 TB/s HBM peak (`frac`); `traffic` = HBM bytes of such a launch from the rocprofv3
counter passes kept under profiles/ (`counters_from` names the file, the entry and the samples per launch of the pass),
with `hbm_frac` from it; `valu_frac` = vector instructions per second (same passes) against the issue rate this run
measured on the device (cpt_valu_issue_rate), `useful_valu_frac` = that x the fraction of lanes active.  Blocks without
a counter pass carry frac = null.  Megakernel rooflines also carry `address_unit`: the traversal's 16-byte lane accesses
of a launch (seven per record visit, three per triangle test, from this run's work counters) over the launch's CU cycles
-- the load on the address unit of the vector memory pipeline, which takes diverged lanes one per cycle.

Prints ONE JSON line on rank 0.  Scene data comes from seeded generators (cuda-path-tracer_amd/scenes.py) and the
committed scene fixtures; nothing outside this repository is read.  The oracle (oracle/) is used only for the
`cpu_baseline` legs.
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

import torch  # noqa: E402  (first: the HIP runtime torch ships must be the one the library binds to)
import torch.distributed as dist  # noqa: E402
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import scenes  # noqa: E402
import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E, vendor peak
SHADER_CLOCK_HZ = 2.4e9                # MI355X engine clock; the render kernels hold it (GRBM_GUI_ACTIVE / 8 XCDs / duration in the counter passes)
COUNTER_FILE = os.path.join("profiles", "r03_counters.json")
KERNEL_OF = {0: "k_render_regen", 1: "k_render", 2: "k_wf_generate + k_wf_trace + k_wf_shade + k_wf_finish", 4: "k_render_flow", 5: "k_render_pool"}
COUNTER_KERNEL = {0: "k_render_regen", 1: "k_render", 2: "k_wf", 4: "k_render_flow", 5: "k_render_pool"}       # first part of a counter entry's key
INTEGRATOR_NAME = {0: "persistent megakernel (lane regeneration)", 1: "megakernel, one lane per pixel", 2: "wavefront (ray queues) + persistent tail",
                   4: "persistent megakernel, traversal state kept across shading",
                   5: "persistent megakernel, finished lanes walk the wave's pooled shadow rays"}
COUNTER_KEYS = ("rays", "node_visits", "tri_tests", "paths", "shades")
WORKLOADS = {  # name -> (generator kind, triangles, width, height, max depth, default steps)
    "heightfield-10m": ("heightfield", 10_000_000, 3840, 2160, 5, 512),
    "soup-10m": ("soup", 10_000_000, 3840, 2160, 5, 128),
    "heightfield-1m": ("heightfield", 1_000_000, 1920, 1080, 5, 512),
    "soup-1m": ("soup", 1_000_000, 1920, 1080, 5, 256),
    "cornell-box-glass": (None, 36, 1024, 1024, 8, 1024),
}


def load_cpt():
    """The package's ctypes binding, imported by path (its directory name has a hyphen).  Raises when libcpt_amd.so
    has not been built: there is no other implementation to fall back to."""
    name = "cuda_path_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "cuda-path-tracer_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def node_record_bytes(exact):
    """Bytes a visit reads: the exact traversal walks two-level records (128 B apart, of which it loads 112: seven 16-byte
    words, the eighth is spare), the ordered one of the batch queries binary records of 64 B."""
    return 112 if exact else 64


def traversal_bytes(c, node_bytes):
    """Per ray 32 B in + 16 B hit out, node_bytes per node record visited, 48 B per triangle record tested."""
    return 48 * c["rays"] + node_bytes * c["node_visits"] + 48 * c["tri_tests"]


def bounce_bytes(c, pixel_samples, node_bytes):
    """Traversal bytes + per shaded hit 3 x (12 + 12 + 8) B of vertex attributes and 112 B of material + 16 B read and
    16 B written per pixel and sample for the accumulation buffer."""
    return traversal_bytes(c, node_bytes) + (96 + 112) * c["shades"] + 32 * pixel_samples


_counter_table = None


def measured_counters(key):
    """Per-sample hardware counter values of one kernel on one workload, from the rocprofv3 --pmc passes kept under
    profiles/ (scripts/profile_round3.sh writes the file); None when there is no entry."""
    global _counter_table
    if _counter_table is None:
        try:
            _counter_table = json.load(open(os.path.join(ROOT, COUNTER_FILE)))
        except Exception:
            _counter_table = {}
    m = _counter_table.get(key)
    if m is not None:
        m = dict(m, key=key)
    return m


def roofline_block(kernel, algorithmic_bytes, avg_ms, samples, m, valu_peak):
    """Roofline object of one launch (or, for the wavefront integrator, of one sample's kernel sequence).
    Live in this run: the duration (HIP events on the launch stream), the algorithmic bytes (work counters) and the vector-issue
    ceiling of the device.  From the counter passes kept under profiles/ (per sample, scaled to the samples this launch held):
    HBM bytes and vector instructions."""
    alg_gbs = algorithmic_bytes / (avg_ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(alg_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "avg_launch_ms": round(avg_ms, 5), "samples_per_launch": round(samples, 2), "algorithmic_bytes_per_launch": int(algorithmic_bytes),
         "algorithmic_frac_of_hbm_peak": round(alg_gbs / HBM_PEAK_GBS, 4)}
    if not m:
        r["note"] = "no counter pass for this launch shape: HBM traffic unknown, frac withheld"
        return r
    traffic = m["hbm_bytes_per_sample"] * samples
    hbm_gbs = traffic / (avg_ms * 1e-3) / 1e9
    r.update(frac=round(alg_gbs / HBM_PEAK_GBS, 4), traffic=int(traffic), hbm_gb_per_s=round(hbm_gbs, 1), hbm_frac=round(hbm_gbs / HBM_PEAK_GBS, 4),
             traffic_over_algorithmic=round(traffic / max(algorithmic_bytes, 1), 4))
    if "valu_insts_per_sample" in m and valu_peak:
        ginst = m["valu_insts_per_sample"] * samples / (avg_ms * 1e-3) / 1e9
        r.update(valu_ginst_per_s=round(ginst, 1), valu_peak_measured_ginst_per_s=round(valu_peak, 1), valu_frac=round(ginst / valu_peak, 4))
        if "lanes_active" in m:
            r["useful_valu_frac"] = round(ginst / valu_peak * m["lanes_active"], 4)
    r["counters"] = {k: m[k] for k in ("valu_busy", "lanes_active", "l1_hit", "l2_hit", "salu_per_valu", "lds_bank_conflict_frac") if k in m}
    r["counters_from"] = {"file": COUNTER_FILE, "entry": m["key"], "samples_per_launch_of_the_pass": m.get("samples_per_dispatch"), "command": m.get("note", ""),
                          "scaling": "per-sample counter values x the samples this run's launch held"}
    return r


def address_unit_block(c, launches, avg_ms, words_per_node, compute_units, clock_hz):
    """The traversal's load on the address unit of the vector memory pipeline, from this run's work counters: a record
    visit is words_per_node 16-byte loads per lane and a triangle test three, and the unit takes lanes whose addresses
    lie in different lines one per cycle (rays that have diverged; lanes of one line share a cycle).  So the lane
    accesses of a launch over its CU cycles bound how busy the unit is from below for coherent rays and state it for
    diverged ones; the counter passes under profiles/ (TA_TA_BUSY over TCP_GATE_EN1) hold the measured figure."""
    accesses = (words_per_node * c["node_visits"] + 3 * c["tri_tests"]) / max(launches, 1)
    cycles = compute_units * clock_hz * avg_ms * 1e-3
    return {"traversal_lane_accesses_per_launch": int(accesses), "cu_cycles_per_launch": int(cycles),
            "lane_accesses_per_cu_cycle": round(accesses / max(cycles, 1.0), 4), "words_per_record_visit": words_per_node,
            "compute_units": compute_units, "shader_clock_mhz_assumed": round(clock_hz / 1e6, 1)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def lbvh_timings(cpt, n, seed=12345, repeats=10, sort_variant=0, kind="soup"):
    """Median device time of the whole build (Morton -> sort -> nodes -> refit) with buffers pre-allocated, after
    two warm-ups; plus the per-stage split of one extra run (hierarchy = leaf gather + topology + boxes).  The
    primitives arrive in random order (kind "soup": every leaf box the hierarchy pass gathers is a memory access of
    its own) or in the order of a mesh (kind "heightfield": neighbours in the file are neighbours in space)."""
    mesh = scenes.synthetic_scene(kind, n, seed)[0][0]
    boxes, merged = scenes.boxes_of_triangles(mesh["positions"], mesh["indices"])
    n = len(boxes)
    del mesh
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


def valu_issue_ceiling(cpt):
    """Vector instructions the device issues per second, measured now: independent wave64 instructions on every SIMD, the
    instruction kinds of a box test and plain v_fma_f32, at 4 and 8 waves per SIMD.  The largest figure is the ceiling."""
    table = {}
    for mix, name in ((1, "box-test mix"), (0, "v_fma_f32")):
        for waves in (1, 4, 8):
            g = C.c_double(0.0)
            cpt.check(cpt.lib.cpt_valu_issue_rate(mix, waves, C.byref(g), None, None))
            table[f"{name}, {waves} waves per SIMD"] = round(g.value, 1)
    return max(table.values()), table


# --------------------------------------------------------------------------- CPU legs (oracle = checker, timed as the baseline)
def oracle_bindings():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bindings
    return bindings


def cpu_render_leg(osc, width, height, max_depth, cores, target_seconds=None, spp_exact=None):
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
    return dict(value=round(width * height * spp / dt / 1e6, 4), unit="Mpaths/s", cores=cores, host_cores_visible=os.cpu_count(), kind="port",
                mrays_per_s=round(int(cnt[0]) / dt / 1e6, 4), seconds=round(dt, 2), spp=spp)


def cpu_baselines(workload, objs, cam, W, H, max_depth):
    """The oracle timed on the host cores: a bounded sample of the headline workload, configuration 1 in full, and the 1M
    build + camera rays.  16 threads at most (the box's CPU share); the number of cores the host shows is reported next to it."""
    B = oracle_bindings()
    cores = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    osc = B.OracleScene(objs, cam)                    # uploads + bottom-level and top-level LBVH builds, one thread
    build_s = time.perf_counter() - t0
    head = cpu_render_leg(osc, W, H, max_depth, cores, target_seconds=12.0)
    head["sample"] = f"{head['spp']} spp of the same {W}x{H} depth-{max_depth} workload ({head['seconds']} s); scene upload + LBVH build on one thread took {build_s:.1f} s"
    head["lbvh_build_ms_one_thread"] = round(build_s * 1e3, 1)
    head["cpu"] = cpu_model()
    del osc
    others = {}
    cobjs, ccam = B.cornell(glass=False, bsdf_type=2)
    c1 = cpu_render_leg(B.OracleScene(cobjs, ccam), 512, 512, 5, cores, spp_exact=64)
    c1["sample"] = "configuration 1 in full: cornell-box 512x512, 64 spp, max-depth 5, blinn-phong"
    others["cornell-box 512x512 64 spp depth 5"] = c1
    gobjs, gcam = B.cornell(glass=True, bsdf_type=2)
    c2 = cpu_render_leg(B.OracleScene(gobjs, gcam), 1024, 1024, 8, cores, target_seconds=6.0)
    c2["sample"] = f"{c2['spp']} spp of configuration 2: cornell-box-glass 1024x1024, max-depth 8"
    others["cornell-box-glass 1024x1024 depth 8"] = c2
    sobjs, scam = scenes.synthetic_scene("heightfield", 1_000_000)
    scam = dict(scam, **scenes.BENCH_CAMERA)
    t0 = time.perf_counter()
    osc = B.OracleScene(sobjs, scam)
    build_s = time.perf_counter() - t0
    rays = osc.camera_rays(1920, 1080, 1)
    t0 = time.perf_counter()
    _, _, _, _, cnt = osc.first_hit(rays, nthreads=cores)
    dt = time.perf_counter() - t0
    others["heightfield 1M: LBVH build + primary rays 1920x1080"] = dict(
        lbvh_build_ms=round(build_s * 1e3, 1), build_threads=1, mrays_per_s=round(len(rays) / dt / 1e6, 3), cores=cores, kind="port",
        node_visits_per_ray=round(int(cnt[1]) / max(int(cnt[0]), 1), 2),
        sample="configuration 3: scene upload + LBVH build of 1M triangles (single thread), 2,073,600 camera rays on all cores")
    return head, others


# --------------------------------------------------------------------------- scenes
def synthetic(cpt, kind, ntris, W, H, keep_arrays=False):
    objs, cam = scenes.synthetic_scene(kind, ntris)            # seeded generator: every rank builds the same scene
    cam = dict(cam, **scenes.BENCH_CAMERA)
    scene = cpt.Scene()
    for o in objs:
        scene.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    scene.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
    ntri = sum(len(o["indices"]) // 3 for o in objs)
    return scene, ntri, (objs, cam) if keep_arrays else None


def primary_ray_block(cpt, tracer, W, H, label, valu_peak):
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
                 "roofline": roofline_block("k_first_hit", traversal_bytes(c, node_record_bytes(mode == 0)) / reps, ms, 1,
                                            measured_counters(f"k_first_hit/{label}/{tag}"), valu_peak)}
        out[tag] = block
    e, f = hits["exact"], hits["fast"]
    out["fast_vs_exact"] = {"id_mismatches": int(((e[0] != f[0]) | (e[1] != f[1])).sum()), "fast_farther": int((f[2] > e[2]).sum()),
                            "fast_nearer": int((f[2] < e[2]).sum()), "speedup": round(out["fast"]["mrays_per_s"] / out["exact"]["mrays_per_s"], 3)}
    tracer.set_traversal(0)
    return out


def keep_stdout_for_the_result():
    """The contract is ONE line on stdout.  Libraries print there too (RCCL writes a version banner when its first
    communicator comes up), so everything this process and its libraries write to file descriptor 1 goes to stderr from
    here on, and the result line is written to the descriptor that stdout was."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    return real


def main():
    result_fd = keep_stdout_for_the_result()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="samples per pixel per GPU in the timed region (default: 512 for the headline workload)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="heightfield-10m", choices=sorted(WORKLOADS),
                    help="heightfield-10m (default: the 10M-triangle 3840x2160 configuration the metric is quoted on), cornell-box-glass, or another synthetic scene")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--integrator", type=int, default=3,
                    help="3 by scene size (default: 5 for the Cornell box, 4 for the synthetic scenes), 0 persistent megakernel, 1 megakernel, "
                         "2 wavefront, 4 persistent megakernel that shades partly finished waves, 5 persistent megakernel with pooled shadow rays")
    ap.add_argument("--traversal", choices=("exact", "fast"), default="exact",
                    help="exact (default): the reference's visit order and arithmetic, bit-exact results; fast: ordered traversal")
    ap.add_argument("--samples-per-launch", type=int, default=None,
                    help="samples (steps) the persistent kernels of integrators 4 and 5 render per launch (default 32, Cornell box 128); the film is bit "
                         "for bit that of one launch per sample (1 = one launch per sample, as the reference launches RenderKernel)")
    ap.add_argument("--variant-spp", type=int, default=32, help="samples per pixel and GPU of each entry of the variants block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lbvh", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the variants, primary_rays and cornell_box_glass blocks")
    ap.add_argument("--single-rank-rccl", action="store_true",
                    help="with --gpus 1: run the N>1 code path on one rank -- process group over nccl, the library's RCCL communicator of one rank, "
                         "samples rendered as sums and passed through ncclReduce (what a one-GPU box can execute of the sharded path)")
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
    sharded = world > 1 or args.single_rank_rccl        # the one-process-per-GPU path: sums, one reduce to rank 0
    if args.single_rank_rccl and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if sharded:
        dist.init_process_group("gloo" if rehearse else "nccl", rank=rank, world_size=world, **({} if rehearse else {"device_id": device}))

    props = torch.cuda.get_device_properties(device)
    compute_units, clock_hz = int(props.multi_processor_count), SHADER_CLOCK_HZ

    cpt = load_cpt()
    cpt.check(cpt.lib.cpt_set_device(local_rank))
    kind, ntris_asked, dW, dH, ddepth, dsteps = WORKLOADS[args.workload]
    W, H = args.width or dW, args.height or dH
    max_depth = ddepth if args.max_depth is None else args.max_depth
    K, WU = (dsteps if args.steps is None else args.steps), args.warmup
    fast = 1 if args.traversal == "fast" else 0
    default_per_launch = 128 if kind is None else 32
    samples_per_launch = args.samples_per_launch or default_per_launch

    comm = None
    comm_note = ""
    rccl_census = None
    if sharded and not rehearse:
        # the communicator id travels by torch.distributed; everything after that is the library's own RCCL use
        ident = [None]
        if rank == 0:
            try:
                ident = [cpt.Comm.unique_id()]
            except Exception as e:                       # librccl could not be opened: say so (the run is then no measurement of the RCCL path)
                ident = [repr(e)]
        dist.broadcast_object_list(ident, src=0)
        ok = torch.tensor([0], dtype=torch.int32, device=device)
        if isinstance(ident[0], bytes):
            try:
                comm = cpt.Comm(ident[0], rank, world)
                rccl_census = comm.census()
                if rccl_census[0] != world or rccl_census[1] != rank:
                    raise RuntimeError(f"RCCL reports {rccl_census[0]} ranks / rank {rccl_census[1]}, expected {world} / {rank}")
                ok += 1
            except Exception as e:
                comm_note = repr(e)
        else:
            comm_note = str(ident[0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:                          # all ranks or none
            comm = None
            comm_note = "FALLBACK: library communicator unavailable (" + (comm_note or "another rank failed") + "); films reduced by torch.distributed, not a measurement of the library's RCCL path"

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(x):
        if not sharded:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=torch.device("cpu") if rehearse else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(counters):
        if not sharded:
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
        elif sharded and rehearse:                      # rehearsal: same blocks, the sum over gloo on the host
            tracer.synchronize()
            host = torch.from_numpy(tracer.film())
            sharding.reduce_sum_to_root(host, dst=0)
            return host.numpy()
        elif sharded:                                   # fallback only: the films as torch tensors, summed by torch.distributed
            tracer.synchronize()
            dev = torch.from_numpy(tracer.film()).to(device)
            sharding.reduce_sum_to_root(dev, dst=0)
            return dev.cpu().numpy()
        return None

    def timed_block(tracer, count):
        """`count` samples per rank inside barriers: rank r renders sample indices r*count+1 .. (r+1)*count.  Returns
        (wall seconds: max over ranks, device ms of this rank's stream, this rank's counters, summed counters)."""
        tracer.synchronize()
        tracer.reset_counters()
        reduced = None
        if not sharded:
            # the reference's own path: PathTracer::Update, one sample per call, running mean in the film
            tracer.reset()
            barrier()
            t0 = time.perf_counter()
            ms_dev = tracer.update_timed(count)      # `count` samples bracketed by HIP events on the launch stream
            barrier()
            dt = time.perf_counter() - t0
        else:
            begin, own = sharding.sample_block(rank, world, count)
            barrier()
            t0 = time.perf_counter()
            tracer.timer_begin()
            reduced = sharded_render(tracer, begin, own)
            ms_dev = tracer.timer_end()              # this rank's launches and its part of the reduce, HIP events on the launch stream
            barrier()
            dt = time.perf_counter() - t0
            if rank == 0:
                film = tracer.film() if reduced is None else reduced
                want = 1.0 if reduced is None else float(count * world)
                if not (np.all(film[..., 3] == want) and np.all(np.isfinite(film))):
                    raise SystemExit("reduced film does not hold count x N samples per pixel")
        own_c = tracer.counters()
        return all_max(dt), ms_dev, own_c, all_sum(own_c)

    valu_peak, valu_table = valu_issue_ceiling(cpt) if rank == 0 else (None, None)
    barrier()

    # ------------------------------------------------------------------ headline workload
    arrays = None
    if kind is None:
        d = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-glass")
        scene = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
        workload = f"cornell-box-glass {W}x{H}, max-depth {max_depth}, glass box (BASELINE configuration 2)"
        scene_triangles = 36
    else:
        want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
        scene, scene_triangles, arrays = synthetic(cpt, kind, ntris_asked, W, H, keep_arrays=want_cpu)
        workload = f"synthetic {kind}, {scene_triangles} triangles (emissive quad included), {W}x{H}, max-depth {max_depth}"
        if args.workload == "heightfield-10m":
            workload += " (BASELINE configuration 4: 4096 spp = 512 spp per GPU on 8 GPUs)"
    tracer = cpt.Tracer(scene, W, H, max_depth=max_depth)
    tracer.set_integrator(args.integrator)
    tracer.set_traversal(fast)
    tracer.set_samples_per_launch(samples_per_launch)
    tracer.build_buffers()                 # the process's first launches load the code object; the timed build is the second one
    tracer.build_buffers(profile=True)
    stages = tracer.build_stage_ms()
    integrator = args.integrator if args.integrator != 3 else (4 if scene_triangles >= 65536 else 5)     # what mode 3 resolves to in the library
    per_launch = samples_per_launch if integrator in (4, 5) else 1

    if not sharded:
        tracer.update(WU)
    else:
        sharded_render(tracer, 1, WU)
    elapsed, kernel_ms, counters, total = timed_block(tracer, K)

    result = None
    if rank == 0:
        paths = W * H * K * world
        kernel_name = KERNEL_OF[integrator]
        if not sharded:
            sharding_note = "replicas only"
        elif comm is not None:
            sharding_note = (f"{world} x disjoint blocks of {K} sample indices, one ncclReduce of the {W * H * 16 / 1e6:.1f} MB films issued by the library; "
                             f"RCCL reports {rccl_census[0]} ranks (this is rank {rccl_census[1]} on device {rccl_census[2]})")
        elif rehearse:
            sharding_note = f"{world} ranks rehearsed on one GPU, sum over gloo"
        else:
            sharding_note = comm_note
        valid = not sharded or comm is not None or rehearse
        result = {
            "metric": f"Mpaths/s per node, {args.workload} {W}x{H} max-depth {max_depth} (Mrays/s, LBVH build ms @1M/10M, cornell-box-glass alongside)",
            "value": round(paths / elapsed / 1e6, 3) if valid else None,
            "unit": "Mpaths/s",
            "n_gpus": world,
            "steps": K,
            "warmup": WU,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic scene from a seeded generator (cuda-path-tracer_amd/scenes.py); reference scene file cornell-box-glass (committed fixture) for the cornell_box_glass block"
                    if kind is not None else "reference scene file cornell-box-glass (committed fixture)",
            "config": {"workload": f"{workload}, 1 spp per step, {K} spp per GPU, {K * world} spp in all", "traversal": args.traversal,
                       "samples_per_launch": per_launch, "integrator": INTEGRATOR_NAME[integrator], "sharding": sharding_note,
                       "rccl_ranks": rccl_census[0] if rccl_census else None},
            "mrays_per_s": round(total["rays"] / elapsed / 1e6, 3),
            "rays_per_path": round(total["rays"] / max(total["paths"], 1), 4),
            "node_records_per_ray": round(total["node_visits"] / max(total["rays"], 1), 3),
            "bvh_ms": {"leaf_boxes": round(stages["leaf_boxes"], 4), "build": round(stages["to_traversable"] - stages["leaf_boxes"] - stages["pack_records"], 4),
                       "pack_records": round(stages["pack_records"], 4), "to_traversable": round(stages["to_traversable"], 4),
                       "note": "device time of the largest mesh's build inside build_buffers (second build of the process; scratch allocation excluded)"},
            "valu_issue_rate_ginst_per_s": valu_table,
        }
        launches = -(-K // per_launch)
        avg_ms = kernel_ms / launches
        samples_in_launch = K / launches                               # what an average timed launch really held
        alg_bytes = bounce_bytes(counters, W * H * K, node_record_bytes(True)) / launches
        m = measured_counters(f"{COUNTER_KERNEL[integrator]}/{args.workload}/{args.traversal}") if (W, H) == (dW, dH) and max_depth == ddepth else None
        roof = roofline_block(kernel_name, alg_bytes, avg_ms, samples_in_launch, m, valu_peak)
        roof["launches"] = launches
        roof["address_unit"] = address_unit_block(counters, launches, avg_ms, 7, compute_units, clock_hz)
        if sharded:
            roof["note_multi_gpu"] = "rank 0's launches; avg_launch_ms includes this rank's share of the reduce"
        result["roofline"] = roof

    # ------------------------------------------------------------------ the other configurations, alongside
    extras = not args.no_extras and args.workload == "heightfield-10m"
    if extras:
        if world == 1:
            try:
                result["primary_rays"] = {"heightfield-10m": primary_ray_block(cpt, tracer, W, H, "heightfield-10m", valu_peak)}
            except Exception as e:                       # the headline number stands on its own
                result["primary_rays"] = {"error": repr(e)}
        # BASELINE configuration 5 (unbounded depth, megakernel against wavefront) and the other variants on the same scene and film
        S = args.variant_spp
        block = {"samples_per_gpu": S, "scaling": "weak", "scene": workload}
        variants = (("depth 5, wavefront", 5, 2, 0), ("unbounded, megakernel", -1, 4, 0), ("unbounded, wavefront", -1, 2, 0),
                    ("depth 5, megakernel, fast traversal", 5, 4, 1))
        try:
            for name, depth, mode, trav in variants:
                tracer.set_max_depth(depth)
                tracer.set_integrator(mode)
                tracer.set_traversal(trav)
                if not sharded:
                    tracer.reset()
                    tracer.update(2)
                else:
                    sharded_render(tracer, 1, 2)
                dt, ms_dev, c_own, c = timed_block(tracer, S)
                if rank == 0:
                    v = {"mpaths_per_s": round(W * H * S * world / dt / 1e6, 1), "mrays_per_s": round(c["rays"] / dt / 1e6, 1), "ms_per_sample": round(dt / S * 1e3, 4),
                         "rays_per_path": round(c["rays"] / max(c["paths"], 1), 3), "node_records_per_ray": round(c["node_visits"] / max(c["rays"], 1), 2)}
                    kname = KERNEL_OF[mode]
                    launches = max(1, -(-S // per_launch)) if mode == 4 else S           # wavefront: one kernel sequence per sample
                    tag = ("unbounded/" if depth < 0 else "") + ("fast" if trav else "exact")
                    m = measured_counters(f"{COUNTER_KERNEL[mode]}/heightfield-10m/{tag}")
                    v["roofline"] = roofline_block(kname, bounce_bytes(c_own, W * H * S, node_record_bytes(True)) / launches, ms_dev / launches, S / launches, m, valu_peak)
                    if mode == 4:
                        v["roofline"]["address_unit"] = address_unit_block(c_own, launches, ms_dev / launches, 7, compute_units, clock_hz)
                    block[name] = v
            tracer.set_max_depth(max_depth)
            tracer.set_integrator(args.integrator)
            tracer.set_traversal(fast)
        except Exception as e:
            block["error"] = repr(e)
        if rank == 0:
            block["depth 5, megakernel"] = "the headline: value, mrays_per_s and roofline of this line"
            result["variants"] = block
    del tracer, scene

    if extras and world == 1:
        try:
            cw, ch, cdepth, cspp = 1024, 1024, 8, 1024
            d = os.path.join(ROOT, "tests", "golden", "scenes", "cornell-box-glass")
            sc = cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))
            tr = cpt.Tracer(sc, cw, ch, max_depth=cdepth)
            tr.set_integrator(3)
            tr.set_samples_per_launch(128)
            tr.build_buffers()
            tr.update(128)
            dt, ms_dev, c_own, c = timed_block(tr, cspp)
            launches = cspp // 128
            result["cornell_box_glass"] = {
                "workload": f"cornell-box-glass {cw}x{ch}, {cspp} spp, max-depth {cdepth}, glass box (BASELINE configuration 2), exact traversal, 128 samples per launch",
                "mpaths_per_s": round(cw * ch * cspp / dt / 1e6, 1), "mrays_per_s": round(c["rays"] / dt / 1e6, 1), "ms_per_sample": round(dt / cspp * 1e3, 5),
                "rays_per_path": round(c["rays"] / max(c["paths"], 1), 3),
                "roofline": roofline_block("k_render_pool", bounce_bytes(c_own, cw * ch * cspp, 128) / launches, ms_dev / launches, 128,
                                           measured_counters("k_render_pool/cornell-box-glass/exact"), valu_peak)}
            del tr, sc
        except Exception as e:
            result["cornell_box_glass"] = {"error": repr(e)}
        try:
            for label, skind, ntris, w, h in (("heightfield-1m", "heightfield", 1_000_000, 1920, 1080), ("soup-1m", "soup", 1_000_000, 1920, 1080)):
                sc, _, _ = synthetic(cpt, skind, ntris, w, h)
                tr = cpt.Tracer(sc, w, h, max_depth=5).build_buffers()
                result["primary_rays"][label] = primary_ray_block(cpt, tr, w, h, label, valu_peak)
                del tr, sc
        except Exception as e:
            result["primary_rays"]["error"] = repr(e)

    if rank == 0:
        if world == 1 and not args.no_lbvh:
            result["lbvh"] = {"1M": lbvh_timings(cpt, 1_000_000), "10M": lbvh_timings(cpt, 10_000_000),
                              "10M, primitives in mesh order": lbvh_timings(cpt, 10_000_000, kind="heightfield")}
            # what a kernel that only streams reaches on this device (1 GiB copied, read + written bytes counted),
            # and the 10M build's 312 bytes per primitive against that instead of the nominal peak
            rate = C.c_double(0.0)
            cpt.check(cpt.lib.cpt_stream_copy_rate(1 << 30, 10, C.byref(rate)))
            result["lbvh"]["stream_copy_gb_per_s"] = round(rate.value, 1)
            med = result["lbvh"]["10M"]["build_ms"]
            result["lbvh"]["10M"]["frac_of_stream_copy"] = round(312.0 * 10_000_000 / (med * 1e-3) / 1e9 / rate.value, 4)
        if world == 1 and not args.no_cpu_baseline:
            if arrays is not None:
                head, others = cpu_baselines(args.workload, arrays[0], arrays[1], W, H, max_depth)
            else:
                B = oracle_bindings()
                gobjs, gcam = B.cornell(glass=True, bsdf_type=2)
                head = cpu_render_leg(B.OracleScene(gobjs, gcam), W, H, max_depth, min(16, os.cpu_count() or 1), target_seconds=10.0)
                head["sample"] = f"{head['spp']} spp of the same {W}x{H} depth-{max_depth} workload ({head['seconds']} s)"
                head["cpu"] = cpu_model()
                others = {}
            result["cpu_baseline"] = head
            result["cpu_baseline_other_configs"] = others
        os.write(result_fd, (json.dumps(result) + "\n").encode())

    if sharded:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
