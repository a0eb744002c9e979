"""Scene ingestion at full size, end to end, on the GPU box: a synthetic scene written as .obj/.mtl/.json
(scripts/gen_scene.py), rendered by the executable twice with --scene-cache 1 (first run parses the text and writes the
binary copy, second run reads it), and the image compared bit for bit with the render of the same triangles handed
over in memory.  Reports seconds (generation, each run's load / build / first sample / total from --timings) and each
run's peak resident set.
usage: python scripts/ingest_scene.py <heightfield|soup> <ntris> <out.json> [width height spp depth]"""
import json, os, re, shutil, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import numpy as np
from conftest import load_cpt
from scenes import synthetic_scene, BENCH_CAMERA

kind, ntris, out_json = sys.argv[1], int(sys.argv[2]), sys.argv[3]
W = int(sys.argv[4]) if len(sys.argv) > 4 else 3840
H = int(sys.argv[5]) if len(sys.argv) > 5 else 2160
spp = int(sys.argv[6]) if len(sys.argv) > 6 else 4
depth = int(sys.argv[7]) if len(sys.argv) > 7 else 5
BINARY = os.path.join(ROOT, "cuda-path-tracer_amd", "cuda-pathtracer")
work = tempfile.mkdtemp(prefix="cpt_ingest_")
prefix = os.path.join(work, "scene")
report = {"scene": f"{kind}, {ntris} triangles asked for", "film": [W, H], "spp": spp, "max_depth": depth}


def read_exr(path):
    """The uncompressed four-channel scanline files Film::SaveTo writes (channels A, B, G, R)."""
    b = open(path, "rb").read()
    p = 8
    attrs = {}
    while b[p] != 0:
        e = b.index(b"\0", p); name = b[p:e].decode(); p = e + 1
        e = b.index(b"\0", p); p = e + 1
        size = struct.unpack("<I", b[p:p + 4])[0]; p += 4
        attrs[name] = b[p:p + size]; p += size
    p += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    img = np.zeros((h, w, 4), np.float32)
    offsets = struct.unpack(f"<{h}Q", b[p:p + 8 * h])
    for y in range(h):
        q = offsets[y]
        yy, size = struct.unpack("<iI", b[q:q + 8])
        line = np.frombuffer(b, np.float32, 4 * w, q + 8).reshape(4, w)
        img[yy, :, 3], img[yy, :, 2], img[yy, :, 1], img[yy, :, 0] = line[0], line[1], line[2], line[3]
    return img


def run_cli(tag):
    t0 = time.perf_counter()
    p = subprocess.Popen([BINARY, prefix + ".obj", prefix + ".json", "--ui", "0", "--max-depth", str(depth), "--max-spp", str(spp), "--scene-cache", "1",
                          "--timings", "1", "-o", os.path.join(work, tag)], stdout=subprocess.PIPE, text=True)
    text = p.stdout.read()
    _, status, usage = os.wait4(p.pid, 0)
    wall = time.perf_counter() - t0
    if status != 0:
        raise SystemExit(f"{tag}: the executable failed ({status}):\n{text}")
    stamps = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"timing: (.*?) after ([0-9.e+-]+) s", text)}
    return {"wall_s": round(wall, 3), "peak_rss_mib": round(usage.ru_maxrss / 1024.0, 1), "from_cache": "scene read from" in text,
            "seconds_after_start": stamps}


try:
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_scene.py"), kind, str(ntris), prefix, str(W), str(H)], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    report["generate_s"] = round(time.perf_counter() - t0, 2)
    report["obj_mib"] = round(os.path.getsize(prefix + ".obj") / 2**20, 1)
    print("generated", report["obj_mib"], "MiB in", report["generate_s"], "s", flush=True)
    report["first_run_parses_the_text"] = run_cli("parsed")
    print("first run", report["first_run_parses_the_text"], flush=True)
    report["cache_mib"] = round(os.path.getsize(prefix + ".obj.cptcache") / 2**20, 1)
    report["second_run_reads_the_cache"] = run_cli("cached")
    print("second run", report["second_run_reads_the_cache"], flush=True)
    parsed, cached = read_exr(os.path.join(work, "parsed.exr")), read_exr(os.path.join(work, "cached.exr"))
    # the same triangles handed over in memory, with the materials the .mtl file gives them
    cpt = load_cpt()
    objs, cam = synthetic_scene(kind, ntris)
    mat = lambda ke, kd: np.array(list(ke) + list(kd) + [0, 0, 0] + [0, 0, 0] + [1.0, 1.0, 1.0, 2], dtype=np.float32)      # Ns 1, Ni 1 (tinyobjloader's default) -> opacity 1
    objs[0]["material"] = mat([0, 0, 0], [0.7, 0.7, 0.7])
    objs[1]["material"] = mat([17, 12, 4], [0, 0, 0])
    sc = cpt.Scene()
    for o in objs:
        sc.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    sc.set_camera(BENCH_CAMERA["pos"], BENCH_CAMERA["look_at"], cam["up"], BENCH_CAMERA["fov"], W, H)
    tr = cpt.Tracer(sc, W, H, max_depth=depth)
    tr.set_integrator(3)
    tr.build_buffers()
    tr.update(spp)
    mem = tr.film()
    report["triangles"] = int(sum(len(o["indices"]) // 3 for o in objs))
    report["image_of_parsed_scene_equals_in_memory_scene_bitwise"] = bool(np.array_equal(parsed.view(np.uint32), mem.view(np.uint32)))
    report["image_of_cached_scene_equals_in_memory_scene_bitwise"] = bool(np.array_equal(cached.view(np.uint32), mem.view(np.uint32)))
    report["lit_pixel_fraction"] = round(float((mem[..., :3] > 0).any(axis=-1).mean()), 4)
finally:
    shutil.rmtree(work, ignore_errors=True)
json.dump(report, open(out_json, "w"), indent=1)
print(json.dumps(report))
if not (report.get("image_of_parsed_scene_equals_in_memory_scene_bitwise") and report.get("image_of_cached_scene_equals_in_memory_scene_bitwise")):
    raise SystemExit("images differ")
