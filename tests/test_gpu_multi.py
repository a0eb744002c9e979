"""GPU: the multi-GPU render path of the C++ product on the GPUs this machine has.  With one GPU the shards share
it: the same threads, streams, replicated builds, per-device sum, RCCL reduce (one rank) and final division run, and
the result must equal the single-tracer render up to the order of fp32 additions (<= 1e-5 relative).  With two or more
GPUs visible the same tests also spread the shards over them."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import *  # noqa: F403
from test_capi_and_host import BINARY, parse_exr

pytestmark = pytest.mark.gpu


def cornell_scene(cpt):
    d = os.path.join(SCENES, "cornell-box-glass")
    return cpt.Scene().load_obj(os.path.join(d, "cornell-box.obj"), 2).load_json(os.path.join(d, "cornell-box.json"))


def single_render(cpt, scene, W, H, depth, spp):
    tr = cpt.Tracer(scene, W, H, max_depth=depth)
    tr.set_integrator(3)
    tr.build_buffers()
    tr.update(spp)
    return tr.film(), tr.counters()


def close_to(img, ref):
    assert np.all(np.isfinite(img))
    assert np.array_equal(img[..., 3], ref[..., 3])
    assert np.allclose(img[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shards,spp", [(1, 8), (3, 10), (4, 3), (5, 64)])
def test_shards_in_one_process_equal_the_single_render(cpt, shards, spp):
    scene = cornell_scene(cpt)
    W, H, depth = 96, 64, 6
    ref, rc = single_render(cpt, scene, W, H, depth, spp)
    gpus = max(1, cpt.device_count())
    multi = cpt.MultiTracer(scene, W, H, [s % gpus for s in range(shards)], max_depth=depth)
    multi.set_integrator(3)
    multi.build_buffers()
    assert multi.num_devices() == min(gpus, shards)
    ms = multi.render(spp)
    assert ms > 0
    close_to(multi.film(), ref)
    c = multi.counters()
    assert c["paths"] == W * H * spp == rc["paths"]
    assert c["rays"] == rc["rays"]                      # the same paths, whoever traced them
    assert c["stack_overflows"] == 0
    # a second render of the same object starts from a cleared film
    multi.reset_counters()
    multi.render(spp)
    close_to(multi.film(), ref)
    multi.close()


def test_one_rank_per_process_interface(cpt):
    """The per-process half (what a one-process-per-GPU launcher drives): block of samples as sums, ncclReduce to the
    root on the tracer's stream, division by the sample count.  One rank here; the call sequence is the N-rank one."""
    scene = cornell_scene(cpt)
    W, H, depth, total = 64, 64, 5, 12
    ref, _ = single_render(cpt, scene, W, H, depth, total)
    comm = cpt.Comm(cpt.Comm.unique_id(), 0, 1)
    assert comm.census() == (1, 0, 0)                     # what RCCL itself reports: one rank, rank 0, on device 0
    tr = cpt.Tracer(scene, W, H, max_depth=depth)
    tr.set_integrator(3)
    tr.build_buffers()
    begin, count = cpt.shard_samples(total, comm.rank, comm.world)
    assert (begin, count) == (1, total)
    tr.reset()
    tr.render_samples(begin, count, True)
    tr.reduce_film(comm, root=0, finish=True)
    close_to(tr.film(), ref)
    # sums of two blocks rendered one after the other, reduced without the division
    tr.reset()
    tr.render_samples(1, 5, True)
    tr.render_samples(6, 7, True)
    tr.reduce_film(comm, root=0, finish=False)
    sums = tr.film()
    assert np.all(sums[..., 3] == float(total))
    assert np.allclose(sums[..., :3] / total, ref[..., :3], rtol=1e-5, atol=1e-6)
    comm.close()


def test_cli_with_shards_writes_the_same_image(cpt, tmp_path):
    d = os.path.join(SCENES, "cornell-box")
    cam = tmp_path / "cam.json"
    cam.write_text('{"camera": {"pos": [6.8, 1, 0], "look_at": [0, 1, 0], "up": [0, 1, 0], "fov": 19.5, "resolution": [64, 48]}}')
    images = []
    for name, extra in (("one", []), ("three", ["--gpus", "1", "--shards", "3"])):
        out = tmp_path / name
        r = subprocess.run([BINARY, os.path.join(d, "cornell-box.obj"), str(cam), "--ui", "0", "--max-depth", "5", "--max-spp", "40", "-o", str(out)] + extra,
                           capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "spp 16" in r.stdout and "spp 32" in r.stdout
        if extra:
            assert "3 shards on 1 GPU(s)" in r.stdout
        images.append(parse_exr(str(out) + ".exr")[0])
    close_to(images[1], images[0])
    assert images[0][..., :3].mean() > 0.05
    r = subprocess.run([BINARY, os.path.join(d, "cornell-box.obj"), str(cam), "--gpus", "999"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "visible" in r.stdout


def test_ten_million_triangles_sharded(cpt):
    """The 10M-triangle configuration through the sharded path (two replicas of the scene on the GPUs present): every
    pixel receives all its samples and the image equals the unsharded one."""
    objs, cam = synthetic_scene("heightfield", 10_000_000)
    scene = cpt.Scene()
    for o in objs:
        scene.add_object(o["name"], o["positions"], o["normals"], o["texcoords"], o["indices"], o["material"])
    W, H, depth, spp = 960, 540, 5, 6
    scene.set_camera(cam["pos"], cam["look_at"], cam["up"], cam["fov"], W, H)
    ref, rc = single_render(cpt, scene, W, H, depth, spp)
    gpus = max(1, cpt.device_count())
    multi = cpt.MultiTracer(scene, W, H, [0, 1 % gpus], max_depth=depth)
    multi.set_integrator(3)
    multi.build_buffers()
    multi.render(spp)
    img = multi.film()
    close_to(img, ref)
    assert img[..., :3].mean() > 1e-3
    c = multi.counters()
    assert c["paths"] == W * H * spp and c["rays"] == rc["rays"] and c["stack_overflows"] == 0
    multi.close()


def test_bench_runs_its_sharded_path_on_one_rank_with_the_library_communicator(cpt):
    """bench.py --single-rank-rccl: everything the N > 1 line does that one GPU can execute -- torch's process group over nccl,
    the library's own RCCL communicator made inside that process from an id, samples rendered as sums, ncclReduce on the
    tracer's stream, division on the root -- and the line says how many ranks RCCL itself reports."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-rank-rccl", "--workload", "cornell-box-glass", "--steps", "8",
                        "--warmup", "2", "--width", "256", "--height", "256", "--no-extras", "--no-lbvh", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is not None and line["value"] > 0 and line["n_gpus"] == 1
    assert line["config"]["rccl_ranks"] == 1 and "RCCL reports 1 ranks" in line["config"]["sharding"]
    assert "FALLBACK" not in line["config"]["sharding"]
