"""GPU: parity at the sizes the benchmark configurations name -- 1M and 10M triangles, tessellated (heightfield) and
random (soup).  Per scene: the whole LBVH bit for bit against the oracle, first hits (instance, primitive, distance,
barycentrics) bit for bit for more than 2M rays (camera rays of two samples and rays from inside the scene), any-hit
answers for 1M short rays, and the count of traversal stack entries dropped, which must be zero: these trees are up
to 36 levels deep.  The 10M heightfield is also rendered: a reduced film against the oracle, and the 3840x2160 film
of the 10M configuration through properties that need no oracle (sample counts, finiteness, the same film bit for
bit from the megakernel and the wavefront integrator, the same number of rays)."""
import gc
import os

import numpy as np
import pytest

from helpers import *  # noqa: F403
from test_gpu_traverse import gpu_scene, interior_rays

pytestmark = pytest.mark.gpu

THREADS = min(16, os.cpu_count() or 1)
_live = {}


def scene_pair(cpt, kind, ntris):
    """(objects, camera, oracle scene, product scene, tracer); one pair resident at a time."""
    key = (kind, ntris)
    if key not in _live:
        _live.clear()
        gc.collect()
        objs, cam = synthetic_scene(kind, ntris)
        osc = OracleScene(objs, cam)
        sc, tr = gpu_scene(cpt, objs, cam, 256, 144, max_depth=5)
        _live[key] = (objs, cam, osc, sc, tr)
    return _live[key]


def same_hits(tr, osc, rays):
    o_inst, o_prim, o_t, o_uv, cnt = osc.first_hit(rays, nthreads=THREADS)
    g_inst, g_prim, g_t, g_uv = tr.first_hit(rays)
    assert np.array_equal(g_inst, o_inst), f"{int((g_inst != o_inst).sum())} instance ids differ"
    assert np.array_equal(g_prim, o_prim), f"{int((g_prim != o_prim).sum())} primitive ids differ"
    assert np.array_equal(g_t.view(np.uint32), o_t.view(np.uint32)), "hit distances differ bitwise"
    assert np.array_equal(g_uv, o_uv), "barycentrics differ"
    return o_inst


SIZES = [("soup", 1_000_000), ("heightfield", 1_000_000), ("soup", 10_000_000), ("heightfield", 10_000_000)]      # the last one stays resident for the render tests


@pytest.mark.parametrize("kind,ntris", SIZES)
def test_lbvh_and_hits_at_benchmark_sizes(cpt, kind, ntris):
    objs, cam, osc, sc, tr = scene_pair(cpt, kind, ntris)
    gn, gb, gk = tr.accel(0)
    on, ob, ok = osc.accel(0)
    assert len(gk) >= ntris * 0.999
    assert np.array_equal(gk, ok), "sorted keys differ"
    assert np.array_equal(gn, on), "nodes differ"
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), "boxes differ bitwise"
    del gn, gb, gk, on, ob, ok
    tr.reset_counters()
    hit_fraction = []
    for spp in (1, 2):                                                  # pixel centres, then jittered
        inst = same_hits(tr, osc, osc.camera_rays(1024, 1024, spp))
        hit_fraction.append(float((inst != 0xFFFFFFFF).mean()))
    assert min(hit_fraction) > 0.3
    lo, hi = np.array([0, -0.2, 0], np.float32), np.array([1, 1.2, 1], np.float32)
    same_hits(tr, osc, interior_rays(1 << 20, 31, lo, hi))
    short = interior_rays(1 << 20, 32, lo, hi, tmax=0.05 if kind == "soup" else 0.5)
    g_occ, o_occ = tr.occlude(short), osc.occlude(short)
    assert np.array_equal(g_occ, o_occ)
    assert 0.02 < float(o_occ.mean()) < 0.98
    assert tr.counters()["rays"] == 3 * (1 << 20) + (1 << 20)
    assert tr.stack_overflows() == 0


def rel_l2(a, b):
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))


def test_render_of_the_10m_scene(cpt):
    objs, cam, osc, sc, tr = scene_pair(cpt, "heightfield", 10_000_000)
    tr.set_integrator(3)
    tr.reset()
    tr.reset_counters()
    tr.update(8)
    img = tr.film()
    ref, ocnt = osc.render(256, 144, 1, 9, 5, nthreads=THREADS)
    assert np.all(np.isfinite(img)) and np.array_equal(img[..., 3], ref[..., 3])
    assert ref[..., :3].mean() > 1e-3
    err = rel_l2(img[..., :3], ref[..., :3])
    assert err <= 1e-3, err
    c = tr.counters()
    assert c["paths"] == 256 * 144 * 8
    assert abs(c["rays"] - int(ocnt[0])) <= 1e-3 * int(ocnt[0])
    assert tr.stack_overflows() == 0


@pytest.mark.parametrize("depth", [5, -1])
def test_full_size_film_of_the_10m_configuration(cpt, depth):
    """3840x2160 on the 10M-triangle scene, bounded depth and Russian roulette only: every pixel holds exactly the
    samples asked for, nothing is NaN or infinite, the persistent megakernel (one launch for all samples) and the
    wavefront integrator (queues per bounce) leave the same film bit for bit and trace the same number of rays, and
    sample blocks rendered as sums add up to the running mean."""
    objs, cam, osc, sc, _ = scene_pair(cpt, "heightfield", 10_000_000)
    W, H, S = 3840, 2160, 3
    films, rays = [], []
    for mode in (4, 2):
        tr = cpt.Tracer(sc, W, H, max_depth=depth)
        tr.set_integrator(mode)
        tr.build_buffers()
        tr.reset_counters()
        tr.update(S)
        films.append(tr.film())
        c = tr.counters()
        assert c["paths"] == W * H * S
        assert tr.stack_overflows() == 0
        rays.append(c["rays"])
        if mode == 4:
            d_sum = cpt.DeviceArray((H, W, 4), np.float32)
            d_sum.zero()
            tr.render_samples(1, 2, True, d_sum.ptr, W, H)
            tr.render_samples(3, 1, True, d_sum.ptr, W, H)
            tr.synchronize()
            sums = d_sum.to_host()
            d_sum.free()
        tr.close()
    assert np.all(np.isfinite(films[0]))
    assert np.all(films[0][..., 3] == 1.0)
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32)), "megakernel and wavefront films differ"
    assert rays[0] == rays[1]
    assert np.all(sums[..., 3] == float(S))
    assert np.allclose(sums[..., :3] / S, films[0][..., :3], rtol=1e-5, atol=1e-6)
    assert films[0][..., :3].mean() > 1e-3


def test_scene_files_of_the_10m_configuration_end_to_end(tmp_path):
    """The 10M-triangle scene as files (/root/reference/src/scene/loader.cpp:144-260, src/scene/mesh.cpp:84-149): about
    1.2 GB of .obj text written from the seeded generator, rendered by the executable twice with --scene-cache 1 (the
    first run parses the text, the second reads the binary copy), both images bit for bit the render of the same triangles
    handed over in memory (scripts/ingest_scene.py asserts that and reports times and peak memory)."""
    import json
    import subprocess
    import sys
    _live.clear()
    gc.collect()
    out = tmp_path / "ingest.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ingest_scene.py"), "heightfield", "10000000", str(out), "1920", "1080", "2", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rep = json.load(open(out))
    assert rep["triangles"] > 9_990_000 and rep["obj_mib"] > 1000
    assert rep["image_of_parsed_scene_equals_in_memory_scene_bitwise"] and rep["image_of_cached_scene_equals_in_memory_scene_bitwise"]
    assert not rep["first_run_parses_the_text"]["from_cache"] and rep["second_run_reads_the_cache"]["from_cache"]
    assert rep["lit_pixel_fraction"] > 0.5
