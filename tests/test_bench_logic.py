"""CPU: the parts of bench.py that decide what a line claims -- the default workload, the sample blocks of the ranks,
and the roofline object (what it withholds without a counter pass, how it scales one)."""
import importlib.util
import os

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_headline_is_the_10m_configuration_at_512_samples_per_gpu(bench):
    kind, ntris, w, h, depth, steps = bench.WORKLOADS["heightfield-10m"]
    assert (kind, ntris, w, h, depth, steps) == ("heightfield", 10_000_000, 3840, 2160, 5, 512)
    assert bench.COUNTER_FILE.startswith("profiles")


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sample_blocks_of_the_ranks_cover_every_index_once(bench, world):
    """Rank r renders sample indices r*K+1 .. (r+1)*K: together 1 .. K*N, each once.  Sample 1, the only one whose camera
    ray goes through the pixel centre instead of a random subpixel (/root/reference/src/kernels/integrator/path.cu:94),
    belongs to rank 0 and to no other."""
    K = 512
    blocks = [bench.sharding.sample_block(r, world, K) for r in range(world)]
    covered = [s for b, n in blocks for s in range(b, b + n)]
    assert covered == list(range(1, K * world + 1))
    owners_of_first = [r for r, (b, n) in enumerate(blocks) if b <= 1 < b + n]
    assert owners_of_first == [0]
    if world == 8:
        assert sum(n for _, n in blocks) == 4096             # BASELINE configuration 4


def test_roofline_without_a_counter_pass_withholds_the_fraction(bench):
    r = bench.roofline_block("k", 8e9, 2.0, 32, None, 1000.0)
    assert r["bound"] == "hbm" and r["frac"] is None and r["traffic"] is None
    assert r["achieved"] == 4000.0 and r["algorithmic_frac_of_hbm_peak"] == 0.5 and "no counter pass" in r["note"]


def test_roofline_with_a_counter_pass(bench):
    m = {"key": "k/w/exact", "hbm_bytes_per_sample": 1e9, "valu_insts_per_sample": 5e7, "lanes_active": 0.5, "samples_per_dispatch": 32.0, "note": "cmd"}
    r = bench.roofline_block("k", 8e9, 2.0, 16, m, 1000.0)          # a launch of 16 samples, 2 ms
    assert r["frac"] == 0.5 and r["traffic"] == 16e9
    assert r["hbm_gb_per_s"] == 8000.0 and r["hbm_frac"] == 1.0 and r["traffic_over_algorithmic"] == 2.0
    assert r["valu_ginst_per_s"] == 400.0 and r["valu_frac"] == 0.4 and r["useful_valu_frac"] == 0.2
    assert r["counters_from"]["entry"] == "k/w/exact" and r["counters_from"]["samples_per_launch_of_the_pass"] == 32.0


def test_algorithmic_bytes_follow_the_work_counters(bench):
    c = {"rays": 10, "node_visits": 100, "tri_tests": 20, "paths": 4, "shades": 6}
    assert bench.traversal_bytes(c, 128) == 48 * 10 + 128 * 100 + 48 * 20
    assert bench.bounce_bytes(c, 4, 128) == bench.traversal_bytes(c, 128) + 208 * 6 + 32 * 4


def test_address_unit_load_follows_the_work_counters(bench):
    # two launches that made 1000 record visits of seven loads and 100 triangle tests of three, on 10 CUs at 1 GHz for 1 us each
    c = {"rays": 50, "node_visits": 1000, "tri_tests": 100, "paths": 4, "shades": 6}
    a = bench.address_unit_block(c, 2, 1e-3, 7, 10, 1e9)
    assert a["traversal_lane_accesses_per_launch"] == (7 * 1000 + 3 * 100) // 2
    assert a["cu_cycles_per_launch"] == 10000 and a["lane_accesses_per_cu_cycle"] == 0.365
