"""CPU, world_size 2 over gloo: the multi-GPU path (disjoint sample blocks per rank, one sum
reduce, no other exchange) reproduces the single-process render.  The oracle stands in for the
renderer here because this machine has no GPU; the sharding and reduction code is the product's."""
import os
import sys

import numpy as np
import pytest

from helpers import *  # noqa: F403
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
import sharding  # noqa: E402


def test_partitions():
    assert sharding.sample_block(0, 8, 512) == (1, 512)
    assert sharding.sample_block(7, 8, 512) == (3585, 512)
    parts = sharding.split_samples(4096, 8)
    assert parts[0] == (1, 512) and parts[-1] == (3585, 512)
    parts = sharding.split_samples(10, 4)
    assert parts == [(1, 3), (4, 3), (7, 2), (9, 2)]
    covered = [s for b, n in parts for s in range(b, b + n)]
    assert covered == list(range(1, 11))
    with pytest.raises(ValueError):
        sharding.sample_block(2, 2, 4)


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    objs, cam = cornell(glass=True)
    osc = OracleScene(objs, cam)
    W, H, per_rank = 32, 32, 3
    begin, count = sharding.sample_block(rank, world, per_rank)
    img, _ = osc.render(W, H, begin, begin + count, 6, sum_mode=1, nthreads=2)
    buf = torch.from_numpy(img)
    sharding.reduce_sum_to_root(buf, dst=0)
    if rank == 0:
        np.save(out_path, buf.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sum_equals_single_process(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "sum.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    objs, cam = cornell(glass=True)
    ref, _ = OracleScene(objs, cam).render(32, 32, 1, 7, 6, sum_mode=1, nthreads=4)
    assert np.array_equal(got[..., 3], np.full((32, 32), 6.0, np.float32))
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)          # fp32 summation order only
    mean = sharding.finish_image(got)
    assert np.all(mean[..., 3] == 1.0)
