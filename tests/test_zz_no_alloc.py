"""GPU: cpt_build_accel_noalloc -- the LBVH build whose scratch lives in a fixed device array inside the library --
leaves bit for bit what cpt_build_accel leaves (nodes and every box), on two calls in a row, the second with a
smaller input (the array keeps the first call's contents), and rejects inputs beyond its stated maximum."""
import numpy as np
import pytest

from helpers import *  # noqa: F403

pytestmark = pytest.mark.gpu


def build_through(cpt, entry, boxes, merged):
    n = len(boxes)
    nn = 2 * n - 1
    host = np.zeros((nn, 8), np.float32)
    host[n - 1:] = boxes
    d_boxes = cpt.DeviceArray.from_host(host)
    d_nodes = cpt.DeviceArray((nn, 2), np.uint32)
    d_nodes.zero()
    cpt.check(entry(d_nodes.ptr, d_boxes.ptr, np.ascontiguousarray(merged, dtype=np.float32), n))
    return d_nodes.to_host(), d_boxes.to_host()


def case(kind, n, seed):
    mesh = random_soup(n, seed=seed) if kind == "soup" else heightfield(int(round((n / 2) ** 0.5)), int(round((n / 2) ** 0.5)))
    return boxes_of_triangles(mesh["positions"], mesh["indices"])


def test_noalloc_build_equals_the_allocating_build_twice_in_a_row(cpt):
    cpt.require_device()
    limit = cpt.lib.cpt_build_accel_noalloc_max_primitives()
    assert limit == 1 << 20
    # first the maximum size (every byte of the array in use), then smaller inputs over its leftovers
    for kind, n, seed in (("soup", limit, 5), ("heightfield", 180000, 0), ("soup", 4097, 6), ("soup", 2, 7), ("soup", 1, 8)):
        boxes, merged = case(kind, n, seed)
        ref_nodes, ref_boxes = build_through(cpt, cpt.lib.cpt_build_accel, boxes, merged)
        got_nodes, got_boxes = build_through(cpt, cpt.lib.cpt_build_accel_noalloc, boxes, merged)
        assert np.array_equal(got_nodes, ref_nodes), f"{kind} {n}: node arrays differ"
        assert np.array_equal(got_boxes.view(np.uint32), ref_boxes.view(np.uint32)), f"{kind} {n}: boxes differ bitwise"


def test_noalloc_build_takes_no_device_memory(cpt):
    """Free device memory reads the same before and after calls of the entry point (its buffers are the caller's and the
    library's fixed array), for a size it has seen and for one it has not."""
    cpt.require_device()
    sizes = (70000, 70000, 300001)
    inputs = []
    for n in sizes:
        boxes, merged = case("soup", n, 11)
        host = np.zeros((2 * n - 1, 8), np.float32)
        host[n - 1:] = boxes
        inputs.append((cpt.DeviceArray.from_host(host), cpt.DeviceArray((2 * n - 1, 2), np.uint32), np.ascontiguousarray(merged, dtype=np.float32), n))
    d_boxes, d_nodes, merged, n = inputs[0]
    cpt.check(cpt.lib.cpt_build_accel_noalloc(d_nodes.ptr, d_boxes.ptr, merged, n))       # first call on this device: the array's address is looked up
    cpt.check(cpt.lib.cpt_synchronize())
    free_before, _ = cpt.memory_info()
    for d_boxes, d_nodes, merged, n in inputs[1:]:
        cpt.check(cpt.lib.cpt_build_accel_noalloc(d_nodes.ptr, d_boxes.ptr, merged, n))
        free_after, _ = cpt.memory_info()
        assert free_after == free_before, f"{n} primitives: free device memory moved by {free_before - free_after} bytes"


def test_noalloc_build_rejects_inputs_beyond_its_array(cpt):
    cpt.require_device()
    n = cpt.lib.cpt_build_accel_noalloc_max_primitives() + 1
    d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    merged = np.array([0, 0, 0, 1, 1, 1, 1, 1], np.float32)
    rc = cpt.lib.cpt_build_accel_noalloc(d_nodes.ptr, d_boxes.ptr, merged, n)
    assert rc == -1
    assert b"at most" in cpt.lib.cpt_last_error()
