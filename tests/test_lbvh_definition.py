"""CPU: the oracle's LBVH against the DEFINITION of the tree the reference builds, with no search code shared with
either builder.  /root/reference/src/kernels/accel/accel_build.cu:57-138 (Karras 2012) produces the binary radix tree
of the sorted 64-bit keys: an internal node covers a contiguous range [first, last] of sorted positions; it splits
where the highest bit in which key[first] and key[last] differ changes (every key left of the split shares one more
leading bit with key[first] than key[last] does); the left child of a split between positions g and g+1 is internal
node g, or leaf n-1+g when its range is the single position g, the right child internal node g+1 or leaf n-1+g+1
(accel_build.cu:127-132); node i lies at one end of its own range; the root is node 0; leaf n-1+i carries the low
key word (accel_build.cu:43-51); parents[child] is the node that names it (accel_build.cu:134-135).

Everything below is recomputed from the keys alone, by linear scans over Python integers."""
import numpy as np
import pytest

from helpers import *  # noqa: F403


def morton_by_definition(boxes, merged):
    """accel_build.cu:17-41 on float32 numpy scalars: centre, normalise by the merged box, 10 bits per axis, x most
    significant; 0/0 on a flat axis is NaN and fmin(NaN, 1023) = 1023."""
    f = np.float32
    keys = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for i, b in enumerate(boxes):
            code = 0
            q = []
            for a in range(3):
                c = f(f(b[a] + b[4 + a]) * f(0.5))
                p = f(f(c - merged[a]) / f(merged[4 + a] - merged[a]))
                v = f(p * f(1024.0))
                v = f(1023.0) if (np.isnan(v) or v > f(1023.0)) else v
                q.append(int(v) if v > 0 else 0)
            for bit in range(10):
                for a in range(3):                          # x lands on bit 3k+2, y on 3k+1, z on 3k
                    code |= ((q[a] >> bit) & 1) << (3 * bit + (2 - a))
            keys.append((code << 32) | i)
    return keys


def lcp64(a, b):
    x = a ^ b
    return 64 if x == 0 else 64 - x.bit_length()


def check_tree_against_definition(nodes, parents, keys, n):
    assert keys == sorted(keys) and len(set(keys)) == n
    if n == 1:
        assert tuple(nodes[0]) == (keys[0] & 0xFFFFFFFF, 0xFFFFFFFF)
        return
    leaf0 = n - 1
    for i in range(n):
        assert tuple(nodes[leaf0 + i]) == (keys[i] & 0xFFFFFFFF, 0xFFFFFFFF)
    seen = set()
    stack = [(0, 0, n - 1)]
    while stack:
        i, first, last = stack.pop()
        assert i not in seen
        seen.add(i)
        assert i == first or i == last, f"node {i} is not an end of its range [{first}, {last}]"
        common = lcp64(keys[first], keys[last])
        # the split by definition: the last position whose key still shares more than `common` bits with key[first]
        g = first
        while g + 1 < last and lcp64(keys[first], keys[g + 1]) > common:
            g += 1
        # every key on the left shares more than `common` leading bits with key[first], none on the right does
        assert all(lcp64(keys[first], keys[k]) > common for k in range(first, g + 1))
        assert all(lcp64(keys[first], keys[k]) == common for k in range(g + 1, last + 1))
        lc = leaf0 + g if g == first else g
        rc = leaf0 + g + 1 if g + 1 == last else g + 1
        assert (int(nodes[i][0]), int(nodes[i][1])) == (lc, rc), f"node {i} over [{first}, {last}]"
        assert int(parents[lc]) == i and int(parents[rc]) == i
        if g != first:
            stack.append((g, first, g))
        if g + 1 != last:
            stack.append((g + 1, g + 1, last))
    assert seen == set(range(n - 1))


def oracle_build(boxes, merged):
    L = oracle()
    n = len(boxes)
    nn = 2 * n - 1
    o_boxes = np.zeros((nn, 8), np.float32)
    o_boxes[n - 1:] = boxes
    o_nodes = np.zeros((nn, 2), np.uint32)
    o_keys = np.zeros(n, np.uint64)
    o_par = np.zeros(nn, np.uint32)
    L.orc_build_accel(o_nodes.reshape(-1), o_boxes.reshape(-1), np.ascontiguousarray(merged), n, o_keys.ctypes.data, o_par.ctypes.data)
    return o_nodes, o_boxes, o_keys, o_par


def random_boxes(n, seed, mode):
    rng = np.random.default_rng(seed)
    lo = rng.random((n, 3), dtype=np.float32)
    if mode == "clustered":            # few distinct Morton cells: most splits are decided by the index bits
        lo = (np.floor(lo * 3) / 3).astype(np.float32)
    elif mode == "identical":
        lo[:] = np.float32(0.25)
    elif mode == "flat":               # a flat axis: 0/0 -> NaN -> 1023 on it
        lo[:, 1] = np.float32(0.5)
    ext = (rng.random((n, 3), dtype=np.float32) * np.float32(0.01)).astype(np.float32)
    if mode in ("identical", "flat"):
        ext[:, 1] = 0
    if mode == "identical":
        ext[:] = np.float32(0.125)
    boxes = np.ones((n, 8), np.float32)
    boxes[:, 0:3] = lo
    boxes[:, 4:7] = lo + ext
    merged = np.ones(8, np.float32)
    merged[0:3] = boxes[:, 0:3].min(axis=0)
    merged[4:7] = boxes[:, 4:7].max(axis=0)
    return boxes, merged


@pytest.mark.parametrize("mode", ["uniform", "clustered", "identical", "flat"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 33, 256, 1000, 4096])
def test_oracle_lbvh_is_the_radix_tree_of_the_definition(n, mode):
    boxes, merged = random_boxes(n, 100 + n, mode)
    nodes, out_boxes, keys, parents = oracle_build(boxes, merged)
    expect = sorted(morton_by_definition(boxes, merged))
    assert [int(k) for k in keys] == expect, "keys or their order differ from the definition"
    check_tree_against_definition(nodes, parents, expect, n)
    # boxes: leaves are the inputs in key order; an internal box is the exact min/max over its leaves, which (every
    # internal node having two children) is the min/max over its two children's boxes
    order = [k & 0xFFFFFFFF for k in expect]
    assert np.array_equal(out_boxes[n - 1:, [0, 1, 2, 4, 5, 6]], boxes[order][:, [0, 1, 2, 4, 5, 6]])
    for i in range(n - 1):
        lc, rc = int(nodes[i][0]), int(nodes[i][1])
        assert np.array_equal(out_boxes[i, 0:3], np.minimum(out_boxes[lc, 0:3], out_boxes[rc, 0:3]))
        assert np.array_equal(out_boxes[i, 4:7], np.maximum(out_boxes[lc, 4:7], out_boxes[rc, 4:7]))


def test_definition_checker_rejects_a_wrong_tree():
    """The checker is not vacuous: swapping two children, or moving a split by one, is caught."""
    boxes, merged = random_boxes(64, 5, "uniform")
    nodes, _, keys, parents = oracle_build(boxes, merged)
    keys = [int(k) for k in keys]
    check_tree_against_definition(nodes, parents, keys, 64)
    bad = nodes.copy()
    bad[3] = bad[3][::-1]
    with pytest.raises(AssertionError):
        check_tree_against_definition(bad, parents, keys, 64)
    bad = nodes.copy()
    inner = [i for i in range(63) if bad[i][0] < 63 and bad[i][1] < 63][0]
    bad[inner] = (bad[inner][0] + 1, bad[inner][1] + 1)
    with pytest.raises(AssertionError):
        check_tree_against_definition(bad, parents, keys, 64)
