"""GPU: the LBVH builder behind cpt_build_accel is bit-exact against the oracle -- sorted 64-bit
Morton keys, node topology, parent links and every box (compared as bit patterns)."""
import numpy as np
import pytest

from helpers import *  # noqa: F403

pytestmark = pytest.mark.gpu


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


def assert_same_build(cpt, boxes, merged):
    o_nodes, o_boxes, o_keys, o_par = oracle_build(boxes, merged)
    g_nodes, g_boxes, g_keys, g_par, _ = cpt.build_accel(boxes, merged)
    assert np.array_equal(g_keys, o_keys), "sorted Morton keys differ"
    assert np.array_equal(g_nodes, o_nodes), "node arrays differ"
    assert np.array_equal(g_par[1:], o_par[1:]), "parent links differ"
    assert np.array_equal(g_boxes.view(np.uint32), o_boxes.view(np.uint32)), "boxes differ bitwise"
    return g_nodes, g_boxes, g_keys


@pytest.mark.parametrize("glass", [False, True])
def test_cornell_meshes(cpt, glass):
    objs, _ = cornell(glass=glass)
    for o in objs:
        boxes, merged = boxes_of_triangles(o["positions"], o["indices"])
        assert_same_build(cpt, boxes, merged)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 64, 65, 1000, 4096, 4097, 12289, 100000])
def test_random_soup(cpt, n):
    soup = random_soup(n, seed=1000 + n)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    assert_same_build(cpt, boxes, merged)


def test_random_soup_above_the_small_sort_limit(cpt):
    """Inputs above 3M keys take the other launch shapes of the build (one key per lane in the key kernel, 16 keys per
    lane in the sort, no fused digit count): the same bit-for-bit comparison with the oracle, at 3.2M triangles."""
    soup = random_soup(3_200_000, seed=77)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    del soup
    assert_same_build(cpt, boxes, merged)


def test_heightfield(cpt):
    hf = heightfield(300, 300)
    boxes, merged = boxes_of_triangles(hf["positions"], hf["indices"])
    assert_same_build(cpt, boxes, merged)


def test_identical_boxes_fall_back_to_the_index_bits(cpt):
    # equal Morton codes everywhere: order and topology are decided by the low 32 key bits
    n = 3000
    boxes = np.tile(np.array([[0.25, 0.25, 0.25, 1, 0.5, 0.5, 0.5, 1]], np.float32), (n, 1))
    merged = np.array([0, 0, 0, 1, 1, 1, 1, 1], np.float32)
    nodes, _, keys = assert_same_build(cpt, boxes, merged)
    assert np.array_equal((keys & np.uint64(0xFFFFFFFF)).astype(np.uint32), np.arange(n, dtype=np.uint32))


def test_flat_axes_and_signed_zeros(cpt):
    rng = np.random.default_rng(3)
    n = 2000
    lo = rng.random((n, 3), dtype=np.float32) - np.float32(0.5)
    hi = lo + rng.random((n, 3), dtype=np.float32) * np.float32(0.1)
    lo[:, 1] = np.float32(-0.0)       # the whole set is flat in y: 0/0 -> NaN -> 1023
    hi[:, 1] = np.float32(0.0)
    lo[::7, 0] = np.float32(-0.0)
    boxes = np.ones((n, 8), np.float32)
    boxes[:, 0:3] = lo
    boxes[:, 4:7] = hi
    merged = np.ones(8, np.float32)
    merged[0:3] = lo.min(axis=0)
    merged[4:7] = hi.max(axis=0)
    merged[1] = np.float32(-0.0)
    assert_same_build(cpt, boxes, merged)


def test_in_place_entry_point_matches_ex(cpt):
    soup = random_soup(5000, seed=5)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    n = len(boxes)
    all_boxes = np.zeros((2 * n - 1, 8), np.float32)
    all_boxes[n - 1:] = boxes
    d_boxes = cpt.DeviceArray.from_host(all_boxes)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    cpt.check(cpt.lib.cpt_build_accel(d_nodes.ptr, d_boxes.ptr, merged, n))
    o_nodes, o_boxes, _, _ = oracle_build(boxes, merged)
    assert np.array_equal(d_nodes.to_host(), o_nodes)
    assert np.array_equal(d_boxes.to_host().view(np.uint32), o_boxes.view(np.uint32))
    assert cpt.lib.cpt_build_accel(d_nodes.ptr, d_boxes.ptr, merged, 0) == 0           # empty input is a no-op
    assert cpt.lib.cpt_build_accel(None, d_boxes.ptr, merged, n) < 0                   # null pointer is rejected


def check_tree_properties(nodes, boxes, keys, leaf_boxes, merged):
    """Size-independent invariants of a finished build."""
    n = len(keys)
    assert np.all(keys[1:] > keys[:-1]), "keys not strictly ascending"
    prim = (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert np.array_equal(np.sort(prim), np.arange(n, dtype=np.uint32)), "not a permutation of the primitives"
    assert np.array_equal(nodes[n - 1:, 0], prim) and np.all(nodes[n - 1:, 1] == 0xFFFFFFFF)
    assert np.array_equal(boxes[n - 1:].view(np.uint32), leaf_boxes[prim].view(np.uint32)), "leaf boxes not permuted with the keys"
    lc, rc = nodes[:n - 1, 0].astype(np.int64), nodes[:n - 1, 1].astype(np.int64)
    kids = np.concatenate([lc, rc])
    assert np.array_equal(np.sort(kids), np.arange(1, 2 * n - 1)), "every node except the root must have exactly one parent"
    lo = np.minimum(boxes[lc, 0:3], boxes[rc, 0:3])
    hi = np.maximum(boxes[lc, 4:7], boxes[rc, 4:7])
    assert np.array_equal(boxes[:n - 1, 0:3], lo) and np.array_equal(boxes[:n - 1, 4:7], hi), "internal box != union of children"
    assert np.array_equal(boxes[0, 0:3], leaf_boxes[:, 0:3].min(axis=0)) and np.array_equal(boxes[0, 4:7], leaf_boxes[:, 4:7].max(axis=0))


@pytest.mark.parametrize("n,kind", [(1_000_000, "soup"), (1_000_000, "heightfield"), (10_000_000, "soup"), (17_000_000, "soup")])     # the last one: exchange words without the packed prefix length (n >= 2^24)
def test_full_size_builds(cpt, n, kind):
    if kind == "soup":
        mesh = random_soup(n, seed=12345)
    else:
        mesh = heightfield(int(round((n / 2) ** 0.5)), int(round((n / 2) ** 0.5)))
    boxes, merged = boxes_of_triangles(mesh["positions"], mesh["indices"])
    del mesh
    g_nodes, g_boxes, g_keys, g_par, stage = cpt.build_accel(boxes, merged)
    check_tree_properties(g_nodes, g_boxes, g_keys, boxes, merged)
    o_nodes, o_boxes, o_keys, o_par = oracle_build(boxes, merged)
    assert np.array_equal(g_keys, o_keys)
    assert np.array_equal(g_nodes, o_nodes)
    assert np.array_equal(g_par[1:], o_par[1:])
    assert np.array_equal(g_boxes.view(np.uint32), o_boxes.view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 4097, 300000, 3_000_000])
def test_onesweep_sort_variant_gives_the_same_tree(cpt, n):
    """The one-kernel-per-pass sort (digit histograms with the keys, decoupled look-back) against the
    oracle: same nodes and boxes as the default sort, bit for bit."""
    soup = random_soup(n, seed=5 + n)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    d_leaf = cpt.DeviceArray.from_host(boxes)
    d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    b = cpt.lib.cpt_builder_create()
    cpt.check(cpt.lib.cpt_builder_set_sort(b, 1))
    ms = np.zeros(1, np.float32)
    for _ in range(2):
        cpt.check(cpt.lib.cpt_builder_build(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, ms.ctypes.data, None))
    cpt.lib.cpt_builder_destroy(b)
    o_nodes, o_boxes, _, _ = oracle_build(boxes, merged)
    assert np.array_equal(d_nodes.to_host(), o_nodes)
    assert np.array_equal(d_boxes.to_host().view(np.uint32), o_boxes.view(np.uint32))


@pytest.mark.parametrize("limit", [0, 1, 5])
def test_hierarchy_nodes_that_do_not_fit_the_second_pass_climb_in_place(cpt, limit):
    """With the hand-over list of the first hierarchy pass cut short, most nodes that span workgroups take
    the in-kernel climb through memory, and mix with the ones that were handed over."""
    for n, kind in ((100000, "soup"), (70000, "heightfield")):
        soup = random_soup(n, seed=31) if kind == "soup" else heightfield(int((n / 2) ** 0.5), int((n / 2) ** 0.5), 3)
        boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
        n = len(boxes)
        d_leaf = cpt.DeviceArray.from_host(boxes)
        d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
        d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
        b = cpt.lib.cpt_builder_create()
        cpt.check(cpt.lib.cpt_builder_set_carrier_limit(b, limit))
        ms = np.zeros(1, np.float32)
        cpt.check(cpt.lib.cpt_builder_build(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, ms.ctypes.data, None))
        cpt.lib.cpt_builder_destroy(b)
        o_nodes, o_boxes, _, _ = oracle_build(boxes, merged)
        assert np.array_equal(d_nodes.to_host(), o_nodes)
        assert np.array_equal(d_boxes.to_host().view(np.uint32), o_boxes.view(np.uint32))


def test_reusable_builder_is_deterministic(cpt):
    soup = random_soup(200000, seed=77)
    boxes, merged = boxes_of_triangles(soup["positions"], soup["indices"])
    n = len(boxes)
    d_leaf = cpt.DeviceArray.from_host(boxes)
    d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    b = cpt.lib.cpt_builder_create()
    first = None
    for _ in range(3):
        ms = np.zeros(1, np.float32)
        cpt.check(cpt.lib.cpt_builder_build(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, ms.ctypes.data, None))
        got = (d_nodes.to_host(), d_boxes.to_host().view(np.uint32))
        if first is None:
            first = got
        assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], first[1]) and ms[0] > 0
    cpt.lib.cpt_builder_destroy(b)
    o_nodes, o_boxes, _, _ = oracle_build(boxes, merged)
    assert np.array_equal(first[0], o_nodes) and np.array_equal(first[1], o_boxes.view(np.uint32))


def records_from_arrays(nodes, boxes, n, top_level):
    """The traversal records of a finished tree, restated with numpy from the reference-format arrays: binary records
    (children's boxes and references), two-level records and the root record.  A two-level record holds a cut through the
    subtree below its node: (right child, left child) to begin with, then twice the internal member with the largest
    box surface (fp32, dx*dy + dy*dz + dz*dx; the earlier member on equality) is replaced by its children, right one
    first; members in that order in slots 0..3, unused slots with reference 0xffffffff and a zero box."""
    LEAF, NONE = np.uint32(0x80000000), np.uint32(0xFFFFFFFF)
    ids = np.arange(2 * n - 1, dtype=np.uint32)
    leaf_ref = LEAF | (nodes[:, 0] if top_level else ids - np.uint32(n - 1))
    ref = np.where(ids >= n - 1, leaf_ref, ids).astype(np.uint32)
    box6 = boxes[:, [0, 1, 2, 4, 5, 6]].view(np.uint32)                     # min.xyz, max.xyz
    lc, rc = nodes[: n - 1, 0], nodes[: n - 1, 1]

    def half(a, ra, b, rb):
        out = np.zeros((len(a), 16), np.uint32)
        out[:, 0:6], out[:, 6:12], out[:, 12], out[:, 13] = a, b, ra, rb
        return out

    binary = half(box6[lc], ref[lc], box6[rc], ref[rc])
    d = boxes[:, 4:7] - boxes[:, 0:3]                                       # float32 throughout, as on the device
    surface = (d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2]) + d[:, 2] * d[:, 0]
    all_lc, all_rc = nodes[:, 0].astype(np.int64), nodes[:, 1].astype(np.int64)
    internal = np.arange(2 * n - 1) < n - 1
    m = n - 1
    cut = np.full((m, 4), -1, np.int64)
    cut[:, 0], cut[:, 1] = rc, lc
    rows = np.arange(m)
    for _ in range(2):
        member = np.maximum(cut, 0)
        s_of = np.where((cut >= 0) & internal[member], surface[member], np.float32(-1.0))
        best = s_of.argmax(axis=1)                                          # first of equal maxima
        can = s_of[rows, best] > -1.0
        opened = cut[rows, best]
        new = cut.copy()
        for col in range(4):
            new[:, col] = np.where(col < best, cut[:, col],
                          np.where(col == best, all_rc[np.maximum(opened, 0)],
                          np.where(col == best + 1, all_lc[np.maximum(opened, 0)], cut[:, max(col - 1, 0)])))
        cut = np.where(can[:, None], new, cut)
    wide = np.zeros((m, 32), np.uint32)
    for slot in range(4):
        used = cut[:, slot] >= 0
        member = np.maximum(cut[:, slot], 0)
        base = 16 * (slot // 2) + 6 * (slot % 2)
        wide[:, base:base + 6] = np.where(used[:, None], box6[member], 0)
        wide[:, 16 * (slot // 2) + 12 + slot % 2] = np.where(used, ref[member], NONE)
    wide[:, 14:16] = wide[:, 28:30]                                         # the second half's references once more: one 16-byte word has all four
    root = np.zeros(10, np.uint32)
    root[0:6] = box6[0]
    root[6] = (LEAF | (nodes[0, 0] if top_level else 0)) if n == 1 else 0
    root[7] = n
    return binary, wide, root


@pytest.mark.parametrize("n,kind,limit,top_level", [(1, "soup", 64, 0), (2, "soup", 64, 1), (3, "soup", 64, 0), (513, "soup", 64, 0), (5000, "soup", 64, 1),
                                                    (100000, "soup", 64, 0), (100000, "soup", 0, 0), (70000, "heightfield", 3, 1), (1_200_000, "heightfield", 64, 0),
                                                    (3_300_000, "soup", 64, 0), (3_300_000, "heightfield", 2, 0),      # above 3M keys: groups of 1024 leaves
                                                    (20000, "stacked", 64, 0), (700, "stacked", 1, 1)])                  # many equal keys: chains instead of balanced subtrees
def test_traversal_records_written_by_the_build(cpt, n, kind, limit, top_level):
    """cpt_builder_build_records: the binary and the two-level records that the hierarchy passes write themselves (from
    LDS for nodes finished inside a workgroup, from memory for the others, also when the hand-over list is cut short)
    equal a numpy packing of the tree's reference-format arrays, bit for bit; so does the root record."""
    if kind == "stacked":
        # a few hundred distinct triangles, each repeated many times: equal Morton codes, told apart by the index bits
        # of the key only, so the tree has long runs of nodes whose boxes are equal (equal surfaces: the cut's tie rule)
        mesh = random_soup(max(n // 64, 4), seed=5 + n)
        mesh["indices"] = np.tile(mesh["indices"].reshape(-1, 3), (64, 1)).reshape(-1)[: 3 * n]
    else:
        mesh = random_soup(n, seed=91 + n) if kind == "soup" else heightfield(int((n / 2) ** 0.5), int((n / 2) ** 0.5), 5)
    boxes, merged = boxes_of_triangles(mesh["positions"], mesh["indices"])
    n = len(boxes)
    d_leaf = cpt.DeviceArray.from_host(boxes)
    d_boxes = cpt.DeviceArray((2 * n - 1, 8), np.float32)
    d_nodes = cpt.DeviceArray((2 * n - 1, 2), np.uint32)
    slots = max(n - 1, 1)
    d_binary = cpt.DeviceArray.from_host(np.full((slots, 16), 0xDEADBEEF, np.uint32))
    d_wide = cpt.DeviceArray.from_host(np.full((slots, 32), 0xDEADBEEF, np.uint32))
    d_root = cpt.DeviceArray.from_host(np.full(10, 0xDEADBEEF, np.uint32))
    b = cpt.lib.cpt_builder_create()
    cpt.check(cpt.lib.cpt_builder_set_carrier_limit(b, limit))
    for _ in range(2):                                                     # the second build reuses the scratch
        cpt.check(cpt.lib.cpt_builder_build_records(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, top_level, d_binary.ptr, d_wide.ptr, d_root.ptr))
    cpt.check(cpt.lib.cpt_synchronize())
    cpt.lib.cpt_builder_destroy(b)
    nodes, gboxes = d_nodes.to_host(), d_boxes.to_host()
    o_nodes, o_boxes, _, _ = oracle_build(boxes, merged)
    assert np.array_equal(nodes, o_nodes) and np.array_equal(gboxes.view(np.uint32), o_boxes.view(np.uint32))
    binary, wide, root = records_from_arrays(nodes, gboxes, n, top_level)
    g_root = d_root.to_host()
    assert np.array_equal(g_root[:8], root[:8]), (g_root, root)             # flags 0: every coordinate is moderate
    # either kind of record alone
    b = cpt.lib.cpt_builder_create()
    d_wide_only = cpt.DeviceArray.from_host(np.full((slots, 32), 0xDEADBEEF, np.uint32))
    cpt.check(cpt.lib.cpt_builder_build_records(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, top_level, None, d_wide_only.ptr, d_root.ptr))
    assert cpt.lib.cpt_builder_build_records(b, d_nodes.ptr, d_boxes.ptr, d_leaf.ptr, merged, n, top_level, None, None, d_root.ptr) < 0
    cpt.check(cpt.lib.cpt_synchronize())
    cpt.lib.cpt_builder_destroy(b)
    if n > 1:
        assert np.array_equal(d_wide_only.to_host()[:, :30], wide[:, :30])
        g_binary, g_wide = d_binary.to_host(), d_wide.to_host()
        assert np.array_equal(g_binary[:, :14], binary[:, :14]), int((g_binary[:, :14] != binary[:, :14]).any(axis=1).sum())
        assert np.array_equal(g_wide[:, :30], wide[:, :30]), int((g_wide[:, :30] != wide[:, :30]).any(axis=1).sum())
