"""The opt-in refined triangle tree (rth_refine_bvh_triangles, Scene.refined): a valid
description of the same triangles whose leaves hold at most `max_leaf` of them.  It is not the
reference's tree — the stop rule of /root/reference/src/bvh.rs:169 is what it removes — so
what is checked is what the header promises: the triangles are permuted inside the reference's
leaves only, every hit distance is the same number as in the reference's tree (the oracle
traverses both), and ids map back through `order` wherever no two triangles tie."""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F

from . import scenes

LEAF = np.uint32(F.RT_LEAF_BIT)


def leaves(nodes):
    is_leaf = (nodes["b"] & LEAF) != 0
    return nodes["a"][is_leaf], nodes["b"][is_leaf] & ~LEAF


def reachable_leaves(nodes):
    out, stack = [], [0]
    while stack:
        n = nodes[stack.pop()]
        if n["b"] & LEAF:
            out.append((int(n["a"]), int(n["b"] & ~LEAF)))
        else:
            assert n["a"] < len(nodes) and n["b"] < len(nodes)
            stack += [int(n["a"]), int(n["b"])]
    return out


def tri_bounds(t):
    a, b, c = t["a"], t["a"] + t["e1"], t["a"] + t["e2"]
    return np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)


@pytest.mark.parametrize("max_leaf", [1, 4, 8])
def test_refined_tree_describes_the_same_triangles(asset_root, max_leaf):
    s = scenes.tree(asset_root)
    first, count = leaves(s.triangle_nodes)
    assert count.max() > 8  # the stop rule leaves long leaves on this scene
    r, order = s.refined(max_leaf)
    # a permutation, and only inside the reference's leaves
    assert np.array_equal(np.sort(order), np.arange(s.triangles.size))
    assert np.array_equal(r.triangles, s.triangles[order])
    leaf_of = np.repeat(np.arange(first.size), count)  # the reference leaf each position belongs to
    position_leaf = np.empty(s.triangles.size, dtype=np.int64)
    position_leaf[np.concatenate([np.arange(f, f + c) for f, c in zip(first, count)])] = leaf_of
    assert np.array_equal(position_leaf[order], position_leaf)
    # the reference's nodes keep their boxes; leaves tile the triangle array; boxes hold their triangles
    n0 = s.triangle_nodes.size
    assert np.array_equal(r.triangle_nodes["min"][:n0], s.triangle_nodes["min"])
    assert np.array_equal(r.triangle_nodes["max"][:n0], s.triangle_nodes["max"])
    got = sorted(reachable_leaves(r.triangle_nodes))
    assert got[0][0] == 0 and all(a + c == b for (a, c), (b, _) in zip(got, got[1:])) and sum(got[-1]) == s.triangles.size
    assert max(c for _, c in got) <= max_leaf
    lo, hi = tri_bounds(r.triangles)
    nodes = r.triangle_nodes
    leaf_nodes = nodes[(nodes["b"] & LEAF) != 0]
    lf, lc = leaf_nodes["a"].astype(np.int64), (leaf_nodes["b"] & ~LEAF).astype(np.int64)
    keep = lc > 0
    # per leaf: the fold of its triangles' boxes (segments of the triangle array), against the node's box
    starts = lf[keep]
    order_by_start = np.argsort(starts)
    starts, boxes = starts[order_by_start], leaf_nodes[keep][order_by_start]
    assert np.array_equal(starts, np.array([f for f, c in got if c > 0]))
    assert np.all(boxes["min"] <= np.minimum.reduceat(lo, starts, axis=0))
    assert np.all(boxes["max"] >= np.maximum.reduceat(hi, starts, axis=0))
    assert scenes.bvh_depth(r.triangle_nodes) < 64


def test_hit_distances_are_the_reference_trees(asset_root):
    s = scenes.tree(asset_root, w=40, h=40)
    r, order = s.refined(4)
    a, b = O.trace_primary(s, s.params, 0, True), O.trace_primary(r, r.params, 0, True)
    assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    assert np.array_equal(a["kind"], b["kind"])
    tri = b["kind"] == F.RT_SHAPE_TRIANGLE
    mapped = b["prim"].copy()
    mapped[tri] = order[b["prim"][tri]]
    assert (mapped != a["prim"]).mean() < 0.01  # ids agree except where two triangles tie on the distance
    assert tri.mean() > 0.2


def test_degenerate_inputs():
    s = scenes.stacked_duplicates(n=40)  # forty copies of one triangle: no plane separates them
    r, order = s.refined(2)
    assert max(c for _, c in reachable_leaves(r.triangle_nodes)) <= 2
    assert np.array_equal(np.sort(order), np.arange(40))
    a, b = O.trace_primary(s, s.params), O.trace_primary(r, r.params)
    assert np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))
    e = scenes.empty()
    r, order = e.refined(8)
    assert order.size == 0 and r.triangle_nodes.size == e.triangle_nodes.size
    lib = F.host_lib()
    assert not lib.rth_refine_bvh_triangles(None, 0, None, 0, 8, None, None)
