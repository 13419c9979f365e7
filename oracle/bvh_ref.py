"""Independent restatement, in numpy binary32 scalars, of the reference's BVH
builder (/root/reference/src/bvh.rs:46-224) and of Rust core's
`Iterator::partition_in_place`.

TEST INFRASTRUCTURE ONLY (see oracle/rt_oracle.c).  It exists so that the C++
builder of the product (raytracer_amd/csrc/host/bvh_build.cpp) can be compared,
node for node and swap for swap, with a second implementation written
separately from the same source lines.  Slow on purpose (pure Python loops):
use it on a few thousand shapes at most.

Parity unpinned by the reference: the Rust crate cannot be built here, so both
implementations are transcriptions; they pin each other, not the original.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
INF = f32(np.inf)
LEAF_BIT = 0x80000000


def _fmin(a, b):  # f32::min: a NaN operand is ignored
    return np.fmin(a, b)


def _fmax(a, b):
    return np.fmax(a, b)


def triangle_bounds(a, e1, e2):
    """shapes.rs:331-347 — centroid, min, max (b and c rebuilt as a+e1, a+e2)."""
    a, e1, e2 = (np.asarray(v, f32) for v in (a, e1, e2))
    b, c = a + e1, a + e2
    centroid = a + (e1 + e2) / f32(3.0)
    return centroid, _fmin(_fmin(a, b), c), _fmax(_fmax(a, b), c)


def sphere_bounds(center, radius):
    """shapes.rs:105-115"""
    c = np.asarray(center, f32)
    r = np.full(3, radius, f32)
    return c, c - r, c + r


def plane_bounds(point):
    """shapes.rs:171-181"""
    return np.asarray(point, f32), np.full(3, -INF, f32), np.full(3, INF, f32)


def surface_area(lo, hi):
    """bvh.rs:54-55,129-132: 2*((ex*ey + ex*ez) + ey*ez) in binary32."""
    with np.errstate(all="ignore"):
        e = (hi - lo).astype(f32)
        return f32(2.0) * ((e[0] * e[1] + e[0] * e[2]) + e[1] * e[2])


def partition_in_place(items: list, pred) -> int:
    """Rust core `Iterator::partition_in_place`: repeatedly find the first
    element for which `pred` is false (from the front) and swap it with the last
    for which it is true (from the back).  Returns the number of trues."""
    i, j, true_count = 0, len(items), 0
    while True:
        while i < j and pred(items[i]):  # find(|x| !pred(x)) consuming the trues
            true_count += 1
            i += 1
        if i == j:
            break
        head = i
        i += 1
        found = False
        while j > i:  # rfind(pred)
            j -= 1
            if pred(items[j]):
                found = True
                break
        if not found:
            break
        items[head], items[j] = items[j], items[head]
        true_count += 1
    return true_count


def build(bounds: list):
    """bounds: list of (centroid, lo, hi) per shape, in input order.
    Returns (nodes, order): nodes = list of dicts {min, max, a, b} in push order
    with the encoding of RtNode; order = permutation applied to the shapes
    (order[k] = original index of the shape now at position k)."""
    n = len(bounds)
    order = list(range(n))
    nodes = []

    def fold(first, end):
        lo, hi = np.full(3, INF, f32), np.full(3, -INF, f32)
        for k in range(first, end):
            _, blo, bhi = bounds[order[k]]
            lo, hi = _fmin(lo, blo), _fmax(hi, bhi)
        return lo, hi

    def get_split(node, node_sa):
        first, count = node["a"], node["b"] & ~LEAF_BIT
        best = (0, f32(0), INF, [f32(np.nan), f32(np.nan)])
        with np.errstate(all="ignore"):
            per_bin = ((node["max"] - node["min"]) / f32(16.0)).astype(f32)
            for k in range(1, 16):
                offsets = (node["min"] + per_bin * f32(k)).astype(f32)
                for axis in range(3):
                    cand = offsets[axis]
                    costs, sas = [], []
                    for child in range(2):
                        lo, hi = np.full(3, INF, f32), np.full(3, -INF, f32)
                        num = 0
                        for idx in range(first, first + count):
                            c, blo, bhi = bounds[order[idx]]
                            take = (c[axis] < cand) if child == 0 else (c[axis] >= cand)
                            if take:
                                lo, hi = _fmin(lo, blo), _fmax(hi, bhi)
                                num += 1
                        if num == 0:
                            costs.append(INF)
                            sas.append(f32(np.nan))
                        else:
                            sa = surface_area(lo, hi)
                            costs.append(f32(f32(f32(sa / node_sa) * f32(2.0)) * f32(num)))
                            sas.append(sa)
                    total = f32(costs[0] + costs[1])
                    if total < best[2]:
                        best = (axis, cand, total, sas)
        return best

    def subdivide(index, node_sa):
        node = nodes[index]
        first, count = node["a"], node["b"] & ~LEAF_BIT
        axis, split, cost, child_sa = get_split(node, node_sa)
        with np.errstate(all="ignore"):
            if f32(f32(15.0) + cost) >= f32(f32(2.0) * f32(count)):
                return
        window = order[first:first + count]
        mid = first + partition_in_place(window, lambda s: bounds[s][0][axis] < split)
        order[first:first + count] = window
        children = []
        for child, (cf, ce) in enumerate(((first, mid), (mid, first + count))):
            lo, hi = fold(cf, ce)
            children.append(len(nodes))
            nodes.append({"min": lo, "max": hi, "a": cf, "b": (ce - cf) | LEAF_BIT})
            subdivide(children[-1], child_sa[child])
        nodes[index]["a"], nodes[index]["b"] = children[0], children[1]

    lo, hi = fold(0, n)
    nodes.append({"min": lo, "max": hi, "a": 0, "b": n | LEAF_BIT})
    subdivide(0, surface_area(lo, hi))
    return nodes, order
