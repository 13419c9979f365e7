// Binned surface-area-heuristic BVH builder: the host-side construction of
// /root/reference/src/bvh.rs:46-224, reproduced operation for operation.
//
// Primitive ids handed to the GPU are indices into the shape arrays *after*
// this build has permuted them (/root/reference/src/config.rs:113-134), and the
// traversal's tie-breaking depends on node order, so three things have to match
// the reference exactly: the floating-point values that pick a split, the
// swap sequence of the in-place partition, and the order nodes are appended.
//
//   * split candidates: for k = 1..15 (outer) and axis x,y,z (inner) the plane
//     min[axis] + (extent[axis]/16)*k; cost = ((SA_lt/SA_node)*2)*n_lt +
//     ((SA_ge/SA_node)*2)*n_ge; an empty side costs +inf; a candidate replaces
//     the best only on strict `<` (bvh.rs:99-146);
//   * a node stays a leaf when 15 + cost >= 2*n (bvh.rs:169);
//   * shapes are partitioned by centroid[axis] < split with the algorithm of
//     Rust's `Iterator::partition_in_place` (first `false` from the front is
//     swapped with the last `true` from the back);
//   * child 0 is appended and fully subdivided before child 1 is appended
//     (bvh.rs:198-221), giving a depth-first pre-order array.
//
// Per-shape centroid/min/max are computed once and permuted alongside the
// shapes; they are pure functions of the shape, so this changes nothing.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include "rt_host.h"
#include "vec3.h"

namespace rth {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kNan = std::numeric_limits<float>::quiet_NaN();

struct Bounds {
    V3 centroid, lo, hi;
};

// shapes.rs:105-115
Bounds bounds_of(const RtSphere& s) {
    const V3 c = v3(s.center);
    const V3 r = v3(s.radius, s.radius, s.radius);
    return Bounds{c, sub(c, r), add(c, r)};
}
// shapes.rs:171-181
Bounds bounds_of(const RtPlane& p) {
    return Bounds{v3(p.point), v3(-kInf, -kInf, -kInf), v3(kInf, kInf, kInf)};
}
// shapes.rs:331-347 — b and c are rebuilt as a+e1, a+e2, not the file's vertices.
Bounds bounds_of(const RtTriangle& t) {
    const V3 a = v3(t.a), e1 = v3(t.e1), e2 = v3(t.e2);
    const V3 b = add(a, e1), c = add(a, e2);
    return Bounds{add(a, div(add(e1, e2), 3.0f)), vmin(vmin(a, b), c), vmax(vmax(a, b), c)};
}

// bvh.rs:54-55 / 129-132
float surface_area(V3 lo, V3 hi) {
    const V3 e = sub(hi, lo);
    return 2.0f * ((e.x * e.y + e.x * e.z) + e.y * e.z);
}

struct Split {
    int axis = 0;
    float value = 0.0f;
    float cost = kInf;
    float sa[2] = {kNan, kNan};
};

template <class Shape>
class Builder {
   public:
    Builder(Shape* shapes, uint32_t n) : shapes_(shapes), n_(n), b_(n) {
        for (uint32_t i = 0; i < n; ++i) b_[i] = bounds_of(shapes[i]);
    }

    std::vector<RtNode> build() {
        V3 lo, hi;
        fold_bounds(0, n_, lo, hi);
        nodes_.push_back(leaf(0, n_, lo, hi));
        subdivide(0, surface_area(lo, hi));
        return std::move(nodes_);
    }

   private:
    static RtNode leaf(uint32_t first, uint32_t end, V3 lo, V3 hi) {
        RtNode n;
        store(n.min, lo);
        store(n.max, hi);
        n.a = first;
        n.b = (end - first) | RT_LEAF_BIT;
        return n;
    }

    // bvh.rs:70-85
    void fold_bounds(uint32_t first, uint32_t end, V3& lo, V3& hi) const {
        lo = v3(kInf, kInf, kInf);
        hi = v3(-kInf, -kInf, -kInf);
        for (uint32_t i = first; i < end; ++i) {
            lo = vmin(lo, b_[i].lo);
            hi = vmax(hi, b_[i].hi);
        }
    }

    // bvh.rs:87-150
    Split get_split(const RtNode& node, float node_sa) const {
        const uint32_t first = node.a, end = first + (node.b & ~RT_LEAF_BIT);
        Split best;
        const V3 lo = v3(node.min), hi = v3(node.max);
        const V3 per_bin = div(sub(hi, lo), 16.0f);
        for (int k = 1; k < 16; ++k) {
            const V3 offsets = add(lo, mul(per_bin, static_cast<float>(k)));
            for (int axis = 0; axis < 3; ++axis) {
                const float candidate = offsets[axis];
                V3 clo[2] = {v3(kInf, kInf, kInf), v3(kInf, kInf, kInf)};
                V3 chi[2] = {v3(-kInf, -kInf, -kInf), v3(-kInf, -kInf, -kInf)};
                uint32_t num[2] = {0, 0};
                for (uint32_t i = first; i < end; ++i) {
                    const float c = b_[i].centroid[axis];
                    // child 0 takes `c < candidate`, child 1 takes `c >= candidate`;
                    // a NaN on either side satisfies neither.
                    const bool lt = c < candidate, ge = c >= candidate;
                    if (lt) {
                        clo[0] = vmin(clo[0], b_[i].lo);
                        chi[0] = vmax(chi[0], b_[i].hi);
                        ++num[0];
                    } else if (ge) {
                        clo[1] = vmin(clo[1], b_[i].lo);
                        chi[1] = vmax(chi[1], b_[i].hi);
                        ++num[1];
                    }
                }
                float cost[2], sa[2];
                for (int c = 0; c < 2; ++c) {
                    if (num[c] == 0) {
                        cost[c] = kInf;
                        sa[c] = kNan;
                    } else {
                        sa[c] = surface_area(clo[c], chi[c]);
                        cost[c] = ((sa[c] / node_sa) * 2.0f) * static_cast<float>(num[c]);
                    }
                }
                const float total = cost[0] + cost[1];
                if (total < best.cost) {
                    best.axis = axis;
                    best.value = candidate;
                    best.cost = total;
                    best.sa[0] = sa[0];
                    best.sa[1] = sa[1];
                }
            }
        }
        return best;
    }

    // Rust core `Iterator::partition_in_place` over shapes[first..end].
    uint32_t partition(uint32_t first, uint32_t end, int axis, float split) {
        auto pred = [&](uint32_t i) { return b_[i].centroid[axis] < split; };
        uint32_t i = first, j = end, true_count = 0;
        for (;;) {
            // find(): consume from the front up to and including the first `false`
            while (i < j && pred(i)) {
                ++true_count;
                ++i;
            }
            if (i == j) break;
            const uint32_t head = i++;
            // rfind(): consume from the back up to and including the last `true`
            bool found = false;
            while (j > i) {
                --j;
                if (pred(j)) {
                    found = true;
                    break;
                }
            }
            if (!found) break;
            std::swap(shapes_[head], shapes_[j]);
            std::swap(b_[head], b_[j]);
            ++true_count;
        }
        return true_count;
    }

    // bvh.rs:152-224
    void subdivide(uint32_t index, float node_sa) {
        const uint32_t first = nodes_[index].a;
        const uint32_t num = nodes_[index].b & ~RT_LEAF_BIT;
        const uint32_t end = first + num;

        const Split s = get_split(nodes_[index], node_sa);
        if (15.0f + s.cost >= 2.0f * static_cast<float>(num)) return;

        const uint32_t mid = first + partition(first, end, s.axis, s.value);
        const uint32_t range[2][2] = {{first, mid}, {mid, end}};
        uint32_t child[2];
        for (int c = 0; c < 2; ++c) {
            V3 lo, hi;
            fold_bounds(range[c][0], range[c][1], lo, hi);
            child[c] = static_cast<uint32_t>(nodes_.size());
            nodes_.push_back(leaf(range[c][0], range[c][1], lo, hi));
            subdivide(child[c], s.sa[c]);
        }
        nodes_[index].a = child[0];
        nodes_[index].b = child[1];
    }

    Shape* shapes_;
    uint32_t n_;
    std::vector<Bounds> b_;
    std::vector<RtNode> nodes_;
};

template <class Shape>
RtNode* build_c(Shape* shapes, uint32_t n, uint32_t* n_nodes) {
    std::vector<RtNode> nodes = Builder<Shape>(shapes, n).build();
    RtNode* out = static_cast<RtNode*>(std::malloc(nodes.size() * sizeof(RtNode)));
    if (!out) return nullptr;
    std::memcpy(out, nodes.data(), nodes.size() * sizeof(RtNode));
    if (n_nodes) *n_nodes = static_cast<uint32_t>(nodes.size());
    return out;
}

}  // namespace

std::vector<RtNode> build_bvh(RtSphere* s, uint32_t n) { return Builder<RtSphere>(s, n).build(); }
std::vector<RtNode> build_bvh(RtPlane* s, uint32_t n) { return Builder<RtPlane>(s, n).build(); }
std::vector<RtNode> build_bvh(RtTriangle* s, uint32_t n) { return Builder<RtTriangle>(s, n).build(); }

}  // namespace rth

extern "C" {
RtNode* rth_build_bvh_spheres(RtSphere* shapes, uint32_t n, uint32_t* n_nodes) {
    return rth::build_c(shapes, n, n_nodes);
}
RtNode* rth_build_bvh_planes(RtPlane* shapes, uint32_t n, uint32_t* n_nodes) {
    return rth::build_c(shapes, n, n_nodes);
}
RtNode* rth_build_bvh_triangles(RtTriangle* shapes, uint32_t n, uint32_t* n_nodes) {
    return rth::build_c(shapes, n, n_nodes);
}
void rth_free(void* p) { std::free(p); }
}
