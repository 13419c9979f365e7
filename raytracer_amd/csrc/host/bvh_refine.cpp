// Refinement of a finished triangle BVH: an opt-in, NON-PARITY acceleration structure.
//
// The reference stops subdividing a node when `15 + cost >= 2 * num`
// (/root/reference/src/bvh.rs:169).  Over clusters of overlapping shapes (foliage quads,
// stacked sheets) no candidate plane brings the summed child areas down, so such clusters
// stay leaves of dozens to hundreds of triangles, and a ray that enters one tests them all.
// rth_refine_bvh_triangles keeps the reference's tree as it is and goes on splitting every
// leaf above `max_leaf` triangles without that stop rule: binned surface-area heuristic
// over the centroids (16 bins per axis), restricted to planes that leave each side at
// least an eighth of the triangles, the median of the widest axis when no plane
// qualifies.  Triangles are permuted inside their leaf's range only, so every node of the
// reference's tree still covers the same set; `order` records where each came from.
//
// What changes and what does not: ray/box and ray/triangle arithmetic are untouched, so a
// ray's nearest hit DISTANCE is the same number unless the ray grazes the box of one of the
// new, smaller leaves (the slab test is not conservative: a hit on the very edge of a
// triangle can lie in a box the ray numerically misses).  Which of several triangles at
// exactly that distance is reported can differ (the reference keeps the lowest index inside a
// leaf and the first leaf visited; both depend on the tree), and primitive ids are
// positions after this permutation: map them back through `order` to compare.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "rt_host.h"
#include "vec3.h"

namespace rth {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr int kBins = 16;

struct Item {
    V3 c, lo, hi;
};

// shapes.rs:331-347: the box of a, a + e1, a + e2, as the reference's builder sees the triangle
Item item_of(const RtTriangle& t) {
    const V3 a = v3(t.a), e1 = v3(t.e1), e2 = v3(t.e2);
    const V3 b = add(a, e1), c = add(a, e2);
    return Item{add(a, div(add(e1, e2), 3.0f)), vmin(vmin(a, b), c), vmax(vmax(a, b), c)};
}

float half_area(V3 lo, V3 hi) {
    const V3 e = sub(hi, lo);
    return (e.x * e.y + e.x * e.z) + e.y * e.z;
}

struct Box {
    V3 lo = v3(kInf, kInf, kInf), hi = v3(-kInf, -kInf, -kInf);
    uint32_t n = 0;
    void add_item(const Item& it) {
        lo = vmin(lo, it.lo);
        hi = vmax(hi, it.hi);
        ++n;
    }
    void add_box(const Box& b) {
        lo = vmin(lo, b.lo);
        hi = vmax(hi, b.hi);
        n += b.n;
    }
};

class Refiner {
   public:
    Refiner(std::vector<RtNode>& nodes, RtTriangle* tris, uint32_t n, uint32_t* order, uint32_t max_leaf)
        : nodes_(nodes), tris_(tris), order_(order), items_(n), left_(n), max_leaf_(std::max(1u, max_leaf)) {
        for (uint32_t i = 0; i < n; ++i) items_[i] = item_of(tris[i]);
    }

    void run() {
        const size_t reference_nodes = nodes_.size();
        for (size_t i = 0; i < reference_nodes; ++i)
            if (nodes_[i].b & RT_LEAF_BIT) split(static_cast<uint32_t>(i));
    }

   private:
    int bin_of(float c, float lo, float scale) const {
        const int b = static_cast<int>((c - lo) * scale);
        return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
    }

    void split(uint32_t index) {
        // (children are appended while this runs: nodes_[index] must be re-read, never held by reference)
        const uint32_t first = nodes_[index].a, num = nodes_[index].b & ~RT_LEAF_BIT;
        if (num <= max_leaf_) return;
        const uint32_t end = first + num;
        V3 clo = v3(kInf, kInf, kInf), chi = v3(-kInf, -kInf, -kInf);
        for (uint32_t i = first; i < end; ++i) {
            clo = vmin(clo, items_[i].c);
            chi = vmax(chi, items_[i].c);
        }
        const uint32_t least = std::max(1u, num / 8u);
        float best_cost = kInf;
        int best_axis = -1, best_k = 0;
        float best_lo = 0.0f, best_scale = 0.0f;
        for (int axis = 0; axis < 3; ++axis) {
            const float lo = clo[axis], extent = chi[axis] - lo;
            if (!(extent > 0.0f) || !std::isfinite(extent)) continue;
            const float scale = static_cast<float>(kBins) / extent;
            Box bins[kBins];
            for (uint32_t i = first; i < end; ++i) bins[bin_of(items_[i].c[axis], lo, scale)].add_item(items_[i]);
            Box suffix[kBins + 1];
            for (int k = kBins - 1; k >= 0; --k) {
                suffix[k] = suffix[k + 1];
                suffix[k].add_box(bins[k]);
            }
            Box prefix;
            for (int k = 1; k < kBins; ++k) {  // plane k: bins [0, k) left, [k, 16) right
                prefix.add_box(bins[k - 1]);
                const Box& right = suffix[k];
                if (prefix.n < least || right.n < least) continue;
                const float cost = half_area(prefix.lo, prefix.hi) * static_cast<float>(prefix.n) +
                                   half_area(right.lo, right.hi) * static_cast<float>(right.n);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_k = k;
                    best_lo = lo;
                    best_scale = scale;
                }
            }
        }
        uint32_t n_left = 0;
        if (best_axis >= 0) {
            for (uint32_t i = first; i < end; ++i) {
                left_[i] = bin_of(items_[i].c[best_axis], best_lo, best_scale) < best_k;
                n_left += left_[i];
            }
        } else {
            // no plane qualifies: the median of the widest centroid axis (equal centroids: by position)
            int axis = 0;
            const V3 e = sub(chi, clo);
            if (e.y > e[axis]) axis = 1;
            if (e.z > e[axis]) axis = 2;
            std::vector<uint32_t> idx(num);
            for (uint32_t i = 0; i < num; ++i) idx[i] = first + i;
            n_left = num / 2;
            std::nth_element(idx.begin(), idx.begin() + n_left, idx.end(), [&](uint32_t a, uint32_t b) {
                const float ca = items_[a].c[axis], cb = items_[b].c[axis];
                return ca < cb || (ca == cb && a < b);
            });
            for (uint32_t i = first; i < end; ++i) left_[i] = 0;
            for (uint32_t i = 0; i < n_left; ++i) left_[idx[i]] = 1;
        }
        // partition in place (triangles, their boxes and their provenance together)
        uint32_t i = first, j = end;
        while (true) {
            while (i < j && left_[i]) ++i;
            while (i < j && !left_[j - 1]) --j;
            if (i >= j) break;
            std::swap(tris_[i], tris_[j - 1]);
            std::swap(items_[i], items_[j - 1]);
            std::swap(order_[i], order_[j - 1]);
            std::swap(left_[i], left_[j - 1]);
        }
        const uint32_t mid = first + n_left;
        const uint32_t range[2][2] = {{first, mid}, {mid, end}};
        uint32_t child[2];
        for (int c = 0; c < 2; ++c) {
            Box b;
            for (uint32_t t = range[c][0]; t < range[c][1]; ++t) b.add_item(items_[t]);
            RtNode n;
            store(n.min, b.lo);
            store(n.max, b.hi);
            n.a = range[c][0];
            n.b = (range[c][1] - range[c][0]) | RT_LEAF_BIT;
            child[c] = static_cast<uint32_t>(nodes_.size());
            nodes_.push_back(n);
        }
        nodes_[index].a = child[0];
        nodes_[index].b = child[1];
        split(child[0]);  // (depth is logarithmic: each side holds at least an eighth, or half)
        split(child[1]);
    }

    std::vector<RtNode>& nodes_;
    RtTriangle* tris_;
    uint32_t* order_;
    std::vector<Item> items_;
    std::vector<uint8_t> left_;
    uint32_t max_leaf_;
};

}  // namespace
}  // namespace rth

extern "C" RtNode* rth_refine_bvh_triangles(const RtNode* nodes, uint32_t n_nodes, RtTriangle* triangles, uint32_t n_triangles,
                                            uint32_t max_leaf, uint32_t* order, uint32_t* n_nodes_out) {
    if (!nodes || n_nodes == 0 || (n_triangles && !triangles) || !order || !n_nodes_out) return nullptr;
    try {
        for (uint32_t i = 0; i < n_nodes; ++i) {
            if (!(nodes[i].b & RT_LEAF_BIT)) continue;
            const uint64_t first = nodes[i].a, count = nodes[i].b & ~RT_LEAF_BIT;
            if (first + count > n_triangles) return nullptr;  // not a description of these triangles
        }
        for (uint32_t i = 0; i < n_triangles; ++i) order[i] = i;
        std::vector<RtNode> out(nodes, nodes + n_nodes);
        rth::Refiner(out, triangles, n_triangles, order, max_leaf).run();
        if (out.size() >= 0x80000000ull) return nullptr;
        RtNode* result = static_cast<RtNode*>(std::malloc(out.size() * sizeof(RtNode)));
        if (!result) return nullptr;
        std::memcpy(result, out.data(), out.size() * sizeof(RtNode));
        *n_nodes_out = static_cast<uint32_t>(out.size());
        return result;
    } catch (...) {
        return nullptr;
    }
}
