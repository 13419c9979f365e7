// Binned surface-area-heuristic BVH construction on the GPU: builds, for the
// triangle family, exactly the tree /root/reference/src/bvh.rs:46-224 builds —
// the same splits, the same in-place permutation of the shapes, the same
// depth-first node order — level by level instead of by recursion.
//
// Why the result is the same although nothing here runs in the reference's
// order:
//   * a split candidate of bvh.rs:99-146 is a plane `min + (extent/16)*k` and a
//     shape goes left iff `centroid < plane`.  The fifteen planes of an axis are
//     non-decreasing in k (rounded multiply and add are monotone), so "left of
//     plane k" is "bin < k" for bin = number of planes <= centroid, found with
//     the same comparisons against the same rounded planes.  Counts and boxes
//     of both sides of all 45 candidates follow from 3 x 16 bins by merging;
//   * boxes are min/max folds and counts are integer sums: associative and
//     commutative, so atomics give the fold's value (a zero may come out with
//     the other sign than the sequential fold's, which nothing downstream
//     distinguishes: every use compares or subtracts);
//   * costs are evaluated per candidate with the reference's scalar formula
//     and the winner is the lowest cost, first in (k outer, axis inner) order;
//   * `Iterator::partition_in_place` swaps the j-th `false` found from the
//     front with the j-th `true` found from the back until the cursors meet:
//     the falses left of the final boundary in ascending order pair with the
//     trues right of it in descending order.  Ranks come from prefix sums;
//   * the array order of bvh.rs:198-221 (child 0 and its whole subtree, then
//     child 1) is depth-first pre-order, assigned from subtree sizes afterwards.
//
// Only finite coordinates are accepted (an infinite or NaN box makes NaN planes,
// for which "left of plane k" is no longer a threshold on a bin); the caller
// keeps the host builder for anything else.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "abi_guard.h"
#include "rt_mi355.h"

namespace rtb {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kMinSplit = 8;  // 15 + cost >= 2*n always holds for n <= 7 (cost >= 0): such a node is a leaf
constexpr int kBins = 16;
constexpr int kBinWords = 7;  // count, lo xyz, hi xyz
constexpr int kNodeBinWords = 3 * kBins * kBinWords;
constexpr int kScanBlock = 1024;  // positions per scan block (256 threads x 4)

struct BNode {
    float lo[3], hi[3];
    uint32_t first, count;
    float sa;          // surface area handed down by the parent's split (bvh.rs:205), own for the root
    uint32_t child0;   // pool index of child 0 (child 1 follows it); 0 = leaf
    uint32_t axis;     // of the split
    float split;
    uint32_t mid;      // first position of child 1
    uint32_t slot;     // row of the bin table while the node is being split, else kNone
};

struct ShapeBounds {  // shapes.rs:331-347 per triangle, by original index
    float4 centroid, lo, hi;
};

// Monotone map of binary32 onto unsigned integers (so that integer atomics
// order like the floats; -0 sorts below +0).
__device__ __forceinline__ uint32_t enc(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ float surface_area(const float* lo, const float* hi) {  // bvh.rs:54-55
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    return 2.0f * ((ex * ey + ex * ez) + ey * ez);
}

// ---- per-shape bounds, root box

// Centroid and box of one shape, /root/reference/src/shapes.rs:331-347 (triangle: b and c
// rebuilt as a+e1, a+e2) and :105-115 (sphere: centre -/+ radius); false if not all finite.
__device__ __forceinline__ bool shape_bounds(const RtTriangle& t, float* c, float* lo, float* hi) {
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        const float a = t.a[k], pb = t.a[k] + t.e1[k], pc = t.a[k] + t.e2[k];
        c[k] = a + (t.e1[k] + t.e2[k]) / 3.0f;
        lo[k] = fminf(fminf(a, pb), pc);
        hi[k] = fmaxf(fmaxf(a, pb), pc);
        finite = finite && isfinite(a) && isfinite(pb) && isfinite(pc) && isfinite(c[k]);
    }
    return finite;
}
__device__ __forceinline__ bool shape_bounds(const RtSphere& s, float* c, float* lo, float* hi) {
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        c[k] = s.center[k];
        lo[k] = s.center[k] - s.radius;
        hi[k] = s.center[k] + s.radius;
        finite = finite && isfinite(lo[k]) && isfinite(hi[k]) && isfinite(c[k]);
    }
    return finite;
}

template <class Shape>
__global__ void k_bounds(const Shape* shapes, uint32_t n, ShapeBounds* out, uint32_t* idx, uint32_t* root_box,
                         uint32_t* bad) {
    __shared__ uint32_t s_box[6];
    if (threadIdx.x < 3) s_box[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) s_box[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        ShapeBounds b;
        float lo[3], hi[3], c[3];
        const bool finite = shape_bounds(shapes[i], c, lo, hi);
        if (!finite) atomicAdd(bad, 1u);
        b.centroid = make_float4(c[0], c[1], c[2], 0.0f);
        b.lo = make_float4(lo[0], lo[1], lo[2], 0.0f);
        b.hi = make_float4(hi[0], hi[1], hi[2], 0.0f);
        out[i] = b;
        idx[i] = i;
        for (int k = 0; k < 3; ++k) {
            atomicMin(&s_box[k], enc(lo[k]));
            atomicMax(&s_box[3 + k], enc(hi[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&root_box[threadIdx.x], s_box[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&root_box[threadIdx.x], s_box[threadIdx.x]);
}

// counters: [0] nodes in the pool, [1] nodes being split at the next level
__global__ void k_root(const uint32_t* root_box, uint32_t n, BNode* pool, uint32_t* node_of, uint32_t* active, uint32_t* counters) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        BNode r{};
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = dec(root_box[k]);
            r.hi[k] = dec(root_box[3 + k]);
        }
        r.first = 0;
        r.count = n;
        r.sa = surface_area(r.lo, r.hi);
        r.child0 = 0;
        r.slot = n >= kMinSplit ? 0u : kNone;
        pool[0] = r;
        counters[0] = 1;
        counters[1] = n >= kMinSplit ? 1u : 0u;
        active[0] = 0;
    }
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) node_of[i] = n >= kMinSplit ? 0u : kNone;
}

// ---- binning: every position of a node being split adds its shape to one bin per axis

__global__ void k_init_bins(uint32_t* bins, uint32_t n_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t w = i % kBinWords;
    bins[i] = w == 0 ? 0u : (w < 4 ? 0xFFFFFFFFu : 0u);  // count, lo (min identity), hi (max identity)
}

__global__ void __launch_bounds__(256) k_bin(const ShapeBounds* sb, const uint32_t* idx, const uint32_t* node_of, const BNode* pool,
                                             uint32_t n, uint32_t* bins) {
    __shared__ uint32_t s_bins[kNodeBinWords];
    const uint32_t base = blockIdx.x * blockDim.x;
    const uint32_t i = base + threadIdx.x;
    const uint32_t last = min(base + blockDim.x, n) - 1u;
    // positions of a node are contiguous: equal ends mean the whole block is in one node
    const uint32_t node_first = node_of[base], node_last = node_of[last];
    const bool uniform = node_first == node_last && node_first != kNone;
    if (uniform) {
        for (uint32_t w = threadIdx.x; w < kNodeBinWords; w += blockDim.x) {
            const uint32_t k = w % kBinWords;
            s_bins[w] = k == 0 ? 0u : (k < 4 ? 0xFFFFFFFFu : 0u);
        }
        __syncthreads();
    }
    const uint32_t node = i < n ? node_of[i] : kNone;
    if (node != kNone) {
        const BNode nd = pool[node];
        const ShapeBounds b = sb[idx[i]];
        const float c[3] = {b.centroid.x, b.centroid.y, b.centroid.z};
        const uint32_t lo_e[3] = {enc(b.lo.x), enc(b.lo.y), enc(b.lo.z)};
        const uint32_t hi_e[3] = {enc(b.hi.x), enc(b.hi.y), enc(b.hi.z)};
        uint32_t* dst = uniform ? s_bins : bins + static_cast<size_t>(nd.slot) * kNodeBinWords;
        for (int axis = 0; axis < 3; ++axis) {
            const float per_bin = (nd.hi[axis] - nd.lo[axis]) / 16.0f;  // bvh.rs:96-98
            int bin = 0;
            for (int k = 1; k < kBins; ++k) bin += (nd.lo[axis] + per_bin * static_cast<float>(k)) <= c[axis] ? 1 : 0;
            uint32_t* w = dst + (axis * kBins + bin) * kBinWords;
            atomicAdd(&w[0], 1u);
            for (int k = 0; k < 3; ++k) {
                atomicMin(&w[1 + k], lo_e[k]);
                atomicMax(&w[4 + k], hi_e[k]);
            }
        }
    }
    if (uniform) {
        __syncthreads();
        uint32_t* g = bins + static_cast<size_t>(pool[node_first].slot) * kNodeBinWords;
        for (uint32_t w = threadIdx.x; w < kNodeBinWords; w += blockDim.x) {
            const uint32_t k = w % kBinWords;
            const uint32_t v = s_bins[w];
            if (k == 0) { if (v) atomicAdd(&g[w], v); }
            else if (k < 4) { if (v != 0xFFFFFFFFu) atomicMin(&g[w], v); }
            else { if (v != 0u) atomicMax(&g[w], v); }
        }
    }
}

// ---- split selection: one wave per node, lane = (k-1)*3 + axis = the candidate's place in bvh.rs:99-101's loop order

struct Side {
    uint32_t count;
    uint32_t lo[3], hi[3];  // encoded
};

__device__ __forceinline__ void merge_bin(Side& s, const uint32_t* w) {
    s.count += w[0];
    for (int k = 0; k < 3; ++k) {
        s.lo[k] = min(s.lo[k], w[1 + k]);
        s.hi[k] = max(s.hi[k], w[4 + k]);
    }
}

__global__ void __launch_bounds__(64) k_split(BNode* pool, const uint32_t* active, uint32_t n_active, const uint32_t* bins,
                                              uint32_t* next_active, uint32_t* counters, uint32_t pool_cap) {
    const uint32_t a = blockIdx.x;
    if (a >= n_active) return;
    const uint32_t node = active[a];
    BNode nd = pool[node];
    const uint32_t lane = threadIdx.x;
    const uint32_t* nb = bins + static_cast<size_t>(nd.slot) * kNodeBinWords;

    float total = __builtin_huge_valf();
    float sa[2] = {__builtin_nanf(""), __builtin_nanf("")};
    Side side[2];
    for (int c = 0; c < 2; ++c) {
        side[c].count = 0;
        for (int k = 0; k < 3; ++k) {
            side[c].lo[k] = 0xFFFFFFFFu;
            side[c].hi[k] = 0u;
        }
    }
    float plane = 0.0f;
    const uint32_t axis = lane % 3u, k = lane / 3u + 1u;
    if (lane < 45u) {
        const float per_bin = (nd.hi[axis] - nd.lo[axis]) / 16.0f;
        plane = nd.lo[axis] + per_bin * static_cast<float>(k);
        for (uint32_t b = 0; b < kBins; ++b) merge_bin(side[b < k ? 0 : 1], nb + (axis * kBins + b) * kBinWords);
        float cost[2];
        for (int c = 0; c < 2; ++c) {
            if (side[c].count == 0) {  // bvh.rs:124-128
                cost[c] = __builtin_huge_valf();
            } else {
                float lo[3], hi[3];
                for (int j = 0; j < 3; ++j) {
                    lo[j] = dec(side[c].lo[j]);
                    hi[j] = dec(side[c].hi[j]);
                }
                sa[c] = surface_area(lo, hi);
                cost[c] = ((sa[c] / nd.sa) * 2.0f) * static_cast<float>(side[c].count);  // bvh.rs:133-136
            }
        }
        total = cost[0] + cost[1];
    }
    // lowest total; among equals the first in loop order (strict `<`, bvh.rs:138).  A NaN total is never taken.
    float best = (total == total) ? total : __builtin_huge_valf();
    uint32_t best_lane = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(best, off);
        const uint32_t ol = __shfl_xor(best_lane, off);
        if (o < best || (o == best && ol < best_lane)) {
            best = o;
            best_lane = ol;
        }
    }
    // bvh.rs:169: a leaf unless the best split beats testing every shape
    const bool leaf = !(best < __builtin_huge_valf()) || (15.0f + best >= 2.0f * static_cast<float>(nd.count));
    if (lane == 0 && leaf) {
        nd.slot = kNone;
        pool[node] = nd;
    }
    if (leaf || lane != best_lane) return;

    const uint32_t c0 = atomicAdd(&counters[0], 2u);
    if (c0 + 2u > pool_cap) {  // cannot happen: a binary tree over n shapes has fewer than 2n nodes
        atomicExch(&counters[2], 1u);
        return;
    }
    uint32_t begin = nd.first;
    for (int c = 0; c < 2; ++c) {
        BNode ch{};
        for (int j = 0; j < 3; ++j) {
            ch.lo[j] = dec(side[c].lo[j]);
            ch.hi[j] = dec(side[c].hi[j]);
        }
        ch.first = begin;
        ch.count = side[c].count;
        ch.sa = sa[c];
        ch.child0 = 0;
        ch.slot = kNone;
        if (ch.count >= kMinSplit) {
            ch.slot = atomicAdd(&counters[1], 1u);
            next_active[ch.slot] = c0 + c;
        }
        pool[c0 + c] = ch;
        begin += ch.count;
    }
    nd.child0 = c0;
    nd.axis = axis;
    nd.split = plane;
    nd.mid = nd.first + side[0].count;
    pool[node] = nd;  // slot stays set: the partition kernels recognise the node by it
}

// ---- partition (Iterator::partition_in_place) of every node split at this level

// flags: low word counts misplaced falses, high word misplaced trues
__device__ __forceinline__ unsigned long long misplaced(const ShapeBounds* sb, const uint32_t* idx, const uint32_t* node_of,
                                                        const BNode* pool, uint32_t i) {
    const uint32_t node = node_of[i];
    if (node == kNone) return 0ull;
    const BNode nd = pool[node];
    if (nd.child0 == 0) return 0ull;  // became a leaf
    const float4 c = sb[idx[i]].centroid;
    const float v = nd.axis == 0 ? c.x : (nd.axis == 1 ? c.y : c.z);
    const bool pred = v < nd.split;
    if (!pred && i < nd.mid) return 1ull;
    if (pred && i >= nd.mid) return 1ull << 32;
    return 0ull;
}

// exclusive prefix sums of the flags inside blocks of kScanBlock positions, block totals aside
__global__ void __launch_bounds__(256) k_scan_local(const ShapeBounds* sb, const uint32_t* idx, const uint32_t* node_of,
                                                    const BNode* pool, uint32_t n, unsigned long long* scan,
                                                    unsigned long long* totals) {
    __shared__ unsigned long long s_wave[4];
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * 4u;
    unsigned long long f[4], sum = 0;
    for (int k = 0; k < 4; ++k) {
        f[k] = base + k < n ? misplaced(sb, idx, node_of, pool, base + k) : 0ull;
        sum += f[k];
    }
    // inclusive scan of the per-thread sums across the workgroup
    unsigned long long inc = sum;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= static_cast<uint32_t>(off)) inc += o;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long before = inc - sum;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) scan[base + k] = before;
        before += f[k];
    }
    if (threadIdx.x == 255u) totals[blockIdx.x] = before;
}

// one workgroup: exclusive prefix sums of the block totals, in place; grand total in totals[n_blocks]
__global__ void __launch_bounds__(256) k_scan_totals(unsigned long long* totals, uint32_t n_blocks) {
    __shared__ unsigned long long s_wave[4];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_blocks; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n_blocks ? totals[i] : 0ull;
        unsigned long long inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(inc, off);
            if (lane >= static_cast<uint32_t>(off)) inc += o;
        }
        if (lane == 63u) s_wave[wave] = inc;
        __syncthreads();
        unsigned long long before = s_carry + inc - v;
        for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n_blocks) totals[i] = before;
        __syncthreads();
        if (threadIdx.x == 255u) s_carry = before + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[n_blocks] = s_carry;
}

__device__ __forceinline__ unsigned long long scan_at(const unsigned long long* scan, const unsigned long long* totals,
                                                      uint32_t n, uint32_t n_blocks, uint32_t i) {
    return i < n ? scan[i] + totals[i / kScanBlock] : totals[n_blocks];
}

// every misplaced true records its position under its rank
__global__ void k_true_positions(const ShapeBounds* sb, const uint32_t* idx, const uint32_t* node_of, const BNode* pool,
                                 uint32_t n, uint32_t n_blocks, const unsigned long long* scan,
                                 const unsigned long long* totals, uint32_t* true_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (misplaced(sb, idx, node_of, pool, i) >> 32)
        true_pos[static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, i) >> 32)] = i;
}

// every misplaced false swaps with its partner; every position moves on to its child (or is finished)
__global__ void k_swap(const ShapeBounds*, uint32_t* idx, const uint32_t* node_of, uint32_t* node_next, const BNode* pool,
                       uint32_t n, uint32_t n_blocks, const unsigned long long* scan, const unsigned long long* totals,
                       const uint32_t* true_pos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = node_of[i];
    if (node == kNone) {
        node_next[i] = kNone;
        return;
    }
    const BNode nd = pool[node];
    if (nd.child0 == 0) {
        node_next[i] = kNone;
        return;
    }
    const uint32_t child = nd.child0 + (i >= nd.mid ? 1u : 0u);
    node_next[i] = pool[child].slot != kNone ? child : kNone;
    // (whether this position holds a misplaced false is read off the prefix sums, not recomputed
    // from idx[], which other threads of this launch are exchanging)
    const uint32_t here = static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, i));
    if (static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, i + 1)) != here) {
        const uint32_t j = here - static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, nd.first));
        const uint32_t t_first = static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, nd.first) >> 32);
        const uint32_t t_end = static_cast<uint32_t>(scan_at(scan, totals, n, n_blocks, nd.first + nd.count) >> 32);
        const uint32_t partner = true_pos[t_first + (t_end - t_first - 1u - j)];  // j-th from the back
        const uint32_t mine = idx[i];
        idx[i] = idx[partner];
        idx[partner] = mine;
    }
}

}  // namespace rtb

// ------------------------------------------------------------------ host side

namespace {

struct DeviceBuffers {
    std::vector<void*> ptrs;
    ~DeviceBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t alloc(T** out, size_t count) {
        *out = nullptr;
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(out), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

}  // namespace

// Declared in rt_context.hip (the context owns the stream and the error text).
int rt_context_fail(RtContext* ctx, int code, const std::string& msg);
int rt_context_device(RtContext* ctx, hipStream_t* stream);

#define BVH_TRY(expr)                                                                                          \
    do {                                                                                                       \
        const hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                                \
            (void)hipGetLastError();                                                                           \
            return rt_context_fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,                   \
                                   std::string(#expr) + ": " + hipGetErrorString(e_));                        \
        }                                                                                                      \
    } while (0)

namespace {

template <class Shape>
int build_bvh_on_gpu(RtContext* ctx, Shape* shapes, uint32_t n, RtNode** nodes_out, uint32_t* n_nodes) {
    using namespace rtb;
    if (!ctx || !nodes_out || !n_nodes || (n && !shapes)) return RT_ERR_INVALID;
    *nodes_out = nullptr;
    *n_nodes = 0;
    hipStream_t stream = nullptr;
    int rc = rt_context_device(ctx, &stream);
    if (rc) return rc;
    if (n > 0x07FFFFFFu) return rt_context_fail(ctx, RT_ERR_INVALID, "too many shapes");

    std::vector<BNode> pool_h;
    std::vector<uint32_t> idx_h(n);
    if (n == 0) {
        // bvh.rs:49-63: an empty family is one leaf with the fold's identities as its box
        BNode r{};
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = INFINITY;
            r.hi[k] = -INFINITY;
        }
        pool_h.push_back(r);
    } else {
        DeviceBuffers mem;
        Shape* d_tris;
        ShapeBounds* d_sb;
        uint32_t *d_idx, *d_node[2], *d_active[2], *d_bins, *d_counters, *d_root, *d_true_pos;
        BNode* d_pool;
        unsigned long long *d_scan, *d_totals;
        const uint32_t pool_cap = 2u * n + 1u;
        const uint32_t max_active = n / kMinSplit + 1u;
        const uint32_t n_blocks = (n + kScanBlock - 1) / kScanBlock;
        BVH_TRY(mem.alloc(&d_tris, n));
        BVH_TRY(mem.alloc(&d_sb, n));
        BVH_TRY(mem.alloc(&d_idx, n));
        BVH_TRY(mem.alloc(&d_node[0], n));
        BVH_TRY(mem.alloc(&d_node[1], n));
        BVH_TRY(mem.alloc(&d_active[0], max_active));
        BVH_TRY(mem.alloc(&d_active[1], max_active));
        BVH_TRY(mem.alloc(&d_pool, pool_cap));
        BVH_TRY(mem.alloc(&d_counters, 4));
        BVH_TRY(mem.alloc(&d_root, 8));
        BVH_TRY(mem.alloc(&d_true_pos, n));
        BVH_TRY(mem.alloc(&d_scan, n));
        BVH_TRY(mem.alloc(&d_totals, n_blocks + 1));
        BVH_TRY(mem.alloc(&d_bins, static_cast<size_t>(max_active) * kNodeBinWords));

        BVH_TRY(hipMemcpyAsync(d_tris, shapes, static_cast<size_t>(n) * sizeof(Shape), hipMemcpyHostToDevice, stream));
        const uint32_t root_init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};  // [6] = non-finite shapes
        BVH_TRY(hipMemcpyAsync(d_root, root_init, sizeof root_init, hipMemcpyHostToDevice, stream));
        BVH_TRY(hipMemsetAsync(d_counters, 0, 4 * sizeof(uint32_t), stream));
        const dim3 per_shape((n + 255) / 256), block(256);
        hipLaunchKernelGGL(k_bounds<Shape>, per_shape, block, 0, stream, static_cast<const Shape*>(d_tris), n, d_sb, d_idx, d_root,
                           d_root + 6);
        hipLaunchKernelGGL(k_root, per_shape, block, 0, stream, d_root, n, d_pool, d_node[0], d_active[0], d_counters);
        BVH_TRY(hipGetLastError());
        uint32_t h[8];
        BVH_TRY(hipMemcpyAsync(h, d_root, sizeof h, hipMemcpyDeviceToHost, stream));
        BVH_TRY(hipStreamSynchronize(stream));
        if (h[6] != 0)
            return rt_context_fail(ctx, RT_ERR_INVALID, "non-finite coordinates: build this family on the host");

        uint32_t n_active = n >= kMinSplit ? 1u : 0u;
        int cur = 0;
        for (uint32_t level = 0; n_active > 0; ++level) {
            if (level > 4096) return rt_context_fail(ctx, RT_ERR_HIP, "BVH build did not terminate");
            const uint32_t bin_words = n_active * kNodeBinWords;
            hipLaunchKernelGGL(k_init_bins, dim3((bin_words + 255) / 256), block, 0, stream, d_bins, bin_words);
            hipLaunchKernelGGL(k_bin, per_shape, block, 0, stream, d_sb, d_idx, d_node[cur], d_pool, n, d_bins);
            BVH_TRY(hipMemsetAsync(d_counters + 1, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_split, dim3(n_active), dim3(64), 0, stream, d_pool, d_active[cur], n_active, d_bins,
                               d_active[cur ^ 1], d_counters, pool_cap);
            hipLaunchKernelGGL(k_scan_local, dim3(n_blocks), block, 0, stream, d_sb, d_idx, d_node[cur], d_pool, n, d_scan, d_totals);
            hipLaunchKernelGGL(k_scan_totals, dim3(1), block, 0, stream, d_totals, n_blocks);
            hipLaunchKernelGGL(k_true_positions, per_shape, block, 0, stream, d_sb, d_idx, d_node[cur], d_pool, n, n_blocks,
                               d_scan, d_totals, d_true_pos);
            hipLaunchKernelGGL(k_swap, per_shape, block, 0, stream, d_sb, d_idx, d_node[cur], d_node[cur ^ 1], d_pool, n, n_blocks,
                               d_scan, d_totals, d_true_pos);
            BVH_TRY(hipGetLastError());
            BVH_TRY(hipMemcpyAsync(h, d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            BVH_TRY(hipStreamSynchronize(stream));
            if (h[2]) return rt_context_fail(ctx, RT_ERR_HIP, "BVH node pool overflow");
            n_active = h[1];
            cur ^= 1;
        }
        BVH_TRY(hipMemcpyAsync(h, d_counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        BVH_TRY(hipStreamSynchronize(stream));
        pool_h.resize(h[0]);
        BVH_TRY(hipMemcpyAsync(pool_h.data(), d_pool, pool_h.size() * sizeof(BNode), hipMemcpyDeviceToHost, stream));
        BVH_TRY(hipMemcpyAsync(idx_h.data(), d_idx, static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        BVH_TRY(hipStreamSynchronize(stream));
    }

    // bvh.rs:198-221: array order is depth-first pre-order.  Children were appended to the pool
    // after their parents, so sizes fold backwards and numbers unfold forwards.
    const uint32_t total = static_cast<uint32_t>(pool_h.size());
    std::vector<uint32_t> size(total, 1), number(total, 0);
    for (uint32_t i = total; i-- > 0;)
        if (pool_h[i].child0) size[i] = 1 + size[pool_h[i].child0] + size[pool_h[i].child0 + 1];
    for (uint32_t i = 0; i < total; ++i)
        if (pool_h[i].child0) {
            number[pool_h[i].child0] = number[i] + 1;
            number[pool_h[i].child0 + 1] = number[i] + 1 + size[pool_h[i].child0];
        }
    RtNode* out = static_cast<RtNode*>(std::malloc(static_cast<size_t>(total) * sizeof(RtNode)));
    if (!out) return rt_context_fail(ctx, RT_ERR_OOM, "node array");
    for (uint32_t i = 0; i < total; ++i) {
        const BNode& b = pool_h[i];
        RtNode& r = out[number[i]];
        std::memcpy(r.min, b.lo, sizeof r.min);
        std::memcpy(r.max, b.hi, sizeof r.max);
        if (b.child0) {
            r.a = number[b.child0];
            r.b = number[b.child0 + 1];
        } else {
            r.a = b.first;
            r.b = b.count | RT_LEAF_BIT;
        }
    }
    // the in-place permutation of config.rs:113-134
    if (n) {
        std::vector<Shape> moved(n);
        for (uint32_t i = 0; i < n; ++i) moved[i] = shapes[idx_h[i]];
        std::memcpy(shapes, moved.data(), static_cast<size_t>(n) * sizeof(Shape));
    }
    *nodes_out = out;
    *n_nodes = total;
    return RT_OK;
}

}  // namespace

extern "C" int rt_build_bvh_triangles(RtContext* ctx, RtTriangle* shapes, uint32_t n, RtNode** nodes, uint32_t* n_nodes) try {
    return build_bvh_on_gpu(ctx, shapes, n, nodes, n_nodes);
} RT_ABI_END_CTX(ctx)

extern "C" int rt_build_bvh_spheres(RtContext* ctx, RtSphere* shapes, uint32_t n, RtNode** nodes, uint32_t* n_nodes) try {
    return build_bvh_on_gpu(ctx, shapes, n, nodes, n_nodes);
} RT_ABI_END_CTX(ctx)

extern "C" void rt_free(void* p) try { std::free(p); } RT_ABI_END_VOID
