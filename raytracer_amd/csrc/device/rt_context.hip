// C ABI of librt_mi355.so (include/rt_mi355.h): context, scene upload with the
// device re-layout, and the host-side driver of the wavefront bounce loop.
#include <hip/hip_runtime.h>
#include <link.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "abi_guard.h"
#include "kernels.h"
#include "layout.h"
#include "rt_mi355.h"

using namespace rtd;

namespace {

struct Family {
    DNode* nodes = nullptr;
    DLongLeaf* long_leaves = nullptr;
    void* shapes = nullptr;
    uint32_t root_ref = kLeafBit;  // empty leaf
    uint32_t n_shapes = 0;
    uint32_t n_branches = 0;  // branch records
    uint32_t max_stack = 1;
    bool bounds_fast = false;  // all box coordinates are 0 or of magnitude in [2^-60, 2^59)
};

constexpr int kMaxChains = 4;

struct Workspace {
    size_t cap_paths = 0;   // records per queue
    size_t cap_pixels = 0;  // pixels of sums / rgb8
    size_t cap_bounces = 0; // entries of the counter block
    DQueue q[2] = {};
    float4* result = nullptr;
    float* sums = nullptr;
    uint8_t* rgb8 = nullptr;
    uint32_t* counters = nullptr;  // [0..B] queue lengths, then 3 cursors per bounce
    uint2* spill = nullptr;      // kMaxChains areas of spill_entries each: traversal launches of different chains run side by side
    size_t spill_entries = 0;    // entries of one area
};

}  // namespace

// ---- scratch of the entry point that allocates nothing (rt_trace_rays_noalloc)
//
// One array per GPU, part of the loaded code object, held until the process ends.
// In host code the name `scratch` is only a handle for hipGetSymbolAddress: the
// device address of this GPU's copy is looked up once per context (rt_create) and
// kept in RtContext::scratch; kernels, memsets and copies are given that pointer.
constexpr size_t kScratchBytes = size_t{64} << 20;
constexpr uint64_t kNoAllocMaxRays = uint64_t{1} << 19;
// layout: [0, 256) counters; then seven per-ray arrays of kNoAllocMaxRays entries (origin,
// direction, hit distance, hit id, result records, and the two staging arrays for rays handed
// over in host memory: 68 bytes per ray, 34 MiB); the rest (from 36 MiB) is the overflow area of
// deep traversal stacks
constexpr size_t kScratchQueue = 256;
constexpr size_t kScratchSpill = size_t{36} << 20;
constexpr size_t kScratchSpillEntries = (kScratchBytes - kScratchSpill) / sizeof(uint2);
static_assert(kScratchQueue + kNoAllocMaxRays * 68 <= kScratchSpill, "per-ray arrays overlap the overflow area");
static_assert(sizeof(RtHitRecord) == sizeof(HitOut) && offsetof(RtHitRecord, prim) == offsetof(HitOut, prim),
              "k_hit_records writes RtHitRecord");
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

struct RtContext {
    int device = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    hipStream_t chain_stream[4] = {nullptr, nullptr, nullptr, nullptr};  // [0] = stream; the others carry further chains of a batch
    hipEvent_t chain_event[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t launch_stream = nullptr;  // where traversal launches go (stream, or a chain's stream)
    uint32_t launch_chain = 0;            // ... and which chain's stack overflow area they use
    uint32_t wave_cap_now = 0;            // traversal waves per CU while chains run side by side (0 = no cap)
    uint32_t tune_shade_groups = 8;       // shading workgroups (256 threads) per compute unit
    uint32_t n_chains = 2;                // independent chains a large batch is split into (1 = none)
    std::string error;

    // samples traced by rt_trace_samples and not folded yet: their region of the frame, how many (0 = none)
    uint32_t pending_rect[6] = {0, 0, 0, 0, 0, 0};  // x0, y0, w, h, frame width, frame height
    uint32_t pending_samples = 0;

    bool has_scene = false;
    Family fam[3];
    uint32_t* sphere_material = nullptr;
    uint32_t* plane_material = nullptr;
    DTriSurface* tri_surface = nullptr;
    DMaterial* materials = nullptr;
    uint32_t* texels = nullptr;
    uint32_t n_materials = 0;
    uint32_t inline_count[2] = {0, 0};  // spheres / planes scanned where rays are created

    Workspace ws;
    unsigned long long* d_stats = nullptr;  // hits, then {branch visits, shape tests} per family, then
                                            // {sphere tests, plane tests} of inlined families
    unsigned long long* d_inline_tests = nullptr;  // = d_stats + 7
    PathConsts* d_path_consts = nullptr;           // argument block of k_path's shading half
    uint32_t* h_pinned = nullptr;           // small pinned read-back area
    size_t h_pinned_words = 0;

    // (kernel, LDS bytes) -> workgroups per CU; a small fixed table (oldest entry replaced), so
    // that a launch never allocates host memory
    struct Occupancy {
        const void* fn = nullptr;
        size_t lds = 0;
        int per_cu = 0;
    } trace_occupancy[32];
    uint32_t trace_occupancy_next = 0;
    // overrides for the entry point that allocates nothing: where deep traversal stacks overflow to
    // (null = the workspace's area of the launching chain) and how many threads that area serves
    uint2* launch_spill = nullptr;
    uint32_t launch_max_threads = 0;
    unsigned char* scratch = nullptr;  // this GPU's copy of the file-scope scratch array (see rt_trace_rays_noalloc)
    uint32_t deepest_stack = 1;    // largest max_stack over the uploaded scene's three families
    uint32_t max_grid_threads = 0; // upper bound of the traversal grid's threads (sizes the spill area)
    uint32_t tune_lds_slots = 6, tune_phase_min = 24, tune_chunk_max = 256, tune_pair_min = 384;
    uint32_t tune_waves_per_cu = 0; // traversal waves resident per CU (0 = all the kernel's occupancy allows)
    bool tune_guided = true;        // queue reservations shrink towards the end of a launch
    uint32_t tune_refill_min = 12;  // idle lanes that trigger a refill
    uint32_t tune_waves = 8;        // waves per traversal workgroup (1, 8 or 16)
    uint32_t tune_cache_nodes = 144;  // branch records each multi-wave workgroup copies to LDS
    bool tune_share = true;  // deal pending (ray, triangle) pairs out to all lanes
    bool opt_count = false, opt_time = false, opt_fast_slab = false;
    uint32_t tune_path_group = 16384;  // paths a wave of k_path takes through all bounces at a time
    uint32_t opt_fused = 2;  // the whole bounce loop in one launch (k_path): 0 never, 1 whenever the scene allows, 2 for batches below kFusedBelowPaths and, above, for scenes on which it measured faster
    // which loop large batches of this scene and frame take when opt_fused == 2: measured once (render_sums_device)
    enum class LargeLoop : uint8_t { Unmeasured, PerBounce, OneLaunch } large_loop = LargeLoop::Unmeasured, forced_loop = LargeLoop::Unmeasured;
    uint64_t large_loop_key = 0;  // (bounces, pixels) the measurement belongs to
    uint32_t pilot_phase = 0;     // large batches measured so far (two decide)
    double pilot_ms_per_sample[2] = {0.0, 0.0};  // best time of the per-bounce launches / of the one launch
    bool debug_tallies = std::getenv("RT_MI355_DEBUG_TALLIES") != nullptr;  // slow diagnostics of the counting kernels
    uint64_t opt_samples_in_flight = 0;  // 0 = automatic
    uint64_t opt_passes_in_flight = 0;   // rt_render: passes traced together (0 = automatic, 1 = one at a time)
    std::vector<hipEvent_t> events;      // pairs around timed launches
    size_t events_used = 0;

    RtStats stats{};
    unsigned long long tallies[38] = {};  // diagnostics: see k_trace
};

namespace {

struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

constexpr size_t kStatWords = 48;  // + 38 diagnostic tallies of the triangle traversal + rays traced by k_path
constexpr size_t kStatSegments = 47;
constexpr uint32_t kFusedBelowPaths = 3u << 22;  // batch size below which the one-launch bounce loop is the faster one

int fail(RtContext* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError(); /* do not let it resurface at a later check */          \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,            \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
        }                                                                                    \
    } while (0)

template <class T>
int upload(RtContext* ctx, T** dst, const void* src, size_t count) {
    *dst = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(dst), bytes));
    if (count) HIP_TRY(ctx, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

void free_scene(RtContext* ctx) {
    for (Family& f : ctx->fam) {
        (void)hipFree(f.nodes);
        (void)hipFree(f.long_leaves);
        (void)hipFree(f.shapes);
        f = Family{};
    }
    (void)hipFree(ctx->sphere_material);
    (void)hipFree(ctx->plane_material);
    (void)hipFree(ctx->tri_surface);
    (void)hipFree(ctx->materials);
    (void)hipFree(ctx->texels);
    ctx->sphere_material = ctx->plane_material = nullptr;
    ctx->tri_surface = nullptr;
    ctx->materials = nullptr;
    ctx->texels = nullptr;
    ctx->inline_count[0] = ctx->inline_count[1] = 0;
    ctx->has_scene = false;
}

void free_queue(DQueue& q) {
    (void)hipFree(q.origin);
    (void)hipFree(q.direction);
    (void)hipFree(q.throughput);
    (void)hipFree(q.rng);
    (void)hipFree(q.slot);
    (void)hipFree(q.hit_t);
    (void)hipFree(q.hit_id);
    q = DQueue{};
}

void free_workspace(RtContext* ctx) {
    Workspace& w = ctx->ws;
    free_queue(w.q[0]);
    free_queue(w.q[1]);
    (void)hipFree(w.result);
    (void)hipFree(w.sums);
    (void)hipFree(w.rgb8);
    (void)hipFree(w.counters);
    (void)hipFree(w.spill);
    w = Workspace{};
}

// Re-lay-out one family's node array (reference order, 32-byte RtNode) into
// branch records and child references; validates everything a kernel will
// dereference so that a malformed description is refused here, not on the GPU.
int build_family(RtContext* ctx, const char* name, const RtNode* nodes, uint32_t n_nodes, uint32_t n_shapes,
                 Family& out, std::vector<DNode>& branches, std::vector<DLongLeaf>& long_leaves) {
    branches.clear();
    long_leaves.clear();
    out.n_shapes = n_shapes;
    out.n_branches = 0;
    out.root_ref = kLeafBit;
    out.max_stack = 1;
    out.bounds_fast = false;
    if (n_nodes == 0) {
        if (n_shapes != 0) return fail(ctx, RT_ERR_INVALID, std::string(name) + ": shapes without a BVH");
        return RT_OK;
    }
    if (!nodes) return fail(ctx, RT_ERR_INVALID, std::string(name) + ": null node array");
    if (n_shapes > kLeafFirstMask) return fail(ctx, RT_ERR_INVALID, std::string(name) + ": too many shapes");

    std::vector<uint32_t> branch_id(n_nodes, 0xFFFFFFFFu), depth(n_nodes, 0);
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const RtNode& nd = nodes[i];
        if (nd.b & RT_LEAF_BIT) {
            const uint64_t first = nd.a, count = nd.b & ~RT_LEAF_BIT;
            if (first + count > n_shapes)
                return fail(ctx, RT_ERR_INVALID, std::string(name) + ": leaf range outside the shape array");
        } else {
            // children are appended after their parent (bvh.rs:209-211); requiring
            // that also rules out cycles
            if (nd.a <= i || nd.b <= i || nd.a >= n_nodes || nd.b >= n_nodes || nd.a == nd.b)
                return fail(ctx, RT_ERR_INVALID, std::string(name) + ": bad child index");
        }
    }
    // a tree: no node has two parents (the depth computed below, which sizes the
    // traversal stack, relies on it)
    {
        std::vector<uint8_t> has_parent(n_nodes, 0);
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const RtNode& nd = nodes[i];
            if (nd.b & RT_LEAF_BIT) continue;
            for (uint32_t c : {nd.a, nd.b}) {
                if (has_parent[c]) return fail(ctx, RT_ERR_INVALID, std::string(name) + ": node with two parents");
                has_parent[c] = 1;
            }
        }
    }
    // Branch records are numbered breadth-first from the root, so that the top of
    // the tree, which every ray walks, is one contiguous run at the start of the
    // array.  Nodes the root cannot reach get no record.
    uint32_t n_branch = 0;
    {
        std::vector<uint32_t> order;
        order.reserve(n_nodes);
        if (!(nodes[0].b & RT_LEAF_BIT)) order.push_back(0);
        for (size_t head = 0; head < order.size(); ++head) {
            const uint32_t i = order[head];
            if (branch_id[i] != 0xFFFFFFFFu) continue;  // a node with two parents keeps its first number
            branch_id[i] = n_branch++;
            const uint32_t c[2] = {nodes[i].a, nodes[i].b};
            for (int k = 0; k < 2; ++k)
                if (!(nodes[c[k]].b & RT_LEAF_BIT)) order.push_back(c[k]);
        }
    }
    auto ref_of = [&](uint32_t i) -> uint32_t {
        const RtNode& nd = nodes[i];
        if (!(nd.b & RT_LEAF_BIT)) return branch_id[i];
        const uint32_t first = nd.a, count = nd.b & ~RT_LEAF_BIT;
        if (count < kLeafLong) return kLeafBit | (count << kLeafCountShift) | first;
        long_leaves.push_back(DLongLeaf{first, count});
        return kLeafBit | (kLeafLong << kLeafCountShift) | static_cast<uint32_t>(long_leaves.size() - 1);
    };
    branches.resize(n_branch);
    out.n_branches = n_branch;
    uint32_t max_depth = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) {
        const RtNode& nd = nodes[i];
        if ((nd.b & RT_LEAF_BIT) || branch_id[i] == 0xFFFFFFFFu) continue;
        const uint32_t c[2] = {nd.a, nd.b};
        DNode& r = branches[branch_id[i]];
        float* axis[3] = {r.x, r.y, r.z};
        for (int k = 0; k < 3; ++k) {
            axis[k][0] = nodes[c[0]].min[k];
            axis[k][1] = nodes[c[1]].min[k];
            axis[k][2] = nodes[c[0]].max[k];
            axis[k][3] = nodes[c[1]].max[k];
        }
        r.ref0 = ref_of(c[0]);
        r.ref1 = ref_of(c[1]);
        r.pad0 = r.pad1 = 0;
        for (int k = 0; k < 2; ++k) {
            depth[c[k]] = depth[i] + 1;
            max_depth = std::max(max_depth, depth[c[k]]);
        }
    }
    if (long_leaves.size() >= kLeafFirstMask) return fail(ctx, RT_ERR_INVALID, std::string(name) + ": too many long leaves");
    out.root_ref = ref_of(0);
    out.max_stack = max_depth + 2;
    // The kernels' cheaper exact division (div_by in kernels.h) is valid only when
    // no slab-test numerator can be tiny or huge; that holds if every box
    // coordinate is zero or of ordinary magnitude.  Infinite boxes (planes, empty
    // families) and extreme scenes keep the plain division.
    bool ordinary = true;
    for (const DNode& r : branches) {
        const float* c = r.x;  // the twelve box floats are contiguous
        for (int k = 0; k < 12; ++k) {
            const float m = std::fabs(c[k]);
            if (!(m == 0.0f || (m >= 0x1p-60f && m < 0x1p59f))) ordinary = false;
        }
    }
    out.bounds_fast = ordinary;
    return RT_OK;
}

int ensure_pinned(RtContext* ctx, size_t words) {
    if (ctx->h_pinned_words >= words) return RT_OK;
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    ctx->h_pinned = nullptr;
    ctx->h_pinned_words = 0;
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pinned), words * sizeof(uint32_t), hipHostMallocDefault));
    ctx->h_pinned_words = words;
    return RT_OK;
}

int alloc_queue(RtContext* ctx, DQueue& q, size_t n) {
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.origin), n * sizeof(P3)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.direction), n * sizeof(P3)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.throughput), n * sizeof(P3)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.rng), n * sizeof(uint2)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.slot), n * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.hit_t), n * sizeof(float)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&q.hit_id), n * sizeof(uint32_t)));
    return RT_OK;
}

int ensure_workspace(RtContext* ctx, size_t n_paths, size_t n_pixels, size_t n_bounces) {
    Workspace& w = ctx->ws;
    if (w.cap_paths < n_paths) {
        ctx->pending_samples = 0;  // traced samples waiting in the result array go with it
        free_queue(w.q[0]);
        free_queue(w.q[1]);
        (void)hipFree(w.result);
        w.result = nullptr;
        w.cap_paths = 0;
        // (entries beyond the paths: k_path stores every wave's share from a multiple of 64 on)
        int rc = alloc_queue(ctx, w.q[0], n_paths + ctx->max_grid_threads);
        if (rc) return rc;
        rc = alloc_queue(ctx, w.q[1], n_paths + ctx->max_grid_threads);
        if (rc) return rc;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&w.result), n_paths * sizeof(float4)));
        w.cap_paths = n_paths;
    }
    if (w.cap_pixels < n_pixels) {
        (void)hipFree(w.sums);
        (void)hipFree(w.rgb8);
        w.sums = nullptr;
        w.rgb8 = nullptr;
        w.cap_pixels = 0;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&w.sums), n_pixels * 3 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&w.rgb8), n_pixels * 3));
        w.cap_pixels = n_pixels;
    }
    if (!w.counters || w.cap_bounces < n_bounces) {
        (void)hipFree(w.counters);
        w.counters = nullptr;
        w.cap_bounces = 0;
        // queue lengths [0..B], then 3 cursors for each of B bounces; one such block per chain
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&w.counters), kMaxChains * (n_bounces + 1 + 3 * n_bounces + 4) * sizeof(uint32_t)));
        w.cap_bounces = n_bounces;
    }
    return ensure_pinned(ctx, kMaxChains * (n_bounces + 1 + 3 * n_bounces + 4) + 8);
}

struct Region {
    uint32_t x0, y0, w, h;
    size_t n_pixels;
};

int check_params(RtContext* ctx, const RtRenderParams* p, Region& r) {
    if (!p) return fail(ctx, RT_ERR_INVALID, "null render parameters");
    if (p->width < 2 || p->height < 2) return fail(ctx, RT_ERR_INVALID, "width and height must be at least 2");
    if (p->rng_mode != RT_RNG_SUBSTREAM)
        return fail(ctx, RT_ERR_INVALID, "the GPU implements RT_RNG_SUBSTREAM only");
    if (p->tile_w) {
        if (!p->tile_h || static_cast<uint64_t>(p->tile_x) + p->tile_w > p->width ||
            static_cast<uint64_t>(p->tile_y) + p->tile_h > p->height)
            return fail(ctx, RT_ERR_INVALID, "tile outside the image");
        r = Region{p->tile_x, p->tile_y, p->tile_w, p->tile_h, static_cast<size_t>(p->tile_w) * p->tile_h};
    } else {
        r = Region{0, 0, p->width, p->height, static_cast<size_t>(p->width) * p->height};
    }
    if (r.n_pixels > 0x7FFFFFFFull) return fail(ctx, RT_ERR_INVALID, "region too large: render it in tiles");
    return RT_OK;
}

FrameArgs frame_args(const RtRenderParams* p, const Region& r) {
    FrameArgs f{};
    // lib.rs:189-190, evaluated on the host in binary32 exactly as the kernels would
    const float wm1 = static_cast<float>(p->width - 1), hm1 = static_cast<float>(p->height - 1);
    for (int k = 0; k < 3; ++k) {
        f.top_left[k] = p->top_left[k];
        f.camera[k] = p->camera[k];
        f.row_step[k] = p->top_edge[k] / wm1;
        f.column_step[k] = p->left_edge[k] / hm1;
    }
    f.width = p->width;
    f.height = p->height;
    f.x0 = r.x0;
    f.y0 = r.y0;
    f.tile_w = r.w;
    f.tile_h = r.h;
    f.seed = p->rng_seed;
    f.stride = 2ull + 3ull * p->max_bounces;
    return f;
}

DScene device_scene(const RtContext* ctx) {
    DScene s{};
    for (int k = 0; k < 3; ++k) {
        s.family[k].nodes = ctx->fam[k].nodes;
        s.family[k].long_leaves = ctx->fam[k].long_leaves;
        s.family[k].shapes = ctx->fam[k].shapes;
        s.family[k].root_ref = ctx->fam[k].root_ref;
        s.family[k].n_shapes = ctx->fam[k].n_shapes;
        s.family[k].max_stack = ctx->fam[k].max_stack;
    }
    s.sphere_material = ctx->sphere_material;
    s.plane_material = ctx->plane_material;
    s.tri_surface = ctx->tri_surface;
    s.materials = ctx->materials;
    s.texels = ctx->texels;
    s.inline_count[0] = ctx->inline_count[0];
    s.inline_count[1] = ctx->inline_count[1];
    return s;
}

// Where the kernels that scan inlined families add their shape-test counts
// (the "shape tests" words of the sphere and plane families); null unless
// counting is on.  d_stats: [0] hits, then {branch visits, shape tests} x 3.
unsigned long long* inline_counters(RtContext* ctx) { return ctx->opt_count ? ctx->d_inline_tests : nullptr; }

// Launch one traversal kernel variant over a grid that is exactly what the
// device keeps resident for its register and LDS use (the kernel is persistent:
// its waves loop until the queue is drained).
template <int KIND, bool COUNT, bool SHARE, int WAVES, bool FASTSLAB = false>
int launch_trace_variant(RtContext* ctx, TraceArgs a) {
    const void* fn = reinterpret_cast<const void*>(&k_trace<KIND, COUNT, SHARE, WAVES, FASTSLAB>);
    const int block = kTraceBlock * WAVES;
    a.cache_nodes = WAVES > 1 ? std::min(ctx->tune_cache_nodes, ctx->fam[KIND].n_branches) : 0u;
    const size_t lds = static_cast<size_t>(a.cache_nodes) * sizeof(DNode) + WAVES * trace_lds_per_wave(a.lds_slots);
    const RtContext::Occupancy* it = nullptr;
    for (const RtContext::Occupancy& o : ctx->trace_occupancy)
        if (o.fn == fn && o.lds == lds) it = &o;
    if (!it) {
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        int per_cu = 0;
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds));
        if (per_cu < 1) return fail(ctx, RT_ERR_INVALID, "traversal workgroup does not fit a compute unit (waves, LDS slots, cached nodes)");
        per_cu = std::min(per_cu, 32 / WAVES);
        if (std::getenv("RT_MI355_DEBUG_TALLIES"))
            std::fprintf(stderr, "[rt_mi355] traversal kind %d: %d waves per workgroup, %zu bytes of LDS, %d workgroups per CU\n",
                         KIND, WAVES, lds, per_cu);
        RtContext::Occupancy& slot = ctx->trace_occupancy[ctx->trace_occupancy_next++ % 32u];
        slot = RtContext::Occupancy{fn, lds, per_cu};
        it = &slot;
    }
    // share of the CU's 32 wave slots: the caller's setting, or the overlapped render loop's
    uint32_t wave_cap = ctx->tune_waves_per_cu ? ctx->tune_waves_per_cu : (ctx->wave_cap_now ? ctx->wave_cap_now : 32u);
    const uint32_t per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(static_cast<uint32_t>(it->per_cu), wave_cap / WAVES));
    uint32_t grid = static_cast<uint32_t>(ctx->n_cus) * per_cu;
    const uint32_t thread_limit = ctx->launch_max_threads ? ctx->launch_max_threads : ctx->max_grid_threads;
    if (ctx->launch_max_threads) grid = std::min(grid, thread_limit / static_cast<uint32_t>(block));
    if (grid == 0) return fail(ctx, RT_ERR_INVALID, "stack overflow area too small for one traversal workgroup");
    a.spill_stride = grid * static_cast<uint32_t>(block);
    if (a.spill_stride > thread_limit) return fail(ctx, RT_ERR_INVALID, "traversal grid larger than the stack overflow area");
    hipLaunchKernelGGL((k_trace<KIND, COUNT, SHARE, WAVES, FASTSLAB>), dim3(grid), dim3(block), lds, ctx->launch_stream, a);
    return RT_OK;
}

template <int KIND>
int launch_trace_kind(RtContext* ctx, const TraceArgs& a) {
    if constexpr (KIND == 2) {
        // the triangle family: shared shape step, optionally several waves per workgroup
        if (ctx->tune_share) {
            // the reciprocal slab test exists for the uncounted kernels of the triangle family
            if (ctx->opt_fast_slab && !ctx->opt_count) {
                switch (ctx->tune_waves) {
                    case 8: return launch_trace_variant<2, false, true, 8, true>(ctx, a);
                    case 16: return launch_trace_variant<2, false, true, 16, true>(ctx, a);
                    default: return launch_trace_variant<2, false, true, 1, true>(ctx, a);
                }
            }
            switch (ctx->tune_waves) {
                case 8: return ctx->opt_count ? launch_trace_variant<2, true, true, 8>(ctx, a) : launch_trace_variant<2, false, true, 8>(ctx, a);
                case 16: return ctx->opt_count ? launch_trace_variant<2, true, true, 16>(ctx, a) : launch_trace_variant<2, false, true, 16>(ctx, a);
                default: return ctx->opt_count ? launch_trace_variant<2, true, true, 1>(ctx, a) : launch_trace_variant<2, false, true, 1>(ctx, a);
            }
        }
    }
    return ctx->opt_count ? launch_trace_variant<KIND, true, false, 1>(ctx, a) : launch_trace_variant<KIND, false, false, 1>(ctx, a);
}

// Nearest hit of every ray in `q` over the three families, in the reference's
// order (lib.rs:314-333).  `cursors` points at three zeroed words.
int launch_trace(RtContext* ctx, const DQueue& q, const uint32_t* n_rays, uint32_t* cursors) {
    const DScene sc = device_scene(ctx);
    for (int kind = 0; kind < 3; ++kind) {
        if (ctx->fam[kind].n_shapes == 0) continue;  // an empty family can never be hit
        if (kind < 2 && ctx->inline_count[kind]) continue;  // already scanned when the ray was created
        TraceArgs a{};
        a.fam = sc.family[kind];
        a.q = q;
        a.n_rays = n_rays;
        a.cursor = cursors + kind;
        a.spill = ctx->launch_spill ? ctx->launch_spill : ctx->ws.spill + ctx->ws.spill_entries * ctx->launch_chain;
        a.lds_slots = ctx->tune_lds_slots;
        a.phase_min = ctx->tune_phase_min;
        a.chunk_max = ctx->tune_chunk_max;
        a.pair_min = ctx->tune_pair_min;
        a.refill_min = ctx->tune_refill_min;
        a.guided = ctx->tune_guided ? 1u : 0u;
        a.bounds_fast = ctx->fam[kind].bounds_fast ? 1 : 0;
        a.counters = ctx->d_stats + 1 + 2 * kind;  // [branch visits, shape tests] of this family
        a.wave_tallies = kind == RT_SHAPE_TRIANGLE ? ctx->d_stats + 9 : nullptr;
        a.visit_bands = kind == RT_SHAPE_TRIANGLE && ctx->debug_tallies ? ctx->d_stats + 15 : nullptr;
        const bool timed = ctx->opt_time && kind == RT_SHAPE_TRIANGLE;
        if (timed) {
            if (ctx->events_used + 2 > ctx->events.size()) {
                for (int k = 0; k < 2; ++k) {
                    hipEvent_t ev;
                    HIP_TRY(ctx, hipEventCreate(&ev));
                    ctx->events.push_back(ev);
                }
            }
            HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used], ctx->launch_stream));
        }
        const int rc = kind == 0 ? launch_trace_kind<0>(ctx, a) : kind == 1 ? launch_trace_kind<1>(ctx, a) : launch_trace_kind<2>(ctx, a);
        if (rc) return rc;
        HIP_TRY(ctx, hipGetLastError());
        if (timed) {
            HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used + 1], ctx->launch_stream));
            ctx->events_used += 2;
            ctx->stats.trace_launches++;
        }
    }
    return RT_OK;
}

// After a stream synchronisation: fold event timings and device counters into
// the host-side statistics.
int harvest(RtContext* ctx) {
    for (size_t i = 0; i + 1 < ctx->events_used; i += 2) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->events[i], ctx->events[i + 1]));
        ctx->stats.trace_ms += ms;
    }
    ctx->events_used = 0;
    // d_stats: [0] hits, then [branch visits, shape tests] per family
    unsigned long long h[kStatWords];
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.hits += h[0];
    for (int kind = 0; kind < 3; ++kind) {
        ctx->stats.branch_visits += h[1 + 2 * kind];
        ctx->stats.prim_tests[kind] += h[2 + 2 * kind];
    }
    ctx->stats.prim_tests[0] += h[7];
    ctx->stats.prim_tests[1] += h[8];
    for (int k = 0; k < 38; ++k) ctx->tallies[k] += h[9 + k];
    ctx->stats.segments += h[kStatSegments];
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof h, ctx->stream));
    return RT_OK;
}

size_t choose_samples_in_flight(const RtContext* ctx, size_t n_pixels, uint32_t count) {
    size_t s = ctx->opt_samples_in_flight;
    if (s == 0) {
        // Long queues amortise the thin tail of late bounces and the ends of launches: 2^28 paths
        // (128 samples of a 1080p frame, 35 GB of queues) render 2.7 % faster than 2^26 and 2^29
        // another 1 %, and the memory is there (288 GB); the caller of this function backs off
        // when it is not.
        const size_t target_paths = size_t{1} << 29;
        s = std::max<size_t>(1, target_paths / std::max<size_t>(n_pixels, 1));
    }
    s = std::min<size_t>(s, count);
    // path indices and queue cursors are 32-bit
    while (s > 1 && s * n_pixels > 0x7FFFFFFFull) --s;
    return std::max<size_t>(s, 1);
}

int render_batch_overlapped(RtContext* ctx, const FrameArgs& f, const DScene& sc, uint32_t B, uint32_t n_paths,
                            uint32_t n_pix, unsigned long long first_sample, uint64_t* segments_out);

// Can the whole bounce loop run as one launch of k_path?  Only the triangle family may need a
// traversal of its own: spheres and planes must be absent or scanned where rays are created.
bool fused_loop_possible(const RtContext* ctx) {
    if (ctx->opt_count || !ctx->tune_share || (ctx->tune_waves != 8 && ctx->tune_waves != 16)) return false;
    if (ctx->fam[RT_SHAPE_TRIANGLE].n_shapes == 0) return false;
    for (int kind = 0; kind < 2; ++kind)
        if (ctx->fam[kind].n_shapes != 0 && ctx->inline_count[kind] == 0) return false;
    return true;
}

template <int WAVES, bool FASTSLAB>
int launch_path_variant(RtContext* ctx, PathArgs a, const FrameArgs& f, uint32_t n_pix, unsigned long long first_sample) {
    const void* fn = reinterpret_cast<const void*>(&k_path<WAVES, FASTSLAB>);
    const int block = kTraceBlock * WAVES;
    a.t.cache_nodes = std::min(ctx->tune_cache_nodes, ctx->fam[RT_SHAPE_TRIANGLE].n_branches);
    const size_t lds = static_cast<size_t>(a.t.cache_nodes) * sizeof(DNode) + WAVES * trace_lds_per_wave(a.t.lds_slots);
    const RtContext::Occupancy* it = nullptr;
    for (const RtContext::Occupancy& o : ctx->trace_occupancy)
        if (o.fn == fn && o.lds == lds) it = &o;
    if (!it) {
        if (lds > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        int per_cu = 0;
        HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, lds));
        if (per_cu < 1) return fail(ctx, RT_ERR_INVALID, "bounce-loop workgroup does not fit a compute unit (waves, LDS slots, cached nodes)");
        per_cu = std::min(per_cu, 32 / WAVES);
        RtContext::Occupancy& slot = ctx->trace_occupancy[ctx->trace_occupancy_next++ % 32u];
        slot = RtContext::Occupancy{fn, lds, per_cu};
        it = &slot;
    }
    // the caller's cap on resident traversal waves (RT_OPT_TRACE_WAVES_PER_CU) holds for this loop too
    const uint32_t wave_cap = ctx->tune_waves_per_cu ? ctx->tune_waves_per_cu : 32u;
    const uint32_t groups_per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(static_cast<uint32_t>(it->per_cu), wave_cap / WAVES));
    uint32_t grid = static_cast<uint32_t>(ctx->n_cus) * groups_per_cu;
    // a small batch: no more waves than blocks of 64 paths
    const uint32_t blocks_of_64 = (a.n_paths + 63u) / 64u;
    grid = std::max(1u, std::min(grid, (blocks_of_64 + WAVES - 1) / WAVES));
    const uint32_t n_waves = grid * WAVES;
    a.per_wave = ((blocks_of_64 + n_waves - 1) / n_waves) * 64u;
    a.t.spill_stride = grid * static_cast<uint32_t>(block);
    if (a.t.spill_stride > ctx->max_grid_threads) return fail(ctx, RT_ERR_INVALID, "bounce-loop grid larger than the stack overflow area");
    if (static_cast<uint64_t>(a.per_wave) * n_waves > ctx->ws.cap_paths + ctx->max_grid_threads) return fail(ctx, RT_ERR_INVALID, "path queues too short for the waves' blocks");
    if (ctx->debug_tallies)
        std::fprintf(stderr, "[rt_mi355] bounce loop: %u workgroups of %d waves (%d per CU), %zu bytes of LDS, %u paths per wave\n", grid,
                     WAVES, it->per_cu, lds, a.per_wave);
    // camera rays, dealt out to the waves in blocks of 64
    hipLaunchKernelGGL(k_raygen, dim3((a.n_paths + 255) / 256), dim3(256), 0, ctx->stream, f, device_scene(ctx), ctx->ws.q[0], 0u, a.n_paths, n_pix,
                       first_sample, 1, ctx->ws.counters, inline_counters(ctx), n_waves, a.per_wave);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->opt_time) HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used], ctx->stream));
    hipLaunchKernelGGL((k_path<WAVES, FASTSLAB>), dim3(grid), dim3(block), lds, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

// One batch of the bounce loop as a single launch (k_path).
int render_batch_fused(RtContext* ctx, const FrameArgs& f, const DScene& sc, uint32_t B, uint32_t n_paths, uint32_t n_pix,
                       unsigned long long first_sample) {
    Workspace& w = ctx->ws;
    HIP_TRY(ctx, hipMemsetAsync(w.counters, 0, 4 * sizeof(uint32_t), ctx->stream));
    PathArgs a{};
    a.t.fam = sc.family[RT_SHAPE_TRIANGLE];
    a.t.spill = ctx->ws.spill;
    a.t.lds_slots = ctx->tune_lds_slots;
    a.t.phase_min = ctx->tune_phase_min;
    a.t.chunk_max = ctx->tune_chunk_max;
    a.t.pair_min = ctx->tune_pair_min;
    a.t.refill_min = ctx->tune_refill_min;
    a.t.bounds_fast = ctx->fam[RT_SHAPE_TRIANGLE].bounds_fast ? 1 : 0;
    if (!ctx->d_path_consts) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_path_consts), sizeof(PathConsts)));
    PathConsts pc{};
    pc.scene = sc;
    pc.q[0] = w.q[0];
    pc.q[1] = w.q[1];
    pc.result = w.result;
    pc.hits = ctx->d_stats;
    pc.segments = ctx->d_stats + kStatSegments;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_path_consts, &pc, sizeof pc, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // `pc` leaves scope
    a.consts = (PathConstsPtr)ctx->d_path_consts;
    a.group = ctx->tune_path_group;
    a.n_paths = n_paths;
    a.max_bounces = B;
    if (ctx->opt_time) {
        while (ctx->events_used + 2 > ctx->events.size()) {
            hipEvent_t ev;
            HIP_TRY(ctx, hipEventCreate(&ev));
            ctx->events.push_back(ev);
        }
    }
    int rc;
    if (ctx->opt_fast_slab)
        rc = ctx->tune_waves == 16 ? launch_path_variant<16, true>(ctx, a, f, n_pix, first_sample)
                                   : launch_path_variant<8, true>(ctx, a, f, n_pix, first_sample);
    else
        rc = ctx->tune_waves == 16 ? launch_path_variant<16, false>(ctx, a, f, n_pix, first_sample)
                                   : launch_path_variant<8, false>(ctx, a, f, n_pix, first_sample);
    if (rc) return rc;
    if (ctx->opt_time) {
        HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used + 1], ctx->stream));
        ctx->events_used += 2;
        ctx->stats.trace_launches++;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return harvest(ctx);
}

// One batch of the bounce loop: samples [first_sample, first_sample + nb) of every pixel
// of the region, each path's final radiance left in the workspace's result array
// ([sample in batch][pixel]).  The workspace must hold nb * n_pixels paths.
int trace_batch(RtContext* ctx, const FrameArgs& f, const DScene& sc, uint32_t B, uint32_t n_pix, unsigned long long first_sample,
                uint32_t nb) {
    Workspace& w = ctx->ws;
    uint32_t* counts = w.counters;           // [0..B]
    uint32_t* cursors = w.counters + B + 1;  // [B][3]
    const size_t counter_words = B + 1 + 3 * static_cast<size_t>(B);
    const int shade_grid = ctx->n_cus * static_cast<int>(ctx->tune_shade_groups);
    const uint32_t n_paths = nb * n_pix;
    int rc = RT_OK;
    ctx->stats.paths += n_paths;
    // Batches below about 12 M paths run as one launch (k_path): 3.8x faster at 65 k paths, 2x at 2 M (one sample
    // of a 1080p frame), equal at 16 M.  Larger ones are faster launch by launch: shading needs far more
    // memory requests in flight than the few waves that shade at any moment inside k_path can have.
    // Above that size the per-bounce launches win on scenes whose rays are expensive (the shading of k_path's waves has
    // too few memory requests in flight); on scenes whose rays mostly miss or die young the one launch stays ahead
    // (the reference's boombox frame: 21.7 against 14.5 Grays/s), so the choice for large batches is measured once
    // per scene and frame (render_sums_device) rather than assumed.
    const RtContext::LargeLoop large = ctx->forced_loop != RtContext::LargeLoop::Unmeasured ? ctx->forced_loop : ctx->large_loop;
    if (B > 0 && fused_loop_possible(ctx) &&
        (ctx->opt_fused == 1 || (ctx->opt_fused == 2 && (n_paths < kFusedBelowPaths || large == RtContext::LargeLoop::OneLaunch))))
        return render_batch_fused(ctx, f, sc, B, n_paths, n_pix, first_sample);
    // large batches run as two overlapped half-batches (not while counting: the counting kernels are slow anyway)
    const bool overlapped = ctx->n_chains > 1 && !ctx->opt_count && B > 0 && n_paths >= (1u << 21);
    if (overlapped) {
        uint64_t overlapped_segments = 0;
        rc = render_batch_overlapped(ctx, f, sc, B, n_paths, n_pix, first_sample, &overlapped_segments);
        if (rc) return rc;
        ctx->stats.segments += overlapped_segments;
        return harvest(ctx);
    }
    HIP_TRY(ctx, hipMemsetAsync(counts, 0, std::max<size_t>(counter_words, 1) * sizeof(uint32_t), ctx->stream));
    if (B == 0) {
        // lib.rs:313,374: no bounce, no colour -> black
        HIP_TRY(ctx, hipMemsetAsync(w.result, 0, static_cast<size_t>(n_paths) * sizeof(float4), ctx->stream));
    } else {
        hipLaunchKernelGGL(k_raygen, dim3((n_paths + 255) / 256), dim3(256), 0, ctx->stream, f, sc, w.q[0], 0u, n_paths,
                           n_pix, first_sample, 1, counts, inline_counters(ctx));
        HIP_TRY(ctx, hipGetLastError());
    }
    for (uint32_t b = 0; b < B; ++b) {
        const DQueue& cur = w.q[b & 1];
        const DQueue& next = w.q[(b + 1) & 1];
        rc = launch_trace(ctx, cur, counts + b, cursors + 3 * static_cast<size_t>(b));
        if (rc) return rc;
        ShadeArgs sa{};
        sa.scene = sc;
        sa.cur = cur;
        sa.next = next;
        sa.n_cur = counts + b;
        sa.n_next = counts + b + 1;
        sa.result = w.result;
        sa.hits = ctx->d_stats;
        sa.test_counters = inline_counters(ctx);
        sa.last_bounce = b + 1 == B;
        hipLaunchKernelGGL(k_shade, dim3(shade_grid), dim3(256), 0, ctx->stream, sa);
        HIP_TRY(ctx, hipGetLastError());
        // every few bounces look whether any path is still alive
        if ((b & 3) == 3 && b + 1 < B) {
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, counts + b + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->h_pinned[0] == 0) break;
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, counts, (B + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t b = 0; b < B; ++b) ctx->stats.segments += ctx->h_pinned[b];
    return harvest(ctx);
}

// Workspace for up to `count` samples of the region at once (fewer when the
// automatic size does not fit the memory); returns the number that fit in *in_flight.
int workspace_for_samples(RtContext* ctx, const Region& r, uint32_t count, uint32_t B, size_t* in_flight) {
    size_t S = choose_samples_in_flight(ctx, r.n_pixels, count);
    int rc = ensure_workspace(ctx, S * r.n_pixels, r.n_pixels, B);
    while (rc == RT_ERR_OOM && S > 1 && ctx->opt_samples_in_flight == 0) {  // automatic size: take what fits
        S = (S + 1) / 2;  // (ensure_workspace releases the queues of the failed attempt first)
        rc = ensure_workspace(ctx, S * r.n_pixels, r.n_pixels, B);
    }
    *in_flight = S;
    return rc;
}

// Which loop do large batches of this scene and frame take?  Measured: the first two large batches are each
// rendered as two halves, one through each loop (both are the product's loops and give the same bits, so nothing
// is thrown away) — per-bounce launches first in the first batch, the one launch first in the second, so that
// what a first batch pays once (fresh queues, first launches) falls on each loop once — every half of at least
// 2^24 paths, above the size at which the one launch wins anyway.  Each loop's better time per sample decides, and
// the one launch must win by 8 % to be chosen; the choice holds until the scene, the bounce limit or the region's
// size changes.
//
// `first_batch`: samples the caller's next batch would take; `unit`: the halves are multiples of this many samples
// (a pass's samples where passes are finished one by one); `batch(n)`: traces the next n samples and does whatever
// the caller does with a traced batch.  When a measurement is due and the batch is large enough, the two halves are
// rendered here; otherwise nothing happens and the caller goes on as usual.
template <class Batch>
int measure_large_loop(RtContext* ctx, const Region& r, uint32_t B, uint32_t first_batch, uint32_t unit, Batch&& batch) {
    const uint64_t key = (static_cast<uint64_t>(B) << 40) ^ r.n_pixels;
    if (ctx->large_loop_key != key) {
        ctx->large_loop_key = key;
        ctx->large_loop = RtContext::LargeLoop::Unmeasured;
        ctx->pilot_phase = 0;
        ctx->pilot_ms_per_sample[0] = ctx->pilot_ms_per_sample[1] = 0.0;
    }
    const uint32_t half = first_batch / 2 / unit * unit;
    if (ctx->opt_fused != 2 || ctx->large_loop != RtContext::LargeLoop::Unmeasured || B == 0 || !fused_loop_possible(ctx) || half == 0 ||
        static_cast<uint64_t>(half) * r.n_pixels < (uint64_t{1} << 24))
        return RT_OK;
    for (int k = 0; k < 2; ++k) {
        const int loop = (k + static_cast<int>(ctx->pilot_phase)) & 1;  // 0 = per-bounce launches, 1 = one launch
        ctx->forced_loop = loop ? RtContext::LargeLoop::OneLaunch : RtContext::LargeLoop::PerBounce;
        Timer timer;  // (trace_batch returns with the batch's stream drained)
        const int rc = batch(half);
        ctx->forced_loop = RtContext::LargeLoop::Unmeasured;
        if (rc) return rc;
        const double ms = timer.ms() / half;
        double& best = ctx->pilot_ms_per_sample[loop];
        best = best > 0.0 ? std::min(best, ms) : ms;
    }
    if (++ctx->pilot_phase == 2) {
        const double* ms = ctx->pilot_ms_per_sample;
        ctx->large_loop = ms[1] < 0.92 * ms[0] ? RtContext::LargeLoop::OneLaunch : RtContext::LargeLoop::PerBounce;
        if (ctx->debug_tallies)
            std::fprintf(stderr, "[rt_mi355] large batches: per-bounce launches %.3f ms per sample, one launch %.3f -> %s\n", ms[0], ms[1],
                         ctx->large_loop == RtContext::LargeLoop::OneLaunch ? "one launch" : "per-bounce launches");
    }
    return RT_OK;
}

// The bounce loop over samples [s_begin, s_begin + count) of the region into
// the workspace's running sums (lib.rs:260-279 for every pixel).
// `initial` (optional): the fold of the samples before s_begin, which the fold goes on from.
int render_sums_device(RtContext* ctx, const RtRenderParams* p, const Region& r, uint32_t s_begin, uint32_t count,
                       const float* initial = nullptr, int initial_on_device = 0) {
    if (count == 0) return fail(ctx, RT_ERR_INVALID, "sample_count must be at least 1");
    const uint32_t B = p->max_bounces;
    size_t S = 0;
    int rc = workspace_for_samples(ctx, r, count, B, &S);
    if (rc) return rc;
    ctx->pending_samples = 0;  // the result array is about to be overwritten
    Workspace& w = ctx->ws;
    const FrameArgs f = frame_args(p, r);
    const DScene sc = device_scene(ctx);
    const uint32_t n_pix = static_cast<uint32_t>(r.n_pixels);
    if (initial)
        HIP_TRY(ctx, hipMemcpyAsync(w.sums, initial, r.n_pixels * 3 * sizeof(float),
                                    initial_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    uint32_t b0 = 0;
    auto batch = [&](uint32_t nb) -> int {
        const int r = trace_batch(ctx, f, sc, B, n_pix, static_cast<unsigned long long>(s_begin) + b0, nb);
        if (r) return r;
        hipLaunchKernelGGL(k_fold, dim3((n_pix + 255) / 256), dim3(256), 0, ctx->stream, w.result, w.sums, n_pix, nb,
                           b0 == 0 && !initial ? 1 : 0);
        HIP_TRY(ctx, hipGetLastError());
        b0 += nb;
        return RT_OK;
    };
    // the first large batches of a scene and frame also measure which loop suits them (measure_large_loop)
    rc = measure_large_loop(ctx, r, B, std::min<uint32_t>(static_cast<uint32_t>(S), count), 1u, batch);
    if (rc) return rc;
    while (b0 < count) {
        rc = batch(std::min<uint32_t>(static_cast<uint32_t>(S), count - b0));
        if (rc) return rc;
    }
    return RT_OK;
}

// Queue `q` seen from entry `first` on.
DQueue queue_from(const DQueue& q, size_t first) {
    DQueue r = q;
    r.origin += first;
    r.direction += first;
    r.throughput += first;
    r.rng += first;
    r.slot += first;
    r.hit_t += first;
    r.hit_id += first;
    return r;
}

// The same bounce loop for one batch, run as several independent chains.
//
// A chain of launches (trace, shade, trace, shade, ...) leaves the GPU partly idle
// at every step: the persistent traversal kernel ends with a tail in which the last
// waves finish their rays alone, a launch cannot begin before the previous one has
// wholly ended, and every few bounces the host looks at a counter.  Here the batch's
// paths are split into `n_chains` parts with queues and counters of their own, each
// on its own stream with nothing ordering them against each other, and each chain's
// traversal is held to its share of the compute units' wave slots so that the
// chains' kernels are resident side by side: one chain's tails and gaps are filled
// by the others, and memory-bound shading of one runs beside instruction-bound
// traversal of another.  Every path is still traced and shaded by the same kernels
// in the same order and its result lands in the slot of its number in the batch:
// nothing changes in what is computed.
int render_batch_overlapped(RtContext* ctx, const FrameArgs& f, const DScene& sc, uint32_t B, uint32_t n_paths,
                            uint32_t n_pix, unsigned long long first_sample, uint64_t* segments_out) {
    Workspace& w = ctx->ws;
    const int K = static_cast<int>(ctx->n_chains);
    const size_t block_words = B + 1 + 3 * static_cast<size_t>(B) + 4;
    uint32_t n_part[kMaxChains], first_path[kMaxChains];
    uint32_t* counts[kMaxChains];
    uint32_t* cursors[kMaxChains];
    DQueue q[kMaxChains][2];
    const uint32_t share = (n_paths / static_cast<uint32_t>(K)) & ~63u;
    for (int h = 0; h < K; ++h) {
        first_path[h] = share * static_cast<uint32_t>(h);
        n_part[h] = h + 1 < K ? share : n_paths - first_path[h];
        counts[h] = w.counters + block_words * static_cast<size_t>(h);
        cursors[h] = counts[h] + B + 1;
        for (int k = 0; k < 2; ++k) q[h][k] = queue_from(w.q[k], first_path[h]);
    }
    const int shade_grid = ctx->n_cus * static_cast<int>(ctx->tune_shade_groups);
    HIP_TRY(ctx, hipMemsetAsync(w.counters, 0, K * block_words * sizeof(uint32_t), ctx->chain_stream[0]));
    HIP_TRY(ctx, hipEventRecord(ctx->chain_event[0], ctx->chain_stream[0]));
    for (int h = 1; h < K; ++h) HIP_TRY(ctx, hipStreamWaitEvent(ctx->chain_stream[h], ctx->chain_event[0], 0));
    for (int h = 0; h < K; ++h) {
        hipLaunchKernelGGL(k_raygen, dim3((n_part[h] + 255) / 256), dim3(256), 0, ctx->chain_stream[h], f, sc, q[h][0],
                           first_path[h], n_part[h], n_pix, first_sample, 1, counts[h], inline_counters(ctx));
        HIP_TRY(ctx, hipGetLastError());
    }
    ctx->wave_cap_now = std::max(8u, (32u / static_cast<uint32_t>(K)) & ~7u);
    int rc = RT_OK;
    bool alive[kMaxChains] = {true, true, true, true};
    bool any_alive = true;
    for (uint32_t b = 0; b < B && rc == RT_OK && any_alive; ++b) {
        for (int h = 0; h < K && rc == RT_OK; ++h) {
            if (!alive[h]) continue;
            ctx->launch_stream = ctx->chain_stream[h];
            ctx->launch_chain = static_cast<uint32_t>(h);
            rc = launch_trace(ctx, q[h][b & 1], counts[h] + b, cursors[h] + 3 * static_cast<size_t>(b));
            if (rc != RT_OK) break;
            ShadeArgs sa{};
            sa.scene = sc;
            sa.cur = q[h][b & 1];
            sa.next = q[h][(b + 1) & 1];
            sa.n_cur = counts[h] + b;
            sa.n_next = counts[h] + b + 1;
            sa.result = w.result;
            sa.hits = ctx->d_stats;
            sa.test_counters = inline_counters(ctx);
            sa.last_bounce = b + 1 == B;
            hipLaunchKernelGGL(k_shade, dim3(shade_grid), dim3(256), 0, ctx->chain_stream[h], sa);
            if (hipGetLastError() != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "shade launch failed");
        }
        // every few bounces look whether any path is still alive
        if (rc == RT_OK && (b & 7) == 7 && b + 1 < B) {
            for (int h = 0; h < K; ++h)
                if (hipMemcpyAsync(ctx->h_pinned + h, counts[h] + b + 1, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                   ctx->chain_stream[h]) != hipSuccess)
                    rc = fail(ctx, RT_ERR_HIP, "count read-back failed");
            for (int h = 0; h < K; ++h)
                if (hipStreamSynchronize(ctx->chain_stream[h]) != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "stream synchronisation failed");
            any_alive = false;
            for (int h = 0; h < K && rc == RT_OK; ++h) {
                alive[h] = alive[h] && ctx->h_pinned[h] != 0;
                any_alive = any_alive || alive[h];
            }
        }
    }
    ctx->launch_stream = ctx->stream;
    ctx->launch_chain = 0;
    ctx->wave_cap_now = 0;
    if (rc != RT_OK) {
        for (int h = 0; h < K; ++h) (void)hipStreamSynchronize(ctx->chain_stream[h]);
        return rc;
    }
    // the first chain's stream (the context's own) goes on once the others have finished
    for (int h = 1; h < K; ++h) {
        HIP_TRY(ctx, hipEventRecord(ctx->chain_event[h], ctx->chain_stream[h]));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->chain_event[h], 0));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pinned, w.counters, K * block_words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t segments = 0;
    for (int h = 0; h < K; ++h)
        for (uint32_t b = 0; b < B; ++b) segments += ctx->h_pinned[h * block_words + b];
    *segments_out = segments;
    return RT_OK;
}

int quantize_device(RtContext* ctx, const float* d_sums, size_t n_pixels, uint32_t n_samples, int running, uint32_t k) {
    const uint32_t n_values = static_cast<uint32_t>(n_pixels * 3);
    hipLaunchKernelGGL(k_quantize, dim3((n_values + 255) / 256), dim3(256), 0, ctx->stream, d_sums, ctx->ws.rgb8,
                       n_values, n_samples, running, k);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}


int require_scene(RtContext* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RT_OK;
}

// Hit queue entries -> RtHitRecord on the host.
int download_hits(RtContext* ctx, const DQueue& q, size_t n, RtHitRecord* out) {
    std::vector<float> t(n);
    std::vector<uint32_t> id(n);
    HIP_TRY(ctx, hipMemcpy(t.data(), q.hit_t, n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(id.data(), q.hit_id, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
        out[i].t = t[i];
        if (id[i] == kHitNone) {
            out[i].kind = RT_SHAPE_NONE;
            out[i].prim = RT_NO_INDEX;
        } else {
            out[i].kind = id[i] >> kHitKindShift;
            out[i].prim = id[i] & kHitPrimMask;
        }
    }
    return RT_OK;
}

bool create_chains(RtContext* ctx) {
    for (int h = 1; h < kMaxChains; ++h)
        if (hipStreamCreateWithFlags(&ctx->chain_stream[h], hipStreamNonBlocking) != hipSuccess) return false;
    for (int h = 0; h < kMaxChains; ++h)
        if (hipEventCreateWithFlags(&ctx->chain_event[h], hipEventDisableTiming) != hipSuccess) return false;
    return true;
}

}  // namespace

// For the other translation units of this library (bvh_build.hip): record an
// error on the context; make the context's GPU current and hand out its stream.
int rt_context_fail(RtContext* ctx, int code, const std::string& msg) { return fail(ctx, code, msg); }
int rt_abi_caught(RtContext* ctx, bool out_of_memory) noexcept {
    // the message buffer was reserved at creation: a literal of this length is stored without allocating
    try {
        if (ctx) ctx->error.assign(out_of_memory ? "out of host memory" : "internal error (an exception was caught at the C boundary)");
    } catch (...) {
    }
    return out_of_memory ? RT_ERR_OOM : RT_ERR_HIP;
}
int rt_context_device(RtContext* ctx, hipStream_t* stream) {
    if (!ctx) return RT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *stream = ctx->stream;
    return RT_OK;
}

extern "C" {

uint32_t rt_abi_version(void) try { return RT_ABI_VERSION; } RT_ABI_END_VALUE(0u)

// Message of the last failed rt_create of this thread (there is no context to hold it).
static thread_local char g_create_error[512] = "";

// Number of HIP runtimes mapped into this process, their paths in `paths`.
static int count_hip_runtimes(char* paths, size_t cap) {
    struct Found {
        int n;
        char* paths;
        size_t cap;
    } found{0, paths, cap};
    paths[0] = 0;
    dl_iterate_phdr(
        [](struct dl_phdr_info* info, size_t, void* data) -> int {
            Found* f = static_cast<Found*>(data);
            const char* name = info->dlpi_name ? info->dlpi_name : "";
            const char* base = std::strrchr(name, '/');
            base = base ? base + 1 : name;
            if (std::strncmp(base, "libamdhip64.so", 14) == 0) {
                ++f->n;
                const size_t used = std::strlen(f->paths);
                std::snprintf(f->paths + used, f->cap - used, "%s%s", used ? ", " : "", name);
            }
            return 0;
        },
        &found);
    return found.n;
}

int rt_create(int device_ordinal, RtContext** out) try {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    g_create_error[0] = 0;
    {
        // Two HIP runtimes in one process (typically ROCm's, which this library is linked against,
        // and the copy a Python package such as PyTorch brings) do not share devices, streams or
        // allocations: a device pointer or stream of one is garbage to the other.  Refuse to start.
        char paths[400];
        if (count_hip_runtimes(paths, sizeof paths) > 1) {
            std::snprintf(g_create_error, sizeof g_create_error,
                          "two HIP runtimes are mapped in this process (%s): load the one the process's other GPU "
                          "code uses before librt_mi355.so (in Python: import torch first, or load the library "
                          "through raytracer_amd._ffi, which does this)", paths);
            return RT_ERR_HIP;
        }
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return RT_ERR_NO_DEVICE;
    RtContext* ctx = new (std::nothrow) RtContext;
    if (!ctx) return RT_ERR_OOM;
    ctx->device = device_ordinal;
    hipDeviceProp_t prop;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        !create_chains(ctx) ||
        hipMalloc(reinterpret_cast<void**>(&ctx->d_stats), kStatWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->d_stats, 0, kStatWords * sizeof(unsigned long long)) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    // this GPU's copy of the scratch array (the device is current here)
    void* scratch_ptr = nullptr;
    if (hipGetSymbolAddress(&scratch_ptr, HIP_SYMBOL(scratch)) != hipSuccess || !scratch_ptr) {
        (void)hipGetLastError();
        rt_destroy(ctx);
        return RT_ERR_HIP;
    }
    ctx->scratch = static_cast<unsigned char*>(scratch_ptr);
    ctx->error.reserve(256);  // later messages of that length are stored without allocating
    ctx->events.reserve(4096);  // timed launches take their event pairs from here without growing it
    ctx->d_inline_tests = ctx->d_stats + 7;
    ctx->launch_stream = ctx->stream;
    ctx->chain_stream[0] = ctx->stream;
    ctx->n_cus = prop.multiProcessorCount;
    ctx->max_grid_threads = static_cast<uint32_t>(ctx->n_cus) * 32u * kTraceBlock;  // 32 waves per CU at most
    *out = ctx;
    return RT_OK;
} RT_ABI_END_INT

void rt_destroy(RtContext* ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    free_workspace(ctx);
    for (int h = 1; h < kMaxChains; ++h)
        if (ctx->chain_stream[h]) (void)hipStreamSynchronize(ctx->chain_stream[h]);
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    for (int h = 0; h < kMaxChains; ++h)
        if (ctx->chain_event[h]) (void)hipEventDestroy(ctx->chain_event[h]);
    for (int h = 1; h < kMaxChains; ++h)
        if (ctx->chain_stream[h]) (void)hipStreamDestroy(ctx->chain_stream[h]);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_path_consts);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
} RT_ABI_END_VOID

const char* rt_last_error(const RtContext* ctx) try {
    if (ctx) return ctx->error.c_str();
    return g_create_error[0] ? g_create_error : "null context";
} RT_ABI_END_VALUE("internal error")

int rt_set_option(RtContext* ctx, uint32_t option, uint64_t value) try {
    if (!ctx) return RT_ERR_INVALID;
    switch (option) {
        case RT_OPT_COUNT_TRAVERSAL: ctx->opt_count = value != 0; return RT_OK;
        case RT_OPT_TIME_KERNELS: ctx->opt_time = value != 0; return RT_OK;
        case RT_OPT_FAST_SLAB: ctx->opt_fast_slab = value != 0; return RT_OK;
        case RT_OPT_SAMPLES_IN_FLIGHT: ctx->opt_samples_in_flight = value; return RT_OK;
        case RT_OPT_TRACE_LDS_SLOTS:
            if (value < 2 || value > 64) return fail(ctx, RT_ERR_INVALID, "lds slots must be in 2..64");
            ctx->tune_lds_slots = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_PHASE_MIN:
            if (value < 1 || value > 64) return fail(ctx, RT_ERR_INVALID, "phase threshold must be in 1..64");
            ctx->tune_phase_min = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_SHARE: ctx->tune_share = value != 0; return RT_OK;
        case RT_OPT_TRACE_WAVES_PER_CU:
            if (value > 32) return fail(ctx, RT_ERR_INVALID, "a compute unit holds at most 32 waves");
            ctx->tune_waves_per_cu = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_SHADE_GROUPS_PER_CU:
            if (value < 1 || value > 64) return fail(ctx, RT_ERR_INVALID, "shading workgroups per CU must be in 1..64");
            ctx->tune_shade_groups = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_CHAINS:
            if (value < 1 || value > kMaxChains) return fail(ctx, RT_ERR_INVALID, "chains per batch must be in 1..4");
            ctx->n_chains = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_REFILL_MIN:
            if (value < 1 || value > 64) return fail(ctx, RT_ERR_INVALID, "refill threshold must be in 1..64");
            ctx->tune_refill_min = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_WAVES:
            if (value != 1 && value != 8 && value != 16) return fail(ctx, RT_ERR_INVALID, "waves per workgroup must be 1, 8 or 16");
            ctx->tune_waves = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_CACHE_NODES:
            if (value > 2048) return fail(ctx, RT_ERR_INVALID, "at most 2048 branch records can be kept in LDS");
            ctx->tune_cache_nodes = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_PAIR_MIN:
            if (value < 1 || value > 4096) return fail(ctx, RT_ERR_INVALID, "pair threshold must be in 1..4096");
            ctx->tune_pair_min = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_CHUNK:
            if (value < 64 || value > 4096 || value % 64) return fail(ctx, RT_ERR_INVALID, "chunk must be a multiple of 64 in 64..4096");
            ctx->tune_chunk_max = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_PATH_GROUP:
            if (value < 64 || value > (1u << 24) || value % 64) return fail(ctx, RT_ERR_INVALID, "path group must be a multiple of 64 in 64..2^24");
            ctx->tune_path_group = static_cast<uint32_t>(value);
            return RT_OK;
        case RT_OPT_TRACE_GUIDED: ctx->tune_guided = value != 0; return RT_OK;
        case RT_OPT_PASSES_IN_FLIGHT: ctx->opt_passes_in_flight = value; return RT_OK;
        case RT_OPT_DEBUG_THROW:
            // test hook of the boundary's exception guard: the exception leaves through this very function
            if (value == 1) throw std::bad_alloc();
            if (value == 2) throw std::runtime_error("RT_OPT_DEBUG_THROW");
            return RT_OK;
        case RT_OPT_FUSED_LOOP:
            if (value > 2) return fail(ctx, RT_ERR_INVALID, "fused bounce loop: 0 (never), 1 (whenever possible) or 2 (for small batches)");
            ctx->opt_fused = static_cast<uint32_t>(value);
            return RT_OK;
        default: return fail(ctx, RT_ERR_INVALID, "unknown option");
    }
} RT_ABI_END_CTX(ctx)

// Words a texture takes on the device: whole tiles (see layout.h).
static uint64_t tiled_texels(uint32_t width, uint32_t height) {
    return ((static_cast<uint64_t>(width) + kTexTileW - 1) / kTexTileW) * ((static_cast<uint64_t>(height) + kTexTileH - 1) / kTexTileH) *
           kTexTileTexels;
}

int rt_upload_scene(RtContext* ctx, const RtSceneDesc* s) try {
    if (!ctx || !s) return RT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    ctx->pending_samples = 0;  // samples traced against the previous scene are not folded into a new one's sums
    ctx->large_loop = RtContext::LargeLoop::Unmeasured;
    ctx->large_loop_key = 0;  // (the next render starts the measurement afresh)

    // ---- validation of everything a kernel indexes with
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_triangles && !s->triangles) ||
        (s->n_materials && !s->materials) || (s->n_vertex_normals && !s->vertex_normals) ||
        (s->n_tex_coords && !s->tex_coords) || (s->n_bary && !s->bary_precomputed) || (s->n_texel_bytes && !s->texels))
        return fail(ctx, RT_ERR_INVALID, "null array with a non-zero count");
    if (s->n_spheres > kHitPrimMask || s->n_planes > kHitPrimMask || s->n_triangles > kHitPrimMask)
        return fail(ctx, RT_ERR_INVALID, "too many shapes");
    // Spheres and planes take solid colours only: the scene file cannot give them a texture
    // (config.rs:147-155), so their texture coordinates (shapes.rs:91-99,157-165) are not computed.
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        if (s->spheres[i].material >= s->n_materials) return fail(ctx, RT_ERR_INVALID, "sphere material index out of range");
        if (s->materials[s->spheres[i].material].color_kind == RT_COLOR_TEXTURE)
            return fail(ctx, RT_ERR_INVALID, "textured materials are supported on triangles only (sphere)");
    }
    for (uint32_t i = 0; i < s->n_planes; ++i) {
        if (s->planes[i].material >= s->n_materials) return fail(ctx, RT_ERR_INVALID, "plane material index out of range");
        if (s->materials[s->planes[i].material].color_kind == RT_COLOR_TEXTURE)
            return fail(ctx, RT_ERR_INVALID, "textured materials are supported on triangles only (plane)");
    }
    for (uint32_t i = 0; i < s->n_triangles; ++i) {
        const RtTriangle& t = s->triangles[i];
        if (t.material >= s->n_materials) return fail(ctx, RT_ERR_INVALID, "triangle material index out of range");
        const bool any = t.normals_idx != RT_NO_INDEX || t.uv_idx != RT_NO_INDEX;
        if ((t.normals_idx != RT_NO_INDEX && t.normals_idx >= s->n_vertex_normals) ||
            (t.uv_idx != RT_NO_INDEX && t.uv_idx >= s->n_tex_coords) ||
            (t.bary_idx != RT_NO_INDEX && t.bary_idx >= s->n_bary) || (any && t.bary_idx == RT_NO_INDEX))
            return fail(ctx, RT_ERR_INVALID, "triangle side-table index out of range");
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const RtMaterial& m = s->materials[i];
        if (m.kind > RT_MAT_LIGHT || m.color_kind > RT_COLOR_TEXTURE) return fail(ctx, RT_ERR_INVALID, "bad material kind");
        if (m.color_kind == RT_COLOR_TEXTURE) {
            const uint64_t texels = static_cast<uint64_t>(m.tex_width) * m.tex_height;  // (no overflow: two 32-bit factors)
            if (m.tex_width == 0 || m.tex_height == 0 || m.tex_offset > s->n_texel_bytes ||
                texels > (s->n_texel_bytes - m.tex_offset) / 3)
                return fail(ctx, RT_ERR_INVALID, "texture outside the texel array");
        }
    }

    // ---- acceleration structures
    const RtNode* node_arrays[3] = {s->sphere_nodes, s->plane_nodes, s->triangle_nodes};
    const uint32_t node_counts[3] = {s->n_sphere_nodes, s->n_plane_nodes, s->n_triangle_nodes};
    const uint32_t shape_counts[3] = {s->n_spheres, s->n_planes, s->n_triangles};
    const char* names[3] = {"spheres", "planes", "triangles"};
    uint32_t deepest = 1;
    for (int k = 0; k < 3; ++k) {
        std::vector<DNode> branches;
        std::vector<DLongLeaf> longs;
        int rc = build_family(ctx, names[k], node_arrays[k], node_counts[k], shape_counts[k], ctx->fam[k], branches, longs);
        if (rc) { free_scene(ctx); return rc; }
        rc = upload(ctx, &ctx->fam[k].nodes, branches.data(), branches.size());
        if (!rc) rc = upload(ctx, &ctx->fam[k].long_leaves, longs.data(), longs.size());
        if (rc) { free_scene(ctx); return rc; }
        deepest = std::max(deepest, ctx->fam[k].max_stack);
    }
    ctx->deepest_stack = deepest;

    // ---- shapes
    {
        std::vector<DSphere> sp(s->n_spheres);
        std::vector<uint32_t> sm(s->n_spheres);
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            std::memcpy(sp[i].center, s->spheres[i].center, 12);
            sp[i].radius = s->spheres[i].radius;
            sm[i] = s->spheres[i].material;
        }
        std::vector<DPlane> pl(s->n_planes);
        std::vector<uint32_t> pm(s->n_planes);
        for (uint32_t i = 0; i < s->n_planes; ++i) {
            std::memcpy(pl[i].point, s->planes[i].point, 12);
            std::memcpy(pl[i].normal, s->planes[i].normal, 12);
            pl[i].pad0 = pl[i].pad1 = 0.0f;
            pm[i] = s->planes[i].material;
        }
        // the traversal's three triangle arrays, one after the other (layout.h)
        const size_t nt = s->n_triangles;
        std::vector<unsigned char> tr(triangle_array_bytes(nt));
        DTriHalf* const tr0 = reinterpret_cast<DTriHalf*>(tr.data());
        DTriHalf* const tr1 = tr0 + nt;
        float* const trz = reinterpret_cast<float*>(tr1 + nt);
        std::vector<DTriSurface> ts(s->n_triangles);
        for (uint32_t i = 0; i < s->n_triangles; ++i) {
            const RtTriangle& t = s->triangles[i];
            tr0[i] = DTriHalf{{t.e2[0], t.e1[0], t.e2[1], t.e1[1]}};  // ex, ey
            tr1[i] = DTriHalf{{t.e2[2], t.e1[2], t.a[0], t.a[1]}};    // ez, a.x, a.y
            trz[i] = t.a[2];
            DTriSurface& d = ts[i];
            std::memset(&d, 0, sizeof d);
            std::memcpy(d.a, t.a, 12);
            std::memcpy(d.e1, t.e1, 12);
            std::memcpy(d.e2, t.e2, 12);
            if (t.material > kTriMaterialMask) { free_scene(ctx); return fail(ctx, RT_ERR_INVALID, "too many materials"); }
            d.material_flags = t.material;
            if (t.bary_idx != RT_NO_INDEX) {
                const float* b = s->bary_precomputed + 4 * static_cast<size_t>(t.bary_idx);
                d.d00 = b[0]; d.d01 = b[1]; d.d11 = b[2]; d.den = b[3];
                d.material_flags |= kTriHasBary;
            }
            if (t.normals_idx != RT_NO_INDEX) {
                const float* n = s->vertex_normals + 9 * static_cast<size_t>(t.normals_idx);
                std::memcpy(d.n0, n, 12);
                std::memcpy(d.n1, n + 3, 12);
                std::memcpy(d.n2, n + 6, 12);
                d.material_flags |= kTriHasNormals;
            }
            if (t.uv_idx != RT_NO_INDEX) {
                const float* c = s->tex_coords + 6 * static_cast<size_t>(t.uv_idx);
                std::memcpy(d.uv0, c, 8);
                std::memcpy(d.uv1, c + 2, 8);
                std::memcpy(d.uv2, c + 4, 8);
                d.material_flags |= kTriHasUv;
            }
        }
        // textures: one 32-bit word per texel, each texture contiguous
        std::vector<DMaterial> mats(s->n_materials);
        uint64_t n_texels = 0;
        for (uint32_t i = 0; i < s->n_materials; ++i) {
            const RtMaterial& m = s->materials[i];
            mats[i] = DMaterial{m.kind, m.param, m.color_kind, {m.rgb[0], m.rgb[1], m.rgb[2]}, m.tex_width, m.tex_height, 0};
            if (m.color_kind == RT_COLOR_TEXTURE) {
                mats[i].tex_first = n_texels;
                n_texels += tiled_texels(m.tex_width, m.tex_height);
            }
        }
        {
            uint8_t* d_rgb8 = nullptr;
            int rc = upload(ctx, &d_rgb8, s->texels, static_cast<size_t>(s->n_texel_bytes));
            if (!rc) {
                hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->texels), std::max<uint64_t>(n_texels, 1) * sizeof(uint32_t));
                if (e != hipSuccess) rc = fail(ctx, RT_ERR_OOM, "texture memory");
            }
            for (uint32_t i = 0; !rc && i < s->n_materials; ++i) {
                const RtMaterial& m = s->materials[i];
                if (m.color_kind != RT_COLOR_TEXTURE) continue;
                const uint64_t nt = tiled_texels(m.tex_width, m.tex_height);
                if ((nt + 255) / 256 > 0x7FFFFFFFull) { rc = fail(ctx, RT_ERR_INVALID, "texture too large"); break; }
                hipLaunchKernelGGL(k_expand_texels, dim3(static_cast<uint32_t>((nt + 255) / 256)), dim3(256), 0, nullptr,
                                   d_rgb8 + m.tex_offset, ctx->texels + mats[i].tex_first, m.tex_width, m.tex_height);
                if (hipGetLastError() != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "texture expansion launch failed");
            }
            if (hipDeviceSynchronize() != hipSuccess && !rc) rc = fail(ctx, RT_ERR_HIP, "texture expansion failed");
            (void)hipFree(d_rgb8);
            if (rc) { free_scene(ctx); return rc; }
        }
        DSphere* dsp = nullptr;
        DPlane* dpl = nullptr;
        unsigned char* dtr = nullptr;
        int rc = upload(ctx, &dsp, sp.data(), sp.size());
        ctx->fam[0].shapes = dsp;
        if (!rc) { rc = upload(ctx, &dpl, pl.data(), pl.size()); ctx->fam[1].shapes = dpl; }
        if (!rc) { rc = upload(ctx, &dtr, tr.data(), tr.size()); ctx->fam[2].shapes = dtr; }
        if (!rc) rc = upload(ctx, &ctx->sphere_material, sm.data(), sm.size());
        if (!rc) rc = upload(ctx, &ctx->plane_material, pm.data(), pm.size());
        if (!rc) rc = upload(ctx, &ctx->tri_surface, ts.data(), ts.size());
        if (!rc) rc = upload(ctx, &ctx->materials, mats.data(), mats.size());
        if (rc) { free_scene(ctx); return rc; }
        ctx->n_materials = s->n_materials;
        // Sphere / plane families that are one small leaf are scanned where rays are
        // created instead of getting traversal launches of their own.
        constexpr uint32_t kInlineMax = 16;
        for (int k = 0; k < 2; ++k) {
            const bool single_leaf = node_counts[k] == 1 && (node_arrays[k][0].b & RT_LEAF_BIT) && node_arrays[k][0].a == 0 &&
                                     (node_arrays[k][0].b & ~RT_LEAF_BIT) == shape_counts[k];
            ctx->inline_count[k] = single_leaf && shape_counts[k] <= kInlineMax ? shape_counts[k] : 0;
        }
        // A seeded plane hit would be found before the sphere family's traversal has run, and an
        // equal distance must go to the sphere (lib.rs:333 keeps the first family): planes are
        // scanned at ray creation only when no sphere traversal comes after it.
        if (shape_counts[0] != 0 && ctx->inline_count[0] == 0) ctx->inline_count[1] = 0;
    }
    HIP_TRY(ctx, hipDeviceSynchronize());  // uploads above went through the default stream

    // ---- stack overflow area for trees deeper than the LDS part of the stack
    {
        Workspace& w = ctx->ws;
        // sized for the whole stack so that any LDS share (down to none) is safe
        const size_t entries = std::max<size_t>(1, static_cast<size_t>(deepest) * ctx->max_grid_threads);
        if (w.spill_entries < entries) {
            (void)hipFree(w.spill);
            w.spill = nullptr;
            w.spill_entries = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&w.spill), kMaxChains * entries * sizeof(uint2)));
            w.spill_entries = entries;
        }
    }
    ctx->has_scene = true;
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_render_sums(RtContext* ctx, const RtRenderParams* p, uint32_t sample_begin, uint32_t sample_count, float* sums,
                   int sums_on_device) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!sums) return fail(ctx, RT_ERR_INVALID, "null output");
    Timer timer;
    rc = render_sums_device(ctx, p, r, sample_begin, sample_count);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(sums, ctx->ws.sums, r.n_pixels * 3 * sizeof(float),
                                sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.render_ms += timer.ms();
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_render_sums_onto(RtContext* ctx, const RtRenderParams* p, uint32_t sample_begin, uint32_t sample_count, float* sums,
                        int sums_on_device) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!sums) return fail(ctx, RT_ERR_INVALID, "null accumulator");
    Timer timer;
    rc = render_sums_device(ctx, p, r, sample_begin, sample_count, sample_begin ? sums : nullptr, sums_on_device);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(sums, ctx->ws.sums, r.n_pixels * 3 * sizeof(float),
                                sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stats.render_ms += timer.ms();
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_trace_samples(RtContext* ctx, const RtRenderParams* p, uint32_t sample_begin, uint32_t sample_count, uint32_t* traced) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!traced || sample_count == 0) return fail(ctx, RT_ERR_INVALID, "bad arguments");
    Timer timer;
    size_t S = 0;
    rc = workspace_for_samples(ctx, r, sample_count, p->max_bounces, &S);
    if (rc) return rc;
    ctx->pending_samples = 0;
    const uint32_t nb = std::min<uint32_t>(static_cast<uint32_t>(S), sample_count);
    rc = trace_batch(ctx, frame_args(p, r), device_scene(ctx), p->max_bounces, static_cast<uint32_t>(r.n_pixels), sample_begin, nb);
    if (rc) return rc;
    const uint32_t rect[6] = {r.x0, r.y0, r.w, r.h, p->width, p->height};
    std::copy(rect, rect + 6, ctx->pending_rect);
    ctx->pending_samples = nb;
    *traced = nb;
    ctx->stats.render_ms += timer.ms();
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_reserve_samples(RtContext* ctx, const RtRenderParams* p, uint32_t sample_count, uint32_t* in_flight) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!in_flight || sample_count == 0) return fail(ctx, RT_ERR_INVALID, "bad arguments");
    size_t S = 0;
    rc = workspace_for_samples(ctx, r, sample_count, p->max_bounces, &S);
    if (rc) return rc;
    *in_flight = std::min<uint32_t>(static_cast<uint32_t>(S), sample_count);
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_fold_samples(RtContext* ctx, const RtRenderParams* p, int carry_on, float* sums, int sums_on_device) try {
    if (!ctx) return RT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Region r;
    int rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!sums) return fail(ctx, RT_ERR_INVALID, "null accumulator");
    const uint32_t rect[6] = {r.x0, r.y0, r.w, r.h, p->width, p->height};
    if (ctx->pending_samples == 0 || !std::equal(rect, rect + 6, ctx->pending_rect))
        return fail(ctx, RT_ERR_STATE, "no traced samples of this region are waiting to be folded");
    Timer timer;
    Workspace& w = ctx->ws;
    const hipMemcpyKind in = sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out = sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (carry_on) HIP_TRY(ctx, hipMemcpyAsync(w.sums, sums, r.n_pixels * 3 * sizeof(float), in, ctx->stream));
    const uint32_t n_pix = static_cast<uint32_t>(r.n_pixels);
    hipLaunchKernelGGL(k_fold, dim3((n_pix + 255) / 256), dim3(256), 0, ctx->stream, w.result, w.sums, n_pix, ctx->pending_samples,
                       carry_on ? 0 : 1);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(sums, w.sums, r.n_pixels * 3 * sizeof(float), out, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pending_samples = 0;
    ctx->stats.render_ms += timer.ms();
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_resolve(RtContext* ctx, const RtRenderParams* p, const float* sums, int sums_on_device, uint32_t n_samples,
               int running_average, uint32_t sample_iteration, uint8_t* rgb8) try {
    if (!ctx) return RT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Region r;
    int rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!sums || !rgb8 || n_samples == 0) return fail(ctx, RT_ERR_INVALID, "bad resolve arguments");
    rc = ensure_workspace(ctx, 0, r.n_pixels, 0);
    if (rc) return rc;
    Workspace& w = ctx->ws;
    HIP_TRY(ctx, hipMemcpyAsync(w.sums, sums, r.n_pixels * 3 * sizeof(float),
                                sums_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (running_average) HIP_TRY(ctx, hipMemcpyAsync(w.rgb8, rgb8, r.n_pixels * 3, hipMemcpyHostToDevice, ctx->stream));
    rc = quantize_device(ctx, w.sums, r.n_pixels, n_samples, running_average, sample_iteration);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(rgb8, w.rgb8, r.n_pixels * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_render(RtContext* ctx, const RtRenderParams* p, uint8_t* rgb8, RtProgressFn progress, void* user) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!rgb8) return fail(ctx, RT_ERR_INVALID, "null pixel buffer");
    // lib.rs:206-211
    const uint32_t chunk = p->incremental ? p->incremental : p->samples_per_pixel;
    if (chunk == 0 || p->samples_per_pixel % chunk != 0)
        return fail(ctx, RT_ERR_INVALID, "incremental must be non-zero and divide the samples per pixel");
    const uint32_t n_passes = p->samples_per_pixel / chunk;
    const int running = p->incremental != 0 || p->has_continue != 0;  // lib.rs:286
    const uint32_t first_pass = p->has_continue ? p->continue_sampling : 0;
    Timer timer;
    rc = ensure_workspace(ctx, 0, r.n_pixels, 0);
    if (rc) return rc;
    if (running) HIP_TRY(ctx, hipMemcpyAsync(ctx->ws.rgb8, rgb8, r.n_pixels * 3, hipMemcpyHostToDevice, ctx->stream));
    // Passes of few samples (the reference's shipped scenes use `incremental(1)`) are short batches of paths,
    // which render at a fraction of the rate of long ones.  What a pass depends on in the pass before it is
    // only the stored bytes it blends into (lib.rs:286-292), not its paths: the paths of several passes are
    // therefore traced together, and the passes are then finished one after the other from the per-sample
    // radiance — fold of the pass's samples, mean, gamma, running average, quantisation — in pass order, which
    // is the same arithmetic on the same values as pass by pass.  The image is rewritten after each such group of
    // passes (at most 2^26 paths, a fraction of a second) and progress is reported for each of its passes.
    const uint64_t pass_paths = static_cast<uint64_t>(chunk) * r.n_pixels;
    uint32_t group = 1;
    if (n_passes > 1 && p->max_bounces > 0 && pass_paths < (uint64_t{1} << 26)) {
        const uint64_t budget = ctx->opt_passes_in_flight ? ctx->opt_passes_in_flight : (uint64_t{1} << 26) / pass_paths;
        group = static_cast<uint32_t>(std::max<uint64_t>(1, std::min<uint64_t>(budget, n_passes)));
    }
    const uint32_t n_pix = static_cast<uint32_t>(r.n_pixels);
    for (uint32_t pass = 0; pass < n_passes;) {
        const uint32_t k = pass + first_pass;  // sample_iteration, lib.rs:233-234
        uint32_t together = std::min(group, n_passes - pass);
        if ((static_cast<uint64_t>(k) + together) * chunk > 0xFFFFFFFFull) return fail(ctx, RT_ERR_INVALID, "sample index overflow");
        if (together > 1) {
            size_t S = 0;
            rc = workspace_for_samples(ctx, r, together * chunk, p->max_bounces, &S);
            if (rc) return rc;
            together = static_cast<uint32_t>(S / chunk);  // whole passes of what fits in flight
        }
        if (together <= 1) {
            rc = render_sums_device(ctx, p, r, k * chunk, chunk);
            if (rc) return rc;
            rc = quantize_device(ctx, ctx->ws.sums, r.n_pixels, chunk, running, k);
            if (rc) return rc;
            together = 1;
        } else {
            ctx->pending_samples = 0;  // the result array is about to be overwritten
            // `n` samples (whole passes) from pass k + done on: traced together, then finished pass by pass
            uint32_t done = 0;
            auto passes = [&](uint32_t n) -> int {
                const uint32_t n_now = n / chunk;
                int e = trace_batch(ctx, frame_args(p, r), device_scene(ctx), p->max_bounces, n_pix,
                                    (static_cast<unsigned long long>(k) + done) * chunk, n);
                if (e) return e;
                for (uint32_t j = 0; j < n_now; ++j) {
                    hipLaunchKernelGGL(k_fold, dim3((n_pix + 255) / 256), dim3(256), 0, ctx->stream,
                                       ctx->ws.result + static_cast<size_t>(j) * chunk * n_pix, ctx->ws.sums, n_pix, chunk, 1);
                    HIP_TRY(ctx, hipGetLastError());
                    e = quantize_device(ctx, ctx->ws.sums, r.n_pixels, chunk, running, k + done + j);
                    if (e) return e;
                }
                done += n_now;
                return RT_OK;
            };
            rc = measure_large_loop(ctx, r, p->max_bounces, together * chunk, chunk, passes);
            if (rc) return rc;
            if (done < together) rc = passes((together - done) * chunk);
            if (rc) return rc;
        }
        // the image file is readable after every group of passes (live preview)
        HIP_TRY(ctx, hipMemcpyAsync(rgb8, ctx->ws.rgb8, r.n_pixels * 3, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t j = 0; j < together; ++j)
            if (progress) progress(first_pass + (pass + j + 1) * chunk, user);  // lib.rs:238-246
        pass += together;
    }
    ctx->stats.render_ms += timer.ms();
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_trace_primary(RtContext* ctx, const RtRenderParams* p, uint32_t sample_index, int jitter, RtHitRecord* out) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    Region r;
    rc = check_params(ctx, p, r);
    if (rc) return rc;
    if (!out) return fail(ctx, RT_ERR_INVALID, "null output");
    rc = ensure_workspace(ctx, r.n_pixels, r.n_pixels, 1);
    if (rc) return rc;
    Workspace& w = ctx->ws;
    const FrameArgs f = frame_args(p, r);
    const uint32_t n = static_cast<uint32_t>(r.n_pixels);
    HIP_TRY(ctx, hipMemsetAsync(w.counters, 0, 8 * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_raygen, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, f, device_scene(ctx), w.q[0], 0u, n, n,
                       static_cast<unsigned long long>(sample_index), jitter ? 1 : 0, w.counters, inline_counters(ctx));
    HIP_TRY(ctx, hipGetLastError());
    rc = launch_trace(ctx, w.q[0], w.counters, w.counters + 2);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = harvest(ctx);
    if (rc) return rc;
    return download_hits(ctx, w.q[0], n, out);
} RT_ABI_END_CTX(ctx)

int rt_trace_rays(RtContext* ctx, const float* origins, const float* directions, uint64_t n_rays, RtHitRecord* out) try {
    int rc = require_scene(ctx);
    if (rc) return rc;
    if (n_rays == 0) return RT_OK;
    if (!origins || !directions || !out) return fail(ctx, RT_ERR_INVALID, "null ray arrays");
    if (n_rays > 0x7FFFFFFFull) return fail(ctx, RT_ERR_INVALID, "too many rays in one call");
    rc = ensure_workspace(ctx, n_rays, 1, 1);
    if (rc) return rc;
    Workspace& w = ctx->ws;
    const uint32_t n = static_cast<uint32_t>(n_rays);
    // the second queue is unused here: stage the packed rays in two of its arrays
    float* d_o = reinterpret_cast<float*>(w.q[1].origin);
    float* d_d = reinterpret_cast<float*>(w.q[1].direction);
    HIP_TRY(ctx, hipMemcpyAsync(d_o, origins, n_rays * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_d, directions, n_rays * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(w.counters, 0, 8 * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_load_rays, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_o, d_d, device_scene(ctx), w.q[0], n,
                       w.counters, inline_counters(ctx));
    HIP_TRY(ctx, hipGetLastError());
    rc = launch_trace(ctx, w.q[0], w.counters, w.counters + 2);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = harvest(ctx);
    if (rc) return rc;
    return download_hits(ctx, w.q[0], n, out);
} RT_ABI_END_CTX(ctx)

// Error paths of the entry point below: a literal copied into the context's message
// buffer (capacity reserved at creation), no temporary strings.
static int fail_literal(RtContext* ctx, int code, const char* msg) {
    ctx->error.assign(msg);
    return code;
}
#define HIP_TRY_LITERAL(ctx, expr)                                                        \
    do {                                                                                  \
        if ((expr) != hipSuccess) {                                                       \
            (void)hipGetLastError();                                                      \
            ctx->launch_spill = nullptr;                                                  \
            ctx->launch_max_threads = 0;                                                  \
            return fail_literal(ctx, RT_ERR_HIP, "HIP call failed: " #expr);              \
        }                                                                                 \
    } while (0)

int rt_trace_rays_noalloc(RtContext* ctx, const float* origins, const float* directions, uint64_t n_rays, RtHitRecord* out,
                          int on_device) try {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->has_scene) return fail_literal(ctx, RT_ERR_STATE, "no scene uploaded");
    if (n_rays == 0) return RT_OK;
    if (!origins || !directions || !out) return fail_literal(ctx, RT_ERR_INVALID, "null ray arrays");
    if (n_rays > kNoAllocMaxRays)
        return fail_literal(ctx, RT_ERR_INVALID, "rt_trace_rays_noalloc takes at most 524288 rays per call (fixed scratch array)");
    static_assert(kNoAllocMaxRays == 524288, "the message above states the limit");
    HIP_TRY_LITERAL(ctx, hipSetDevice(ctx->device));
    const uint32_t n = static_cast<uint32_t>(n_rays);
    // carve this GPU's scratch array (always through the looked-up device pointer)
    unsigned char* const base = ctx->scratch;
    uint32_t* const counters = reinterpret_cast<uint32_t*>(base);
    unsigned char* at = base + kScratchQueue;
    auto take = [&](size_t bytes_per_ray) {
        unsigned char* p = at;
        at += kNoAllocMaxRays * bytes_per_ray;
        return p;
    };
    DQueue q{};
    q.origin = reinterpret_cast<P3*>(take(sizeof(P3)));
    q.direction = reinterpret_cast<P3*>(take(sizeof(P3)));
    q.hit_t = reinterpret_cast<float*>(take(sizeof(float)));
    q.hit_id = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t)));
    HitOut* const records = reinterpret_cast<HitOut*>(take(sizeof(HitOut)));
    float* const stage_o = reinterpret_cast<float*>(take(3 * sizeof(float)));
    float* const stage_d = reinterpret_cast<float*>(take(3 * sizeof(float)));
    // the overflow area serves as many traversal threads as the scene's deepest stack leaves room for
    const uint32_t spill_threads = static_cast<uint32_t>(std::min<size_t>(kScratchSpillEntries / ctx->deepest_stack, ctx->max_grid_threads));
    // counters must start at zero; everything else is written before it is read
    HIP_TRY_LITERAL(ctx, hipMemsetAsync(counters, 0, kScratchQueue, ctx->stream));
    const float* d_o = origins;
    const float* d_d = directions;
    if (!on_device) {
        HIP_TRY_LITERAL(ctx, hipMemcpyAsync(stage_o, origins, n_rays * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY_LITERAL(ctx, hipMemcpyAsync(stage_d, directions, n_rays * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        d_o = stage_o;
        d_d = stage_d;
    }
    hipLaunchKernelGGL(k_load_rays, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_o, d_d, device_scene(ctx), q, n, counters,
                       inline_counters(ctx));
    HIP_TRY_LITERAL(ctx, hipGetLastError());
    ctx->launch_spill = reinterpret_cast<uint2*>(base + kScratchSpill);
    ctx->launch_max_threads = spill_threads;
    const bool opt_time = ctx->opt_time;
    ctx->opt_time = false;  // timing a launch creates events
    const int rc = launch_trace(ctx, q, counters, counters + 2);
    ctx->opt_time = opt_time;
    ctx->launch_spill = nullptr;
    ctx->launch_max_threads = 0;
    if (rc) return rc;
    hipLaunchKernelGGL(k_hit_records, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, q.hit_t, q.hit_id, records, n);
    HIP_TRY_LITERAL(ctx, hipGetLastError());
    HIP_TRY_LITERAL(ctx, hipMemcpyAsync(out, records, n_rays * sizeof(RtHitRecord),
                                        on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY_LITERAL(ctx, hipStreamSynchronize(ctx->stream));
    return harvest(ctx);
} RT_ABI_END_CTX(ctx)

int rt_self_test(RtContext* ctx, uint32_t which, uint64_t n_cases, uint64_t* failures) try {
    if (!ctx || !failures) return RT_ERR_INVALID;
    if (which != RT_SELF_TEST_DIVISION && which != RT_SELF_TEST_RECIPROCAL && which != RT_SELF_TEST_UNIT_BYTE)
        return fail(ctx, RT_ERR_INVALID, "unknown self test");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long), ctx->stream));
    const uint32_t threads = 256u * 1024u;
    if (which == RT_SELF_TEST_DIVISION) {
        const uint32_t per_thread = static_cast<uint32_t>(std::min<uint64_t>((n_cases / 2 + threads - 1) / threads, 1u << 20));
        hipLaunchKernelGGL(k_selftest_division, dim3(threads / 256), dim3(256), 0, ctx->stream, 0x1234567ull, per_thread, ctx->d_stats);
    } else if (which == RT_SELF_TEST_RECIPROCAL) {
        hipLaunchKernelGGL(k_selftest_reciprocal, dim3(threads / 256), dim3(256), 0, ctx->stream, ctx->d_stats);
    } else {
        hipLaunchKernelGGL(k_selftest_unit, dim3(1), dim3(256), 0, ctx->stream, ctx->d_stats);
    }
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long bad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&bad, ctx->d_stats, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long), ctx->stream));
    *failures = bad;
    return RT_OK;
} RT_ABI_END_CTX(ctx)

// Page-locked host memory, usable with every GPU of the process.
void* rt_host_alloc(size_t bytes) try {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
} RT_ABI_END_VALUE(nullptr)

void rt_host_free(void* p) try {
    if (p) (void)hipHostFree(p);
} RT_ABI_END_VOID

int rt_get_stats(RtContext* ctx, RtStats* out) try {
    if (!ctx || !out) return RT_ERR_INVALID;
    *out = ctx->stats;
    return RT_OK;
} RT_ABI_END_CTX(ctx)

int rt_reset_stats(RtContext* ctx) try {
    if (!ctx) return RT_ERR_INVALID;
    if (std::getenv("RT_MI355_DEBUG_TALLIES"))
        std::fprintf(stderr, "[rt_mi355] iters %llu active_lanes %llu branch_runs %llu branch_lanes %llu shape_runs %llu shape_lanes %llu\n",
                     ctx->tallies[0], ctx->tallies[1], ctx->tallies[2], ctx->tallies[3], ctx->tallies[4], ctx->tallies[5]);
    if (std::getenv("RT_MI355_DEBUG_TALLIES")) {
        std::fprintf(stderr, "[rt_mi355] branch visits by depth band:");
        for (int k = 0; k < 32; ++k) std::fprintf(stderr, " %llu", ctx->tallies[6 + k]);
        std::fprintf(stderr, "\n");
    }
    for (unsigned long long& t : ctx->tallies) t = 0;
    ctx->stats = RtStats{};
    return RT_OK;
} RT_ABI_END_CTX(ctx)

}  // extern "C"
