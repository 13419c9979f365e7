// HIP kernels of the render loop for gfx950 (wave64).
//
// One path = one camera sample.  A batch of paths moves through the bounce
// loop of /root/reference/src/lib.rs:304-375 as a wavefront:
//
//   k_raygen                      camera rays + generator state      (once)
//   per bounce:
//     k_trace<SPHERE|PLANE|TRIANGLE>  nearest hit in that family's BVH
//     k_shade                     sky / scatter / texture, compaction of the
//                                 surviving paths into the next queue
//   k_fold                        ordered per-pixel sum of the batch's samples
//   k_quantize                    mean, gamma, running average, u8
//
// All arithmetic repeats the reference's operation order in IEEE binary32:
// this file must be compiled with -ffp-contract=off (no fused multiply-add),
// correctly rounded division and sqrt, and denormals preserved, so that every
// value is bit-identical to the scalar CPU evaluation of the same formulas.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace rtd {

// ------------------------------------------------------------------ vectors
// Operation orders of /root/reference/src/vec3.rs.

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 f3(const float* p) { return F3{p[0], p[1], p[2]}; }
__device__ __forceinline__ F3 f3(float4 v) { return F3{v.x, v.y, v.z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
// vec3.rs:224-233
__device__ __forceinline__ float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vec3.rs:303-321
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vec3.rs:285-291
__device__ __forceinline__ F3 normalize(F3 a) { return a / sqrtf(dot(a, a)); }
// vec3.rs:564-575
__device__ __forceinline__ F3 reflect(F3 d, F3 n) { return d - (n * 2.0f) * dot(d, n); }

// ---------------------------------------------------------------- generator
// wyrand, /root/reference/src/rng.rs:19-41.

constexpr unsigned long long kWy0 = 0x2d358dccaa6c78a5ull;
constexpr unsigned long long kWy1 = 0x8bb84b93962eacc9ull;

__device__ __forceinline__ float rng_f32(unsigned long long& s) {
    s += kWy0;
    const unsigned long long a = s, b = s ^ kWy1;
    const unsigned long long lo = a * b, hi = __umul64hi(a, b);
    const uint32_t u = static_cast<uint32_t>(lo ^ hi);
    return __uint_as_float(0x3F800000u + (u >> 9)) - 1.0f;
}

// vec3.rs:549-558
__device__ __forceinline__ F3 random_unit(unsigned long long& s) {
    const float x = rng_f32(s) - 0.5f;
    const float y = rng_f32(s) - 0.5f;
    const float z = rng_f32(s) - 0.5f;
    return normalize(f3(x, y, z));
}

// ------------------------------------------------------------ intersections

constexpr float kMinDistance = 0.001f;        // shapes.rs:18
constexpr float kEpsilon = 1.1920929e-7f;     // f32::EPSILON
constexpr float kInfinity = __builtin_huge_valf();

// bvh.rs:25-43
__device__ __forceinline__ bool box_hit(const float* mn, const float* mx, F3 o, F3 d, float& tmin_out) {
    const float t1x = (mn[0] - o.x) / d.x, t1y = (mn[1] - o.y) / d.y, t1z = (mn[2] - o.z) / d.z;
    const float t2x = (mx[0] - o.x) / d.x, t2y = (mx[1] - o.y) / d.y, t2z = (mx[2] - o.z) / d.z;
    const float tmin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
    const float tmax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
    tmin_out = tmin;
    return tmax >= tmin && tmax > 0.0f;
}

// shapes.rs:60-87
__device__ __forceinline__ bool sphere_hit(const DSphere& s, F3 o, F3 d, float& t) {
    const F3 delta = o - f3(s.center);
    const float b = dot(delta, d);
    const float disc = b * b - dot(delta, delta) + s.radius * s.radius;
    if (disc < 0.0f) return false;
    const float root = sqrtf(disc);
    const float t1 = -b - root;
    if (t1 > kMinDistance) {
        t = t1;
        return true;
    }
    const float t2 = -b + root;
    t = t2;
    return t2 > kMinDistance;
}

// shapes.rs:141-154
__device__ __forceinline__ bool plane_hit(const DPlane& p, F3 o, F3 d, float& t) {
    const F3 n = f3(p.normal);
    const float den = dot(n, d);
    if (fabsf(den) < kEpsilon) return false;
    const float num = dot(n, o - f3(p.point));
    t = -(num / den);
    return t > kMinDistance;
}

// shapes.rs:232-259.  The reference returns early after each failed test; here
// every quantity is computed and the tests are combined at the end, which gives
// the same answer (a failed earlier test makes the result false whatever the
// later values, NaN or infinite, turn out to be) without branches inside a step
// that other lanes of the wave are executing anyway.
//
// The two cross products h = d x e2 and q = (o - a) x e1 and the four dot
// products run two at a time in packed instructions (v_pk_mul_f32 /
// v_pk_add_f32: two binary32 operations per lane, each rounded exactly like its
// scalar form; nothing is fused): ex/ey/ez hold (e2, e1) of one axis, so the
// first half of every pair below belongs to h and the second to q.
typedef float v2f __attribute__((ext_vector_type(2)));

// 1.0f / b for |b| in [2^-20, 2^60): hipcc expands a correctly rounded division
// into v_div_scale (both operands), a reciprocal refined by one Newton step,
// the quotient refined by two residual steps, v_div_fmas and v_div_fixup.  With
// numerator 1 and b in this range nothing is scaled and no special case
// applies, so the seven instructions below are the same operations on the same
// values, hence the same bits (k_selftest_reciprocal compares all of them).
constexpr float kRcpLo = 0x1p-20f, kRcpHi = 0x1p60f;
__device__ __forceinline__ float rcp_midrange(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r = __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
    const float q = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-b, q, 1.0f), r, q);
}

__device__ __forceinline__ v2f swapped(v2f v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ bool triangle_hit(v2f ex, v2f ey, v2f ez, F3 a, F3 o, F3 d, float& t) {
    const F3 s = o - a;
    const v2f px = {d.x, s.x}, py = {d.y, s.y}, pz = {d.z, s.z};
    const v2f cx = py * ez - pz * ey;  // (h.x, q.x)
    const v2f cy = pz * ex - px * ez;  // (h.y, q.y)
    const v2f cz = px * ey - py * ex;  // (h.z, q.z)
    const v2f qhx = swapped(cx), qhy = swapped(cy), qhz = swapped(cz);
    const v2f t_det = (ex * qhx + ey * qhy) + ez * qhz;  // (dot(e2, q), dot(e1, h))
    v2f v_u = (px * qhx + py * qhy) + pz * qhz;          // (dot(d, q), dot(s, h))
    asm volatile("" : "+v"(v_u));  // keeps the packed sums above the branch below (sunk past it, they lose their operand selects)
    const float det = t_det.y;
    // A determinant below 2e-6 fails the test whatever `inv` is, so only large
    // (or NaN) determinants need the general division; the choice is made for
    // the whole wave, so the common case runs no division at all.
    static_assert(kRcpLo < 2e-6f, "every determinant that can pass must be in the reciprocal's range");
    const float inv = __any(!(fabsf(det) < kRcpHi)) ? 1.0f / det : rcp_midrange(det);
    const v2f vu = v2f{inv, inv} * v_u;
    const float u = vu.y, v = vu.x;
    t = inv * t_det.x;
    const bool ok_det = !(fabsf(det) < 2e-6f);
    const bool ok_u = u >= 0.0f && u <= 1.0f;
    const bool ok_v = !(v < 0.0f || u + v > 1.0f);
    return ok_det & ok_u & ok_v & (t > kMinDistance);
}
// Triangle `i` of a family's three triangle arrays (layout.h).
__device__ __forceinline__ bool triangle_hit(const DFamily& fam, uint32_t i, F3 o, F3 d, float& t) {
    // (32-bit byte offsets from wave-uniform bases — a family holds fewer than 2^27 shapes — so that
    // the loads take their base from scalar registers and no 64-bit address is computed per lane)
    const char* b0 = static_cast<const char*>(fam.shapes);
    const char* b1 = b0 + static_cast<size_t>(fam.n_shapes) * sizeof(DTriHalf);
    const char* bz = b1 + static_cast<size_t>(fam.n_shapes) * sizeof(DTriHalf);
    const uint32_t o16 = i << 4, o4 = i << 2;
    const float4 v0 = *reinterpret_cast<const float4*>(b0 + o16), v1 = *reinterpret_cast<const float4*>(b1 + o16);
    const float az = *reinterpret_cast<const float*>(bz + o4);
    return triangle_hit(v2f{v0.x, v0.y}, v2f{v0.z, v0.w}, v2f{v1.x, v1.y}, f3(v1.z, v1.w, az), o, d, t);
}

// ------------------------------------------------------------------- trace

// Exact division for the slab test at about a third of the cost of `/`.
//
// The slab test divides twelve differences by the same three direction
// components per node, and the reference's results depend on every quotient
// being the correctly rounded one.  hipcc expands `a / b` into eleven
// instructions; here the divisor's part is done once per ray and a quotient
// costs a multiply and three FMAs:
//   per ray   zh = RN(1/b) (one IEEE division), zl = RN((1 - b*zh) * zh), so that
//             zh + zl is 1/b to about 46 bits;
//   quotient  q0 = RN(a*zh + RN(a*zl))     within 1/2 ulp + 2^-22 ulp of a/b, i.e.
//                                          one of the two neighbours of a/b,
//             r  = a - b*q0                exact in one FMA for such a q0,
//             q  = RN(q0 + r*zh)           = RN(a/b): Markstein's theorem (a
//                                          correctly rounded reciprocal corrects a
//                                          faithful quotient to the correctly
//                                          rounded one).
// This holds while nothing under- or overflows on the way; the caller guarantees
// |b| in [2^-30, 2], a == 0 or |a| in [2^-84, 2^60) (see `fast` in k_trace) and
// takes the plain `/` otherwise.  A zero quotient may come out with either sign;
// the slab test only compares.  k_selftest_division compares with `/`.
struct Reciprocal {
    float hi, lo;
};
__device__ __forceinline__ Reciprocal reciprocal_of(float b) {
    const float zh = 1.0f / b;
    return Reciprocal{zh, __builtin_fmaf(-b, zh, 1.0f) * zh};
}
// Two quotients with a common divisor in one packed instruction stream
// (v_pk_mul_f32 / v_pk_fma_f32 carry two binary32 operations per lane, each
// rounded exactly like its scalar form).
__device__ __forceinline__ v2f div_by2(v2f a, float b, Reciprocal z) {
    const v2f nb = {-b, -b}, hh = {z.hi, z.hi}, ll = {z.lo, z.lo};
    const v2f q = __builtin_elementwise_fma(a, hh, a * ll);
    const v2f r = __builtin_elementwise_fma(nb, q, a);
    return __builtin_elementwise_fma(r, hh, q);
}

// True when |v| is in [lo, hi) — false for NaN.
__device__ __forceinline__ bool magnitude_in(float v, float lo, float hi) {
    const float m = fabsf(v);
    return m >= lo && m < hi;
}

// Persistent traversal kernel: workgroups of WAVES waves (1, 8 or 16; the waves
// of a workgroup share only the LDS copy of the tree's top, see below k_trace);
// every lane owns one ray and performs that ray's ordered traversal exactly as
// bvh.rs:227-298 does.
// Lanes that finish take new rays from a block of the queue the wave has
// reserved (one atomic on the queue cursor per block of rays: a single shared
// counter cannot serve one atomic per ray), so a wave stays busy until the
// queue is drained.
//
// What a lane holds between iterations is either a branch to expand, a leaf
// with shapes left to test, or nothing (then it pops its stack).  Expanding a
// branch pushes only the farther child: the nearer one would be pushed last and
// popped first with the cull test of bvh.rs:252 already known to pass (it was
// pushed because its distance is below the closest hit, which has not changed),
// so it is kept in hand instead of making the round trip through the stack.
// Every iteration the wave votes: the branch step (one 64-byte record, two slab
// tests) and the shape step (one shape per lane, or — triangles — all pending
// (ray, triangle) pairs of the wave dealt out to its 64 lanes) each run only when
// enough lanes wait for them.  A lane's own sequence of operations is never reordered,
// only delayed, so results are those of the sequential algorithm.
//
// The stack's first `lds_slots` entries per lane live in LDS, laid out
// [slot][lane] as 8-byte entries: lane l always touches banks 2l and 2l+1
// (mod 64), whatever slot it is at, so divergent stack depths never conflict.
// Deeper entries overflow to a private strip of global memory.

// Streaming accesses to the path queues: every element is touched once per
// kernel and the queues are far larger than the caches, so they are marked
// non-temporal to leave cache space to the scene data (nodes, surface records,
// texels), which is re-read.
//
// Every pointer these helpers and in_global() are given points into device memory.  Saying so
// (address space 1) matters where the pointer itself was read from memory rather than passed as
// a kernel argument (k_path): the compiler then cannot tell, and would emit flat loads, which
// reach memory the slow way round and count against the LDS wait counter as well.
#define RT_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T* in_global(T* p) {
    __builtin_assume(!__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void*)p));
    __builtin_assume(!__builtin_amdgcn_is_private((const __attribute__((address_space(0))) void*)p));
    return p;
}
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
__device__ inline float4 stream_load(const float4* p) {
    const nt_f4 v = __builtin_nontemporal_load((const RT_GLOBAL nt_f4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef float nt_f3 __attribute__((ext_vector_type(3)));
typedef nt_f3 nt_f3_packed __attribute__((aligned(4)));  // 12 bytes in memory, dword-aligned: one global_load_dwordx3
__device__ inline F3 stream_load(const P3* p) {
    const nt_f3 v = __builtin_nontemporal_load((const RT_GLOBAL nt_f3_packed*)p);
    return F3{v.x, v.y, v.z};
}
__device__ inline void stream_store(P3* p, F3 v) {
    __builtin_nontemporal_store(nt_f3{v.x, v.y, v.z}, (RT_GLOBAL nt_f3_packed*)p);
}
__device__ inline uint2 stream_load(const uint2* p) {
    const nt_u2 v = __builtin_nontemporal_load((const RT_GLOBAL nt_u2*)p);
    return make_uint2(v.x, v.y);
}
__device__ inline uint32_t stream_load(const uint32_t* p) { return __builtin_nontemporal_load((const RT_GLOBAL uint32_t*)p); }
__device__ inline float stream_load(const float* p) { return __builtin_nontemporal_load((const RT_GLOBAL float*)p); }
__device__ inline void stream_store(float4* p, float4 v) {
    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, (RT_GLOBAL nt_f4*)p);
}
__device__ inline void stream_store(uint2* p, uint2 v) {
    __builtin_nontemporal_store(nt_u2{v.x, v.y}, (RT_GLOBAL nt_u2*)p);
}
__device__ inline void stream_store(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, (RT_GLOBAL uint32_t*)p); }
__device__ inline void stream_store(float* p, float v) { __builtin_nontemporal_store(v, (RT_GLOBAL float*)p); }

// Inclusive scans over the 64 lanes of a wave with data-parallel-primitive moves
// (row shifts inside each row of 16 lanes, then the last lane of row 0/2 and of
// lane 31 broadcast forward), which cost a few VALU cycles each instead of a
// trip through the LDS crossbar.  Every lane of the wave must be executing.
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp_or_zero(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
// Number of set bits of `mask` below this lane (one v_mbcnt pair, no per-lane mask to keep).
__device__ __forceinline__ uint32_t bits_below_lane(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
// Value of `v` in lane `src` (0..63) of this wave.
__device__ __forceinline__ uint32_t from_lane(uint32_t v, uint32_t src) {
    return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src << 2), static_cast<int>(v)));
}
__device__ __forceinline__ float from_lane(float v, uint32_t src) { return __uint_as_float(from_lane(__float_as_uint(v), src)); }

__device__ inline uint32_t wave_scan_add(uint32_t v) {
    v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8
    v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ inline uint32_t wave_scan_max(uint32_t v) {
    v = max(v, dpp_or_zero<0x111, 0xf>(v));
    v = max(v, dpp_or_zero<0x112, 0xf>(v));
    v = max(v, dpp_or_zero<0x114, 0xf>(v));
    v = max(v, dpp_or_zero<0x118, 0xf>(v));
    v = max(v, dpp_or_zero<0x142, 0xa>(v));
    v = max(v, dpp_or_zero<0x143, 0xc>(v));
    return v;
}

constexpr int kTraceBlock = 64;
constexpr uint32_t kItemNone = 0xFFFFFFFFu;

struct TraceArgs {
    DFamily fam;
    DQueue q;
    const uint32_t* n_rays;  // device: rays in the queue
    uint32_t* cursor;        // device: next unclaimed ray (zero before launch)
    uint2* spill;            // [(slot - lds_slots)][grid threads]
    uint32_t spill_stride;   // grid threads
    uint32_t lds_slots;      // stack entries per lane kept in LDS (dynamic LDS = lds_slots*64*8 bytes)
    uint32_t phase_min;      // a step runs for this many waiting lanes, or for the majority
    uint32_t chunk_max;      // most rays a wave reserves from the queue per atomic
    uint32_t pair_min;       // SHARE: pending (ray, triangle) pairs that trigger a shared shape step
    uint32_t cache_nodes;    // branch records [0, cache_nodes) are copied to LDS by each workgroup
    uint32_t refill_min;     // idle lanes that trigger a refill from the queue (1..64)
    uint32_t guided;         // 1: reservations shrink towards the end of the queue (see trace_loop)
    int bounds_fast;         // every box coordinate is 0 or has magnitude in [2^-60, 2^59)
    unsigned long long* counters;  // [0] branch visits, [1] shape tests (COUNT only)
    unsigned long long* wave_tallies;  // six diagnostic words, triangle family only (COUNT only)
    unsigned long long* visit_bands;   // 32 words: branch visits by depth band (COUNT only; null unless asked for: one atomic per visit)
};

// A workgroup is WAVES independent waves that share one thing: a copy in LDS of
// the first `cache_nodes` branch records (the array is numbered breadth-first,
// so these are the top levels of the tree, where most branch visits land).
// The copy is laid out [16-byte quarter][record], so that lanes reading the
// same quarter of different records spread over the LDS banks.  Dynamic LDS:
// cache_nodes*64 bytes, then per wave lds_slots*512 (stack) + 512 (result keys)
// + 256 (marks).
__host__ __device__ inline size_t trace_lds_per_wave(uint32_t lds_slots) {
    return static_cast<size_t>(lds_slots) * kTraceBlock * sizeof(uint2) + kTraceBlock * (sizeof(unsigned long long) + sizeof(uint32_t));
}

// Eight waves per SIMD leave each wave 96 scalar registers, of which the
// system's trap handler keeps 16: the kernel is held to the 80 that remain.
//
// FASTSLAB (an option, off by default, not bit-exact): the slab test of an ordinary
// ray multiplies by the reciprocal of the direction instead of dividing, fused:
// t = fma(bound, 1/d, -origin/d), six packed instructions for the twelve values in
// place of thirty.  Distances of boxes then differ from the reference's in the last
// places, which can change the order of visits and the outcome of a cull where a
// box's distance and the closest hit's agree to rounding; triangle tests and
// everything else stay exact.
// The traversal loop of one wave (see the comment above): rays are entries of `q`,
// taken from the block [pool_first, pool_last) the wave holds and then, unless
// `no_more`, from further blocks of `chunk` entries reserved on a.cursor up to
// entry `n`.  The wave's LDS areas and the workgroup's copy of the tree's top are
// set up by the calling kernel.
template <int KIND, bool COUNT, bool SHARE, int WAVES, bool FASTSLAB>
__device__ __forceinline__ void trace_loop(const TraceArgs& a, const DQueue& q, const float4* s_nodes, uint2* s_stack,
                                           unsigned long long* s_res, uint32_t* s_mark, uint32_t lane, uint32_t gtid,
                                           uint32_t cache_nodes, uint32_t n, uint32_t chunk, uint32_t pool_first,
                                           uint32_t pool_last, bool no_more, uint32_t tail_shift = 31u) {
    const int lds_slots = static_cast<int>(a.lds_slots);
    const DNode* __restrict__ nodes = a.fam.nodes;

    auto stack_store = [&](int slot, float dist, uint32_t ref) {
        // One 64-bit value, so that neither store can be split in two and the halves
        // merged across the branches into a flat store through a selected address.
        const unsigned long long e = (static_cast<unsigned long long>(ref) << 32) | __float_as_uint(dist);
        if (slot < lds_slots) reinterpret_cast<unsigned long long*>(s_stack)[slot * kTraceBlock + lane] = e;
        else reinterpret_cast<unsigned long long*>(a.spill)[static_cast<size_t>(slot - lds_slots) * a.spill_stride + gtid] = e;
    };
    auto stack_load = [&](int slot) -> uint2 {
        uint2 e;
        if (slot < lds_slots) {
            e = s_stack[slot * kTraceBlock + lane];
            // keeps the two loads apart: merged into one load through a selected address
            // they become a flat load, which reaches LDS the slow way round
            asm volatile("" : "+v"(e.x), "+v"(e.y));
        } else {
            e = a.spill[static_cast<size_t>(slot - lds_slots) * a.spill_stride + gtid];
        }
        return e;
    };

    bool active = false, exhausted = no_more, fast = false;
    uint32_t pool_next = pool_first, pool_end = pool_last;  // wave-uniform: rays reserved for this wave
    uint32_t reserve = chunk;  // size of this wave's next reservation
    uint32_t ray = 0, best_i = kHitNone;
    int sp = 0;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0);
    Reciprocal zx = {0, 0}, zy = {0, 0}, zz = {0, 0};  // of d.x, d.y, d.z
    float best_t = kInfinity;
    float prior_t = kInfinity;              // distance already recorded for this ray by an earlier family
    uint32_t item = kItemNone;              // node reference in hand (branch or leaf)
    uint32_t shape_cur = 0, shape_end = 0;  // shapes of the current leaf still to test
    unsigned long long n_branch = 0, n_tests = 0;
    // wave-level tallies of how full the steps run (COUNT only)
    uint32_t w_iters = 0, w_branch_runs = 0, w_branch_lanes = 0, w_shape_runs = 0, w_shape_lanes = 0, w_active_lanes = 0;

    for (;;) {
        // ---- pop: a lane with nothing in hand takes the first stack entry that
        //      survives the cull test of bvh.rs:252, or finishes its ray
        if (active && item == kItemNone && shape_cur == shape_end) {
            while (sp > 0) {
                const uint2 e = stack_load(--sp);
                if (!(best_t <= __uint_as_float(e.x))) {
                    item = e.y;
                    break;
                }
            }
            if (item == kItemNone) {
                // bvh.rs:300 + lib.rs:333: keep the earlier family on equal distance.
                if (best_t < prior_t) {
                    stream_store(&q.hit_t[ray], best_t);
                    stream_store(&q.hit_id[ray], (static_cast<uint32_t>(KIND) << kHitKindShift) | best_i);
                }
                active = false;
            }
        }

        // ---- refill: idle lanes (including those that have just finished) take new rays
        if (!exhausted || pool_next != pool_end) {
            const unsigned long long idle = __ballot(!active);
            const uint32_t n_idle = __popcll(idle);
            // New rays come from HBM and the wave waits for them with its next loads, so
            // idle lanes are refilled in batches rather than one by one.
            if (n_idle >= a.refill_min || n_idle == 64u) {
                if (pool_next == pool_end) {
                    // reserve the next block of the queue for this wave
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(a.cursor, reserve);
                    base = __builtin_amdgcn_readfirstlane(base);
                    pool_next = min(base, n);
                    pool_end = min(base + reserve, n);
                    if (base + reserve >= n) exhausted = true;
                    // Reservations shrink as the queue runs out (down to one wave's worth), so that
                    // the launch's waves finish at about the same time instead of each working off a
                    // full last block alone: what is left, shared among all waves, halved.
                    if (a.guided) reserve = min(chunk, max(64u, ((n - pool_end) >> tail_shift) & ~63u));
                }
                const uint32_t take = min(n_idle, pool_end - pool_next);
                const uint32_t rank = bits_below_lane(idle);
                if (!active && rank < take) {
                    const uint32_t idx = pool_next + rank;
                    ray = idx;
                    o = stream_load(&q.origin[idx]);  // read once: kept out of the way of the node data in the caches
                    d = stream_load(&q.direction[idx]);
                    prior_t = stream_load(&q.hit_t[idx]);
                    best_t = kInfinity;
                    best_i = kHitNone;
                    sp = 0;
                    item = a.fam.root_ref;  // bvh.rs:246: the root is entered with its box untested
                    active = true;
                    // The per-ray part of the slab test is prepared here, at the price of waiting
                    // for the ray's loads (other waves of the SIMD cover that): a block deferred
                    // to the ray's first branch step measured 1.3 % slower, as every extra
                    // conditional region in this loop costs register moves where its values merge;
                    // reciprocals prepared by the kernel that creates the ray and read from the queue
                    // here (12 more bytes per ray, 90 fewer instructions per refill) measured 6 % slower
                    // over the whole render.
                    // regime in which div_by2 equals `/` (see reciprocal_of)
                    const int d_ok = int(magnitude_in(d.x, 0x1p-30f, 2.0f)) & int(magnitude_in(d.y, 0x1p-30f, 2.0f)) &
                                     int(magnitude_in(d.z, 0x1p-30f, 2.0f));
                    const int o_ok = (int(o.x == 0.0f) | int(magnitude_in(o.x, 0x1p-60f, 0x1p59f))) &
                                     (int(o.y == 0.0f) | int(magnitude_in(o.y, 0x1p-60f, 0x1p59f))) &
                                     (int(o.z == 0.0f) | int(magnitude_in(o.z, 0x1p-60f, 0x1p59f)));
                    fast = (int(a.bounds_fast != 0) & d_ok & o_ok) != 0;
                    if (FASTSLAB) {
                        // hi = 1/d, lo = -o/d (as -(o * (1/d)))
                        zx.hi = 1.0f / d.x; zx.lo = -(o.x * zx.hi);
                        zy.hi = 1.0f / d.y; zy.lo = -(o.y * zy.hi);
                        zz.hi = 1.0f / d.z; zz.lo = -(o.z * zz.hi);
                    } else {
                        zx = reciprocal_of(d.x);
                        zy = reciprocal_of(d.y);
                        zz = reciprocal_of(d.z);
                    }
                }
                pool_next += take;
            }
        }
        // (no `continue` on an idle wave: with nothing in hand both votes below fail by themselves,
        // and one way round the loop keeps the loop-carried state in one set of registers)
        if (!__any(active) && exhausted && pool_next == pool_end) break;

        if (COUNT && __any(active)) {
            ++w_iters;
            w_active_lanes += __popcll(__ballot(active));
        }
        // ---- branch step, bvh.rs:259-286
        const bool want_branch = item < kLeafBit;  // kItemNone and leaves have the top bit set
        {
            const uint32_t n_b = __popcll(__ballot(want_branch));
            const uint32_t n_s = __popcll(__ballot(shape_cur < shape_end));
            if (COUNT && n_b > 0 && (n_b >= a.phase_min || n_b >= n_s)) {
                ++w_branch_runs;
                w_branch_lanes += n_b;
            }
            if (n_b > 0 && (n_b >= a.phase_min || n_b >= n_s) && want_branch) {
                float4 qx, qy, qz;  // (c0min, c1min, c0max, c1max) per axis
                uint32_t ref0, ref1;
                if (WAVES > 1 && item < cache_nodes) {
                    qx = s_nodes[item];
                    qy = s_nodes[cache_nodes + item];
                    qz = s_nodes[2u * cache_nodes + item];
                    const float4 r = s_nodes[3u * cache_nodes + item];
                    ref0 = __float_as_uint(r.x);
                    ref1 = __float_as_uint(r.y);
                    // (keeps these LDS loads from being merged with the global ones below
                    // into flat loads through a selected address)
                    asm volatile("" : "+v"(qx.x), "+v"(qx.y), "+v"(qx.z), "+v"(qx.w), "+v"(qy.x), "+v"(qy.y), "+v"(qy.z), "+v"(qy.w));
                    asm volatile("" : "+v"(qz.x), "+v"(qz.y), "+v"(qz.z), "+v"(qz.w), "+v"(ref0), "+v"(ref1));
                } else {
                    const float4* np = reinterpret_cast<const float4*>(nodes + item);
                    qx = np[0];
                    qy = np[1];
                    qz = np[2];
                    uint4 q3 = reinterpret_cast<const uint4*>(np)[3];
                    // All sixteen loaded words are kept alive up to here, the two padding words
                    // included: otherwise the register allocator may lay the last load over the
                    // dead half of its neighbour, and the loads then run one after the other.
                    asm volatile("" : "+v"(qx.x), "+v"(qx.y), "+v"(qx.z), "+v"(qx.w), "+v"(qy.x), "+v"(qy.y), "+v"(qy.z), "+v"(qy.w));
                    asm volatile("" : "+v"(qz.x), "+v"(qz.y), "+v"(qz.z), "+v"(qz.w), "+v"(q3.x), "+v"(q3.y), "+v"(q3.z), "+v"(q3.w));
                    ref0 = q3.x;
                    ref1 = q3.y;
                }
                float d0, d1;
                bool h0, h1;
                if (fast) {
                    v2f t1x, t2x, t1y, t2y, t1z, t2z;
                    if (FASTSLAB) {
                        const v2f ix = {zx.hi, zx.hi}, iy = {zy.hi, zy.hi}, iz = {zz.hi, zz.hi};
                        const v2f nx = {zx.lo, zx.lo}, ny = {zy.lo, zy.lo}, nz = {zz.lo, zz.lo};
                        t1x = __builtin_elementwise_fma(v2f{qx.x, qx.y}, ix, nx);
                        t2x = __builtin_elementwise_fma(v2f{qx.z, qx.w}, ix, nx);
                        t1y = __builtin_elementwise_fma(v2f{qy.x, qy.y}, iy, ny);
                        t2y = __builtin_elementwise_fma(v2f{qy.z, qy.w}, iy, ny);
                        t1z = __builtin_elementwise_fma(v2f{qz.x, qz.y}, iz, nz);
                        t2z = __builtin_elementwise_fma(v2f{qz.z, qz.w}, iz, nz);
                    } else {
                        const v2f ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
                        t1x = div_by2(v2f{qx.x, qx.y} - ox, d.x, zx), t2x = div_by2(v2f{qx.z, qx.w} - ox, d.x, zx);
                        t1y = div_by2(v2f{qy.x, qy.y} - oy, d.y, zy), t2y = div_by2(v2f{qy.z, qy.w} - oy, d.y, zy);
                        t1z = div_by2(v2f{qz.x, qz.y} - oz, d.z, zz), t2z = div_by2(v2f{qz.z, qz.w} - oz, d.z, zz);
                    }
                    d0 = fmaxf(fmaxf(fminf(t1x.x, t2x.x), fminf(t1y.x, t2y.x)), fminf(t1z.x, t2z.x));
                    d1 = fmaxf(fmaxf(fminf(t1x.y, t2x.y), fminf(t1y.y, t2y.y)), fminf(t1z.y, t2z.y));
                    const float f0 = fminf(fminf(fmaxf(t1x.x, t2x.x), fmaxf(t1y.x, t2y.x)), fmaxf(t1z.x, t2z.x));
                    const float f1 = fminf(fminf(fmaxf(t1x.y, t2x.y), fmaxf(t1y.y, t2y.y)), fmaxf(t1z.y, t2z.y));
                    h0 = f0 >= d0 && f0 > 0.0f;
                    h1 = f1 >= d1 && f1 > 0.0f;
                } else {
                    const float c0min[3] = {qx.x, qy.x, qz.x}, c0max[3] = {qx.z, qy.z, qz.z};
                    const float c1min[3] = {qx.y, qy.y, qz.y}, c1max[3] = {qx.w, qy.w, qz.w};
                    h0 = box_hit(c0min, c0max, o, d, d0);
                    h1 = box_hit(c1min, c1max, o, d, d1);
                }
                h0 = h0 && d0 < best_t;  // bvh.rs:263-266
                h1 = h1 && d1 < best_t;
                if (COUNT) {
                    ++n_branch;
                    // diagnostic: visits by depth band of the breadth-first numbering (records 2^k - 1 .. 2^(k+1) - 2)
                    if (a.visit_bands) atomicAdd(&a.visit_bands[31 - __clz(static_cast<int>(item + 1u))], 1ull);
                }
                // bvh.rs:268-285.  With both children hit the farther one is pushed and
                // the nearer kept in hand; on equal distances child 1 is the nearer.
                const bool zero_near = d0 < d1;
                if (h0 && h1) {
                    stack_store(sp++, zero_near ? d1 : d0, zero_near ? ref1 : ref0);
                    item = zero_near ? ref0 : ref1;
                } else {
                    item = h0 ? ref0 : (h1 ? ref1 : kItemNone);
                }
            }
        }

        // ---- a leaf in hand becomes a range of shapes to test
        if (item != kItemNone && item >= kLeafBit) {
            uint32_t first = item & kLeafFirstMask;
            uint32_t count = (item >> kLeafCountShift) & 15u;
            if (count == kLeafLong) {
                const DLongLeaf ll = a.fam.long_leaves[first];
                first = ll.first;
                count = ll.count;
            }
            if (COUNT) n_tests += count;
            shape_cur = first;
            shape_end = first + count;
            item = kItemNone;
        }

        // ---- shape step, bvh.rs:288-295
        if (SHARE && KIND == 2) {
            // All pending (ray, triangle) pairs of the wave are dealt out to its 64 lanes,
            // so a leaf of k triangles costs its ray one step instead of k, and the step
            // runs with full lanes.  The scan of a leaf keeps the smallest distance and, among
            // equal distances, the lowest index, then replaces the closest hit on strict `<`:
            // exactly what min over the key (distance bits, index) followed by one strict
            // comparison gives (distances are positive, so their bit patterns order like
            // the values).
            const uint32_t c = shape_end - shape_cur;
            const uint32_t P = wave_scan_add(c);  // inclusive prefix sum of the pending counts
            const uint32_t N = __builtin_amdgcn_readlane(P, 63);
            const uint32_t n_s = __popcll(__ballot(c != 0));
            const uint32_t n_b = __popcll(__ballot(item < kLeafBit));
            if (N > 0 && (N >= a.pair_min || n_s >= n_b)) {  // wave-uniform
                if (c) s_res[lane] = (static_cast<unsigned long long>(__float_as_uint(kInfinity)) << 32) | 0xFFFFFFFFull;
                const uint32_t base = shape_cur - (P - c);  // pair j of this lane is triangle base + j
                for (uint32_t cb = 0; cb < N; cb += 64) {
                    const uint32_t j = cb + lane;
                    // The lane that owns pair j: every lane whose run of pairs begins inside
                    // this group of 64 (or covers its first pair) marks that position with its
                    // number + 1; a running maximum then carries each mark forward over the run
                    // (runs are laid out in lane order, so later marks are larger).
                    // (atomic accesses: the word a lane reads back was written by another lane,
                    // which plain loads and stores would let the compiler assume away)
                    __hip_atomic_store(&s_mark[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (c && P > cb && P - c < cb + 64)
                        __hip_atomic_store(&s_mark[P - c > cb ? P - c - cb : 0], lane + 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t mark = __hip_atomic_load(&s_mark[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const uint32_t owner = min(wave_scan_max(mark) - 1u, 63u);
                    const uint32_t tri = from_lane(base, owner) + j;
                    const F3 oo = f3(from_lane(o.x, owner), from_lane(o.y, owner), from_lane(o.z, owner));
                    const F3 dd = f3(from_lane(d.x, owner), from_lane(d.y, owner), from_lane(d.z, owner));
                    if (COUNT) {
                        ++w_shape_runs;
                        w_shape_lanes += min(64u, N - cb);
                    }
                    if (j < N && tri < a.fam.n_shapes) {  // the second test only keeps a logic error from reading outside the table
                        float t;
                        if (triangle_hit(a.fam, tri, oo, dd, t))
                            atomicMin(&s_res[owner], (static_cast<unsigned long long>(__float_as_uint(t)) << 32) | tri);
                    }
                }
                if (c) {
                    const unsigned long long key = s_res[lane];
                    const float t = __uint_as_float(static_cast<uint32_t>(key >> 32));
                    if (t < best_t) {
                        best_t = t;
                        best_i = static_cast<uint32_t>(key);
                    }
                    shape_cur = shape_end;
                }
            }
        } else {
            // one shape per lane and step: ascending index, replace on strict `<`
            const bool want_shape = shape_cur < shape_end;
            const uint32_t n_s = __popcll(__ballot(want_shape));
            const uint32_t n_b = __popcll(__ballot(item < kLeafBit));
            if (COUNT && n_s > 0 && (n_s >= a.phase_min || n_s >= n_b)) {
                ++w_shape_runs;
                w_shape_lanes += n_s;
            }
            if (n_s > 0 && (n_s >= a.phase_min || n_s >= n_b) && want_shape) {
                const uint32_t i = shape_cur++;
                float t;
                bool hit;
                if (KIND == 0) {
                    hit = sphere_hit(static_cast<const DSphere*>(a.fam.shapes)[i], o, d, t);
                } else if (KIND == 1) {
                    hit = plane_hit(static_cast<const DPlane*>(a.fam.shapes)[i], o, d, t);
                } else {
                    hit = triangle_hit(a.fam, i, o, d, t);
                }
                if (hit && t < best_t) {
                    best_t = t;
                    best_i = i;
                }
            }
        }
    }
    if (COUNT) {
        atomicAdd(&a.counters[0], n_branch);
        atomicAdd(&a.counters[1], n_tests);
        if (lane == 0 && a.wave_tallies) {
            atomicAdd(&a.wave_tallies[0], static_cast<unsigned long long>(w_iters));
            atomicAdd(&a.wave_tallies[1], static_cast<unsigned long long>(w_active_lanes));
            atomicAdd(&a.wave_tallies[2], static_cast<unsigned long long>(w_branch_runs));
            atomicAdd(&a.wave_tallies[3], static_cast<unsigned long long>(w_branch_lanes));
            atomicAdd(&a.wave_tallies[4], static_cast<unsigned long long>(w_shape_runs));
            atomicAdd(&a.wave_tallies[5], static_cast<unsigned long long>(w_shape_lanes));
        }
    }
}

template <int KIND, bool COUNT, bool SHARE, int WAVES, bool FASTSLAB = false>
__global__ void __launch_bounds__(kTraceBlock * WAVES) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(COUNT ? 4 : 8))) k_trace(TraceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float4* const s_nodes = reinterpret_cast<float4*>(s_raw);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const s_wave = s_raw + static_cast<size_t>(a.cache_nodes) * sizeof(DNode) + wave * trace_lds_per_wave(a.lds_slots);
    uint2* const s_stack = reinterpret_cast<uint2*>(s_wave);  // [lds_slots][64]; then 64 result keys, 64 marks (SHARE)
    unsigned long long* const s_res = reinterpret_cast<unsigned long long*>(s_stack + a.lds_slots * kTraceBlock);
    uint32_t* const s_mark = reinterpret_cast<uint32_t*>(s_res + kTraceBlock);
    const uint32_t gtid = blockIdx.x * (kTraceBlock * WAVES) + threadIdx.x;
    const uint32_t n = *a.n_rays;
    const uint32_t cache_nodes = WAVES > 1 ? a.cache_nodes : 0u;
    if (WAVES > 1) {
        const float4* src = reinterpret_cast<const float4*>(a.fam.nodes);
        for (uint32_t k = threadIdx.x; k < cache_nodes * 4u; k += kTraceBlock * WAVES)
            s_nodes[(k & 3u) * cache_nodes + (k >> 2)] = src[k];
        __syncthreads();
    }
    // Reservation size: large for long queues (few atomics), one wave's worth for
    // short ones so that the rays still spread over all waves of the grid.
    const uint32_t chunk = min(a.chunk_max, max(64u, (n / (2u * gridDim.x * WAVES)) & ~63u));
    // log2 of twice the number of waves, rounded up: "what is left / (2 x waves)" as a shift
    const uint32_t tail_shift = 32u - static_cast<uint32_t>(__builtin_clz(2u * gridDim.x * WAVES - 1u));
    trace_loop<KIND, COUNT, SHARE, WAVES, FASTSLAB>(a, a.q, s_nodes, s_stack, s_res, s_mark, lane, gtid, cache_nodes, n, chunk, 0u, 0u, false,
                                                    tail_shift);
}

// --------------------------------------------------- inlined small families

// Nearest hit of a new ray among the sphere and plane families that are small
// enough to be scanned on the spot (DScene::inline_count).  A family whose BVH
// is a single leaf is traversed by bvh.rs:288-295 as exactly this loop: shapes
// in ascending index, replace on strict `<`; the later family replaces the
// earlier only on strict `<` as well (lib.rs:333).  The result seeds the ray's
// hit record; the traversal kernels of the remaining families merge into it.
__device__ __forceinline__ void seed_hit(const DScene& sc, F3 o, F3 d, float& t_out, uint32_t& id_out,
                                         unsigned long long* test_counters) {
    float best_t = kInfinity;
    uint32_t best_id = kHitNone;
    const uint32_t ns = sc.inline_count[0], np = sc.inline_count[1];
    for (uint32_t i = 0; i < ns; ++i) {
        float t;
        if (sphere_hit(in_global(static_cast<const DSphere*>(sc.family[0].shapes))[i], o, d, t) && t < best_t) {
            best_t = t;
            best_id = i;  // kind 0
        }
    }
    float plane_t = kInfinity;
    uint32_t plane_i = kHitNone;
    for (uint32_t i = 0; i < np; ++i) {
        float t;
        if (plane_hit(in_global(static_cast<const DPlane*>(sc.family[1].shapes))[i], o, d, t) && t < plane_t) {
            plane_t = t;
            plane_i = i;
        }
    }
    if (plane_t < best_t) {
        best_t = plane_t;
        best_id = (1u << kHitKindShift) | plane_i;
    }
    t_out = best_t;
    id_out = best_id;
    if (test_counters) {  // counting mode only
        if (ns) atomicAdd(&test_counters[0], static_cast<unsigned long long>(ns));
        if (np) atomicAdd(&test_counters[1], static_cast<unsigned long long>(np));
    }
}

// ------------------------------------------------------------------ raygen

struct FrameArgs {
    float top_left[3], row_step[3], column_step[3], camera[3];
    uint32_t width, height;          // full image
    uint32_t x0, y0, tile_w, tile_h; // region being rendered
    unsigned long long seed;
    unsigned long long stride;       // draws reserved per sample: 2 + 3*max_bounces
};

// lib.rs:261-268.  Path i of the batch is sample (first_sample + i / n_pixels)
// of tile pixel (i % n_pixels); its generator starts at the sub-stream of
// (sample, image pixel) and the two jitter draws come first, x then y.
//
// The launch covers paths [first_path, first_path + n_paths) of the batch and
// writes them to queue entries [0, n_paths): a batch may be split over several
// queues; a path's result slot is its number in the whole batch.
//
// `deal_waves` (k_path's launches; 0 otherwise): the paths are dealt out in blocks of 64 to that
// many waves in turn, wave w's blocks stored one after the other from entry w * per_wave on, so
// that every wave's share is spread over the whole frame (neighbouring pixels have paths of
// similar length; contiguous shares would leave some waves with all the long ones).
__global__ void k_raygen(FrameArgs f, DScene sc, DQueue q, uint32_t first_path, uint32_t n_paths, uint32_t n_pixels,
                         unsigned long long first_sample, int jitter, uint32_t* n_rays_out,
                         unsigned long long* test_counters, uint32_t deal_waves = 0, uint32_t per_wave = 0) {
    uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local == 0) *n_rays_out = n_paths;
    if (local >= n_paths) return;
    const uint32_t path = first_path + local;
    if (deal_waves) {
        const uint32_t block = local >> 6;
        local = (block % deal_waves) * per_wave + (block / deal_waves) * 64u + (local & 63u);
    }
    const uint32_t j = path / n_pixels, pix = path - j * n_pixels;
    const uint32_t x = f.x0 + pix % f.tile_w, y = f.y0 + pix / f.tile_w;
    const unsigned long long sample = first_sample + j;
    const unsigned long long k =
        sample * static_cast<unsigned long long>(f.width) * f.height +
        (static_cast<unsigned long long>(y) * f.width + x);
    unsigned long long s = f.seed + k * f.stride * kWy0;
    float jx = 0.0f, jy = 0.0f;
    if (jitter) {
        jx = rng_f32(s);
        jy = rng_f32(s);
    }
    const F3 a = f3(f.row_step) * (static_cast<float>(x) + jx / 2.0f);
    const F3 b = f3(f.column_step) * (static_cast<float>(y) + jy / 2.0f);
    const F3 pixel = (f3(f.top_left) + a) + b;
    const F3 cam = f3(f.camera);
    const F3 dir = normalize(pixel - cam);
    q.origin[local] = P3{cam.x, cam.y, cam.z};
    q.direction[local] = P3{dir.x, dir.y, dir.z};
    q.throughput[local] = P3{1.0f, 1.0f, 1.0f};
    q.rng[local] = make_uint2(static_cast<uint32_t>(s), static_cast<uint32_t>(s >> 32));
    q.slot[local] = path;
    float ht;
    uint32_t hid;
    seed_hit(sc, cam, dir, ht, hid, test_counters);
    q.hit_t[local] = ht;
    q.hit_id[local] = hid;
}

// Caller-supplied rays (rt_trace_rays).
__global__ void k_load_rays(const float* origins, const float* directions, DScene sc, DQueue q, uint32_t n,
                            uint32_t* n_rays_out, unsigned long long* test_counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *n_rays_out = n;
    if (i >= n) return;
    q.origin[i] = P3{origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]};
    q.direction[i] = P3{directions[3 * i], directions[3 * i + 1], directions[3 * i + 2]};
    float ht;
    uint32_t hid;
    seed_hit(sc, f3(&q.origin[i].x), f3(&q.direction[i].x), ht, hid, test_counters);
    q.hit_t[i] = ht;
    q.hit_id[i] = hid;
}

// Hit queue entries -> {distance, family, primitive} records (the layout of the
// C ABI's RtHitRecord), on the device.
struct HitOut {
    float t;
    uint32_t kind, prim;
};
__global__ void k_hit_records(const float* hit_t, const uint32_t* hit_id, HitOut* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = hit_id[i];
    HitOut h;
    h.t = hit_t[i];
    h.kind = id == kHitNone ? 3u : id >> kHitKindShift;
    h.prim = id == kHitNone ? 0xFFFFFFFFu : id & kHitPrimMask;
    out[i] = h;
}

// ------------------------------------------------------------------- shade

// byte / 255.0 (vec3.rs:515-520: u8 -> f32, then true division) for byte k of a texel word,
// without a division and without a table: with z = RN(1/255), q0 = RN(b*z) is one of the two
// neighbours of b/255, r = b - 255*q0 is exact in one FMA, and RN(q0 + r*z) is the correctly
// rounded quotient (Markstein's correction; k_selftest_unit compares all 256 bytes with `/`).
// v_cvt_f32_ubyteK takes the byte straight out of the word.
template <int K>
__device__ __forceinline__ float unit_of_byte(uint32_t texel) {
    const float b = static_cast<float>((texel >> (8 * K)) & 255u);
    constexpr float z = 1.0f / 255.0f;
    const float q0 = b * z;
    const float r = __builtin_fmaf(-255.0f, q0, b);
    return __builtin_fmaf(r, z, q0);
}

// material.rs:192-239
__device__ __forceinline__ F3 color_sample(const DScene& sc, const DMaterial& m, float u_in, float v_in) {
    if (m.color_kind == 0) return f3(m.rgb);
    float c[2] = {u_in, v_in};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float fr = c[k] - truncf(c[k]);  // fract
        float r = fmodf(fr, 1.0f);             // rem_euclid(1.)
        if (r < 0.0f) r += 1.0f;
        c[k] = r;
    }
    c[1] = 1.0f - c[1];
    const uint32_t dim[2] = {m.tex_width, m.tex_height};
    uint32_t i0[2], i1[2];
    float de[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float e = c[k] * static_cast<float>(dim[k] - 1u);
        const float e0f = floorf(e);
        // `as usize` saturates; indices past the edge would be an out-of-bounds
        // panic in the reference and are clamped here instead of faulting.
        uint32_t e0 = !(e0f > 0.0f) ? 0u : (e0f >= static_cast<float>(dim[k] - 1u) ? dim[k] - 1u
                                                                                   : static_cast<uint32_t>(e0f));
        i0[k] = e0;
        i1[k] = min(e0 + 1u, dim[k] - 1u);
        de[k] = e - e0f;
    }
    const uint32_t* tex = in_global(sc.texels) + m.tex_first;
    const size_t tiles_per_row = (static_cast<size_t>(m.tex_width) + (kTexTileW - 1u)) / kTexTileW;
    auto texel = [&](uint32_t x, uint32_t y) {
        const uint32_t t = tex[((y / kTexTileH) * tiles_per_row + x / kTexTileW) * kTexTileTexels + (y % kTexTileH) * kTexTileW + x % kTexTileW];
        return f3(unit_of_byte<0>(t), unit_of_byte<1>(t), unit_of_byte<2>(t));
    };
    const F3 c00 = texel(i0[0], i0[1]), c01 = texel(i0[0], i1[1]);
    const F3 c10 = texel(i1[0], i0[1]), c11 = texel(i1[0], i1[1]);
    const float dx = de[0], dy = de[1];
    const F3 c0 = c00 * (1.0f - dx) + c10 * dx;  // vec3.rs:164-177
    const F3 c1 = c01 * (1.0f - dx) + c11 * dx;
    return c0 * (1.0f - dy) + c1 * dy;
}

struct ShadeArgs {
    DScene scene;
    DQueue cur, next;
    const uint32_t* n_cur;
    uint32_t* n_next;       // zero before launch
    float4* result;         // [slot] final radiance of a finished path
    unsigned long long* hits;  // segments that hit something
    unsigned long long* test_counters;  // [sphere tests, plane tests] of inlined families; null unless counting
    int last_bounce;        // this bounce is max_bounces-1
};

// One iteration of the bounce loop for queue entry i after its nearest hit is
// known, lib.rs:335-371, with Material::scatter (material.rs:23-100) and the
// surface attributes of shapes.rs:91-99,157-165,264-326 inlined.  A path that
// ends here leaves its radiance in result[slot]; for one that goes on (returns
// true) the next ray's origin and direction, the throughput, the generator state
// and the result slot are handed back.
__device__ __forceinline__ bool shade_entry(const DScene& scene, const DQueue& cur, uint32_t i, float4* result, int last_bounce,
                                            F3& no, F3& nd, F3& thr, unsigned long long& rng,
                                            uint32_t& slot, unsigned long long& hits) {
    bool alive = false;  // continues into the next bounce
    const F3 o = stream_load(&cur.origin[i]), d = stream_load(&cur.direction[i]);
    thr = stream_load(&cur.throughput[i]);
    slot = stream_load(&cur.slot[i]);
    const uint32_t id = stream_load(&cur.hit_id[i]);
    // (the generator state and the distance are only needed after a hit, but fetched with the rest of the
    // entry: a load that waited for the id would be one more trip to memory for nearly every path)
    uint2 rs = stream_load(&cur.rng[i]);
    float t = stream_load(&cur.hit_t[i]);
    asm volatile("" : "+v"(rs.x), "+v"(rs.y), "+v"(t));  // (keeps these loads above the test of `id`)
    if (id == kHitNone) {
        // lib.rs:337-345
        const float s = 0.5f * (d.y + 1.0f);
        const F3 sky = f3(0.2f, 0.2f, 0.8f) * (1.0f - s) + f3(1.0f, 1.0f, 1.0f) * s;
        thr = thr * sky;
        in_global(result)[slot] = make_float4(thr.x, thr.y, thr.z, 0.0f);
    } else {
        ++hits;
        rng = (static_cast<unsigned long long>(rs.y) << 32) | rs.x;
        const uint32_t kind = id >> kHitKindShift, prim = id & kHitPrimMask;
        const F3 p = o + d * t;  // bvh.rs:303
        F3 normal;
        float uv0 = 0.0f, uv1 = 0.0f;
        uint32_t mat_i;
        if (kind == 0) {
            const DSphere s = in_global(static_cast<const DSphere*>(scene.family[0].shapes))[prim];
            normal = normalize(p - f3(s.center));
            mat_i = in_global(scene.sphere_material)[prim];
        } else if (kind == 1) {
            const DPlane pl = in_global(static_cast<const DPlane*>(scene.family[1].shapes))[prim];
            normal = f3(pl.normal);
            mat_i = in_global(scene.plane_material)[prim];
        } else {
            const float4* sp4 = in_global(reinterpret_cast<const float4*>(scene.tri_surface + prim));
            const float4 w0 = sp4[0], w1 = sp4[1], w2 = sp4[2], w7 = sp4[7];
            const F3 ta = f3(w0.x, w0.y, w0.z), e1 = f3(w0.w, w1.x, w1.y), e2 = f3(w1.z, w1.w, w2.x);
            const uint32_t mf = __float_as_uint(w7.x);
            mat_i = mf & kTriMaterialMask;
            if (!(mf & kTriHasBary)) {
                normal = normalize(cross(e1, e2));
            } else {
                // shapes.rs:210-227
                const float4 w3 = sp4[3], w4 = sp4[4], w5 = sp4[5], pre = sp4[6];
                const F3 ap = p - ta;
                const float d20 = dot(ap, e1), d21 = dot(ap, e2);
                const float bv = (pre.z * d20 - pre.y * d21) / pre.w;
                const float bw = (pre.x * d21 - pre.y * d20) / pre.w;
                const float bu = 1.0f - bv - bw;
                if (mf & kTriHasNormals) {
                    const F3 n0 = f3(w2.y, w2.z, w2.w), n1 = f3(w3.x, w3.y, w3.z), n2 = f3(w3.w, w4.x, w4.y);
                    normal = normalize((n0 * bu + n1 * bv) + n2 * bw);
                } else {
                    normal = normalize(cross(e1, e2));
                }
                if (mf & kTriHasUv) {
                    uv0 = (w4.z * bu + w5.x * bv) + w5.z * bw;
                    uv1 = (w4.w * bu + w5.y * bv) + w5.w * bw;
                }
            }
        }
        const DMaterial m = in_global(scene.materials)[mat_i];
        no = p + normal * 1e-4f;  // material.rs:29
        int what = 0;            // 0 scattered, 1 absorbed, 2 light
        if (m.kind == 0) {
            const F3 dir = normalize(normal + random_unit(rng));
            const bool near_zero = fabsf(dir.x) < kEpsilon && fabsf(dir.y) < kEpsilon && fabsf(dir.z) < kEpsilon;
            nd = near_zero ? normal : dir;
        } else if (m.kind == 1) {
            const F3 refl = reflect(d, normal);
            if (m.param == 0.0f) {
                nd = refl;
            } else {
                nd = normalize(refl + random_unit(rng) * m.param);
                if (!(dot(nd, normal) > 0.0f)) what = 1;
            }
        } else if (m.kind == 2) {
            float eta;
            F3 nn;
            if (dot(d, normal) < 0.0f) {
                eta = 1.0f / m.param;
                nn = normal;
            } else {
                eta = m.param;
                nn = -normal;
            }
            const float cosv = fminf(dot(-d, nn), 1.0f);
            const float sinv = sqrtf(1.0f - cosv * cosv);
            float r0 = (1.0f - eta) / (1.0f + eta);
            r0 = r0 * r0;
            const float x = 1.0f - cosv;
            const float x5 = x * ((x * x) * (x * x));
            const float reflectance = r0 + (1.0f - r0) * x5;
            bool do_reflect = eta * sinv > 1.0f;
            if (!do_reflect) do_reflect = rng_f32(rng) < reflectance;  // draw only without total reflection
            if (do_reflect) {
                nd = reflect(d, nn);
            } else {
                const F3 perp = (d + nn * cosv) * eta;
                const float disc = 1.0f - eta * eta * (1.0f - cosv * cosv);
                nd = perp + nn * (-sqrtf(disc));
            }
        } else {
            what = 2;
        }
        if (what == 1) {
            in_global(result)[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // lib.rs:359-362
        } else {
            thr = thr * color_sample(scene, m, uv0, uv1);
            if (what == 2 || last_bounce) {
                // a light ends the path; so does the bounce limit, which
                // returns the throughput reached so far (lib.rs:313,374)
                in_global(result)[slot] = make_float4(thr.x, thr.y, thr.z, 0.0f);
            } else {
                alive = true;
            }
        }
    }
    return alive;
}

// A surviving path's entry in the next bounce's queue: the new ray, the state that
// travels with the path, and the nearest hit among the families scanned on the spot.
__device__ __forceinline__ void store_survivor(const DScene& scene, const DQueue& next, uint32_t dst, F3 no, F3 nd, F3 thr,
                                               unsigned long long rng, uint32_t slot, unsigned long long* test_counters) {
    stream_store(&next.origin[dst], no);
    stream_store(&next.direction[dst], nd);
    stream_store(&next.throughput[dst], thr);
    stream_store(&next.rng[dst], make_uint2(static_cast<uint32_t>(rng), static_cast<uint32_t>(rng >> 32)));
    stream_store(&next.slot[dst], slot);
    float ht;
    uint32_t hid;
    seed_hit(scene, no, nd, ht, hid, test_counters);
    stream_store(&next.hit_t[dst], ht);
    stream_store(&next.hit_id[dst], hid);
}

// shade_entry over a whole queue; the surviving paths are compacted into `next`.
__global__ void __launch_bounds__(256) k_shade(ShadeArgs a) {
    __shared__ uint32_t s_wave_live[4];
    __shared__ uint32_t s_block_base;
    const uint32_t n = *a.n_cur;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1) / stride;
    unsigned long long my_hits = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        F3 no = f3(0, 0, 0), nd = f3(0, 0, 0), thr = f3(0, 0, 0);
        unsigned long long rng = 0;
        uint32_t slot = 0;
        const bool alive = i < n && shade_entry(a.scene, a.cur, i, a.result, a.last_bounce, no, nd, thr, rng, slot, my_hits);
        // compaction: one atomic per workgroup and round; survivors keep their order
        const unsigned long long live = __ballot(alive);
        if (lane == 0) s_wave_live[wave] = __popcll(live);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = s_wave_live[0] + s_wave_live[1] + s_wave_live[2] + s_wave_live[3];
            s_block_base = total ? atomicAdd(a.n_next, total) : 0u;
        }
        __syncthreads();
        if (alive) {
            uint32_t dst = s_block_base + bits_below_lane(live);
            for (uint32_t k = 0; k < wave; ++k) dst += s_wave_live[k];
            store_survivor(a.scene, a.next, dst, no, nd, thr, rng, slot, a.test_counters);
        }
        __syncthreads();  // s_wave_live / s_block_base are rewritten next round
    }
    // one atomic per wave for the hit counter
    for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_down(my_hits, off);
    if (lane == 0 && my_hits) atomicAdd(a.hits, my_hits);
}

// ------------------------------------------------------- the whole bounce loop
//
// One launch for all bounces of a batch of paths: every wave owns a share of the
// batch's paths (blocks of 64 dealt to the waves in turn, stored in a stretch of
// the queues that is the wave's own) and takes it through the bounce loop of
// lib.rs:313-373 by itself — trace_loop over its rays, then shade_entry over the
// same entries 64 at a time with the survivors written, packed, to the same
// stretch of the other queue, and so on until none of its paths is alive.
// Waves never wait for each other: there is no launch per bounce, no queue
// counter, no compaction atomic, and on every SIMD some waves traverse (bound by
// instruction issue) while others shade (bound by memory), with all of the
// SIMD's wave slots open to both.  Each path goes through exactly the operations
// of the launch-per-bounce loop, so the results are the same bits.
//
// For scenes whose only traversed family is the triangles (spheres and planes
// absent or scanned where rays are created).  Dynamic LDS: the traversal
// kernel's layout.

// What the shading half of k_path needs, kept in device memory rather than in the kernel's
// argument block: arguments are fetched into scalar registers at the kernel's entry and would
// stay there across the traversal loop, which has none to spare.
struct PathConsts {
    DScene scene;
    DQueue q[2];   // [0] holds the batch's camera rays on entry
    float4* result;
    unsigned long long* hits;      // segments that hit something
    unsigned long long* segments;  // rays traced
};

// (address space 4: memory no kernel writes, so that a wave-uniform read is a scalar load)
typedef const __attribute__((address_space(4))) PathConsts* PathConstsPtr;

struct PathArgs {
    TraceArgs t;   // family, tuning and stack overflow area of the traversal (its queue and cursor fields are unused)
    PathConstsPtr consts;
    uint32_t n_paths;
    uint32_t per_wave;     // queue entries set aside for a wave's paths (a multiple of 64)
    uint32_t max_bounces;
    uint32_t group;        // paths a wave takes through all bounces before it starts on the next ones (a multiple of 64)
};

// A copy of a plain struct that lies in that address space (word by word: class types cannot be
// copied across address spaces by assignment); fields that are not used cost nothing.
template <class T>
__device__ __forceinline__ T read_const(const __attribute__((address_space(4))) T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    T out;
    const __attribute__((address_space(4))) uint32_t* w = (const __attribute__((address_space(4))) uint32_t*)p;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (size_t k = 0; k < sizeof(T) / 4; ++k) o[k] = w[k];
    return out;
}

// (the compiler may not move loads through `p` above this point)
__device__ __forceinline__ PathConstsPtr fetched_here(PathConstsPtr p) {
    asm volatile("" : "+s"(p));
    return p;
}

template <int WAVES, bool FASTSLAB>
__global__ void __launch_bounds__(kTraceBlock * WAVES) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8))) k_path(PathArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float4* const s_nodes = reinterpret_cast<float4*>(s_raw);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const s_wave = s_raw + static_cast<size_t>(a.t.cache_nodes) * sizeof(DNode) + wave * trace_lds_per_wave(a.t.lds_slots);
    uint2* const s_stack = reinterpret_cast<uint2*>(s_wave);
    unsigned long long* const s_res = reinterpret_cast<unsigned long long*>(s_stack + a.t.lds_slots * kTraceBlock);
    uint32_t* const s_mark = reinterpret_cast<uint32_t*>(s_res + kTraceBlock);
    const uint32_t gtid = blockIdx.x * (kTraceBlock * WAVES) + threadIdx.x;
    const uint32_t cache_nodes = a.t.cache_nodes;
    {
        const float4* src = reinterpret_cast<const float4*>(a.t.fam.nodes);
        for (uint32_t k = threadIdx.x; k < cache_nodes * 4u; k += kTraceBlock * WAVES)
            s_nodes[(k & 3u) * cache_nodes + (k >> 2)] = src[k];
        __syncthreads();
    }
    // this wave's paths: blocks g, g + W, g + 2W, ... of the batch's blocks of 64 (see k_raygen), stored from
    // entry g * per_wave on; only the batch's last block can be short
    const uint32_t g = blockIdx.x * WAVES + wave, n_waves = gridDim.x * WAVES;
    const uint32_t n_blocks = (a.n_paths + 63u) >> 6;
    const uint32_t first = g * a.per_wave;
    uint32_t n = g < n_blocks ? ((n_blocks - 1u - g) / n_waves + 1u) * 64u : 0u;
    if (n && g == (n_blocks - 1u) % n_waves && (a.n_paths & 63u)) n -= 64u - (a.n_paths & 63u);
    // The wave's paths go through the bounce loop in groups: a group is traversed and shaded bounce
    // after bounce until none of its paths is alive, then the next group starts.  Waves would
    // otherwise all traverse and all shade at the same times (they have the same amount of work),
    // and traversal and shading would not overlap on a SIMD; with groups, and first groups of
    // different sizes from wave to wave, the waves' phases are spread evenly.
    unsigned long long my_hits = 0, my_segments = 0;
    const uint32_t n_wave = n;
    uint32_t group = max(64u, (a.group * (1u + (g * 5u) % 8u) / 8u) & ~63u);  // the first group: 1/8 .. 8/8 of a group
    for (uint32_t done = 0; done < n_wave; done += group, group = a.group) {
        n = min(group, n_wave - done);
        const uint32_t at = first + done;  // the group's entries, in both queues
        uint32_t c = 0;
        for (uint32_t b = 0; b < a.max_bounces && n > 0; ++b) {
            {
                PathConstsPtr pc = fetched_here(a.consts);
                DQueue cur{};  // the four arrays the traversal touches
                cur.origin = pc->q[c].origin;
                cur.direction = pc->q[c].direction;
                cur.hit_t = pc->q[c].hit_t;
                cur.hit_id = pc->q[c].hit_id;
                trace_loop<2, false, true, WAVES, FASTSLAB>(a.t, cur, s_nodes, s_stack, s_res, s_mark, lane, gtid, cache_nodes, at + n, 64u, at,
                                                            at + n, true);
            }
            if (lane == 0) my_segments += n;
            PathConstsPtr pc = fetched_here(a.consts);
            const DScene scene = read_const(&pc->scene);
            const DQueue cur = read_const(&pc->q[c]), next = read_const(&pc->q[c ^ 1u]);
            float4* const result = pc->result;
            const int last_bounce = b + 1 == a.max_bounces;
            uint32_t n_out = 0;
            for (uint32_t blk = 0; blk < n; blk += kTraceBlock) {
                const uint32_t i = blk + lane;
                F3 no = f3(0, 0, 0), nd = f3(0, 0, 0), thr = f3(0, 0, 0);
                unsigned long long rng = 0;
                uint32_t slot = 0;
                const bool alive = i < n && shade_entry(scene, cur, at + i, result, last_bounce, no, nd, thr, rng, slot, my_hits);
                const unsigned long long live = __ballot(alive);
                if (alive) store_survivor(scene, next, at + n_out + bits_below_lane(live), no, nd, thr, rng, slot, nullptr);
                n_out += __popcll(live);
            }
            n = n_out;
            c ^= 1u;
        }
    }
    for (int off = 32; off > 0; off >>= 1) my_hits += __shfl_down(my_hits, off);
    PathConstsPtr pc = fetched_here(a.consts);
    if (lane == 0 && my_hits) atomicAdd(pc->hits, my_hits);
    if (lane == 0 && my_segments) atomicAdd(pc->segments, my_segments);
}

// ------------------------------------------------------------ accumulation

// lib.rs:274-279: left fold of a pixel's samples in sample order.  `result`
// holds the batch as [sample][pixel]; `first` says the running sum is empty.
__global__ void k_fold(const float4* result, float* sums, uint32_t n_pixels, uint32_t n_samples, int first) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    float r, g, b;
    uint32_t j = 0;
    if (first) {
        const float4 c = result[p];
        r = c.x; g = c.y; b = c.z;
        j = 1;
    } else {
        r = sums[3 * p]; g = sums[3 * p + 1]; b = sums[3 * p + 2];
    }
    for (; j < n_samples; ++j) {
        const float4 c = result[static_cast<size_t>(j) * n_pixels + p];
        r += c.x; g += c.y; b += c.z;
    }
    sums[3 * p] = r; sums[3 * p + 1] = g; sums[3 * p + 2] = b;
}

// `as u8` on f32 (convert.rs:53-55): toward zero, saturating, NaN to 0.
__device__ __forceinline__ uint8_t to_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return static_cast<uint8_t>(v);
}

// lib.rs:283-295 per channel.
__global__ void k_quantize(const float* sums, uint8_t* rgb8, uint32_t n_values, uint32_t n_samples,
                           int running_average, uint32_t sample_iteration) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    const float mean = sums[i] / static_cast<float>(n_samples);
    const float corrected = sqrtf(mean);
    float v = corrected;
    if (running_average) {
        const float old = static_cast<float>(rgb8[i]) / 255.0f;
        const float k = static_cast<float>(sample_iteration);
        v = (old * k + corrected) / (k + 1.0f);
    }
    rgb8[i] = to_u8(v * 255.0f);
}

// Upload-time repacking of one texture: packed RGB8, row by row -> one 32-bit word per texel
// in tiles of kTexTileW x kTexTileH texels (see layout.h); one thread per word of the tiled image.
__global__ void k_expand_texels(const uint8_t* rgb8, uint32_t* out, uint32_t width, uint32_t height) {
    const unsigned long long tiles_per_row = (static_cast<unsigned long long>(width) + (kTexTileW - 1u)) / kTexTileW;
    const unsigned long long tile_rows = (static_cast<unsigned long long>(height) + (kTexTileH - 1u)) / kTexTileH;
    const unsigned long long i = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= tiles_per_row * tile_rows * kTexTileTexels) return;
    const unsigned long long tile = i / kTexTileTexels;
    const uint32_t in_tile = static_cast<uint32_t>(i % kTexTileTexels);
    const unsigned long long x = (tile % tiles_per_row) * kTexTileW + in_tile % kTexTileW;
    const unsigned long long y = (tile / tiles_per_row) * kTexTileH + in_tile / kTexTileW;
    uint32_t word = 0;  // padding of the edge tiles: never sampled
    if (x < width && y < height) {
        const uint8_t* t = rgb8 + 3 * (y * width + x);
        word = static_cast<uint32_t>(t[0]) | (static_cast<uint32_t>(t[1]) << 8) | (static_cast<uint32_t>(t[2]) << 16);
    }
    out[i] = word;
}

// Self-test of the shared-reciprocal division: compares div_by2 with `/` bit for
// bit on pseudo-random operands drawn over the whole regime the traversal kernel
// admits (|b| in [2^-30, 2), a == 0 or |a| in [2^-84, 2^60), either sign), on
// operands at the edges of that regime, on divisors whose significand is all
// ones, and on dividends built so that the quotient lies next to the midpoint of
// two neighbouring values (where a less careful sequence rounds the wrong way).
// Counts quotients that differ (other than in the sign of a zero).
__global__ void k_selftest_division(unsigned long long seed, uint32_t per_thread, unsigned long long* mismatches) {
    unsigned long long s = seed + (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        s += kWy0;
        const unsigned long long a = s, b = s ^ kWy1;
        return (a * b) ^ __umul64hi(a, b);
    };
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < per_thread; ++i) {
        const unsigned long long r0 = next(), r1 = next();
        // b: exponent 97..127 (2^-30 .. 2^0), random mantissa and sign
        const uint32_t be = 97u + static_cast<uint32_t>(r0 % 31u);
        float b = __uint_as_float((static_cast<uint32_t>(r0 >> 63) << 31) | (be << 23) | (static_cast<uint32_t>(r0 >> 8) & 0x7FFFFFu));
        // a: exponent 43..186 (2^-84 .. 2^59), random mantissa and sign; sometimes 0
        const uint32_t ae = 43u + static_cast<uint32_t>(r1 % 144u);
        float a0 = __uint_as_float((static_cast<uint32_t>(r1 >> 63) << 31) | (ae << 23) | (static_cast<uint32_t>(r1 >> 8) & 0x7FFFFFu));
        if ((i & 1023u) == 0u) a0 = 0.0f;
        if ((i & 1023u) == 2u) { b = (i & 1024u) ? 0x1p-30f : -0x1.fffffep0f; }
        if ((i & 7u) == 3u) { b = __uint_as_float((__float_as_uint(b) & 0xFF800000u) | 0x7FFFFFu); }  // significand all ones
        // a1: b times a value half-way between two neighbours (25 significant bits, the
        // last one set), rounded, then moved by -1..+2 units in the last place
        float a1;
        {
            const uint32_t me = 70u + static_cast<uint32_t>((r1 >> 40) % 90u);  // quotient 2^-57 .. 2^32
            const double mid = static_cast<double>(__uint_as_float((me << 23) | (static_cast<uint32_t>(r0 >> 33) & 0x7FFFFFu))) *
                               (1.0 + 0x1p-24 / (1.0 + static_cast<double>(static_cast<uint32_t>(r0 >> 33) & 0x7FFFFFu) * 0x1p-23));
            a1 = static_cast<float>(mid * static_cast<double>(b));
            a1 = __uint_as_float(__float_as_uint(a1) + static_cast<uint32_t>(r1 >> 20 & 3u) - 1u);
            if (!(fabsf(a1) >= 0x1p-84f && fabsf(a1) < 0x1p60f)) a1 = a0;
        }
        if ((i & 1023u) == 1u) { a0 = 0x1p-84f; a1 = -0x1.fffffep59f; }
        const v2f q = div_by2(v2f{a0, a1}, b, reciprocal_of(b));
        const float w0 = a0 / b, w1 = a1 / b;
        const bool ok0 = __float_as_uint(q.x) == __float_as_uint(w0) || (q.x == 0.0f && w0 == 0.0f);
        const bool ok1 = __float_as_uint(q.y) == __float_as_uint(w1) || (q.y == 0.0f && w1 == 0.0f);
        bad += (ok0 ? 0u : 1u) + (ok1 ? 0u : 1u);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Self-test of unit_of_byte: all 256 bytes, in each of a word's three byte positions, against
// `static_cast<float>(b) / 255.0f`.  Counts the differences.
__global__ void k_selftest_unit(unsigned long long* mismatches) {
    const uint32_t b = threadIdx.x & 255u;
    const float want = static_cast<float>(b) / 255.0f;
    const uint32_t noise = 0xA5u ^ b;  // other bytes of the word must not matter
    unsigned long long bad = 0;
    bad += __float_as_uint(unit_of_byte<0>(b | (noise << 8) | (noise << 16) | (noise << 24))) != __float_as_uint(want);
    bad += __float_as_uint(unit_of_byte<1>((b << 8) | noise | (noise << 16) | (noise << 24))) != __float_as_uint(want);
    bad += __float_as_uint(unit_of_byte<2>((b << 16) | noise | (noise << 8) | (noise << 24))) != __float_as_uint(want);
    if (bad) atomicAdd(mismatches, bad);
}

// Self-test of rcp_midrange: every binary32 value of either sign with magnitude
// in [2^-20, 2^60) against `1.0f / b`, bit for bit.  Counts the differences.
__global__ void k_selftest_reciprocal(unsigned long long* mismatches) {
    const uint32_t lo = __float_as_uint(kRcpLo), hi = __float_as_uint(kRcpHi);
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (uint32_t u = lo + blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += stride) {
        const float b = __uint_as_float(u);
        bad += __float_as_uint(rcp_midrange(b)) != __float_as_uint(1.0f / b);
        bad += __float_as_uint(rcp_midrange(-b)) != __float_as_uint(1.0f / -b);
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace rtd
