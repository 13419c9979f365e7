/*
 * rt_mi355.h — C ABI of the MI355X render-loop library (librt_mi355.so).
 *
 * This is the seam a host program binds instead of calling the reference's
 * `Scene::render` (/root/reference/src/lib.rs:188).  The host keeps scene
 * parsing, OBJ/MTL/PPM loading and BVH construction and hands the finished,
 * immutable scene over as flat little-endian f32/u32 arrays (RtSceneDesc);
 * the library owns everything from ray generation to the quantised pixels.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative RtStatus; the message
 *     for the last failure on a context is available from rt_last_error();
 *     nothing aborts and nothing throws across this boundary;
 *   - the caller owns every pointer it passes in; rt_upload_scene copies, so
 *     the arrays may be freed as soon as it returns;
 *   - a context is single-caller (not re-entrant) and bound to one GPU;
 *   - all structs are plain C, naturally aligned, no padding surprises:
 *     every field is 4 bytes wide except the 8-byte ones, which sit at
 *     8-byte offsets.
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1u

/* Leaf marker in RtNode.b, "absent" marker for optional per-triangle tables. */
#define RT_LEAF_BIT 0x80000000u
#define RT_NO_INDEX 0xFFFFFFFFu

/* Seed of the reference's generator, /root/reference/src/rng.rs:10. */
#define RT_RNG_SEED 0x0ef6f79ed30ba75aull

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument / inconsistent scene description   */
    RT_ERR_NO_DEVICE = -2, /* no usable GPU (the library never falls back)     */
    RT_ERR_HIP = -3,       /* a HIP runtime call failed                        */
    RT_ERR_OOM = -4,       /* device or host allocation failed                 */
    RT_ERR_STATE = -5      /* call order (e.g. render before upload)           */
} RtStatus;

/* MaterialKind, /root/reference/src/material.rs:113-119. */
typedef enum RtMaterialKind {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1, /* param = fuzziness        */
    RT_MAT_GLASS = 2, /* param = refractive index */
    RT_MAT_LIGHT = 3
} RtMaterialKind;

/* ColorKind, /root/reference/src/material.rs:140-148. */
typedef enum RtColorKind { RT_COLOR_SOLID = 0, RT_COLOR_TEXTURE = 1 } RtColorKind;

/* Shape families in the order `ray_color` consults them,
 * /root/reference/src/lib.rs:314-333 (first wins on equal distance). */
typedef enum RtShapeKind {
    RT_SHAPE_SPHERE = 0,
    RT_SHAPE_PLANE = 1,
    RT_SHAPE_TRIANGLE = 2,
    RT_SHAPE_NONE = 3
} RtShapeKind;

/* How pseudo-random draws are assigned to samples. */
typedef enum RtRngMode {
    /* One wyrand substream per (sample, pixel): stream k = sample*W*H + y*W + x
     * starts at state seed + k*(2+3*max_bounces)*0x2d358dccaa6c78a5 (mod 2^64),
     * i.e. a jump inside the reference's single Weyl sequence.  The only mode
     * the GPU implements; results do not depend on GPU count or scheduling. */
    RT_RNG_SUBSTREAM = 0,
    /* One stream consumed in the literal order of the reference running on a
     * single thread (/root/reference/src/lib.rs:226-296).  CPU oracle only. */
    RT_RNG_SERIAL = 1
} RtRngMode;

/* BvhNode + BvhNodeKind, /root/reference/src/bvh.rs:15-21,318-325.
 * Array in the reference's push order (depth-first pre-order, root = 0).
 *   branch: a = children[0], b = children[1]           (b < RT_LEAF_BIT)
 *   leaf:   a = shapes_range.start, b = count | RT_LEAF_BIT              */
typedef struct RtNode {
    float min[3];
    uint32_t a;
    float max[3];
    uint32_t b;
} RtNode;

/* Sphere, /root/reference/src/shapes.rs:43-47. */
typedef struct RtSphere {
    float center[3];
    float radius;
    uint32_t material;
} RtSphere;

/* Plane, /root/reference/src/shapes.rs:119-123 (normal already normalised,
 * /root/reference/src/config.rs:74). */
typedef struct RtPlane {
    float point[3];
    float normal[3];
    uint32_t material;
} RtPlane;

/* Triangle + NormalsTextureCoordinates, /root/reference/src/shapes.rs:185-193,
 * 354-370.  Stored as vertex a and edges e1 = b-a, e2 = c-a (:203-205).
 * An index equal to RT_NO_INDEX means that table is absent for the triangle;
 * bary_idx is present iff normals_idx or uv_idx is.                        */
typedef struct RtTriangle {
    float a[3];
    float e1[3];
    float e2[3];
    uint32_t normals_idx; /* into RtSceneDesc.vertex_normals   */
    uint32_t uv_idx;      /* into RtSceneDesc.tex_coords       */
    uint32_t bary_idx;    /* into RtSceneDesc.bary_precomputed */
    uint32_t material;
} RtTriangle;

/* Material, /root/reference/src/material.rs:11-15.  A textured material may be
 * referenced by triangles only: rt_upload_scene refuses a sphere or plane with
 * one (the reference's scene file cannot express it, config.rs:147-155, and
 * their texture coordinates, shapes.rs:91-99,157-165, are not computed).  For a texture,
 * tex_offset is the byte offset of its first texel in RtSceneDesc.texels
 * (packed RGB8, row-major, top row first, as read from a P6 file). */
typedef struct RtMaterial {
    uint32_t kind;       /* RtMaterialKind */
    float param;         /* fuzziness / refractive index, else 0 */
    uint32_t color_kind; /* RtColorKind */
    float rgb[3];        /* solid colour */
    uint32_t tex_width;
    uint32_t tex_height;
    uint64_t tex_offset;
} RtMaterial;

/* The immutable part of `Scene`, /root/reference/src/lib.rs:103-150:
 * Shapes (three shape arrays + three per-triangle side tables), Bvhs (three
 * node arrays) and the interned material table.  Shape arrays are in
 * post-BVH-build order (the build permutes them in place,
 * /root/reference/src/config.rs:113-134), so "primitive id" means the index
 * into these arrays. */
typedef struct RtSceneDesc {
    const RtNode* sphere_nodes;
    const RtNode* plane_nodes;
    const RtNode* triangle_nodes;
    const RtSphere* spheres;
    const RtPlane* planes;
    const RtTriangle* triangles;
    const float* vertex_normals;   /* [n_vertex_normals][3][3] */
    const float* tex_coords;       /* [n_tex_coords][3][2]     */
    const float* bary_precomputed; /* [n_bary][4] = d00,d01,d11,denominator */
    const RtMaterial* materials;
    const uint8_t* texels;
    uint64_t n_texel_bytes;
    uint32_t n_sphere_nodes;
    uint32_t n_plane_nodes;
    uint32_t n_triangle_nodes;
    uint32_t n_spheres;
    uint32_t n_planes;
    uint32_t n_triangles;
    uint32_t n_vertex_normals;
    uint32_t n_tex_coords;
    uint32_t n_bary;
    uint32_t n_materials;
} RtSceneDesc;

/* Screen + Camera + the two optional scene-file switches,
 * /root/reference/src/lib.rs:103-107,378-418. */
typedef struct RtRenderParams {
    float top_left[3];
    float top_edge[3];
    float left_edge[3];
    float camera[3];
    uint32_t width;
    uint32_t height;
    uint32_t samples_per_pixel;
    uint32_t max_bounces;
    uint32_t incremental;       /* 0 = None, else samples per pass          */
    uint32_t has_continue;      /* 0 = None                                  */
    uint32_t continue_sampling; /* pass number to resume at (if has_continue) */
    uint32_t rng_mode;          /* RtRngMode                                 */
    uint64_t rng_seed;          /* RT_RNG_SEED reproduces the reference      */
    /* Sub-rectangle of the image to work on; tile_w == 0 means the full
     * image.  Pixel buffers passed with a tile are tile-sized, row-major. */
    uint32_t tile_x;
    uint32_t tile_y;
    uint32_t tile_w;
    uint32_t tile_h;
} RtRenderParams;

/* Result of a nearest-hit query for one ray. */
typedef struct RtHitRecord {
    float t;       /* +inf when nothing was hit */
    uint32_t kind; /* RtShapeKind               */
    uint32_t prim; /* index into that family's shape array, RT_NO_INDEX if none */
} RtHitRecord;

/* Work counters since creation or the last rt_reset_stats.
 * A "segment" is one iteration of the bounce loop,
 * /root/reference/src/lib.rs:313 — the unit of the Mrays/s metric.
 * The three traversal counters are filled only while counting is enabled
 * with rt_set_option(RT_OPT_COUNT_TRAVERSAL, 1) (slower kernels). */
typedef struct RtStats {
    uint64_t paths;          /* camera samples started                         */
    uint64_t segments;       /* bounce-loop iterations                         */
    uint64_t hits;           /* segments that hit a shape                      */
    uint64_t branch_visits;  /* Branch nodes expanded (two box tests each)     */
    uint64_t prim_tests[3];  /* sphere / plane / triangle intersection tests   */
    uint64_t trace_launches; /* launches of the triangle traversal kernel      */
    double trace_ms;         /* their summed duration (HIP events); needs
                                rt_set_option(RT_OPT_TIME_KERNELS, 1)          */
    double render_ms;        /* wall time inside rt_render / rt_render_sums    */
} RtStats;

typedef enum RtOption {
    RT_OPT_COUNT_TRAVERSAL = 1, /* 0/1: fill branch_visits / prim_tests          */
    RT_OPT_TIME_KERNELS = 2,    /* 0/1: bracket traversal launches with events   */
    RT_OPT_SAMPLES_IN_FLIGHT = 3, /* samples per pixel traced concurrently (>=1) */
    /* Tuning knobs of the traversal kernel; results never depend on them. */
    RT_OPT_TRACE_LDS_SLOTS = 4, /* stack entries per ray kept in LDS (2..64)      */
    RT_OPT_TRACE_PHASE_MIN = 5, /* lanes that must wait for a step to run it      */
    RT_OPT_TRACE_CHUNK = 6,     /* rays a wave reserves per queue atomic          */
    RT_OPT_TRACE_SHARE = 7,     /* 0/1: deal a wave's pending triangle tests out to all its lanes */
    RT_OPT_TRACE_PAIR_MIN = 8,  /* pending tests that trigger such a shared step  */
    RT_OPT_TRACE_WAVES = 9,     /* waves per traversal workgroup: 1, 8 or 16      */
    RT_OPT_TRACE_CACHE_NODES = 10, /* top-of-tree branch records a multi-wave workgroup keeps in LDS (0..2048) */
    RT_OPT_TRACE_REFILL_MIN = 11, /* idle lanes of a wave that trigger a refill from the queue (1..64) */
    RT_OPT_CHAINS = 13,         /* 1..4: independent chains (own queues, own stream) a large batch of paths is split into; each chain's tails and launch gaps are filled by the others (default 2) */
    RT_OPT_SHADE_GROUPS_PER_CU = 14, /* grid of the shading kernel: workgroups of 256 threads per compute unit (1..64, default 8) */
    /* 0/1, default 0.  NOT bit-exact when on: the triangle family's ray/box test
     * (/root/reference/src/bvh.rs:25-43) multiplies by the reciprocal of the direction,
     * fused with the subtraction, instead of dividing.  Box distances then differ from the
     * reference's in their last places, which can reorder visits and flip a cull where a box
     * and the closest hit are equally far to rounding; hit distances themselves, triangle
     * tests and shading stay exact.  Ignored while RT_OPT_COUNT_TRAVERSAL is on. */
    RT_OPT_FAST_SLAB = 15,
    /* The whole bounce loop of a batch as one launch in which every wave takes its own block of
     * paths through all bounces (no launch, counter or barrier per bounce): 0 never, 1 whenever
     * the scene allows it (triangles traversed; spheres and planes absent or few enough to be
     * scanned where rays are created), 2 (default) then and for batches below 3 * 2^22 paths,
     * where it is the faster loop (several times faster for previews and single-sample passes),
     * and for larger batches on scenes where it measures faster: the first two large batches after
     * an upload are each rendered as two halves, one through each loop, and the faster loop is kept
     * (per-bounce launches on scenes whose rays are expensive, the one launch where most rays
     * miss or end early).  Results do not depend on it. */
    RT_OPT_FUSED_LOOP = 17,
    RT_OPT_PASSES_IN_FLIGHT = 21, /* rt_render: passes (of `incremental` samples) whose paths are traced together before the passes are finished one by one, in order — same bytes; the image is rewritten once per such group. 0 (default) = as many as make 2^26 paths, 1 = one pass at a time */
    RT_OPT_TRACE_GUIDED = 20, /* 0/1 (default 1): a wave's reservations from the ray queue shrink as the queue runs out, so that a launch's waves finish together */
    RT_OPT_DEBUG_THROW = 19, /* test hook: 1 makes this very rt_set_option call fail inside with std::bad_alloc, 2 with another
                              * exception; the call must come back with RT_ERR_OOM / RT_ERR_HIP (nothing throws across the boundary) */
    RT_OPT_PATH_GROUP = 18, /* tuning of that launch: paths a wave takes through all bounces at a time (multiple of 64; default 16384) */
    RT_OPT_TRACE_WAVES_PER_CU = 12 /* traversal waves resident per compute unit (0 = as many as fit, else 1..32), for the per-bounce traversal launches and the one-launch bounce loop alike: leaves room for kernels of another context on the same GPU */
} RtOption;

typedef struct RtContext RtContext;

/* Called on the calling thread after each completed pass with the number of
 * samples per pixel finished so far; stands in for the reference's
 * `print!("\rSamples: {}")`, /root/reference/src/lib.rs:238-246. */
typedef void (*RtProgressFn)(uint32_t samples_done, void* user);

uint32_t rt_abi_version(void);

/* Create a context on GPU `device_ordinal`.  Fails with RT_ERR_NO_DEVICE when
 * there is none: there is no CPU fallback behind this library.  Fails with
 * RT_ERR_HIP when two HIP runtimes are mapped in the process (this library's and
 * another package's private copy): rt_last_error(NULL) then names both.  One
 * context drives one GPU; a host that renders on several GPUs creates one
 * context per GPU (one thread or process each) and adds the radiance sums of
 * rt_render_sums itself or with a collective. */
int rt_create(int device_ordinal, RtContext** out);
void rt_destroy(RtContext* ctx);
/* Message of the last failure on `ctx`; with NULL, of this thread's last failed rt_create. */
const char* rt_last_error(const RtContext* ctx);

int rt_set_option(RtContext* ctx, uint32_t option, uint64_t value);

/* Validate, re-lay-out and copy the scene to the GPU.  Replaces the hand-over
 * of `&'static Scene` through `SCENE`, /root/reference/src/lib.rs:40. */
int rt_upload_scene(RtContext* ctx, const RtSceneDesc* scene);

/* Drop-in for `Scene::render`, /root/reference/src/lib.rs:188-301.
 * `rgb8` is the pixel area of the output image (width*height*3 bytes, or the
 * tile); it may be the mmap'd file itself.  It is read first when
 * `incremental` or `continue` is set (the running average of :286-292 uses
 * the stored bytes) and rewritten after every pass — or, when the passes are
 * short (few samples of a small frame, e.g. the `incremental(1)` of the
 * reference's shipped scenes), after every group of passes: the paths of up to
 * 2^26 / (paths per pass) passes are traced together and the passes then
 * finished one by one in order, which stores the same bytes several times as
 * fast (RT_OPT_PASSES_IN_FLIGHT = 1 turns the grouping off).  `progress` is
 * called once per pass either way. */
int rt_render(RtContext* ctx, const RtRenderParams* params, uint8_t* rgb8,
              RtProgressFn progress, void* user);

/* Linear radiance, summed (left fold in sample order, :274-279) over samples
 * [sample_begin, sample_begin + sample_count) of every pixel of the tile:
 * 3 floats per pixel.  `sums` is a host pointer, or a device pointer on this
 * context's GPU when `sums_on_device` is non-zero.  This is the piece a
 * multi-GPU driver partitions by sample range and reduces. */
int rt_render_sums(RtContext* ctx, const RtRenderParams* params,
                   uint32_t sample_begin, uint32_t sample_count, float* sums,
                   int sums_on_device);

/* The same fold carried on from an earlier call: on entry `sums` holds the
 * fold of samples [0, sample_begin) (ignored when sample_begin is 0), on return
 * that of [0, sample_begin + sample_count) — bit for bit what one call over the
 * whole range gives, since the fold of :274-279 is sequential in sample order.
 * With rt_resolve on the running total this is a progressive render whose
 * preview does not go through the 8-bit running average of :286-292 (which
 * loses precision every pass and drifts dark) and whose final image equals
 * the non-incremental render of the same sample count.  The reference has no
 * such mode; the caller keeps `sums` (for instance in a file beside the image). */
int rt_render_sums_onto(RtContext* ctx, const RtRenderParams* params,
                        uint32_t sample_begin, uint32_t sample_count, float* sums,
                        int sums_on_device);

/* The two halves of rt_render_sums_onto as separate calls, for a multi-GPU
 * driver that wants the single-GPU image bit for bit: every GPU traces its
 * contiguous share of a pass's samples at the same time (rt_trace_samples: the
 * per-sample radiance stays on the device), then the running sums travel from
 * GPU to GPU in sample order and each folds its samples onto them
 * (rt_fold_samples with carry_on = 1; the first GPU starts the fold with
 * carry_on = 0).  The fold of :274-279 is then the same sequence of additions
 * as on one GPU.  rt_trace_samples traces as many of the `sample_count` samples
 * as fit in flight at once and reports the number in `*traced`; the caller
 * folds them and calls again for the rest.  A later render or trace call on the
 * context discards samples not folded yet. */
int rt_trace_samples(RtContext* ctx, const RtRenderParams* params,
                     uint32_t sample_begin, uint32_t sample_count, uint32_t* traced);
int rt_fold_samples(RtContext* ctx, const RtRenderParams* params, int carry_on,
                    float* sums, int sums_on_device);
/* Sizes the context's device workspace for up to `sample_count` samples of the
 * region in flight and reports in `*in_flight` how many one rt_trace_samples
 * call will then take (fewer than asked when the memory or RT_OPT_SAMPLES_IN_FLIGHT
 * says so).  A driver of several GPUs asks every GPU first and then hands out
 * pieces that all of them trace in one call, so that they stay in step.  The
 * same allocation happens inside the render and trace calls when they need it. */
int rt_reserve_samples(RtContext* ctx, const RtRenderParams* params, uint32_t sample_count,
                       uint32_t* in_flight);

/* Mean, gamma and quantisation of :283-295: rgb8 = u8(sqrt(sums/n)*255), or
 * with `running_average` the blend with the bytes already in `rgb8` weighted
 * by `sample_iteration`.  `sums` as in rt_render_sums; `rgb8` is host memory. */
int rt_resolve(RtContext* ctx, const RtRenderParams* params, const float* sums,
               int sums_on_device, uint32_t n_samples, int running_average,
               uint32_t sample_iteration, uint8_t* rgb8);

/* Nearest hit of the camera ray of sample `sample_index` for every pixel of
 * the tile (one iteration of :313-333): the parity hook for primitive ids.
 * With `jitter` zero the two sub-pixel offsets of :262-263 are taken as 0. */
int rt_trace_primary(RtContext* ctx, const RtRenderParams* params,
                     uint32_t sample_index, int jitter, RtHitRecord* out);

/* Nearest hit for caller-supplied rays (origin[3], direction[3] per ray,
 * host memory): `Scene::ray_color`'s three `closest_shape` calls plus the
 * `min_by`, /root/reference/src/lib.rs:314-333. */
int rt_trace_rays(RtContext* ctx, const float* origins, const float* directions,
                  uint64_t n_rays, RtHitRecord* out);

/* rt_trace_rays for callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  Same results as rt_trace_rays, bit for bit, but every intermediate
 * (ray queue, hit records, traversal stack overflow) lives in a fixed 64 MiB
 * array that is part of the library's code object — one copy per GPU, held until
 * the process ends — so the call performs no hipMalloc / hipFree and no host
 * allocation.  At most 524288 rays per call (RT_ERR_INVALID beyond that).
 * `origins`, `directions` and `out` are host pointers, or device pointers on this
 * context's GPU when `on_device` is non-zero.  The call works on the context's
 * own non-blocking stream, waits for it and copies results before it returns:
 * it is a blocking call (not one to make while a stream capture is active on the
 * calling thread), and device inputs must be complete — and `out` no longer in
 * use — by the caller's own synchronisation before the call, since the library's
 * stream is not ordered against any stream of the caller.  All contexts on one
 * GPU share that GPU's array: calls that overlap in time on one GPU (from
 * different contexts or threads) race; calls in sequence are independent of
 * each other. */
int rt_trace_rays_noalloc(RtContext* ctx, const float* origins, const float* directions,
                          uint64_t n_rays, RtHitRecord* out, int on_device);

/* `BvhNode::new` for the triangle family, /root/reference/src/bvh.rs:48-224, on
 * the GPU: the same contract as the host builder (rth_build_bvh_triangles in
 * rt_host.h) and the same result — `shapes` is permuted in place exactly as the
 * reference permutes its slice (/root/reference/src/config.rs:113-134) and
 * `*nodes` receives the node array in the reference's push order (malloc'd;
 * release with rt_free).  Box coordinates equal the host builder's as numbers
 * (a zero may carry the other sign).  Needs no scene.  Coordinates must be
 * finite: otherwise RT_ERR_INVALID, and the family is built on the host. */
int rt_build_bvh_triangles(RtContext* ctx, RtTriangle* shapes, uint32_t n, RtNode** nodes,
                           uint32_t* n_nodes);
/* The same for the sphere family (boxes centre -/+ radius, /root/reference/src/shapes.rs:105-115).
 * Planes have infinite boxes and are never split, so they need no builder. */
int rt_build_bvh_spheres(RtContext* ctx, RtSphere* shapes, uint32_t n, RtNode** nodes,
                         uint32_t* n_nodes);
void rt_free(void* p);

/* Device self-tests (no counterpart in the reference).  RT_SELF_TEST_DIVISION
 * checks, on about `n_cases` operand pairs, that the traversal kernel's
 * shared-reciprocal division returns exactly the bits of an IEEE division over
 * the whole operand range it is used for; `failures` receives the number of
 * differing quotients (0 on a correct build).  RT_SELF_TEST_RECIPROCAL does the
 * same for the triangle test's reciprocal of the determinant, exhaustively
 * (every binary32 value in its range; `n_cases` is ignored). */
#define RT_SELF_TEST_DIVISION 1u
#define RT_SELF_TEST_RECIPROCAL 2u
#define RT_SELF_TEST_UNIT_BYTE 3u /* texel byte / 255 without a division: all 256 bytes against `/` */
int rt_self_test(RtContext* ctx, uint32_t which, uint64_t n_cases, uint64_t* failures);

/* Page-locked host memory for pixel buffers handed to rt_render / rt_resolve /
 * rt_group_render (no counterpart in the reference, whose image is its memory
 * mapping, /root/reference/src/mmap.rs:41-77).  Copies between such a buffer and
 * the GPU run at the full rate of the host link; from ordinary pageable memory —
 * the mapping itself included — the runtime stages them through small buffers of
 * its own at a fraction of that.  Worth it for tiles of hundreds of megabytes
 * (the tile loop of a 16K image: copy the tile's rows from the mapping into such
 * a buffer, render, copy back).  NULL when the memory cannot be had; release
 * with rt_host_free.  Needs a GPU in the process, no context. */
void* rt_host_alloc(size_t bytes);
void rt_host_free(void* p);

int rt_get_stats(RtContext* ctx, RtStats* out);
int rt_reset_stats(RtContext* ctx);

/* ---- several GPUs of one node behind one handle ------------------------------
 *
 * The reference's seam is one `Scene::render(&self)` call
 * (/root/reference/src/lib.rs:188-301) whose threads share the pixels of a frame
 * (:195-235).  A group is that call over a list of GPUs: the scene is replicated,
 * a pass's samples are split into contiguous shares that the GPUs trace at the
 * same time (one host thread per GPU inside the library), and the per-pixel sums
 * then travel from GPU to GPU in sample order, each folding its samples onto
 * them.  The fold of :274-279 is therefore the same sequence of additions as on
 * one GPU: rt_group_render stores the bytes rt_render stores, and
 * rt_group_render_sums returns the floats rt_render_sums returns, whatever the
 * number of GPUs (a GPU may be listed more than once).  The calls of one group
 * must not overlap in time. */
typedef struct RtGroup RtGroup;
/* One context per listed ordinal (rt_create's conditions apply to each). */
int rt_group_create(const int* device_ordinals, uint32_t n_devices, RtGroup** out);
void rt_group_destroy(RtGroup* group);
/* Message of the last failure on `group`; with NULL, of this thread's last failed rt_group_create. */
const char* rt_group_last_error(const RtGroup* group);
uint32_t rt_group_size(const RtGroup* group);
/* The context of member `index` (owned by the group), e.g. for rt_trace_primary; NULL beyond the end. */
RtContext* rt_group_member(RtGroup* group, uint32_t index);
/* rt_set_option / rt_upload_scene on every member (uploads run side by side). */
int rt_group_set_option(RtGroup* group, uint32_t option, uint64_t value);
int rt_group_upload_scene(RtGroup* group, const RtSceneDesc* scene);
/* rt_render over the group: same arguments, same bytes.  A pass of many samples is
 * shared out by samples (the sums handed from GPU to GPU in sample order); a render
 * of many short passes is shared out by rows instead — bands of the region, four per
 * GPU, pulled off a counter, each band's passes rendered by one GPU straight into its
 * rows of `rgb8`, in groups of passes after each of which the buffer is a whole image —
 * so that no GPU idles over a one-sample pass and nothing moves between GPUs. */
int rt_group_render(RtGroup* group, const RtRenderParams* params, uint8_t* rgb8,
                    RtProgressFn progress, void* user);
/* rt_render_sums over the group (`sums` is host memory): same floats. */
int rt_group_render_sums(RtGroup* group, const RtRenderParams* params,
                         uint32_t sample_begin, uint32_t sample_count, float* sums);
/* Counters summed over the members; render_ms is the largest member's. */
int rt_group_get_stats(RtGroup* group, RtStats* out);
int rt_group_reset_stats(RtGroup* group);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H */
