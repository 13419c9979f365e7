// Device-side data layout of the scene and of the path queues in HBM.
//
// Scene (built once by rt_upload_scene, read-only afterwards)
//   * BVH: one 64-byte record per *branch* node holding both children's boxes
//     and both child references, so expanding a branch is one aligned 64-byte
//     fetch (four 16-byte loads) instead of the parent plus two child records.
//     The reference visits a branch by box-testing its two children
//     (/root/reference/src/bvh.rs:259-267); this record is exactly the data
//     that step reads.  Branch records keep the reference's depth-first order.
//   * a child reference is 32 bits:
//        bit 31 = 0: index of a branch record
//        bit 31 = 1: leaf; bits 27..30 = shape count 0..14 and bits 0..26 the
//                    first shape index, or count field 15 = "long leaf" with
//                    bits 0..26 indexing a {first, count} table
//     so a traversal stack entry is 8 bytes: entry distance + reference.
//   * triangles: 36 bytes each in three arrays (two 16-byte loads and one
//     4-byte load, consecutive triangles consecutive in each): a, e1, e2 — the
//     operands of /root/reference/src/shapes.rs:232-259.
//     Shading attributes live in separate arrays, untouched by traversal.
//
// Path queues (one record per live path, structure-of-arrays so that a wave
// reading 64 consecutive paths issues full-width coalesced loads)
//   origin float4 | direction float4 | throughput float4 | rng u64 | slot u32
//   | hit distance f32 | hit id u32
//   Two copies (current / next bounce); the shade kernel compacts survivors
//   into the next one.
#pragma once
#include <cstdint>

namespace rtd {

constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLeafCountShift = 27;
constexpr uint32_t kLeafFirstMask = (1u << kLeafCountShift) - 1u;
constexpr uint32_t kLeafLong = 15u;
constexpr uint32_t kHitNone = 0xFFFFFFFFu;
constexpr uint32_t kHitKindShift = 30;
constexpr uint32_t kHitPrimMask = (1u << kHitKindShift) - 1u;

struct alignas(16) DNode {
    // per axis: child 0 min, child 1 min, child 0 max, child 1 max — the two
    // children side by side, so that each loaded register pair is one operand
    // of the packed (two-wide) arithmetic of the slab test
    float x[4];
    float y[4];
    float z[4];
    uint32_t ref0, ref1, pad0, pad1;
};
static_assert(sizeof(DNode) == 64, "branch record must be one 64-byte line");

// The triangles of the traversal: 36 bytes each, as three arrays one after the other in one
// allocation of n triangles — n x {ex[2], ey[2]}, n x {ez[2], a.x, a.y}, n x a.z — where
// ex/ey/ez hold (e2, e1) of one axis side by side: the two cross products of the intersection
// test, d x e2 and (o - a) x e1, run as the two halves of the packed (two-wide) arithmetic, so
// each loaded register pair is one operand.  Arrays rather than one record per triangle because
// the lanes of a shared triangle step read runs of consecutive triangles (a leaf's): a load
// instruction then touches a third of the memory lines it would touch in 48-byte records, and
// the level-one cache's work goes by the line.
struct alignas(16) DTriHalf {
    float v[4];
};
__host__ __device__ inline size_t triangle_array_bytes(size_t n) { return n * (2 * sizeof(DTriHalf) + sizeof(float)); }

struct alignas(16) DSphere {
    float center[3];
    float radius;
};

struct alignas(16) DPlane {
    float point[3];
    float pad0;
    float normal[3];
    float pad1;
};

struct DLongLeaf {
    uint32_t first, count;
};

// Everything the shade step needs about a triangle — geometry, vertex normals,
// texture coordinates, the barycentric precomputation and the material — in
// one 128-byte, 128-byte-aligned record addressed by the triangle's own index,
// so shading a hit costs one memory line.  (The reference reaches the same
// values through three per-triangle indices into side tables,
// /root/reference/src/shapes.rs:264-326 and src/lib.rs:114-122; traversal keeps
// its own compact 48-byte geometry array.)
constexpr uint32_t kTriHasBary = 0x80000000u;     // barycentric table row present
constexpr uint32_t kTriHasNormals = 0x40000000u;  // vertex normals present
constexpr uint32_t kTriHasUv = 0x20000000u;       // texture coordinates present
constexpr uint32_t kTriMaterialMask = 0x1FFFFFFFu;
struct alignas(128) DTriSurface {
    float a[3], e1[3], e2[3];      // words 0..8
    float n0[3], n1[3], n2[3];     // words 9..17
    float uv0[2], uv1[2], uv2[2];  // words 18..23
    float d00, d01, d11, den;      // words 24..27, /root/reference/src/obj.rs:142-153
    uint32_t material_flags;       // word 28: material index | kTriHas* bits
    uint32_t pad[3];
};
static_assert(sizeof(DTriSurface) == 128, "surface record is one 128-byte line");

// Textures are repacked at upload from 3 to 4 bytes per texel (R, G, B, unused), so that a
// texel is one aligned 32-bit load, and into tiles of 8 x 4 texels, one 128-byte memory line
// each (tiles row by row, texels row by row inside a tile; edge tiles padded): the 2 x 2
// texels of a bilinear sample then lie in 1.4 lines on average instead of 2, and a random
// gather costs by the line, not by the byte.  Each texture is contiguous and starts on a line.
constexpr uint32_t kTexTileW = 8, kTexTileH = 4, kTexTileTexels = kTexTileW * kTexTileH;
struct DMaterial {
    uint32_t kind;
    float param;
    uint32_t color_kind;
    float rgb[3];
    uint32_t tex_width, tex_height;
    uint64_t tex_first;  // index of the texture's first tile's first texel in DScene::texels (a multiple of 32)
};

// One shape family's acceleration structure on the device.
struct DFamily {
    const DNode* nodes;
    const DLongLeaf* long_leaves;
    const void* shapes;  // DSphere / DPlane / the three triangle arrays (DTriHalf x n, DTriHalf x n, float x n)
    uint32_t root_ref;
    uint32_t n_shapes;
    uint32_t max_stack;  // deepest stack the traversal can need (tree depth + 1)
};

struct DScene {
    DFamily family[3];
    const uint32_t* sphere_material;
    const uint32_t* plane_material;
    const DTriSurface* tri_surface;
    const DMaterial* materials;
    const uint32_t* texels;  // R | G<<8 | B<<16
    // A sphere or plane family whose whole BVH is one small leaf is not given a
    // traversal launch of its own: the kernel that creates a ray scans those few
    // shapes at once (same ascending order, same strict `<`).  0 = not inlined.
    uint32_t inline_count[2];
};

// Three packed floats: the vectors of a path record are stored without a
// fourth lane (every bounce streams all of them in and out once).
struct P3 {
    float x, y, z;
};
static_assert(sizeof(P3) == 12, "path vectors are 12 bytes apart");

// One bounce's worth of live paths.
struct DQueue {
    P3* origin;
    P3* direction;
    P3* throughput;
    uint2* rng;
    uint32_t* slot;
    float* hit_t;
    uint32_t* hit_id;
};

}  // namespace rtd
