// Scalar float3 helpers for the host-side ingest code.
//
// Every function spells out the operation order of the reference's generic
// vector type (/root/reference/src/vec3.rs) so that values computed here are
// bit-identical to what the Rust crate computes: no fused multiply-add (build
// with -ffp-contract=off), true division, NaN-ignoring min/max.
#pragma once
#include <cmath>

namespace rth {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }
inline void store(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

inline V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 mul(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 div(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }

// vec3.rs:224-233 — element-wise product, then a left fold of `+`.
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// vec3.rs:303-321 — yzx(a)*zxy(b) - zxy(a)*yzx(b).
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// vec3.rs:285-291 — every component divided by sqrt(length_squared).
inline V3 normalize(V3 a) {
    const float len = std::sqrt(dot(a, a));
    return V3{a.x / len, a.y / len, a.z / len};
}

// vec3.rs:268-281 via MinMax -> f32::min / f32::max (a NaN operand is ignored).
inline V3 vmin(V3 a, V3 b) { return V3{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return V3{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

}  // namespace rth
