// In-memory scene of the host-side ingest library: the flat arrays that
// RtSceneDesc points into, plus the render parameters read from the scene file.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_host.h"
#include "rt_mi355.h"

namespace rth {

// Material table with the reference's interning rule
// (/root/reference/src/config.rs:157-178): a material equal to an existing one
// reuses its index, otherwise it is appended.  Equality is the derived
// `PartialEq` of Material (/root/reference/src/material.rs:11-15,113-148):
// same kind and parameter, and either the same solid colour or textures of
// equal size and equal texel bytes.
struct MaterialTable {
    std::vector<RtMaterial> items;
    std::vector<uint8_t> texels;

    uint32_t intern_solid(uint32_t kind, float param, const float rgb[3]);
    uint32_t intern_texture(uint32_t kind, float param, uint32_t w, uint32_t h,
                            const uint8_t* rgb8);
};

struct Scene {
    RtRenderParams params{};
    std::vector<RtSphere> spheres;
    std::vector<RtPlane> planes;
    std::vector<RtTriangle> triangles;
    std::vector<float> vertex_normals;    // [n][3][3]
    std::vector<float> tex_coords;        // [n][3][2]
    std::vector<float> bary_precomputed;  // [n][4]
    MaterialTable materials;
    std::vector<RtNode> sphere_nodes, plane_nodes, triangle_nodes;
};

// /root/reference/src/obj.rs:12-192.  Appends triangles and side-table rows.
// Throws std::runtime_error where the reference panics.
void load_obj(const std::string& path, Scene& scene, std::vector<RtTriangle>& out);

// /root/reference/src/material.rs:150-189.  Throws std::runtime_error.
std::vector<uint8_t> load_ppm_p6(const std::string& path, uint32_t& width, uint32_t& height);

// /root/reference/src/config.rs:15-145.  Throws std::runtime_error.
// `triangle_builder` (may be null) replaces the host builder for the triangle family.
void parse_scene(const std::string& text, const std::string& base_dir, Scene& scene,
                 RthTriangleBvhFn triangle_builder = nullptr, void* builder_user = nullptr);

std::vector<RtNode> build_bvh(RtSphere* s, uint32_t n);
std::vector<RtNode> build_bvh(RtPlane* s, uint32_t n);
std::vector<RtNode> build_bvh(RtTriangle* s, uint32_t n);

// Number parsing with Rust's `str::parse` strictness (no surrounding blanks,
// whole token consumed).  Throw std::runtime_error on failure.
float parse_f32(const std::string& token);
uint32_t parse_u32(const std::string& token);

}  // namespace rth
