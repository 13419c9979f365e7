// C entry points of the host-side ingest library (see include/rt_host.h).
#include <cstdio>
#include <exception>
#include <new>

#include "rt_host.h"
#include "scene.h"

struct RthScene {
    rth::Scene scene;
};

extern "C" {

RthScene* rth_parse_scene(const char* text, const char* base_dir, char* err, size_t err_len) {
    return rth_parse_scene_with(text, base_dir, nullptr, nullptr, err, err_len);
}

RthScene* rth_parse_scene_with(const char* text, const char* base_dir, RthTriangleBvhFn build_triangles, void* user,
                               char* err, size_t err_len) {
    RthScene* s = new (std::nothrow) RthScene;
    if (!s) {
        if (err && err_len) std::snprintf(err, err_len, "out of memory");
        return nullptr;
    }
    try {
        rth::parse_scene(text ? text : "", base_dir ? base_dir : ".", s->scene, build_triangles, user);
        return s;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        delete s;
        return nullptr;
    }
}

void rth_free_scene(RthScene* scene) { delete scene; }

void rth_scene_desc(const RthScene* scene, RtSceneDesc* out) {
    const rth::Scene& s = scene->scene;
    *out = RtSceneDesc{};
    out->sphere_nodes = s.sphere_nodes.data();
    out->plane_nodes = s.plane_nodes.data();
    out->triangle_nodes = s.triangle_nodes.data();
    out->spheres = s.spheres.data();
    out->planes = s.planes.data();
    out->triangles = s.triangles.data();
    out->vertex_normals = s.vertex_normals.data();
    out->tex_coords = s.tex_coords.data();
    out->bary_precomputed = s.bary_precomputed.data();
    out->materials = s.materials.items.data();
    out->texels = s.materials.texels.data();
    out->n_texel_bytes = s.materials.texels.size();
    out->n_sphere_nodes = static_cast<uint32_t>(s.sphere_nodes.size());
    out->n_plane_nodes = static_cast<uint32_t>(s.plane_nodes.size());
    out->n_triangle_nodes = static_cast<uint32_t>(s.triangle_nodes.size());
    out->n_spheres = static_cast<uint32_t>(s.spheres.size());
    out->n_planes = static_cast<uint32_t>(s.planes.size());
    out->n_triangles = static_cast<uint32_t>(s.triangles.size());
    out->n_vertex_normals = static_cast<uint32_t>(s.vertex_normals.size() / 9);
    out->n_tex_coords = static_cast<uint32_t>(s.tex_coords.size() / 6);
    out->n_bary = static_cast<uint32_t>(s.bary_precomputed.size() / 4);
    out->n_materials = static_cast<uint32_t>(s.materials.items.size());
}

void rth_scene_params(const RthScene* scene, RtRenderParams* out) { *out = scene->scene.params; }
}
