// Scene-file parser: the line-oriented `key(value)` format of
// /root/reference/src/config.rs:15-202.
//
// Grammar, as the reference implements it:
//   * one entry per line; the last character of the line (the closing
//     parenthesis) is dropped and the rest is cut at the first '(' into key and
//     value (config.rs:40);
//   * keys: continue, incremental, screen, camera, spheres, planes, triangles,
//     obj; anything else is an error (config.rs:41-108);
//   * lines are consumed only until screen, camera, spheres, planes and
//     triangles have all been seen — `obj(...)` counts as having seen
//     triangles — so trailing lines are never read (config.rs:32-37,93);
//   * a list value is "(item), (item), ..."; items are cut at "), (" and fields
//     at ", "; a vector is three numbers separated by single blanks
//     (config.rs:180-202, vec3.rs:541-547);
//   * material strings: lambertian | metal <fuzz> | glass <index> | light
//     (material.rs:121-138); scene-file materials always carry a solid colour
//     (config.rs:147-155) and share one interned table with OBJ materials;
//   * `incremental(n)` must be non-zero and divide the sample count
//     (config.rs:120-123);
//   * the three BVHs are built last and permute the shape lists
//     (config.rs:113-134).
#include <cstring>
#include <stdexcept>

#include <cstdlib>

#include "scene.h"
#include "text.h"
#include "vec3.h"

namespace rth {

static bool same_material_head(const RtMaterial& m, uint32_t kind, float param) {
    // derived PartialEq on MaterialKind: the variant, then its f32 field by `==`
    // (so a NaN parameter never equals itself).
    if (m.kind != kind) return false;
    if (kind == RT_MAT_METAL || kind == RT_MAT_GLASS) return m.param == param;
    return true;
}

uint32_t MaterialTable::intern_solid(uint32_t kind, float param, const float rgb[3]) {
    for (size_t i = 0; i < items.size(); ++i) {
        const RtMaterial& m = items[i];
        if (same_material_head(m, kind, param) && m.color_kind == RT_COLOR_SOLID &&
            m.rgb[0] == rgb[0] && m.rgb[1] == rgb[1] && m.rgb[2] == rgb[2])
            return static_cast<uint32_t>(i);
    }
    RtMaterial m{};
    m.kind = kind;
    m.param = param;
    m.color_kind = RT_COLOR_SOLID;
    m.rgb[0] = rgb[0];
    m.rgb[1] = rgb[1];
    m.rgb[2] = rgb[2];
    items.push_back(m);
    return static_cast<uint32_t>(items.size() - 1);
}

uint32_t MaterialTable::intern_texture(uint32_t kind, float param, uint32_t w, uint32_t h,
                                       const uint8_t* rgb8) {
    const size_t bytes = static_cast<size_t>(w) * h * 3;
    for (size_t i = 0; i < items.size(); ++i) {
        const RtMaterial& m = items[i];
        if (same_material_head(m, kind, param) && m.color_kind == RT_COLOR_TEXTURE &&
            m.tex_width == w && m.tex_height == h &&
            (bytes == 0 || std::memcmp(&texels[m.tex_offset], rgb8, bytes) == 0))
            return static_cast<uint32_t>(i);
    }
    RtMaterial m{};
    m.kind = kind;
    m.param = param;
    m.color_kind = RT_COLOR_TEXTURE;
    m.tex_width = w;
    m.tex_height = h;
    m.tex_offset = texels.size();
    texels.insert(texels.end(), rgb8, rgb8 + bytes);
    items.push_back(m);
    return static_cast<uint32_t>(items.size() - 1);
}

namespace {

V3 parse_v3(const std::string& s) {
    const std::vector<std::string> tok = split(s, " ");
    if (tok.size() < 3) throw std::runtime_error("expected three numbers in '" + s + "'");
    return v3(parse_f32(tok[0]), parse_f32(tok[1]), parse_f32(tok[2]));
}

// material.rs:121-138
void parse_material_kind(const std::string& s, uint32_t& kind, float& param) {
    const std::vector<std::string> tok = split_whitespace(s);
    if (tok.empty()) throw std::runtime_error("empty material");
    param = 0.0f;
    if (tok[0] == "lambertian") {
        kind = RT_MAT_LAMBERTIAN;
    } else if (tok[0] == "metal" || tok[0] == "glass") {
        if (tok.size() < 2) throw std::runtime_error("material '" + tok[0] + "' needs a parameter");
        kind = tok[0] == "metal" ? RT_MAT_METAL : RT_MAT_GLASS;
        param = parse_f32(tok[1]);
    } else if (tok[0] == "light") {
        kind = RT_MAT_LIGHT;
    } else {
        throw std::runtime_error("Unknown material: " + tok[0]);
    }
}

// config.rs:147-155: the next two fields are the material string and its colour.
uint32_t material_from_fields(const std::vector<std::string>& f, size_t at, MaterialTable& table) {
    if (f.size() != at + 2) throw std::runtime_error("wrong number of fields in a shape entry");
    uint32_t kind;
    float param;
    parse_material_kind(f[at], kind, param);
    const V3 c = parse_v3(f[at + 1]);
    const float rgb[3] = {c.x, c.y, c.z};
    return table.intern_solid(kind, param, rgb);
}

// config.rs:190-202
std::vector<std::vector<std::string>> items_of(const std::string& value) {
    std::vector<std::vector<std::string>> out;
    if (value.size() > 1)
        for (const std::string& item : split(value.substr(1, value.size() - 2), "), ("))
            out.push_back(split(item, ", "));
    return out;
}

}  // namespace

void parse_scene(const std::string& text, const std::string& base_dir, Scene& scene,
                 RthTriangleBvhFn triangle_builder, void* builder_user) {
    const std::vector<std::string> ls = lines(text);
    size_t next = 0;
    bool have_screen = false, have_camera = false, have_spheres = false, have_planes = false,
         have_triangles = false;
    RtRenderParams& p = scene.params;
    p = RtRenderParams{};
    p.rng_seed = RT_RNG_SEED;
    p.rng_mode = RT_RNG_SUBSTREAM;

    while (!(have_screen && have_camera && have_spheres && have_planes && have_triangles)) {
        if (next >= ls.size()) throw std::runtime_error("scene file ended before all entries were seen");
        const std::string& line = ls[next++];
        if (line.empty()) throw std::runtime_error("empty line in scene file");
        const std::string body = line.substr(0, line.size() - 1);
        const size_t open = body.find('(');
        if (open == std::string::npos) throw std::runtime_error("missing '(' in: " + line);
        const std::string key = body.substr(0, open), value = body.substr(open + 1);

        if (key == "continue") {
            p.has_continue = 1;
            p.continue_sampling = parse_u32(value);
        } else if (key == "incremental") {
            p.incremental = parse_u32(value);
            if (p.incremental == 0) throw std::runtime_error("incremental(0) is not allowed");
        } else if (key == "screen") {
            const std::vector<std::string> f = split(value, ", ");
            if (f.size() != 7) throw std::runtime_error("screen needs 7 fields");
            store(p.top_left, parse_v3(f[0]));
            store(p.top_edge, parse_v3(f[1]));
            store(p.left_edge, parse_v3(f[2]));
            p.width = parse_u32(f[3]);
            p.height = parse_u32(f[4]);
            p.samples_per_pixel = parse_u32(f[5]);
            p.max_bounces = parse_u32(f[6]);
            have_screen = true;
        } else if (key == "camera") {
            store(p.camera, parse_v3(value));
            have_camera = true;
        } else if (key == "spheres") {
            scene.spheres.clear();
            for (const auto& f : items_of(value)) {
                if (f.size() < 2) throw std::runtime_error("sphere needs 4 fields");
                RtSphere s;
                store(s.center, parse_v3(f[0]));
                s.radius = parse_f32(f[1]);
                s.material = material_from_fields(f, 2, scene.materials);
                scene.spheres.push_back(s);
            }
            have_spheres = true;
        } else if (key == "planes") {
            scene.planes.clear();
            for (const auto& f : items_of(value)) {
                if (f.size() < 2) throw std::runtime_error("plane needs 4 fields");
                RtPlane pl;
                store(pl.point, parse_v3(f[0]));
                store(pl.normal, normalize(parse_v3(f[1])));
                pl.material = material_from_fields(f, 2, scene.materials);
                scene.planes.push_back(pl);
            }
            have_planes = true;
        } else if (key == "triangles") {
            for (const auto& f : items_of(value)) {
                if (f.size() < 3) throw std::runtime_error("triangle needs 5 fields");
                const V3 a = parse_v3(f[0]), b = parse_v3(f[1]), c = parse_v3(f[2]);
                RtTriangle t;
                store(t.a, a);
                store(t.e1, sub(b, a));
                store(t.e2, sub(c, a));
                t.normals_idx = t.uv_idx = t.bary_idx = RT_NO_INDEX;
                t.material = material_from_fields(f, 3, scene.materials);
                scene.triangles.push_back(t);
            }
            have_triangles = true;
        } else if (key == "obj") {
            for (const auto& f : items_of(value)) {
                if (f.size() != 1) throw std::runtime_error("obj entry takes one name");
                load_obj(base_dir + "/obj/" + f[0] + ".obj", scene, scene.triangles);
            }
            have_triangles = true;
        } else {
            throw std::runtime_error("Unknown entry " + key + " with value " + value);
        }
    }

    if (p.incremental != 0 && p.samples_per_pixel % p.incremental != 0)
        throw std::runtime_error("incremental must divide the samples per pixel");

    scene.sphere_nodes = build_bvh(scene.spheres.data(), static_cast<uint32_t>(scene.spheres.size()));
    scene.plane_nodes = build_bvh(scene.planes.data(), static_cast<uint32_t>(scene.planes.size()));
    // The triangle family is where construction time goes; a caller may supply
    // another builder with the same contract (the GPU one) and this falls back to
    // the host builder when that one declines.
    const uint32_t n_tris = static_cast<uint32_t>(scene.triangles.size());
    bool built = false;
    if (triangle_builder) {
        uint32_t n_nodes = 0;
        RtNode* nodes = triangle_builder(builder_user, scene.triangles.data(), n_tris, &n_nodes);
        if (nodes) {
            scene.triangle_nodes.assign(nodes, nodes + n_nodes);
            std::free(nodes);
            built = true;
        }
    }
    if (!built) scene.triangle_nodes = build_bvh(scene.triangles.data(), n_tris);
}

}  // namespace rth
