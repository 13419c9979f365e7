// Wavefront OBJ / MTL loader following /root/reference/src/obj.rs:12-235,
// including the parts of its behaviour that are accidents of how it is written
// but decide material indices and triangle order:
//
//   * the file is cut at every occurrence of the text "usemtl "; the first line
//     of each piece is taken as a material name, and a piece whose name is
//     unknown (always true for the piece before the first `usemtl`) interns the
//     default grey Lambertian even if it holds no faces (obj.rs:66-81);
//   * a face is any line starting with 'f'; polygons are fan-triangulated
//     around their first vertex (obj.rs:84,112-117);
//   * indices are 1-based, negative ones count from the end (obj.rs:93-100);
//   * a triangle gets texture coordinates / vertex normals only if all three
//     corners have them; normals are normalised on load (obj.rs:119-141);
//   * MTL materials are always Lambertian; `map_Kd` (a P6 file next to the
//     OBJ) wins over `Kd`, which wins over grey 0.5 (obj.rs:219-226).
#include <array>
#include <cstdio>
#include <map>
#include <stdexcept>

#include "scene.h"
#include "text.h"
#include "vec3.h"

namespace rth {
namespace {

V3 parse_v3(const std::string& s) {
    // vec3.rs:541-547 — split on single blanks, first three tokens.
    const std::vector<std::string> tok = split(s, " ");
    if (tok.size() < 3) throw std::runtime_error("expected three numbers in '" + s + "'");
    return v3(parse_f32(tok[0]), parse_f32(tok[1]), parse_f32(tok[2]));
}

struct Corner {
    V3 position;
    bool has_uv = false, has_normal = false;
    float uv[2] = {0, 0};
    V3 normal{0, 0, 0};
};

// obj.rs:196-235
std::map<std::string, uint32_t> parse_materials(const std::string* mtl, const std::string& parent,
                                                MaterialTable& materials) {
    std::map<std::string, uint32_t> name_index;
    if (!mtl) return name_index;
    const std::vector<std::string> sections = split(*mtl, "newmtl ");
    for (size_t s = 1; s < sections.size(); ++s) {
        const std::vector<std::string> ls = lines(sections[s]);
        if (ls.empty()) throw std::runtime_error("newmtl without a name");
        const std::string& name = ls[0];
        const std::string* kd = nullptr;
        const std::string* map_kd = nullptr;
        for (size_t i = 1; i < ls.size(); ++i) {
            if (!kd && starts_with(ls[i], "Kd")) kd = &ls[i];
            if (!map_kd && starts_with(ls[i], "map_Kd")) map_kd = &ls[i];
        }
        uint32_t index;
        if (map_kd) {
            if (map_kd->size() < 7) throw std::runtime_error("map_Kd without a file name");
            uint32_t w = 0, h = 0;
            const std::vector<uint8_t> px = load_ppm_p6(parent + "/" + map_kd->substr(7), w, h);
            index = materials.intern_texture(RT_MAT_LAMBERTIAN, 0.0f, w, h, px.data());
        } else if (kd) {
            if (kd->size() < 3) throw std::runtime_error("Kd without a colour");
            const V3 c = parse_v3(kd->substr(3));
            const float rgb[3] = {c.x, c.y, c.z};
            index = materials.intern_solid(RT_MAT_LAMBERTIAN, 0.0f, rgb);
        } else {
            const float rgb[3] = {0.5f, 0.5f, 0.5f};
            index = materials.intern_solid(RT_MAT_LAMBERTIAN, 0.0f, rgb);
        }
        name_index[name] = index;
    }
    return name_index;
}

}  // namespace

void load_obj(const std::string& path, Scene& scene, std::vector<RtTriangle>& out) {
    const std::string text = read_text_file(path);
    const std::vector<std::string> all = lines(text);

    const size_t slash = path.rfind('/');
    if (slash == std::string::npos) throw std::runtime_error("obj path has no directory: " + path);
    const std::string parent = path.substr(0, slash + 1);

    std::string mtl_text;
    const std::string* mtl = nullptr;
    for (const std::string& l : all)
        if (starts_with(l, "mtllib")) {
            if (l.size() < 7) throw std::runtime_error("mtllib without a file name");
            mtl_text = read_text_file(parent + "/" + l.substr(7));
            mtl = &mtl_text;
            break;
        }
    const std::map<std::string, uint32_t> name_index = parse_materials(mtl, parent, scene.materials);

    std::vector<V3> positions, normals;
    std::vector<std::array<float, 2>> uvs;
    for (const std::string& l : all) {
        if (starts_with(l, "v ")) {
            positions.push_back(parse_v3(trim(l.substr(2))));
        } else if (starts_with(l, "vt")) {
            const std::vector<std::string> tok = split(trim(l.size() > 3 ? l.substr(3) : ""), " ");
            if (tok.size() < 2) throw std::runtime_error("vt needs two numbers: " + l);
            uvs.push_back({parse_f32(tok[0]), parse_f32(tok[1])});
            if (tok.size() > 2 && parse_f32(tok[2]) != 0.0f)
                std::fprintf(stderr, "Warning: only 2d texture coordinates are currently supported\n");
        } else if (starts_with(l, "vn")) {
            normals.push_back(parse_v3(trim(l.size() > 3 ? l.substr(3) : "")));
        }
    }

    auto resolve = [](const std::string& s, size_t len, size_t& index) -> bool {
        // `str::parse::<isize>().ok()?` then obj.rs:96-100.
        if (s.empty()) return false;
        size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
        if (i == s.size()) return false;
        long long v = 0;
        for (; i < s.size(); ++i) {
            if (s[i] < '0' || s[i] > '9') return false;
            v = v * 10 + (s[i] - '0');
            if (v > (1ll << 40)) return false;
        }
        if (s[0] == '-') {
            if (static_cast<size_t>(v) > len) throw std::runtime_error("negative index out of range");
            index = len - static_cast<size_t>(v);
        } else {
            if (v == 0) throw std::runtime_error("index 0 in a face");
            index = static_cast<size_t>(v) - 1;
        }
        return true;
    };

    const std::vector<std::string> pieces = split(text, "usemtl ");
    for (const std::string& piece : pieces) {
        const std::vector<std::string> ls = lines(piece);
        uint32_t material;
        auto it = ls.empty() ? name_index.end() : name_index.find(ls[0]);
        if (it != name_index.end()) {
            material = it->second;
        } else {
            const float grey[3] = {0.5f, 0.5f, 0.5f};
            material = scene.materials.intern_solid(RT_MAT_LAMBERTIAN, 0.0f, grey);
        }

        for (size_t li = 1; li < ls.size(); ++li) {
            const std::string& l = ls[li];
            if (l.empty() || l[0] != 'f') continue;
            std::vector<Corner> corners;
            for (const std::string& part : split_whitespace(l.substr(1))) {
                const std::vector<std::string> f = split(part, "/");
                Corner c;
                size_t idx;
                if (f.empty() || !resolve(f[0], positions.size(), idx))
                    throw std::runtime_error("face corner without a vertex index: " + l);
                if (idx >= positions.size()) throw std::runtime_error("vertex index out of range: " + l);
                c.position = positions[idx];
                if (f.size() > 1 && resolve(f[1], uvs.size(), idx)) {
                    if (idx >= uvs.size()) throw std::runtime_error("vt index out of range: " + l);
                    c.has_uv = true;
                    c.uv[0] = uvs[idx][0];
                    c.uv[1] = uvs[idx][1];
                }
                if (f.size() > 2 && resolve(f[2], normals.size(), idx)) {
                    if (idx >= normals.size()) throw std::runtime_error("vn index out of range: " + l);
                    c.has_normal = true;
                    c.normal = normals[idx];
                }
                corners.push_back(c);
            }
            if (corners.empty()) throw std::runtime_error("face without corners: " + l);

            const Corner& c1 = corners[0];
            for (size_t k = 1; k + 1 < corners.size(); ++k) {
                const Corner& c2 = corners[k];
                const Corner& c3 = corners[k + 1];
                RtTriangle t;
                t.normals_idx = t.uv_idx = t.bary_idx = RT_NO_INDEX;
                if (c1.has_uv && c2.has_uv && c3.has_uv) {
                    t.uv_idx = static_cast<uint32_t>(scene.tex_coords.size() / 6);
                    for (const Corner* c : {&c1, &c2, &c3}) {
                        scene.tex_coords.push_back(c->uv[0]);
                        scene.tex_coords.push_back(c->uv[1]);
                    }
                }
                if (c1.has_normal && c2.has_normal && c3.has_normal) {
                    t.normals_idx = static_cast<uint32_t>(scene.vertex_normals.size() / 9);
                    for (const Corner* c : {&c1, &c2, &c3}) {
                        const V3 n = normalize(c->normal);
                        scene.vertex_normals.push_back(n.x);
                        scene.vertex_normals.push_back(n.y);
                        scene.vertex_normals.push_back(n.z);
                    }
                }
                const V3 e1 = sub(c2.position, c1.position);
                const V3 e2 = sub(c3.position, c1.position);
                if (t.uv_idx != RT_NO_INDEX || t.normals_idx != RT_NO_INDEX) {
                    // obj.rs:142-153
                    const float d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2);
                    t.bary_idx = static_cast<uint32_t>(scene.bary_precomputed.size() / 4);
                    scene.bary_precomputed.push_back(d00);
                    scene.bary_precomputed.push_back(d01);
                    scene.bary_precomputed.push_back(d11);
                    scene.bary_precomputed.push_back(d00 * d11 - d01 * d01);
                }
                store(t.a, c1.position);
                store(t.e1, e1);
                store(t.e2, e2);
                t.material = material;
                out.push_back(t);
            }
        }
    }
}

}  // namespace rth
