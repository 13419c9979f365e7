// Binary PPM ("P6") texture loader with the reference's exact header grammar,
// /root/reference/src/material.rs:150-189: the bytes "P6", one separator byte,
// the width up to the next ' ', the height up to the next '\n', the literal
// "255", one separator byte, then packed RGB8 texels.  Comments and other
// white-space layouts that the Netpbm format allows are not accepted, just as
// the reference does not accept them.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#include "rt_host.h"
#include "scene.h"

namespace rth {

static std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
    std::fclose(f);
    return data;
}

std::vector<uint8_t> load_ppm_p6(const std::string& path, uint32_t& width, uint32_t& height) {
    const std::vector<uint8_t> c = read_file(path);
    if (c.size() < 2 || c[0] != 'P' || c[1] != '6') throw std::runtime_error(path + ": not a P6 file");

    size_t base = 3;
    uint32_t dims[2];
    const char stop[2] = {' ', '\n'};
    for (int d = 0; d < 2; ++d) {
        size_t len = 0;
        while (base + len < c.size() && c[base + len] != static_cast<uint8_t>(stop[d])) ++len;
        if (base > c.size()) throw std::runtime_error(path + ": truncated header");
        dims[d] = parse_u32(std::string(c.begin() + base, c.begin() + base + len));
        base += len + 1;
    }
    if (base + 3 > c.size() || std::memcmp(&c[base], "255", 3) != 0)
        throw std::runtime_error(path + ": maximum value must be 255");

    const size_t start = base + 4;
    const size_t n_pixels = start <= c.size() ? (c.size() - start) / 3 : 0;
    if (n_pixels != static_cast<size_t>(dims[0]) * dims[1])
        throw std::runtime_error(path + ": pixel count does not match the header");
    width = dims[0];
    height = dims[1];
    return std::vector<uint8_t>(c.begin() + start, c.begin() + start + n_pixels * 3);
}

}  // namespace rth

extern "C" uint8_t* rth_load_ppm_p6(const char* path, uint32_t* width, uint32_t* height, char* err,
                                    size_t err_len) {
    try {
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> px = rth::load_ppm_p6(path, w, h);
        uint8_t* out = static_cast<uint8_t*>(std::malloc(px.size() ? px.size() : 1));
        if (!out) throw std::runtime_error("out of memory");
        std::memcpy(out, px.data(), px.size());
        if (width) *width = w;
        if (height) *height = h;
        return out;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        return nullptr;
    }
}
