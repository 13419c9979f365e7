// rt_render — command-line stand-in for the reference's binary
// (/root/reference/src/main.rs:5-8): read the file `scene` from the working
// directory, parse it, render into the memory-mapped `target/out.ppm`.
//
//   rt_render [scene-file [output.ppm [gpu-ordinal [tile-edge]]]]
//
// `obj/<name>.obj` paths in the scene file are resolved against the working
// directory, as in the reference.  The output file is never truncated, so a
// later run with `continue(n)` in the scene file resumes from its pixels.
//
// Images of any size: the frame is rendered pass by pass (one pass = the
// `incremental` samples, or all of them) and, within a pass, tile by tile, each
// tile going through the device and back into its rows of the mapping.  Only a
// tile's worth of paths is ever resident on the GPU; the image itself lives in
// the page cache, as in the reference (README.md:18).  tile-edge 0 (default)
// picks one tile for frames up to 2^24 pixels and 4096 x 4096 tiles beyond.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rt_host.h"
#include "rt_mi355.h"

static void progress(uint32_t samples_done, void*) {
    std::printf("\rSamples: %u", samples_done);  // lib.rs:238-246
    std::fflush(stdout);
}

// RthTriangleBvhFn over rt_build_bvh_triangles; NULL (decline) sends the parser to the host builder.
static RtNode* build_triangles_on_gpu(void* user, RtTriangle* shapes, uint32_t n, uint32_t* n_nodes) {
    RtNode* nodes = nullptr;
    if (rt_build_bvh_triangles(static_cast<RtContext*>(user), shapes, n, &nodes, n_nodes) != RT_OK) return nullptr;
    return nodes;
}

int main(int argc, char** argv) {
    const std::string scene_path = argc > 1 ? argv[1] : "scene";
    const std::string out_path = argc > 2 ? argv[2] : "target/out.ppm";
    const int gpu = argc > 3 ? std::atoi(argv[3]) : 0;
    uint32_t tile_edge = argc > 4 ? static_cast<uint32_t>(std::strtoul(argv[4], nullptr, 10)) : 0;

    std::ifstream in(scene_path);
    if (!in) {
        std::fprintf(stderr, "cannot read %s\n", scene_path.c_str());
        return 1;
    }
    std::stringstream text;
    text << in.rdbuf();

    RtContext* ctx = nullptr;
    int rc = rt_create(gpu, &ctx);
    if (rc != RT_OK) {
        std::fprintf(stderr, "no usable GPU (rt_create returned %d)\n", rc);
        return 1;
    }

    // scene ingest on the host, with the triangle BVH built on the GPU (same tree)
    char err[1024] = {0};
    RthScene* scene = rth_parse_scene_with(text.str().c_str(), ".", build_triangles_on_gpu, ctx, err, sizeof err);
    if (!scene) {
        std::fprintf(stderr, "scene error: %s\n", err);
        return 1;
    }
    RtSceneDesc desc;
    RtRenderParams params;
    rth_scene_desc(scene, &desc);
    rth_scene_params(scene, &params);

    const size_t slash = out_path.rfind('/');
    if (slash != std::string::npos) ::mkdir(out_path.substr(0, slash).c_str(), 0755);
    RthImage* image = rth_image_open(out_path.c_str(), params.width, params.height, err, sizeof err);
    if (!image) {
        std::fprintf(stderr, "image error: %s\n", err);
        return 1;
    }

    rc = rt_upload_scene(ctx, &desc);
    const uint64_t n_pixels = static_cast<uint64_t>(params.width) * params.height;
    if (tile_edge == 0 && n_pixels > (1ull << 24)) tile_edge = 4096;
    if (rc == RT_OK && (tile_edge == 0 || (tile_edge >= params.width && tile_edge >= params.height))) {
        rc = rt_render(ctx, &params, rth_image_pixels(image), progress, nullptr);
    } else if (rc == RT_OK) {
        // lib.rs:206-211,233-235: passes outermost, so the file is a whole image after each
        const uint32_t chunk = params.incremental ? params.incremental : params.samples_per_pixel;
        if (chunk == 0 || params.samples_per_pixel % chunk != 0) {
            std::fprintf(stderr, "incremental must divide the samples per pixel\n");
            return 1;
        }
        const uint32_t n_passes = params.samples_per_pixel / chunk;
        const bool running = params.incremental != 0 || params.has_continue != 0;
        const uint32_t first_pass = params.has_continue ? params.continue_sampling : 0;
        std::vector<uint8_t> tile(static_cast<size_t>(std::min(tile_edge, params.width)) * std::min(tile_edge, params.height) * 3);
        for (uint32_t pass = 0; pass < n_passes && rc == RT_OK; ++pass) {
            // one pass of one tile = a render of `chunk` samples that resumes at pass k
            RtRenderParams tp = params;
            tp.samples_per_pixel = chunk;
            tp.incremental = 0;
            tp.has_continue = running ? 1 : 0;
            tp.continue_sampling = pass + first_pass;
            for (uint32_t y = 0; y < params.height && rc == RT_OK; y += tile_edge)
                for (uint32_t x = 0; x < params.width && rc == RT_OK; x += tile_edge) {
                    tp.tile_x = x;
                    tp.tile_y = y;
                    tp.tile_w = std::min(tile_edge, params.width - x);
                    tp.tile_h = std::min(tile_edge, params.height - y);
                    if (running) rth_image_read_tile(image, x, y, tp.tile_w, tp.tile_h, tile.data());
                    rc = rt_render(ctx, &tp, tile.data(), nullptr, nullptr);
                    if (rc == RT_OK) rth_image_write_tile(image, x, y, tp.tile_w, tp.tile_h, tile.data());
                }
            if (rc == RT_OK) progress(first_pass + (pass + 1) * chunk, nullptr);  // lib.rs:238-246
        }
    }
    if (rc != RT_OK) std::fprintf(stderr, "\nrender failed (%d): %s\n", rc, rt_last_error(ctx));
    else std::printf("\n");

    RtStats st;
    if (rc == RT_OK && rt_get_stats(ctx, &st) == RT_OK && st.render_ms > 0)
        std::fprintf(stderr, "%llu rays in %.1f ms: %.1f Mrays/s\n", (unsigned long long)st.segments, st.render_ms,
                     st.segments / st.render_ms / 1e3);
    rt_destroy(ctx);
    rth_image_flush(image);
    rth_image_close(image);
    rth_free_scene(scene);
    return rc == RT_OK ? 0 : 1;
}
