// rt_render — command-line stand-in for the reference's binary
// (/root/reference/src/main.rs:5-8): read the file `scene` from the working
// directory, parse it, render into the memory-mapped `target/out.ppm`.
//
//   rt_render [scene-file [output.ppm [gpu-ordinal]]]
//
// `obj/<name>.obj` paths in the scene file are resolved against the working
// directory, as in the reference.  The output file is never truncated, so a
// later run with `continue(n)` in the scene file resumes from its pixels.
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "rt_host.h"
#include "rt_mi355.h"

static void progress(uint32_t samples_done, void*) {
    std::printf("\rSamples: %u", samples_done);  // lib.rs:238-246
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const std::string scene_path = argc > 1 ? argv[1] : "scene";
    const std::string out_path = argc > 2 ? argv[2] : "target/out.ppm";
    const int gpu = argc > 3 ? std::atoi(argv[3]) : 0;

    std::ifstream in(scene_path);
    if (!in) {
        std::fprintf(stderr, "cannot read %s\n", scene_path.c_str());
        return 1;
    }
    std::stringstream text;
    text << in.rdbuf();

    char err[1024] = {0};
    RthScene* scene = rth_parse_scene(text.str().c_str(), ".", err, sizeof err);
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

    RtContext* ctx = nullptr;
    int rc = rt_create(gpu, &ctx);
    if (rc != RT_OK) {
        std::fprintf(stderr, "no usable GPU (rt_create returned %d)\n", rc);
        return 1;
    }
    rc = rt_upload_scene(ctx, &desc);
    if (rc == RT_OK) rc = rt_render(ctx, &params, rth_image_pixels(image), progress, nullptr);
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
