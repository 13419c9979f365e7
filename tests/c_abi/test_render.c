/* The C ABI driven the way a compiled host drives it — the steps of the
 * reference's `main` (/root/reference/src/main.rs:5-8: read the scene file,
 * `config::parse`, `Scene::render` into the memory-mapped target/out.ppm) in
 * plain C against include/rt_host.h and include/rt_mi355.h — with the stored
 * bytes compared against the CPU oracle's `oracle_render` (test infrastructure,
 * linked into this test program only).
 *
 *   test_render <scene-file> <base-dir> <out.ppm>
 *
 * Exit status 0 when every byte of the image equals the oracle's, for the scene
 * as given and again for a second run that resumes from the file with
 * `continue`. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"
#include "rt_mi355.h"

typedef struct OracleCounters {
    uint64_t paths, segments, hits, branch_visits, prim_tests[3];
} OracleCounters;
int oracle_render(const RtSceneDesc* scene, const RtRenderParams* p, uint8_t* rgb8, float* mean_out, int n_threads,
                  OracleCounters* counters);

static int passes_seen = 0;
static uint32_t last_samples = 0;
static void on_pass(uint32_t samples_done, void* user) {
    (void)user;
    ++passes_seen;
    last_samples = samples_done;
}

static char* read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

int main(int argc, char** argv) {
    CHECK(argc == 4, "usage: %s <scene-file> <base-dir> <out.ppm>", argv[0]);
    char err[512] = "";
    char* text = read_file(argv[1]);
    CHECK(text, "cannot read %s", argv[1]);
    RthScene* scene = rth_parse_scene(text, argv[2], err, sizeof err);
    CHECK(scene, "parse: %s", err);
    RtSceneDesc desc;
    RtRenderParams params;
    rth_scene_desc(scene, &desc);
    rth_scene_params(scene, &params);

    CHECK(rt_abi_version() == RT_ABI_VERSION, "ABI version mismatch");
    RtContext* ctx = NULL;
    int rc = rt_create(0, &ctx);
    CHECK(rc == RT_OK, "rt_create: %d %s", rc, rt_last_error(NULL));
    rc = rt_upload_scene(ctx, &desc);
    CHECK(rc == RT_OK, "rt_upload_scene: %s", rt_last_error(ctx));

    remove(argv[3]);
    RthImage* image = rth_image_open(argv[3], params.width, params.height, err, sizeof err);
    CHECK(image, "image: %s", err);
    uint8_t* pixels = rth_image_pixels(image);
    const size_t n_bytes = (size_t)params.width * params.height * 3;
    uint8_t* want = (uint8_t*)calloc(n_bytes, 1);

    /* run 1: the scene as written */
    rc = rt_render(ctx, &params, pixels, on_pass, NULL);
    CHECK(rc == RT_OK, "rt_render: %s", rt_last_error(ctx));
    const uint32_t chunk = params.incremental ? params.incremental : params.samples_per_pixel;
    CHECK(passes_seen == (int)(params.samples_per_pixel / chunk) && last_samples == params.samples_per_pixel,
          "progress callback: %d passes, %u samples", passes_seen, last_samples);
    CHECK(oracle_render(&desc, &params, want, NULL, 4, NULL) == 0, "oracle_render failed");
    size_t differing = 0;
    for (size_t i = 0; i < n_bytes; ++i) differing += pixels[i] != want[i];
    CHECK(differing == 0, "run 1: %zu of %zu bytes differ from the oracle", differing, n_bytes);

    /* run 2: resume from the file (continue = number of passes done so far), as the reference does
     * when its scene file says continue(N) (/root/reference/src/lib.rs:233-234,286-292) */
    rth_image_flush(image);
    rth_image_close(image);
    image = rth_image_open(argv[3], params.width, params.height, err, sizeof err);
    CHECK(image, "image (reopen): %s", err);
    pixels = rth_image_pixels(image);
    for (size_t i = 0; i < n_bytes; ++i) differing += pixels[i] != want[i];
    CHECK(differing == 0, "the reopened file does not hold the image");
    RtRenderParams more = params;
    more.has_continue = 1;
    more.continue_sampling = params.samples_per_pixel / chunk;
    rc = rt_render(ctx, &more, pixels, NULL, NULL);
    CHECK(rc == RT_OK, "rt_render (continue): %s", rt_last_error(ctx));
    CHECK(oracle_render(&desc, &more, want, NULL, 4, NULL) == 0, "oracle_render (continue) failed");
    for (size_t i = 0; i < n_bytes; ++i) differing += pixels[i] != want[i];
    CHECK(differing == 0, "run 2 (continue): %zu bytes differ from the oracle", differing);

    /* error reporting across the boundary: a bad parameter block is refused with a message */
    RtRenderParams bad = params;
    bad.tile_w = params.width + 1;
    bad.tile_h = 1;
    CHECK(rt_render(ctx, &bad, pixels, NULL, NULL) == RT_ERR_INVALID && strlen(rt_last_error(ctx)) > 0, "bad tile not refused");

    RtStats stats;
    CHECK(rt_get_stats(ctx, &stats) == RT_OK && stats.paths > 0 && stats.segments >= stats.paths, "statistics");
    printf("ok: %ux%u, %u spp, %u passes + continue: all %zu bytes equal the oracle's; %llu rays\n", params.width, params.height,
           params.samples_per_pixel, (unsigned)passes_seen, n_bytes, (unsigned long long)stats.segments);
    rth_image_close(image);
    rt_destroy(ctx);
    rth_free_scene(scene);
    free(want);
    free(text);
    return 0;
}
