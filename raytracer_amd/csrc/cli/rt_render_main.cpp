// rt_render — command-line stand-in for the reference's binary
// (/root/reference/src/main.rs:5-8): read the file `scene` from the working
// directory, parse it, render into the memory-mapped `target/out.ppm`.
//
//   rt_render [scene-file [output.ppm [gpu-ordinals [tile-edge [accumulator-file]]]]]
//
// gpu-ordinals is one ordinal or a comma-separated list ("0,1,2,3"): with a list
// the GPUs share every pass's samples (rt_group_render) and the image is the one
// a single GPU stores, byte for byte.  The accumulator-file mode runs on the
// first GPU of the list.
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
// picks one tile for frames up to 2^24 pixels and 8192 x 8192 tiles beyond (a pass of
// few samples renders 12 % faster than in 4096 x 4096 tiles: larger batches of paths).
//
// RT_RENDER_MAX_LEAF=n in the environment renders with the refined tree of rt_host.h's
// rth_refine_bvh_triangles (opt-in; not the reference's tree).
//
// With an accumulator file the run is progressive at full precision instead of
// averaging the stored bytes (lib.rs:286-292): the file keeps the per-pixel f32
// fold of all samples so far (32-byte header: "RTACC1", width, height, samples,
// pass-in-progress mark, generator seed; then height x width x 3 f32), every run
// adds the scene's samples per pixel in passes of `incremental` (or all at once)
// and rewrites the image from the total, so after n samples the image is the
// non-incremental render of n samples.  `continue` in the scene file is ignored
// in this mode: the file knows where it stands.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// The accumulator file, mapped (layout in the header comment).
struct Accumulator {
    struct Header {
        char magic[8];
        uint32_t width, height, samples, in_progress;
        uint64_t seed;
    };
    static_assert(sizeof(Header) == 32, "header layout");
    void* map = nullptr;
    size_t size = 0;
    Header* header() { return static_cast<Header*>(map); }
    float* sums() { return reinterpret_cast<float*>(static_cast<char*>(map) + sizeof(Header)); }
    void sync() { ::msync(map, size, MS_SYNC); }

    bool open(const std::string& path, uint32_t w, uint32_t h, uint64_t seed) {
        size = sizeof(Header) + static_cast<size_t>(w) * h * 12;
        const int fd = ::open(path.c_str(), O_RDWR | O_CREAT, 0644);
        if (fd < 0) return false;
        struct stat st;
        Header on_disk{};
        bool fresh = ::fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) != size ||
                     ::pread(fd, &on_disk, sizeof on_disk, 0) != static_cast<ssize_t>(sizeof on_disk);
        fresh = fresh || std::memcmp(on_disk.magic, "RTACC1\0\0", 8) != 0 || on_disk.width != w || on_disk.height != h ||
                on_disk.seed != seed;
        if (fresh && (::ftruncate(fd, 0) != 0 || ::ftruncate(fd, static_cast<off_t>(size)) != 0)) {  // missing, or of another image
            ::close(fd);
            return false;
        }
        map = ::mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (map == MAP_FAILED) return false;
        if (fresh) {
            std::memcpy(header()->magic, "RTACC1\0\0", 8);
            header()->width = w;
            header()->height = h;
            header()->samples = 0;
            header()->in_progress = 0;
            header()->seed = seed;
        }
        return true;
    }
    void close() {
        if (map && map != MAP_FAILED) {
            sync();
            ::munmap(map, size);
        }
        map = nullptr;
    }
};

// A tile's pixel bytes: rt_host_alloc'd if possible, heap memory otherwise.
struct PixelBuffer {
    explicit PixelBuffer(size_t bytes) : pinned(static_cast<uint8_t*>(rt_host_alloc(bytes))) {
        if (!pinned) heap.resize(bytes);
    }
    ~PixelBuffer() { rt_host_free(pinned); }
    PixelBuffer(const PixelBuffer&) = delete;
    PixelBuffer& operator=(const PixelBuffer&) = delete;
    uint8_t* data() { return pinned ? pinned : heap.data(); }
    uint8_t* pinned;
    std::vector<uint8_t> heap;
};

int main(int argc, char** argv) {
    const std::string scene_path = argc > 1 ? argv[1] : "scene";
    const std::string out_path = argc > 2 ? argv[2] : "target/out.ppm";
    std::vector<int> gpus;
    for (const char* c = argc > 3 ? argv[3] : "0"; *c;) {
        char* end = nullptr;
        gpus.push_back(static_cast<int>(std::strtol(c, &end, 10)));
        if (end == c) {
            std::fprintf(stderr, "bad GPU list %s\n", argv[3]);
            return 1;
        }
        c = *end == ',' ? end + 1 : end;
    }
    uint32_t tile_edge = argc > 4 ? static_cast<uint32_t>(std::strtoul(argv[4], nullptr, 10)) : 0;
    const std::string acc_path = argc > 5 ? argv[5] : "";

    std::ifstream in(scene_path);
    if (!in) {
        std::fprintf(stderr, "cannot read %s\n", scene_path.c_str());
        return 1;
    }
    std::stringstream text;
    text << in.rdbuf();

    RtGroup* group = nullptr;
    int rc = rt_group_create(gpus.data(), static_cast<uint32_t>(gpus.size()), &group);
    if (rc != RT_OK) {
        std::fprintf(stderr, "no usable GPU (%d): %s\n", rc, rt_group_last_error(nullptr));
        return 1;
    }
    RtContext* ctx = rt_group_member(group, 0);

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

    // Opt-in (RT_RENDER_MAX_LEAF=n in the environment): not the reference's tree but its leaves split on down to n
    // triangles (rth_refine_bvh_triangles) — faster on scenes whose leaves are long; the image can differ where two
    // triangles are hit at exactly the same distance.
    std::vector<RtTriangle> refined_triangles;
    std::vector<uint32_t> refined_order;
    RtNode* refined_nodes = nullptr;
    if (const char* leaf = std::getenv("RT_RENDER_MAX_LEAF")) {
        const uint32_t max_leaf = static_cast<uint32_t>(std::strtoul(leaf, nullptr, 10));
        if (max_leaf > 0 && desc.n_triangles > 0) {
            refined_triangles.assign(desc.triangles, desc.triangles + desc.n_triangles);
            refined_order.resize(desc.n_triangles);
            uint32_t n_nodes = 0;
            refined_nodes = rth_refine_bvh_triangles(desc.triangle_nodes, desc.n_triangle_nodes, refined_triangles.data(),
                                                     desc.n_triangles, max_leaf, refined_order.data(), &n_nodes);
            if (!refined_nodes) {
                std::fprintf(stderr, "tree refinement failed\n");
                return 1;
            }
            std::fprintf(stderr, "refined tree: %u -> %u nodes, leaves of at most %u triangles\n", desc.n_triangle_nodes, n_nodes, max_leaf);
            desc.triangles = refined_triangles.data();
            desc.triangle_nodes = refined_nodes;
            desc.n_triangle_nodes = n_nodes;
        }
    }

    const size_t slash = out_path.rfind('/');
    if (slash != std::string::npos) ::mkdir(out_path.substr(0, slash).c_str(), 0755);
    RthImage* image = rth_image_open(out_path.c_str(), params.width, params.height, err, sizeof err);
    if (!image) {
        std::fprintf(stderr, "image error: %s\n", err);
        return 1;
    }

    rc = rt_group_upload_scene(group, &desc);
    const uint64_t n_pixels = static_cast<uint64_t>(params.width) * params.height;
    if (tile_edge == 0 && n_pixels > (1ull << 24)) tile_edge = 8192;
    if (rc == RT_OK && !acc_path.empty()) {
        Accumulator acc;
        if (!acc.open(acc_path, params.width, params.height, params.rng_seed)) {
            std::fprintf(stderr, "cannot open accumulator file %s\n", acc_path.c_str());
            return 1;
        }
        if (acc.header()->in_progress) {
            std::fprintf(stderr, "%s was cut off in the middle of a pass; delete it to start over\n", acc_path.c_str());
            return 1;
        }
        const uint32_t chunk = params.incremental ? params.incremental : params.samples_per_pixel;
        if (chunk == 0 || params.samples_per_pixel % chunk != 0) {
            std::fprintf(stderr, "incremental must divide the samples per pixel\n");
            return 1;
        }
        const uint32_t edge_x = tile_edge ? std::min(tile_edge, params.width) : params.width;
        const uint32_t edge_y = tile_edge ? std::min(tile_edge, params.height) : params.height;
        std::vector<float> sums(static_cast<size_t>(edge_x) * edge_y * 3);
        std::vector<uint8_t> tile(static_cast<size_t>(edge_x) * edge_y * 3);
        for (uint32_t pass = 0; pass < params.samples_per_pixel / chunk && rc == RT_OK; ++pass) {
            const uint32_t done = acc.header()->samples;
            acc.header()->in_progress = 1;
            acc.sync();
            RtRenderParams tp = params;
            tp.samples_per_pixel = chunk;
            tp.incremental = tp.has_continue = tp.continue_sampling = 0;
            for (uint32_t y = 0; y < params.height && rc == RT_OK; y += edge_y)
                for (uint32_t x = 0; x < params.width && rc == RT_OK; x += edge_x) {
                    tp.tile_x = x;
                    tp.tile_y = y;
                    tp.tile_w = std::min(edge_x, params.width - x);
                    tp.tile_h = std::min(edge_y, params.height - y);
                    const size_t row = static_cast<size_t>(tp.tile_w) * 3;
                    for (uint32_t j = 0; j < tp.tile_h; ++j)
                        std::memcpy(&sums[j * row], acc.sums() + (static_cast<size_t>(y + j) * params.width + x) * 3, row * sizeof(float));
                    rc = rt_render_sums_onto(ctx, &tp, done, chunk, sums.data(), 0);
                    if (rc != RT_OK) break;
                    for (uint32_t j = 0; j < tp.tile_h; ++j)
                        std::memcpy(acc.sums() + (static_cast<size_t>(y + j) * params.width + x) * 3, &sums[j * row], row * sizeof(float));
                    rc = rt_resolve(ctx, &tp, sums.data(), 0, done + chunk, 0, 0, tile.data());
                    if (rc == RT_OK) rth_image_write_tile(image, x, y, tp.tile_w, tp.tile_h, tile.data());
                }
            if (rc == RT_OK) {
                acc.sync();  // the sums before the count that vouches for them
                acc.header()->samples = done + chunk;
                acc.header()->in_progress = 0;
                acc.sync();
                rth_image_flush(image);
                progress(done + chunk, nullptr);
            }
        }
        acc.close();
    } else if (rc == RT_OK && (tile_edge == 0 || (tile_edge >= params.width && tile_edge >= params.height))) {
        rc = rt_group_render(group, &params, rth_image_pixels(image), progress, nullptr);
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
        // the tile's bytes go through page-locked memory when it can be had: copies to and from
        // the GPU at the full rate of the host link (ordinary memory otherwise)
        const size_t tile_bytes = static_cast<size_t>(std::min(tile_edge, params.width)) * std::min(tile_edge, params.height) * 3;
        PixelBuffer tile(tile_bytes);
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
                    rc = rt_group_render(group, &tp, tile.data(), nullptr, nullptr);
                    if (rc == RT_OK) rth_image_write_tile(image, x, y, tp.tile_w, tp.tile_h, tile.data());
                }
            if (rc == RT_OK) progress(first_pass + (pass + 1) * chunk, nullptr);  // lib.rs:238-246
        }
    }
    if (rc != RT_OK) {
        const char* why = rt_group_last_error(group);
        std::fprintf(stderr, "\nrender failed (%d): %s\n", rc, why[0] ? why : rt_last_error(ctx));
    }
    else std::printf("\n");

    RtStats st;
    if (rc == RT_OK && rt_group_get_stats(group, &st) == RT_OK && st.render_ms > 0)
        std::fprintf(stderr, "%llu rays in %.1f ms: %.1f Mrays/s\n", (unsigned long long)st.segments, st.render_ms,
                     st.segments / st.render_ms / 1e3);
    rt_group_destroy(group);
    rth_image_flush(image);
    rth_image_close(image);
    rth_free(refined_nodes);
    rth_free_scene(scene);
    return rc == RT_OK ? 0 : 1;
}
