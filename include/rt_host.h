/*
 * rt_host.h — C ABI of the host-side ingest library (librt_host.so).
 *
 * In the reference the host work is done by the Rust crate itself: scene-file
 * parsing (/root/reference/src/config.rs), OBJ/MTL loading (src/obj.rs), the
 * P6 texture loader (src/material.rs:150-189), the binned-SAH BVH builder
 * (src/bvh.rs:46-224) and the mmap'd output image (src/mmap.rs, src/lib.rs:
 * 64-89).  A Rust host keeps all of that and only needs rt_mi355.h.  This
 * library is the same work written in C++ for hosts without that crate (the
 * CLI, the Python bindings and the tests of this repository); it produces
 * exactly the RtSceneDesc / RtRenderParams a Rust host would hand over.
 * No GPU code, no dependency on HIP.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RthScene RthScene;

/* `config::parse`, /root/reference/src/config.rs:15-145.  `text` is the scene
 * file's content; `base_dir` is the directory `obj/...` paths are resolved
 * against (the reference resolves them against the process's working
 * directory, /root/reference/src/config.rs:97).  On failure returns NULL and
 * copies a message into `err` (the reference panics instead). */
RthScene* rth_parse_scene(const char* text, const char* base_dir, char* err, size_t err_len);
void rth_free_scene(RthScene* scene);

/* The same, with the triangle family's `BvhNode::new`
 * (/root/reference/src/config.rs:132-134) handed to `build_triangles`: a function
 * with the contract of rth_build_bvh_triangles (permute `shapes` in place, return
 * a malloc'd node array and its length) that may return NULL to decline, in
 * which case the host builder is used.  This is how rt_build_bvh_triangles of
 * rt_mi355.h (construction on the GPU) is plugged in without this library
 * depending on HIP. */
typedef RtNode* (*RthTriangleBvhFn)(void* user, RtTriangle* shapes, uint32_t n, uint32_t* n_nodes);
RthScene* rth_parse_scene_with(const char* text, const char* base_dir, RthTriangleBvhFn build_triangles,
                               void* user, char* err, size_t err_len);

/* Views into the parsed scene; valid until rth_free_scene. */
void rth_scene_desc(const RthScene* scene, RtSceneDesc* out);
void rth_scene_params(const RthScene* scene, RtRenderParams* out);

/* `BvhNode::new`, /root/reference/src/bvh.rs:48-68, for each shape family.
 * Permutes `shapes` in place exactly as the reference does and returns a
 * malloc'd node array (free with rth_free) and its length. */
RtNode* rth_build_bvh_spheres(RtSphere* shapes, uint32_t n, uint32_t* n_nodes);
RtNode* rth_build_bvh_planes(RtPlane* shapes, uint32_t n, uint32_t* n_nodes);
RtNode* rth_build_bvh_triangles(RtTriangle* shapes, uint32_t n, uint32_t* n_nodes);
void rth_free(void* p);

/* Opt-in, NON-PARITY faster tree (no counterpart in the reference): keeps the node array
 * of `BvhNode::new` as it is and goes on splitting every leaf that holds more than
 * `max_leaf` triangles, i.e. without the stop rule `15 + cost >= 2 * num` of
 * /root/reference/src/bvh.rs:169, which leaves clusters of overlapping shapes as leaves of
 * dozens to hundreds of triangles.  `triangles` is permuted in place (inside each such
 * leaf's range only) and `order[i]` receives the position, before the call, of the
 * triangle now at i; the returned node array (malloc'd, free with rth_free; the first
 * `n_nodes` entries are the reference's nodes, new ones follow) describes the same
 * triangles.  Intersection arithmetic is untouched, so nearest-hit distances are the
 * reference tree's numbers except where a ray grazes the box of a new, smaller leaf (the
 * slab test of /root/reference/src/bvh.rs:25-43 is not conservative; about one primary ray
 * in four million on the test frames); which of several triangles at exactly the same
 * distance is reported may differ from the reference's tree, and primitive ids are
 * positions after this call (map them back through `order`).  Returns NULL on bad
 * arguments or lack of memory. */
RtNode* rth_refine_bvh_triangles(const RtNode* nodes, uint32_t n_nodes, RtTriangle* triangles, uint32_t n_triangles,
                                 uint32_t max_leaf, uint32_t* order, uint32_t* n_nodes_out);

/* `ColorKind::texture_from_ppm_p6`, /root/reference/src/material.rs:150-189.
 * Returns malloc'd packed RGB8 texels (free with rth_free) or NULL. */
uint8_t* rth_load_ppm_p6(const char* path, uint32_t* width, uint32_t* height, char* err,
                         size_t err_len);

/* `Image::new` + `MmapFile::new`, /root/reference/src/lib.rs:70-83 and
 * /root/reference/src/mmap.rs:41-77: open (create, never truncate) `path`,
 * size it to header + width*height*3, map it shared, write the P6 header.
 * Existing pixel bytes are preserved — that is the reference's checkpoint. */
typedef struct RthImage RthImage;
RthImage* rth_image_open(const char* path, uint32_t width, uint32_t height, char* err,
                         size_t err_len);
uint8_t* rth_image_pixels(RthImage* image); /* width*height*3 bytes after the header */
size_t rth_image_header_len(const RthImage* image);
/* Copy the rows of tile (x, y, w, h) of the mapped image into / out of a packed
 * w*h*3 buffer: what a renderer working tile by tile reads for the running
 * average of /root/reference/src/lib.rs:286-292 and writes back afterwards (the
 * reference works on per-chunk slices of the mapping, :201-204,248-295).
 * Return 0, or -1 when the tile is not inside the image. */
int rth_image_read_tile(const RthImage* image, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                        uint8_t* tile);
int rth_image_write_tile(RthImage* image, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                         const uint8_t* tile);
void rth_image_flush(RthImage* image);
void rth_image_close(RthImage* image);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
