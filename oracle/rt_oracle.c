/*
 * rt_oracle.c — CPU restatement of the reference's render loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the GPU library is
 * compared against (tests/, __graft_entry__.smoke(), and the `cpu_baseline`
 * leg of bench.py).  Nothing in the product path links, loads or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (a nightly-only Rust crate)
 * cannot be built here, ships no tests, no golden vectors and no recorded
 * outputs, so this restatement cannot be checked against the real program.
 * It is pinned only by (a) known answers derived by hand from the reference's
 * formulas (generator outputs, analytic ray/shape cases — tests/golden/), and
 * (b) being a function-by-function transcription of the cited lines.
 *
 * Every function below follows one piece of the reference, cited as
 * file:line under /root/reference/.  Arithmetic is IEEE binary32 with no
 * fused multiply-add (compile with -ffp-contract=off, no -ffast-math), true
 * division and correctly rounded sqrt, as rustc emits it.
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rt_mi355.h"

/* ------------------------------------------------------------------ vectors */
/* src/vec3.rs: operation orders of the generic BaseVector. */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 Vp(const float* p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* vec3.rs:224-233 */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* vec3.rs:303-321 */
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.rs:285-291 */
static inline v3 vnormalize(v3 a) { return vdivs(a, sqrtf(vdot(a, a))); }
/* vec3.rs:564-575: this - normal*2*dot(this, normal), evaluated left to right */
static inline v3 vreflect(v3 d, v3 n) { return vsub(d, vscale(vscale(n, 2.0f), vdot(d, n))); }

/* ---------------------------------------------------------------- generator */
/* src/rng.rs:19-41 (wyrand). */

#define WY0 0x2d358dccaa6c78a5ull
#define WY1 0x8bb84b93962eacc9ull

uint64_t oracle_rng_u64(uint64_t* state) {
    const uint64_t s = *state + WY0;
    *state = s;
    const unsigned __int128 t = (unsigned __int128)s * (unsigned __int128)(s ^ WY1);
    return (uint64_t)t ^ (uint64_t)(t >> 64);
}

float oracle_rng_f32(uint64_t* state) {
    const uint32_t u = (uint32_t)oracle_rng_u64(state);
    const uint32_t bits = 0x3F800000u + (u >> 9); /* (1<<30) - (1<<23) + (u >> 9) */
    float f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}

/* vec3.rs:549-558: three draws in x, y, z order, each minus 0.5, normalised. */
static v3 random_unit(uint64_t* rng) {
    const float x = oracle_rng_f32(rng) - 0.5f;
    const float y = oracle_rng_f32(rng) - 0.5f;
    const float z = oracle_rng_f32(rng) - 0.5f;
    return vnormalize(V(x, y, z));
}

/* ----------------------------------------------------------------- counters */

typedef struct OracleCounters {
    uint64_t paths, segments, hits, branch_visits, prim_tests[3];
} OracleCounters;

/* --------------------------------------------------------------- traversal */

typedef struct { float dist; uint32_t node; } StackEntry;
typedef struct { StackEntry* e; size_t len, cap; } Stack;

static void stack_push(Stack* s, float dist, uint32_t node) {
    if (s->len == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 64;
        s->e = (StackEntry*)realloc(s->e, s->cap * sizeof(StackEntry));
    }
    s->e[s->len].dist = dist;
    s->e[s->len].node = node;
    s->len++;
}

/* src/bvh.rs:25-43: slab test with true division; returns 1 and *tmin on a hit. */
static inline int box_hit(const RtNode* n, v3 o, v3 d, float* tmin_out) {
    const v3 t1 = vdiv(vsub(Vp(n->min), o), d);
    const v3 t2 = vdiv(vsub(Vp(n->max), o), d);
    const float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    const float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    if (tmax >= tmin && tmax > 0.0f) {
        *tmin_out = tmin;
        return 1;
    }
    return 0;
}

#define MIN_DISTANCE 0.001f /* src/shapes.rs:18 */

/* src/shapes.rs:60-87 */
static inline int sphere_hit(const RtSphere* s, v3 o, v3 d, float* t) {
    const v3 delta = vsub(o, Vp(s->center));
    const float b = vdot(delta, d);
    const float disc = b * b - vdot(delta, delta) + s->radius * s->radius;
    if (disc < 0.0f) return 0;
    const float t1 = -b - sqrtf(disc);
    if (t1 > MIN_DISTANCE) {
        *t = t1;
        return 1;
    }
    const float t2 = -b + sqrtf(disc);
    if (t2 > MIN_DISTANCE) {
        *t = t2;
        return 1;
    }
    return 0;
}

/* src/shapes.rs:141-154 */
static inline int plane_hit(const RtPlane* p, v3 o, v3 d, float* t) {
    const v3 n = Vp(p->normal);
    const float den = vdot(n, d);
    if (fabsf(den) < 1.1920929e-7f) return 0; /* f32::EPSILON */
    const float num = vdot(n, vsub(o, Vp(p->point)));
    const float tt = -(num / den);
    if (tt > MIN_DISTANCE) {
        *t = tt;
        return 1;
    }
    return 0;
}

/* src/shapes.rs:232-259 (Moeller-Trumbore, two-sided) */
static inline int triangle_hit(const RtTriangle* tr, v3 o, v3 d, float* t) {
    const v3 e1 = Vp(tr->e1), e2 = Vp(tr->e2);
    const v3 h = vcross(d, e2);
    const float det = vdot(e1, h);
    if (fabsf(det) < 2e-6f) return 0;
    const float inv = 1.0f / det;
    const v3 s = vsub(o, Vp(tr->a));
    const float u = inv * vdot(s, h);
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    const v3 q = vcross(s, e1);
    const float v = inv * vdot(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float tt = inv * vdot(e2, q);
    if (tt > MIN_DISTANCE) {
        *t = tt;
        return 1;
    }
    return 0;
}

typedef struct {
    const RtSceneDesc* scene;
    Stack stack; /* reused across shape families, src/lib.rs:224,308 */
    OracleCounters* counters;
} Tracer;

/* src/bvh.rs:227-298: ordered stack traversal of one family's BVH.
 * Returns the nearest distance (INFINITY if none) and its shape index. */
static float closest_shape(Tracer* tr, int kind, v3 o, v3 d, uint32_t* index_out) {
    const RtSceneDesc* sc = tr->scene;
    const RtNode* nodes = kind == RT_SHAPE_SPHERE  ? sc->sphere_nodes
                          : kind == RT_SHAPE_PLANE ? sc->plane_nodes
                                                   : sc->triangle_nodes;
    Stack* st = &tr->stack;
    st->len = 0;
    float best_t = INFINITY;
    uint32_t best_i = 0xFFFFFFFFu;

    stack_push(st, 0.0f, 0); /* the root box is never tested */
    while (st->len > 0) {
        const StackEntry entry = st->e[--st->len];
        if (best_t <= entry.dist) continue;
        const RtNode* node = &nodes[entry.node];
        if (!(node->b & RT_LEAF_BIT)) {
            const uint32_t child[2] = {node->a, node->b};
            float dist[2];
            int hit[2];
            for (int c = 0; c < 2; ++c) {
                hit[c] = box_hit(&nodes[child[c]], o, d, &dist[c]);
                if (hit[c] && !(dist[c] < best_t)) hit[c] = 0;
            }
            if (tr->counters) tr->counters->branch_visits++;
            if (hit[0] && hit[1]) {
                if (dist[0] < dist[1]) {
                    stack_push(st, dist[1], child[1]);
                    stack_push(st, dist[0], child[0]);
                } else {
                    stack_push(st, dist[0], child[0]);
                    stack_push(st, dist[1], child[1]);
                }
            } else if (hit[0]) {
                stack_push(st, dist[0], child[0]);
            } else if (hit[1]) {
                stack_push(st, dist[1], child[1]);
            }
        } else {
            const uint32_t first = node->a, count = node->b & ~RT_LEAF_BIT;
            if (tr->counters) tr->counters->prim_tests[kind] += count;
            for (uint32_t i = first; i < first + count; ++i) {
                float t;
                int hit;
                if (kind == RT_SHAPE_SPHERE) hit = sphere_hit(&sc->spheres[i], o, d, &t);
                else if (kind == RT_SHAPE_PLANE) hit = plane_hit(&sc->planes[i], o, d, &t);
                else hit = triangle_hit(&sc->triangles[i], o, d, &t);
                if (hit && t < best_t) {
                    best_t = t;
                    best_i = i;
                }
            }
        }
    }
    *index_out = best_i;
    return best_t;
}

/* src/lib.rs:314-333: the three families in order; `min_by` keeps the first of
 * equal distances.  A family's result exists only if its distance is finite
 * (src/bvh.rs:300). */
static void nearest_hit(Tracer* tr, v3 o, v3 d, RtHitRecord* rec) {
    rec->t = INFINITY;
    rec->kind = RT_SHAPE_NONE;
    rec->prim = RT_NO_INDEX;
    for (int kind = 0; kind < 3; ++kind) {
        uint32_t idx;
        const float t = closest_shape(tr, kind, o, d, &idx);
        if (isfinite(t) && (rec->kind == RT_SHAPE_NONE || t < rec->t)) {
            rec->t = t;
            rec->kind = (uint32_t)kind;
            rec->prim = idx;
        }
    }
}

/* ----------------------------------------------------- surface attributes */

/* src/shapes.rs:91-99,157-165,264-326.  Sphere and plane texture coordinates
 * are computed by the reference but can never reach a texture (scene-file
 * materials are solid colours, src/config.rs:147-155), so they are returned
 * as zero here. */
static void surface_at(const RtSceneDesc* sc, const RtHitRecord* rec, v3 p, v3* normal, float uv[2],
                       uint32_t* material) {
    uv[0] = uv[1] = 0.0f;
    if (rec->kind == RT_SHAPE_SPHERE) {
        const RtSphere* s = &sc->spheres[rec->prim];
        *normal = vnormalize(vsub(p, Vp(s->center)));
        *material = s->material;
    } else if (rec->kind == RT_SHAPE_PLANE) {
        const RtPlane* pl = &sc->planes[rec->prim];
        *normal = Vp(pl->normal);
        *material = pl->material;
    } else {
        const RtTriangle* t = &sc->triangles[rec->prim];
        const v3 e1 = Vp(t->e1), e2 = Vp(t->e2);
        *material = t->material;
        if (t->bary_idx == RT_NO_INDEX) {
            *normal = vnormalize(vcross(e1, e2));
            return;
        }
        /* src/shapes.rs:210-227 */
        const float* pre = &sc->bary_precomputed[4 * (size_t)t->bary_idx];
        const v3 ap = vsub(p, Vp(t->a));
        const float d20 = vdot(ap, e1), d21 = vdot(ap, e2);
        const float bv = (pre[2] * d20 - pre[1] * d21) / pre[3];
        const float bw = (pre[0] * d21 - pre[1] * d20) / pre[3];
        const float bu = 1.0f - bv - bw;
        if (t->normals_idx != RT_NO_INDEX) {
            const float* n = &sc->vertex_normals[9 * (size_t)t->normals_idx];
            const v3 w0 = vscale(Vp(n), bu), w1 = vscale(Vp(n + 3), bv), w2 = vscale(Vp(n + 6), bw);
            *normal = vnormalize(vadd(vadd(w0, w1), w2));
        } else {
            *normal = vnormalize(vcross(e1, e2));
        }
        if (t->uv_idx != RT_NO_INDEX) {
            const float* c = &sc->tex_coords[6 * (size_t)t->uv_idx];
            for (int k = 0; k < 2; ++k) uv[k] = (c[k] * bu + c[2 + k] * bv) + c[4 + k] * bw;
        }
    }
}

/* src/material.rs:192-239 */
static v3 color_sample(const RtSceneDesc* sc, const RtMaterial* m, const float uv_in[2]) {
    if (m->color_kind == RT_COLOR_SOLID) return Vp(m->rgb);
    float c[2];
    for (int k = 0; k < 2; ++k) {
        const float fr = uv_in[k] - truncf(uv_in[k]); /* f32::fract */
        float r = fmodf(fr, 1.0f);                    /* rem_euclid(1.) */
        if (r < 0.0f) r += 1.0f;
        c[k] = r;
    }
    c[1] = 1.0f - c[1];
    const uint32_t dim[2] = {m->tex_width, m->tex_height};
    size_t i0[2], i1[2];
    float de[2];
    for (int k = 0; k < 2; ++k) {
        const float e = c[k] * (float)(dim[k] - 1);
        const float e0f = floorf(e);
        /* `as usize`: NaN and negatives to 0, saturating above.  c is in [0, 1], so e0 never exceeds
         * dim - 1 (the reference would panic on the index if it did); the GPU kernel clamps there
         * instead of faulting, and so does this function, so that the two state the same thing. */
        size_t e0 = !(e0f > 0.0f) ? 0 : (e0f >= (float)(dim[k] - 1) ? (size_t)dim[k] - 1 : (size_t)e0f);
        size_t e1 = e0 + 1 < (size_t)dim[k] - 1 ? e0 + 1 : (size_t)dim[k] - 1;
        i0[k] = e0;
        i1[k] = e1;
        de[k] = e - e0f;
    }
    const uint8_t* tex = sc->texels + m->tex_offset;
    const size_t w = m->tex_width;
    const size_t at[4][2] = {{i0[0], i0[1]}, {i0[0], i1[1]}, {i1[0], i0[1]}, {i1[0], i1[1]}};
    v3 px[4];
    for (int k = 0; k < 4; ++k) {
        const uint8_t* t = tex + 3 * (at[k][0] + at[k][1] * w);
        px[k] = V((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f);
    }
    /* lerp(a, b, x) = a*(1-x) + b*x, src/vec3.rs:164-177 */
    const float dx = de[0], dy = de[1];
    const v3 c0 = vadd(vscale(px[0], 1.0f - dx), vscale(px[2], dx)); /* c00 -> c10 */
    const v3 c1 = vadd(vscale(px[1], 1.0f - dx), vscale(px[3], dx)); /* c01 -> c11 */
    return vadd(vscale(c0, 1.0f - dy), vscale(c1, dy));
}

enum { SCATTERED = 0, ABSORBED = 1, LIGHT = 2 };

/* src/material.rs:23-100.  On SCATTERED fills the new ray. */
static int scatter(const RtMaterial* m, v3 o_in, v3 d, v3 normal, v3 hit_point, uint64_t* rng,
                   v3* o_out, v3* d_out) {
    (void)o_in;
    const float EPS = 1.1920929e-7f;
    *o_out = vadd(hit_point, vscale(normal, 1e-4f));
    switch (m->kind) {
        case RT_MAT_LAMBERTIAN: {
            const v3 dir = vnormalize(vadd(normal, random_unit(rng)));
            const int near_zero = fabsf(dir.x) < EPS && fabsf(dir.y) < EPS && fabsf(dir.z) < EPS;
            *d_out = near_zero ? normal : dir;
            return SCATTERED;
        }
        case RT_MAT_METAL: {
            const v3 refl = vreflect(d, normal);
            if (m->param == 0.0f) {
                *d_out = refl;
                return SCATTERED;
            }
            const v3 dir = vnormalize(vadd(refl, vscale(random_unit(rng), m->param)));
            if (vdot(dir, normal) > 0.0f) {
                *d_out = dir;
                return SCATTERED;
            }
            return ABSORBED;
        }
        case RT_MAT_GLASS: {
            float eta;
            v3 n;
            if (vdot(d, normal) < 0.0f) {
                eta = 1.0f / m->param;
                n = normal;
            } else {
                eta = m->param;
                n = vneg(normal);
            }
            const float cosv = fminf(vdot(vneg(d), n), 1.0f);
            const float sinv = sqrtf(1.0f - cosv * cosv);
            float r0 = (1.0f - eta) / (1.0f + eta);
            r0 = r0 * r0;
            const float x = 1.0f - cosv;
            const float x5 = x * ((x * x) * (x * x)); /* powi(5) */
            const float reflectance = r0 + (1.0f - r0) * x5;
            /* `||` short-circuits: the draw happens only without total reflection */
            if (eta * sinv > 1.0f || oracle_rng_f32(rng) < reflectance) {
                *d_out = vreflect(d, n);
            } else {
                const v3 perp = vscale(vadd(d, vscale(n, cosv)), eta);
                const float disc = 1.0f - eta * eta * (1.0f - cosv * cosv);
                const v3 par = vscale(n, -sqrtf(disc));
                *d_out = vadd(perp, par);
            }
            return SCATTERED;
        }
        default:
            return LIGHT;
    }
}

/* src/lib.rs:304-375 */
static v3 ray_color(Tracer* tr, v3 o, v3 d, uint32_t max_bounces, uint64_t* rng) {
    const RtSceneDesc* sc = tr->scene;
    int have_color = 0;
    v3 color = V(0, 0, 0);
    for (uint32_t bounce = 0; bounce < max_bounces; ++bounce) {
        RtHitRecord rec;
        nearest_hit(tr, o, d, &rec);
        if (tr->counters) tr->counters->segments++;
        if (rec.kind == RT_SHAPE_NONE) {
            const float a = 0.5f * (d.y + 1.0f);
            if (!have_color) { color = V(1, 1, 1); have_color = 1; }
            const v3 sky = vadd(vscale(V(0.2f, 0.2f, 0.8f), 1.0f - a), vscale(V(1, 1, 1), a));
            color = vmul(color, sky);
            break;
        }
        if (tr->counters) tr->counters->hits++;
        const v3 p = vadd(o, vscale(d, rec.t)); /* src/bvh.rs:303 */
        v3 normal;
        float uv[2];
        uint32_t mat_i;
        surface_at(sc, &rec, p, &normal, uv, &mat_i);
        const RtMaterial* m = &sc->materials[mat_i];
        v3 no, nd;
        const int what = scatter(m, o, d, normal, p, rng, &no, &nd);
        if (what == ABSORBED) {
            color = V(0, 0, 0);
            have_color = 1;
            break;
        }
        if (!have_color) { color = V(1, 1, 1); have_color = 1; }
        color = vmul(color, color_sample(sc, m, uv));
        if (what == LIGHT) break;
        o = no;
        d = nd;
    }
    return have_color ? color : V(0, 0, 0);
}

/* ------------------------------------------------------------- render loop */

typedef struct {
    v3 top_left, row_step, column_step, camera;
    uint32_t W, H, x0, y0, tw, th;
} Frame;

static Frame make_frame(const RtRenderParams* p) {
    Frame f;
    f.top_left = Vp(p->top_left);
    f.camera = Vp(p->camera);
    f.W = p->width;
    f.H = p->height;
    /* src/lib.rs:189-190 */
    f.row_step = vdivs(Vp(p->top_edge), (float)(p->width - 1));
    f.column_step = vdivs(Vp(p->left_edge), (float)(p->height - 1));
    if (p->tile_w) {
        f.x0 = p->tile_x; f.y0 = p->tile_y; f.tw = p->tile_w; f.th = p->tile_h;
    } else {
        f.x0 = 0; f.y0 = 0; f.tw = p->width; f.th = p->height;
    }
    return f;
}

/* src/lib.rs:261-268 */
static void camera_ray(const Frame* f, uint32_t x, uint32_t y, float jx, float jy, v3* o, v3* d) {
    const v3 a = vscale(f->row_step, (float)x + jx / 2.0f);
    const v3 b = vscale(f->column_step, (float)y + jy / 2.0f);
    const v3 pixel = vadd(vadd(f->top_left, a), b);
    *o = f->camera;
    *d = vnormalize(vsub(pixel, f->camera));
}

static uint64_t substream_state(const RtRenderParams* p, uint64_t sample, uint64_t pixel) {
    const uint64_t k = sample * (uint64_t)p->width * (uint64_t)p->height + pixel;
    const uint64_t stride = 2ull + 3ull * (uint64_t)p->max_bounces;
    return p->rng_seed + k * stride * WY0;
}

/* `as u8` on f32: toward zero, saturating, NaN to 0 (src/convert.rs:53-55). */
static inline uint8_t to_u8(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* src/lib.rs:283-295 for one pixel: sum of n samples -> stored bytes. */
static void resolve_pixel(const float sum[3], uint32_t n, int running_average, uint32_t k, uint8_t px[3]) {
    for (int c = 0; c < 3; ++c) {
        const float mean = sum[c] / (float)n;
        const float corrected = sqrtf(mean); /* vec3.rs:504-509 */
        float v;
        if (running_average) {
            const float old = (float)px[c] / 255.0f; /* vec3.rs:515-520 */
            v = (old * (float)k + corrected) / ((float)k + 1.0f);
        } else {
            v = corrected;
        }
        px[c] = to_u8(v * 255.0f); /* vec3.rs:528-538 */
    }
}

int oracle_resolve(const RtRenderParams* p, const float* sums, uint32_t n_samples, int running_average,
                   uint32_t sample_iteration, uint8_t* rgb8) {
    const Frame f = make_frame(p);
    const size_t n = (size_t)f.tw * f.th;
    for (size_t i = 0; i < n; ++i) resolve_pixel(&sums[3 * i], n_samples, running_average, sample_iteration, &rgb8[3 * i]);
    return 0;
}

/* One pixel's samples [s0, s0+n): left fold of ray colours (src/lib.rs:260-279). */
static void pixel_sum(Tracer* tr, const RtRenderParams* p, const Frame* f, uint32_t x, uint32_t y,
                      uint64_t s0, uint32_t n, uint64_t* serial_rng, float out[3]) {
    v3 acc = V(0, 0, 0);
    for (uint32_t j = 0; j < n; ++j) {
        uint64_t local, *rng;
        if (serial_rng) {
            rng = serial_rng;
        } else {
            local = substream_state(p, s0 + j, (uint64_t)y * f->W + x);
            rng = &local;
        }
        const float jx = oracle_rng_f32(rng);
        const float jy = oracle_rng_f32(rng);
        v3 o, d;
        camera_ray(f, x, y, jx, jy, &o, &d);
        if (tr->counters) tr->counters->paths++;
        const v3 c = ray_color(tr, o, d, p->max_bounces, rng);
        acc = j == 0 ? c : vadd(acc, c);
    }
    out[0] = acc.x; out[1] = acc.y; out[2] = acc.z;
}

typedef struct {
    const RtSceneDesc* scene;
    const RtRenderParams* params;
    Frame frame;
    uint64_t s0;
    uint32_t n;
    float* sums;
    atomic_size_t* next_chunk;
    size_t chunk_size, n_chunks, n_pixels;
    OracleCounters counters;
    int count;
    int cpu;
    uint32_t row_stride; /* every row_stride-th row of the region is rendered (1 = all) */
} Worker;

/* The reference's worker body (src/lib.rs:219-297) for one pass: take pixel
 * chunks from a shared counter until none are left. */
static void* worker_main(void* arg) {
    Worker* w = (Worker*)arg;
    if (w->cpu >= 0) {
        /* src/cpu_affinity.rs:33-45: thread i pinned to core i */
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(w->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = w->scene;
    tr.counters = w->count ? &w->counters : NULL;
    for (;;) {
        const size_t chunk = atomic_fetch_add_explicit(w->next_chunk, 1, memory_order_relaxed);
        if (chunk >= w->n_chunks) break;
        const size_t begin = chunk * w->chunk_size;
        size_t end = begin + w->chunk_size;
        if (end > w->n_pixels) end = w->n_pixels;
        for (size_t i = begin; i < end; ++i) {
            const uint32_t x = w->frame.x0 + (uint32_t)(i % w->frame.tw);
            const uint32_t y = w->frame.y0 + (uint32_t)(i / w->frame.tw) * w->row_stride;
            pixel_sum(&tr, w->params, &w->frame, x, y, w->s0, w->n, NULL, &w->sums[3 * i]);
        }
    }
    free(tr.stack.e);
    return NULL;
}

static void add_counters(OracleCounters* dst, const OracleCounters* src) {
    dst->paths += src->paths;
    dst->segments += src->segments;
    dst->hits += src->hits;
    dst->branch_visits += src->branch_visits;
    for (int k = 0; k < 3; ++k) dst->prim_tests[k] += src->prim_tests[k];
}

/* Per-pixel radiance sums over samples [sample_begin, sample_begin+count) in
 * sub-stream mode, on `n_threads` pinned workers with the reference's chunking
 * (T*T chunks of W*H/(T*T) pixels, src/lib.rs:195-204).  `counters` may be
 * NULL. */
int oracle_render_sums_rows(const RtSceneDesc* scene, const RtRenderParams* p, uint32_t sample_begin,
                            uint32_t sample_count, uint32_t row_stride, float* sums, int n_threads,
                            OracleCounters* counters);
int oracle_render_sums(const RtSceneDesc* scene, const RtRenderParams* p, uint32_t sample_begin,
                       uint32_t sample_count, float* sums, int n_threads, OracleCounters* counters) {
    return oracle_render_sums_rows(scene, p, sample_begin, sample_count, 1, sums, n_threads, counters);
}

/* The same over every `row_stride`-th row of the region only (rows y0, y0 +
 * row_stride, ...): `sums` holds ceil(rows / row_stride) rows.  A sample of the
 * whole frame for timing: the mix of paths is the frame's, the work a fraction. */
int oracle_render_sums_rows(const RtSceneDesc* scene, const RtRenderParams* p, uint32_t sample_begin,
                            uint32_t sample_count, uint32_t row_stride, float* sums, int n_threads,
                            OracleCounters* counters) {
    if (p->rng_mode != RT_RNG_SUBSTREAM || row_stride == 0) return -1;
    if (n_threads < 1) n_threads = 1;
    const Frame f = make_frame(p);
    const size_t n_pixels = (size_t)f.tw * ((f.th + row_stride - 1) / row_stride);
    size_t chunk_size = n_pixels / ((size_t)n_threads * (size_t)n_threads);
    if (chunk_size == 0) chunk_size = 1; /* the reference would panic here */
    const size_t n_chunks = (n_pixels + chunk_size - 1) / chunk_size;
    atomic_size_t next;
    atomic_init(&next, 0);

    Worker* ws = (Worker*)calloc((size_t)n_threads, sizeof(Worker));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int i = 0; i < n_threads; ++i) {
        ws[i].scene = scene; ws[i].params = p; ws[i].frame = f;
        ws[i].s0 = sample_begin; ws[i].n = sample_count; ws[i].sums = sums;
        ws[i].next_chunk = &next; ws[i].chunk_size = chunk_size; ws[i].n_chunks = n_chunks;
        ws[i].n_pixels = n_pixels; ws[i].count = counters != NULL;
        ws[i].cpu = n_threads > 1 ? i : -1;
        ws[i].row_stride = row_stride;
    }
    if (n_threads == 1) {
        worker_main(&ws[0]);
    } else {
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker_main, &ws[i]);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    }
    if (counters)
        for (int i = 0; i < n_threads; ++i) add_counters(counters, &ws[i].counters);
    free(ws);
    free(th);
    return 0;
}

/* `Scene::render`, src/lib.rs:188-301, writing the stored bytes into `rgb8`
 * (tile-sized).  In RT_RNG_SERIAL mode this is the literal single-thread
 * order: one generator, passes outermost, pixels in index order, the samples
 * of a pixel consecutive.  In RT_RNG_SUBSTREAM mode each pass may use several
 * threads; the result does not depend on their number.  `mean_out` (may be
 * NULL) receives the last pass's per-pixel mean radiance, 3 floats each. */
int oracle_render(const RtSceneDesc* scene, const RtRenderParams* p, uint8_t* rgb8, float* mean_out,
                  int n_threads, OracleCounters* counters) {
    const Frame f = make_frame(p);
    const size_t n_pixels = (size_t)f.tw * f.th;
    const uint32_t chunk = p->incremental ? p->incremental : p->samples_per_pixel;
    if (chunk == 0 || p->samples_per_pixel % chunk != 0) return -1;
    const uint32_t n_passes = p->samples_per_pixel / chunk;
    const int running = p->incremental != 0 || p->has_continue != 0;
    const uint32_t first_pass = p->has_continue ? p->continue_sampling : 0;

    float* sums = (float*)malloc(n_pixels * 3 * sizeof(float));
    if (!sums) return -1;
    uint64_t serial = p->rng_seed;
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = scene;
    tr.counters = counters;

    for (uint32_t pass = 0; pass < n_passes; ++pass) {
        const uint32_t k = pass + first_pass; /* sample_iteration, src/lib.rs:233-234 */
        if (p->rng_mode == RT_RNG_SERIAL) {
            for (size_t i = 0; i < n_pixels; ++i) {
                const uint32_t x = f.x0 + (uint32_t)(i % f.tw), y = f.y0 + (uint32_t)(i / f.tw);
                pixel_sum(&tr, p, &f, x, y, 0, chunk, &serial, &sums[3 * i]);
            }
        } else {
            const int rc = oracle_render_sums(scene, p, k * chunk, chunk, sums, n_threads, counters);
            if (rc) { free(sums); return rc; }
        }
        for (size_t i = 0; i < n_pixels; ++i) resolve_pixel(&sums[3 * i], chunk, running, k, &rgb8[3 * i]);
    }
    if (mean_out)
        for (size_t i = 0; i < n_pixels * 3; ++i) mean_out[i] = sums[i] / (float)chunk;
    free(sums);
    free(tr.stack.e);
    return 0;
}

/* Nearest hit of each pixel's camera ray for sample `sample_index` (one
 * iteration of src/lib.rs:313-333). */
int oracle_trace_primary(const RtSceneDesc* scene, const RtRenderParams* p, uint32_t sample_index,
                         int jitter, RtHitRecord* out, OracleCounters* counters) {
    const Frame f = make_frame(p);
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = scene;
    tr.counters = counters;
    for (uint32_t ty = 0; ty < f.th; ++ty)
        for (uint32_t tx = 0; tx < f.tw; ++tx) {
            const uint32_t x = f.x0 + tx, y = f.y0 + ty;
            float jx = 0.0f, jy = 0.0f;
            if (jitter) {
                uint64_t s = substream_state(p, sample_index, (uint64_t)y * f.W + x);
                jx = oracle_rng_f32(&s);
                jy = oracle_rng_f32(&s);
            }
            v3 o, d;
            camera_ray(&f, x, y, jx, jy, &o, &d);
            nearest_hit(&tr, o, d, &out[(size_t)ty * f.tw + tx]);
        }
    free(tr.stack.e);
    return 0;
}

int oracle_trace_rays(const RtSceneDesc* scene, const float* origins, const float* directions,
                      uint64_t n, RtHitRecord* out, OracleCounters* counters) {
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = scene;
    tr.counters = counters;
    for (uint64_t i = 0; i < n; ++i) nearest_hit(&tr, Vp(&origins[3 * i]), Vp(&directions[3 * i]), &out[i]);
    free(tr.stack.e);
    return 0;
}

/* Single pieces exposed for known-answer tests. */
int oracle_sphere_hit(const RtSphere* s, const float o[3], const float d[3], float* t) { return sphere_hit(s, Vp(o), Vp(d), t); }
int oracle_plane_hit(const RtPlane* s, const float o[3], const float d[3], float* t) { return plane_hit(s, Vp(o), Vp(d), t); }
int oracle_triangle_hit(const RtTriangle* s, const float o[3], const float d[3], float* t) { return triangle_hit(s, Vp(o), Vp(d), t); }
int oracle_box_hit(const RtNode* n, const float o[3], const float d[3], float* t) { return box_hit(n, Vp(o), Vp(d), t); }
void oracle_color_sample(const RtSceneDesc* sc, uint32_t material, const float uv[2], float out[3]) {
    const v3 c = color_sample(sc, &sc->materials[material], uv);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
/* One scatter event: returns 0 scattered / 1 absorbed / 2 light. */
int oracle_scatter(const RtSceneDesc* sc, uint32_t material, const float d[3], const float normal[3],
                   const float hit_point[3], uint64_t* rng, float o_out[3], float d_out[3]) {
    v3 no = V(0, 0, 0), nd = V(0, 0, 0);
    const int r = scatter(&sc->materials[material], V(0, 0, 0), Vp(d), Vp(normal), Vp(hit_point), rng, &no, &nd);
    o_out[0] = no.x; o_out[1] = no.y; o_out[2] = no.z;
    d_out[0] = nd.x; d_out[1] = nd.y; d_out[2] = nd.z;
    return r;
}
void oracle_surface_at(const RtSceneDesc* sc, const RtHitRecord* rec, const float p[3], float normal[3],
                       float uv[2], uint32_t* material) {
    v3 n;
    surface_at(sc, rec, Vp(p), &n, uv, material);
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
}
void oracle_ray_color(const RtSceneDesc* sc, const float o[3], const float d[3], uint32_t max_bounces,
                      uint64_t* rng, float out[3]) {
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = sc;
    const v3 c = ray_color(&tr, Vp(o), Vp(d), max_bounces, rng);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
    free(tr.stack.e);
}
