"""Host-side mirror of the reference's `config::parse` / `Scene` API.

`parse(text)` stands where /root/reference/src/config.rs:15 stands (scene text
in, immutable scene out, BVHs built), `Scene.render()` where
/root/reference/src/lib.rs:188 stands (render into the memory-mapped
`target/out.ppm`).  The work is done by the two native libraries; this module
only owns buffers and passes pointers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _ffi as F

# numpy views of the C structs (same layout as include/rt_mi355.h)
NODE_DT = np.dtype([("min", "<f4", 3), ("a", "<u4"), ("max", "<f4", 3), ("b", "<u4")])
SPHERE_DT = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4")])
PLANE_DT = np.dtype([("point", "<f4", 3), ("normal", "<f4", 3), ("material", "<u4")])
TRIANGLE_DT = np.dtype([("a", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("normals_idx", "<u4"),
                        ("uv_idx", "<u4"), ("bary_idx", "<u4"), ("material", "<u4")])
MATERIAL_DT = np.dtype([("kind", "<u4"), ("param", "<f4"), ("color_kind", "<u4"), ("rgb", "<f4", 3),
                        ("tex_width", "<u4"), ("tex_height", "<u4"), ("tex_offset", "<u8")])
HIT_DT = np.dtype([("t", "<f4"), ("kind", "<u4"), ("prim", "<u4")])

assert NODE_DT.itemsize == C.sizeof(F.RtNode) == 32
assert SPHERE_DT.itemsize == C.sizeof(F.RtSphere) == 20
assert PLANE_DT.itemsize == C.sizeof(F.RtPlane) == 28
assert TRIANGLE_DT.itemsize == C.sizeof(F.RtTriangle) == 52
assert MATERIAL_DT.itemsize == C.sizeof(F.RtMaterial) == 40
assert HIT_DT.itemsize == C.sizeof(F.RtHitRecord) == 12


def _ptr(arr: np.ndarray, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr.size else C.POINTER(ctype)()


def _view(ptr, count: int, dtype: np.dtype) -> np.ndarray:
    """Copy `count` records starting at a ctypes pointer into a numpy array."""
    if not count:
        return np.zeros(0, dtype=dtype)
    addr = C.cast(ptr, C.c_void_p).value
    buf = (C.c_char * (count * dtype.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def make_params(top_left, top_edge, left_edge, camera, width, height, samples_per_pixel, max_bounces,
                incremental=0, continue_sampling=None, rng_mode=F.RT_RNG_SUBSTREAM,
                rng_seed=F.RT_RNG_SEED) -> F.RtRenderParams:
    """Screen + Camera of /root/reference/src/lib.rs:378-418 as a parameter block."""
    p = F.RtRenderParams()
    p.top_left[:] = [float(v) for v in top_left]
    p.top_edge[:] = [float(v) for v in top_edge]
    p.left_edge[:] = [float(v) for v in left_edge]
    p.camera[:] = [float(v) for v in camera]
    p.width, p.height = int(width), int(height)
    p.samples_per_pixel, p.max_bounces = int(samples_per_pixel), int(max_bounces)
    p.incremental = int(incremental)
    p.has_continue = 0 if continue_sampling is None else 1
    p.continue_sampling = 0 if continue_sampling is None else int(continue_sampling)
    p.rng_mode, p.rng_seed = int(rng_mode), int(rng_seed)
    return p


class Scene:
    """The immutable scene of /root/reference/src/lib.rs:103-111 as flat arrays.

    Attribute names follow the reference's fields; every array is a numpy array
    laid out exactly as include/rt_mi355.h describes, in post-BVH-build order.
    """

    def __init__(self, params, spheres, planes, triangles, vertex_normals, tex_coords,
                 bary_precomputed, materials, texels, sphere_nodes, plane_nodes, triangle_nodes):
        self.params = params
        self.spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DT)
        self.planes = np.ascontiguousarray(planes, dtype=PLANE_DT)
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE_DT)
        self.vertex_normals = np.ascontiguousarray(vertex_normals, dtype="<f4").reshape(-1, 3, 3)
        self.tex_coords = np.ascontiguousarray(tex_coords, dtype="<f4").reshape(-1, 3, 2)
        self.bary_precomputed = np.ascontiguousarray(bary_precomputed, dtype="<f4").reshape(-1, 4)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        self.texels = np.ascontiguousarray(texels, dtype=np.uint8).reshape(-1)
        self.sphere_nodes = np.ascontiguousarray(sphere_nodes, dtype=NODE_DT)
        self.plane_nodes = np.ascontiguousarray(plane_nodes, dtype=NODE_DT)
        self.triangle_nodes = np.ascontiguousarray(triangle_nodes, dtype=NODE_DT)

    # ------------------------------------------------------------------ build
    @classmethod
    def from_shapes(cls, params, spheres=None, planes=None, triangles=None, vertex_normals=None,
                    tex_coords=None, bary_precomputed=None, materials=None, texels=None) -> "Scene":
        """Build the three BVHs (permuting the shape arrays, as
        /root/reference/src/config.rs:113-134 does) around caller-made shapes."""
        lib = F.host_lib()

        def arr(a, dt):
            return np.array(a if a is not None else [], dtype=dt).reshape(-1).copy()

        spheres, planes, triangles = arr(spheres, SPHERE_DT), arr(planes, PLANE_DT), arr(triangles, TRIANGLE_DT)

        def build(fn, shapes):
            n_nodes = F.u32(0)
            raw = fn(shapes.ctypes.data if shapes.size else None, shapes.size, C.byref(n_nodes))
            if not raw:
                raise MemoryError("BVH build failed")
            nodes = _view(C.c_void_p(raw), n_nodes.value, NODE_DT)
            lib.rth_free(raw)
            return nodes

        sn = build(lib.rth_build_bvh_spheres, spheres)
        pn = build(lib.rth_build_bvh_planes, planes)
        tn = build(lib.rth_build_bvh_triangles, triangles)
        empty_f = np.zeros(0, dtype="<f4")
        return cls(params, spheres, planes, triangles,
                   vertex_normals if vertex_normals is not None else empty_f,
                   tex_coords if tex_coords is not None else empty_f,
                   bary_precomputed if bary_precomputed is not None else empty_f,
                   materials if materials is not None else np.zeros(0, dtype=MATERIAL_DT),
                   texels if texels is not None else np.zeros(0, dtype=np.uint8), sn, pn, tn)

    def refined(self, max_leaf: int = 8) -> tuple["Scene", np.ndarray]:
        """The opt-in, non-parity faster tree (rth_refine_bvh_triangles of include/rt_host.h):
        the reference's triangle tree with every leaf above `max_leaf` triangles split on,
        past the stop rule of /root/reference/src/bvh.rs:169.  Returns (scene, order):
        triangle i of the new scene is triangle order[i] of this one.  Hit distances are the
        same numbers (but for rays that graze a new leaf's box); primitive ids are positions in
        the new order."""
        lib = F.host_lib()
        tris = self.triangles.copy()
        order = np.zeros(tris.size, dtype=np.uint32)
        n_out = F.u32(0)
        raw = lib.rth_refine_bvh_triangles(self.triangle_nodes.ctypes.data, self.triangle_nodes.size,
                                           tris.ctypes.data if tris.size else None, tris.size, int(max_leaf),
                                           order.ctypes.data, C.byref(n_out))
        if not raw:
            raise MemoryError("BVH refinement failed")
        nodes = _view(C.c_void_p(raw), n_out.value, NODE_DT)
        lib.rth_free(raw)
        return Scene(self.params, self.spheres, self.planes, tris, self.vertex_normals, self.tex_coords,
                     self.bary_precomputed, self.materials, self.texels, self.sphere_nodes, self.plane_nodes, nodes), order

    # ------------------------------------------------------------------- view
    def desc(self) -> F.RtSceneDesc:
        """RtSceneDesc pointing into this object's arrays (keep `self` alive)."""
        d = F.RtSceneDesc()
        d.sphere_nodes = _ptr(self.sphere_nodes, F.RtNode)
        d.plane_nodes = _ptr(self.plane_nodes, F.RtNode)
        d.triangle_nodes = _ptr(self.triangle_nodes, F.RtNode)
        d.spheres = _ptr(self.spheres, F.RtSphere)
        d.planes = _ptr(self.planes, F.RtPlane)
        d.triangles = _ptr(self.triangles, F.RtTriangle)
        d.vertex_normals = _ptr(self.vertex_normals, F.f32)
        d.tex_coords = _ptr(self.tex_coords, F.f32)
        d.bary_precomputed = _ptr(self.bary_precomputed, F.f32)
        d.materials = _ptr(self.materials, F.RtMaterial)
        d.texels = _ptr(self.texels, F.u8)
        d.n_texel_bytes = self.texels.size
        d.n_sphere_nodes, d.n_plane_nodes = self.sphere_nodes.size, self.plane_nodes.size
        d.n_triangle_nodes = self.triangle_nodes.size
        d.n_spheres, d.n_planes, d.n_triangles = self.spheres.size, self.planes.size, self.triangles.size
        d.n_vertex_normals = self.vertex_normals.shape[0]
        d.n_tex_coords = self.tex_coords.shape[0]
        d.n_bary = self.bary_precomputed.shape[0]
        d.n_materials = self.materials.size
        return d

    # ----------------------------------------------------------------- render
    def render(self, path: str = "target/out.ppm", device: int = 0, progress=None) -> None:
        """`Scene::render`, /root/reference/src/lib.rs:188-301: render into the
        memory-mapped P6 file at `path` (created if needed, never truncated)."""
        from .renderer import Renderer  # local import: needs the GPU library

        lib = F.host_lib()
        err = C.create_string_buffer(512)
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        img = lib.rth_image_open(path.encode(), self.params.width, self.params.height, err, len(err))
        if not img:
            raise OSError(err.value.decode())
        try:
            with Renderer(device) as r:
                r.upload(self)
                r.render_into(self.params, lib.rth_image_pixels(img), progress)
            lib.rth_image_flush(img)
        finally:
            lib.rth_image_close(img)


def parse(text: str, base_dir: str = ".", renderer=None) -> Scene:
    """`config::parse`, /root/reference/src/config.rs:15: scene text to Scene.
    `base_dir` is where `obj/<name>.obj` is looked up (the reference uses the
    working directory).  Raises ValueError where the reference panics.
    With a `renderer` (raytracer_amd.Renderer) the triangle BVH is built on its
    GPU (same tree; the host builder remains the fallback for non-finite data)."""
    lib = F.host_lib()
    err = C.create_string_buffer(1024)
    if renderer is None:
        handle = lib.rth_parse_scene(text.encode(), base_dir.encode(), err, len(err))
    else:
        dev = F.device_lib()

        def build(_user, shapes, n, n_nodes):
            nodes = C.c_void_p()
            rc = dev.rt_build_bvh_triangles(renderer._ctx, shapes, n, C.byref(nodes), n_nodes)
            return nodes.value if rc == 0 else None

        hook = F.RthTriangleBvhFn(build)
        handle = lib.rth_parse_scene_with(text.encode(), base_dir.encode(), hook, None, err, len(err))
    if not handle:
        raise ValueError(err.value.decode())
    try:
        d = F.RtSceneDesc()
        p = F.RtRenderParams()
        lib.rth_scene_desc(handle, C.byref(d))
        lib.rth_scene_params(handle, C.byref(p))
        return Scene(
            p,
            _view(d.spheres, d.n_spheres, SPHERE_DT), _view(d.planes, d.n_planes, PLANE_DT),
            _view(d.triangles, d.n_triangles, TRIANGLE_DT),
            _view(d.vertex_normals, d.n_vertex_normals * 9, np.dtype("<f4")),
            _view(d.tex_coords, d.n_tex_coords * 6, np.dtype("<f4")),
            _view(d.bary_precomputed, d.n_bary * 4, np.dtype("<f4")),
            _view(d.materials, d.n_materials, MATERIAL_DT),
            _view(d.texels, d.n_texel_bytes, np.dtype(np.uint8)),
            _view(d.sphere_nodes, d.n_sphere_nodes, NODE_DT),
            _view(d.plane_nodes, d.n_plane_nodes, NODE_DT),
            _view(d.triangle_nodes, d.n_triangle_nodes, NODE_DT),
        )
    finally:
        lib.rth_free_scene(handle)
