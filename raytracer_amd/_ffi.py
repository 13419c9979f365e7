"""ctypes mirrors of include/rt_mi355.h and include/rt_host.h, and the loaders
for the two in-tree shared libraries.

The loaders fail loudly: there is no Python or CPU fallback for either library.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_MI355_LIB_DIR points at another build of the two libraries (A/B timing of kernel variants).
LIB_DIR = os.environ.get("RT_MI355_LIB_DIR") or os.path.join(_HERE, "lib")

RT_ABI_VERSION = 1
RT_LEAF_BIT = 0x80000000
RT_NO_INDEX = 0xFFFFFFFF
RT_RNG_SEED = 0x0EF6F79ED30BA75A

RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_GLASS, RT_MAT_LIGHT = 0, 1, 2, 3
RT_COLOR_SOLID, RT_COLOR_TEXTURE = 0, 1
RT_SHAPE_SPHERE, RT_SHAPE_PLANE, RT_SHAPE_TRIANGLE, RT_SHAPE_NONE = 0, 1, 2, 3
RT_RNG_SUBSTREAM, RT_RNG_SERIAL = 0, 1
RT_OPT_COUNT_TRAVERSAL, RT_OPT_TIME_KERNELS, RT_OPT_SAMPLES_IN_FLIGHT = 1, 2, 3
RT_OPT_TRACE_LDS_SLOTS, RT_OPT_TRACE_PHASE_MIN, RT_OPT_TRACE_CHUNK = 4, 5, 6
RT_OPT_TRACE_SHARE, RT_OPT_TRACE_PAIR_MIN = 7, 8
RT_OPT_TRACE_WAVES, RT_OPT_TRACE_CACHE_NODES, RT_OPT_TRACE_REFILL_MIN = 9, 10, 11
RT_OPT_TRACE_WAVES_PER_CU, RT_OPT_CHAINS, RT_OPT_SHADE_GROUPS_PER_CU, RT_OPT_FAST_SLAB = 12, 13, 14, 15
RT_OPT_FUSED_LOOP, RT_OPT_PATH_GROUP, RT_OPT_DEBUG_THROW, RT_OPT_TRACE_GUIDED, RT_OPT_PASSES_IN_FLIGHT = 17, 18, 19, 20, 21

f32, u32, u64, u8 = C.c_float, C.c_uint32, C.c_uint64, C.c_uint8


class RtNode(C.Structure):
    _fields_ = [("min", f32 * 3), ("a", u32), ("max", f32 * 3), ("b", u32)]


class RtSphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", u32)]


class RtPlane(C.Structure):
    _fields_ = [("point", f32 * 3), ("normal", f32 * 3), ("material", u32)]


class RtTriangle(C.Structure):
    _fields_ = [("a", f32 * 3), ("e1", f32 * 3), ("e2", f32 * 3), ("normals_idx", u32),
                ("uv_idx", u32), ("bary_idx", u32), ("material", u32)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", u32), ("param", f32), ("color_kind", u32), ("rgb", f32 * 3),
                ("tex_width", u32), ("tex_height", u32), ("tex_offset", u64)]


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("sphere_nodes", C.POINTER(RtNode)), ("plane_nodes", C.POINTER(RtNode)),
        ("triangle_nodes", C.POINTER(RtNode)), ("spheres", C.POINTER(RtSphere)),
        ("planes", C.POINTER(RtPlane)), ("triangles", C.POINTER(RtTriangle)),
        ("vertex_normals", C.POINTER(f32)), ("tex_coords", C.POINTER(f32)),
        ("bary_precomputed", C.POINTER(f32)), ("materials", C.POINTER(RtMaterial)),
        ("texels", C.POINTER(u8)), ("n_texel_bytes", u64),
        ("n_sphere_nodes", u32), ("n_plane_nodes", u32), ("n_triangle_nodes", u32),
        ("n_spheres", u32), ("n_planes", u32), ("n_triangles", u32),
        ("n_vertex_normals", u32), ("n_tex_coords", u32), ("n_bary", u32), ("n_materials", u32),
    ]


class RtRenderParams(C.Structure):
    _fields_ = [
        ("top_left", f32 * 3), ("top_edge", f32 * 3), ("left_edge", f32 * 3), ("camera", f32 * 3),
        ("width", u32), ("height", u32), ("samples_per_pixel", u32), ("max_bounces", u32),
        ("incremental", u32), ("has_continue", u32), ("continue_sampling", u32), ("rng_mode", u32),
        ("rng_seed", u64), ("tile_x", u32), ("tile_y", u32), ("tile_w", u32), ("tile_h", u32),
    ]

    def copy(self) -> "RtRenderParams":
        out = RtRenderParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(RtRenderParams))
        return out

    def tile_shape(self) -> tuple[int, int]:
        """(height, width) of the region pixel buffers cover."""
        if self.tile_w:
            return int(self.tile_h), int(self.tile_w)
        return int(self.height), int(self.width)


class RtHitRecord(C.Structure):
    _fields_ = [("t", f32), ("kind", u32), ("prim", u32)]


class RtStats(C.Structure):
    _fields_ = [("paths", u64), ("segments", u64), ("hits", u64), ("branch_visits", u64),
                ("prim_tests", u64 * 3), ("trace_launches", u64), ("trace_ms", C.c_double),
                ("render_ms", C.c_double)]


RtProgressFn = C.CFUNCTYPE(None, u32, C.c_void_p)

_DEVICE_SYMBOLS = {
    "rt_abi_version": (u32, []),
    "rt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_set_option": (C.c_int, [C.c_void_p, u32, u64]),
    "rt_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(RtSceneDesc)]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_render_sums": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, u32, C.c_void_p, C.c_int]),
    "rt_render_sums_onto": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, u32, C.c_void_p, C.c_int]),
    "rt_trace_samples": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, u32, C.POINTER(u32)]),
    "rt_reserve_samples": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, C.POINTER(u32)]),
    "rt_fold_samples": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_int, C.c_void_p, C.c_int]),
    "rt_resolve": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_int, u32, C.c_int,
                             u32, C.c_void_p]),
    "rt_trace_primary": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, C.c_int, C.c_void_p]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_void_p]),
    "rt_trace_rays_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64, C.c_void_p, C.c_int]),
    "rt_build_bvh_triangles": (C.c_int, [C.c_void_p, C.c_void_p, u32, C.POINTER(C.c_void_p), C.POINTER(u32)]),
    "rt_build_bvh_spheres": (C.c_int, [C.c_void_p, C.c_void_p, u32, C.POINTER(C.c_void_p), C.POINTER(u32)]),
    "rt_free": (None, [C.c_void_p]),
    "rt_host_alloc": (C.c_void_p, [C.c_size_t]),
    "rt_host_free": (None, [C.c_void_p]),
    "rt_self_test": (C.c_int, [C.c_void_p, u32, u64, C.POINTER(u64)]),
    "rt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(RtStats)]),
    "rt_reset_stats": (C.c_int, [C.c_void_p]),
    "rt_group_create": (C.c_int, [C.POINTER(C.c_int), u32, C.POINTER(C.c_void_p)]),
    "rt_group_destroy": (None, [C.c_void_p]),
    "rt_group_last_error": (C.c_char_p, [C.c_void_p]),
    "rt_group_size": (u32, [C.c_void_p]),
    "rt_group_member": (C.c_void_p, [C.c_void_p, u32]),
    "rt_group_set_option": (C.c_int, [C.c_void_p, u32, u64]),
    "rt_group_upload_scene": (C.c_int, [C.c_void_p, C.POINTER(RtSceneDesc)]),
    "rt_group_render": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_group_render_sums": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), u32, u32, C.c_void_p]),
    "rt_group_get_stats": (C.c_int, [C.c_void_p, C.POINTER(RtStats)]),
    "rt_group_reset_stats": (C.c_int, [C.c_void_p]),
}

RthTriangleBvhFn = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, u32, C.POINTER(u32))

_HOST_SYMBOLS = {
    "rth_parse_scene": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "rth_parse_scene_with": (C.c_void_p, [C.c_char_p, C.c_char_p, RthTriangleBvhFn, C.c_void_p, C.c_char_p, C.c_size_t]),
    "rth_free_scene": (None, [C.c_void_p]),
    "rth_scene_desc": (None, [C.c_void_p, C.POINTER(RtSceneDesc)]),
    "rth_scene_params": (None, [C.c_void_p, C.POINTER(RtRenderParams)]),
    "rth_build_bvh_spheres": (C.c_void_p, [C.c_void_p, u32, C.POINTER(u32)]),
    "rth_build_bvh_planes": (C.c_void_p, [C.c_void_p, u32, C.POINTER(u32)]),
    "rth_build_bvh_triangles": (C.c_void_p, [C.c_void_p, u32, C.POINTER(u32)]),
    "rth_free": (None, [C.c_void_p]),
    "rth_refine_bvh_triangles": (C.c_void_p, [C.c_void_p, u32, C.c_void_p, u32, u32, C.c_void_p, C.POINTER(u32)]),
    "rth_load_ppm_p6": (C.c_void_p, [C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.c_char_p, C.c_size_t]),
    "rth_image_open": (C.c_void_p, [C.c_char_p, u32, u32, C.c_char_p, C.c_size_t]),
    "rth_image_pixels": (C.c_void_p, [C.c_void_p]),
    "rth_image_header_len": (C.c_size_t, [C.c_void_p]),
    "rth_image_read_tile": (C.c_int, [C.c_void_p, u32, u32, u32, u32, C.c_void_p]),
    "rth_image_write_tile": (C.c_int, [C.c_void_p, u32, u32, u32, u32, C.c_void_p]),
    "rth_image_flush": (None, [C.c_void_p]),
    "rth_image_close": (None, [C.c_void_p]),
}


def _bind(lib: C.CDLL, table: dict) -> C.CDLL:
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_host = None
_device = None


def host_lib() -> C.CDLL:
    """librt_host.so (scene ingest).  Raises if it has not been built."""
    global _host
    if _host is None:
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C raytracer_amd/csrc` "
                               "(or __graft_entry__.build()) first")
        _host = _bind(C.CDLL(path), _HOST_SYMBOLS)
    return _host


def _map_torch_hip_runtime_first() -> None:
    """One HIP runtime per process, whatever the import order.

    PyTorch ships its own libamdhip64.so and asks the loader for it under that file
    name; librt_mi355.so asks for the runtime under its soname, libamdhip64.so.7.
    If ROCm's copy is mapped first (ours loaded before torch), the loader does not
    recognise it under torch's name and maps torch's copy as well: two runtimes,
    each with its own idea of devices, streams and allocations, and a pointer
    handed from one to the other is invalid.  Mapping torch's copy first makes both
    requests resolve to it (ours matches its soname).  torch itself is not imported."""
    import importlib.util

    if "torch" in sys.modules:
        return  # its runtime is mapped already
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return  # no torch in this environment: ROCm's runtime is the only one
    runtime = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(runtime):
        C.CDLL(runtime, mode=C.RTLD_GLOBAL)


def device_lib() -> C.CDLL:
    """librt_mi355.so (the HIP render loop).  Raises if it has not been built;
    nothing in this package renders without it."""
    global _device
    if _device is None:
        path = os.path.join(LIB_DIR, "librt_mi355.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C raytracer_amd/csrc` "
                               "(or __graft_entry__.build()) first; there is no CPU fallback")
        _map_torch_hip_runtime_first()
        _device = _bind(C.CDLL(path), _DEVICE_SYMBOLS)
        if _device.rt_abi_version() != RT_ABI_VERSION:
            raise RuntimeError("librt_mi355.so ABI version mismatch")
    return _device
