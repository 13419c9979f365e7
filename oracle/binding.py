"""ctypes binding of the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the `cpu_baseline` leg of bench.py, always as the checker, never by the product
package.  Parity is unpinned by the reference (see the header of rt_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import _ffi as F
from raytracer_amd.scene import HIT_DT, Scene

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")


class OracleCounters(C.Structure):
    _fields_ = [("paths", F.u64), ("segments", F.u64), ("hits", F.u64), ("branch_visits", F.u64),
                ("prim_tests", F.u64 * 3)]

    def as_dict(self) -> dict:
        return {"paths": int(self.paths), "segments": int(self.segments), "hits": int(self.hits),
                "branch_visits": int(self.branch_visits), "prim_tests": [int(v) for v in self.prim_tests]}


def build() -> str:
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _LIB_PATH


def build_native(out_dir: str) -> str | None:
    """The same source compiled for the machine this runs on (-O3 -march=native,
    still without FMA contraction or fast-math: same results), as the timed CPU
    baseline wants it.  Returns the library's path, or None when no compiler is at hand."""
    out = os.path.join(out_dir, "librt_oracle_native.so")
    cmd = ["gcc", "-std=gnu11", "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fno-unsafe-math-optimizations", f"-I{os.path.join(os.path.dirname(_HERE), 'include')}", "-shared", "-o", out,
           os.path.join(_HERE, "rt_oracle.c"), "-lm", "-lpthread"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
    except (OSError, subprocess.SubprocessError):
        return None
    return out


_lib = None
_libs = {}


def lib(path: str | None = None) -> C.CDLL:
    """The oracle library: the -O2 build next to this file, or the one at `path`."""
    global _lib
    if path is not None:
        if path not in _libs:
            _libs[path] = _declare(C.CDLL(path))
        return _libs[path]
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _declare(C.CDLL(_LIB_PATH))
    return _lib


def _declare(L: C.CDLL) -> C.CDLL:
    P = C.POINTER
    L.oracle_rng_u64.restype, L.oracle_rng_u64.argtypes = F.u64, [P(F.u64)]
    L.oracle_rng_f32.restype, L.oracle_rng_f32.argtypes = F.f32, [P(F.u64)]
    L.oracle_render_sums.restype = C.c_int
    L.oracle_render_sums.argtypes = [P(F.RtSceneDesc), P(F.RtRenderParams), F.u32, F.u32, C.c_void_p,
                                     C.c_int, P(OracleCounters)]
    L.oracle_render_sums_rows.restype = C.c_int
    L.oracle_render_sums_rows.argtypes = [P(F.RtSceneDesc), P(F.RtRenderParams), F.u32, F.u32, F.u32, C.c_void_p,
                                          C.c_int, P(OracleCounters)]
    L.oracle_render.restype = C.c_int
    L.oracle_render.argtypes = [P(F.RtSceneDesc), P(F.RtRenderParams), C.c_void_p, C.c_void_p, C.c_int,
                                P(OracleCounters)]
    L.oracle_resolve.restype = C.c_int
    L.oracle_resolve.argtypes = [P(F.RtRenderParams), C.c_void_p, F.u32, C.c_int, F.u32, C.c_void_p]
    L.oracle_trace_primary.restype = C.c_int
    L.oracle_trace_primary.argtypes = [P(F.RtSceneDesc), P(F.RtRenderParams), F.u32, C.c_int, C.c_void_p,
                                       P(OracleCounters)]
    L.oracle_trace_rays.restype = C.c_int
    L.oracle_trace_rays.argtypes = [P(F.RtSceneDesc), C.c_void_p, C.c_void_p, F.u64, C.c_void_p,
                                    P(OracleCounters)]
    for name, shape in (("oracle_sphere_hit", F.RtSphere), ("oracle_plane_hit", F.RtPlane),
                        ("oracle_triangle_hit", F.RtTriangle), ("oracle_box_hit", F.RtNode)):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = C.c_int, [P(shape), P(F.f32 * 3), P(F.f32 * 3), P(F.f32)]
    L.oracle_color_sample.restype = None
    L.oracle_color_sample.argtypes = [P(F.RtSceneDesc), F.u32, P(F.f32 * 2), P(F.f32 * 3)]
    L.oracle_scatter.restype = C.c_int
    L.oracle_scatter.argtypes = [P(F.RtSceneDesc), F.u32, P(F.f32 * 3), P(F.f32 * 3), P(F.f32 * 3),
                                 P(F.u64), P(F.f32 * 3), P(F.f32 * 3)]
    L.oracle_surface_at.restype = None
    L.oracle_surface_at.argtypes = [P(F.RtSceneDesc), P(F.RtHitRecord), P(F.f32 * 3), P(F.f32 * 3),
                                    P(F.f32 * 2), P(F.u32)]
    L.oracle_ray_color.restype = None
    L.oracle_ray_color.argtypes = [P(F.RtSceneDesc), P(F.f32 * 3), P(F.f32 * 3), F.u32, P(F.u64),
                                   P(F.f32 * 3)]
    return L


def _vec3(v):
    return (F.f32 * 3)(*[float(x) for x in v])


class Rng:
    """The reference generator, one stream (src/rng.rs:9-41)."""

    def __init__(self, state: int = F.RT_RNG_SEED):
        self.state = F.u64(state)

    def u64(self) -> int:
        return int(lib().oracle_rng_u64(C.byref(self.state)))

    def f32(self) -> np.float32:
        return np.float32(lib().oracle_rng_f32(C.byref(self.state)))


def render_sums(scene: Scene, params, sample_begin: int, sample_count: int, threads: int = 1,
                counters: OracleCounters | None = None) -> np.ndarray:
    h, w = params.tile_shape()
    out = np.zeros((h, w, 3), dtype=np.float32)
    d = scene.desc()
    rc = lib().oracle_render_sums(C.byref(d), C.byref(params), sample_begin, sample_count, out.ctypes.data,
                                  threads, C.byref(counters) if counters is not None else None)
    if rc:
        raise RuntimeError(f"oracle_render_sums failed ({rc})")
    return out


def render_sums_rows(scene: Scene, params, sample_begin: int, sample_count: int, row_stride: int, threads: int = 1,
                     counters: OracleCounters | None = None, library: C.CDLL | None = None) -> np.ndarray:
    """render_sums over every `row_stride`-th row of the region only."""
    h, w = params.tile_shape()
    out = np.zeros(((h + row_stride - 1) // row_stride, w, 3), dtype=np.float32)
    d = scene.desc()
    rc = (library or lib()).oracle_render_sums_rows(C.byref(d), C.byref(params), sample_begin, sample_count, row_stride,
                                                    out.ctypes.data, threads, C.byref(counters) if counters is not None else None)
    if rc:
        raise RuntimeError(f"oracle_render_sums_rows failed ({rc})")
    return out


def render(scene: Scene, params, rgb8: np.ndarray | None = None, threads: int = 1,
           counters: OracleCounters | None = None):
    """Full `Scene::render` semantics.  Returns (rgb8, mean radiance of the last pass)."""
    h, w = params.tile_shape()
    if rgb8 is None:
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    assert rgb8.dtype == np.uint8 and rgb8.shape == (h, w, 3) and rgb8.flags.c_contiguous
    mean = np.zeros((h, w, 3), dtype=np.float32)
    d = scene.desc()
    rc = lib().oracle_render(C.byref(d), C.byref(params), rgb8.ctypes.data, mean.ctypes.data, threads,
                             C.byref(counters) if counters is not None else None)
    if rc:
        raise RuntimeError(f"oracle_render failed ({rc})")
    return rgb8, mean


def resolve(params, sums: np.ndarray, n_samples: int, running_average: bool = False, sample_iteration: int = 0,
            rgb8: np.ndarray | None = None) -> np.ndarray:
    h, w = params.tile_shape()
    sums = np.ascontiguousarray(sums, dtype=np.float32)
    if rgb8 is None:
        rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
    lib().oracle_resolve(C.byref(params), sums.ctypes.data, n_samples, int(running_average), sample_iteration,
                         rgb8.ctypes.data)
    return rgb8


def trace_primary(scene: Scene, params, sample_index: int = 0, jitter: bool = False,
                  counters: OracleCounters | None = None) -> np.ndarray:
    h, w = params.tile_shape()
    out = np.zeros((h, w), dtype=HIT_DT)
    d = scene.desc()
    lib().oracle_trace_primary(C.byref(d), C.byref(params), sample_index, int(jitter), out.ctypes.data,
                               C.byref(counters) if counters is not None else None)
    return out


def trace_rays(scene: Scene, origins: np.ndarray, directions: np.ndarray,
               counters: OracleCounters | None = None) -> np.ndarray:
    origins = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    directions = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(origins.shape[0], dtype=HIT_DT)
    d = scene.desc()
    lib().oracle_trace_rays(C.byref(d), origins.ctypes.data, directions.ctypes.data, origins.shape[0],
                            out.ctypes.data, C.byref(counters) if counters is not None else None)
    return out


def shape_hit(shape, origin, direction):
    """Single ray/shape test; returns t or None."""
    fn = {F.RtSphere: "oracle_sphere_hit", F.RtPlane: "oracle_plane_hit", F.RtTriangle: "oracle_triangle_hit",
          F.RtNode: "oracle_box_hit"}[type(shape)]
    t = F.f32(0)
    hit = getattr(lib(), fn)(C.byref(shape), C.byref(_vec3(origin)), C.byref(_vec3(direction)), C.byref(t))
    return np.float32(t.value) if hit else None


def color_sample(scene: Scene, material: int, uv) -> np.ndarray:
    out = (F.f32 * 3)()
    d = scene.desc()
    lib().oracle_color_sample(C.byref(d), material, C.byref((F.f32 * 2)(float(uv[0]), float(uv[1]))), C.byref(out))
    return np.array(out, dtype=np.float32)


def scatter(scene: Scene, material: int, direction, normal, hit_point, rng: Rng):
    """One `Material::scatter`; returns (kind, origin, direction), kind 0/1/2 =
    scattered/absorbed/light."""
    o, dd = (F.f32 * 3)(), (F.f32 * 3)()
    d = scene.desc()
    kind = lib().oracle_scatter(C.byref(d), material, C.byref(_vec3(direction)), C.byref(_vec3(normal)),
                                C.byref(_vec3(hit_point)), C.byref(rng.state), C.byref(o), C.byref(dd))
    return kind, np.array(o, dtype=np.float32), np.array(dd, dtype=np.float32)


def surface_at(scene: Scene, kind: int, prim: int, point):
    rec = F.RtHitRecord(0.0, kind, prim)
    n, uv, m = (F.f32 * 3)(), (F.f32 * 2)(), F.u32(0)
    d = scene.desc()
    lib().oracle_surface_at(C.byref(d), C.byref(rec), C.byref(_vec3(point)), C.byref(n), C.byref(uv), C.byref(m))
    return np.array(n, dtype=np.float32), np.array(uv, dtype=np.float32), int(m.value)


def ray_color(scene: Scene, origin, direction, max_bounces: int, rng: Rng) -> np.ndarray:
    out = (F.f32 * 3)()
    d = scene.desc()
    lib().oracle_ray_color(C.byref(d), C.byref(_vec3(origin)), C.byref(_vec3(direction)), max_bounces,
                           C.byref(rng.state), C.byref(out))
    return np.array(out, dtype=np.float32)
