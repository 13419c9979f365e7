"""Thin Python wrapper over the C ABI of librt_mi355.so (include/rt_mi355.h).

Every method is one call across the boundary; nothing is computed here.  There
is no fallback: constructing a Renderer without the library or without a GPU
raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi as F
from .scene import HIT_DT, Scene


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"librt_mi355 error {code}: {message}")
        self.code = code


class Renderer:
    def __init__(self, device: int = 0):
        self._lib = F.device_lib()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(device, C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise RtError(rc, "rt_create failed (no usable GPU?)")
        self._scene = None

    # ---------------------------------------------------------------- plumbing
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def close(self) -> None:
        if self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option: int, value: int) -> None:
        self._check(self._lib.rt_set_option(self._ctx, option, value))

    # ------------------------------------------------------------------- scene
    def upload(self, scene: Scene) -> None:
        d = scene.desc()
        self._check(self._lib.rt_upload_scene(self._ctx, C.byref(d)))
        self._scene = scene

    def build_bvh_triangles(self, triangles: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """`BvhNode::new` (/root/reference/src/bvh.rs:48) for triangles, on the GPU.
        Returns (nodes, triangles in post-build order); the input is not modified."""
        from .scene import TRIANGLE_DT

        return self._build_bvh(self._lib.rt_build_bvh_triangles, triangles, TRIANGLE_DT)

    def build_bvh_spheres(self, spheres: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """The same for spheres."""
        from .scene import SPHERE_DT

        return self._build_bvh(self._lib.rt_build_bvh_spheres, spheres, SPHERE_DT)

    def _build_bvh(self, fn, shapes, dtype):
        from .scene import NODE_DT

        tris = np.ascontiguousarray(shapes, dtype=dtype).copy()
        raw, n_nodes = C.c_void_p(), F.u32(0)
        self._check(fn(self._ctx, tris.ctypes.data if tris.size else None, tris.size, C.byref(raw), C.byref(n_nodes)))
        try:
            buf = (C.c_char * (n_nodes.value * NODE_DT.itemsize)).from_address(raw.value)
            nodes = np.frombuffer(buf, dtype=NODE_DT, count=n_nodes.value).copy()
        finally:
            self._lib.rt_free(raw)
        return nodes, tris

    # ------------------------------------------------------------------ render
    def render_into(self, params, pixels_ptr, progress=None) -> None:
        """rt_render into caller memory (an address or ctypes pointer)."""
        cb = F.RtProgressFn(lambda done, _user: progress(int(done))) if progress else None
        self._check(self._lib.rt_render(self._ctx, C.byref(params), pixels_ptr,
                                        C.cast(cb, C.c_void_p) if cb else None, None))

    def render(self, params, rgb8: np.ndarray | None = None, progress=None) -> np.ndarray:
        """`Scene::render` into a numpy image (read first if incremental/continue)."""
        h, w = params.tile_shape()
        if rgb8 is None:
            rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        assert rgb8.dtype == np.uint8 and rgb8.shape == (h, w, 3) and rgb8.flags.c_contiguous
        self.render_into(params, rgb8.ctypes.data, progress)
        return rgb8

    def render_sums(self, params, sample_begin: int, sample_count: int) -> np.ndarray:
        h, w = params.tile_shape()
        out = np.empty((h, w, 3), dtype=np.float32)
        self._check(self._lib.rt_render_sums(self._ctx, C.byref(params), sample_begin, sample_count,
                                             out.ctypes.data, 0))
        return out

    def render_sums_device(self, params, sample_begin: int, sample_count: int, device_ptr: int) -> None:
        """Same, into device memory on this context's GPU (3 floats per pixel)."""
        self._check(self._lib.rt_render_sums(self._ctx, C.byref(params), sample_begin, sample_count,
                                             C.c_void_p(device_ptr), 1))

    def render_sums_onto(self, params, sample_begin: int, sample_count: int, sums: np.ndarray) -> np.ndarray:
        """Carries the fold in `sums` (samples [0, sample_begin) of every pixel) on over
        sample_count more samples, in place; equals render_sums(params, 0, begin + count)."""
        h, w = params.tile_shape()
        if sums.dtype != np.float32 or sums.shape != (h, w, 3) or not sums.flags.c_contiguous or not sums.flags.writeable:
            raise ValueError("sums must be a writable C-contiguous float32 array of shape (h, w, 3)")
        self._check(self._lib.rt_render_sums_onto(self._ctx, C.byref(params), sample_begin, sample_count,
                                                  sums.ctypes.data, 0))
        return sums

    def trace_samples(self, params, sample_begin: int, sample_count: int) -> int:
        """Traces as many of the samples as fit in flight (their radiance stays on the
        device until fold_samples) and returns how many that was."""
        traced = F.u32(0)
        self._check(self._lib.rt_trace_samples(self._ctx, C.byref(params), sample_begin, sample_count, C.byref(traced)))
        return int(traced.value)

    def reserve_samples(self, params, sample_count: int) -> int:
        """Sizes the device workspace for up to `sample_count` samples of the region in
        flight and returns how many one trace_samples call will take."""
        fit = F.u32(0)
        self._check(self._lib.rt_reserve_samples(self._ctx, C.byref(params), sample_count, C.byref(fit)))
        return int(fit.value)

    def fold_samples(self, params, carry_on: bool, sums, sums_on_device: bool = False):
        """Folds the traced samples, in sample order, onto `sums` (carry_on) or into it
        from nothing.  `sums`: float32 (h, w, 3) array, or a device pointer."""
        if sums_on_device:
            ptr = C.c_void_p(int(sums))
        else:
            h, w = params.tile_shape()
            if sums.dtype != np.float32 or sums.shape != (h, w, 3) or not sums.flags.c_contiguous or not sums.flags.writeable:
                raise ValueError("sums must be a writable C-contiguous float32 array of shape (h, w, 3)")
            ptr = C.c_void_p(sums.ctypes.data)
        self._check(self._lib.rt_fold_samples(self._ctx, C.byref(params), int(carry_on), ptr, int(sums_on_device)))
        return sums

    def resolve(self, params, sums, n_samples: int, running_average: bool = False, sample_iteration: int = 0,
                rgb8: np.ndarray | None = None, sums_on_device: bool = False) -> np.ndarray:
        h, w = params.tile_shape()
        if rgb8 is None:
            rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        if sums_on_device:
            ptr = C.c_void_p(int(sums))
        else:
            sums = np.ascontiguousarray(sums, dtype=np.float32)
            ptr = C.c_void_p(sums.ctypes.data)
        self._check(self._lib.rt_resolve(self._ctx, C.byref(params), ptr, int(sums_on_device), n_samples,
                                         int(running_average), sample_iteration, rgb8.ctypes.data))
        return rgb8

    # ------------------------------------------------------------------- trace
    def trace_primary(self, params, sample_index: int = 0, jitter: bool = False) -> np.ndarray:
        h, w = params.tile_shape()
        out = np.zeros((h, w), dtype=HIT_DT)
        self._check(self._lib.rt_trace_primary(self._ctx, C.byref(params), sample_index, int(jitter),
                                               out.ctypes.data))
        return out

    def trace_rays(self, origins, directions) -> np.ndarray:
        origins = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        directions = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert origins.shape == directions.shape
        out = np.zeros(origins.shape[0], dtype=HIT_DT)
        self._check(self._lib.rt_trace_rays(self._ctx, origins.ctypes.data, directions.ctypes.data,
                                            origins.shape[0], out.ctypes.data))
        return out

    def trace_rays_noalloc(self, origins, directions) -> np.ndarray:
        """rt_trace_rays_noalloc: the same answers out of the library's fixed scratch
        array, without any allocation on the device (at most 524288 rays)."""
        origins = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        directions = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        assert origins.shape == directions.shape
        out = np.zeros(origins.shape[0], dtype=HIT_DT)
        self._check(self._lib.rt_trace_rays_noalloc(self._ctx, origins.ctypes.data, directions.ctypes.data,
                                                    origins.shape[0], out.ctypes.data, 0))
        return out

    def self_test_division(self, n_cases: int = 1 << 28) -> int:
        """Number of quotients (of about n_cases) on which the kernels' shared-
        reciprocal division differs from IEEE division; 0 on a correct build."""
        bad = F.u64(0)
        self._check(self._lib.rt_self_test(self._ctx, 1, n_cases, C.byref(bad)))
        return int(bad.value)

    def self_test_reciprocal(self) -> int:
        """Number of binary32 values in the triangle test's reciprocal range (all are
        tried) whose reciprocal differs from `1.0f / x`; 0 on a correct build."""
        bad = F.u64(0)
        self._check(self._lib.rt_self_test(self._ctx, 2, 0, C.byref(bad)))
        return int(bad.value)

    def self_test_unit_byte(self) -> int:
        """Number of (byte, position in the texel word) cases whose texel conversion differs
        from `byte / 255.0f`; 0 on a correct build."""
        bad = F.u64(0)
        self._check(self._lib.rt_self_test(self._ctx, 3, 0, C.byref(bad)))
        return int(bad.value)

    # ------------------------------------------------------------------- stats
    def stats(self) -> dict:
        s = F.RtStats()
        self._check(self._lib.rt_get_stats(self._ctx, C.byref(s)))
        return {"paths": int(s.paths), "segments": int(s.segments), "hits": int(s.hits),
                "branch_visits": int(s.branch_visits), "prim_tests": [int(v) for v in s.prim_tests],
                "trace_launches": int(s.trace_launches), "trace_ms": float(s.trace_ms),
                "render_ms": float(s.render_ms)}

    def reset_stats(self) -> None:
        self._check(self._lib.rt_reset_stats(self._ctx))


class RendererGroup:
    """Several GPUs of one node behind one handle (rt_group_* of include/rt_mi355.h):
    `Scene::render` over a list of GPUs, storing the bytes one GPU stores."""

    def __init__(self, devices):
        self._lib = F.device_lib()
        self._group = C.c_void_p()
        devices = [int(d) for d in devices]
        arr = (C.c_int * len(devices))(*devices)
        rc = self._lib.rt_group_create(arr, len(devices), C.byref(self._group))
        if rc != 0:
            self._group = C.c_void_p()
            raise RtError(rc, self._lib.rt_group_last_error(None).decode())
        self._scene = None

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RtError(rc, self._lib.rt_group_last_error(self._group).decode())

    def close(self) -> None:
        if self._group:
            self._lib.rt_group_destroy(self._group)
            self._group = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.rt_group_size(self._group))

    def set_option(self, option: int, value: int) -> None:
        self._check(self._lib.rt_group_set_option(self._group, option, value))

    def upload(self, scene: Scene) -> None:
        d = scene.desc()
        self._check(self._lib.rt_group_upload_scene(self._group, C.byref(d)))
        self._scene = scene

    def render(self, params, rgb8: np.ndarray | None = None, progress=None) -> np.ndarray:
        h, w = params.tile_shape()
        if rgb8 is None:
            rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        assert rgb8.dtype == np.uint8 and rgb8.shape == (h, w, 3) and rgb8.flags.c_contiguous
        cb = F.RtProgressFn(lambda done, _user: progress(int(done))) if progress else None
        self._check(self._lib.rt_group_render(self._group, C.byref(params), rgb8.ctypes.data,
                                              C.cast(cb, C.c_void_p) if cb else None, None))
        return rgb8

    def render_sums(self, params, sample_begin: int, sample_count: int) -> np.ndarray:
        h, w = params.tile_shape()
        out = np.empty((h, w, 3), dtype=np.float32)
        self._check(self._lib.rt_group_render_sums(self._group, C.byref(params), sample_begin, sample_count,
                                                   out.ctypes.data))
        return out

    def stats(self) -> dict:
        s = F.RtStats()
        self._check(self._lib.rt_group_get_stats(self._group, C.byref(s)))
        return {"paths": int(s.paths), "segments": int(s.segments), "hits": int(s.hits),
                "branch_visits": int(s.branch_visits), "prim_tests": [int(v) for v in s.prim_tests],
                "trace_launches": int(s.trace_launches), "trace_ms": float(s.trace_ms),
                "render_ms": float(s.render_ms)}

    def reset_stats(self) -> None:
        self._check(self._lib.rt_group_reset_stats(self._group))
