"""Tile-by-tile rendering into the memory-mapped image, on one GPU or many.

The reference renders "infinitely large images" (README.md:18) because the
framebuffer is only ever the mmap'd file: its workers lock one pixel chunk at a
time and write it in place (/root/reference/src/lib.rs:201-204,248-295).  Here
the unit is a rectangular tile.  A pass (the `incremental` samples, or all of
them) is rendered tile by tile; each tile's bytes are read from the mapping
when the running average of lib.rs:286-292 needs them, and written back when
the tile is done, so the file is a complete image after every pass and
`continue(n)` works on it exactly as on an untiled render.

Several GPUs: ranks form groups of `group_size`; tiles are dealt to groups
round-robin.  Inside a group every rank renders its share of the pass's samples
of the tile and the group's radiance sums are reduced (RCCL) onto the group's
first rank, which resolves and writes the tile.  With group_size 1 there is no
collective at all: every rank maps the same file and writes its own tiles.
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Protocol

import numpy as np

from . import _ffi as F
from .distributed import fold_in_sample_order, sample_share


class Image:
    """The memory-mapped P6 file of /root/reference/src/lib.rs:64-89 +
    src/mmap.rs:41-77 (created if missing, never truncated)."""

    def __init__(self, path: str, width: int, height: int):
        self._lib = F.host_lib()
        err = C.create_string_buffer(512)
        d = os.path.dirname(os.path.abspath(path))
        os.makedirs(d, exist_ok=True)
        self.width, self.height = int(width), int(height)
        self._h = self._lib.rth_image_open(path.encode(), self.width, self.height, err, len(err))
        if not self._h:
            raise OSError(err.value.decode())

    def pixels(self) -> np.ndarray:
        """(height, width, 3) uint8 view of the mapping (valid until close)."""
        n = self.width * self.height * 3
        buf = (C.c_uint8 * n).from_address(self._lib.rth_image_pixels(self._h))
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3)

    def read_tile(self, x: int, y: int, w: int, h: int, out: np.ndarray | None = None) -> np.ndarray:
        """The tile's bytes, into `out` (a C-contiguous (h, w, 3) uint8 array) when given."""
        if out is None:
            out = np.empty((h, w, 3), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.shape == (h, w, 3) and out.flags.c_contiguous
        if self._lib.rth_image_read_tile(self._h, x, y, w, h, out.ctypes.data) != 0:
            raise ValueError("tile outside the image")
        return out

    def write_tile(self, x: int, y: int, tile: np.ndarray) -> None:
        h, w, _ = tile.shape
        tile = np.ascontiguousarray(tile, dtype=np.uint8)
        if self._lib.rth_image_write_tile(self._h, x, y, w, h, tile.ctypes.data) != 0:
            raise ValueError("tile outside the image")

    def flush(self) -> None:
        self._lib.rth_image_flush(self._h)

    def close(self) -> None:
        if self._h:
            self._lib.rth_image_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def tile_grid(width: int, height: int, tile_w: int, tile_h: int) -> list[tuple[int, int, int, int]]:
    """(x, y, w, h) of every tile, row-major; edge tiles are smaller."""
    if tile_w < 1 or tile_h < 1:
        raise ValueError("tile size must be positive")
    return [(x, y, min(tile_w, width - x), min(tile_h, height - y))
            for y in range(0, height, tile_h) for x in range(0, width, tile_w)]


def tiles_of_group(n_tiles: int, n_groups: int, group: int) -> range:
    """Indices of the tiles group `group` of `n_groups` renders (round-robin, so
    neighbouring tiles, which cost about the same, go to different groups)."""
    if n_groups < 1 or not 0 <= group < n_groups:
        raise ValueError("bad group")
    return range(group, n_tiles, n_groups)


def passes(params) -> tuple[int, int, bool, int]:
    """(samples per pass, number of passes, running average?, first pass number),
    /root/reference/src/lib.rs:206-211,233-234,286."""
    chunk = int(params.incremental) or int(params.samples_per_pixel)
    if chunk == 0 or params.samples_per_pixel % chunk:
        raise ValueError("incremental must be non-zero and divide the samples per pixel")
    running = bool(params.incremental) or bool(params.has_continue)
    return chunk, int(params.samples_per_pixel) // chunk, running, int(params.continue_sampling) if params.has_continue else 0


class TileBackend(Protocol):
    """What the driver needs from a renderer.  `sums` objects are whatever the
    group's reduce works on (torch tensors); only the backend looks inside.
    A backend may also offer `pixel_buffer(h, w)` -> an (h, w, 3) uint8 array that is cheap
    for it to read and write (page-locked memory for a GPU); the tile loop then keeps
    three of them going round: one being filled from the file, one being rendered,
    one being written back."""

    def render_tile_pass(self, params, old: np.ndarray | None, out: np.ndarray | None = None) -> np.ndarray:
        """All of one pass of one tile on this rank alone: `params` describes the
        tile and the pass (see pass_params); returns the tile's new bytes (written over
        `old`, or into `out` when given: buffers of the tile loop's pool)."""

    def render_sums(self, params, sample_begin: int, sample_count: int): ...

    def zeros(self, params): ...

    def reserve(self, params, sample_count: int) -> int:
        """(exact groups only) Room for up to `sample_count` samples of the tile in flight; returns how many one trace takes."""

    def trace(self, params, sample_begin: int, sample_count: int) -> int:
        """(exact groups only) Trace as many of the samples as fit; their radiance waits for fold."""

    def fold(self, params, carry_on: bool):
        """(exact groups only) Fold the traced samples onto the sums tensor (or start it); returns it."""

    def resolve(self, params, sums, n_samples: int, running: bool, k: int, old: np.ndarray | None,
                out: np.ndarray | None = None) -> np.ndarray:
        """Mean, gamma, running average with `old` (when running) and quantisation of the tile's sums;
        the bytes are written over `old`, or into `out` when given (a buffer of the tile loop's pool)."""


def pass_params(params, rect, chunk: int, running: bool, k: int):
    """Render parameters for pass number `k` of tile `rect` as a render of its
    own: `chunk` samples that resume at pass k (sample indices k*chunk ...,
    running-average weight k), which is what lib.rs:233-234,286-292 do for the
    pixels of one chunk."""
    tp = params.copy()
    tp.tile_x, tp.tile_y, tp.tile_w, tp.tile_h = rect
    tp.samples_per_pixel = chunk
    tp.incremental = 0
    tp.has_continue = 1 if running else 0
    tp.continue_sampling = k if running else 0
    return tp


def render_tiled(backend: TileBackend, image: Image, params, tile_w: int, tile_h: int, world: int = 1, rank: int = 0,
                 group_size: int = 1, reduce_in_group: Callable | None = None, end_of_pass: Callable | None = None,
                 progress: Callable | None = None, exact_in_group: tuple[Callable, Callable, Callable] | None = None) -> None:
    """Render `params` (full image) into `image`.

    exact_in_group = (send, recv, agree_min), instead of reduce_in_group:
    `send(sums, member)` / `recv(sums, member)` move a tile's sums to / from another member
    of this rank's group (dist.send / dist.recv to the member's rank).  The members then
    trace pieces of the tile's samples at the same time and fold them in sample order, the
    sums going from member to member; the group's last member writes the tile, and its
    bytes are those of a group of one (raytracer_amd.distributed.fold_in_sample_order, per
    tile; `agree_min(n)` -> the minimum of n over the group's members).

    reduce_in_group(sums): sum over the ranks of this rank's group onto the
    group's first rank (dist.reduce on the group's process group); required when
    group_size > 1.  end_of_pass(): called by every rank after each pass (a
    barrier, so that the file is a whole image before anyone starts the next
    pass).  progress(samples_done) is called on rank 0 like the reference's
    "Samples: n" line.
    """
    if world % group_size:
        raise ValueError("group size must divide the number of ranks")
    if group_size > 1 and reduce_in_group is None and exact_in_group is None:
        raise ValueError("groups of several ranks need a reduce or the exact hand-over")
    if group_size > 1 and exact_in_group is not None and len(exact_in_group) != 3:
        raise ValueError("exact_in_group is (send, recv, agree_min): the members must agree on one piece size")
    n_groups, group, member = world // group_size, rank // group_size, rank % group_size
    chunk, n_passes, running, first_pass = passes(params)
    grid = tile_grid(int(params.width), int(params.height), tile_w, tile_h)
    mine = tiles_of_group(len(grid), n_groups, group)
    # Tile copies between the mapping and packed buffers run on a helper thread, so
    # that the GPU renders tile i while tile i-1 is written and tile i+1 is read
    # (the copies are plain memcpy over up to hundreds of MB of page cache).
    copier = ThreadPoolExecutor(max_workers=1)
    try:
        _render_passes(backend, image, params, grid, mine, copier, chunk, n_passes, running, first_pass, rank, group_size,
                       member, reduce_in_group, end_of_pass, progress, exact_in_group)
    finally:
        copier.shutdown()
    image.flush()


def _pool_of(backend) -> "_BufferPool":
    """The backend's pool of pixel buffers, kept on it from one render to the next (page-locked
    memory is dear to allocate and is not given back until the backend goes)."""
    pool = getattr(backend, "tile_pool", None)
    if pool is None:
        pool = _BufferPool(backend, 3)
        try:
            backend.tile_pool = pool
        except AttributeError:
            pass
    return pool


class _BufferPool:
    """Up to `n` pixel buffers of the backend's kind (see TileBackend), re-used from tile to
    tile; a tile smaller than a buffer gets a view of its start.  Without `pixel_buffer` on
    the backend every request is a fresh numpy array."""

    def __init__(self, backend, n: int):
        self._make = getattr(backend, "pixel_buffer", None)
        self._free: list[np.ndarray] = []
        self._bases: dict[int, np.ndarray] = {}
        self._left = n

    def take(self, h: int, w: int) -> np.ndarray:
        if self._make is None:
            return np.empty((h, w, 3), dtype=np.uint8)
        need = h * w * 3
        for i, b in enumerate(self._free):
            if b.size >= need:
                base = self._free.pop(i)
                break
        else:
            if self._left == 0 and self._free:
                self._free.pop(0)  # too small for this tile: let it go
                self._left += 1
            self._left -= 1
            base = self._make(h, w).reshape(-1)
        view = base[:need].reshape(h, w, 3)
        self._bases[view.ctypes.data] = base
        return view

    def give_back(self, tile: np.ndarray) -> None:
        base = self._bases.pop(tile.ctypes.data, None)
        if base is not None:
            self._free.append(base)


def _exact_tile_sums(backend, tp, begin, count, group_size, member, send, recv, agree_min):
    """The tile's sums by the hand-over of render_pass_exact; complete on the last member."""
    return fold_in_sample_order(lambda first, n: backend.trace(tp, first, n), lambda carry_on: backend.fold(tp, carry_on),
                                lambda: backend.sums(tp), lambda: backend.zeros(tp), lambda n: backend.reserve(tp, n),
                                begin, count, group_size, member, send, recv, agree_min)


def _render_passes(backend, image, params, grid, mine, copier, chunk, n_passes, running, first_pass, rank, group_size,
                   member, reduce_in_group, end_of_pass, progress, exact_in_group=None) -> None:
    for p in range(n_passes):
        k = p + first_pass
        if group_size == 1:
            rects = [grid[ti] for ti in mine]
            pool = _pool_of(backend)

            def read(rect):
                return image.read_tile(*rect, out=pool.take(rect[3], rect[2]))

            def write(rect, tile):
                image.write_tile(rect[0], rect[1], tile)
                pool.give_back(tile)

            # Per tile: the radiance sums are rendered while the helper thread writes the previous
            # tile back and reads this tile's old bytes (a tile's read is queued behind every earlier
            # write, so a pass of one single tile reads what the pass before stored); then mean,
            # gamma, running average and quantisation on those bytes (rt_render's own sequence, taken
            # in two calls so that the file traffic hides behind the rendering).
            ahead = copier.submit(read, rects[0]) if running and rects else None
            writing = None
            for n, rect in enumerate(rects):
                tp = pass_params(params, rect, chunk, running, k)
                if running and k * chunk + chunk > 0xFFFFFFFF:
                    raise ValueError("sample index overflow")
                sums = backend.render_sums(tp, k * chunk if running else 0, chunk)
                if writing is not None:  # (before the next read is queued: the pool's third buffer comes back with it)
                    writing.result()
                old = ahead.result() if ahead is not None else pool.take(rect[3], rect[2])
                ahead = copier.submit(read, rects[n + 1]) if running and n + 1 < len(rects) else None
                new = backend.resolve(tp, sums, chunk, running, k, old if running else None, out=old)
                writing = copier.submit(write, rect, new)
            if writing is not None:
                writing.result()
        for ti in (mine if group_size > 1 else ()):
            rect = grid[ti]
            x, y, w, h = rect
            tp = pass_params(params, rect, chunk, running, k)
            if running:
                if k * chunk + chunk > 0xFFFFFFFF:
                    raise ValueError("sample index overflow")
                begin = k * chunk
            else:
                begin = 0
            if exact_in_group is not None:
                sums = _exact_tile_sums(backend, tp, begin, chunk, group_size, member, *exact_in_group)
            else:
                first, n = sample_share(begin, chunk, group_size, member)
                sums = backend.render_sums(tp, first, n) if n > 0 else backend.zeros(tp)
                reduce_in_group(sums)
            if member == (group_size - 1 if exact_in_group is not None else 0):
                old = image.read_tile(x, y, w, h) if running else None
                image.write_tile(x, y, backend.resolve(tp, sums, chunk, running, k, old))
        if end_of_pass is not None:
            end_of_pass()
        if progress is not None and rank == 0:
            progress(first_pass + (p + 1) * chunk)


class _Pinned:
    """Owner of one rt_host_alloc block."""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        try:
            self._lib.rt_host_free(self._ptr)
        except Exception:
            pass


class GpuTileBackend:
    """TileBackend over librt_mi355.so.  Radiance sums stay on the GPU: the
    library writes them into a torch CUDA tensor (what RCCL reduces) through its
    raw device pointer.  (raytracer_amd._ffi maps torch's own HIP runtime before the
    library, so both use one runtime whatever the import order.)

    Ordering: the library returns from every call with its own stream drained, so
    torch may use the tensor at once; in the other direction render_sums, fold and
    zeros first wait for whatever torch / RCCL have enqueued on the device
    (`settle`), so callbacks that are bare dist.reduce / dist.send / dist.recv
    (which return once enqueued) are safe with this backend."""

    def __init__(self, renderer, device=None):
        self.renderer = renderer
        self.device = device
        self._buffer = None
        self._pinned = []

    def pixel_buffer(self, h: int, w: int) -> np.ndarray:
        """An (h, w, 3) uint8 array in page-locked host memory (rt_host_alloc): tile copies to and
        from the GPU then run at the full rate of the host link instead of being staged by the
        runtime.  Ordinary memory if the allocation is refused."""
        lib = F.device_lib()
        n = h * w * 3
        ptr = lib.rt_host_alloc(n)
        if not ptr:
            return np.empty((h, w, 3), dtype=np.uint8)
        keeper = _Pinned(lib, ptr)
        arr = np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=np.uint8).reshape(h, w, 3)
        self._pinned.append((arr, keeper))  # released with the backend
        return arr

    def render_tile_pass(self, params, old, out=None):
        h, w = params.tile_shape()
        if old is not None:
            rgb8 = np.ascontiguousarray(old)
        elif out is not None:
            rgb8 = out
        else:
            rgb8 = np.zeros((h, w, 3), dtype=np.uint8)
        return self.renderer.render(params, rgb8)

    def _tensor(self, params):
        import torch

        h, w = params.tile_shape()
        n = h * w * 3
        if self._buffer is None or self._buffer.numel() < n:
            self._buffer = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._buffer[:n]

    def settle(self):
        import torch

        torch.cuda.synchronize(self.device)

    def render_sums(self, params, sample_begin, sample_count):
        t = self._tensor(params)
        self.settle()
        self.renderer.render_sums_device(params, sample_begin, sample_count, t.data_ptr())
        return t

    def zeros(self, params):
        t = self._tensor(params).zero_()
        self.settle()
        return t

    def sums(self, params):
        return self._tensor(params)

    def reserve(self, params, sample_count):
        return self.renderer.reserve_samples(params, sample_count)

    def trace(self, params, sample_begin, sample_count):
        return self.renderer.trace_samples(params, sample_begin, sample_count)

    def fold(self, params, carry_on):
        t = self._tensor(params)
        self.settle()
        self.renderer.fold_samples(params, carry_on, t.data_ptr(), sums_on_device=True)
        return t

    def resolve(self, params, sums, n_samples, running, k, old, out=None):
        h, w = params.tile_shape()
        if old is not None:
            rgb8 = np.ascontiguousarray(old)
        else:
            rgb8 = out if out is not None else np.zeros((h, w, 3), dtype=np.uint8)
        return self.renderer.resolve(params, sums.data_ptr(), n_samples, running_average=running, sample_iteration=k,
                                     rgb8=rgb8, sums_on_device=True)
