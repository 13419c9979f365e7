"""Progressive rendering on a linear accumulator kept beside the image.

The reference's `incremental` / `continue` modes average the *stored bytes* of
the image with every new pass (/root/reference/src/lib.rs:286-292): gamma-space
8-bit values, re-truncated each time, so a long progressive render loses
precision and drifts dark.  This module is the alternative for hosts that want
live preview and resume without that loss: the running left fold of the samples
(lib.rs:274-279) is kept as 3 x f32 per pixel in a side file `<image>.acc`, every
pass carries the fold on (`rt_render_sums_onto`) and rewrites the image from the
total (`rt_resolve`).  After n samples the image bytes are exactly those of the
non-incremental render with n samples per pixel, however the n were split into
passes and across restarts.  It is an addition: `Scene.render` and `rt_render`
keep the reference's byte-averaging behaviour.

Side file: 32-byte header (magic, width, height, samples folded so far, a
"pass in progress" mark, the generator seed), then height x width x 3
little-endian f32.  A pass sets the mark before it touches the sums and clears
it with the new count afterwards; a file found with the mark set was cut off
mid-pass (some tiles ahead of the count) and is refused rather than resumed.
"""
from __future__ import annotations

import os
import struct
from typing import Callable

import numpy as np

from .tiles import Image, tile_grid

_MAGIC = b"RTACC1\0\0"
_HEADER = struct.Struct("<8sIIIIQ")  # magic, width, height, samples, pass in progress, seed
assert _HEADER.size == 32


class Accumulator:
    """The mapped side file.  `sums` is the (height, width, 3) f32 fold, `samples`
    the number of samples per pixel it holds."""

    def __init__(self, path: str, width: int, height: int, seed: int):
        self.path, self.width, self.height, self.seed = path, int(width), int(height), int(seed)
        size = _HEADER.size + self.width * self.height * 12
        fresh = not os.path.exists(path) or os.path.getsize(path) != size
        if not fresh:
            with open(path, "rb") as f:
                magic, w, h, _n, _r, s = _HEADER.unpack(f.read(_HEADER.size))
            fresh = (magic, w, h, s) != (_MAGIC, self.width, self.height, self.seed)
        if fresh:  # missing, or of another image: start over
            with open(path, "wb") as f:
                f.write(_HEADER.pack(_MAGIC, self.width, self.height, 0, 0, self.seed))
                f.truncate(size)
        self._map = np.memmap(path, dtype=np.uint8, mode="r+")
        self.sums = self._map[_HEADER.size:].view("<f4").reshape(self.height, self.width, 3)

    @property
    def samples(self) -> int:
        return _HEADER.unpack(bytes(self._map[:_HEADER.size]))[3]

    @property
    def in_progress(self) -> bool:
        return _HEADER.unpack(bytes(self._map[:_HEADER.size]))[4] != 0

    def mark(self, samples: int, in_progress: bool) -> None:
        """Writes the header and flushes the file (sums first, as they were written first)."""
        self._map.flush()
        self._map[:_HEADER.size] = np.frombuffer(
            _HEADER.pack(_MAGIC, self.width, self.height, int(samples), int(in_progress), self.seed), dtype=np.uint8)
        self._map.flush()

    def flush(self) -> None:
        self._map.flush()

    def close(self) -> None:
        self.flush()
        self.sums = None
        self._map = None


def render_progressive(renderer, image: Image, params, passes: int, samples_per_pass: int,
                       accumulator_path: str, tile: int | None = None,
                       progress: Callable[[int, int], None] | None = None) -> int:
    """Adds `passes` x `samples_per_pass` samples per pixel to the image at `image`,
    continuing whatever `accumulator_path` already holds for it.  After every pass the
    image file is a complete picture of all samples so far.  `tile`: edge of the square
    tiles a pass is cut into (None = the whole frame at once).  Returns the total
    number of samples per pixel now in the image.  `params` describes the whole frame
    (its tile_* and samples_per_pixel fields are set here)."""
    if passes < 0 or samples_per_pass < 1:
        raise ValueError("passes must be >= 0 and samples_per_pass >= 1")
    w, h = image.width, image.height
    if (params.width, params.height) != (w, h):
        raise ValueError("params describe another frame than the image")
    acc = Accumulator(accumulator_path, w, h, params.rng_seed)
    try:
        if acc.in_progress:
            raise RuntimeError(f"{accumulator_path} was cut off in the middle of a pass; delete it to start over")
        tiles = tile_grid(w, h, tile or w, tile or h)
        for k in range(passes):
            done = acc.samples
            acc.mark(done, True)
            for (x, y, tw, th) in tiles:
                p = params.copy()
                p.tile_x, p.tile_y, p.tile_w, p.tile_h = x, y, tw, th
                p.samples_per_pixel = samples_per_pass
                p.incremental, p.has_continue, p.continue_sampling = 0, 0, 0
                sums = np.ascontiguousarray(acc.sums[y:y + th, x:x + tw])
                renderer.render_sums_onto(p, done, samples_per_pass, sums)
                acc.sums[y:y + th, x:x + tw] = sums
                image.write_tile(x, y, renderer.resolve(p, sums, done + samples_per_pass))
            acc.mark(done + samples_per_pass, False)
            image.flush()
            if progress:
                progress(k + 1, acc.samples)
        return acc.samples
    finally:
        acc.close()
