"""rt_trace_rays_noalloc (the entry point that allocates nothing: its queue, hit
records and stack overflow area live in a fixed array inside the library) against
rt_trace_rays, through the C ABI, bit for bit: twice in a row on one context, the
second call smaller, so that leftovers of the first call in the shared array would
show; then on a tree deep enough to use the overflow area, from a second context
on the same GPU (which shares the array), with device pointers, and at the limit."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import Renderer, RtError
from raytracer_amd.scene import HIT_DT

from . import scenes

pytestmark = pytest.mark.gpu


def rays(n, seed, lo, hi):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::13, 0] = 0.0
    return o, d


def same(a, b):
    return (np.array_equal(a["kind"], b["kind"]) and np.array_equal(a["prim"], b["prim"])
            and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32)))


def test_two_calls_in_a_row_equal_the_allocating_entry_point(renderer, asset_root):
    s = scenes.room(asset_root)
    renderer.upload(s)
    o1, d1 = rays(200_000, 5, [-2.9, 0.1, -3.9], [2.9, 3.1, 4.9])
    o2, d2 = rays(7_001, 6, [-2.9, 0.1, -3.9], [2.9, 3.1, 4.9])
    want1, want2 = renderer.trace_rays(o1, d1), renderer.trace_rays(o2, d2)
    got1 = renderer.trace_rays_noalloc(o1, d1)
    got2 = renderer.trace_rays_noalloc(o2, d2)  # smaller: the array still holds the first call's rays
    assert same(got1, want1) and same(got2, want2)
    assert (got1["kind"] == F.RT_SHAPE_TRIANGLE).mean() > 0.5
    assert same(got2, O.trace_rays(s, o2, d2))
    # and the allocating entry point is unchanged by having shared kernels with it
    assert same(renderer.trace_rays(o2, d2), want2)


def test_no_device_memory_is_allocated_or_released(renderer, asset_root):
    """The property the entry point exists for: after a warm-up call, calls of different
    sizes leave the GPU's free memory exactly where it was (hipMemGetInfo through torch)."""
    import torch

    s = scenes.room(asset_root)
    renderer.upload(s)
    o, d = rays(300_000, 9, [-2.9, 0.1, -3.9], [2.9, 3.1, 4.9])
    renderer.trace_rays_noalloc(o[:1000], d[:1000])  # warm-up: code object, host-side staging of the runtime
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info(0)[0]
    for n in (300_000, 17, 65_536, 300_000, 1):
        renderer.trace_rays_noalloc(o[:n], d[:n])
        assert torch.cuda.mem_get_info(0)[0] == before, n
    # the allocating entry point, for contrast, grows its workspace for the larger call
    fresh_o, fresh_d = rays(400_000, 10, [-2.9, 0.1, -3.9], [2.9, 3.1, 4.9])
    with Renderer(0) as other:
        other.upload(s)
        torch.cuda.synchronize()
        start = torch.cuda.mem_get_info(0)[0]
        other.trace_rays(fresh_o, fresh_d)
        assert torch.cuda.mem_get_info(0)[0] < start


def test_deep_tree_overflows_into_the_fixed_array_and_a_second_context_shares_it(renderer):
    s = scenes.deep_chain(n=60, w=16, h=16)
    o = np.tile(np.array([[-1.0, 0.0, -3.0]], np.float32), (4096, 1))
    d = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (4096, 1))
    d[:, 1] = np.linspace(-0.1, 0.1, 4096)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    renderer.upload(s)
    renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 2)  # nearly the whole stack goes to the overflow area
    try:
        want = renderer.trace_rays(o, d)
        assert same(renderer.trace_rays_noalloc(o, d), want)
        with Renderer(0) as other:  # same GPU, same array, used one after the other
            other.upload(s)
            other.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 2)
            assert same(other.trace_rays_noalloc(o[:1000], d[:1000]), want[:1000])
        assert same(renderer.trace_rays_noalloc(o[:33], d[:33]), want[:33])
    finally:
        renderer.set_option(F.RT_OPT_TRACE_LDS_SLOTS, 6)
    assert same(want, O.trace_rays(s, o, d))


def test_device_pointers_limit_and_errors(renderer):
    import torch

    s = scenes.many_spheres()  # sphere family traversed, planes too: three launches share the array
    renderer.upload(s)
    n = 524288  # the limit
    o, d = rays(n, 8, [-3, -3, -4], [3, 3, 6])
    want = renderer.trace_rays(o, d)
    t_o, t_d = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    t_out = torch.zeros(n * 3, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    lib = F.device_lib()
    rc = lib.rt_trace_rays_noalloc(renderer._ctx, C.c_void_p(t_o.data_ptr()), C.c_void_p(t_d.data_ptr()), n,
                                   C.c_void_p(t_out.data_ptr()), 1)
    assert rc == 0, lib.rt_last_error(renderer._ctx)
    got = t_out.cpu().numpy().view(HIT_DT).reshape(-1)
    assert same(got, want)
    assert same(renderer.trace_rays_noalloc(o, d), want)
    with pytest.raises(RtError, match="at most 524288"):
        renderer.trace_rays_noalloc(np.zeros((n + 1, 3), np.float32), np.ones((n + 1, 3), np.float32))
    assert renderer.trace_rays_noalloc(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)).size == 0
    with Renderer(0) as fresh:
        with pytest.raises(RtError) as e:
            fresh.trace_rays_noalloc(o[:4], d[:4])
        assert e.value.code == -5  # no scene uploaded
