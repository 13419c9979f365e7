"""Several GPUs behind one handle (rt_group_*): the image and the radiance sums of a group
are the single-GPU ones, bit for bit, whatever the number of members.  On a one-GPU box the
members are several contexts on GPU 0 (a GPU may be listed more than once), which runs the
same code: member threads, the hand-over of the sums from member to member, the fold order."""
import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd.renderer import RendererGroup, RtError

from . import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("members", [1, 2, 3])
def test_group_sums_and_bytes_equal_the_oracle(members, asset_root):
    s = scenes.room(asset_root, w=64, h=36, spp=7, bounces=12)  # 7 samples: uneven shares (3, 2, 2)
    with RendererGroup([0] * members) as g:
        assert len(g) == members
        g.upload(s)
        sums = g.render_sums(s.params, 0, 7)
        assert np.array_equal(bits(sums), bits(O.render_sums(s, s.params, 0, 7, threads=4)))
        assert np.array_equal(g.render(s.params), O.render(s, s.params, threads=4)[0])
        st = g.stats()
        assert st["paths"] == 2 * 64 * 36 * 7


def test_group_equals_one_context_with_pieces_and_incremental(renderer, asset_root):
    s = scenes.room(asset_root, w=48, h=32, spp=12, bounces=10, incremental=4)
    renderer.upload(s)
    seen = []
    one = renderer.render(s.params, progress=seen.append)
    with RendererGroup([0, 0]) as g:
        g.upload(s)
        g.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 1)  # a member's share of 2 samples does not fit at once
        got = []
        two = g.render(s.params, progress=got.append)
        assert got == seen == [4, 8, 12]
        assert np.array_equal(one, two)
        # fewer samples than members: the later members have nothing to do
        assert np.array_equal(bits(g.render_sums(s.params, 5, 1)), bits(renderer.render_sums(s.params, 5, 1)))


def test_group_takes_a_large_pass_in_rounds(renderer, asset_root):
    # 16 samples, three members holding two samples in flight each: rounds of 2+2+2, 2+2+2, 2+2+0
    s = scenes.room(asset_root, w=40, h=24, spp=16, bounces=8)
    renderer.upload(s)
    one = renderer.render_sums(s.params, 3, 16)
    with RendererGroup([0, 0, 0]) as g:
        g.upload(s)
        g.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 2)
        assert np.array_equal(bits(g.render_sums(s.params, 3, 16)), bits(one))
        assert g.stats()["paths"] == 40 * 24 * 16
        g.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
        assert np.array_equal(bits(g.render_sums(s.params, 3, 16)), bits(one))
    # the reserved room is what one trace call takes
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 2)
    assert renderer.reserve_samples(s.params, 9) == 2 and renderer.trace_samples(s.params, 0, 9) == 2
    renderer.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 0)
    assert renderer.reserve_samples(s.params, 9) == 9 and renderer.trace_samples(s.params, 0, 9) == 9
    with pytest.raises(RtError):
        renderer.reserve_samples(s.params, 0)


def test_group_errors():
    with pytest.raises(RtError):
        RendererGroup([])
    with pytest.raises(RtError):
        RendererGroup([0, 4096])  # no such GPU
    s = scenes.sphere_plane(w=16, h=8, spp=2)
    with RendererGroup([0]) as g:
        with pytest.raises(RtError):
            g.render_sums(s.params, 0, 2)  # no scene uploaded


def test_hand_over_of_a_full_frame_is_ordered_before_the_fold(renderer, asset_root):
    """A 1920x1080 frame is 24.9 MB of sums per hand-over: a copy between members that the
    receiving member's fold did not wait for would fold onto stale sums.  Three members, one
    sample in flight each, so that every sample's sums travel."""
    s = scenes.room(asset_root, w=1920, h=1080, spp=6, bounces=4)
    renderer.upload(s)
    one = renderer.render_sums(s.params, 0, 6)
    with RendererGroup([0, 0, 0]) as g:
        g.upload(s)
        g.set_option(F.RT_OPT_SAMPLES_IN_FLIGHT, 1)
        for _ in range(2):
            assert np.array_equal(bits(g.render_sums(s.params, 0, 6)), bits(one))
        assert np.array_equal(g.render(s.params), renderer.render(s.params))


def test_exceptions_inside_the_library_come_back_as_status_codes(renderer, asset_root):
    """The boundary's guard, exercised for real: an exception raised inside an exported call
    (RT_OPT_DEBUG_THROW: std::bad_alloc, then any other exception) returns a status."""
    with pytest.raises(RtError) as e:
        renderer.set_option(F.RT_OPT_DEBUG_THROW, 1)  # std::bad_alloc inside rt_set_option
    assert e.value.code == -4 and "memory" in str(e.value)
    with pytest.raises(RtError) as e:
        renderer.set_option(F.RT_OPT_DEBUG_THROW, 2)  # any other exception
    assert e.value.code == -3
    with RendererGroup([0, 0]) as g:
        with pytest.raises(RtError) as e:
            g.set_option(F.RT_OPT_DEBUG_THROW, 1)
        assert e.value.code == -4
    s = scenes.room(asset_root, w=192, h=108, spp=4, bounces=4)
    renderer.upload(s)
    # the context still renders afterwards
    assert renderer.render_sums(s.params, 0, 1).shape == (108, 192, 3)


@pytest.mark.parametrize("members", [2, 3])
def test_many_short_passes_are_dealt_out_as_bands_of_rows(renderer, asset_root, members):
    """incremental(1) over several GPUs: a pass of one sample cannot be shared out, its rows can.  The
    group cuts the frame into bands, each member renders groups of passes of the bands it pulls, and the
    bytes are the single-GPU (and the oracle's) pass-by-pass bytes; progress is reported per pass; a
    resumed render and a tile work the same."""
    s = scenes.room(asset_root, w=80, h=45, spp=6, bounces=10, incremental=1)
    cpu, _ = O.render(s, s.params, threads=4)
    with RendererGroup([0] * members) as g:
        g.upload(s)
        seen = []
        assert np.array_equal(g.render(s.params, progress=seen.append), cpu)
        assert seen == [1, 2, 3, 4, 5, 6]
        assert g.stats()["paths"] == 80 * 45 * 6
        more = scenes.room(asset_root, w=80, h=45, spp=4, bounces=10, incremental=2, cont=3)
        cpu2, _ = O.render(more, more.params, rgb8=cpu.copy(), threads=4)
        assert np.array_equal(g.render(more.params, rgb8=cpu.copy()), cpu2)
        t = s.params.copy()
        t.tile_x, t.tile_y, t.tile_w, t.tile_h = 8, 5, 50, 7  # fewer rows than bands
        cpu_t, _ = O.render(s, t, threads=4)
        assert np.array_equal(g.render(t), cpu_t)
