"""The CPU oracle against known answers.

The reference ships no tests or vectors, so these are answers derived by hand
from its formulas (cited per test) or recomputed here independently in numpy
binary32, step by step in the reference's operation order.  They pin the
oracle; the GPU is then compared with the oracle (test_gpu_parity.py).
"""
import json
import os

import numpy as np
import pytest

from oracle import binding as O
from raytracer_amd import _ffi as F
from raytracer_amd import make_params
from raytracer_amd.scene import MATERIAL_DT, Scene

from . import scenes

f32 = np.float32
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(x):
    return np.asarray(x, dtype=f32).view(np.uint32)


# ------------------------------------------------------------------ generator


def test_rng_known_answers():
    """src/rng.rs:19-41 against vectors computed with Python integers."""
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    a, b = O.Rng(), O.Rng()
    for d in kat["draws"]:
        assert a.u64() == int(d["u64"], 16)
        assert int(bits(b.f32())) == int(d["f32_bits"], 16)
    for j in kat["jump"]:
        r = O.Rng(int(j["state"], 16))
        assert r.u64() == int(j["next_u64"], 16)


def test_rng_state_is_a_weyl_counter():
    r = O.Rng()
    for _ in range(1000):
        r.u64()
    assert r.state.value == (F.RT_RNG_SEED + 1000 * 0x2D358DCCAA6C78A5) % 2 ** 64


def test_rng_f32_range():
    r = O.Rng()
    v = np.array([r.f32() for _ in range(20000)], dtype=f32)
    assert v.min() >= 0.0 and v.max() < 1.0
    assert abs(float(v.mean()) - 0.5) < 0.01


# --------------------------------------------------------------- intersections


def sphere(c=(0, 0, 0), r=1.0):
    return F.RtSphere((F.f32 * 3)(*c), r, 0)


def test_sphere_known_answers():
    """src/shapes.rs:60-87.  Unit sphere at the origin."""
    s = sphere()
    assert O.shape_hit(s, (0, 0, 5), (0, 0, -1)) == f32(4.0)      # t1 = -b - sqrt(disc) = 5 - 1
    assert O.shape_hit(s, (0, 0, 0), (0, 0, -1)) == f32(1.0)      # inside: t1 < 0, t2 = 1
    assert O.shape_hit(s, (0, 0, 5), (0, 0, 1)) is None           # behind the origin
    assert O.shape_hit(s, (0, 2, 5), (0, 0, -1)) is None          # disc = 0 - 29 + 1 < 0
    assert O.shape_hit(s, (0, 1, 5), (0, 0, -1)) == f32(5.0)      # tangent: disc = 0
    # a hit closer than MIN_DISTANCE = 0.001 is ignored, the far root is taken
    t = O.shape_hit(s, (0, 0, 1.0005), (0, 0, -1))
    assert t is not None and abs(float(t) - 2.0005) < 1e-5


def test_plane_known_answers():
    """src/shapes.rs:141-154."""
    p = F.RtPlane((F.f32 * 3)(0, 0, 0), (F.f32 * 3)(0, 1, 0), 0)
    assert O.shape_hit(p, (0, 1, 0), (0, -1, 0)) == f32(1.0)
    assert O.shape_hit(p, (0, -2, 0), (0, 1, 0)) == f32(2.0)     # two-sided
    assert O.shape_hit(p, (0, 1, 0), (1, 0, 0)) is None          # parallel: |den| < EPSILON
    assert O.shape_hit(p, (0, 1, 0), (0, 1, 0)) is None          # pointing away: t = -1
    assert O.shape_hit(p, (0, 0.0005, 0), (0, -1, 0)) is None    # t below MIN_DISTANCE


def triangle(a, b, c):
    a, b, c = (np.asarray(v, f32) for v in (a, b, c))
    return F.RtTriangle((F.f32 * 3)(*a), (F.f32 * 3)(*(b - a)), (F.f32 * 3)(*(c - a)),
                        F.RT_NO_INDEX, F.RT_NO_INDEX, F.RT_NO_INDEX, 0)


def test_triangle_known_answers():
    """src/shapes.rs:232-259."""
    t = triangle((0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert O.shape_hit(t, (0.25, 0.25, 3), (0, 0, -1)) == f32(3.0)
    assert O.shape_hit(t, (0.25, 0.25, -3), (0, 0, 1)) == f32(3.0)   # no back-face culling
    assert O.shape_hit(t, (0.75, 0.75, 3), (0, 0, -1)) is None       # u + v > 1
    assert O.shape_hit(t, (-0.1, 0.2, 3), (0, 0, -1)) is None        # u < 0
    assert O.shape_hit(t, (0.25, 0.25, 3), (1, 0, 0)) is None        # parallel: |det| < 2e-6
    assert O.shape_hit(t, (0, 0, 3), (0, 0, -1)) == f32(3.0)         # exactly through vertex a
    assert O.shape_hit(t, (0.5, 0.5, 3), (0, 0, -1)) == f32(3.0)     # on the hypotenuse: u + v == 1


def box(lo, hi):
    return F.RtNode((F.f32 * 3)(*lo), 0, (F.f32 * 3)(*hi), 0)


def test_slab_test_known_answers():
    """src/bvh.rs:25-43."""
    b = box((-1, -1, -1), (1, 1, 1))
    assert O.shape_hit(b, (0, 0, 5), (0, 0, -1)) == f32(4.0)
    # origin inside: tmin is negative and still reported (tmax > 0)
    assert O.shape_hit(b, (0, 0, 0), (0, 0, -1)) == f32(-1.0)
    assert O.shape_hit(b, (0, 0, 5), (0, 0, 1)) is None               # box behind
    # zero direction components: (min-o)/0 = -inf / +inf keep the slab open
    assert O.shape_hit(b, (0.5, 0.5, 5), (0, 0, -1)) == f32(4.0)
    assert O.shape_hit(b, (2, 0, 5), (0, 0, -1)) is None              # outside the x slab: +inf/+inf ordering
    # origin exactly on a slab plane with a zero direction component: 0/0 = NaN is
    # dropped by f32::min/max, which leaves the other quotient (-2/0 = -inf) as
    # both the near and the far value of that axis, so tmax = -inf: a miss
    assert O.shape_hit(b, (1, 0, 5), (0, 0, -1)) is None


def test_slab_matches_numpy_restatement():
    rng = np.random.default_rng(5)
    b = box((-1, -2, -0.5), (1.5, 0.25, 3))
    lo, hi = np.array([-1, -2, -0.5], f32), np.array([1.5, 0.25, 3], f32)
    for _ in range(2000):
        o = rng.uniform(-4, 4, 3).astype(f32)
        d = rng.normal(size=3).astype(f32)
        d /= np.sqrt((d * d).sum(dtype=f32))
        with np.errstate(all="ignore"):
            t1, t2 = (lo - o) / d, (hi - o) / d
        tmin = np.fmax(np.fmax(np.fmin(t1[0], t2[0]), np.fmin(t1[1], t2[1])), np.fmin(t1[2], t2[2]))
        tmax = np.fmin(np.fmin(np.fmax(t1[0], t2[0]), np.fmax(t1[1], t2[1])), np.fmax(t1[2], t2[2]))
        want = tmin if (tmax >= tmin and tmax > 0) else None
        got = O.shape_hit(b, o, d)
        assert (got is None) == (want is None)
        if want is not None:
            assert bits(got) == bits(want)


# -------------------------------------------------------------------- shading


def one_material_scene(kind, param=0.0, rgb=(0.5, 0.25, 1.0), texture=None):
    m = np.zeros(1, dtype=MATERIAL_DT)
    m["kind"], m["param"], m["rgb"] = kind, param, rgb
    texels = np.zeros(0, np.uint8)
    if texture is not None:
        m["color_kind"], m["tex_height"], m["tex_width"] = F.RT_COLOR_TEXTURE, texture.shape[0], texture.shape[1]
        texels = texture.reshape(-1)
    p = make_params((-1, 1, 1), (2, 0, 0), (0, -2, 0), (0, 0, 3), 4, 4, 1, 4)
    return Scene.from_shapes(p, materials=m, texels=texels)


def np_normalize(v):
    v = np.asarray(v, f32)
    length = np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    return (v / length).astype(f32)


def np_random_unit(rng):
    return np_normalize([rng.f32() - f32(0.5), rng.f32() - f32(0.5), rng.f32() - f32(0.5)])


def np_dot(a, b):
    return f32(f32(a[0] * b[0] + a[1] * b[1]) + a[2] * b[2])


def test_lambertian_scatter_matches_numpy_restatement():
    """src/material.rs:29-47 with the draws of src/vec3.rs:549-558."""
    s = one_material_scene(F.RT_MAT_LAMBERTIAN)
    n, p, d = np_normalize([0.2, 1.0, -0.1]), np.array([1, 2, 3], f32), np_normalize([0.3, -1, 0.2])
    a, b = O.Rng(12345), O.Rng(12345)
    for _ in range(50):
        kind, o_new, d_new = O.scatter(s, 0, d, n, p, a)
        want_d = np_normalize(n + np_random_unit(b))
        assert kind == 0
        assert np.array_equal(bits(o_new), bits(p + n * f32(1e-4)))
        assert np.array_equal(bits(d_new), bits(want_d))
        assert a.state.value == b.state.value  # exactly three draws each


def test_metal_scatter():
    """src/material.rs:48-66: mirror reflection draws nothing; fuzz draws three
    numbers and absorbs rays that end up below the surface."""
    n, p, d = np.array([0, 1, 0], f32), np.zeros(3, f32), np_normalize([1, -1, 0])
    r = O.Rng(7)
    kind, _, d_new = O.scatter(one_material_scene(F.RT_MAT_METAL, 0.0), 0, d, n, p, r)
    assert kind == 0 and r.state.value == 7
    want = d - (n * f32(2.0)) * np_dot(d, n)
    assert np.array_equal(bits(d_new), bits(want))
    s = one_material_scene(F.RT_MAT_METAL, 0.9)
    grazing = np_normalize([1, -0.05, 0])
    a, b = O.Rng(99), O.Rng(99)
    outcomes = set()
    for _ in range(200):
        kind, _, d_new = O.scatter(s, 0, grazing, n, p, a)
        refl = grazing - (n * f32(2.0)) * np_dot(grazing, n)
        want = np_normalize(refl + np_random_unit(b) * f32(0.9))
        absorbed = not (np_dot(want, n) > 0)
        assert kind == (1 if absorbed else 0)
        if not absorbed:
            assert np.array_equal(bits(d_new), bits(want))
        outcomes.add(kind)
    assert outcomes == {0, 1}


def test_glass_scatter():
    """src/material.rs:67-97: total internal reflection draws nothing; otherwise
    one draw decides between reflection and (un-normalised) refraction."""
    s = one_material_scene(F.RT_MAT_GLASS, 1.5)
    n, p = np.array([0, 1, 0], f32), np.zeros(3, f32)
    # leaving the glass at a grazing angle: eta = 1.5, sin > 1/1.5 -> total reflection
    d = np_normalize([1, 0.3, 0])
    r = O.Rng(3)
    kind, _, d_new = O.scatter(s, 0, d, n, p, r)
    assert kind == 0 and r.state.value == 3
    nn = -n
    assert np.array_equal(bits(d_new), bits(d - (nn * f32(2.0)) * np_dot(d, nn)))
    # entering head-on: reflectance r0 = ((1 - 1/1.5)/(1 + 1/1.5))^2 = 0.04
    d = np.array([0, -1, 0], f32)
    a, b = O.Rng(11), O.Rng(11)
    seen = set()
    for _ in range(300):
        kind, _, d_new = O.scatter(s, 0, d, n, p, a)
        u = b.f32()
        eta = f32(1.0) / f32(1.5)
        r0 = (f32(1.0) - eta) / (f32(1.0) + eta)
        r0 = f32(r0 * r0)
        if u < r0:  # cos = 1 -> (1-cos)^5 = 0 -> reflectance = r0
            want = d - (n * f32(2.0)) * np_dot(d, n)
            seen.add("reflect")
        else:
            want = (d + n * f32(1.0)) * eta + n * (-np.sqrt(f32(1.0) - eta * eta * f32(0.0)))
            seen.add("refract")
        assert kind == 0 and np.array_equal(bits(d_new), bits(want.astype(f32)))
        assert a.state.value == b.state.value
    assert seen == {"reflect", "refract"}


def test_light_terminates():
    r = O.Rng(1)
    kind, _, _ = O.scatter(one_material_scene(F.RT_MAT_LIGHT), 0, (0, -1, 0), (0, 1, 0), (0, 0, 0), r)
    assert kind == 2 and r.state.value == 1


def test_texture_sample_known_answers():
    """src/material.rs:192-239 on a 2x2 texture: rows are stored top first, v is
    flipped, coordinates wrap, texels are byte/255."""
    tex = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 0, 255], [255, 255, 255]]], np.uint8)
    s = one_material_scene(F.RT_MAT_LAMBERTIAN, texture=tex)
    assert np.array_equal(O.color_sample(s, 0, (0.0, 0.0)), f32([0, 0, 0]) + f32([0, 0, 1]))   # bottom-left
    # (0, 0.999..) wraps below 1: top-left texel (red)
    assert np.allclose(O.color_sample(s, 0, (0.0, 0.999999)), [1, 0, 0], atol=1e-5)
    # centre: average of the four texels
    assert np.allclose(O.color_sample(s, 0, (0.5, 0.5)), [0.5, 0.5, 0.5], atol=1e-6)
    # tiling: integer parts are dropped, negatives wrap upwards
    assert np.array_equal(bits(O.color_sample(s, 0, (3.25, -1.75))), bits(O.color_sample(s, 0, (0.25, 0.25))))
    # solid colours ignore the coordinates
    solid = one_material_scene(F.RT_MAT_LAMBERTIAN, rgb=(0.1, 0.2, 0.3))
    assert np.array_equal(O.color_sample(solid, 0, (0.3, 0.9)), f32([0.1, 0.2, 0.3]))


def test_texture_sample_matches_numpy_restatement():
    rng = np.random.default_rng(2)
    tex = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    s = one_material_scene(F.RT_MAT_LAMBERTIAN, texture=tex)
    h, w = 5, 7
    for _ in range(500):
        uv = rng.uniform(-3, 3, 2).astype(f32)
        c = []
        for e in uv:
            fr = f32(e - np.trunc(e))
            r = f32(np.fmod(fr, f32(1.0)))
            c.append(f32(r + f32(1.0)) if r < 0 else r)
        x, y = c[0], f32(f32(1.0) - c[1])
        ex, ey = f32(x * f32(w - 1)), f32(y * f32(h - 1))
        x0, y0 = int(np.floor(ex)), int(np.floor(ey))
        x1, y1 = min(x0 + 1, w - 1), min(y0 + 1, h - 1)
        dx, dy = f32(ex - np.floor(ex)), f32(ey - np.floor(ey))
        t = lambda xx, yy: tex[yy, xx].astype(f32) / f32(255.0)  # noqa: E731
        c0 = t(x0, y0) * f32(f32(1.0) - dx) + t(x1, y0) * dx
        c1 = t(x0, y1) * f32(f32(1.0) - dx) + t(x1, y1) * dx
        want = c0 * f32(f32(1.0) - dy) + c1 * dy
        assert np.array_equal(bits(O.color_sample(s, 0, uv)), bits(want.astype(f32)))


def test_triangle_surface_attributes():
    """src/shapes.rs:210-227,264-326: barycentric interpolation of vertex normals
    and texture coordinates; the geometric normal is never turned to the ray."""
    s = scenes.grid_wall(n=1)
    n, uv, m = O.surface_at(s, F.RT_SHAPE_TRIANGLE, 0, (0.5, -0.5, 0.0))
    t = s.triangles[0]
    want = np_normalize(np.cross(t["e1"].astype(f32), t["e2"].astype(f32)).astype(f32))
    assert np.array_equal(bits(n), bits(want)) and np.array_equal(uv, f32([0, 0])) and m == 0


# ---------------------------------------------------------------- render loop


def test_sky_and_bounce_limit():
    """src/lib.rs:313-345,374."""
    s = scenes.empty()
    d = np_normalize([0.3, 0.4, -1])
    a = f32(0.5) * f32(d[1] + f32(1.0))
    want = f32([0.2, 0.2, 0.8]) * f32(f32(1.0) - a) + f32([1, 1, 1]) * a
    assert np.array_equal(bits(O.ray_color(s, (0, 0, 0), d, 5, O.Rng())), bits(want.astype(f32)))
    assert np.array_equal(O.ray_color(s, (0, 0, 0), d, 0, O.Rng()), f32([0, 0, 0]))  # no bounce, no colour
    # one bounce on a Lambertian sphere: the throughput so far, not black
    sp = scenes.sphere_plane(bounces=1)
    c = O.ray_color(sp, (0, 1, 5), (0, 0, -1), 1, O.Rng())
    assert np.array_equal(c, f32([1, 0, 0]))


def test_quantisation_truncates_and_saturates():
    """src/lib.rs:283-295, src/convert.rs:53-55."""
    p = make_params((-1, 1, 1), (2, 0, 0), (0, -2, 0), (0, 0, 3), 2, 2, 1, 1)
    sums = np.array([[[0.0, 1.0, 4.0], [0.25, np.nan, -1.0]], [[np.inf, 0.999, 1e-9], [0.5, 0.5, 0.5]]], f32)
    out = O.resolve(p, sums, 1)
    # sqrt then *255 then truncation: sqrt(.25)*255 = 127.5 -> 127; NaN -> 0; 4 -> 510 -> 255
    assert out[0, 0].tolist() == [0, 255, 255]
    assert out[0, 1].tolist() == [127, 0, 0]
    assert out[1, 0, 0] == 255 and out[1, 0, 1] == int(np.sqrt(f32(0.999)) * f32(255))
    # running average with the stored byte: ((old/255)*k + new)/(k+1)
    old = np.full((2, 2, 3), 200, np.uint8)
    out = O.resolve(p, sums, 1, True, 3, old)
    want = (f32(200) / f32(255) * f32(3) + np.sqrt(f32(0.5))) / f32(4) * f32(255)
    assert out[1, 1, 0] == int(want)


def test_serial_mode_is_the_single_thread_order():
    """RT_RNG_SERIAL consumes one stream in the order of src/lib.rs:226-296 with
    one worker: passes outermost, pixels in index order, a pixel's samples
    consecutive, jitter x then y, then the path's draws."""
    s = scenes.sphere_plane(w=6, h=4, spp=4, bounces=5, prefix="incremental(2)\n")
    p = s.params.copy()
    p.rng_mode = F.RT_RNG_SERIAL
    img, _ = O.render(s, p)
    rng = O.Rng()
    want = np.zeros((4, 6, 3), np.uint8)
    row = f32([3.5556, 0, 0]) / f32(5)
    col = f32([0, -2, 0]) / f32(3)
    for k in range(2):
        for y in range(4):
            for x in range(6):
                acc = None
                for _ in range(2):
                    jx, jy = rng.f32(), rng.f32()
                    px = (f32([-1.7778, 2, 1]) + row * f32(f32(x) + jx / f32(2))) + col * f32(f32(y) + jy / f32(2))
                    d = np_normalize(px - f32([0, 1, 5]))
                    c = O.ray_color(s, (0, 1, 5), d, 5, rng)
                    acc = c if acc is None else (acc + c).astype(f32)
                want[y, x] = O.resolve(make_params((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), 2, 2, 1, 1),
                                       np.tile(acc, (2, 2, 1)), 2, True, k, np.tile(want[y, x], (2, 2, 1)))[0, 0]
    assert np.array_equal(img, want)


def test_substream_mode_is_thread_count_independent():
    s = scenes.materials(w=24, h=16, spp=3)
    a, _ = O.render(s, s.params, threads=1)
    b, _ = O.render(s, s.params, threads=5)
    assert np.array_equal(a, b)
    x = O.render_sums(s, s.params, 0, 3, threads=1)
    y = O.render_sums(s, s.params, 0, 3, threads=3)
    assert np.array_equal(bits(x), bits(y))


def test_substream_start_states():
    """Sample s of pixel p starts at seed + (s*W*H + p)*(2+3*max_bounces)*C0."""
    s = scenes.empty(w=5, h=3, spp=1, bounces=4)
    p = s.params
    sums = O.render_sums(s, p, 7, 1)
    row, col = f32([2, 0, 0]) / f32(4), f32([0, -2, 0]) / f32(2)
    for (x, y) in ((0, 0), (4, 2), (2, 1)):
        k = 7 * 15 + y * 5 + x
        r = O.Rng((F.RT_RNG_SEED + k * 14 * 0x2D358DCCAA6C78A5) % 2 ** 64)
        jx, jy = r.f32(), r.f32()
        px = (f32([-1, 1, 1]) + row * f32(f32(x) + jx / f32(2))) + col * f32(f32(y) + jy / f32(2))
        d = np_normalize(px - f32([0, 0, 3]))
        assert np.array_equal(bits(sums[y, x]), bits(O.ray_color(s, (0, 0, 3), d, 4, r)))


def test_nearest_hit_prefers_earlier_family_on_ties():
    """src/lib.rs:314-333: spheres, then planes, then triangles; `min_by` keeps
    the first of equal distances."""
    text = ("screen(-1 1 1, 2 0 0, 0 -2 0, 4, 4, 1, 2)\ncamera(0 0 3)\n"
            "spheres((0 0 -1, 1, lambertian, 1 0 0))\n"
            "planes((0 0 0, 0 0 1, lambertian, 0 1 0))\n"
            "triangles((-5 -5 0, 5 -5 0, 0 5 0, lambertian, 0 0 1))\n")
    from raytracer_amd import parse
    s = parse(text)
    # straight down the axis all three are hit at exactly t = 3
    h = O.trace_rays(s, [[0, 0, 3]], [[0, 0, -1]])[0]
    assert h["t"] == f32(3.0) and h["kind"] == F.RT_SHAPE_SPHERE
    # off the sphere: plane and triangle tie, the plane wins
    h = O.trace_rays(s, [[2, 0, 3]], [[0, 0, -1]])[0]
    assert h["t"] == f32(3.0) and h["kind"] == F.RT_SHAPE_PLANE
