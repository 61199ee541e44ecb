"""Known answers, ON THE DEVICE, for the parts of the path that no artefact of the reference pins pixel for pixel (round-2 review:
the Cornell PNG reaches neither ConstantMedium nor the Checker / Noise / Image textures nor sphere_uv's seam).  Closed forms:
  * ConstantMedium's free flight is exponential: hit_distance = -(1/density) ln(xi) (constant_medium.rs:85-91) -- a
    Kolmogorov-Smirnov test over 10^6 rays traced by the product's walk through one big medium;
  * Perlin::noise vanishes at lattice points (perlin.rs:46-72: every weight vector has a zero component product there), is bounded
    by sqrt(3)/2, and turb is |sum 2^-i noise(2^i p)| (perlin.rs:74-86), recomputed here from the device's own noise values;
  * CheckerTexture picks by the sign of sin(10x) sin(10y) sin(10z) (texture.rs:46-55);
  * sphere_uv at the poles and on both sides of the seam (math.rs:67-71), against the literal oracle's values bit for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import orc

sys.path.insert(0, os.path.join(orc.ROOT, "tools"))


def _medium_scene(rt, density, radius=1000.0):
    sc = rt.Scene(1)
    white = sc.solid_color((0.9, 0.9, 0.9))
    boundary = sc.sphere((0.0, 0.0, 0.0), radius, sc.dielectric(1.5))
    medium = sc.constant_medium(boundary, density, white)
    sc.set_world(sc.bvh_node([medium]))
    sc.set_lights([])
    sc.set_background((0.5, 0.5, 0.5))
    sc.set_camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    sc.commit()
    return sc


@pytest.mark.gpu
def test_constant_medium_free_flight_is_exponential_on_the_device(rt, gpu_ctx_factory):
    import walk_lab as wl
    density = 0.05
    ctx = gpu_ctx_factory(_medium_scene(rt, density))
    lab = wl.Lab(ctx)
    n = 1_000_000
    g = np.random.default_rng(12345)
    d = g.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    length = np.where(np.arange(n) % 2 == 0, 1.0, 2.5)           # unnormalised directions too: the flight is in world units
    rays = np.zeros((n, 8))
    rays[:, 3:6] = d * length[:, None]
    lab.set_rays(rays)
    r = lab.trace(0, repeats=1)
    assert (r["prim"] != 0xFFFFFFFF).all()                       # exp(-50): nothing leaves a medium 50 mean free paths deep
    # constant_medium.rs:78-103: the ray starts inside, rec1.t = max(t1, t_min = 0.001) -> t = 0.001 + hit_distance / |d|
    flight = (r["t"] - 0.001) * length
    assert (flight > 0).all()
    x = np.sort(flight)
    cdf = 1.0 - np.exp(-density * x)
    emp_hi = np.arange(1, n + 1) / n
    emp_lo = np.arange(0, n) / n
    ks = max(np.abs(emp_hi - cdf).max(), np.abs(emp_lo - cdf).max())
    print("KS statistic", ks, "mean flight", flight.mean(), "expected", 1 / density)
    assert ks < 1.95 / np.sqrt(n)                                # alpha ~ 0.001
    assert abs(flight.mean() - 1 / density) < 5 * (1 / density) / np.sqrt(n)
    # the two direction lengths separately (the division by ray_length, constant_medium.rs:81,103)
    for sel in (length == 1.0, length == 2.5):
        assert abs(flight[sel].mean() - 1 / density) < 5 * (1 / density) / np.sqrt(sel.sum())
    lab.close()


@pytest.mark.gpu
def test_perlin_noise_checker_and_sphere_uv_known_answers_on_the_device(rt, gpu_ctx_factory):
    sc = rt.Scene(7)
    a, b = sc.solid_color((0.2, 0.3, 0.1)), sc.solid_color((0.9, 0.9, 0.9))
    checker = sc.checker_texture(a, b)                           # CheckerTexture { odd: a, even: b }
    noise = sc.noise_texture(4.0)
    sc.set_world(sc.bvh_node([sc.sphere((0, -1000, 0), 1000, sc.lambertian(checker)), sc.sphere((0, 2, 0), 2, sc.lambertian(noise))]))
    sc.set_lights([])
    sc.set_background((0.7, 0.8, 1.0))
    sc.set_camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 1.5, 0.0, 10.0, 0.0, 1.0)
    sc.commit()
    ctx = gpu_ctx_factory(sc)
    g = np.random.default_rng(99)

    # Perlin::noise at lattice points, negative and beyond the 256-period included
    lattice = g.integers(-600, 600, size=(4096, 3)).astype(np.float64)
    out = ctx.debug_texture(1, 0, np.concatenate([np.zeros((4096, 2)), lattice], axis=1))
    assert (out[:, 0] == 0.0).all() and (out[:, 1] == 0.0).all()   # noise and turb (every octave is a lattice point again)
    # range, and periodicity 256 of the permutation tables (perlin.rs:52-62: i & 255)
    p = g.uniform(-300, 300, size=(100000, 3))
    o1 = ctx.debug_texture(1, 0, np.concatenate([np.zeros((len(p), 2)), p], axis=1))
    assert np.abs(o1[:, 0]).max() <= np.sqrt(3) / 2 + 1e-12 and np.abs(o1[:, 0]).max() > 0.3
    pi = np.floor(g.uniform(-100, 100, size=(20000, 3))) + g.integers(0, 64, size=(20000, 3)) / 64.0   # exact binary fractions: + 256 is exact
    n0 = ctx.debug_texture(1, 0, np.concatenate([np.zeros((len(pi), 2)), pi], axis=1))[:, 0]
    n256 = ctx.debug_texture(1, 0, np.concatenate([np.zeros((len(pi), 2)), pi + np.array([256.0, -512.0, 768.0])], axis=1))[:, 0]
    assert np.array_equal(n0, n256)
    # turb = |sum_i 2^-i noise(2^i p)| over 7 octaves, recomputed from the device's own noise in the reference's order
    q = p[:5000]
    acc = np.zeros(len(q))
    w = 1.0
    t = q.copy()
    for _ in range(7):
        acc = acc + w * ctx.debug_texture(1, 0, np.concatenate([np.zeros((len(t), 2)), t], axis=1))[:, 0]
        w *= 0.5
        t = t * 2.0
    assert np.array_equal(np.abs(acc), o1[:5000, 1])
    # NoiseTexture::value = 0.5 (1 + sin(scale z + 10 turb)) (texture.rs:57-65), within the contract's 1 ulp of sin
    val = ctx.debug_texture(0, noise, np.concatenate([np.zeros((len(q), 2)), q], axis=1))
    want = 0.5 * (1.0 + np.sin(4.0 * q[:, 2] + 10.0 * o1[:5000, 1]))
    assert np.abs(val[:, 0] - want).max() < 1e-15 * 4 and np.array_equal(val[:, 0], val[:, 1]) and np.array_equal(val[:, 0], val[:, 2])

    # CheckerTexture: sines < 0 -> odd (texture.rs:46-55); points a safe distance from the zero surfaces of the product
    pc = g.uniform(-20, 20, size=(200000, 3))
    s = np.sin(10 * pc[:, 0]) * np.sin(10 * pc[:, 1]) * np.sin(10 * pc[:, 2])
    keep = np.abs(s) > 1e-9
    got = ctx.debug_texture(0, checker, np.concatenate([np.zeros((len(pc), 2)), pc], axis=1))
    odd, even = np.array([0.2, 0.3, 0.1]), np.array([0.9, 0.9, 0.9])
    assert keep.sum() > 190000
    assert np.array_equal(got[keep], np.where((s[keep] < 0)[:, None], odd, even))

    # sphere_uv: poles, seam (both signs of zero), and random points -- the literal oracle's bits
    pts = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0.0], [-1, 0, -0.0], [0, 0, 1], [0, 0, -1], [0.6, 0.0, 0.8], [-0.6, 0.8, -0.0]], dtype=np.float64)
    rnd = g.normal(size=(2000, 3))
    rnd /= np.linalg.norm(rnd, axis=1, keepdims=True)
    pts = np.concatenate([pts, rnd])
    dev = ctx.debug_texture(2, 0, np.concatenate([np.zeros((len(pts), 2)), pts], axis=1))[:, :2]
    orc.A.orc_sphere_uv.argtypes = [C.c_void_p, C.c_void_p]
    lit = np.empty((len(pts), 2))
    for i in range(len(pts)):
        orc.A.orc_sphere_uv(pts[i].ctypes.data_as(C.c_void_p), lit[i].ctypes.data_as(C.c_void_p))
    assert np.array_equal(dev.view(np.uint64), lit.view(np.uint64))
    assert dev[0].tolist() == [0.5, 1.0] or dev[0][1] == 1.0          # north pole: v = acos(-1)/pi = 1
    assert dev[1][1] == 0.0                                             # south pole
    assert dev[3][0] == 0.0 and dev[4][0] == 1.0                        # the seam: atan2(-0, -1) = -pi -> u = 0; atan2(+0, -1) = pi -> u = 1
