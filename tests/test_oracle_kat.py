"""Hand-derived known answers for the leaf functions of the literal oracle
(oracle/oracle.cpp).  The reference ships no tests or vectors, so
these closed-form cases -- derived from the cited source lines -- are what pins
the oracle's leaves."""
import ctypes as C
import math

import numpy as np

import orc

_P = C.c_void_p


def arr(*v):
    return np.array(v, dtype=np.float64)


def prim_hit(kind, prm, o, d, time=0.0, t_min=0.001, t_max=math.inf):
    out = np.zeros(11)
    prm, o, d = arr(*prm), arr(*o), arr(*d)
    rc = orc.A.orc_prim_hit(kind, prm.ctypes.data_as(_P), o.ctypes.data_as(_P), d.ctypes.data_as(_P), time, t_min, t_max,
                            out.ctypes.data_as(_P))
    assert rc == 0
    return out


def test_sphere_roots_sphere_rs_31_48():
    # unit sphere at origin, ray from z=-3 along +z: roots 2 and 4; near root wins
    h = prim_hit(0, (0, 0, 0, 1), (0, 0, -3), (0, 0, 1))
    assert h[0] == 1 and h[1] == 2.0 and tuple(h[2:5]) == (0, 0, -1)
    assert tuple(h[5:8]) == (0, 0, -1) and h[10] == 1          # outward normal, front face
    # un-normalised direction: t scales (a = |d|^2 = 4)
    assert prim_hit(0, (0, 0, 0, 1), (0, 0, -3), (0, 0, 2))[1] == 1.0
    # near root below t_min -> far root (ray starts inside): back face, normal flipped against the ray
    h = prim_hit(0, (0, 0, 0, 1), (0, 0, 0), (0, 0, 1))
    assert h[1] == 1.0 and h[10] == 0 and tuple(h[5:8]) == (0, 0, -1)
    # t_max clips: both roots beyond
    assert prim_hit(0, (0, 0, 0, 1), (0, 0, -3), (0, 0, 1), t_max=1.5)[0] == 0
    # root == t_max is accepted (`t_max < root` rejects), sphere.rs:43
    assert prim_hit(0, (0, 0, 0, 1), (0, 0, -3), (0, 0, 1), t_max=2.0)[0] == 1
    # miss: discriminant < 0
    assert prim_hit(0, (0, 0, 0, 1), (0, 2, -3), (0, 0, 1))[0] == 0


def test_sphere_uv_math_rs_67_71():
    out = np.zeros(2)
    for p, uv in (((1, 0, 0), (0.5, 0.5)), ((0, 1, 0), (0.5, 1.0)), ((0, -1, 0), (0.5, 0.0)),
                  ((0, 0, 1), (0.25, 0.5)), ((0, 0, -1), (0.75, 0.5)),
                  # -z = -0.0: atan2(-0.0, -1) = -pi -> u = 0; z = -0.0 gives +pi -> u = 1 (IEEE signed zero)
                  ((-1, 0, 0.0), (0.0, 0.5)), ((-1, 0, -0.0), (1.0, 0.5))):
        orc.A.orc_sphere_uv(arr(*p).ctypes.data_as(_P), out.ctypes.data_as(_P))
        assert abs(out[0] - uv[0]) < 1e-15 and abs(out[1] - uv[1]) < 1e-15, (p, out)


def test_rect_hits_aarect_rs():
    # XY rect at z=5, [0,2]x[0,4]
    h = prim_hit(1, (0, 2, 0, 4, 5), (1, 1, 0), (0, 0, 1))
    assert h[0] == 1 and h[1] == 5.0 and tuple(h[2:5]) == (1, 1, 5) and (h[8], h[9]) == (0.5, 0.25)
    assert tuple(h[5:8]) == (0, 0, -1) and h[10] == 0          # ray along +z hits the back of normal +z
    assert prim_hit(1, (0, 2, 0, 4, 5), (3, 1, 0), (0, 0, 1))[0] == 0        # outside x
    assert prim_hit(1, (0, 2, 0, 4, 5), (2, 4, 0), (0, 0, 1))[0] == 1        # edges inclusive
    # XZ rect (the Cornell light), hit from below: front face is the -y side only via FlipFace
    h = prim_hit(2, (213, 343, 227, 332, 554), (278, 0, 280), (0, 1, 0))
    assert h[0] == 1 and h[1] == 554.0 and tuple(h[5:8]) == (0, -1, 0) and h[10] == 0
    # YZ rect
    h = prim_hit(3, (0, 555, 0, 555, 555), (0, 100, 200), (1, 0, 0))
    assert h[0] == 1 and h[1] == 555.0 and (h[8], h[9]) == (100 / 555, 200 / 555)
    # parallel ray: t = inf or nan -> miss
    assert prim_hit(1, (0, 2, 0, 4, 5), (1, 1, 0), (1, 0, 0))[0] == 0


def test_moving_sphere_center_is_unclamped():
    # moving_sphere.rs:23-26: centre extrapolates linearly in ray.time (quirk Q1 feeds it time = hit t)
    prm = (0, 0, 0, 0, 1, 0, 0.0, 1.0, 0.5)
    assert prim_hit(4, prm, (0, 0, -3), (0, 0, 1), time=0.0)[1] == 2.5
    h = prim_hit(4, prm, (0, 3, -3), (0, 0, 1), time=3.0)      # centre at y = 3
    assert h[0] == 1 and h[1] == 2.5


def aabb(mn, mx, o, d, t_min=0.001, t_max=math.inf):
    return orc.A.orc_aabb_hit(arr(*mn).ctypes.data_as(_P), arr(*mx).ctypes.data_as(_P), arr(*o).ctypes.data_as(_P),
                              arr(*d).ctypes.data_as(_P), t_min, t_max)


def test_aabb_slabs_aabb_rs_13_32():
    assert aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1)) == 1      # zero components: inv_d = inf, inside slabs
    assert aabb((0, 0, 0), (1, 1, 1), (1.5, 0.5, -1), (0, 0, 1)) == 0      # outside the x slab with d.x = 0
    assert aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, 2), (0, 0, -1)) == 1      # negative direction swaps
    assert aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), t_max=0.5) == 0
    assert aabb((0, 0, 0), (1, 1, 1), (0.5, 0.5, -1), (0, 0, 1), t_max=1.0) == 0   # t_max <= t_min is a miss
    # origin exactly on a slab plane with d = 0 there: (0 * inf) = NaN, comparisons false -> slab ignored
    assert aabb((0, 0, 0), (1, 1, 1), (0.0, 0.5, -1), (0, 0, 1)) == 1


def test_schlick_reflectance_material_rs_121_125():
    r0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    assert orc.A.orc_reflectance(1.0, 1.5) == r0
    assert orc.A.orc_reflectance(0.0, 1.5) == 1.0
    c = 0.3
    assert abs(orc.A.orc_reflectance(c, 1.5) - (r0 + (1 - r0) * (1 - c) ** 5)) < 1e-16


def test_reflect_refract():
    out = np.zeros(3)
    orc.A.orc_reflect(arr(1, -1, 0).ctypes.data_as(_P), arr(0, 1, 0).ctypes.data_as(_P), out.ctypes.data_as(_P))
    assert tuple(out) == (1, 1, 0)
    # normal incidence goes straight through
    orc.A.orc_refract(arr(0, -1, 0).ctypes.data_as(_P), arr(0, 1, 0).ctypes.data_as(_P), 1 / 1.5, out.ctypes.data_as(_P))
    assert tuple(out) == (0, -1, 0)
    # Snell: sin(t) = sin(i)/1.5
    i = math.radians(40)
    orc.A.orc_refract(arr(math.sin(i), -math.cos(i), 0).ctypes.data_as(_P), arr(0, 1, 0).ctypes.data_as(_P), 1 / 1.5,
                      out.ctypes.data_as(_P))
    assert abs(out[0] - math.sin(i) / 1.5) < 1e-15 and abs(np.linalg.norm(out) - 1) < 1e-15


def test_onb_branch_onb_rs_15_19():
    out = np.zeros(9)
    orc.A.orc_onb(arr(0, 0, 2).ctypes.data_as(_P), out.ctypes.data_as(_P))      # |w.x| <= 0.9 -> a = x axis
    u, v, w = out[0:3], out[3:6], out[6:9]
    assert tuple(w) == (0, 0, 1) and tuple(v) == (0, 1, 0) and tuple(u) == (-1, 0, 0)
    orc.A.orc_onb(arr(1, 0, 0).ctypes.data_as(_P), out.ctypes.data_as(_P))      # |w.x| > 0.9 -> a = y axis
    u, v, w = out[0:3], out[3:6], out[6:9]
    assert tuple(w) == (1, 0, 0) and tuple(v) == (0, 0, 1) and tuple(u) == (0, -1, 0)


def test_light_pdfs_closed_form():
    # XZRect::pdf_value aarect.rs:119-138: straight up from below the centre: d^2 / (cos * A), cos = 1
    prm = arr(213, 343, 227, 332, 554)
    o, v = arr(278, 54, 279.5), arr(0, 1, 0)
    got = orc.A.orc_light_pdf_value(2, prm.ctypes.data_as(_P), o.ctypes.data_as(_P), v.ctypes.data_as(_P))
    assert got == 500.0 ** 2 / (130.0 * 105.0)
    # unnormalised v gives the same density (t shrinks, |v| grows)
    v2 = arr(0, 4, 0)
    assert orc.A.orc_light_pdf_value(2, prm.ctypes.data_as(_P), o.ctypes.data_as(_P), v2.ctypes.data_as(_P)) == got
    # direction that misses the rectangle
    vm = arr(1, 0.1, 0)
    assert orc.A.orc_light_pdf_value(2, prm.ctypes.data_as(_P), o.ctypes.data_as(_P), vm.ctypes.data_as(_P)) == 0.0
    # Sphere::pdf_value sphere.rs:72-90: 1 / (2 pi (1 - sqrt(1 - r^2/d^2)))
    sp = arr(190, 90, 190, 90)
    o = arr(190, 90, 490)
    v = arr(0, 0, -1)
    exp = 1.0 / (2.0 * math.pi * (1.0 - math.sqrt(1.0 - 90.0 * 90.0 / (300.0 * 300.0))))
    got = orc.A.orc_light_pdf_value(0, sp.ctypes.data_as(_P), o.ctypes.data_as(_P), v.ctypes.data_as(_P))
    assert abs(got - exp) <= 2e-16 * exp
    # origin inside the sphere: sqrt of a negative -> NaN (reference quirk Q15)
    oi = arr(190, 90, 200)
    assert math.isnan(orc.A.orc_light_pdf_value(0, sp.ctypes.data_as(_P), oi.ctypes.data_as(_P), v.ctypes.data_as(_P)))


def test_into_sampled_and_quantiser_color_rs():
    rt = orc.rt()
    sums = np.array([[4.0, float("nan"), 0.0], [float("inf"), -1.0, 2.0]])
    m = rt.resolve(sums, 4)                              # NaN scrub on the SUM, then * 1/spp
    assert m.tolist() == [[1.0, 0.0, 0.0], [float("inf"), -0.25, 0.5]]
    q = rt.quantize(np.array([0.0, 1.0, 4.0, 0.25, float("nan"), -1.0, float("inf"), 0.999 ** 2, 1e-300]))
    # sqrt -> clamp [0, 0.999] -> *256 -> truncate; sqrt(-1) = NaN -> 0
    assert q.tolist() == [0, 255, 255, 128, 0, 0, 255, 255, 0]
    # oracle's own quantiser agrees with the library's on a dense sweep
    x = np.linspace(0, 1.2, 100001)
    qo = np.empty(x.size, dtype=np.uint8)
    orc.A.orc_quantize(x.ctypes.data_as(_P), C.c_uint64(x.size), qo.ctypes.data_as(_P))
    assert np.array_equal(qo, rt.quantize(x))


def test_camera_ray_camera_rs_61_73():
    sc = orc.OracleScene(5, 1)
    out = np.zeros(7)
    orc.A.orc_camera_ray(sc._h, 0.5, 0.5, 1234, 0, out.ctypes.data_as(_P))
    # aperture 0: origin = look_from; centre of the viewport looks down +z (look_at - look_from), length focus_dist = 10
    assert tuple(out[0:3]) == (278.0, 278.0, -800.0)
    assert abs(out[3]) < 1e-12 and abs(out[4]) < 1e-12 and abs(out[5] - 10.0) < 1e-12
    assert 0.0 <= out[6] < 1.0
    # corner (0,0) is the lower-left: vfov 40 deg, aspect 1 -> half extent = 10 * tan(20 deg)
    orc.A.orc_camera_ray(sc._h, 0.0, 0.0, 1234, 0, out.ctypes.data_as(_P))
    half = 10.0 * math.tan(math.radians(20.0))
    # u = vup x w with w = -z: u = -x ... lower-left is at +x? cgmath: u = vup.cross(w) = (0,1,0)x(0,0,-1) = (-1,0,0)
    assert abs(out[3] - half) < 1e-12 and abs(out[4] + half) < 1e-12
