"""Builds the same scene, call for call, through the product's C ABI (rt.Scene) and in the literal oracle
(oracle.cpp's generic builder), and a seeded random scene-graph generator on top.  Test infrastructure."""
import ctypes as C

import numpy as np

import orc

_P = C.c_void_p
A = orc.A
A.orcb_new.restype = _P
A.orcb_new.argtypes = [C.c_uint64]
A.orcb_free.argtypes = [_P]
A.orcb_finish.restype = _P
A.orcb_finish.argtypes = [_P]
A.orcb_rng_f64.restype = C.c_double
A.orcb_rng_f64.argtypes = [_P]
A.orcb_rng_range.restype = C.c_double
A.orcb_rng_range.argtypes = [_P, C.c_double, C.c_double]
for name, args in {
    "orcb_tex_solid": [_P, C.c_double, C.c_double, C.c_double], "orcb_tex_checker": [_P, C.c_int, C.c_int],
    "orcb_tex_noise": [_P, C.c_double], "orcb_tex_image": [_P, _P, C.c_uint32, C.c_uint32],
    "orcb_mat_lambertian": [_P, C.c_int], "orcb_mat_metal": [_P, C.c_double, C.c_double, C.c_double, C.c_double],
    "orcb_mat_dielectric": [_P, C.c_double], "orcb_mat_diffuse_light": [_P, C.c_int], "orcb_mat_null": [_P],
    "orcb_sphere": [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int],
    "orcb_moving_sphere": [_P, _P, _P, C.c_double, C.c_double, C.c_double, C.c_int],
    "orcb_rect": [_P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int],
    "orcb_aabox": [_P, _P, _P, C.c_int], "orcb_translate": [_P, C.c_int, C.c_double, C.c_double, C.c_double],
    "orcb_rotate_y": [_P, C.c_int, C.c_double], "orcb_flip_face": [_P, C.c_int],
    "orcb_constant_medium": [_P, C.c_int, C.c_double, C.c_int], "orcb_bvh": [_P, C.POINTER(C.c_int), C.c_uint32],
}.items():
    getattr(A, name).restype = C.c_int
    getattr(A, name).argtypes = args
A.orcb_set_world.argtypes = [_P, C.c_int]
A.orcb_set_lights.argtypes = [_P, C.POINTER(C.c_int), C.c_uint32]
A.orcb_set_background.argtypes = [_P, C.c_double, C.c_double, C.c_double]
A.orcb_set_camera.argtypes = [_P, _P, _P, _P] + [C.c_double] * 6


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleBuilt(orc.OracleScene):
    def __init__(self, handle):
        self._h = handle
        self.defaults = None
        self.earth = None


class Dual:
    """Every method returns a (product id, oracle id) pair."""

    def __init__(self, build_seed, best_axis=True):
        self.rt = orc.rt()
        self.p = self.rt.Scene(build_seed)
        self.o = A.orcb_new(build_seed)
        self.best_axis = best_axis  # the product's default build (RT1W_BVH_BEST_AXIS); False: both sides draw the axes (RT1W_BVH_REFERENCE)
        A.orc_set_bvh_axis_rule(1 if best_axis else 0)
        self._img = []

    def rng_f64(self):
        a, b = self.p.rng_f64(), A.orcb_rng_f64(self.o)
        assert a == b
        return a

    def solid(self, c): return (self.p.solid_color(c), A.orcb_tex_solid(self.o, *map(float, c)))
    def checker(self, odd, even): return (self.p.checker_texture(odd[0], even[0]), A.orcb_tex_checker(self.o, odd[1], even[1]))
    def noise(self, scale): return (self.p.noise_texture(scale), A.orcb_tex_noise(self.o, scale))

    def image(self, rgb8):
        rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
        self._img.append(rgb8)
        h, w = rgb8.shape[:2]
        return (self.p.image_texture(rgb8), A.orcb_tex_image(self.o, rgb8.ctypes.data_as(_P), w, h))

    def lambertian(self, t): return (self.p.lambertian(t[0]), A.orcb_mat_lambertian(self.o, t[1]))
    def metal(self, c, fuzz): return (self.p.metal(c, fuzz), A.orcb_mat_metal(self.o, float(c[0]), float(c[1]), float(c[2]), fuzz))
    def dielectric(self, ir): return (self.p.dielectric(ir), A.orcb_mat_dielectric(self.o, ir))
    def diffuse_light(self, t): return (self.p.diffuse_light(t[0]), A.orcb_mat_diffuse_light(self.o, t[1]))
    def null_material(self): return (self.p.null_material(), A.orcb_mat_null(self.o))
    def sphere(self, c, r, m): return (self.p.sphere(c, r, m[0]), A.orcb_sphere(self.o, float(c[0]), float(c[1]), float(c[2]), r, m[1]))

    def moving_sphere(self, c0, c1, t0, t1, r, m):
        return (self.p.moving_sphere(c0, c1, t0, t1, r, m[0]), A.orcb_moving_sphere(self.o, _d3(c0), _d3(c1), t0, t1, r, m[1]))

    def rect(self, axis, a0, a1, b0, b1, k, m):
        f = (self.p.xy_rect, self.p.xz_rect, self.p.yz_rect)[axis]
        return (f(a0, a1, b0, b1, k, m[0]), A.orcb_rect(self.o, axis, a0, a1, b0, b1, k, m[1]))

    def aabox(self, p0, p1, m): return (self.p.aabox(p0, p1, m[0]), A.orcb_aabox(self.o, _d3(p0), _d3(p1), m[1]))
    def translate(self, h, off): return (self.p.translate(h[0], off), A.orcb_translate(self.o, h[1], float(off[0]), float(off[1]), float(off[2])))
    def rotate_y(self, h, deg): return (self.p.rotate_y(h[0], deg), A.orcb_rotate_y(self.o, h[1], deg))
    def flip_face(self, h): return (self.p.flip_face(h[0]), A.orcb_flip_face(self.o, h[1]))
    def constant_medium(self, h, d, t): return (self.p.constant_medium(h[0], d, t[0]), A.orcb_constant_medium(self.o, h[1], d, t[1]))

    def bvh(self, hs):
        arr = (C.c_int * len(hs))(*[h[1] for h in hs])
        oid = A.orcb_bvh(self.o, arr, len(hs))
        assert oid >= 0
        return (self.p.bvh_node([h[0] for h in hs]), oid)

    def finish(self, world, lights, background, cam):
        self.p.set_world(world[0])
        self.p.set_lights([l[0] for l in lights])
        self.p.set_background(background)
        self.p.set_camera(*cam)
        self.p.commit()
        if not self.best_axis:
            self.p.set_bvh_build("reference")
        A.orcb_set_world(self.o, world[1])
        arr = (C.c_int * max(1, len(lights)))(*[l[1] for l in lights])
        A.orcb_set_lights(self.o, arr, len(lights))
        A.orcb_set_background(self.o, *map(float, background))
        A.orcb_set_camera(self.o, _d3(cam[0]), _d3(cam[1]), _d3(cam[2]), *[float(x) for x in cam[3:]])
        h = A.orcb_finish(self.o)
        A.orc_set_bvh_axis_rule(0)
        self.o = None
        return self.p, OracleBuilt(h)


def random_scene_pair(seed, n_objects=None, best_axis=True):
    """A random but well-formed scene graph: every primitive, wrapper nesting up to 3, media with sphere / box /
    wrapped-box boundaries, nested BVHs (incl. 1- and 2-element ones), textures of every kind, 0-3 lights incl. kinds
    that keep the trait defaults."""
    g = np.random.default_rng(seed)
    d = Dual(int(g.integers(1, 1 << 30)), best_axis)
    img = g.integers(0, 256, (8, 16, 3), dtype=np.uint8)

    def tex(depth=0):
        k = g.integers(0, 4 if depth < 2 else 1)
        if k == 0: return d.solid(g.uniform(0.05, 0.95, 3))
        if k == 1: return d.checker(tex(depth + 1), tex(depth + 1))
        if k == 2: return d.noise(float(g.uniform(0.5, 5.0)))
        return d.image(img)

    def material():
        k = g.integers(0, 10)
        if k < 5: return d.lambertian(tex())
        if k < 7: return d.metal(g.uniform(0.4, 1.0, 3), float(g.choice([0.0, 0.3, 1.0])))
        if k < 9: return d.dielectric(float(g.choice([1.5, 1.3, 2.4])))
        return d.diffuse_light(d.solid(g.uniform(1.0, 6.0, 3)))

    def primitive():
        k = g.integers(0, 6)
        c = g.uniform(-3, 3, 3)
        if k == 0: return d.sphere(c, float(g.uniform(0.2, 1.0)), material())
        if k == 1:
            return d.moving_sphere(c, c + g.uniform(-0.5, 0.5, 3), 0.0, 1.0, float(g.uniform(0.2, 0.8)), material())
        if k <= 4:
            a0, b0 = g.uniform(-3, 1, 2)
            return d.rect(int(k - 2), float(a0), float(a0 + g.uniform(0.5, 3)), float(b0), float(b0 + g.uniform(0.5, 3)),
                          float(g.uniform(-3, 3)), material())
        p0 = g.uniform(-3, 1, 3)
        return d.aabox(p0, p0 + g.uniform(0.4, 2.0, 3), material())

    def wrapped(h, depth):
        for _ in range(int(g.integers(0, 4 - depth))):
            k = g.integers(0, 3)
            if k == 0: h = d.translate(h, g.uniform(-1.5, 1.5, 3))
            elif k == 1: h = d.rotate_y(h, float(g.uniform(-60, 60)))
            else: h = d.flip_face(h)
        return h

    def obj(depth=0):
        k = g.integers(0, 10)
        if k < 6: return wrapped(primitive(), depth)
        if k < 8 and depth == 0:  # a medium; boundary = sphere, box, or wrapped box (no media inside)
            kb = g.integers(0, 3)
            c = g.uniform(-2, 2, 3)
            if kb == 0: b = d.sphere(c, float(g.uniform(0.6, 1.5)), d.dielectric(1.5))
            else:
                b = d.aabox(c, c + g.uniform(0.6, 2.0, 3), d.dielectric(1.5))
                if kb == 2: b = d.translate(d.rotate_y(b, float(g.uniform(-40, 40))), g.uniform(-1, 1, 3))
            return d.constant_medium(b, float(g.uniform(0.2, 2.0)), tex())
        n = int(g.integers(1, 5))
        group = d.bvh([wrapped(primitive(), 2) for _ in range(n)])   # nested BVH (len 1..4); <= 1 wrapper inside, <= 2 outside
        return wrapped(group, 1) if depth == 0 else group

    n = n_objects or int(g.integers(2, 9))
    world = d.bvh([obj() for _ in range(n)])
    lights = []
    null = None
    for _ in range(int(g.integers(0, 4))):
        null = null or d.null_material()
        k = g.integers(0, 4)
        if k == 0: lights.append(d.rect(1, -1.0, 1.0, -1.0, 1.0, 4.0, null))
        elif k == 1: lights.append(d.sphere(g.uniform(-2, 2, 3), float(g.uniform(0.3, 1.0)), null))
        elif k == 2: lights.append(d.rect(0, -1.0, 1.0, -1.0, 1.0, 3.0, null))          # XYRect: trait defaults
        else: lights.append(d.flip_face(d.rect(1, -1.0, 1.0, -1.0, 1.0, 3.5, null)))    # FlipFace does not forward pdf/random
    bg = g.uniform(0.0, 1.0, 3) if g.random() < 0.7 else np.zeros(3)
    cam = (g.uniform(-1, 1, 3) + np.array([0, 0.5, 9.0]), g.uniform(-0.5, 0.5, 3), (0.0, 1.0, 0.0), float(g.uniform(30, 60)),
           float(g.choice([1.0, 1.5])), float(g.choice([0.0, 0.2])), float(g.uniform(5, 10)), 0.0, 1.0)
    return d.finish(world, lights, bg, cam)
