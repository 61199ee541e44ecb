"""raytracing-1w_amd -- Python host binding of librt1w.so (the MI355X path tracer).

This is plumbing over the C ABI in ``include/rt1w.h``: every call goes to the
shared library, whose render path is the HIP kernel.  There is no Python or CPU
fallback -- if the library is missing the import fails, and without a GPU
``Context`` raises.

The package name has a hyphen (mandated layout), so import it with::

    import importlib
    rt = importlib.import_module("raytracing-1w_amd")

Method names follow the reference's constructors (src/sphere.rs, src/aarect.rs,
src/material.rs, src/texture.rs, src/bvh.rs, src/camera.rs of hatoo/raytracing-1w).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT1W_LIB") or os.path.join(_HERE, "librt1w.so")  # RT1W_LIB: diagnostic builds only

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "librt1w.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C raytracing-1w_amd/csrc` (needs hipcc, gfx950). There is no fallback path."
    )

_lib = C.CDLL(LIB_PATH)
_lab = None


def load_lab():
    """Load the diagnostics library librt1w_lab.so (the trace-only harness and the wavefront form: measured opt-ins that are not in the
    product library).  Loading it registers the wavefront form with librt1w.so; RT1W_WAVEFRONT renders need it."""
    global _lab
    if _lab is None:
        _lab = C.CDLL(os.path.join(os.path.dirname(LIB_PATH), "librt1w_lab.so"))
    return _lab


OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM, ERR_STATE, ERR_CANCELLED = -1, -2, -3, -4, -5, -6
ROWS_F64, ROWS_U8 = 0, 1
OUT_SUM = 1
UNSORTED = 2
PROBE_COHERENT = 0x40000000  # measurement only (include/rt1w.h): every wave traces one path 64 times; the frame is not the image
LDS_NODES = 4
GENERIC = 8  # do not use a scene-specialised kernel for this render
OUT_FRAME = 32  # rt1w_render: `out` is the whole image; only the tile's pixels are written, at their image positions
RNG_REFERENCE = 64  # parity mode: the reference's own ChaCha12 stream per pixel (main.rs:964)
NO_NODE_CACHE = 0x10000  # big scenes: the stack-walk kernels without the most visited node records in LDS (rt_walk_table.h)
CLASSIC_WALK = 128  # sphere scenes: the one-entry-per-step walk instead of the pair walk (rt_walk_pair.h)
WAVEFRONT = 16  # big scenes: path state queued in HBM, trace / shade kernels per bounce
SPECIALISE_CACHED_ONLY = 1


class Rt1wError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt1w error {code}: {msg}")
        self.code = code


class RenderParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "width", "height", "x0", "y0", "tile_w", "tile_h", "spp", "sample_offset",
        "max_depth", "global_seed", "chunk", "flags", "strip_rows", "strip_period", "precision", "partial_mib")]


class SpecialiseInfo(C.Structure):
    _fields_ = [("key", C.c_char * 24), ("active", C.c_uint32), ("from_cache", C.c_uint32), ("compile_ms", C.c_double),
                ("grid", C.c_uint32), ("vgprs", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("kernel_ms", C.c_double),
                ("total_ms", C.c_double), ("chunk", C.c_uint32), ("n_chunks", C.c_uint32),
                ("grid", C.c_uint32), ("block", C.c_uint32), ("variant", C.c_uint32), ("sorted", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_lights", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_textures", C.c_uint32), ("n_perlin", C.c_uint32), ("stack_need", C.c_uint32),
                ("scope_depth", C.c_uint32), ("has_media", C.c_uint32), ("has_textures", C.c_uint32),
                ("has_moving", C.c_uint32), ("variant", C.c_uint32), ("bytes", C.c_uint64)]


_P = C.c_void_p
_D3 = C.c_double * 3


def _sig(name, restype, *argtypes):
    f = getattr(_lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("rt1w_last_error", C.c_char_p)
_sig("rt1w_version", C.c_char_p)
_sig("rt1w_scene_create", C.c_int, C.c_uint64, C.POINTER(_P))
_sig("rt1w_scene_destroy", None, _P)
_sig("rt1w_scene_rng_f64", C.c_int, _P, C.POINTER(C.c_double))
_sig("rt1w_scene_rng_range", C.c_int, _P, C.c_double, C.c_double, C.POINTER(C.c_double))
_sig("rt1w_texture_solid", C.c_int, _P, _D3)
_sig("rt1w_texture_checker", C.c_int, _P, C.c_int, C.c_int)
_sig("rt1w_texture_noise", C.c_int, _P, C.c_double)
_sig("rt1w_texture_noise_tables", C.c_int, _P, C.c_double, _P, _P, _P, _P)
_sig("rt1w_texture_image", C.c_int, _P, _P, C.c_uint32, C.c_uint32)
_sig("rt1w_material_lambertian", C.c_int, _P, C.c_int)
_sig("rt1w_material_metal", C.c_int, _P, _D3, C.c_double)
_sig("rt1w_material_dielectric", C.c_int, _P, C.c_double)
_sig("rt1w_material_diffuse_light", C.c_int, _P, C.c_int)
_sig("rt1w_material_null", C.c_int, _P)
_sig("rt1w_hittable_sphere", C.c_int, _P, _D3, C.c_double, C.c_int)
_sig("rt1w_hittable_moving_sphere", C.c_int, _P, _D3, _D3, C.c_double, C.c_double, C.c_double, C.c_int)
for _n in ("xy", "xz", "yz"):
    _sig(f"rt1w_hittable_{_n}_rect", C.c_int, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int)
_sig("rt1w_hittable_aabox", C.c_int, _P, _D3, _D3, C.c_int)
_sig("rt1w_hittable_translate", C.c_int, _P, C.c_int, _D3)
_sig("rt1w_hittable_rotate_y", C.c_int, _P, C.c_int, C.c_double, C.c_double, C.c_double)
_sig("rt1w_hittable_flip_face", C.c_int, _P, C.c_int)
_sig("rt1w_hittable_constant_medium", C.c_int, _P, C.c_int, C.c_double, C.c_int)
_sig("rt1w_hittable_bvh", C.c_int, _P, C.POINTER(C.c_int), C.c_uint32, C.c_double, C.c_double)
_sig("rt1w_scene_set_world", C.c_int, _P, C.c_int)
_sig("rt1w_scene_set_lights", C.c_int, _P, C.POINTER(C.c_int), C.c_uint32)
_sig("rt1w_scene_set_background", C.c_int, _P, _D3)
_sig("rt1w_scene_set_camera", C.c_int, _P, _D3, _D3, _D3, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)
_sig("rt1w_scene_commit", C.c_int, _P)
_sig("rt1w_scene_build_reference", C.c_int, C.c_int, C.c_uint64, C.c_double, _P, C.c_uint32, C.c_uint32,
     C.POINTER(_P), C.POINTER(C.c_uint32 * 3))
_sig("rt1w_scene_set_walk_order", C.c_int, _P, C.c_uint32)
_sig("rt1w_scene_set_bvh_build", C.c_int, _P, C.c_uint32)
_sig("rt1w_scene_get_bvh_topology", C.c_int64, _P, _P, C.c_uint64)
_sig("rt1w_scene_get_info", C.c_int, _P, C.POINTER(SceneInfo))
_sig("rt1w_scene_copy_flat", C.c_int64, _P, C.c_int, _P, C.c_uint64)
_sig("rt1w_device_count", C.c_int)
_sig("rt1w_context_create", C.c_int, C.c_int, _P, C.POINTER(_P))
_sig("rt1w_context_destroy", None, _P)
_sig("rt1w_context_specialise", C.c_int, _P, C.c_uint32, C.POINTER(SpecialiseInfo))
_sig("rt1w_scene_kernel_key", C.c_int, _P, C.c_char * 24)
_sig("rt1w_default_chunk", C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32)
_sig("rt1w_scene_default_chunk", C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32)
_sig("rt1w_render", C.c_int, _P, C.POINTER(RenderParams), _P, C.POINTER(Stats))
_sig("rt1w_render_device", C.c_int, _P, C.POINTER(RenderParams), _P, C.POINTER(Stats))
_sig("rt1w_render_u8", C.c_int, _P, C.POINTER(RenderParams), _P, C.POINTER(Stats))
PROGRESS_FN = C.CFUNCTYPE(C.c_int, _P, C.c_uint32, C.c_uint32)
_sig("rt1w_render_rows", C.c_int, _P, C.POINTER(RenderParams), C.c_uint32, C.c_int, _P, PROGRESS_FN, _P, C.POINTER(Stats))
_sig("rt1w_abi_sizeof", C.c_uint32, C.c_int)
_sig("rt1w_host_alloc", C.c_int, C.c_uint64, C.POINTER(_P))
_sig("rt1w_host_free", C.c_int, _P)
_sig("rt1w_host_register", C.c_int, _P, C.c_uint64)
_sig("rt1w_host_unregister", C.c_int, _P)
_sig("rt1w_resolve", C.c_int, _P, C.c_uint64, C.c_uint32, _P)
_sig("rt1w_quantize", C.c_int, _P, C.c_uint64, _P)
_sig("rt1w_format_ppm", C.c_int64, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64)
_sig("rt1w_debug_eval", C.c_int, _P, C.c_int, _P, _P, _P, C.c_uint64)
_sig("rt1w_debug_aabb", C.c_int, _P, _P, _P, _P, C.c_uint64)
_sig("rt1w_debug_stamps", C.c_int, _P, C.POINTER(C.c_uint64 * 16), C.c_int)
_sig("rt1w_debug_texture", C.c_int, _P, C.c_int, C.c_uint32, _P, _P, C.c_uint64)


for _i, _t in enumerate((RenderParams, Stats, SceneInfo, SpecialiseInfo)):
    if _lib.rt1w_abi_sizeof(_i) != C.sizeof(_t):
        raise ImportError(f"librt1w.so and this binding disagree on the layout of {_t.__name__}: rebuild the library")


def last_error():
    return _lib.rt1w_last_error().decode()


def version():
    return _lib.rt1w_version().decode()


def _ck(rc):
    if rc < 0:
        raise Rt1wError(rc, last_error())
    return rc


def _v3(v):
    return _D3(float(v[0]), float(v[1]), float(v[2]))


def device_count():
    return _lib.rt1w_device_count()


def default_chunk(tile_w, tile_h, spp):
    return _lib.rt1w_default_chunk(tile_w, tile_h, spp)


_EARTH = None


def earth_rgb8():
    """Decoded 1024x512 RGB8 earth map used by scene arms 3 and 7 (host one-shot).

    The reference decodes assets/earthmap.jpg with the `image` crate (src/main.rs:347-348);
    here PIL decodes the same asset; decoders may differ by 1 LSB per texel (unpinned).
    """
    global _EARTH
    if _EARTH is None:
        from PIL import Image
        im = Image.open(os.path.join(_HERE, "assets", "earthmap.jpg")).convert("RGB")
        _EARTH = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    return _EARTH


class Scene:
    """Scene under construction / committed (rt1w_scene)."""

    def __init__(self, build_seed=1, _handle=None, _defaults=None):
        if _handle is None:
            h = _P()
            _ck(_lib.rt1w_scene_create(C.c_uint64(build_seed), C.byref(h)))
            _handle = h
        self._h = _handle
        self.defaults = _defaults  # (image_width, image_height, samples_per_pixel) of a reference arm
        self._keep = []

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.rt1w_scene_destroy(self._h)
            self._h = None

    @classmethod
    def reference(cls, arm, build_seed=1, aspect_ratio=None):
        """The scene table of the reference's main (src/main.rs:815-937)."""
        if aspect_ratio is None:
            aspect_ratio = 1.0 if (arm in (5, 6) or arm < 0 or arm > 6) else 16.0 / 9.0
        earth = None
        ew = eh = 0
        if arm == 3 or arm < 0 or arm > 6:
            earth = earth_rgb8()
            eh, ew = earth.shape[:2]
        h = _P()
        d = (C.c_uint32 * 3)()
        _ck(_lib.rt1w_scene_build_reference(arm, C.c_uint64(build_seed), aspect_ratio,
                                            earth.ctypes.data_as(_P) if earth is not None else None,
                                            ew, eh, C.byref(h), C.byref(d)))
        return cls(_handle=h, _defaults=(d[0], d[1], d[2]))

    # draws from the build stream
    def rng_f64(self):
        x = C.c_double()
        _ck(_lib.rt1w_scene_rng_f64(self._h, C.byref(x)))
        return x.value

    def rng_range(self, lo, hi):
        x = C.c_double()
        _ck(_lib.rt1w_scene_rng_range(self._h, lo, hi, C.byref(x)))
        return x.value

    # textures
    def solid_color(self, rgb): return _ck(_lib.rt1w_texture_solid(self._h, _v3(rgb)))
    def checker_texture(self, odd, even): return _ck(_lib.rt1w_texture_checker(self._h, odd, even))
    def noise_texture(self, scale): return _ck(_lib.rt1w_texture_noise(self._h, scale))

    def noise_texture_tables(self, scale, ranvec, perm_x, perm_y, perm_z):
        rv = np.ascontiguousarray(ranvec, dtype=np.float64).reshape(768)
        ps = [np.ascontiguousarray(p, dtype=np.uint32).reshape(256) for p in (perm_x, perm_y, perm_z)]
        return _ck(_lib.rt1w_texture_noise_tables(self._h, scale, rv.ctypes.data_as(_P), *[p.ctypes.data_as(_P) for p in ps]))

    def image_texture(self, rgb8):
        a = np.ascontiguousarray(rgb8, dtype=np.uint8)
        h, w = a.shape[:2]
        return _ck(_lib.rt1w_texture_image(self._h, a.ctypes.data_as(_P), w, h))

    # materials
    def lambertian(self, tex): return _ck(_lib.rt1w_material_lambertian(self._h, tex))
    def metal(self, albedo, fuzz): return _ck(_lib.rt1w_material_metal(self._h, _v3(albedo), fuzz))
    def dielectric(self, ir): return _ck(_lib.rt1w_material_dielectric(self._h, ir))
    def diffuse_light(self, tex): return _ck(_lib.rt1w_material_diffuse_light(self._h, tex))
    def null_material(self): return _ck(_lib.rt1w_material_null(self._h))

    # hittables
    def sphere(self, center, radius, mat): return _ck(_lib.rt1w_hittable_sphere(self._h, _v3(center), radius, mat))

    def moving_sphere(self, c0, c1, t0, t1, radius, mat):
        return _ck(_lib.rt1w_hittable_moving_sphere(self._h, _v3(c0), _v3(c1), t0, t1, radius, mat))

    def xy_rect(self, x0, x1, y0, y1, k, mat): return _ck(_lib.rt1w_hittable_xy_rect(self._h, x0, x1, y0, y1, k, mat))
    def xz_rect(self, x0, x1, z0, z1, k, mat): return _ck(_lib.rt1w_hittable_xz_rect(self._h, x0, x1, z0, z1, k, mat))
    def yz_rect(self, y0, y1, z0, z1, k, mat): return _ck(_lib.rt1w_hittable_yz_rect(self._h, y0, y1, z0, z1, k, mat))
    def aabox(self, p0, p1, mat): return _ck(_lib.rt1w_hittable_aabox(self._h, _v3(p0), _v3(p1), mat))
    def translate(self, child, offset): return _ck(_lib.rt1w_hittable_translate(self._h, child, _v3(offset)))
    def rotate_y(self, child, angle_deg, time0=0.0, time1=1.0): return _ck(_lib.rt1w_hittable_rotate_y(self._h, child, time0, time1, angle_deg))
    def flip_face(self, child): return _ck(_lib.rt1w_hittable_flip_face(self._h, child))
    def constant_medium(self, boundary, density, tex): return _ck(_lib.rt1w_hittable_constant_medium(self._h, boundary, density, tex))

    def bvh_node(self, children, time0=0.0, time1=1.0):
        arr = (C.c_int * len(children))(*children)
        return _ck(_lib.rt1w_hittable_bvh(self._h, arr, len(children), time0, time1))

    # scene level
    def set_world(self, hid): _ck(_lib.rt1w_scene_set_world(self._h, hid))

    def set_lights(self, ids):
        arr = (C.c_int * max(1, len(ids)))(*ids)
        _ck(_lib.rt1w_scene_set_lights(self._h, arr, len(ids)))

    def set_background(self, rgb): _ck(_lib.rt1w_scene_set_background(self._h, _v3(rgb)))

    def set_camera(self, look_from, look_at, vup, vfov_deg, aspect_ratio, aperture, focus_dist, time0, time1):
        _ck(_lib.rt1w_scene_set_camera(self._h, _v3(look_from), _v3(look_at), _v3(vup), vfov_deg, aspect_ratio,
                                       aperture, focus_dist, time0, time1))

    def commit(self): _ck(_lib.rt1w_scene_commit(self._h))

    def set_bvh_build(self, mode):
        """Opt-in BVH build (rt1w_scene_set_bvh_build): False / 0 = BVHNode::new as written (random axis, median split; default),
        True / 1 / "sah" = the trees rebuilt by surface-area heuristic (statistically the same frames, not bit for bit),
        2 / "best_axis" = BVHNode::new as written with the axis of bvh.rs:84 chosen by the lowest cost of its median split instead of
        drawn (a tree the reference itself can build).  Returns self."""
        mode = {"reference": 0, "sah": 1, "best_axis": 2}.get(mode, mode)
        _ck(_lib.rt1w_scene_set_bvh_build(self._h, int(mode)))
        return self

    def set_walk_order(self, near_far):
        """Opt-in traversal order (rt1w_scene_set_walk_order): False = the reference's left-then-right (default),
        True = near child first in media-free subtrees.  On a committed scene, before creating contexts."""
        _ck(_lib.rt1w_scene_set_walk_order(self._h, int(near_far)))   # 0 reference, 1 near-far (frames identical on all tested scenes, not provably; segment counts may differ), 2 near-far everywhere
        return self

    def default_chunk(self, tile_w, tile_h, spp):
        """Samples per work item the renders of THIS scene use when `chunk` is 0 (rt1w_scene_default_chunk): 1 for scenes on the
        stack-walk kernels, the scene-independent rule (default_chunk) for the others."""
        return int(_lib.rt1w_scene_default_chunk(self._h, tile_w, tile_h, spp))

    def bvh_topology(self):
        """The trees of the opt-in SAH / best-axis builds as one int32 stream (rt1w_scene_get_bvh_topology); empty for the reference's build."""
        n = int(_lib.rt1w_scene_get_bvh_topology(self._h, None, 0))
        _ck(n)
        out = np.zeros(max(n, 1), dtype=np.int32)
        _ck(int(_lib.rt1w_scene_get_bvh_topology(self._h, out.ctypes.data_as(_P), n)))
        return out[:n]

    def info(self):
        i = SceneInfo()
        _ck(_lib.rt1w_scene_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in SceneInfo._fields_}

    def kernel_key(self):
        """Cache key of this scene's specialised kernel (sweep_<key>.hsaco); raises ERR_UNSUPPORTED for big scenes."""
        buf = (C.c_char * 24)()
        _ck(_lib.rt1w_scene_kernel_key(self._h, buf))
        return buf.value.decode()

    def flat(self, what):
        """Bytes of one flat array (0 nodes,1 lights,2 materials,3 textures,4 perlin,5 images,6 camera+bg)."""
        n = _lib.rt1w_scene_copy_flat(self._h, what, None, 0)
        _ck(int(n))
        buf = np.zeros(max(int(n), 1), dtype=np.uint8)
        _ck(int(_lib.rt1w_scene_copy_flat(self._h, what, buf.ctypes.data_as(_P), int(n))))
        return buf[:int(n)]


class Context:
    """One GPU + one HIP stream with the scene uploaded (rt1w_context)."""

    def __init__(self, scene, device=0):
        h = _P()
        _ck(_lib.rt1w_context_create(device, scene._h, C.byref(h)))
        self._h = h
        self.scene = scene
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            _lib.rt1w_context_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _params(width, height, spp, max_depth, tile, sample_offset, global_seed, chunk, out_sum, variant=None, unsorted=False, lds_nodes=False, generic=False, wavefront=False,
                strips=None, out_frame=False, reference_stream=False, f32=False, classic_walk=False, probe_coherent=False, partial_mib=0, no_node_cache=False):
        x0, y0, tw, th = tile if tile is not None else (0, 0, width, height)
        if wavefront:
            load_lab()
        flags = (OUT_SUM if out_sum else 0) | (OUT_FRAME if out_frame else 0) | (RNG_REFERENCE if reference_stream else 0) | (UNSORTED if unsorted else 0) | (LDS_NODES if lds_nodes else 0) | (GENERIC if generic else 0) | (WAVEFRONT if wavefront else 0) | (CLASSIC_WALK if classic_walk else 0) | (NO_NODE_CACHE if no_node_cache else 0) | (PROBE_COHERENT if probe_coherent else 0) | (((variant + 1) << 8) if variant is not None else 0)
        sr, sp = strips if strips is not None else (0, 0)
        return RenderParams(width, height, x0, y0, tw, th, spp, sample_offset, max_depth, global_seed, chunk, flags, sr, sp, 1 if f32 else 0, int(partial_mib))

    def specialise(self, cached_only=False):
        """Load (from the kernel cache) or compile (hiprtc, 3-5 s) the kernel specialised for this scene's topology
        (rt1w_context_specialise).  Returns the info dict; raises Rt1wError for scenes of more than 256 nodes
        (ERR_UNSUPPORTED) or, with cached_only, on a cache miss (ERR_STATE)."""
        info = SpecialiseInfo()
        _ck(_lib.rt1w_context_specialise(self._h, SPECIALISE_CACHED_ONLY if cached_only else 0, C.byref(info)))
        return {"key": info.key.decode(), "active": bool(info.active), "from_cache": bool(info.from_cache),
                "compile_ms": info.compile_ms, "grid": info.grid, "vgprs": info.vgprs}

    def specialised(self):
        """True if renders on this context use a scene-specialised kernel (cache hit at creation or specialise())."""
        info = SpecialiseInfo()
        rc = _lib.rt1w_context_specialise(self._h, SPECIALISE_CACHED_ONLY, C.byref(info))
        return rc == 0 and bool(info.active)

    def render(self, width, height, spp, max_depth=50, tile=None, sample_offset=0, global_seed=0, chunk=0, out_sum=False,
               variant=None, unsorted=False, lds_nodes=False, generic=False, wavefront=False, strips=None, out=None, frame=None,
               reference_stream=False, f32=False, classic_walk=False, probe_coherent=False, partial_mib=0, no_node_cache=False):
        """Returns (image[tile_h, tile_w, 3] float64 with row 0 = reference row j = y0, stats dict).
        probe_coherent: measurement mode RT1W_PROBE_COHERENT -- the returned array is NOT the image.
        strips=(strip_rows, strip_period): row-interleaved tile (tile row r = image row y0 + r//strip_rows*strip_period +
        r%strip_rows).  out: caller's array for the packed tile (e.g. pinned_empty).  frame: caller's WHOLE image
        [height, width, 3]; the tile's pixels are written at their image positions (RT1W_OUT_FRAME) and `frame` is returned."""
        p = self._params(width, height, spp, max_depth, tile, sample_offset, global_seed, chunk, out_sum, variant, unsorted, lds_nodes, generic, wavefront,
                         strips, frame is not None, reference_stream, f32, classic_walk, probe_coherent, partial_mib, no_node_cache)
        if frame is not None:
            assert frame.dtype == np.float64 and frame.shape == (height, width, 3) and frame.flags.c_contiguous
            out = frame
        elif out is None:
            out = np.empty((p.tile_h, p.tile_w, 3), dtype=np.float64)
        else:
            assert out.dtype == np.float64 and out.shape == (p.tile_h, p.tile_w, 3) and out.flags.c_contiguous
        st = Stats()
        _ck(_lib.rt1w_render(self._h, C.byref(p), out.ctypes.data_as(_P), C.byref(st)))
        return out, {n: getattr(st, n) for n, _ in Stats._fields_}

    def render_u8(self, width, height, spp, max_depth=50, tile=None, sample_offset=0, global_seed=0, chunk=0, reference_stream=False):
        """Quantised on the device, rows top-down as the reference prints them: uint8 [tile_h, tile_w, 3]."""
        p = self._params(width, height, spp, max_depth, tile, sample_offset, global_seed, chunk, False, reference_stream=reference_stream)
        out = np.empty((p.tile_h, p.tile_w, 3), dtype=np.uint8)
        st = Stats()
        _ck(_lib.rt1w_render_u8(self._h, C.byref(p), out.ctypes.data_as(_P), C.byref(st)))
        return out, {n: getattr(st, n) for n, _ in Stats._fields_}

    def render_rows(self, width, height, spp, strip_rows=0, u8=False, progress=None, max_depth=50, tile=None,
                    sample_offset=0, global_seed=0, chunk=0, out_sum=False, out=None, no_node_cache=False):
        """Strip-wise render from the top row down with D2H overlapped (rt1w_render_rows).  `progress(rows_done, rows_total)`
        is called as strips land; returning a true value cancels (raises Rt1wError with code ERR_CANCELLED).
        Returns the same arrays as render() (u8=False) or render_u8() (u8=True)."""
        p = self._params(width, height, spp, max_depth, tile, sample_offset, global_seed, chunk, out_sum, no_node_cache=no_node_cache)
        if out is None:
            out = np.empty((p.tile_h, p.tile_w, 3), dtype=np.uint8 if u8 else np.float64)
        assert out.flags.c_contiguous and out.shape == (p.tile_h, p.tile_w, 3) and out.dtype == (np.uint8 if u8 else np.float64)
        st = Stats()
        cb = PROGRESS_FN((lambda user, done, total: 1 if progress(done, total) else 0) if progress else 0)
        _ck(_lib.rt1w_render_rows(self._h, C.byref(p), strip_rows, ROWS_U8 if u8 else ROWS_F64, out.ctypes.data_as(_P), cb,
                                  None, C.byref(st)))
        return out, {n: getattr(st, n) for n, _ in Stats._fields_}

    def render_device(self, d_ptr, width, height, spp, max_depth=50, tile=None, sample_offset=0, global_seed=0, chunk=0,
                      out_sum=False, variant=None, unsorted=False, generic=False, strips=None, f32=False):
        """Same, into device memory `d_ptr` (int address, e.g. torch tensor .data_ptr()); packed tile rows."""
        p = self._params(width, height, spp, max_depth, tile, sample_offset, global_seed, chunk, out_sum, variant, unsorted, False, generic,
                         False, strips, f32=f32)
        st = Stats()
        _ck(_lib.rt1w_render_device(self._h, C.byref(p), C.c_void_p(d_ptr), C.byref(st)))
        return {n: getattr(st, n) for n, _ in Stats._fields_}

    def debug_aabb(self, cases):
        """cases[n, 14] = min3, max3, origin3, direction3, t_min, t_max -> (literal[n], fast[n]) from the device."""
        a = np.ascontiguousarray(cases, dtype=np.float64).reshape(-1, 14)
        lit = np.empty(a.shape[0], dtype=np.int32)
        fast = np.empty(a.shape[0], dtype=np.int32)
        _ck(_lib.rt1w_debug_aabb(self._h, a.ctypes.data_as(_P), lit.ctypes.data_as(_P), fast.ctypes.data_as(_P), a.shape[0]))
        return lit, fast

    def debug_texture(self, mode, tex, uvp):
        """uvp[n, 5] = u, v, p.xyz -> out[n, 3] from the device: mode 0 Texture::value of texture `tex`, 1 Perlin noise / turb of
        table `tex`, 2 sphere_uv(p) (rt1w_debug_texture)."""
        a = np.ascontiguousarray(uvp, dtype=np.float64).reshape(-1, 5)
        out = np.empty((a.shape[0], 3), dtype=np.float64)
        _ck(_lib.rt1w_debug_texture(self._h, mode, tex, a.ctypes.data_as(_P), out.ctypes.data_as(_P), a.shape[0]))
        return out

    def debug_stamps(self, reset=True):
        out = (C.c_uint64 * 16)()
        rc = _ck(_lib.rt1w_debug_stamps(self._h, C.byref(out), 1 if reset else 0))
        return rc, [int(v) for v in out]

    def debug_eval(self, fn, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty_like(a)
        _ck(_lib.rt1w_debug_eval(self._h, fn, a.ctypes.data_as(_P), b.ctypes.data_as(_P), out.ctypes.data_as(_P), a.size))
        return out


class _Pinned:
    def __init__(self, ptr): self.ptr = ptr
    def __del__(self):
        if self.ptr: _lib.rt1w_host_free(self.ptr); self.ptr = None


def pinned_empty(shape, dtype=np.float64):
    """numpy array over page-locked host memory (rt1w_host_alloc): device->host copies into it run at full PCIe rate."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = _P()
    _ck(_lib.rt1w_host_alloc(n, C.byref(p)))
    keep = _Pinned(p)
    buf = (C.c_uint8 * n).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype).reshape(shape)
    _PINNED_KEEP[id(buf)] = keep
    import weakref
    weakref.finalize(buf, _PINNED_KEEP.pop, id(buf), None)
    return a


_PINNED_KEEP = {}


def host_register(array):
    """Pin memory the caller owns (e.g. a shared-memory mapping several single-GPU processes fill)."""
    _ck(_lib.rt1w_host_register(C.c_void_p(array.ctypes.data), array.nbytes))


def host_unregister(array):
    _ck(_lib.rt1w_host_unregister(C.c_void_p(array.ctypes.data)))


def resolve(sums, spp):
    s = np.ascontiguousarray(sums, dtype=np.float64)
    out = np.empty_like(s)
    _ck(_lib.rt1w_resolve(s.ctypes.data_as(_P), s.size // 3, spp, out.ctypes.data_as(_P)))
    return out


def quantize(means):
    m = np.ascontiguousarray(means, dtype=np.float64)
    out = np.empty(m.shape, dtype=np.uint8)
    _ck(_lib.rt1w_quantize(m.ctypes.data_as(_P), m.size, out.ctypes.data_as(_P)))
    return out


def format_ppm(means):
    """P3 text exactly as the reference prints it (src/main.rs:953,1003-1007); means[j, i, 3], row 0 = j = 0."""
    m = np.ascontiguousarray(means, dtype=np.float64)
    h, w = m.shape[:2]
    n = int(_lib.rt1w_format_ppm(m.ctypes.data_as(_P), w, h, None, 0))
    _ck(n)
    buf = C.create_string_buffer(n + 1)
    _ck(int(_lib.rt1w_format_ppm(m.ctypes.data_as(_P), w, h, buf, n + 1)))
    return buf.raw[:n].decode()
