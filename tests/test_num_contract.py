"""include/rt1w_num.h on the host: Philox known answers, draw shapes, elementary functions."""
import ctypes as C

import mpmath as mp
import numpy as np

import orc

_P = C.c_void_p


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    orc.A.orc_philox(c.ctypes.data_as(_P), k.ctypes.data_as(_P), o.ctypes.data_as(_P))
    return [int(v) for v in o]


def test_philox4x32_10_random123_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def stream(seed, shape, lo=0.0, hi=1.0, m=2, n=4096):
    out = np.empty(n)
    orc.A.orc_stream(seed, shape, lo, hi, m, out.ctypes.data_as(_P), n)
    return out


def test_word_stream_is_philox_blocks_in_order():
    w = stream(7, 0, n=8).astype(np.uint64)
    b0 = philox([0, 0, 0, 0x424C4453], [7, 0])
    b1 = philox([1, 0, 0, 0x424C4453], [7, 0])
    assert [int(x) for x in w] == b0 + b1


def test_draw_shapes():
    f = stream(3, 1, n=20000)
    assert f.min() >= 0.0 and f.max() < 1.0
    assert np.all(f * 2.0**53 == np.floor(f * 2.0**53))          # 53-bit grid, rand 0.8 Standard f64
    assert abs(f.mean() - 0.5) < 0.01
    r = stream(4, 2, lo=-1.0, hi=1.0, n=20000)
    assert r.min() >= -1.0 and r.max() < 1.0
    assert np.all((r + 1.0) * 2.0**51 == np.floor((r + 1.0) * 2.0**51))  # 52-bit mantissa * scale 2
    b = stream(5, 3, m=3, n=30000)
    assert set(np.unique(b)) == {0.0, 1.0, 2.0}
    assert np.all(np.abs(np.bincount(b.astype(int)) / b.size - 1 / 3) < 0.01)
    two = stream(6, 3, m=2, n=20000)
    assert abs(two.mean() - 0.5) < 0.02
    # u64 draws are even-aligned word pairs: f64 #0 of a stream = words 0,1
    w = stream(9, 0, n=4).astype(np.uint64)
    f0 = stream(9, 1, n=1)[0]
    assert f0 == float(((int(w[1]) << 32 | int(w[0])) >> 11)) * 2.0**-53


def num_eval(fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(a if b is None else b, dtype=np.float64)
    out = np.empty_like(a)
    orc.A.orc_num_eval(fn, a.ctypes.data_as(_P), b.ctypes.data_as(_P), out.ctypes.data_as(_P), a.size)
    return out


def ulps(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    sp = np.spacing(np.abs(ref))
    return np.abs(got - ref) / sp


def mpref(f, xs):
    mp.mp.prec = 200
    return np.array([float(f(mp.mpf(float(x)))) for x in xs])


def test_sin_cos_within_1ulp():
    rng = np.random.default_rng(1)
    for lo, hi in ((-7, 7), (-1e3, 1e3), (-1e5, 1e5)):
        x = rng.uniform(lo, hi, 4000)
        assert ulps(num_eval(2, x), mpref(mp.sin, x)).max() <= 1.0
        assert ulps(num_eval(3, x), mpref(mp.cos, x)).max() <= 1.0
    # documented domain edge: |x| >= 2^30 -> NaN on both targets
    assert np.all(np.isnan(num_eval(2, np.array([2.0**30, -1e300, np.inf, np.nan]))))


def test_atan2_acos_log_within_2ulp():
    rng = np.random.default_rng(2)
    a, b = rng.normal(size=4000), rng.normal(size=4000)
    mp.mp.prec = 200
    ref = np.array([float(mp.atan2(mp.mpf(float(p)), mp.mpf(float(q)))) for p, q in zip(a, b)])
    assert ulps(num_eval(5, a, b), ref).max() <= 2.0
    x = rng.uniform(-50, 50, 4000)
    arg = x / (np.abs(x) + 1.0)
    assert ulps(num_eval(4, x), mpref(mp.acos, arg)).max() <= 2.0
    y = np.concatenate([rng.uniform(1e-12, 1, 2000), 10.0 ** rng.uniform(-300, 300, 2000)])
    assert ulps(num_eval(6, y, y), mpref(mp.log, y)).max() <= 2.0
    # special values
    assert num_eval(5, np.array([0.0]), np.array([-1.0]))[0] == np.pi
    assert num_eval(6, np.array([1.0]), np.array([1.0]))[0] == 0.0
    assert num_eval(6, np.array([1.0]), np.array([0.0]))[0] == -np.inf


def test_floor_and_tan():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 5000), [0.0, -0.0, -1.0, 2.0, -2.5, 1e300, -1e300]])
    assert np.array_equal(num_eval(9, x), np.floor(x))
    t = rng.uniform(-1.5, 1.5, 2000)
    assert ulps(num_eval(10, t), mpref(mp.tan, t)).max() <= 3.0


def test_rng_mark_and_rewind_reproduce_the_word_stream():
    """rt_rng_mark / rt_rng_rewind (the phased walk's restart, csrc/rt_walk2.h): the buffered words are regenerated from the
    counter, so drawing on from a rewound stream gives the very words drawn after the mark -- at every buffer position, for mixed
    32- and 64-bit draws, in the Philox build and in the reference-stream (ChaCha12) build of the header."""
    import ctypes as C
    for lib in (orc.B, orc.flat_ref_lib()):
        lib.orcflat_rng_rewind_check.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        g = np.random.default_rng(5)
        for before in list(range(0, 24)) + [100, 257]:
            for pattern in (0, 0xFFFFFFFF, int(g.integers(0, 1 << 32)), int(g.integers(0, 1 << 32))):
                a = np.zeros(40, dtype=np.uint64)
                b = np.ones(40, dtype=np.uint64)
                lib.orcflat_rng_rewind_check(int(g.integers(0, 1 << 40)), pattern, before, 40, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))
                assert np.array_equal(a, b), (before, hex(pattern))
                assert len(set(a.tolist())) > 30
