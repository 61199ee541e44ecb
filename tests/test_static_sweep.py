"""The sweep unrolled along a scene's own tree (rt_sweep_static, what the library's run-time compiler builds for the
GPU) against the generic sweep, on the CPU: the flat core is compiled once more with the topologies of a few scenes as
compile-time arrays (generated here exactly as jit.cpp generates them) and must give identical bits."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
from dual import random_scene_pair

NODE = np.dtype([('kind', '<u4'), ('skip', '<u4'), ('d', '<f8', 6), ('b', '<u4'), ('mat', '<u4'), ('e', '<f8', 3), ('a', '<u4'), ('pad', '<u4')])


def scenes(rt):
    out = [("cornell", rt.Scene.reference(5, build_seed=1)), ("cornell_seed3_drawn_axes", rt.Scene.reference(5, build_seed=3).set_bvh_build("reference")),
           ("cornel_smoke", rt.Scene.reference(6, build_seed=1)), ("simple_light", rt.Scene.reference(4, build_seed=1)),
           ("two_perlin", rt.Scene.reference(2, build_seed=1))]
    n = 0
    for seed in range(3000, 3100):
        prod, _ = random_scene_pair(seed)
        info = prod.info()
        if info["n_nodes"] <= 64 and info["scope_depth"] >= 2 and (info["has_media"] or n % 2 == 0):
            out.append((f"random{seed}", prod))
            n += 1
            if n == 4:
                break
    return out


@pytest.fixture(scope="module")
def static_lib(rt, tmp_path_factory):
    work = tmp_path_factory.mktemp("static_sweep")
    cases = scenes(rt)
    hdr = []
    sw = []
    for k, (name, sc) in enumerate(cases):
        nodes = np.frombuffer(sc.flat(0).tobytes(), dtype=NODE)
        info = sc.info()
        n = len(nodes)
        root = int(np.frombuffer(sc.flat(6).tobytes()[-8:-4], dtype='<u4')[0])
        hdr.append(f"struct Topo{k} {{ static constexpr uint32_t n = {n}u, root = {root}u;\n"
                   f"  static constexpr uint32_t kind[{n}] = {{{', '.join(str(int(x)) + 'u' for x in nodes['kind'])}}};\n"
                   f"  static constexpr uint32_t skip[{n}] = {{{', '.join(str(int(x)) + 'u' for x in nodes['skip'])}}}; }};\n"
                   f"typedef RtCfg<{'true' if info['has_media'] else 'false'}, {'true' if info['has_textures'] else 'false'}, "
                   f"{'true' if info['has_moving'] else 'false'}, true, {max(2, info['scope_depth'])}, Topo{k}> CfgS{k};\n")
        sw.append(f"case {100 + k}: run_path<CfgS{k}>(sc, f, px, py, s, stk, sum, segs, path); break;")
    (work / "topo_gen.h").write_text("".join(hdr) + f"#define ORC_N_STATIC {len(cases)}\n#define ORC_STATIC_CASES " + " ".join(sw) + "\n")
    so = work / "liborc_flat_static.so"
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-Wno-unknown-pragmas",
           "-I" + os.path.join(orc.ROOT, "include"), "-I" + os.path.join(orc.ROOT, "raytracing-1w_amd", "csrc"), "-I" + str(work),
           "-DRT_RNG_CHECK", '-DORC_STATIC_TOPO_H="topo_gen.h"', "-shared", os.path.join(orc.ROOT, "oracle", "oracle_flat.cpp"), "-o", str(so)]
    subprocess.check_call(cmd)
    return orc.declare_flat(C.CDLL(str(so))), cases


def test_unrolled_sweep_equals_generic_sweep(static_lib):
    lib, cases = static_lib
    assert lib.orcflat_n_static() == len(cases) >= 8 and orc.B.orcflat_n_static() == 0
    for k, (name, sc) in enumerate(cases):
        W, H, spp = (48, 48, 6) if "cornel" in name else (28, 20, 4)
        a, sa = orc.flat_render(sc, W, H, spp, chunk=3)
        b, sb = orc.flat_render(sc, W, H, spp, chunk=3, variant=100 + k, lib=lib)
        assert sa["segments"] == sb["segments"], name
        assert np.array_equal(a, b, equal_nan=True), name


def test_kernel_key_does_not_depend_on_the_host_compiler(rt):
    """The kernels under <package>/kernels were compiled by the build for this very library: their key is source + headers + options,
    NOT the hiprtc of the host that runs them (round-2 advice: with the toolchain id in the key a host without libhiprtc, or with
    another minor release, silently fell back to the generic kernels).  The key is the same with the run-time compiler hidden
    (RT1W_NO_HIPRTC), and the build's precompiled kernel of the Cornell arm is there under that key."""
    import os
    import subprocess
    import sys
    code = ("import importlib, sys; sys.path.insert(0, %r); rt = importlib.import_module('raytracing-1w_amd'); "
            "print(rt.Scene.reference(5, build_seed=1).kernel_key())" % orc.ROOT)
    keys = []
    for hide in (False, True):
        env = dict(os.environ)
        env.pop("RT1W_NO_HIPRTC", None)
        if hide:
            env["RT1W_NO_HIPRTC"] = "1"
        keys.append(subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip())
    assert keys[0] == keys[1] and len(keys[0]) == 16
    kdir = os.path.join(orc.ROOT, "raytracing-1w_amd", "kernels")
    if os.path.isdir(kdir):   # filled by build(); absent only if hiprtc was unavailable at build time
        assert os.path.exists(os.path.join(kdir, "sweep_%s.hsaco" % keys[0])), os.listdir(kdir)
