"""A Cornell box with N rotated boxes (13 nodes each): the generic kernel (stack walk above 64 nodes) against the
scene-specialised sweep (up to RT_JIT_MAX_NODES = 256 nodes).  usage: python tools/midsize.py 4 8 16"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import orc
rt = orc.rt()
for nbox in [int(a) for a in sys.argv[1:]] or [8]:
    s = rt.Scene(build_seed=1)
    red = s.lambertian(s.solid_color((0.65, 0.05, 0.05))); white = s.lambertian(s.solid_color((0.73, 0.73, 0.73)))
    green = s.lambertian(s.solid_color((0.12, 0.45, 0.15))); light = s.diffuse_light(s.solid_color((15, 15, 15)))
    objs = [s.yz_rect(0, 555, 0, 555, 555, green), s.yz_rect(0, 555, 0, 555, 0, red),
            s.flip_face(s.xz_rect(213, 343, 227, 332, 554, light)), s.xz_rect(0, 555, 0, 555, 0, white),
            s.xz_rect(0, 555, 0, 555, 555, white), s.xy_rect(0, 555, 0, 555, 555, white)]
    rng = np.random.default_rng(3)
    for i in range(nbox):
        b = s.aabox((0, 0, 0), (60, float(rng.uniform(60, 250)), 60), white)
        b = s.rotate_y(b, float(rng.uniform(-40, 40)))
        b = s.translate(b, (float(rng.uniform(30, 460)), 0.0, float(rng.uniform(30, 460))))
        objs.append(b)
    s.set_world(s.bvh_node(objs))
    nm = s.null_material()
    s.set_lights([s.xz_rect(213, 343, 227, 332, 554, nm)])
    s.set_background((0, 0, 0))
    s.set_camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    s.commit()
    info = s.info()
    ctx = rt.Context(s, 0)
    g, sg = ctx.render(400, 400, 50)
    best_g = max(ctx.render(400, 400, 50)[1]["paths"] / ctx.render(400, 400, 50)[1]["kernel_ms"] / 1e3 for _ in range(1))
    line = f"boxes {nbox:3d} nodes {info['n_nodes']:4d} generic V{sg['variant']} {sg['paths'] / sg['kernel_ms'] / 1e3:8.1f} Mpaths/s"
    try:
        sp = ctx.specialise()
        a, sa = ctx.render(400, 400, 50)
        a, sa = ctx.render(400, 400, 50)
        line += f" | specialised {sa['paths'] / sa['kernel_ms'] / 1e3:8.1f} Mpaths/s (compile {sp['compile_ms'] / 1e3:.1f} s, {sp['vgprs']} VGPRs, grid {sp['grid']}) bit-exact {np.array_equal(a, g, equal_nan=True)}"
    except rt.Rt1wError as e:
        line += f" | not specialised: {e}"
    print(line, flush=True)
