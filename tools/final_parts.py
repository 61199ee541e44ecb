"""Where final_scene's time goes: the scene of main.rs:635-795 rebuilt through the constructors with parts left out
(global fog, smoke ball, floor boxes, sphere cluster, textures), kernel-time Mpaths/s and segments/path of each (800x800, 16 spp)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-1w_amd")


def build(fog=True, smoke=True, floor=True, cluster=True, textures=True, sah=False):
    s = rt.Scene(1)
    lam = lambda rgb: s.lambertian(s.solid_color(rgb))
    ground = lam((0.48, 0.83, 0.53))
    objects = []
    boxes1 = []
    for i in range(20):
        for j in range(20):
            x0, z0 = -1000.0 + i * 100.0, -1000.0 + j * 100.0
            y1 = s.rng_range(1.0, 101.0)
            if floor:
                boxes1.append(s.aabox((x0, 0.0, z0), (x0 + 100.0, y1, z0 + 100.0), ground))
    if floor:
        objects.append(s.bvh_node(boxes1))
    else:
        objects.append(s.xz_rect(-1000.0, 1000.0, -1000.0, 1000.0, 50.0, ground))
    light = s.diffuse_light(s.solid_color((7.0, 7.0, 7.0)))
    objects.append(s.flip_face(s.xz_rect(123.0, 423.0, 147.0, 412.0, 554.0, light)))
    objects.append(s.moving_sphere((400.0, 400.0, 200.0), (430.0, 400.0, 200.0), 0.0, 1.0, 50.0, lam((0.7, 0.3, 0.1))))
    objects.append(s.sphere((260.0, 150.0, 45.0), 50.0, s.dielectric(1.5)))
    objects.append(s.sphere((0.0, 150.0, 145.0), 50.0, s.metal((0.8, 0.8, 0.9), 1.0)))
    objects.append(s.sphere((360.0, 150.0, 145.0), 70.0, s.dielectric(1.5)))
    if smoke:
        objects.append(s.constant_medium(s.sphere((360.0, 150.0, 145.0), 70.0, s.dielectric(1.5)), 0.2, s.solid_color((0.2, 0.4, 0.9))))
    if fog:
        objects.append(s.constant_medium(s.sphere((0.0, 0.0, 0.0), 5000.0, s.dielectric(1.5)), 0.0001, s.solid_color((1.0, 1.0, 1.0))))
    emat = s.lambertian(s.image_texture(rt.earth_rgb8())) if textures else lam((0.3, 0.4, 0.7))
    objects.append(s.sphere((400.0, 200.0, 400.0), 100.0, emat))
    pertext = s.lambertian(s.noise_texture(0.1)) if textures else lam((0.5, 0.5, 0.5))
    objects.append(s.sphere((220.0, 280.0, 300.0), 80.0, pertext))
    if cluster:
        white = lam((0.73, 0.73, 0.73))
        boxes2 = [s.sphere((s.rng_range(0.0, 165.0), s.rng_range(0.0, 165.0), s.rng_range(0.0, 165.0)), 10.0, white) for _ in range(1000)]
        objects.append(s.translate(s.rotate_y(s.bvh_node(boxes2), 15.0), (-100.0, 270.0, 395.0)))
    s.set_world(s.bvh_node(objects))
    s.set_lights([s.xz_rect(123.0, 423.0, 147.0, 412.0, 554.0, s.null_material())])
    s.set_background((0.0, 0.0, 0.0))
    s.set_camera((478.0, 278.0, -600.0), (278.0, 278.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    s.commit()
    if sah:
        s.set_bvh_build(True)
    return s


CASES = [("all (= final_scene)", {}), ("no global fog", dict(fog=False)), ("no media at all", dict(fog=False, smoke=False)),
         ("no floor boxes (one rect)", dict(floor=False)), ("no sphere cluster", dict(cluster=False)), ("no textures", dict(textures=False)),
         ("no media, no textures", dict(fog=False, smoke=False, textures=False)),
         ("only floor + lights + spheres (no media/cluster/textures)", dict(fog=False, smoke=False, cluster=False, textures=False))]
for name, kw in CASES:
    for sah in (False, True):
        sc = build(sah=sah, **kw)
        ctx = rt.Context(sc, 0)
        ctx.render(800, 800, 2)
        best = 0
        for _ in range(2):
            g, st = ctx.render(800, 800, 64)
            best = max(best, st["paths"] / st["kernel_ms"] / 1e3)
        info = sc.info()
        print(f"{name:58s} {'SAH' if sah else 'ref'} nodes {info['n_nodes']:5d} V{st['variant']} seg/path {st['segments'] / st['paths']:.2f} {best:7.1f} Mpaths/s "
              f"{best * st['segments'] / st['paths']:7.1f} Msegments/s", flush=True)
        ctx.close()
