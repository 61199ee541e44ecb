"""Generates the committed golden fixtures.  Run in the build container only:
    python tests/golden/make_golden.py
(0) pixel-level fixture of the reference's own render rest_of_your_life.png (hash, row checksums, a crop);
(1) block means of the reference's own artefacts rest_of_your_life.png (Cornell, master) and next_week.png
    (final_scene, an earlier revision) read from /root/reference -- the only outputs of the reference's own runs
    that match a scene arm of master (one_weekend.png is the book-1 scene: gradient sky, no checker, no motion blur);
(2) golden framebuffers of the literal CPU oracle (oracle/oracle.cpp) for small
    configurations of every scene arm, incl. BASELINE config C1 (Cornell 200x200x64);
(3) block means of the same oracle built with `type Float = f32` (oracle/oracle_f32.cpp) -- `make_golden.py f32` makes only these.
The reference (Rust) cannot be run here, so (2) are oracle outputs, not reference
outputs; (1) is the reference's.
"""
import hashlib, json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc

def png_blocks():
    from PIL import Image
    im = np.asarray(Image.open('/root/reference/rest_of_your_life.png').convert('RGB'), dtype=np.float64)
    assert im.shape == (600, 600, 3)
    b = im.reshape(6, 100, 6, 100, 3).mean(axis=(1, 3))  # [block_row(top first)][block_col][rgb]
    return {"source": "rest_of_your_life.png (hatoo/raytracing-1w master, README.md:19)", "shape": [600, 600],
            "block": 100, "channel_means": im.mean(axis=(0, 1)).tolist(), "block_means_top_down": b.tolist()}

def png_blocks_final():
    """next_week.png = the reference's own render of the final_scene arm (README.md:15), made by an EARLIER revision:
    its light emits from both faces (master: front face only, material.rs:168-181 + FlipFace, main.rs:655-662), so the
    fog-lit region above the light plane (image rows 0-1) is brighter than master's, and the ground boxes / the cube of
    small spheres are placed by thread_rng.  Pinned here: the blocks below the light plane that hold neither saturated
    pixels nor randomly placed geometry -- linear means (the PNG's 8-bit gamma values mapped back through color.rs:56-65)."""
    from PIL import Image
    im = np.asarray(Image.open('/root/reference/next_week.png').convert('RGB'), dtype=np.float64)
    assert im.shape == (800, 800, 3)
    lin = ((im + 0.5) / 256.0) ** 2
    b = lin.reshape(8, 100, 8, 100, 3).mean(axis=(1, 3))
    sel = [(2, 0), (2, 1), (2, 2), (2, 3), (2, 5), (2, 6), (2, 7), (3, 0), (3, 1), (3, 2), (3, 3), (3, 6), (3, 7),
           (4, 0), (4, 1), (4, 2), (4, 3), (4, 4), (4, 5), (4, 6), (4, 7)]
    # Objects whose geometry and appearance are LITERALS of final_scene (no thread_rng, no Perlin table): the earth sphere
    # (main.rs:747-753: centre (400,200,400), r 100, the image texture), the moving sphere (main.rs:693-707: (400,400,200) ->
    # (430,400,200), r 50, lambertian (0.7,0.3,0.1)) and the blue subsurface ball (main.rs:715-733: glass sphere (360,150,145)
    # r 70 filled with a density-0.2 medium).  Their projected discs (camera main.rs:931-934: from (478,278,-600) at (278,278,0),
    # vfov 40, square) shrunk to 0.7 of the radius, the PNG's linear mean over each disc, and the PNG's own noise there:
    # sigma_pixel = rms of (pixel - mean of its 3x3 neighbourhood) * sqrt(9/8), so sigma of the disc mean = sigma_pixel / sqrt(n).
    look_from, look_at, vup = np.array([478.0, 278.0, -600.0]), np.array([278.0, 278.0, 0.0]), np.array([0.0, 1.0, 0.0])
    w = (look_from - look_at) / np.linalg.norm(look_from - look_at)
    u = np.cross(vup, w); u /= np.linalg.norm(u)
    v = np.cross(w, u)
    half = np.tan(np.radians(40.0) / 2.0)
    discs = []
    for name, centre, radius in (("earth sphere", (400.0, 200.0, 400.0), 100.0), ("moving sphere", (415.0, 400.0, 200.0), 50.0),
                                 ("blue subsurface ball", (360.0, 150.0, 145.0), 70.0)):
        d = np.array(centre) - look_from
        depth = -np.dot(d, w)
        cx = (0.5 + np.dot(d, u) / (2.0 * half * depth)) * 799.0
        cy_up = (0.5 + np.dot(d, v) / (2.0 * half * depth)) * 799.0
        rad = 0.7 * radius / (2.0 * half * depth) * 800.0 - (15.0 if name == "moving sphere" else 0.0) * 800.0 / (2.0 * half * depth)
        cy = 799.0 - cy_up
        yy, xx = np.mgrid[0:800, 0:800]
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad ** 2
        k = np.ones((3, 3)) / 9.0
        loc = sum(np.roll(np.roll(lin, dy, 0), dx, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)) / 9.0
        resid = (lin - loc)[m]
        sigma_pixel = np.sqrt((resid ** 2).mean(axis=0) * 9.0 / 8.0)
        discs.append({"name": name, "cx": float(cx), "cy_top_down": float(cy), "radius_px": float(rad), "pixels": int(m.sum()),
                      "png_linear_mean": lin[m].mean(axis=0).tolist(), "png_sigma_of_mean": (sigma_pixel / np.sqrt(m.sum())).tolist(),
                      "saturated_fraction": float((im[m] >= 255).any(axis=1).mean())})
    return {"source": "next_week.png (hatoo/raytracing-1w, README.md:15; rendered by an earlier revision, see make_golden.py)",
            "shape": [800, 800], "block": 100, "linear_block_means_top_down": b.tolist(), "selected_blocks": sel,
            "literal_object_discs": discs}

def png_pixels():
    """Pixel-level pin: rest_of_your_life.png IS what `cargo run` of master prints (Cornell arm, 600x600, 100 spp,
    main.rs:868-870), converted losslessly from the P3 text.  The per-pixel stream is deterministic (main.rs:964) and the
    scene has no build-time randomness that reaches a pixel, so the literal oracle in reference-stream mode
    (oracle/refstream.h: ChaCha12 + rand 0.8.4 shapes + libm) must reproduce it pixel for pixel.  Committed: sha256 of
    the RGB bytes (rows top-down), crc32 of every row, and a 128x128 crop (sphere, box edge, floor, wall)."""
    import zlib
    from PIL import Image
    im = np.ascontiguousarray(np.asarray(Image.open('/root/reference/rest_of_your_life.png').convert('RGB'), dtype=np.uint8))
    assert im.shape == (600, 600, 3)
    x0, y0 = 150, 330   # top-down coordinates of the crop
    np.save(os.path.join(HERE, 'cornell_png_crop.npy'), im[y0:y0 + 128, x0:x0 + 128])
    return {"source": "rest_of_your_life.png (hatoo/raytracing-1w master, README.md:19)", "shape": [600, 600, 3], "spp": 100,
            "depth": 50, "arm": 5, "sha256_rgb_top_down": hashlib.sha256(im.tobytes()).hexdigest(),
            "row_crc32_top_down": [zlib.crc32(im[r].tobytes()) for r in range(600)], "crop_top_down": [x0, y0, 128, 128]}

CASES = {  # name: (arm, W, H, spp, depth)
    "c1_cornell_200x200x64": (5, 200, 200, 64, 50),
    "random_scene_96x64x8": (0, 96, 64, 8, 50),
    "two_spheres_64x36x8": (1, 64, 36, 8, 50),
    "two_perlin_64x36x8": (2, 64, 36, 8, 50),
    "earth_64x36x8": (3, 64, 36, 8, 50),
    "simple_light_64x36x16": (4, 64, 36, 16, 50),
    "cornel_smoke_64x64x16": (6, 64, 64, 16, 50),
    "final_scene_64x64x16": (7, 64, 64, 16, 50),
    "cornell_depth3_48x48x8": (5, 48, 48, 8, 3),
}

F32_CASES = {  # name: (arm, W, H, spp, block, aspect)
    "cornell_600x600x256": (5, 600, 600, 256, 100, None),
    "random_scene_300x200x64": (0, 300, 200, 64, 50, 1.5),
    "final_scene_200x200x64": (7, 200, 200, 64, 50, None),
}

def f32_frames():
    """(3) Block means of the literal oracle built with the reference's precision switch thrown, `type Float = f32` (main.rs:1;
    oracle/oracle_f32.cpp): the expected side of the product's RT1W_PRECISION_F32 mode.  Linear means per block and channel, NaN
    pixels (zeroed by into_sampled, color.rs:16-18, before they get here) counted through their zeros."""
    out = {}
    for name, (arm, W, H, spp, blk, aspect) in F32_CASES.items():
        img, st = orc.OracleScene(arm, build_seed=1, aspect_ratio=aspect, f32=True).render(W, H, spp)
        bm = img.reshape(H // blk, blk, W // blk, blk, 3).mean(axis=(1, 3))
        out[name] = {"arm": arm, "W": W, "H": H, "spp": spp, "block": blk, "aspect": aspect, "segments": st["segments"],
                     "mean": float(img.mean()), "channel_means": img.mean(axis=(0, 1)).tolist(),
                     "zero_channels": int((img == 0.0).sum()), "block_means_bottom_up": bm.tolist()}
        print(name, st["segments"], out[name]["mean"])
    with open(os.path.join(HERE, 'oracle_f32_blocks.json'), 'w') as f:
        json.dump({"build_seed": 1, "global_seed": 0, "source": "oracle/oracle_f32.cpp (liborc_f32.so)", "cases": out}, f, indent=1)

def main():
    if sys.argv[1:] == ["f32"]:
        return f32_frames()
    f32_frames()
    with open(os.path.join(HERE, 'cornell_png_blocks.json'), 'w') as f:
        json.dump(png_blocks(), f, indent=1)
    with open(os.path.join(HERE, 'final_scene_png_blocks.json'), 'w') as f:
        json.dump(png_blocks_final(), f, indent=1)
    with open(os.path.join(HERE, 'cornell_png_pixels.json'), 'w') as f:
        json.dump(png_pixels(), f, indent=1)
    meta = {}
    arrays = {}
    for name, (arm, W, H, spp, depth) in CASES.items():
        sc = orc.OracleScene(arm, build_seed=1)
        img, st = sc.render(W, H, spp, max_depth=depth)
        if img.size > 96 * 64 * 3:   # keep the repo small: centre crop + block means + hash of the full buffer
            y0, x0 = H // 2 - 16, W // 2 - 16
            arrays[name + "__crop"] = img[y0:y0 + 32, x0:x0 + 32].copy()
            bm = img.reshape(H // 20, 20, W // 20, 20, 3).mean(axis=(1, 3))
            arrays[name + "__block20"] = bm
            meta[name] = {"arm": arm, "W": W, "H": H, "spp": spp, "depth": depth, "crop": [x0, y0, 32, 32]}
        else:
            arrays[name] = img
            meta[name] = {"arm": arm, "W": W, "H": H, "spp": spp, "depth": depth}
        meta[name]["sha256_f64"] = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()
        meta[name]["segments"] = st["segments"]
        meta[name]["mean"] = float(np.nanmean(img))
        print(name, meta[name]["segments"], meta[name]["mean"])
    np.savez_compressed(os.path.join(HERE, 'oracle_frames.npz'), **arrays)
    with open(os.path.join(HERE, 'oracle_frames.json'), 'w') as f:
        json.dump({"build_seed": 1, "global_seed": 0, "cases": meta}, f, indent=1)

if __name__ == '__main__':
    main()
