/* scenes.cpp -- the scene table of the reference's `main` (src/main.rs:192-795 scene
 * functions, :815-937 table) written against the C ABI, the way the Rust host
 * would call it.  Same construction order, hence the same draws from the build
 * random stream as the reference's constructors make from `rng`. */
#include <vector>

#include "rt1w.h"
#include "scene.h"

namespace {

struct B { /* tiny builder: first error sticks */
    rt1w_scene* s;
    int err = RT1W_OK;
    int ck(int v) { if (v < 0 && err == RT1W_OK) err = v; return v; }
    int solid(double r, double g, double b) { double c[3] = {r, g, b}; return ck(rt1w_texture_solid(s, c)); }
    int lambert_solid(double r, double g, double b) { return ck(rt1w_material_lambertian(s, solid(r, g, b))); }
    int light(double r, double g, double b) { return ck(rt1w_material_diffuse_light(s, solid(r, g, b))); }
    int metal(double r, double g, double b, double fuzz) { double c[3] = {r, g, b}; return ck(rt1w_material_metal(s, c, fuzz)); }
    int glass(double ir) { return ck(rt1w_material_dielectric(s, ir)); }
    int sphere(double x, double y, double z, double r, int m) { double c[3] = {x, y, z}; return ck(rt1w_hittable_sphere(s, c, r, m)); }
    int xy(double a0, double a1, double b0, double b1, double k, int m) { return ck(rt1w_hittable_xy_rect(s, a0, a1, b0, b1, k, m)); }
    int xz(double a0, double a1, double b0, double b1, double k, int m) { return ck(rt1w_hittable_xz_rect(s, a0, a1, b0, b1, k, m)); }
    int yz(double a0, double a1, double b0, double b1, double k, int m) { return ck(rt1w_hittable_yz_rect(s, a0, a1, b0, b1, k, m)); }
    int box(double x0, double y0, double z0, double x1, double y1, double z1, int m) {
        double p0[3] = {x0, y0, z0}, p1[3] = {x1, y1, z1};
        return ck(rt1w_hittable_aabox(s, p0, p1, m));
    }
    int rotate_y(int c, double deg) { return ck(rt1w_hittable_rotate_y(s, c, 0.0, 1.0, deg)); }
    int translate(int c, double x, double y, double z) { double o[3] = {x, y, z}; return ck(rt1w_hittable_translate(s, c, o)); }
    int flip(int c) { return ck(rt1w_hittable_flip_face(s, c)); }
    int medium(int boundary, double d, int tex) { return ck(rt1w_hittable_constant_medium(s, boundary, d, tex)); }
    int bvh(const std::vector<int>& v) { return ck(rt1w_hittable_bvh(s, v.data(), (uint32_t)v.size(), 0.0, 1.0)); }
    double gen() { double x = 0; ck(rt1w_scene_rng_f64(s, &x)); return x; }
    double range(double lo, double hi) { double x = 0; ck(rt1w_scene_rng_range(s, lo, hi, &x)); return x; }
};

/* random_scene main.rs:192-295 */
int random_scene(B& b) {
    int checker = b.ck(rt1w_texture_checker(b.s, /*odd*/ b.solid(0.9, 0.9, 0.9), /*even*/ b.solid(0.2, 0.3, 0.1)));
    int ground_material = b.ck(rt1w_material_lambertian(b.s, checker));
    std::vector<int> world;
    world.push_back(b.sphere(0.0, -1000.0, 0.0, 1000.0, ground_material));
    for (int a = -11; a < 11; ++a) {
        for (int bb = -11; bb < 11; ++bb) {
            double choose_mat = b.gen();
            double cx = (double)a + 0.9 * b.gen();
            double cy = 0.2;
            double cz = (double)bb + 0.9 * b.gen();
            RtV3 dlt = rt_v3(cx, cy, cz) - rt_v3(4.0, 0.2, 0.0);
            if (rt_mag(dlt) > 0.9) {
                if (choose_mat < 0.8) {
                    double r0 = b.gen(), g0 = b.gen(), b0 = b.gen();
                    double r1 = b.gen(), g1 = b.gen(), b1 = b.gen();
                    double c2y = cy + b.range(0.0, 0.5);
                    int material = b.lambert_solid(r0 * r1, g0 * g1, b0 * b1);
                    double c0[3] = {cx, cy, cz}, c1[3] = {cx + 0.0, c2y, cz + 0.0};
                    world.push_back(b.ck(rt1w_hittable_moving_sphere(b.s, c0, c1, 0.0, 1.0, 0.2, material)));
                } else if (choose_mat < 0.95) {
                    double r = b.range(0.5, 1.0), g = b.range(0.5, 1.0), bl = b.range(0.5, 1.0);
                    double fuzz = b.range(0.5, 1.0);
                    world.push_back(b.sphere(cx, cy, cz, 0.2, b.metal(r, g, bl, fuzz)));
                } else {
                    world.push_back(b.sphere(cx, cy, cz, 0.2, b.glass(1.5)));
                }
            }
        }
    }
    world.push_back(b.sphere(0.0, 1.0, 0.0, 1.0, b.glass(1.5)));
    world.push_back(b.sphere(-4.0, 1.0, 0.0, 1.0, b.lambert_solid(0.4, 0.2, 0.1)));
    world.push_back(b.sphere(4.0, 1.0, 0.0, 1.0, b.metal(0.7, 0.6, 0.5, 0.0)));
    return b.bvh(world);
}

/* two_spheres main.rs:297-323 */
int two_spheres(B& b) {
    int checker = b.ck(rt1w_texture_checker(b.s, b.solid(0.9, 0.9, 0.9), b.solid(0.2, 0.3, 0.1)));
    int m = b.ck(rt1w_material_lambertian(b.s, checker));
    return b.bvh({b.sphere(0.0, -10.0, 0.0, 10.0, m), b.sphere(0.0, 10.0, 0.0, 10.0, m)});
}
/* two_perlin_spheres main.rs:325-344 */
int two_perlin_spheres(B& b) {
    int m = b.ck(rt1w_material_lambertian(b.s, b.ck(rt1w_texture_noise(b.s, 4.0))));
    return b.bvh({b.sphere(0.0, -1000.0, 0.0, 1000.0, m), b.sphere(0.0, 2.0, 0.0, 2.0, m)});
}
/* earth main.rs:346-358 */
int earth(B& b, const uint8_t* rgb, uint32_t w, uint32_t h) {
    int m = b.ck(rt1w_material_lambertian(b.s, b.ck(rt1w_texture_image(b.s, rgb, w, h))));
    return b.bvh({b.sphere(0.0, 0.0, 0.0, 2.0, m)});
}
/* simple_light main.rs:360-393 */
int simple_light(B& b) {
    int pertext = b.ck(rt1w_material_lambertian(b.s, b.ck(rt1w_texture_noise(b.s, 4.0))));
    int difflight = b.light(4.0, 4.0, 4.0);
    return b.bvh({b.sphere(0.0, -1000.0, 0.0, 1000.0, pertext), b.sphere(0.0, 2.0, 0.0, 2.0, pertext),
                  b.xy(3.0, 5.0, 1.0, 3.0, -2.0, difflight)});
}
/* cornel_box main.rs:395-512 */
int cornel_box(B& b) {
    int red = b.lambert_solid(0.65, 0.05, 0.05);
    int white = b.lambert_solid(0.73, 0.73, 0.73);
    int green = b.lambert_solid(0.12, 0.45, 0.15);
    int light = b.light(15.0, 15.0, 15.0);
    int aluminum = b.metal(0.8, 0.85, 0.88, 0.0);
    int box1 = b.box(0.0, 0.0, 0.0, 165.0, 330.0, 165.0, aluminum);
    box1 = b.rotate_y(box1, 15.0);
    box1 = b.translate(box1, 265.0, 0.0, 295.0);
    int grass = b.glass(1.5);
    std::vector<int> world = {
        b.yz(0.0, 555.0, 0.0, 555.0, 555.0, green),
        b.yz(0.0, 555.0, 0.0, 555.0, 0.0, red),
        b.flip(b.xz(213.0, 343.0, 227.0, 332.0, 554.0, light)),
        b.xz(0.0, 555.0, 0.0, 555.0, 0.0, white),
        b.xz(0.0, 555.0, 0.0, 555.0, 555.0, white),
        b.xy(0.0, 555.0, 0.0, 555.0, 555.0, white),
        box1,
        b.sphere(190.0, 90.0, 190.0, 90.0, grass),
    };
    return b.bvh(world);
}
/* cornel_smoke main.rs:514-633 */
int cornel_smoke(B& b) {
    int red = b.lambert_solid(0.65, 0.05, 0.05);
    int white = b.lambert_solid(0.73, 0.73, 0.73);
    int green = b.lambert_solid(0.12, 0.45, 0.15);
    int light = b.light(7.0, 7.0, 7.0);
    int box1 = b.box(0.0, 0.0, 0.0, 165.0, 330.0, 165.0, white);
    box1 = b.translate(b.rotate_y(box1, 15.0), 265.0, 0.0, 295.0);
    int box2 = b.box(0.0, 0.0, 0.0, 165.0, 165.0, 165.0, white);
    box2 = b.translate(b.rotate_y(box2, -18.0), 130.0, 0.0, 65.0);
    int smoke1 = b.medium(box1, 0.01, b.solid(0.0, 0.0, 0.0));
    int smoke2 = b.medium(box2, 0.01, b.solid(1.0, 1.0, 1.0));
    std::vector<int> world = {
        b.yz(0.0, 555.0, 0.0, 555.0, 555.0, green),
        b.yz(0.0, 555.0, 0.0, 555.0, 0.0, red),
        b.flip(b.xz(113.0, 443.0, 127.0, 432.0, 554.0, light)),
        b.xz(0.0, 555.0, 0.0, 555.0, 0.0, white),
        b.xz(0.0, 555.0, 0.0, 555.0, 555.0, white),
        b.xy(0.0, 555.0, 0.0, 555.0, 555.0, white),
        smoke1,
        smoke2,
    };
    return b.bvh(world);
}
/* final_scene main.rs:635-795 */
int final_scene(B& b, const uint8_t* rgb, uint32_t w, uint32_t h) {
    int ground = b.lambert_solid(0.48, 0.83, 0.53);
    const int BOXES_PER_SIDE = 20;
    std::vector<int> boxes1;
    for (int i = 0; i < BOXES_PER_SIDE; ++i) {
        for (int j = 0; j < BOXES_PER_SIDE; ++j) {
            double wd = 100.0;
            double x0 = -1000.0 + (double)i * wd;
            double z0 = -1000.0 + (double)j * wd;
            double y0 = 0.0;
            double x1 = x0 + wd;
            double y1 = b.range(1.0, 101.0);
            double z1 = z0 + wd;
            boxes1.push_back(b.box(x0, y0, z0, x1, y1, z1, ground));
        }
    }
    std::vector<int> objects;
    objects.push_back(b.bvh(boxes1));
    int light = b.light(7.0, 7.0, 7.0);
    objects.push_back(b.flip(b.xz(123.0, 423.0, 147.0, 412.0, 554.0, light)));
    double center1[3] = {400.0, 400.0, 200.0};
    double center2[3] = {400.0 + 30.0, 400.0 + 0.0, 200.0 + 0.0};
    int moving_sphere_material = b.lambert_solid(0.7, 0.3, 0.1);
    objects.push_back(b.ck(rt1w_hittable_moving_sphere(b.s, center1, center2, 0.0, 1.0, 50.0, moving_sphere_material)));
    objects.push_back(b.sphere(260.0, 150.0, 45.0, 50.0, b.glass(1.5)));
    objects.push_back(b.sphere(0.0, 150.0, 145.0, 50.0, b.metal(0.8, 0.8, 0.9, 1.0)));
    int boundary = b.sphere(360.0, 150.0, 145.0, 70.0, b.glass(1.5));
    objects.push_back(b.sphere(360.0, 150.0, 145.0, 70.0, b.glass(1.5)));
    objects.push_back(b.medium(boundary, 0.2, b.solid(0.2, 0.4, 0.9)));
    boundary = b.sphere(0.0, 0.0, 0.0, 5000.0, b.glass(1.5));
    objects.push_back(b.medium(boundary, 0.0001, b.solid(1.0, 1.0, 1.0)));
    int emat = b.ck(rt1w_material_lambertian(b.s, b.ck(rt1w_texture_image(b.s, rgb, w, h))));
    objects.push_back(b.sphere(400.0, 200.0, 400.0, 100.0, emat));
    int pertext = b.ck(rt1w_material_lambertian(b.s, b.ck(rt1w_texture_noise(b.s, 0.1))));
    objects.push_back(b.sphere(220.0, 280.0, 300.0, 80.0, pertext));
    std::vector<int> boxes2;
    int white = b.lambert_solid(0.73, 0.73, 0.73);
    const int ns = 1000;
    for (int i = 0; i < ns; ++i) {
        double x = b.range(0.0, 165.0), y = b.range(0.0, 165.0), z = b.range(0.0, 165.0);
        boxes2.push_back(b.sphere(x, y, z, 10.0, white));
    }
    int rot = b.rotate_y(b.bvh(boxes2), 15.0);
    objects.push_back(b.translate(rot, -100.0, 270.0, 395.0));
    return b.bvh(objects);
}

} // namespace

extern "C" int rt1w_scene_build_reference(int arm, uint64_t build_seed, double aspect_ratio,
                                          const uint8_t* earth_rgb8, uint32_t earth_w, uint32_t earth_h,
                                          rt1w_scene** out, uint32_t defaults[3]) {
    if (!out) { rt1w::set_error("null out"); return RT1W_ERR_INVALID; }
    bool needs_earth = (arm == 3) || (arm < 0 || arm > 6);
    if (needs_earth && (!earth_rgb8 || !earth_w || !earth_h)) {
        rt1w::set_error("this scene arm needs the decoded earth texture (RGB8)");
        return RT1W_ERR_INVALID;
    }
    rt1w_scene* s = nullptr;
    int rc = rt1w_scene_create(build_seed, &s);
    if (rc < 0) return rc;
    B b{s};
    /* main.rs:798-800 defaults, overridden per arm */
    uint32_t image_width = 400, samples_per_pixel = 100;
    double bg[3] = {0.70, 0.80, 1.00};
    double look_from[3] = {13.0, 2.0, 3.0}, look_at[3] = {0.0, 0.0, 0.0};
    double vfov = 20.0, aperture = 0.0;
    std::vector<int> lights;
    int null_mat = b.ck(rt1w_material_null(s)); /* main.rs:805 */
    int world = -1;
    switch (arm) {
        case 0: samples_per_pixel = 500; world = random_scene(b); aperture = 0.1; break;        /* main.rs:816-827 */
        case 1: world = two_spheres(b); break;                                                  /* :828-836 */
        case 2: world = two_perlin_spheres(b); break;                                           /* :837-845 */
        case 3: world = earth(b, earth_rgb8, earth_w, earth_h); break;                          /* :846-854 */
        case 4:                                                                                 /* :855-866 */
            samples_per_pixel = 400; world = simple_light(b);
            bg[0] = bg[1] = bg[2] = 0.0;
            look_from[0] = 26.0; look_from[1] = 3.0; look_from[2] = 6.0; look_at[1] = 2.0;
            break;
        case 5:                                                                                 /* :867-894 */
            image_width = 600; samples_per_pixel = 100; world = cornel_box(b);
            lights.push_back(b.xz(213.0, 343.0, 227.0, 332.0, 554.0, null_mat));
            lights.push_back(b.sphere(190.0, 90.0, 190.0, 90.0, null_mat));
            bg[0] = bg[1] = bg[2] = 0.0;
            look_from[0] = 278.0; look_from[1] = 278.0; look_from[2] = -800.0;
            look_at[0] = 278.0; look_at[1] = 278.0; look_at[2] = 0.0;
            vfov = 40.0;
            break;
        case 6:                                                                                 /* :895-915 */
            image_width = 600; samples_per_pixel = 200; world = cornel_smoke(b);
            lights.push_back(b.xz(113.0, 443.0, 127.0, 432.0, 554.0, null_mat));
            bg[0] = bg[1] = bg[2] = 0.0;
            look_from[0] = 278.0; look_from[1] = 278.0; look_from[2] = -800.0;
            look_at[0] = 278.0; look_at[1] = 278.0; look_at[2] = 0.0;
            vfov = 40.0;
            break;
        default:                                                                                /* :916-936 */
            image_width = 800; samples_per_pixel = 10000; world = final_scene(b, earth_rgb8, earth_w, earth_h);
            lights.push_back(b.xz(123.0, 423.0, 147.0, 412.0, 554.0, null_mat));
            bg[0] = bg[1] = bg[2] = 0.0;
            look_from[0] = 478.0; look_from[1] = 278.0; look_from[2] = -600.0;
            look_at[0] = 278.0; look_at[1] = 278.0; look_at[2] = 0.0;
            vfov = 40.0;
            break;
    }
    double vup[3] = {0.0, 1.0, 0.0};
    if (b.err == RT1W_OK) b.ck(rt1w_scene_set_world(s, world));
    if (b.err == RT1W_OK) b.ck(rt1w_scene_set_lights(s, lights.data(), (uint32_t)lights.size()));
    if (b.err == RT1W_OK) b.ck(rt1w_scene_set_background(s, bg));
    /* Camera::new(look_from, look_at, vup, vfov, aspect_ratio, aperture, 10.0, 0.0, 1.0) main.rs:941-951 */
    if (b.err == RT1W_OK) b.ck(rt1w_scene_set_camera(s, look_from, look_at, vup, vfov, aspect_ratio, aperture, 10.0, 0.0, 1.0));
    if (b.err == RT1W_OK) b.ck(rt1w_scene_commit(s));
    if (b.err != RT1W_OK) { rt1w_scene_destroy(s); return b.err; }
    if (defaults) {
        defaults[0] = image_width;
        defaults[1] = (uint32_t)((double)image_width / aspect_ratio); /* main.rs:939 */
        defaults[2] = samples_per_pixel;
    }
    *out = s;
    return RT1W_OK;
}
