/* rt_flat.h -- the flattened (SoA-of-records) scene the device kernel walks.
 *
 * The reference keeps the scene as a tree of `Box<dyn Hittable>` /
 * `Arc<Box<dyn Material>>` / `Box<dyn Texture>` objects (src/hittable.rs:63-72,
 * src/material.rs:25-50, src/texture.rs:8-10).  The host builds the same graph
 * (scene_graph.h), then flattens it ONCE into the plain arrays below, which are
 * uploaded to HBM at context creation and never change.
 *
 * Record sizes are multiples of 16 B so one lane fetches a record with
 * dwordx4 loads.
 */
#ifndef RT1W_FLAT_H
#define RT1W_FLAT_H

#include "rt1w_num.h"

#define RT_NONE 0xFFFFFFFFu

/* node kinds: every `impl Hittable` of the reference */
enum {
    RT_BVH2 = 0,      /* BVHChild::Two   src/bvh.rs:11,38-47   d[0..5]=aabb min,max  a=left b=right */
    RT_BVH1 = 1,      /* BVHChild::One   src/bvh.rs:10,37      d[0..5]=aabb          a=child */
    RT_SPHERE = 2,    /* src/sphere.rs:16-20      d[0..2]=center d[3]=radius  mat */
    RT_MSPHERE = 3,   /* src/moving_sphere.rs:13-20 d[0..2]=c0 d[3..5]=c1 e[0]=t0 e[1]=t1 e[2]=radius mat */
    RT_XY = 4,        /* src/aarect.rs:15-22  d[0..4]=x0,x1,y0,y1,k mat */
    RT_XZ = 5,        /* src/aarect.rs:25-32  d[0..4]=x0,x1,z0,z1,k mat */
    RT_YZ = 6,        /* src/aarect.rs:35-42  d[0..4]=y0,y1,z0,z1,k mat */
    RT_TRANSLATE = 7, /* src/hittable.rs:49-52   d[0..2]=offset   a=child b=parent scope */
    RT_ROTATE_Y = 8,  /* src/hittable.rs:54-59   d[0]=sin d[1]=cos a=child b=parent scope */
    RT_FLIP = 9,      /* src/hittable.rs:61      a=child b=parent scope */
    RT_MEDIUM = 10,   /* src/constant_medium.rs:15-19 d[0]=neg_inv_density mat=phase fn a=boundary root */
    RT_DEFAULT = 11,  /* lights table only: a hittable without pdf_value/random overrides
                         (src/hittable.rs:66-71 defaults) */
    RT_KIND_MASK = 0xFF,
    /* OPT-IN walk order (rt1w_scene_set_walk_order; all zero by default = the reference's left-then-right, bvh.rs:38-47):
     * bits 9-10 of a BVH2 node = 1 + the axis along which its children are furthest apart, bit 11 = the left child is the
     * lower one on that axis.  The stack walk then visits the child on the ray's near side first. */
    RT_BVH_ORDER_SHIFT = 9, RT_BVH_ORDER_MASK = 3, RT_BVH_LEFT_LOWER = 0x800,
    RT_LEAF_FLIPPED = 0x100 /* leaf wrapped directly by FlipFace (hittable.rs:286-292): the flattener folds the
                               wrapper into this flag; the flip is applied right after the leaf's own record,
                               i.e. exactly where the innermost wrapper's fix-up would run */
};

/* Layout: everything a traversal step reads (kind, skip, a whole AABB / rect / sphere / wrapper,
 * the right child and the material) is the first 64 bytes -- one s_load_dwordx16 when the index is
 * wave-uniform (sweep), four dwordx4 when it is per-lane (stack).  The left child of a BVH node, the
 * child of a wrapper and the boundary of a medium are always the next node (pre-order). */
struct RtNode {
    uint32_t kind;
    uint32_t skip; /* nodes are stored in depth-first pre-order: [index, skip) is this node's subtree */
    double d[6];
    uint32_t b;    /* BVH2: right child; wrapper / leaf / medium: the wrapper above it (RT_NONE at top level) */
    uint32_t mat;
    double e[3];   /* MovingSphere only: time0, time1, radius */
    uint32_t a;    /* == index + 1 for BVH / wrapper / medium nodes */
    uint32_t pad;
}; /* 96 bytes */
/* the first 64 bytes of an RtNode */
struct RtNodeHot {
    uint32_t kind;
    uint32_t skip;
    double d[6];
    uint32_t b;
    uint32_t mat;
};

/* material kinds: every `impl Material` */
enum {
    RT_MAT_NULL = 0,         /* impl Material for ()      src/material.rs:68 */
    RT_MAT_LAMBERTIAN = 1,   /* src/material.rs:70-92   tex */
    RT_MAT_METAL = 2,        /* src/material.rs:98-112  d[0..2]=albedo d[3]=fuzz */
    RT_MAT_DIELECTRIC = 3,   /* src/material.rs:132-161 d[0]=ir */
    RT_MAT_DIFFUSE_LIGHT = 4,/* src/material.rs:163-182 tex */
    RT_MAT_ISOTROPIC = 5     /* src/constant_medium.rs:31-51 tex */
};
#define RT_MAT_NEEDS_UV 0x100u /* texture tree of this material reads (u,v): image texture */
#define RT_MAT_SOLID 0x200u    /* the material's texture is a SolidColor and its colour is also in d[0..2] */
/* a node's `mat` = material index in the low 16 bits, that material's kind word (kind | flags) above: the path class of a
 * hit and the u,v question are answered by the node record alone, without a dependent fetch of the material */
#define RT_MAT_INDEX(m) ((m) & 0xFFFFu)
#define RT_MAT_KINDF(m) ((m) >> 16)

struct RtMaterial {
    double d[4];
    uint32_t kind; /* low 8 bits kind, RT_MAT_NEEDS_UV / RT_MAT_SOLID flags */
    uint32_t tex;
    uint32_t pad0, pad1;
}; /* 48 bytes */

/* texture kinds: every `impl Texture` used by a scene */
enum {
    RT_TEX_SOLID = 0,   /* src/texture.rs:40-44   d=rgb */
    RT_TEX_CHECKER = 1, /* src/texture.rs:46-55   a=odd b=even */
    RT_TEX_NOISE = 2,   /* src/texture.rs:57-65   d[0]=scale a=perlin table index */
    RT_TEX_IMAGE = 3    /* src/texture.rs:67-89   a=width b=height c=byte offset into image pool */
};
struct RtTexture {
    double d[3];
    uint32_t kind, a, b, c;
    uint32_t pad0, pad1;
}; /* 48 bytes */

/* src/perlin.rs:8-13 with POINT_COUNT = 256 */
struct RtPerlin {
    double ranvec[256 * 3];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
};

/* src/camera.rs:7-18 */
struct RtCamera {
    RtV3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    double lens_radius, time0, time1;
};

/* what the kernel sees (device pointers on the GPU, host pointers in the CPU
 * test build of the same core) */
struct RtSceneView {
    const RtNode* nodes;
    const RtNode* lights;
    const RtMaterial* materials;
    const RtTexture* textures;
    const RtPerlin* perlin;
    const uint8_t* images;
    uint32_t root;
    uint32_t n_nodes;
    uint32_t n_lights;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t pad;
    RtCamera camera;
    RtV3 background;
};

/* one render call (mirrors the loop bounds of src/main.rs:957-992) */
struct RtFrame {
    uint32_t width, height;       /* full image, src/main.rs:799,939 */
    uint32_t x0, y0, tile_w, tile_h; /* tile rendered by this call; y counts reference rows j */
    uint32_t spp;                 /* samples rendered by this call */
    uint32_t sample_offset;       /* first absolute sample index (sample-range sharding) */
    uint32_t max_depth;           /* MAX_DEPTH, src/main.rs:801 */
    uint32_t global_seed;
    uint32_t chunk;               /* samples per work item; pixel sum = sum over chunks of chunk sums */
    uint32_t n_chunks;
    /* row-interleaved tile (image tiling over GPUs: strips of `strip_rows` image rows dealt round-robin): tile row r is
     * image row y0 + (r / strip_rows) * strip_period + r % strip_rows.  strip_rows == 0: contiguous rows y0 + r. */
    uint32_t strip_rows, strip_period;
    /* MEASUREMENT MODE (RT1W_PROBE_COHERENT; the frame is NOT rendered): the 64 lanes of a wave are given the SAME work item -- one pixel
     * of every 8x8 block -- so that a wave executes exactly the instructions ONE path needs in this kernel: its VALU instruction count
     * per traced segment is the necessary work the `roofline.valu` block of bench.py sets the executed work against */
    uint32_t probe;
};
/* image row j (the reference's row index, main.rs:957-964) of tile row py */
RT_HD uint32_t rt_frame_row(const RtFrame& f, uint32_t py) {
    if (f.strip_rows == 0u) return f.y0 + py;
    const uint32_t k = py / f.strip_rows;
    return f.y0 + k * f.strip_period + (py - k * f.strip_rows);
}

#define RT_STACK_CAP 32      /* traversal stack entries per lane (LDS) */
#define RT_SWEEP_MAX_NODES 64 /* scenes up to this many nodes use the stackless wave-uniform sweep */
/* scenes up to this many nodes can get a kernel with the sweep unrolled along their own tree (jit.cpp).  Measured on
 * Cornell boxes with N rotated boxes against the stack walk: 77 nodes 2.3x, 141 2.1x, 205 1.7x, 273 1.5x, 529 1.1x;
 * the compile takes 1.7 s at 77 nodes, 8 s at 273, 23 s at 529 */
#define RT_JIT_MAX_NODES 256

/* compile-time feature set of a kernel variant: code for absent features is not
 * generated, which is what keeps the register budget of the simple scenes low */
/* `Topo_`: void, or a type with the scene's node kinds and subtree ends as compile-time arrays (rt_sweep_static) */
template <bool MEDIA_, bool TEX_, bool MSPHERE_, bool SWEEP_, int SCOPE_DEPTH_ = 3, class Topo_ = void, bool ORDERED_ = false>
struct RtCfg {
    static constexpr bool ordered = ORDERED_; /* stack walk honours the opt-in near-far bits of BVH2 nodes (rt1w_scene_set_walk_order) */
    typedef Topo_ Topo;
    static constexpr int scope_depth = SCOPE_DEPTH_; /* deepest wrapper nesting the variant handles; 0: the scene has no Translate / RotateY / FlipFace
                                                        node at all, and the walk carries neither a wrapper's ray nor its entry/exit code */
    static constexpr bool media = MEDIA_;     /* scene contains ConstantMedium nodes */
    static constexpr bool tex = TEX_;         /* scene has non-solid textures (checker/noise/image) */
    static constexpr bool msphere = MSPHERE_; /* scene has MovingSphere primitives */
    static constexpr bool sweep = SWEEP_;     /* stackless pre-order sweep instead of the LDS stack */
    static constexpr bool sphere_media = false; /* see RtCfgSphereMedia */
};
/* A variant for scenes whose every ConstantMedium is bounded by a bare Sphere (final_scene: main.rs:730-752): the kernel then
 * carries the two sphere roots of constant_medium.rs:62-69 only, not the two complete boundary WALKS a general boundary needs --
 * 1400 instructions and two more walk states that the stack walk's loop otherwise holds for nothing (final_scene +5 %).  Chosen
 * by the host from the flattened scene (context.hip); the same arithmetic on the path that runs, so the same bits. */
template <class Base>
struct RtCfgSphereMedia : Base {
    static constexpr bool sphere_media = true;
};
/* the variants that are built (host picks the cheapest one that covers the scene) */
typedef RtCfg<false, false, false, true, 2> RtCfgV0; /* small scene, solid colours only, no media, no moving spheres (Cornell box) */
typedef RtCfg<true, true, true, true> RtCfgV1;    /* small scene, every feature */
typedef RtCfg<false, true, true, false> RtCfgV2;  /* large scene without media (random_scene) */
typedef RtCfg<true, true, true, false> RtCfgV3;   /* large scene, every feature (final_scene) */
typedef RtCfg<true, true, true, false, 3, void, true> RtCfgV4; /* V3 + the opt-in near-far walk order (only scenes that asked for it) */
typedef RtCfg<false, true, true, false, 0> RtCfgV5; /* large scene without media and without wrappers (random_scene): one ray space */
#define RT_N_VARIANTS 6
/* cheapest valid variant for a scene; `force` >= 0 overrides when valid */
inline int rt_pick_variant(uint32_t n_nodes, bool media, bool tex, bool msphere, uint32_t scope_depth, bool ordered = false) {
    if (n_nodes <= RT_SWEEP_MAX_NODES) return (!media && !tex && !msphere && scope_depth <= 2u) ? 0 : 1;
    if (ordered) return 4;
    if (!media && scope_depth == 0u) return 5;
    return media ? 3 : 2;
}
inline bool rt_variant_valid(int v, uint32_t n_nodes, bool media, bool tex, bool msphere, uint32_t scope_depth) {
    (void)n_nodes;
    switch (v) {
        case 0: return !media && !tex && !msphere && scope_depth <= 2u;
        case 1: return true;
        case 2: return !media;
        case 3: return true;
        case 4: return true; /* reference order unless the scene's nodes carry order bits */
        case 5: return !media && scope_depth == 0u;
        default: return false;
    }
}
#define RT_MAX_SCOPE_DEPTH 3 /* nested Translate/RotateY/FlipFace wrappers above a primitive */

#endif
