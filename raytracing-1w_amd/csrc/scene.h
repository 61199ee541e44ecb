/* scene.h -- host-side scene graph of librt1w: the constructors of the
 * reference's Hittable / Material / Texture implementors, BVHNode::new, and the
 * flattener that turns the graph into the record arrays of rt_flat.h.
 *
 * Host code here only BUILDS (bounding boxes, BVH, camera frame).  It has no
 * `hit`/`scatter`: intersection and shading exist only in the device kernel
 * (rt_core.h compiled by hipcc) -- there is no CPU render path in the library.
 */
#ifndef RT1W_SCENE_H
#define RT1W_SCENE_H

#include <map>
#include <string>
#include <vector>

#include "rt1w.h"
#include "rt_flat.h"

namespace rt1w {

struct AABB { RtV3 minimum, maximum; }; /* aabb.rs:7-10 */

/* host-only kinds on top of RT_* */
enum { H_AABOX = 100, H_BVH = 101 };

struct HostHittable {
    uint32_t kind = RT_DEFAULT;
    double d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int mat = -1;
    int child = -1;          /* wrapper child / medium boundary / AABox side BVH */
    int left = -1, right = -1; /* H_BVH: BVHChild::One -> left only */
    AABB box{};              /* H_BVH aabb, RotateY cached aabb, AABox (box_min, box_max) */
    bool has_box = false;
    bool used = false;       /* already owned by a parent (Box ownership) */
};

void set_error(const std::string& msg);

} // namespace rt1w

struct rt1w_scene {
    RtRng rng;
    bool committed = false;
    std::vector<RtTexture> textures;
    std::vector<RtPerlin> perlin;
    std::vector<uint8_t> images;
    std::vector<RtMaterial> materials;
    std::vector<rt1w::HostHittable> hittables;
    int world = -1;
    std::vector<int> lights;
    RtV3 background{0, 0, 0};
    RtCamera camera{};
    bool has_camera = false;
    /* flat form, valid after commit */
    std::vector<RtNode> flat_nodes, flat_lights;
    uint32_t flat_root = RT_NONE;
    uint32_t stack_need = 0, scope_depth = 0;
    bool has_media = false, has_tex = false, has_msphere = false;
    bool media_bare_spheres = false; /* every ConstantMedium's boundary is a bare Sphere node (the kernels of RtCfgSphereMedia) */
    uint32_t walk_order = 0; /* RT1W_WALK_* */
    uint32_t walk_annotated = 0; /* BVH nodes that carry an order annotation (0: the plain kernels serve) */
    uint32_t bvh_build = RT1W_BVH_BEST_AXIS;  /* RT1W_BVH_*: what rt1w_scene_commit flattens with (include/rt1w.h says why) */
    std::vector<int32_t> bvh_topology; /* RT1W_BVH_SAH / RT1W_BVH_BEST_AXIS: the rebuilt trees, see rt1w_scene_get_bvh_topology */
    /* every `BVHNode::new` call of the host (rt1w_hittable_bvh, an AABox's sides): root hittable id -> the objects in the order they
     * were handed over (what RT1W_BVH_BEST_AXIS re-runs the reference's build rule on, bvh.rs:84-87) */
    std::map<int, std::vector<int>> bvh_calls;
};

#endif
