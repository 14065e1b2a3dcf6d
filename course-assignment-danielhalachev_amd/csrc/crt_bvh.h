// crt_bvh.h -- the CANDIDATE FILTER: a 4-wide bounding-volume hierarchy over the scene's triangles, built by crt_create.
//
// The reference's trees are the contract: which triangles a ray TESTS is decided by the leaves of the reference's two kd-trees whose
// boxes pass the reference's exact slab test (KDTree.cpp:48-87,127-167; BoundingBox.h:85-108), and a hit only counts when one of
// those leaves lists the triangle ("a better BVH would be wrong").  But which triangles a ray can be
// ACCEPTED by (Ray.cpp:9-31, Triangle.cpp:37-57) is geometry: an accepted hit with a finite distance puts the computed hit point
// inside the triangle grown by the test's own tolerance (FLT_EPSILON over the edge length, plus rounding), on the ray up to
// rounding.  So the production kernels (kernel_bvh.h)
//   1. find the CANDIDATES with a hierarchy of their own -- boxes around every triangle grown by that tolerance (triangle_margin
//      below), walked nearest-first with distance pruning, a test that can only err towards "pass" (bvh_child_test);
//   2. run the reference's exact triangle test on the candidates;
//   3. VERIFY an accepted candidate against the reference's trees: its mesh must be listed in a top-level leaf whose box the ray
//      passes, and the triangle in a leaf of the mesh's tree whose box the ray passes -- the exact slab test on those few boxes
//      (a leaf is reached exactly when its own box passes: nested boxes + monotone slab test, kernel_heavy.h);
//   4. take the reference's winner: smallest finite distance, ties to the hit collected first (KDTree.cpp:75-86,156-167) = smallest
//      (position of the mesh's first reached entry in leaf_meshes, position of the triangle's first reached entry in leaf_triangles).
// What the filter cannot see -- accepted hits with an infinite or NaN distance (a ray parallel to a plane; they matter only when
// a ray has no finite hit) -- is left to the reference-order kernels: a closest-hit ray without a verified finite hit is walked again
// by them.  A scene the margins cannot be bounded for (degenerate triangles, normals that are not the triangles') has no filter
// at all (SceneArgs::bvh_ok = 0) and renders on the reference-order kernels alone.
#pragma once

#include <stdint.h>
#include <vector>

#include "../../include/crt_hip.h"

// one node: four children, boxes in SoA form, 128 bytes
struct BvhNode {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    uint32_t child[4];  // inner node: its index; leaf: BVH_LEAF | (triangles - 1) << 24 | first entry of bvh_tris; BVH_EMPTY: no child (its box is inverted)
    uint32_t pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode is one 128-byte line");
constexpr uint32_t BVH_LEAF = 0x80000000u, BVH_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t BVH_LEAF_MAX = 2;          // triangles per leaf: one step of a walk (kernel_bvh.h: two triangles per step).  Measured 1 / 2 / 4: HW14 2.75 / 2.65 / 2.76 ms, HW11 3.38 / 3.29 / 3.48
constexpr uint32_t BVH_ID_REFRACTIVE = 0x80000000u;  // bvh_ids: the triangle's mesh is refractive (shadow rays skip it outside the GI mode)
// A triangle listed by more reference leaves than this has no leaf list (where more than eight triangles share a vertex -- the pole of a
// sphere -- the reference's tree subdivides down to its depth limit and lists each of them in thousands of leaves): its hits are
// verified by walking the mesh's tree itself, pruned to the nodes whose box overlaps the triangle's (kernel_bvh.h: bvh_leaf_walk;
// bvh_build checks that every leaf listing such a triangle does overlap it, or there is no filter).
constexpr uint32_t BVH_LIST_MAX = 24;
constexpr uint32_t BVH_TRI_WALK = 0x80000000u;       // tri_mesh: verify by the pruned tree walk, not by the leaf list

struct BvhHost {
    bool trees_are_ranges = false;   // every tree of the description occupies one index range starting at its root (kernel_heavy.h's leaf sequences assume it)
    bool ok = false;                 // false: no filter for this scene (`why` says why)
    const char *why = "";
    std::vector<BvhNode> nodes;      // nodes[0] = root
    std::vector<float> tris;         // 12 floats per entry: {v0, nx} {v1, ny} {v2, nz}, in leaf order
    std::vector<uint32_t> ids;       // per entry: global triangle index | BVH_ID_REFRACTIVE
    std::vector<uint32_t> tri_mesh;  // per global triangle: its mesh (NONE: in no leaf) | BVH_TRI_WALK
    float overlap_eps = 0;           // slack of the box-overlap predicate of the pruned tree walk
    // CSR: the reference leaves listing triangle t, in visit order, 8 floats each: {box lo, entry position in leaf_triangles (bits)} {box hi, 0}
    std::vector<uint32_t> tri_leaf_first;
    std::vector<float> tri_leaf_list;
    // CSR: the top-level leaves listing mesh m, in visit order, the same form with the entry position in leaf_meshes
    std::vector<uint32_t> mesh_top_first;
    std::vector<float> mesh_top_list;
    // The same hierarchy as the MISS CHECK reads it (kernel_bvh.h: bvh_miss_step): per node, per child,
    //   vnodes  the box of everything the reference could reach the child's triangles THROUGH: the union of the boxes of the
    //           reference leaves listing them (and of the triangles themselves) -- 24 floats lo.x[4] .. hi.z[4];
    //   cones   a cone around the child's triangle normals -- axis x[4] y[4] z[4] and k[4] = sin(half-angle + 1e-4): a direction d
    //           with |d . axis| > k is more than 1e-4 rad away from being parallel to any of those triangles' planes.
    std::vector<float> vnodes, cones;
    float extent = 0;                // largest absolute coordinate of any box
    uint32_t max_depth = 0;          // of the binary build
    uint32_t wide_depth = 0;         // inner nodes on the longest root-to-leaf path of the 4-wide hierarchy
    double max_margin = 0;
    uint32_t walk_triangles = 0;     // triangles verified by the pruned tree walk
};

// Builds the filter for a validated scene description (crt_scene.hip: validate_scene has checked every index).
void bvh_build(const crt_scene_desc *s, bool nested_boxes, BvhHost &out);
