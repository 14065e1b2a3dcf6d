// Host-side builder of the two-level spatial-median tree and its flattened device form.
//
// The tree is part of the parity contract: the reference's tree mode is NOT an
// exact closest-hit search, so the builder reproduces the reference's topology and box floats
// exactly (reference: SourceCode/src/KDTree.cpp:10-46,89-125; include/tracer/BoundingBox.h:24-83;
// src/AccelerationStructure.cpp:12-50) and then flattens it into the stackless hit/miss-link form
// described at include/crt_hip.h (crt_node).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/crt_hip.h"
#include "Scene.h"

namespace crt {

struct BoundingBox {  // reference: BoundingBox.h:9-22
  Vector minPoint, maxPoint;
  BoundingBox();  // empty: min = +FLT_MAX, max = lowest
  void include(const Vector &p);
  bool intersects(const BoundingBox &box) const;  // BoundingBox.h:75-83 (inclusive)
  void split(unsigned short axis, BoundingBox &first, BoundingBox &second) const;  // BoundingBox.h:60-69
};

// The tree exactly as the reference builds it: nodes in creation (pre-order) numbering.
struct KDTree {
  struct TreeNode {  // reference: KDTree.h:16-21
    BoundingBox box;
    unsigned int children[2];
    unsigned int parent;
    std::vector<uint32_t> indexes;
  };
  static constexpr unsigned int INVALID_INDEX = 0xFFFFFFFFu;
  std::vector<TreeNode> nodes;
  // elementBoxes[i] = box of element i; rootBox = box of the root node
  void build(const std::vector<BoundingBox> &elementBoxes, const BoundingBox &rootBox, unsigned short maxDepth,
             unsigned short maxElementsInLeaf);
  // The same tree, built on GPU `device` (crt_hip.h: crt_build_tree_device) -- identical node for node.
  void buildOnDevice(int device, const std::vector<BoundingBox> &elementBoxes, const BoundingBox &rootBox, unsigned short maxDepth,
                     unsigned short maxElementsInLeaf);
};

// reference: AccelerationStructure.h:6-33 -- TriangleKDTree per mesh (depth 25, leaf 8) under an
// ObjectKDTree over the meshes (depth 25, leaf 4).
struct AccelerationStructure {
  std::vector<KDTree> meshTrees;
  KDTree objectTree;
  // buildDevice >= 0: the mesh trees are built on that GPU (meshes of fewer than 4096 triangles, and the tiny tree over the
  // meshes, stay on the host: a launch sequence per level costs more than building them here)
  explicit AccelerationStructure(const Scene &scene, int buildDevice = -1);
};

// Flat, device-ready copy of a scene and its tree; `desc` points into the vectors below.
struct FlatScene {
  std::vector<crt_node> nodes;
  std::vector<uint32_t> leafTriangles, leafMeshes;
  std::vector<crt_triangle> triangles;
  std::vector<uint32_t> triangleVertices;
  std::vector<float> vertexNormals, vertexUVs;
  std::vector<crt_mesh> meshes;
  std::vector<crt_material> materials;
  std::vector<crt_texture> textures;
  std::vector<uint8_t> texels;
  std::vector<crt_light> lights;
  std::vector<uint32_t> meshNodeBase;  // first global node index of each mesh tree (top tree starts at 0)
  crt_scene_desc desc{};
  FlatScene() = default;
  FlatScene(const FlatScene &) = delete;
  FlatScene &operator=(const FlatScene &) = delete;
};

void flattenScene(const Scene &scene, const AccelerationStructure &accel, FlatScene &out);

}  // namespace crt
