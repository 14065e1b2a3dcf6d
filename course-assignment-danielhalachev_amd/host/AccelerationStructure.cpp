#include "AccelerationStructure.h"

#include <algorithm>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>

namespace crt {

BoundingBox::BoundingBox()  // reference: BoundingBox.h:14-20
    : minPoint(std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()),
      maxPoint(std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
               std::numeric_limits<float>::lowest()) {}

void BoundingBox::include(const Vector &p) {  // reference: BoundingBox.h:27-33
  for (unsigned short i = 0; i < 3; i++) {
    minPoint[i] = std::min(minPoint[i], p[i]);
    maxPoint[i] = std::max(maxPoint[i], p[i]);
  }
}

bool BoundingBox::intersects(const BoundingBox &box) const {  // reference: BoundingBox.h:75-83
  for (unsigned short i = 0; i < 3; i++) {
    bool notOverlapI = (minPoint[i] > box.maxPoint[i]) || (maxPoint[i] < box.minPoint[i]);
    if (notOverlapI) return false;
  }
  return true;
}

void BoundingBox::split(unsigned short axis, BoundingBox &first, BoundingBox &second) const {  // BoundingBox.h:60-69
  float middle = (maxPoint[axis] - minPoint[axis]) / 2;
  float splitPlaneCoordinate = minPoint[axis] + middle;
  first = *this;
  second = *this;
  first.maxPoint[axis] = splitPlaneCoordinate;
  second.minPoint[axis] = splitPlaneCoordinate;
}

// reference: KDTree<T>::build, KDTree.cpp:10-46 (triangles) and :89-125 (meshes) -- the two bodies are
// the same algorithm over different element boxes.  Work list instead of recursion; the reference's
// numbering (child[0] and its whole subtree are created before child[1]) is kept by processing
// pending jobs depth-first, first child first.
void KDTree::build(const std::vector<BoundingBox> &elementBoxes, const BoundingBox &rootBox, unsigned short maxDepth,
                   unsigned short maxElementsInLeaf) {
  struct Job {
    unsigned int parent;  // node whose children[slot] this job creates (INVALID for the root)
    int slot;
    unsigned short depth;
    BoundingBox box;
    std::vector<uint32_t> elements;
  };
  nodes.clear();
  std::vector<Job> stack;
  {
    Job root;
    root.parent = INVALID_INDEX;
    root.slot = 0;
    root.depth = 0;
    root.box = rootBox;
    root.elements.resize(elementBoxes.size());
    std::iota(root.elements.begin(), root.elements.end(), 0u);
    stack.push_back(std::move(root));
  }
  while (!stack.empty()) {
    Job job = std::move(stack.back());
    stack.pop_back();
    const unsigned int index = (unsigned int)nodes.size();
    nodes.push_back(TreeNode{job.box, {INVALID_INDEX, INVALID_INDEX}, job.parent, {}});
    if (job.parent != INVALID_INDEX) nodes[job.parent].children[job.slot] = index;
    if (job.depth >= maxDepth || job.elements.size() <= maxElementsInLeaf) {
      nodes[index].indexes = std::move(job.elements);
      continue;
    }
    Job first, second;
    job.box.split(job.depth % 3, first.box, second.box);
    for (uint32_t e : job.elements) {
      if (first.box.intersects(elementBoxes[e])) first.elements.push_back(e);
      if (second.box.intersects(elementBoxes[e])) second.elements.push_back(e);
    }
    first.parent = second.parent = index;
    first.slot = 0;
    second.slot = 1;
    first.depth = second.depth = (unsigned short)(job.depth + 1);
    // LIFO: push the second child first so that the first child (and all of its subtree) is numbered first
    if (!second.elements.empty()) stack.push_back(std::move(second));
    if (!first.elements.empty()) stack.push_back(std::move(first));
  }
}

void KDTree::buildOnDevice(int device, const std::vector<BoundingBox> &elementBoxes, const BoundingBox &rootBox, unsigned short maxDepth,
                           unsigned short maxElementsInLeaf) {
  std::vector<float> flat(elementBoxes.size() * 6);
  for (size_t i = 0; i < elementBoxes.size(); i++)
    for (unsigned short a = 0; a < 3; a++) { flat[6 * i + a] = elementBoxes[i].minPoint[a]; flat[6 * i + 3 + a] = elementBoxes[i].maxPoint[a]; }
  const float root[6] = {rootBox.minPoint.x, rootBox.minPoint.y, rootBox.minPoint.z, rootBox.maxPoint.x, rootBox.maxPoint.y, rootBox.maxPoint.z};
  crt_built_tree *t = nullptr;
  if (crt_build_tree_device(device, flat.data(), (uint32_t)elementBoxes.size(), root, maxDepth, maxElementsInLeaf, &t) != CRT_OK)
    throw std::runtime_error(std::string("crt_build_tree_device failed: ") + crt_build_last_error());
  const uint32_t n = crt_built_tree_node_count(t);
  const float *boxes = crt_built_tree_boxes(t);
  const uint32_t *links = crt_built_tree_links(t), *idx = crt_built_tree_indexes(t);
  nodes.clear();
  nodes.resize(n);
  size_t at = 0;
  for (uint32_t i = 0; i < n; i++) {
    TreeNode &N = nodes[i];
    N.box.minPoint = Vector(boxes[6 * (size_t)i], boxes[6 * (size_t)i + 1], boxes[6 * (size_t)i + 2]);
    N.box.maxPoint = Vector(boxes[6 * (size_t)i + 3], boxes[6 * (size_t)i + 4], boxes[6 * (size_t)i + 5]);
    N.children[0] = links[4 * (size_t)i];
    N.children[1] = links[4 * (size_t)i + 1];
    N.parent = links[4 * (size_t)i + 2];
    N.indexes.assign(idx + at, idx + at + links[4 * (size_t)i + 3]);
    at += links[4 * (size_t)i + 3];
  }
  crt_built_tree_free(t);
}

AccelerationStructure::AccelerationStructure(const Scene &scene, int buildDevice) {
  // reference: AccelerationStructure.cpp:27-50
  const size_t n = scene.objects.size();
  meshTrees.resize(n);
  std::vector<BoundingBox> meshBoxes(n);
  BoundingBox sceneBox;  // BoundingBox(const Scene&), BoundingBox.h:36-48: over the vertices triangles refer to
  for (size_t m = 0; m < n; m++) {
    const Mesh &mesh = scene.objects[m];
    std::vector<BoundingBox> triBoxes(mesh.triangles.size());
    BoundingBox meshBox;  // BoundingBox(const std::vector<Triangle>&), BoundingBox.h:24-34
    for (size_t t = 0; t < mesh.triangles.size(); t++)
      for (int k = 0; k < 3; k++) {
        const Vector &p = mesh.vertices[mesh.triangles[t].indexes[k]].position;
        triBoxes[t].include(p);  // BoundingBox(const Triangle&), BoundingBox.h:50-58
        meshBox.include(p);
        sceneBox.include(p);
      }
    // TriangleKDTree defaults, AccelerationStructure.h:10-11
    if (buildDevice >= 0 && triBoxes.size() >= 4096) meshTrees[m].buildOnDevice(buildDevice, triBoxes, meshBox, 25, 8);
    else meshTrees[m].build(triBoxes, meshBox, 25, 8);
    meshBoxes[m] = meshTrees[m].nodes[0].box;       // ObjectKDTreeSubTree::getBoundingBox, AccelerationStructure.cpp:21-23
  }
  objectTree.build(meshBoxes, sceneBox, 25, 4);  // ObjectKDTree defaults, AccelerationStructure.h:25-26
}

namespace {

// Lays one tree out in the reference's visit order (a node, then the subtree of children[1], then the
// subtree of children[0] -- KDTree.cpp:66-73 pushes children[0] first, so children[1] is popped first)
// and turns the stack walk into hit/miss links.
void threadTree(const KDTree &tree, uint32_t nodeBase, uint32_t elementBase, std::vector<crt_node> &nodes,
                std::vector<uint32_t> &leafEntries) {
  const size_t n = tree.nodes.size();
  std::vector<uint32_t> size(n, 1), pos(n, 0), after(n, CRT_LINK_END);
  for (size_t i = n; i-- > 0;)  // children are always numbered after their parent
    for (int c = 0; c < 2; c++)
      if (tree.nodes[i].children[c] != KDTree::INVALID_INDEX) size[i] += size[tree.nodes[i].children[c]];
  pos[0] = nodeBase;
  for (size_t i = 0; i < n; i++) {
    const unsigned int c0 = tree.nodes[i].children[0], c1 = tree.nodes[i].children[1];
    uint32_t next = pos[i] + 1;
    if (c1 != KDTree::INVALID_INDEX) { pos[c1] = next; next += size[c1]; }
    if (c0 != KDTree::INVALID_INDEX) { pos[c0] = next; after[c0] = after[i]; }
    if (c1 != KDTree::INVALID_INDEX) after[c1] = (c0 != KDTree::INVALID_INDEX) ? pos[c0] : after[i];
  }
  std::vector<uint32_t> byPos(n);
  for (size_t i = 0; i < n; i++) byPos[pos[i] - nodeBase] = (uint32_t)i;
  nodes.resize(std::max(nodes.size(), (size_t)nodeBase + n));
  for (size_t k = 0; k < n; k++) {
    const KDTree::TreeNode &src = tree.nodes[byPos[k]];
    crt_node &dst = nodes[nodeBase + k];
    for (unsigned short a = 0; a < 3; a++) { dst.lo[a] = src.box.minPoint[a]; dst.hi[a] = src.box.maxPoint[a]; }
    dst.miss = after[byPos[k]];
    if (!src.indexes.empty()) {  // leaf iff it holds indexes (KDTree.cpp:56,135)
      dst.link = CRT_LINK_LEAF | (uint32_t)leafEntries.size();
      for (size_t e = 0; e < src.indexes.size(); e++)
        leafEntries.push_back((elementBase + src.indexes[e]) | (e + 1 == src.indexes.size() ? CRT_ENTRY_LAST : 0u));
    } else {
      const unsigned int c0 = src.children[0], c1 = src.children[1];
      dst.link = c1 != KDTree::INVALID_INDEX ? pos[c1] : (c0 != KDTree::INVALID_INDEX ? pos[c0] : dst.miss);
    }
  }
}

}  // namespace

void flattenScene(const Scene &scene, const AccelerationStructure &accel, FlatScene &out) {
  const size_t nMesh = scene.objects.size();
  if (accel.meshTrees.size() != nMesh) throw std::invalid_argument("acceleration structure does not match the scene");
  out.nodes.clear(); out.leafTriangles.clear(); out.leafMeshes.clear(); out.triangles.clear();
  out.triangleVertices.clear(); out.vertexNormals.clear(); out.vertexUVs.clear(); out.meshes.clear();
  out.materials.clear(); out.textures.clear(); out.texels.clear(); out.lights.clear(); out.meshNodeBase.clear();

  // top-level tree first (its leaf entries are mesh indices), then every mesh tree
  threadTree(accel.objectTree, 0, 0, out.nodes, out.leafMeshes);
  uint64_t totalTriangles = 0, totalLeafEntries = 0;
  for (size_t m = 0; m < nMesh; m++) {
    totalTriangles += scene.objects[m].triangles.size();
    for (auto &node : accel.meshTrees[m].nodes) totalLeafEntries += node.indexes.size();
  }
  if (totalTriangles >= 0x7FFFFFFFull || totalLeafEntries >= 0x7FFFFFFFull || out.nodes.size() >= 0x7FFFFFFFull)
    throw std::length_error("scene too large for 31-bit indices");

  uint32_t triBase = 0, vertBase = 0;
  for (size_t m = 0; m < nMesh; m++) {
    const Mesh &mesh = scene.objects[m];
    if (mesh.material >= scene.materials.size()) throw std::out_of_range("mesh material index");
    const uint32_t nodeBase = (uint32_t)out.nodes.size();
    out.meshNodeBase.push_back(nodeBase);
    threadTree(accel.meshTrees[m], nodeBase, triBase, out.nodes, out.leafTriangles);
    crt_mesh cm{};
    cm.root = nodeBase;
    cm.material = mesh.material;
    cm.flags = scene.materials[mesh.material].type == Refractive ? 1u : 0u;
    out.meshes.push_back(cm);
    for (const Triangle &t : mesh.triangles) {
      const Vector &a = mesh.vertices[t.indexes[0]].position, &b = mesh.vertices[t.indexes[1]].position,
                   &c = mesh.vertices[t.indexes[2]].position;
      crt_triangle ct{};
      ct.v0[0] = a.x; ct.v0[1] = a.y; ct.v0[2] = a.z; ct.nx = t.normal.x;
      ct.v1[0] = b.x; ct.v1[1] = b.y; ct.v1[2] = b.z; ct.ny = t.normal.y;
      ct.v2[0] = c.x; ct.v2[1] = c.y; ct.v2[2] = c.z; ct.nz = t.normal.z;
      ct.plane = -(a.dot(t.normal));  // distanceToPlane, Ray.cpp:17
      out.triangles.push_back(ct);
      for (int k = 0; k < 3; k++) out.triangleVertices.push_back(vertBase + t.indexes[k]);
    }
    for (const Vertex &v : mesh.vertices) {
      out.vertexNormals.insert(out.vertexNormals.end(), {v.normal.x, v.normal.y, v.normal.z});
      out.vertexUVs.insert(out.vertexUVs.end(), {v.UV.x, v.UV.y, v.UV.z});
    }
    triBase += (uint32_t)mesh.triangles.size();
    vertBase += (uint32_t)mesh.vertices.size();
  }

  for (const Material &m : scene.materials) {
    crt_material cm{};
    cm.albedo[0] = m.albedo.x; cm.albedo[1] = m.albedo.y; cm.albedo[2] = m.albedo.z;
    cm.ior = m.ior;
    cm.type = (uint32_t)m.type;
    cm.smooth = m.smoothShading ? 1u : 0u;
    cm.texture = m.texture;
    if (m.texture >= (int)scene.textures.size()) throw std::out_of_range("material texture index");
    out.materials.push_back(cm);
  }
  for (const Texture &t : scene.textures) {
    crt_texture ct{};
    ct.kind = (uint32_t)t.kind;
    ct.color_a[0] = t.colorA.x; ct.color_a[1] = t.colorA.y; ct.color_a[2] = t.colorA.z;
    ct.color_b[0] = t.colorB.x; ct.color_b[1] = t.colorB.y; ct.color_b[2] = t.colorB.z;
    ct.scalar = t.scalar;
    if (t.kind == BitmapTexture) {
      if (t.rgb8.size() != (size_t)t.width * t.height * 3) throw std::invalid_argument("bitmap size mismatch");
      ct.width = (uint32_t)t.width;
      ct.height = (uint32_t)t.height;
      ct.texel_offset = out.texels.size() / 3;
      out.texels.insert(out.texels.end(), t.rgb8.begin(), t.rgb8.end());
    }
    out.textures.push_back(ct);
  }
  for (const Light &l : scene.lights) {
    crt_light cl{};
    cl.position[0] = l.position.x; cl.position[1] = l.position.y; cl.position[2] = l.position.z;
    cl.intensity = l.intentsity;
    out.lights.push_back(cl);
  }

  crt_scene_desc &d = out.desc;
  d = crt_scene_desc{};
  d.width = scene.sceneSettings.image.width;
  d.height = scene.sceneSettings.image.height;
  d.background[0] = scene.sceneSettings.sceneBackgroundColor.x;
  d.background[1] = scene.sceneSettings.sceneBackgroundColor.y;
  d.background[2] = scene.sceneSettings.sceneBackgroundColor.z;
  d.nodes = out.nodes.data(); d.n_nodes = (uint32_t)out.nodes.size();
  d.top_root = 0;
  d.leaf_triangles = out.leafTriangles.data(); d.n_leaf_triangles = out.leafTriangles.size();
  d.leaf_meshes = out.leafMeshes.data(); d.n_leaf_meshes = (uint32_t)out.leafMeshes.size();
  d.triangles = out.triangles.data(); d.n_triangles = (uint32_t)out.triangles.size();
  d.triangle_vertices = out.triangleVertices.data();
  d.vertex_normals = out.vertexNormals.data();
  d.vertex_uvs = out.vertexUVs.data();
  d.n_vertices = (uint32_t)(out.vertexNormals.size() / 3);
  d.meshes = out.meshes.data(); d.n_meshes = (uint32_t)out.meshes.size();
  d.materials = out.materials.data(); d.n_materials = (uint32_t)out.materials.size();
  d.textures = out.textures.data(); d.n_textures = (uint32_t)out.textures.size();
  d.texels = out.texels.data(); d.n_texels = out.texels.size() / 3;
  d.lights = out.lights.data(); d.n_lights = (uint32_t)out.lights.size();
}

}  // namespace crt
