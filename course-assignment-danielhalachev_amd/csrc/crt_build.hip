// crt_build.hip -- the reference's spatial-median tree built on the MI355X (crt_hip.h: crt_build_tree_device).
//
// KDTree<T>::build (reference: SourceCode/src/KDTree.cpp:10-46 for triangles, :89-125 for meshes) recurses: a node of more
// than maxLeaf elements above depth maxDepth is split at min + (max - min) / 2 on axis depth % 3 (BoundingBox.h:60-69),
// an element goes into a child when its box overlaps the child's box, bounds included (BoundingBox.h:71-83), a child
// that gets no element is not created, and child 0 with its whole subtree is numbered before child 1.
//
// On the GPU the same tree is built LEVEL BY LEVEL -- all nodes of a depth at once, one thread per (node, element) entry:
//   classify   per node: leaf, or inner with its two child boxes (the same float operations as the reference's split);
//   flags      per entry: does the element's box overlap child 0 / child 1;
//   scan       exclusive sums of the two flag arrays over the level's entries (hipCUB);
//   scatter    per entry: its place in the next level's entry array -- entries stay in their list order, node by node,
//              child 0's list before child 1's -- and per node: its children's records.
// The level arrays come back to the host, which numbers the nodes in the reference's pre-order (subtree sizes bottom-up,
// indices top-down) and fills the same structure the host builder fills.  Every float in it is produced by the same
// operation on the same operands as in the reference: the trees are identical, node for node and list for list, which
// tests/test_gpu_build.py checks through crt_host_tree_dump.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/crt_hip.h"

namespace {

struct BNode {           // one node of the level under construction
    float lo[3], hi[3];
    uint32_t parent;     // index of the parent in the PREVIOUS level's array (0xFFFFFFFF for the root)
    uint32_t slot;       // 0 / 1: which child of its parent
    uint32_t begin, end; // its entries in the level's entry array
    uint32_t leaf;       // set by classify
    float c0hi, c1lo;    // inner: the split plane as child 0's max / child 1's min on the split axis (the same value)
};

__global__ void classify_kernel(BNode *nodes, uint32_t n, uint32_t depth, uint32_t max_depth, uint32_t max_leaf) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BNode &N = nodes[i];
    const uint32_t count = N.end - N.begin;
    N.leaf = (depth >= max_depth || count <= max_leaf) ? 1u : 0u;  // KDTree.cpp:13-16
    const uint32_t axis = depth % 3u;
    const float middle = (N.hi[axis] - N.lo[axis]) / 2;            // BoundingBox.h:61-62
    const float plane = N.lo[axis] + middle;
    N.c0hi = plane;
    N.c1lo = plane;
}

// owner[k] = the node entry k belongs to: nodes' ranges are consecutive, so a binary search over `begin`
__device__ __forceinline__ uint32_t owner_of(const BNode *nodes, uint32_t n, uint32_t k) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (nodes[mid].begin <= k) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool overlaps(const float *alo, const float *ahi, const float *b) {  // BoundingBox.h:75-83, bounds included
    for (int a = 0; a < 3; a++)
        if (alo[a] > b[3 + a] || ahi[a] < b[a]) return false;
    return true;
}

__global__ void flags_kernel(const BNode *nodes, uint32_t n_nodes, const uint32_t *entries, uint32_t n_entries, const float *boxes,
                             uint32_t depth, uint32_t *owner, uint32_t *f0, uint32_t *f1) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    const uint32_t i = owner_of(nodes, n_nodes, k);
    owner[k] = i;
    const BNode &N = nodes[i];
    uint32_t a0 = 0, a1 = 0;
    if (!N.leaf) {
        const uint32_t axis = depth % 3u;
        float lo0[3] = {N.lo[0], N.lo[1], N.lo[2]}, hi0[3] = {N.hi[0], N.hi[1], N.hi[2]};
        float lo1[3] = {N.lo[0], N.lo[1], N.lo[2]}, hi1[3] = {N.hi[0], N.hi[1], N.hi[2]};
        hi0[axis] = N.c0hi;
        lo1[axis] = N.c1lo;
        const float *b = boxes + 6 * (size_t)entries[k];
        a0 = overlaps(lo0, hi0, b) ? 1u : 0u;   // KDTree.cpp:24-31: `firstBox.intersects(elementBox)`
        a1 = overlaps(lo1, hi1, b) ? 1u : 0u;
    }
    f0[k] = a0;
    f1[k] = a1;
}

// child slots per node: has child 0 / has child 1 (a child without elements is not created, KDTree.cpp:33-44)
__global__ void child_count_kernel(const BNode *nodes, uint32_t n, const uint32_t *s0, const uint32_t *s1, uint32_t n_entries,
                                   uint32_t total0, uint32_t total1, uint32_t *kids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BNode &N = nodes[i];
    const uint32_t e0 = N.end < n_entries ? s0[N.end] : total0, e1 = N.end < n_entries ? s1[N.end] : total1;
    const uint32_t b0 = N.begin < n_entries ? s0[N.begin] : total0, b1 = N.begin < n_entries ? s1[N.begin] : total1;
    kids[i] = (N.leaf ? 0u : ((e0 - b0) ? 1u : 0u) + ((e1 - b1) ? 1u : 0u));
}

__global__ void scatter_kernel(const BNode *nodes, const uint32_t *entries, uint32_t n_entries, const uint32_t *owner, const uint32_t *f0,
                               const uint32_t *f1, const uint32_t *s0, const uint32_t *s1, uint32_t total0, uint32_t total1,
                               uint32_t *next_entries) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    const BNode &N = nodes[owner[k]];
    if (N.leaf) return;
    // everything the nodes before this one pass on lies before this node's lists: S0[begin] + S1[begin]
    const uint32_t b0 = s0[N.begin], b1 = s1[N.begin];
    const uint32_t e0 = N.end < n_entries ? s0[N.end] : total0;
    const uint32_t off0 = b0 + b1, off1 = off0 + (e0 - b0);
    if (f0[k]) next_entries[off0 + (s0[k] - b0)] = entries[k];
    if (f1[k]) next_entries[off1 + (s1[k] - b1)] = entries[k];
}

__global__ void children_kernel(const BNode *nodes, uint32_t n, uint32_t depth, const uint32_t *s0, const uint32_t *s1, uint32_t n_entries,
                                uint32_t total0, uint32_t total1, const uint32_t *kid_offset, BNode *next) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BNode &N = nodes[i];
    if (N.leaf) return;
    const uint32_t axis = depth % 3u;
    const uint32_t b0 = s0[N.begin], b1 = s1[N.begin];
    const uint32_t e0 = N.end < n_entries ? s0[N.end] : total0, e1 = N.end < n_entries ? s1[N.end] : total1;
    const uint32_t c0 = e0 - b0, c1 = e1 - b1;
    const uint32_t off0 = b0 + b1, off1 = off0 + c0;
    uint32_t at = kid_offset[i];
    if (c0) {
        BNode C = N;
        C.hi[axis] = N.c0hi;
        C.parent = i; C.slot = 0; C.begin = off0; C.end = off0 + c0; C.leaf = 0;
        next[at++] = C;
    }
    if (c1) {
        BNode C = N;
        C.lo[axis] = N.c1lo;
        C.parent = i; C.slot = 1; C.begin = off1; C.end = off1 + c1; C.leaf = 0;
        next[at] = C;
    }
}

struct DeviceBuffers {
    std::vector<void *> all;
    ~DeviceBuffers() { for (void *p : all) (void)hipFree(p); }
    template <typename T> T *alloc(size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
        all.push_back(p);
        return (T *)p;
    }
};

std::string g_build_error;

}  // namespace

struct crt_built_tree {
    std::vector<float> boxes;        // 6 per node (min xyz, max xyz), nodes in the reference's creation order
    std::vector<uint32_t> links;     // 4 per node: children[0], children[1], parent, number of leaf indexes
    std::vector<uint32_t> indexes;   // the leaves' index lists, concatenated in node order
};

extern "C" const char *crt_build_last_error(void) { return g_build_error.c_str(); }

extern "C" void crt_built_tree_free(crt_built_tree *t) { delete t; }

extern "C" uint32_t crt_built_tree_node_count(const crt_built_tree *t) { return t ? (uint32_t)(t->links.size() / 4) : 0u; }
extern "C" uint64_t crt_built_tree_index_total(const crt_built_tree *t) { return t ? (uint64_t)t->indexes.size() : 0u; }
extern "C" const float *crt_built_tree_boxes(const crt_built_tree *t) { return t ? t->boxes.data() : nullptr; }
extern "C" const uint32_t *crt_built_tree_links(const crt_built_tree *t) { return t ? t->links.data() : nullptr; }
extern "C" const uint32_t *crt_built_tree_indexes(const crt_built_tree *t) { return t ? t->indexes.data() : nullptr; }

extern "C" int crt_build_tree_device(int device, const float *element_boxes, uint32_t n_elements, const float root_box[6],
                                     uint32_t max_depth, uint32_t max_leaf, crt_built_tree **out) {
    if (!out || !root_box || (n_elements && !element_boxes)) return CRT_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_build_error = "no usable HIP device"; return CRT_ERR_NO_DEVICE; }
#define BK(expr)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) { g_build_error = std::string(#expr) + ": " + hipGetErrorString(e_); return CRT_ERR_HIP; } \
    } while (0)
    BK(hipSetDevice(device));
    DeviceBuffers D;
    float *d_boxes = D.alloc<float>((size_t)n_elements * 6);
    if (!d_boxes) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; }
    if (n_elements) BK(hipMemcpy(d_boxes, element_boxes, (size_t)n_elements * 6 * sizeof(float), hipMemcpyHostToDevice));

    // level 0: the root holds every element, in order (KDTree.cpp:11: iota)
    std::vector<BNode> h_nodes(1);
    memset(&h_nodes[0], 0, sizeof(BNode));
    for (int a = 0; a < 3; a++) { h_nodes[0].lo[a] = root_box[a]; h_nodes[0].hi[a] = root_box[3 + a]; }
    h_nodes[0].parent = 0xFFFFFFFFu;
    h_nodes[0].begin = 0; h_nodes[0].end = n_elements;
    std::vector<uint32_t> h_entries(n_elements);
    for (uint32_t i = 0; i < n_elements; i++) h_entries[i] = i;

    struct Level { std::vector<BNode> nodes; std::vector<uint32_t> entries; };
    std::vector<Level> levels;
    BNode *d_nodes = D.alloc<BNode>(1);
    uint32_t *d_entries = D.alloc<uint32_t>(n_elements);
    if (!d_nodes || !d_entries) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; }
    BK(hipMemcpy(d_nodes, h_nodes.data(), sizeof(BNode), hipMemcpyHostToDevice));
    if (n_elements) BK(hipMemcpy(d_entries, h_entries.data(), (size_t)n_elements * 4, hipMemcpyHostToDevice));
    uint32_t n_nodes = 1, n_entries = n_elements;
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    for (uint32_t depth = 0;; depth++) {
        const uint32_t nb = (n_nodes + 255) / 256, eb = (n_entries + 255) / 256;
        hipLaunchKernelGGL(classify_kernel, dim3(nb), dim3(256), 0, 0, d_nodes, n_nodes, depth, max_depth, max_leaf);
        uint32_t *d_owner = D.alloc<uint32_t>(n_entries), *d_f0 = D.alloc<uint32_t>(n_entries), *d_f1 = D.alloc<uint32_t>(n_entries);
        uint32_t *d_s0 = D.alloc<uint32_t>(n_entries), *d_s1 = D.alloc<uint32_t>(n_entries), *d_kids = D.alloc<uint32_t>(n_nodes),
                 *d_koff = D.alloc<uint32_t>(n_nodes);
        if (!d_owner || !d_f0 || !d_f1 || !d_s0 || !d_s1 || !d_kids || !d_koff) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; }
        uint32_t total0 = 0, total1 = 0;
        if (n_entries) {
            hipLaunchKernelGGL(flags_kernel, dim3(eb), dim3(256), 0, 0, d_nodes, n_nodes, d_entries, n_entries, d_boxes, depth, d_owner, d_f0, d_f1);
            size_t need = 0;
            BK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_f0, d_s0, (int)n_entries));
            if (need > temp_bytes) { d_temp = D.alloc<char>(need); temp_bytes = need; if (!d_temp) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; } }
            size_t tb = temp_bytes;
            BK(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_f0, d_s0, (int)n_entries));
            tb = temp_bytes;
            BK(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_f1, d_s1, (int)n_entries));
            uint32_t last[4];
            BK(hipMemcpy(&last[0], d_s0 + (n_entries - 1), 4, hipMemcpyDeviceToHost));
            BK(hipMemcpy(&last[1], d_f0 + (n_entries - 1), 4, hipMemcpyDeviceToHost));
            BK(hipMemcpy(&last[2], d_s1 + (n_entries - 1), 4, hipMemcpyDeviceToHost));
            BK(hipMemcpy(&last[3], d_f1 + (n_entries - 1), 4, hipMemcpyDeviceToHost));
            total0 = last[0] + last[1];
            total1 = last[2] + last[3];
        }
        hipLaunchKernelGGL(child_count_kernel, dim3(nb), dim3(256), 0, 0, d_nodes, n_nodes, d_s0, d_s1, n_entries, total0, total1, d_kids);
        {
            size_t need = 0;
            BK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_kids, d_koff, (int)n_nodes));
            if (need > temp_bytes) { d_temp = D.alloc<char>(need); temp_bytes = need; if (!d_temp) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; } }
            size_t tb = temp_bytes;
            BK(hipcub::DeviceScan::ExclusiveSum(d_temp, tb, d_kids, d_koff, (int)n_nodes));
        }
        uint32_t lastk[2];
        BK(hipMemcpy(&lastk[0], d_koff + (n_nodes - 1), 4, hipMemcpyDeviceToHost));
        BK(hipMemcpy(&lastk[1], d_kids + (n_nodes - 1), 4, hipMemcpyDeviceToHost));
        const uint32_t next_nodes = lastk[0] + lastk[1];
        const uint64_t next_entries64 = (uint64_t)total0 + total1;
        if (next_entries64 > 0x7FFFFFFFull) { g_build_error = "tree too large"; return CRT_ERR_INVALID; }
        const uint32_t next_entries = (uint32_t)next_entries64;
        BNode *d_next = D.alloc<BNode>(next_nodes);
        uint32_t *d_next_entries = D.alloc<uint32_t>(next_entries);
        if (!d_next || !d_next_entries) { g_build_error = "out of device memory"; return CRT_ERR_NOMEM; }
        if (n_entries && next_entries)
            hipLaunchKernelGGL(scatter_kernel, dim3(eb), dim3(256), 0, 0, d_nodes, d_entries, n_entries, d_owner, d_f0, d_f1, d_s0, d_s1, total0, total1,
                               d_next_entries);
        if (next_nodes)
            hipLaunchKernelGGL(children_kernel, dim3(nb), dim3(256), 0, 0, d_nodes, n_nodes, depth, d_s0, d_s1, n_entries, total0, total1, d_koff, d_next);
        BK(hipGetLastError());
        // this level back to the host (its classify results are in place now)
        levels.emplace_back();
        levels.back().nodes.resize(n_nodes);
        levels.back().entries.resize(n_entries);
        BK(hipMemcpy(levels.back().nodes.data(), d_nodes, (size_t)n_nodes * sizeof(BNode), hipMemcpyDeviceToHost));
        if (n_entries) BK(hipMemcpy(levels.back().entries.data(), d_entries, (size_t)n_entries * 4, hipMemcpyDeviceToHost));
        if (next_nodes == 0) break;
        d_nodes = d_next; d_entries = d_next_entries; n_nodes = next_nodes; n_entries = next_entries;
        if (depth > max_depth + 2u) { g_build_error = "tree deeper than its depth limit"; return CRT_ERR_INVALID; }
    }
#undef BK

    // ---- the reference's numbering: a node, then child 0's whole subtree, then child 1's (KDTree.cpp:33-44)
    const size_t L = levels.size();
    std::vector<std::vector<uint32_t>> child(L), size(L), index(L);
    size_t total_nodes = 0;
    for (size_t l = 0; l < L; l++) {
        child[l].assign(levels[l].nodes.size() * 2, 0xFFFFFFFFu);
        size[l].assign(levels[l].nodes.size(), 1u);
        index[l].assign(levels[l].nodes.size(), 0u);
        total_nodes += levels[l].nodes.size();
    }
    for (size_t l = L; l-- > 1;)
        for (size_t i = 0; i < levels[l].nodes.size(); i++) {
            const BNode &N = levels[l].nodes[i];
            child[l - 1][2 * (size_t)N.parent + N.slot] = (uint32_t)i;
            size[l - 1][N.parent] += size[l][i];
        }
    for (size_t l = 0; l + 1 < L; l++)
        for (size_t i = 0; i < levels[l].nodes.size(); i++) {
            const uint32_t c0 = child[l][2 * i], c1 = child[l][2 * i + 1];
            uint32_t next = index[l][i] + 1;
            if (c0 != 0xFFFFFFFFu) { index[l + 1][c0] = next; next += size[l + 1][c0]; }
            if (c1 != 0xFFFFFFFFu) index[l + 1][c1] = next;
        }
    crt_built_tree *T = new crt_built_tree();
    T->boxes.resize(total_nodes * 6);
    T->links.assign(total_nodes * 4, 0xFFFFFFFFu);
    std::vector<uint32_t> leaf_level(total_nodes, 0), leaf_at(total_nodes, 0);
    for (size_t l = 0; l < L; l++)
        for (size_t i = 0; i < levels[l].nodes.size(); i++) {
            const BNode &N = levels[l].nodes[i];
            const uint32_t id = index[l][i];
            for (int a = 0; a < 3; a++) { T->boxes[6 * (size_t)id + a] = N.lo[a]; T->boxes[6 * (size_t)id + 3 + a] = N.hi[a]; }
            const uint32_t c0 = child[l][2 * i], c1 = child[l][2 * i + 1];
            T->links[4 * (size_t)id + 0] = c0 != 0xFFFFFFFFu ? index[l + 1][c0] : 0xFFFFFFFFu;
            T->links[4 * (size_t)id + 1] = c1 != 0xFFFFFFFFu ? index[l + 1][c1] : 0xFFFFFFFFu;
            T->links[4 * (size_t)id + 2] = l ? index[l - 1][N.parent] : 0xFFFFFFFFu;
            T->links[4 * (size_t)id + 3] = N.leaf ? N.end - N.begin : 0u;
            leaf_level[id] = (uint32_t)l;
            leaf_at[id] = (uint32_t)i;
        }
    for (size_t id = 0; id < total_nodes; id++) {
        const BNode &N = levels[leaf_level[id]].nodes[leaf_at[id]];
        if (!N.leaf) continue;
        const std::vector<uint32_t> &e = levels[leaf_level[id]].entries;
        T->indexes.insert(T->indexes.end(), e.begin() + N.begin, e.begin() + N.end);
    }
    *out = T;
    return CRT_OK;
}
