// crt_scene.hip -- crt_create / crt_destroy: the flattened scene (crt_scene_desc) validated, copied to HBM and turned into what the
// kernels read: leaf-order triangle records, compact links, leaf sequences, the plan of the top-level tree, the
// candidate filter (crt_bvh.h).  replaces RayTracer::RayTracer (RayTracer.cpp:45-51), setCamera (:57-59).
#include "crt_internal.h"

std::string g_create_error;

extern "C" int crt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int validate_scene(const crt_scene_desc *s, std::string &err) {
    auto bad = [&](const char *m) { err = m; return CRT_ERR_INVALID; };
    if (!s) return bad("scene is NULL");
    if (s->width == 0 || s->height == 0) return bad("empty image");
    if (!s->nodes || s->n_nodes == 0 || s->top_root >= s->n_nodes) return bad("missing tree nodes");
    if (s->n_triangles && (!s->triangles || !s->triangle_vertices)) return bad("missing triangle arrays");
    if (s->n_vertices && !s->vertex_normals) return bad("missing vertex normals");
    if (s->n_meshes && !s->meshes) return bad("missing meshes");
    if (s->n_materials == 0 && s->n_meshes) return bad("meshes without materials");
    if (s->n_leaf_triangles && !s->leaf_triangles) return bad("missing leaf_triangles");
    if (s->n_leaf_meshes && !s->leaf_meshes) return bad("missing leaf_meshes");
    if (s->n_materials && !s->materials) return bad("missing materials");
    if (s->n_textures && !s->textures) return bad("missing textures");
    if (s->n_texels && !s->texels) return bad("missing texels");
    if (s->n_lights && !s->lights) return bad("missing lights");
    if (s->n_leaf_triangles > 0x7FFFFFFFull) return bad("too many leaf entries");
    // every index the kernel will follow is checked here, on the host, before anything is launched
    // Links must point FORWARD (nodes stored in visit order): this makes every walk finite whatever the
    // data, and it is what lets the packet kernel park a ray until `node index >= miss`.
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const crt_node &n = s->nodes[i];
        if (n.miss != CRT_LINK_END && (n.miss >= s->n_nodes || n.miss <= i)) return bad("node miss link must point forward");
        if (is_leaf_link(n.link)) continue;
        if (n.link != CRT_LINK_END && (n.link >= s->n_nodes || n.link <= i)) return bad("node hit link must point forward");
        if (n.link != CRT_LINK_END && n.miss != CRT_LINK_END && n.link > n.miss) return bad("node hit link beyond its miss link");
    }
    for (uint64_t i = 0; i < s->n_leaf_triangles; i++)
        if ((s->leaf_triangles[i] & ~CRT_ENTRY_LAST) >= s->n_triangles) return bad("leaf triangle index out of range");
    if (s->n_leaf_triangles && !(s->leaf_triangles[s->n_leaf_triangles - 1] & CRT_ENTRY_LAST)) return bad("unterminated triangle leaf");
    for (uint32_t i = 0; i < s->n_leaf_meshes; i++)
        if ((s->leaf_meshes[i] & ~CRT_ENTRY_LAST) >= s->n_meshes) return bad("leaf mesh index out of range");
    if (s->n_leaf_meshes && !(s->leaf_meshes[s->n_leaf_meshes - 1] & CRT_ENTRY_LAST)) return bad("unterminated mesh leaf");
    for (uint32_t i = 0; i < s->n_meshes; i++) {
        if (s->meshes[i].root >= s->n_nodes) return bad("mesh root out of range");
        if (s->meshes[i].material >= s->n_materials) return bad("mesh material out of range");
    }
    for (uint64_t i = 0; i < (uint64_t)s->n_triangles * 3; i++)
        if (s->triangle_vertices[i] >= s->n_vertices) return bad("triangle vertex index out of range");
    bool any_uv_texture = false;
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const crt_material &m = s->materials[i];
        if (m.texture >= 0) {
            if ((uint32_t)m.texture >= s->n_textures) return bad("material texture out of range");
            uint32_t k = s->textures[m.texture].kind;
            if (k == CRT_TEX_CHECKER || k == CRT_TEX_BITMAP) any_uv_texture = true;
        }
    }
    if (any_uv_texture && !s->vertex_uvs) return bad("textured material without vertex uvs");
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const crt_texture &t = s->textures[i];
        if (t.kind > CRT_TEX_BITMAP) return bad("unknown texture kind");
        if (t.kind == CRT_TEX_BITMAP) {
            if (t.width == 0 || t.height == 0) return bad("empty bitmap");
            if (t.texel_offset + (uint64_t)t.width * t.height > s->n_texels) return bad("bitmap texels out of range");
        }
    }
    // Leaf links must point inside THEIR entry array: a leaf of the top-level tree lists meshes, a leaf of a mesh tree
    // lists triangles.  Which tree a node belongs to is decided by reachability from top_root (links point forward,
    // so the walk below is finite); both arrays end with a terminated entry (checked above), so a leaf that begins
    // inside its array also ends inside it.
    {
        std::vector<bool> is_top(s->n_nodes, false);
        std::vector<uint32_t> stack{s->top_root};
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (i >= s->n_nodes || is_top[i]) continue;
            is_top[i] = true;
            const crt_node &n = s->nodes[i];
            if (is_leaf_link(n.link) || n.link == CRT_LINK_END) continue;
            stack.push_back(n.link);
            const uint32_t c2 = s->nodes[n.link].miss;
            if (c2 != n.miss && c2 != CRT_LINK_END) stack.push_back(c2);
        }
        for (uint32_t i = 0; i < s->n_nodes; i++) {
            const crt_node &n = s->nodes[i];
            if (!is_leaf_link(n.link)) continue;
            const uint64_t b = n.link & ~CRT_LINK_LEAF;
            if (is_top[i]) { if (b >= s->n_leaf_meshes) return bad("top-level leaf begins outside leaf_meshes"); }
            else if (b >= s->n_leaf_triangles) return bad("mesh-tree leaf begins outside leaf_triangles");
        }
        for (uint32_t i = 0; i < s->n_meshes; i++)
            if (is_top[s->meshes[i].root]) return bad("mesh root inside the top-level tree");
    }
    return CRT_OK;
}


extern "C" void crt_tuning_defaults(crt_tuning *t) {
    if (!t) return;
    memset(t, 0, sizeof(*t));
    t->size = (uint32_t)sizeof(*t);
    t->mode = CRT_MODE_STREAM;
    t->step_budget = 384; t->shadow_budget = 4096; t->level0_budget = 0;
    t->heavy_level = 100000; t->side_blocks = 3;
    t->node_cap = t->ray_cap = t->shadow_cap = 0;
    t->bvh = 1;
    t->level_queue = 1;
    t->fetch_chunk = 256u | (64u << 16);
}

extern "C" int crt_create(const crt_scene_desc *s, int device, crt_ctx **out) { return crt_create_tuned(s, device, nullptr, out); }

extern "C" int crt_create_tuned(const crt_scene_desc *s, int device, const crt_tuning *tuning, crt_ctx **out) {
    if (!out) return CRT_ERR_INVALID;
    *out = nullptr;
    crt_tuning tune;
    crt_tuning_defaults(&tune);
    if (tuning) {
        if (tuning->size == 0 || tuning->size > sizeof(crt_tuning) || (tuning->size & 3u)) {
            g_create_error = "crt_tuning.size is not set (call crt_tuning_defaults first)";
            return CRT_ERR_INVALID;
        }
        memcpy(&tune, tuning, tuning->size);  // an older, shorter struct keeps the defaults of the newer fields
        tune.size = (uint32_t)sizeof(crt_tuning);
        if (tune.mode > CRT_MODE_LANES) { g_create_error = "crt_tuning.mode out of range"; return CRT_ERR_INVALID; }
        const uint32_t c_lo = tune.fetch_chunk & 0xFFFFu, c_hi = tune.fetch_chunk >> 16;
        if (c_lo < 64u || c_hi < 64u || (c_lo & 63u) || (c_hi & 63u)) {
            g_create_error = "crt_tuning.fetch_chunk: both halves must be multiples of 64, at least 64";
            return CRT_ERR_INVALID;
        }
    }
    int rc = validate_scene(s, g_create_error);
    if (rc != CRT_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device available (this library has no CPU fallback)";
        return CRT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "device index out of range";
        return CRT_ERR_NO_DEVICE;
    }
    crt_ctx *ctx = new (std::nothrow) crt_ctx();
    if (!ctx) return CRT_ERR_NOMEM;
    ctx->device = device;
    auto fail = [&](int code) {
        g_create_error = ctx->error;
        crt_destroy(ctx);
        return code;
    };
#define CK(expr)                                                                  \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_);       \
            return fail(CRT_ERR_HIP);                                             \
        }                                                                         \
    } while (0)
    CK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, device));
    ctx->num_cus = prop.multiProcessorCount;
    CK(hipStreamCreate(&ctx->stream));
    {   // the side stream (bulk shadow pass) at the lowest stream priority: nothing waits for it until the levels are done
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CK(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, least));
        CK(hipStreamCreateWithPriority(&ctx->early, hipStreamNonBlocking, greatest));   // (a hardware queue of its own: the level queue's launch must run BESIDE level 0)
    }
    for (int i = 0; i < crt_ctx::EV_RING; i++) {
        CK(hipEventCreate(&ctx->ev0[i]));
        CK(hipEventCreate(&ctx->ev1[i]));
        CK(hipEventCreate(&ctx->ev2[i]));
        CK(hipEventCreate(&ctx->ev3[i]));
        CK(hipEventCreate(&ctx->ev4[i]));
        CK(hipEventCreate(&ctx->ev_fork[i]));
        CK(hipEventCreate(&ctx->ev_s0[i]));
        CK(hipEventCreate(&ctx->ev_s1[i]));
        CK(hipEventCreate(&ctx->ev_s2[i]));
        CK(hipEventCreateWithFlags(&ctx->ev_reset[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&ctx->ev_queue[i], hipEventDisableTiming));
    }

    ctx->width = s->width;
    ctx->height = s->height;
    ctx->tiles_x = (s->width + TILE - 1) / TILE;
    ctx->tiles_y = (s->height + TILE - 1) / TILE;
    SceneArgs &A = ctx->scene;
    static_assert(sizeof(crt_node) == 32 && sizeof(crt_triangle) == 64, "record sizes");
    if (upload(ctx, (const float4 *)s->nodes, (size_t)s->n_nodes * 2, &A.nodes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_triangles, (size_t)s->n_leaf_triangles, &A.leaf_tris)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->leaf_meshes, (size_t)s->n_leaf_meshes, &A.leaf_meshes)) return fail(CRT_ERR_HIP);
    if (upload(ctx, (const float4 *)s->triangles, (size_t)s->n_triangles * 4, &A.tris)) return fail(CRT_ERR_HIP);
    {
        // leaf-order triangle records: what a walk reads inside a leaf, without the index indirection
        std::vector<float4> lt((size_t)s->n_leaf_triangles * 4);
        for (uint64_t e = 0; e < s->n_leaf_triangles; e++) {
            const uint32_t ent = s->leaf_triangles[e], tri = ent & ~CRT_ENTRY_LAST;
            const crt_triangle &T = s->triangles[tri];
            lt[4 * e + 0] = make_float4(T.v0[0], T.v0[1], T.v0[2], T.nx);
            lt[4 * e + 1] = make_float4(T.v1[0], T.v1[1], T.v1[2], T.ny);
            lt[4 * e + 2] = make_float4(T.v2[0], T.v2[1], T.v2[2], T.nz);
            float idbits, lastbits;
            const uint32_t last = (ent & CRT_ENTRY_LAST) ? 1u : 0u;
            memcpy(&idbits, &tri, 4);
            memcpy(&lastbits, &last, 4);
            lt[4 * e + 3] = make_float4(T.plane, idbits, lastbits, 0.0f);
        }
        if (upload(ctx, lt.data(), lt.size(), &A.ltris)) return fail(CRT_ERR_HIP);
    }
    // Compact forms for the plan kernels (kernel_plan.h), whose cost is the number of vector-memory instructions per step:
    // a triangle as 3 x float4 {v0,nx} {v1,ny} {v2,nz} (the plane offset is recomputed, -(v0 . n) as on the host, Ray.cpp:17;
    // the triangle's id is looked up in leaf_tris only for a hit) and a leaf node's link as LEAF | (count - 1) << 24 | first
    // entry, so that a walk knows where its leaf ends without the record's `last` flag.
    std::vector<uint32_t> compact_link(s->n_nodes);
    {
        bool ok = s->n_leaf_triangles < (1ull << 24);
        std::vector<float4> pn((size_t)s->n_nodes * 2);
        for (uint32_t i = 0; i < s->n_nodes; i++) {
            const crt_node &n = s->nodes[i];
            uint32_t link = n.link;
            if (is_leaf_link(n.link)) {
                const uint64_t begin = n.link & ~CRT_LINK_LEAF;
                uint64_t count = 0;
                if (begin < s->n_leaf_triangles) {  // (a top-level leaf's link indexes leaf_meshes: whatever this gives is not used)
                    uint64_t e = begin;
                    do { count++; } while (!(s->leaf_triangles[e++] & CRT_ENTRY_LAST) && e < s->n_leaf_triangles);
                }
                if (count >= 1 && count <= 128 && begin < (1ull << 24)) link = CRT_LINK_LEAF | (uint32_t)((count - 1) << 24) | (uint32_t)begin;
                else if (begin < s->n_leaf_triangles) ok = false;
            }
            compact_link[i] = link;
            float lb, mb;
            memcpy(&lb, &link, 4);
            memcpy(&mb, &n.miss, 4);
            pn[2 * (size_t)i] = make_float4(n.lo[0], n.lo[1], n.lo[2], mb);
            pn[2 * (size_t)i + 1] = make_float4(n.hi[0], n.hi[1], n.hi[2], lb);
        }
        std::vector<float4> pt(ok ? (size_t)s->n_leaf_triangles * 3 : 0);
        for (uint64_t e = 0; ok && e < s->n_leaf_triangles; e++) {
            const crt_triangle &T = s->triangles[s->leaf_triangles[e] & ~CRT_ENTRY_LAST];
            pt[3 * e + 0] = make_float4(T.v0[0], T.v0[1], T.v0[2], T.nx);
            pt[3 * e + 1] = make_float4(T.v1[0], T.v1[1], T.v1[2], T.ny);
            pt[3 * e + 2] = make_float4(T.v2[0], T.v2[1], T.v2[2], T.nz);
            // the kernels recompute the plane offset: it must be the stored one, bit for bit, or this form is not used
            const float plane = -(T.v0[0] * T.nx + T.v0[1] * T.ny + T.v0[2] * T.nz);
            if (memcmp(&plane, &T.plane, 4) != 0) ok = false;
        }
        if (!ok) { pt.clear(); pn.clear(); }
        A.plan_compact = ok ? 1u : 0u;
        if (upload(ctx, pt.data(), pt.size(), &A.ptris)) return fail(CRT_ERR_HIP);
        if (upload(ctx, pn.data(), pn.size(), &A.pnodes)) return fail(CRT_ERR_HIP);
    }
    std::vector<HeavyMesh> hmesh_host;  // filled with the leaf sequences below, read again for the single-leaf mesh table
    std::vector<bool> is_top(s->n_nodes, false);
    bool top_is_range = false;  // the top-level tree's nodes are ONE index range [top_first, top_first + top_count)
    {
        // the top-level tree's nodes: reachable from top_root (links point forward, so the walk is finite)
        std::vector<uint32_t> stack{s->top_root};
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            if (i >= s->n_nodes || is_top[i]) continue;
            is_top[i] = true;
            const crt_node &n = s->nodes[i];
            if (is_leaf_link(n.link) || n.link == CRT_LINK_END) continue;
            stack.push_back(n.link);
            const uint32_t c2 = s->nodes[n.link].miss;
            if (c2 != n.miss && c2 != CRT_LINK_END) stack.push_back(c2);
        }
        uint32_t lo = UINT32_MAX, hi = 0, cnt = 0;
        for (uint32_t i = 0; i < s->n_nodes; i++)
            if (is_top[i]) { lo = i < lo ? i : lo; hi = i > hi ? i : hi; cnt++; }
        top_is_range = cnt > 0 && hi - lo + 1 == cnt;
        A.top_fast = (top_is_range && cnt <= 64u && s->n_leaf_meshes <= 128u && s->n_meshes <= 64u) ? 1u : 0u;
        A.top_first = cnt ? lo : 0u;
        A.top_count = cnt;
        A.top_leaf_entries = s->n_leaf_meshes;
        A.top_meshes = s->n_meshes;
    }
    {
        // Leaf sequence of every mesh tree (kernel_heavy.h): the leaves' own boxes in visit order, then union
        // boxes of 64 entries per level until at most 64 remain.  With forward links the nodes of a mesh tree
        // are the index range [root, next tree's root), already in visit order.
        std::vector<uint32_t> roots;
        roots.push_back(s->top_root);
        for (uint32_t m = 0; m < s->n_meshes; m++) roots.push_back(s->meshes[m].root);
        std::sort(roots.begin(), roots.end());
        std::vector<float4> hbox;
        std::vector<HeavyMesh> &hm = hmesh_host;
        hm.assign(s->n_meshes, HeavyMesh{});
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            HeavyMesh &H = hm[m];
            memset(&H, 0, sizeof(H));
            const uint32_t root = s->meshes[m].root;
            auto it = std::upper_bound(roots.begin(), roots.end(), root);
            const uint32_t end = it == roots.end() ? s->n_nodes : *it;
            std::vector<float4> level;  // 2 x float4 per entry
            for (uint32_t i = root; i < end; i++) {
                const crt_node &n = s->nodes[i];
                if (!is_leaf_link(n.link)) continue;
                const uint32_t begin = n.link & ~CRT_LINK_LEAF;
                uint32_t count = 0;
                if (begin < s->n_leaf_triangles) {
                    uint64_t e = begin;
                    do { count++; } while (!(s->leaf_triangles[e++] & CRT_ENTRY_LAST) && e < s->n_leaf_triangles);
                }
                float bb, cb;
                memcpy(&bb, &begin, 4);
                memcpy(&cb, &count, 4);
                level.push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], bb));
                level.push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], cb));
            }
            uint32_t nl = 0;
            while (!level.empty() && nl < 4) {
                const uint32_t cnt = (uint32_t)(level.size() / 2);
                H.first[nl] = (uint32_t)(hbox.size() / 2);
                H.count[nl] = cnt;
                hbox.insert(hbox.end(), level.begin(), level.end());
                nl++;
                if (cnt <= 64) break;
                std::vector<float4> up;
                for (uint32_t g = 0; g < cnt; g += 64) {
                    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    for (uint32_t k = g; k < cnt && k < g + 64; k++) {
                        const float4 &a = level[2 * k], &b = level[2 * k + 1];
                        lo[0] = a.x < lo[0] ? a.x : lo[0]; lo[1] = a.y < lo[1] ? a.y : lo[1]; lo[2] = a.z < lo[2] ? a.z : lo[2];
                        hi[0] = b.x > hi[0] ? b.x : hi[0]; hi[1] = b.y > hi[1] ? b.y : hi[1]; hi[2] = b.z > hi[2] ? b.z : hi[2];
                    }
                    up.push_back(make_float4(lo[0], lo[1], lo[2], 0.0f));
                    up.push_back(make_float4(hi[0], hi[1], hi[2], 0.0f));
                }
                level.swap(up);
            }
            // more than 64^4 leaves: leave n_levels = 0 for this mesh -> the heavy path is switched off below
            H.n_levels = (!level.empty() && H.count[nl ? nl - 1 : 0] <= 64) ? nl : 0;
            if (H.n_levels == 0 && !level.empty()) ctx->step_budget = 0;
        }
        if (upload(ctx, hbox.data(), hbox.size(), &A.hbox)) return fail(CRT_ERR_HIP);
        if (upload(ctx, hm.data(), hm.size(), &A.hmesh)) return fail(CRT_ERR_HIP);
    }
    if (upload(ctx, s->triangle_vertices, (size_t)s->n_triangles * 3, &A.tri_verts)) return fail(CRT_ERR_HIP);
    if (upload(ctx, s->vertex_normals, (size_t)s->n_vertices * 3, &A.vnormals)) return fail(CRT_ERR_HIP);
    if (s->vertex_uvs) {
        if (upload(ctx, s->vertex_uvs, (size_t)s->n_vertices * 3, &A.vuvs)) return fail(CRT_ERR_HIP);
    } else A.vuvs = nullptr;
    {
        // the meshes' device copy marks single-leaf meshes (kernel_heavy.h: TinyResults): at most 64 of them, at most 512 triangles in all
        std::vector<crt_mesh> dm(s->meshes, s->meshes + s->n_meshes);
        std::vector<uint32_t> tiny_at, tiny_flags;
        uint64_t tiny_tris = 0;
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            dm[m].pad = 0;
            const crt_node &root = s->nodes[s->meshes[m].root];
            const HeavyMesh &H = hmesh_host[m];
            if (!is_leaf_link(root.link) || H.n_levels != 1 || H.count[0] != 1 || tiny_at.size() >= 64) continue;
            uint32_t cnt = 0;
            for (uint64_t e = root.link & ~CRT_LINK_LEAF; e < s->n_leaf_triangles; e++) { cnt++; if (s->leaf_triangles[e] & CRT_ENTRY_LAST) break; }
            if (tiny_tris + cnt > 512) continue;
            tiny_tris += cnt;
            dm[m].pad = (uint32_t)tiny_at.size() + 1u;
            tiny_at.push_back(H.first[0]);
            tiny_flags.push_back(s->meshes[m].flags);
        }
        A.tiny_count = (uint32_t)tiny_at.size();
        if (upload(ctx, tiny_at.data(), tiny_at.size(), &A.tiny_at)) return fail(CRT_ERR_HIP);
        if (upload(ctx, tiny_flags.data(), tiny_flags.size(), &A.tiny_flags)) return fail(CRT_ERR_HIP);
        if (upload(ctx, dm.data(), dm.size(), &A.meshes)) return fail(CRT_ERR_HIP);
    }
    {
        std::vector<DMaterial> mats(s->n_materials);
        for (uint32_t i = 0; i < s->n_materials; i++) {
            const crt_material &m = s->materials[i];
            mats[i] = DMaterial{m.albedo[0], m.albedo[1], m.albedo[2], m.ior, m.type, m.smooth, m.texture, 0};
        }
        if (upload(ctx, mats.data(), mats.size(), &A.materials)) return fail(CRT_ERR_HIP);
        std::vector<DTexture> tex(s->n_textures);
        for (uint32_t i = 0; i < s->n_textures; i++) {
            const crt_texture &t = s->textures[i];
            tex[i] = DTexture{t.kind, t.color_a[0], t.color_a[1], t.color_a[2], t.color_b[0], t.color_b[1], t.color_b[2],
                              t.scalar, t.width, t.height, t.texel_offset};
        }
        if (upload(ctx, tex.data(), tex.size(), &A.textures)) return fail(CRT_ERR_HIP);
        std::vector<uint32_t> px((size_t)s->n_texels);
        for (uint64_t i = 0; i < s->n_texels; i++)
            px[i] = (uint32_t)s->texels[3 * i] | ((uint32_t)s->texels[3 * i + 1] << 8) | ((uint32_t)s->texels[3 * i + 2] << 16);
        if (upload(ctx, px.data(), px.size(), &A.texels)) return fail(CRT_ERR_HIP);
        std::vector<float4> lights(s->n_lights);
        for (uint32_t i = 0; i < s->n_lights; i++)
            lights[i] = make_float4(s->lights[i].position[0], s->lights[i].position[1], s->lights[i].position[2],
                                    (float)s->lights[i].intensity);  // static_cast<float>(light.intentsity), RayTracer.cpp:320
        if (upload(ctx, lights.data(), lights.size(), &A.lights)) return fail(CRT_ERR_HIP);
    }
    A.n_lights = s->n_lights;
    A.top_root = s->top_root;
    {
        // Are child boxes nested in their parent's box?  (True for every tree the reference builds.)  With
        // forward links, the nodes in (i, miss_i) are exactly the descendants of inner node i.
        bool nested = true;
        std::vector<uint32_t> stack;  // enclosing inner nodes of the current position
        std::vector<bool> root(s->n_nodes, false);
        root[s->top_root] = true;
        for (uint32_t m = 0; m < s->n_meshes; m++) root[s->meshes[m].root] = true;
        for (uint32_t i = 0; i < s->n_nodes && nested; i++) {
            while (!stack.empty() && s->nodes[stack.back()].miss != CRT_LINK_END && s->nodes[stack.back()].miss <= i) stack.pop_back();
            // a tree root starts a new nesting chain: it is not a descendant of the previous tree's nodes
            if (root[i]) stack.clear();
            if (!stack.empty()) {
                const crt_node &p = s->nodes[stack.back()], &c = s->nodes[i];
                for (int a = 0; a < 3; a++)
                    if (!(c.lo[a] >= p.lo[a] && c.hi[a] <= p.hi[a])) nested = false;
            }
            if (!is_leaf_link(s->nodes[i].link)) stack.push_back(i);
        }
        A.nested_boxes = nested ? 1u : 0u;
    }
    {
        // The plan of the top-level tree (kernel_plan.h; kernel_heavy.h walks the same table as a leaf sequence): its leaves in visit order, which is index order.
        std::vector<float4> boxes, boxes_all;
        std::vector<uint32_t> order, order_all;  // non-refractive meshes / every mesh (the GI mode's shadow rays), most leaves first
        for (uint32_t m = 0; m < s->n_meshes; m++) {
            if (!(s->meshes[m].flags & 1u)) order.push_back(m);
            order_all.push_back(m);
        }
        auto more_leaves = [&](uint32_t a, uint32_t b) { return hmesh_host[a].count[0] > hmesh_host[b].count[0]; };
        std::stable_sort(order.begin(), order.end(), more_leaves);
        std::stable_sort(order_all.begin(), order_all.end(), more_leaves);
        constexpr uint32_t MAX_BITS = 256;  // shadow mask bits of the wide plan (kernel_plan.h: 8 words); the plan proper uses the first 64
        std::vector<uint32_t> bit_of(s->n_meshes, MAX_BITS), bit_of_all(s->n_meshes, MAX_BITS);
        for (size_t b = 0; b < order.size() && b < MAX_BITS; b++) bit_of[order[b]] = (uint32_t)b;
        for (size_t b = 0; b < order_all.size() && b < MAX_BITS; b++) bit_of_all[order_all[b]] = (uint32_t)b;
        // the plan reads the top-level leaves off the index range [top_first, top_first + top_count): a scene whose top-level nodes are
        // interleaved with mesh-tree nodes (forward links allow it) has no plan and stays on the faithful kernels
        const bool contiguous = top_is_range && A.nested_boxes;
        uint32_t n_leaves = 0;
        std::vector<float4> groups;
        for (uint32_t i = A.top_first; contiguous && i < A.top_first + A.top_count; i++) {
            const crt_node &n = s->nodes[i];
            if (!is_leaf_link(n.link)) continue;
            const uint32_t begin = n.link & ~CRT_LINK_LEAF;
            uint32_t count = 0;
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mask_all[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t e = begin; e < s->n_leaf_meshes; e++) {
                const uint32_t mi = s->leaf_meshes[e] & ~CRT_ENTRY_LAST;
                if (bit_of[mi] < MAX_BITS) mask[bit_of[mi] >> 5] |= 1u << (bit_of[mi] & 31u);
                if (bit_of_all[mi] < MAX_BITS) mask_all[bit_of_all[mi] >> 5] |= 1u << (bit_of_all[mi] & 31u);
                count++;
                if (s->leaf_meshes[e] & CRT_ENTRY_LAST) break;
            }
            float bb, cb, mf[8];
            memcpy(&bb, &begin, 4);
            memcpy(&cb, &count, 4);
            for (std::vector<float4> *table : {&boxes, &boxes_all}) {
                memcpy(mf, table == &boxes ? mask : mask_all, sizeof(mf));
                table->push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], bb));  // PLAN_LEAF_DWORDS = 16 per leaf (kernel_plan.h)
                table->push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], cb));
                table->push_back(make_float4(mf[0], mf[1], mf[2], mf[3]));
                table->push_back(make_float4(mf[4], mf[5], mf[6], mf[7]));
            }
            // the wide plan's groups: PLAN_GROUP_LEAVES consecutive leaves under their union box
            if (n_leaves % PLAN_GROUP_LEAVES == 0) {
                float first_leaf, zero = 0.0f;
                memcpy(&first_leaf, &n_leaves, 4);
                groups.push_back(make_float4(n.lo[0], n.lo[1], n.lo[2], first_leaf));
                groups.push_back(make_float4(n.hi[0], n.hi[1], n.hi[2], zero));
            }
            float4 &glo = groups[groups.size() - 2], &ghi = groups[groups.size() - 1];
            glo.x = std::min(glo.x, n.lo[0]); glo.y = std::min(glo.y, n.lo[1]); glo.z = std::min(glo.z, n.lo[2]);
            ghi.x = std::max(ghi.x, n.hi[0]); ghi.y = std::max(ghi.y, n.hi[1]); ghi.z = std::max(ghi.z, n.hi[2]);
            const uint32_t in_group = n_leaves % PLAN_GROUP_LEAVES + 1u;
            memcpy(&ghi.w, &in_group, 4);
            n_leaves++;
        }
        // the per-lane plan kernels keep a ray's meshes in two 32-bit words (the plan proper) or eight (the wide plan)
        A.plan_ok = (contiguous && A.top_fast && n_leaves <= 64u && s->n_meshes <= 64u && A.plan_compact) ? 1u : 0u;
        A.plan_wide = (!A.plan_ok && contiguous && n_leaves > 0 && s->n_meshes <= MAX_BITS && A.plan_compact) ? 1u : 0u;
        A.plan_leaves = n_leaves;
        A.plan_seq = (contiguous && n_leaves > 0) ? 1u : 0u;  // the wave-per-ray kernels walk the leaf sequence (kernel_heavy.h)
        A.plan_group_count = (uint32_t)(groups.size() / 2);
        A.plan_shadow_bits = (uint32_t)std::min<size_t>(order.size(), A.plan_ok ? 64u : MAX_BITS);
        A.plan_list_words = std::min((s->n_meshes + 3u) / 4u, 32u);
        if (upload(ctx, groups.data(), groups.size(), &A.plan_groups)) return fail(CRT_ERR_HIP);
        if (upload(ctx, boxes.data(), boxes.size(), &A.plan_boxes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, order.data(), order.size(), &A.plan_shadow_mesh)) return fail(CRT_ERR_HIP);
        A.plan_shadow_bits_all = (uint32_t)std::min<size_t>(order_all.size(), A.plan_ok ? 64u : MAX_BITS);
        if (upload(ctx, boxes_all.data(), boxes_all.size(), &A.plan_boxes_all)) return fail(CRT_ERR_HIP);
        if (upload(ctx, order_all.data(), order_all.size(), &A.plan_shadow_mesh_all)) return fail(CRT_ERR_HIP);
    }
    {
        // The candidate filter (crt_bvh.h): built on the host, once; a scene it cannot be built for renders on the reference-order kernels
        BvhHost bvh;
        bvh_build(s, A.nested_boxes != 0, bvh);
        if (bvh.ok && !A.plan_compact) { bvh.ok = false; bvh.why = "no compact leaf links"; }  // (kernel_bvh.h: bvh_leaf_walk reads them)
        A.bvh_stack = 3u * bvh.wide_depth + 1u;   // what a walk can push: three children per inner node of a path
        if (bvh.ok && A.bvh_stack > 1024u) { bvh.ok = false; bvh.why = "hierarchy too deep"; }
        ctx->bvh_note = bvh.ok ? "" : bvh.why;
        if (bvh.ok) {
            char note[160];
            snprintf(note, sizeof(note), "nodes:%zu,entries:%zu,depth:%u/%u,walk_triangles:%u,max_margin:%.3g", bvh.nodes.size(), bvh.ids.size(),
                     bvh.max_depth, bvh.wide_depth, bvh.walk_triangles, bvh.max_margin);
            ctx->bvh_stats = note;
        }
        // the leaf sequences of the wave-per-ray kernels were read off index ranges [root, next root): a description whose trees are
        // interleaved (forward links allow it) is walked by the faithful kernels, which only follow links
        if (!bvh.trees_are_ranges) ctx->step_budget = 0;
        A.bvh_ok = bvh.ok ? 1u : 0u;
        A.bvh_extent = bvh.extent;
        A.bvh_overlap_eps = bvh.overlap_eps;
        A.n_bvh_nodes = (uint32_t)bvh.nodes.size(); A.n_bvh_entries = (uint32_t)bvh.ids.size(); A.n_triangles = s->n_triangles; A.n_nodes = s->n_nodes;
        A.n_leaf_tris = (uint32_t)s->n_leaf_triangles; A.n_tri_leaf_entries = (uint32_t)(bvh.tri_leaf_list.size() / 8);
        A.n_mesh_top_entries = (uint32_t)(bvh.mesh_top_list.size() / 8); A.n_meshes = s->n_meshes;
        if (!bvh.ok) bvh = BvhHost{};
        static_assert(sizeof(BvhNode) == 8 * sizeof(float4), "BvhNode = 8 x float4");
        if (upload(ctx, (const float4 *)bvh.nodes.data(), bvh.nodes.size() * 8, &A.bvh_nodes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.vnodes.data(), bvh.vnodes.size() / 4, &A.bvh_vnodes)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.cones.data(), bvh.cones.size() / 4, &A.bvh_cones)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.tris.data(), bvh.tris.size() / 4, &A.bvh_tris)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.ids.data(), bvh.ids.size(), &A.bvh_ids)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.tri_mesh.data(), bvh.tri_mesh.size(), &A.tri_mesh)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.tri_leaf_first.data(), bvh.tri_leaf_first.size(), &A.tri_leaf_first)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.tri_leaf_list.data(), bvh.tri_leaf_list.size() / 4, &A.tri_leaf_list)) return fail(CRT_ERR_HIP);
        if (upload(ctx, bvh.mesh_top_first.data(), bvh.mesh_top_first.size(), &A.mesh_top_first)) return fail(CRT_ERR_HIP);
        if (upload(ctx, (const float4 *)bvh.mesh_top_list.data(), bvh.mesh_top_list.size() / 4, &A.mesh_top_list)) return fail(CRT_ERR_HIP);
    }
    A.bgx = s->background[0]; A.bgy = s->background[1]; A.bgz = s->background[2];
    A.width = s->width; A.height = s->height; A.tiles_x = ctx->tiles_x;
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(ctx->frame.cam, ident, sizeof(ident));
    ctx->frame.cam_pos[0] = ctx->frame.cam_pos[1] = ctx->frame.cam_pos[2] = 0;

    size_t frame_bytes = (size_t)s->width * s->height * 3 * sizeof(float);
    CK(hipMalloc((void **)&ctx->d_frame, frame_bytes));
    CK(hipMemset(ctx->d_frame, 0, frame_bytes));  // colorBuffer starts as Color() = (0,0,0), RayTracer.cpp:46-50
    CK(hipMalloc((void **)&ctx->d_quant, (size_t)s->width * s->height * 3));
    CK(hipMalloc((void **)&ctx->d_sync, 4 * sizeof(uint32_t)));
    ctx->tuning = tune;
    ctx->mode = tune.mode == CRT_MODE_LANES ? crt_ctx::MODE_LANES : crt_ctx::MODE_STREAM;
    if (ctx->step_budget) ctx->step_budget = tune.step_budget;  // (0: a mesh with too many leaves switched the wave-per-ray path off)
    ctx->lean_ok = s->n_nodes < (1u << 27) && s->n_leaf_triangles < (1ull << 26);
    CK(hipMalloc((void **)&ctx->d_lq_words, LQ_WORDS * sizeof(uint32_t)));
    CK(hipMemset(ctx->d_lq_words, 0, LQ_WORDS * sizeof(uint32_t)));
    CK(hipMalloc((void **)&ctx->d_exec, 6 * sizeof(unsigned long long)));
    CK(hipMemset(ctx->d_exec, 0, 6 * sizeof(unsigned long long)));
    CK(hipMalloc((void **)&ctx->d_scounts, SC_ALLOC_WORDS * sizeof(uint32_t)));
    CK(hipMemset(ctx->d_scounts, 0, SC_ALLOC_WORDS * sizeof(uint32_t)));
    static_assert(crt_ctx::H_SLOT_WORDS == SC_ALLOC_WORDS + 1, "pinned slot = counter block + fallback total");
    CK(hipHostMalloc((void **)&ctx->h_ring, (size_t)crt_ctx::EV_RING * crt_ctx::H_SLOT_WORDS * sizeof(uint32_t)));
    memset(ctx->h_ring, 0, (size_t)crt_ctx::EV_RING * crt_ctx::H_SLOT_WORDS * sizeof(uint32_t));
    ctx->last_counts.assign(SC_ALLOC_WORDS, 0u);
    CK(hipMalloc((void **)&ctx->d_fallback_total, sizeof(uint32_t)));
    CK(hipMemset(ctx->d_fallback_total, 0, sizeof(uint32_t)));
    ctx->n_lights = s->n_lights;
    CK(hipMalloc((void **)&ctx->d_counters, 3 * C_N * sizeof(unsigned long long)));  // [levels | shadow pass 0 | the rest]
    // persistent grid: 8 blocks of 256 threads per CU gives every CU its 32 waves if registers allow
    ctx->grid_blocks = (uint32_t)ctx->num_cus * 8u;
    if (ctx->scene.bvh_ok && ctx->scene.bvh_stack > BVH_LDS_STACK)
        CK(hipMalloc((void **)&ctx->d_bvh_spill, 2u * (size_t)ctx->grid_blocks * BLOCK * (ctx->scene.bvh_stack - BVH_LDS_STACK) * sizeof(uint32_t)));
    // the argument blocks (kernel_common.h): the scene's once, a slot per frame in flight for the frames'
    CK(hipMalloc((void **)&ctx->d_scene, sizeof(SceneArgs)));
    CK(hipMemcpy(ctx->d_scene, &ctx->scene, sizeof(SceneArgs), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&ctx->d_frame_ring, (size_t)crt_ctx::EV_RING * sizeof(FrameArgs)));
    CK(hipHostMalloc((void **)&ctx->h_frame_ring, (size_t)crt_ctx::EV_RING * sizeof(FrameArgs)));
#undef CK
    *out = ctx;
    return CRT_OK;
}

extern "C" void crt_destroy(crt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (void *p : ctx->allocs) (void)hipFree(p);
    if (ctx->d_frame) (void)hipFree(ctx->d_frame);
    if (ctx->d_quant) (void)hipFree(ctx->d_quant);
    if (ctx->d_items) (void)hipFree(ctx->d_items);
    if (ctx->d_sync) (void)hipFree(ctx->d_sync);
    for (int i = 0; i < 2; i++) if (ctx->d_rayq[i]) (void)hipFree(ctx->d_rayq[i]);
    if (ctx->d_shadowq) (void)hipFree(ctx->d_shadowq);
    if (ctx->d_occluded) (void)hipFree(ctx->d_occluded);
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_scounts) (void)hipFree(ctx->d_scounts);
    if (ctx->d_exec) (void)hipFree(ctx->d_exec);
    if (ctx->d_heavy) (void)hipFree(ctx->d_heavy);
    if (ctx->d_sheavy) (void)hipFree(ctx->d_sheavy);
    if (ctx->d_hits) (void)hipFree(ctx->d_hits);
    if (ctx->d_hits_all) (void)hipFree(ctx->d_hits_all);
    if (ctx->d_bvh_spill) (void)hipFree(ctx->d_bvh_spill);
    if (ctx->d_lq) (void)hipFree(ctx->d_lq);
    if (ctx->d_lq_words) (void)hipFree(ctx->d_lq_words);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->h_frame_ring) (void)hipHostFree(ctx->h_frame_ring);
    if (ctx->d_frame_ring) (void)hipFree(ctx->d_frame_ring);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    if (ctx->d_fallback_total) (void)hipFree(ctx->d_fallback_total);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_frames) (void)hipFree(ctx->d_frames);
    for (int i = 0; i < crt_ctx::EV_RING; i++) {
        if (ctx->ev0[i]) (void)hipEventDestroy(ctx->ev0[i]);
        if (ctx->ev1[i]) (void)hipEventDestroy(ctx->ev1[i]);
        if (ctx->ev2[i]) (void)hipEventDestroy(ctx->ev2[i]);
        if (ctx->ev3[i]) (void)hipEventDestroy(ctx->ev3[i]);
        if (ctx->ev4[i]) (void)hipEventDestroy(ctx->ev4[i]);
        if (ctx->ev_fork[i]) (void)hipEventDestroy(ctx->ev_fork[i]);
        if (ctx->ev_s0[i]) (void)hipEventDestroy(ctx->ev_s0[i]);
        if (ctx->ev_s1[i]) (void)hipEventDestroy(ctx->ev_s1[i]);
        if (ctx->ev_s2[i]) (void)hipEventDestroy(ctx->ev_s2[i]);
        if (ctx->ev_reset[i]) (void)hipEventDestroy(ctx->ev_reset[i]);
        if (ctx->ev_queue[i]) (void)hipEventDestroy(ctx->ev_queue[i]);
    }
    if (ctx->ev_call0) (void)hipEventDestroy(ctx->ev_call0);
    if (ctx->ev_call1) (void)hipEventDestroy(ctx->ev_call1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->early) (void)hipStreamDestroy(ctx->early);
    delete ctx;
}

extern "C" const char *crt_last_error(const crt_ctx *ctx) {
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

extern "C" int crt_set_camera(crt_ctx *ctx, const float position[3], const float matrix[9]) {
    if (!ctx || !position || !matrix) return CRT_ERR_INVALID;
    memcpy(ctx->frame.cam_pos, position, 3 * sizeof(float));
    memcpy(ctx->frame.cam, matrix, 9 * sizeof(float));
    return CRT_OK;
}
