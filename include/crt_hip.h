/* crt_hip.h -- C ABI of the MI355X-native per-pixel hot path (libcrt_hip.so).
 *
 * This is the drop-in boundary.  The reference has no FFI; its seam is the C++ class
 * `RayTracer` (reference: SourceCode/include/tracer/RayTracer.h:96-101).  Each entry point below
 * names the reference interface it replaces.  Plain pointers and sizes only: no C++ types, no
 * torch types, no exceptions across the boundary (the reference uses assert / throw,
 * SceneParser.cpp:41,203, Vector.cpp:21-23; here every call returns an int status).
 *
 * Ownership: the caller owns every array passed in and every output buffer; the context owns its
 * device copies; no caller pointer is retained after a call returns.
 * Threading: one context = one caller thread at a time (the reference's render() is not
 * re-entrant either, RayTracer.cpp:205-213).
 * There is NO CPU fallback: without a usable HIP device crt_create fails with CRT_ERR_NO_DEVICE.
 */
#ifndef CRT_HIP_H
#define CRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CRT_OK = 0,
    CRT_ERR_INVALID = 1,   /* bad argument / inconsistent description */
    CRT_ERR_NO_DEVICE = 2, /* no HIP device, or the device is not usable */
    CRT_ERR_HIP = 3,       /* a HIP runtime call failed (see crt_last_error) */
    CRT_ERR_NOMEM = 4,
    CRT_ERR_IO = 5,
    CRT_ERR_PARSE = 6
};

/* Material types -- same numbering as the reference's enum MaterialType (Material.h:7). */
enum { CRT_MAT_DIFFUSE = 0, CRT_MAT_REFLECTIVE = 1, CRT_MAT_CONSTANT = 2, CRT_MAT_REFRACTIVE = 3 };
/* Texture kinds (Texture.h:24-58). */
enum { CRT_TEX_ALBEDO = 0, CRT_TEX_EDGES = 1, CRT_TEX_CHECKER = 2, CRT_TEX_BITMAP = 3 };

#define CRT_LINK_END 0xFFFFFFFFu  /* "no next node" */
#define CRT_LINK_LEAF 0x80000000u /* bit 31 of crt_node.link: the node is a leaf (unless link == CRT_LINK_END) */
#define CRT_ENTRY_LAST 0x80000000u /* bit 31 of a leaf entry: last entry of its leaf */

/* One node of the flattened ("threaded") two-level tree, 32 bytes.
 * The reference walks its KD/AABB trees with an explicit stack, pushing children[0] then
 * children[1] so that children[1] is visited first, and never prunes by distance
 * (KDTree.cpp:53-74,132-155).  The visit order is therefore a fixed property of the tree, and it
 * is encoded here as two links per node instead of a stack:
 *   box test passes, inner node : continue at `link`            (first child in visit order)
 *   box test passes, leaf       : process entries starting at (link & ~CRT_LINK_LEAF), then `miss`
 *   box test fails              : continue at `miss`            (skip the whole subtree)
 * CRT_LINK_END terminates the walk.  Nodes are stored in visit order: `link` and `miss` always point
 * to a HIGHER index (crt_create rejects anything else), so a subtree is the index range [i, miss).  Node indices are global (top-level tree and all mesh trees
 * share one array). */
typedef struct crt_node {
    float lo[3];
    uint32_t miss;
    float hi[3];
    uint32_t link;
} crt_node;

/* Triangle record used by the intersection test, 64 bytes (Ray.cpp:9-31, Triangle.cpp:37-57):
 * positions, the unit face normal (Triangle.cpp:13-16) and plane = -(v0 . normal) (Ray.cpp:17). */
typedef struct crt_triangle {
    float v0[3]; float nx;
    float v1[3]; float ny;
    float v2[3]; float nz;
    float plane; uint32_t pad[3];
} crt_triangle;

typedef struct crt_mesh {
    uint32_t root;      /* global node index of this mesh's tree root */
    uint32_t material;  /* index into materials */
    uint32_t flags;     /* bit 0: material is refractive (shadow rays skip it, AccelerationStructure.cpp:67-71) */
    uint32_t pad;
} crt_mesh;

typedef struct crt_material {
    float albedo[3];
    float ior;
    uint32_t type;    /* CRT_MAT_* */
    uint32_t smooth;  /* smooth_shading */
    int32_t texture;  /* index into textures, -1 = constant albedo */
    uint32_t pad;
} crt_material;

typedef struct crt_texture {
    uint32_t kind;      /* CRT_TEX_* */
    float color_a[3];   /* albedo | inner_color | color_A */
    float color_b[3];   /*        | edge_color  | color_B */
    float scalar;       /*        | edge_width  | square_size */
    uint32_t width, height; /* bitmap only */
    uint64_t texel_offset;  /* bitmap only: first texel in crt_scene_desc.texels */
} crt_texture;

typedef struct crt_light {
    float position[3];
    uint32_t intensity; /* unsigned, as in the reference (Scene.h:22) */
} crt_light;

/* Everything the hot path reads (the reference's `Scene` + its AccelerationStructure, flattened). */
typedef struct crt_scene_desc {
    uint32_t width, height;
    float background[3];

    const crt_node *nodes;          uint32_t n_nodes;
    uint32_t top_root;              /* node index of the top-level (object) tree root */
    const uint32_t *leaf_triangles; uint64_t n_leaf_triangles; /* mesh-tree leaf entries: global triangle index | CRT_ENTRY_LAST */
    const uint32_t *leaf_meshes;    uint32_t n_leaf_meshes;    /* top-tree leaf entries: mesh index | CRT_ENTRY_LAST */

    const crt_triangle *triangles;  uint32_t n_triangles;
    const uint32_t *triangle_vertices; /* 3 global vertex indices per triangle */
    const float *vertex_normals;    /* 3 floats per vertex (Scene.cpp:21-29) */
    const float *vertex_uvs;        /* 3 floats per vertex, may be NULL when no material has a texture */
    uint32_t n_vertices;

    const crt_mesh *meshes;         uint32_t n_meshes;
    const crt_material *materials;  uint32_t n_materials;
    const crt_texture *textures;    uint32_t n_textures;
    const uint8_t *texels;          uint64_t n_texels;  /* RGB8, 3 bytes per texel, all bitmaps concatenated */
    const crt_light *lights;        uint32_t n_lights;
} crt_scene_desc;

/* RenderOptions (RayTracer.h:25-50).  use_gi selects the reference's GI / multi-sample mode (RayTracer.cpp:90-104, 331-354):
 * RAYS_PER_PIXEL primary rays per pixel (the first through the pixel centre, the others jittered) averaged, GI_SAMPLE_SIZE
 * diffuse reflection rays at every diffuse hit, and refractive meshes no longer skipped by shadow rays
 * (AccelerationStructure.cpp:67-71).  The reference seeds that mode's generator from clock() ^ thread id (RayTracer.cpp:28-30):
 * only the distribution of its images is defined.  Here the random numbers come from a counter-based generator keyed by
 * (gi_seed, pixel, sample, position in the ray tree) -- csrc/gi_random.h -- so a frame is a function of its options, whatever
 * the device count or tile order (the GI mode's frames go through the ray-stream kernels like any other: csrc/kernel_stream.h,
 * csrc/kernel_plan.h; pixel by pixel -- csrc/kernel_lane.h -- only when level 0 does not fit 31-bit ray indices). */
typedef struct crt_options {
    uint32_t max_depth;      /* MAX_DEPTH, default 5 */
    float shadow_bias;       /* SHADOW_BIAS, default 1e-4 */
    float reflection_bias;   /* REFLECTION_BIAS */
    float refraction_bias;   /* REFRACTION_BIAS */
    uint32_t use_gi;         /* USE_GI, default 0 */
    uint32_t collect_counters; /* 1: run the counting build (every ray walked the reference's way; fills crt_stats);
                                * 2: run the production kernels and tally the tests they execute (crt_get_executed_counters) */
    uint32_t gi_sample_size; /* GI_SAMPLE_SIZE, default 2 (used when use_gi) */
    uint32_t rays_per_pixel; /* RAYS_PER_PIXEL, default 1 (used when use_gi; 0 renders like 1, as in the reference) */
    float monte_carlo_bias;  /* MONTE_CARLO_BIAS, default 1e-4 */
    uint32_t gi_seed;        /* the frame's seed of the GI generator */
} crt_options;

/* A pixel rectangle = the reference's unit of work, RayTracer::renderRectangle(row, col, w, h)
 * (RayTracer.cpp:82-112); clamped to the image like the reference does (:84-85). */
typedef struct crt_rect {
    uint32_t row, col, width, height;
} crt_rect;

/* Work counters of the last counted render: properties of (scene, camera, options) under the
 * reference's traversal semantics; they price the algorithmic bytes of SURVEY.md §8d. */
typedef struct crt_stats {
    double kernel_ms;          /* device time of the last render kernel (HIP events on the render stream) */
    double total_ms;           /* kernel + device->host copy of the last crt_render */
    uint64_t box_tests;        /* BoundingBox::hasIntersection calls */
    uint64_t tri_tests;        /* Ray::intersectWithTriangle calls */
    uint64_t leaf_index_reads; /* leaf index entries read (both tree levels) */
    uint64_t shaded_hits;      /* closest hits shaded */
    uint64_t light_evals;      /* light-loop iterations */
    uint64_t texel_fetches;    /* bitmap texel reads */
    uint64_t primary_rays, secondary_rays, shadow_rays;
    uint64_t pixels;           /* pixels rendered by the last call */
    uint32_t counters_valid;   /* 1 when the last render ran with collect_counters */
    uint32_t fallback_frames;  /* frames since crt_create whose ray queues overflowed and that were redone by the
                                * queue-less kernel (same pixels, much slower): the last resort, for a frame that outgrows
                                * queues an earlier frame of its size had fitted, and for explicit crt_tuning capacities.
                                * Updated by the synchronous calls (crt_render, crt_kernel_times_ms, crt_synchronize). */
    uint64_t queue_bytes;      /* device memory of the per-frame ray queues as allocated now (they follow the frames: they
                                * grow when a frame came close to a capacity, and after an attempt that overflowed) */
    uint64_t queue_regrows;    /* attempts since crt_create that overflowed their queues and were repeated, inside the same
                                * call, with larger ones (a context's first frame of a size is probed this way) */
} crt_stats;

typedef struct crt_ctx crt_ctx;

/* Kernel selection and sizing.  Nothing here changes a pixel: every combination renders the same frame bit for
 * bit (tests/test_gpu_parity.py runs the matrix); the defaults are what bench.py measures.  The reference has no
 * counterpart (its only tunables are RenderOptions, above); the library reads NO environment variables.
 * (Round 2's 43 fields selected between ~20 kernel variants and fed a wall-clock autotuner; the variants that lost their
 * measurements were removed -- DESIGN.md section 7 keeps the numbers, git keeps the code -- their constants are now constants,
 * and the defaults below are within 1.5 % of the best setting on every BASELINE scene.) */
enum { CRT_MODE_STREAM = 0, CRT_MODE_LANES = 1 };
typedef struct crt_tuning {
    uint32_t size;            /* sizeof(crt_tuning), filled in by crt_tuning_defaults */
    uint32_t mode;            /* CRT_MODE_STREAM (ray stream, default) | CRT_MODE_LANES (full recursion per lane, queue-less) */
    uint32_t step_budget;     /* 384: steps after which a closest-hit walk goes to the wave-per-ray kernel; 0 = faithful kernels only */
    uint32_t shadow_budget;   /* 4096: cap of the same for the bulk shadow pass (the launch scales it down with its size) */
    uint32_t level0_budget;   /* 0 (= min(step_budget, what a lane gets through in the launch)): the same for PRIMARY rays */
    uint32_t heavy_level;     /* 100000: recursion levels with fewer rays skip the per-lane kernel */
    uint32_t side_blocks;     /* 3: workgroups per CU of the bulk shadow pass on the side stream; 0 = no side stream */
    uint32_t node_cap, ray_cap, shadow_cap; /* 0 = the queues follow the frames; explicit values make
                                             * queue overflow -- and the fallback -- reachable in tests */
    uint32_t bvh;             /* 1: rays are walked through the candidate filter (csrc/crt_bvh.h, csrc/kernel_bvh.h) where the scene has one;
                               * 0: by the reference-order kernels alone; 2: the filter kernels' bounds-checked build (development) */
    uint32_t level_queue;     /* 1: with the filter kernels, every recursion level below level 0 in ONE launch that feeds itself through a queue
                               * (csrc/kernel_bvh.h: bvh_trace_queue), on this many workgroups per CU -- for frames whose levels held at most
                               * 250 k rays each a frame ago (wider levels are throughput: one launch per level is as fast or faster);
                               * 0: always one launch per level.  Development bits: 8 no child ray continues in its parent's lane; 9 the launch behind
                               * level 0 instead of beside it; 12 (tests) level 0 held back until the launch beside it has given up */
    uint32_t fetch_chunk;     /* 256 | 64 << 16: work indices a wave claims from a launch's cursor with ONE atomic -- bits 0-15 the bulk shadow
                               * pass's slots, bits 16-31 level 0's primary rays (both >= 64, multiples of 64); a cursor word serves ~100
                               * atomics per microsecond however many waves ask, which bounded the pass until it claimed in chunks */
} crt_tuning;
void crt_tuning_defaults(crt_tuning *tuning);

/* replaces RayTracer::RayTracer(Scene&) (RayTracer.cpp:45-51): copies the flattened scene + tree to
 * HBM on `device` and allocates the persistent H*W colour buffer (zero-initialised like
 * colorBuffer, RayTracer.h:69). */
int crt_create(const crt_scene_desc *scene, int device, crt_ctx **out);
/* the same with explicit tuning (NULL = defaults) */
int crt_create_tuned(const crt_scene_desc *scene, int device, const crt_tuning *tuning, crt_ctx **out);

/* replaces RayTracer::setCamera() (RayTracer.cpp:57-59): position + row-major 3x3 matrix
 * (Camera.h:7-8).  The tree is not rebuilt. */
int crt_set_camera(crt_ctx *ctx, const float position[3], const float matrix[9]);

/* replaces RayTracer::render's bucket scheduling + renderRectangle (RayTracer.cpp:141-158,82-112):
 * renders the pixels covered by `rects` into the context's persistent colour buffer (pixels not
 * covered keep their previous value) and copies the whole H*W*3 float buffer (row 0 = top) to
 * `out_rgb` (host memory, may be NULL to skip the copy). */
int crt_render(crt_ctx *ctx, const crt_options *options, const crt_rect *rects, uint32_t n_rects, float *out_rgb);

/* The same without waiting for the device: the frame (and the copies to out_rgb -- float, H*W*3 -- and / or out_rgb8 --
 * quantised bytes, PPMColor rule; either may be NULL; pinned host memory keeps the copies asynchronous) is enqueued and the
 * call returns.  crt_wait finishes it and fills the statistics.  One frame per context at a time: a second crt_render_async
 * first waits for the previous frame.  Frames IN FLIGHT together need one context each (the reference's animation driver,
 * app/animation.cpp:24-38, renders frame after frame; crt::RayTracer::renderAsync alternates two contexts).
 * THE FIRST FRAME OF A SIZE / DEPTH ON A CONTEXT IS NOT ASYNCHRONOUS: its queue capacities are probed -- the call waits for the
 * attempt (hipEventSynchronize on the frame's stream) and repeats it with larger queues if it overflowed (crt_stats::queue_regrows) --
 * and so is the frame after one that overflowed.  That also holds for crt_render_tiles_device on a caller's stream: such a call
 * cannot be captured into a hipGraph; render one frame first, capture the following ones.  The library also uses two streams of its
 * own beside the caller's (the bulk shadow pass; the level queue's launch), joined to it by events before the call's last launch. */
int crt_render_async(crt_ctx *ctx, const crt_options *options, const crt_rect *rects, uint32_t n_rects, float *out_rgb,
                     uint8_t *out_rgb8);
int crt_wait(crt_ctx *ctx);
/* page-locked host memory for those outputs (NULL when it cannot be had) */
void *crt_alloc_pinned(size_t bytes);
void crt_free_pinned(void *p);

/* Device-resident variants used by the multi-GPU tile partition.
 * The frame is cut into 8x8 pixel tiles, numbered row-major; this call renders tiles
 * first, first+stride, first+2*stride, ... and writes them PACKED, tile after tile, 64 pixels x 3
 * floats each (pixel k of a tile = row k/8, column k%8), to `d_packed` (device memory of at least
 * crt_packed_tile_count(...)*192 floats).  Asynchronous on `stream` (a hipStream_t, NULL = default). */
int crt_render_tiles_device(crt_ctx *ctx, const crt_options *options, uint32_t first, uint32_t stride,
                            float *d_packed, void *stream);
/* number of tiles the call above renders */
uint32_t crt_packed_tile_count(const crt_ctx *ctx, uint32_t first, uint32_t stride);
/* Scatter `n_parts` packed buffers (part p holds tiles p, p+n_parts, ...; laid out one after another in
 * d_packed_all with `part_stride_floats` between parts) into the row-major H*W*3 device frame d_frame. */
int crt_unpack_tiles_device(crt_ctx *ctx, const float *d_packed_all, uint32_t n_parts, uint64_t part_stride_floats,
                            float *d_frame, void *stream);

/* PPMColor quantiser (Color.cpp:12-16) on the device: out[i] = (uint8)(clamp(rgb[i],0,1)*255), truncating. */
int crt_quantize_device(crt_ctx *ctx, const float *d_rgb, uint64_t n_values, uint8_t *d_out, void *stream);
/* Quantised copy of the context's persistent colour buffer to host memory (H*W*3 bytes). */
int crt_read_quantized(crt_ctx *ctx, uint8_t *out_rgb8);

/* Device time of the most recent render kernel, from the HIP events recorded around it on the stream
 * it was launched on (waits for that kernel to finish). */
int crt_kernel_elapsed_ms(crt_ctx *ctx, double *ms);
/* Device times of the most recent renders (up to 64, oldest first), from HIP events recorded around the
 * kernels on the streams they were launched on; 5 doubles per render:
 *   [0] whole render, first kernel to last
 *   [1] the recursion levels: stream_trace_shade + heavy_trace_closest + stream_shade_evicted, all levels
 *   [2] stream_trace_shadow pass 0 (level-0 shadow rays; runs on a side stream BESIDE [1])
 *   [3] stream_trace_shadow pass 1 + heavy_trace_shadow   [4] stream_resolve (+ fallback)
 * Waits for those kernels to finish. */
int crt_kernel_times_ms(crt_ctx *ctx, double *out_phase_ms, uint32_t max_count, uint32_t *count);
int crt_get_stats(crt_ctx *ctx, crt_stats *out);
/* The counters of the last counted render split by kernel, in crt_stats order: box_tests, tri_tests,
 * leaf_index_reads, shaded_hits, light_evals, texel_fetches, primary_rays, secondary_rays, shadow_rays.
 * closest = stream_trace_shade of all recursion levels, shadow = stream_trace_shadow pass 0 (the level-0 shadow
 * rays); shadow pass 1 and the resolve are only in crt_stats' totals. */
int crt_get_kernel_counters(crt_ctx *ctx, uint64_t closest[9], uint64_t shadow[9]);
int crt_synchronize(crt_ctx *ctx);
void crt_destroy(crt_ctx *ctx);
/* last error text of a context (or of the last failed crt_create when ctx == NULL) */
const char *crt_last_error(const crt_ctx *ctx);
int crt_device_count(void);

/* ---- one scene on several devices of one node, behind the same call (SURVEY.md section 8b "multi-GPU handled inside the
 * context"; the reference's counterpart is the bucket thread pool, RayTracer.cpp:141-158).  One context, host thread and
 * stream per listed device (a device may be listed more than once); the covered 8x8 tiles are dealt round-robin, every
 * device copies its packed tiles to devices[0] over xGMI (peer copy, no collective) and devices[0] scatters them into its
 * persistent colour buffer.  crt_multi_render has crt_render's contract: same pixels, bit for bit. */
typedef struct crt_multi crt_multi;
int crt_multi_create(const crt_scene_desc *scene, const int *devices, uint32_t n_devices, const crt_tuning *tuning, crt_multi **out);
int crt_multi_set_camera(crt_multi *multi, const float position[3], const float matrix[9]);
int crt_multi_render(crt_multi *multi, const crt_options *options, const crt_rect *rects, uint32_t n_rects, float *out_rgb);
int crt_multi_read_quantized(crt_multi *multi, uint8_t *out_rgb8);
/* counters summed over the devices, kernel_ms = the slowest device's, total_ms = wall time of the call */
int crt_multi_get_stats(crt_multi *multi, crt_stats *out);
uint32_t crt_multi_device_count(const crt_multi *multi);
crt_ctx *crt_multi_context(crt_multi *multi, uint32_t part); /* part 0 holds the frame */
const char *crt_multi_last_error(const crt_multi *multi);
/* Parts whose device cannot store into device[0]'s memory (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess said so at
 * crt_multi_create) copy their tiles through pinned host memory instead of over xGMI: how many there are, and what the
 * runtime answered for each (one line per part, empty when every part has peer access).  Same pixels either way. */
uint32_t crt_multi_staged_parts(const crt_multi *multi);
const char *crt_multi_peer_note(const crt_multi *multi);
int crt_debug_multi_force_staged(crt_multi *multi, int on); /* tests: stage every part, as if no device had peer access */
int crt_debug_multi_fail_next_alloc(crt_multi *multi);       /* tests: the next re-partition stops at a part's buffer with CRT_ERR_NOMEM */
void crt_multi_destroy(crt_multi *multi);

/* ---- the reference's tree built on the GPU (KDTree<T>::build, KDTree.cpp:10-46 / :89-125; BoundingBox.h:60-83): level by
 * level, one thread per (node, element) entry, the same float operations as the reference's split and overlap test, so the
 * result is the reference's tree node for node: same creation-order numbering, same boxes, same leaf lists.
 * element_boxes: 6 floats per element (min xyz, max xyz); root_box likewise.  The built tree is read back through the
 * accessors: boxes 6 floats per node, links 4 words per node (children[0], children[1], parent, number of leaf indexes; none
 * = 0xFFFFFFFF), indexes = the leaves' lists concatenated in node order (the layout of crt_host_tree_dump, crt_host.h). */
typedef struct crt_built_tree crt_built_tree;
int crt_build_tree_device(int device, const float *element_boxes, uint32_t n_elements, const float root_box[6],
                          uint32_t max_depth, uint32_t max_leaf, crt_built_tree **out);
uint32_t crt_built_tree_node_count(const crt_built_tree *tree);
uint64_t crt_built_tree_index_total(const crt_built_tree *tree);
const float *crt_built_tree_boxes(const crt_built_tree *tree);
const uint32_t *crt_built_tree_links(const crt_built_tree *tree);
const uint32_t *crt_built_tree_indexes(const crt_built_tree *tree);
void crt_built_tree_free(crt_built_tree *tree);
const char *crt_build_last_error(void);

/* After a render with collect_counters == 2 on the default (ray-stream) path: out = {box tests, triangle tests} the
 * production kernels executed in the whole render, then the same two for shadow pass 0 alone (the largest kernel).  Fewer than crt_stats' box_tests / tri_tests, which are the reference's: the kernels leave
 * out work that cannot change the result (shadow early exit, one walk per mesh and ray). */
int crt_get_executed_counters(crt_ctx *ctx, uint64_t out[4]);
/* ... and the box tests of the plan loops (csrc/kernel_plan.h: a ray against the top-level leaves, whose boxes sit in scalar
 * registers loaded once per wave -- executed per ray, but with no per-ray fetch): {whole render, of which shadow pass 0}.
 * They are NOT part of the counts above. */
int crt_get_executed_plan_tests(crt_ctx *ctx, uint64_t out[2]);

/* ---- Test hooks: exported by libcrt_hip_test.so only (the product's objects + csrc/crt_testhooks.hip; libcrt_hip.so has none of them):
 * crt_test_pow5, crt_test_gi, crt_bvh_selftest, crt_debug_multi_force_staged, crt_debug_multi_fail_next_alloc. ---- */
/* Test hook: out[i] = the device build of the restated glibc powf(x[i], 5) (the Fresnel term, RayTracer.cpp:407). */
int crt_test_pow5(int device, const float *x, float *out, uint64_t n);
/* Test hook for the GI mode's arithmetic (csrc/glibc_sincosf.h, csrc/gi_random.h), evaluated on `device`, or by the host
 * build of the same headers when device < 0.  what = 0: out[i] = bits of sinf(a[i] as float); 1: cosf; 2: bits of the
 * uniform number u(key a[i], draw b[i]); 3: mix(a[i], b[i]).  b may be NULL for 0 and 1. */
int crt_test_gi(int device, uint32_t what, const uint32_t *a, const uint32_t *b, uint32_t *out, uint64_t n);

/* Test hook, HOST ONLY (no device is touched): builds the candidate filter of `scene` (csrc/crt_bvh.h) and checks its two promises
 * against brute force for n_rays rays (6 floats each: origin, direction): out = {rays, triangles the reference's test accepts with a
 * finite distance, of which the conservative walk does not reach (must be 0), triangles it accepts with an infinite or NaN distance in
 * a leaf the ray's line passes, of which the miss check does not reach (must be 0), nodes visited by the two walks, structural
 * errors (must be 0)}.  CRT_ERR_INVALID when the scene has no filter. */
int crt_bvh_selftest(const crt_scene_desc *scene, const float *rays, uint32_t n_rays, int primary, uint64_t out[8]);

/* Diagnostics for the development tools under tools/ (no counterpart in the reference; not needed to render):
 * the ray-stream pass's queue counters of the last frame (rays per recursion level, walks handed to the
 * wave-per-ray kernels, ...: the SC_* layout of csrc/kernel_stream.h, at most 512 words). */
int crt_debug_stream_counts(crt_ctx *ctx, uint32_t *out_words, uint32_t max_words);
/* which kernels a production frame of this context runs, e.g. "level0=stream_trace_shade_plan<true>;shadow0=stream_trace_shadow_plan<0u>;
 * levels=heavy_trace_closest<5>" (names as rocprofv3 prints them) */
int crt_describe_kernels(const crt_ctx *ctx, char *out, size_t size);

#ifdef __cplusplus
}
#endif
#endif
