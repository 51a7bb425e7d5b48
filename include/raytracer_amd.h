/*
 * raytracer_amd.h -- C ABI of the MI355X ray-intersection engine (libraytracer_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of PaulThythy/raytracer: ray generation,
 * acceleration-structure traversal and ray/primitive closest-hit + any-hit.  The reference has no
 * FFI; the path is reached through four seams inside VkRenderer.  Every entry point
 * below names the reference seam it replaces (paths relative to /root/reference).
 *
 * Conventions (mirroring the reference: single caller thread, blocking calls, owner keeps its arrays):
 *   - return 0 on success, a negative RT_E_* code on failure; nothing is thrown across the ABI (every entry point
 *     catches: std::bad_alloc and thread-creation failures become RT_E_NOMEM, anything else RT_E_INVALID).
 *     raytracer_amd/host/raytracer.hpp converts non-zero into std::runtime_error(rt_last_error()),
 *     the reference's own convention (src/main.cpp:8-14).
 *   - one caller thread per rt_ctx.  host_* buffers stay owned by the caller.
 *   - *_dev entry points take DEVICE pointers valid on the context's device `slot` and enqueue on the
 *     given hipStream_t, passed as void* (NULL = HIP's default stream); they do not synchronise.
 *   - all arithmetic is fp32; hit records are bit-exact against oracle/ (see DESIGN.md).
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;   /* opaque; owns all device memory */

/* shaders/frag.glsl:17-20 `struct Ray` plus an explicit interval (reference: tmin 0, tmax 1e20). */
typedef struct { float ox, oy, oz, tmin, dx, dy, dz, tmax; } rt_ray;          /* 32 B */
/* Result of shaders/frag.glsl:265-301 traceRay.  prim = index into the uploaded triangle array;
 * bit 31 set = index into the sphere array; 0xFFFFFFFF = miss (then t = ray.tmax, u = v = 0). */
typedef struct { float t, u, v; uint32_t prim; } rt_hit;                     /* 16 B */

#define RT_MISS        0xFFFFFFFFu
#define RT_SPHERE_BIT  0x80000000u

enum {
    RT_OK = 0,
    RT_E_INVALID = -1,      /* bad argument (null pointer, wrong stride, zero size, ...) */
    RT_E_DEVICE = -2,       /* HIP runtime error; text in rt_last_error */
    RT_E_NO_SCENE = -3,     /* trace before rt_scene_upload */
    RT_E_NO_CAMERA = -4,    /* primary rays before rt_set_camera */
    RT_E_NOMEM = -5,        /* host allocation (or thread creation) failed; the context stays usable */
    RT_E_TOO_LARGE = -6
};

/* rt_set_option keys */
enum {
    RT_OPT_TRAVERSAL = 1,   /* 0 auto (default), 1 brute force (the reference's loop), 2 BVH2 nodes, 3 BVH4q nodes
                               (the node layout is chosen at rt_scene_upload: set this first) */
    RT_OPT_BVH_THREADS = 2, /* host threads for the BVH build; 0 = hardware concurrency capped at 16 */
    RT_OPT_COUNTERS = 3,    /* 1: *_dev closest-hit launches run the instrumented kernel and accumulate rt_counters */
    RT_OPT_PAD_LOG2 = 4,    /* BVH boxes are padded by max|vertex coordinate| * 2^-value at upload, and every ray widens them by
                               max|origin coordinate| * 2^-value on its own (default 15, range 4..20: below 2^-20 the
                               conservative-traversal argument of DESIGN.md does not hold) */
    RT_OPT_MAX_LEAF = 5,    /* triangles per BVH leaf, 1..8 (default 4); takes effect at upload */
    RT_OPT_STRIP = 6,       /* ray-buffer kernel: rays per chunk a persistent wave draws (default 256); 0 = one ray per lane, no refill */
    RT_OPT_REFILL = 7,      /* ray-buffer kernel: low byte = idle lanes that trigger a refill (default 16),
                               bits 8.. = lanes holding a parked leaf before the triangle block runs (default 12) */
    RT_OPT_RAY_BINNING = 8, /* ray-buffer kernel: hand rays out grouped by origin cell (16^3 Morton grid over the scene,
                               dead rays last): 0 off, 1 on, 2 auto = when nodes + triangles exceed 48 MB and the buffer
                               holds >= 2^18 rays (default).  Output order and values are unaffected.  Uses per-slot
                               scratch (4 B per ray): calls on one slot that bin must be stream-ordered. */
    RT_OPT_LDS_STACK = 9,   /* ray-buffer kernel, BVH4q nodes: traversal-stack entries per lane kept in LDS (default 20, 2..64);
                               deeper entries spill to a per-slot area in HBM, which lets 8 waves per SIMD run instead of
                               3-4.  0 = size LDS for the tree's worst case, no spill.  Results are unaffected.  Left at 20 the
                               library picks per launch (12 beside the ray queue on structures the L2s serve, 32 for binned
                               launches below 2^25 rays); any other value holds for every launch. */
    RT_OPT_PRIMARY = 10,    /* primary rays.  1 = tile kernel, one ray per lane with its own traversal stack (generation fused
                               with traversal and, in rt_primary_shade_stripes_dev, with the first shading segment);
                               3 = the same kernel with PACKET traversal: the 64 rays of a wave (a 4x4 .. 8x8 pixel tile)
                               walk the tree together, node by node, from an fp32 copy of the BVH4q nodes fetched with
                               scalar loads (128 B per node more in HBM; set before rt_scene_upload; trees of 2^25 nodes
                               and more -- over 100 M triangles -- go without the copy and take 1 or 2 by the rule below);
                               2 = ray generation into a buffer + the ray-buffer kernel (+ the first-segment shade);
                               0 auto (default) = 3 for BVH4q trees (ahead on every scene measured, tessellated or soup);
                               for BVH2 trees 2 when the tree's SAH cost (rt_scene_stats.sah_cost) exceeds 250, i.e. for
                               soup-like scenes whose neighbouring rays walk unrelated paths, else 1.
                               Same rays, hits and path state in every case. */
    RT_OPT_SAH_BINS = 12,   /* BVH builder: SAH bins per axis, 4..32 (default 16); takes effect at upload.  Results are unaffected. */
    RT_OPT_SAH_TRAV = 13,   /* BVH builder: SAH cost of one interior step in hundredths of a triangle test (default 60:
                               the BVH4q step costs about as much as a triangle test, and half the children of a node are culled) */
    RT_OPT_RAY_PACKETS = 15,/* rt_trace_dev / rt_trace (closest hit) and rt_shadow_dev, BVH4q trees: 1 = every 64 consecutive rays of
                               the buffer walk the tree together as one packet, like the primary rays of the tile kernel.  Same hit records for any
                               buffer; faster only when consecutive rays really travel together (same region, similar
                               direction), slower -- possibly much -- when they do not.  Default 0. */
    RT_OPT_FUSED_PATHS = 18, /* rt_render on scenes traced by the brute-force loop (<= 32 triangles, e.g. the reference's Cornell box):
                               1 (default) = the whole path loop of a sample in one kernel, ray / throughput / radiance in registers;
                               0 = the staged pipeline (trace and shade kernels per segment).  Same frame, bit for bit. */
    RT_OPT_FRAME_EXCHANGE = 16, /* contexts with several slots, rt_render / rt_trace_primary: how the slots' interleaved 8-row stripes
                               reach slot 0's device, where they are put in image order and leave in ONE device-to-host copy.
                               1 = RCCL: grouped ncclSend / ncclRecv, every slot over its own xGMI link to slot 0 (librccl.so.1 is
                               loaded on first use; needs distinct device ordinals -- RCCL refuses a communicator with a GPU twice);
                               2 = hipMemcpyPeerAsync (plain device copies when the ordinals repeat);
                               0 auto (default) = 1 when all ordinals differ and RCCL loads, else 2.  Same frame in every case. */
    RT_OPT_RAY_QUEUE = 19,  /* ray-buffer kernel, BVH4q nodes: 1 = every wave keeps a queue of 32 prepared rays (origin, direction, interval,
                               slab constants, index) in LDS, filled by half the wave at once; idle lanes start their next ray from it
                               (refills from 6 idle lanes on instead of 16, unless RT_OPT_REFILL says otherwise).  0 = idle lanes
                               load and prepare their next ray themselves.  2 auto (default) = 1 for launches whose rays are not
                               binned (structures the L2s serve) and for binned launches below 2^25 rays, else 0.
                               Results are unaffected. */
    RT_OPT_RAY_PAD = 11     /* 1 (default): the per-ray box pad described at RT_OPT_PAD_LOG2.  0 switches it off -- for
                               measuring what it protects against only: rays that start hundreds of scene extents away
                               then lose hits the reference loop finds (tests/test_gpu_parity.py, far-origin test). */
};

/* Scene / acceleration-structure statistics after rt_scene_upload. */
typedef struct {
    uint32_t n_tris, n_spheres, n_lights;
    uint32_t n_nodes;          /* interior nodes of the flattened BVH */
    uint32_t n_leaves;
    uint32_t max_depth;
    uint32_t node_bytes;       /* bytes per node in HBM */
    uint32_t tri_bytes;        /* bytes per intersection triangle in HBM */
    float    sah_cost;         /* SAH cost of the binary tree: 1.2 per interior step + 1 per triangle test, area-weighted (fixed weights,
                                  independent of RT_OPT_SAH_TRAV) */
    float    build_ms;         /* host BVH build + flatten */
    float    upload_ms;        /* H2D copies */
    uint64_t device_bytes;     /* per replica */
    uint32_t stack_entries;    /* traversal-stack entries per ray the flattened tree can need, sentinel included: the tile
                                  kernel's dynamic LDS is stack_entries * 1 KiB per 256-thread block, the ray-buffer kernel's
                                  min(stack_entries, RT_OPT_LDS_STACK) * 1 KiB */
    uint32_t reserved;
} rt_scene_stats;

/* Work counters of the instrumented kernel (B, N, T per ray). */
typedef struct {
    uint64_t rays;
    uint64_t box_tests;        /* B: child-box slab tests */
    uint64_t nodes;            /* N: interior nodes fetched */
    uint64_t tri_tests;        /* T: triangle tests */
    uint64_t sphere_tests;
    uint64_t wave_max_nodes;   /* sum over waves of the per-wave maximum of `nodes` (divergence probe) */
    uint64_t wave_max_tris;    /* same for triangle tests */
    uint64_t waves;
    /* scheduling probes of the ray-buffer kernel (summed over scheduling rounds of all waves; for that kernel `waves` is
       the number of rounds, wave_max_nodes / wave_max_tris the lanes that took an interior step / a triangle test) */
    uint64_t idle_lane_rounds;     /* lanes without a ray, waiting for the next refill */
    uint64_t blocked_lane_rounds;  /* lanes that hold a parked leaf and stand on a second one: cannot walk on */
    uint64_t drained_lane_rounds;  /* lanes with an empty stack waiting for the triangle block */
    uint64_t tri_rounds;           /* rounds in which the triangle block ran */
} rt_counters;

/* ---- life cycle --------------------------------------------------------------------------------
 * Replaces VkRenderer::initVulkan's device pick + cleanupVulkan for the compute path
 * (src/vulkan/VkRenderer.cpp:15-38, :40-122).  device_ids[i] are HIP device ordinals; each gets a
 * replica of the scene ("slot" i).  n_devices >= 1; the same ordinal may appear more than once. */
int  rt_create(rt_ctx** out, const int* device_ids, int n_devices);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);   /* never NULL; "" when the last call succeeded */
int  rt_set_option(rt_ctx* ctx, int key, int64_t value);
int  rt_num_slots(const rt_ctx* ctx);
/* How the last frame of a context with several slots (rt_render / rt_trace_primary) was gathered on slot 0:
 * 1 = RCCL send / recv, 2 = peer copies, 0 = no such frame yet (RT_OPT_FRAME_EXCHANGE).  No counterpart in the reference. */
int  rt_frame_exchange_used(const rt_ctx* ctx);

/* ---- scene upload (once) -----------------------------------------------------------------------
 * Replaces VkRenderer::createData's three SSBO uploads (src/vulkan/VkRenderer.cpp:260-326).  Takes
 * the reference's host structs as they are: Triangle 144 B (src/math/Triangle.h:10-15), Sphere 64 B
 * (src/math/Sphere.h:10-15), Light 32 B (src/math/Light.h:6-10); strides must be >= those sizes and
 * multiples of 4.  Counts are explicit: 0 means 0 (the reference uploads one zeroed element for an
 * empty vector, :262-270; a zero triangle/sphere can never be hit, so results agree).
 * Builds the BVH on the host, flattens it to the 16-B-aligned SoA layout of DESIGN.md and copies it
 * to every slot.  The caller keeps its arrays. */
int  rt_scene_upload(rt_ctx* ctx,
                     const void* tris,    uint32_t n_tris,    uint32_t tri_stride,
                     const void* spheres, uint32_t n_spheres, uint32_t sphere_stride,
                     const void* lights,  uint32_t n_lights,  uint32_t light_stride);
int  rt_scene_stats_get(const rt_ctx* ctx, rt_scene_stats* out);
/* The host half of rt_scene_upload on its own (no device needed): BVH build + flatten.  layout 2 = BVH2 nodes,
 * 3 = BVH4q nodes (16 words per node, DESIGN.md "Data layout").  nodes_out / tri_order_out may be NULL (then only
 * the statistics are produced); stats->stack_entries carries the traversal-stack bound. */
int  rt_bvh_build_host(const void* tris, uint32_t n_tris, uint32_t tri_stride, int layout, int max_leaf, int pad_log2,
                       int threads, rt_scene_stats* stats, uint32_t* nodes_out, uint64_t nodes_cap_words,
                       uint32_t* tri_order_out);

/* ---- camera (per frame) ------------------------------------------------------------------------
 * Replaces VkRenderer::updateUniformBuffer (src/vulkan/VkRenderer.cpp:855-862): 112-byte
 * Camera::UniformBufferObject (src/application/Camera.h:6-18). */
int  rt_set_camera(rt_ctx* ctx, const void* camera_ubo112);

/* Optional: pinned (page-locked) host memory for the arrays a renderer keeps for its lifetime -- rays in, hits out --
 * the way VkRenderer keeps its host-visible buffers mapped (src/vulkan/VkRenderer.cpp:260-326).  The host-pointer entry
 * points below DMA from / to such blocks directly; ordinary (pageable) memory is staged through pinned buffers by a few
 * host threads instead (slower, same bits).  Blocks live until rt_host_free or rt_destroy. */
int  rt_host_alloc(rt_ctx* ctx, size_t bytes, void** out);
int  rt_host_free(rt_ctx* ctx, void* block);

/* ---- launch (per frame) ------------------------------------------------------------------------
 * Replace the vkCmdDrawIndexed that runs frag.glsl:main once per pixel
 * (src/vulkan/VkRenderer.cpp:1123), restricted to the intersection part of main().
 * rt_trace_primary: getCameraRay with jitter off (frag.glsl:89-128) + traceRay (:265-301) for every
 * pixel; uv = ((x+.5)/w, (y+.5)/h), row 0 = top (shaders/vert.glsl:9-12).  host_out[y*w+x]. */
int  rt_trace_primary(rt_ctx* ctx, uint32_t w, uint32_t h, rt_hit* host_out);
/* closest hit for caller-supplied rays (frag.glsl:265-301).  Rays are split in contiguous ranges
 * over the slots.  (rt_trace_primary and rt_render split the image rows instead: 8-row stripes dealt round-robin to the
 * slots -- cost is spatially uneven, SURVEY.md 8e -- gathered on slot 0's device, RT_OPT_FRAME_EXCHANGE.) */
int  rt_trace(rt_ctx* ctx, const rt_ray* host_rays, uint64_t n, rt_hit* host_out);
/* any-hit: host_out[i] = 1 iff some primitive is hit with tmin <= t < tmax.  The reference has no
 * any-hit routine; its shadow query (frag.glsl:325-330) is rt_shadow_dev. */
int  rt_occluded(rt_ctx* ctx, const rt_ray* host_rays, uint64_t n, uint8_t* host_out);

/* ---- device-resident pipeline stages (inputs/outputs stay in HBM) ------------------------------ */
/* Primary rays for the pixel rows [y0, y0+rows) of a w x h image, spp samples per pixel, and their
 * closest hits.  jitter != 0 applies frag.glsl:94-103 with the seeded hash of DESIGN.md instead of
 * frag.glsl:13-15.  Ray index r = ((y-y0)*w + x)*spp + s; RNG key uses the full-image index.
 * d_rays_out may be NULL when the rays themselves are not needed. */
int  rt_trace_primary_dev(rt_ctx* ctx, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                          uint32_t spp, uint32_t seed, int jitter,
                          rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream);
/* Same, for an interleaved share of the image: local row yl maps to image row
 * y0 + (yl / stripe_h) * stripe_h * stripe_step + yl % stripe_h (rows beyond h are skipped), so rank r
 * of N takes y0 = r*stripe_h, stripe_step = N.  stripe_h should be a multiple of 8 (tile height).
 * Output index ((yl*w + x)*spp + s) is compact in the local rows. */
int  rt_trace_primary_stripes_dev(rt_ctx* ctx, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                  uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t seed, int jitter,
                                  rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream);
int  rt_trace_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits_out, void* stream);
int  rt_occluded_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, uint64_t n, uint8_t* d_out, void* stream);
/* rt_trace_dev that allocates nothing.  No counterpart in the reference: it is for callers that cannot call hipMalloc
 * -- code under HIP graph capture, tight serving loops, harnesses that forbid it.  Same rays in, same hit records out
 * (bit for bit) as rt_trace_dev, but the kernel's scratch (chunk cursors, spill columns of the traversal stack, the
 * ray-binning permutation) is a static device array inside the library (about 36 MiB per GPU, held until the process
 * ends) instead of per-context allocations, and the call performs no hipMalloc / hipFree.  n <= RT_NOALLOC_MAX_RAYS,
 * else RT_E_TOO_LARGE (split the buffer).  Every call on one GPU shares that array: calls that overlap on one GPU --
 * different streams, contexts or threads -- race; keep them stream-ordered.  RT_OPT_COUNTERS is ignored. */
#define RT_NOALLOC_MAX_RAYS (1u << 22)
int  rt_trace_dev_noalloc(rt_ctx* ctx, int slot, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits_out, void* stream);
/* frag.glsl:321-330 for one light: from (rays, hits) build the shadow ray at p + N*0.001 towards
 * light_pos, trace closest-hit, d_lit[i] = 1 lit / 0 shadowed / 2 primary miss.
 * d_shadow_rays_out may be NULL. */
int  rt_shadow_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                   const float light_pos[3], rt_ray* d_shadow_rays_out, uint8_t* d_lit, void* stream);
/* frag.glsl:185-203 + :340-343: cosine-weighted bounce rays from hits (key0 = RNG key of element 0). */
int  rt_bounce_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                   uint32_t key0, uint32_t seed, rt_ray* d_rays_out, void* stream);

/* ---- shading + path loop (frag.glsl:205-263, :303-360) ---------------------------- */
/* throughput = (1,1,1), radiance = 0 for n paths (frag.glsl:304, :308); 4 floats per path each. */
int  rt_path_init_dev(rt_ctx* ctx, int slot, uint64_t n, float* d_throughput4, float* d_radiance4, void* stream);
/* One iteration of main()'s bounce loop (frag.glsl:310-354) for n path segments: emission (:313), every
 * light's shadow test + Cook-Torrance term (:319-338), the next cosine-weighted ray (:340-343) and the
 * throughput update (:345-350).  A miss ends the path: the output ray is dead (tmax 0).
 * RNG key of element i = key0 + i.  d_rays_out may be NULL for the last segment of a path: then only the
 * radiance is updated (no next ray, no throughput update). */
int  rt_shade_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                  uint32_t seed, float* d_throughput4, float* d_radiance4, rt_ray* d_rays_out, void* stream);
/* Same for an interleaved share of a frame (see rt_trace_primary_stripes_dev): local element i lies in local row
 * yl = i / row_len (row_len = w*spp); its RNG key is key0 + (image row of yl relative to y0) * row_len + i % row_len,
 * i.e. with key0 = y0*row_len the index of the path in the full frame -- a striped render draws the same numbers. */
int  rt_shade_stripes_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                          uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step, uint32_t seed,
                          float* d_throughput4, float* d_radiance4, rt_ray* d_rays_out, void* stream);

/* Segment 0 of the path loop in ONE launch: rt_trace_primary_stripes_dev + rt_path_init_dev + rt_shade_stripes_dev
 * (frag.glsl:89-128, :265-301, :306-354 with throughput = 1, radiance = 0) without the primary rays and hits going
 * through HBM.  Writes d_throughput4 / d_radiance4 for every local ray slot and, unless NULL, the bounce rays to
 * d_next_out (dead rays where the path ended); d_rays_out and d_hits_out are optional copies of the primary rays
 * and hits.  Bit-identical to the three separate calls with `seed` (primary jitter) and `shade_seed`. */
int  rt_primary_shade_stripes_dev(rt_ctx* ctx, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                  uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t seed, int jitter,
                                  uint32_t shade_seed, rt_ray* d_rays_out, rt_hit* d_hits_out, float* d_throughput4,
                                  float* d_radiance4, rt_ray* d_next_out, void* stream);
/* frag.glsl:357-359: per pixel, sum of its spp radiance records / spp, then pow(c, 1/2.6) -> RGBA8 (a = 255).
 * d_linear4 (4 floats per pixel, pre-gamma) and d_rgba8 may each be NULL, not both. */
int  rt_resolve_dev(rt_ctx* ctx, int slot, const float* d_radiance4, uint64_t n_pixels, uint32_t spp,
                    float* d_linear4, uint8_t* d_rgba8, void* stream);

/* Last segment of the path loop + resolve in one launch: rt_shade_stripes_dev(..., d_out = NULL) followed by
 * rt_resolve_dev (frag.glsl:310-338, :357-359) without the final radiance going through HBM.  n = pixels * spp rays in
 * (pixel, sample) order.  d_radiance4 is read but NOT updated (with spp not a power of two <= 64 the two separate
 * kernels run instead and it is).  Same bits as the two calls.
 * Precondition: the hit record of a dead ray (tmax <= tmin) is a miss (prim = RT_MISS) -- what every trace entry point of
 * this library writes for such a ray.  Without lights the kernel does not read the rays at all (48 instead of 80 bytes per
 * path) and decides from the hit record alone; a dead ray paired with a stale non-miss record would add that record's
 * emission. */
int  rt_shade_resolve_dev(rt_ctx* ctx, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t spp,
                          float* d_throughput4, float* d_radiance4, float* d_linear4, uint8_t* d_rgba8, void* stream);
/* The whole of frag.glsl:main for a w x h frame: spp jittered samples per pixel (the reference: 10), `bounces`
 * path segments (the reference: 8), lights as uploaded.  host_rgba8[(y*w+x)*4 ..], row 0 = top.  The RNG is the
 * seeded hash of DESIGN.md (frag.glsl:13-15 is parity-unpinned), sample s of pixel p uses key (p*spp+s). */
int  rt_render(rt_ctx* ctx, uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces, uint32_t seed, uint8_t* host_rgba8);

int  rt_counters_get(rt_ctx* ctx, int slot, rt_counters* out, int reset);
int  rt_sync(rt_ctx* ctx, int slot);

/* ---- scene generators (host) -------------------------------------------------------------------
 * The reference compiles its scene into VkRenderer::createData (src/vulkan/VkRenderer.cpp:164-258)
 * and has one tessellator, Sphere::sphereGeometry (src/math/Sphere.h:17-79).  These write arrays of
 * the reference's 144-B Triangle / 64-B Sphere structs.  Each returns the number of elements the
 * scene has; it writes at most `cap` of them (call with cap = 0 to size). */
uint32_t rt_gen_cornell_triangles(void* tris144, uint32_t cap);               /* 12, VkRenderer.cpp:171-238 */
uint32_t rt_gen_cornell_spheres(void* spheres64, uint32_t cap);               /* 3,  VkRenderer.cpp:240-250 */
uint32_t rt_gen_sphere_geometry(const float center[3], float radius, uint32_t stacks, uint32_t slices,
                                const float material12[12], void* tris144, uint32_t cap); /* Sphere.h:17-79; 0 = count
                                                                    exceeds 2^32-1 or out of host memory */
uint32_t rt_gen_soup(uint64_t seed, uint32_t n, float half_extent, void* tris144, uint32_t cap); /* SURVEY C4 */
void     rt_gen_random_rays(uint64_t seed, uint64_t first, uint64_t n, rt_ray* out);              /* SURVEY C4 */
/* Default camera of VkRenderer::VkRenderer (src/vulkan/VkRenderer.cpp:3-12) with the basis of
 * Camera::updateCameraVectors (src/application/Camera.h:36-41). */
void     rt_camera_look_at(const float pos[3], const float look_at[3], float fov_deg, float aspect,
                           void* camera_ubo112);
/* the camera mutators of src/application/Camera.h:56-90 applied to a 112-byte UBO in place -- what the
 * reference's GLFW callbacks (src/application/Application.h:53-174) do between frames; pass the result to rt_set_camera.
 * rt_camera_move = moveForward / moveRight / moveUp (position and look-at point translate together along the basis
 * vector), rt_camera_rotate = rotateAroundUp (about worldUp) / rotateAroundRight, angles in degrees; zero arguments
 * are skipped.  The C++ class itself (same member names as the reference's Camera) is raytracer_amd/host/rt_camera.hpp. */
void     rt_camera_move(void* camera_ubo112, float forward, float right, float up);
void     rt_camera_rotate(void* camera_ubo112, float around_up_deg, float around_right_deg);

/* minimal Wavefront OBJ reader -> the reference's 144-B Triangle records (src/math/Triangle.h),
 * ready for rt_scene_upload.  The reference has no loader (its geometry is the literal scene of
 * src/vulkan/VkRenderer.cpp:164-258 and Sphere::sphereGeometry); BASELINE's configs name real meshes.
 * Reads `v`, `vn`, `f` (polygons as fans; `a`, `a/b`, `a//c`, `a/b/c`; negative indices); faces without normals get
 * the geometric normal; every triangle gets `material48` (the reference's 48-B Material; NULL = grey, roughness 0.5).
 * Writes at most `cap` triangles, always reports the file's triangle count (call with tris144 = NULL to size).
 * RT_E_INVALID: unreadable file, malformed record or index out of range. */
int      rt_load_obj(const char* path, const void* material48, void* tris144, uint64_t cap, uint64_t* n_tris_out);

#ifdef __cplusplus
}
#endif
#endif
