// kernels.hpp -- launch interface between the C ABI (rt_api.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include "../../include/raytracer_amd.h"

namespace rt {

// Device view of one scene replica (all pointers 16-B aligned, see DESIGN.md "Data layout").
struct DevScene {
    const float4* nodes;     // 4 x float4 per interior node (BVH2 or BVH4q, see bvh.hpp)
    const float4* tris;      // 3 x float4 per triangle in leaf order: (v0, bits id) (e1, 0) (e2, 0)
    const float4* normals;   // 3 x float4 per triangle in ORIGINAL order: n0, n1, n2
    const float4* spheres;   // (centre, radius)
    const float4* materials; // 3 x float4 per distinct material: (albedo, strength) (emission, roughness) (metallic,0,0,0)
    const uint32_t* sphere_mat;  // material id per sphere; triangles keep theirs in normals[3*i].w
    const float4* lights;    // 2 x float4 per light: (position, intensity) (colour, 0)
    const uint32_t* wide;    // BVH4q only: 32 words per node with fp32 planes (bvh.hpp), null = no packet traversal
    uint32_t n_nodes, n_tris, n_spheres, stack_depth, n_lights;
    float ray_pad;           // per-ray box pad = ray_pad * max|origin coordinate| (kernels.hip make_slab); 2^-pad_log2, 0 = off
};

// Camera constants precomputed on the host by rt_set_camera (frag.glsl:106-110: tan of the half fov
// is evaluated with the host libm so that kernel and oracle see the same bits).
struct DevCamera {
    float px, py, pz, half_w;
    float rx, ry, rz, half_h;
    float ux, uy, uz, pad0;
    float fx, fy, fz, pad1;
};

enum TraceMode { kNone = 0 /* launch_primary only: generate the rays, trace nothing */, kBrute = 1, kBvh2 = 2, kBvh4q = 3 };

// Ray binning (order in which the strip kernel hands rays out; results do not depend on it): origins are
// quantised to a 16^3 grid over the scene bounds, cells in Morton order, dead rays (tmax <= tmin) last.
constexpr uint32_t kBinCells = 1u << 12;
constexpr uint32_t kBins = kBinCells + 1;
constexpr uint32_t kBinBlocksMax = 1024;
struct BinGrid { float lox, loy, loz, sx, sy, sz; };        // cell = clamp((o - lo) * s, 0, 15)
// scratch words (uint32) that launch_bin needs for n rays: the permutation + per-block histograms
// two_level (launch_bin): + the first level's index array (n words) and two arrays of 16-bit bins (n/2 words each, rounded up)
constexpr uint32_t kBinLoBits = 6;                                   // second-level digit; the first level sorts by bin >> 6
constexpr uint32_t kBinsHi = ((kBins - 1) >> kBinLoBits) + 1;       // 65
constexpr uint64_t kBinTwoLevelMin = 1ull << 22;                     // buffers from this size on take the two-level sort
inline uint64_t bin_scratch_words(uint64_t n, uint32_t max_blocks = kBinBlocksMax, bool two_level = false) {
    return n + (uint64_t)max_blocks * kBins + kBins + 4 + (two_level ? n + 2 * ((n + 1) / 2) + 16 : 0);      // + alignment slack
}
// word 0 behind the bin totals: 1 if the permutation is to be used (see launch_bin), 0 if the input was coherent already
inline const uint32_t* bin_flag(const uint32_t* d_scratch, uint64_t n, uint32_t max_blocks = kBinBlocksMax) { return d_scratch + n + (uint64_t)max_blocks * kBins + kBins; }
// d_scratch[0 .. n) receives the permutation (ray indices grouped by bin); n must be < 2^32
// min_run: the rays are only permuted when a (64 Ki-ray block, cell) pair holds fewer than min_run rays on average,
// i.e. when the buffer is not grouped by origin already (0 = always permute).  The decision is taken on the device.
// max_blocks: histogram rows the scratch was sized for (bin_scratch_words with the same value)
// two_level: sort by the upper bits of the bin first (large buffers: see k_bin2_count); needs bin_scratch_words(n, max_blocks, true)
hipError_t launch_bin(const rt_ray* d_rays, uint64_t n, const BinGrid& g, uint32_t* d_scratch, uint32_t min_run, hipStream_t st,
                      uint32_t max_blocks = kBinBlocksMax, bool two_level = false);

// Scratch of one strip-kernel launch (the caller owns it and keeps launches that share it ordered):
//   work       kWorkWords u64: 8 chunk cursors + the mask of exhausted eighths, kCursorStride apart (zeroed by launch_trace)
//   spill      BVH4q: spill_words() ints, enables the capped LDS stack (null = whole stack in LDS)
//   lds_cap    stack entries per lane kept in LDS when spill is given
//   max_blocks bound on the persistent grid (0 = none): the spill area holds one column per thread of the grid
//   refill_perm  (leaf-vote threshold << 8 | refill threshold) for large buffers that are handed out permuted; 0 = as the others
//   stream_nt  ray loads and hit stores go past the caches (structures that do not fit the L2s)
//   queue      BVH4q: staged ray queues in LDS (k_trace_strip<QUEUE>): refills read prepared rays instead of loading and preparing them
struct StripScratch { unsigned long long* work = nullptr; int* spill = nullptr; int lds_cap = 0; uint32_t max_blocks = 0; int refill_perm = 0; bool queue = false; bool stream_nt = false; };
// closest hit for a buffer whose consecutive rays (after d_perm, if given) are coherent: 64 rays per wave walk as one packet
hipError_t launch_trace_packets(const DevScene& s, bool count, const rt_ray* d_rays, uint64_t n, const uint32_t* d_perm,
                                rt_hit* d_hits, unsigned long long* d_counters, hipStream_t st);
hipError_t launch_trace(const DevScene& s, int mode, bool any_hit, bool count, const rt_ray* d_rays, uint64_t n,
                        rt_hit* d_hits, uint8_t* d_occ, unsigned long long* d_counters, hipStream_t st,
                        uint32_t strip /* rays per chunk, 0 = one ray per lane */, int refill /* idle lanes that trigger a refill */,
                        const uint32_t* d_perm /* order in which rays are handed out (strip kernel only) or null */,
                        const uint32_t* d_perm_on /* device flag: 0 = ignore d_perm */, const StripScratch& sc);

// Allocation-free closest hit (rt_trace_dev_noalloc): the strip kernel with every scratch buffer in a static device
// array of noalloc_bytes() bytes per GPU.  noalloc_init returns that array's address on the CURRENT device.
constexpr uint64_t kNoAllocMaxRays = 1ull << 22;      // rays per call (the binning permutation is part of the array)
constexpr uint32_t kNoAllocSpillEntries = 8;          // stack entries per thread that may spill; deeper trees use more LDS
constexpr uint32_t kNoAllocMaxBlocks = 2048;          // persistent grid bound (8 blocks x 256 CUs)
constexpr uint32_t kNoAllocBinBlocks = 256;           // histogram rows of the binning pass
size_t noalloc_bytes();
hipError_t noalloc_init(void** d_scratch_out);
hipError_t launch_trace_noalloc(void* d_scratch, const DevScene& s, int mode, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits,
                                hipStream_t st, uint32_t strip, int refill, int lds_cap, bool bin, const BinGrid& grid, uint32_t min_run,
                                bool queue);

// Capped LDS stack of the strip kernel (kernels.hip "Capped stack"): entries beyond lds_cap spill to global memory,
// one column per thread of the persistent grid (at most 8 blocks of kThreads per CU).
constexpr int kThreads = 256;
inline uint64_t spill_words(uint32_t stack_depth, int lds_cap, int cus) {
    return stack_depth + 2 > (uint32_t)lds_cap ? (uint64_t)(stack_depth + 2 - lds_cap) * 8u * (uint64_t)cus * kThreads : 0;
}
int device_cus();          // multiprocessors of the current device

// With `shade` the first segment of the path loop runs in the same kernel (throughput 1, radiance 0 going in):
// d_throughput / d_radiance are written for every local ray slot, d_next (may be null) receives the bounce rays,
// and d_rays_out / d_hits become optional.  Counters are not collected in that variant.
struct PrimaryShade { float4* d_throughput; float4* d_radiance; rt_ray* d_next; uint32_t seed; };
hipError_t launch_primary(const DevScene& s, const DevCamera& cam, int mode, bool count, uint32_t w, uint32_t h,
                          uint32_t y0, uint32_t rows, uint32_t stripe_h, uint32_t stripe_step, uint32_t spp,
                          uint32_t seed, int jitter, rt_ray* d_rays_out, rt_hit* d_hits,
                          unsigned long long* d_counters, hipStream_t st, const PrimaryShade* shade = nullptr);

hipError_t launch_shadow(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                         float lx, float ly, float lz, rt_ray* d_shadow_rays, uint8_t* d_lit, hipStream_t st,
                         bool packets = false /* 64 consecutive shadow rays walk as one packet (needs s.wide) */);

hipError_t launch_bounce(const DevScene& s, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                         uint32_t seed, rt_ray* d_out, hipStream_t st);

// frag.glsl:310-354, one bounce of the path loop for a batch (emission, direct light, next ray, throughput)
hipError_t launch_shade(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                        uint32_t key0, uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step, uint32_t seed,
                        float4* d_throughput, float4* d_radiance, rt_ray* d_out, hipStream_t st,
                        bool first = false /* segment 0: throughput 1 / radiance 0 implied, both written for every slot */);
// frag.glsl:357-359: average over spp, gamma 1/2.6, RGBA8 (+ optional linear float4)
hipError_t launch_resolve(const float4* d_radiance, uint64_t n_pixels, uint32_t spp, float4* d_linear, uint32_t* d_rgba8,
                          hipStream_t st);
// last path segment + resolve in one kernel; spp must be a power of two <= 64 that divides n (else hipErrorInvalidValue)
hipError_t launch_shade_resolve(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                                uint32_t spp, const float4* d_throughput, const float4* d_radiance, float4* d_linear,
                                uint32_t* d_rgba8, hipStream_t st);
hipError_t launch_fill(float4* d, uint64_t n, float x, float y, float z, float w, hipStream_t st);
// frag.glsl:306-354 for every sample of the rows [..] in one kernel, brute-force scenes only (mode kBrute): per-sample radiance
// (ray index ((yl*w + x)*spp + s), as the staged pipeline leaves it), to be resolved by launch_resolve
hipError_t launch_path_brute(const DevScene& s, const DevCamera& cam, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows, uint32_t stripe_h,
                             uint32_t stripe_step, uint32_t spp, uint32_t bounces, uint32_t seed, float4* d_radiance, hipStream_t st);
// Interleaved shares -> image order (multi-slot rt_render / rt_trace_primary): `stage` holds n_shares blocks of
// share_rows rows x row_words 32-bit words; share s owns the stripes s, s + n_shares, ... of stripe_h rows each.
// Row y of the h-row frame is row (y / stripe_h / n_shares) * stripe_h + y % stripe_h of share (y / stripe_h) % n_shares.
hipError_t launch_gather_rows(const uint32_t* stage, uint32_t* frame, uint32_t h, uint32_t row_words, uint32_t stripe_h,
                              uint32_t n_shares, uint32_t share_rows, hipStream_t st);

constexpr int kStackDepth = 32;   // BVH2 depth budget + 2: bounds the per-lane traversal stack of the BVH2 layout
constexpr int kWorkSlots = 4;     // chunk-cursor sets behind the counters: one per scratch slot of a replica (rt_api.cpp)
constexpr int kCursorStride = 16;       // u64 words between chunk cursors: one 128-B line each, so they sit in different L2 channels
constexpr int kWorkWords = 9 * kCursorStride;   // 8 cursors + the mask of exhausted eighths
constexpr int kCountersPad = 16;        // the counters take the first line of the buffer, the cursor sets follow
constexpr int kNumCounters = 12;  // rays, box_tests, nodes, tri_tests, sphere_tests, sum wave-max nodes, sum wave-max tris, waves, + 4 strip-kernel scheduling probes

}  // namespace rt
