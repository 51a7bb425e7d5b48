// rt_api.cpp -- implementation of the C ABI declared in include/raytracer_amd.h.
//
// Host-side orchestration only: argument checks, BVH build + flatten, HBM replicas, chunked
// H2D/D2H for the host-pointer entry points, and kernel launches (kernels.hip).  Seams replaced in the
// reference are cited next to each declaration in the header.
#include "../../include/raytracer_amd.h"
#include "../host/rt_types.hpp"
#include "bvh.hpp"
#include "kernels.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <array>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

using namespace rt;

namespace {

constexpr uint64_t kChunkRays = 1ull << 22;   // host-pointer entry points stream rays in 4 Mi-ray pieces

struct Replica {
    int device = 0;
    hipStream_t stream = nullptr;
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    float4* d_normals = nullptr;
    float4* d_spheres = nullptr;
    float4* d_materials = nullptr;
    float4* d_sphere_mat = nullptr;   // uint32 ids, allocated through the float4 helper
    float4* d_lights = nullptr;
    // scratch for rt_render
    void* d_render = nullptr; size_t render_cap = 0;
    unsigned long long* d_counters = nullptr;
    // scratch for host-pointer entry points
    void* d_rays = nullptr;  size_t rays_cap = 0;
    void* d_out = nullptr;   size_t out_cap = 0;
    // scratch for ray binning (permutation + histograms, uint32 words)
    uint32_t* d_bin = nullptr; uint64_t bin_cap = 0;
    // spill area of the strip kernel's capped stack: kSpillSlots buffers used round-robin (launches in flight)
    int* d_spill = nullptr; uint64_t spill_cap = 0; uint32_t spill_seq = 0;
    // primary rays / hits of the ray-buffer primary path when the caller does not ask for them
    void* d_prim = nullptr; size_t prim_cap = 0;
    std::string err;
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct rt_ctx {
    std::vector<Replica> reps;
    std::string err;
    // options
    int traversal = 0;          // 0 auto, 1 brute force, 2 BVH2 nodes, 3 BVH4q nodes (RT_OPT_TRAVERSAL)
    int bvh_threads = 0;
    int counters = 0;
    int pad_log2 = 15;          // box padding = max|coordinate| * 2^-pad_log2
    int max_leaf = 4;
    uint32_t strip = 256;       // rays per chunk drawn by a wave of the ray-buffer kernel (0 = plain one-ray-per-lane kernel)
    int refill = 16 | (24 << 8); // low byte: idle lanes that trigger a refill; high: leaf-vote threshold
    int lds_stack = 20;         // ray-buffer kernel, BVH4q: stack entries per lane kept in LDS, deeper ones spill to HBM (0 = all in LDS)
    int primary_path = 0;       // primary rays: 0 auto, 1 tile kernel (k_primary), 2 ray generation + ray-buffer kernel
    int binning = 2;            // ray binning before the ray-buffer kernel: 0 off, 1 on, 2 auto (structure larger than the L2s)
    BinGrid bin_grid{};         // 16^3 grid over the triangle bounds of the uploaded scene
    // scene
    bool has_scene = false, has_camera = false;
    rt_scene_stats stats{};
    DevCamera cam{};
    uint32_t stack_need = 0;    // traversal-stack entries the uploaded tree can need
    int layout = kBvh4q;        // node layout the uploaded scene was flattened to
    int mode() const {
        if (traversal == 1) return kBrute;
        if (traversal == 2 || traversal == 3) return layout;
        return stats.n_tris <= 32 ? kBrute : layout;
    }
    DevScene dev_scene(const Replica& r) const {
        DevScene s;
        s.nodes = r.d_nodes; s.tris = r.d_tris; s.normals = r.d_normals; s.spheres = r.d_spheres;
        s.materials = r.d_materials; s.sphere_mat = reinterpret_cast<const uint32_t*>(r.d_sphere_mat); s.lights = r.d_lights;
        s.n_lights = stats.n_lights;
        s.n_nodes = stats.n_nodes; s.n_tris = stats.n_tris; s.n_spheres = stats.n_spheres;
        s.stack_depth = stack_need;
        return s;
    }
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int fail_hip(rt_ctx* c, hipError_t e, const char* what) {
    return fail(c, RT_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
#define RT_HIP(ctx, call)                                              \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return fail_hip((ctx), e__, #call);     \
    } while (0)

void free_scene(Replica& r) {
    (void)hipSetDevice(r.device);
    if (r.d_nodes) (void)hipFree(r.d_nodes);
    if (r.d_tris) (void)hipFree(r.d_tris);
    if (r.d_normals) (void)hipFree(r.d_normals);
    if (r.d_spheres) (void)hipFree(r.d_spheres);
    if (r.d_materials) (void)hipFree(r.d_materials);
    if (r.d_sphere_mat) (void)hipFree(r.d_sphere_mat);
    if (r.d_lights) (void)hipFree(r.d_lights);
    r.d_nodes = r.d_tris = r.d_normals = r.d_spheres = r.d_materials = r.d_sphere_mat = r.d_lights = nullptr;
}

// Closest-hit / any-hit over a ray buffer in HBM.  With binning the rays are handed out grouped by origin cell
// (launch_bin), which keeps a wave's node fetches close together when the structure does not fit the L2s.
constexpr uint64_t kBinMinRays = 1ull << 18;
constexpr uint64_t kBinMinSceneBytes = 48ull << 20;
constexpr uint32_t kSpillSlots = 4;       // strip launches of one slot that may be in flight at once
constexpr uint32_t kBinMinRun = 128;      // mean rays per non-empty (64 Ki-ray block, cell) pair above which a buffer counts as coherent
hipError_t trace_rays(rt_ctx* c, Replica& r, bool any_hit, bool count, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits,
                      uint8_t* d_occ, hipStream_t st) {
    const int mode = c->mode();
    const uint64_t scene_bytes = (uint64_t)c->stats.n_nodes * 64 + (uint64_t)c->stats.n_tris * 48;
    const bool bin = mode != kBrute && c->strip >= 64 && n >= kBinMinRays && n < (1ull << 32) &&
                     (c->binning == 1 || (c->binning == 2 && scene_bytes > kBinMinSceneBytes));
    const uint32_t* perm = nullptr;
    if (bin) {
        const uint64_t words = bin_scratch_words(n);
        if (words > r.bin_cap) {
            if (r.d_bin) (void)hipFree(r.d_bin);
            r.d_bin = nullptr; r.bin_cap = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&r.d_bin), words * sizeof(uint32_t));
            if (e != hipSuccess) return e;
            r.bin_cap = words;
        }
        // auto: permute only buffers that are not grouped by origin already (camera-ordered bounce rays are)
        hipError_t e = launch_bin(d_rays, n, c->bin_grid, r.d_bin, c->binning == 1 ? 0u : kBinMinRun, st);
        if (e != hipSuccess) return e;
        perm = r.d_bin;
    }
    int* spill = nullptr;
    if (mode == kBvh4q && c->strip >= 64) {
        const uint64_t words = c->lds_stack ? spill_words(c->stack_need, c->lds_stack, device_cus()) : 0;
        if (words) {
            if (words * kSpillSlots > r.spill_cap) {
                if (r.d_spill) (void)hipFree(r.d_spill);
                r.d_spill = nullptr; r.spill_cap = 0;
                hipError_t e = hipMalloc(reinterpret_cast<void**>(&r.d_spill), words * kSpillSlots * sizeof(int));
                if (e != hipSuccess) return e;
                r.spill_cap = words * kSpillSlots;
            }
            spill = r.d_spill + (uint64_t)(r.spill_seq++ % kSpillSlots) * words;
        }
    }
    return launch_trace(c->dev_scene(r), mode, any_hit, count, d_rays, n, d_hits, d_occ, r.d_counters, st, c->strip, c->refill, perm,
                        perm ? bin_flag(perm, n) : nullptr, spill, c->lds_stack);
}

// Primary rays of a "hard" tree go through the ray-buffer kernel (ray generation -> k_trace_strip).  What decides is
// not the size of the scene but how long and how divergent traversals are, which the tree's SAH cost predicts: on
// random soups (SAH cost 300..1800) neighbouring pixels walk unrelated paths and the tile kernel, which walks each ray
// of a wave to the end, loses (10 M triangles: 50 ms against 17 ms); on tessellated surfaces (SAH cost 8..72, up to
// 8 M triangles measured) neighbouring rays stay together and the tile kernel is ahead (C3: 2.7 against 3.7 ms).
// Crossover measured at SAH cost ~250 (scripts/primary_threshold*.py).  Same rays, same hits either way.
constexpr float kPrimaryBufferSah = 250.0f;
bool primary_via_buffer(const rt_ctx* c) {
    const int mode = c->mode();
    if (mode == kBrute || c->strip < 64 || c->counters || c->primary_path == 1) return false;
    if (c->primary_path == 2) return true;
    return c->stats.sah_cost > kPrimaryBufferSah;
}

// rays (32 B) and / or hits (16 B) for n local ray slots from the slot's scratch when the caller passes none
hipError_t primary_scratch(Replica& r, uint64_t n, rt_ray** rays, rt_hit** hits) {
    const size_t need = (size_t)n * ((*rays ? 0 : sizeof(rt_ray)) + (*hits ? 0 : sizeof(rt_hit)));
    if (need > r.prim_cap) {
        if (r.d_prim) (void)hipFree(r.d_prim);
        r.d_prim = nullptr; r.prim_cap = 0;
        hipError_t e = hipMalloc(&r.d_prim, need);
        if (e != hipSuccess) return e;
        r.prim_cap = need;
    }
    char* p = static_cast<char*>(r.d_prim);
    if (!*rays) { *rays = reinterpret_cast<rt_ray*>(p); p += (size_t)n * sizeof(rt_ray); }
    if (!*hits) *hits = reinterpret_cast<rt_hit*>(p);
    return hipSuccess;
}

bool valid_slot(const rt_ctx* c, int slot) { return c && slot >= 0 && slot < (int)c->reps.size(); }

// Run fn(slot) on every replica; one host thread per replica when there are several (they drive
// different devices, and pageable-memory copies block the calling thread).
template <class F>
int for_each_slot(rt_ctx* c, F fn) {
    const int n = (int)c->reps.size();
    std::vector<int> rc(n, RT_OK);
    if (n == 1) {
        rc[0] = fn(0);
    } else {
        std::vector<std::thread> th;
        for (int s = 0; s < n; ++s) th.emplace_back([&, s] { rc[s] = fn(s); });
        for (auto& t : th) t.join();
    }
    for (int s = 0; s < n; ++s)
        if (rc[s] != RT_OK) { c->err = c->reps[s].err; return rc[s]; }
    return RT_OK;
}

int slot_fail(Replica& r, hipError_t e, const char* what) {
    r.err = std::string(what) + ": " + hipGetErrorString(e);
    return RT_E_DEVICE;
}
#define RT_HIP_SLOT(r, call)                                           \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) return slot_fail((r), e__, #call);      \
    } while (0)

}  // namespace

extern "C" {

int rt_create(rt_ctx** out, const int* device_ids, int n_devices) {
    if (!out) return RT_E_INVALID;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return RT_E_INVALID;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return RT_E_DEVICE;
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return RT_E_NOMEM;
    c->reps.resize(n_devices);
    for (int i = 0; i < n_devices; ++i) {
        int d = device_ids ? device_ids[i] : i;
        if (d < 0 || d >= ndev) { delete c; return RT_E_INVALID; }
        Replica& r = c->reps[i];
        r.device = d;
        if (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&r.d_counters, (kCountersPad + (size_t)kWorkSlots * kWorkWords) * sizeof(unsigned long long)) != hipSuccess ||
            hipMemset(r.d_counters, 0, (kCountersPad + (size_t)kWorkSlots * kWorkWords) * sizeof(unsigned long long)) != hipSuccess) {
            rt_destroy(c);
            return RT_E_DEVICE;
        }
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    for (Replica& r : c->reps) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        free_scene(r);
        if (r.d_counters) (void)hipFree(r.d_counters);
        if (r.d_rays) (void)hipFree(r.d_rays);
        if (r.d_out) (void)hipFree(r.d_out);
        if (r.d_bin) (void)hipFree(r.d_bin);
        if (r.d_spill) (void)hipFree(r.d_spill);
        if (r.d_prim) (void)hipFree(r.d_prim);
        if (r.d_render) (void)hipFree(r.d_render);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rt_num_slots(const rt_ctx* c) { return c ? (int)c->reps.size() : 0; }

int rt_set_option(rt_ctx* c, int key, int64_t v) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    switch (key) {
        case RT_OPT_TRAVERSAL:   if (v < 0 || v > 3) return fail(c, RT_E_INVALID, "RT_OPT_TRAVERSAL: 0 auto, 1 brute, 2 BVH2, 3 BVH4q"); c->traversal = (int)v; return RT_OK;
        case RT_OPT_BVH_THREADS: if (v < 0 || v > 256) return fail(c, RT_E_INVALID, "RT_OPT_BVH_THREADS out of range"); c->bvh_threads = (int)v; return RT_OK;
        case RT_OPT_COUNTERS:    c->counters = v ? 1 : 0; return RT_OK;
        case RT_OPT_LDS_STACK:   if (v != 0 && (v < 2 || v > 64)) return fail(c, RT_E_INVALID, "RT_OPT_LDS_STACK: 0 or 2..64 entries"); c->lds_stack = (int)v; return RT_OK;
        case RT_OPT_PRIMARY:     if (v < 0 || v > 2) return fail(c, RT_E_INVALID, "RT_OPT_PRIMARY: 0 auto, 1 tile kernel, 2 ray-buffer kernel"); c->primary_path = (int)v; return RT_OK;
        case RT_OPT_RAY_BINNING: if (v < 0 || v > 2) return fail(c, RT_E_INVALID, "RT_OPT_RAY_BINNING: 0 off, 1 on, 2 auto"); c->binning = (int)v; return RT_OK;
        case RT_OPT_PAD_LOG2:    if (v < 4 || v > 64) return fail(c, RT_E_INVALID, "RT_OPT_PAD_LOG2 in [4,64]"); c->pad_log2 = (int)v; return RT_OK;
        case RT_OPT_MAX_LEAF:    if (v < 1 || v > 8) return fail(c, RT_E_INVALID, "RT_OPT_MAX_LEAF in [1,8]"); c->max_leaf = (int)v; return RT_OK;
        case RT_OPT_STRIP:       if (v != 0 && (v < 64 || v > (1 << 24))) return fail(c, RT_E_INVALID, "RT_OPT_STRIP: 0 or 64..2^24"); c->strip = (uint32_t)v; return RT_OK;
        case RT_OPT_REFILL:      if ((v & 0xff) < 1 || (v & 0xff) > 64 || (v >> 8) > 64) return fail(c, RT_E_INVALID, "RT_OPT_REFILL: low byte 1..64 idle lanes, bits 8.. = leaf-vote threshold 0..64"); c->refill = (int)v; return RT_OK;
        default: return fail(c, RT_E_INVALID, "unknown option key");
    }
}

int rt_scene_upload(rt_ctx* c, const void* tris, uint32_t n_tris, uint32_t tri_stride, const void* spheres,
                    uint32_t n_spheres, uint32_t sphere_stride, const void* lights, uint32_t n_lights,
                    uint32_t light_stride) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (n_tris && (!tris || tri_stride < sizeof(Triangle) || (tri_stride & 3))) return fail(c, RT_E_INVALID, "triangles: null pointer or stride < 144 / not a multiple of 4");
    if (n_spheres && (!spheres || sphere_stride < sizeof(Sphere) || (sphere_stride & 3))) return fail(c, RT_E_INVALID, "spheres: null pointer or stride < 64 / not a multiple of 4");
    if (n_lights && (!lights || light_stride < sizeof(Light) || (light_stride & 3))) return fail(c, RT_E_INVALID, "lights: null pointer or stride < 32 / not a multiple of 4");
    if (n_tris >= (1u << 28)) return fail(c, RT_E_TOO_LARGE, "more than 2^28-1 triangles");
    if (n_spheres >= (1u << 31)) return fail(c, RT_E_TOO_LARGE, "more than 2^31-1 spheres");

    const double t0 = now_ms();
    // 1. exact triangle bounds (positions at byte offsets 0 / 32 / 64 of the 144-B struct)
    std::vector<Aabb> boxes(n_tris);
    float maxabs = 0.f;
    const char* tb = static_cast<const char*>(tris);
    Aabb scene_box; scene_box.reset();
    for (uint32_t i = 0; i < n_tris; ++i) {
        const char* t = tb + (size_t)i * tri_stride;
        Aabb b; b.reset();
        for (int v = 0; v < 3; ++v) {
            float p[3];
            std::memcpy(p, t + 32 * v, 12);
            for (int a = 0; a < 3; ++a) {
                if (!std::isfinite(p[a])) return fail(c, RT_E_INVALID, "triangle " + std::to_string(i) + " has a non-finite position");
                maxabs = std::max(maxabs, std::fabs(p[a]));
            }
            b.grow(p);
        }
        boxes[i] = b;
        scene_box.grow(b);
    }
    // 2. BVH2 build + flatten
    BuildParams bp;
    bp.max_leaf = c->max_leaf;
    bp.max_depth = kStackDepth - 2;
    bp.threads = c->bvh_threads;
    BuildResult br;
    build_bvh2(boxes.data(), n_tris, bp, br);
    for (const BuildNode& nd : br.nodes)
        if (nd.count > 8) return fail(c, RT_E_TOO_LARGE, "internal: leaf with more than 8 triangles (depth budget exceeded)");
    // traversal option 2 asks for the plain BVH2 layout; everything else gets the 4-wide quantised one (faster on
    // every measured config since the sign-selected plane test, scripts/cmp_layouts.py)
    int layout = (c->traversal == 2) ? kBvh2 : kBvh4q;
    const float pad = std::ldexp(std::max(maxabs, 1e-30f), -c->pad_log2);
    struct { std::vector<float> nodes; std::vector<uint32_t> tri_order; uint32_t n_nodes = 0, n_leaves = 0, depth = 0, stack_need = 0; } flat;
    if (layout == kBvh4q) {
        FlatBvh4q f4;
        flatten_bvh4q(br, n_tris ? pad : 0.f, f4);
        if (f4.stack_need + 2 > 64) {
            layout = kBvh2;                     // pathological tree: its 4-wide stack would not fit 64 KB of LDS
        } else {
            flat.nodes.resize(f4.nodes.size());
            if (!f4.nodes.empty()) std::memcpy(flat.nodes.data(), f4.nodes.data(), f4.nodes.size() * 4);
            flat.tri_order.swap(f4.tri_order);
            flat.n_nodes = f4.n_nodes; flat.n_leaves = f4.n_leaves; flat.depth = f4.depth; flat.stack_need = f4.stack_need;
        }
    }
    if (layout == kBvh2) {
        FlatBvh2 f2;
        flatten_bvh2(br, n_tris ? pad : 0.f, f2);
        flat.nodes.swap(f2.nodes); flat.tri_order.swap(f2.tri_order);
        flat.n_nodes = f2.n_nodes; flat.n_leaves = f2.n_leaves; flat.depth = f2.depth;
        flat.stack_need = std::max(f2.depth, bvh2_stack_need(br));
    }
    if (flat.stack_need + 2 > 64) return fail(c, RT_E_TOO_LARGE, "internal: traversal stack would exceed the LDS budget");

    // 3. device records
    std::vector<float> tri48((size_t)n_tris * 12), nrm48((size_t)n_tris * 12), sph((size_t)n_spheres * 4);
    for (uint32_t k = 0; k < n_tris; ++k) {
        const uint32_t id = flat.tri_order[k];
        const char* t = tb + (size_t)id * tri_stride;
        float p0[3], p1[3], p2[3];
        std::memcpy(p0, t, 12); std::memcpy(p1, t + 32, 12); std::memcpy(p2, t + 64, 12);
        float* o = &tri48[(size_t)k * 12];
        o[0] = p0[0]; o[1] = p0[1]; o[2] = p0[2]; std::memcpy(&o[3], &id, 4);
        // frag.glsl:132-133: edge1 = v1 - v0, edge2 = v2 - v0 (plain fp32 subtraction, no contraction possible)
        o[4] = p1[0] - p0[0]; o[5] = p1[1] - p0[1]; o[6] = p1[2] - p0[2]; o[7] = 0.f;
        o[8] = p2[0] - p0[0]; o[9] = p2[1] - p0[1]; o[10] = p2[2] - p0[2]; o[11] = 0.f;
    }
    // distinct materials (src/math/Material.h:6-11, 9 floats used) -> table; ids ride in normals[3*i].w / sphere_mat
    std::vector<float> mats;                       // 12 floats per material: (albedo, strength) (emission, roughness) (metallic,0,0,0)
    std::map<std::array<uint32_t, 9>, uint32_t> mat_ids;
    auto material_id = [&](const char* m48) -> uint32_t {
        float f[10];
        std::memcpy(f, m48, 40);                   // albedo@0, emission@16, strength@28, roughness@32, metallic@36
        std::array<uint32_t, 9> key;
        const float vals[9] = { f[0], f[1], f[2], f[4], f[5], f[6], f[7], f[8], f[9] };
        std::memcpy(key.data(), vals, 36);
        auto it = mat_ids.find(key);
        if (it != mat_ids.end()) return it->second;
        const uint32_t id = (uint32_t)mat_ids.size();
        mat_ids.emplace(key, id);
        const float rec[12] = { f[0], f[1], f[2], f[7], f[4], f[5], f[6], f[8], f[9], 0.f, 0.f, 0.f };
        mats.insert(mats.end(), rec, rec + 12);
        return id;
    };
    for (uint32_t i = 0; i < n_tris; ++i) {
        const char* t = tb + (size_t)i * tri_stride;
        float* o = &nrm48[(size_t)i * 12];
        for (int v = 0; v < 3; ++v) { std::memcpy(o + 4 * v, t + 16 + 32 * v, 12); o[4 * v + 3] = 0.f; }
        const uint32_t id = material_id(t + 96);
        std::memcpy(&o[3], &id, 4);
    }
    std::vector<float> sphere_mat((size_t)n_spheres), lights8((size_t)n_lights * 8);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const uint32_t id = material_id(static_cast<const char*>(spheres) + (size_t)i * sphere_stride + 16);
        std::memcpy(&sphere_mat[i], &id, 4);
    }
    for (uint32_t i = 0; i < n_lights; ++i) {      // src/math/Light.h:6-10: position@0, colour@16, intensity@28
        const char* l = static_cast<const char*>(lights) + (size_t)i * light_stride;
        float* o = &lights8[(size_t)i * 8];
        std::memcpy(o, l, 12); std::memcpy(o + 3, l + 28, 4);
        std::memcpy(o + 4, l + 16, 12); o[7] = 0.f;
    }
    const char* sb = static_cast<const char*>(spheres);
    for (uint32_t i = 0; i < n_spheres; ++i) std::memcpy(&sph[(size_t)i * 4], sb + (size_t)i * sphere_stride, 16);
    const double t1 = now_ms();

    // 4. replicate
    c->has_scene = false;
    for (Replica& r : c->reps) {
        RT_HIP(c, hipSetDevice(r.device));
        RT_HIP(c, hipStreamSynchronize(r.stream));
        free_scene(r);
        auto up = [&](float4** dst, const std::vector<float>& src) -> hipError_t {
            const size_t bytes = std::max<size_t>(src.size() * sizeof(float), 64);
            hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
            if (e != hipSuccess) return e;
            e = hipMemset(*dst, 0, bytes);
            if (e != hipSuccess || src.empty()) return e;
            return hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
        };
        RT_HIP(c, up(&r.d_nodes, flat.nodes));
        RT_HIP(c, up(&r.d_tris, tri48));
        RT_HIP(c, up(&r.d_normals, nrm48));
        RT_HIP(c, up(&r.d_spheres, sph));
        RT_HIP(c, up(&r.d_materials, mats));
        RT_HIP(c, up(&r.d_sphere_mat, sphere_mat));
        RT_HIP(c, up(&r.d_lights, lights8));
    }
    const double t2 = now_ms();

    rt_scene_stats& st = c->stats;
    st = rt_scene_stats{};
    st.n_tris = n_tris; st.n_spheres = n_spheres; st.n_lights = n_lights;
    st.n_nodes = flat.n_nodes; st.n_leaves = flat.n_leaves; st.max_depth = flat.depth;
    c->stack_need = flat.stack_need; c->layout = layout;
    {
        float* lo = &c->bin_grid.lox; float* sc = &c->bin_grid.sx;
        for (int a = 0; a < 3; ++a) {
            const float ext = n_tris ? scene_box.hi[a] - scene_box.lo[a] : 0.f;
            lo[a] = n_tris ? scene_box.lo[a] : 0.f;
            sc[a] = ext > 0.f ? 16.0f / ext : 0.f;
        }
    }
    st.node_bytes = 64; st.tri_bytes = 48;
    st.sah_cost = br.sah_cost;
    st.build_ms = (float)(t1 - t0); st.upload_ms = (float)(t2 - t1);
    st.device_bytes = (uint64_t)flat.nodes.size() * 4 + (uint64_t)tri48.size() * 4 + (uint64_t)nrm48.size() * 4 + (uint64_t)sph.size() * 4;
    c->has_scene = true;
    return RT_OK;
}

// Host-only BVH build + flatten (no device): what rt_scene_upload runs before copying to HBM.
int rt_bvh_build_host(const void* tris, uint32_t n_tris, uint32_t tri_stride, int layout, int max_leaf, int pad_log2,
                      int threads, rt_scene_stats* stats, uint32_t* nodes_out, uint64_t nodes_cap_words,
                      uint32_t* tri_order_out) {
    if (!stats || (n_tris && (!tris || tri_stride < sizeof(Triangle) || (tri_stride & 3)))) return RT_E_INVALID;
    if (n_tris >= (1u << 28)) return RT_E_TOO_LARGE;
    if (layout != kBvh2 && layout != kBvh4q) return RT_E_INVALID;
    const double t0 = now_ms();
    std::vector<Aabb> boxes(n_tris);
    float maxabs = 0.f;
    const char* tb = static_cast<const char*>(tris);
    for (uint32_t i = 0; i < n_tris; ++i) {
        Aabb b; b.reset();
        for (int v = 0; v < 3; ++v) {
            float p[3];
            std::memcpy(p, tb + (size_t)i * tri_stride + 32 * v, 12);
            for (int a = 0; a < 3; ++a) { if (!std::isfinite(p[a])) return RT_E_INVALID; maxabs = std::max(maxabs, std::fabs(p[a])); }
            b.grow(p);
        }
        boxes[i] = b;
    }
    BuildParams bp;
    bp.max_leaf = max_leaf > 0 ? max_leaf : 4;
    bp.max_depth = kStackDepth - 2;
    bp.threads = threads;
    BuildResult br;
    build_bvh2(boxes.data(), n_tris, bp, br);
    const float pad = std::ldexp(std::max(maxabs, 1e-30f), -(pad_log2 > 0 ? pad_log2 : 15));
    *stats = rt_scene_stats{};
    stats->n_tris = n_tris; stats->sah_cost = br.sah_cost; stats->node_bytes = 64; stats->tri_bytes = 48;
    std::vector<uint32_t> words, order;
    if (layout == kBvh2) {
        FlatBvh2 f; flatten_bvh2(br, n_tris ? pad : 0.f, f);
        words.resize(f.nodes.size());
        if (!f.nodes.empty()) std::memcpy(words.data(), f.nodes.data(), f.nodes.size() * 4);
        order.swap(f.tri_order);
        stats->n_nodes = f.n_nodes; stats->n_leaves = f.n_leaves; stats->max_depth = f.depth;
    } else {
        FlatBvh4q f; flatten_bvh4q(br, n_tris ? pad : 0.f, f);
        words.swap(f.nodes); order.swap(f.tri_order);
        stats->n_nodes = f.n_nodes; stats->n_leaves = f.n_leaves; stats->max_depth = f.depth;
        stats->n_lights = f.stack_need;          // host-only entry: reports the 4-wide stack bound here
    }
    stats->build_ms = (float)(now_ms() - t0);
    stats->device_bytes = (uint64_t)words.size() * 4 + (uint64_t)n_tris * 96;
    if (nodes_out) {
        if (nodes_cap_words < words.size()) return RT_E_TOO_LARGE;
        if (!words.empty()) std::memcpy(nodes_out, words.data(), words.size() * 4);
    }
    if (tri_order_out && !order.empty()) std::memcpy(tri_order_out, order.data(), order.size() * 4);
    return RT_OK;
}

int rt_scene_stats_get(const rt_ctx* c, rt_scene_stats* out) {
    if (!c || !out) return RT_E_INVALID;
    if (!c->has_scene) return RT_E_NO_SCENE;
    *out = c->stats;
    return RT_OK;
}

int rt_set_camera(rt_ctx* c, const void* ubo) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!ubo) return fail(c, RT_E_INVALID, "camera: null pointer");
    CameraUBO u;
    std::memcpy(&u, ubo, sizeof u);
    DevCamera d{};
    d.px = u.m_position.x; d.py = u.m_position.y; d.pz = u.m_position.z;
    d.rx = u.m_right.x; d.ry = u.m_right.y; d.rz = u.m_right.z;
    d.ux = u.m_up.x; d.uy = u.m_up.y; d.uz = u.m_up.z;
    d.fx = u.m_front.x; d.fy = u.m_front.y; d.fz = u.m_front.z;
    // frag.glsl:106-110.  radians() = deg * pi/180 in fp32; tan from the host libm.
    const float fov = u.m_fov * 0.017453292519943295f;
    d.half_h = tanf(fov / 2.0f);
    d.half_w = d.half_h * u.m_aspectRatio;
    c->cam = d;
    c->has_camera = true;
    return RT_OK;
}

// ---- device-resident stages -------------------------------------------------------------------------
static int check_stage(rt_ctx* c, int slot, bool need_cam) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (need_cam && !c->has_camera) return fail(c, RT_E_NO_CAMERA, "no camera set");
    return RT_OK;
}

int rt_trace_primary_stripes_dev(rt_ctx* c, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                 uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t seed, int jitter,
                                 rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream) {
    int rc = check_stage(c, slot, true);
    if (rc) return rc;
    if (!d_hits_out || w == 0 || h == 0 || spp == 0 || stripe_h == 0 || stripe_step == 0) return fail(c, RT_E_INVALID, "primary: bad arguments");
    if ((uint64_t)w * h * spp > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "primary: w*h*spp exceeds 2^32-1 (RNG key width)");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    if (primary_via_buffer(c)) {
        const uint64_t n = (uint64_t)rows * w * spp;
        rt_ray* rays = d_rays_out;
        rt_hit* hits = d_hits_out;
        RT_HIP(c, primary_scratch(r, n, &rays, &hits));
        RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, kNone, false, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter,
                                 rays, nullptr, r.d_counters, st));
        RT_HIP(c, trace_rays(c, r, false, false, rays, n, hits, nullptr, st));
        return RT_OK;
    }
    RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, c->mode(), c->counters != 0, w, h, y0, rows, stripe_h, stripe_step,
                             spp, seed, jitter, d_rays_out, d_hits_out, r.d_counters, st));
    return RT_OK;
}

int rt_primary_shade_stripes_dev(rt_ctx* c, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                 uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t seed, int jitter,
                                 uint32_t shade_seed, rt_ray* d_rays_out, rt_hit* d_hits_out, float* d_throughput4,
                                 float* d_radiance4, rt_ray* d_next_out, void* stream) {
    int rc = check_stage(c, slot, true);
    if (rc) return rc;
    if (!d_throughput4 || !d_radiance4 || w == 0 || h == 0 || spp == 0 || stripe_h == 0 || stripe_step == 0)
        return fail(c, RT_E_INVALID, "primary+shade: bad arguments");
    if ((uint64_t)w * h * spp > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "primary: w*h*spp exceeds 2^32-1 (RNG key width)");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    if (primary_via_buffer(c)) {
        const uint64_t n = (uint64_t)rows * w * spp;
        rt_ray* rays = d_rays_out;
        rt_hit* hits = d_hits_out;
        RT_HIP(c, primary_scratch(r, n, &rays, &hits));
        const DevScene ds = c->dev_scene(r);
        RT_HIP(c, launch_primary(ds, c->cam, kNone, false, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter, rays, nullptr,
                                 r.d_counters, st));
        RT_HIP(c, trace_rays(c, r, false, false, rays, n, hits, nullptr, st));
        RT_HIP(c, launch_shade(ds, c->mode(), rays, hits, n, (uint32_t)((uint64_t)y0 * w * spp), w * spp, stripe_h, stripe_step, shade_seed,
                               reinterpret_cast<float4*>(d_throughput4), reinterpret_cast<float4*>(d_radiance4), d_next_out, st, true));
        return RT_OK;
    }
    const PrimaryShade ps = { reinterpret_cast<float4*>(d_throughput4), reinterpret_cast<float4*>(d_radiance4), d_next_out, shade_seed };
    RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, c->mode(), false, w, h, y0, rows, stripe_h, stripe_step,
                             spp, seed, jitter, d_rays_out, d_hits_out, r.d_counters, st, &ps));
    return RT_OK;
}

int rt_trace_primary_dev(rt_ctx* c, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows, uint32_t spp,
                         uint32_t seed, int jitter, rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream) {
    if (c && (uint64_t)y0 + rows > h) return fail(c, RT_E_INVALID, "primary: y0 + rows > h");
    return rt_trace_primary_stripes_dev(c, slot, w, h, y0, rows, rows ? rows : 1, 1, spp, seed, jitter, d_rays_out, d_hits_out, stream);
}

int rt_trace_dev(rt_ctx* c, int slot, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits)) return fail(c, RT_E_INVALID, "trace: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, trace_rays(c, r, false, c->counters != 0, d_rays, n, d_hits, nullptr, st));
    return RT_OK;
}

int rt_occluded_dev(rt_ctx* c, int slot, const rt_ray* d_rays, uint64_t n, uint8_t* d_out, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_out)) return fail(c, RT_E_INVALID, "occluded: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, trace_rays(c, r, true, false, d_rays, n, nullptr, d_out, st));
    return RT_OK;
}

int rt_shadow_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, const float lp[3],
                  rt_ray* d_srays, uint8_t* d_lit, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (!lp || (n && (!d_rays || !d_hits || !d_lit))) return fail(c, RT_E_INVALID, "shadow: null pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, launch_shadow(c->dev_scene(r), c->mode(), d_rays, d_hits, n, lp[0], lp[1], lp[2], d_srays, d_lit, st));
    return RT_OK;
}

int rt_bounce_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                  uint32_t seed, rt_ray* d_out, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits || !d_out)) return fail(c, RT_E_INVALID, "bounce: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, launch_bounce(c->dev_scene(r), d_rays, d_hits, n, key0, seed, d_out, st));
    return RT_OK;
}

int rt_shade_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                 uint32_t seed, float* d_throughput4, float* d_radiance4, rt_ray* d_out, void* stream) {
    return rt_shade_stripes_dev(c, slot, d_rays, d_hits, n, key0, 0, 1, 1, seed, d_throughput4, d_radiance4, d_out, stream);
}

int rt_shade_stripes_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                         uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step, uint32_t seed, float* d_throughput4,
                         float* d_radiance4, rt_ray* d_out, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits || !d_throughput4 || !d_radiance4)) return fail(c, RT_E_INVALID, "shade: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    RT_HIP(c, launch_shade(c->dev_scene(r), c->mode(), d_rays, d_hits, n, key0, row_len, stripe_h, stripe_step, seed, reinterpret_cast<float4*>(d_throughput4),
                           reinterpret_cast<float4*>(d_radiance4), d_out, (hipStream_t)stream));
    return RT_OK;
}

int rt_path_init_dev(rt_ctx* c, int slot, uint64_t n, float* d_throughput4, float* d_radiance4, void* stream) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    if (n && (!d_throughput4 || !d_radiance4)) return fail(c, RT_E_INVALID, "path_init: null device pointer");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, launch_fill(reinterpret_cast<float4*>(d_throughput4), n, 1.f, 1.f, 1.f, 0.f, (hipStream_t)stream));   // frag.glsl:308
    RT_HIP(c, launch_fill(reinterpret_cast<float4*>(d_radiance4), n, 0.f, 0.f, 0.f, 0.f, (hipStream_t)stream));     // :304
    return RT_OK;
}

int rt_resolve_dev(rt_ctx* c, int slot, const float* d_radiance4, uint64_t n_pixels, uint32_t spp, float* d_linear4,
                   uint8_t* d_rgba8, void* stream) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    if (spp == 0 || (n_pixels && (!d_radiance4 || (!d_linear4 && !d_rgba8)))) return fail(c, RT_E_INVALID, "resolve: bad arguments");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, launch_resolve(reinterpret_cast<const float4*>(d_radiance4), n_pixels, spp, reinterpret_cast<float4*>(d_linear4),
                             reinterpret_cast<uint32_t*>(d_rgba8), (hipStream_t)stream));
    return RT_OK;
}

int rt_shade_resolve_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t spp,
                         float* d_throughput4, float* d_radiance4, float* d_linear4, uint8_t* d_rgba8, void* stream) {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (spp == 0 || n % spp) return fail(c, RT_E_INVALID, "shade+resolve: n must be a multiple of spp");
    if (n && (!d_rays || !d_hits || !d_throughput4 || !d_radiance4 || (!d_linear4 && !d_rgba8)))
        return fail(c, RT_E_INVALID, "shade+resolve: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;
    const DevScene ds = c->dev_scene(r);
    float4* thr = reinterpret_cast<float4*>(d_throughput4);
    float4* rad = reinterpret_cast<float4*>(d_radiance4);
    if (spp <= 64 && (spp & (spp - 1)) == 0) {
        RT_HIP(c, launch_shade_resolve(ds, c->mode(), d_rays, d_hits, n, spp, thr, rad, reinterpret_cast<float4*>(d_linear4),
                                       reinterpret_cast<uint32_t*>(d_rgba8), st));
    } else {                                     // samples of a pixel would straddle waves: the two separate kernels
        RT_HIP(c, launch_shade(ds, c->mode(), d_rays, d_hits, n, 0, 0, 1, 1, 0, thr, rad, nullptr, st));
        RT_HIP(c, launch_resolve(rad, n / spp, spp, reinterpret_cast<float4*>(d_linear4), reinterpret_cast<uint32_t*>(d_rgba8), st));
    }
    return RT_OK;
}

// frag.glsl:303-360 for a whole frame, as a wavefront pipeline: k_primary, then per bounce k_shade + closest hit.
int rt_render(rt_ctx* c, uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces, uint32_t seed, uint8_t* host_rgba8) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (!c->has_camera) return fail(c, RT_E_NO_CAMERA, "no camera set");
    if (!host_rgba8 || w == 0 || h == 0 || spp == 0 || bounces == 0) return fail(c, RT_E_INVALID, "render: null output or zero size");
    if ((uint64_t)w * h * spp > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "render: w*h*spp exceeds 2^32-1");
    const int ns = (int)c->reps.size();
    const uint32_t per = ((h + ns - 1) / ns + 7) / 8 * 8;
    const int mode = c->mode();
    return for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        const uint32_t y0 = std::min<uint64_t>((uint64_t)per * s, h), y1 = std::min<uint64_t>((uint64_t)y0 + per, h);
        if (y1 <= y0) return RT_OK;
        const uint32_t rows = y1 - y0;
        const uint64_t n = (uint64_t)rows * w * spp, npx = (uint64_t)rows * w;
        // rays A/B (32 B), hits (16 B), throughput (16 B), radiance (16 B) per ray + RGBA8 per pixel
        const size_t bytes = n * (32 + 32 + 16 + 16 + 16) + npx * 4;
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        if (bytes > r.render_cap) {
            if (r.d_render) (void)hipFree(r.d_render);
            r.d_render = nullptr; r.render_cap = 0;
            RT_HIP_SLOT(r, hipMalloc(&r.d_render, bytes));
            r.render_cap = bytes;
        }
        char* base = static_cast<char*>(r.d_render);
        rt_ray* ra = reinterpret_cast<rt_ray*>(base);
        rt_ray* rb = reinterpret_cast<rt_ray*>(base + n * 32);
        rt_hit* hits = reinterpret_cast<rt_hit*>(base + n * 64);
        float4* thr = reinterpret_cast<float4*>(base + n * 80);
        float4* rad = reinterpret_cast<float4*>(base + n * 96);
        uint32_t* rgba = reinterpret_cast<uint32_t*>(base + n * 112);
        const DevScene ds = c->dev_scene(r);
        // segment 0: primary rays + their shading in one kernel (the primary rays and hits stay in registers)
        const uint32_t key0 = (uint32_t)((uint64_t)y0 * w * spp);
        if (primary_via_buffer(c)) {             // big tree: ray generation -> ray-buffer kernel -> first-segment shade
            RT_HIP_SLOT(r, launch_primary(ds, c->cam, kNone, false, w, h, y0, rows, rows, 1, spp, seed, 1, rb, nullptr, r.d_counters, r.stream));
            RT_HIP_SLOT(r, trace_rays(c, r, false, false, rb, n, hits, nullptr, r.stream));
            RT_HIP_SLOT(r, launch_shade(ds, mode, rb, hits, n, key0, 0, 1, 1, seed + 1u, thr, rad, bounces > 1 ? ra : nullptr, r.stream, true));
        } else {
            const PrimaryShade ps = { thr, rad, bounces > 1 ? ra : nullptr, seed + 1u };
            RT_HIP_SLOT(r, launch_primary(ds, c->cam, mode, false, w, h, y0, rows, rows, 1, spp, seed, 1, nullptr, nullptr, r.d_counters, r.stream, &ps));
        }
        const bool fuse_last = bounces > 1 && spp <= 64 && (spp & (spp - 1)) == 0;    // last segment + resolve in one kernel
        for (uint32_t b = 1; b < bounces; ++b) {
            RT_HIP_SLOT(r, trace_rays(c, r, false, false, ra, n, hits, nullptr, r.stream));
            if (b + 1 == bounces && fuse_last) {
                RT_HIP_SLOT(r, launch_shade_resolve(ds, mode, ra, hits, n, spp, thr, rad, nullptr, rgba, r.stream));
                break;
            }
            RT_HIP_SLOT(r, launch_shade(ds, mode, ra, hits, n, key0, 0, 1, 1, seed + 1u + b, thr, rad, b + 1 < bounces ? rb : nullptr, r.stream));
            std::swap(ra, rb);
        }
        if (!fuse_last) RT_HIP_SLOT(r, launch_resolve(rad, npx, spp, nullptr, rgba, r.stream));
        RT_HIP_SLOT(r, hipMemcpyAsync(host_rgba8 + (size_t)y0 * w * 4, rgba, npx * 4, hipMemcpyDeviceToHost, r.stream));
        RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
        return RT_OK;
    });
}

int rt_counters_get(rt_ctx* c, int slot, rt_counters* out, int reset) {
    if (!c || !out) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    RT_HIP(c, hipDeviceSynchronize());
    unsigned long long v[kNumCounters];
    RT_HIP(c, hipMemcpy(v, r.d_counters, sizeof v, hipMemcpyDeviceToHost));
    out->rays = v[0]; out->box_tests = v[1]; out->nodes = v[2]; out->tri_tests = v[3]; out->sphere_tests = v[4];
    out->wave_max_nodes = v[5]; out->wave_max_tris = v[6]; out->waves = v[7];
    if (reset) RT_HIP(c, hipMemset(r.d_counters, 0, sizeof v));
    return RT_OK;
}

int rt_sync(rt_ctx* c, int slot) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, hipDeviceSynchronize());     // *_dev stages may have been enqueued on any stream
    return RT_OK;
}

// ---- host-pointer entry points (blocking; what the reference's single-threaded frame loop expects) ---
int rt_trace_primary(rt_ctx* c, uint32_t w, uint32_t h, rt_hit* host_out) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (!c->has_camera) return fail(c, RT_E_NO_CAMERA, "no camera set");
    if (!host_out || w == 0 || h == 0) return fail(c, RT_E_INVALID, "trace_primary: null output or empty image");
    if ((uint64_t)w * h > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "trace_primary: w*h exceeds 2^32-1");
    const int ns = (int)c->reps.size();
    // contiguous row blocks, multiples of 8 rows so that 8x8 tiles stay whole
    const uint32_t per = ((h + ns - 1) / ns + 7) / 8 * 8;
    const int mode = c->mode();
    return for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        const uint32_t y0 = std::min<uint64_t>((uint64_t)per * s, h), y1 = std::min<uint64_t>((uint64_t)y0 + per, h);
        if (y1 <= y0) return RT_OK;
        const uint32_t rows = y1 - y0;
        const size_t bytes = (size_t)rows * w * sizeof(rt_hit);
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        if (bytes > r.out_cap) {
            if (r.d_out) (void)hipFree(r.d_out);
            r.d_out = nullptr; r.out_cap = 0;
            RT_HIP_SLOT(r, hipMalloc(&r.d_out, bytes));
            r.out_cap = bytes;
        }
        RT_HIP_SLOT(r, launch_primary(c->dev_scene(r), c->cam, mode, false, w, h, y0, rows, rows, 1, 1, 0, 0, nullptr,
                                      static_cast<rt_hit*>(r.d_out), r.d_counters, r.stream));
        RT_HIP_SLOT(r, hipMemcpyAsync(host_out + (size_t)y0 * w, r.d_out, bytes, hipMemcpyDeviceToHost, r.stream));
        RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
        return RT_OK;
    });
}

static int trace_host(rt_ctx* c, const rt_ray* rays, uint64_t n, void* out, bool any_hit) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!rays || !out) return fail(c, RT_E_INVALID, "trace: null host pointer");
    const int ns = (int)c->reps.size();
    const uint64_t per = (n + ns - 1) / ns;
    const size_t osz = any_hit ? 1 : sizeof(rt_hit);
    return for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        const uint64_t b = std::min(per * s, n), e = std::min(b + per, n);
        if (e <= b) return RT_OK;
        const uint64_t chunk = std::min<uint64_t>(kChunkRays, e - b);
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        if (chunk * sizeof(rt_ray) > r.rays_cap) {
            if (r.d_rays) (void)hipFree(r.d_rays);
            r.d_rays = nullptr; r.rays_cap = 0;
            RT_HIP_SLOT(r, hipMalloc(&r.d_rays, chunk * sizeof(rt_ray)));
            r.rays_cap = chunk * sizeof(rt_ray);
        }
        if (chunk * osz > r.out_cap) {
            if (r.d_out) (void)hipFree(r.d_out);
            r.d_out = nullptr; r.out_cap = 0;
            RT_HIP_SLOT(r, hipMalloc(&r.d_out, chunk * osz));
            r.out_cap = chunk * osz;
        }
        for (uint64_t i = b; i < e; i += chunk) {
            const uint64_t m = std::min(chunk, e - i);
            RT_HIP_SLOT(r, hipMemcpyAsync(r.d_rays, rays + i, m * sizeof(rt_ray), hipMemcpyHostToDevice, r.stream));
            RT_HIP_SLOT(r, trace_rays(c, r, any_hit, false, static_cast<const rt_ray*>(r.d_rays), m,
                                      any_hit ? nullptr : static_cast<rt_hit*>(r.d_out),
                                      any_hit ? static_cast<uint8_t*>(r.d_out) : nullptr, r.stream));
            RT_HIP_SLOT(r, hipMemcpyAsync(static_cast<char*>(out) + i * osz, r.d_out, m * osz, hipMemcpyDeviceToHost, r.stream));
            RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
        }
        return RT_OK;
    });
}

int rt_trace(rt_ctx* c, const rt_ray* host_rays, uint64_t n, rt_hit* host_out) {
    return trace_host(c, host_rays, n, host_out, false);
}
int rt_occluded(rt_ctx* c, const rt_ray* host_rays, uint64_t n, uint8_t* host_out) {
    return trace_host(c, host_rays, n, host_out, true);
}

}  // extern "C"
