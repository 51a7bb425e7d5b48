// rt_api.cpp -- implementation of the C ABI declared in include/raytracer_amd.h.
//
// Host-side orchestration only: argument checks, BVH build + flatten, HBM replicas, chunked
// H2D/D2H for the host-pointer entry points, and kernel launches (kernels.hip).  Seams replaced in the
// reference are cited next to each declaration in the header.
#include "../../include/raytracer_amd.h"
#include "../host/rt_types.hpp"
#include "abi_guard.hpp"
#include "bvh.hpp"
#include "kernels.hpp"
#include "scene_host.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>        // types and prototypes only: the library is loaded with dlopen on first use (Rccl below)
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <array>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace rt;

namespace {

constexpr uint64_t kPieceRays = 1ull << 21;   // host-pointer entry points stream rays in 2 Mi-ray pieces (64 MB in, 32 MB out)
constexpr int kPipeRing = 3;                  // pieces in flight: staging copy | H2D + kernel | D2H + copy-out
constexpr uint32_t kSpillSlots = 4;           // strip-kernel scratch sets (cursors + spill columns) of one slot, used round-robin
constexpr int kEvBin = kSpillSlots, kEvPrim = kSpillSlots + 1, kNumScratchEvents = kSpillSlots + 2;

// Splits one large memcpy over a few persistent host threads: a single thread moves ~10 GB/s between pageable and pinned
// memory, a fifth of what the PCIe link takes.  Used by the host-pointer entry points only.
class CopyPool {
public:
    explicit CopyPool(unsigned helpers) {
        th_.reserve(helpers);
        try { for (unsigned k = 0; k < helpers; ++k) th_.emplace_back([this] { worker(); }); }
        catch (...) {}                                 // fewer helpers; the caller's thread always takes part
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void copy(void* dst, const void* src, size_t bytes) {
        const size_t parts = th_.size() + 1, min_part = 1u << 20;
        if (th_.empty() || bytes < 2 * min_part) { std::memcpy(dst, src, bytes); return; }
        const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = static_cast<char*>(dst); src_ = static_cast<const char*>(src); bytes_ = bytes; per_ = per;
            next_ = 1; todo_ = (int)th_.size(); ++gen_;
        }
        cv_.notify_all();
        std::memcpy(dst, src, std::min(per, bytes));                                       // part 0 on this thread
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [&] { return todo_ == 0; });
    }
private:
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            size_t part;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return quit_ || gen_ != seen; });
                if (quit_) return;
                seen = gen_;
                part = next_++;
            }
            const size_t b = part * per_;
            if (b < bytes_) std::memcpy(dst_ + b, src_ + b, std::min(per_, bytes_ - b));
            { std::lock_guard<std::mutex> g(mu_); if (--todo_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    char* dst_ = nullptr; const char* src_ = nullptr;
    size_t bytes_ = 0, per_ = 0, next_ = 0;
    int todo_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

// Staging of the host-pointer entry points (rt_trace / rt_occluded): pinned host buffers and device buffers for
// kPipeRing pieces, four streams (inbound copies, outbound copies, kernels of even / odd pieces), three events per ring slot.
struct HostPipe {
    void* pin_in[kPipeRing] = {};  void* pin_out[kPipeRing] = {};
    void* d_in[kPipeRing] = {};    void* d_out[kPipeRing] = {};
    hipEvent_t done[kPipeRing] = {};
    hipEvent_t got[kPipeRing] = {}, ran[kPipeRing] = {};     // piece is on the device / its kernels have run
    hipStream_t st[4] = {};                                  // inbound copies, outbound copies, kernels of even / odd pieces
    uint64_t cap_rays = 0;
    bool ready = false;
    std::unique_ptr<CopyPool> pool;            // staging copies (created on first use)
};

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
    float4* d_wide = nullptr;         // fp32 copy of the BVH4q nodes for the packet traversal (null when not built)
    // scratch for rt_render
    void* d_render = nullptr; size_t render_cap = 0;
    unsigned long long* d_counters = nullptr;      // kCountersPad words of counters, then kSpillSlots cursor sets
    // scratch for rt_trace_primary
    void* d_out = nullptr;   size_t out_cap = 0;
    // scratch for ray binning (permutation + histograms, uint32 words)
    uint32_t* d_bin = nullptr; uint64_t bin_cap = 0;
    // spill area of the strip kernel's capped stack: kSpillSlots buffers used round-robin with the cursor sets
    int* d_spill = nullptr; uint64_t spill_cap = 0; uint32_t scratch_seq = 0;
    uint64_t spill_words = 0;           // size of one spill set at the last launch
    // primary rays / hits of the ray-buffer primary path when the caller does not ask for them
    void* d_prim = nullptr; size_t prim_cap = 0;
    // Launches of one slot share the scratch above.  Stream order protects launches on one stream; for launches on
    // different streams every scratch piece has an event: recorded after its last user, waited on (on the device)
    // by its next user.  Any number of launches may therefore be in flight on any streams.
    hipEvent_t sc_ev[kNumScratchEvents] = {};
    bool sc_used[kNumScratchEvents] = {};
    void* d_noalloc = nullptr;         // this device's copy of the static scratch array of rt_trace_dev_noalloc
    // multi-slot frames (rt_render / rt_trace_primary): slot 0 keeps the shares of all slots + the frame in image order here
    void* d_gather = nullptr; size_t gather_cap = 0;
    hipEvent_t ev_share = nullptr;     // "this slot's share is complete" (recorded on r.stream, waited for by slot 0's stream)
    HostPipe pipe;
    std::string err;
};

hipError_t scratch_acquire(Replica& r, int k, hipStream_t st) {
    return r.sc_used[k] ? hipStreamWaitEvent(st, r.sc_ev[k], 0) : hipSuccess;
}
hipError_t scratch_release(Replica& r, int k, hipStream_t st) {
    r.sc_used[k] = true;
    return hipEventRecord(r.sc_ev[k], st);
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// RCCL for the in-process frame gather (north_star: "framebuffer gathered over xGMI with RCCL").  Loaded on first use, so that
// one-GPU users neither link nor load it; one communicator per slot, rank = slot, made by ncclCommInitAll.
struct Rccl {
    void* so = nullptr;
    bool tried = false;
    std::string why;                                   // why it cannot be used (after a failed attempt)
    bool given_up = false;                             // RT_OPT_FRAME_EXCHANGE 0 (auto) fell back to peer copies for good
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<ncclComm_t> comms;
    bool load() {
        if (tried) return so != nullptr;
        tried = true;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
            if ((so = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!so) { const char* e = dlerror(); why = std::string("librccl.so.1 not loadable: ") + (e ? e : "?"); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(so, n); if (!p && why.empty()) why = std::string("RCCL symbol missing: ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!why.empty()) { dlclose(so); so = nullptr; return false; }
        return true;
    }
    void destroy() {
        for (ncclComm_t cm : comms) if (cm && CommDestroy) (void)CommDestroy(cm);
        comms.clear();
        // the library stays loaded: unloading RCCL while its proxy threads wind down is not worth the risk
    }
};

}  // namespace

struct rt_ctx {
    std::vector<Replica> reps;
    std::vector<void*> host_blocks;   // rt_host_alloc
    std::string err;
    // options
    int traversal = 0;          // 0 auto, 1 brute force, 2 BVH2 nodes, 3 BVH4q nodes (RT_OPT_TRAVERSAL)
    int bvh_threads = 0;
    int counters = 0;
    int pad_log2 = 15;          // box padding = max|coordinate| * 2^-pad_log2; the per-ray pad uses the same factor
    int ray_pad = 1;            // per-ray box pad (kernels.hip make_slab): 1 on, 0 off (measurement only: far origins then lose hits)
    int max_leaf = 4;
    int sah_bins = 16;          // SAH bins per axis of the builder
    int ray_packets = 0;        // closest-hit ray buffers walk as packets of 64 consecutive rays (RT_OPT_RAY_PACKETS)
    int sah_trav_x100 = 60;     // SAH cost of an interior step, in hundredths of a triangle test (profiles/r05w_sah_trav.txt)
    uint32_t strip = 256;       // rays per chunk drawn by a wave of the ray-buffer kernel (0 = plain one-ray-per-lane kernel)
    int refill = 16 | (12 << 8); // (= kDefaultRefill) low byte: idle lanes that trigger a refill; high: leaf-vote threshold (profiles/r02v_sweep_refill.txt)
    int lds_stack = 20;         // (= kDefaultLdsStack) ray-buffer kernel, BVH4q: stack entries per lane kept in LDS, deeper ones spill to HBM (0 = all in LDS)
    int primary_path = 0;       // primary rays: 0 auto, 1 tile kernel (k_primary), 2 ray generation + ray-buffer kernel
    int binning = 2;            // ray binning before the ray-buffer kernel: 0 off, 1 on, 2 auto (structure larger than the L2s)
    int ray_queue = 2;          // ray-buffer kernel: staged ray queue in LDS: 0 off, 1 on, 2 auto = launches that are not binned (RT_OPT_RAY_QUEUE)
    int fused_paths = 1;        // rt_render on brute-force scenes: 1 = the whole path loop in one kernel, 0 = the staged pipeline (RT_OPT_FUSED_PATHS)
    int frame_exchange = 0;     // multi-slot frames: 0 auto, 1 RCCL send/recv, 2 peer copies (RT_OPT_FRAME_EXCHANGE)
    Rccl rccl;
    int last_exchange = 0;      // what the last multi-slot frame used (1 RCCL, 2 peer copies; 0 = none yet)
    BinGrid bin_grid{};         // 16^3 grid over the triangle bounds of the uploaded scene
    // scene
    bool has_scene = false, has_camera = false;
    bool has_wide = false;      // the fp32 node copy for packet traversal was uploaded with the scene
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
        // packet traversal of the tile kernel: RT_OPT_PRIMARY 3 forces it, 0 (auto) takes it when the wide nodes exist
        s.wide = (primary_path == 0 || primary_path == 3) ? reinterpret_cast<const uint32_t*>(r.d_wide) : nullptr;
        s.n_nodes = stats.n_nodes; s.n_tris = stats.n_tris; s.n_spheres = stats.n_spheres;
        s.stack_depth = stack_need;
        s.ray_pad = ray_pad ? std::ldexp(1.0f, -pad_log2) : 0.0f;
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
    if (r.d_wide) (void)hipFree(r.d_wide);
    r.d_wide = nullptr;
    r.d_nodes = r.d_tris = r.d_normals = r.d_spheres = r.d_materials = r.d_sphere_mat = r.d_lights = nullptr;
}

// Closest-hit / any-hit over a ray buffer in HBM.  With binning the rays are handed out grouped by origin cell
// (launch_bin), which keeps a wave's node fetches close together when the structure does not fit the L2s.
constexpr uint64_t kBinMinRays = 1ull << 18;
constexpr uint64_t kBinMinSceneBytes = 48ull << 20;
constexpr int kDefaultRefill = 16 | (12 << 8);   // RT_OPT_REFILL as rt_create leaves it
constexpr int kDefaultLdsStack = 20;      // RT_OPT_LDS_STACK as rt_create leaves it
constexpr int kQueueLdsStack = 12;        // stack entries per lane in LDS when the staged ray queue takes its 8 KB per block
constexpr uint32_t kBinMinRun = 128;      // mean rays per non-empty (64 Ki-ray block, cell) pair above which a buffer counts as coherent
bool wants_binning(const rt_ctx* c, int mode, uint64_t n) {
    const uint64_t scene_bytes = (uint64_t)c->stats.n_nodes * 64 + (uint64_t)c->stats.n_tris * 48;
    return mode != kBrute && c->strip >= 64 && n >= kBinMinRays && n < (1ull << 32) &&
           (c->binning == 1 || (c->binning == 2 && scene_bytes > kBinMinSceneBytes));
}
hipError_t trace_rays(rt_ctx* c, Replica& r, bool any_hit, bool count, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits,
                      uint8_t* d_occ, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const int mode = c->mode();
    if (c->ray_packets && !any_hit && mode == kBvh4q && c->has_wide && c->stack_need + 2 <= 64) {
        DevScene ds = c->dev_scene(r);
        ds.wide = reinterpret_cast<const uint32_t*>(r.d_wide);
        return launch_trace_packets(ds, count, d_rays, n, nullptr, d_hits, r.d_counters, st);
    }
    const bool strip_kernel = mode != kBrute && c->strip >= 64;
    const bool bin = wants_binning(c, mode, n);
    const uint32_t* perm = nullptr;
    hipError_t e;
    if (bin) {
        const bool two_level = n >= kBinTwoLevelMin;
        const uint64_t words = bin_scratch_words(n, kBinBlocksMax, two_level);
        if (words > r.bin_cap) {
            if (r.d_bin) (void)hipFree(r.d_bin);          // hipFree waits for the device: no launch still reads it
            r.d_bin = nullptr; r.bin_cap = 0;
            e = hipMalloc(reinterpret_cast<void**>(&r.d_bin), words * sizeof(uint32_t));
            if (e != hipSuccess) return e;
            r.bin_cap = words;
        }
        if ((e = scratch_acquire(r, kEvBin, st)) != hipSuccess) return e;
        // auto: permute only buffers that are not grouped by origin already (camera-ordered bounce rays are)
        e = launch_bin(d_rays, n, c->bin_grid, r.d_bin, c->binning == 1 ? 0u : kBinMinRun, st, kBinBlocksMax, two_level);
        if (e != hipSuccess) return e;
        perm = r.d_bin;
    }
    StripScratch sc;
    int set = -1;
    int refill = c->refill;
    if (strip_kernel) {
        set = (int)(r.scratch_seq++ % kSpillSlots);
        sc.work = r.d_counters + kCountersPad + (size_t)set * kWorkWords;
        sc.lds_cap = c->lds_stack;
        // Staged ray queue (k_trace_strip<QUEUE>): refills cost a few LDS reads, so they run from 6 idle lanes on and the triangle
        // block whenever a leaf is parked.  Pays when the structure is served by the L2s (bench frame: bounce launch -3 %) and
        // for binned launches below 2^25 rays (10 M-triangle soup: 2 Mi rays +8 %, 8 Mi +4 %, 16 Mi +3 %); from 2^25 rays on a
        // binned launch loses with it (-1 .. -7 %): profiles/r05_ray_queue.txt, r05y_c4_cap.txt.
        // The queue takes 8 KB of LDS per block.  Structures in the L2s: 12 stack entries in LDS instead of 20 cost nothing there
        // and keep 7 blocks per CU.  Binned launches below 2^25 rays want FEWER resident waves anyway: the rays in flight at
        // once are a large part of the buffer, so the region of the scene they touch outgrows the caches (10 M-triangle soup,
        // 8 Mi rays, round 2's tree: L2 hit 52 %, HBM 5.8 TB/s; profiles/r04j_c4_8mi_summary.md) -- 32 entries in LDS, 4 blocks
        // per CU with the queue (r05y_c4_cap.txt; 2 Mi rays +12 %, 4-16 Mi +3-5 % before the queue, r04k_c4_cap_sweep.txt).
        // Explicit RT_OPT_LDS_STACK / RT_OPT_REFILL values hold.
        sc.stream_nt = bin;                   // structure larger than the L2s: the ray / hit records stream past them (kernels.hip load_stream)
        const bool small_binned = bin && n < (1ull << 25);
        sc.queue = mode == kBvh4q && n < (1ull << 32) && (c->ray_queue == 1 || (c->ray_queue == 2 && (!bin || small_binned)));
        if (c->lds_stack == kDefaultLdsStack) {
            if (small_binned) sc.lds_cap = 32;
            else if (sc.queue) sc.lds_cap = kQueueLdsStack;
        }
        if (sc.queue && c->refill == kDefaultRefill) refill = 6 | (1 << 8);
        if (c->refill == kDefaultRefill) sc.refill_perm = 32 | (8 << 8);      // the default thresholds: large permuted buffers get their own (kernels.hip launch_trace); an explicit RT_OPT_REFILL holds for all
        if (mode == kBvh4q) {
            // the spill columns are sized for the smallest LDS part any launch of this context may choose
            const int cap_lo = c->lds_stack == kDefaultLdsStack && c->ray_queue ? kQueueLdsStack : c->lds_stack;
            const uint64_t words = cap_lo ? spill_words(c->stack_need, cap_lo, device_cus()) : 0;
            if (words != r.spill_words && r.spill_words) {
                // an option changed the size of a spill set: the sets of launches still running would overlap the new ones
                if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
            }
            r.spill_words = words;
            if (words) {
                if (words * kSpillSlots > r.spill_cap) {
                    if (r.d_spill) (void)hipFree(r.d_spill);
                    r.d_spill = nullptr; r.spill_cap = 0;
                    e = hipMalloc(reinterpret_cast<void**>(&r.d_spill), words * kSpillSlots * sizeof(int));
                    if (e != hipSuccess) return e;
                    r.spill_cap = words * kSpillSlots;
                }
                sc.spill = r.d_spill + (uint64_t)set * words;
            }
        }
        if ((e = scratch_acquire(r, set, st)) != hipSuccess) return e;
    }
    e = launch_trace(c->dev_scene(r), mode, any_hit, count, d_rays, n, d_hits, d_occ, r.d_counters, st, c->strip, refill, perm,
                     perm ? bin_flag(perm, n) : nullptr, sc);
    if (e != hipSuccess) return e;
    if (set >= 0 && (e = scratch_release(r, set, st)) != hipSuccess) return e;
    if (bin && (e = scratch_release(r, kEvBin, st)) != hipSuccess) return e;
    return hipSuccess;
}

// Which kernel takes the primary rays.  With the wide node copy uploaded (BVH4q trees, unless RT_OPT_PRIMARY is 1 or 2)
// it is the packet form of the tile kernel: ahead of both others on every scene measured, tessellated surfaces up to 8.4 M
// triangles (2.2 ms against 6.2 / 5.1 ms) and random soups up to 10 M (3.6 ms against 13.0 / 6.1 ms; scripts/primary_threshold*.py,
// profiles/r03b_*).  Without it (BVH2 trees, RT_OPT_PRIMARY 1) the round-1 rule holds: a "hard" tree, recognised by its SAH
// cost, sends its primary rays through ray generation + the ray-buffer kernel, because the one-ray-per-lane tile kernel walks
// every ray of a wave to the end (10 M-triangle soup: 50 ms against 17 ms); crossover at SAH cost ~250.  Same rays and hits
// in every case.
constexpr float kPrimaryBufferSah = 250.0f;
bool primary_via_buffer(const rt_ctx* c) {
    const int mode = c->mode();
    if (mode == kBrute || c->strip < 64 || c->counters || c->primary_path == 1 || c->primary_path == 3) return false;
    if (c->primary_path == 2) return true;
    if (mode == kBvh4q && c->has_wide && c->stack_need + 2 <= 64) return false;      // packet traversal
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
        // nothing may escape a std::thread; a thread that cannot be started runs its slot on this one
        auto run = [&](int s) { try { rc[s] = fn(s); } catch (...) { rc[s] = abi_exception(&c->reps[s].err); } };
        std::vector<std::thread> th;
        th.reserve(n);
        for (int s = 0; s < n; ++s) {
            try { th.emplace_back(run, s); } catch (...) { run(s); }
        }
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

// Handler of every entry point's function-try-block: nothing is thrown across the ABI (abi_guard.hpp).
#define RT_ABI_CATCH(ctx) catch (...) { rt_ctx* c__ = const_cast<rt_ctx*>(static_cast<const rt_ctx*>(ctx)); return abi_exception(c__ ? &c__->err : nullptr); }

// ---- frames of contexts with several slots ---------------------------------------------------------------------------------
// The image rows are dealt to the slots in interleaved stripes of kFrameStripe rows (cost is spatially uneven -- floor against
// geometry -- so contiguous blocks would last as long as the slowest one).  Every slot fills a "share": its
// stripes, compact, padded to the same number of rows on every slot.  The shares meet on slot 0's device, are put in image
// order there (launch_gather_rows) and leave in one device-to-host copy.
constexpr uint32_t kFrameStripe = 8;                 // rows per stripe: the tile height of the primary kernels
struct FrameSplit {
    uint32_t n_slots, share_rows;                    // rows of every slot's share (whole stripes, the last ones may be padding)
    uint32_t y0(int s) const { return (uint32_t)s * kFrameStripe; }
};
FrameSplit frame_split(uint32_t h, int n_slots) {
    const uint32_t n_stripes = (h + kFrameStripe - 1) / kFrameStripe;
    return FrameSplit{ (uint32_t)n_slots, ((n_stripes + n_slots - 1) / n_slots) * kFrameStripe };
}

hipError_t grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return hipSuccess;
    if (*p) (void)hipFree(*p);                       // hipFree waits for the device
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(p, need);
    if (e == hipSuccess) *cap = need;
    return e;
}

// slot 0's staging for a frame of h rows x row_bytes: [n_slots shares][frame in image order]; slot 0 writes its own share in
// place.  (A context of one slot keeps room for a second copy of its share: the target of the RCCL self-send, frame_gather.)
int frame_stage(rt_ctx* c, const FrameSplit& sp, uint32_t h, size_t row_bytes, char** shares, char** frame) {
    Replica& r0 = c->reps[0];
    const size_t share_bytes = (size_t)sp.share_rows * row_bytes;
    const size_t blocks = sp.n_slots == 1 ? 2 : sp.n_slots;
    RT_HIP(c, hipSetDevice(r0.device));
    RT_HIP(c, grow(&r0.d_gather, &r0.gather_cap, share_bytes * blocks + (size_t)h * row_bytes));
    *shares = static_cast<char*>(r0.d_gather);
    *frame = *shares + share_bytes * blocks;
    return RT_OK;
}

bool distinct_devices(const rt_ctx* c) {
    for (size_t a = 0; a < c->reps.size(); ++a)
        for (size_t b = a + 1; b < c->reps.size(); ++b)
            if (c->reps[a].device == c->reps[b].device) return false;
    return true;
}

int rccl_fail(rt_ctx* c, ncclResult_t e, const char* what) {
    return fail(c, RT_E_DEVICE, std::string(what) + ": " + (c->rccl.GetErrorString ? c->rccl.GetErrorString(e) : "RCCL error"));
}
#define RT_RCCL(ctx, call)                                             \
    do {                                                               \
        ncclResult_t e__ = (call);                                     \
        if (e__ != ncclSuccess) return rccl_fail((ctx), e__, #call);   \
    } while (0)

// one communicator per slot (rank = slot) over the context's devices
int rccl_prepare(rt_ctx* c) {
    Rccl& rc = c->rccl;
    if (!rc.comms.empty()) return RT_OK;
    if (!distinct_devices(c)) return fail(c, RT_E_INVALID, "RT_OPT_FRAME_EXCHANGE 1 (RCCL) needs distinct device ordinals: a communicator cannot hold one GPU twice");
    if (!rc.load()) return fail(c, RT_E_DEVICE, rc.why);
    std::vector<int> devs;
    for (const Replica& r : c->reps) devs.push_back(r.device);
    std::vector<ncclComm_t> comms(devs.size(), nullptr);
    RT_RCCL(c, rc.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
    rc.comms = comms;
    return RT_OK;
}

// The slots' shares -- d_share[s] on slot s's device, complete in the order of that slot's stream; d_share[0] is the first block
// of slot 0's staging -- to slot 0, into image order, to host_out.  Blocking, like the entry points that call it.
int frame_gather(rt_ctx* c, const FrameSplit& sp, uint32_t h, size_t row_bytes, void* const* d_share, char* shares, char* frame,
                 void* host_out) {
    const int ns = (int)sp.n_slots;
    const size_t share_bytes = (size_t)sp.share_rows * row_bytes;
    Replica& r0 = c->reps[0];
    bool use_rccl = c->frame_exchange == 1;
    if (c->frame_exchange == 0 && ns > 1 && !c->rccl.given_up && distinct_devices(c)) {
        use_rccl = rccl_prepare(c) == RT_OK;        // auto: RCCL when it is there, else peer copies (the reason stays in rccl.why)
        if (!use_rccl) { c->rccl.given_up = true; if (c->rccl.why.empty()) c->rccl.why = c->err; }     // not tried again frame after frame
        c->err.clear();
    }
    const char* src = shares;                       // where launch_gather_rows reads the shares
    if (use_rccl) {
        int rcode = rccl_prepare(c);
        if (rcode != RT_OK) return rcode;
        Rccl& rc = c->rccl;
        // Grouped point-to-point: slot s sends its share once, over its own xGMI link to slot 0; no ring, no reduction.
        RT_RCCL(c, rc.GroupStart());
        struct GroupGuard {                         // an error between GroupStart and GroupEnd must not leave RCCL in group mode
            Rccl& rc; bool open = true;
            ~GroupGuard() { if (open) (void)rc.GroupEnd(); }
        } group{ rc };
        for (int s = 1; s < ns; ++s) {
            RT_RCCL(c, rc.Send(d_share[s], share_bytes, ncclUint8, 0, rc.comms[s], c->reps[s].stream));
            RT_RCCL(c, rc.Recv(shares + (size_t)s * share_bytes, share_bytes, ncclUint8, s, rc.comms[0], r0.stream));
        }
        if (ns == 1) {
            // A context of ONE slot that asks for RCCL sends its share to itself, into the second block of the staging, and the
            // frame is assembled from that copy: the whole plumbing -- dlopen, communicator, send / recv on the slot's stream,
            // stream order against the kernels -- then runs on a one-GPU box too (tests/).
            RT_RCCL(c, rc.Send(shares, share_bytes, ncclUint8, 0, rc.comms[0], r0.stream));
            RT_RCCL(c, rc.Recv(shares + share_bytes, share_bytes, ncclUint8, 0, rc.comms[0], r0.stream));
            src = shares + share_bytes;
        }
        group.open = false;
        RT_RCCL(c, rc.GroupEnd());
        c->last_exchange = 1;
    } else {
        // peer copies on slot 0's stream, each behind the event that marks its source share complete
        for (int s = 1; s < ns; ++s) {
            Replica& r = c->reps[s];
            RT_HIP(c, hipSetDevice(r.device));
            RT_HIP(c, hipEventRecord(r.ev_share, r.stream));
            RT_HIP(c, hipSetDevice(r0.device));
            RT_HIP(c, hipStreamWaitEvent(r0.stream, r.ev_share, 0));
            char* dst = shares + (size_t)s * share_bytes;
            if (r.device == r0.device) RT_HIP(c, hipMemcpyAsync(dst, d_share[s], share_bytes, hipMemcpyDeviceToDevice, r0.stream));
            else                       RT_HIP(c, hipMemcpyPeerAsync(dst, r0.device, d_share[s], r.device, share_bytes, r0.stream));
        }
        c->last_exchange = 2;
    }
    RT_HIP(c, hipSetDevice(r0.device));
    RT_HIP(c, launch_gather_rows(reinterpret_cast<const uint32_t*>(src), reinterpret_cast<uint32_t*>(frame), h, (uint32_t)(row_bytes / 4),
                                 kFrameStripe, sp.n_slots, sp.share_rows, r0.stream));
    RT_HIP(c, hipMemcpyAsync(host_out, frame, (size_t)h * row_bytes, hipMemcpyDeviceToHost, r0.stream));
    RT_HIP(c, hipStreamSynchronize(r0.stream));
    for (int s = 1; s < ns; ++s) {                  // the senders' buffers are free again, and their errors surface here
        RT_HIP(c, hipSetDevice(c->reps[s].device));
        RT_HIP(c, hipStreamSynchronize(c->reps[s].stream));
    }
    return RT_OK;
}

void free_pipe(Replica& r) {
    HostPipe& p = r.pipe;
    p.pool.reset();
    for (int k = 0; k < kPipeRing; ++k) {
        if (p.pin_in[k]) (void)hipHostFree(p.pin_in[k]);
        if (p.pin_out[k]) (void)hipHostFree(p.pin_out[k]);
        if (p.d_in[k]) (void)hipFree(p.d_in[k]);
        if (p.d_out[k]) (void)hipFree(p.d_out[k]);
        if (p.done[k]) (void)hipEventDestroy(p.done[k]);
        if (p.got[k]) (void)hipEventDestroy(p.got[k]);
        if (p.ran[k]) (void)hipEventDestroy(p.ran[k]);
        p.pin_in[k] = p.pin_out[k] = p.d_in[k] = p.d_out[k] = nullptr; p.done[k] = p.got[k] = p.ran[k] = nullptr;
    }
    for (int k = 0; k < 4; ++k) { if (p.st[k]) (void)hipStreamDestroy(p.st[k]); p.st[k] = nullptr; }
    p.cap_rays = 0; p.ready = false;
}

}  // namespace

extern "C" {

int rt_create(rt_ctx** out, const int* device_ids, int n_devices) try {
    if (!out) return RT_E_INVALID;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return RT_E_INVALID;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return RT_E_DEVICE;
    std::unique_ptr<rt_ctx> c(new rt_ctx());
    c->reps.resize(n_devices);
    const size_t counter_words = kCountersPad + (size_t)kSpillSlots * kWorkWords;
    static_assert(kWorkSlots == (int)kSpillSlots, "one cursor set per scratch slot");
    for (int i = 0; i < n_devices; ++i) {
        int d = device_ids ? device_ids[i] : i;
        if (d < 0 || d >= ndev) { rt_destroy(c.release()); return RT_E_INVALID; }
        Replica& r = c->reps[i];
        r.device = d;
        bool ok = hipSetDevice(d) == hipSuccess && hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipMalloc(&r.d_counters, counter_words * sizeof(unsigned long long)) == hipSuccess &&
                  hipMemset(r.d_counters, 0, counter_words * sizeof(unsigned long long)) == hipSuccess &&
                  noalloc_init(&r.d_noalloc) == hipSuccess &&        // this device's address of the static scratch array
                  hipEventCreateWithFlags(&r.ev_share, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; ok && k < kNumScratchEvents; ++k) ok = hipEventCreateWithFlags(&r.sc_ev[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) { rt_destroy(c.release()); return RT_E_DEVICE; }
    }
    *out = c.release();
    return RT_OK;
} catch (...) {
    return abi_exception(nullptr);
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    for (Replica& r : c->reps) { (void)hipSetDevice(r.device); (void)hipDeviceSynchronize(); }   // *_dev launches may still be running on the caller's streams
    c->rccl.destroy();
    for (Replica& r : c->reps) {
        (void)hipSetDevice(r.device);
        free_scene(r);
        free_pipe(r);
        if (r.d_counters) (void)hipFree(r.d_counters);
        if (r.d_out) (void)hipFree(r.d_out);
        if (r.d_bin) (void)hipFree(r.d_bin);
        if (r.d_spill) (void)hipFree(r.d_spill);
        if (r.d_prim) (void)hipFree(r.d_prim);
        if (r.d_render) (void)hipFree(r.d_render);
        if (r.d_gather) (void)hipFree(r.d_gather);
        if (r.ev_share) (void)hipEventDestroy(r.ev_share);
        for (hipEvent_t ev : r.sc_ev) if (ev) (void)hipEventDestroy(ev);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    for (void* p : c->host_blocks) (void)hipHostFree(p);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rt_num_slots(const rt_ctx* c) { return c ? (int)c->reps.size() : 0; }

int rt_frame_exchange_used(const rt_ctx* c) { return c ? c->last_exchange : 0; }

int rt_set_option(rt_ctx* c, int key, int64_t v) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    switch (key) {
        case RT_OPT_TRAVERSAL:   if (v < 0 || v > 3) return fail(c, RT_E_INVALID, "RT_OPT_TRAVERSAL: 0 auto, 1 brute, 2 BVH2, 3 BVH4q"); c->traversal = (int)v; return RT_OK;
        case RT_OPT_BVH_THREADS: if (v < 0 || v > 256) return fail(c, RT_E_INVALID, "RT_OPT_BVH_THREADS out of range"); c->bvh_threads = (int)v; return RT_OK;
        case RT_OPT_COUNTERS:    c->counters = v ? 1 : 0; return RT_OK;
        case RT_OPT_LDS_STACK:   if (v != 0 && (v < 2 || v > 64)) return fail(c, RT_E_INVALID, "RT_OPT_LDS_STACK: 0 or 2..64 entries"); c->lds_stack = (int)v; return RT_OK;
        case RT_OPT_PRIMARY:     if (v < 0 || v > 3) return fail(c, RT_E_INVALID, "RT_OPT_PRIMARY: 0 auto, 1 tile kernel (one ray per lane), 2 ray-buffer kernel, 3 tile kernel (packet traversal)"); c->primary_path = (int)v; return RT_OK;
        case RT_OPT_RAY_BINNING: if (v < 0 || v > 2) return fail(c, RT_E_INVALID, "RT_OPT_RAY_BINNING: 0 off, 1 on, 2 auto"); c->binning = (int)v; return RT_OK;
        // 2^-4 .. 2^-20: below 2^-20 (16 ulp) the conservative-traversal argument of DESIGN.md no longer holds
        case RT_OPT_PAD_LOG2:    if (v < 4 || v > 20) return fail(c, RT_E_INVALID, "RT_OPT_PAD_LOG2 in [4,20]"); c->pad_log2 = (int)v; return RT_OK;
        case RT_OPT_MAX_LEAF:    if (v < 1 || v > 8) return fail(c, RT_E_INVALID, "RT_OPT_MAX_LEAF in [1,8]"); c->max_leaf = (int)v; return RT_OK;
        case RT_OPT_STRIP:       if (v != 0 && (v < 64 || v > (1 << 24))) return fail(c, RT_E_INVALID, "RT_OPT_STRIP: 0 or 64..2^24"); c->strip = (uint32_t)v; return RT_OK;
        case RT_OPT_REFILL:      if ((v & 0xff) < 1 || (v & 0xff) > 64 || (v >> 8) > 64) return fail(c, RT_E_INVALID, "RT_OPT_REFILL: low byte 1..64 idle lanes, bits 8.. = leaf-vote threshold 0..64"); c->refill = (int)v; return RT_OK;
        case RT_OPT_SAH_BINS:    if (v < 4 || v > 32) return fail(c, RT_E_INVALID, "RT_OPT_SAH_BINS in [4,32]"); c->sah_bins = (int)v; return RT_OK;
        case RT_OPT_SAH_TRAV:    if (v < 1 || v > 10000) return fail(c, RT_E_INVALID, "RT_OPT_SAH_TRAV in [1,10000] (hundredths)"); c->sah_trav_x100 = (int)v; return RT_OK;
        case RT_OPT_RAY_PACKETS: if (v < 0 || v > 1) return fail(c, RT_E_INVALID, "RT_OPT_RAY_PACKETS: 0 off, 1 on"); c->ray_packets = (int)v; return RT_OK;
        case RT_OPT_RAY_QUEUE:   if (v < 0 || v > 2) return fail(c, RT_E_INVALID, "RT_OPT_RAY_QUEUE: 0 off, 1 on, 2 auto"); c->ray_queue = (int)v; return RT_OK;
        case RT_OPT_FUSED_PATHS: if (v < 0 || v > 1) return fail(c, RT_E_INVALID, "RT_OPT_FUSED_PATHS: 0 staged, 1 fused"); c->fused_paths = (int)v; return RT_OK;
        case RT_OPT_FRAME_EXCHANGE: if (v < 0 || v > 2) return fail(c, RT_E_INVALID, "RT_OPT_FRAME_EXCHANGE: 0 auto, 1 RCCL, 2 peer copies"); c->frame_exchange = (int)v; return RT_OK;
        case RT_OPT_RAY_PAD:     if (v < 0 || v > 1) return fail(c, RT_E_INVALID, "RT_OPT_RAY_PAD: 0 off, 1 on"); c->ray_pad = (int)v; return RT_OK;
        default: return fail(c, RT_E_INVALID, "unknown option key");
    }
} RT_ABI_CATCH(c)

int rt_scene_upload(rt_ctx* c, const void* tris, uint32_t n_tris, uint32_t tri_stride, const void* spheres,
                    uint32_t n_spheres, uint32_t sphere_stride, const void* lights, uint32_t n_lights,
                    uint32_t light_stride) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (n_tris && (!tris || tri_stride < sizeof(Triangle) || (tri_stride & 3))) return fail(c, RT_E_INVALID, "triangles: null pointer or stride < 144 / not a multiple of 4");
    if (n_spheres && (!spheres || sphere_stride < sizeof(Sphere) || (sphere_stride & 3))) return fail(c, RT_E_INVALID, "spheres: null pointer or stride < 64 / not a multiple of 4");
    if (n_lights && (!lights || light_stride < sizeof(Light) || (light_stride & 3))) return fail(c, RT_E_INVALID, "lights: null pointer or stride < 32 / not a multiple of 4");
    if (n_tris >= (1u << 28)) return fail(c, RT_E_TOO_LARGE, "more than 2^28-1 triangles");
    if (n_spheres >= (1u << 31)) return fail(c, RT_E_TOO_LARGE, "more than 2^31-1 spheres");

    const double t0 = now_ms();
    // host half (scene_host.cpp): bounds, BVH build, flatten, device records.  Traversal option 2 asks for the plain
    // BVH2 layout; everything else gets the 4-wide quantised one (faster on every measured config, scripts/cmp_layouts.py)
    HostSceneParams hp;
    hp.layout = (c->traversal == 2) ? kBvh2 : kBvh4q;
    hp.max_leaf = c->max_leaf; hp.pad_log2 = c->pad_log2; hp.threads = c->bvh_threads; hp.max_depth = kStackDepth - 2;
    hp.wide = c->primary_path != 1 && c->primary_path != 2;      // the tile kernel's packet traversal wants fp32 nodes
    hp.bins = c->sah_bins; hp.cost_trav = 0.01f * (float)c->sah_trav_x100;
    HostScene hs;
    {
        std::string err;
        const int rc = prepare_scene(tris, n_tris, tri_stride, spheres, n_spheres, sphere_stride, lights, n_lights, light_stride, hp, hs, err);
        if (rc != RT_OK) return fail(c, rc, err);
    }
    if (hs.layout == kBvh4q && hs.n_nodes >= (1u << 26))      // the BVH4q step addresses nodes by a 32-bit byte offset
        return fail(c, RT_E_TOO_LARGE, "more than 2^26-1 BVH4q nodes (about 200 M triangles); use RT_OPT_TRAVERSAL 2");
    if (hs.n_nodes >= (1u << 25)) { std::vector<uint32_t>().swap(hs.wide); }   // the packet walk addresses the 128-byte copy by a 32-bit byte offset: larger trees go without it
    const double t1 = now_ms();

    // replicate
    c->has_scene = false;
    for (Replica& r : c->reps) {
        RT_HIP(c, hipSetDevice(r.device));
        RT_HIP(c, hipDeviceSynchronize());       // launches on any stream may still read the old scene
        free_scene(r);
        auto up = [&](float4** dst, const void* src, size_t src_bytes) -> hipError_t {
            const size_t bytes = std::max<size_t>(src_bytes, 64);
            hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), bytes);
            if (e != hipSuccess) return e;
            if (src_bytes < bytes) { e = hipMemset(*dst, 0, bytes); if (e != hipSuccess) return e; }
            return src_bytes ? hipMemcpy(*dst, src, src_bytes, hipMemcpyHostToDevice) : hipSuccess;
        };
        RT_HIP(c, up(&r.d_nodes, hs.nodes.data(), hs.nodes.size() * 4));
        RT_HIP(c, up(&r.d_tris, hs.tri48.data(), hs.tri48.size() * 4));
        RT_HIP(c, up(&r.d_normals, hs.nrm48.data(), hs.nrm48.size() * 4));
        RT_HIP(c, up(&r.d_spheres, hs.sph.data(), hs.sph.size() * 4));
        RT_HIP(c, up(&r.d_materials, hs.mats.data(), hs.mats.size() * 4));
        RT_HIP(c, up(&r.d_sphere_mat, hs.sphere_mat.data(), hs.sphere_mat.size() * 4));
        RT_HIP(c, up(&r.d_lights, hs.lights8.data(), hs.lights8.size() * 4));
        if (!hs.wide.empty()) RT_HIP(c, up(&r.d_wide, hs.wide.data(), hs.wide.size() * 4));
    }
    const double t2 = now_ms();

    rt_scene_stats& st = c->stats;
    st = rt_scene_stats{};
    st.n_tris = n_tris; st.n_spheres = n_spheres; st.n_lights = n_lights;
    st.n_nodes = hs.n_nodes; st.n_leaves = hs.n_leaves; st.max_depth = hs.depth;
    c->stack_need = hs.stack_need; c->layout = hs.layout; c->has_wide = !hs.wide.empty();
    {
        float* lo = &c->bin_grid.lox; float* sc = &c->bin_grid.sx;
        for (int a = 0; a < 3; ++a) {
            const float ext = n_tris ? hs.box.hi[a] - hs.box.lo[a] : 0.f;
            lo[a] = n_tris ? hs.box.lo[a] : 0.f;
            sc[a] = ext > 0.f ? 16.0f / ext : 0.f;
        }
    }
    st.node_bytes = 64; st.tri_bytes = 48;
    st.stack_entries = hs.stack_need + 2;
    st.sah_cost = hs.sah_cost;
    st.build_ms = (float)(t1 - t0); st.upload_ms = (float)(t2 - t1);
    st.device_bytes = (uint64_t)hs.nodes.size() * 4 + (uint64_t)hs.tri48.size() * 4 + (uint64_t)hs.nrm48.size() * 4 + (uint64_t)hs.sph.size() * 4 + (uint64_t)hs.wide.size() * 4;
    c->has_scene = true;
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_scene_stats_get(const rt_ctx* c, rt_scene_stats* out) try {
    if (!c || !out) return RT_E_INVALID;
    if (!c->has_scene) return RT_E_NO_SCENE;
    *out = c->stats;
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_set_camera(rt_ctx* c, const void* ubo) try {
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
} RT_ABI_CATCH(c)

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
                                 rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream) try {
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
        const bool own = !rays || !hits;            // the slot's scratch stands in for what the caller did not pass
        RT_HIP(c, primary_scratch(r, n, &rays, &hits));
        if (own) RT_HIP(c, scratch_acquire(r, kEvPrim, st));
        RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, kNone, false, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter,
                                 rays, nullptr, r.d_counters, st));
        RT_HIP(c, trace_rays(c, r, false, false, rays, n, hits, nullptr, st));
        if (own) RT_HIP(c, scratch_release(r, kEvPrim, st));
        return RT_OK;
    }
    RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, c->mode(), c->counters != 0, w, h, y0, rows, stripe_h, stripe_step,
                             spp, seed, jitter, d_rays_out, d_hits_out, r.d_counters, st));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_primary_shade_stripes_dev(rt_ctx* c, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                 uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t seed, int jitter,
                                 uint32_t shade_seed, rt_ray* d_rays_out, rt_hit* d_hits_out, float* d_throughput4,
                                 float* d_radiance4, rt_ray* d_next_out, void* stream) try {
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
        const bool own = !rays || !hits;
        RT_HIP(c, primary_scratch(r, n, &rays, &hits));
        if (own) RT_HIP(c, scratch_acquire(r, kEvPrim, st));
        const DevScene ds = c->dev_scene(r);
        RT_HIP(c, launch_primary(ds, c->cam, kNone, false, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter, rays, nullptr,
                                 r.d_counters, st));
        RT_HIP(c, trace_rays(c, r, false, false, rays, n, hits, nullptr, st));
        RT_HIP(c, launch_shade(ds, c->mode(), rays, hits, n, (uint32_t)((uint64_t)y0 * w * spp), w * spp, stripe_h, stripe_step, shade_seed,
                               reinterpret_cast<float4*>(d_throughput4), reinterpret_cast<float4*>(d_radiance4), d_next_out, st, true));
        if (own) RT_HIP(c, scratch_release(r, kEvPrim, st));
        return RT_OK;
    }
    const PrimaryShade ps = { reinterpret_cast<float4*>(d_throughput4), reinterpret_cast<float4*>(d_radiance4), d_next_out, shade_seed };
    RT_HIP(c, launch_primary(c->dev_scene(r), c->cam, c->mode(), false, w, h, y0, rows, stripe_h, stripe_step,
                             spp, seed, jitter, d_rays_out, d_hits_out, r.d_counters, st, &ps));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_trace_primary_dev(rt_ctx* c, int slot, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows, uint32_t spp,
                         uint32_t seed, int jitter, rt_ray* d_rays_out, rt_hit* d_hits_out, void* stream) try {
    if (c && (uint64_t)y0 + rows > h) return fail(c, RT_E_INVALID, "primary: y0 + rows > h");
    return rt_trace_primary_stripes_dev(c, slot, w, h, y0, rows, rows ? rows : 1, 1, spp, seed, jitter, d_rays_out, d_hits_out, stream);
} RT_ABI_CATCH(c)

int rt_trace_dev(rt_ctx* c, int slot, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits)) return fail(c, RT_E_INVALID, "trace: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, trace_rays(c, r, false, c->counters != 0, d_rays, n, d_hits, nullptr, st));
    return RT_OK;
} RT_ABI_CATCH(c)

// rt_trace_dev for callers that cannot allocate: no hipMalloc / hipFree, no allocating library routine.  The strip
// kernel's scratch is this device's copy of a static device array (kernels.hip, noalloc_bytes()); calls that overlap
// on one GPU share it and race.
int rt_trace_dev_noalloc(rt_ctx* c, int slot, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits)) return fail(c, RT_E_INVALID, "trace_noalloc: null device pointer");
    if (n > RT_NOALLOC_MAX_RAYS) return fail(c, RT_E_TOO_LARGE, "trace_noalloc: more than RT_NOALLOC_MAX_RAYS (4194304) rays in one call; split the buffer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    const int mode = c->mode();
    // at most 2^22 rays: the launches the staged ray queue pays for (trace_rays), with its thresholds unless RT_OPT_REFILL is set
    const bool queue = c->ray_queue != 0 && mode == kBvh4q;
    const int refill = queue && c->refill == kDefaultRefill ? (6 | (1 << 8)) : c->refill;
    RT_HIP(c, launch_trace_noalloc(r.d_noalloc, c->dev_scene(r), mode, d_rays, n, d_hits, (hipStream_t)stream, c->strip, refill,
                                   c->lds_stack, wants_binning(c, mode, n), c->bin_grid, c->binning == 1 ? 0u : kBinMinRun, queue));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_occluded_dev(rt_ctx* c, int slot, const rt_ray* d_rays, uint64_t n, uint8_t* d_out, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_out)) return fail(c, RT_E_INVALID, "occluded: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, trace_rays(c, r, true, false, d_rays, n, nullptr, d_out, st));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_shadow_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, const float lp[3],
                  rt_ray* d_srays, uint8_t* d_lit, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (!lp || (n && (!d_rays || !d_hits || !d_lit))) return fail(c, RT_E_INVALID, "shadow: null pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    DevScene ds = c->dev_scene(r);
    if (c->ray_packets) ds.wide = reinterpret_cast<const uint32_t*>(r.d_wide);
    RT_HIP(c, launch_shadow(ds, c->mode(), d_rays, d_hits, n, lp[0], lp[1], lp[2], d_srays, d_lit, st, c->ray_packets != 0));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_bounce_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                  uint32_t seed, rt_ray* d_out, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits || !d_out)) return fail(c, RT_E_INVALID, "bounce: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    hipStream_t st = (hipStream_t)stream;   // NULL = HIP's default (null) stream
    RT_HIP(c, launch_bounce(c->dev_scene(r), d_rays, d_hits, n, key0, seed, d_out, st));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_shade_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                 uint32_t seed, float* d_throughput4, float* d_radiance4, rt_ray* d_out, void* stream) try {
    return rt_shade_stripes_dev(c, slot, d_rays, d_hits, n, key0, 0, 1, 1, seed, d_throughput4, d_radiance4, d_out, stream);
} RT_ABI_CATCH(c)

int rt_shade_stripes_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                         uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step, uint32_t seed, float* d_throughput4,
                         float* d_radiance4, rt_ray* d_out, void* stream) try {
    int rc = check_stage(c, slot, false);
    if (rc) return rc;
    if (n && (!d_rays || !d_hits || !d_throughput4 || !d_radiance4)) return fail(c, RT_E_INVALID, "shade: null device pointer");
    Replica& r = c->reps[slot];
    RT_HIP(c, hipSetDevice(r.device));
    RT_HIP(c, launch_shade(c->dev_scene(r), c->mode(), d_rays, d_hits, n, key0, row_len, stripe_h, stripe_step, seed, reinterpret_cast<float4*>(d_throughput4),
                           reinterpret_cast<float4*>(d_radiance4), d_out, (hipStream_t)stream));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_path_init_dev(rt_ctx* c, int slot, uint64_t n, float* d_throughput4, float* d_radiance4, void* stream) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    if (n && (!d_throughput4 || !d_radiance4)) return fail(c, RT_E_INVALID, "path_init: null device pointer");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, launch_fill(reinterpret_cast<float4*>(d_throughput4), n, 1.f, 1.f, 1.f, 0.f, (hipStream_t)stream));   // frag.glsl:308
    RT_HIP(c, launch_fill(reinterpret_cast<float4*>(d_radiance4), n, 0.f, 0.f, 0.f, 0.f, (hipStream_t)stream));     // :304
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_resolve_dev(rt_ctx* c, int slot, const float* d_radiance4, uint64_t n_pixels, uint32_t spp, float* d_linear4,
                   uint8_t* d_rgba8, void* stream) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    if (spp == 0 || (n_pixels && (!d_radiance4 || (!d_linear4 && !d_rgba8)))) return fail(c, RT_E_INVALID, "resolve: bad arguments");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, launch_resolve(reinterpret_cast<const float4*>(d_radiance4), n_pixels, spp, reinterpret_cast<float4*>(d_linear4),
                             reinterpret_cast<uint32_t*>(d_rgba8), (hipStream_t)stream));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_shade_resolve_dev(rt_ctx* c, int slot, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t spp,
                         float* d_throughput4, float* d_radiance4, float* d_linear4, uint8_t* d_rgba8, void* stream) try {
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
} RT_ABI_CATCH(c)

// frag.glsl:303-360 for a whole frame, as a wavefront pipeline: k_primary, then per bounce k_shade + closest hit.
// Several slots: each renders its interleaved share (FrameSplit), the shares are gathered on slot 0's device (frame_gather).
int rt_render(rt_ctx* c, uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces, uint32_t seed, uint8_t* host_rgba8) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (!c->has_camera) return fail(c, RT_E_NO_CAMERA, "no camera set");
    if (!host_rgba8 || w == 0 || h == 0 || spp == 0 || bounces == 0) return fail(c, RT_E_INVALID, "render: null output or zero size");
    if ((uint64_t)w * h * spp > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "render: w*h*spp exceeds 2^32-1");
    const int ns = (int)c->reps.size();
    const int mode = c->mode();
    const bool gathered = ns > 1 || c->frame_exchange == 1;       // the frame is assembled on slot 0's device
    if (gathered && h > 65535u) return fail(c, RT_E_TOO_LARGE, "render: more than 65535 rows with several slots");
    const FrameSplit sp = frame_split(h, ns);
    char* shares = nullptr; char* frame = nullptr;
    if (gathered) { const int rc = frame_stage(c, sp, h, (size_t)w * 4, &shares, &frame); if (rc != RT_OK) return rc; }
    std::vector<void*> d_share(ns, nullptr);
    const int rc = for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        // one slot: the whole frame, rows in order.  Several: local row yl is image row y0 + (yl / 8) * 8 * ns + yl % 8.
        const uint32_t y0 = gathered ? sp.y0(s) : 0, rows = gathered ? sp.share_rows : h;
        const uint32_t stripe_h = gathered ? kFrameStripe : rows, step = gathered ? (uint32_t)ns : 1u;
        const uint32_t row_len = gathered ? w * spp : 0u;        // RNG keys are path indices in the FULL frame either way
        const uint64_t n = (uint64_t)rows * w * spp, npx = (uint64_t)rows * w;
        const bool in_place = gathered && s == 0;                // slot 0's RGBA8 share is the first block of the staging
        // rays A/B (32 B), hits (16 B), throughput (16 B), radiance (16 B) per ray + RGBA8 per pixel
        const size_t bytes = n * (32 + 32 + 16 + 16 + 16) + (in_place ? 0 : npx * 4);
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        RT_HIP_SLOT(r, grow(&r.d_render, &r.render_cap, bytes));
        char* base = static_cast<char*>(r.d_render);
        rt_ray* ra = reinterpret_cast<rt_ray*>(base);
        rt_ray* rb = reinterpret_cast<rt_ray*>(base + n * 32);
        rt_hit* hits = reinterpret_cast<rt_hit*>(base + n * 64);
        float4* thr = reinterpret_cast<float4*>(base + n * 80);
        float4* rad = reinterpret_cast<float4*>(base + n * 96);
        uint32_t* rgba = in_place ? reinterpret_cast<uint32_t*>(shares) : reinterpret_cast<uint32_t*>(base + n * 112);
        const DevScene ds = c->dev_scene(r);
        if (mode == kBrute && c->fused_paths) {
            // the reference's own kind of scene (a few triangles, its brute-force loop): the whole path loop in one kernel
            RT_HIP_SLOT(r, launch_path_brute(ds, c->cam, w, h, y0, rows, stripe_h, step, spp, bounces, seed, rad, r.stream));
            RT_HIP_SLOT(r, launch_resolve(rad, npx, spp, nullptr, rgba, r.stream));
            d_share[s] = rgba;
            if (!gathered) {
                RT_HIP_SLOT(r, hipMemcpyAsync(host_rgba8, rgba, npx * 4, hipMemcpyDeviceToHost, r.stream));
                RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
            }
            return RT_OK;
        }
        // segment 0: primary rays + their shading in one kernel (the primary rays and hits stay in registers)
        const uint32_t key0 = (uint32_t)((uint64_t)y0 * w * spp);
        if (primary_via_buffer(c)) {             // big tree: ray generation -> ray-buffer kernel -> first-segment shade
            RT_HIP_SLOT(r, launch_primary(ds, c->cam, kNone, false, w, h, y0, rows, stripe_h, step, spp, seed, 1, rb, nullptr, r.d_counters, r.stream));
            RT_HIP_SLOT(r, trace_rays(c, r, false, false, rb, n, hits, nullptr, r.stream));
            RT_HIP_SLOT(r, launch_shade(ds, mode, rb, hits, n, key0, row_len, stripe_h, step, seed + 1u, thr, rad, bounces > 1 ? ra : nullptr, r.stream, true));
        } else {
            const PrimaryShade ps = { thr, rad, bounces > 1 ? ra : nullptr, seed + 1u };
            RT_HIP_SLOT(r, launch_primary(ds, c->cam, mode, false, w, h, y0, rows, stripe_h, step, spp, seed, 1, nullptr, nullptr, r.d_counters, r.stream, &ps));
        }
        const bool fuse_last = bounces > 1 && spp <= 64 && (spp & (spp - 1)) == 0;    // last segment + resolve in one kernel
        for (uint32_t b = 1; b < bounces; ++b) {
            RT_HIP_SLOT(r, trace_rays(c, r, false, false, ra, n, hits, nullptr, r.stream));
            if (b + 1 == bounces && fuse_last) {
                RT_HIP_SLOT(r, launch_shade_resolve(ds, mode, ra, hits, n, spp, thr, rad, nullptr, rgba, r.stream));
                break;
            }
            RT_HIP_SLOT(r, launch_shade(ds, mode, ra, hits, n, key0, row_len, stripe_h, step, seed + 1u + b, thr, rad, b + 1 < bounces ? rb : nullptr, r.stream));
            std::swap(ra, rb);
        }
        if (!fuse_last) RT_HIP_SLOT(r, launch_resolve(rad, npx, spp, nullptr, rgba, r.stream));
        d_share[s] = rgba;
        if (!gathered) {
            RT_HIP_SLOT(r, hipMemcpyAsync(host_rgba8, rgba, npx * 4, hipMemcpyDeviceToHost, r.stream));
            RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
        }
        return RT_OK;
    });
    if (rc != RT_OK || !gathered) return rc;
    return frame_gather(c, sp, h, (size_t)w * 4, d_share.data(), shares, frame, host_rgba8);
} RT_ABI_CATCH(c)

int rt_counters_get(rt_ctx* c, int slot, rt_counters* out, int reset) try {
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
    out->idle_lane_rounds = v[8]; out->blocked_lane_rounds = v[9]; out->drained_lane_rounds = v[10]; out->tri_rounds = v[11];
    if (reset) RT_HIP(c, hipMemset(r.d_counters, 0, sizeof v));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_sync(rt_ctx* c, int slot) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!valid_slot(c, slot)) return fail(c, RT_E_INVALID, "slot out of range");
    RT_HIP(c, hipSetDevice(c->reps[slot].device));
    RT_HIP(c, hipDeviceSynchronize());     // *_dev stages may have been enqueued on any stream
    return RT_OK;
} RT_ABI_CATCH(c)

// ---- host-pointer entry points (blocking; what the reference's single-threaded frame loop expects) ---
int rt_trace_primary(rt_ctx* c, uint32_t w, uint32_t h, rt_hit* host_out) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (!c->has_camera) return fail(c, RT_E_NO_CAMERA, "no camera set");
    if (!host_out || w == 0 || h == 0) return fail(c, RT_E_INVALID, "trace_primary: null output or empty image");
    if ((uint64_t)w * h > 0xFFFFFFFFull) return fail(c, RT_E_TOO_LARGE, "trace_primary: w*h exceeds 2^32-1");
    const int ns = (int)c->reps.size();
    const int mode = c->mode();
    const bool gathered = ns > 1 || c->frame_exchange == 1;       // interleaved 8-row stripes per slot, assembled on slot 0's device
    if (gathered && h > 65535u) return fail(c, RT_E_TOO_LARGE, "trace_primary: more than 65535 rows with several slots");
    const FrameSplit sp = frame_split(h, ns);
    const size_t row_bytes = (size_t)w * sizeof(rt_hit);
    char* shares = nullptr; char* frame = nullptr;
    if (gathered) { const int rc = frame_stage(c, sp, h, row_bytes, &shares, &frame); if (rc != RT_OK) return rc; }
    std::vector<void*> d_share(ns, nullptr);
    const int rc = for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        const uint32_t y0 = gathered ? sp.y0(s) : 0, rows = gathered ? sp.share_rows : h;
        const uint32_t stripe_h = gathered ? kFrameStripe : rows, step = gathered ? (uint32_t)ns : 1u;
        const size_t bytes = (size_t)rows * row_bytes;
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        rt_hit* out = reinterpret_cast<rt_hit*>(shares);          // slot 0 of a gathered frame writes its share in place
        if (!(gathered && s == 0)) {
            RT_HIP_SLOT(r, grow(&r.d_out, &r.out_cap, bytes));
            out = static_cast<rt_hit*>(r.d_out);
        }
        if (primary_via_buffer(c)) {             // trees whose primary rays go through ray generation + the ray-buffer kernel
            rt_ray* rays = nullptr;
            rt_hit* hh = out;
            const uint64_t n = (uint64_t)rows * w;
            RT_HIP_SLOT(r, primary_scratch(r, n, &rays, &hh));
            RT_HIP_SLOT(r, scratch_acquire(r, kEvPrim, r.stream));
            RT_HIP_SLOT(r, launch_primary(c->dev_scene(r), c->cam, kNone, false, w, h, y0, rows, stripe_h, step, 1, 0, 0, rays, nullptr, r.d_counters, r.stream));
            RT_HIP_SLOT(r, trace_rays(c, r, false, false, rays, n, out, nullptr, r.stream));
            RT_HIP_SLOT(r, scratch_release(r, kEvPrim, r.stream));
        } else {
            RT_HIP_SLOT(r, launch_primary(c->dev_scene(r), c->cam, mode, false, w, h, y0, rows, stripe_h, step, 1, 0, 0, nullptr, out, r.d_counters, r.stream));
        }
        d_share[s] = out;
        if (!gathered) {
            RT_HIP_SLOT(r, hipMemcpyAsync(host_out, out, bytes, hipMemcpyDeviceToHost, r.stream));
            RT_HIP_SLOT(r, hipStreamSynchronize(r.stream));
        }
        return RT_OK;
    });
    if (rc != RT_OK || !gathered) return rc;
    return frame_gather(c, sp, h, row_bytes, d_share.data(), shares, frame, host_out);
} RT_ABI_CATCH(c)

// Is p host memory the GPU can DMA from / to directly (rt_host_alloc, hipHostMalloc)?
static bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int pipe_prepare(Replica& r, uint64_t piece, bool stage_in, bool stage_out, size_t osz) {
    HostPipe& p = r.pipe;
    if (!p.ready) {                                 // a call that failed part-way left some handles behind: create only the missing ones
        for (int k = 0; k < 4; ++k) if (!p.st[k]) RT_HIP_SLOT(r, hipStreamCreateWithFlags(&p.st[k], hipStreamNonBlocking));
        for (int k = 0; k < kPipeRing; ++k) {
            if (!p.done[k]) RT_HIP_SLOT(r, hipEventCreateWithFlags(&p.done[k], hipEventDisableTiming));
            if (!p.got[k]) RT_HIP_SLOT(r, hipEventCreateWithFlags(&p.got[k], hipEventDisableTiming));
            if (!p.ran[k]) RT_HIP_SLOT(r, hipEventCreateWithFlags(&p.ran[k], hipEventDisableTiming));
        }
        p.ready = true;
    }
    if (piece > p.cap_rays) {                       // device buffers sized for hits (16 B) so that any-hit fits too
        for (int k = 0; k < kPipeRing; ++k) {
            if (p.d_in[k]) (void)hipFree(p.d_in[k]);
            if (p.d_out[k]) (void)hipFree(p.d_out[k]);
            if (p.pin_in[k]) (void)hipHostFree(p.pin_in[k]);
            if (p.pin_out[k]) (void)hipHostFree(p.pin_out[k]);
            p.d_in[k] = p.d_out[k] = p.pin_in[k] = p.pin_out[k] = nullptr;
        }
        p.cap_rays = 0;
        for (int k = 0; k < kPipeRing; ++k) {
            RT_HIP_SLOT(r, hipMalloc(&p.d_in[k], piece * sizeof(rt_ray)));
            RT_HIP_SLOT(r, hipMalloc(&p.d_out[k], piece * sizeof(rt_hit)));
        }
        p.cap_rays = piece;
    }
    (void)osz;
    for (int k = 0; k < kPipeRing; ++k) {           // pinned staging only for the directions that need it
        if (stage_in && !p.pin_in[k]) RT_HIP_SLOT(r, hipHostMalloc(&p.pin_in[k], p.cap_rays * sizeof(rt_ray), hipHostMallocDefault));
        if (stage_out && !p.pin_out[k]) RT_HIP_SLOT(r, hipHostMalloc(&p.pin_out[k], p.cap_rays * sizeof(rt_hit), hipHostMallocDefault));
    }
    if ((stage_in || stage_out) && !p.pool) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        p.pool.reset(new CopyPool(std::min(7u, std::max(1u, hw / 2) - 0u)));
    }
    return RT_OK;
}

// Host-pointer closest hit / any-hit (what VkRenderer::drawFrame would call, src/vulkan/VkRenderer.cpp:1287-1374): the
// rays go through the GPU in pieces of kPieceRays, kPipeRing pieces in flight on four streams, so that the staging copy
// of piece q+1, the H2D copy + kernel of piece q and the D2H copy of piece q-1 overlap.  Pageable caller memory is
// staged through pinned buffers by a few host threads; pinned memory (rt_host_alloc / hipHostMalloc) is read and written
// by DMA directly.
static int trace_host(rt_ctx* c, const rt_ray* rays, uint64_t n, void* out, bool any_hit) {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!c->has_scene) return fail(c, RT_E_NO_SCENE, "no scene uploaded");
    if (n == 0) return RT_OK;
    if (!rays || !out) return fail(c, RT_E_INVALID, "trace: null host pointer");
    const int ns = (int)c->reps.size();
    const uint64_t per = (n + ns - 1) / ns;
    const size_t osz = any_hit ? 1 : sizeof(rt_hit);
    const bool stage_in = !is_pinned(rays), stage_out = !is_pinned(out);
    return for_each_slot(c, [&](int s) -> int {
        Replica& r = c->reps[s];
        const uint64_t b = std::min(per * s, n), e = std::min(b + per, n);
        if (e <= b) return RT_OK;
        // 2 Mi-ray pieces keep the link busy when the kernels are fast; structures that do not fit the L2s (the ones whose rays get
        // binned) run small launches at well under their resident rate, so their pieces are twice as long (10 M-triangle soup:
        // 1.03 -> 1.16 Grays/s; the bench scene would lose 3 %: profiles/r03t_host_path.txt)
        const uint64_t scene_bytes = (uint64_t)c->stats.n_nodes * 64 + (uint64_t)c->stats.n_tris * 48;
        const uint64_t piece = std::min<uint64_t>(scene_bytes > kBinMinSceneBytes ? 2 * kPieceRays : kPieceRays, e - b);
        RT_HIP_SLOT(r, hipSetDevice(r.device));
        int rc = pipe_prepare(r, piece, stage_in, stage_out, osz);
        if (rc != RT_OK) return rc;
        HostPipe& p = r.pipe;
        // Whatever way the loop below is left, no copy may outlive the call: earlier pieces still read the caller's rays (or
        // pin_in) and write the caller's `out` (or pin_out), and the next call reuses the ring.  On success every piece has
        // been waited for already; on an error return the four streams are drained first.
        struct Drain {
            HostPipe& p; bool armed = true;
            ~Drain() { if (armed) for (hipStream_t s : p.st) if (s) (void)hipStreamSynchronize(s); }
        } drain{ p };
        const uint64_t n_pieces = (e - b + piece - 1) / piece;
        auto first = [&](uint64_t q) { return b + q * piece; };
        auto count = [&](uint64_t q) { return std::min(piece, e - first(q)); };
        auto finish = [&](uint64_t q) -> int {               // piece q has left the GPU: hand its results to the caller
            const int k = (int)(q % kPipeRing);
            RT_HIP_SLOT(r, hipEventSynchronize(p.done[k]));
            if (stage_out) p.pool->copy(static_cast<char*>(out) + first(q) * osz, p.pin_out[k], count(q) * osz);
            return RT_OK;
        };
        for (uint64_t q = 0; q < n_pieces; ++q) {
            const int k = (int)(q % kPipeRing);
            if (q >= (uint64_t)kPipeRing && (rc = finish(q - kPipeRing)) != RT_OK) return rc;
            const uint64_t m = count(q);
            const rt_ray* src = rays + first(q);
            if (stage_in) { p.pool->copy(p.pin_in[k], src, m * sizeof(rt_ray)); src = static_cast<const rt_ray*>(p.pin_in[k]); }
            void* dst = stage_out ? p.pin_out[k] : static_cast<void*>(static_cast<char*>(out) + first(q) * osz);
            // One stream per direction and two for the kernels (even / odd pieces), chained by events: the inbound copies follow
            // one another without waiting for anything but their ring slot, and the kernels of neighbouring pieces may overlap
            // (small launches have tails; the 10 M-triangle soup is kernel-bound here).  (Pieces alternating over two streams,
            // each with H2D | kernels | D2H, reached 65 % of the link rate from pinned memory: copies of both directions queued
            // on one stream do not overlap with the other stream's.  Now 86-89 %: profiles/r03t_host_path.txt.)
            hipStream_t s_in = p.st[0], s_out = p.st[1], s_k = p.st[2 + (q & 1)];
            RT_HIP_SLOT(r, hipMemcpyAsync(p.d_in[k], src, m * sizeof(rt_ray), hipMemcpyHostToDevice, s_in));
            RT_HIP_SLOT(r, hipEventRecord(p.got[k], s_in));
            RT_HIP_SLOT(r, hipStreamWaitEvent(s_k, p.got[k], 0));
            RT_HIP_SLOT(r, trace_rays(c, r, any_hit, false, static_cast<const rt_ray*>(p.d_in[k]), m,
                                      any_hit ? nullptr : static_cast<rt_hit*>(p.d_out[k]),
                                      any_hit ? static_cast<uint8_t*>(p.d_out[k]) : nullptr, s_k));
            RT_HIP_SLOT(r, hipEventRecord(p.ran[k], s_k));
            RT_HIP_SLOT(r, hipStreamWaitEvent(s_out, p.ran[k], 0));
            RT_HIP_SLOT(r, hipMemcpyAsync(dst, p.d_out[k], m * osz, hipMemcpyDeviceToHost, s_out));
            RT_HIP_SLOT(r, hipEventRecord(p.done[k], s_out));
        }
        for (uint64_t q = n_pieces > (uint64_t)kPipeRing ? n_pieces - kPipeRing : 0; q < n_pieces; ++q)
            if ((rc = finish(q)) != RT_OK) return rc;
        drain.armed = false;
        return RT_OK;
    });
}

// Pinned host memory for the caller's ray / hit arrays (the arrays a renderer keeps for its lifetime, VkRenderer.h:80-90):
// rt_trace / rt_occluded DMA from and to such buffers directly, without staging copies.  (hipHostRegister of ordinary
// memory is refused by the driver on the target pool, so the library hands the memory out instead.)
int rt_host_alloc(rt_ctx* c, size_t bytes, void** out) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    if (!out || !bytes) return fail(c, RT_E_INVALID, "host_alloc: null pointer or zero size");
    *out = nullptr;
    RT_HIP(c, hipSetDevice(c->reps[0].device));
    c->host_blocks.reserve(c->host_blocks.size() + 1);
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); return fail(c, RT_E_NOMEM, "host_alloc: out of pinned host memory"); }
    RT_HIP(c, e);
    c->host_blocks.push_back(p);
    *out = p;
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_host_free(rt_ctx* c, void* p) try {
    if (!c) return RT_E_INVALID;
    c->err.clear();
    auto it = std::find(c->host_blocks.begin(), c->host_blocks.end(), p);
    if (!p || it == c->host_blocks.end()) return fail(c, RT_E_INVALID, "host_free: not a block of rt_host_alloc");
    for (Replica& r : c->reps) { RT_HIP(c, hipSetDevice(r.device)); RT_HIP(c, hipDeviceSynchronize()); }   // no copy still uses it
    c->host_blocks.erase(it);
    RT_HIP(c, hipHostFree(p));
    return RT_OK;
} RT_ABI_CATCH(c)

int rt_trace(rt_ctx* c, const rt_ray* host_rays, uint64_t n, rt_hit* host_out) try {
    return trace_host(c, host_rays, n, host_out, false);
} RT_ABI_CATCH(c)
int rt_occluded(rt_ctx* c, const rt_ray* host_rays, uint64_t n, uint8_t* host_out) try {
    return trace_host(c, host_rays, n, host_out, true);
} RT_ABI_CATCH(c)

}  // extern "C"
