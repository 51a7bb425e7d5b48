// bvh.cpp -- binned-SAH BVH2 build (multi-threaded, depth-bounded) and flattening.
// Build-original; see bvh.hpp for the contract.
//
// How the build is organised (round 3; the first builder walked an index array, three binning passes and one bounds pass per
// node, the top of the tree on one thread: 4.4 s for 10 M triangles on 16 threads):
//   * primitives are 32-byte records (box + id) that are PARTITIONED IN PLACE, so every pass streams through memory instead of
//     chasing indices into a 240 MB box array;
//   * one pass bins a node's primitives along all three axes; the partition pass that follows also accumulates the children's
//     boxes and centroid boxes, so a child never re-reads its primitives for bounds: two passes per node;
//   * nodes above kParNode primitives are split DATA-PARALLEL (per-chunk bins merged in chunk order, stable partition by
//     per-chunk counts + scatter), breadth first, on a fork-join pool; the subtrees below are tasks of a deque pool and split
//     further into tasks while both halves exceed kSpawnThreshold;
//   * every partition is STABLE, so the primitive order -- and with it the flattened tree -- is the same for any number of
//     threads (tests/test_bvh_host.py); subtree statistics are combined along the tree, not in completion order.
// The split rule is the first builder's, value for value: per axis 4..32 bins over the centroid box, cost =
// cost_trav + cost_isect * (A_l n_l + A_r n_r) / A, axes and planes scanned in ascending order with strict '<', object median
// along the widest centroid axis when no plane separates anything or when the depth budget forces a balanced split.
#include "bvh.hpp"
#include "parallel.hpp"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <system_error>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace rt {
namespace {

constexpr int kMaxBins = 32;                  // BuildParams::bins is clamped to [4, kMaxBins]
constexpr uint32_t kSpawnThreshold = 16384;   // subtrees larger than this become pool tasks
constexpr uint32_t kParNode = 1u << 19;       // nodes with more primitives are split data-parallel
constexpr uint32_t kParChunk = 1u << 15;      // primitives per chunk of a data-parallel pass
constexpr uint32_t kNodeBlock = 1024;         // node slots a thread takes at a time (even)

// 4-lane boxes for the hot loops (lane 3 unused): SSE on x86-64, plain arrays elsewhere.  min / max are exact, so both
// forms give the same boxes bit for bit.
#if defined(__SSE2__)
typedef __m128 v4;
inline v4 v4_load(const float* p) { return _mm_loadu_ps(p); }
inline void v4_store(float* p, v4 v) { _mm_storeu_ps(p, v); }
inline v4 v4_set1(float x) { return _mm_set1_ps(x); }
inline v4 v4_min(v4 a, v4 b) { return _mm_min_ps(a, b); }
inline v4 v4_max(v4 a, v4 b) { return _mm_max_ps(a, b); }
inline v4 v4_xyz(v4 a) { return _mm_and_ps(a, _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1))); }   // lane 3 (id bits) -> +0: never a denormal operand
// bins of the three axes: (int)((0.5 * (lo + hi) - clo) * scale), the scalar expression lane by lane (no contraction)
inline void v4_bins(v4 lo, v4 hi, v4 clo, v4 scale, int k[4]) {
    const v4 c = _mm_mul_ps(_mm_set1_ps(0.5f), _mm_add_ps(lo, hi));
    _mm_storeu_si128(reinterpret_cast<__m128i*>(k), _mm_cvttps_epi32(_mm_mul_ps(_mm_sub_ps(c, clo), scale)));
}
inline v4 v4_centroid(v4 lo, v4 hi) { return _mm_mul_ps(_mm_set1_ps(0.5f), _mm_add_ps(lo, hi)); }
// dx*dy + dy*dz + dz*dx of the box (lo, hi), added in the order Aabb::half_area adds them: the same bits
inline float v4_half_area(v4 lo, v4 hi) {
    const v4 d = _mm_sub_ps(hi, lo);
    const v4 m = _mm_mul_ps(d, _mm_shuffle_ps(d, d, _MM_SHUFFLE(3, 0, 2, 1)));        // (dx*dy, dy*dz, dz*dx, -)
    const float xy = _mm_cvtss_f32(m), yz = _mm_cvtss_f32(_mm_shuffle_ps(m, m, 1)), zx = _mm_cvtss_f32(_mm_shuffle_ps(m, m, 2));
    return xy + yz + zx;
}
#else
struct v4 { float f[4]; };
inline v4 v4_load(const float* p) { v4 r; for (int i = 0; i < 4; ++i) r.f[i] = p[i]; return r; }
inline void v4_store(float* p, v4 v) { for (int i = 0; i < 4; ++i) p[i] = v.f[i]; }
inline v4 v4_set1(float x) { v4 r; for (int i = 0; i < 4; ++i) r.f[i] = x; return r; }
inline v4 v4_min(v4 a, v4 b) { v4 r; for (int i = 0; i < 4; ++i) r.f[i] = b.f[i] < a.f[i] ? b.f[i] : a.f[i]; return r; }
inline v4 v4_max(v4 a, v4 b) { v4 r; for (int i = 0; i < 4; ++i) r.f[i] = b.f[i] > a.f[i] ? b.f[i] : a.f[i]; return r; }
inline v4 v4_xyz(v4 a) { a.f[3] = 0.f; return a; }
inline void v4_bins(v4 lo, v4 hi, v4 clo, v4 scale, int k[4]) { for (int i = 0; i < 4; ++i) k[i] = (int)((0.5f * (lo.f[i] + hi.f[i]) - clo.f[i]) * scale.f[i]); }
inline v4 v4_centroid(v4 lo, v4 hi) { v4 r; for (int i = 0; i < 4; ++i) r.f[i] = 0.5f * (lo.f[i] + hi.f[i]); return r; }
inline float v4_half_area(v4 lo, v4 hi) { const float dx = hi.f[0] - lo.f[0], dy = hi.f[1] - lo.f[1], dz = hi.f[2] - lo.f[2]; return dx * dy + dy * dz + dz * dx; }
#endif

struct Prim { float lo[3]; uint32_t id; float hi[3]; uint32_t pad; };       // 32 B
static_assert(sizeof(Prim) == 32, "Prim must be 32 bytes");
inline v4 prim_lo(const Prim& p) { return v4_xyz(v4_load(p.lo)); }
inline v4 prim_hi(const Prim& p) { return v4_xyz(v4_load(p.hi)); }

struct Box4 {                      // lanes 0..2 = x, y, z
    v4 lo, hi;
    void reset() { lo = v4_set1(3.0e38f); hi = v4_set1(-3.0e38f); }
    void grow(v4 l, v4 h) { lo = v4_min(lo, l); hi = v4_max(hi, h); }
    void grow(const Box4& b) { grow(b.lo, b.hi); }
    Aabb aabb() const {
        float l[4], h[4]; v4_store(l, lo); v4_store(h, hi);
        Aabb r; for (int a = 0; a < 3; ++a) { r.lo[a] = l[a]; r.hi[a] = h[a]; }
        return r;
    }
    float half_area() const { return v4_half_area(lo, hi); }     // the same products added in the same order, hence the same bits, as Aabb::half_area
};

struct Task { uint32_t node, begin, end; int depth; Aabb box, cbox; bool in_tmp; };     // in_tmp: which of the two record buffers holds [begin, end)
struct Stats { double cost = 0.0; uint32_t height = 0, leaves = 0; };        // of a subtree; height 0 = the node is a leaf

struct Bins {
    Box4 bb[3][kMaxBins];
    uint32_t bc[3][kMaxBins];
    void reset(int n) { for (int a = 0; a < 3; ++a) for (int k = 0; k < n; ++k) { bb[a][k].reset(); bc[a][k] = 0; } }
    void merge(const Bins& o, int n) { for (int a = 0; a < 3; ++a) for (int k = 0; k < n; ++k) { bb[a][k].grow(o.bb[a][k]); bc[a][k] += o.bc[a][k]; } }
};
struct Split { int axis = -1, plane = -1; float cost = 3.0e38f; bool leaf = false, median = false; };
struct Sides {
    Box4 box[2], cbox[2];
    void reset() { for (int i = 0; i < 2; ++i) { box[i].reset(); cbox[i].reset(); } }
    void add(int side, const Prim& p) { const v4 l = prim_lo(p), h = prim_hi(p), c = v4_centroid(l, h); box[side].grow(l, h); cbox[side].grow(c, c); }
    void merge(const Sides& o) { for (int i = 0; i < 2; ++i) { box[i].grow(o.box[i]); cbox[i].grow(o.cbox[i]); } }
};

struct Builder {
    Prim* prims = nullptr;        // the records; a subtree's range lives in one of the two buffers (Task::in_tmp) ...
    Prim* tmp = nullptr;          // ... and the other one is the target / holding space of its stable partitions
    Prim* buf(bool in_tmp) const { return in_tmp ? tmp : prims; }
    BuildParams P;
    NodeArray* nodes = nullptr;
    std::atomic<uint32_t> n_nodes{1};
    std::atomic<bool> oversize_leaf{false};       // a leaf with more than 8 primitives was made (depth budget exhausted)
    ForkJoin* fj = nullptr;

    // Node slots are handed out to a thread in blocks: one shared counter bumped for every node pair (3 M times for 10 M
    // triangles, from 16 threads) and neighbouring nodes written by different threads cost the subtree phase half its
    // speed on a 16-thread box.  Slots a thread does not use stay uninitialised; nothing walks the array linearly.
    struct NodeAlloc {
        uint32_t next = 0, end = 0;
        uint32_t pair(std::atomic<uint32_t>& counter) {
            if (next + 2 > end) { next = counter.fetch_add(kNodeBlock); end = next + kNodeBlock; }
            const uint32_t c = next; next += 2; return c;
        }
    };

    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    int pending = 0;          // tasks queued or running
    bool parallel = false;
    std::exception_ptr failure;   // first exception of a worker (queue growth can throw); rethrown by build_bvh2
    std::unordered_map<uint32_t, Stats> done;     // statistics of subtrees that were built without spawning (guarded by mu)

    static int ceil_log2(uint32_t x) { int l = 0; uint32_t v = 1; while (v < x) { v <<= 1; ++l; } return l; }

    void spawn(const Task& t) {
        std::lock_guard<std::mutex> g(mu);
        queue.push_back(t);
        ++pending;
        cv.notify_one();
    }

    void worker() {
        NodeAlloc na;
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return !queue.empty() || pending == 0; });
                if (queue.empty()) return;
                t = queue.front();
                queue.pop_front();
            }
            // nothing may escape a std::thread: keep the first exception, drain the queue without building
            bool skip;
            { std::lock_guard<std::mutex> g(mu); skip = (bool)failure; }
            if (!skip) {
                try {
                    Stats st;
                    if (build(t, na, st)) { std::lock_guard<std::mutex> g(mu); done.emplace(t.node, st); }
                }
                catch (...) { std::lock_guard<std::mutex> g(mu); if (!failure) failure = std::current_exception(); }
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (--pending == 0) cv.notify_all();
            }
        }
    }

    void make_leaf(BuildNode& n, uint32_t begin, uint32_t end) {
        n.first = begin;
        n.count = end - begin;
        if (n.count > 8) oversize_leaf.store(true, std::memory_order_relaxed);
    }
    static constexpr float kStatTrav = 1.2f;
    Stats leaf_stats(const BuildNode& n) const {
        Stats s; s.cost = (double)P.cost_isect * n.count * n.box.half_area(); s.leaves = 1; return s;       // scaled by 1 / root area at the end
    }
    static Stats join_stats(const BuildParams& P, const BuildNode& n, const Stats& l, const Stats& r) {
        Stats s;
        // the REPORTED cost always weighs an interior step as kStatTrav triangle tests, whatever the build was steered by: the
        // figure classifies trees (rt_scene_stats.sah_cost, the primary-path rule for BVH2 trees) and must not move with a tuning option
        (void)P;
        s.cost = (double)kStatTrav * n.box.half_area() + l.cost + r.cost;
        s.height = 1 + std::max(l.height, r.height);
        s.leaves = l.leaves + r.leaves;
        return s;
    }

    // bin index of a primitive along axis a (the one expression both the binning and the partition use)
    static inline int bin_of(const Prim& p, int a, float lo, float scale, int n_bins) {
        int k = (int)((0.5f * (p.lo[a] + p.hi[a]) - lo) * scale);
        return k < 0 ? 0 : (k >= n_bins ? n_bins - 1 : k);
    }

    void bin_range(const Prim* prims, uint32_t b, uint32_t e, const Aabb& cbox, const float scale[3], const bool use[3], Bins& out) const {
        const int nb = P.bins;
        const float cl[4] = { cbox.lo[0], cbox.lo[1], cbox.lo[2], 0.f }, sc[4] = { scale[0], scale[1], scale[2], 0.f };
        const v4 clo = v4_load(cl), scl = v4_load(sc);
        for (uint32_t i = b; i < e; ++i) {
            const v4 l = prim_lo(prims[i]), h = prim_hi(prims[i]);
            int k[4];
            v4_bins(l, h, clo, scl, k);
            for (int a = 0; a < 3; ++a) {
                if (!use[a]) continue;
                const int ka = k[a] < 0 ? 0 : (k[a] >= nb ? nb - 1 : k[a]);
                out.bb[a][ka].grow(l, h);
                ++out.bc[a][ka];
            }
        }
    }

    // The split of a node from its bins (the first builder's rule, see the header).  Only the non-empty bins are swept:
    // between two of them every plane separates the same two sets, and of equal costs the lowest plane wins (strict '<'),
    // so the candidates are the non-empty bins but the last -- the same choice as a sweep over all planes, at a cost that
    // follows the node's size (most nodes hold a handful of primitives).
    Split choose(const Task& t, const Bins& bins, const bool use[3], bool force_balanced) const {
        Split s;
        const uint32_t n = t.end - t.begin;
        const int nb = P.bins;
        if (!force_balanced) {
            const float inv_area = 1.0f / std::max(t.box.half_area(), 1e-30f);
            for (int a = 0; a < 3; ++a) {
                if (!use[a]) continue;
                int idx[kMaxBins + 1], m = 0;
                for (int k = 0; k < nb; ++k) { idx[m] = k; m += bins.bc[a][k] != 0u; }        // no branch: which bins are empty is a coin toss for small nodes
                float ra[kMaxBins]; uint32_t rc[kMaxBins];
                Box4 acc; acc.reset(); uint32_t cnt = 0;
                for (int j = m - 1; j > 0; --j) { acc.grow(bins.bb[a][idx[j]]); cnt += bins.bc[a][idx[j]]; ra[j] = acc.half_area(); rc[j] = cnt; }
                acc.reset(); cnt = 0;
                for (int j = 0; j + 1 < m; ++j) {
                    acc.grow(bins.bb[a][idx[j]]); cnt += bins.bc[a][idx[j]];
                    const float cost = P.cost_trav + P.cost_isect * inv_area * (acc.half_area() * (float)cnt + ra[j + 1] * (float)rc[j + 1]);
                    const bool better = cost < s.cost;
                    s.cost = better ? cost : s.cost; s.axis = better ? a : s.axis; s.plane = better ? idx[j] : s.plane;
                }
            }
            const float leaf_cost = P.cost_isect * (float)n;
            if (n <= (uint32_t)P.max_leaf && (s.axis < 0 || leaf_cost <= s.cost)) { s.leaf = true; return s; }
        }
        if (s.axis < 0) {
            if (n <= (uint32_t)P.max_leaf) { s.leaf = true; return s; }
            s.median = true;
            int a = 0;
            float e0 = t.cbox.hi[0] - t.cbox.lo[0], e1 = t.cbox.hi[1] - t.cbox.lo[1], e2 = t.cbox.hi[2] - t.cbox.lo[2];
            if (e1 > e0) { a = 1; e0 = e1; }
            if (e2 > e0) a = 2;
            s.axis = a;
        }
        return s;
    }

    void sides_of(const Prim* prims, uint32_t begin, uint32_t mid, uint32_t end, Sides& sd) const {
        sd.reset();
        for (uint32_t i = begin; i < mid; ++i) sd.add(0, prims[i]);
        for (uint32_t i = mid; i < end; ++i) sd.add(1, prims[i]);
    }

    // object median along axis a (also the forced-balanced path); ties broken by id
    uint32_t median_split(Prim* prims, uint32_t begin, uint32_t end, int a) {
        const uint32_t mid = begin + (end - begin) / 2;
        std::nth_element(prims + begin, prims + mid, prims + end, [a](const Prim& x, const Prim& y) {
            const float cx = x.lo[a] + x.hi[a], cy = y.lo[a] + y.hi[a];
            return cx < cy || (cx == cy && x.id < y.id);
        });
        return mid;
    }

    // stable partition of [begin, end) by "bin <= plane", one thread; accumulates both sides' bounds on the way
    uint32_t partition_seq(const Task& t, const Split& s, float scale, Sides& sd) {
        const int a = s.axis, nb = P.bins;
        const float lo = t.cbox.lo[a];
        Prim* cur = buf(t.in_tmp);
        uint32_t w = t.begin, r = 0;                 // left records move down in place, right ones wait in the other buffer
        Prim* hold = buf(!t.in_tmp) + t.begin;
        sd.reset();
        for (uint32_t i = t.begin; i < t.end; ++i) {             // no branch on the side: it is a coin toss near the top of the tree
            const Prim p = cur[i];
            const uint32_t right = bin_of(p, a, lo, scale, nb) <= s.plane ? 0u : 1u;
            cur[w] = p; hold[r] = p;                               // w <= i: never ahead of the read position
            w += 1u - right; r += right;
            sd.add((int)right, p);
        }
        if (r) std::memcpy(cur + w, hold, (size_t)r * sizeof(Prim));
        return w;
    }

    // The same partition over the fork-join pool: every chunk scatters to its place in the OTHER buffer (the children live
    // there: no copy back).  How many of a chunk's records go left is already in the chunk's bins.
    uint32_t partition_par(const Task& t, const Split& s, float scale, const std::vector<Bins>& chunk_bins, Sides& sd) {
        const int a = s.axis, nb = P.bins;
        const float lo = t.cbox.lo[a];
        const uint32_t n = t.end - t.begin, nc = (n + kParChunk - 1) / kParChunk;
        std::vector<uint32_t> left(nc + 1, 0);
        for (uint32_t c = 0; c < nc; ++c) {
            uint32_t cnt = 0;
            for (int k = 0; k <= s.plane; ++k) cnt += chunk_bins[c].bc[a][k];
            left[c + 1] = left[c] + cnt;                                          // left[c] = left records before chunk c
        }
        const uint32_t n_left = left[nc];
        const Prim* src = buf(t.in_tmp);
        Prim* dst = buf(!t.in_tmp);
        std::vector<Sides> part(nc);
        fj->run(nc, [&](uint32_t c) {
            const uint32_t b = t.begin + c * kParChunk, e = std::min(t.end, b + kParChunk);
            Prim* wl = dst + t.begin + left[c];
            Prim* wr = dst + t.begin + n_left + ((b - t.begin) - left[c]);
            Sides sc; sc.reset();
            for (uint32_t i = b; i < e; ++i) {
                const Prim p = src[i];
                if (bin_of(p, a, lo, scale, nb) <= s.plane) { *wl++ = p; sc.add(0, p); }
                else { *wr++ = p; sc.add(1, p); }
            }
            part[c] = sc;
        });
        sd.reset();
        for (const Sides& sc : part) sd.merge(sc);
        return t.begin + n_left;
    }

    // One node: leaf, or split into two child tasks.  par = use the fork-join pool for the passes (big nodes, phase 1).
    // Returns false for a leaf.
    bool split_node(const Task& t, bool par, NodeAlloc& na, Task& lt, Task& rt_) {
        BuildNode& node = (*nodes)[t.node];
        const uint32_t n = t.end - t.begin;
        node.box = t.box;
        node.count = 0; node.first = 0; node.left = node.right = 0;
        if (n <= 1 || t.depth >= P.max_depth) { make_leaf(node, t.begin, t.end); return false; }
        // depth budget: a balanced split of n prims needs ceil_log2(ceil(n/max_leaf)) more levels
        const int need = ceil_log2((n + P.max_leaf - 1) / P.max_leaf);
        const bool force_balanced = (t.depth + need >= P.max_depth);
        const int nb = P.bins;
        bool use[3]; float scale[3];
        for (int a = 0; a < 3; ++a) {
            const float ext = t.cbox.hi[a] - t.cbox.lo[a];
            use[a] = ext > 0.f;
            scale[a] = use[a] ? (float)nb / ext : 0.f;
        }
        Bins bins;
        bins.reset(nb);
        Prim* cur = buf(t.in_tmp);
        std::vector<Bins> part;                       // per-chunk bins of a data-parallel pass
        if (!force_balanced) {
            if (par) {
                const uint32_t nc = (n + kParChunk - 1) / kParChunk;
                part.resize(nc);
                fj->run(nc, [&](uint32_t c) {
                    const uint32_t b = t.begin + c * kParChunk, e = std::min(t.end, b + kParChunk);
                    part[c].reset(nb);
                    bin_range(cur, b, e, t.cbox, scale, use, part[c]);
                });
                for (const Bins& pb : part) bins.merge(pb, nb);
            } else {
                bin_range(cur, t.begin, t.end, t.cbox, scale, use, bins);
            }
        }
        const Split s = choose(t, bins, use, force_balanced);
        if (s.leaf) { make_leaf(node, t.begin, t.end); return false; }
        Sides sd;
        uint32_t mid;
        bool child_in_tmp = t.in_tmp;
        if (s.median) {
            mid = median_split(cur, t.begin, t.end, s.axis);
            sides_of(cur, t.begin, mid, t.end, sd);
        } else if (par) {
            mid = partition_par(t, s, scale[s.axis], part, sd);       // a plane is only chosen with both sides populated
            child_in_tmp = !t.in_tmp;
        } else {
            mid = partition_seq(t, s, scale[s.axis], sd);
        }
        const uint32_t c = na.pair(n_nodes);
        node.left = c; node.right = c + 1;
        lt = Task{ c, t.begin, mid, t.depth + 1, sd.box[0].aabb(), sd.cbox[0].aabb(), child_in_tmp };
        rt_ = Task{ c + 1, mid, t.end, t.depth + 1, sd.box[1].aabb(), sd.cbox[1].aabb(), child_in_tmp };
        return true;
    }

    // A subtree on one thread (spawning its right halves while they are large).  Returns true with the subtree's statistics
    // when nothing below was handed to another task; otherwise the pieces report on their own (worker()).
    bool build(const Task& t, NodeAlloc& na, Stats& st) {
        Task lt, rt_;
        if (!split_node(t, false, na, lt, rt_)) { st = leaf_stats((*nodes)[t.node]); return true; }
        Stats sl, sr;
        if (parallel && (lt.end - lt.begin) > kSpawnThreshold && (rt_.end - rt_.begin) > kSpawnThreshold) {
            spawn(rt_);
            if (build(lt, na, sl)) { std::lock_guard<std::mutex> g(mu); done.emplace(lt.node, sl); }
            return false;
        }
        const bool cl = build(lt, na, sl), cr = build(rt_, na, sr);
        if (cl && cr) { st = join_stats(P, (*nodes)[t.node], sl, sr); return true; }
        // below the spawn threshold nothing is handed over, so both sides are complete or neither side of THIS node spawned:
        // a side that did spawn has reported its complete pieces itself; report the other one
        std::lock_guard<std::mutex> g(mu);
        if (cl) done.emplace(lt.node, sl);
        if (cr) done.emplace(rt_.node, sr);
        return false;
    }
};

}  // namespace

void build_bvh2(const Aabb* prim_boxes, uint32_t n, const BuildParams& p, BuildResult& out) {
    out.nodes.clear(); out.prim_index.clear();
    out.depth = 0; out.n_leaves = 0; out.sah_cost = 0.f;
    if (n == 0) return;
    out.prim_index.resize(n);

    const bool timing = std::getenv("RT_HOST_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt host]   build: %-20s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
        tprev = now;
    };
    Builder b;
    b.P = p; b.nodes = &out.nodes;
    if (b.P.max_leaf < 1) b.P.max_leaf = 1;
    if (b.P.max_leaf > 8) b.P.max_leaf = 8;
    b.P.bins = std::min(std::max(b.P.bins, 4), kMaxBins);
    int nt = host_threads(p.threads);
    if (n < 4 * kSpawnThreshold) nt = 1;
    b.parallel = nt > 1;
    ForkJoin fj(nt);
    b.fj = &fj;
    // uninitialised: every node is written when it is handed out; room for the unused end of every thread's last block
    out.nodes.allocate(2 * (size_t)n + 1 + (size_t)kNodeBlock * ((size_t)nt + 2));

    lap("allocate nodes");
    // primitive records + the root's bounds
    std::unique_ptr<Prim[]> prims(new Prim[n]), tmp(new Prim[n]);      // uninitialised (320 MB each for 10 M triangles)
    b.prims = prims.get(); b.tmp = tmp.get();
    const uint32_t nc = (n + kParChunk - 1) / kParChunk;
    std::vector<Sides> part(nc);
    fj.run(nc, [&](uint32_t c) {
        const uint32_t lo = c * kParChunk, hi = std::min(n, lo + kParChunk);
        Sides s; s.reset();
        for (uint32_t i = lo; i < hi; ++i) {
            Prim& q = prims[i];
            for (int a = 0; a < 3; ++a) { q.lo[a] = prim_boxes[i].lo[a]; q.hi[a] = prim_boxes[i].hi[a]; }
            q.id = i; q.pad = 0;
            s.add(0, q);
        }
        part[c] = s;
    });
    Task root{ 0, 0, n, 0, Aabb{}, Aabb{}, false };
    root.box.reset(); root.cbox.reset();
    { Sides all; all.reset(); for (const Sides& s : part) all.merge(s); root.box = all.box[0].aabb(); root.cbox = all.cbox[0].aabb(); }

    lap("records + root bounds");
    // phase 1: the big nodes, breadth first, every pass over the whole pool; phase 2: the subtrees below as pool tasks
    Builder::NodeAlloc na0;                   // of this thread: phase 1, and phase 2 when there is no pool
    std::deque<Task> big;
    std::vector<Task> small;
    std::vector<size_t> done_leaves;          // entries of `small` that are leaves of phase 1 already (kept there for the final order)
    (n > kParNode && b.parallel ? (void)big.push_back(root) : (void)small.push_back(root));
    while (!big.empty()) {
        const Task t = big.front(); big.pop_front();
        Task lt, rt_;
        if (!b.split_node(t, true, na0, lt, rt_)) { b.done.emplace(t.node, b.leaf_stats(out.nodes[t.node])); small.push_back(t); done_leaves.push_back(small.size() - 1); continue; }
        for (const Task& ch : { lt, rt_ })
            if (ch.end - ch.begin > kParNode) big.push_back(ch); else small.push_back(ch);
    }
    lap("phase 1 (big nodes)");
    std::vector<Task> todo;
    for (size_t i = 0; i < small.size(); ++i)
        if (std::find(done_leaves.begin(), done_leaves.end(), i) == done_leaves.end()) todo.push_back(small[i]);
    if (!b.parallel) {
        for (const Task& t : todo) { Stats st; if (b.build(t, na0, st)) b.done.emplace(t.node, st); }
    } else {
        std::sort(todo.begin(), todo.end(), [](const Task& x, const Task& y) { return (x.end - x.begin) > (y.end - y.begin); });
        for (const Task& t : todo) b.spawn(t);
        fj.run((uint32_t)fj.size(), [&](uint32_t) { b.worker(); });
        if (b.failure) std::rethrow_exception(b.failure);
    }
    lap("phase 2 (subtrees)");
    out.nodes.resize_down(b.n_nodes.load());
    // the order of the primitives: every range below phase 1 stayed in the buffer it was handed over in
    fj.run((uint32_t)small.size(), [&](uint32_t k) {
        const Task& t = small[k];
        const Prim* src = b.buf(t.in_tmp);
        for (uint32_t i = t.begin; i < t.end; ++i) out.prim_index[i] = src[i].id;
    });

    // statistics: walk the few nodes above the subtrees that reported their own, combine along the tree
    struct Rec { uint32_t node; bool expanded; };
    std::vector<Rec> st;
    std::vector<Stats> val;            // value stack
    st.push_back({ 0u, false });
    while (!st.empty()) {
        Rec r = st.back(); st.pop_back();
        const BuildNode& nd = out.nodes[r.node];
        if (r.expanded) {
            const Stats sr = val.back(); val.pop_back();
            const Stats sl = val.back(); val.pop_back();
            val.push_back(Builder::join_stats(b.P, nd, sl, sr));
            continue;
        }
        auto it = b.done.find(r.node);
        if (it != b.done.end()) { val.push_back(it->second); continue; }
        if (nd.count) { val.push_back(b.leaf_stats(nd)); continue; }     // (every leaf above lies inside a reported subtree; kept for safety)
        st.push_back({ r.node, true });
        st.push_back({ nd.right, false });
        st.push_back({ nd.left, false });
    }
    lap("order + statistics");
    const Stats total = val.back();
    const float root_area = std::max(out.nodes[0].box.half_area(), 1e-30f);
    out.depth = total.height;
    out.n_leaves = total.leaves;
    out.sah_cost = (float)(total.cost / (double)root_area);
    out.oversize_leaf = b.oversize_leaf.load();
}

static inline int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

void flatten_bvh2(const BuildResult& b, float pad, FlatBvh2& out) {
    out.nodes.clear(); out.tri_order.clear();
    out.n_nodes = 0; out.n_leaves = b.n_leaves; out.depth = b.depth;
    if (b.nodes.empty()) return;

    // DFS pre-order numbering of interior nodes; leaves emit their triangles in visit order.
    const size_t nb = b.nodes.size();
    std::vector<int32_t> ref(nb, 0);
    std::vector<uint32_t> order;          // interior build nodes in flat order
    order.reserve(nb / 2 + 1);
    out.tri_order.reserve(b.prim_index.size());
    std::vector<uint32_t> st;
    st.push_back(0);
    while (!st.empty()) {
        uint32_t i = st.back(); st.pop_back();
        const BuildNode& nd = b.nodes[i];
        if (nd.count) {
            ref[i] = leaf_ref((uint32_t)out.tri_order.size(), nd.count);
            for (uint32_t k = 0; k < nd.count; ++k) out.tri_order.push_back(b.prim_index[nd.first + k]);
        } else {
            ref[i] = (int32_t)order.size();
            order.push_back(i);
            st.push_back(nd.right);
            st.push_back(nd.left);
        }
    }

    auto emit_child = [&](float* q, int c, const Aabb* box) {
        float lo[3], hi[3];
        if (box) { for (int a = 0; a < 3; ++a) { lo[a] = box->lo[a] - pad; hi[a] = box->hi[a] + pad; } }
        else     { for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }   // inverted: never hit
        float* qa = q + 4 * c;                  // q0 or q1
        qa[0] = lo[0]; qa[1] = hi[0]; qa[2] = lo[1]; qa[3] = hi[1];
        q[8 + 2 * c] = lo[2]; q[8 + 2 * c + 1] = hi[2];
    };

    if (order.empty()) {
        // the root itself is a leaf: wrap it in one interior node whose second child is empty
        out.n_nodes = 1;
        out.nodes.assign(16, 0.f);
        emit_child(out.nodes.data(), 0, &b.nodes[0].box);
        emit_child(out.nodes.data(), 1, nullptr);
        int32_t r0 = ref[0], r1 = leaf_ref(0, 1);
        std::memcpy(&out.nodes[12], &r0, 4);
        std::memcpy(&out.nodes[13], &r1, 4);
        out.depth = 1;
        return;
    }

    out.n_nodes = (uint32_t)order.size();
    out.nodes.assign((size_t)out.n_nodes * 16, 0.f);
    ForkJoin fj(out.n_nodes < 65536 ? 1 : host_threads(0));
    fj.range(out.n_nodes, 1u << 14, [&](uint64_t k_lo, uint64_t k_hi) {
        for (uint64_t k = k_lo; k < k_hi; ++k) {
            const BuildNode& nd = b.nodes[order[k]];
            float* q = &out.nodes[(size_t)k * 16];
            emit_child(q, 0, &b.nodes[nd.left].box);
            emit_child(q, 1, &b.nodes[nd.right].box);
            int32_t r0 = ref[nd.left], r1 = ref[nd.right];
            std::memcpy(q + 12, &r0, 4);
            std::memcpy(q + 13, &r1, 4);
        }
    });
}

uint32_t bvh2_stack_need(const BuildResult& b) {
    if (b.nodes.empty()) return 0;
    uint32_t need = 0;
    std::vector<std::pair<uint32_t, uint32_t>> st;      // (node, entries on the stack when it is entered)
    st.push_back({ 0u, 0u });
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        need = std::max(need, d);
        const BuildNode& nd = b.nodes[i];
        if (!nd.count) { st.push_back({ nd.left, d + 1 }); st.push_back({ nd.right, d + 1 }); }
    }
    return need;
}

namespace {

// Children of a wide node: start from the two children of `i` and keep opening the interior child with
// the largest surface area until there are four (or only leaves are left).
int gather4(const BuildResult& b, uint32_t i, uint32_t out[4]) {
    const BuildNode& nd = b.nodes[i];
    int n = 0;
    out[n++] = nd.left; out[n++] = nd.right;
    while (n < 4) {
        int best = -1; float best_area = -1.f;
        for (int k = 0; k < n; ++k) {
            const BuildNode& c = b.nodes[out[k]];
            if (c.count) continue;
            float a = c.box.half_area();
            if (a > best_area) { best_area = a; best = k; }
        }
        if (best < 0) break;
        const BuildNode& c = b.nodes[out[best]];
        out[best] = c.left;
        out[n++] = c.right;
    }
    return n;
}

}  // namespace

void flatten_bvh4q(const BuildResult& b, float pad, FlatBvh4q& out, bool want_wide, int threads) {
    out.nodes.clear(); out.tri_order.clear(); out.wide.clear();
    out.n_nodes = 0; out.n_leaves = b.n_leaves; out.depth = 0; out.stack_need = 0;
    if (b.nodes.empty()) return;

    // Wide nodes are numbered in DFS pre-order, and a wide node's leaf children put their triangles behind those of all wide
    // nodes before it.  Sequentially that is one walk; here the walk is cut below kTopDepth levels into independent subtrees
    // ("blocks"): a short sequential walk over the top, the blocks' sizes in parallel, a prefix over the pre-order, the
    // blocks' walks in parallel, and finally one flat parallel loop that quantises and writes every node.  Same numbering,
    // same bytes, for any thread count.
    struct Wide { uint32_t src; uint32_t child[4]; int n; };
    struct Item { uint32_t src, depth, stack; bool block; uint32_t n_wide, n_tris, base_wide, base_tri; };
    constexpr uint32_t kTopDepth = 6;
    ForkJoin fj(b.nodes.size() < 65536 ? 1 : host_threads(threads));
    std::vector<int32_t> ref(b.nodes.size(), 0);       // reference of every build node that becomes a child
    std::vector<Wide> wide;
    out.tri_order.resize(b.prim_index.size());

    auto emit_leaf = [&](uint32_t i, uint32_t& tri_at) {
        const BuildNode& nd = b.nodes[i];
        ref[i] = leaf_ref(tri_at, nd.count);
        for (uint32_t k = 0; k < nd.count; ++k) out.tri_order[tri_at + k] = b.prim_index[nd.first + k];
        tri_at += nd.count;
    };
    if (b.nodes[0].count) {
        // the root is a leaf: one wide node with a single child
        uint32_t at = 0;
        emit_leaf(0, at);
        Wide w{}; w.src = 0; w.n = 1; w.child[0] = 0;
        wide.push_back(w);
        out.depth = 1;
        out.stack_need = 3;
    } else {
        // 1. the top of the tree, in pre-order; interior children at kTopDepth become blocks
        std::vector<Item> items;
        {
            std::vector<Item> st;
            st.push_back(Item{ 0u, 1u, 0u, false, 0, 0, 0, 0 });
            while (!st.empty()) {
                Item it = st.back(); st.pop_back();
                it.block = it.depth > kTopDepth;
                items.push_back(it);
                if (it.block) continue;
                uint32_t ch[4];
                const int n = gather4(b, it.src, ch);
                for (int k = n - 1; k >= 0; --k)
                    if (b.nodes[ch[k]].count == 0) st.push_back(Item{ ch[k], it.depth + 1, it.stack + 3u, false, 0, 0, 0, 0 });
            }
        }
        // 2. sizes of the blocks: wide nodes and triangles below each
        fj.run((uint32_t)items.size(), [&](uint32_t i) {
            Item& it = items[i];
            if (!it.block) return;
            uint32_t nw = 0, nt = 0;
            std::vector<uint32_t> st{ it.src };
            while (!st.empty()) {
                const uint32_t src = st.back(); st.pop_back();
                ++nw;
                uint32_t ch[4];
                const int n = gather4(b, src, ch);
                for (int k = 0; k < n; ++k) { const BuildNode& c = b.nodes[ch[k]]; if (c.count) nt += c.count; else st.push_back(ch[k]); }
            }
            it.n_wide = nw; it.n_tris = nt;
        });
        // 3. pre-order prefix: index of every top node and base of every block; the top nodes' own records and leaves
        uint32_t run_w = 0, run_t = 0;
        for (Item& it : items) {
            it.base_wide = run_w; it.base_tri = run_t;
            if (it.block) { run_w += it.n_wide; run_t += it.n_tris; continue; }
            ++run_w;
            uint32_t ch[4];
            const int n = gather4(b, it.src, ch);
            for (int k = 0; k < n; ++k) run_t += b.nodes[ch[k]].count;
        }
        wide.resize(run_w);
        uint32_t depth = 0, stack_need = 0;
        for (const Item& it : items) {
            if (it.block) continue;
            Wide w{}; w.src = it.src;
            w.n = gather4(b, it.src, w.child);
            ref[it.src] = (int32_t)it.base_wide;
            wide[it.base_wide] = w;
            depth = std::max(depth, it.depth);
            // A visit may push up to THREE slots before descending, whatever the node's child count: the per-ray pad
            // of the kernels (kernels.hip make_slab) can exceed a small node's whole extent for a ray that starts very far
            // away, and then even its empty (inverted) slots pass the slab test and are pushed (they refer to triangle
            // 0, which is harmless to test).  The bound must cover that, or such a ray overruns its stack.
            stack_need = std::max(stack_need, it.stack + 3u);
            uint32_t at = it.base_tri;
            for (int k = 0; k < w.n; ++k) if (b.nodes[w.child[k]].count) emit_leaf(w.child[k], at);     // children of one node stay adjacent in memory
        }
        // 4. the blocks: the same walk, numbering from the block's base
        std::vector<uint32_t> bdepth(items.size(), 0), bstack(items.size(), 0);
        fj.run((uint32_t)items.size(), [&](uint32_t i) {
            const Item& blk = items[i];
            if (!blk.block) return;
            struct Ent { uint32_t src, depth, stack; };
            std::vector<Ent> st{ Ent{ blk.src, blk.depth, blk.stack } };
            uint32_t at_w = blk.base_wide, at_t = blk.base_tri, d = 0, sn = 0;
            while (!st.empty()) {
                const Ent e = st.back(); st.pop_back();
                Wide w{}; w.src = e.src;
                w.n = gather4(b, e.src, w.child);
                ref[e.src] = (int32_t)at_w;
                wide[at_w++] = w;
                d = std::max(d, e.depth);
                sn = std::max(sn, e.stack + 3u);
                for (int k = 0; k < w.n; ++k) if (b.nodes[w.child[k]].count) emit_leaf(w.child[k], at_t);
                for (int k = w.n - 1; k >= 0; --k)
                    if (b.nodes[w.child[k]].count == 0) st.push_back(Ent{ w.child[k], e.depth + 1, e.stack + 3u });
            }
            bdepth[i] = d; bstack[i] = sn;
        });
        for (size_t i = 0; i < items.size(); ++i) { depth = std::max(depth, bdepth[i]); stack_need = std::max(stack_need, bstack[i]); }
        out.depth = depth;
        out.stack_need = stack_need;
    }

    // 5. quantise and write every node (independent of each other once the references are known)
    out.n_nodes = (uint32_t)wide.size();
    out.nodes.assign((size_t)out.n_nodes * 16, 0u);
    if (want_wide) out.wide.assign((size_t)out.n_nodes * 32, 0u);
    fj.range(out.n_nodes, 4096, [&](uint64_t w_lo, uint64_t w_hi) {
    for (uint32_t wi = (uint32_t)w_lo; wi < (uint32_t)w_hi; ++wi) {
        const Wide& w = wide[wi];
        uint32_t* q = &out.nodes[(size_t)wi * 16];
        float lo[4][3], hi[4][3], org[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, top[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (int k = 0; k < w.n; ++k) {
            const Aabb& bx = b.nodes[w.child[k]].box;
            for (int a = 0; a < 3; ++a) {
                lo[k][a] = bx.lo[a] - pad; hi[k][a] = bx.hi[a] + pad;
                org[a] = std::min(org[a], lo[k][a]); top[a] = std::max(top[a], hi[k][a]);
            }
        }
        float scale[3];
        for (int a = 0; a < 3; ++a) {
            // grid step of the axis: the smallest float sc with origin + 255 * sc >= top (checked in double; all inputs are
            // floats).  Not a power of two: a step of ext/255 instead of up to twice that keeps the quantised boxes tight.
            const double ext = (double)top[a] - (double)org[a];
            float sc = 1.0e-30f;
            if (ext > 0.0) {
                const double want = std::min(ext / 255.0, 1.0e36);
                sc = (float)want;
                if ((double)sc < want) sc = std::nextafterf(sc, 3.0e38f);
                if (!(sc > 0.f)) sc = 1.0e-30f;
                while ((double)sc * 255.0 < ext && sc < 1.0e36f) sc = std::nextafterf(sc, 3.0e38f);
            }
            scale[a] = sc;
            const double scd = (double)sc;
            for (int k = 0; k < w.n; ++k) {
                double ql = std::floor(((double)lo[k][a] - (double)org[a]) / scd);
                double qh = std::ceil(((double)hi[k][a] - (double)org[a]) / scd);
                if (ql < 0) ql = 0;
                if (ql > 255) ql = 255;
                if (qh > 255) qh = 255;
                if (qh < 0) qh = 0;
                // guard the conservative property against the rounding of the divisions above
                while (ql > 0 && (double)org[a] + ql * scd > (double)lo[k][a]) ql -= 1;
                while (qh < 255 && (double)org[a] + qh * scd < (double)hi[k][a]) qh += 1;
                const uint32_t l8 = (uint32_t)ql, h8 = (uint32_t)qh;
                const int lo_word = 8 + a;                       // q2.x/y/z = lo planes
                const int hi_word = (a == 0) ? 11 : 12 + (a - 1);  // q2.w = hi.x, q3.x = hi.y, q3.y = hi.z
                q[lo_word] |= l8 << (8 * k);
                q[hi_word] |= h8 << (8 * k);
            }
        }
        if (want_wide) {
            uint32_t* wq = &out.wide[(size_t)wi * 32];
            wq[4] = (uint32_t)w.n;
            const float far = 3.0e38f;        // empty slot: a point at "infinity", missed by arithmetic whatever the ray's signs
            for (int k = 0; k < 4; ++k)
                for (int a = 0; a < 3; ++a) {
                    std::memcpy(&wq[8 + 6 * k + 2 * a], k < w.n ? &lo[k][a] : &far, 4);
                    std::memcpy(&wq[8 + 6 * k + 2 * a + 1], k < w.n ? &hi[k][a] : &far, 4);
                }
        }
        std::memcpy(&q[0], &org[0], 4); std::memcpy(&q[1], &org[1], 4); std::memcpy(&q[2], &org[2], 4);
        std::memcpy(&q[3], &scale[0], 4); std::memcpy(&q[14], &scale[1], 4); std::memcpy(&q[15], &scale[2], 4);
        for (int k = 0; k < 4; ++k) {
            int32_t r = (k < w.n) ? ref[w.child[k]] : leaf_ref(0, 1);
            std::memcpy(&q[4 + k], &r, 4);
            if (want_wide) std::memcpy(&out.wide[(size_t)wi * 32 + k], &r, 4);
            if (k >= w.n) {                                   // empty slot: inverted box (lo 255, hi 0) never hits
                for (int a = 0; a < 3; ++a) q[8 + a] |= 0xffu << (8 * k);
            }
        }
    }
    });
}

}  // namespace rt
