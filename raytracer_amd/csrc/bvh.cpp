// bvh.cpp -- binned-SAH BVH2 build (multi-threaded, depth-bounded) and flattening.
// Build-original; see bvh.hpp for the contract.
#include "bvh.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <exception>
#include <system_error>
#include <mutex>
#include <thread>

namespace rt {
namespace {

constexpr int kMaxBins = 32;        // BuildParams::bins is clamped to [4, kMaxBins]
constexpr uint32_t kSpawnThreshold = 16384;   // subtrees larger than this become pool tasks

struct Task { uint32_t node, begin, end; int depth; };

struct Builder {
    const Aabb* pb;
    BuildParams P;
    std::vector<BuildNode>* nodes;
    std::vector<uint32_t>* idx;
    std::atomic<uint32_t> n_nodes{1};

    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    int pending = 0;          // tasks queued or running
    bool parallel = false;
    std::exception_ptr failure;   // first exception of a worker (queue growth can throw); rethrown by build_bvh2

    static int ceil_log2(uint32_t x) { int l = 0; uint32_t v = 1; while (v < x) { v <<= 1; ++l; } return l; }

    void spawn(const Task& t) {
        std::lock_guard<std::mutex> g(mu);
        queue.push_back(t);
        ++pending;
        cv.notify_one();
    }

    void worker() {
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
                try { build(t.node, t.begin, t.end, t.depth); }
                catch (...) { std::lock_guard<std::mutex> g(mu); if (!failure) failure = std::current_exception(); }
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (--pending == 0) cv.notify_all();
            }
        }
    }

    inline float centroid(uint32_t prim, int a) const { return 0.5f * (pb[prim].lo[a] + pb[prim].hi[a]); }

    void make_leaf(BuildNode& n, uint32_t begin, uint32_t end) {
        n.first = begin;
        n.count = end - begin;
    }

    void build(uint32_t node_i, uint32_t begin, uint32_t end, int depth) {
        uint32_t* I = idx->data();
        BuildNode& node = (*nodes)[node_i];
        const uint32_t n = end - begin;

        Aabb box, cbox;
        box.reset(); cbox.reset();
        for (uint32_t i = begin; i < end; ++i) {
            const Aabb& b = pb[I[i]];
            box.grow(b);
            float c[3] = { 0.5f * (b.lo[0] + b.hi[0]), 0.5f * (b.lo[1] + b.hi[1]), 0.5f * (b.lo[2] + b.hi[2]) };
            cbox.grow(c);
        }
        node.box = box;
        node.count = 0;

        if (n <= 1 || depth >= P.max_depth) { make_leaf(node, begin, end); return; }

        // depth budget: a balanced split of n prims needs ceil_log2(ceil(n/max_leaf)) more levels
        const int need = ceil_log2((n + P.max_leaf - 1) / P.max_leaf);
        const bool force_balanced = (depth + need >= P.max_depth);

        int best_axis = -1, best_plane = -1;
        float best_cost = 3.0e38f;
        if (!force_balanced) {
            const float inv_area = 1.0f / std::max(box.half_area(), 1e-30f);
            for (int a = 0; a < 3; ++a) {
                const float ext = cbox.hi[a] - cbox.lo[a];
                if (!(ext > 0.f)) continue;
                const int kBins = P.bins;
                const float scale = (float)kBins / ext;
                Aabb bb[kMaxBins]; uint32_t bc[kMaxBins];
                for (int k = 0; k < kBins; ++k) { bb[k].reset(); bc[k] = 0; }
                for (uint32_t i = begin; i < end; ++i) {
                    const Aabb& b = pb[I[i]];
                    int k = (int)((0.5f * (b.lo[a] + b.hi[a]) - cbox.lo[a]) * scale);
                    k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                    bb[k].grow(b); ++bc[k];
                }
                float ra[kMaxBins]; uint32_t rc[kMaxBins];
                Aabb acc; acc.reset(); uint32_t cnt = 0;
                for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); cnt += bc[k]; ra[k] = cnt ? acc.half_area() : 0.f; rc[k] = cnt; }
                acc.reset(); cnt = 0;
                for (int k = 0; k < kBins - 1; ++k) {
                    acc.grow(bb[k]); cnt += bc[k];
                    if (cnt == 0 || rc[k + 1] == 0) continue;
                    float cost = P.cost_trav + P.cost_isect * inv_area * (acc.half_area() * (float)cnt + ra[k + 1] * (float)rc[k + 1]);
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = k; }
                }
            }
            const float leaf_cost = P.cost_isect * (float)n;
            if (n <= (uint32_t)P.max_leaf && (best_axis < 0 || leaf_cost <= best_cost)) { make_leaf(node, begin, end); return; }
        }

        uint32_t mid;
        if (best_axis >= 0) {
            const int a = best_axis;
            const int kBins = P.bins;
            const float scale = (float)kBins / (cbox.hi[a] - cbox.lo[a]);
            const float lo = cbox.lo[a];
            const int plane = best_plane;
            uint32_t* m = std::partition(I + begin, I + end, [&](uint32_t prim) {
                int k = (int)((0.5f * (pb[prim].lo[a] + pb[prim].hi[a]) - lo) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= plane;
            });
            mid = (uint32_t)(m - I);
        } else {
            // object-median split along the widest centroid axis (also the forced-balanced path)
            int a = 0;
            float e0 = cbox.hi[0] - cbox.lo[0], e1 = cbox.hi[1] - cbox.lo[1], e2 = cbox.hi[2] - cbox.lo[2];
            if (e1 > e0) { a = 1; e0 = e1; }
            if (e2 > e0) a = 2;
            if (n <= (uint32_t)P.max_leaf && !force_balanced) { make_leaf(node, begin, end); return; }
            if (n <= (uint32_t)P.max_leaf) { make_leaf(node, begin, end); return; }
            mid = begin + n / 2;
            std::nth_element(I + begin, I + mid, I + end, [&](uint32_t x, uint32_t y) {
                float cx = pb[x].lo[a] + pb[x].hi[a], cy = pb[y].lo[a] + pb[y].hi[a];
                return cx < cy || (cx == cy && x < y);
            });
        }
        if (mid == begin || mid == end) mid = begin + n / 2;

        const uint32_t c = n_nodes.fetch_add(2);
        node.left = c; node.right = c + 1;
        if (parallel && (mid - begin) > kSpawnThreshold && (end - mid) > kSpawnThreshold) {
            spawn(Task{ c + 1, mid, end, depth + 1 });
            build(c, begin, mid, depth + 1);
        } else {
            build(c, begin, mid, depth + 1);
            build(c + 1, mid, end, depth + 1);
        }
    }
};

}  // namespace

void build_bvh2(const Aabb* prim_boxes, uint32_t n, const BuildParams& p, BuildResult& out) {
    out.nodes.clear(); out.prim_index.clear();
    out.depth = 0; out.n_leaves = 0; out.sah_cost = 0.f;
    if (n == 0) return;
    out.nodes.resize(2 * (size_t)n + 1);
    out.prim_index.resize(n);
    for (uint32_t i = 0; i < n; ++i) out.prim_index[i] = i;

    Builder b;
    b.pb = prim_boxes; b.P = p; b.nodes = &out.nodes; b.idx = &out.prim_index;
    if (b.P.max_leaf < 1) b.P.max_leaf = 1;
    if (b.P.max_leaf > 8) b.P.max_leaf = 8;
    b.P.bins = std::min(std::max(b.P.bins, 4), kMaxBins);
    int nt = p.threads > 0 ? p.threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 4 * kSpawnThreshold) nt = 1;
    b.parallel = nt > 1;

    if (!b.parallel) {
        b.build(0, 0, n, 0);
    } else {
        b.spawn(Task{ 0, 0, n, 0 });
        // nt - 1 helpers + this thread; a helper that cannot be started is simply missing
        std::vector<std::thread> pool;
        pool.reserve(nt);
        try { for (int t = 1; t < nt; ++t) pool.emplace_back([&b] { b.worker(); }); } catch (...) {}      // std::system_error or std::bad_alloc
        b.worker();
        for (auto& t : pool) t.join();
        if (b.failure) std::rethrow_exception(b.failure);
    }
    out.nodes.resize(b.n_nodes.load());

    // stats (iterative DFS)
    const float root_area = std::max(out.nodes[0].box.half_area(), 1e-30f);
    std::vector<std::pair<uint32_t, uint32_t>> st;
    st.push_back({ 0u, 0u });
    double cost = 0.0;
    while (!st.empty()) {
        auto [i, d] = st.back(); st.pop_back();
        const BuildNode& nd = out.nodes[i];
        if (nd.count) {
            ++out.n_leaves;
            out.depth = std::max(out.depth, d);
            cost += (double)p.cost_isect * nd.count * nd.box.half_area() / root_area;
        } else {
            cost += (double)p.cost_trav * nd.box.half_area() / root_area;
            st.push_back({ nd.left, d + 1 });
            st.push_back({ nd.right, d + 1 });
        }
    }
    out.sah_cost = (float)cost;
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
    for (uint32_t k = 0; k < out.n_nodes; ++k) {
        const BuildNode& nd = b.nodes[order[k]];
        float* q = &out.nodes[(size_t)k * 16];
        emit_child(q, 0, &b.nodes[nd.left].box);
        emit_child(q, 1, &b.nodes[nd.right].box);
        int32_t r0 = ref[nd.left], r1 = ref[nd.right];
        std::memcpy(q + 12, &r0, 4);
        std::memcpy(q + 13, &r1, 4);
    }
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

void flatten_bvh4q(const BuildResult& b, float pad, FlatBvh4q& out, bool want_wide) {
    out.nodes.clear(); out.tri_order.clear(); out.wide.clear();
    out.n_nodes = 0; out.n_leaves = b.n_leaves; out.depth = 0; out.stack_need = 0;
    if (b.nodes.empty()) return;

    struct Wide { uint32_t src; uint32_t child[4]; int n; };
    std::vector<Wide> wide;
    std::vector<int32_t> ref(b.nodes.size(), 0);       // reference of every build node that becomes a child
    out.tri_order.reserve(b.prim_index.size());

    // DFS pre-order over wide nodes; (build node, depth, stack entries at entry)
    struct Item { uint32_t src, depth, stack; };
    std::vector<Item> st;
    auto emit_leaf = [&](uint32_t i) {
        const BuildNode& nd = b.nodes[i];
        ref[i] = leaf_ref((uint32_t)out.tri_order.size(), nd.count);
        for (uint32_t k = 0; k < nd.count; ++k) out.tri_order.push_back(b.prim_index[nd.first + k]);
    };
    if (b.nodes[0].count) {
        // the root is a leaf: one wide node with a single child
        emit_leaf(0);
        Wide w{}; w.src = 0; w.n = 1; w.child[0] = 0;
        wide.push_back(w);
        out.depth = 1;
        out.stack_need = 3;
    } else {
        st.push_back({ 0u, 1u, 0u });
        while (!st.empty()) {
            Item it = st.back(); st.pop_back();
            Wide w{}; w.src = it.src;
            w.n = gather4(b, it.src, w.child);
            ref[it.src] = (int32_t)wide.size();
            wide.push_back(w);
            out.depth = std::max(out.depth, it.depth);
            // A visit may push up to THREE slots before descending, whatever the node's child count: the per-ray pad
            // of the kernels (kernels.hip make_slab) can exceed a small node's whole extent for a ray that starts very far
            // away, and then even its empty (inverted) slots pass the slab test and are pushed (they refer to triangle
            // 0, which is harmless to test).  The bound must cover that, or such a ray overruns its stack.
            out.stack_need = std::max(out.stack_need, it.stack + 3u);
            for (int k = w.n - 1; k >= 0; --k) {
                if (b.nodes[w.child[k]].count == 0) st.push_back({ w.child[k], it.depth + 1, it.stack + 3u });
            }
            // leaves are emitted when the wide node is written (below) to keep triangles of one node adjacent
        }
    }
    // second pass in the same order: leaves' triangle ranges (children of node k stay adjacent in memory)
    if (!b.nodes[0].count)
        for (const Wide& w : wide)
            for (int k = 0; k < w.n; ++k)
                if (b.nodes[w.child[k]].count) emit_leaf(w.child[k]);

    out.n_nodes = (uint32_t)wide.size();
    out.nodes.assign((size_t)out.n_nodes * 16, 0u);
    if (want_wide) out.wide.assign((size_t)out.n_nodes * 32, 0u);
    for (uint32_t wi = 0; wi < out.n_nodes; ++wi) {
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
}

}  // namespace rt
