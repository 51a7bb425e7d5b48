// scene_host.cpp -- host half of rt_scene_upload (src/vulkan/VkRenderer.cpp:260-326 hands over the reference's
// Triangle / Sphere / Light arrays as they are): exact bounds, BVH build, flatten, and the device records.
// No HIP in this file (scene_host.hpp).
#include "scene_host.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

#include "../../include/raytracer_amd.h"
#include "../host/rt_types.hpp"
#include "abi_guard.hpp"
#include "parallel.hpp"

namespace rt {

int prepare_scene(const void* tris, uint32_t n_tris, uint32_t tri_stride, const void* spheres, uint32_t n_spheres,
                  uint32_t sphere_stride, const void* lights, uint32_t n_lights, uint32_t light_stride,
                  const HostSceneParams& p, HostScene& out, std::string& err) {
    out = HostScene{};
    out.box.reset();
    // RT_HOST_TIMING=1: phase times of the host half on stderr (measurement aid)
    const bool timing = std::getenv("RT_HOST_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rt host] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
        tprev = now;
    };
    // 1. exact triangle bounds (positions at byte offsets 0 / 32 / 64 of the 144-B struct)
    // (the loops over all triangles run on a small fork-join pool: per-piece results, combined in piece order)
    ForkJoin fj(n_tris < 65536 ? 1 : host_threads(p.threads));
    std::unique_ptr<Aabb[]> boxes(new Aabb[std::max<uint32_t>(n_tris, 1)]);
    float maxabs = 0.f;
    const char* tb = static_cast<const char*>(tris);
    {
        constexpr uint32_t kPiece = 1u << 16;
        const uint32_t n_pieces = (n_tris + kPiece - 1) / kPiece;
        struct Piece { Aabb box; float maxabs; uint32_t bad; };
        std::vector<Piece> pieces(n_pieces);
        fj.run(n_pieces, [&](uint32_t c) {
            Piece pc; pc.box.reset(); pc.maxabs = 0.f; pc.bad = 0xFFFFFFFFu;
            const uint32_t lo = c * kPiece, hi = std::min(n_tris, lo + kPiece);
            for (uint32_t i = lo; i < hi; ++i) {
                const char* t = tb + (size_t)i * tri_stride;
                Aabb b; b.reset();
                for (int v = 0; v < 3; ++v) {
                    float q[3];
                    std::memcpy(q, t + 32 * v, 12);
                    for (int a = 0; a < 3; ++a) {
                        if (!std::isfinite(q[a]) && pc.bad == 0xFFFFFFFFu) pc.bad = i;
                        pc.maxabs = std::max(pc.maxabs, std::fabs(q[a]));
                    }
                    b.grow(q);
                }
                boxes[i] = b;
                pc.box.grow(b);
            }
            pieces[c] = pc;
        });
        for (const Piece& pc : pieces) {
            if (pc.bad != 0xFFFFFFFFu) { err = "triangle " + std::to_string(pc.bad) + " has a non-finite position"; return RT_E_INVALID; }
            maxabs = std::max(maxabs, pc.maxabs);
            out.box.grow(pc.box);
        }
    }
    out.maxabs = maxabs;
    lap("triangle bounds");
    // 2. BVH2 build + flatten
    BuildParams bp;
    bp.max_leaf = p.max_leaf;
    bp.max_depth = p.max_depth;
    bp.threads = p.threads;
    bp.bins = p.bins;
    bp.cost_trav = p.cost_trav;
    BuildResult br;
    build_bvh2(boxes.get(), n_tris, bp, br);
    lap("build_bvh2");
    boxes.reset();
    if (br.oversize_leaf) { err = "internal: leaf with more than 8 triangles (depth budget exceeded)"; return RT_E_TOO_LARGE; }
    out.sah_cost = br.sah_cost;
    int layout = p.layout;
    const float pad = n_tris ? std::ldexp(std::max(maxabs, 1e-30f), -p.pad_log2) : 0.f;
    if (layout == 3) {
        FlatBvh4q f4;
        flatten_bvh4q(br, pad, f4, p.wide && p.records, p.threads);
        if (f4.stack_need + 2 > 64) {
            layout = 2;                         // pathological tree: its 4-wide stack would not fit 64 KB of LDS
        } else {
            out.nodes.swap(f4.nodes); out.tri_order.swap(f4.tri_order); out.wide.swap(f4.wide);
            out.n_nodes = f4.n_nodes; out.n_leaves = f4.n_leaves; out.depth = f4.depth; out.stack_need = f4.stack_need;
        }
    }
    if (layout == 2) {
        FlatBvh2 f2;
        flatten_bvh2(br, pad, f2);
        out.nodes.resize(f2.nodes.size());
        if (!f2.nodes.empty()) std::memcpy(out.nodes.data(), f2.nodes.data(), f2.nodes.size() * 4);
        out.tri_order.swap(f2.tri_order);
        out.n_nodes = f2.n_nodes; out.n_leaves = f2.n_leaves; out.depth = f2.depth;
        out.stack_need = std::max(f2.depth, bvh2_stack_need(br));
    }
    out.layout = layout;
    lap("flatten");
    if (out.stack_need + 2 > 64) { err = "internal: traversal stack would exceed the LDS budget"; return RT_E_TOO_LARGE; }
    if (!p.records) return RT_OK;
    br.nodes.clear();

    // 3. device records
    out.tri48.resize((size_t)n_tris * 12);
    out.nrm48.resize((size_t)n_tris * 12);
    out.sph.resize((size_t)n_spheres * 4);
    fj.range(n_tris, 1u << 15, [&](uint64_t k_lo, uint64_t k_hi) {
    for (uint64_t k = k_lo; k < k_hi; ++k) {
        const uint32_t id = out.tri_order[k];
        const char* t = tb + (size_t)id * tri_stride;
        float p0[3], p1[3], p2[3];
        std::memcpy(p0, t, 12); std::memcpy(p1, t + 32, 12); std::memcpy(p2, t + 64, 12);
        float* o = &out.tri48[(size_t)k * 12];
        o[0] = p0[0]; o[1] = p0[1]; o[2] = p0[2]; std::memcpy(&o[3], &id, 4);
        // frag.glsl:132-133: edge1 = v1 - v0, edge2 = v2 - v0 (plain fp32 subtraction, no contraction possible)
        o[4] = p1[0] - p0[0]; o[5] = p1[1] - p0[1]; o[6] = p1[2] - p0[2]; o[7] = 0.f;
        o[8] = p2[0] - p0[0]; o[9] = p2[1] - p0[1]; o[10] = p2[2] - p0[2]; o[11] = 0.f;
    }
    });
    lap("triangle records");
    // distinct materials (src/math/Material.h:6-11, 9 floats used) -> table; ids ride in normals[3*i].w / sphere_mat.
    // Ids follow the order of first appearance (triangles, then spheres), whatever the thread count: every piece collects
    // its distinct materials with the first triangle that uses each, the pieces are merged in order, then a second
    // sweep writes normals and ids against the finished (read-only) table.
    typedef std::array<uint32_t, 9> MatKey;
    auto mat_key = [](const char* m48, float rec[12]) -> MatKey {
        float f[10];
        std::memcpy(f, m48, 40);                   // albedo@0, emission@16, strength@28, roughness@32, metallic@36
        MatKey key;
        const float vals[9] = { f[0], f[1], f[2], f[4], f[5], f[6], f[7], f[8], f[9] };
        std::memcpy(key.data(), vals, 36);
        const float r[12] = { f[0], f[1], f[2], f[7], f[4], f[5], f[6], f[8], f[9], 0.f, 0.f, 0.f };
        std::memcpy(rec, r, sizeof r);
        return key;
    };
    std::map<MatKey, uint32_t> mat_ids;
    auto material_id = [&](const char* m48) -> uint32_t {          // sequential use only (merge, spheres)
        float rec[12];
        const MatKey key = mat_key(m48, rec);
        auto it = mat_ids.find(key);
        if (it != mat_ids.end()) return it->second;
        const uint32_t id = (uint32_t)mat_ids.size();
        mat_ids.emplace(key, id);
        out.mats.insert(out.mats.end(), rec, rec + 12);
        return id;
    };
    {
        constexpr uint32_t kPiece = 1u << 16;
        const uint32_t n_pieces = (n_tris + kPiece - 1) / kPiece;
        std::vector<std::vector<uint32_t>> firsts(n_pieces);      // per piece: triangles that introduce a material, ascending
        fj.run(n_pieces, [&](uint32_t c) {
            std::map<MatKey, uint32_t> seen;
            const uint32_t lo = c * kPiece, hi = std::min(n_tris, lo + kPiece);
            MatKey last{}; bool have_last = false;
            for (uint32_t i = lo; i < hi; ++i) {
                float rec[12];
                const MatKey key = mat_key(tb + (size_t)i * tri_stride + 96, rec);
                if (have_last && key == last) continue;
                last = key; have_last = true;
                if (seen.emplace(key, i).second) firsts[c].push_back(i);
            }
        });
        for (const auto& v : firsts) for (uint32_t i : v) (void)material_id(tb + (size_t)i * tri_stride + 96);
        fj.range(n_tris, 1u << 15, [&](uint64_t i_lo, uint64_t i_hi) {
            MatKey last{}; uint32_t last_id = 0; bool have_last = false;
            for (uint64_t i = i_lo; i < i_hi; ++i) {
                const char* t = tb + (size_t)i * tri_stride;
                float* o = &out.nrm48[(size_t)i * 12];
                for (int v = 0; v < 3; ++v) { std::memcpy(o + 4 * v, t + 16 + 32 * v, 12); o[4 * v + 3] = 0.f; }
                float rec[12];
                const MatKey key = mat_key(t + 96, rec);
                if (!have_last || !(key == last)) { last = key; last_id = mat_ids.find(key)->second; have_last = true; }
                std::memcpy(&o[3], &last_id, 4);
            }
        });
    }
    lap("normals + materials");
    out.sphere_mat.resize(n_spheres);
    out.lights8.resize((size_t)n_lights * 8);
    const char* sb = static_cast<const char*>(spheres);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const uint32_t id = material_id(sb + (size_t)i * sphere_stride + 16);
        std::memcpy(&out.sphere_mat[i], &id, 4);
        std::memcpy(&out.sph[(size_t)i * 4], sb + (size_t)i * sphere_stride, 16);
    }
    for (uint32_t i = 0; i < n_lights; ++i) {      // src/math/Light.h:6-10: position@0, colour@16, intensity@28
        const char* l = static_cast<const char*>(lights) + (size_t)i * light_stride;
        float* o = &out.lights8[(size_t)i * 8];
        std::memcpy(o, l, 12); std::memcpy(o + 3, l + 28, 4);
        std::memcpy(o + 4, l + 16, 12); o[7] = 0.f;
    }
    return RT_OK;
}

}  // namespace rt

using namespace rt;

// Host-only BVH build + flatten (no device): what rt_scene_upload runs before copying to HBM.
extern "C" int rt_bvh_build_host(const void* tris, uint32_t n_tris, uint32_t tri_stride, int layout, int max_leaf, int pad_log2,
                                 int threads, rt_scene_stats* stats, uint32_t* nodes_out, uint64_t nodes_cap_words,
                                 uint32_t* tri_order_out) try {
    if (!stats || (n_tris && (!tris || tri_stride < sizeof(Triangle) || (tri_stride & 3)))) return RT_E_INVALID;
    if (n_tris >= (1u << 28)) return RT_E_TOO_LARGE;
    if (layout != 2 && layout != 3) return RT_E_INVALID;
    if (pad_log2 > 20 || max_leaf > 8) return RT_E_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    HostSceneParams p;
    p.layout = layout; p.max_leaf = max_leaf > 0 ? max_leaf : 4; p.pad_log2 = pad_log2 > 0 ? std::max(pad_log2, 4) : 15;
    p.threads = threads; p.records = false;
    HostScene hs;
    std::string err;
    const int rc = prepare_scene(tris, n_tris, tri_stride, nullptr, 0, 0, nullptr, 0, 0, p, hs, err);
    if (rc != RT_OK) return rc;
    if (hs.layout != layout) return RT_E_TOO_LARGE;        // the 4-wide stack bound was exceeded
    *stats = rt_scene_stats{};
    stats->n_tris = n_tris; stats->sah_cost = hs.sah_cost; stats->node_bytes = 64; stats->tri_bytes = 48;
    stats->n_nodes = hs.n_nodes; stats->n_leaves = hs.n_leaves; stats->max_depth = hs.depth;
    stats->stack_entries = hs.stack_need + 2;              // the traversal-stack bound (pushes + sentinel + spare)
    stats->build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    stats->device_bytes = (uint64_t)hs.nodes.size() * 4 + (uint64_t)n_tris * 96;
    if (nodes_out) {
        if (nodes_cap_words < hs.nodes.size()) return RT_E_TOO_LARGE;
        if (!hs.nodes.empty()) std::memcpy(nodes_out, hs.nodes.data(), hs.nodes.size() * 4);
    }
    if (tri_order_out && !hs.tri_order.empty()) std::memcpy(tri_order_out, hs.tri_order.data(), hs.tri_order.size() * 4);
    return RT_OK;
} catch (...) {
    return abi_exception(nullptr);
}
