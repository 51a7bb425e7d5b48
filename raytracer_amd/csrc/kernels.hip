// kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the ray hot path.
//
// What is computed follows shaders/frag.glsl of the reference (paths relative to /root/reference):
//   getCameraRay :89-128, rayIntersectsTriangle :130-156, rayIntersectsSphere :158-183,
//   traceRay :265-301, shadow query :321-330, sampleHemisphere :185-203 + offset :340-343.
// How it is computed is build-original: one ray per lane of a wave64, 8x8-pixel tiles per wave for
// primary rays, a per-lane traversal stack in LDS ([entry][thread] so every lane owns one bank
// column), interior nodes as 64-B SoA child pairs, triangles as 48-B (v0,e1,e2) records.
//
// Canonical fp32 evaluation order (must be compiled with -ffp-contract=off; the CPU oracle fixes the
// same order independently, DESIGN.md "Arithmetic contract"):
//   dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));  cross(a,b).x = fma(a.y,b.z, -(a.z*b.y)) ...
//   normalize(v) = v / sqrt(dot(v,v));  o + t*d = fma(t,d,o)
// Division and sqrt are IEEE correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the default).
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace rt {
namespace {

constexpr int kBlock = 256;
constexpr float kEps = 1e-6f;              // frag.glsl:131
constexpr int kDone = (int)0x80000000;   // stack-bottom sentinel (never a valid leaf reference: first tri < 2^28)
constexpr uint32_t kKeyFlip = 0x80000000u; // tie-break key = prim ^ kKeyFlip: spheres sort before triangles

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk(__fmaf_rn(a.y, b.z, -(a.z * b.y)), __fmaf_rn(a.z, b.x, -(a.x * b.z)), __fmaf_rn(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ f3 norm3(f3 a) { float l = sqrtf(dot3(a, a)); return mk(a.x / l, a.y / l, a.z / l); }
__device__ __forceinline__ f3 madd3(float t, f3 d, f3 o) { return mk(__fmaf_rn(t, d.x, o.x), __fmaf_rn(t, d.y, o.y), __fmaf_rn(t, d.z, o.z)); }

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// Seeded integer hash -> [0,1): substitute for frag.glsl:13-15 (sin-based, parity unpinned).
__device__ __forceinline__ float rand01(uint32_t index, uint32_t dim, uint32_t seed) {
    uint32_t k = hash32(seed * 0x9E3779B9u + dim + 1u);
    uint32_t h = hash32(hash32(index ^ k) + dim);
    return (float)(h >> 8) * 0x1.0p-24f;
}

struct Best {
    float t, u, v;
    uint32_t key;     // prim ^ kKeyFlip of the current best; 0 while nothing is hit (so ties at tmax lose)
};

struct Counters { uint32_t box, nodes, tris, spheres; };

// frag.glsl:130-156 with e1/e2 precomputed at upload (same fp32 subtraction, :132-133).  All four
// rejections are evaluated in the reference's comparison forms, so NaNs take the same side.
__device__ __forceinline__ bool tri_hit(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& t, float& u, float& v) {
    f3 h = cross3(d, e2);
    float a = dot3(e1, h);
    float f = 1.0f / a;
    f3 s = sub3(o, v0);
    u = f * dot3(s, h);
    f3 q = cross3(s, e1);
    v = f * dot3(d, q);
    t = f * dot3(e2, q);
    bool rej = (fabsf(a) < kEps) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f);
    return !rej & (t > kEps);
}

// traceRay's replacement rule (frag.glsl:272/285, strict '<' in index order) in order-independent form.
__device__ __forceinline__ void consider(Best& b, float tmin, float t, float u, float v, uint32_t key) {
    if (t >= tmin && (t < b.t || (t == b.t && key < b.key))) { b.t = t; b.u = u; b.v = v; b.key = key; }
}

// frag.glsl:158-183
__device__ __forceinline__ bool sphere_hit(f3 o, f3 d, float4 sph, float& t) {
    f3 oc = sub3(o, mk(sph.x, sph.y, sph.z));
    float a = dot3(d, d);
    float b = 2.0f * dot3(oc, d);
    float c = __fmaf_rn(-sph.w, sph.w, dot3(oc, oc));
    float disc = __fmaf_rn(b, b, -((4.0f * a) * c));
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float t0 = (-b - sq) / (2.0f * a);
    float t1 = (-b + sq) / (2.0f * a);
    if (t0 > 0.0f) { t = t0; return true; }
    if (t1 > 0.0f) { t = t1; return true; }
    return false;
}

template <bool COUNT>
__device__ __forceinline__ void spheres_pass(const DevScene& s, f3 o, f3 d, float tmin, Best& b, Counters& c) {
    for (uint32_t i = 0; i < s.n_spheres; ++i) {         // frag.glsl:269-280 (uniform index: scalar loads)
        float t;
        if (sphere_hit(o, d, s.spheres[i], t)) consider(b, tmin, t, 0.0f, 0.0f, i);
        if (COUNT) ++c.spheres;
    }
}

// ---- brute force: the reference's own loop (frag.glsl:282-298) over the whole triangle array ----
template <bool ANY, bool COUNT>
__device__ __forceinline__ void trace_brute(const DevScene& s, f3 o, f3 d, float tmin, Best& b, Counters& c) {
    const float4* __restrict__ tp = s.tris;
    for (uint32_t i = 0; i < s.n_tris; ++i) {             // uniform index: the three records come in as scalar loads
        float4 q0 = tp[3 * i], q1 = tp[3 * i + 1], q2 = tp[3 * i + 2];
        float t, u, v;
        if (tri_hit(o, d, mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), t, u, v))
            consider(b, tmin, t, u, v, __float_as_uint(q0.w) | kKeyFlip);
        if (COUNT) ++c.tris;
    }
}

// ---- BVH2 traversal, one ray per lane, stack in LDS ------------------------------------------------
// Box planes are padded at build time (DESIGN.md "Conservative traversal"), so the slab test may use
// the single-fma form t = plane*idir + (-o*idir).  A node is entered when tn <= tf with tf clipped to
// the current best t: equality is NOT culled, which the tie rule needs.
__device__ __forceinline__ float safe_rcp(float d) {
    const float tiny = 0x1.0p-80f;
    float dd = fabsf(d) > tiny ? d : copysignf(tiny, d);
    return 1.0f / dd;
}

// One interior step for this lane: test both children, descend into the nearer hit child, push the other,
// or pop when both miss.  Returns the next node reference.
template <bool COUNT>
__device__ __forceinline__ int bvh2_step(const float4* __restrict__ nodes, int node, f3 id, f3 od, float tmin, float bt,
                                         int* __restrict__ stk, int& sp, Counters& c) {
    const float4* np = nodes + (size_t)node * 4;
    float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    if (COUNT) { ++c.nodes; c.box += 2; }
    float a0 = __fmaf_rn(q0.x, id.x, od.x), a1 = __fmaf_rn(q0.y, id.x, od.x);
    float a2 = __fmaf_rn(q0.z, id.y, od.y), a3 = __fmaf_rn(q0.w, id.y, od.y);
    float a4 = __fmaf_rn(q2.x, id.z, od.z), a5 = __fmaf_rn(q2.y, id.z, od.z);
    float b0 = __fmaf_rn(q1.x, id.x, od.x), b1 = __fmaf_rn(q1.y, id.x, od.x);
    float b2 = __fmaf_rn(q1.z, id.y, od.y), b3 = __fmaf_rn(q1.w, id.y, od.y);
    float b4 = __fmaf_rn(q2.z, id.z, od.z), b5 = __fmaf_rn(q2.w, id.z, od.z);
    float tn0 = fmaxf(fmaxf(fminf(a0, a1), fminf(a2, a3)), fmaxf(fminf(a4, a5), tmin));
    float tf0 = fminf(fminf(fmaxf(a0, a1), fmaxf(a2, a3)), fminf(fmaxf(a4, a5), bt));
    float tn1 = fmaxf(fmaxf(fminf(b0, b1), fminf(b2, b3)), fmaxf(fminf(b4, b5), tmin));
    float tf1 = fminf(fminf(fmaxf(b0, b1), fmaxf(b2, b3)), fminf(fmaxf(b4, b5), bt));
    bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
    if (h0 & h1) {
        bool sw = tn1 < tn0;
        stk[sp * kBlock] = sw ? c0 : c1;
        ++sp;
        return sw ? c1 : c0;
    }
    if (h0 | h1) return h0 ? c0 : c1;
    --sp;
    return stk[sp * kBlock];
}

template <bool COUNT>
__device__ __forceinline__ void bvh2_leaf(const float4* __restrict__ tris, int node, f3 o, f3 d, float tmin, Best& b,
                                          Counters& c) {
    uint32_t ref = ~(uint32_t)node;
    uint32_t first = ref >> 3, cnt = (ref & 7u) + 1u;
    for (uint32_t k = 0; k < cnt; ++k) {
        const float4* tp = tris + (size_t)(first + k) * 3;
        float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
        float t, u, v;
        if (tri_hit(o, d, mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), t, u, v))
            consider(b, tmin, t, u, v, __float_as_uint(q0.w) | kKeyFlip);
        if (COUNT) ++c.tris;
    }
}

// while-while traversal: every lane walks interior nodes (popping when both children miss) until it holds a
// leaf or is done; then the wave intersects leaves together.  kDone sits at the stack bottom.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void trace_bvh2(const DevScene& s, f3 o, f3 d, float tmin, float tmax, Best& b, Counters& c,
                                           int* __restrict__ stk /* this lane's column, stride kBlock */) {
    if (s.n_nodes == 0) return;
    const float4* __restrict__ nodes = s.nodes;
    const float4* __restrict__ tris = s.tris;
    const f3 id = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const f3 od = mk(-o.x * id.x, -o.y * id.y, -o.z * id.z);
    stk[0] = kDone;
    int sp = 1;
    int node = 0;
    while (node != kDone) {
        while (node >= 0) node = bvh2_step<COUNT>(nodes, node, id, od, tmin, b.t, stk, sp, c);
        if (node != kDone) {
            bvh2_leaf<COUNT>(tris, node, o, d, tmin, b, c);
            if (ANY && b.t < tmax) return;
            --sp;
            node = stk[sp * kBlock];
        }
    }
}

template <int MODE, bool ANY, bool COUNT>
__device__ __forceinline__ void trace_ray(const DevScene& s, f3 o, f3 d, float tmin, float tmax, Best& b, Counters& c,
                                          int* stk) {
    b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;       // frag.glsl:266 (closestT = tmax)
    if (!(tmax > tmin)) return;                            // empty interval: nothing can be accepted
    spheres_pass<COUNT>(s, o, d, tmin, b, c);
    if (ANY && b.t < tmax) return;
    if (MODE == kBrute) trace_brute<ANY, COUNT>(s, o, d, tmin, b, c);
    else                trace_bvh2<ANY, COUNT>(s, o, d, tmin, tmax, b, c, stk);
}

// b.key == 0 can also be sphere 0 (key 0): a hit is recognised by t < tmax instead.
__device__ __forceinline__ float4 pack_hit(const Best& b, float tmax) {
    bool hit = b.t < tmax;
    uint32_t prim = hit ? (b.key ^ kKeyFlip) : RT_MISS;
    return make_float4(b.t, hit ? b.u : 0.0f, hit ? b.v : 0.0f, __uint_as_float(prim));
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const Counters& c, bool active, unsigned long long* g) {
    if (!COUNT) return;
    uint32_t v[5] = { active ? 1u : 0u, c.box, c.nodes, c.tris, c.spheres };
    #pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint32_t x = v[k];
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g[k], (unsigned long long)x);
    }
    // divergence probes: per-wave maxima of nodes / triangle tests (what a lock-step wave has to sit through)
    uint32_t m[2] = { c.nodes, c.tris };
    #pragma unroll
    for (int k = 0; k < 2; ++k) {
        uint32_t x = m[k];
        #pragma unroll
        for (int off = 32; off > 0; off >>= 1) x = max(x, (uint32_t)__shfl_down(x, off, 64));
        if ((threadIdx.x & 63) == 0) atomicAdd(&g[5 + k], (unsigned long long)x);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&g[7], 1ull);
}

// sphere 0 has key 0, the same as "nothing yet": any-hit tests must look at t, not key
__device__ __forceinline__ bool found(const Best& b, float tmax) { return b.t < tmax; }

// ---- kernels ---------------------------------------------------------------------------------------
template <int MODE, bool ANY, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace(DevScene s, const rt_ray* __restrict__ rays, uint64_t n,
                                                  float4* __restrict__ hits, uint8_t* __restrict__ occ,
                                                  unsigned long long* counters) {
    extern __shared__ int stack[];      // (depth + 2) entries x kBlock threads, sized at launch
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = i < n;
    Counters c = { 0, 0, 0, 0 };
    if (active) {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        float4 r0 = rp[0], r1 = rp[1];
        Best b;
        trace_ray<MODE, ANY, COUNT>(s, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), r0.w, r1.w, b, c,
                                    stack + (MODE == kBvh2 ? threadIdx.x : 0));
        if (ANY) occ[i] = found(b, r1.w) ? 1 : 0;
        else     hits[i] = pack_hit(b, r1.w);
    }
    flush_counters<COUNT>(c, active, counters);
}

// Ray-buffer closest-hit / any-hit with in-wave refill ("strip" kernel).  Incoherent rays finish at very
// different times (C3 bounce rays: 30 nodes on average, 65 for the slowest lane of a wave), so a lock-step
// wave idles most lanes.  Here a wave owns a contiguous strip of `strip` rays; whenever at least `refill`
// lanes are idle they take the next rays of the strip (rank by mbcnt over the idle ballot: coalesced reads,
// no atomics), while the other lanes keep their traversal state.  Each lane writes its hit when it finishes.
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_trace_strip(DevScene s, const rt_ray* __restrict__ rays, uint64_t n,
                                                        uint32_t strip, int refill, int tth,
                                                        float4* __restrict__ hits, uint8_t* __restrict__ occ,
                                                        unsigned long long* counters) {
    extern __shared__ int stack[];
    int* stk = stack + threadIdx.x;
    const float4* __restrict__ nodes = s.nodes;
    const float4* __restrict__ tris = s.tris;
    const uint64_t wave = (uint64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    uint64_t cursor = wave * strip;                                   // wave-uniform
    const uint64_t end = (cursor + strip < n) ? cursor + strip : n;
    stk[0] = kDone;
    int node = kDone, sp = 1;
    uint64_t my = 0;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 0), id = mk(0, 0, 0), od = mk(0, 0, 0);
    float tmin = 0.0f, tmax = 0.0f;
    Best b = { 0.0f, 0.0f, 0.0f, 0u };
    Counters c = { 0, 0, 0, 0 };
    uint32_t n_mine = 0;
    for (;;) {
        const bool idle = node == kDone;
        const unsigned long long im = __ballot(idle);
        const int n_idle = __popcll(im);
        if (cursor < end && n_idle >= refill) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            const uint64_t cand = cursor + rank;
            if (idle && cand < end) {
                my = cand;
                const float4* rp = reinterpret_cast<const float4*>(rays + my);
                float4 r0 = rp[0], r1 = rp[1];
                o = mk(r0.x, r0.y, r0.z); d = mk(r1.x, r1.y, r1.z); tmin = r0.w; tmax = r1.w;
                b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
                if (COUNT) ++n_mine;
                bool go = tmax > tmin;
                if (go) {
                    spheres_pass<COUNT>(s, o, d, tmin, b, c);
                    go = !(ANY && b.t < tmax) && s.n_nodes != 0;
                }
                if (go) {
                    id = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
                    od = mk(-o.x * id.x, -o.y * id.y, -o.z * id.z);
                    node = 0; sp = 1;
                } else {
                    if (ANY) occ[my] = b.t < tmax ? 1 : 0;
                    else     hits[my] = pack_hit(b, tmax);
                }
            }
            cursor += (uint64_t)n_idle;
        }
        const bool live = node != kDone;
        if (__ballot(live) == 0ull) {
            if (cursor >= end) break;
            continue;
        }
        // if-if scheduling: per round every lane does at most ONE interior step and ONE triangle test, so no
        // lane waits for the slowest lane's whole descent.  The triangle block is skipped (wave-uniform
        // branch) until at least `tth` lanes hold a leaf or no lane has interior work left.
        const bool need_i = node >= 0;
        const unsigned long long mi = __ballot(need_i);
        if (need_i) node = bvh2_step<COUNT>(nodes, node, id, od, tmin, b.t, stk, sp, c);
        const bool need_t = node < 0 && node != kDone;
        const unsigned long long mt = __ballot(need_t);
        if (__popcll(mt) >= tth || mi == 0ull) {
            if (need_t) {
                const uint32_t ref = ~(uint32_t)node;              // (first << 3) | (remaining - 1)
                const float4* tp = tris + (size_t)(ref >> 3) * 3;
                float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                float t, u, v;
                if (tri_hit(o, d, mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), t, u, v))
                    consider(b, tmin, t, u, v, __float_as_uint(q0.w) | kKeyFlip);
                if (COUNT) ++c.tris;
                if (ANY && b.t < tmax) { node = kDone; sp = 1; }
                else if (ref & 7u) node = (int)~(ref + 7u);          // next triangle of the leaf: first+1, remaining-1
                else { --sp; node = stk[sp * kBlock]; }
            }
        }
        if (live && node == kDone) {
            if (ANY) occ[my] = b.t < tmax ? 1 : 0;
            else     hits[my] = pack_hit(b, tmax);
        }
    }
    if (COUNT) {
        uint32_t v[5] = { n_mine, c.box, c.nodes, c.tris, c.spheres };
        #pragma unroll
        for (int k = 0; k < 5; ++k) {
            uint32_t x = v[k];
            #pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters[k], (unsigned long long)x);
        }
    }
}

// Primary rays (frag.glsl:89-128) fused with traceRay.  One wave = one 8x8 pixel tile for one sample.
template <int MODE, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_primary(DevScene s, DevCamera cam, uint32_t w, uint32_t h, uint32_t y0,
                                                    uint32_t rows, uint32_t stripe_h, uint32_t stripe_step,
                                                    uint32_t spp, uint32_t seed, int jitter, uint32_t tiles_x, uint32_t n_waves, float4* __restrict__ rays_out,
                                                    float4* __restrict__ hits, unsigned long long* counters) {
    extern __shared__ int stack[];      // (depth + 2) entries x kBlock threads, sized at launch
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t tile = wave / spp, smp = wave - tile * spp;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t x = tx * 8 + (lane & 7), yl = ty * 8 + (lane >> 3);
    // local row yl -> image row: stripes of stripe_h rows, every stripe_step-th stripe belongs to this launch
    const uint32_t y = y0 + (yl / stripe_h) * stripe_h * stripe_step + (yl % stripe_h);
    const bool active = wave < n_waves && x < w && yl < rows && y < h;
    Counters c = { 0, 0, 0, 0 };
    if (active) {
        const uint32_t key = (y * w + x) * spp + smp;                 // RNG key: full-image ray index
        float uvx = ((float)x + 0.5f) / (float)w;                     // vert.glsl:9-12 pixel centre
        float uvy = ((float)y + 0.5f) / (float)h;
        float ndcx = __fmaf_rn(uvx, 2.0f, -1.0f), ndcy = __fmaf_rn(uvy, 2.0f, -1.0f);   // :91
        if (jitter) {                                                 // :94-103
            ndcx += (rand01(key, 0u, seed) - 0.5f) * (2.0f / 1920.0f);
            ndcy += (rand01(key, 1u, seed) - 0.5f) * (2.0f / 1080.0f);
        }
        float a = ndcx * cam.half_w, bb = ndcy * cam.half_h;          // :114-115
        f3 d = mk(__fmaf_rn(bb, cam.ux, a * cam.rx) + cam.fx, __fmaf_rn(bb, cam.uy, a * cam.ry) + cam.fy,
                  __fmaf_rn(bb, cam.uz, a * cam.rz) + cam.fz);
        d = norm3(d);                                                 // :113
        d = norm3(d);                                                 // :120
        f3 o = mk(cam.px, cam.py, cam.pz);
        const float tmin = 0.0f, tmax = 1e20f;                        // :266
        Best b;
        trace_ray<MODE, false, COUNT>(s, o, d, tmin, tmax, b, c, stack + (MODE == kBvh2 ? threadIdx.x : 0));
        const size_t r = ((size_t)yl * w + x) * spp + smp;
        if (rays_out) {
            rays_out[2 * r] = make_float4(o.x, o.y, o.z, tmin);
            rays_out[2 * r + 1] = make_float4(d.x, d.y, d.z, tmax);
        }
        hits[r] = pack_hit(b, tmax);
    } else if (wave < n_waves && x < w && yl < rows) {
        // padding row of an interleaved share (image row >= h): a miss record and a dead ray, so that
        // later stages that sweep the whole local buffer see well-defined input
        const size_t r = ((size_t)yl * w + x) * spp + smp;
        if (rays_out) { rays_out[2 * r] = make_float4(0, 0, 0, 0); rays_out[2 * r + 1] = make_float4(0, 0, 0, 0); }
        hits[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(RT_MISS));
    }
    flush_counters<COUNT>(c, active, counters);
}

// Shading point of a hit: position frag.glsl:275/288, normal :276/:289-293, normalised again as :315.
__device__ __forceinline__ bool hit_point(const DevScene& s, f3 o, f3 d, float4 hit, f3& p, f3& n) {
    uint32_t prim = __float_as_uint(hit.w);
    if (prim == RT_MISS) return false;
    p = madd3(hit.x, d, o);
    if (prim & RT_SPHERE_BIT) {
        float4 sp = s.spheres[prim & ~RT_SPHERE_BIT];
        n = norm3(sub3(p, mk(sp.x, sp.y, sp.z)));
    } else {
        const float4* np = s.normals + (size_t)prim * 3;
        float4 n0 = np[0], n1 = np[1], n2 = np[2];
        float w0 = (1.0f - hit.y) - hit.z;
        n = mk(__fmaf_rn(hit.z, n2.x, __fmaf_rn(hit.y, n1.x, w0 * n0.x)),
               __fmaf_rn(hit.z, n2.y, __fmaf_rn(hit.y, n1.y, w0 * n0.y)),
               __fmaf_rn(hit.z, n2.z, __fmaf_rn(hit.y, n1.z, w0 * n0.z)));
        n = norm3(n);
    }
    n = norm3(n);
    return true;
}

// frag.glsl:321-330 for one light, fused: shadow-ray build + closest-hit + lit test.
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_shadow(DevScene s, const rt_ray* __restrict__ rays,
                                                   const float4* __restrict__ hits, uint64_t n, float lx, float ly,
                                                   float lz, float4* __restrict__ srays, uint8_t* __restrict__ lit) {
    extern __shared__ int stack[];      // (depth + 2) entries x kBlock threads, sized at launch
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4* rp = reinterpret_cast<const float4*>(rays + i);
    float4 r0 = rp[0], r1 = rp[1];
    f3 p, nn;
    if (!hit_point(s, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), hits[i], p, nn)) {
        lit[i] = 2;
        if (srays) { srays[2 * i] = make_float4(0, 0, 0, 0); srays[2 * i + 1] = make_float4(0, 0, 0, 0); }
        return;
    }
    f3 toL = sub3(mk(lx, ly, lz), p);
    f3 L = norm3(toL);                                       // :321
    float distance = sqrtf(dot3(toL, toL));                  // :322
    f3 so = madd3(0.001f, nn, p);                            // :326
    Best b; Counters c = { 0, 0, 0, 0 };
    const float tmax = 1e20f;
    trace_ray<MODE, false, false>(s, so, L, 0.0f, tmax, b, c, stack + (MODE == kBvh2 ? threadIdx.x : 0));
    bool is_lit = true;
    if (b.t < tmax) {
        f3 dl = sub3(madd3(b.t, L, so), p);
        is_lit = sqrtf(dot3(dl, dl)) > distance;             // :330
    }
    lit[i] = is_lit ? 1 : 0;
    if (srays) { srays[2 * i] = make_float4(so.x, so.y, so.z, 0.0f); srays[2 * i + 1] = make_float4(L.x, L.y, L.z, tmax); }
}

// sin/cos of 2*pi*x, x in [0,1): exact quarter-turn reduction + fixed fp32 polynomials (same
// coefficients as the oracle; GLSL leaves sin/cos precision implementation-defined).
__device__ __forceinline__ void sincos_2pi(float x, float& s, float& c) {
    float k = floorf(__fmaf_rn(x, 4.0f, 0.5f));
    float r = __fmaf_rn(k, -0.25f, x);
    float a = r * 6.283185307179586f;
    float a2 = a * a;
    float sp = __fmaf_rn(a2, __fmaf_rn(a2, __fmaf_rn(a2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.0f) * a;
    float cp = __fmaf_rn(a2, __fmaf_rn(a2, __fmaf_rn(a2, __fmaf_rn(a2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                                     4.166664568298827e-2f), -0.5f), 1.0f);
    int q = (int)k & 3;
    float ss = (q & 1) ? cp : sp, cs = (q & 1) ? sp : cp;
    if (q == 1) { cs = -cs; }
    else if (q == 2) { ss = -ss; cs = -cs; }
    else if (q == 3) { ss = -ss; }
    s = ss; c = cs;
}

// frag.glsl:185-203 (cos(theta) = sqrt(1-xi1), sin(theta) = sqrt(xi1) folded analytically) + :340-343.
__global__ __launch_bounds__(kBlock) void k_bounce(DevScene s, const rt_ray* __restrict__ rays,
                                                   const float4* __restrict__ hits, uint64_t n, uint32_t key0,
                                                   uint32_t seed, float4* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4* rp = reinterpret_cast<const float4*>(rays + i);
    float4 r0 = rp[0], r1 = rp[1];
    f3 p, N;
    if (!hit_point(s, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), hits[i], p, N)) {
        out[2 * i] = make_float4(0, 0, 0, 0); out[2 * i + 1] = make_float4(0, 0, 0, 0);   // tmax 0: traces to a miss
        return;
    }
    const uint32_t key = key0 + (uint32_t)i;
    float xi1 = rand01(key, 2u, seed), xi2 = rand01(key, 3u, seed);
    float ct = sqrtf(1.0f - xi1), st = sqrtf(xi1);
    float sphi, cphi;
    sincos_2pi(xi2, sphi, cphi);
    float xs = st * cphi, ys = ct, zs = st * sphi;                                  // :193-195
    f3 up = fabsf(N.z) < 0.999f ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);      // :197
    f3 tX = norm3(cross3(up, N));                                                   // :198
    f3 tY = cross3(N, tX);                                                          // :199
    f3 dir = mk(__fmaf_rn(N.x, ys, __fmaf_rn(tY.x, zs, tX.x * xs)), __fmaf_rn(N.y, ys, __fmaf_rn(tY.y, zs, tX.y * xs)),
                __fmaf_rn(N.z, ys, __fmaf_rn(tY.z, zs, tX.z * xs)));                // :201
    dir = norm3(dir);                                                               // :202
    f3 o = madd3(0.001f, N, p);                                                     // :342
    out[2 * i] = make_float4(o.x, o.y, o.z, 0.0f);
    out[2 * i + 1] = make_float4(dir.x, dir.y, dir.z, 1e20f);
}

// LDS for the per-lane stacks: sentinel + one entry per BVH level (+1 spare), [entry][thread]
inline size_t stack_bytes(const DevScene& s, int mode) {
    return mode == kBvh2 ? (size_t)(s.stack_depth + 2) * kBlock * sizeof(int) : 0;
}
inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

}  // namespace

hipError_t launch_trace(const DevScene& s, int mode, bool any_hit, bool count, const rt_ray* d_rays, uint64_t n,
                        rt_hit* d_hits, uint8_t* d_occ, unsigned long long* d_counters, hipStream_t st,
                        uint32_t strip, int refill) {
    if (n == 0) return hipSuccess;
    dim3 b(kBlock);
    float4* hits = reinterpret_cast<float4*>(d_hits);
    if (mode == kBvh2 && strip >= 64) {
        // strip kernel: one wave per `strip` rays
        const uint64_t n_waves = (n + strip - 1) / strip;
        dim3 g((uint32_t)((n_waves + kBlock / 64 - 1) / (kBlock / 64)));
#define RT_LAUNCH(A, C) hipLaunchKernelGGL((k_trace_strip<A, C>), g, b, stack_bytes(s, kBvh2), st, s, d_rays, n, strip, refill & 0xff, refill >> 8, hits, d_occ, d_counters)
        if (any_hit) { if (count) RT_LAUNCH(true, true); else RT_LAUNCH(true, false); }
        else         { if (count) RT_LAUNCH(false, true); else RT_LAUNCH(false, false); }
#undef RT_LAUNCH
        return hipGetLastError();
    }
    dim3 g(grid_for(n));
#define RT_LAUNCH(M, A, C) hipLaunchKernelGGL((k_trace<M, A, C>), g, b, stack_bytes(s, M), st, s, d_rays, n, hits, d_occ, d_counters)
    if (mode == kBrute) {
        if (any_hit) { if (count) RT_LAUNCH(kBrute, true, true); else RT_LAUNCH(kBrute, true, false); }
        else         { if (count) RT_LAUNCH(kBrute, false, true); else RT_LAUNCH(kBrute, false, false); }
    } else {
        if (any_hit) { if (count) RT_LAUNCH(kBvh2, true, true); else RT_LAUNCH(kBvh2, true, false); }
        else         { if (count) RT_LAUNCH(kBvh2, false, true); else RT_LAUNCH(kBvh2, false, false); }
    }
#undef RT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_primary(const DevScene& s, const DevCamera& cam, int mode, bool count, uint32_t w, uint32_t h,
                          uint32_t y0, uint32_t rows, uint32_t stripe_h, uint32_t stripe_step, uint32_t spp,
                          uint32_t seed, int jitter, rt_ray* d_rays_out, rt_hit* d_hits,
                          unsigned long long* d_counters, hipStream_t st) {
    if (rows == 0 || w == 0 || spp == 0) return hipSuccess;
    if (stripe_h == 0 || stripe_step == 0) return hipErrorInvalidValue;
    const uint32_t tiles_x = (w + 7) / 8, tiles_y = (rows + 7) / 8;
    const uint64_t n_waves64 = (uint64_t)tiles_x * tiles_y * spp;
    if (n_waves64 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t n_waves = (uint32_t)n_waves64;
    dim3 g((n_waves + kBlock / 64 - 1) / (kBlock / 64)), b(kBlock);
    float4* rays = reinterpret_cast<float4*>(d_rays_out);
    float4* hits = reinterpret_cast<float4*>(d_hits);
#define RT_LAUNCH(M, C) hipLaunchKernelGGL((k_primary<M, C>), g, b, stack_bytes(s, M), st, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, \
                                           jitter, tiles_x, n_waves, rays, hits, d_counters)
    if (mode == kBrute) { if (count) RT_LAUNCH(kBrute, true); else RT_LAUNCH(kBrute, false); }
    else                { if (count) RT_LAUNCH(kBvh2, true); else RT_LAUNCH(kBvh2, false); }
#undef RT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_shadow(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, float lx,
                         float ly, float lz, rt_ray* d_shadow_rays, uint8_t* d_lit, hipStream_t st) {
    if (n == 0) return hipSuccess;
    dim3 g(grid_for(n)), b(kBlock);
    const float4* hits = reinterpret_cast<const float4*>(d_hits);
    float4* sr = reinterpret_cast<float4*>(d_shadow_rays);
    if (mode == kBrute) hipLaunchKernelGGL((k_shadow<kBrute>), g, b, stack_bytes(s, kBrute), st, s, d_rays, hits, n, lx, ly, lz, sr, d_lit);
    else                hipLaunchKernelGGL((k_shadow<kBvh2>), g, b, stack_bytes(s, kBvh2), st, s, d_rays, hits, n, lx, ly, lz, sr, d_lit);
    return hipGetLastError();
}

hipError_t launch_bounce(const DevScene& s, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, uint32_t key0,
                         uint32_t seed, rt_ray* d_out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    dim3 g(grid_for(n)), b(kBlock);
    hipLaunchKernelGGL(k_bounce, g, b, 0, st, s, d_rays, reinterpret_cast<const float4*>(d_hits), n, key0, seed,
                       reinterpret_cast<float4*>(d_out));
    return hipGetLastError();
}

}  // namespace rt
