// kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the ray hot path.
//
// What is computed follows shaders/frag.glsl of the reference (paths relative to /root/reference):
//   getCameraRay :89-128, rayIntersectsTriangle :130-156, rayIntersectsSphere :158-183,
//   traceRay :265-301, shadow query :321-330, sampleHemisphere :185-203 + offset :340-343.
// How it is computed is build-original.  Incoherent rays (ray buffers: bounce rays, random rays): one ray per lane of a
// wave64, a per-lane traversal stack in LDS ([entry][thread], every lane owns one bank column) with a spill cap, 64-B
// nodes with four 8-bit-quantised children (or two fp32 ones), persistent waves that refill idle lanes (k_trace_strip).
// Coherent rays (the primary rays of a pixel tile, the shadow rays of a tile to one light): the WAVE walks the tree as
// one packet -- nodes through scalar loads from an fp32 copy, the stack in one VGPR, decisions in scalar code
// (trace_packet, k_primary_packet).  Triangles are 48-B (v0, e1, e2) records in leaf order.
//
// Canonical fp32 evaluation order (must be compiled with -ffp-contract=off; the CPU oracle fixes the
// same order independently, DESIGN.md "Arithmetic contract"):
//   dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x));  cross(a,b).x = fma(a.y,b.z, -(a.z*b.y)) ...
//   normalize(v) = v / sqrt(dot(v,v));  o + t*d = fma(t,d,o)
// Division and sqrt are IEEE correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the default).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include "kernels.hpp"

namespace rt {
namespace {

constexpr int kBlock = kThreads;
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
// 1/d for the slab test only (never for hit records): v_rcp_f32 is within 1 ulp, which the box padding
// covers (DESIGN.md "Conservative traversal"), and costs one instruction instead of the ~10 of an IEEE divide.
__device__ __forceinline__ float safe_rcp(float d) {
    const float tiny = 0x1.0p-80f;
    float dd = fabsf(d) > tiny ? d : copysignf(tiny, d);
    return __builtin_amdgcn_rcpf(dd);
}

typedef float v2f __attribute__((ext_vector_type(2)));
// Streaming kernels want all 8 waves per SIMD: the register allocator is told so (the max-ilp scheduling strategy this file is
// compiled with otherwise spends registers on overlap they cannot use: k_shade_resolve 55 -> 70 VGPRs, k_path_brute 64 -> 83)
#define RT_WAVES8 __attribute__((amdgpu_waves_per_eu(8, 8)))
#define RT_WAVES8_IF(cond) __attribute__((amdgpu_waves_per_eu((cond) ? 8 : 1, 8)))
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr uint64_t kStreamMinRays = 1ull << 22;     // buffers of this many records and more are taken to outlive the caches
// Ray records in, hit records out: streamed once.  With `nt` (wave-uniform; launches on structures that do not fit the L2s, i.e.
// the ones whose rays are binned) they go past the caches as nontemporal accesses, so that they do not push nodes and triangles
// out: 10 M-triangle soup 2.97 -> 3.04 Grays/s at 64 Mi rays, C5 46.4 -> 45.9 ms (profiles/r05j_nontemporal.txt).  Not for
// structures the L2s hold: the next stage reads the hit records, and finds them in the cache (bench frame: bounce -0.7 %,
// last segment + resolve +4.5 %, together nothing).
__device__ __forceinline__ float4 load_stream(const float4* p, bool nt) {
    if (nt) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ void store_stream(float4* p, float4 v, bool nt) {
    if (nt) __builtin_nontemporal_store((v4f){ v.x, v.y, v.z, v.w }, reinterpret_cast<v4f*>(p));
    else    *p = v;
}

// Per-ray slab constants: t = plane * idir + (-o * idir).
//
// Per-ray pad.  The boxes are padded at build time by 2^-pad_log2 * max|vertex coordinate|, which covers rounding that
// grows with the SCENE's magnitudes.  The slab arithmetic above and Moeller-Trumbore's s = o - v0 also work on numbers
// of the ORIGIN's magnitude: the slab t is off by up to ~|o * idir| * 2^-23 (the two large terms cancel), and the
// triangle test accepts rays that pass ~|o| * 2^-22 beside a triangle.  A ray that starts hundreds of scene extents away
// would therefore lose leaves whose triangle the reference loop accepts.  So every ray widens every box by
// m = ray_pad * max|o_a| (ray_pad = the same 2^-pad_log2) on its own: the entry t of an axis moves down and the exit t up
// by m * |idir_a|.  Kept as seven values (one more than without the pad, so the kernel still fits 72 VGPRs = 7 waves per
// SIMD): idir, the ENTRY-side constants e_a = -o_a * idir_a - m * |idir_a|, and m2 = 2 m; the exit-side constant of an
// axis is e_a + m2 * |idir_a|, one fma in the node step.  (BVH2 nodes hold (lo, hi) planes instead of (entry, exit):
// there e_a = -o_a * idir_a - m * idir_a belongs to the lo plane and e_a + m2 * idir_a to the hi plane.)
struct Slab { float ix, iy, iz, ex, ey, ez, m2; };
template <int MODE>
__device__ __forceinline__ Slab make_slab(f3 o, f3 d, float ray_pad) {
    const f3 id = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) * ray_pad;
    const f3 w = MODE == kBvh4q ? mk(m * fabsf(id.x), m * fabsf(id.y), m * fabsf(id.z)) : mk(m * id.x, m * id.y, m * id.z);
    Slab s;
    s.ix = id.x; s.iy = id.y; s.iz = id.z;
    s.ex = -o.x * id.x - w.x; s.ey = -o.y * id.y - w.y; s.ez = -o.z * id.z - w.z;
    s.m2 = m + m;
    return s;
}

// One interior step for this lane, branch-free: test both children, descend into the nearer hit child and
// push the other, or pop when both miss.  The would-be-popped entry is read up front (its latency hides
// under the node fetch) and the far child is always written to the free slot; only `sp` decides what counts.
template <bool COUNT>
__device__ __forceinline__ int bvh2_step(const float4* __restrict__ nodes, int node, const Slab& sl, float tmin, float bt,
                                         int* __restrict__ stk, int& sp, Counters& c) {
    const float4* np = nodes + (size_t)node * 4;
    const int popped = stk[(sp - 1) * kBlock];
    float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    if (COUNT) { ++c.nodes; c.box += 2; }
    // (lo, hi) plane constants of the three axes: lo plane moved out by -m, hi plane by +m (per-ray pad, make_slab)
    const v2f ix2 = (v2f){ sl.ix, sl.ix }, iy2 = (v2f){ sl.iy, sl.iy }, iz2 = (v2f){ sl.iz, sl.iz };
    const v2f ox2 = (v2f){ sl.ex, __fmaf_rn(sl.m2, sl.ix, sl.ex) }, oy2 = (v2f){ sl.ey, __fmaf_rn(sl.m2, sl.iy, sl.ey) },
              oz2 = (v2f){ sl.ez, __fmaf_rn(sl.m2, sl.iz, sl.ez) };
    v2f ax = __builtin_elementwise_fma((v2f){ q0.x, q0.y }, ix2, ox2);
    v2f ay = __builtin_elementwise_fma((v2f){ q0.z, q0.w }, iy2, oy2);
    v2f az = __builtin_elementwise_fma((v2f){ q2.x, q2.y }, iz2, oz2);
    v2f bx = __builtin_elementwise_fma((v2f){ q1.x, q1.y }, ix2, ox2);
    v2f by = __builtin_elementwise_fma((v2f){ q1.z, q1.w }, iy2, oy2);
    v2f bz = __builtin_elementwise_fma((v2f){ q2.z, q2.w }, iz2, oz2);
    float tn0 = fmaxf(fmaxf(fminf(ax.x, ax.y), fminf(ay.x, ay.y)), fmaxf(fminf(az.x, az.y), tmin));
    float tf0 = fminf(fminf(fmaxf(ax.x, ax.y), fmaxf(ay.x, ay.y)), fminf(fmaxf(az.x, az.y), bt));
    float tn1 = fmaxf(fmaxf(fminf(bx.x, bx.y), fminf(by.x, by.y)), fmaxf(fminf(bz.x, bz.y), tmin));
    float tf1 = fminf(fminf(fmaxf(bx.x, bx.y), fmaxf(by.x, by.y)), fminf(fmaxf(bz.x, bz.y), bt));
    const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
    const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
    const bool take1 = h1 & ((tn1 < tn0) | !h0);            // child 1 is the one to enter
    const int near = take1 ? c1 : c0, far = take1 ? c0 : c1;
    stk[sp * kBlock] = far;
    const bool both = h0 & h1, any = h0 | h1;
    sp += both ? 1 : (any ? 0 : -1);
    asm volatile("" :: "v"(q3.z), "v"(q3.w));     // keeps the unused half of q3 away from the next load's destination (see bvh4q_step)
    return any ? near : popped;
}

// ---- BVH4q: four 8-bit-quantised children per 64-B node (layout: bvh.hpp) -----------------------------
// plane = origin + 2^e * q  =>  t = q * (2^e * idir) + (origin * idir - o * idir): one cvt + one (packed) fma
// per plane.  Hit children are sorted by entry distance (5-comparator network on (key, ref) pairs); the nearest
// is entered, the others are pushed so that the next nearest pops first.
__device__ __forceinline__ void cswap(uint32_t& ka, int& ra, uint32_t& kb, int& rb) {
    const bool sw = kb < ka;
    const uint32_t k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const int r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}

// Capped stack of the ray-buffer kernel: the first `cap` entries (RT_OPT_LDS_STACK, default 20) of a lane's stack live in LDS, deeper ones in
// a per-thread column of a global spill area.  The worst-case depth of a BVH4q (3 pushes per level) is far above
// what rays use, and sizing LDS for the worst case holds the kernel at 3-4 waves per SIMD; with the cap it runs at 8.
// base and stride are wave-uniform (SGPRs); the thread's column is recomputed where it is needed (the spill path is rare),
// so that the capped stack costs no VGPRs in the traversal loop.
struct Spill { int* base; uint32_t stride; int cap; };   // [entry][thread of the launch], stride = threads of the launch, LDS entries
__device__ __forceinline__ int* spill_slot(const Spill& sp_, int idx) {
    return sp_.base + ((size_t)(uint32_t)(idx - sp_.cap) * sp_.stride + blockIdx.x * (uint32_t)kBlock + threadIdx.x);
}
__device__ __forceinline__ int st_read(const int* __restrict__ stk, const Spill& sp_, int idx) {
    return idx < sp_.cap ? stk[idx * kBlock] : *spill_slot(sp_, idx);
}
__device__ __forceinline__ void st_write(int* __restrict__ stk, const Spill& sp_, int idx, int v) {
    if (idx < sp_.cap) stk[idx * kBlock] = v; else *spill_slot(sp_, idx) = v;
}

template <bool COUNT, bool CAPPED = false>
__device__ __forceinline__ int bvh4q_step(const float4* __restrict__ nodes, int node, const Slab& sl, float tmin0, float bt,
                                          int* __restrict__ stk, int& sp, Counters& c, const Spill& spill = Spill{ nullptr, 0u, 0 }) {
    // 32-bit byte offset from the (wave-uniform) node array: one shift instead of 64-bit address arithmetic per lane
    const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes) + ((uint32_t)node << 6));
    // this step touches stack slots sp-1 .. sp+2: all in LDS unless the lane is unusually deep
    const bool in_lds = !CAPPED || sp + 3 <= spill.cap;
    // The entry that is popped if all four children miss is read up front, always from LDS (index clamped): its latency
    // hides under the node fetch, and the load stays a plain ds_read.  (Selecting between the LDS slot and the HBM spill
    // column here made the compiler emit a flat load behind an address select.)  A lane so deep that the entry lives in
    // the spill column fetches it in the rare branch at the end.
    const int pidx = sp - 1;
    int popped = stk[(CAPPED ? min(pidx, spill.cap - 1) : pidx) * kBlock];
    const uint4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    if (COUNT) {                                  // children = slots that are not stored inverted (lo.x 255, hi.x 0)
        ++c.nodes;
        #pragma unroll
        for (int k = 0; k < 4; ++k) c.box += (((q2.x >> (8 * k)) & 0xffu) == 255u && ((q2.w >> (8 * k)) & 0xffu) == 0u) ? 0u : 1u;
    }
    const float idx = sl.ix, idy = sl.iy, idz = sl.iz;
    // grid step of the node's axes times 1/d: three multiplies (the steps are plain floats in the node)
    const float sx = __uint_as_float(q0.w) * idx, sy = __uint_as_float(q3.z) * idy, sz = __uint_as_float(q3.w) * idz;
    const v2f sx2 = (v2f){ sx, sx }, sy2 = (v2f){ sy, sy }, sz2 = (v2f){ sz, sz };
    // (entry, exit) constants of the node: grid origin * idir + the ray's entry constant, the exit side 2 m |idir| later
    const float aex = __fmaf_rn(__uint_as_float(q0.x), idx, sl.ex), aey = __fmaf_rn(__uint_as_float(q0.y), idy, sl.ey),
                aez = __fmaf_rn(__uint_as_float(q0.z), idz, sl.ez);
    const v2f ax2 = (v2f){ aex, __fmaf_rn(sl.m2, fabsf(idx), aex) };
    const v2f ay2 = (v2f){ aey, __fmaf_rn(sl.m2, fabsf(idy), aey) };
    const v2f az2 = (v2f){ aez, __fmaf_rn(sl.m2, fabsf(idz), aez) };
    // The ray's direction signs pick, once per node and for all four children, which plane word is the entry
    // side and which the exit side: no per-axis min/max later.  Empty slots are stored inverted (lo 255, hi 0),
    // so their entry lies behind their exit whatever the signs and they miss without a count check.
    const bool nx = idx < 0.0f, ny = idy < 0.0f, nz = idz < 0.0f;
    const uint32_t ex = nx ? q2.w : q2.x, fx = nx ? q2.x : q2.w;      // entry / exit plane bytes, x
    const uint32_t ey = ny ? q3.x : q2.y, fy = ny ? q2.y : q3.x;
    const uint32_t ez = nz ? q3.y : q2.z, fz = nz ? q2.z : q3.y;
#ifdef RT_PROBE_DEADPOP
    int probe_hits_unbounded = 0;
#endif
    uint32_t key[4];
    int ref[4] = { (int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w };
    #pragma unroll
    for (int k = 0; k < 4; ++k) {
        const v2f tx = __builtin_elementwise_fma((v2f){ (float)((ex >> (8 * k)) & 0xffu), (float)((fx >> (8 * k)) & 0xffu) }, sx2, ax2);
        const v2f ty = __builtin_elementwise_fma((v2f){ (float)((ey >> (8 * k)) & 0xffu), (float)((fy >> (8 * k)) & 0xffu) }, sy2, ay2);
        const v2f tz = __builtin_elementwise_fma((v2f){ (float)((ez >> (8 * k)) & 0xffu), (float)((fz >> (8 * k)) & 0xffu) }, sz2, az2);
        // the clamps against the loop-carried tmin0 / best t as opaque instructions: written as fmaxf / fminf the compiler first
        // canonicalises both loop-carried values (two more VALU instructions per node step); v_max_f32 / v_min_f32 return
        // the non-NaN operand exactly like fmaxf / fminf
#ifndef RT_NO_ASM_CLAMP
        float tn = fmaxf(fmaxf(tx.x, ty.x), tz.x), tf = fminf(fminf(tx.y, ty.y), tz.y);
        asm("v_max_f32 %0, %1, %2" : "=v"(tn) : "v"(tn), "v"(tmin0));
        asm("v_min_f32 %0, %1, %2" : "=v"(tf) : "v"(tf), "v"(bt));
#else
        const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin0));
        const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, bt));
#endif
        key[k] = (tn <= tf) ? __float_as_uint(tn) : 0xffffffffu;     // tn >= tmin0 >= 0: bits order like the value
#ifdef RT_PROBE_DEADPOP
        if (COUNT && tn <= fminf(fminf(tx.y, ty.y), tz.y)) ++probe_hits_unbounded;
#endif
    }
    cswap(key[0], ref[0], key[1], ref[1]);
    cswap(key[2], ref[2], key[3], ref[3]);
    cswap(key[0], ref[0], key[2], ref[2]);
    cswap(key[1], ref[1], key[3], ref[3]);
    cswap(key[1], ref[1], key[2], ref[2]);
    // push hits 1..3, farthest first (misses sort to the end; dropping the last comparator and relying on the
    // per-slot test below was measured: 8 % more node visits, slower)
    const uint32_t miss = 0xffffffffu;
    int s = sp;
    if (in_lds) {
        if (key[3] != miss) { stk[s * kBlock] = ref[3]; ++s; }
        if (key[2] != miss) { stk[s * kBlock] = ref[2]; ++s; }
        if (key[1] != miss) { stk[s * kBlock] = ref[1]; ++s; }
    } else {
        if (key[3] != miss) { st_write(stk, spill, s, ref[3]); ++s; }
        if (key[2] != miss) { st_write(stk, spill, s, ref[2]); ++s; }
        if (key[1] != miss) { st_write(stk, spill, s, ref[1]); ++s; }
    }
    const bool any = key[0] != miss;             // key[0] is the minimum: a miss there means four misses
    // (All four quarters of the node are fetched as 16-byte loads and every word of them is used.  When half of the last
    // quarter was still unused, the register allocator could hand that dead half to the NEXT load of the group -- seen:
    // v[14:17] / v[16:19] -- and two loads in flight with overlapping destinations cost trees that do not fit the caches
    // 12 %: C4 25.5 -> 28.5 ms, profiles/r02s_c4_regression.txt.  scripts/check_load_overlap.py watches for it.)
#ifdef RT_PROBE_DEADPOP
    if (COUNT && !any && probe_hits_unbounded) ++c.spheres;   // probe build: steps in which all four children miss ONLY because of the current best t
#endif
    if (CAPPED && !any && pidx >= spill.cap) popped = *spill_slot(spill, pidx);
    sp = any ? s : sp - 1;
    return any ? ref[0] : popped;
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

template <int MODE, bool COUNT, bool CAPPED = false>
__device__ __forceinline__ int bvh_step(const float4* __restrict__ nodes, int node, const Slab& sl, float tmin0, float bt,
                                        int* __restrict__ stk, int& sp, Counters& c, const Spill& spill = Spill{ nullptr, 0u, 0 }) {
    if (MODE == kBvh4q) return bvh4q_step<COUNT, CAPPED>(nodes, node, sl, tmin0, bt, stk, sp, c, spill);
    return bvh2_step<COUNT>(nodes, node, sl, tmin0, bt, stk, sp, c);
}

// while-while traversal: every lane walks interior nodes (popping when both children miss) until it holds a
// leaf or is done; then the wave intersects leaves together.  kDone sits at the stack bottom.
template <int MODE, bool ANY, bool COUNT>
__device__ __forceinline__ void trace_bvh(const DevScene& s, f3 o, f3 d, float tmin, float tmax, Best& b, Counters& c,
                                          int* __restrict__ stk /* this lane's column, stride kBlock */) {
    if (s.n_nodes == 0) return;
    const float4* __restrict__ nodes = s.nodes;
    const float4* __restrict__ tris = s.tris;
    const Slab sl = make_slab<MODE>(o, d, s.ray_pad);
    const float tmin0 = fmaxf(tmin, 0.0f);      // accepted hits have t > 1e-6 (frag.glsl:151) and t > 0 (:174)
    stk[0] = kDone;
    int sp = 1;
    int node = 0;
    while (node != kDone) {
        while (node >= 0) node = bvh_step<MODE, COUNT>(nodes, node, sl, tmin0, b.t, stk, sp, c);
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
    else                trace_bvh<MODE, ANY, COUNT>(s, o, d, tmin, tmax, b, c, stk);
}

// ---- packet traversal (tile kernel, BVH4q trees with the fp32 "wide" node copy: bvh.hpp) ---------------------------
// The 64 rays of a wave of the tile kernel are a 4x4 .. 8x8 pixel tile from one origin: they visit almost the same
// nodes.  Here the WAVE walks the tree: one node at a time for all lanes, the traversal stack and every decision in
// scalar registers, nodes and triangles fetched with scalar loads (no per-lane node fetch, no byte decoding, no
// per-lane sort, no LDS).  Per node and lane that leaves the slab arithmetic: 12 packed fma and the min/max of four
// children.  A child is entered when ANY lane's ray meets it (ballot), children are ordered by the entry distance of the
// first lane that hits them, and every lane tests every triangle of a leaf the wave visits.  Each lane therefore tests
// a superset of what it would test alone, against its own best t; by the order-independent replacement rule
// (consider) the closest hit of every ray is the one the per-lane walk and the reference loop find, bit for bit.
// The stack is one VGPR used as a 64-entry array (lane i = entry i, v_writelane / v_readlane with a scalar index).
// v_writelane_b32 with a scalar value and a scalar lane index (this compiler has the readlane builtin only); the lane
// select goes through m0, because the two scalar operands may not both come from different SGPRs on gfx9-class VALU
// (m0 is reserved by the compiler and cannot be named as a clobber, so it is saved and restored around the instruction)
__device__ __forceinline__ int write_lane(int vec, int value, int lane) {
    int saved;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
        : "+v"(vec), "=&s"(saved) : "s"(value), "s"(lane));
    return vec;
}
__device__ __forceinline__ void cswap_s(uint32_t& ka, int& ra, uint32_t& kb, int& rb) {
    // s_max_u32 leaves "first operand selected" (ka > kb: swap) in SCC for the two reference selects: 4 scalar instructions
    // where compare + four selects are 5
    uint32_t klo, khi; int rlo, rhi;
    asm("s_min_u32 %0, %4, %5\n\ts_max_u32 %1, %4, %5\n\ts_cselect_b32 %2, %7, %6\n\ts_cselect_b32 %3, %6, %7"
        : "=&s"(klo), "=&s"(khi), "=&s"(rlo), "=s"(rhi) : "s"(ka), "s"(kb), "s"(ra), "s"(rb) : "scc");
    ka = klo; kb = khi; ra = rlo; rb = rhi;
}
// v_min / v_max as opaque instructions: like fminf / fmaxf they return the non-NaN operand, but the compiler does not
// put a canonicalising v_max(x, x) in front of every operand it cannot prove quiet (six per child in this loop)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// OCT: bit a set = every ray of the wave points down axis a (its entry plane is the hi plane), known at compile time, so
// entry and exit t of an axis are simply the two halves of one packed fma; OCT = 8: signs differ inside the wave (tiles on
// the camera's axis lines), entry / exit by min / max (six more instructions per child).
template <bool COUNT, int OCT>
__device__ __forceinline__ void packet_walk(const DevScene& s, f3 o, f3 d, float tmin, bool active, const Slab& sl, Best& b, Counters& c) {
    const uint4* __restrict__ wide = reinterpret_cast<const uint4*>(s.wide);
    const float4* __restrict__ tris = s.tris;
    // (lo, hi) plane constants: the box grown by the per-ray pad on both sides (as in bvh2_step)
    const v2f ix2 = (v2f){ sl.ix, sl.ix }, iy2 = (v2f){ sl.iy, sl.iy }, iz2 = (v2f){ sl.iz, sl.iz };
    const v2f cx = (v2f){ sl.ex, __fmaf_rn(sl.m2, sl.ix, sl.ex) }, cy = (v2f){ sl.ey, __fmaf_rn(sl.m2, sl.iy, sl.ey) },
              cz = (v2f){ sl.ez, __fmaf_rn(sl.m2, sl.iz, sl.ez) };
    const float tmin0 = fmaxf(tmin, 0.0f);
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    const uint32_t miss = 0xffffffffu;
    int stack = kDone;                  // lane i holds stack entry i; entry 0 is the sentinel that ends the walk
    int sp = 1, node = 0;               // wave-uniform
    // the lane whose entry distances order the children: one in the middle of the tile (4x4 pixels x 4 samples, 8x8 pixels or 64
    // rays in a row: lane 27), the first lane with a ray if that one has none.  A corner lane (lane 0) orders as well on average
    // but worse for some tiles, and a small launch lasts as long as its slowest wave (1920x1080 primary rays: 0.18 ms with
    // lane 0, 0.10 ms with lane 27, 0.11 ms ordering by the first lane that meets each child: profiles/r03q_packet_salu.txt)
    const int rep = ((act >> 27) & 1ull) ? 27 : (int)__builtin_ctzll(act | (1ull << 63));
    // One loop, one exit, every branch wave-uniform: with `continue` / `break` / an empty-stack test in the body the compiler
    // turned the uniform branches into mask arithmetic (a dozen scalar instructions per node); popping the sentinel ends it
    // (1.70 -> 1.62 ms; an `if` with a `break` for the sentinel in its else branch instead of the inner loop, and selects
    // instead of branches in the triangle loop, were both slower: profiles/r03q_packet_salu.txt).
    for (;;) {
        while (node >= 0) {
            // the whole 128-byte node in one go (uniform address: scalar loads), then straight-line code: no child is
            // skipped by a branch -- empty slots carry planes at +3e38 and miss by arithmetic
            // 32-bit byte offset (the scalar loads take it as their SGPR offset: one shift instead of 64-bit address arithmetic);
            // rt_api.cpp keeps the wide copy only for trees below 2^25 nodes
            const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wide) + ((uint32_t)node << 7));
            const uint4 w0 = np[0], w1 = np[1], p0 = np[2], p1 = np[3], p2 = np[4], p3 = np[5], p4 = np[6], p5 = np[7];
            if (COUNT && active) { ++c.nodes; c.box += w1.x; }
            const uint32_t pl[24] = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w,
                                      p3.x, p3.y, p3.z, p3.w, p4.x, p4.y, p4.z, p4.w, p5.x, p5.y, p5.z, p5.w };
            int ref[4] = { (int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w };
            uint32_t key[4];
            #pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f tx = __builtin_elementwise_fma((v2f){ __uint_as_float(pl[6 * k]), __uint_as_float(pl[6 * k + 1]) }, ix2, cx);
                const v2f ty = __builtin_elementwise_fma((v2f){ __uint_as_float(pl[6 * k + 2]), __uint_as_float(pl[6 * k + 3]) }, iy2, cy);
                const v2f tz = __builtin_elementwise_fma((v2f){ __uint_as_float(pl[6 * k + 4]), __uint_as_float(pl[6 * k + 5]) }, iz2, cz);
                float tn, tf;
                if (OCT < 8) {
                    tn = vmax(vmax3((OCT & 1) ? tx.y : tx.x, (OCT & 2) ? ty.y : ty.x, (OCT & 4) ? tz.y : tz.x), tmin0);
                    tf = vmin(vmin3((OCT & 1) ? tx.x : tx.y, (OCT & 2) ? ty.x : ty.y, (OCT & 4) ? tz.x : tz.y), b.t);
                } else {
                    tn = vmax(vmax3(vmin(tx.x, tx.y), vmin(ty.x, ty.y), vmin(tz.x, tz.y)), tmin0);
                    tf = vmin(vmin3(vmax(tx.x, tx.y), vmax(ty.x, ty.y), vmax(tz.x, tz.y)), b.t);
                }
                // The children are ordered by the entry distances of ONE lane of the packet (`rep`), whether that
                // lane meets the child or not; "some lane meets it" and the key select are two scalar instructions (s_and_b64
                // leaves "result != 0" in SCC).  The scalar unit is what this loop runs out of first -- one more scalar
                // instruction per node costs as much as four VALU cycles (profiles/r03q_packet_salu.txt) -- and finding the first
                // lane that meets each child (s_ff1 + a second readlane index + compare) took 15 of them: 1.82 -> 1.71 ms.
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(tn <= tf);
                const uint32_t tn_rep = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(tn), rep);   // tn >= 0: bits order like the value
                unsigned long long m;
                asm("s_and_b64 %0, %2, %3\n\ts_cselect_b32 %1, %4, -1" : "=&s"(m), "=s"(key[k]) : "s"(hm), "s"(act), "s"(tn_rep) : "scc");
            }
            cswap_s(key[0], ref[0], key[1], ref[1]);
            cswap_s(key[2], ref[2], key[3], ref[3]);
            cswap_s(key[0], ref[0], key[2], ref[2]);
            cswap_s(key[1], ref[1], key[3], ref[3]);
            cswap_s(key[1], ref[1], key[2], ref[2]);
            if (key[3] != miss) { stack = write_lane(stack, ref[3], sp); ++sp; }
            if (key[2] != miss) { stack = write_lane(stack, ref[2], sp); ++sp; }
            if (key[1] != miss) { stack = write_lane(stack, ref[1], sp); ++sp; }
            if (key[0] != miss) node = ref[0];
            else { --sp; node = __builtin_amdgcn_readlane(stack, sp); }
        }
        if (node == kDone) break;
        {
            const uint32_t r = ~(uint32_t)node;                               // (first << 3) | (count - 1)
            const uint32_t first = r >> 3, cnt = (r & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const float4* tp = tris + (size_t)(first + k) * 3;            // uniform address: scalar loads
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                float t, u, v;
                if (active && tri_hit(o, d, mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), t, u, v))
                    consider(b, tmin, t, u, v, __float_as_uint(q0.w) | kKeyFlip);
                if (COUNT && active) ++c.tris;
            }
            --sp;
            node = __builtin_amdgcn_readlane(stack, sp);
        }
    }
}

template <bool COUNT>
__device__ __forceinline__ void trace_packet(const DevScene& s, f3 o, f3 d, float tmin, bool active, Best& b, Counters& c) {
    if (s.n_nodes == 0) return;
    const Slab sl = make_slab<kBvh2>(o, d, s.ray_pad);
    // direction signs of the wave's rays (of the reciprocals the slabs use): uniform on all three axes for nearly every tile
    const unsigned long long act = __builtin_amdgcn_ballot_w64(active);
    const unsigned long long nx = __builtin_amdgcn_ballot_w64(sl.ix < 0.0f) & act, ny = __builtin_amdgcn_ballot_w64(sl.iy < 0.0f) & act,
                             nz = __builtin_amdgcn_ballot_w64(sl.iz < 0.0f) & act;
    const bool uniform = (nx == 0ull || nx == act) && (ny == 0ull || ny == act) && (nz == 0ull || nz == act);
    const int oct = uniform ? ((nx ? 1 : 0) | (ny ? 2 : 0) | (nz ? 4 : 0)) : 8;
    switch (oct) {
        case 0: packet_walk<COUNT, 0>(s, o, d, tmin, active, sl, b, c); break;
        case 1: packet_walk<COUNT, 1>(s, o, d, tmin, active, sl, b, c); break;
        case 2: packet_walk<COUNT, 2>(s, o, d, tmin, active, sl, b, c); break;
        case 3: packet_walk<COUNT, 3>(s, o, d, tmin, active, sl, b, c); break;
        case 4: packet_walk<COUNT, 4>(s, o, d, tmin, active, sl, b, c); break;
        case 5: packet_walk<COUNT, 5>(s, o, d, tmin, active, sl, b, c); break;
        case 6: packet_walk<COUNT, 6>(s, o, d, tmin, active, sl, b, c); break;
        case 7: packet_walk<COUNT, 7>(s, o, d, tmin, active, sl, b, c); break;
        default: packet_walk<COUNT, 8>(s, o, d, tmin, active, sl, b, c); break;
    }
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
    bool live = false;                                      // counters.rays = LIVE rays (tmax > tmin)
    Counters c = { 0, 0, 0, 0 };
    if (active) {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        float4 r0 = rp[0], r1 = rp[1];
        live = r1.w > r0.w;
        Best b;
        trace_ray<MODE, ANY, COUNT>(s, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), r0.w, r1.w, b, c,
                                    stack + (MODE != kBrute ? threadIdx.x : 0));
        if (ANY) occ[i] = found(b, r1.w) ? 1 : 0;
        else     hits[i] = pack_hit(b, r1.w);
    }
    flush_counters<COUNT>(c, live, counters);
}

// Ray buffers whose consecutive rays are coherent (RT_OPT_RAY_PACKETS): each wave takes 64 consecutive rays and walks the
// tree as one packet (trace_packet).  Correct for any buffer -- every ray still tests a superset of its own nodes -- but
// only fast when the 64 rays of a wave really travel together.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_trace_packet(DevScene s, const rt_ray* __restrict__ rays, uint64_t n, const uint32_t* __restrict__ perm,
                    float4* __restrict__ hits, unsigned long long* counters) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in = i < n;
    const uint64_t my = in ? (perm ? (uint64_t)perm[i] : i) : 0;
    float4 r0 = make_float4(0, 0, 0, 0), r1 = make_float4(0, 0, 0, 0);
    if (in) { const float4* rp = reinterpret_cast<const float4*>(rays + my); r0 = rp[0]; r1 = rp[1]; }
    const bool live = in && r1.w > r0.w;
    const f3 o = mk(r0.x, r0.y, r0.z), d = mk(r1.x, r1.y, r1.z);
    Counters c = { 0, 0, 0, 0 };
    Best b;
    b.t = r1.w; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
    if (live) spheres_pass<COUNT>(s, o, d, r0.w, b, c);
    trace_packet<COUNT>(s, o, d, r0.w, live, b, c);
    if (in) hits[my] = pack_hit(b, r1.w);
    flush_counters<COUNT>(c, live, counters);
}

// Ray-buffer closest-hit / any-hit with in-wave refill ("strip" kernel).  Incoherent rays finish at very
// different times (C3 bounce rays: 14 nodes on average, several times that for the slowest lane of a wave), so a
// lock-step wave idles most lanes.  Here the grid is persistent: a wave draws chunks of `strip` consecutive rays
// from a global cursor (one atomic per chunk, by lane 0); whenever at least `refill` lanes are idle they take the
// next rays of the chunk (rank by mbcnt over the idle ballot: coalesced reads), while the other lanes keep their
// traversal state, and a new chunk is drawn as soon as the current one is handed out, so the only tail is the
// one at the end of the launch.  Each lane writes its hit when it finishes; results do not depend on the order.
// Waves per SIMD the register allocator is asked to make room for (0 = its own choice).  Left alone it takes 80 VGPRs
// (6 waves per SIMD); asked for 7 it fits the kernel into 72 without spilling (8 would need 64 and spills in the loop:
// twice as slow).  One more wave per SIMD is worth 3-4 % on incoherent rays (profiles/r02f_ab.txt).
#ifndef RT_STRIP_WAVES
#define RT_STRIP_WAVES 7
#endif
#if RT_STRIP_WAVES
#define RT_STRIP_ATTR __attribute__((amdgpu_waves_per_eu(RT_STRIP_WAVES, 8)))
#else
#define RT_STRIP_ATTR
#endif
#ifndef RT_SECOND_STEP_LANES
#define RT_SECOND_STEP_LANES 36
#endif
constexpr int kSecondStepLanes = RT_SECOND_STEP_LANES;      // lanes on interior nodes from which a round takes a second interior step
// QUEUE: a staged ray queue per wave in LDS (behind the stacks, at int offset q_off).  Whenever it is empty the lower half of the
// wave loads the next kQueue rays and computes their slab constants -- 32 lanes at work, whatever the wave's traversal state --
// and an idle lane then starts its next ray with four ds_read_b128 instead of a global load + make_slab executed for a
// handful of lanes.  That makes a refill cheap enough to run at a threshold of a few idle lanes instead of 16.
// Entry = 4 float4 in planes [plane][slot]: (o, tmin) (d, tmax) (idir, 2m) (entry constants, ray index); n < 2^32.
constexpr int kQueue = 32;
constexpr size_t kQueueBytesPerBlock = (size_t)(kBlock / 64) * 4 * kQueue * sizeof(float4);
template <int MODE, bool ANY, bool COUNT, bool CAPPED, bool QUEUE = false>
__global__ __launch_bounds__(kBlock) RT_STRIP_ATTR void k_trace_strip(DevScene s, const rt_ray* __restrict__ rays, uint64_t n,
                                                        uint32_t strip, int refill, int tth, int refill_perm /* both thresholds (tth << 8 | refill) for a buffer that IS handed out permuted */,
                                                        unsigned long long* __restrict__ work /* 8 chunk cursors + mask, kCursorStride apart, 0 at launch */,
                                                        const uint32_t* __restrict__ perm /* hand-out order or null */,
                                                        const uint32_t* __restrict__ perm_on /* device flag, null = on */,
                                                        int* __restrict__ spill_base /* CAPPED: [entry][thread of the launch] */, int lds_cap,
                                                        int q_off /* QUEUE: ints from the start of LDS to the ray queues */,
                                                        int nt_arg /* ray loads and hit stores nontemporal */,
                                                        float4* __restrict__ hits, uint8_t* __restrict__ occ,
                                                        unsigned long long* counters) {
    extern __shared__ int stack[];
    int* stk = stack + threadIdx.x;
    float4* const qw = reinterpret_cast<float4*>(stack + (QUEUE ? q_off : 0)) + (threadIdx.x >> 6) * (4 * kQueue);   // this wave's queue
    uint32_t q_head = 0, q_avail = 0;                                  // wave-uniform: next slot to hand out, slots left
    uint32_t my32 = 0;                                                 // QUEUE: the ray's index (32 bits)
    const bool nt = nt_arg != 0;
    const float4* __restrict__ nodes = s.nodes;
    const float4* __restrict__ tris = s.tris;
    if (perm && perm_on && *perm_on == 0u) perm = nullptr;             // launch_bin found the buffer coherent already
    if (perm && refill_perm) { refill = refill_perm & 0xff; tth = refill_perm >> 8; }
    // Wave-uniform state of the chunk being handed out, all 32 bits wide so that it lives in SGPRs and the per-round tests are
    // scalar compares (a 64-bit cursor / end pair needs v_cmp_lt_u64: the compiler then keeps the whole chain -- cursor, end,
    // the eighth's number, the mask of exhausted eighths -- in six VGPRs of the 72 this kernel has): chunk number, rays of
    // the chunk handed out so far, rays in the chunk.  launch_trace keeps n below 2^32 chunks.
    uint32_t chunk = 0, cur = 0, lim = 0;
    bool more = true;                                                 // wave-uniform: some cursor is not exhausted
    const uint32_t n_chunks = (uint32_t)((n + strip - 1) / strip);
    const uint32_t per_q = (n_chunks + 7u) / 8u;                      // chunks per eighth
    uint32_t q = blockIdx.x & 7u, q_done = 0;                         // workgroups go round-robin over the 8 XCDs
    const Spill spill = { CAPPED ? spill_base : nullptr, gridDim.x * (uint32_t)kBlock, lds_cap };
    stk[0] = kDone;
    int node = kDone, sp = 1;
    uint64_t my = 0;
    f3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    Slab sl = make_slab<MODE>(o, d, s.ray_pad);
    float tmin = 0.0f, tmin0 = 0.0f, tmax = 0.0f;
    Best b = { 0.0f, 0.0f, 0.0f, 0u };
    Counters c = { 0, 0, 0, 0 };
    uint32_t n_mine = 0;
    uint32_t w_rounds = 0, w_i = 0, w_t = 0;    // wave-uniform scheduling probes (COUNT builds)
    uint32_t w_idle = 0, w_blocked = 0, w_pend = 0, w_tblocks = 0;
    int pend = 0;                               // postponed leaf (leaf refs are negative; 0 = none)
#ifdef RT_PEND2
    int pend2 = 0;                              // -DRT_PEND2, measured and not the default (profiles/r04m_pend2.txt): a second postponed leaf,
                                                // only ever set while pend is.  Bench frame -1.4 %, 10 M-triangle soup +3.3 %.
#endif
    for (;;) {
        const bool idle = node == kDone && pend == 0;
        bool fresh = false;                                          // this lane started a ray in this round
        const unsigned long long im = __ballot(idle);
        const int n_idle = __popcll(im);
        if (cur >= lim && more && (QUEUE ? q_avail == 0u : n_idle >= refill)) {
            // draw the next chunk: own eighth of the buffer first (one cursor per XCD, so that an XCD's L2 sees one
            // contiguous -- with binning: spatially compact -- part of the rays), then the other eighths
            cur = lim = 0;
            while (more) {
                const uint32_t first = q * per_q;
                const uint32_t left = first >= n_chunks ? 0u : (per_q < n_chunks - first ? per_q : n_chunks - first);
                unsigned long long c0 = ~0ull;
                if (left && (threadIdx.x & 63) == 0) c0 = atomicAdd(&work[q * kCursorStride], 1ull);
                const uint32_t got = __builtin_amdgcn_readfirstlane((uint32_t)c0);
                if (got < left) {
                    chunk = first + got;
                    const uint64_t rest = n - (uint64_t)chunk * strip;
                    lim = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(rest < strip ? rest : strip));
                    break;
                }
                // This eighth is exhausted.  Thousands of waves get here at the end of a launch; failed draws on the
                // other cursors would serialise in the L2 (0.75 ms per launch, measured), so exhausted eighths are
                // published once in a mask word on its own line and later waves skip them after one plain load.
                uint32_t mask = 0;
                if ((threadIdx.x & 63) == 0) {
                    mask = (uint32_t)__hip_atomic_load(&work[8 * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(mask & (1u << q))) atomicOr(&work[8 * kCursorStride], 1ull << q);
                }
                q_done |= __builtin_amdgcn_readfirstlane(mask) | (1u << q);
                const uint32_t cand = ~q_done & 0xffu;
                if (cand == 0u) { more = false; break; }
                const uint32_t rot = ((cand >> q) | (cand << (8u - q))) & 0xffu;        // next open eighth after q, cyclic
                q = (q + (uint32_t)__builtin_ctz(rot)) & 7u;
            }
        }
        if (QUEUE) {
            if (q_avail == 0u && cur < lim) {
                // stage the next rays of the chunk: lane l < cnt takes ray cur + l, whatever else it is doing
                const uint32_t lane = threadIdx.x & 63u;
                const uint32_t left = lim - cur, cnt = left < (uint32_t)kQueue ? left : (uint32_t)kQueue;
                if (lane < cnt) {
                    const uint64_t cand = (uint64_t)chunk * strip + cur + lane;
                    const uint32_t idx = perm ? perm[cand] : (uint32_t)cand;
                    const float4* rp = reinterpret_cast<const float4*>(rays + idx);
                    const float4 r0 = load_stream(rp, nt), r1 = load_stream(rp + 1, nt);
                    const Slab ns = make_slab<MODE>(mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), s.ray_pad);
                    qw[lane] = r0;
                    qw[kQueue + lane] = r1;
                    qw[2 * kQueue + lane] = make_float4(ns.ix, ns.iy, ns.iz, ns.m2);
                    qw[3 * kQueue + lane] = make_float4(ns.ex, ns.ey, ns.ez, __uint_as_float(idx));
                }
                cur += cnt; q_head = 0u; q_avail = cnt;
            }
            if (q_avail != 0u && n_idle >= refill) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                fresh = idle && rank < q_avail;
                if (fresh) {
                    const float4* qe = qw + (q_head + rank);
                    const float4 r0 = qe[0], r1 = qe[kQueue], a = qe[2 * kQueue], e = qe[3 * kQueue];
                    o = mk(r0.x, r0.y, r0.z); d = mk(r1.x, r1.y, r1.z); tmin = r0.w; tmax = r1.w;
                    my32 = __float_as_uint(e.w);
                    tmin0 = fmaxf(tmin, 0.0f);
                    b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
                    sl.ix = a.x; sl.iy = a.y; sl.iz = a.z; sl.m2 = a.w; sl.ex = e.x; sl.ey = e.y; sl.ez = e.z;
                    if (COUNT && tmax > tmin) ++n_mine;
                    if (s.n_spheres && tmax > tmin) spheres_pass<COUNT>(s, o, d, tmin, b, c);
                    // A ray with an empty interval, a sphere hit that settles an any-hit query, a scene without triangles: the
                    // lane is through at once and the write at the end of this round records it (`fresh` keeps it live until then).
                    const bool go = tmax > tmin && !(ANY && b.t < tmax) && s.n_nodes != 0;
                    node = go ? 0 : kDone; sp = 1;
                }
                const uint32_t took = (uint32_t)n_idle < q_avail ? (uint32_t)n_idle : q_avail;
                q_head += took; q_avail -= took;
            }
        } else
        if (cur < lim && n_idle >= refill) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            fresh = idle && cur + rank < lim;
            if (fresh) {
                const uint64_t cand = (uint64_t)chunk * strip + cur + rank;
                my = perm ? (uint64_t)perm[cand] : cand;
                const float4* rp = reinterpret_cast<const float4*>(rays + my);
                float4 r0 = load_stream(rp, nt), r1 = load_stream(rp + 1, nt);
                o = mk(r0.x, r0.y, r0.z); d = mk(r1.x, r1.y, r1.z); tmin = r0.w; tmax = r1.w;
                tmin0 = fmaxf(tmin, 0.0f);
                b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
                if (COUNT && tmax > tmin) ++n_mine;             // counters.rays = LIVE rays (dead slots are handed out too)
                if (s.n_spheres && tmax > tmin) spheres_pass<COUNT>(s, o, d, tmin, b, c);
                sl = make_slab<MODE>(o, d, s.ray_pad);
                // empty interval, settled any-hit query, no triangles: through at once; the write at the end of this round
                // records the lane (`fresh` keeps it live until then)
                const bool go = tmax > tmin && !(ANY && b.t < tmax) && s.n_nodes != 0;
                node = go ? 0 : kDone; sp = 1;
            }
            cur += (uint32_t)n_idle;
        }
        const bool live = !(node == kDone && pend == 0) || fresh;
        // One exit and no `continue`: a round in which no lane is live (between chunks, at the very end) falls through the body,
        // where every block is skipped by its own vote.  With a `continue` here the compiler kept two copies of every value that
        // is carried round the loop (33 more v_mov in the kernel, a dozen of them per round: bounce 3.42 -> 3.39 ms).
        if (__ballot(live) == 0ull && cur >= lim && !more && q_avail == 0u) break;
        // if-if scheduling with one postponed leaf per lane: per round every lane does at most ONE interior
        // step and ONE triangle test, so no lane waits for the slowest lane's whole descent.  A lane that
        // reaches a leaf parks it in `pend` and keeps walking from its stack; the triangle block is skipped
        // (wave-uniform branch) until at least `tth` lanes hold a parked leaf or no lane can walk any further.
        const bool need_i = node >= 0;
        if (COUNT) { ++w_rounds; w_i += __popcll(__ballot(need_i)); w_idle += __popcll(__ballot(node == kDone && pend == 0)); }
        if (need_i) node = bvh_step<MODE, COUNT, CAPPED>(nodes, node, sl, tmin0, b.t, stk, sp, c, spill);
#ifdef RT_PEND2
        if (node < 0 && node != kDone && pend2 == 0) {
            if (pend == 0) pend = node; else pend2 = node;
            --sp;
            node = stk[(CAPPED ? min(sp, lds_cap - 1) : sp) * kBlock];
            if (CAPPED && sp >= lds_cap) node = *spill_slot(spill, sp);
        }
#else
        if (node < 0 && node != kDone && pend == 0) {
            pend = node; --sp;
            node = stk[(CAPPED ? min(sp, lds_cap - 1) : sp) * kBlock];             // plain ds_read; deep lanes fix it up below
            if (CAPPED && sp >= lds_cap) node = *spill_slot(spill, sp);
        }
#endif
        // A second interior step in the same round when most of the wave still stands on interior nodes: the round's fixed
        // work (refill vote, leaf parking, triangle-block vote, result write: ~65 VALU instructions against 114 for the node
        // step) is then spread over two steps.  Below the threshold the step would run half empty and cost more than it saves
        // (thresholds 24..44 measured alike, a third step is slower: profiles/r02u_two_steps.txt).  Bounce 3.67 -> 3.57 ms.
        if ((int)__popcll(__ballot(node >= 0)) >= kSecondStepLanes) {
            if (COUNT) { w_i += __popcll(__ballot(node >= 0)); }
            if (node >= 0) node = bvh_step<MODE, COUNT, CAPPED>(nodes, node, sl, tmin0, b.t, stk, sp, c, spill);
#ifdef RT_PEND2
            if (node < 0 && node != kDone && pend2 == 0) {
                if (pend == 0) pend = node; else pend2 = node;
                --sp;
                node = stk[(CAPPED ? min(sp, lds_cap - 1) : sp) * kBlock];
                if (CAPPED && sp >= lds_cap) node = *spill_slot(spill, sp);
            }
#else
            if (node < 0 && node != kDone && pend == 0) {
                pend = node; --sp;
                node = stk[(CAPPED ? min(sp, lds_cap - 1) : sp) * kBlock];
                if (CAPPED && sp >= lds_cap) node = *spill_slot(spill, sp);
            }
#endif
        }
        const bool need_t = pend != 0;
        const unsigned long long mt = __ballot(need_t);
        if (COUNT) { w_blocked += __popcll(__ballot(pend != 0 && node < 0 && node != kDone)); w_pend += __popcll(__ballot(pend != 0 && node == kDone)); }
        if ((int)__popcll(mt) >= tth || __ballot(node >= 0) == 0ull) {
            if (COUNT) { w_t += __popcll(mt); ++w_tblocks; }
            if (need_t) {
                const uint32_t ref = ~(uint32_t)pend;              // (first << 3) | (remaining - 1)
                const float4* tp = tris + (size_t)(ref >> 3) * 3;
                float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                float t, u, v;
                if (tri_hit(o, d, mk(q0.x, q0.y, q0.z), mk(q1.x, q1.y, q1.z), mk(q2.x, q2.y, q2.z), t, u, v))
                    consider(b, tmin, t, u, v, __float_as_uint(q0.w) | kKeyFlip);
                if (COUNT) ++c.tris;
#ifdef RT_PEND2
                if (ANY && b.t < tmax) { node = kDone; pend = 0; pend2 = 0; sp = 1; }
                else if (ref & 7u) pend = (int)~(ref + 7u);
                else { pend = pend2; pend2 = 0; }
#else
                if (ANY && b.t < tmax) { node = kDone; pend = 0; sp = 1; }
                else pend = (ref & 7u) ? (int)~(ref + 7u) : 0;       // next triangle of the leaf: first+1, remaining-1
#endif
            }
        }
        if (live && node == kDone && pend == 0) {
            const uint64_t at = QUEUE ? (uint64_t)my32 : my;
            if (ANY) occ[at] = b.t < tmax ? 1 : 0;
            else     store_stream(hits + at, pack_hit(b, tmax), nt);
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
        if ((threadIdx.x & 63) == 0) {       // scheduling probes of this kernel (they reuse the divergence slots):
            atomicAdd(&counters[7], (unsigned long long)w_rounds);      // waves      <- scheduling rounds
            atomicAdd(&counters[5], (unsigned long long)w_i);           // wave_max_nodes <- lanes active in interior blocks
            atomicAdd(&counters[6], (unsigned long long)w_t);           // wave_max_tris  <- lanes active in triangle blocks
            atomicAdd(&counters[8], (unsigned long long)w_idle);        // lanes without a ray (waiting for the next refill), summed over rounds
            atomicAdd(&counters[9], (unsigned long long)w_blocked);     // lanes holding a parked leaf AND standing on a second one
            atomicAdd(&counters[10], (unsigned long long)w_pend);       // lanes with an empty stack waiting for the triangle block
            atomicAdd(&counters[11], (unsigned long long)w_tblocks);    // rounds in which the triangle block ran
        }
    }
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

// ---- shading (frag.glsl:205-263) + one bounce of the path loop (:310-354) ------------------------------
// Same evaluation order as oracle.c: pow(x,5) = ((x*x)*(x*x))*x, mix(a,b,t) = fma(b,t,a*(1-t)), PI = fp32(3.14159265...).
constexpr float kPi = 3.14159274f;
__device__ __forceinline__ float max0(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __forceinline__ float schlick_ggx(float ndv, float roughness) {          // :221-228
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    return ndv / __fmaf_rn(ndv, 1.0f - k, k);
}
struct Mat { f3 albedo, emission; float strength, roughness, metallic; };
__device__ __forceinline__ Mat load_mat(const DevScene& s, uint32_t id) {
    const float4* mp = s.materials + (size_t)id * 3;
    float4 a = mp[0], b = mp[1], c = mp[2];
    Mat m; m.albedo = mk(a.x, a.y, a.z); m.strength = a.w; m.emission = mk(b.x, b.y, b.z); m.roughness = b.w; m.metallic = c.x;
    return m;
}
__device__ __forceinline__ f3 brdf(const Mat& m, f3 N, f3 V, f3 L) {               // :239-263
    f3 H = norm3(mk(V.x + L.x, V.y + L.y, V.z + L.z));
    float NdotL = max0(dot3(N, L)), NdotV = max0(dot3(N, V)), NdotH = max0(dot3(N, H)), VdotH = max0(dot3(V, H));
    float om = 1.0f - m.metallic;
    f3 F0 = mk(__fmaf_rn(m.albedo.x, m.metallic, 0.04f * om), __fmaf_rn(m.albedo.y, m.metallic, 0.04f * om),
               __fmaf_rn(m.albedo.z, m.metallic, 0.04f * om));
    float x = 1.0f - VdotH, x2 = x * x, p5 = (x2 * x2) * x;
    f3 F = mk(__fmaf_rn(1.0f - F0.x, p5, F0.x), __fmaf_rn(1.0f - F0.y, p5, F0.y), __fmaf_rn(1.0f - F0.z, p5, F0.z));
    float a = m.roughness * m.roughness, a2 = a * a;
    float dd = __fmaf_rn(NdotH * NdotH, a2 - 1.0f, 1.0f);
    float D = a2 / ((kPi * dd) * dd);
    float G = schlick_ggx(NdotL, m.roughness) * schlick_ggx(NdotV, m.roughness);
    float denominator = __fmaf_rn(4.0f * NdotV, NdotL, 0.001f);
    float kd = 1.0f - m.metallic;
    f3 out;
    out.x = (((1.0f - F.x) * kd) * m.albedo.x) / kPi + ((D * F.x) * G) / denominator;
    out.y = (((1.0f - F.y) * kd) * m.albedo.y) / kPi + ((D * F.y) * G) / denominator;
    out.z = (((1.0f - F.z) * kd) * m.albedo.z) / kPi + ((D * F.z) * G) / denominator;
    return out;
}

__device__ __forceinline__ f3 hemisphere(f3 N, float xi1, float xi2) {             // :185-203
    float ct = sqrtf(1.0f - xi1), st = sqrtf(xi1);
    float sphi, cphi;
    sincos_2pi(xi2, sphi, cphi);
    float xs = st * cphi, ys = ct, zs = st * sphi;
    f3 up = fabsf(N.z) < 0.999f ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);
    f3 tX = norm3(cross3(up, N));
    f3 tY = cross3(N, tX);
    f3 dir = mk(__fmaf_rn(N.x, ys, __fmaf_rn(tY.x, zs, tX.x * xs)), __fmaf_rn(N.y, ys, __fmaf_rn(tY.y, zs, tX.y * xs)),
                __fmaf_rn(N.z, ys, __fmaf_rn(tY.z, zs, tX.z * xs)));
    return norm3(dir);
}

// One segment of the path loop (frag.glsl:310-354) for one path: T, C in/out; returns false when the path ends
// here (miss or dead ray: T and C are left as they are).  With want_next the bounce ray (o, dir) is produced
// and T updated (:340-350); key = index of the path in the FULL frame (decomposition-invariant RNG).
// PACKET (tile kernel with packet traversal only): the function is called by ALL lanes of the wave, `valid` says which
// of them carry a path; the shadow rays of a wave -- neighbouring surface points, one light -- then walk the tree together
// (trace_packet) instead of lane by lane with stacks in LDS.  Same closest hits, so the same lit / unlit decisions.
template <int MODE, bool PACKET = false>
__device__ __forceinline__ bool shade_one(const DevScene& s, float4 r0, float4 r1, float4 hit, f3& T, f3& C, uint32_t key,
                                          uint32_t seed, bool want_next, int* stk, f3& o, f3& dir, bool valid = true) {
    f3 P = mk(0.0f, 0.0f, 0.0f), N = mk(0.0f, 0.0f, 1.0f);
    f3 d = mk(r1.x, r1.y, r1.z);
    const bool ok = valid && (r1.w > r0.w) && hit_point(s, mk(r0.x, r0.y, r0.z), d, hit, P, N);
    if (!PACKET && !ok) return false;                                               // :351-353 the path ends
    if (PACKET && s.n_lights == 0 && !ok) return false;                             // no wave-wide work ahead
    const uint32_t prim = ok ? __float_as_uint(hit.w) : 0u;
    const uint32_t mid = (prim & RT_SPHERE_BIT) ? s.sphere_mat[prim & ~RT_SPHERE_BIT]
                                                : __float_as_uint(s.normals[(size_t)prim * 3].w);
    const Mat m = load_mat(s, mid);
    // lanes that only follow the packet (PACKET && !ok: no path, or a path that missed) read primitive 0's material as a
    // placeholder and must not add its emission: their radiance stays as it is, as on the early-return paths above
    if (ok) C = mk(__fmaf_rn(T.x * m.emission.x, m.strength, C.x), __fmaf_rn(T.y * m.emission.y, m.strength, C.y),
                   __fmaf_rn(T.z * m.emission.z, m.strength, C.z));                 // :313
    f3 V = norm3(mk(-d.x, -d.y, -d.z));                                             // :316
    for (uint32_t li = 0; li < s.n_lights; ++li) {                                  // :319-338
        float4 l0 = s.lights[2 * li], l1 = s.lights[2 * li + 1];
        f3 toL = sub3(mk(l0.x, l0.y, l0.z), P);
        f3 L = norm3(toL);
        float distance = sqrtf(dot3(toL, toL));
        float attenuation = 1.0f / (distance * distance);
        f3 so = madd3(0.001f, N, P);
        Best b; Counters c = { 0, 0, 0, 0 };
        const float tmax = 1e20f;
        if (PACKET) {
            b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
            if (ok) spheres_pass<false>(s, so, L, 0.0f, b, c);
            trace_packet<false>(s, so, L, 0.0f, ok, b, c);
            if (!ok) continue;
        } else {
            trace_ray<MODE, false, false>(s, so, L, 0.0f, tmax, b, c, stk);
        }
        bool lit = true;
        if (b.t < tmax) {
            f3 dl = sub3(madd3(b.t, L, so), P);
            lit = sqrtf(dot3(dl, dl)) > distance;                                   // :330
        }
        if (lit) {
            f3 B = brdf(m, N, V, L);
            float NdotL = max0(dot3(N, L));
            float ia = l0.w * attenuation;
            C = mk(__fmaf_rn((T.x * B.x) * (l1.x * ia), NdotL, C.x), __fmaf_rn((T.y * B.y) * (l1.y * ia), NdotL, C.y),
                   __fmaf_rn((T.z * B.z) * (l1.z * ia), NdotL, C.z));                // :335-336
        }
    }
    if (PACKET && !ok) return false;
    if (!want_next) return true;           // last segment of a path: only its radiance is still needed
    dir = hemisphere(N, rand01(key, 2u, seed), rand01(key, 3u, seed));              // :340
    o = madd3(0.001f, N, P);                                                        // :342
    float NdotR = max0(dot3(N, dir));                                               // :345
    float pdf = NdotR / kPi;                                                        // :346
    f3 B = brdf(m, N, V, dir);                                                      // :348
    T = mk(T.x * ((B.x * NdotR) / pdf), T.y * ((B.y * NdotR) / pdf), T.z * ((B.z * NdotR) / pdf));   // :350
    return true;
}

// Primary rays (frag.glsl:89-128) fused with traceRay: the "tile kernel".  One wave = a 4x4 pixel tile x 4 samples,
// 8x4 x 2 or 8x8 x 1 (the largest sample group that divides spp).  Two forms share this body: one ray per lane with its
// own stack in LDS (k_primary: any layout), and the packet walk of the whole wave (k_primary_packet: BVH4q trees with the
// wide node copy; the default).
// With SHADE the first segment of the path loop (:310-354, throughput 1, radiance 0) runs in the same kernel:
// the primary ray and its hit never travel through HBM (rays_out / hits become optional outputs), and the
// path-state buffers are written once instead of filled, read and written.
// frag.glsl:89-128 for pixel (x, y) of a w x h image; key = index of the path in the full frame (RNG)
__device__ __forceinline__ void camera_ray(const DevCamera& cam, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t key, uint32_t seed,
                                           int jitter, f3& o, f3& d) {
    float uvx = ((float)x + 0.5f) / (float)w;                     // vert.glsl:9-12 pixel centre
    float uvy = ((float)y + 0.5f) / (float)h;
    float ndcx = __fmaf_rn(uvx, 2.0f, -1.0f), ndcy = __fmaf_rn(uvy, 2.0f, -1.0f);   // :91
    if (jitter) {                                                 // :94-103
        ndcx += (rand01(key, 0u, seed) - 0.5f) * (2.0f / 1920.0f);
        ndcy += (rand01(key, 1u, seed) - 0.5f) * (2.0f / 1080.0f);
    }
    float a = ndcx * cam.half_w, bb = ndcy * cam.half_h;          // :114-115
    d = mk(__fmaf_rn(bb, cam.ux, a * cam.rx) + cam.fx, __fmaf_rn(bb, cam.uy, a * cam.ry) + cam.fy,
           __fmaf_rn(bb, cam.uz, a * cam.rz) + cam.fz);
    d = norm3(d);                                                 // :113
    d = norm3(d);                                                 // :120
    o = mk(cam.px, cam.py, cam.pz);
}

struct ShadeOut { float4* thr; float4* rad; float4* next; uint32_t seed; };
template <int MODE, bool COUNT, bool SHADE, bool PACKET>
__device__ __forceinline__ void primary_body(int* stack /* dynamic LDS */, const DevScene& s, const DevCamera& cam, uint32_t w, uint32_t h, uint32_t y0,
                                             uint32_t rows, uint32_t stripe_h, uint32_t stripe_step,
                                             uint32_t spp, uint32_t seed, int jitter, uint32_t tiles_x, uint32_t n_waves, float4* __restrict__ rays_out,
                                             float4* __restrict__ hits, unsigned long long* counters, const ShadeOut& so, int quad) {
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    // lane -> (pixel, sample): a wave takes S = 4, 2 or 1 samples (the largest that divides spp; quad = log2 S) of a
    // 4x4, 8x4 or 8x8 pixel tile -- the samples of one pixel are almost the same ray, so the wave stays together
    // longer (lane utilisation 83 -> 89 % on C3).  tiles_x counts tiles of that width.
    const uint32_t S = 1u << quad, pw = quad == 2 ? 4u : 8u, ph = quad == 0 ? 8u : 4u;
    const uint32_t groups = spp >> quad, tile = wave / groups, sg = wave - tile * groups;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x, pl = lane >> quad;
    const uint32_t x = tx * pw + (pl & (pw - 1u)), yl = ty * ph + pl / pw, smp = sg * S + (lane & (S - 1u));
    // local row yl -> image row: stripes of stripe_h rows, every stripe_step-th stripe belongs to this launch
    const uint32_t y = y0 + (yl / stripe_h) * stripe_h * stripe_step + (yl % stripe_h);
    const bool active = wave < n_waves && x < w && yl < rows && y < h;
    Counters c = { 0, 0, 0, 0 };
    f3 o = mk(0.0f, 0.0f, 0.0f), d = mk(0.0f, 0.0f, 0.0f);
    const float tmin = 0.0f, tmax = 1e20f;                            // :266
    const uint32_t key = (y * w + x) * spp + smp;                     // RNG key: full-image ray index
    Best b;
    if (active) camera_ray(cam, x, y, w, h, key, seed, jitter, o, d);
    int* stk = stack + (MODE == kBvh2 || MODE == kBvh4q ? threadIdx.x : 0);
    if (PACKET) {
        // the whole wave walks the tree together (trace_packet); lanes without a ray only follow
        b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;               // frag.glsl:266
        if (active) spheres_pass<COUNT>(s, o, d, tmin, b, c);
        trace_packet<COUNT>(s, o, d, tmin, active, b, c);
    } else if (active) {
        if (MODE != kNone) trace_ray<MODE, false, COUNT>(s, o, d, tmin, tmax, b, c, stk);
        else { b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u; }       // ray generation only
    }
    // segment 0 of the path loop.  In the packet form the shadow rays of the wave walk the tree together, so every lane
    // makes the call (lanes without a ray or without a hit only follow: `valid`).
    f3 T = mk(1.0f, 1.0f, 1.0f), C = mk(0.0f, 0.0f, 0.0f), no = mk(0.0f, 0.0f, 0.0f), nd = mk(0.0f, 0.0f, 0.0f);   // :306-307
    bool go = false;
    if (SHADE && PACKET) {
        const float4 r0 = make_float4(o.x, o.y, o.z, tmin), r1 = make_float4(d.x, d.y, d.z, tmax), hit = pack_hit(b, tmax);
        go = shade_one<MODE, true>(s, r0, r1, hit, T, C, key, so.seed, so.next != nullptr, stk, no, nd, active);
    }
    if (active) {
        const size_t r = ((size_t)yl * w + x) * spp + smp;
        const float4 r0 = make_float4(o.x, o.y, o.z, tmin), r1 = make_float4(d.x, d.y, d.z, tmax), hit = pack_hit(b, tmax);
        if (rays_out) { rays_out[2 * r] = r0; rays_out[2 * r + 1] = r1; }
        if (hits) hits[r] = hit;
        if (SHADE) {
            if (!PACKET) go = shade_one<MODE>(s, r0, r1, hit, T, C, key, so.seed, so.next != nullptr, stk, no, nd);
            // path state and bounce rays are read again only after the whole frame's worth of them has been written (and the
            // bounce launch wants the tree in the caches, not them): past the caches when the buffers are large.  Bench frame:
            // the BOUNCE launch that follows gains 0.8 %, C5's primary pass 1.5 % (profiles/r05j_nontemporal.txt)
            const bool pnt = (uint64_t)n_waves * 64u >= kStreamMinRays;
            store_stream(so.thr + r, make_float4(T.x, T.y, T.z, 0.0f), pnt);
            store_stream(so.rad + r, make_float4(C.x, C.y, C.z, 0.0f), pnt);
            if (so.next) {
                store_stream(so.next + 2 * r, go ? make_float4(no.x, no.y, no.z, 0.0f) : make_float4(0, 0, 0, 0), pnt);
                store_stream(so.next + 2 * r + 1, go ? make_float4(nd.x, nd.y, nd.z, 1e20f) : make_float4(0, 0, 0, 0), pnt);
            }
        }
    } else if (wave < n_waves && x < w && yl < rows) {
        // padding row of an interleaved share (image row >= h): a miss record and a dead ray, so that
        // later stages that sweep the whole local buffer see well-defined input
        const size_t r = ((size_t)yl * w + x) * spp + smp;
        if (rays_out) { rays_out[2 * r] = make_float4(0, 0, 0, 0); rays_out[2 * r + 1] = make_float4(0, 0, 0, 0); }
        if (hits) hits[r] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(RT_MISS));
        if (SHADE) {
            so.thr[r] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            so.rad[r] = make_float4(0, 0, 0, 0);
            if (so.next) { so.next[2 * r] = make_float4(0, 0, 0, 0); so.next[2 * r + 1] = make_float4(0, 0, 0, 0); }
        }
    }
    flush_counters<COUNT>(c, active, counters);
}

template <int MODE, bool COUNT, bool SHADE>
__global__ __launch_bounds__(kBlock) void k_primary(DevScene s, DevCamera cam, uint32_t w, uint32_t h, uint32_t y0,
                                                    uint32_t rows, uint32_t stripe_h, uint32_t stripe_step,
                                                    uint32_t spp, uint32_t seed, int jitter, uint32_t tiles_x, uint32_t n_waves, float4* __restrict__ rays_out,
                                                    float4* __restrict__ hits, unsigned long long* counters, ShadeOut so, int quad) {
    extern __shared__ int stack[];      // (depth + 2) entries x kBlock threads, sized at launch
    primary_body<MODE, COUNT, SHADE, false>(stack, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter, tiles_x, n_waves, rays_out, hits, counters, so, quad);
}

// The packet form (trace_packet).  A wave waits for one scalar node fetch after the other, so the kernel lives on waves
// per SIMD: the traversal loops need 40 VGPRs, the fused shading segment 108.  Asked for 8 waves (64 VGPRs) the allocator
// spills 33 registers, all of them in the shading code that runs once per ray, none in the traversal loops: primary +
// segment 0 2.12 -> 1.81 ms (5 waves 1.98, 6 waves 1.92; profiles/r02z_packet.txt).
#ifndef RT_PACKET_WAVES
#define RT_PACKET_WAVES 8
#endif
template <bool COUNT, bool SHADE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RT_PACKET_WAVES, 8)))
void k_primary_packet(DevScene s, DevCamera cam, uint32_t w, uint32_t h, uint32_t y0,
                      uint32_t rows, uint32_t stripe_h, uint32_t stripe_step,
                      uint32_t spp, uint32_t seed, int jitter, uint32_t tiles_x, uint32_t n_waves, float4* __restrict__ rays_out,
                      float4* __restrict__ hits, unsigned long long* counters, ShadeOut so, int quad) {
    primary_body<kBvh4q, COUNT, SHADE, true>(nullptr, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter, tiles_x, n_waves, rays_out, hits, counters, so, quad);
}

// The whole path loop of one sample (frag.glsl:306-354) in ONE kernel, for scenes that are traced by the reference's own
// brute-force loop (up to 32 triangles: the reference's Cornell box is 12 + 3 spheres).  Such a frame is bound by the memory
// traffic of the staged pipeline (192 bytes per path segment between the trace and shade kernels), not by its arithmetic, and
// with no traversal there is no divergence to manage: a thread keeps ray, throughput and radiance in registers over all bounces
// and writes 16 bytes at the end.  Same functions in the same order as the stages, hence the same radiance bit for bit.
__global__ __launch_bounds__(kBlock) RT_WAVES8 void k_path_brute(DevScene s, DevCamera cam, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows,
                                                       uint32_t stripe_h, uint32_t stripe_step, uint32_t spp, uint32_t bounces,
                                                       uint32_t seed, float4* __restrict__ rad) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint64_t)rows * w * spp) return;
    const uint32_t smp = (uint32_t)(i % spp), px = (uint32_t)(i / spp), x = px % w, yl = px / w;
    const uint32_t y = y0 + (yl / stripe_h) * stripe_h * stripe_step + (yl % stripe_h);
    f3 C = mk(0.0f, 0.0f, 0.0f);
    if (y < h) {                                                      // rows beyond the image: padding of an interleaved share
        const uint32_t key = (y * w + x) * spp + smp;
        f3 o, d, T = mk(1.0f, 1.0f, 1.0f);
        camera_ray(cam, x, y, w, h, key, seed, 1, o, d);
        for (uint32_t b = 0; b < bounces; ++b) {
            Best bst; Counters c = { 0, 0, 0, 0 };
            trace_ray<kBrute, false, false>(s, o, d, 0.0f, 1e20f, bst, c, nullptr);
            const float4 r0 = make_float4(o.x, o.y, o.z, 0.0f), r1 = make_float4(d.x, d.y, d.z, 1e20f), hit = pack_hit(bst, 1e20f);
            f3 no, nd;
            if (!shade_one<kBrute>(s, r0, r1, hit, T, C, key, seed + 1u + b, b + 1 < bounces, nullptr, no, nd)) break;   // :351-353
            o = no; d = nd;
        }
    }
    rad[i] = make_float4(C.x, C.y, C.z, 0.0f);
}

// frag.glsl:321-330 for one light, fused: shadow-ray build + closest-hit + lit test.
// PACKET (RT_OPT_RAY_PACKETS, BVH4q trees with the wide node copy): the 64 shadow rays of a wave walk the tree together
// (trace_packet) -- right when consecutive rays come from neighbouring surface points, as the hits of primary rays in pixel
// order do; the flags are the same for any buffer.
template <int MODE, bool PACKET>
__device__ __forceinline__ void shadow_body(int* stack, const DevScene& s, const rt_ray* __restrict__ rays,
                                            const float4* __restrict__ hits, uint64_t n, float lx, float ly,
                                            float lz, float4* __restrict__ srays, uint8_t* __restrict__ lit) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in = i < n;
    if (!PACKET && !in) return;
    f3 p = mk(0.0f, 0.0f, 0.0f), nn = mk(0.0f, 0.0f, 1.0f);
    bool valid = false;
    if (in) {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        float4 r0 = rp[0], r1 = rp[1];
        valid = hit_point(s, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z), hits[i], p, nn);
        if (!valid) {
            lit[i] = 2;
            if (srays) { srays[2 * i] = make_float4(0, 0, 0, 0); srays[2 * i + 1] = make_float4(0, 0, 0, 0); }
        }
    }
    if (!PACKET && !valid) return;
    f3 toL = sub3(mk(lx, ly, lz), p);
    f3 L = norm3(toL);                                       // :321
    float distance = sqrtf(dot3(toL, toL));                  // :322
    f3 so = madd3(0.001f, nn, p);                            // :326
    Best b; Counters c = { 0, 0, 0, 0 };
    const float tmax = 1e20f;
    if (PACKET) {
        b.t = tmax; b.u = 0.0f; b.v = 0.0f; b.key = 0u;
        if (valid) spheres_pass<false>(s, so, L, 0.0f, b, c);
        trace_packet<false>(s, so, L, 0.0f, valid, b, c);
        if (!valid) return;
    } else {
        trace_ray<MODE, false, false>(s, so, L, 0.0f, tmax, b, c, stack + (MODE != kBrute ? threadIdx.x : 0));
    }
    bool is_lit = true;
    if (b.t < tmax) {
        f3 dl = sub3(madd3(b.t, L, so), p);
        is_lit = sqrtf(dot3(dl, dl)) > distance;             // :330
    }
    lit[i] = is_lit ? 1 : 0;
    if (srays) { srays[2 * i] = make_float4(so.x, so.y, so.z, 0.0f); srays[2 * i + 1] = make_float4(L.x, L.y, L.z, tmax); }
}
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_shadow(DevScene s, const rt_ray* __restrict__ rays,
                                                   const float4* __restrict__ hits, uint64_t n, float lx, float ly,
                                                   float lz, float4* __restrict__ srays, uint8_t* __restrict__ lit) {
    extern __shared__ int stack[];      // (depth + 2) entries x kBlock threads, sized at launch
    shadow_body<MODE, false>(stack, s, rays, hits, n, lx, ly, lz, srays, lit);
}
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_shadow_packet(DevScene s, const rt_ray* __restrict__ rays, const float4* __restrict__ hits, uint64_t n, float lx, float ly,
                     float lz, float4* __restrict__ srays, uint8_t* __restrict__ lit) {
    shadow_body<kBvh4q, true>(nullptr, s, rays, hits, n, lx, ly, lz, srays, lit);
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

// FIRST: segment 0 of a path -- throughput 1 and radiance 0 are implied (frag.glsl:306-307) instead of read, and both
// buffers are written for every ray slot, dead ones included (what k_primary<SHADE> does in one kernel).
template <int MODE, bool FIRST>
__global__ __launch_bounds__(kBlock) RT_WAVES8_IF(MODE == kBrute) void k_shade(DevScene s, const rt_ray* __restrict__ rays,
                                                  const float4* __restrict__ hits, uint64_t n, uint32_t key0,
                                                  uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step,
                                                  uint32_t seed, float4* __restrict__ thr, float4* __restrict__ rad,
                                                  float4* __restrict__ out) {
    extern __shared__ int stack[];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // (plain loads and stores: nontemporal, the two 16-byte halves of a 32-byte ray record fetch their line twice, and the kernel
    // -- HBM-bound on large buffers -- takes 10-25 % longer: profiles/r05j_nontemporal.txt)
    const float4* rp = reinterpret_cast<const float4*>(rays + i);
    float4 r0 = rp[0], r1 = rp[1];
    float4 hit = hits[i];
    // RNG key = index of this path in the FULL frame, so that a striped share draws the same numbers
    uint32_t key = key0 + (uint32_t)i;
    if (row_len) {
        const uint32_t yl = (uint32_t)(i / row_len), rem = (uint32_t)(i - (uint64_t)yl * row_len);
        key = key0 + ((yl / stripe_h) * stripe_h * stripe_step + yl % stripe_h) * row_len + rem;
    }
    float4 t4 = make_float4(1.0f, 1.0f, 1.0f, 0.0f), c4 = make_float4(0, 0, 0, 0);
    const bool alive = r1.w > r0.w && __float_as_uint(hit.w) != RT_MISS;
    if (!FIRST && alive) { t4 = thr[i]; c4 = rad[i]; }
    f3 T = mk(t4.x, t4.y, t4.z), C = mk(c4.x, c4.y, c4.z), o, dir;
    const bool go = shade_one<MODE>(s, r0, r1, hit, T, C, key, seed, out != nullptr, stack + (MODE != kBrute ? threadIdx.x : 0), o, dir);
    if (FIRST) {
        thr[i] = make_float4(T.x, T.y, T.z, 0.0f);
        rad[i] = make_float4(C.x, C.y, C.z, 0.0f);
    }
    if (!go) {
        if (out) { out[2 * i] = make_float4(0, 0, 0, 0); out[2 * i + 1] = make_float4(0, 0, 0, 0); }
        return;
    }
    if (FIRST) {
        if (out == nullptr) return;
    } else {
        rad[i] = make_float4(C.x, C.y, C.z, 0.0f);
        if (out == nullptr) return;
        thr[i] = make_float4(T.x, T.y, T.z, 0.0f);
    }
    out[2 * i] = make_float4(o.x, o.y, o.z, 0.0f);
    out[2 * i + 1] = make_float4(dir.x, dir.y, dir.z, 1e20f);
}

// frag.glsl:357-359.  powf is the device libm's: the gamma step is compared with a tolerance, the linear
// average is bit-exact.
__global__ __launch_bounds__(kBlock) void k_resolve(const float4* __restrict__ rad, uint64_t n_pixels, uint32_t spp,
                                                    float4* __restrict__ lin, uint32_t* __restrict__ rgba) {
    const uint64_t p = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_pixels) return;
    float c[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t k = 0; k < spp; ++k) {
        float4 r = rad[p * spp + k];
        c[0] += r.x; c[1] += r.y; c[2] += r.z;
    }
    const float fs = (float)spp;
    c[0] = c[0] / fs; c[1] = c[1] / fs; c[2] = c[2] / fs;
    if (lin) lin[p] = make_float4(c[0], c[1], c[2], 1.0f);
    if (rgba) {
        uint32_t px = 0xff000000u;
        #pragma unroll
        for (int a = 0; a < 3; ++a) {
            float g = powf(c[a], 1.0f / 2.6f);
            g = g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g);
            if (!(g == g)) g = 0.0f;
            px |= (uint32_t)(g * 255.0f + 0.5f) << (8 * a);
        }
        rgba[p] = px;
    }
}

// Last segment of the path loop fused with the resolve (frag.glsl:310-338 + :357-359): a path's final radiance never
// goes through HBM.  Rays are laid out (pixel, sample) with spp consecutive samples per pixel and spp divides 64, so a
// pixel's samples sit in adjacent lanes of one wave; its first lane adds them in sample order (the same order, hence
// the same bits, as k_resolve) and writes the pixel.  d_radiance is read, not updated.
__device__ __forceinline__ uint32_t to_rgba8(float r, float g, float b) {
    const float c[3] = { r, g, b };
    uint32_t px = 0xff000000u;
    #pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = powf(c[a], 1.0f / 2.6f);
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        if (!(v == v)) v = 0.0f;
        px |= (uint32_t)(v * 255.0f + 0.5f) << (8 * a);
    }
    return px;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) RT_WAVES8_IF(MODE == kBrute) void k_shade_resolve(DevScene s, const rt_ray* __restrict__ rays,
                                                          const float4* __restrict__ hits, uint64_t n, uint32_t spp,
                                                          const float4* __restrict__ thr, const float4* __restrict__ rad,
                                                          float4* __restrict__ lin, uint32_t* __restrict__ rgba) {
    extern __shared__ int stack[];
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    f3 C = mk(0.0f, 0.0f, 0.0f);
    if (i < n && s.n_lights == 0) {
        // Without lights the last segment adds only the emission of what was hit (:313): neither the hit point nor the ray
        // is needed, so the 32-byte rays are not read at all (80 -> 48 bytes per path).  Relies on what every trace kernel
        // of this library guarantees: the hit record of a dead ray (tmax <= tmin) is a miss.
        const bool nt = n >= kStreamMinRays;
        const float4 hit = load_stream(hits + i, nt);
        const float4 c4 = load_stream(rad + i, nt);
        C = mk(c4.x, c4.y, c4.z);
        const uint32_t prim = __float_as_uint(hit.w);
        if (prim != RT_MISS) {
            const float4 t4 = load_stream(thr + i, nt);
            const uint32_t mid = (prim & RT_SPHERE_BIT) ? s.sphere_mat[prim & ~RT_SPHERE_BIT] : __float_as_uint(s.normals[(size_t)prim * 3].w);
            const Mat m = load_mat(s, mid);
            C = mk(__fmaf_rn(t4.x * m.emission.x, m.strength, C.x), __fmaf_rn(t4.y * m.emission.y, m.strength, C.y),
                   __fmaf_rn(t4.z * m.emission.z, m.strength, C.z));
        }
    } else if (i < n) {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        float4 r0 = rp[0], r1 = rp[1];
        float4 hit = hits[i];
        float4 c4 = rad[i];
        C = mk(c4.x, c4.y, c4.z);
        if (r1.w > r0.w && __float_as_uint(hit.w) != RT_MISS) {
            float4 t4 = thr[i];
            f3 T = mk(t4.x, t4.y, t4.z), o, dir;
            shade_one<MODE>(s, r0, r1, hit, T, C, 0u, 0u, false, stack + (MODE != kBrute ? threadIdx.x : 0), o, dir);
        }
    }
    const uint32_t lane = threadIdx.x & 63u, first = lane & ~(spp - 1u);
    float c[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t k = 0; k < spp; ++k) {                                   // wave-uniform trip count
        c[0] += __shfl(C.x, (int)(first + k), 64);
        c[1] += __shfl(C.y, (int)(first + k), 64);
        c[2] += __shfl(C.z, (int)(first + k), 64);
    }
    const float fs = (float)spp;
    c[0] = c[0] / fs; c[1] = c[1] / fs; c[2] = c[2] / fs;
    // The gamma step (three powf per pixel) was this kernel's largest cost, run by one lane in spp: now the lanes of a pixel
    // share it -- lane j of the pixel takes channel j (channels q*spp + j in pass q when spp < 3), ONE powf per lane and pass,
    // and the pixel's first lane collects the bytes.  Same function on the same value, so the same byte as to_rgba8.
    uint32_t px = 0xff000000u;
    if (rgba) {                                                            // wave-uniform
        const uint32_t j = lane - first, passes = (3u + spp - 1u) / spp;
        for (uint32_t q = 0; q < passes; ++q) {
            const uint32_t ch = q * spp + j;
            const float vin = ch == 0u ? c[0] : (ch == 1u ? c[1] : c[2]);
            float v = powf(vin, 1.0f / 2.6f);
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            if (!(v == v)) v = 0.0f;
            const uint32_t byte = (uint32_t)(v * 255.0f + 0.5f);
            for (uint32_t k = 0; k < spp && q * spp + k < 3u; ++k)       // wave-uniform trip count
                px |= (uint32_t)__shfl((int)byte, (int)(first + k), 64) << (8u * (q * spp + k));
        }
    }
    if (i < n && lane == first) {
        const uint64_t p = i / spp;
        if (lin) lin[p] = make_float4(c[0], c[1], c[2], 1.0f);
        if (rgba) rgba[p] = px;
    }
}

__global__ __launch_bounds__(kBlock) void k_fill(float4* __restrict__ d, uint64_t n, float4 v) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) d[i] = v;
}

// interleaved stripe shares -> image order (launch_gather_rows): one block row per image row, plain word copies
__global__ __launch_bounds__(kBlock) void k_gather_rows(const uint32_t* __restrict__ stage, uint32_t* __restrict__ frame, uint32_t row_words,
                                                        uint32_t stripe_h, uint32_t n_shares, uint32_t share_rows) {
    const uint32_t y = blockIdx.y, stripe = y / stripe_h, share = stripe % n_shares;
    const uint32_t local = (stripe / n_shares) * stripe_h + y % stripe_h;
    const uint32_t* src = stage + ((size_t)share * share_rows + local) * row_words;
    uint32_t* dst = frame + (size_t)y * row_words;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < row_words; i += gridDim.x * kBlock) dst[i] = src[i];
}

// ---- ray binning ------------------------------------------------------------------------------------------
// Counting sort of ray INDICES by origin cell (kernels.hpp), four launches, no global atomics: per-block
// histograms -> per-(block, bin) bases -> scatter with LDS cursors.  Rays of one bin keep no particular order.
__device__ __forceinline__ uint32_t spread4(uint32_t x) { return (x & 1u) | ((x & 2u) << 2) | ((x & 4u) << 4) | ((x & 8u) << 6); }
__device__ __forceinline__ uint32_t ray_bin(const rt_ray* __restrict__ rays, uint64_t i, const BinGrid& g) {
    const float4 r0 = reinterpret_cast<const float4*>(rays + i)[0];
    const float tmax = rays[i].tmax;
    if (!(tmax > r0.w)) return kBinCells;                                   // dead ray
    const int x = min(max((int)((r0.x - g.lox) * g.sx), 0), 15);
    const int y = min(max((int)((r0.y - g.loy) * g.sy), 0), 15);
    const int z = min(max((int)((r0.z - g.loz) * g.sz), 0), 15);
    return spread4((uint32_t)x) | (spread4((uint32_t)y) << 1) | (spread4((uint32_t)z) << 2);
}

__global__ __launch_bounds__(kBlock) void k_bin_count(const rt_ray* __restrict__ rays, uint64_t n, uint64_t per_block,
                                                      BinGrid g, uint32_t* __restrict__ hist /* [block][kBins] */) {
    __shared__ uint32_t h[kBins];
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) h[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock) atomicAdd(&h[ray_bin(rays, i, g)], 1u);
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * kBins;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) row[b] = h[b];
}

// hist[block][bin] -> exclusive prefix over blocks (in place); totals[bin] = rays of the bin.
// A workgroup takes 16 bins x 16 segments of the rows: a thread sums its segment, the segment sums are scanned through LDS,
// then the thread walks its segment again and writes the running prefix.  (One thread per bin over all rows took 57 us for
// 1,024 rows whatever the buffer's size -- a tenth of a 4 Mi-ray call's binning; profiles/r04e_c4_kernels_by_ray_count.txt.)
constexpr uint32_t kColBins = 16, kColSegs = kBlock / kColBins;
__global__ __launch_bounds__(kBlock) void k_bin_columns(uint32_t* __restrict__ hist, uint32_t n_blocks, uint32_t bins, uint32_t* __restrict__ totals,
                                                        uint32_t* __restrict__ meta /* [1] += non-empty (block, bin) pairs; null = do not count */,
                                                        const uint32_t* __restrict__ on /* null, or: return at once when *on == 0 */) {
    __shared__ uint32_t part[kColSegs][kColBins];
    if (on && *on == 0u) return;
    const uint32_t bl = threadIdx.x % kColBins, seg = threadIdx.x / kColBins;
    const uint32_t b = blockIdx.x * kColBins + bl;
    const uint32_t per = (n_blocks + kColSegs - 1) / kColSegs;
    const uint32_t r0 = min(seg * per, n_blocks), r1 = min(r0 + per, n_blocks);
    const bool in = b < bins;
    uint32_t sum = 0;
    if (in) {
        for (uint32_t k0 = r0; k0 < r1; k0 += 16) {             // 16 independent loads in flight
            uint32_t v[16];
            #pragma unroll
            for (uint32_t j = 0; j < 16; ++j) v[j] = (k0 + j < r1) ? hist[(size_t)(k0 + j) * bins + b] : 0u;
            #pragma unroll
            for (uint32_t j = 0; j < 16; ++j) sum += v[j];
        }
    }
    part[seg][bl] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t sgm = 0; sgm < kColSegs; ++sgm) { const uint32_t v = part[sgm][bl]; run += sgm < seg ? v : 0u; total += v; }
    if (!in) return;
    uint32_t pairs = 0;
    for (uint32_t k0 = r0; k0 < r1; k0 += 16) {
        uint32_t v[16];
        #pragma unroll
        for (uint32_t j = 0; j < 16; ++j) v[j] = (k0 + j < r1) ? hist[(size_t)(k0 + j) * bins + b] : 0u;
        #pragma unroll
        for (uint32_t j = 0; j < 16; ++j) {
            if (k0 + j < r1) hist[(size_t)(k0 + j) * bins + b] = run;
            run += v[j];
            pairs += v[j] != 0u;
        }
    }
    if (seg == 0) totals[b] = total;
    if (meta && pairs) atomicAdd(&meta[1], pairs);
}

// totals[bin] -> start of the bin in the permutation (exclusive scan, one block)
// decide: 1 = this launch takes the permute / keep decision (meta[0]) from the pair count; 0 = leaves it as it is
__global__ __launch_bounds__(kBlock) void k_bin_scan(uint32_t* __restrict__ totals, uint32_t bins, uint32_t* __restrict__ meta, uint64_t n,
                                                     uint32_t min_run, int decide /* 2 = the decision is taken already: nothing to do when it is 0 */) {
    if (decide == 2 && meta[0] == 0u) return;
    if (decide == 1 && threadIdx.x == 0) meta[0] = (min_run == 0u || n < (uint64_t)min_run * meta[1]) ? 1u : 0u;   // permute?
    __shared__ uint32_t part[kBlock];
    const uint32_t per = (bins + kBlock - 1) / kBlock;
    const uint32_t lo = threadIdx.x * per;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < per; ++k) if (lo + k < bins) sum += totals[lo + k];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < kBlock; ++k) { const uint32_t v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = 0; k < per; ++k) if (lo + k < bins) { const uint32_t v = totals[lo + k]; totals[lo + k] = run; run += v; }
}

__global__ __launch_bounds__(kBlock) void k_bin_scatter(const rt_ray* __restrict__ rays, uint64_t n, uint64_t per_block, BinGrid g,
                                                        const uint32_t* __restrict__ hist, const uint32_t* __restrict__ starts,
                                                        const uint32_t* __restrict__ meta, uint32_t* __restrict__ perm) {
    __shared__ uint32_t cur[kBins];
    if (meta[0] == 0u) return;                                               // coherent input: keep its order
    const uint32_t* row = hist + (size_t)blockIdx.x * kBins;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) cur[b] = starts[b] + row[b];
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    // four rays per thread and round: the loads of a round are in flight together (one ray per round left the kernel
    // waiting for one load after the other: 1.5 ms for 64 Mi rays)
    for (uint64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * kBlock) {
        uint32_t b[4];
        #pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t i = i0 + (uint64_t)k * kBlock; b[k] = i < hi ? ray_bin(rays, i, g) : 0u; }
        #pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t i = i0 + (uint64_t)k * kBlock; if (i < hi) perm[atomicAdd(&cur[b[k]], 1u)] = (uint32_t)i; }
    }
}

// ---- two-level variant for large buffers --------------------------------------------------------------------
// The one-level scatter keeps blocks x 4097 write streams open at once (16-entry runs that fill over a block's whole
// loop): far more lines than the L2s hold, so they are evicted half-filled and written several times (measured: 4.4 GB
// of HBM traffic for 2.3 GB of algorithmic bytes on 64 Mi rays, 1.5 ms).  Here the sort goes by the upper 7 bits of
// the bin first (65 streams per block, runs of ~1,000 entries) and by the whole bin second, on an array that is
// grouped by the upper bits already, so a block of the second level meets ~100 bins and writes runs of ~600 entries.
// The bin of every ray is computed once (16 bits per ray) instead of re-reading the 32-byte rays.
// Is the buffer grouped by origin already (camera-ordered bounce rays are)?  Decided from every kBinProbeStride-th (64th) ray before
// anything else runs: a buffer that keeps its order then costs a fraction of a counting pass instead of a whole one (C5: two
// bounce buffers of 132 M slots, 1.0 ms each just to find out that nothing is to be done).  meta[2] += non-empty (block, cell)
// pairs among the samples, meta[3] += samples; k_bin_decide sets meta[0] = 1 (permute) when a pair holds fewer than
// min_run / stride samples on average, and every later kernel of the sort returns at once when it is 0.
constexpr uint32_t kBinProbeStride = 64;
__global__ __launch_bounds__(kBlock) void k_bin_probe(const rt_ray* __restrict__ rays, uint64_t n, uint64_t per_block, BinGrid g,
                                                      uint32_t* __restrict__ meta) {
    __shared__ uint32_t h[kBins];
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) h[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    uint32_t mine = 0;
    for (uint64_t i = lo + (uint64_t)threadIdx.x * kBinProbeStride; i < hi; i += (uint64_t)kBlock * kBinProbeStride) { atomicAdd(&h[ray_bin(rays, i, g)], 1u); ++mine; }
    __syncthreads();
    uint32_t pairs = 0;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) pairs += h[b] != 0u;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) { pairs += __shfl_down(pairs, off, 64); mine += __shfl_down(mine, off, 64); }
    // ONE pair of atomics per block (through LDS), not per wave: thousands of atomics on one address serialise in their L2
    // channel -- with one per wave this kernel took 100 us whatever the buffer's size
    __shared__ uint32_t red[2][kBlock / 64];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = pairs; red[1][threadIdx.x >> 6] = mine; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t p = 0, m = 0;
        for (int k = 0; k < kBlock / 64; ++k) { p += red[0][k]; m += red[1][k]; }
        if (p) atomicAdd(&meta[2], p);
        if (m) atomicAdd(&meta[3], m);
    }
}
__global__ void k_bin_decide(uint32_t* __restrict__ meta, uint32_t min_run) {
    if (threadIdx.x == 0) meta[0] = (min_run == 0u || (uint64_t)meta[3] * kBinProbeStride < (uint64_t)min_run * meta[2]) ? 1u : 0u;
}

__global__ __launch_bounds__(kBlock) void k_bin2_count(const rt_ray* __restrict__ rays, uint64_t n, uint64_t per_block, BinGrid g,
                                                       uint16_t* __restrict__ keys, uint32_t* __restrict__ hist_hi /* [block][kBinsHi] */,
                                                       uint32_t* __restrict__ meta) {
    __shared__ uint32_t h[kBins];
    if (meta[0] == 0u) return;                                               // k_bin_probe / k_bin_decide: the buffer keeps its order
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) h[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
        const uint32_t b = ray_bin(rays, i, g);
        keys[i] = (uint16_t)b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    // upper-level histogram row (the coherence decision was taken by k_bin_probe / k_bin_decide before this kernel)
    if (threadIdx.x < kBinsHi) {
        uint32_t sum = 0;
        const uint32_t first = threadIdx.x << kBinLoBits, last = min(first + (1u << kBinLoBits), kBins);
        for (uint32_t b = first; b < last; ++b) sum += h[b];
        hist_hi[(size_t)blockIdx.x * kBinsHi + threadIdx.x] = sum;
    }
}

// The second-level kernels read 2 to 6 bytes per ray: with one ray per thread and round a CU would keep ~2 KB in flight and
// crawl at a tenth of the HBM rate (measured: 0.74 ms for 0.5 GB), so a thread takes 8 consecutive entries per round
// (one 16-byte load of keys, two of indices).  per_block is a multiple of 8 (launch_bin), so the loads stay aligned.
__device__ __forceinline__ void load_keys8(const uint16_t* __restrict__ keys, uint64_t i, uint64_t hi, uint32_t k[8]) {
    if (i + 8 <= hi) {
        const uint4 q = *reinterpret_cast<const uint4*>(keys + i);
        k[0] = q.x & 0xffffu; k[1] = q.x >> 16; k[2] = q.y & 0xffffu; k[3] = q.y >> 16;
        k[4] = q.z & 0xffffu; k[5] = q.z >> 16; k[6] = q.w & 0xffffu; k[7] = q.w >> 16;
    } else {
        #pragma unroll
        for (int j = 0; j < 8; ++j) k[j] = i + j < hi ? keys[i + j] : 0u;
    }
}

__global__ __launch_bounds__(kBlock) void k_bin2_scatter_hi(const uint16_t* __restrict__ keys, uint64_t n, uint64_t per_block,
                                                            const uint32_t* __restrict__ hist_hi, const uint32_t* __restrict__ starts_hi,
                                                            const uint32_t* __restrict__ meta, uint32_t* __restrict__ tmp_idx,
                                                            uint16_t* __restrict__ tmp_key) {
    __shared__ uint32_t cur[kBinsHi];
    if (meta[0] == 0u) return;                                               // coherent input: keep its order
    if (threadIdx.x < kBinsHi) cur[threadIdx.x] = starts_hi[threadIdx.x] + hist_hi[(size_t)blockIdx.x * kBinsHi + threadIdx.x];
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (uint64_t i = lo + (uint64_t)threadIdx.x * 8; i < hi; i += 8 * kBlock) {
        uint32_t k[8];
        load_keys8(keys, i, hi, k);
        #pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i + j < hi) {
                const uint32_t dst = atomicAdd(&cur[k[j] >> kBinLoBits], 1u);
                tmp_idx[dst] = (uint32_t)(i + j);
                tmp_key[dst] = (uint16_t)k[j];
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_bin2_count_keys(const uint16_t* __restrict__ tmp_key, uint64_t n, uint64_t per_block,
                                                            const uint32_t* __restrict__ meta, uint32_t* __restrict__ hist /* [block][kBins] */) {
    __shared__ uint32_t h[kBins];
    if (meta[0] == 0u) return;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) h[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (uint64_t i = lo + (uint64_t)threadIdx.x * 8; i < hi; i += 8 * kBlock) {
        uint32_t k[8];
        load_keys8(tmp_key, i, hi, k);
        #pragma unroll
        for (int j = 0; j < 8; ++j) if (i + j < hi) atomicAdd(&h[k[j]], 1u);
    }
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * kBins;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) row[b] = h[b];
}

__global__ __launch_bounds__(kBlock) void k_bin2_scatter_keys(const uint16_t* __restrict__ tmp_key, const uint32_t* __restrict__ tmp_idx, uint64_t n,
                                                              uint64_t per_block, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ starts,
                                                              const uint32_t* __restrict__ meta, uint32_t* __restrict__ perm) {
    __shared__ uint32_t cur[kBins];
    if (meta[0] == 0u) return;
    const uint32_t* row = hist + (size_t)blockIdx.x * kBins;
    for (uint32_t b = threadIdx.x; b < kBins; b += kBlock) cur[b] = starts[b] + row[b];
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = (lo + per_block < n) ? lo + per_block : n;
    for (uint64_t i = lo + (uint64_t)threadIdx.x * 8; i < hi; i += 8 * kBlock) {
        uint32_t k[8], id[8];
        load_keys8(tmp_key, i, hi, k);
        if (i + 8 <= hi) {
            const uint4 a = *reinterpret_cast<const uint4*>(tmp_idx + i), c = *reinterpret_cast<const uint4*>(tmp_idx + i + 4);
            id[0] = a.x; id[1] = a.y; id[2] = a.z; id[3] = a.w; id[4] = c.x; id[5] = c.y; id[6] = c.z; id[7] = c.w;
        } else {
            #pragma unroll
            for (int j = 0; j < 8; ++j) id[j] = i + j < hi ? tmp_idx[i + j] : 0u;
        }
        #pragma unroll
        for (int j = 0; j < 8; ++j) if (i + j < hi) perm[atomicAdd(&cur[k[j]], 1u)] = id[j];
    }
}

// LDS for the per-lane stacks: sentinel + one entry per BVH level (+1 spare), [entry][thread]
inline size_t stack_bytes(const DevScene& s, int mode) {
    return mode != kBrute ? (size_t)(s.stack_depth + 2) * kBlock * sizeof(int) : 0;
}
inline uint32_t grid_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }
inline int num_cus() {                        // of the current device (cached per device id)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cache[dev].load();
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev].store(v);
    }
    return v;
}

}  // namespace

int device_cus() { return num_cus(); }

hipError_t launch_bin(const rt_ray* d_rays, uint64_t n, const BinGrid& g, uint32_t* d_scratch, uint32_t min_run, hipStream_t st,
                      uint32_t max_blocks, bool two_level) {
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32) || max_blocks == 0) return hipErrorInvalidValue;
    uint32_t* perm = d_scratch;
    uint32_t* hist = d_scratch + n;
    uint32_t* totals = hist + (size_t)max_blocks * kBins;
    uint32_t* meta = totals + kBins;                 // [0] permute flag, [1] non-empty (block, bin) pairs
    const uint32_t n_blocks = (uint32_t)std::min<uint64_t>(max_blocks, (n + 4095) / 4096);
    const uint64_t per_block = (((n + n_blocks - 1) / n_blocks) + 7) & ~7ull;      // multiple of 8: aligned 16-byte loads in the second level
    hipError_t e = hipMemsetAsync(meta, 0, 4 * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (!two_level) {
        hipLaunchKernelGGL(k_bin_count, dim3(n_blocks), dim3(kBlock), 0, st, d_rays, n, per_block, g, hist);
        hipLaunchKernelGGL(k_bin_columns, dim3((kBins + kColBins - 1) / kColBins), dim3(kBlock), 0, st, hist, n_blocks, kBins, totals, meta, (const uint32_t*)nullptr);
        hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kBlock), 0, st, totals, kBins, meta, n, min_run, 1);
        hipLaunchKernelGGL(k_bin_scatter, dim3(n_blocks), dim3(kBlock), 0, st, d_rays, n, per_block, g, hist, totals, meta, perm);
        return hipGetLastError();
    }
    // two levels (see k_bin2_count): behind meta come the index array of the first level and the two 16-bit key arrays
    // (each array starts on a 16-byte boundary of the 256-byte-aligned scratch: the second level loads 16 bytes at a time)
    auto align4 = [](uint64_t words) { return (words + 3) & ~3ull; };
    const uint64_t off_idx = align4(n + (uint64_t)max_blocks * kBins + kBins + 4), off_keys = align4(off_idx + n), off_tmp = align4(off_keys + (n + 1) / 2);
    uint32_t* tmp_idx = d_scratch + off_idx;
    uint16_t* keys = reinterpret_cast<uint16_t*>(d_scratch + off_keys);
    uint16_t* tmp_key = reinterpret_cast<uint16_t*>(d_scratch + off_tmp);
    uint32_t* hist_hi = hist;                                                // rows of kBinsHi words; the full-width rows reuse the area afterwards
    uint32_t* totals_hi = totals;
    hipLaunchKernelGGL(k_bin_probe, dim3(n_blocks), dim3(kBlock), 0, st, d_rays, n, per_block, g, meta);
    hipLaunchKernelGGL(k_bin_decide, dim3(1), dim3(64), 0, st, meta, min_run);
    hipLaunchKernelGGL(k_bin2_count, dim3(n_blocks), dim3(kBlock), 0, st, d_rays, n, per_block, g, keys, hist_hi, meta);
    hipLaunchKernelGGL(k_bin_columns, dim3((kBinsHi + kColBins - 1) / kColBins), dim3(kBlock), 0, st, hist_hi, n_blocks, kBinsHi, totals_hi, (uint32_t*)nullptr, (const uint32_t*)meta);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kBlock), 0, st, totals_hi, kBinsHi, meta, n, min_run, 2);
    hipLaunchKernelGGL(k_bin2_scatter_hi, dim3(n_blocks), dim3(kBlock), 0, st, keys, n, per_block, hist_hi, totals_hi, meta, tmp_idx, tmp_key);
    hipLaunchKernelGGL(k_bin2_count_keys, dim3(n_blocks), dim3(kBlock), 0, st, tmp_key, n, per_block, meta, hist);
    hipLaunchKernelGGL(k_bin_columns, dim3((kBins + kColBins - 1) / kColBins), dim3(kBlock), 0, st, hist, n_blocks, kBins, totals, (uint32_t*)nullptr, (const uint32_t*)meta);
    hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(kBlock), 0, st, totals, kBins, meta, n, min_run, 2);
    hipLaunchKernelGGL(k_bin2_scatter_keys, dim3(n_blocks), dim3(kBlock), 0, st, tmp_key, tmp_idx, n, per_block, hist, totals, meta, perm);
    return hipGetLastError();
}

hipError_t launch_trace_packets(const DevScene& s, bool count, const rt_ray* d_rays, uint64_t n, const uint32_t* d_perm,
                                rt_hit* d_hits, unsigned long long* d_counters, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (!s.wide || s.stack_depth + 2 > 64) return hipErrorInvalidValue;
    dim3 g(grid_for(n)), b(kBlock);
    float4* hits = reinterpret_cast<float4*>(d_hits);
    if (count) hipLaunchKernelGGL((k_trace_packet<true>), g, b, 0, st, s, d_rays, n, d_perm, hits, d_counters);
    else       hipLaunchKernelGGL((k_trace_packet<false>), g, b, 0, st, s, d_rays, n, d_perm, hits, d_counters);
    return hipGetLastError();
}

hipError_t launch_trace(const DevScene& s, int mode, bool any_hit, bool count, const rt_ray* d_rays, uint64_t n,
                        rt_hit* d_hits, uint8_t* d_occ, unsigned long long* d_counters, hipStream_t st,
                        uint32_t strip, int refill, const uint32_t* d_perm, const uint32_t* d_perm_on, const StripScratch& sc) {
    if (n == 0) return hipSuccess;
    dim3 b(kBlock);
    float4* hits = reinterpret_cast<float4*>(d_hits);
    const int rf = refill & 0xff, tth = refill >> 8;
    // Large permuted buffers (random rays sorted by origin cell, 2^25 rays and more) do better with refills in bigger batches and
    // an earlier triangle block: 10 M-triangle soup, 64 Mi rays 24.8 -> 23.7 ms.  Not so smaller ones (8 Mi: 4.05 -> 4.22 ms, the
    // launch is HBM-bound) nor buffers that keep their order (C5's bounce rays: 48.6 -> 50.7 ms): profiles/r04u_refill_sweeps.txt.
    const int rf_perm = (d_perm && sc.refill_perm && n >= (1ull << 25)) ? sc.refill_perm : 0;
    if (mode != kBrute && strip >= 64) {
        // strip kernel: persistent waves, as many blocks as the chip holds at once (LDS-limited), never more than
        // there are chunks.  The chunk cursors live in the caller's scratch (sc.work), zeroed on the stream.
        if (!sc.work) return hipErrorInvalidValue;
        unsigned long long* work = sc.work;
        hipError_t e = hipMemsetAsync(work, 0, kWorkWords * sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
        // chunk size: the option is the maximum; smaller buffers get smaller chunks (down to one wave's worth), so that
        // a full grid still draws >= 8 chunks per wave and the launch does not end in a long tail of half-empty waves
        {
            const uint64_t want = n / (8ull * 8ull * (uint64_t)num_cus() * (kBlock / 64));     // rays per chunk for 8 chunks per wave
            uint32_t fit = 64;
            while ((uint64_t)fit * 2 <= want && fit * 2 <= strip) fit *= 2;
            strip = fit;
        }
        const int lds_cap = sc.lds_cap;
        int* d_spill = sc.spill;
        const bool capped = mode == kBvh4q && d_spill && lds_cap >= 2 && s.stack_depth + 2 > (uint32_t)lds_cap;
        const size_t lds_stack = capped ? (size_t)lds_cap * kBlock * sizeof(int) : stack_bytes(s, mode);
        // the staged ray queues sit behind the stacks (BVH4q, ray indices of 32 bits)
        // (a block may ask for 64 KB of dynamic LDS: a stack part that leaves no room for the queues goes without them)
        const bool queue = sc.queue && mode == kBvh4q && n < (1ull << 32) && lds_stack + kQueueBytesPerBlock <= 64u * 1024u;
        const int q_off = (int)(lds_stack / sizeof(int));
        const size_t lds = lds_stack + (queue ? kQueueBytesPerBlock : 0);
        const uint64_t per_cu = std::min<uint64_t>(8, std::max<uint64_t>(1, (160u * 1024u) / std::max<size_t>(lds, 1)));
        const uint64_t n_chunks = (n + strip - 1) / strip;
        if (n_chunks + 8 > 0xFFFFFFFFull) return hipErrorInvalidValue;      // the kernel counts chunks in 32 bits (2^38 rays at the smallest chunk)
        // a full grid when there are chunks for it (>= 8 per wave by the chunk size above, down to 1 for small buffers,
        // where putting every ray in flight at once matters more than the tail); sc.max_blocks bounds the spill columns
        uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((n_chunks + kBlock / 64 - 1) / (kBlock / 64), per_cu * (uint64_t)num_cus()));
        if (sc.max_blocks) blocks = std::min<uint64_t>(blocks, sc.max_blocks);
        dim3 g((uint32_t)blocks);
#define RT_LAUNCH(M, A, C, P, Q) hipLaunchKernelGGL((k_trace_strip<M, A, C, P, Q>), g, b, lds, st, s, d_rays, n, strip, rf, tth, rf_perm, work, d_perm, d_perm_on, d_spill, lds_cap, q_off, (int)sc.stream_nt, hits, d_occ, d_counters)
#define RT_LAUNCH_M(M, P, Q) do { if (any_hit) { if (count) RT_LAUNCH(M, true, true, P, Q); else RT_LAUNCH(M, true, false, P, Q); } \
                                  else         { if (count) RT_LAUNCH(M, false, true, P, Q); else RT_LAUNCH(M, false, false, P, Q); } } while (0)
        if (mode == kBvh4q) {
            if (queue) { if (capped) RT_LAUNCH_M(kBvh4q, true, true); else RT_LAUNCH_M(kBvh4q, false, true); }
            else       { if (capped) RT_LAUNCH_M(kBvh4q, true, false); else RT_LAUNCH_M(kBvh4q, false, false); }
        } else RT_LAUNCH_M(kBvh2, false, false);
#undef RT_LAUNCH
#undef RT_LAUNCH_M
        return hipGetLastError();
    }
    dim3 g(grid_for(n));
#define RT_LAUNCH(M, A, C) hipLaunchKernelGGL((k_trace<M, A, C>), g, b, stack_bytes(s, M), st, s, d_rays, n, hits, d_occ, d_counters)
#define RT_LAUNCH_M(M) do { if (any_hit) { if (count) RT_LAUNCH(M, true, true); else RT_LAUNCH(M, true, false); } \
                            else         { if (count) RT_LAUNCH(M, false, true); else RT_LAUNCH(M, false, false); } } while (0)
    if (mode == kBrute) RT_LAUNCH_M(kBrute); else if (mode == kBvh4q) RT_LAUNCH_M(kBvh4q); else RT_LAUNCH_M(kBvh2);
#undef RT_LAUNCH
#undef RT_LAUNCH_M
    return hipGetLastError();
}

// ---- allocation-free closest hit (rt_trace_dev_noalloc) -------------------------------------------------------
// Callers that cannot call hipMalloc (code under graph capture, serving loops) get the strip kernel with its scratch
// -- chunk cursors, spill columns of the capped stack, the binning permutation and histograms -- in this file-scope
// device array instead of per-context allocations.  Every GPU the library runs on holds its own copy (kNoAllocBytes,
// about 36 MiB) until the process ends.  `scratch` below is only a host-side stand-in for the symbol: host code uses
// the per-device address that hipGetSymbolAddress returned (noalloc_init), never the name.
constexpr size_t kNoAllocWorkBytes = 4096;                                                   // cursor set (kWorkWords u64), own lines
constexpr size_t kNoAllocSpillWords = (size_t)kNoAllocSpillEntries * kNoAllocMaxBlocks * kThreads;
constexpr size_t kNoAllocBinWords = (size_t)kNoAllocMaxRays + (size_t)kNoAllocBinBlocks * kBins + kBins + 4;
constexpr size_t kNoAllocBytes = ((kNoAllocWorkBytes + (kNoAllocSpillWords + kNoAllocBinWords) * 4) + 255) / 256 * 256;
static_assert(kWorkWords * sizeof(unsigned long long) <= kNoAllocWorkBytes, "cursor set does not fit its slice");
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

size_t noalloc_bytes() { return kNoAllocBytes; }

hipError_t noalloc_init(void** out) {                 // the device that is current; rt_create calls this once per slot
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    *out = p;
    return hipSuccess;
}

hipError_t launch_trace_noalloc(void* d_scratch, const DevScene& s, int mode, const rt_ray* d_rays, uint64_t n, rt_hit* d_hits,
                                hipStream_t st, uint32_t strip, int refill, int lds_cap, bool bin, const BinGrid& grid, uint32_t min_run,
                                bool queue) {
    if (n == 0) return hipSuccess;
    if (!d_scratch || n > kNoAllocMaxRays) return hipErrorInvalidValue;
    unsigned char* base = static_cast<unsigned char*>(d_scratch);
    StripScratch sc;
    sc.work = reinterpret_cast<unsigned long long*>(base);
    sc.spill = reinterpret_cast<int*>(base + kNoAllocWorkBytes);
    sc.max_blocks = kNoAllocMaxBlocks;
    sc.queue = queue;                      // the staged ray queue lives in LDS: nothing to allocate
    // the spill slice holds kNoAllocSpillEntries entries per thread: trees that are deeper keep more of the stack in LDS
    const int need = (int)s.stack_depth + 2;
    sc.lds_cap = std::max(lds_cap >= 2 ? lds_cap : need, need - (int)kNoAllocSpillEntries);
    const uint32_t* perm = nullptr;
    if (bin && mode != kBrute && strip >= 64) {
        uint32_t* bw = reinterpret_cast<uint32_t*>(base + kNoAllocWorkBytes) + kNoAllocSpillWords;
        hipError_t e = launch_bin(d_rays, n, grid, bw, min_run, st, kNoAllocBinBlocks);
        if (e != hipSuccess) return e;
        perm = bw;
    }
    return launch_trace(s, mode, false, false, d_rays, n, d_hits, nullptr, nullptr, st, strip, refill, perm,
                        perm ? bin_flag(perm, n, kNoAllocBinBlocks) : nullptr, sc);
}

hipError_t launch_primary(const DevScene& s, const DevCamera& cam, int mode, bool count, uint32_t w, uint32_t h,
                          uint32_t y0, uint32_t rows, uint32_t stripe_h, uint32_t stripe_step, uint32_t spp,
                          uint32_t seed, int jitter, rt_ray* d_rays_out, rt_hit* d_hits,
                          unsigned long long* d_counters, hipStream_t st, const PrimaryShade* shade) {
    if (rows == 0 || w == 0 || spp == 0) return hipSuccess;
    if (stripe_h == 0 || stripe_step == 0) return hipErrorInvalidValue;
    if (mode == kNone && (shade || count || !d_rays_out)) return hipErrorInvalidValue;      // kNone: ray generation only
    if (mode != kNone && !shade && !d_hits) return hipErrorInvalidValue;
    if (shade && (!shade->d_throughput || !shade->d_radiance)) return hipErrorInvalidValue;
    const int quad = (spp % 4 == 0) ? 2 : (spp % 2 == 0) ? 1 : 0;             // log2 of the samples one wave takes
    const uint32_t pw = quad == 2 ? 4 : 8, ph = quad == 0 ? 8 : 4;
    const uint32_t tiles_x = (w + pw - 1) / pw, tiles_y = (rows + ph - 1) / ph;
    const uint64_t n_waves64 = (uint64_t)tiles_x * tiles_y * (spp >> quad);
    if (n_waves64 > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t n_waves = (uint32_t)n_waves64;
    dim3 g((n_waves + kBlock / 64 - 1) / (kBlock / 64)), b(kBlock);
    float4* rays = reinterpret_cast<float4*>(d_rays_out);
    float4* hits = reinterpret_cast<float4*>(d_hits);
    ShadeOut so = { nullptr, nullptr, nullptr, 0u };
    if (shade) so = ShadeOut{ shade->d_throughput, shade->d_radiance, reinterpret_cast<float4*>(shade->d_next), shade->seed };
#define RT_LAUNCH(M, C, S) hipLaunchKernelGGL((k_primary<M, C, S>), g, b, stack_bytes(s, M), st, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, \
                                              jitter, tiles_x, n_waves, rays, hits, d_counters, so, quad)
#define RT_LAUNCH_M(M) do { if (shade) RT_LAUNCH(M, false, true); else if (count) RT_LAUNCH(M, true, false); else RT_LAUNCH(M, false, false); } while (0)
    // packet traversal: BVH4q trees with the wide node copy whose stack fits the 64 lanes of the stack register
    if (mode == kBvh4q && s.wide && s.stack_depth + 2 <= 64) {
        const size_t lds = 0;          // the shadow rays of the fused shading stage walk as packets too: no per-lane stacks
#define RT_LAUNCH_P(C, S) hipLaunchKernelGGL((k_primary_packet<C, S>), g, b, lds, st, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, \
                                              jitter, tiles_x, n_waves, rays, hits, d_counters, so, quad)
        if (shade) RT_LAUNCH_P(false, true); else if (count) RT_LAUNCH_P(true, false); else RT_LAUNCH_P(false, false);
#undef RT_LAUNCH_P
    } else
    if (mode == kNone) hipLaunchKernelGGL((k_primary<kNone, false, false>), g, b, 0, st, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, seed, jitter, tiles_x, n_waves, rays, nullptr, d_counters, so, quad);
    else if (mode == kBrute) RT_LAUNCH_M(kBrute); else if (mode == kBvh4q) RT_LAUNCH_M(kBvh4q); else RT_LAUNCH_M(kBvh2);
#undef RT_LAUNCH
#undef RT_LAUNCH_M
    return hipGetLastError();
}

hipError_t launch_shadow(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n, float lx,
                         float ly, float lz, rt_ray* d_shadow_rays, uint8_t* d_lit, hipStream_t st, bool packets) {
    if (n == 0) return hipSuccess;
    dim3 g(grid_for(n)), b(kBlock);
    const float4* hits = reinterpret_cast<const float4*>(d_hits);
    float4* sr = reinterpret_cast<float4*>(d_shadow_rays);
#define RT_LAUNCH_M(M) hipLaunchKernelGGL((k_shadow<M>), g, b, stack_bytes(s, M), st, s, d_rays, hits, n, lx, ly, lz, sr, d_lit)
    if (packets && mode == kBvh4q && s.wide && s.stack_depth + 2 <= 64)
        hipLaunchKernelGGL(k_shadow_packet, g, b, 0, st, s, d_rays, hits, n, lx, ly, lz, sr, d_lit);
    else
    if (mode == kBrute) RT_LAUNCH_M(kBrute); else if (mode == kBvh4q) RT_LAUNCH_M(kBvh4q); else RT_LAUNCH_M(kBvh2);
#undef RT_LAUNCH_M
    return hipGetLastError();
}

hipError_t launch_shade(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                        uint32_t key0, uint32_t row_len, uint32_t stripe_h, uint32_t stripe_step, uint32_t seed,
                        float4* d_throughput, float4* d_radiance, rt_ray* d_out, hipStream_t st, bool first) {
    if (n == 0) return hipSuccess;
    if (row_len && (stripe_h == 0 || stripe_step == 0)) return hipErrorInvalidValue;
    dim3 g(grid_for(n)), b(kBlock);
    const float4* hits = reinterpret_cast<const float4*>(d_hits);
    float4* out = reinterpret_cast<float4*>(d_out);
    // the traversal inside k_shade only runs when the scene has lights
    const int m = s.n_lights ? mode : kBrute;
#define RT_LAUNCH_M(M) do { if (first) hipLaunchKernelGGL((k_shade<M, true>), g, b, stack_bytes(s, M), st, s, d_rays, hits, n, key0, row_len, stripe_h, stripe_step, seed, d_throughput, d_radiance, out); \
                            else       hipLaunchKernelGGL((k_shade<M, false>), g, b, stack_bytes(s, M), st, s, d_rays, hits, n, key0, row_len, stripe_h, stripe_step, seed, d_throughput, d_radiance, out); } while (0)
    if (m == kBrute) RT_LAUNCH_M(kBrute); else if (m == kBvh4q) RT_LAUNCH_M(kBvh4q); else RT_LAUNCH_M(kBvh2);
#undef RT_LAUNCH_M
    return hipGetLastError();
}

hipError_t launch_shade_resolve(const DevScene& s, int mode, const rt_ray* d_rays, const rt_hit* d_hits, uint64_t n,
                                uint32_t spp, const float4* d_throughput, const float4* d_radiance, float4* d_linear,
                                uint32_t* d_rgba8, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (spp == 0 || spp > 64 || (spp & (spp - 1)) || n % spp) return hipErrorInvalidValue;
    dim3 g(grid_for(n)), b(kBlock);
    const float4* hits = reinterpret_cast<const float4*>(d_hits);
    const int m = s.n_lights ? mode : kBrute;
#define RT_LAUNCH_M(M) hipLaunchKernelGGL((k_shade_resolve<M>), g, b, stack_bytes(s, M), st, s, d_rays, hits, n, spp, d_throughput, d_radiance, d_linear, d_rgba8)
    if (m == kBrute) RT_LAUNCH_M(kBrute); else if (m == kBvh4q) RT_LAUNCH_M(kBvh4q); else RT_LAUNCH_M(kBvh2);
#undef RT_LAUNCH_M
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* d_radiance, uint64_t n_pixels, uint32_t spp, float4* d_linear, uint32_t* d_rgba8,
                          hipStream_t st) {
    if (n_pixels == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(n_pixels)), dim3(kBlock), 0, st, d_radiance, n_pixels, spp, d_linear, d_rgba8);
    return hipGetLastError();
}

hipError_t launch_fill(float4* d, uint64_t n, float x, float y, float z, float w, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(kBlock), 0, st, d, n, make_float4(x, y, z, w));
    return hipGetLastError();
}

hipError_t launch_path_brute(const DevScene& s, const DevCamera& cam, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows, uint32_t stripe_h,
                             uint32_t stripe_step, uint32_t spp, uint32_t bounces, uint32_t seed, float4* d_radiance, hipStream_t st) {
    if (rows == 0 || w == 0 || spp == 0) return hipSuccess;
    if (stripe_h == 0 || stripe_step == 0 || bounces == 0 || !d_radiance) return hipErrorInvalidValue;
    const uint64_t n = (uint64_t)rows * w * spp;
    if (n > 0xFFFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_path_brute, dim3(grid_for(n)), dim3(kBlock), 0, st, s, cam, w, h, y0, rows, stripe_h, stripe_step, spp, bounces, seed, d_radiance);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const uint32_t* stage, uint32_t* frame, uint32_t h, uint32_t row_words, uint32_t stripe_h,
                              uint32_t n_shares, uint32_t share_rows, hipStream_t st) {
    if (h == 0 || row_words == 0) return hipSuccess;
    if (!stage || !frame || stripe_h == 0 || n_shares == 0 || h > 65535u) return hipErrorInvalidValue;
    if ((uint64_t)((h - 1) / stripe_h / n_shares + 1) * stripe_h > share_rows) return hipErrorInvalidValue;   // last row must lie inside its share
    const uint32_t bx = std::min<uint32_t>(8u, (row_words + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_gather_rows, dim3(bx, h), dim3(kBlock), 0, st, stage, frame, row_words, stripe_h, n_shares, share_rows);
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
