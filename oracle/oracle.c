/*
 * oracle.c -- plain-C CPU restatement of the reference hot path.  TEST INFRASTRUCTURE: see oracle.h.
 *
 * Canonical arithmetic (fp32, compile with -ffp-contract=off; the HIP kernels implement the same
 * order independently in raytracer_amd/csrc/kernels.hip):
 *   dot(a,b)    = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   cross(a,b)  = ( fma(a.y,b.z, -(a.z*b.y)), fma(a.z,b.x, -(a.x*b.z)), fma(a.x,b.y, -(a.y*b.x)) )
 *   normalize v = v / sqrt(dot(v,v))            (three IEEE divisions)
 *   o + t*d     = fma(t, d, o)
 * GLSL (no `precise`) permits this contraction, so it is a valid evaluation of shaders/frag.glsl.
 * Division and sqrt are IEEE correctly rounded on both sides (hipcc default for HIP).
 *
 * All citations are relative to /root/reference.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld3(const void* base, size_t off) {
    const float* p = (const float*)((const char*)base + off);
    return V(p[0], p[1], p[2]);
}
static inline float ldf(const void* base, size_t off) { return *(const float*)((const char*)base + off); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 normalize(v3 a) { float l = sqrtf(dot(a, a)); return V(a.x / l, a.y / l, a.z / l); }
static inline v3 madd(float t, v3 d, v3 o) { return V(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z)); }

/* Byte offsets inside the reference's host structs (src/math/Vertex.h:38-41, Material.h:6-11,
 * Triangle.h:10-15, Sphere.h:10-15, Light.h:6-10; sizes probed in SURVEY.md section 2). */
enum { TRI_P0 = 0, TRI_N0 = 16, TRI_P1 = 32, TRI_N1 = 48, TRI_P2 = 64, TRI_N2 = 80, TRI_MAT = 96,
       SPH_C = 0, SPH_R = 12, SPH_MAT = 16, LIGHT_P = 0,
       /* Camera::UniformBufferObject, src/application/Camera.h:6-18 */
       CAM_POS = 0, CAM_FRONT = 32, CAM_UP = 48, CAM_RIGHT = 64, CAM_FOV = 92, CAM_ASPECT = 96 };

/* ---- seeded integer hash: substitute for frag.glsl:13-15 (sin-based rand is parity unpinned) ---- */
static inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
float orc_rand(uint32_t index, uint32_t dim, uint32_t seed) {
    uint32_t k = hash32(seed * 0x9E3779B9u + dim + 1u);
    uint32_t h = hash32(hash32(index ^ k) + dim);
    return (float)(h >> 8) * 0x1.0p-24f;
}

/* ---- frag.glsl:130-156 ---- */
static inline int ray_tri(v3 o, v3 d, v3 p0, v3 p1, v3 p2, float* t, float* u, float* v) {
    const float EPSILON = 1e-6f;
    v3 e1 = sub(p1, p0);                       /* :132 */
    v3 e2 = sub(p2, p0);                       /* :133 */
    v3 h = cross(d, e2);                       /* :134 */
    float a = dot(e1, h);                      /* :135 */
    if (fabsf(a) < EPSILON) return 0;          /* :136 */
    float f = 1.0f / a;                        /* :139 */
    v3 s = sub(o, p0);                         /* :140 */
    float uu = f * dot(s, h);                  /* :141 */
    if (uu < 0.0f || uu > 1.0f) return 0;      /* :142 */
    v3 q = cross(s, e1);                       /* :145 */
    float vv = f * dot(d, q);                  /* :146 */
    if (vv < 0.0f || uu + vv > 1.0f) return 0; /* :147 */
    float tt = f * dot(e2, q);                 /* :150 */
    *t = tt; *u = uu; *v = vv;
    return tt > EPSILON;                       /* :151 */
}

/* ---- frag.glsl:158-183 ---- */
static inline int ray_sphere(v3 o, v3 d, v3 c, float r, float* t) {
    v3 oc = sub(o, c);                                     /* :159 */
    float a = dot(d, d);                                   /* :160 */
    float b = 2.0f * dot(oc, d);                           /* :161 */
    float cc = fmaf(-r, r, dot(oc, oc));                   /* :162 */
    float disc = fmaf(b, b, -((4.0f * a) * cc));           /* :164 */
    if (disc < 0.0f) return 0;                             /* :165 */
    float sq = sqrtf(disc);                                /* :169 */
    float t0 = (-b - sq) / (2.0f * a);                     /* :170 */
    float t1 = (-b + sq) / (2.0f * a);                     /* :171 */
    if (t0 > 0.0f) { *t = t0; return 1; }                  /* :174 */
    if (t1 > 0.0f) { *t = t1; return 1; }                  /* :178 */
    return 0;
}

int orc_ray_triangle(const orc_ray* r, const float v0[3], const float v1[3], const float v2[3],
                     float* t, float* u, float* v) {
    return ray_tri(V(r->ox, r->oy, r->oz), V(r->dx, r->dy, r->dz),
                   V(v0[0], v0[1], v0[2]), V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2]), t, u, v);
}
int orc_ray_sphere(const orc_ray* r, const float c[3], float radius, float* t) {
    return ray_sphere(V(r->ox, r->oy, r->oz), V(r->dx, r->dy, r->dz), V(c[0], c[1], c[2]), radius, t);
}

/* ---- frag.glsl:265-301: spheres first, then triangles, index order, strict '<' ----
 * Build extension (the reference ray has no interval): closestT starts at ray.tmax (reference: 1e20)
 * and a candidate must also satisfy t >= ray.tmin (reference: tmin = 0, which changes nothing). */
void orc_trace_one(const orc_scene* s, const orc_ray* r, orc_hit* out) {
    v3 o = V(r->ox, r->oy, r->oz), d = V(r->dx, r->dy, r->dz);
    float closest = r->tmax;                                         /* :266 */
    orc_hit best = { r->tmax, 0.0f, 0.0f, ORC_MISS };
    for (uint32_t i = 0; i < s->n_spheres; ++i) {                    /* :269 */
        const char* sp = (const char*)s->spheres + (size_t)i * s->sphere_stride;
        float t;
        if (ray_sphere(o, d, ld3(sp, SPH_C), ldf(sp, SPH_R), &t)) {  /* :271 */
            if (t < closest && t >= r->tmin) {                       /* :272 */
                closest = t;
                best.t = t; best.u = 0.0f; best.v = 0.0f; best.prim = ORC_SPHERE_BIT | i;
            }
        }
    }
    for (uint32_t i = 0; i < s->n_tris; ++i) {                       /* :282 */
        const char* tp = (const char*)s->tris + (size_t)i * s->tri_stride;
        float t, u, v;
        if (ray_tri(o, d, ld3(tp, TRI_P0), ld3(tp, TRI_P1), ld3(tp, TRI_P2), &t, &u, &v)) { /* :284 */
            if (t < closest && t >= r->tmin) {                       /* :285 */
                closest = t;
                best.t = t; best.u = u; best.v = v; best.prim = i;
            }
        }
    }
    *out = best;
}

void orc_trace(const orc_scene* s, const orc_ray* rays, uint64_t n, orc_hit* out) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) orc_trace_one(s, &rays[i], &out[i]);
}

/* ---- "build CPU BVH, not reference" (second CPU row) --------------------------------------
 * The same closest-hit query answered by walking a BVH2 on the CPU: the FAIR CPU figure beside the brute-force one.
 * `nodes` is the flattened BVH2 the product's host builder emits (rt_bvh_build_host, layout BVH2: 16 words per node,
 * (c0.lo.x,c0.hi.x,c0.lo.y,c0.hi.y) (c1 ...) (c0.lo.z,c0.hi.z,c1.lo.z,c1.hi.z) (ref0,ref1,-,-); ref >= 0 interior,
 * ~ref = first<<3 | count-1 a leaf over tri_order[first ..]); it is input data here, the oracle links nothing of the
 * product.  Triangles are tested with the oracle's own ray_tri on the caller's records; candidates are merged with
 * the order-independent form of the reference's rule (smaller t, then spheres before triangles, then lower index),
 * so the result equals orc_trace whenever no accepted triangle's leaf is culled (the boxes are padded for that). */
static void trace_one_bvh2(const orc_scene* s, const uint32_t* nodes, uint32_t n_nodes, const uint32_t* tri_order,
                           const orc_ray* r, orc_hit* out) {
    v3 o = V(r->ox, r->oy, r->oz), d = V(r->dx, r->dy, r->dz);
    orc_hit best = { r->tmax, 0.0f, 0.0f, ORC_MISS };
    if (!(r->tmax > r->tmin)) { *out = best; return; }
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const char* sp = (const char*)s->spheres + (size_t)i * s->sphere_stride;
        float t;
        if (ray_sphere(o, d, ld3(sp, SPH_C), ldf(sp, SPH_R), &t) && t < best.t && t >= r->tmin) {
            best.t = t; best.u = 0.0f; best.v = 0.0f; best.prim = ORC_SPHERE_BIT | i;
        }
    }
    if (n_nodes == 0) { *out = best; return; }
    const float idir[3] = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };     /* +-inf for axis-parallel rays: slabs still work */
    const float org[3] = { o.x, o.y, o.z };
    const float tmin0 = r->tmin > 0.0f ? r->tmin : 0.0f;
    int32_t stack[64];
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        if (node >= 0) {
            const float* q = (const float*)(nodes + (size_t)node * 16);
            const int32_t* refs = (const int32_t*)(nodes + (size_t)node * 16 + 12);
            float tn[2], tf[2];
            for (int c = 0; c < 2; ++c) {
                const float lo[3] = { q[4 * c], q[4 * c + 2], q[8 + 2 * c] }, hi[3] = { q[4 * c + 1], q[4 * c + 3], q[9 + 2 * c] };
                float a = tmin0, b = best.t;
                for (int k = 0; k < 3; ++k) {
                    float t0 = (lo[k] - org[k]) * idir[k], t1 = (hi[k] - org[k]) * idir[k];
                    if (t0 != t0 || t1 != t1) continue;               /* 0 * inf: the ray lies in the slab plane */
                    if (t0 > t1) { float x = t0; t0 = t1; t1 = x; }
                    if (t0 > a) a = t0;
                    if (t1 < b) b = t1;
                }
                tn[c] = a; tf[c] = b;
            }
            const int h0 = tn[0] <= tf[0], h1 = tn[1] <= tf[1];
            if (h0 && h1) {
                const int near = tn[1] < tn[0];
                stack[sp++] = refs[!near];
                node = refs[near];
            } else if (h0 || h1) {
                node = refs[h1];
            } else {
                if (sp == 0) break;
                node = stack[--sp];
            }
        } else {
            const uint32_t ref = ~(uint32_t)node, first = ref >> 3, cnt = (ref & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t id = tri_order[first + k];
                const char* tp = (const char*)s->tris + (size_t)id * s->tri_stride;
                float t, u, v;
                if (ray_tri(o, d, ld3(tp, TRI_P0), ld3(tp, TRI_P1), ld3(tp, TRI_P2), &t, &u, &v) && t >= r->tmin &&
                    (t < best.t || (t == best.t && best.prim != ORC_MISS && !(best.prim & ORC_SPHERE_BIT) && id < best.prim))) {
                    best.t = t; best.u = u; best.v = v; best.prim = id;
                }
            }
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    *out = best;
}

void orc_trace_bvh2(const orc_scene* s, const uint32_t* nodes, uint32_t n_nodes, const uint32_t* tri_order,
                    const orc_ray* rays, uint64_t n, orc_hit* out) {
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; ++i) trace_one_bvh2(s, nodes, n_nodes, tri_order, &rays[i], &out[i]);
}

/* any-hit: 1 iff some primitive is accepted in [tmin, tmax) -- by construction identical to
 * "closest-hit found something". */
void orc_occluded(const orc_scene* s, const orc_ray* rays, uint64_t n, uint8_t* out) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        orc_hit h;
        orc_trace_one(s, &rays[i], &h);
        out[i] = (h.prim != ORC_MISS);
    }
}

/* ---- vert.glsl:9-12 + VkRenderer.h:148-155: fragment centre -> uv, uv(0,0) = top-left ---- */
void orc_pixel_uv(uint32_t x, uint32_t y, uint32_t w, uint32_t h, float uv[2]) {
    uv[0] = ((float)x + 0.5f) / (float)w;
    uv[1] = ((float)y + 0.5f) / (float)h;
}

/* ---- frag.glsl:89-128 ---- */
void orc_camera_ray(const void* cam, uint32_t w, uint32_t h, uint32_t x, uint32_t y,
                    uint32_t ray_index, uint32_t seed, int jitter, orc_ray* out) {
    float uv[2];
    orc_pixel_uv(x, y, w, h, uv);
    float ndcx = fmaf(uv[0], 2.0f, -1.0f);                 /* :91 (x2 is exact, so == uv*2-1) */
    float ndcy = fmaf(uv[1], 2.0f, -1.0f);
    if (jitter) {                                          /* :94-103, rand replaced by orc_rand */
        const float psx = 2.0f / 1920.0f, psy = 2.0f / 1080.0f;   /* hard-coded in the reference */
        /* key = index of this ray in the [pixel][sample] buffer */
        ndcx += (orc_rand(ray_index, 0u, seed) - 0.5f) * psx;
        ndcy += (orc_rand(ray_index, 1u, seed) - 0.5f) * psy;
    }
    float fov = ldf(cam, CAM_FOV) * 0.017453292519943295f; /* :106 radians() */
    float halfH = tanf(fov / 2.0f);                        /* :109 */
    float halfW = halfH * ldf(cam, CAM_ASPECT);            /* :110 */
    v3 right = ld3(cam, CAM_RIGHT), up = ld3(cam, CAM_UP), front = ld3(cam, CAM_FRONT);
    float a = ndcx * halfW, b = ndcy * halfH;              /* :114-115 */
    v3 dir = V(fmaf(b, up.x, a * right.x) + front.x,
               fmaf(b, up.y, a * right.y) + front.y,
               fmaf(b, up.z, a * right.z) + front.z);      /* :113-117 */
    dir = normalize(dir);                                  /* :113 */
    dir = normalize(dir);                                  /* :120 */
    v3 pos = ld3(cam, CAM_POS);
    out->ox = pos.x; out->oy = pos.y; out->oz = pos.z; out->tmin = 0.0f;
    out->dx = dir.x; out->dy = dir.y; out->dz = dir.z; out->tmax = 1e20f;   /* :266 */
}

void orc_primary_rays(const void* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t seed,
                      int jitter, orc_ray* out) {
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)w * h; ++p) {
        uint32_t x = (uint32_t)(p % w), y = (uint32_t)(p / w);
        for (uint32_t s = 0; s < spp; ++s) {
            uint32_t ray_index = (uint32_t)(p * spp + s);
            orc_camera_ray(cam, w, h, x, y, ray_index, seed, jitter, &out[(size_t)p * spp + s]);
        }
    }
}

/* ---- hit geometry: frag.glsl:275-276 (sphere), :288-293 (triangle), :315 (second normalize) ---- */
int orc_hit_point(const orc_scene* s, const orc_ray* r, const orc_hit* h, float p[3], float n[3]) {
    if (h->prim == ORC_MISS) return 0;
    v3 o = V(r->ox, r->oy, r->oz), d = V(r->dx, r->dy, r->dz);
    v3 pos = madd(h->t, d, o);
    v3 nn;
    if (h->prim & ORC_SPHERE_BIT) {
        const char* sp = (const char*)s->spheres + (size_t)(h->prim & ~ORC_SPHERE_BIT) * s->sphere_stride;
        nn = normalize(sub(pos, ld3(sp, SPH_C)));
    } else {
        const char* tp = (const char*)s->tris + (size_t)h->prim * s->tri_stride;
        v3 n0 = ld3(tp, TRI_N0), n1 = ld3(tp, TRI_N1), n2 = ld3(tp, TRI_N2);
        float w0 = (1.0f - h->u) - h->v;
        nn = V(fmaf(h->v, n2.x, fmaf(h->u, n1.x, w0 * n0.x)),
               fmaf(h->v, n2.y, fmaf(h->u, n1.y, w0 * n0.y)),
               fmaf(h->v, n2.z, fmaf(h->u, n1.z, w0 * n0.z)));
        nn = normalize(nn);
    }
    nn = normalize(nn);                                   /* :315 */
    p[0] = pos.x; p[1] = pos.y; p[2] = pos.z;
    n[0] = nn.x; n[1] = nn.y; n[2] = nn.z;
    return 1;
}

/* ---- frag.glsl:321-330, one light ---- */
void orc_shadow_lit(const orc_scene* s, const orc_ray* rays, const orc_hit* hits, uint64_t n,
                    const float lp[3], orc_ray* out_sr, uint8_t* out_lit) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float p[3], nn[3];
        orc_ray sr; memset(&sr, 0, sizeof sr);
        if (!orc_hit_point(s, &rays[i], &hits[i], p, nn)) {
            out_lit[i] = 2;
            if (out_sr) out_sr[i] = sr;
            continue;
        }
        v3 P = V(p[0], p[1], p[2]), N = V(nn[0], nn[1], nn[2]);
        v3 toL = sub(V(lp[0], lp[1], lp[2]), P);
        v3 L = normalize(toL);                            /* :321 */
        float distance = sqrtf(dot(toL, toL));            /* :322 */
        v3 so = madd(0.001f, N, P);                       /* :326  p + N*0.001 */
        sr.ox = so.x; sr.oy = so.y; sr.oz = so.z; sr.tmin = 0.0f;
        sr.dx = L.x; sr.dy = L.y; sr.dz = L.z; sr.tmax = 1e20f;
        orc_hit sh;
        orc_trace_one(s, &sr, &sh);                       /* :330 closest hit */
        int lit = 1;
        if (sh.prim != ORC_MISS) {
            v3 hp = madd(sh.t, L, so);                    /* shadowHit.position :275/288 */
            v3 dlt = sub(hp, P);
            lit = sqrtf(dot(dlt, dlt)) > distance;        /* :330 */
        }
        out_lit[i] = (uint8_t)lit;
        if (out_sr) out_sr[i] = sr;
    }
}

/* sin/cos of 2*pi*x for x in [0,1): exact quadrant reduction, then fixed fp32 polynomials on
 * [-pi/4, pi/4] evaluated with fmaf (same coefficients in the HIP kernel => bit-identical). */
static inline void sincos_2pi(float x, float* s, float* c) {
    float k = floorf(fmaf(x, 4.0f, 0.5f));           /* nearest quarter turn, 0..4 */
    float r = fmaf(k, -0.25f, x);                    /* exact: x multiple of 2^-24 */
    float a = r * 6.283185307179586f;                /* [-pi/4, pi/4] */
    float a2 = a * a;
    float sp = fmaf(a2, fmaf(a2, fmaf(a2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.0f) * a;
    float cp = fmaf(a2, fmaf(a2, fmaf(a2, fmaf(a2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                       4.166664568298827e-2f), -0.5f), 1.0f);
    int q = (int)k & 3;
    float ss = (q & 1) ? cp : sp, cs = (q & 1) ? sp : cp;
    if (q == 1) { cs = -cs; }
    else if (q == 2) { ss = -ss; cs = -cs; }
    else if (q == 3) { ss = -ss; }
    *s = ss; *c = cs;
}

/* frag.glsl:185-203 with xi supplied.  theta = acos(sqrt(1-xi1)) is folded analytically:
 * cos(theta) = sqrt(1-xi1), sin(theta) = sqrt(xi1) (GLSL leaves acos/sin/cos precision
 * implementation-defined; orc_hemisphere_libm is the literal form, compared in tests to 2e-6). */
static inline v3 hemisphere(v3 N, float xi1, float xi2) {
    float ct = sqrtf(1.0f - xi1), st = sqrtf(xi1);
    float sphi, cphi;
    sincos_2pi(xi2, &sphi, &cphi);
    float xs = st * cphi, ys = ct, zs = st * sphi;                       /* :193-195 */
    v3 up = fabsf(N.z) < 0.999f ? V(0.0f, 0.0f, 1.0f) : V(1.0f, 0.0f, 0.0f);   /* :197 */
    v3 tX = normalize(cross(up, N));                                     /* :198 */
    v3 tY = cross(N, tX);                                                /* :199 */
    v3 dir = V(fmaf(N.x, ys, fmaf(tY.x, zs, tX.x * xs)),
               fmaf(N.y, ys, fmaf(tY.y, zs, tX.y * xs)),
               fmaf(N.z, ys, fmaf(tY.z, zs, tX.z * xs)));                /* :201 */
    return normalize(dir);                                               /* :202 */
}
void orc_hemisphere(const float n[3], float xi1, float xi2, float dir[3]) {
    v3 r = hemisphere(V(n[0], n[1], n[2]), xi1, xi2);
    dir[0] = r.x; dir[1] = r.y; dir[2] = r.z;
}
void orc_hemisphere_libm(const float n[3], float xi1, float xi2, float dir[3]) {
    v3 N = V(n[0], n[1], n[2]);
    float theta = acosf(sqrtf(1.0f - xi1));                              /* :190 */
    float phi = 2.0f * 3.14159265358979323846f * xi2;                    /* :191 */
    float xs = sinf(theta) * cosf(phi), ys = cosf(theta), zs = sinf(theta) * sinf(phi);
    v3 up = fabsf(N.z) < 0.999f ? V(0.0f, 0.0f, 1.0f) : V(1.0f, 0.0f, 0.0f);
    v3 tX = normalize(cross(up, N));
    v3 tY = cross(N, tX);
    v3 d = V(tX.x * xs + tY.x * zs + N.x * ys, tX.y * xs + tY.y * zs + N.y * ys,
             tX.z * xs + tY.z * zs + N.z * ys);
    d = normalize(d);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* ---- frag.glsl:340-343 ---- */
void orc_bounce(const orc_scene* s, const orc_ray* rays, const orc_hit* hits, uint64_t n,
                uint32_t seed, orc_ray* out) {
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float p[3], nn[3];
        orc_ray br; memset(&br, 0, sizeof br);
        if (orc_hit_point(s, &rays[i], &hits[i], p, nn)) {
            v3 P = V(p[0], p[1], p[2]), N = V(nn[0], nn[1], nn[2]);
            float xi1 = orc_rand((uint32_t)i, 2u, seed), xi2 = orc_rand((uint32_t)i, 3u, seed);
            v3 dir = hemisphere(N, xi1, xi2);                            /* :340 */
            v3 o = madd(0.001f, N, P);                                   /* :342 */
            br.ox = o.x; br.oy = o.y; br.oz = o.z; br.tmin = 0.0f;
            br.dx = dir.x; br.dy = dir.y; br.dz = dir.z; br.tmax = 1e20f;
        }
        out[i] = br;
    }
}

/* ---- frag.glsl:205-263 Cook-Torrance ------------------------------------------------------------
 * pow(x, 5.0) is evaluated as ((x*x)*(x*x))*x (GLSL leaves pow precision open); PI is the fp32 value
 * of the shader's literal (:5); mix(a, b, t) = a*(1-t) + b*t as GLSL defines it. */
#define ORC_PI 3.14159274f
static inline float max0(float x) { return x > 0.0f ? x : 0.0f; }          /* max(x, 0.0) */
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
static inline float geometry_schlick_ggx(float ndv, float roughness) {     /* :221-228 */
    float r = roughness + 1.0f;
    float k = (r * r) / 8.0f;
    float denom = fmaf(ndv, 1.0f - k, k);
    return ndv / denom;
}
static v3 brdf(const float* m, v3 N, v3 Vv, v3 L) {
    v3 albedo = V(m[0], m[1], m[2]);
    float roughness = m[8], metallic = m[9];
    v3 H = normalize(V(Vv.x + L.x, Vv.y + L.y, Vv.z + L.z));               /* :240 */
    float NdotL = max0(dot(N, L)), NdotV = max0(dot(N, Vv));               /* :242-243 */
    float NdotH = max0(dot(N, H)), VdotH = max0(dot(Vv, H));               /* :244-245 */
    float om = 1.0f - metallic;
    v3 F0 = V(fmaf(albedo.x, metallic, 0.04f * om), fmaf(albedo.y, metallic, 0.04f * om),
              fmaf(albedo.z, metallic, 0.04f * om));                       /* :247 mix */
    float p5 = pow5(1.0f - VdotH);                                         /* :206 */
    v3 F = V(fmaf(1.0f - F0.x, p5, F0.x), fmaf(1.0f - F0.y, p5, F0.y), fmaf(1.0f - F0.z, p5, F0.z));
    float a = roughness * roughness, a2 = a * a;                           /* :210-211 */
    float dd = fmaf(NdotH * NdotH, a2 - 1.0f, 1.0f);                       /* :215 */
    float D = a2 / ((ORC_PI * dd) * dd);                                   /* :216-218 */
    float G = geometry_schlick_ggx(NdotL, roughness) * geometry_schlick_ggx(NdotV, roughness);  /* :230-237 */
    float denominator = fmaf(4.0f * NdotV, NdotL, 0.001f);                 /* :254 */
    float kd = 1.0f - metallic;                                            /* :258 */
    v3 out;
    {   float num = (D * F.x) * G; float spec = num / denominator;         /* :253-255 */
        float kD = (1.0f - F.x) * kd; out.x = (kD * albedo.x) / ORC_PI + spec; }   /* :257-262 */
    {   float num = (D * F.y) * G; float spec = num / denominator;
        float kD = (1.0f - F.y) * kd; out.y = (kD * albedo.y) / ORC_PI + spec; }
    {   float num = (D * F.z) * G; float spec = num / denominator;
        float kD = (1.0f - F.z) * kd; out.z = (kD * albedo.z) / ORC_PI + spec; }
    return out;
}
void orc_brdf(const float m[12], const float n[3], const float v[3], const float l[3], float out[3]) {
    v3 r = brdf(m, V(n[0], n[1], n[2]), V(v[0], v[1], v[2]), V(l[0], l[1], l[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static const float* hit_material(const orc_scene* s, uint32_t prim) {
    if (prim & ORC_SPHERE_BIT)
        return (const float*)((const char*)s->spheres + (size_t)(prim & ~ORC_SPHERE_BIT) * s->sphere_stride + SPH_MAT);
    return (const float*)((const char*)s->tris + (size_t)prim * s->tri_stride + TRI_MAT);
}

/* frag.glsl:310-354, one bounce for a batch */
void orc_shade(const orc_scene* s, const orc_ray* rays, const orc_hit* hits, uint64_t n, uint32_t key0,
               uint32_t seed, float* thr, float* rad, orc_ray* out) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float p[3], nn[3];
        orc_ray br; memset(&br, 0, sizeof br);
        if (!(rays[i].tmax > rays[i].tmin) || !orc_hit_point(s, &rays[i], &hits[i], p, nn)) { out[i] = br; continue; }   /* :351-353 */
        const float* m = hit_material(s, hits[i].prim);
        v3 T = V(thr[4 * i], thr[4 * i + 1], thr[4 * i + 2]);
        v3 C = V(rad[4 * i], rad[4 * i + 1], rad[4 * i + 2]);
        float es = m[7];
        C = V(fmaf(T.x * m[4], es, C.x), fmaf(T.y * m[5], es, C.y), fmaf(T.z * m[6], es, C.z));      /* :313 */
        v3 P = V(p[0], p[1], p[2]), N = V(nn[0], nn[1], nn[2]);                                      /* :315 */
        v3 Vv = normalize(V(-rays[i].dx, -rays[i].dy, -rays[i].dz));                                 /* :316 */
        for (uint32_t li = 0; li < s->n_lights; ++li) {                                              /* :319 */
            const char* lp = (const char*)s->lights + (size_t)li * s->light_stride;
            v3 lpos = ld3(lp, 0), lcol = ld3(lp, 16);
            float inten = ldf(lp, 28);
            v3 toL = sub(lpos, P);
            v3 L = normalize(toL);                                                                   /* :321 */
            float distance = sqrtf(dot(toL, toL));                                                   /* :322 */
            float attenuation = 1.0f / (distance * distance);                                        /* :323 */
            v3 so = madd(0.001f, N, P);                                                              /* :326 */
            orc_ray sr = { so.x, so.y, so.z, 0.0f, L.x, L.y, L.z, 1e20f };
            orc_hit sh;
            orc_trace_one(s, &sr, &sh);
            int lit = 1;
            if (sh.prim != ORC_MISS) {
                v3 dl = sub(madd(sh.t, L, so), P);
                lit = sqrtf(dot(dl, dl)) > distance;                                                 /* :330 */
            }
            if (lit) {
                v3 B = brdf(m, N, Vv, L);                                                            /* :331 */
                float NdotL = max0(dot(N, L));                                                       /* :333 */
                float ia = inten * attenuation;
                v3 radiance = V(lcol.x * ia, lcol.y * ia, lcol.z * ia);                              /* :335 */
                C = V(fmaf((T.x * B.x) * radiance.x, NdotL, C.x), fmaf((T.y * B.y) * radiance.y, NdotL, C.y),
                      fmaf((T.z * B.z) * radiance.z, NdotL, C.z));                                   /* :336 */
            }
        }
        uint32_t key = key0 + (uint32_t)i;
        float xi1 = orc_rand(key, 2u, seed), xi2 = orc_rand(key, 3u, seed);
        v3 dir = hemisphere(N, xi1, xi2);                                                            /* :340 */
        v3 o = madd(0.001f, N, P);                                                                   /* :342 */
        br.ox = o.x; br.oy = o.y; br.oz = o.z; br.tmin = 0.0f;
        br.dx = dir.x; br.dy = dir.y; br.dz = dir.z; br.tmax = 1e20f;
        float NdotR = max0(dot(N, dir));                                                             /* :345 */
        float pdf = NdotR / ORC_PI;                                                                  /* :346 */
        v3 B = brdf(m, N, Vv, dir);                                                                  /* :348 */
        T = V(T.x * ((B.x * NdotR) / pdf), T.y * ((B.y * NdotR) / pdf), T.z * ((B.z * NdotR) / pdf));   /* :350 */
        thr[4 * i] = T.x; thr[4 * i + 1] = T.y; thr[4 * i + 2] = T.z;
        rad[4 * i] = C.x; rad[4 * i + 1] = C.y; rad[4 * i + 2] = C.z;
        out[i] = br;
    }
}

/* frag.glsl:357-359 */
void orc_resolve(const float* rad, uint64_t n_pixels, uint32_t spp, float* lin, uint8_t* rgba) {
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)n_pixels; ++p) {
        float c[3] = { 0.0f, 0.0f, 0.0f };
        for (uint32_t k = 0; k < spp; ++k)
            for (int a = 0; a < 3; ++a) c[a] += rad[4 * ((size_t)p * spp + k) + a];      /* color += ... over samples */
        for (int a = 0; a < 3; ++a) c[a] = c[a] / (float)spp;                             /* :357 */
        if (lin) { lin[4 * p] = c[0]; lin[4 * p + 1] = c[1]; lin[4 * p + 2] = c[2]; lin[4 * p + 3] = 1.0f; }
        if (rgba) {
            for (int a = 0; a < 3; ++a) {
                float g = powf(c[a], 1.0f / 2.6f);                                        /* :358 */
                g = g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g);
                if (!(g == g)) g = 0.0f;
                rgba[4 * p + a] = (uint8_t)(g * 255.0f + 0.5f);
            }
            rgba[4 * p + 3] = 255;                                                        /* :359 alpha 1.0 */
        }
    }
}

void orc_render(const orc_scene* s, const void* cam, uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces,
                uint32_t seed, float* lin, uint8_t* rgba) {
    size_t n = (size_t)w * h * spp;
    orc_ray* rays = (orc_ray*)malloc(n * sizeof(orc_ray));
    orc_ray* next = (orc_ray*)malloc(n * sizeof(orc_ray));
    orc_hit* hits = (orc_hit*)malloc(n * sizeof(orc_hit));
    float* thr = (float*)malloc(n * 16);
    float* rad = (float*)calloc(n, 16);
    for (size_t i = 0; i < n; ++i) { thr[4 * i] = thr[4 * i + 1] = thr[4 * i + 2] = 1.0f; thr[4 * i + 3] = 0.0f; }   /* :308 */
    orc_primary_rays(cam, w, h, spp, seed, 1, rays);                                      /* :307 */
    for (uint32_t b = 0; b < bounces; ++b) {                                              /* :310 */
        orc_trace(s, rays, n, hits);                                                      /* :312 */
        orc_shade(s, rays, hits, n, 0u, seed + 1u + b, thr, rad, next);
        orc_ray* t = rays; rays = next; next = t;
    }
    orc_resolve(rad, (uint64_t)w * h, spp, lin, rgba);
    free(rays); free(next); free(hits); free(thr); free(rad);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
