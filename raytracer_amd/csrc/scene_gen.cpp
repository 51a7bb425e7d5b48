// scene_gen.cpp -- host scene generators behind the C ABI (rt_gen_*).
//
// The reference compiles its one scene into VkRenderer::createData (src/vulkan/VkRenderer.cpp:164-258)
// and owns one tessellator, Sphere::sphereGeometry (src/math/Sphere.h:17-79).  The Cornell data below
// is that scene value for value; the tessellator follows the same fp32 arithmetic; the "stand-in" and
// "soup" scenes are the synthetic inputs of SURVEY.md 8(d) (real Bunny/Sponza meshes are not in the
// reference and cannot be fetched).
#include "../../include/raytracer_amd.h"
#include "../host/rt_camera.hpp"
#include "../host/rt_types.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

using namespace rt;

namespace {

Material make_mat(float ar, float ag, float ab, float er, float eg, float eb, float es, float rough, float metal) {
    Material m;
    std::memset(&m, 0, sizeof m);
    m.m_albedo = { ar, ag, ab };
    m.m_emission = { er, eg, eb };
    m.m_emissionStrength = es; m.m_roughness = rough; m.m_metallic = metal;
    return m;
}

Triangle make_tri(const Vec3& p0, const Vec3& p1, const Vec3& p2, const Vec3& n0, const Vec3& n1, const Vec3& n2,
                  const Material& m) {
    Triangle t;
    std::memset(&t, 0, sizeof t);
    t.m_v0.m_position = p0; t.m_v0.m_normal = n0;
    t.m_v1.m_position = p1; t.m_v1.m_normal = n1;
    t.m_v2.m_position = p2; t.m_v2.m_normal = n2;
    t.m_material = m;
    return t;
}

// glm::normalize = v * inversesqrt(dot(v,v)); input-data generation only (parity unpinned).
Vec3 normalized(const Vec3& v) {
    float inv = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return { v.x * inv, v.y * inv, v.z * inv };
}

// VkRenderer.cpp:166-169 materials, :171-238 triangles.  Row: 3 positions, shared normal, material id.
const Material kWhite = make_mat(1, 1, 1, 0, 0, 0, 0.0f, 0.0f, 0.0f);
const Material kLeft  = make_mat(1, 0, 0, 0, 0, 0, 0.0f, 0.0f, 0.0f);
const Material kRight = make_mat(0, 1, 0, 0, 0, 0, 0.0f, 0.0f, 0.0f);
const Material kEmit  = make_mat(1, 1, 1, 1, 1, 1, 0.5f, 0.0f, 0.0f);

struct CornellRow { float p[9]; float n[3]; int mat; };
const CornellRow kCornell[12] = {
    { { -2,  2, 0,      2,  2, 0,     -2, -2, 0    }, {  0, 0,  1 }, 0 },   // ground
    { { -2, -2, 0,      2,  2, 0,      2, -2, 0    }, {  0, 0,  1 }, 0 },
    { {  2, -2, 0,      2, -2, 2,     -2, -2, 0    }, {  0, 1,  0 }, 0 },   // back wall
    { { -2, -2, 0,      2, -2, 2,     -2, -2, 2    }, {  0, 1,  0 }, 0 },
    { { -2,  2, 0,     -2, -2, 0,     -2, -2, 2    }, {  1, 0,  0 }, 1 },   // left wall
    { { -2,  2, 0,     -2, -2, 2,     -2,  2, 2    }, {  1, 0,  0 }, 1 },
    { {  2, -2, 0,      2,  2, 0,      2, -2, 2    }, { -1, 0,  0 }, 2 },   // right wall
    { {  2, -2, 2,      2,  2, 0,      2,  2, 2    }, { -1, 0,  0 }, 2 },
    { { -2,  2, 2,     -2, -2, 2,      2,  2, 2    }, {  0, 0, -1 }, 0 },   // ceiling
    { { -2, -2, 2,      2,  2, 2,      2, -2, 2    }, {  0, 0, -1 }, 0 },
    { { -1,  1, 1.99f, -1, -1, 1.99f,  1,  1, 1.99f }, {  0, 0, -1 }, 3 },  // light
    { { -1, -1, 1.99f,  1,  1, 1.99f,  1, -1, 1.99f }, {  0, 0, -1 }, 3 },
};

inline uint64_t splitmix(uint64_t seed, uint64_t k) {       // k-th output of SplitMix64(seed)
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float u01(uint64_t seed, uint64_t k) { return (float)(splitmix(seed, k) >> 40) * 0x1.0p-24f; }

// The generators are counter-based, so ranges can be filled by independent host threads.
template <class F>
void parallel_ranges(uint64_t n, F fn) {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned nt = n < (1u << 16) ? 1u : hw;
    if (nt == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    try { th.reserve(nt); } catch (...) { fn(0, n); return; }          // emplace_back below then never allocates
    const uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const uint64_t b = std::min<uint64_t>(per * t, n), e = std::min<uint64_t>(b + per, n);
        if (b >= e) continue;
        // fn only writes caller-owned memory and cannot throw; a thread that cannot be started runs inline
        try { th.emplace_back([=] { fn(b, e); }); } catch (...) { fn(b, e); }
    }
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

uint32_t rt_gen_cornell_triangles(void* out, uint32_t cap) {
    const Material* mats[4] = { &kWhite, &kLeft, &kRight, &kEmit };
    Triangle* t = static_cast<Triangle*>(out);
    for (uint32_t i = 0; i < 12 && i < cap; ++i) {
        const CornellRow& r = kCornell[i];
        Vec3 n = { r.n[0], r.n[1], r.n[2] };
        t[i] = make_tri({ r.p[0], r.p[1], r.p[2] }, { r.p[3], r.p[4], r.p[5] }, { r.p[6], r.p[7], r.p[8] }, n, n, n,
                        *mats[r.mat]);
    }
    return 12;
}

uint32_t rt_gen_cornell_spheres(void* out, uint32_t cap) {
    // VkRenderer.cpp:240-245 (double literals narrowed to float by the Sphere/Material members)
    const float cx[3] = { -1.0f, 0.0f, 1.0f };
    const Material m[3] = { make_mat(1.0f, 0.9f, 0.0f, 0, 0, 0, 0.0f, 0.1f, 1.0f),     // gold
                            make_mat(0.7f, 0.7f, 0.7f, 0, 0, 0, 0.0f, 0.1f, 1.0f),     // silver
                            make_mat(0.0f, 0.0f, 1.0f, 0, 0, 0, 0.0f, 1.0f, 0.0f) };   // flat blue
    Sphere* s = static_cast<Sphere*>(out);
    for (uint32_t i = 0; i < 3 && i < cap; ++i) {
        std::memset(&s[i], 0, sizeof(Sphere));
        s[i].m_center = { cx[i], 0.0f, 0.2f };
        s[i].m_radius = 0.2f;
        s[i].m_material = m[i];
    }
    return 3;
}

// Latitude/longitude tessellation with the fp32 arithmetic of src/math/Sphere.h:17-79:
// (stacks+1) x (slices+1) grid, stack angle from +pi/2 down, two triangles per quad
// (v[i][j], v[i+1][j], v[i+1][j+1]) and (v[i][j], v[i+1][j+1], v[i][j+1]), normals = normalize(v - c).
uint32_t rt_gen_sphere_geometry(const float center[3], float radius, uint32_t stacks, uint32_t slices,
                                const float material12[12], void* out, uint32_t cap) try {
    const uint64_t total64 = 2ull * stacks * slices;
    if (total64 > 0xFFFFFFFFull) return 0;
    const uint32_t total = (uint32_t)total64;
    if (cap == 0 || out == nullptr || stacks == 0 || slices == 0) return total;

    Material mat;
    std::memcpy(&mat, material12, sizeof mat);
    const float pi = 3.14159265358979323846f;
    const float stack_step = pi / stacks;
    const float slice_step = (2.0f * pi) / slices;
    const Vec3 c = { center[0], center[1], center[2] };

    std::vector<Vec3> grid((size_t)(stacks + 1) * (slices + 1));
    for (uint32_t i = 0; i <= stacks; ++i) {
        const float stack_angle = pi / 2.0f - i * stack_step;
        const float xy = radius * cosf(stack_angle);
        const float z = radius * sinf(stack_angle);
        for (uint32_t j = 0; j <= slices; ++j) {
            const float slice_angle = j * slice_step;
            grid[(size_t)i * (slices + 1) + j] = { c.x + xy * cosf(slice_angle), c.y + xy * sinf(slice_angle), c.z + z };
        }
    }
    auto nrm = [&](const Vec3& p) { return normalized({ p.x - c.x, p.y - c.y, p.z - c.z }); };

    Triangle* t = static_cast<Triangle*>(out);
    uint32_t k = 0;
    for (uint32_t i = 0; i < stacks; ++i) {
        for (uint32_t j = 0; j < slices; ++j) {
            const Vec3& a = grid[(size_t)i * (slices + 1) + j];
            const Vec3& b = grid[(size_t)(i + 1) * (slices + 1) + j];
            const Vec3& d = grid[(size_t)(i + 1) * (slices + 1) + j + 1];
            const Vec3& e = grid[(size_t)i * (slices + 1) + j + 1];
            if (k < cap) t[k] = make_tri(a, b, d, nrm(a), nrm(b), nrm(d), mat);
            ++k;
            if (k < cap) t[k] = make_tri(a, d, e, nrm(a), nrm(d), nrm(e), mat);
            ++k;
        }
    }
    return total;
} catch (...) {
    return 0;            // out of host memory for the vertex grid: nothing written (a real count is never 0 here)
}

// centre uniform in [0,1)^3, vertices = centre + uniform in [-h, h)^3.
// Counter-based SplitMix64 (12 draws per triangle) so any sub-range can be generated independently.
uint32_t rt_gen_soup(uint64_t seed, uint32_t n, float half_extent, void* out, uint32_t cap) {
    Triangle* t = static_cast<Triangle*>(out);
    const uint32_t m = n < cap ? n : cap;
    parallel_ranges(m, [=](uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; ++i) {
        const uint64_t b = 12ull * i;
        Vec3 c = { u01(seed, b), u01(seed, b + 1), u01(seed, b + 2) };
        Vec3 p[3];
        for (int v = 0; v < 3; ++v) {
            p[v].x = c.x + (2.0f * u01(seed, b + 3 + 3 * v) - 1.0f) * half_extent;
            p[v].y = c.y + (2.0f * u01(seed, b + 4 + 3 * v) - 1.0f) * half_extent;
            p[v].z = c.z + (2.0f * u01(seed, b + 5 + 3 * v) - 1.0f) * half_extent;
        }
        Vec3 e1 = { p[1].x - p[0].x, p[1].y - p[0].y, p[1].z - p[0].z };
        Vec3 e2 = { p[2].x - p[0].x, p[2].y - p[0].y, p[2].z - p[0].z };
        Vec3 g = { e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x };
        float l2 = g.x * g.x + g.y * g.y + g.z * g.z;
        Vec3 nn = l2 > 0.f ? normalized(g) : Vec3{ 0.f, 0.f, 1.f };
        t[i] = make_tri(p[0], p[1], p[2], nn, nn, nn, kWhite);
    }
    });
    return n;
}

// SURVEY.md 8(d) C4 rays: origin uniform in [0,1)^3, direction uniform on the sphere (normalised
// Gaussian triple via Box-Muller), tmin 1e-6, tmax 1e20.  8 draws per ray, counter-based.
void rt_gen_random_rays(uint64_t seed, uint64_t first, uint64_t n, rt_ray* out) {
    parallel_ranges(n, [=](uint64_t i0, uint64_t i1) {
    for (uint64_t i = i0; i < i1; ++i) {
        const uint64_t b = 8ull * (first + i);
        rt_ray r;
        r.ox = u01(seed, b); r.oy = u01(seed, b + 1); r.oz = u01(seed, b + 2);
        float g[4];
        for (int k = 0; k < 2; ++k) {
            float u1 = 1.0f - u01(seed, b + 3 + 2 * k);          // (0,1]
            float u2 = u01(seed, b + 4 + 2 * k);
            float rad = std::sqrt(-2.0f * std::log(u1)), ang = 6.283185307179586f * u2;
            g[2 * k] = rad * std::cos(ang); g[2 * k + 1] = rad * std::sin(ang);
        }
        float l = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        if (!(l > 0.f)) { g[0] = 0.f; g[1] = 0.f; g[2] = 1.f; l = 1.f; }
        r.dx = g[0] / l; r.dy = g[1] / l; r.dz = g[2] / l;
        r.tmin = 1e-6f; r.tmax = 1e20f;
        out[i] = r;
    }
    });
}

// Camera::Camera + updateCameraVectors (src/application/Camera.h:24-41): worldUp is forced to
// (0,0,-1); front = normalize(lookAt - pos); right = normalize(front x worldUp); up = normalize(right x front).
void rt_camera_look_at(const float pos[3], const float look_at[3], float fov_deg, float aspect, void* ubo) {
    CameraUBO c;
    std::memset(&c, 0, sizeof c);
    c.m_position = { pos[0], pos[1], pos[2] };
    c.m_lookAt = { look_at[0], look_at[1], look_at[2] };
    c.m_worldUp = { 0.0f, 0.0f, -1.0f };
    auto cross = [](const Vec3& a, const Vec3& b) { return Vec3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; };
    c.m_front = normalized({ look_at[0] - pos[0], look_at[1] - pos[1], look_at[2] - pos[2] });
    c.m_right = normalized(cross(c.m_front, c.m_worldUp));
    c.m_up = normalized(cross(c.m_right, c.m_front));
    c.m_fov = fov_deg; c.m_aspectRatio = aspect;
    c.m_nearPlane = 0.1f; c.m_farPlane = 100.0f;           // VkRenderer.cpp:10-11 (unused by the shader)
    std::memcpy(ubo, &c, sizeof c);
}

// Camera mutators of the reference (src/application/Camera.h:56-90) on a 112-byte UBO in place, for hosts without the C++
// header (raytracer_amd/host/rt_camera.hpp holds the class itself): what the GLFW callbacks do between frames
// (src/application/Application.h:53-174) before updateUniformBuffer hands the UBO over.  Order within one call:
// moveForward, moveRight, moveUp / rotateAroundUp, rotateAroundRight; zero arguments are skipped.
void rt_camera_move(void* ubo, float forward, float right, float up) {
    Camera c;
    std::memcpy(&c.m_cameraUBO, ubo, sizeof(CameraUBO));
    if (forward != 0.0f) c.moveForward(forward);
    if (right != 0.0f) c.moveRight(right);
    if (up != 0.0f) c.moveUp(up);
    std::memcpy(ubo, &c.m_cameraUBO, sizeof(CameraUBO));
}

void rt_camera_rotate(void* ubo, float around_up_deg, float around_right_deg) {
    Camera c;
    std::memcpy(&c.m_cameraUBO, ubo, sizeof(CameraUBO));
    if (around_up_deg != 0.0f) c.rotateAroundUp(around_up_deg);
    if (around_right_deg != 0.0f) c.rotateAroundRight(around_right_deg);
    std::memcpy(ubo, &c.m_cameraUBO, sizeof(CameraUBO));
}

}  // extern "C"
