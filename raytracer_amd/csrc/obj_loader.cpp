// obj_loader.cpp -- SURVEY.md 8(f2): minimal Wavefront OBJ reader behind the C ABI (rt_load_obj).
//
// The reference has no mesh loader (its only geometry sources are the literal scene of
// src/vulkan/VkRenderer.cpp:164-258 and Sphere::sphereGeometry, src/math/Sphere.h:17-79), but BASELINE's configs
// name real meshes (Bunny, Sponza, San Miguel).  This reader turns `v`, `vn` and `f` records into the reference's
// 144-B Triangle records (src/math/Triangle.h: three Vertex3D {position, normal} + Material), which is what
// rt_scene_upload takes.  Polygons are triangulated as fans; indices may be negative (relative) and may carry
// texture/normal parts (`a`, `a/b`, `a//c`, `a/b/c`); faces without normals get the geometric normal
// normalize(cross(v1-v0, v2-v0)), degenerate ones (0,0,0).  Everything else (`vt`, groups, materials) is skipped:
// every triangle gets the one material the caller passes.
#include "../../include/raytracer_amd.h"
#include "../host/rt_types.hpp"
#include "abi_guard.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace rt;

namespace {

struct Corner { long v, n; };      // 1-based OBJ indices, 0 = absent, negative = relative to the end

// parses "a", "a/b", "a//c", "a/b/c"; returns false at end of line
bool next_corner(const char*& p, Corner& c) {
    while (*p == ' ' || *p == '\t') ++p;
    if (*p == '\0' || *p == '\n' || *p == '\r' || *p == '#') return false;
    char* e = nullptr;
    c.v = std::strtol(p, &e, 10);
    c.n = 0;
    if (e == p) return false;
    p = e;
    if (*p == '/') {
        ++p;
        if (*p != '/') { (void)std::strtol(p, &e, 10); p = e; }          // texture index, unused
        if (*p == '/') { ++p; c.n = std::strtol(p, &e, 10); p = e; }
    }
    return true;
}

// a keyword ends at white space or at the end of the line / file (a bare `f` is a malformed record, not a comment)
bool is_end(char c) { return c == ' ' || c == '\t' || c == '\0' || c == '\n' || c == '\r'; }

bool resolve(long idx, size_t count, size_t& out) {
    if (idx > 0 && (size_t)idx <= count) { out = (size_t)idx - 1; return true; }
    if (idx < 0 && (size_t)(-idx) <= count) { out = count - (size_t)(-idx); return true; }
    return false;
}

}  // namespace

extern "C" int rt_load_obj(const char* path, const void* material48, void* tris144, uint64_t cap, uint64_t* n_tris_out) try {
    if (!path || !n_tris_out) return RT_E_INVALID;
    *n_tris_out = 0;
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return RT_E_INVALID;
    struct Closer { std::FILE* f; ~Closer() { std::fclose(f); } } closer{ f };     // also when a vector throws
    Material mat;
    std::memset(&mat, 0, sizeof mat);
    if (material48) std::memcpy(&mat, material48, sizeof mat);
    else { mat.m_albedo = { 0.8f, 0.8f, 0.8f }; mat.m_roughness = 0.5f; }
    std::vector<Vec3> pos, nrm;
    std::vector<Corner> face;
    Triangle* out = static_cast<Triangle*>(tris144);
    uint64_t n = 0;
    int rc = RT_OK;
    std::string line;
    char buf[4096];
    while (std::fgets(buf, sizeof buf, f)) {
        line.assign(buf);
        while (!line.empty() && line.back() != '\n' && std::fgets(buf, sizeof buf, f)) line.append(buf);   // long lines
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && is_end(p[1])) {
            Vec3 v{ 0, 0, 0 };
            if (std::sscanf(p + 1, "%f %f %f", &v.x, &v.y, &v.z) != 3) { rc = RT_E_INVALID; break; }
            pos.push_back(v);
        } else if (p[0] == 'v' && p[1] == 'n' && is_end(p[2])) {
            Vec3 v{ 0, 0, 0 };
            if (std::sscanf(p + 2, "%f %f %f", &v.x, &v.y, &v.z) != 3) { rc = RT_E_INVALID; break; }
            nrm.push_back(v);
        } else if (p[0] == 'f' && is_end(p[1])) {
            face.clear();
            const char* q = p + 1;
            Corner c;
            while (next_corner(q, c)) face.push_back(c);
            if (face.size() < 3) { rc = RT_E_INVALID; break; }
            for (size_t k = 1; k + 1 < face.size(); ++k) {                  // fan: (0, k, k+1)
                const Corner cs[3] = { face[0], face[k], face[k + 1] };
                size_t vi[3], ni[3];
                bool ok = true, has_n = true;
                for (int a = 0; a < 3; ++a) {
                    ok = ok && resolve(cs[a].v, pos.size(), vi[a]);
                    if (cs[a].n == 0) has_n = false;
                    else ok = ok && resolve(cs[a].n, nrm.size(), ni[a]);
                }
                if (!ok) { rc = RT_E_INVALID; break; }
                if (out && n < cap) {
                    Triangle t;
                    std::memset(&t, 0, sizeof t);
                    Vertex3D* vs[3] = { &t.m_v0, &t.m_v1, &t.m_v2 };
                    for (int a = 0; a < 3; ++a) vs[a]->m_position = pos[vi[a]];
                    if (has_n) {
                        for (int a = 0; a < 3; ++a) vs[a]->m_normal = nrm[ni[a]];
                    } else {
                        const Vec3 &a0 = pos[vi[0]], &a1 = pos[vi[1]], &a2 = pos[vi[2]];
                        const float e1[3] = { a1.x - a0.x, a1.y - a0.y, a1.z - a0.z }, e2[3] = { a2.x - a0.x, a2.y - a0.y, a2.z - a0.z };
                        float g[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                        const float len = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                        if (len > 0.f) { g[0] /= len; g[1] /= len; g[2] /= len; } else { g[0] = g[1] = g[2] = 0.f; }
                        for (int a = 0; a < 3; ++a) vs[a]->m_normal = { g[0], g[1], g[2] };
                    }
                    t.m_material = mat;
                    out[n] = t;
                }
                ++n;
            }
            if (rc != RT_OK) break;
        }
    }
    *n_tris_out = n;
    return rc;
} catch (...) {
    return rt::abi_exception(nullptr);       // RT_E_NOMEM for std::bad_alloc
}
