// raytracer.hpp -- C++ host wrapper over the C ABI (include/raytracer_amd.h).
//
// Keeps the reference's error convention: every failure becomes std::runtime_error, caught once in main
// (src/main.cpp:8-14).  Single caller thread, blocking calls, caller keeps ownership of its vectors -- the
// same contract VkRenderer has with its SSBOs (src/vulkan/VkRenderer.h:80-90, VkRenderer.cpp:260-326).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "raytracer_amd.h"
#include "rt_types.hpp"

namespace rt {

class Engine {
public:
    explicit Engine(const std::vector<int>& devices = {0}) {
        int rc = rt_create(&ctx_, devices.data(), (int)devices.size());
        if (rc != RT_OK) throw std::runtime_error("rt_create failed (" + std::to_string(rc) + "): no usable HIP device");
    }
    ~Engine() { rt_destroy(ctx_); }
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;

    // VkRenderer::createData's upload (VkRenderer.cpp:260-326)
    void upload(const void* tris, size_t n_tris, size_t tri_stride, const void* spheres, size_t n_spheres,
                size_t sphere_stride, const void* lights, size_t n_lights, size_t light_stride) {
        check(rt_scene_upload(ctx_, tris, (uint32_t)n_tris, (uint32_t)tri_stride, spheres, (uint32_t)n_spheres,
                              (uint32_t)sphere_stride, lights, (uint32_t)n_lights, (uint32_t)light_stride));
    }
    void upload(const std::vector<Triangle>& t, const std::vector<Sphere>& s, const std::vector<Light>& l) {
        upload(t.data(), t.size(), sizeof(Triangle), s.data(), s.size(), sizeof(Sphere), l.data(), l.size(), sizeof(Light));
    }
    // VkRenderer::updateUniformBuffer (VkRenderer.cpp:855-862)
    void setCamera(const void* ubo112) { check(rt_set_camera(ctx_, ubo112)); }
    // the draw (VkRenderer.cpp:1123), intersection part of frag.glsl:main
    void tracePrimary(uint32_t w, uint32_t h, rt_hit* out) { check(rt_trace_primary(ctx_, w, h, out)); }
    void trace(const rt_ray* rays, uint64_t n, rt_hit* out) { check(rt_trace(ctx_, rays, n, out)); }
    void occluded(const rt_ray* rays, uint64_t n, uint8_t* out) { check(rt_occluded(ctx_, rays, n, out)); }
    // the draw, whole shader (frag.glsl:303-360): spp x bounces path loop -> RGBA8, row 0 = top
    void render(uint32_t w, uint32_t h, uint32_t spp, uint32_t bounces, uint32_t seed, uint8_t* rgba8) {
        check(rt_render(ctx_, w, h, spp, bounces, seed, rgba8));
    }
    // Wavefront OBJ -> Triangle records with one material
    static std::vector<Triangle> loadObj(const std::string& path, const Material* material = nullptr) {
        uint64_t n = 0;
        int rc = rt_load_obj(path.c_str(), material, nullptr, 0, &n);
        std::vector<Triangle> t(n);
        if (rc == RT_OK && n) rc = rt_load_obj(path.c_str(), material, t.data(), n, &n);
        if (rc != RT_OK) throw std::runtime_error("rt_load_obj failed for " + path + " (code " + std::to_string(rc) + ")");
        return t;
    }
    void setOption(int key, int64_t v) { check(rt_set_option(ctx_, key, v)); }
    rt_scene_stats stats() const { rt_scene_stats s; check(rt_scene_stats_get(ctx_, &s)); return s; }
    rt_ctx* raw() { return ctx_; }

private:
    void check(int rc) const {
        if (rc != RT_OK) throw std::runtime_error(std::string(rt_last_error(ctx_)) + " (code " + std::to_string(rc) + ")");
    }
    rt_ctx* ctx_ = nullptr;
};

}  // namespace rt
