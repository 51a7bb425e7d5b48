// scene_host.hpp -- the host half of rt_scene_upload (no HIP): bounds, BVH build, flatten, device records.
// Kept free of the HIP runtime so that it runs under host sanitizers (tests/host_san) and in rt_bvh_build_host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "bvh.hpp"

namespace rt {

// The two big record arrays (480 MB each for 10 M triangles) are filled completely by parallel loops; a std::vector would
// zero them on one thread first.  Minimal owner of uninitialised storage with the few members the callers use.
template <class T>
class UninitArray {
public:
    void resize(size_t n) { p_.reset(n ? new T[n] : nullptr); n_ = n; }      // contents indeterminate
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T* data() { return p_.get(); }
    const T* data() const { return p_.get(); }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
private:
    std::unique_ptr<T[]> p_;
    size_t n_ = 0;
};

struct HostSceneParams {
    int layout = 3;          // kBvh2 (2) or kBvh4q (3); a 4-wide tree whose stack would not fit falls back to 2
    int max_leaf = 4;
    int pad_log2 = 15;       // boxes are padded by max|coordinate| * 2^-pad_log2
    int threads = 0;
    int max_depth = 30;
    int bins = 16;           // SAH bins per axis
    float cost_trav = 0.6f;  // SAH cost of one interior step relative to one triangle test
    bool wide = false;       // also flatten the fp32 "wide" copy of a BVH4q tree (packet traversal of the tile kernel)
    bool records = true;     // false: nodes + triangle order only (rt_bvh_build_host)
};

struct HostScene {
    std::vector<uint32_t> nodes;       // 16 words per node, layout `layout`
    std::vector<uint32_t> wide;        // 32 words per node (bvh.hpp FlatBvh4q::wide) or empty
    std::vector<uint32_t> tri_order;   // tri_order[k] = original index of the triangle stored at slot k
    UninitArray<float> tri48;          // 12 floats per triangle in leaf order: (v0, bits id) (e1, 0) (e2, 0)
    UninitArray<float> nrm48;          // 12 floats per triangle in ORIGINAL order: n0, n1, n2; material id in n0.w
    std::vector<float> sph;            // (centre, radius)
    std::vector<float> mats;           // 12 floats per distinct material
    std::vector<float> sphere_mat;     // uint32 bits: material id per sphere
    std::vector<float> lights8;        // (position, intensity) (colour, 0)
    uint32_t n_nodes = 0, n_leaves = 0, depth = 0, stack_need = 0;
    int layout = 3;
    float sah_cost = 0.f, maxabs = 0.f;
    Aabb box;                          // exact bounds of the triangles
};

// RT_OK or a negative RT_E_* code with `err` set.  May throw std::bad_alloc / std::system_error (the ABI entry
// points catch: abi_guard.hpp).  Strides are validated by the caller.
int prepare_scene(const void* tris, uint32_t n_tris, uint32_t tri_stride, const void* spheres, uint32_t n_spheres,
                  uint32_t sphere_stride, const void* lights, uint32_t n_lights, uint32_t light_stride,
                  const HostSceneParams& p, HostScene& out, std::string& err);

}  // namespace rt
