// bvh.hpp -- host-side acceleration-structure build for the MI355X ray engine.
//
// Build-original: the reference has no acceleration structure (shaders/frag.glsl:265-301 is a linear
// loop over every primitive).  The correctness contract is "return exactly what that loop returns"
// (consequence 2); DESIGN.md explains how boxes are padded so that traversal
// never culls the primitive the loop would have chosen.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

namespace rt {

struct Aabb {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
    void grow(const Aabb& b) { for (int a = 0; a < 3; ++a) { if (b.lo[a] < lo[a]) lo[a] = b.lo[a]; if (b.hi[a] > hi[a]) hi[a] = b.hi[a]; } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { if (p[a] < lo[a]) lo[a] = p[a]; if (p[a] > hi[a]) hi[a] = p[a]; } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// Binary build tree.  count > 0 => leaf over prim_index[first, first+count).
struct BuildNode {            // trivially constructible on purpose: see NodeArray
    Aabb box;
    uint32_t left, right;
    uint32_t first, count;
};

// The build nodes of a tree: up to 2n - 1 records for n primitives, of which a build with 4-triangle leaves uses a third.
// Storage is allocated UNINITIALISED (a std::vector would write all 800 MB of a 10 M-triangle build's worst case before the
// first node exists); the builder fills every node it hands out, size() is the number handed out.
class NodeArray {
public:
    void allocate(size_t capacity) { p_.reset(capacity ? new BuildNode[capacity] : nullptr); cap_ = capacity; n_ = capacity; }
    void resize_down(size_t n) { if (n < n_) n_ = n; }
    void clear() { p_.reset(); cap_ = n_ = 0; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    BuildNode& operator[](size_t i) { return p_[i]; }
    const BuildNode& operator[](size_t i) const { return p_[i]; }
    const BuildNode* begin() const { return p_.get(); }
    const BuildNode* end() const { return p_.get() + n_; }
    void swap(NodeArray& o) { p_.swap(o.p_); std::swap(cap_, o.cap_); std::swap(n_, o.n_); }
private:
    std::unique_ptr<BuildNode[]> p_;
    size_t cap_ = 0, n_ = 0;
};

struct BuildParams {
    int   max_leaf = 4;       // primitives per leaf (<= 8: leaf references carry a 3-bit count)
    int   max_depth = 30;     // leaves sit at depth <= max_depth (root = 0); bounds the traversal stack
    float cost_trav = 0.6f;   // SAH: one interior step
    float cost_isect = 1.0f;  // SAH: one triangle test
    int   threads = 0;        // 0 = hardware concurrency, capped at 16
    int   bins = 16;          // SAH bins per axis (4..32)
};

struct BuildResult {
    NodeArray nodes;                   // nodes[0] = root
    std::vector<uint32_t>  prim_index; // permutation of [0, n)
    uint32_t depth = 0;
    uint32_t n_leaves = 0;
    float    sah_cost = 0.f;
    bool     oversize_leaf = false;   // some leaf holds more than 8 primitives (the depth budget ran out): not flattenable
};

// prim_boxes: n exact (unpadded) primitive bounds.
void build_bvh2(const Aabb* prim_boxes, uint32_t n, const BuildParams& p, BuildResult& out);

// ---- flattened HBM layout (see DESIGN.md "Data layout") ---------------------------------------
// Interior node, 64 B = 4 x float4 (16-B aligned SoA of the two children):
//   q0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//   q1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   q2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//   q3 = (bits child0, bits child1, 0, 0)
// child reference: >= 0 interior node index; < 0 leaf: ~ref = (first_tri << 3) | (count - 1).
struct FlatBvh2 {
    std::vector<float>    nodes;      // 16 floats per interior node
    std::vector<uint32_t> tri_order;  // tri_order[k] = original triangle index stored at slot k
    uint32_t n_nodes = 0, n_leaves = 0, depth = 0;
};
static constexpr int32_t kEmptyLeaf = 0x7fffffff;   // never visited: its box is inverted

void flatten_bvh2(const BuildResult& b, float pad, FlatBvh2& out);

// ---- 4-wide, quantised layout ("BVH4q") ---------------------------------------------------------
// Incoherent traversal on CDNA4 is bound by the vector-memory path (one pass per cache line per load
// instruction), not by VALU, so the node packs FOUR children into the 64 B that BVH2 spends on two:
//   q0 = (origin.x, origin.y, origin.z, scale.x)           fp32 grid step per axis, about extent / 255 (NOT a power of two)
//   q1 = child references 0..3 (same encoding as BVH2)
//   q2 = (lo.x[0..3], lo.y[0..3], lo.z[0..3], hi.x[0..3])     one byte per child, plane = origin + scale * q
//   q3 = (hi.y[0..3], hi.z[0..3], scale.y, scale.z)
// Empty slots are stored inverted (lo bytes 255, hi bytes 0).
// Quantisation rounds lo down and hi up from the PADDED child box, so every quantised box contains the box
// BVH2 would have used: traversal stays conservative and results stay identical to the brute-force loop.
struct FlatBvh4q {
    std::vector<uint32_t> nodes;      // 16 words per node
    std::vector<uint32_t> tri_order;
    uint32_t n_nodes = 0, n_leaves = 0, depth = 0;
    uint32_t stack_need = 0;          // exact bound on traversal-stack entries (excluding the sentinel)
    // The same nodes with fp32 planes for the packet traversal of the tile kernel (a wave walks ONE node at a time and
    // fetches it with scalar loads, so nothing has to be decoded per lane), 32 words = 128 B per node:
    //   [0..3] child references, [4] number of children, [5..7] 0,
    //   [8 + 6k ..] child k: lo.x, hi.x, lo.y, hi.y, lo.z, hi.z  (the PADDED box, not the quantised one; empty slots: every plane at +3e38)
    std::vector<uint32_t> wide;       // filled only when flatten_bvh4q is asked to (want_wide)
};

void flatten_bvh4q(const BuildResult& b, float pad, FlatBvh4q& out, bool want_wide = false, int threads = 0);

// exact bound on the BVH2 traversal stack (entries pushed along the worst root-to-leaf path)
uint32_t bvh2_stack_need(const BuildResult& b);

}  // namespace rt
