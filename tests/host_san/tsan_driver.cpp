// ThreadSanitizer driver (CPU): the host half of rt_scene_upload with its fork-join passes and task pool --
// data-parallel top of the build (n > 2^19), subtree tasks, parallel flatten, record loops, material table.
#include "../../raytracer_amd/csrc/scene_host.hpp"
#include "../../include/raytracer_amd.h"
#include <cstdio>
#include <vector>
using namespace rt;
int main() {
    for (uint32_t n : { 700000u, 90000u }) {
        std::vector<char> tris((size_t)n * 144);
        rt_gen_soup(0x5EED0001, n, 0.004f, tris.data(), n);
        HostSceneParams p; p.threads = 6; p.wide = true; p.max_depth = 30;
        HostScene hs; std::string err;
        int rc = prepare_scene(tris.data(), n, 144, nullptr, 0, 0, nullptr, 0, 0, p, hs, err);
        printf("n %u rc %d nodes %u\n", n, rc, hs.n_nodes);
        if (rc != 0 || hs.n_nodes == 0) return 1;
    }
}
