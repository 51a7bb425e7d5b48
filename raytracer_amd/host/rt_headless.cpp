// rt_headless.cpp -- headless stand-in for Application::run + VkRenderer::drawFrame
// (src/application/Application.cpp:15-21, src/vulkan/VkRenderer.cpp:1287-1374): builds the reference's
// Cornell scene and default camera, traces one frame of primary rays through the C ABI and writes a PPM
// that visualises the hit buffer (primitive id + distance).  Error handling mirrors src/main.cpp:5-17.
//
// With spp and bounces it runs the whole shader instead (rt_render, frag.glsl:303-360) and writes the RGBA8 frame --
// the headless form of the presentation hook.  A Wavefront OBJ can replace the three
// Cornell spheres.
//
// The camera is the reference's interactive Camera (rt_camera.hpp, src/application/Camera.h): optional yaw / pitch (degrees)
// and dolly arguments apply rotateAroundUp / rotateAroundRight / moveForward before the frame, the way the GLFW callbacks
// (src/application/Application.h:53-174) would between frames.
//
// usage: rt_headless [width height [out.ppm [spp bounces [mesh.obj|- [yaw_deg pitch_deg dolly]]]]]
// RT_HEADLESS_DEVICES="0,1,2,3" (environment): the HIP device ordinals of the engine's slots -- with several, the frame is dealt
// out in interleaved 8-row stripes and gathered on the first device (RT_OPT_FRAME_EXCHANGE); an ordinal may repeat.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "raytracer.hpp"
#include "rt_camera.hpp"

int main(int argc, char** argv) {
    try {
        const uint32_t w = argc > 2 ? (uint32_t)std::atoi(argv[1]) : 1200;     // Config::INIT_WINDOW_WIDTH/HEIGHT
        const uint32_t h = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1000;     // (src/globals/globals.h:12-13)
        const char* path = argc > 3 ? argv[3] : "frame.ppm";

        std::vector<rt::Triangle> tris(rt_gen_cornell_triangles(nullptr, 0));
        rt_gen_cornell_triangles(tris.data(), (uint32_t)tris.size());
        std::vector<rt::Sphere> spheres(rt_gen_cornell_spheres(nullptr, 0));
        rt_gen_cornell_spheres(spheres.data(), (uint32_t)spheres.size());
        std::vector<rt::Light> lights;                                         // VkRenderer.cpp:252-258: none
        const uint32_t spp = argc > 5 ? (uint32_t)std::atoi(argv[4]) : 0, bounces = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 0;
        if (argc > 6 && std::string(argv[6]) != "-") {                         // mesh instead of the spheres
            const std::vector<rt::Triangle> mesh = rt::Engine::loadObj(argv[6]);
            tris.insert(tris.end(), mesh.begin(), mesh.end());
            spheres.clear();
        }

        std::vector<int> devices;
        if (const char* list = std::getenv("RT_HEADLESS_DEVICES")) {
            for (const char* p = list; *p;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) throw std::runtime_error(std::string("RT_HEADLESS_DEVICES: not a list of device ordinals: ") + list);
                devices.push_back((int)v);
                p = (*end == ',') ? end + 1 : end;
            }
        }
        if (devices.empty()) devices.push_back(0);
        rt::Engine engine(devices);
        engine.upload(tris, spheres, lights);
        // VkRenderer::VkRenderer (VkRenderer.cpp:3-12): position (0,4,1), lookAt (0,0,1), up (0,0,-1), vfov 45, near 0.1, far 100
        rt::Camera camera({0.0f, 4.0f, 1.0f}, {0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, -1.0f}, 45.0f, (float)w / (float)h, 0.1f, 100.0f);
        if (argc > 9) {
            camera.rotateAroundUp((float)std::atof(argv[7]));
            camera.rotateAroundRight((float)std::atof(argv[8]));
            camera.moveForward((float)std::atof(argv[9]));
        }
        rt::Camera::UniformBufferObject cam;
        camera.updateCameraUBO(cam, 0.0f);                                     // VkRenderer::updateUniformBuffer, :855-862
        engine.setCamera(&cam);

        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error(std::string("failed to open ") + path);
        f << "P6\n" << w << " " << h << "\n255\n";
        if (spp) {
            std::vector<uint8_t> frame((size_t)w * h * 4);
            engine.render(w, h, spp, bounces, 1u, frame.data());
            size_t lit = 0;
            for (size_t p = 0; p < (size_t)w * h; ++p) {
                f.write(reinterpret_cast<const char*>(&frame[4 * p]), 3);
                lit += (frame[4 * p] | frame[4 * p + 1] | frame[4 * p + 2]) != 0;
            }
            std::cout << "rendered " << w << "x" << h << " at " << spp << " spp, " << bounces << " bounces, " << lit
                      << " non-black pixels, " << engine.stats().n_tris << " triangles -> " << path << std::endl;
            return EXIT_SUCCESS;
        }
        std::vector<rt_hit> hits((size_t)w * h);
        engine.tracePrimary(w, h, hits.data());
        size_t n_hit = 0;
        for (const rt_hit& hit : hits) {
            unsigned char rgb[3] = {0, 0, 0};
            if (hit.prim != RT_MISS) {
                ++n_hit;
                const uint32_t id = (hit.prim & RT_SPHERE_BIT) ? 100u + (hit.prim & ~RT_SPHERE_BIT) : hit.prim;
                const float shade = 1.0f / (1.0f + 0.15f * hit.t);
                rgb[0] = (unsigned char)(255 * shade * (0.35f + 0.65f * ((id * 37u) % 7u) / 6.0f));
                rgb[1] = (unsigned char)(255 * shade * (0.35f + 0.65f * ((id * 17u) % 5u) / 4.0f));
                rgb[2] = (unsigned char)(255 * shade * (0.35f + 0.65f * ((id * 11u) % 3u) / 2.0f));
            }
            f.write(reinterpret_cast<const char*>(rgb), 3);
        }
        const rt_scene_stats st = engine.stats();
        std::cout << "traced " << w << "x" << h << " primary rays, " << n_hit << " hits, " << st.n_tris << " triangles, "
                  << st.n_spheres << " spheres -> " << path << std::endl;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
