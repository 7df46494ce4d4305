// examples/main.cpp -- a PRT-style driver written against prt_amd/csrc/host/prt.h.
//
// It follows the call sequence of the reference's src/main.cpp (scene set-up -> Bvh::build -> Scene::add ->
// Camera::create -> Image -> PathTracer::TraceBlock -> save) to show that the C++ surface is a drop-in for that
// caller; the one deliberate difference is that the image is handed to the GPU in ONE TraceBlock call instead of one
// call per 16x16 tile from a thread pool (both work; per-tile calls pay a launch + download each).
//
//   ./prt_main [cornell|bunny|atrium] [width height spp [tiles]]      (needs an MI355X; there is no CPU path)
//   `tiles` drives the image the way the reference's main.cpp does: one TraceBlock per 16x16 tile from a thread pool
//   ./prt_main twoscenes [width height spp]   two different scenes rendered back to back from the SAME stack slot, the way
//   the reference's main() calls raytrace_scene() repeatedly (main.cpp:107-118, 192-200): render_a.pfm, render_b.pfm
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <mutex>

#include "../prt_amd/csrc/host/prt.h"

using namespace prt;

static void setupCornellBox(Scene& scene, Camera& camera, float& exposure, uint32_t width, uint32_t height, const char* teapotObj)
{
    auto cbox = new Bvh;
    cbox->build(SampleModels::getCornellBox(true));
    scene.add(cbox);
    if (teapotObj) { // main.cpp:28-50
        Material mat;
        mat.init();
        mat.diffuse = {0.9f, 0.9f, 0.9f};
        mat.reflectionType = ReflectionType::kSpecular;
        Mesh teapot;
        teapot.loadObj(teapotObj, mat);
        auto pos = teapot.getPositionBuffer();
        for (uint32_t i = 0; i < teapot.getVertexCount(); i++) {
            float s = 0.005f;
            pos[i] = s * pos[i] + Vector3f(-0.5f, 0.0f, 0.5f);
        }
        teapot.calculateVertexNormals();
        teapot.calculateBounds();
        auto teapotBvh = new Bvh;
        teapotBvh->build(std::move(teapot));
        scene.add(teapotBvh);
    }
    camera.create({0, 0.965, 2.6}, {0, 0, -1.0f}, width, height);
    exposure = 1.0f;
}

static void setupBunnyStandIn(Scene& scene, Camera& camera, float& exposure, uint32_t width, uint32_t height)
{
    auto cbox = new Bvh;
    cbox->build(SampleModels::getCornellBox(true));
    scene.add(cbox);
    Material mat;
    mat.init();
    mat.diffuse = {0.8f, 0.75f, 0.7f};
    Mesh m = SampleModels::getDisplacedSphere(69451, 0.3f, Vector3f(-0.45f, 0.36f, 0.45f), mat, 1);
    m.calculateVertexNormals();
    m.calculateBounds();
    auto b = new Bvh;
    b->buildOnDevice(std::move(m)); // the same tree as Bvh::build, built on the GPU (prt_hip_build_bvh)
    scene.add(b);
    scene.setDirectionalLight(normalize(Vector3f(0.2f, 1.0f, 0.2f)), Vector3f(16.7f, 15.6f, 11.7f)); // main.cpp:84
    camera.create({0, 0.965, 2.6}, {0, 0, -1.0f}, width, height);
    exposure = 1.0f;
}

static void setupAtriumStandIn(Scene& scene, Camera& camera, float& exposure, uint32_t width, uint32_t height)
{
    Mesh m = SampleModels::getAtrium(262000, 1, true, true, 0.0f);
    m.calculateVertexNormals();
    auto b = new Bvh;
    b->build(std::move(m));
    scene.setDirectionalLight(normalize(Vector3f(0.05f, 1.0f, 0.1f)), Vector3f(16.7f, 15.6f, 11.7f)); // main.cpp:66
    scene.add(b);
    camera.create({-15.0f, 4.0f, 0.5f}, {1.0f, 0.08f, -0.05f}, width, height);
    exposure = 1.0f;
}

// main.cpp:107-190 in short: a stack Scene, per-tile TraceBlock calls, save.  Called twice by `twoscenes`.
static void raytraceScene(uint32_t width, uint32_t height, uint32_t samples, const char* imagePath, bool bunny)
{
    Scene scene;
    Camera camera;
    float exposure;
    scene.init();
    if (bunny) setupBunnyStandIn(scene, camera, exposure, width, height);
    else setupCornellBox(scene, camera, exposure, width, height, nullptr);
    Image image(width, height, true, exposure);
    const uint32_t kTile = 16;
    for (uint32_t y = 0; y < height; y += kTile)
        for (uint32_t x = 0; x < width; x += kTile) {
            PathTracer t;
            t.TraceBlock(image, x, y, std::min(x + kTile, width - 1), std::min(y + kTile, height - 1), scene, camera, samples);
        }
    image.savePfm(imagePath);
}

int main(int argc, char** argv)
{
    const char* which = argc > 1 ? argv[1] : "cornell";
    if (!strcmp(which, "twoscenes")) {
        uint32_t w = argc > 2 ? (uint32_t)atoi(argv[2]) : 128, h = argc > 3 ? (uint32_t)atoi(argv[3]) : 128;
        uint32_t spp = argc > 4 ? (uint32_t)atoi(argv[4]) : 16;
        raytraceScene(w, h, spp, "render_a.pfm", true);
        raytraceScene(w, h, spp, "render_b.pfm", false); // same camera, same stack addresses, another scene
        PathTracer::releaseDevice();
        return 0;
    }
    uint32_t width = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024, height = argc > 3 ? (uint32_t)atoi(argv[3]) : 1024;
    const uint32_t kSamples = argc > 4 ? (uint32_t)atoi(argv[4]) : 64; // main.cpp:125

    Scene scene;
    Camera camera;
    float exposure;
    scene.init();
    if (!strcmp(which, "bunny")) setupBunnyStandIn(scene, camera, exposure, width, height);
    else if (!strcmp(which, "atrium")) setupAtriumStandIn(scene, camera, exposure, width, height);
    else setupCornellBox(scene, camera, exposure, width, height, getenv("PRT_TEAPOT_OBJ"));

    Image image(width, height, true, exposure);
    auto start = std::chrono::steady_clock::now();
    PathTracer tracer;
    Stats st;
    st.clear();
    double kernelMs = 0.0;
    if (argc > 5 && !strcmp(argv[5], "tiles")) {
        // the reference's own loop (main.cpp:121-176): one task per 16x16 tile on the thread pool, a PathTracer per task
        ThreadPool threadPool;
        threadPool.create(8);
        std::mutex statsMutex;
        const uint32_t kTile = 16;
        for (uint32_t y = 0; y < height; y += kTile) {
            auto y0 = y, y1 = std::min(y + kTile, height - 1);
            for (uint32_t x = 0; x < width; x += kTile) {
                auto x0 = x, x1 = std::min(x + kTile, width - 1);
                threadPool.queue([&image, x0, y0, x1, y1, &scene, &camera, &st, &statsMutex, &kernelMs, kSamples]() {
                    PathTracer t;
                    t.TraceBlock(image, x0, y0, x1, y1, scene, camera, kSamples);
                    std::lock_guard<std::mutex> g(statsMutex);
                    st.merge(t.getStats());
                    kernelMs += t.getKernelMs();
                });
            }
        }
        threadPool.waitAllTasksDone();
    } else {
        tracer.TraceBlock(image, 0, 0, width - 1, height - 1, scene, camera, kSamples);
        st = tracer.getStats();
        kernelMs = tracer.getKernelMs();
    }
    auto end = std::chrono::steady_clock::now();
    auto ms = (float)std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count();
    printf("%.3fms @%uspp, %llu rays (%llu occlusion), kernels %.3f ms => %.1f Mray/s\n", ms, kSamples, (unsigned long long)st.raysTraced,
           (unsigned long long)st.occludedTraced, kernelMs, st.raysTraced / kernelMs / 1e3);
    image.saveExr("render.exr");  // main.cpp:189
    image.savePfm("render.pfm");  // the same pixels as raw floats
    image.savePpm("render.ppm");
    PathTracer::releaseDevice();
    return 0;
}
