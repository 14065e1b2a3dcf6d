// crt_main -- still-image driver.  Replaces the reference's app/main.cpp (which hard-codes
// /home/daniel paths, app/main.cpp:12,16) with the same flow taking its paths from the command line:
//   parse scene -> RayTracer(scene) -> render(path, options) -> PPM.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../host/RayTracer.h"
#include "../host/SceneParser.h"

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr,
                 "usage: %s scene.crtscene out.ppm [--folder DIR] [--depth N] [--mode 0..9] [--device D | --devices 0-7 | --devices 0,2,5] [--repeat K]\n"
                 "       [--gi GI_SAMPLE_SIZE RAYS_PER_PIXEL [--seed S]]   (RenderOptions::USE_GI, RayTracer.h:27-30)\n",
                 argv[0]);
    return 2;
  }
  std::string scenePath = argv[1], outPath = argv[2], folder;
  unsigned depth = 5;
  int mode = crt::BVHBucketsThreadPool, device = 0, repeat = 1;
  std::vector<int> devices;  // --devices: the frame's tiles over several GPUs (first one gathers)
  bool useGI = false;
  unsigned giSamples = 2, raysPerPixel = 1, seed = 0;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--folder") && i + 1 < argc) folder = argv[++i];
    else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--mode") && i + 1 < argc) mode = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
      const char *p = argv[++i];
      while (*p) {
        char *end = nullptr;
        long a = strtol(p, &end, 10), b = a;
        if (end == p) break;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); if (end == p) break; }
        for (long d = a; d <= b && devices.size() < 64; d++) devices.push_back((int)d);
        p = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',') break;
      }
    }
    else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--gi") && i + 2 < argc) { useGI = true; giSamples = (unsigned)atoi(argv[++i]); raysPerPixel = (unsigned)atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)strtoul(argv[++i], nullptr, 0);
  }
  try {
    crt::SceneParser parser;
    crt::Scene scene = parser.parseScene(scenePath, folder);
    auto t0 = std::chrono::high_resolution_clock::now();
    std::unique_ptr<crt::RayTracer> tracerPtr(devices.size() > 0 ? new crt::RayTracer(scene, devices) : new crt::RayTracer(scene, device));
    crt::RayTracer &tracer = *tracerPtr;
    auto t1 = std::chrono::high_resolution_clock::now();
    crt::RenderOptions options((crt::RenderOptimization)mode, depth, useGI, giSamples, raysPerPixel);
    tracer.setGISeed(seed);  // frame r of this run uses seed + r (the reference's GI frames all differ: clock() ^ thread id)
    double best = 1e30;
    for (int r = 0; r < repeat; r++) {
      auto a = std::chrono::high_resolution_clock::now();
      tracer.render(r + 1 == repeat ? outPath : std::string(), options);
      auto b = std::chrono::high_resolution_clock::now();
      best = std::min(best, std::chrono::duration<double>(b - a).count());
    }
    if (!tracer.renderNote().empty()) std::fprintf(stderr, "note: %s\n", tracer.renderNote().c_str());
    crt_stats st = tracer.stats();
    // the reference prints the elapsed seconds of the render window (MEASURE_TIME, RayTracer.cpp:289-293)
    std::printf("%.6fs\n", best);
    std::printf("{\"build_s\": %.6f, \"render_s\": %.6f, \"kernel_ms\": %.4f, \"width\": %u, \"height\": %u}\n",
                std::chrono::duration<double>(t1 - t0).count(), best, st.kernel_ms, scene.sceneSettings.image.width,
                scene.sceneSettings.image.height);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
