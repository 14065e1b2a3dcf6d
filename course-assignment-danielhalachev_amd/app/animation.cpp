// crt_animation -- camera-orbit driver.  Same flow as the reference's app/animation.cpp:8-40: one
// RayTracer (tree built and uploaded once), then FPS*SECONDS+1 frames of
//   setPosition(orbit) ; reset matrix ; pan(lookAtAngle) ; render(frame path)
// with the scene and tree resident on the GPU between frames (only the 12 camera floats change).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../host/RayTracer.h"
#include "../host/SceneParser.h"

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s scene.crtscene out_prefix [--folder DIR] [--depth N] [--fps F] [--seconds S] [--radius R] [--device D]\n", argv[0]);
    return 2;
  }
  std::string scenePath = argv[1], prefix = argv[2], folder;
  unsigned depth = 5;
  short FPS = 30, SECONDS = 10;  // app/animation.cpp:16-17
  float radius = 5.12f;          // app/animation.cpp:20
  int device = 0;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--folder") && i + 1 < argc) folder = argv[++i];
    else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--fps") && i + 1 < argc) FPS = (short)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) SECONDS = (short)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--radius") && i + 1 < argc) radius = (float)atof(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
  }
  try {
    crt::SceneParser parser;
    crt::Scene scene = parser.parseScene(scenePath, folder);
    crt::RayTracer tracer(scene, device);
    crt::RenderOptions options(crt::BVHBucketsThreadPool, depth, false);
    const float DEG_CHANGE = 360.0f / (FPS * SECONDS);
    float degrees = 0;
    for (float t = 0; t <= FPS * SECONDS; ++t) {  // app/animation.cpp:24-38
      float radians = degrees * (M_PIf / 180.0f);
      float x = sinf(radians) * radius;
      float z = cosf(radians) * radius - 3;
      tracer.setCamera().setPosition() = crt::Vector(x, 0, z);
      float deltaX = x - 0;
      float deltaZ = z + 3;
      tracer.setCamera().setRotationMatrix() = crt::Matrix3::identity();
      float lookAtAngle = std::atan2(deltaX, deltaZ) * (180.0f / M_PIf);
      tracer.setCamera().pan(lookAtAngle);
      tracer.render(prefix + std::to_string(t) + ".ppm", options);
      degrees += DEG_CHANGE;
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
