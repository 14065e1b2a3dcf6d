// crt::RayTracer -- the seam.  Same public surface as the reference's class
// (reference: SourceCode/include/tracer/RayTracer.h:12-50,96-101):
//     explicit RayTracer(Scene&);  getCamera();  setCamera();
//     std::vector<std::vector<Color>> render(const std::string& pathToImage, RenderOptions = RenderOptions());
//     void exportPPM(const std::string&, const std::vector<std::vector<Color>>&);
// The per-pixel work (renderRectangle -> getRay -> shootRay -> ...) runs on the MI355X through the C
// ABI of include/crt_hip.h; scheduling (the std::thread pools of RayTracer.cpp:114-202) is replaced by
// the GPU's own pixel queue, but WHICH pixels a render covers is still decided by the reference's bucket
// arithmetic (RayTracer.cpp:141-152), because that is observable in the output.
#pragma once

#include <string>
#include <vector>

#include "../../include/crt_hip.h"
#include "AccelerationStructure.h"
#include "Scene.h"

namespace crt {

enum RenderOptimization {  // RayTracer.h:12-23
  NoOptimization,
  Regions,
  BucketsThreadPool,
  BucketsQueue,
  AABB,
  BucketsThreadPoolAABB,
  BucketsQueueAABB,
  BVH,
  BVHBucketsThreadPool,
  BVHBucketsQueue
};

struct RenderOptions {  // RayTracer.h:25-50, same fields, defaults and constructor order
  RenderOptimization optimization = BVHBucketsThreadPool;
  bool USE_GI = false;
  unsigned int MAX_DEPTH = 5;
  unsigned int GI_SAMPLE_SIZE = 2;
  unsigned int RAYS_PER_PIXEL = 1;
  float SHADOW_BIAS = 1e-4;
  float REFLECTION_BIAS = 1e-4;
  float REFRACTION_BIAS = 1e-4;
  float MONTE_CARLO_BIAS = 1e-4;

  explicit RenderOptions(const RenderOptimization optimization = BVHBucketsThreadPool, const unsigned int maxDepth = 5,
                         const bool useGI = false, const unsigned int sampleSize = 2, const unsigned int raysPerPixel = 1,
                         const float shadowBias = 1e-4, const float reflectionBias = 1e-4,
                         const float refractionBias = 1e-4, const float monteCarloBias = 1e-4)
      : optimization{optimization}, USE_GI{useGI}, MAX_DEPTH{maxDepth}, GI_SAMPLE_SIZE{sampleSize},
        RAYS_PER_PIXEL{raysPerPixel}, SHADOW_BIAS{shadowBias}, REFLECTION_BIAS{reflectionBias},
        REFRACTION_BIAS{refractionBias}, MONTE_CARLO_BIAS{monteCarloBias} {}
};

// The rectangles RayTracer::render hands to renderRectangle for a given optimisation mode
// (RayTracer.cpp:209-286 + renderRegions :114-139 / renderBucketsThreadpool :141-158 / renderBucketsQueue
// :160-202; the queue variant shuffles the same set).  `hardwareConcurrency` stands for
// std::thread::hardware_concurrency() (used by the Regions mode only).
std::vector<crt_rect> bucketRectangles(unsigned int width, unsigned int height, unsigned int bucketSize,
                                       RenderOptimization optimization, unsigned int hardwareConcurrency);

// P3 writer, byte-identical to RayTracer::exportPPM + PPMColor (RayTracer.cpp:540-552, Color.cpp:12-21).
void writePPM(const std::string &path, const float *rgb, unsigned int width, unsigned int height);
void writePPMQuantized(const std::string &path, const uint8_t *rgb8, unsigned int width, unsigned int height);

class RayTracer {
 public:
  // Builds the two-level tree once (as RayTracer::RayTracer does, RayTracer.cpp:45-51), flattens it and
  // uploads scene + tree to the GPU.  Throws std::runtime_error when no usable GPU exists: there is no
  // CPU fallback.
  explicit RayTracer(Scene &scene, int device = 0, const crt_tuning *tuning = nullptr);
  // The same on several GPUs of one node: the frame's 8x8 tiles are dealt over `devices`, gathered on devices[0]
  // (crt_hip.h: crt_multi).  This is what replaces the reference's thread pool over buckets (RayTracer.cpp:141-158).
  RayTracer(Scene &scene, const std::vector<int> &devices, const crt_tuning *tuning = nullptr);
  ~RayTracer();
  RayTracer(const RayTracer &) = delete;
  RayTracer &operator=(const RayTracer &) = delete;

  const Camera &getCamera() const { return camera; }
  Camera &setCamera() { return camera; }
  std::vector<std::vector<Color>> render(const std::string &pathToImage, RenderOptions renderOptions = RenderOptions());
  void exportPPM(const std::string &pathToImage, const std::vector<std::vector<Color>> &colorBuffer);

  // ---- frames in flight (animation: app/animation.cpp:24-38 renders frame after frame with a new camera each).
  // setFramesInFlight(k) keeps the scene resident in k contexts; renderAsync() enqueues a frame with the CURRENT camera on
  // the next one and returns its slot at once; finishFrame(slot) waits for that frame and returns its quantised pixels
  // (H*W*3 bytes, PPMColor rule, valid until the slot is used again).  The GPU then works on frame k+1's primary rays while
  // frame k's deepest recursion levels and its copy to the host drain.  render() is unaffected.
  // USE_GI frames (RayTracer.cpp:90-104, 331-354): the reference seeds its generator from clock() ^ thread id, so no two of its
  // renders agree; here frame k of a tracer uses seed k (csrc/gi_random.h) unless setGISeed() says otherwise
  void setGISeed(unsigned int seed) { giSeed = seed; }
  void setFramesInFlight(unsigned int k);
  unsigned int framesInFlight() const { return (unsigned int)ring.size(); }
  int renderAsync(const RenderOptions &renderOptions);
  const uint8_t *finishFrame(int slot);

  // flat access for callers that do not want the vector-of-vectors copy
  int renderFlat(const std::string &pathToImage, const RenderOptions &renderOptions, float *outRGB, unsigned int counters = 0);
  crt_ctx *context() const { return multi ? crt_multi_context(multi, 0) : ctx; }
  crt_multi *multiContext() const { return multi; }
  const FlatScene &flatScene() const { return flat; }
  const AccelerationStructure &acceleration() const { return accelerationStructure; }
  crt_stats stats() const;
  // What the last render did differently from the reference, in words ("" when nothing): the reference's seven non-tree
  // RenderOptimization modes (brute force, single AABB) accept a handful of hits its tree modes drop; every mode renders
  // with the TREE's semantics here, the mode only selects the pixel coverage -- a caller that asks for one of them is told so.
  const std::string &renderNote() const { return note; }

 private:
  std::string note;
  const AccelerationStructure accelerationStructure;
  const Scene &scene;
  Camera camera;
  FlatScene flat;
  crt_ctx *ctx = nullptr;
  crt_multi *multi = nullptr;  // set instead of ctx when the tracer was built for several devices
  struct InFlight { crt_ctx *ctx = nullptr; uint8_t *rgb8 = nullptr; bool busy = false; };
  std::vector<InFlight> ring;  // ring[0].ctx == ctx
  unsigned int ringNext = 0;
  unsigned int giSeed = 0;
  int deviceIndex = 0;
  crt_tuning tuningCopy{};
  bool haveTuning = false;
  std::vector<float> frame;  // persistent colorBuffer (RayTracer.h:69)
};

}  // namespace crt
