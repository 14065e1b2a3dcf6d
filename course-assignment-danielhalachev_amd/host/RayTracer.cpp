#include "RayTracer.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace crt {

std::vector<crt_rect> bucketRectangles(unsigned int width, unsigned int height, unsigned int bucketSize,
                                       RenderOptimization optimization, unsigned int hardwareConcurrency) {
  // RayTracer::render's switch (RayTracer.cpp:209-286); both counters are `unsigned short` members
  // (RayTracer.h:74-75), so the values wrap at 65536 like the reference's do.
  unsigned short threadCount = 1, rectangleCount = 1;
  bool regions = true;
  switch (optimization) {
    case NoOptimization: case AABB: case BVH:
      break;
    case Regions:
      threadCount = (unsigned short)hardwareConcurrency;
      rectangleCount = (unsigned short)hardwareConcurrency;
      break;
    default:  // the six bucket modes
      threadCount = (unsigned short)hardwareConcurrency;
      rectangleCount = (unsigned short)bucketSize;
      regions = false;
      break;
  }
  std::vector<crt_rect> rects;
  unsigned int threadNumY = static_cast<unsigned int>(std::sqrt(rectangleCount));  // RayTracer.cpp:115,143
  if (threadNumY == 0) threadNumY = 1;
  unsigned int threadNumX = rectangleCount / threadNumY;
  if (threadNumX == 0 || width == 0 || height == 0) return rects;  // the reference divides by zero here
  unsigned int regionWidth = width / threadNumX;
  unsigned int regionHeight = height / threadNumY;
  if (regions && threadCount == 1) {  // RayTracer.cpp:123-126
    rects.push_back(crt_rect{0, 0, regionWidth, regionHeight});
    return rects;
  }
  const int count = regions ? threadCount : rectangleCount;  // RayTracer.cpp:130 vs :149
  for (int i = 0; i < count; i++) {
    unsigned column = (i * regionWidth) % width;
    unsigned row = (i / threadNumX) * regionHeight;
    rects.push_back(crt_rect{row, column, regionWidth, regionHeight});
  }
  return rects;
}

// ------------------------------------------------------------------------------------------------ PPM
namespace {
struct DecimalTable {
  char text[256][4];
  unsigned char len[256];
  DecimalTable() {
    for (int i = 0; i < 256; i++) len[i] = (unsigned char)snprintf(text[i], 4, "%d", i);
  }
};
const DecimalTable &decimals() {
  static const DecimalTable t;
  return t;
}
inline unsigned short quantise(float c) {  // PPMColor, Color.cpp:12-16: clamp, scale, TRUNCATE
  const float cl = (c < 0.0f) ? 0.0f : ((1.0f < c) ? 1.0f : c);
  return static_cast<unsigned short>(cl * 255);
}
template <typename Fetch>
void writePPMImpl(const std::string &path, unsigned int width, unsigned int height, Fetch fetch) {
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot open " + path + " for writing");
  // RayTracer.cpp:541-551: "P3\n{W} {H}\n255\n", then per row "{r} {g} {b}\t" per pixel and "\n"
  fprintf(f, "P3\n%u %u\n%d\n", width, height, 255);
  const DecimalTable &t = decimals();
  std::vector<char> line((size_t)width * 12 + 2);
  for (unsigned int row = 0; row < height; row++) {
    char *p = line.data();
    for (unsigned int col = 0; col < width; col++) {
      for (int k = 0; k < 3; k++) {
        const unsigned v = fetch(((size_t)row * width + col) * 3 + k);
        memcpy(p, t.text[v], t.len[v]);
        p += t.len[v];
        *p++ = (k == 2) ? '\t' : ' ';
      }
    }
    *p++ = '\n';
    fwrite(line.data(), 1, (size_t)(p - line.data()), f);
  }
  fclose(f);
}
}  // namespace

void writePPM(const std::string &path, const float *rgb, unsigned int width, unsigned int height) {
  writePPMImpl(path, width, height, [rgb](size_t i) -> unsigned { return quantise(rgb[i]); });
}

void writePPMQuantized(const std::string &path, const uint8_t *rgb8, unsigned int width, unsigned int height) {
  writePPMImpl(path, width, height, [rgb8](size_t i) -> unsigned { return rgb8[i]; });
}

// ------------------------------------------------------------------------------------------------ RayTracer
RayTracer::RayTracer(Scene &scene, int device, const crt_tuning *tuning)
    : accelerationStructure(scene), scene(scene), camera(scene.camera) {
  flattenScene(scene, accelerationStructure, flat);
  int rc = crt_create_tuned(&flat.desc, device, tuning, &ctx);
  if (rc != CRT_OK) throw std::runtime_error(std::string("crt_create failed: ") + crt_last_error(nullptr));
  deviceIndex = device;
  if (tuning) { tuningCopy = *tuning; haveTuning = true; }
  {  // the context starts with the scene's camera, as RayTracer::RayTracer copies scene.camera (RayTracer.cpp:46)
    const float pos[3] = {camera.getPosition().x, camera.getPosition().y, camera.getPosition().z};
    crt_set_camera(ctx, pos, &camera.getRotationMatrix().m[0][0]);
  }
  frame.assign((size_t)scene.sceneSettings.image.width * scene.sceneSettings.image.height * 3, 0.0f);
}

RayTracer::RayTracer(Scene &scene, const std::vector<int> &devices, const crt_tuning *tuning)
    : accelerationStructure(scene), scene(scene), camera(scene.camera) {
  flattenScene(scene, accelerationStructure, flat);
  int rc = crt_multi_create(&flat.desc, devices.data(), (uint32_t)devices.size(), tuning, &multi);
  if (rc != CRT_OK) throw std::runtime_error(std::string("crt_multi_create failed: ") + crt_multi_last_error(nullptr));
  const float pos[3] = {camera.getPosition().x, camera.getPosition().y, camera.getPosition().z};
  crt_multi_set_camera(multi, pos, &camera.getRotationMatrix().m[0][0]);
  frame.assign((size_t)scene.sceneSettings.image.width * scene.sceneSettings.image.height * 3, 0.0f);
}

RayTracer::~RayTracer() {
  for (size_t i = 0; i < ring.size(); i++) {
    if (ring[i].busy) crt_wait(ring[i].ctx);
    crt_free_pinned(ring[i].rgb8);
    if (i > 0) crt_destroy(ring[i].ctx);
  }
  crt_destroy(ctx);
  crt_multi_destroy(multi);
}

void RayTracer::setFramesInFlight(unsigned int k) {
  if (multi) throw std::runtime_error("frames in flight: not available on a multi-device tracer");
  if (k < 1) k = 1;
  if (k > 8) k = 8;
  const size_t bytes = (size_t)scene.sceneSettings.image.width * scene.sceneSettings.image.height * 3;
  while (ring.size() < k) {
    InFlight f;
    if (ring.empty()) f.ctx = ctx;
    else if (crt_create_tuned(&flat.desc, deviceIndex, haveTuning ? &tuningCopy : nullptr, &f.ctx) != CRT_OK)
      throw std::runtime_error(std::string("crt_create failed: ") + crt_last_error(nullptr));
    f.rgb8 = static_cast<uint8_t *>(crt_alloc_pinned(bytes));
    if (!f.rgb8) { if (!ring.empty()) crt_destroy(f.ctx); throw std::runtime_error("out of pinned host memory"); }
    ring.push_back(f);
  }
}

int RayTracer::renderAsync(const RenderOptions &ro) {
  if (ring.empty()) setFramesInFlight(2);
  const int slot = (int)(ringNext % ring.size());
  ringNext++;
  InFlight &f = ring[(size_t)slot];
  if (f.busy) { crt_wait(f.ctx); f.busy = false; }  // the slot's previous frame (its pixels are overwritten from here on)
  crt_options o{};
  o.max_depth = ro.MAX_DEPTH;
  o.shadow_bias = ro.SHADOW_BIAS;
  o.reflection_bias = ro.REFLECTION_BIAS;
  o.refraction_bias = ro.REFRACTION_BIAS;
  o.use_gi = ro.USE_GI ? 1u : 0u;
  o.gi_sample_size = ro.GI_SAMPLE_SIZE;
  o.rays_per_pixel = ro.RAYS_PER_PIXEL;
  o.monte_carlo_bias = ro.MONTE_CARLO_BIAS;
  o.gi_seed = ro.USE_GI ? giSeed++ : 0u;  // every GI frame its own seed, as every GI render of the reference differs
  const Matrix3 &m = camera.getRotationMatrix();
  const float pos[3] = {camera.getPosition().x, camera.getPosition().y, camera.getPosition().z};
  int rc = crt_set_camera(f.ctx, pos, &m.m[0][0]);
  const unsigned int W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
  std::vector<crt_rect> rects = bucketRectangles(W, H, scene.sceneSettings.bucketSize, ro.optimization, std::thread::hardware_concurrency());
  if (rc == CRT_OK) rc = crt_render_async(f.ctx, &o, rects.data(), (uint32_t)rects.size(), nullptr, f.rgb8);
  if (rc != CRT_OK) throw std::runtime_error(std::string("renderAsync failed: ") + crt_last_error(f.ctx));
  f.busy = true;
  return slot;
}

const uint8_t *RayTracer::finishFrame(int slot) {
  if (slot < 0 || (size_t)slot >= ring.size()) throw std::runtime_error("finishFrame: no such slot");
  InFlight &f = ring[(size_t)slot];
  if (f.busy) {
    if (crt_wait(f.ctx) != CRT_OK) throw std::runtime_error(std::string("finishFrame failed: ") + crt_last_error(f.ctx));
    f.busy = false;
  }
  return f.rgb8;
}

crt_stats RayTracer::stats() const {
  crt_stats s{};
  if (multi) crt_multi_get_stats(multi, &s);
  else crt_get_stats(ctx, &s);
  return s;
}

int RayTracer::renderFlat(const std::string &pathToImage, const RenderOptions &ro, float *outRGB, unsigned int counters) {
  const unsigned int W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
  crt_options o{};
  o.max_depth = ro.MAX_DEPTH;
  o.shadow_bias = ro.SHADOW_BIAS;
  o.reflection_bias = ro.REFLECTION_BIAS;
  o.refraction_bias = ro.REFRACTION_BIAS;
  o.use_gi = ro.USE_GI ? 1u : 0u;
  o.gi_sample_size = ro.GI_SAMPLE_SIZE;
  o.rays_per_pixel = ro.RAYS_PER_PIXEL;
  o.monte_carlo_bias = ro.MONTE_CARLO_BIAS;
  o.gi_seed = ro.USE_GI ? giSeed++ : 0u;  // every GI frame its own seed, as every GI render of the reference differs
  o.collect_counters = counters;  // 0, 1 (counting build) or 2 (production kernels with tallies), see crt_hip.h
  const Matrix3 &m = camera.getRotationMatrix();
  const float pos[3] = {camera.getPosition().x, camera.getPosition().y, camera.getPosition().z};
  int rc = multi ? crt_multi_set_camera(multi, pos, &m.m[0][0]) : crt_set_camera(ctx, pos, &m.m[0][0]);
  if (rc) return rc;
  // All ten modes render with the tree's semantics: the three BVH* modes are pixel-identical in the
  // reference, the non-tree modes differ from them in a handful of pixels and are not
  // part of this path.  The mode still selects the pixel coverage.
  note = ro.optimization < BVH ? "RenderOptimization " + std::to_string((int)ro.optimization) + " is rendered with the tree modes' semantics (the reference's brute-force and "
                                 "single-box modes differ from its tree modes in a handful of pixels); the mode selects the pixel coverage only" : std::string();
  std::vector<crt_rect> rects = bucketRectangles(W, H, scene.sceneSettings.bucketSize, ro.optimization,
                                                 std::thread::hardware_concurrency());
  rc = multi ? crt_multi_render(multi, &o, rects.data(), (uint32_t)rects.size(), outRGB)
             : crt_render(ctx, &o, rects.data(), (uint32_t)rects.size(), outRGB);
  if (rc) return rc;
  if (!pathToImage.empty()) {  // RayTracer.cpp:294-296; quantised on the device with the same rule
    std::vector<uint8_t> q((size_t)W * H * 3);
    rc = multi ? crt_multi_read_quantized(multi, q.data()) : crt_read_quantized(ctx, q.data());
    if (rc) return rc;
    writePPMQuantized(pathToImage, q.data(), W, H);
  }
  return CRT_OK;
}

std::vector<std::vector<Color>> RayTracer::render(const std::string &pathToImage, RenderOptions ro) {
  const unsigned int W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
  int rc = renderFlat(pathToImage, ro, frame.data());
  if (rc != CRT_OK) throw std::runtime_error(std::string("render failed: ") + (multi ? crt_multi_last_error(multi) : crt_last_error(ctx)));
  std::vector<std::vector<Color>> colorBuffer(H, std::vector<Color>(W));
  for (unsigned int y = 0; y < H; y++)
    for (unsigned int x = 0; x < W; x++) {
      const float *p = &frame[((size_t)y * W + x) * 3];
      colorBuffer[y][x] = Color(p[0], p[1], p[2]);
    }
  return colorBuffer;
}

void RayTracer::exportPPM(const std::string &pathToImage, const std::vector<std::vector<Color>> &colorBuffer) {
  const unsigned int W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
  std::vector<float> flatRGB((size_t)W * H * 3, 0.0f);
  for (unsigned int y = 0; y < H && y < colorBuffer.size(); y++)
    for (unsigned int x = 0; x < W && x < colorBuffer[y].size(); x++) {
      flatRGB[((size_t)y * W + x) * 3 + 0] = colorBuffer[y][x].x;
      flatRGB[((size_t)y * W + x) * 3 + 1] = colorBuffer[y][x].y;
      flatRGB[((size_t)y * W + x) * 3 + 2] = colorBuffer[y][x].z;
    }
  writePPM(pathToImage, flatRGB.data(), W, H);
}

}  // namespace crt
