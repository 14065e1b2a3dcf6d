// TEST INFRASTRUCTURE -- not part of the product, never shipped, never on the product path.
//
// Driver for the REAL reference, compiled where its sources lie under /root/reference by
// oracle/Makefile into oracle/_ref/ (git-ignored).  It replaces only what cannot be built in this
// image: the reference's app/main.cpp hard-codes /home/daniel paths (app/main.cpp:12,16) and its
// SceneParser needs RapidJSON, which the image lacks (cmake/FindRapidJSON.cmake:3-17 fetches it
// from the network).  The reference's Scene/Mesh/Material/Light/Camera/Texture types are all
// publicly constructible (Scene.h:50-69, Material.h:15-24, Texture.h:24-58), so this file fills a
// `Scene` from a CRTS blob (oracle/scene_blob.h) and calls the reference's own
// `RayTracer::render` (RayTracer.cpp:204-298).  Nothing of the hot path is restated here.
//
// usage: ref_render <scene.crts> <out.f32> [--depth N] [--mode NAME] [--ppm out.ppm] [--repeat K] [--gi SAMPLES RAYS_PER_PIXEL] [--all-frames]
//   out.f32 = H*W*3 little-endian float32, row 0 = top  (the reference's colorBuffer)
//   prints one JSON line: {"render_s": ..., "build_s": ..., "width": W, "height": H, ...}
// usage: ref_render --camera-ops <ops.txt> <out.txt>
//   drives the reference's own Camera (Camera.cpp:33-70).  ops.txt: first line = position (3) and row-major matrix (9)
//   as hexadecimal float bit patterns; every further line = "<truck|pan|tilt|roll> <3 bit patterns>", applied one after the
//   other to the same camera.  out.txt: the 12 bit patterns of the camera after every operation, one line each.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "tracer/RayTracer.h"
#include "tracer/Scene.h"

#include "scene_blob.h"

static RenderOptimization parseMode(const std::string &s) {
  if (s == "none") return NoOptimization;
  if (s == "regions") return Regions;
  if (s == "pool") return BucketsThreadPool;
  if (s == "queue") return BucketsQueue;
  if (s == "aabb") return AABB;
  if (s == "aabbpool") return BucketsThreadPoolAABB;
  if (s == "aabbqueue") return BucketsQueueAABB;
  if (s == "bvh") return BVH;
  if (s == "bvhpool") return BVHBucketsThreadPool;
  if (s == "bvhqueue") return BVHBucketsQueue;
  std::fprintf(stderr, "unknown mode %s\n", s.c_str());
  std::exit(2);
}

static float bitsToFloat(const std::string &hex) {
  const uint32_t u = (uint32_t)std::strtoul(hex.c_str(), nullptr, 16);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

static int cameraOps(const char *opsPath, const char *outPath) {
  std::ifstream in(opsPath);
  std::ofstream out(outPath);
  std::string tok;
  float v[12];
  for (int i = 0; i < 12; i++) {
    if (!(in >> tok)) return 2;
    v[i] = bitsToFloat(tok);
  }
  Camera camera(Vector(v[0], v[1], v[2]));
  camera.setRotationMatrix() = Matrix<3>(std::vector<float>(v + 3, v + 12));
  std::string op;
  while (in >> op) {
    float a[3];
    for (int i = 0; i < 3; i++) {
      if (!(in >> tok)) return 2;
      a[i] = bitsToFloat(tok);
    }
    if (op == "truck") camera.truck(Vector(a[0], a[1], a[2]));
    else if (op == "pan") camera.pan(a[0]);
    else if (op == "tilt") camera.tilt(a[0]);
    else if (op == "roll") camera.roll(a[0]);
    else return 2;
    float r[12] = {camera.getPosition()[0], camera.getPosition()[1], camera.getPosition()[2]};
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r[3 + 3 * i + j] = camera.getRotationMatrix()[i][j];
    char line[160];
    int n = 0;
    for (int i = 0; i < 12; i++) {
      uint32_t u;
      std::memcpy(&u, &r[i], 4);
      n += std::snprintf(line + n, sizeof(line) - n, "%08x%c", u, i == 11 ? '\n' : ' ');
    }
    out << line;
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc == 4 && std::string(argv[1]) == "--camera-ops") return cameraOps(argv[2], argv[3]);
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s scene.crts out.f32 [--depth N] [--mode NAME] [--ppm P] [--repeat K] [--gi SAMPLES RAYS_PER_PIXEL] [--all-frames]\n", argv[0]);
    return 2;
  }
  std::string blobPath = argv[1], outPath = argv[2], ppmPath, mode = "bvhpool";
  unsigned depth = 5;
  int repeat = 1;
  bool useGI = false, allFrames = false;
  unsigned giSamples = 2, raysPerPixel = 1;
  for (int i = 3; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--depth" && i + 1 < argc) depth = std::atoi(argv[++i]);
    else if (a == "--mode" && i + 1 < argc) mode = argv[++i];
    else if (a == "--ppm" && i + 1 < argc) ppmPath = argv[++i];
    else if (a == "--repeat" && i + 1 < argc) repeat = std::atoi(argv[++i]);
    else if (a == "--gi" && i + 2 < argc) { useGI = true; giSamples = (unsigned)std::atoi(argv[++i]); raysPerPixel = (unsigned)std::atoi(argv[++i]); }
    else if (a == "--all-frames") allFrames = true;
  }

  std::ifstream in(blobPath, std::ios::binary);
  std::vector<char> data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  blob_cursor c;
  blob_init(&c, data.data(), data.size());
  const void *magic = blob_take(&c, 4);
  if (!magic || std::memcmp(magic, "CRTS", 4) != 0 || blob_u32(&c) != 1) {
    std::fprintf(stderr, "bad blob\n");
    return 2;
  }

  Scene scene;
  scene.sceneSettings.image.width = blob_u32(&c);
  scene.sceneSettings.image.height = blob_u32(&c);
  scene.sceneSettings.bucketSize = blob_u32(&c);
  float f[9];
  blob_f32v(&c, f, 3);
  scene.sceneSettings.sceneBackgroundColor = Color(f[0], f[1], f[2]);
  blob_f32v(&c, f, 3);
  scene.camera.setPosition() = Vector(f[0], f[1], f[2]);
  blob_f32v(&c, f, 9);
  scene.camera.setRotationMatrix() = Matrix<3>(std::vector<float>(f, f + 9));

  // ---- textures
  struct TexRec { uint32_t kind; float a[3], b[3], s; uint32_t w, h; std::string file; };
  std::vector<TexRec> texRecs;
  uint32_t nTex = blob_u32(&c);
  for (uint32_t t = 0; t < nTex; t++) {
    TexRec r;
    r.kind = blob_u32(&c);
    blob_f32v(&c, r.a, 3);
    blob_f32v(&c, r.b, 3);
    r.s = blob_f32(&c);
    r.w = blob_u32(&c);
    r.h = blob_u32(&c);
    if (r.kind == 3) {
      const void *px = blob_take(&c, (size_t)r.w * r.h * 3);
      r.file = outPath + ".tex" + std::to_string(t) + ".ppm";  // decoded by the reference's own stb_image
      std::ofstream o(r.file, std::ios::binary);
      o << "P6\n" << r.w << " " << r.h << "\n255\n";
      o.write((const char *)px, (std::streamsize)r.w * r.h * 3);
    }
    texRecs.push_back(r);
  }
#if (defined USE_TEXTURES) && USE_TEXTURES
  for (uint32_t t = 0; t < nTex; t++) {
    const TexRec &r = texRecs[t];
    std::string name = "tex" + std::to_string(t);
    Color a(r.a[0], r.a[1], r.a[2]), b(r.b[0], r.b[1], r.b[2]);
    if (r.kind == 0) scene.textures.push_back(new AlbedoTexture(name, a));
    else if (r.kind == 1) scene.textures.push_back(new EdgeTexture(name, a, b, r.s));
    else if (r.kind == 2) scene.textures.push_back(new CheckerTexture(name, a, b, r.s));
    else scene.textures.push_back(new BitmapTexture(name, r.file));
  }
#else
  if (nTex != 0) {
    std::fprintf(stderr, "scene has textures: use the USE_TEXTURES build (ref_render_tex)\n");
    return 2;
  }
#endif

  // ---- materials (reserve first: Mesh keeps `const Material&`, Scene.h:27)
  uint32_t nMat = blob_u32(&c);
  scene.materials.reserve(nMat);
  for (uint32_t m = 0; m < nMat; m++) {
    uint32_t type = blob_u32(&c);
    blob_f32v(&c, f, 3);
    uint32_t smooth = blob_u32(&c);
    float ior = blob_f32(&c);
    int32_t tex = blob_i32(&c);
    Albedo albedo(f[0], f[1], f[2]);
#if (defined USE_TEXTURES) && USE_TEXTURES
    const Texture *texture;
    if (tex >= 0) {
      texture = scene.textures[tex];
    } else {  // constant albedo expressed the way the textured build does it (Texture.cpp:14-16)
      scene.textures.push_back(new AlbedoTexture("albedo" + std::to_string(m), albedo));
      texture = scene.textures.back();
    }
    scene.materials.push_back(Material(*texture, albedo, (MaterialType)type, smooth != 0, ior));
#else
    (void)tex;
    scene.materials.push_back(Material(albedo, (MaterialType)type, smooth != 0, ior));
#endif
  }

  uint32_t nLights = blob_u32(&c);
  for (uint32_t l = 0; l < nLights; l++) {
    blob_f32v(&c, f, 3);
    uint32_t intensity = blob_u32(&c);
    scene.lights.push_back(Light{Vector(f[0], f[1], f[2]), intensity});
  }

  // ---- meshes (reserve + rvalue push_back: triangles point into their own mesh's vertices, Triangle.h:10)
  uint32_t nMesh = blob_u32(&c);
  scene.objects.reserve(nMesh);
  for (uint32_t m = 0; m < nMesh; m++) {
    uint32_t mat = blob_u32(&c), nv = blob_u32(&c), nt = blob_u32(&c), hasUV = blob_u32(&c);
    const float *pos = (const float *)blob_take(&c, (size_t)nv * 12);
    const float *uvs = hasUV ? (const float *)blob_take(&c, (size_t)nv * 12) : nullptr;
    const uint32_t *idx = (const uint32_t *)blob_take(&c, (size_t)nt * 12);
    if (!c.ok) break;
    std::vector<Vertex> vertices;
    vertices.reserve(nv);
    for (uint32_t v = 0; v < nv; v++) {
      vertices.push_back(Vertex(Vector(pos[3 * v], pos[3 * v + 1], pos[3 * v + 2])));
#if (defined USE_TEXTURES) && USE_TEXTURES
      if (uvs) vertices.back().UV = Vector(uvs[3 * v], uvs[3 * v + 1], uvs[3 * v + 2]);
#endif
    }
    (void)uvs;
    std::vector<unsigned int> triples(idx, idx + (size_t)nt * 3);
    scene.objects.push_back(Mesh{scene.materials[mat], vertices, triples});
  }
  if (!c.ok) {
    std::fprintf(stderr, "truncated blob\n");
    return 2;
  }

  const unsigned W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
  auto t0 = std::chrono::high_resolution_clock::now();
  RayTracer tracer(scene);
  auto t1 = std::chrono::high_resolution_clock::now();
  // --gi N R: the GI / multi-sample mode (RayTracer.h:27-30) -- every render then differs (the reference seeds its generator
  // from clock() ^ thread id), so --all-frames writes each of the --repeat frames, for statistics over them
  RenderOptions options(parseMode(mode), depth, useGI, giSamples, raysPerPixel);
  std::vector<std::vector<Color>> buffer;
  double best = 1e30, total = 0;
  std::ofstream out(outPath, std::ios::binary);
  std::vector<float> flat((size_t)W * H * 3);
  for (int r = 0; r < repeat; r++) {
    auto a = std::chrono::high_resolution_clock::now();
    buffer = tracer.render(r == repeat - 1 ? ppmPath : std::string(), options);
    auto b = std::chrono::high_resolution_clock::now();
    double s = std::chrono::duration<double>(b - a).count();
    total += s;
    if (s < best) best = s;
    if (allFrames || r == repeat - 1) {
      for (unsigned y = 0; y < H; y++)
        for (unsigned x = 0; x < W; x++)
          for (unsigned k = 0; k < 3; k++) flat[((size_t)y * W + x) * 3 + k] = buffer[y][x][k];
      out.write((const char *)flat.data(), (std::streamsize)flat.size() * 4);
    }
  }
  for (auto &r : texRecs)
    if (!r.file.empty()) std::remove(r.file.c_str());

  std::printf("\n{\"render_s\": %.6f, \"render_s_mean\": %.6f, \"build_s\": %.6f, \"width\": %u, \"height\": %u, "
              "\"depth\": %u, \"mode\": \"%s\", \"threads\": %u, \"repeat\": %d}\n",
              best, total / repeat, std::chrono::duration<double>(t1 - t0).count(), W, H, depth, mode.c_str(),
              std::thread::hardware_concurrency(), repeat);
  return 0;
}
