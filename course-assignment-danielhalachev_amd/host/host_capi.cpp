// extern "C" wrappers of include/crt_host.h over the C++ host layer.  No exception leaves this file.
#include <chrono>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <thread>

#include "../../include/crt_host.h"
#include "AccelerationStructure.h"
#include "RayTracer.h"
#include "Scene.h"
#include "SceneParser.h"

struct crt_host_scene {
  double build_seconds = 0;  // tree build alone (AccelerationStructure's constructor)
  crt::Scene scene;
  std::unique_ptr<crt::AccelerationStructure> accel;
  crt::FlatScene flat;
};

struct crt_host_tracer {
  crt_host_scene *scene = nullptr;
  std::unique_ptr<crt::RayTracer> tracer;
};

static thread_local std::string g_error;

template <typename F>
static int guarded(F &&f) {
  try {
    return f();
  } catch (const crt::SceneParseError &e) {
    g_error = e.what();
    return CRT_ERR_PARSE;
  } catch (const std::bad_alloc &) {
    g_error = "out of memory";
    return CRT_ERR_NOMEM;
  } catch (const std::exception &e) {
    g_error = e.what();
    return CRT_ERR_INVALID;
  } catch (...) {
    g_error = "unknown error";
    return CRT_ERR_INVALID;
  }
}

static int finishScene(std::unique_ptr<crt_host_scene> hs, crt_host_scene **out, int build_device = -1) {
  const auto t0 = std::chrono::steady_clock::now();
  hs->accel.reset(new crt::AccelerationStructure(hs->scene, build_device));
  hs->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  crt::flattenScene(hs->scene, *hs->accel, hs->flat);
  *out = hs.release();
  return CRT_OK;
}

extern "C" const char *crt_host_last_error(void) { return g_error.c_str(); }

extern "C" int crt_host_scene_parse_file(const char *path, const char *folder, crt_host_scene **out) {
  if (!path || !out) return CRT_ERR_INVALID;
  *out = nullptr;
  return guarded([&]() {
    std::unique_ptr<crt_host_scene> hs(new crt_host_scene());
    crt::SceneParser parser;
    hs->scene = parser.parseScene(path, folder ? folder : "");
    return finishScene(std::move(hs), out);
  });
}

extern "C" int crt_host_scene_parse_text(const char *json, size_t length, const char *folder, crt_host_scene **out) {
  if (!json || !out) return CRT_ERR_INVALID;
  *out = nullptr;
  return guarded([&]() {
    std::unique_ptr<crt_host_scene> hs(new crt_host_scene());
    crt::SceneParser parser;
    hs->scene = parser.parseSceneText(std::string(json, length), folder ? folder : "");
    return finishScene(std::move(hs), out);
  });
}

extern "C" int crt_host_scene_parse_text_ex(const char *json, size_t length, const char *folder, int build_device, crt_host_scene **out) {
  if (!json || !out) return CRT_ERR_INVALID;
  *out = nullptr;
  return guarded([&]() {
    std::unique_ptr<crt_host_scene> hs(new crt_host_scene());
    crt::SceneParser parser;
    hs->scene = parser.parseSceneText(std::string(json, length), folder ? folder : "");
    return finishScene(std::move(hs), out, build_device);
  });
}

extern "C" double crt_host_scene_build_seconds(const crt_host_scene *scene) { return scene ? scene->build_seconds : 0.0; }

extern "C" void crt_host_scene_free(crt_host_scene *scene) { delete scene; }

extern "C" const crt_scene_desc *crt_host_scene_desc(const crt_host_scene *scene) { return scene ? &scene->flat.desc : nullptr; }

extern "C" void crt_host_scene_settings(const crt_host_scene *s, uint32_t *width, uint32_t *height, uint32_t *bucket) {
  if (width) *width = s->scene.sceneSettings.image.width;
  if (height) *height = s->scene.sceneSettings.image.height;
  if (bucket) *bucket = s->scene.sceneSettings.bucketSize;
}

extern "C" void crt_host_scene_camera(const crt_host_scene *s, float position[3], float matrix[9]) {
  const crt::Vector &p = s->scene.camera.getPosition();
  position[0] = p.x; position[1] = p.y; position[2] = p.z;
  memcpy(matrix, &s->scene.camera.getRotationMatrix().m[0][0], 9 * sizeof(float));
}

extern "C" uint32_t crt_host_scene_mesh_count(const crt_host_scene *s) { return (uint32_t)s->scene.objects.size(); }

static const crt::KDTree &pickTree(const crt_host_scene *s, int mesh) {
  return mesh < 0 ? s->accel->objectTree : s->accel->meshTrees[(size_t)mesh];
}

extern "C" uint32_t crt_host_tree_node_count(const crt_host_scene *s, int mesh) { return (uint32_t)pickTree(s, mesh).nodes.size(); }

extern "C" uint64_t crt_host_tree_index_total(const crt_host_scene *s, int mesh) {
  uint64_t n = 0;
  for (auto &node : pickTree(s, mesh).nodes) n += node.indexes.size();
  return n;
}

extern "C" void crt_host_tree_dump(const crt_host_scene *s, int mesh, float *boxes, uint32_t *links, uint32_t *indexes) {
  const crt::KDTree &t = pickTree(s, mesh);
  size_t k = 0;
  for (size_t i = 0; i < t.nodes.size(); i++) {
    const auto &n = t.nodes[i];
    for (unsigned short a = 0; a < 3; a++) { boxes[6 * i + a] = n.box.minPoint[a]; boxes[6 * i + 3 + a] = n.box.maxPoint[a]; }
    links[4 * i] = n.children[0]; links[4 * i + 1] = n.children[1]; links[4 * i + 2] = n.parent;
    links[4 * i + 3] = (uint32_t)n.indexes.size();
    for (uint32_t e : n.indexes) indexes[k++] = e;
  }
}

extern "C" void crt_host_mesh_sizes(const crt_host_scene *s, uint32_t mesh, uint32_t *nv, uint32_t *nt) {
  *nv = (uint32_t)s->scene.objects[mesh].vertices.size();
  *nt = (uint32_t)s->scene.objects[mesh].triangles.size();
}

extern "C" void crt_host_mesh_normals(const crt_host_scene *s, uint32_t mesh, float *face, float *vertex) {
  const crt::Mesh &m = s->scene.objects[mesh];
  for (size_t i = 0; i < m.triangles.size(); i++) { face[3 * i] = m.triangles[i].normal.x; face[3 * i + 1] = m.triangles[i].normal.y; face[3 * i + 2] = m.triangles[i].normal.z; }
  for (size_t i = 0; i < m.vertices.size(); i++) { vertex[3 * i] = m.vertices[i].normal.x; vertex[3 * i + 1] = m.vertices[i].normal.y; vertex[3 * i + 2] = m.vertices[i].normal.z; }
}

extern "C" uint32_t crt_host_bucket_rects(uint32_t width, uint32_t height, uint32_t bucket_size, int optimization,
                                          uint32_t hardware_concurrency, crt_rect *out, uint32_t max_rects) {
  if (optimization < 0 || optimization > CRT_OPT_BVH_BUCKETS_QUEUE) return 0;
  std::vector<crt_rect> r = crt::bucketRectangles(width, height, bucket_size, (crt::RenderOptimization)optimization,
                                                  hardware_concurrency ? hardware_concurrency : std::thread::hardware_concurrency());
  uint32_t n = (uint32_t)r.size() < max_rects ? (uint32_t)r.size() : max_rects;
  if (out && n) memcpy(out, r.data(), n * sizeof(crt_rect));
  return (uint32_t)r.size();
}

extern "C" int crt_host_camera_apply(float position[3], float matrix[9], int op, const float v[3]) {
  if (!position || !matrix || !v) return CRT_ERR_INVALID;
  crt::Camera cam(crt::Vector(position[0], position[1], position[2]));
  memcpy(&cam.setRotationMatrix().m[0][0], matrix, 9 * sizeof(float));
  switch (op) {
    case 0: cam.truck(crt::Vector(v[0], v[1], v[2])); break;
    case 1: cam.pan(v[0]); break;
    case 2: cam.tilt(v[0]); break;
    case 3: cam.roll(v[0]); break;
    default: return CRT_ERR_INVALID;
  }
  position[0] = cam.getPosition().x; position[1] = cam.getPosition().y; position[2] = cam.getPosition().z;
  memcpy(matrix, &cam.getRotationMatrix().m[0][0], 9 * sizeof(float));
  return CRT_OK;
}

extern "C" int crt_host_tracer_create(crt_host_scene *scene, int device, crt_host_tracer **out) {
  return crt_host_tracer_create_tuned(scene, device, nullptr, out);
}

extern "C" int crt_host_tracer_create_tuned(crt_host_scene *scene, int device, const crt_tuning *tuning, crt_host_tracer **out) {
  if (!scene || !out) return CRT_ERR_INVALID;
  *out = nullptr;
  int ndev = crt_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) {
    g_error = "no usable HIP device (this library has no CPU fallback)";
    return CRT_ERR_NO_DEVICE;
  }
  return guarded([&]() {
    std::unique_ptr<crt_host_tracer> t(new crt_host_tracer());
    t->scene = scene;
    t->tracer.reset(new crt::RayTracer(scene->scene, device, tuning));
    *out = t.release();
    return CRT_OK;
  });
}

extern "C" int crt_host_tracer_create_multi(crt_host_scene *scene, const int *devices, uint32_t n_devices, const crt_tuning *tuning,
                                            crt_host_tracer **out) {
  if (!scene || !out || !devices || n_devices == 0) return CRT_ERR_INVALID;
  *out = nullptr;
  int ndev = crt_device_count();
  for (uint32_t i = 0; i < n_devices; i++)
    if (ndev <= 0 || devices[i] < 0 || devices[i] >= ndev) {
      g_error = "no usable HIP device (this library has no CPU fallback)";
      return CRT_ERR_NO_DEVICE;
    }
  return guarded([&]() {
    std::unique_ptr<crt_host_tracer> t(new crt_host_tracer());
    t->scene = scene;
    t->tracer.reset(new crt::RayTracer(scene->scene, std::vector<int>(devices, devices + n_devices), tuning));
    *out = t.release();
    return CRT_OK;
  });
}

extern "C" void crt_host_tracer_free(crt_host_tracer *tracer) { delete tracer; }

extern "C" int crt_host_tracer_set_camera(crt_host_tracer *t, const float position[3], const float matrix[9]) {
  if (!t || !position || !matrix) return CRT_ERR_INVALID;
  t->tracer->setCamera().setPosition() = crt::Vector(position[0], position[1], position[2]);
  memcpy(&t->tracer->setCamera().setRotationMatrix().m[0][0], matrix, 9 * sizeof(float));
  if (t->tracer->multiContext()) return crt_multi_set_camera(t->tracer->multiContext(), position, matrix);
  return crt_set_camera(t->tracer->context(), position, matrix);  // also for callers of the device-level API
}

extern "C" int crt_host_tracer_render(crt_host_tracer *t, const char *ppm_path, int optimization, const crt_options *o,
                                      float *out_rgb) {
  if (!t || !o || optimization < 0 || optimization > CRT_OPT_BVH_BUCKETS_QUEUE) return CRT_ERR_INVALID;
  return guarded([&]() {
    crt::RenderOptions ro((crt::RenderOptimization)optimization, o->max_depth, o->use_gi != 0, o->gi_sample_size, o->rays_per_pixel,
                          o->shadow_bias, o->reflection_bias, o->refraction_bias, o->monte_carlo_bias);
    if (o->use_gi) t->tracer->setGISeed(o->gi_seed);
    int rc = t->tracer->renderFlat(ppm_path ? ppm_path : "", ro, out_rgb, o->collect_counters);
    if (rc) g_error = t->tracer->multiContext() ? crt_multi_last_error(t->tracer->multiContext()) : crt_last_error(t->tracer->context());
    return rc;
  });
}

extern "C" const char *crt_host_tracer_note(const crt_host_tracer *t) { return t ? t->tracer->renderNote().c_str() : ""; }
extern "C" crt_ctx *crt_host_tracer_ctx(crt_host_tracer *t) { return t ? t->tracer->context() : nullptr; }
extern "C" crt_multi *crt_host_tracer_multi(crt_host_tracer *t) { return t ? t->tracer->multiContext() : nullptr; }

extern "C" int crt_host_tracer_stats(crt_host_tracer *t, crt_stats *out) {
  if (!t || !out) return CRT_ERR_INVALID;
  *out = t->tracer->stats();
  return CRT_OK;
}

extern "C" int crt_host_export_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height) {
  if (!path || !rgb) return CRT_ERR_INVALID;
  return guarded([&]() {
    crt::writePPM(path, rgb, width, height);
    return CRT_OK;
  });
}
