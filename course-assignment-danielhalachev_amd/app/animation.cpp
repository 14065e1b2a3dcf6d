// crt_animation -- camera-orbit driver.  Same flow as the reference's app/animation.cpp:8-40: one
// RayTracer (tree built and uploaded once), then FPS*SECONDS+1 frames of
//   setPosition(orbit) ; reset matrix ; pan(lookAtAngle) ; render(frame path)
// with the scene and tree resident on the GPU between frames (only the 12 camera floats change).
// What differs is the plumbing around the same frames: --in-flight K (default 2) keeps K frames in flight on the GPU
// (crt::RayTracer::renderAsync), and the P3 text files -- about 25 MB each at 1920x1080, an order of magnitude more host
// time than the frame takes to render -- are formatted by a few writer threads while the next frames render.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../host/RayTracer.h"
#include "../host/SceneParser.h"

namespace {
struct Job { std::string path; std::vector<uint8_t> rgb8; };
struct Writers {  // a few threads that turn quantised frames into PPM files
  std::mutex m;
  std::condition_variable cv, room;
  std::deque<Job> jobs;
  bool closing = false;
  unsigned int W, H;
  std::vector<std::thread> threads;
  Writers(unsigned int n, unsigned int w, unsigned int h) : W(w), H(h) {
    for (unsigned int i = 0; i < n; i++) threads.emplace_back([this] { run(); });
  }
  void run() {
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [this] { return closing || !jobs.empty(); });
        if (jobs.empty()) return;
        job = std::move(jobs.front());
        jobs.pop_front();
      }
      room.notify_one();
      crt::writePPMQuantized(job.path, job.rgb8.data(), W, H);
    }
  }
  void push(Job &&job) {
    std::unique_lock<std::mutex> lock(m);
    room.wait(lock, [this] { return jobs.size() < 2 * threads.size() + 2; });  // bounded: a frame is 6 MB
    jobs.push_back(std::move(job));
    cv.notify_one();
  }
  void close() {  // (idempotent: the destructor calls it again)
    { std::lock_guard<std::mutex> lock(m); closing = true; }
    cv.notify_all();
    for (auto &t : threads)
      if (t.joinable()) t.join();
  }
  // a frame that throws unwinds through here: joinable threads destroyed un-joined would end the process before main's handler prints the error
  ~Writers() { close(); }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s scene.crtscene out_prefix [--folder DIR] [--depth N] [--fps F] [--seconds S] [--radius R] [--device D] "
                         "[--in-flight K] [--writers N] [--no-ppm]\n", argv[0]);
    return 2;
  }
  std::string scenePath = argv[1], prefix = argv[2], folder;
  unsigned depth = 5;
  short FPS = 30, SECONDS = 10;  // app/animation.cpp:16-17
  float radius = 5.12f;          // app/animation.cpp:20
  int device = 0;
  unsigned inFlight = 2, nWriters = 4;
  bool writeFiles = true;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--folder") && i + 1 < argc) folder = argv[++i];
    else if (!strcmp(argv[i], "--depth") && i + 1 < argc) depth = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--fps") && i + 1 < argc) FPS = (short)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) SECONDS = (short)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--radius") && i + 1 < argc) radius = (float)atof(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--in-flight") && i + 1 < argc) inFlight = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--writers") && i + 1 < argc) nWriters = (unsigned)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-ppm")) writeFiles = false;
  }
  try {
    crt::SceneParser parser;
    crt::Scene scene = parser.parseScene(scenePath, folder);
    crt::RayTracer tracer(scene, device);
    tracer.setFramesInFlight(inFlight ? inFlight : 1);
    const unsigned W = scene.sceneSettings.image.width, H = scene.sceneSettings.image.height;
    crt::RenderOptions options(crt::BVHBucketsThreadPool, depth, false);
    Writers writers(writeFiles ? (nWriters ? nWriters : 1) : 0, W, H);
    const float DEG_CHANGE = 360.0f / (FPS * SECONDS);
    float degrees = 0;
    std::deque<std::pair<int, std::string>> inflight;  // (slot, file) of the frames enqueued and not yet collected
    auto collect = [&]() {
      const std::pair<int, std::string> f = inflight.front();
      inflight.pop_front();
      const uint8_t *rgb8 = tracer.finishFrame(f.first);
      if (writeFiles) writers.push(Job{f.second, std::vector<uint8_t>(rgb8, rgb8 + (size_t)W * H * 3)});
    };
    const auto t0 = std::chrono::steady_clock::now();
    unsigned frames = 0;
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
      if (inflight.size() >= tracer.framesInFlight()) collect();  // the ring is full: the oldest frame first
      inflight.emplace_back(tracer.renderAsync(options), prefix + std::to_string(t) + ".ppm");
      frames++;
      degrees += DEG_CHANGE;
    }
    while (!inflight.empty()) collect();
    const auto t1 = std::chrono::steady_clock::now();
    writers.close();
    const auto t2 = std::chrono::steady_clock::now();
    const double render_s = std::chrono::duration<double>(t1 - t0).count(), total_s = std::chrono::duration<double>(t2 - t0).count();
    std::printf("{\"frames\": %u, \"width\": %u, \"height\": %u, \"depth\": %u, \"frames_in_flight\": %u, \"render_s\": %.6f, "
                "\"frames_per_s\": %.3f, \"with_ppm_s\": %.6f, \"frames_per_s_with_ppm\": %.3f, \"ppm_files\": %s}\n",
                frames, W, H, depth, tracer.framesInFlight(), render_s, frames / render_s, total_s, frames / total_s,
                writeFiles ? "true" : "false");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
