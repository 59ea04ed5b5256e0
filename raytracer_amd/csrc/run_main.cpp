// run_main.cpp -- `run`, the command line of the render path.
//
// Same contract as the reference binary (parsing/parsing.cu:132-175, main.cu:14-70):
//   run -s scene.json [-w W] [-h H] [-m maxDepth] [-r resources.json] [-i image.png]
// flags may appear anywhere and are matched exactly; without -s the usage text is printed and the
// exit status is 0; the banner of defaults, "Oct taken seconds: x" and "Rendering seconds: y" go to
// stdout; mesh paths inside the scene are relative to the working directory.
// Additions (do not change the above):
//   --device N      HIP device of a single-GPU render (default 0)
//   --gpus N        render the frame on devices 0..N-1 of this node: 16-row bands dealt round-robin, one RCCL gather
//   --stats         one JSON line: ray counts, time per phase (parse, octree, upload, kernel, copy, png), Mrays/s and the
//                   algorithmic fraction of the FP32 vector roofline (from one more, instrumented, launch outside the timed part)
//   --counters F    with --stats: a hardware-counter record of this very frame (profiles/r*_counters.json); the FP32
//                   arithmetic the vector units executed is then printed beside the algorithmic figure
//   --frames K      after the image: the same frame K more times through a pipeline set up once (rt_frame_ctx): with --stats the
//                   mean time of a frame rendered alone (first launch to host memory) and the time per frame with frames in flight
//   --dump-ids F    primary-hit primitive ids, W*H little-endian int32 in row order (triangle >= 0, sphere s = -2-s, miss = -1)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"

static const char* option_value(int ac, char** av, const char* name) {
  for (int i = 0; i < ac; i++)
    if (!strcmp(av[i], name)) return i + 1 < ac ? av[i + 1] : nullptr;
  return nullptr;
}
static bool has_option(int ac, char** av, const char* name) {
  for (int i = 0; i < ac; i++)
    if (!strcmp(av[i], name)) return true;
  return false;
}

int main(int ac, char** av) {
  using clk = std::chrono::steady_clock;
  auto seconds_since = [](clk::time_point t) { return std::chrono::duration<double>(clk::now() - t).count(); };
  auto t0 = clk::now();
  if (ac < 2 || !has_option(ac, av, "-s") || !option_value(ac, av, "-s")) {
    printf("Usage: ./raytracer -w W -h H -s scene.json ");
    printf("-r resources.txt -i image.png");
    return 0;
  }
  int W = 512, H = 512, depth = 3, device = 0, gpus = 0;
  std::string res = "resources.json", img = "image.png", banner = "Using following defaults:\n";
  if (option_value(ac, av, "-w")) W = atoi(option_value(ac, av, "-w")); else banner += "+ Width: 512\n";
  if (option_value(ac, av, "-h")) H = atoi(option_value(ac, av, "-h")); else banner += "+ Height: 512\n";
  if (option_value(ac, av, "-m")) depth = atoi(option_value(ac, av, "-m")); else banner += "+ MaxRayDepth: 3\n";
  if (option_value(ac, av, "-r")) res = option_value(ac, av, "-r"); else banner += "+ Resources: resources.json\n";
  if (option_value(ac, av, "-i")) img = option_value(ac, av, "-i"); else banner += "+ Image: image.png\n";
  if (option_value(ac, av, "--device")) device = atoi(option_value(ac, av, "--device"));
  if (option_value(ac, av, "--gpus")) gpus = atoi(option_value(ac, av, "--gpus"));
  const int frames = option_value(ac, av, "--frames") ? atoi(option_value(ac, av, "--frames")) : 0;
  if (frames < 0 || frames > 100000) { fprintf(stderr, "run: --frames must be 0..100000\n"); return 1; }
  const bool want_stats = has_option(ac, av, "--stats");
  const char* ids_path = option_value(ac, av, "--dump-ids");
  const char* counters_path = option_value(ac, av, "--counters");
  const char* scene_path = option_value(ac, av, "-s");
  printf("%s", banner.c_str());
  if (gpus < 0 || gpus > 64) { fprintf(stderr, "run: --gpus must be 1..64\n"); return 1; }
  if (gpus > 0 && ids_path) { fprintf(stderr, "run: --dump-ids is a single-GPU option\n"); return 1; }

  // the HIP runtime comes up on a thread of its own while this one parses the scene and builds the octree
  // (a failure there is reported by rt_scene_upload below, which does the same steps again)
  std::thread bring_up([&] { for (int r = 0; r < (gpus > 0 ? gpus : 1); r++) (void)rt_device_prepare(gpus > 0 ? r : device); });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{bring_up};
  rt_scene* sc = nullptr;
  int rc = rt_scene_load(option_value(ac, av, "-s"), res.c_str(), "", W, H, depth, &sc);
  if (rc != RT_OK) {
    // the reference prints "No <file> has been found" for a missing mesh and exits with status 1 (parsing.cu:26-29)
    printf("%s\n", rt_last_error());
    return 1;
  }
  rt_scene_info info;
  rt_scene_get_info(sc, &info);
  const auto t_up = clk::now();
  bring_up.join();
  const int n_dev = gpus > 0 ? gpus : 1;
  std::vector<rt_dscene*> ds(n_dev, nullptr);
  std::vector<rt_comm*> comms(n_dev, nullptr);
  std::vector<int> devs(n_dev);
  for (int r = 0; r < n_dev; r++) {
    devs[r] = gpus > 0 ? r : device;
    rc = rt_scene_upload(sc, devs[r], &ds[r]);
    if (rc != RT_OK) { fprintf(stderr, "run: %s\n", rt_last_error()); return 1; }
  }
  if (gpus > 0 && (rc = rt_comm_init_all(n_dev, devs.data(), comms.data())) != RT_OK) { fprintf(stderr, "run: %s\n", rt_last_error()); return 1; }
  const double upload_s = seconds_since(t_up);
  printf("Oct taken seconds: %g\n", seconds_since(t0));
  t0 = clk::now();

  // page-locked, so that the copy engine writes the frame straight into it
  const size_t px = (size_t)W * H;
  uint8_t* rgb = (uint8_t*)rt_host_alloc(px * 3);
  int32_t* ids = ids_path ? (int32_t*)rt_host_alloc(px * 4) : nullptr;
  if (!rgb || (ids_path && !ids)) { fprintf(stderr, "run: %s\n", rt_last_error()); return 1; }
  memset(rgb, 0, px * 3);
  rt_stats st = {};
  if (gpus > 0) rc = rt_render_frame_gathered(ds.data(), comms.data(), n_dev, 16, rgb, 0, &st);
  else rc = rt_render_rows_host(ds[0], 0, H, rgb, ids, nullptr, RT_RENDER_TIME, &st);
  const bool rendered = rc == RT_OK;
  if (!rendered) printf("kernel launch failed with error \"%s\".\n", rt_last_error());   // main.cu:51-54: report, carry on
  int status = 0;
  // --frames K: the same frame K more times through a pipeline that is set up once (rt_frame_ctx), first one frame at a
  // time (latency: first launch to the frame in host memory), then with frames in flight (throughput)
  double frame_latency_ms = 0.0, pipelined_ms = 0.0, on_device_ms = 0.0; int in_flight = 0; bool frames_equal = true;
  if (frames > 0 && rendered) {
    in_flight = frames < 3 ? frames : (n_dev >= 6 ? 3 : (n_dev > 1 ? 2 : 3));
    rt_frame_ctx* fc = nullptr;
    rc = rt_frame_ctx_create(ds.data(), gpus > 0 ? comms.data() : nullptr, n_dev, 16, in_flight, &fc);
    if (rc != RT_OK) { fprintf(stderr, "run: %s\n", rt_last_error()); status = 1; }
    else {
      const uint8_t* got = nullptr; float ms = 0.f; bool ok = true;
      for (int s = 0; s < in_flight && ok; s++) ok = rt_frame_enqueue(fc, s, 0) == RT_OK && rt_frame_wait(fc, s, &got, &ms) == RT_OK;   // every slot once: streams, RCCL channels
      double sum = 0.0;
      for (int k = 0; k < frames && ok; k++) {
        ok = rt_frame_enqueue(fc, 0, 0) == RT_OK && rt_frame_wait(fc, 0, &got, &ms) == RT_OK;
        sum += ms;
        if (ok && memcmp(got, rgb, px * 3) != 0) frames_equal = false;
      }
      frame_latency_ms = sum / frames;
      const auto tp = clk::now();
      for (int k = 0; k < frames + in_flight && ok; k++) {
        const int s = k % in_flight;
        if (k >= in_flight) { ok = rt_frame_wait(fc, s, &got, nullptr) == RT_OK; if (ok && memcmp(got, rgb, px * 3) != 0) frames_equal = false; }
        if (k < frames && ok) ok = rt_frame_enqueue(fc, s, in_flight > 1 ? RT_RENDER_STRIPS : 0) == RT_OK;
      }
      pipelined_ms = seconds_since(tp) * 1e3 / frames;
      // and once more with the frames left in device memory (no 3 bytes per pixel over PCIe per frame: what the GPUs themselves keep up)
      const auto td = clk::now();
      for (int k = 0; k < frames + in_flight && ok; k++) {
        const int s = k % in_flight;
        if (k >= in_flight) ok = rt_frame_wait(fc, s, nullptr, nullptr) == RT_OK;
        if (k < frames && ok) ok = rt_frame_enqueue(fc, s, (in_flight > 1 ? RT_RENDER_STRIPS : 0) | RT_FRAME_ON_DEVICE) == RT_OK;
      }
      on_device_ms = seconds_since(td) * 1e3 / frames;
      if (!ok) { fprintf(stderr, "run: %s\n", rt_last_error()); status = 1; }
      if (!frames_equal) { fprintf(stderr, "run: a frame of the pipeline differs from the first frame\n"); status = 1; }
      rt_frame_ctx_destroy(fc);
    }
  }
  const auto t_png = clk::now();
  rc = rt_write_png(img.c_str(), W, H, rgb);
  if (rc != RT_OK) { fprintf(stderr, "run: %s\n", rt_last_error()); status = 1; }
  const double png_s = seconds_since(t_png);
  if (ids_path && rendered) {
    FILE* f = fopen(ids_path, "wb");
    if (!f || fwrite(ids, 4, px, f) != px) { fprintf(stderr, "run: cannot write %s\n", ids_path); status = 1; }
    if (f) fclose(f);
  }
  // the roofline figure of --stats needs the test counts of the frame: one instrumented launch, outside "Rendering seconds"
  rt_stats counted = {};
  const double render_s = seconds_since(t0);
  if (want_stats && rendered && rt_render_rows_host(ds[0], 0, H, rgb, nullptr, nullptr, RT_RENDER_COUNT_TESTS, &counted) != RT_OK) counted = rt_stats{};
  for (int r = 0; r < n_dev; r++) { rt_comm_destroy(comms[r]); rt_dscene_destroy(ds[r]); }
  rt_scene_destroy(sc);
  rt_host_free(rgb); rt_host_free(ids);
  printf("Rendering seconds: %g\n", render_s);
  if (want_stats) {
    if (!rendered) { printf("{\"error\": \"render failed\", \"version\": \"%s\"}\n", rt_version()); return status ? status : 1; }
    // algorithmic work of the minimal equivalent formulation per test (slab node test 24, Cramer triangle solve 61, sphere
    // quadratic 20, shading 150 per ray), over the kernel time, against the 157.3 TFLOP/s FP32 vector peak of an MI355X
    const double flop = 24.0 * counted.node_tests + 61.0 * counted.tri_tests + 20.0 * counted.sphere_tests + 150.0 * counted.rays;
    const double tflops = st.kernel_ms > 0 ? flop / (st.kernel_ms * 1e-3) / 1e12 : 0.0;
    // `algorithmic_frac` prices the tests the REFERENCE makes for this frame; the kernels prove most of them away without
    // executing them.  What the vector units executed can only be read from hardware counters: --counters FILE names a
    // counter record as tools/profile_to_json.py writes them (profiles/r*_counters.json), and when its workload is this
    // very frame its executed FLOP are printed beside the algorithmic figure.
    double executed_tflops = -1.0;
    if (counters_path) {
      std::string text;
      if (FILE* f = fopen(counters_path, "rb")) { char buf[4096]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k); fclose(f); }
      std::string base = scene_path ? scene_path : "";
      const size_t slash = base.rfind('/');
      if (slash != std::string::npos) base = base.substr(slash + 1);
      char want[512];
      snprintf(want, sizeof want, "\"workload\": \"scenes/%s %dx%d depth %d\"", base.c_str(), W, H, depth);
      const size_t at = text.find("\"executed_fp32_flop_per_launch\":");
      if (text.find(want) != std::string::npos && at != std::string::npos && st.kernel_ms > 0)
        executed_tflops = strtod(text.c_str() + at + 33, nullptr) / (st.kernel_ms * 1e-3) / 1e12;
    }
    printf("{\"rays\": %llu, \"shadow_rays\": %llu, \"rays_walked\": %llu, \"gpus\": %d, \"parse_ms\": %.3f, \"octree_ms\": %.3f, \"upload_ms\": %.3f, "
           "\"kernel_ms\": %.3f, \"copy_ms\": %.3f, \"png_ms\": %.3f, \"mrays_per_s\": %.2f, \"node_tests\": %llu, \"tri_tests\": %llu, "
           "\"algorithmic_tflops\": %.2f, \"algorithmic_frac\": %.4f, "
           "\"algorithmic_frac_is\": \"FLOP of the tests and shading the reference performs for this frame over kernel time and the 157.3 TFLOP/s FP32 peak: work delivered, not work executed\", ",
           (unsigned long long)st.rays, (unsigned long long)st.shadow_rays, (unsigned long long)st.rays_traced, n_dev, info.parse_seconds * 1e3, info.build_seconds * 1e3,
           upload_s * 1e3, st.kernel_ms, st.copy_ms, png_s * 1e3, st.kernel_ms > 0 ? st.rays / (st.kernel_ms * 1e3) : 0.0,
           (unsigned long long)counted.node_tests, (unsigned long long)counted.tri_tests, tflops, tflops / (157.3 * n_dev));
    if (executed_tflops >= 0.0)
      printf("\"executed_tflops\": %.2f, \"executed_frac\": %.4f, \"executed_frac_of_fma_free_peak\": %.4f, ", executed_tflops, executed_tflops / (157.3 * n_dev),
             executed_tflops / (78.65 * n_dev));
    else printf("\"executed_tflops\": null, ");
    if (frames > 0) printf("\"frames\": %d, \"frame_latency_ms\": %.4f, \"frames_in_flight\": %d, \"pipelined_ms_per_frame\": %.4f, \"pipelined_on_device_ms_per_frame\": %.4f, \"frames_identical\": %s, ", frames,
                           frame_latency_ms, in_flight, pipelined_ms, on_device_ms, frames_equal ? "true" : "false");
    printf("\"version\": \"%s\"}\n", rt_version());
  }
  return status;
}
