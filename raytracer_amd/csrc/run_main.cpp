// run_main.cpp -- `run`, the command line of the render path.
//
// Same contract as the reference binary (parsing/parsing.cu:132-175, main.cu:14-70):
//   run -s scene.json [-w W] [-h H] [-m maxDepth] [-r resources.json] [-i image.png]
// flags may appear anywhere and are matched exactly; without -s the usage text is printed and the
// exit status is 0; the banner of defaults, "Oct taken seconds: x" and "Rendering seconds: y" go to
// stdout; mesh paths inside the scene are relative to the working directory.
// Additions (do not change the above): --device N, --stats (one JSON line with ray counts and rates).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
  auto t0 = clk::now();
  if (ac < 2 || !has_option(ac, av, "-s") || !option_value(ac, av, "-s")) {
    printf("Usage: ./raytracer -w W -h H -s scene.json ");
    printf("-r resources.txt -i image.png");
    return 0;
  }
  int W = 512, H = 512, depth = 3, device = 0;
  std::string res = "resources.json", img = "image.png", banner = "Using following defaults:\n";
  if (option_value(ac, av, "-w")) W = atoi(option_value(ac, av, "-w")); else banner += "+ Width: 512\n";
  if (option_value(ac, av, "-h")) H = atoi(option_value(ac, av, "-h")); else banner += "+ Height: 512\n";
  if (option_value(ac, av, "-m")) depth = atoi(option_value(ac, av, "-m")); else banner += "+ MaxRayDepth: 3\n";
  if (option_value(ac, av, "-r")) res = option_value(ac, av, "-r"); else banner += "+ Resources: resources.json\n";
  if (option_value(ac, av, "-i")) img = option_value(ac, av, "-i"); else banner += "+ Image: image.png\n";
  if (option_value(ac, av, "--device")) device = atoi(option_value(ac, av, "--device"));
  const bool want_stats = has_option(ac, av, "--stats");
  printf("%s", banner.c_str());

  rt_scene* sc = nullptr;
  int rc = rt_scene_load(option_value(ac, av, "-s"), res.c_str(), "", W, H, depth, &sc);
  if (rc != RT_OK) {
    // the reference prints "No <file> has been found" for a missing mesh and exits with status 1 (parsing.cu:26-29)
    printf("%s\n", rt_last_error());
    return 1;
  }
  rt_dscene* ds = nullptr;
  rc = rt_scene_upload(sc, device, &ds);
  if (rc != RT_OK) { fprintf(stderr, "run: %s\n", rt_last_error()); return 1; }
  printf("Oct taken seconds: %g\n", std::chrono::duration<double>(clk::now() - t0).count());
  t0 = clk::now();

  std::vector<uint8_t> rgb((size_t)W * H * 3);
  rt_stats st;
  rc = rt_render_rows_host(ds, 0, H, rgb.data(), nullptr, nullptr, RT_RENDER_TIME, &st);
  if (rc != RT_OK) printf("kernel launch failed with error \"%s\".\n", rt_last_error());   // main.cu:51-54: report, carry on
  rc = rt_write_png(img.c_str(), W, H, rgb.data());
  if (rc != RT_OK) fprintf(stderr, "run: %s\n", rt_last_error());
  rt_dscene_destroy(ds);
  rt_scene_destroy(sc);
  const double render_s = std::chrono::duration<double>(clk::now() - t0).count();
  printf("Rendering seconds: %g\n", render_s);
  if (want_stats) {
    rt_scene_info info;
    printf("{\"rays\": %llu, \"shadow_rays\": %llu, \"kernel_ms\": %.3f, \"mrays_per_s\": %.2f, \"version\": \"%s\"}\n",
           (unsigned long long)st.rays, (unsigned long long)st.shadow_rays, st.kernel_ms,
           st.kernel_ms > 0 ? st.rays / (st.kernel_ms * 1e3) : 0.0, rt_version());
    (void)info;
  }
  return 0;
}
