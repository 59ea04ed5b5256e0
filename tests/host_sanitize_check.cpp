// host_sanitize_check.cpp -- test harness, not part of the product.
// The host half of the library (scene and resource parsing, OBJ ingest, octree build, scene from arrays, image decoders, PNG
// writer: rt_host.cpp + rt_image.cpp, no HIP) built with AddressSanitizer and UndefinedBehaviorSanitizer and driven over
//   * every scene the assets hold (load, describe, create again from the description, dump, compare),
//   * the texture images, and a PNG written and read back,
//   * thousands of damaged inputs (scene JSON, resources JSON, OBJ, PNG, JPEG: truncated, bytes flipped, spliced), which must
//     end in a status code - any memory error or undefined operation ends the process with the sanitizer's report.
// Build and run: tests/test_host_sanitize.py.   Usage: host_sanitize_check ASSETS_DIR TMP_DIR N_DAMAGED SEED
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../include/rt_amd.h"

static std::string slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); std::stringstream s; s << f.rdbuf(); return s.str(); }
static void spit(const std::string& p, const std::string& d) { std::ofstream f(p, std::ios::binary); f.write(d.data(), (std::streamsize)d.size()); }
static int fails = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAILED: %s (%s)\n", what, rt_last_error()); fails++; } } while (0)

static std::string damage(const std::string& src, std::mt19937& rng) {
  std::string d = src;
  if (d.empty()) return d;
  switch (rng() % 5) {
    case 0: d.resize(rng() % d.size()); break;                                               // truncated
    case 1: for (int k = 0, n = 1 + (int)(rng() % 8); k < n; k++) d[rng() % d.size()] = (char)(rng() & 0xff); break;   // bytes replaced
    case 2: { const size_t a = rng() % d.size(), n = rng() % 64; d.erase(a, n); break; }    // a piece missing
    case 3: { const size_t a = rng() % d.size(), b = rng() % d.size(), n = rng() % 256; d.insert(a, d.substr(b, n)); break; }   // a piece twice
    default: { const size_t a = rng() % d.size(); d[a] = (char)(d[a] ^ (1 << (rng() % 8))); break; }   // one bit
  }
  return d;
}

int main(int ac, char** av) {
  if (ac < 5) return 2;
  const std::string assets = av[1], tmp = av[2];
  const int n_damaged = atoi(av[3]);
  std::mt19937 rng((unsigned)atoi(av[4]));
  const std::string res = assets + "/resources.json";
  const char* scenes[] = {"boxedSpheres", "heliumAttenuatedSphere", "heliumSphere", "lambertBPHSpheres", "meshMedBunny", "meshPlane", "meshReflectedSquircle",
                          "meshSphere", "meshTeapot", "meshTorus", "meshTransparentBunny", "meshTwoBunnies", "sphere", "twoSpheres", "meshDragon"};
  // ---- every scene: load, describe, create from the description, same tree and triangles ----
  for (const char* name : scenes) {
    rt_scene* a = nullptr; rt_scene* b = nullptr;
    const std::string path = assets + "/scenes/" + name + ".json";
    CHECK(rt_scene_load(path.c_str(), res.c_str(), assets.c_str(), 96, 64, 3, &a) == RT_OK, name);
    if (!a) continue;
    rt_scene_desc d;
    CHECK(rt_scene_describe(a, &d) == RT_OK, "describe");
    CHECK(rt_scene_create(&d, &b) == RT_OK, "create");
    rt_scene_info ia, ib; rt_scene_get_info(a, &ia); if (b) rt_scene_get_info(b, &ib);
    if (b) {
      CHECK(ia.n_nodes == ib.n_nodes && ia.n_leaf_refs == ib.n_leaf_refs && ia.n_triangles == ib.n_triangles, "same tree size");
      std::vector<int32_t> ka(ia.n_nodes + 1), kb(ia.n_nodes + 1), ra(ia.n_leaf_refs + 1), rb(ia.n_leaf_refs + 1);
      std::vector<float> ba(6 * (size_t)ia.n_nodes + 6), bb(6 * (size_t)ia.n_nodes + 6);
      CHECK(rt_scene_dump_tree(a, ka.data(), ba.data(), ra.data()) == RT_OK && rt_scene_dump_tree(b, kb.data(), bb.data(), rb.data()) == RT_OK, "dump tree");
      CHECK(ka == kb && ra == rb && !memcmp(ba.data(), bb.data(), ba.size() * 4), "same tree");
      std::vector<float> va(9 * (size_t)ia.n_triangles + 9), na(va.size()), vb(va.size()), nb(va.size());
      std::vector<int32_t> ma(ia.n_triangles + 1), mb(ia.n_triangles + 1);
      CHECK(rt_scene_dump_triangles(a, va.data(), na.data(), ma.data()) == RT_OK && rt_scene_dump_triangles(b, vb.data(), nb.data(), mb.data()) == RT_OK, "dump triangles");
      CHECK(!memcmp(va.data(), vb.data(), va.size() * 4) && !memcmp(na.data(), nb.data(), na.size() * 4) && ma == mb, "same triangles");
      float ca[19], cb[19];
      CHECK(rt_scene_dump_camera(a, ca) == RT_OK && rt_scene_dump_camera(b, cb) == RT_OK && !memcmp(ca, cb, sizeof ca), "same camera");
    }
    rt_scene_destroy(b); rt_scene_destroy(a);
  }
  // ---- images: the shipped textures, and a PNG written and read back ----
  for (const char* t : {"bricks.jpg", "bricks_SPEC.jpg", "chess.png", "earthmap.jpg", "red.png"}) {
    int w = 0, h = 0;
    const std::string p = assets + "/textures/" + t;
    CHECK(rt_image_load(p.c_str(), &w, &h, nullptr, 0) == RT_OK && w > 0 && h > 0, t);
    std::vector<uint8_t> px((size_t)w * h * 3);
    CHECK(rt_image_load(p.c_str(), &w, &h, px.data(), px.size()) == RT_OK, t);
    CHECK(rt_image_load(p.c_str(), &w, &h, px.data(), px.size() - 1) == RT_ERR_ARG, "short buffer refused");
  }
  {
    const int w = 37, h = 23; std::vector<uint8_t> px((size_t)w * h * 3), back(px.size());
    for (auto& v : px) v = (uint8_t)rng();
    const std::string p = tmp + "/roundtrip.png";
    int w2 = 0, h2 = 0;
    CHECK(rt_write_png(p.c_str(), w, h, px.data()) == RT_OK, "write png");
    CHECK(rt_image_load(p.c_str(), &w2, &h2, back.data(), back.size()) == RT_OK && w2 == w && h2 == h && px == back, "png round trip");
    CHECK(rt_write_png((tmp + "/no/such/dir/x.png").c_str(), w, h, px.data()) != RT_OK, "unwritable png refused");
  }
  // ---- bad arguments ----
  { rt_scene* s = nullptr;
    CHECK(rt_scene_load((assets + "/scenes/none.json").c_str(), res.c_str(), assets.c_str(), 64, 64, 3, &s) == RT_ERR_IO, "missing scene");
    CHECK(rt_scene_load((assets + "/scenes/sphere.json").c_str(), res.c_str(), assets.c_str(), 0, 64, 3, &s) == RT_ERR_ARG, "zero width");
    CHECK(rt_scene_load((assets + "/scenes/sphere.json").c_str(), res.c_str(), assets.c_str(), 64, 64, 10, &s) == RT_ERR_LIMIT, "depth 10");
    CHECK(rt_scene_load(nullptr, res.c_str(), assets.c_str(), 64, 64, 3, &s) == RT_ERR_ARG, "null path"); }
  // ---- damaged inputs: any status, no memory error, nothing undefined ----
  const std::string scene_src = slurp(assets + "/scenes/meshTransparentBunny.json"), scene2_src = slurp(assets + "/scenes/boxedSpheres.json");
  const std::string res_src = slurp(res), obj_src = slurp(assets + "/meshes/box_back.obj"), obj2_src = slurp(assets + "/meshes/torus.obj");
  const std::string png_src = slurp(assets + "/textures/chess.png"), jpg_src = slurp(assets + "/textures/bricks_SPEC.jpg");
  // a scene of its own whose mesh is the file that gets damaged
  const std::string obj_scene = "{\"camera\":{\"fov\":45.0,\"position\":[0,2,5],\"up\":[0,1,0],\"target\":[0,0,0]},\"objects\":[{\"__type__\":\"mesh\",\"file_path\":\"" + tmp +
                                "/damaged.obj\",\"materials\":[\"red_lambert\"]}],\"lights\":[{\"__type__\":\"point_light\",\"position\":[0,5,5],\"color\":[1,1,1]}],\"params\":{}}";
  spit(tmp + "/obj_scene.json", obj_scene);
  long ok_loads = 0, refused = 0;
  for (int k = 0; k < n_damaged; k++) {
    rt_scene* s = nullptr; int rc = 0;
    switch (k % 6) {
      case 0: spit(tmp + "/damaged.json", damage(k % 12 ? scene_src : scene2_src, rng));
              rc = rt_scene_load((tmp + "/damaged.json").c_str(), res.c_str(), assets.c_str(), 48, 32, 3, &s); break;
      case 1: spit(tmp + "/damaged_res.json", damage(res_src, rng));
              rc = rt_scene_load((assets + "/scenes/lambertBPHSpheres.json").c_str(), (tmp + "/damaged_res.json").c_str(), assets.c_str(), 48, 32, 3, &s); break;
      case 2: case 3: spit(tmp + "/damaged.obj", damage(k % 4 == 2 ? obj_src : obj2_src, rng));
              rc = rt_scene_load((tmp + "/obj_scene.json").c_str(), res.c_str(), "", 48, 32, 3, &s); break;
      case 4: { spit(tmp + "/damaged.png", damage(png_src, rng)); int w = 0, h = 0; std::vector<uint8_t> px(4096 * 3);
                rc = rt_image_load((tmp + "/damaged.png").c_str(), &w, &h, px.data(), px.size()); break; }
      default: { spit(tmp + "/damaged.jpg", damage(jpg_src, rng)); int w = 0, h = 0; std::vector<uint8_t> px(1 << 22);
                 rc = rt_image_load((tmp + "/damaged.jpg").c_str(), &w, &h, px.data(), px.size()); break; }
    }
    if (rc == RT_OK) ok_loads++; else refused++;
    if (s) { rt_scene_info i; rt_scene_get_info(s, &i); rt_scene_desc d; if (rt_scene_describe(s, &d) == RT_OK) { rt_scene* t = nullptr; if (rt_scene_create(&d, &t) == RT_OK) rt_scene_destroy(t); } }
    rt_scene_destroy(s);
  }
  printf("damaged inputs: %ld accepted, %ld refused; failures %d\n", ok_loads, refused, fails);
  return fails ? 1 : 0;
}
