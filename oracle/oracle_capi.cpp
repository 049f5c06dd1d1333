// C entry points of the CPU oracle (test infrastructure; see oracle.hpp).
//
// Scene programs restate examples/spheres.rs, spheres_bvh.rs, cornel_box.rs
// and final_scene.rs of the reference, with the scene-construction randomness
// taken from a seeded generator in source order.  The render loop restates
// renderer.rs:43-92,193-229 (strips over threads; every row is rendered).
#include "oracle.hpp"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

using namespace oracle;

namespace {

struct SceneDefaults {
  uint32_t width, height, spp, depth;
  float aspect;
};

struct Scene {
  std::unique_ptr<Hittable> hittable;  // World or BVH
  World* world = nullptr;              // set when hittable is a World
  BVH* bvh = nullptr;                  // set when hittable is a BVH
  CulledBVH* culled = nullptr;         // set when hittable is the culling container (large-scene baseline, not in the reference)
  std::unique_ptr<Hittable> lights;    // may be null
  Point3 look_from, look_at;
  Vec3 v_up;
  float vfov = 40.0f, aperture = 0.0f, focus = 10.0f, time0 = 0.0f, time1 = 1.0f;
  Color background;
  SceneDefaults defaults{};
  uint32_t n_media = 0;
  std::string error;
};

using LambSolid = Lambertian<SolidTexture>;

// examples/spheres.rs:42-107 (same generator for spheres_bvh.rs:44-111).
// `half` = 11 reproduces `for a in -11..11`; `inclusive` extends the grid to
// [-half, half] for the large synthetic variant.
World weekend_scene(SceneRng& rng, int half, bool inclusive) {
  World world;
  world.add(Sphere<Lambertian<CheckerTexture<SolidTexture, SolidTexture>>>(
      Point3(0.0f, -1000.0f, 0.0f), 1000.0f,
      Lambertian<CheckerTexture<SolidTexture, SolidTexture>>(
          checker_from_colors(Color(0.2f, 0.3f, 0.1f), Color(0.9f, 0.9f, 0.9f)))));
  const int hi = inclusive ? half + 1 : half;
  for (int a = -half; a < hi; ++a) {
    for (int b = -half; b < hi; ++b) {
      const float choose_mat = rng.uniform01();
      const float cx = (float)a + 0.9f * rng.uniform01();
      const float cz = (float)b + 0.9f * rng.uniform01();
      const Point3 center(cx, 0.2f, cz);
      if ((center - Point3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
        if (choose_mat < 0.8f) {
          const LambSolid m(SolidTexture(rng.vec3(0.0f, 0.5f)));
          const Point3 center2 = center + Vec3(0.0f, rng.uniform(0.0f, 0.5f), 0.0f);
          world.add(MovingSphere<LambSolid>(center, center2, 0.0f, 1.0f, 0.2f, m));
        } else if (choose_mat < 0.95f) {
          const SolidTexture albedo(rng.vec3(0.5f, 1.0f));
          const float fuzz = rng.uniform01();
          world.add(Sphere<Metal<SolidTexture>>(center, 0.2f, Metal<SolidTexture>(albedo, fuzz)));
        } else {
          world.add(Sphere<Dielectric>(center, 0.2f, Dielectric(1.5f)));
        }
      }
    }
  }
  world.add(Sphere<Dielectric>(Point3(0.0f, 1.0f, 0.0f), 1.0f, Dielectric(1.5f)));
  world.add(Sphere<LambSolid>(Point3(-4.0f, 1.0f, 0.0f), 1.0f, LambSolid(SolidTexture::from_rgb(0.4f, 0.2f, 0.1f))));
  world.add(Sphere<Metal<SolidTexture>>(Point3(4.0f, 1.0f, 0.0f), 1.0f,
                                        Metal<SolidTexture>(SolidTexture::from_rgb(0.7f, 0.6f, 0.5f), 0.0f)));
  return world;
}

void weekend_camera(Scene& s) {  // examples/spheres.rs:3-37
  s.look_from = Point3(13.0f, 2.0f, 3.0f);
  s.look_at = Point3(0.0f, 0.0f, 0.0f);
  s.v_up = Vec3(0.0f, 1.0f, 0.0f);
  s.vfov = 40.0f; s.aperture = 0.1f; s.focus = 10.0f; s.time0 = 0.0f; s.time1 = 1.0f;
  s.background = Color(0.8f, 0.8f, 0.8f);
  const float aspect = 16.0f / 9.0f;
  s.defaults = {600, (uint32_t)(600.0f / aspect), 10, 10, aspect};
}

void cornell_camera(Scene& s) {  // examples/cornel_box.rs:28-52, final_scene.rs:18-42
  s.look_from = Point3(478.0f, 278.0f, -600.0f);
  s.look_at = Point3(278.0f, 278.0f, 0.0f);
  s.v_up = Vec3(0.0f, 1.0f, 0.0f);
  s.vfov = 40.0f; s.aperture = 0.0f; s.focus = 10.0f; s.time0 = 0.0f; s.time1 = 1.0f;
  s.background = Color(0.0f, 0.0f, 0.0f);
}

World cornell_box() {  // examples/cornel_box.rs:58-105
  World world;
  const LambSolid red(SolidTexture::from_rgb(0.65f, 0.05f, 0.05f));
  const LambSolid white(SolidTexture::from_rgb(0.73f, 0.73f, 0.73f));
  const LambSolid green(SolidTexture::from_rgb(0.12f, 0.45f, 0.15f));
  const DiffuseLight<SolidTexture> light(SolidTexture::from_rgb(15.0f, 15.0f, 15.0f));
  world.add(YZRect<LambSolid>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, green));
  world.add(YZRect<LambSolid>(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, red));
  world.add(FlipFace<XZRect<DiffuseLight<SolidTexture>>>(
      XZRect<DiffuseLight<SolidTexture>>(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, light)));
  world.add(XZRect<LambSolid>(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, white));
  world.add(XZRect<LambSolid>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white));
  world.add(XYRect<LambSolid>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white));
  world.add(Translate<Rotate<Box3d<LambSolid>>>(
      Rotate<Box3d<LambSolid>>(Box3d<LambSolid>(Point3(0.0f, 0.0f, 0.0f), Point3(165.0f, 330.0f, 165.0f), white), 15.0f),
      Vec3(265.0f, 0.0f, 295.0f)));
  world.add(Sphere<Dielectric>(Point3(190.0f, 90.0f, 190.0f), 90.0f, Dielectric(2.0f)));
  return world;
}

bool load_rgb(const std::string& path, size_t expect, std::vector<uint8_t>& out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  out.resize(expect);
  const size_t got = std::fread(out.data(), 1, expect, f);
  std::fclose(f);
  return got == expect;
}

bool final_scene(SceneRng& rng, const std::string& asset_dir, World& world, std::string& err) {
  // examples/final_scene.rs:48-171
  World boxes;
  const LambSolid ground(SolidTexture::from_rgb(0.48f, 0.83f, 0.53f));
  const uint32_t boxes_per_side = 20;
  for (uint32_t i = 0; i < boxes_per_side; ++i) {
    for (uint32_t j = 0; j < boxes_per_side; ++j) {
      const float w = 100.0f;
      const float x0 = -1000.0f + (float)i * w;
      const float z0 = -1000.0f + (float)j * w;
      const float y0 = 0.0f;
      const float x1 = x0 + w;
      const float y1 = (float)(1 + rng.below(100));  // gen_range(1..101)
      const float z1 = z0 + w;
      boxes.add(Box3d<LambSolid>(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
    }
  }
  world.add(std::move(boxes));
  const DiffuseLight<SolidTexture> light(SolidTexture::from_rgb(7.0f, 7.0f, 7.0f));
  world.add(FlipFace<XZRect<DiffuseLight<SolidTexture>>>(
      XZRect<DiffuseLight<SolidTexture>>(123.0f, 423.0f, 147.0f, 412.0f, 554.0f, light)));
  const Point3 center1(400.0f, 400.0f, 400.0f);
  const Point3 center2 = center1 + Point3(30.0f, 0.0f, 0.0f);
  world.add(MovingSphere<LambSolid>(center1, center2, 0.0f, 1.0f, 50.0f,
                                    LambSolid(SolidTexture::from_rgb(0.7f, 0.3f, 0.1f))));
  world.add(Sphere<Dielectric>(Point3(260.0f, 150.0f, 45.0f), 50.0f, Dielectric(1.5f)));
  world.add(Sphere<Metal<SolidTexture>>(Point3(0.0f, 150.0f, 145.0f), 50.0f,
                                        Metal<SolidTexture>(SolidTexture::from_rgb(0.8f, 0.8f, 0.8f), 1.0f)));
  using Medium = ConstantMedium<Sphere<Dielectric>, LambSolid>;
  world.add(Medium(Sphere<Dielectric>(Point3(360.0f, 150.0f, 145.0f), 70.0f, Dielectric(1.5f)), 0.2f,
                   LambSolid(SolidTexture::from_rgb(0.2f, 0.4f, 0.9f))));
  world.add(Medium(Sphere<Dielectric>(Point3(0.0f, 0.0f, 0.0f), 5000.0f, Dielectric(1.5f)), 0.0001f,
                   LambSolid(SolidTexture::from_rgb(1.0f, 1.0f, 1.0f))));
  auto earth = std::make_shared<std::vector<uint8_t>>();
  if (!load_rgb(asset_dir + "/earthmap_1024x512.rgb", 1024u * 512u * 3u, *earth)) {
    err = "cannot read " + asset_dir + "/earthmap_1024x512.rgb";
    return false;
  }
  world.add(Sphere<Lambertian<ImageTexture>>(Point3(400.0f, 200.0f, 400.0f), 100.0f,
                                             Lambertian<ImageTexture>(ImageTexture(earth, 1024, 512))));
  world.add(Sphere<Lambertian<NoiseTexture>>(Point3(220.0f, 280.0f, 300.0f), 80.0f,
                                             Lambertian<NoiseTexture>(NoiseTexture(0.1f, rng))));
  World boxes2;
  const LambSolid white(SolidTexture::from_rgb(0.73f, 0.73f, 0.73f));
  for (int i = 0; i < 1000; ++i) boxes2.add(Sphere<LambSolid>(rng.vec3(0.0f, 165.0f), 10.0f, white));
  world.add(Translate<Rotate<World>>(Rotate<World>(std::move(boxes2), 15.0f), Vec3(-100.0f, 270.0f, 395.0f)));
  return true;
}

Scene* build_scene(const std::string& name, uint64_t scene_seed, const std::string& asset_dir) {
  auto s = std::make_unique<Scene>();
  SceneRng rng(scene_seed);
  const LambSolid dummy(SolidTexture(Color(0.4f, 0.2f, 0.1f)));
  if (name == "spheres" || name == "spheres_bvh" || name.rfind("synthetic:", 0) == 0 ||
      name.rfind("synthetic_bvh:", 0) == 0 || name.rfind("synthetic_culled:", 0) == 0) {
    int half = 11;
    bool inclusive = false;
    bool use_bvh = (name == "spheres_bvh");
    if (name.rfind("synthetic", 0) == 0) {
      half = std::atoi(name.c_str() + name.find(':') + 1);
      inclusive = true;
      use_bvh = name.rfind("synthetic_bvh:", 0) == 0;
      if (half < 1) { s->error = "bad synthetic size"; return s.release(); }
    }
    World w = weekend_scene(rng, half, inclusive);
    weekend_camera(*s);
    if (name.rfind("synthetic_culled:", 0) == 0) {  // the same World behind a box test that culls (see CulledBVH)
      auto c = std::make_unique<CulledBVH>(CulledBVH::from_world(std::move(w)));
      s->culled = c.get();
      s->hittable = std::move(c);
    } else if (use_bvh) {
      auto b = std::make_unique<BVH>(BVH::from_world(std::move(w), 0.0f, 1.0f));  // spheres_bvh.rs:12
      s->bvh = b.get();
      s->hittable = std::move(b);
    } else {
      auto wp = std::make_unique<World>(std::move(w));
      s->world = wp.get();
      s->hittable = std::move(wp);
    }
  } else if (name == "cornel_box") {
    auto wp = std::make_unique<World>(cornell_box());
    s->world = wp.get();
    s->hittable = std::move(wp);
    auto lights = std::make_unique<World>();  // cornel_box.rs:12-26
    lights->add(XZRect<LambSolid>(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, dummy));
    lights->add(Sphere<LambSolid>(Point3(190.0f, 90.0f, 190.0f), 90.0f, dummy));
    s->lights = std::move(lights);
    cornell_camera(*s);
    s->defaults = {1000, 1000, 5, 5, 1.0f};
  } else if (name == "final_scene") {
    auto wp = std::make_unique<World>();
    if (!final_scene(rng, asset_dir, *wp, s->error)) return s.release();
    s->world = wp.get();
    s->hittable = std::move(wp);
    s->lights = std::make_unique<XZRect<LambSolid>>(123.0f, 423.0f, 147.0f, 412.0f, 554.0f, dummy);  // final_scene.rs:16
    cornell_camera(*s);
    s->defaults = {600, 600, 5, 5, 1.0f};
  } else {
    s->error = "unknown scene '" + name + "'";
    return s.release();
  }
  uint32_t n = 0;
  s->hittable->number_media(n);
  s->n_media = n;
  return s.release();
}

Camera make_camera(const Scene& s, float aspect) {
  return Camera(s.look_from, s.look_at, s.v_up, s.vfov, aspect > 0.0f ? aspect : s.defaults.aspect, s.aperture,
                s.focus, s.time0, s.time1);
}

}  // namespace

extern "C" {

void* oracle_scene_create(const char* name, uint64_t scene_seed, const char* asset_dir) {
  return build_scene(name ? name : "", scene_seed, asset_dir ? asset_dir : ".");
}
void oracle_scene_destroy(void* h) { delete static_cast<Scene*>(h); }
const char* oracle_scene_error(void* h) {
  Scene* s = static_cast<Scene*>(h);
  return s->error.empty() ? nullptr : s->error.c_str();
}
// out[0..3] = width, height, spp, depth; aspect_out = default aspect
void oracle_scene_defaults(void* h, uint32_t out[4], float* aspect_out) {
  Scene* s = static_cast<Scene*>(h);
  out[0] = s->defaults.width; out[1] = s->defaults.height; out[2] = s->defaults.spp; out[3] = s->defaults.depth;
  *aspect_out = s->defaults.aspect;
}
// cam[0..21]: origin, lower_left_corner, horizontal, vertical, u, v, w, lens_radius, time0, time1 ... (camera.rs:4-15)
void oracle_scene_camera(void* h, float aspect, float cam[24], float background[3]) {
  Scene* s = static_cast<Scene*>(h);
  const Camera c = make_camera(*s, aspect);
  const Vec3* v[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
  for (int i = 0; i < 7; ++i) { cam[3 * i] = v[i]->x; cam[3 * i + 1] = v[i]->y; cam[3 * i + 2] = v[i]->z; }
  cam[21] = c.lens_radius; cam[22] = c.time0; cam[23] = c.time1;
  background[0] = s->background.x; background[1] = s->background.y; background[2] = s->background.z;
}
uint32_t oracle_scene_object_count(void* h) {
  Scene* s = static_cast<Scene*>(h);
  const World* w = s->world ? s->world : (s->culled ? &s->culled->world : &s->bvh->world);
  uint32_t n = 0;
  for (const auto& col : w->data) n += (uint32_t)col.size();
  return n;
}

// In-order leaves of the bvh.rs tree (left to right); returns the leaf count.
static void collect_leaves(const BVH& b, const BVH::Child& c, std::vector<WorldIndex>& out) {
  if (!c.is_node) { out.push_back(c.index); return; }
  collect_leaves(b, b.nodes[c.node].left, out);
  collect_leaves(b, b.nodes[c.node].right, out);
}
uint32_t oracle_scene_bvh_leaf_order(void* h, uint32_t* type_index, uint32_t* object_index, uint32_t cap) {
  Scene* s = static_cast<Scene*>(h);
  if (!s->bvh || s->bvh->nodes.empty()) return 0;
  std::vector<WorldIndex> leaves;
  BVH::Child root;
  root.is_node = true;
  root.node = s->bvh->nodes.size() - 1;
  collect_leaves(*s->bvh, root, leaves);
  for (size_t i = 0; i < leaves.size() && i < cap; ++i) {
    type_index[i] = leaves[i].type_index;
    object_index[i] = leaves[i].object_index;
  }
  return (uint32_t)leaves.size();
}

// Renders samples [first_sample, first_sample+spp) of every pixel.
// rgb_sum (optional): W*H*3 f32 sums, row 0 = image top (the layout
// renderer.rs:254-258 gives the RGB24 buffer).  rgb24 (optional): the encoded
// image for `encode_spp` samples per pixel.  stats[0] = path segments traced.
int oracle_scene_render(void* h, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_sample,
                        uint32_t max_depth, float aspect, uint64_t seed, uint32_t threads, uint32_t estimator,
                        uint8_t* rgb24, float* rgb_sum, uint64_t* stats) {
  Scene* s = static_cast<Scene*>(h);
  if (!s || !s->error.empty() || width < 2 || height < 2) return 1;
  const Camera camera = make_camera(*s, aspect);
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  threads = std::min(threads, height);
  std::vector<float> local_sum;
  float* sum = rgb_sum;
  if (!sum) { local_sum.assign((size_t)width * height * 3, 0.0f); sum = local_sum.data(); }
  std::atomic<uint32_t> next_row{0};
  std::atomic<uint64_t> segments{0};
  const Hittable& world = *s->hittable;
  const Hittable* lights = s->lights.get();
  const Color background = s->background;
  auto worker = [&]() {
    Sampler sampler;
    sampler.key[0] = (uint32_t)seed;
    sampler.key[1] = (uint32_t)(seed >> 32);
    tls_sampler() = &sampler;
    tls_segments() = 0;
    tls_estimator() = estimator;
    const uint32_t strata = (estimator & EST_STRATA_MASK) >> EST_STRATA_SHIFT;
    for (;;) {
      const uint32_t y = next_row.fetch_add(1);
      if (y >= height) break;
      for (uint32_t x = 0; x < width; ++x) {
        Color color(0.0f, 0.0f, 0.0f);
        sampler.pixel = y * width + x;
        for (uint32_t k = 0; k < spp; ++k) {
          sampler.sample = first_sample + k;
          sampler.begin(0, STAGE_CAMERA);
          float jx = rand01(), jy = rand01();
          if (strata != 0) {  // opt-in: sample k stays inside cell k mod strata^2 of the pixel
            const uint32_t cell = sampler.sample % (strata * strata);
            jx = ((float)(cell % strata) + jx) / (float)strata;
            jy = ((float)(cell / strata) + jy) / (float)strata;
          }
          const float u = ((float)x + jx) / (float)(width - 1);   // renderer.rs:216
          const float v = ((float)y + jy) / (float)(height - 1);  // renderer.rs:217
          const Ray r = camera.get_ray(u, v);
          const Color c = ray_color(r, world, max_depth, 0, background, lights);
          if ((estimator & EST_DROP_NONFINITE) &&
              !(std::fabs(c.x) < INF_F && std::fabs(c.y) < INF_F && std::fabs(c.z) < INF_F))
            continue;  // opt-in: a sample that is not finite adds nothing
          color += c;
        }
        const size_t off = ((size_t)(height - 1 - y) * width + x) * 3;  // renderer.rs:254-255
        sum[off] = color.x; sum[off + 1] = color.y; sum[off + 2] = color.z;
      }
    }
    segments.fetch_add(tls_segments());
    tls_sampler() = nullptr;
    tls_estimator() = 0;
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (rgb24)
    for (size_t i = 0; i < (size_t)width * height * 3; ++i) rgb24[i] = encode_channel(sum[i], spp);
  if (stats) stats[0] = segments.load();
  return 0;
}

// Encodes f32 sums the way renderer.rs:232-259 does.
void oracle_encode(const float* rgb_sum, size_t n_values, uint32_t spp, uint8_t* out) {
  for (size_t i = 0; i < n_values; ++i) out[i] = encode_channel(rgb_sum[i], spp);
}

// Closest hit of each ray through the scene's own container (bvh.rs tournament
// for BVH scenes, world.rs scan otherwise) with t in (0.001, inf).
// rays: n x 7 floats (origin, direction, time).  Miss: indices 0xFFFFFFFF, t = inf.
// Media read draw `ordinal` of stream (pixel = ray index, sample 0, bounce 0).
int oracle_scene_trace(void* h, const float* rays, size_t n, uint64_t seed, uint32_t threads, uint32_t* type_index,
                       uint32_t* object_index, float* t_out) {
  Scene* s = static_cast<Scene*>(h);
  if (!s || !s->error.empty()) return 1;
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    Sampler sampler;
    sampler.key[0] = (uint32_t)seed;
    sampler.key[1] = (uint32_t)(seed >> 32);
    tls_sampler() = &sampler;
    for (;;) {
      const size_t begin = next.fetch_add(1024);
      if (begin >= n) break;
      const size_t end = std::min(n, begin + 1024);
      for (size_t i = begin; i < end; ++i) {
        const float* p = rays + 7 * i;
        const Ray r(Point3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5]), p[6]);
        sampler.pixel = (uint32_t)i;
        sampler.sample = 0;
        sampler.begin(0, STAGE_MEDIA);
        HitRecord rec;
        WorldIndex who;
        const bool ok = s->culled ? s->culled->hit_indexed(r, 0.001f, INF_F, rec, who)
                        : s->bvh  ? s->bvh->hit_indexed(r, 0.001f, INF_F, rec, who)
                                  : s->world->hit_indexed(r, 0.001f, INF_F, rec, who);
        type_index[i] = ok ? who.type_index : 0xFFFFFFFFu;
        object_index[i] = ok ? who.object_index : 0xFFFFFFFFu;
        t_out[i] = ok ? rec.t : INF_F;
      }
    }
    tls_sampler() = nullptr;
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  return 0;
}

// Full hit record of each ray (for record-level checks): out is n x 12 floats
// = point(3), normal(3), t, u, v, front_face, hit flag, pad.
int oracle_scene_hit_records(void* h, const float* rays, size_t n, uint64_t seed, float* out) {
  Scene* s = static_cast<Scene*>(h);
  if (!s || !s->error.empty()) return 1;
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  tls_sampler() = &sampler;
  for (size_t i = 0; i < n; ++i) {
    const float* p = rays + 7 * i;
    const Ray r(Point3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5]), p[6]);
    sampler.pixel = (uint32_t)i;
    sampler.sample = 0;
    sampler.begin(0, STAGE_MEDIA);
    HitRecord rec;
    const bool ok = s->hittable->hit(r, 0.001f, INF_F, rec);
    float* o = out + 12 * i;
    o[0] = rec.point.x; o[1] = rec.point.y; o[2] = rec.point.z;
    o[3] = rec.normal.x; o[4] = rec.normal.y; o[5] = rec.normal.z;
    o[6] = ok ? rec.t : INF_F; o[7] = rec.u; o[8] = rec.v;
    o[9] = rec.front_face ? 1.0f : 0.0f; o[10] = ok ? 1.0f : 0.0f; o[11] = 0.0f;
  }
  tls_sampler() = nullptr;
  return 0;
}

// The same with the object's WorldIndex and what the record's material makes of the hit before any random draw
// matters: 16 words per ray in the layout of the device library's mirt_hit_record --
// point(3), t, normal(3), front_face (u32), u, v, type_index (u32), object_index (u32), colour(3), hit flag (u32).
// colour = ScatterRecord::attenuation when the material scatters (material.rs:40-49, 66-90, 111-137), else what it
// emits towards the ray (material.rs:151-158; renderer.rs:151-154 evaluates emit before scatter).
int oracle_scene_hit_records_full(void* h, const float* rays, size_t n, uint64_t seed, float* out) {
  Scene* s = static_cast<Scene*>(h);
  if (!s || !s->error.empty()) return 1;
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  tls_sampler() = &sampler;
  auto put_u32 = [](float* dst, uint32_t v) { std::memcpy(dst, &v, 4); };
  for (size_t i = 0; i < n; ++i) {
    const float* p = rays + 7 * i;
    const Ray r(Point3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5]), p[6]);
    sampler.pixel = (uint32_t)i;
    sampler.sample = 0;
    sampler.begin(0, STAGE_MEDIA);
    HitRecord rec;
    WorldIndex who;
    const bool ok = s->culled ? s->culled->hit_indexed(r, 0.001f, INF_F, rec, who)
                    : s->bvh  ? s->bvh->hit_indexed(r, 0.001f, INF_F, rec, who)
                              : s->world->hit_indexed(r, 0.001f, INF_F, rec, who);
    float* o = out + 16 * i;
    for (int k = 0; k < 16; ++k) o[k] = 0.0f;
    o[3] = INF_F;
    put_u32(o + 10, 0xFFFFFFFFu);
    put_u32(o + 11, 0xFFFFFFFFu);
    if (!ok) continue;
    o[0] = rec.point.x; o[1] = rec.point.y; o[2] = rec.point.z; o[3] = rec.t;
    o[4] = rec.normal.x; o[5] = rec.normal.y; o[6] = rec.normal.z;
    put_u32(o + 7, rec.front_face ? 1u : 0u);
    o[8] = rec.u; o[9] = rec.v;
    put_u32(o + 10, who.type_index);
    put_u32(o + 11, who.object_index);
    sampler.begin(0, STAGE_SHADE);
    ScatterRecord sr;
    Color c;
    if (rec.material->scatter(r, rec, sr)) c = sr.attenuation;
    else c = rec.material->emit(r, rec, rec.u, rec.v, rec.point);
    o[12] = c.x; o[13] = c.y; o[14] = c.z;
    put_u32(o + 15, 1u);
  }
  tls_sampler() = nullptr;
  return 0;
}

// ---- small known-answer helpers -------------------------------------------
void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { Philox::block(ctr, key, out); }

// kind: 0 sphere(center, radius) p = cx,cy,cz,r ; 1 xy ; 2 xz ; 3 yz rect p = a0,a1,b0,b1,k ;
// 4 box p = min, max ; 5 moving sphere p = c0, c1, t0, t1, r
// out = t, u, v, front_face, point(3), normal(3)
int oracle_kat_hit(int kind, const float* p, const float ray[7], float t_min, float t_max, float out[10]) {
  const LambSolid m(SolidTexture(Color(0.5f, 0.5f, 0.5f)));
  const Ray r(Point3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]);
  HitRecord h;
  bool ok = false;
  switch (kind) {
    case 0: ok = Sphere<LambSolid>(Point3(p[0], p[1], p[2]), p[3], m).hit(r, t_min, t_max, h); break;
    case 1: ok = XYRect<LambSolid>(p[0], p[1], p[2], p[3], p[4], m).hit(r, t_min, t_max, h); break;
    case 2: ok = XZRect<LambSolid>(p[0], p[1], p[2], p[3], p[4], m).hit(r, t_min, t_max, h); break;
    case 3: ok = YZRect<LambSolid>(p[0], p[1], p[2], p[3], p[4], m).hit(r, t_min, t_max, h); break;
    case 4: ok = Box3d<LambSolid>(Point3(p[0], p[1], p[2]), Point3(p[3], p[4], p[5]), m).hit(r, t_min, t_max, h); break;
    case 5:
      ok = MovingSphere<LambSolid>(Point3(p[0], p[1], p[2]), Point3(p[3], p[4], p[5]), p[6], p[7], p[8], m)
               .hit(r, t_min, t_max, h);
      break;
    default: return -1;
  }
  if (!ok) return 0;
  out[0] = h.t; out[1] = h.u; out[2] = h.v; out[3] = h.front_face ? 1.0f : 0.0f;
  out[4] = h.point.x; out[5] = h.point.y; out[6] = h.point.z;
  out[7] = h.normal.x; out[8] = h.normal.y; out[9] = h.normal.z;
  return 1;
}

void oracle_kat_onb(const float n[3], float out[9]) {
  const Onb o = Onb::new_from_w(Vec3(n[0], n[1], n[2]));
  out[0] = o.u.x; out[1] = o.u.y; out[2] = o.u.z;
  out[3] = o.v.x; out[4] = o.v.y; out[5] = o.v.z;
  out[6] = o.w.x; out[7] = o.w.y; out[8] = o.w.z;
}

// Dielectric::scatter with the random draw replaced by `xi`; out = direction.
void oracle_kat_dielectric(float ior, const float dir[3], const float normal[3], int front_face, uint64_t seed,
                           float out[3]) {
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  sampler.begin(0, STAGE_SHADE);
  tls_sampler() = &sampler;
  HitRecord h;
  h.normal = Vec3(normal[0], normal[1], normal[2]);
  h.front_face = front_face != 0;
  ScatterRecord sr;
  Dielectric(ior).scatter(Ray(Point3(), Vec3(dir[0], dir[1], dir[2]), 0.0f), h, sr);
  out[0] = sr.specular_ray.direction.x; out[1] = sr.specular_ray.direction.y; out[2] = sr.specular_ray.direction.z;
  tls_sampler() = nullptr;
}

uint8_t oracle_kat_encode(float sum, uint32_t spp) { return encode_channel(sum, spp); }

// The direction samplers (vec3.rs:62-88) on the stream of path (pixel 0, sample 0, bounce 0, shading stage) of
// `seed`: kind 0 = random_cosine_direction, 1 = random_to_sphere(radius, distance_squared).  out = direction.
void oracle_kat_direction(int kind, float radius, float distance_squared, uint64_t seed, float out[3]) {
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  sampler.begin(0, STAGE_SHADE);
  tls_sampler() = &sampler;
  const Vec3 d = kind == 0 ? random_cosine_direction() : random_to_sphere(radius, distance_squared);
  out[0] = d.x; out[1] = d.y; out[2] = d.z;
  tls_sampler() = nullptr;
}

// Schlick's approximation as the Dielectric uses it (material.rs:105-108).
float oracle_kat_reflectance(float cosine, float ref_idx) { return Dielectric::reflectance(cosine, ref_idx); }

static void put_hit(const HitRecord& h, float out[10]) {
  out[0] = h.t; out[1] = h.u; out[2] = h.v; out[3] = h.front_face ? 1.0f : 0.0f;
  out[4] = h.point.x; out[5] = h.point.y; out[6] = h.point.z;
  out[7] = h.normal.x; out[8] = h.normal.y; out[9] = h.normal.z;
}

// A unit of transforms.rs around a sphere (p = cx,cy,cz,r):
// wrapper 1 Translate(offset = wp[0..3]); 2 Rotate(angle = wp[3] degrees); 3 FlipFace;
//         4 Translate(Rotate(sphere, wp[3]), wp[0..3]) as cornel_box.rs:90-99 nests them.
// out as oracle_kat_hit.
int oracle_kat_wrapped_sphere(int wrapper, const float wp[4], const float p[4], const float ray[7], float t_min,
                              float t_max, float out[10]) {
  const LambSolid m(SolidTexture(Color(0.5f, 0.5f, 0.5f)));
  using S = Sphere<LambSolid>;
  const S sphere(Point3(p[0], p[1], p[2]), p[3], m);
  const Vec3 offset(wp[0], wp[1], wp[2]);
  const Ray r(Point3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]);
  HitRecord h;
  bool ok = false;
  switch (wrapper) {
    case 1: ok = Translate<S>(sphere, offset).hit(r, t_min, t_max, h); break;
    case 2: ok = Rotate<S>(sphere, wp[3]).hit(r, t_min, t_max, h); break;
    case 3: ok = FlipFace<S>(sphere).hit(r, t_min, t_max, h); break;
    case 4: ok = Translate<Rotate<S>>(Rotate<S>(sphere, wp[3]), offset).hit(r, t_min, t_max, h); break;
    default: return -1;
  }
  if (ok) put_hit(h, out);
  return ok ? 1 : 0;
}

// ConstantMedium (transforms.rs:117-185) over a sphere boundary; the free-flight draw is draw 0 of the media
// stream of (pixel, sample 0, bounce 0) under `seed`, which oracle_philox lets a test work out for itself.
int oracle_kat_medium(const float sphere[4], float density, const float ray[7], float t_min, float t_max,
                      uint64_t seed, uint32_t pixel, float out[10]) {
  const LambSolid m(SolidTexture(Color(0.5f, 0.5f, 0.5f)));
  ConstantMedium<Sphere<Dielectric>, LambSolid> medium(
      Sphere<Dielectric>(Point3(sphere[0], sphere[1], sphere[2]), sphere[3], Dielectric(1.5f)), density, m);
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  sampler.pixel = pixel;
  sampler.begin(0, STAGE_MEDIA);
  tls_sampler() = &sampler;
  HitRecord h;
  const bool ok = medium.hit(Ray(Point3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]), t_min, t_max, h);
  tls_sampler() = nullptr;
  if (ok) put_hit(h, out);
  return ok ? 1 : 0;
}

// Hittable::pdf_value: kind 0 sphere p = cx,cy,cz,r (objects.rs:82-94); 2 XZ rect p = x0,x1,z0,z1,k (objects.rs:292-301);
// 1 XY rect, which keeps the trait's default (hittable.rs:52-54).
float oracle_kat_pdf_value(int kind, const float* p, const float origin[3], const float dir[3]) {
  const LambSolid m(SolidTexture(Color(0.5f, 0.5f, 0.5f)));
  const Point3 o(origin[0], origin[1], origin[2]);
  const Vec3 d(dir[0], dir[1], dir[2]);
  if (kind == 0) return Sphere<LambSolid>(Point3(p[0], p[1], p[2]), p[3], m).pdf_value(o, d);
  if (kind == 1) return XYRect<LambSolid>(p[0], p[1], p[2], p[3], p[4], m).pdf_value(o, d);
  return XZRect<LambSolid>(p[0], p[1], p[2], p[3], p[4], m).pdf_value(o, d);
}

// MixturePdf(HittablePdf(XZ rect light, origin), CosinePdf(normal)).value(dir)   pdf.rs:25-32,60-62,81-83
float oracle_kat_mixture_value(const float rect[5], const float origin[3], const float normal[3], const float dir[3]) {
  const LambSolid m(SolidTexture(Color(0.5f, 0.5f, 0.5f)));
  const XZRect<LambSolid> light(rect[0], rect[1], rect[2], rect[3], rect[4], m);
  const HittablePdf light_pdf(&light, Point3(origin[0], origin[1], origin[2]));
  const CosinePdf cosine_pdf(Vec3(normal[0], normal[1], normal[2]));
  return MixturePdf(&light_pdf, &cosine_pdf).value(Vec3(dir[0], dir[1], dir[2]));
}

// Perlin::noise (depth 0) or Perlin::turb(p, depth) with tables the caller supplies (perlin.rs:26-60, 75-93).
float oracle_kat_perlin(const float vec[256 * 3], const uint32_t perm_x[256], const uint32_t perm_y[256],
                        const uint32_t perm_z[256], const float p[3], uint32_t depth) {
  SceneRng rng(1);
  Perlin pn(rng);
  for (int i = 0; i < 256; ++i) {
    pn.random_vec[i] = Vec3(vec[3 * i], vec[3 * i + 1], vec[3 * i + 2]);
    pn.perm_x[i] = perm_x[i]; pn.perm_y[i] = perm_y[i]; pn.perm_z[i] = perm_z[i];
  }
  const Point3 q(p[0], p[1], p[2]);
  return depth == 0 ? pn.noise(q) : pn.turb(q, depth);
}

// ImageTexture::color(u, v) over a caller-supplied RGB8 image (texture.rs:101-122).
void oracle_kat_image_texel(const uint8_t* rgb8, uint32_t width, uint32_t height, float u, float v, float out[3]) {
  auto data = std::make_shared<std::vector<uint8_t>>(rgb8, rgb8 + (size_t)width * height * 3);
  const Color c = ImageTexture(data, width, height).color(u, v, Point3());
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

void oracle_kat_sphere_uv(const float p[3], float out[2]) { get_sphere_uv(Point3(p[0], p[1], p[2]), out[0], out[1]); }

// CheckerTexture over two solid colours (texture.rs:51-60): returns 1 when the odd texture answers.
int oracle_kat_checker_is_odd(const float p[3]) {
  const auto t = checker_from_colors(Color(1.0f, 0.0f, 0.0f), Color(0.0f, 0.0f, 0.0f));
  return t.color(0.0f, 0.0f, Point3(p[0], p[1], p[2])).x == 1.0f ? 1 : 0;
}

// AABB::hit as written in aabb.rs (always true for t_max > t_min on this path).
int oracle_kat_aabb_hit(const float box[6], const float ray[7], float t_min, float t_max) {
  const AABB b(Point3(box[0], box[1], box[2]), Point3(box[3], box[4], box[5]));
  return b.hit(Ray(Point3(ray[0], ray[1], ray[2]), Vec3(ray[3], ray[4], ray[5]), ray[6]), t_min, t_max) ? 1 : 0;
}

// renderer.rs:140-190 in a furnace: a Lambertian floor (albedo a) under a sky of radiance b and a lamp that emits the
// same b downwards (FlipFace(XZRect(DiffuseLight)), as in cornel_box.rs:79-86), sampled through the light list --
// the lamp alone, or with_sphere a World of the lamp and a glowing ball, one object of each type as in
// cornel_box.rs:12-26.  Every direction the mixture picks brings back b, so the mean of ray_color over the floor is
// a * b whatever the densities are, as long as pdf.rs:81-91 and objects.rs:82-100,292-309 agree with each other.
// Returns the mean of n paths that start above the floor and look down at it.
float oracle_kat_furnace(int with_sphere, float a, float b, uint32_t n, uint64_t seed) {
  const LambSolid grey(SolidTexture::from_rgb(a, a, a));
  const DiffuseLight<SolidTexture> glow(SolidTexture::from_rgb(b, b, b));
  World world;
  world.add(XZRect<LambSolid>(-40.0f, 40.0f, -80.0f, 0.0f, -1.0f, grey));
  world.add(FlipFace<XZRect<DiffuseLight<SolidTexture>>>(
      XZRect<DiffuseLight<SolidTexture>>(-1.0f, 1.0f, -5.0f, -3.0f, 1.5f, glow)));
  if (with_sphere) world.add(Sphere<DiffuseLight<SolidTexture>>(Point3(2.0f, 0.0f, -4.0f), 0.5f, glow));
  const XZRect<LambSolid> lamp(-1.0f, 1.0f, -5.0f, -3.0f, 1.5f, grey);
  World list;
  list.add(XZRect<LambSolid>(-1.0f, 1.0f, -5.0f, -3.0f, 1.5f, grey));
  list.add(Sphere<LambSolid>(Point3(2.0f, 0.0f, -4.0f), 0.5f, grey));
  const Hittable* lights = with_sphere ? static_cast<const Hittable*>(&list) : static_cast<const Hittable*>(&lamp);
  Sampler sampler;
  sampler.key[0] = (uint32_t)seed;
  sampler.key[1] = (uint32_t)(seed >> 32);
  tls_sampler() = &sampler;
  double sum = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    sampler.pixel = i;
    sampler.sample = 0;
    sampler.begin(0, STAGE_CAMERA);
    const float x = rand_range(-1.0f, 1.0f), z = rand_range(-1.0f, -0.3f);
    const Ray r(Point3(0.0f, 0.5f, 0.0f), Vec3(x, -1.0f, z), 0.0f);
    sum += ray_color(r, world, 3, 0, Color(b, b, b), lights).x;
  }
  tls_sampler() = nullptr;
  return (float)(sum / n);
}

}  // extern "C"
