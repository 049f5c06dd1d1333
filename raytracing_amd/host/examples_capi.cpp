// C entry points that build one of the example scenes with the host mirror,
// upload it through libmirt and render / trace it.  Used by the Python
// package (tests, bench.py); the flow is the one of the reference's example
// programs: scene function -> Camera::new -> Renderer::render.
#include <cstring>
#include <memory>
#include <string>

#include "scenes.hpp"

namespace {

struct Example {
  rt::DeviceScene device;
  std::string name;
  std::string error;
  float default_aspect = 1.0f;
  uint32_t width = 0, height = 0, spp = 0, depth = 0;
  rt::Color background;
  bool weekend_camera = false;
  size_t top_level_objects = 0;
};

thread_local std::string g_error;

}  // namespace

extern "C" {

const char* rt_example_last_error(void) { return g_error.c_str(); }

}  // extern "C"

// name: spheres | spheres_bvh | cornel_box | final_scene | synthetic:<half> | synthetic_bvh:<half>
// Builds the scene with the host mirror and hands (hittable, lights) to `sink`.
template <class Sink>
static bool with_example(const std::string& name, uint64_t scene_seed, const char* asset_dir, Example& ex, Sink&& sink) {
  ex.name = name;
  rt::seed(scene_seed);
  if (name == "spheres" || name == "spheres_bvh" || name.rfind("synthetic:", 0) == 0 ||
      name.rfind("synthetic_bvh:", 0) == 0) {
    int half = 11;
    bool inclusive = false;
    bool use_bvh = name == "spheres_bvh";
    if (name.rfind("synthetic", 0) == 0) {
      half = std::atoi(name.c_str() + name.find(':') + 1);
      inclusive = true;
      use_bvh = name.rfind("synthetic_bvh:", 0) == 0;
      if (half < 1) { g_error = "bad synthetic size"; return false; }
    }
    rt::World world = rt::scenes::random_spheres(half, inclusive);
    ex.top_level_objects = world.len();
    ex.default_aspect = 16.0f / 9.0f;
    ex.width = 600; ex.height = (uint32_t)(600.0f / ex.default_aspect); ex.spp = 10; ex.depth = 10;
    ex.background = rt::Color(0.8f, 0.8f, 0.8f);
    ex.weekend_camera = true;
    if (use_bvh) {
      rt::BVH bvh = rt::BVH::from_world(std::move(world), 0.0f, 1.0f);
      return sink(bvh, (const rt::BVH*)nullptr);
    }
    return sink(world, (const rt::World*)nullptr);
  }
  if (name == "cornel_box") {
    rt::World world = rt::scenes::cornell_box();
    rt::World lights = rt::scenes::cornell_lights();
    ex.top_level_objects = world.len();
    ex.default_aspect = 1.0f;
    ex.width = 1000; ex.height = 1000; ex.spp = 5; ex.depth = 5;
    return sink(world, (const rt::World*)&lights);
  }
  if (name == "final_scene") {
    auto world = rt::scenes::final_scene(std::string(asset_dir ? asset_dir : ".") + "/earthmap_1024x512.rgb");
    if (!world.is_ok()) { g_error = world.error; return false; }
    auto light = rt::scenes::final_scene_light();
    ex.top_level_objects = world.unwrap().len();
    ex.default_aspect = 1.0f;
    ex.width = 600; ex.height = 600; ex.spp = 5; ex.depth = 5;
    return sink(world.unwrap(), &light);
  }
  g_error = "unknown scene '" + name + "'";
  return false;
}

// lowering only (no GPU): lets the host logic be checked on a CPU-only machine
struct Lowered {
  Example ex;
  rt::Lowerer L;
};

extern "C" {

void* rt_example_open(const char* name_c, uint64_t scene_seed, const char* asset_dir, int device) {
  auto ex = std::make_unique<Example>();
  Example& e = *ex;
  const bool ok = with_example(name_c ? name_c : "", scene_seed, asset_dir, e, [&](const auto& hittable, const auto* lights) {
    const rt::Status st = e.device.build(hittable, lights, 0.0f, 1.0f, device);
    if (!st.is_ok()) g_error = st.error;
    return st.is_ok();
  });
  return ok ? ex.release() : nullptr;
}

void* rt_lowered_open(const char* name_c, uint64_t scene_seed, const char* asset_dir) {
  auto lo = std::make_unique<Lowered>();
  Lowered& l = *lo;
  const bool ok = with_example(name_c ? name_c : "", scene_seed, asset_dir, l.ex, [&](const auto& hittable, const auto* lights) {
    rt::lower_root(hittable, l.L);
    if (lights) lights->lower_lights_argument(l.L);
    if (!l.L.error.empty()) { g_error = l.L.error; return false; }
    return true;
  });
  return ok ? lo.release() : nullptr;
}
void rt_lowered_close(void* h) { delete static_cast<Lowered*>(h); }
// Uploads the lowered scene through a context the caller made (e.g. one over several devices).
int rt_lowered_upload(void* h, mirt_context* ctx, mirt_scene** out) {
  const mirt_scene_desc desc = rt::make_desc(static_cast<Lowered*>(h)->L, 0.0f, 1.0f);
  return mirt_scene_upload(ctx, &desc, out);
}
// out: spheres, moving spheres, rects, boxes, chains, chain ops, media, materials, textures, perlins, images,
//      lights, light groups, lights_mode, top-level objects, 0
void rt_lowered_counts(void* h, uint32_t out[16]) {
  const rt::Lowerer& L = static_cast<Lowered*>(h)->L;
  const uint32_t v[16] = {(uint32_t)L.spheres.size(), (uint32_t)L.moving_spheres.size(), (uint32_t)L.rects.size(),
                          (uint32_t)L.boxes.size(), (uint32_t)L.chains.size(), (uint32_t)L.chain_ops.size(),
                          (uint32_t)L.media.size(), (uint32_t)L.materials.size(), (uint32_t)L.textures.size(),
                          (uint32_t)L.perlins.size(), (uint32_t)L.images.size(), (uint32_t)L.lights.size(),
                          (uint32_t)L.light_groups.size(), L.lights_mode,
                          (uint32_t)static_cast<Lowered*>(h)->ex.top_level_objects, 0};
  std::memcpy(out, v, sizeof(v));
}
// Every surface primitive as rows of (tag, type_index, object_index, order, chain, material kind); returns the count.
uint32_t rt_lowered_prims(void* h, uint32_t* rows, uint32_t cap) {
  const rt::Lowerer& L = static_cast<Lowered*>(h)->L;
  uint32_t n = 0;
  auto put = [&](uint32_t tag, const mirt_prim_header& ph) {
    if (ph.flags & MIRT_PRIM_FLAG_BOUNDARY) return;
    if (n < cap) {
      uint32_t* r = rows + 6 * (size_t)n;
      r[0] = tag; r[1] = ph.type_index; r[2] = ph.object_index; r[3] = ph.order; r[4] = ph.chain;
      r[5] = L.materials[ph.material].kind;
    }
    ++n;
  };
  for (const auto& p : L.spheres) put(MIRT_PRIM_SPHERE, p.h);
  for (const auto& p : L.moving_spheres) put(MIRT_PRIM_MOVING_SPHERE, p.h);
  for (const auto& p : L.rects) put(MIRT_PRIM_RECT, p.h);
  for (const auto& p : L.boxes) put(MIRT_PRIM_BOX, p.h);
  return n;
}

void rt_example_close(void* h) { delete static_cast<Example*>(h); }

void rt_example_defaults(void* h, uint32_t out[4], float* aspect) {
  Example* ex = static_cast<Example*>(h);
  out[0] = ex->width; out[1] = ex->height; out[2] = ex->spp; out[3] = ex->depth;
  *aspect = ex->default_aspect;
}
uint64_t rt_example_primitive_count(void* h) { return static_cast<Example*>(h)->device.primitive_count(); }
uint64_t rt_example_object_count(void* h) { return static_cast<Example*>(h)->top_level_objects; }
mirt_context* rt_example_context(void* h) { return static_cast<Example*>(h)->device.context(); }
mirt_scene* rt_example_scene(void* h) { return static_cast<Example*>(h)->device.scene(); }

void rt_example_camera(void* h, float aspect, mirt_camera* cam, float background[3]) {
  Example* ex = static_cast<Example*>(h);
  const float a = aspect > 0.0f ? aspect : ex->default_aspect;
  const rt::Camera c = ex->weekend_camera ? rt::scenes::weekend_camera(a) : rt::scenes::cornell_camera(a);
  *cam = c.lower();
  background[0] = ex->background.x; background[1] = ex->background.y; background[2] = ex->background.z;
}

static mirt_render_params make_params(Example* ex, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_sample,
                                      uint32_t depth, uint64_t seed, uint32_t path_pool, uint32_t flags) {
  mirt_render_params p{};
  p.width = width; p.height = height;
  p.samples_per_pixel = spp;
  p.first_sample = first_sample;
  p.max_depth = depth;
  p.background[0] = ex->background.x; p.background[1] = ex->background.y; p.background[2] = ex->background.z;
  p.seed = seed;
  p.path_pool = path_pool;
  p.flags = flags;
  return p;
}

// Host-buffer render (mirt_render).  aspect <= 0 selects the example's own.
int rt_example_render(void* h, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_sample, uint32_t depth,
                      float aspect, uint64_t seed, uint32_t path_pool, uint32_t flags, uint8_t* rgb24, float* rgb_sum,
                      mirt_stats* stats) {
  Example* ex = static_cast<Example*>(h);
  mirt_camera cam;
  float bg[3];
  rt_example_camera(h, aspect, &cam, bg);
  const mirt_render_params p = make_params(ex, width, height, spp, first_sample, depth, seed, path_pool, flags);
  const int rc = mirt_render(ex->device.context(), ex->device.scene(), &cam, &p, rgb24, rgb_sum, stats);
  if (rc != MIRT_OK) g_error = mirt_last_error(ex->device.context());
  return rc;
}

// The frame as the reference leaves it on `threads` threads (renderer.rs:55-87): the rows it renders, moved to the
// top of the buffer, the remainder black.
int rt_example_render_as_reference(void* h, uint32_t width, uint32_t height, uint32_t spp, uint32_t depth, float aspect,
                                   uint64_t seed, uint32_t threads, uint8_t* rgb24, mirt_stats* stats) {
  Example* ex = static_cast<Example*>(h);
  const rt::ReferenceStrips strips{height, threads};
  if (!strips.valid()) { g_error = "more strip threads than rows"; return MIRT_ERR_INVALID_ARGUMENT; }
  mirt_camera cam;
  float bg[3];
  rt_example_camera(h, aspect, &cam, bg);
  mirt_render_params p = make_params(ex, width, height, spp, 0, depth, seed, 0, 0);
  p.first_row = strips.first_row();
  p.n_rows = strips.rows_kept();
  const int rc = mirt_render(ex->device.context(), ex->device.scene(), &cam, &p, rgb24, nullptr, stats);
  if (rc != MIRT_OK) { g_error = mirt_last_error(ex->device.context()); return rc; }
  strips.place(rgb24, width);
  return rc;
}

// ReferenceStrips on a caller's frame (no GPU involved): moves rows [height - kept, height) to the top and blackens
// the rest; returns the rows kept, or -1 when the reference would not run with that many threads.
int rt_reference_strips_place(uint8_t* rgb24, uint32_t width, uint32_t height, uint32_t threads) {
  const rt::ReferenceStrips strips{height, threads};
  if (!strips.valid()) return -1;
  strips.place(rgb24, width);
  return (int)strips.rows_kept();
}

// Device-buffer render (mirt_render_accumulate) for one rank's sample range.
int rt_example_render_accumulate(void* h, uint32_t width, uint32_t height, uint32_t spp, uint32_t first_sample,
                                 uint32_t depth, float aspect, uint64_t seed, uint32_t path_pool, uint32_t flags,
                                 float* rgb_sum_device, void* stream, mirt_stats* stats) {
  Example* ex = static_cast<Example*>(h);
  mirt_camera cam;
  float bg[3];
  rt_example_camera(h, aspect, &cam, bg);
  const mirt_render_params p = make_params(ex, width, height, spp, first_sample, depth, seed, path_pool, flags);
  const int rc = mirt_render_accumulate(ex->device.context(), ex->device.scene(), &cam, &p, rgb_sum_device, stream, stats);
  if (rc != MIRT_OK) g_error = mirt_last_error(ex->device.context());
  return rc;
}

int rt_example_resolve(void* h, const float* rgb_sum_device, uint32_t width, uint32_t height, uint32_t spp,
                       uint8_t* rgb24_device, void* stream) {
  Example* ex = static_cast<Example*>(h);
  const int rc = mirt_resolve(ex->device.context(), rgb_sum_device, width, height, spp, rgb24_device, stream);
  if (rc != MIRT_OK) g_error = mirt_last_error(ex->device.context());
  return rc;
}

int rt_example_trace(void* h, const mirt_ray* rays, size_t n, uint64_t seed, uint32_t* type_index,
                     uint32_t* object_index, float* t) {
  Example* ex = static_cast<Example*>(h);
  const int rc = mirt_trace_primary(ex->device.context(), ex->device.scene(), rays, n, seed, type_index, object_index, t);
  if (rc != MIRT_OK) g_error = mirt_last_error(ex->device.context());
  return rc;
}

}  // extern "C"
