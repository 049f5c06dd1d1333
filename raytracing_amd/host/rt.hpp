// Host-side mirror of the reference's public API (World / Camera / Renderer /
// Hittable / Material / Texture) for programs that render through libmirt.
//
// The reference is a Rust crate (src/lib.rs:1-33 re-exports everything); no
// Rust toolchain exists in this build, so the drop-in host layer is C++ with
// the same type names, constructor arguments and error behaviour.  Objects
// here only hold parameters: Renderer::render lowers the object tree to the
// flat arrays of include/mirt.h and the GPU does the rest.  Nothing in this
// file intersects a ray.
//
// Differences a caller sees, all forced by the environment:
//  * `Type::new(..)` is spelled as a constructor (`new` is a C++ keyword);
//    fallible constructors return rt::Result<T>.
//  * scene-building randomness comes from rt::rng(), a seeded generator
//    (rt::seed), instead of rand::thread_rng().
//  * ImageTexture reads raw RGB8 files (no JPEG decoder is linked).
//  * Renderer::present() writes a PPM instead of opening an SDL window.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <thread>
#include <unordered_map>
#include <memory>
#include <string>
#include <typeindex>
#include <typeinfo>
#include <utility>
#include <vector>

#include "../../include/mirt.h"

namespace rt {

// ---- Result<T, String> ------------------------------------------------------------
template <class T>
struct Result {
  std::unique_ptr<T> value;
  std::string error;
  bool is_ok() const { return value != nullptr; }
  T& unwrap() { return *value; }
  static Result ok(T v) { Result r; r.value = std::make_unique<T>(std::move(v)); return r; }
  static Result err(std::string e) { Result r; r.error = std::move(e); return r; }
};
struct Status {
  std::string error;
  bool is_ok() const { return error.empty(); }
  static Status ok() { return Status{}; }
  static Status err(std::string e) { return Status{std::move(e)}; }
};

// ---- vec3.rs ----------------------------------------------------------------------
struct Vec3 {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  Vec3() = default;
  Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  float length_squared() const { return x * x + y * y + z * z; }
  float length() const { return std::sqrt(length_squared()); }
  float dot(const Vec3& r) const { return x * r.x + y * r.y + z * r.z; }
  Vec3 cross(const Vec3& r) const { return Vec3(y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x); }
  Vec3 unit() const { const float l = length(); return Vec3(x / l, y / l, z / l); }
  static Vec3 random(float lo, float hi);  // vec3.rs:20-28, from rt::rng()
};
using Point3 = Vec3;
using Color = Vec3;
inline Vec3 operator-(const Vec3& a) { return Vec3(-a.x, -a.y, -a.z); }
inline Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(float s, const Vec3& v) { return Vec3(v.x * s, v.y * s, v.z * s); }
inline Vec3 operator*(const Vec3& v, float s) { return Vec3(v.x * s, v.y * s, v.z * s); }

// ---- seeded stand-in for rand::thread_rng() at scene-build time ----------------------
struct SceneRng {
  uint64_t state = 1;
  uint64_t next_u64() {
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  float uniform01() { return (float)(next_u64() >> 40) * (1.0f / 16777216.0f); }
  float gen_range(float lo, float hi) { return lo + (hi - lo) * uniform01(); }
  uint32_t gen_below(uint32_t n) { return (uint32_t)(((next_u64() >> 32) * (uint64_t)n) >> 32); }
  // integer gen_range(lo..hi), hi exclusive
  int32_t gen_range_i(int32_t lo, int32_t hi) { return lo + (int32_t)gen_below((uint32_t)(hi - lo)); }
};
inline SceneRng& rng() { static thread_local SceneRng r; return r; }
inline void seed(uint64_t s) { rng().state = s; }
inline Vec3 Vec3::random(float lo, float hi) {
  const float x = rng().gen_range(lo, hi), y = rng().gen_range(lo, hi), z = rng().gen_range(lo, hi);
  return Vec3(x, y, z);
}

// ---- aabb.rs (data only) -------------------------------------------------------------
struct AABB {
  Point3 minimum, maximum;
  AABB() = default;
  AABB(Point3 a, Point3 b) : minimum(a), maximum(b) {}
  static AABB surrounding_box(const AABB& a, const AABB& b) {  // aabb.rs:53-65
    return AABB(Point3(std::fmin(a.minimum.x, b.minimum.x), std::fmin(a.minimum.y, b.minimum.y),
                       std::fmin(a.minimum.z, b.minimum.z)),
                Point3(std::fmax(a.maximum.x, b.maximum.x), std::fmax(a.maximum.y, b.maximum.y),
                       std::fmax(a.maximum.z, b.maximum.z)));
  }
};

// ---- lowering state --------------------------------------------------------------------
struct Lowerer {
  std::vector<mirt_sphere> spheres;
  std::vector<mirt_moving_sphere> moving_spheres;
  std::vector<mirt_rect> rects;
  std::vector<mirt_box> boxes;
  std::vector<mirt_chain> chains{mirt_chain{0, 0}};
  std::vector<mirt_chain_op> chain_ops;
  std::vector<mirt_medium> media;
  std::vector<mirt_material> materials;
  std::vector<mirt_texture> textures;
  std::vector<mirt_perlin> perlins;
  std::vector<mirt_image> images;
  std::vector<std::shared_ptr<const std::vector<uint8_t>>> image_owners;
  std::vector<mirt_light> lights;
  std::vector<mirt_light_group> light_groups;
  uint32_t lights_mode = MIRT_LIGHTS_NONE;
  std::string error;  // first construct the device path cannot express

  // state of the walk
  std::vector<mirt_chain_op> wrappers;  // outermost first
  uint32_t type_index = 0, object_index = 0;
  uint32_t visit_counter = 0;
  bool boundary_mode = false;  // lowering the boundary of a constant medium
  int32_t last_boundary_tag = -1;
  uint32_t last_boundary_index = 0;
  uint32_t boundary_prims = 0;
  // Equal chains / materials / textures are stored once.  Keys are the records' bytes (all of them are
  // zero-initialised before they are filled in, so padding compares equal); a scene of millions of
  // differently coloured spheres makes millions of entries, hence hashing.
  template <class T>
  struct BytesHash {
    size_t operator()(const T& t) const {
      const unsigned char* p = reinterpret_cast<const unsigned char*>(&t);
      uint64_t h = 1469598103934665603ull;  // FNV-1a
      for (size_t i = 0; i < sizeof(T); ++i) { h ^= p[i]; h *= 1099511628211ull; }
      return (size_t)h;
    }
  };
  template <class T>
  struct BytesEqual {
    bool operator()(const T& a, const T& b) const { return std::memcmp(&a, &b, sizeof(T)) == 0; }
  };
  struct ChainKey {
    uint32_t n;
    mirt_chain_op ops[MIRT_MAX_CHAIN_OPS];
  };
  std::unordered_map<ChainKey, uint32_t, BytesHash<ChainKey>, BytesEqual<ChainKey>> chain_cache;
  std::unordered_map<mirt_material, uint32_t, BytesHash<mirt_material>, BytesEqual<mirt_material>> material_cache;
  std::unordered_map<mirt_texture, uint32_t, BytesHash<mirt_texture>, BytesEqual<mirt_texture>> texture_cache;

  void unsupported(const std::string& what) { if (error.empty()) error = what; }

  uint32_t current_chain() {
    if (wrappers.empty()) return 0;
    if (wrappers.size() > MIRT_MAX_CHAIN_OPS) { unsupported("more than MIRT_MAX_CHAIN_OPS nested wrappers"); return 0; }
    ChainKey key;
    std::memset(&key, 0, sizeof(key));
    key.n = (uint32_t)wrappers.size();
    std::memcpy(key.ops, wrappers.data(), wrappers.size() * sizeof(mirt_chain_op));
    auto it = chain_cache.find(key);
    if (it != chain_cache.end()) return it->second;
    mirt_chain c{(uint32_t)chain_ops.size(), (uint32_t)wrappers.size()};
    chain_ops.insert(chain_ops.end(), wrappers.begin(), wrappers.end());
    chains.push_back(c);
    chain_cache[key] = (uint32_t)chains.size() - 1;
    return (uint32_t)chains.size() - 1;
  }
  uint32_t add_texture(const mirt_texture& key) {
    const mirt_texture& t = key;
    auto it = texture_cache.find(key);
    if (it != texture_cache.end()) return it->second;
    textures.push_back(t);
    texture_cache[key] = (uint32_t)textures.size() - 1;
    return (uint32_t)textures.size() - 1;
  }
  uint32_t add_material(const mirt_material& key) {
    const mirt_material& m = key;
    auto it = material_cache.find(key);
    if (it != material_cache.end()) return it->second;
    materials.push_back(m);
    material_cache[key] = (uint32_t)materials.size() - 1;
    return (uint32_t)materials.size() - 1;
  }
  mirt_prim_header header(uint32_t material) {
    mirt_prim_header h{};
    h.material = material;
    h.chain = current_chain();
    h.order = visit_counter++;
    h.type_index = type_index;
    h.object_index = object_index;
    h.flags = boundary_mode ? (uint32_t)MIRT_PRIM_FLAG_BOUNDARY : 0u;
    return h;
  }
  void note_boundary(uint32_t tag, uint32_t index) {
    if (!boundary_mode) return;
    last_boundary_tag = (int32_t)tag;
    last_boundary_index = index;
    ++boundary_prims;
  }
};

// ---- texture.rs --------------------------------------------------------------------------
struct SolidTexture {
  Color color;
  SolidTexture() = default;
  static SolidTexture from_color(Color c) { SolidTexture t; t.color = c; return t; }
  static SolidTexture from_rgb(float r, float g, float b) { return from_color(Color(r, g, b)); }
  uint32_t lower(Lowerer& L) const {
    mirt_texture t{};
    t.kind = MIRT_TEX_SOLID;
    t.rgb[0] = color.x; t.rgb[1] = color.y; t.rgb[2] = color.z;
    return L.add_texture(t);
  }
};

template <class O, class E>
struct CheckerTexture {
  O odd;
  E even;
  static CheckerTexture from_textures(O o, E e) { return CheckerTexture{std::move(o), std::move(e)}; }
  uint32_t lower(Lowerer& L) const {
    mirt_texture t{};
    t.kind = MIRT_TEX_CHECKER;
    t.odd = odd.lower(L);
    t.even = even.lower(L);
    return L.add_texture(t);
  }
};
inline CheckerTexture<SolidTexture, SolidTexture> checker_from_colors(Color odd, Color even) {  // texture.rs:42-49
  return CheckerTexture<SolidTexture, SolidTexture>{SolidTexture::from_color(odd), SolidTexture::from_color(even)};
}

struct Perlin {  // perlin.rs:5-23, 62-73 (tables only)
  std::shared_ptr<mirt_perlin> tables;
  Perlin() : tables(std::make_shared<mirt_perlin>()) {
    for (int i = 0; i < 256; ++i) {
      const Vec3 v = Vec3::random(-1.0f, 1.0f).unit();
      tables->random_vec[i][0] = v.x; tables->random_vec[i][1] = v.y; tables->random_vec[i][2] = v.z;
    }
    generate_perm(tables->perm_x);
    generate_perm(tables->perm_y);
    generate_perm(tables->perm_z);
  }
  static void generate_perm(uint32_t p[256]) {
    for (uint32_t i = 0; i < 256; ++i) p[i] = i;
    for (uint32_t i = 255; i >= 1; --i) std::swap(p[i], p[rng().gen_below(i)]);  // Uniform::new(0, i)
  }
};

struct NoiseTexture {
  Perlin noise;
  float scale;
  explicit NoiseTexture(float scale_) : scale(scale_) {}  // texture.rs:67-74
  uint32_t lower(Lowerer& L) const {
    L.perlins.push_back(*noise.tables);
    mirt_texture t{};
    t.kind = MIRT_TEX_NOISE;
    t.scale = scale;
    t.index = (uint32_t)L.perlins.size() - 1;
    return L.add_texture(t);
  }
};

struct ImageTexture {
  std::shared_ptr<const std::vector<uint8_t>> data;  // RGB8, row 0 = top
  uint32_t width = 0, height = 0;
  // texture.rs:90-99.  `path` names a raw RGB8 file "<stem>_<W>x<H>.rgb".
  static Result<ImageTexture> open(const std::string& path) {
    const size_t us = path.rfind('_'), xs = path.rfind('x'), dot = path.rfind('.');
    if (us == std::string::npos || xs == std::string::npos || dot == std::string::npos || !(us < xs && xs < dot))
      return Result<ImageTexture>::err("image path must look like name_<W>x<H>.rgb: " + path);
    ImageTexture t;
    t.width = (uint32_t)std::atoi(path.substr(us + 1, xs - us - 1).c_str());
    t.height = (uint32_t)std::atoi(path.substr(xs + 1, dot - xs - 1).c_str());
    if (t.width == 0 || t.height == 0) return Result<ImageTexture>::err("bad image size in " + path);
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return Result<ImageTexture>::err("cannot open " + path);
    auto buf = std::make_shared<std::vector<uint8_t>>((size_t)t.width * t.height * 3);
    const size_t got = std::fread(buf->data(), 1, buf->size(), f);
    std::fclose(f);
    if (got != buf->size()) return Result<ImageTexture>::err("short read on " + path);
    t.data = buf;
    return Result<ImageTexture>::ok(std::move(t));
  }
  uint32_t lower(Lowerer& L) const {
    L.image_owners.push_back(data);
    L.images.push_back(mirt_image{data->data(), width, height});
    mirt_texture t{};
    t.kind = MIRT_TEX_IMAGE;
    t.index = (uint32_t)L.images.size() - 1;
    return L.add_texture(t);
  }
};

// ---- material.rs ---------------------------------------------------------------------------
template <class T>
struct Lambertian {
  T albedo;
  explicit Lambertian(T a) : albedo(std::move(a)) {}
  uint32_t lower(Lowerer& L) const { return L.add_material(mirt_material{MIRT_MAT_LAMBERTIAN, albedo.lower(L), 0.0f}); }
};
template <class T>
struct Metal {
  T albedo;
  float fuzz;
  Metal(T a, float f) : albedo(std::move(a)), fuzz(f < 1.0f ? f : 1.0f) {}  // material.rs:67-72
  uint32_t lower(Lowerer& L) const { return L.add_material(mirt_material{MIRT_MAT_METAL, albedo.lower(L), fuzz}); }
};
struct Dielectric {
  float index_of_refraction;
  explicit Dielectric(float ior) : index_of_refraction(ior) {}
  uint32_t lower(Lowerer& L) const { return L.add_material(mirt_material{MIRT_MAT_DIELECTRIC, 0, index_of_refraction}); }
};
template <class T>
struct DiffuseLight {
  T emit;
  explicit DiffuseLight(T e) : emit(std::move(e)) {}
  uint32_t lower(Lowerer& L) const { return L.add_material(mirt_material{MIRT_MAT_DIFFUSE_LIGHT, emit.lower(L), 0.0f}); }
};
template <class T>
struct Isotropic {
  T albedo;
  explicit Isotropic(T a) : albedo(std::move(a)) {}
  uint32_t lower(Lowerer& L) const { return L.add_material(mirt_material{MIRT_MAT_ISOTROPIC, albedo.lower(L), 0.0f}); }
};

// ---- hittable.rs ------------------------------------------------------------------------------
struct Hittable {
  virtual ~Hittable() = default;
  virtual AABB bounding_box() const = 0;
  // Appends this object (and everything inside it) to the lowered scene.
  virtual void lower(Lowerer& L) const = 0;
  // Appends this object as one member of a light list (pdf_value / random of hittable.rs:52-57).
  virtual void lower_light(Lowerer& L) const { L.lights.push_back(mirt_light{MIRT_LIGHT_OTHER, {0, 0, 0, 0, 0}}); }
  // Lowers this object as the whole `lights` argument of Renderer::render.
  virtual void lower_lights_argument(Lowerer& L) const {
    L.lights_mode = MIRT_LIGHTS_OBJECT;
    lower_light(L);
  }
};

// ---- objects.rs -----------------------------------------------------------------------------------
template <class M>
struct Sphere final : Hittable {
  Point3 center;
  float radius;
  M material;
  Sphere(Point3 c, float r, M m) : center(c), radius(r), material(std::move(m)) {}
  AABB bounding_box() const override {
    return AABB(center - Vec3(radius, radius, radius), center + Vec3(radius, radius, radius));
  }
  void lower(Lowerer& L) const override {
    mirt_sphere s{};
    s.h = L.header(material.lower(L));
    s.center[0] = center.x; s.center[1] = center.y; s.center[2] = center.z;
    s.radius = radius;
    L.spheres.push_back(s);
    L.note_boundary(MIRT_PRIM_SPHERE, (uint32_t)L.spheres.size() - 1);
  }
  void lower_light(Lowerer& L) const override {
    L.lights.push_back(mirt_light{MIRT_LIGHT_SPHERE, {center.x, center.y, center.z, radius, 0}});
  }
};

template <class M>
struct MovingSphere final : Hittable {
  Point3 center0, center1;
  float time0, time1, radius;
  M material;
  MovingSphere(Point3 c0, Point3 c1, float t0, float t1, float r, M m)
      : center0(c0), center1(c1), time0(t0), time1(t1), radius(r), material(std::move(m)) {}
  AABB bounding_box() const override {  // objects.rs:173-180
    const Vec3 r(radius, radius, radius);
    return AABB::surrounding_box(AABB(center0 - r, center0 + r), AABB(center1 - r, center1 + r));
  }
  void lower(Lowerer& L) const override {
    mirt_moving_sphere s{};
    s.h = L.header(material.lower(L));
    s.center0[0] = center0.x; s.center0[1] = center0.y; s.center0[2] = center0.z;
    s.center1[0] = center1.x; s.center1[1] = center1.y; s.center1[2] = center1.z;
    s.time0 = time0; s.time1 = time1; s.radius = radius;
    L.moving_spheres.push_back(s);
    L.note_boundary(MIRT_PRIM_MOVING_SPHERE, (uint32_t)L.moving_spheres.size() - 1);
  }
};

template <class M, uint32_t AXIS>
struct RectBase : Hittable {
  float a0, a1, b0, b1, k;
  M material;
  RectBase(float a0_, float a1_, float b0_, float b1_, float k_, M m)
      : a0(a0_), a1(a1_), b0(b0_), b1(b1_), k(k_), material(std::move(m)) {}
  AABB bounding_box() const override {  // objects.rs:230-235, 285-290, 359-364
    if (AXIS == MIRT_RECT_XY) return AABB(Point3(a0, b0, k - 0.00001f), Point3(a1, b1, k + 0.00001f));
    if (AXIS == MIRT_RECT_XZ) return AABB(Point3(a0, k - 0.00001f, b0), Point3(a1, k + 0.00001f, b1));
    return AABB(Point3(k - 0.00001f, a0, b0), Point3(k + 0.00001f, a1, b1));
  }
  void lower(Lowerer& L) const override {
    mirt_rect r{};
    r.h = L.header(material.lower(L));
    r.axis = AXIS;
    r.a0 = a0; r.a1 = a1; r.b0 = b0; r.b1 = b1; r.k = k;
    L.rects.push_back(r);
    L.note_boundary(MIRT_PRIM_RECT, (uint32_t)L.rects.size() - 1);
  }
};
template <class M>
struct XYRect final : RectBase<M, MIRT_RECT_XY> {
  XYRect(float x0, float x1, float y0, float y1, float k, M m) : RectBase<M, MIRT_RECT_XY>(x0, x1, y0, y1, k, std::move(m)) {}
};
template <class M>
struct XZRect final : RectBase<M, MIRT_RECT_XZ> {
  XZRect(float x0, float x1, float z0, float z1, float k, M m) : RectBase<M, MIRT_RECT_XZ>(x0, x1, z0, z1, k, std::move(m)) {}
  void lower_light(Lowerer& L) const override {  // objects.rs:292-309
    L.lights.push_back(mirt_light{MIRT_LIGHT_XZ_RECT, {this->a0, this->a1, this->b0, this->b1, this->k}});
  }
};
template <class M>
struct YZRect final : RectBase<M, MIRT_RECT_YZ> {
  YZRect(float y0, float y1, float z0, float z1, float k, M m) : RectBase<M, MIRT_RECT_YZ>(y0, y1, z0, z1, k, std::move(m)) {}
};

template <class M>
struct Box3d final : Hittable {
  Point3 min, max;
  M material;
  Box3d(Point3 mn, Point3 mx, M m) : min(mn), max(mx), material(std::move(m)) {}
  AABB bounding_box() const override { return AABB(min, max); }
  void lower(Lowerer& L) const override {
    mirt_box b{};
    b.h = L.header(material.lower(L));
    b.min[0] = min.x; b.min[1] = min.y; b.min[2] = min.z;
    b.max[0] = max.x; b.max[1] = max.y; b.max[2] = max.z;
    L.boxes.push_back(b);
    L.note_boundary(MIRT_PRIM_BOX, (uint32_t)L.boxes.size() - 1);
  }
};

// ---- transforms.rs ---------------------------------------------------------------------------------
template <class T>
struct Translate final : Hittable {
  T object;
  Vec3 offset;
  Translate(T o, Vec3 off) : object(std::move(o)), offset(off) {}
  AABB bounding_box() const override {
    const AABB b = object.bounding_box();
    return AABB(b.minimum + offset, b.maximum + offset);
  }
  void lower(Lowerer& L) const override {
    L.wrappers.push_back(mirt_chain_op{MIRT_OP_TRANSLATE, {offset.x, offset.y, offset.z}});
    object.lower(L);
    L.wrappers.pop_back();
  }
};

inline float to_radians(float deg) { return deg * (3.14159265358979323846264338327950288f / 180.0f); }

template <class T>
struct Rotate final : Hittable {
  T object;
  float sin_theta, cos_theta;
  AABB aabb;
  Rotate(T o, float angle) : object(std::move(o)) {  // transforms.rs:40-77
    const float radians = to_radians(angle);
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
    const AABB b = object.bounding_box();
    const float inf = std::numeric_limits<float>::infinity();
    Point3 mn(inf, inf, inf), mx(-inf, -inf, -inf);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int k = 0; k < 2; ++k) {
          const float x = (float)i * b.maximum.x + (1.0f - (float)i) * b.minimum.x;
          const float y = (float)j * b.maximum.y + (1.0f - (float)j) * b.minimum.y;
          const float z = (float)k * b.maximum.z + (1.0f - (float)k) * b.minimum.z;
          const float nx = cos_theta * x + sin_theta * z;
          const float nz = -sin_theta * x + cos_theta * z;
          mn.x = std::fmin(mn.x, nx); mn.y = std::fmin(mn.y, y); mn.z = std::fmin(mn.z, nz);
          mx.x = std::fmax(mx.x, nx); mx.y = std::fmax(mx.y, y); mx.z = std::fmax(mx.z, nz);
        }
    aabb = AABB(mn, mx);
  }
  AABB bounding_box() const override { return aabb; }
  void lower(Lowerer& L) const override {
    L.wrappers.push_back(mirt_chain_op{MIRT_OP_ROTATE_Y, {sin_theta, cos_theta, 0.0f}});
    object.lower(L);
    L.wrappers.pop_back();
  }
};

template <class T>
struct FlipFace final : Hittable {
  T object;
  explicit FlipFace(T o) : object(std::move(o)) {}
  AABB bounding_box() const override { return object.bounding_box(); }
  void lower(Lowerer& L) const override {
    L.wrappers.push_back(mirt_chain_op{MIRT_OP_FLIP_FACE, {0, 0, 0}});
    object.lower(L);
    L.wrappers.pop_back();
  }
};

template <class T, class M>
struct ConstantMedium final : Hittable {
  T boundary;
  M phase_function;
  float neg_inv_density;
  ConstantMedium(T b, float density, M m)
      : boundary(std::move(b)), phase_function(std::move(m)), neg_inv_density(-1.0f / density) {}
  AABB bounding_box() const override { return boundary.bounding_box(); }
  void lower(Lowerer& L) const override {
    if (L.boundary_mode) { L.unsupported("a constant medium used as the boundary of another medium"); return; }
    mirt_medium m{};
    m.neg_inv_density = neg_inv_density;
    m.material = phase_function.lower(L);
    m.chain = L.current_chain();
    m.order = L.visit_counter++;
    m.type_index = L.type_index;
    m.object_index = L.object_index;
    L.boundary_mode = true;
    L.boundary_prims = 0;
    L.last_boundary_tag = -1;
    const uint32_t keep = L.visit_counter;
    boundary.lower(L);
    L.visit_counter = keep;
    L.boundary_mode = false;
    if (L.boundary_prims != 1 || L.last_boundary_tag < 0) {
      L.unsupported("constant medium whose boundary is not a single primitive");
      return;
    }
    m.boundary_tag = (uint32_t)L.last_boundary_tag;
    m.boundary_index = L.last_boundary_index;
    L.media.push_back(m);
  }
};

// ---- world.rs ------------------------------------------------------------------------------------------
struct WorldIndex {
  uint32_t type_index = 0, object_index = 0;
};

struct World final : Hittable {
  AABB aabb;  // starts at the default box, world.rs:19-24
  std::vector<std::type_index> types;
  std::vector<std::vector<std::shared_ptr<Hittable>>> data;  // one column per concrete type

  World() = default;
  World(World&&) = default;
  World& operator=(World&&) = default;
  World(const World&) = delete;

  template <class T>
  void add(T object) {  // world.rs:27-40
    const std::type_index id(typeid(T));
    size_t index = 0;
    for (; index < types.size(); ++index)
      if (types[index] == id) break;
    if (index == types.size()) {
      types.push_back(id);
      data.emplace_back();
    }
    aabb = AABB::surrounding_box(aabb, object.bounding_box());
    data[index].push_back(std::make_shared<T>(std::move(object)));
  }
  std::vector<std::pair<WorldIndex, AABB>> volumes() const {  // world.rs:42-59, columns in type_index order
    std::vector<std::pair<WorldIndex, AABB>> v;
    for (uint32_t ti = 0; ti < data.size(); ++ti)
      for (uint32_t oi = 0; oi < data[ti].size(); ++oi)
        v.push_back({WorldIndex{ti, oi}, data[ti][oi]->bounding_box()});
    return v;
  }
  size_t len() const {
    size_t n = 0;
    for (const auto& c : data) n += c.size();
    return n;
  }
  AABB bounding_box() const override { return aabb; }
  // nested inside another object: members keep the enclosing top-level WorldIndex
  void lower(Lowerer& L) const override {
    for (const auto& col : data)
      for (const auto& obj : col) obj->lower(L);
  }
  // as the scene root: every member is a top-level object of its own
  void lower_root(Lowerer& L) const {
    for (uint32_t ti = 0; ti < data.size(); ++ti)
      for (uint32_t oi = 0; oi < data[ti].size(); ++oi) {
        L.type_index = ti;
        L.object_index = oi;
        data[ti][oi]->lower(L);
      }
  }
  void lower_lights_argument(Lowerer& L) const override {  // world.rs:100-117
    L.lights_mode = MIRT_LIGHTS_WORLD;
    for (const auto& col : data) {
      mirt_light_group g{(uint32_t)L.lights.size(), (uint32_t)col.size()};
      for (const auto& obj : col) obj->lower_light(L);
      L.light_groups.push_back(g);
    }
  }
};

// ---- bvh.rs -----------------------------------------------------------------------------------------------
// Keeps the World and the order in which the reference's tree lists its
// leaves; the device builds its own acceleration structure at upload.
struct BVH final : Hittable {
  World world;
  std::vector<WorldIndex> leaf_order;  // in-order sequence of the leaves of bvh.rs's tree
  AABB root_box;

  static BVH from_world(World w, float time0, float time1) {  // bvh.rs:24-29
    BVH b;
    b.world = std::move(w);
    auto volumes = b.world.volumes();
    b.leaf_order.resize(volumes.size());
    if (!volumes.empty()) b.root_box = order_leaves(b.leaf_order.data(), volumes.data(), volumes.size(), 0, 0);
    (void)time0; (void)time1;  // accepted and ignored, as in the reference
    return b;
  }
  static float key(const std::pair<WorldIndex, AABB>& v, int axis) {
    return axis == 0 ? v.second.minimum.x : (axis == 1 ? v.second.minimum.y : v.second.minimum.z);
  }
  // The recursion of create_node (bvh.rs:31-81) reduced to what decides a
  // tie between equal hit distances: the left-to-right order of the leaves.
  // `out` receives the n leaves of this subtree.  The two halves are independent, so for big inputs the first
  // few levels hand one half to another thread (the reference is single-threaded; the order is the same).
  static AABB order_leaves(WorldIndex* out, std::pair<WorldIndex, AABB>* v, size_t n, int axis, int depth) {
    const int cmp_axis = axis;
    axis = (axis + 1) % 3;
    if (n == 1) { out[0] = v[0].first; return v[0].second; }
    if (n == 2) {
      const bool swap = key(v[0], cmp_axis) > key(v[1], cmp_axis);  // Less/Equal keep, Greater swaps
      out[0] = v[swap ? 1 : 0].first;
      out[1] = v[swap ? 0 : 1].first;
      return AABB::surrounding_box(v[0].second, v[1].second);
    }
    if (n <= 24) {  // a stable insertion sort: same order as the library's stable sort, no scratch allocation
      for (size_t i = 1; i < n; ++i) {
        const auto item = v[i];
        const float k = key(item, cmp_axis);
        size_t j = i;
        for (; j > 0 && key(v[j - 1], cmp_axis) > k; --j) v[j] = v[j - 1];
        v[j] = item;
      }
    } else {
      std::stable_sort(v, v + n, [cmp_axis](const auto& a, const auto& b) { return key(a, cmp_axis) < key(b, cmp_axis); });
    }
    const size_t half = n / 2;
    AABB l, r;
    if (depth < 4 && n >= (size_t)1 << 16) {
      std::thread other([&] { r = order_leaves(out + half, v + half, n - half, axis, depth + 1); });
      l = order_leaves(out, v, half, axis, depth + 1);
      other.join();
    } else {
      l = order_leaves(out, v, half, axis, depth + 1);
      r = order_leaves(out + half, v + half, n - half, axis, depth + 1);
    }
    return AABB::surrounding_box(l, r);
  }
  AABB bounding_box() const override { return root_box; }
  void lower(Lowerer& L) const override { world.lower(L); }
  void lower_root(Lowerer& L) const {
    // visiting in leaf order makes `order` rank equal-t hits the way the
    // tournament of bvh.rs:109-131 does (ties go to the right-hand leaf)
    for (const WorldIndex& wi : leaf_order) {
      L.type_index = wi.type_index;
      L.object_index = wi.object_index;
      world.data[wi.type_index][wi.object_index]->lower(L);
    }
    // media keep the numbering of a column-order walk whatever order the leaves were visited in
    std::stable_sort(L.media.begin(), L.media.end(), [](const mirt_medium& a, const mirt_medium& b) {
      return a.type_index != b.type_index ? a.type_index < b.type_index : a.object_index < b.object_index;
    });
  }
};

// ---- camera.rs ------------------------------------------------------------------------------------------------
struct Camera {
  Point3 origin, lower_left_corner;
  Vec3 horizontal, vertical, u, v, w;
  float lens_radius = 0.0f, time0 = 0.0f, time1 = 0.0f;
  Camera(const Point3& look_from, const Point3& look_at, const Vec3& v_up, float vfov, float aspect_ratio,
         float aperture, float focus_dist, float time0_, float time1_) {  // camera.rs:19-55
    const float theta = to_radians(vfov);
    const float h = std::tan(theta / 2.0f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect_ratio * viewport_height;
    w = (look_from - look_at).unit();
    u = v_up.cross(w).unit();
    v = w.cross(u);
    origin = look_from;
    horizontal = (focus_dist * viewport_width) * u;
    vertical = (focus_dist * viewport_height) * v;
    lower_left_corner = origin - 0.5f * horizontal - 0.5f * vertical - focus_dist * w;
    lens_radius = aperture / 2.0f;
    time0 = time0_;
    time1 = time1_;
  }
  mirt_camera lower() const {
    mirt_camera c{};
    const Vec3* src[7] = {&origin, &lower_left_corner, &horizontal, &vertical, &u, &v, &w};
    float* dst[7] = {c.origin, c.lower_left_corner, c.horizontal, c.vertical, c.u, c.v, c.w};
    for (int i = 0; i < 7; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    c.lens_radius = lens_radius; c.time0 = time0; c.time1 = time1;
    return c;
  }
};

// ---- lowering entry points -----------------------------------------------------------------------------------------
template <class H>
inline void lower_root(const H& hittable, Lowerer& L) {  // any single Hittable: WorldIndex (0,0)
  L.type_index = 0;
  L.object_index = 0;
  hittable.lower(L);
}
inline void lower_root(const World& w, Lowerer& L) { w.lower_root(L); }
inline void lower_root(const BVH& b, Lowerer& L) { b.lower_root(L); }

inline mirt_scene_desc make_desc(const Lowerer& L, float time0, float time1) {
  mirt_scene_desc d{};
  d.spheres = L.spheres.data(); d.n_spheres = (uint32_t)L.spheres.size();
  d.moving_spheres = L.moving_spheres.data(); d.n_moving_spheres = (uint32_t)L.moving_spheres.size();
  d.rects = L.rects.data(); d.n_rects = (uint32_t)L.rects.size();
  d.boxes = L.boxes.data(); d.n_boxes = (uint32_t)L.boxes.size();
  d.chains = L.chains.data(); d.n_chains = (uint32_t)L.chains.size();
  d.chain_ops = L.chain_ops.data(); d.n_chain_ops = (uint32_t)L.chain_ops.size();
  d.media = L.media.data(); d.n_media = (uint32_t)L.media.size();
  d.materials = L.materials.data(); d.n_materials = (uint32_t)L.materials.size();
  d.textures = L.textures.data(); d.n_textures = (uint32_t)L.textures.size();
  d.perlins = L.perlins.data(); d.n_perlins = (uint32_t)L.perlins.size();
  d.images = L.images.data(); d.n_images = (uint32_t)L.images.size();
  d.lights_mode = L.lights_mode;
  d.lights = L.lights.data(); d.n_lights = (uint32_t)L.lights.size();
  d.light_groups = L.light_groups.data(); d.n_light_groups = (uint32_t)L.light_groups.size();
  d.time0 = time0; d.time1 = time1;
  return d;
}

// A scene lowered and resident on one GPU.
class DeviceScene {
 public:
  DeviceScene() = default;
  DeviceScene(const DeviceScene&) = delete;
  DeviceScene& operator=(const DeviceScene&) = delete;
  ~DeviceScene() { reset(); }
  void reset() {
    if (scene_) mirt_scene_destroy(ctx_, scene_);
    if (ctx_) mirt_context_destroy(ctx_);
    scene_ = nullptr;
    ctx_ = nullptr;
  }
  template <class H, class LT>
  Status build(const H& hittable, const LT* lights, float time0, float time1, int device = 0) {
    return build(hittable, lights, time0, time1, std::vector<int>{device}, 0);
  }
  // On several GPUs of the node (the device-side counterpart of the strip threads of renderer.rs:55-87): the scene
  // is kept on every device and a render is shared out between them.
  template <class H, class LT>
  Status build(const H& hittable, const LT* lights, float time0, float time1, const std::vector<int>& devices,
               uint32_t row_bands) {
    reset();
    if (devices.empty()) return Status::err("no device given");
    const auto t0 = std::chrono::steady_clock::now();
    Lowerer L;
    lower_root(hittable, L);
    if (lights) lights->lower_lights_argument(L);
    if (!L.error.empty()) return Status::err("cannot lower scene: " + L.error);
    const mirt_scene_desc desc = make_desc(L, time0, time1);
    const auto t1 = std::chrono::steady_clock::now();
    if (mirt_context_create_multi(devices.data(), (int)devices.size(), row_bands, &ctx_) != MIRT_OK)
      return Status::err(mirt_last_error(nullptr));
    const auto t2 = std::chrono::steady_clock::now();
    if (mirt_scene_upload(ctx_, &desc, &scene_) != MIRT_OK) {
      const std::string e = mirt_last_error(ctx_);
      reset();
      return Status::err(e);
    }
    if (std::getenv("MIRT_VERBOSE")) {
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      std::fprintf(stderr, "rt: lowering %.0f ms, context %.0f ms, scene upload (incl. BVH) %.0f ms\n", ms(t0, t1),
                   ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
    }
    n_media_ = (uint32_t)L.media.size();
    n_prims_ = L.spheres.size() + L.moving_spheres.size() + L.rects.size() + L.boxes.size();
    return Status::ok();
  }
  mirt_context* context() const { return ctx_; }
  mirt_scene* scene() const { return scene_; }
  size_t primitive_count() const { return n_prims_; }
  uint32_t media_count() const { return n_media_; }

 private:
  mirt_context* ctx_ = nullptr;
  mirt_scene* scene_ = nullptr;
  size_t n_prims_ = 0;
  uint32_t n_media_ = 0;
};

// ---- renderer.rs -------------------------------------------------------------------------------------------------------
// The frame the reference leaves when it runs on `threads` threads (renderer.rs:55-87): every thread renders
// tile_height = height / threads rows, the buffer's first row shows scene row threads * tile_height - 1, and the
// height % threads rows at the end of the buffer are never written.  rows_kept() of the frame's rows hold an image;
// place() moves a frame whose rows [height - rows_kept(), height) were rendered (row 0 = top) into that layout.
struct ReferenceStrips {
  uint32_t height, threads;
  bool valid() const { return threads != 0 && threads <= height; }  // (tile_height 0 panics in the reference, :67)
  uint32_t rows_kept() const { return threads * (height / threads); }
  uint32_t first_row() const { return height - rows_kept(); }
  void place(uint8_t* rgb24, uint32_t width) const {
    const size_t pitch = (size_t)width * 3;
    std::memmove(rgb24, rgb24 + pitch * first_row(), pitch * rows_kept());
    std::memset(rgb24 + pitch * rows_kept(), 0, pitch * first_row());
  }
};

class Renderer {
 public:
  static Result<Renderer> create(uint32_t screen_width, uint32_t screen_height, uint32_t samples_per_pixel,
                                 uint32_t max_depth, Color background) {  // renderer.rs:25-41
    Renderer r;
    r.screen_width_ = screen_width;
    r.screen_height_ = screen_height;
    r.samples_per_pixel_ = samples_per_pixel;
    r.max_depth_ = max_depth;
    r.background_ = background;
    r.pitch_ = screen_width * 3;
    r.buffer_.assign((size_t)screen_width * screen_height * 3, 0);
    return Result<Renderer>::ok(std::move(r));
  }

  // renderer.rs:43-92.  `lights` may be null (the reference's None).
  template <class H, class LT>
  Status render(const H& hittable, const Camera& camera, const LT* lights) {
    const auto now = std::chrono::steady_clock::now();
    DeviceScene scene;
    Status st = scene.build(hittable, lights, camera.time0, camera.time1, devices_, row_bands_);
    if (!st.is_ok()) return st;
    const mirt_camera cam = camera.lower();
    mirt_render_params p{};
    p.width = screen_width_;
    p.height = screen_height_;
    p.samples_per_pixel = samples_per_pixel_;
    p.first_sample = 0;
    p.max_depth = max_depth_;
    p.background[0] = background_.x; p.background[1] = background_.y; p.background[2] = background_.z;
    p.seed = seed_;
    p.flags = estimator_;
    const ReferenceStrips strips{screen_height_, strip_threads_};
    if (strip_threads_ != 0) {
      if (!strips.valid()) return Status::err("more strip threads than rows");
      p.first_row = strips.first_row();
      p.n_rows = strips.rows_kept();
    }
    if (mirt_render(scene.context(), scene.scene(), &cam, &p, buffer_.data(), nullptr, &stats_) != MIRT_OK)
      return Status::err(mirt_last_error(scene.context()));
    if (strip_threads_ != 0) strips.place(buffer_.data(), screen_width_);
    const auto delta = std::chrono::steady_clock::now() - now;
    std::printf("Rendered in %lldms\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(delta).count());
    return Status::ok();
  }
  template <class H>
  Status render(const H& hittable, const Camera& camera) {
    return render<H, World>(hittable, camera, nullptr);
  }

  // renderer.rs:94-138 shows the buffer in an SDL window; here it is written
  // as a binary PPM (RT_PRESENT_PATH, default "render.ppm").
  Status present() {
    const char* env = std::getenv("RT_PRESENT_PATH");
    return write_ppm(env && *env ? env : "render.ppm");
  }
  Status write_ppm(const std::string& path) const {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return Status::err("cannot open " + path);
    std::fprintf(f, "P6\n%u %u\n255\n", screen_width_, screen_height_);
    const size_t n = std::fwrite(buffer_.data(), 1, buffer_.size(), f);
    std::fclose(f);
    return n == buffer_.size() ? Status::ok() : Status::err("short write on " + path);
  }

  // Not in the reference (its buffer is private): access for tests and tools.
  const std::vector<uint8_t>& buffer() const { return buffer_; }
  const mirt_stats& stats() const { return stats_; }
  void set_seed(uint64_t s) { seed_ = s; }
  void set_device(int d) { devices_.assign(1, d); }
  // Share every render out between these GPUs (row_bands bands of rows times sample ranges; 0 = sample ranges only).
  void set_devices(const std::vector<int>& devices, uint32_t row_bands = 0) {
    if (!devices.empty()) devices_ = devices;
    row_bands_ = row_bands;
  }
  // 0 (the default): every row of the frame is rendered.  t > 0: the frame as the reference leaves it on a machine
  // with t hardware threads (renderer.rs:56-71): the height % t rows it never renders stay black at the end of the
  // buffer and the image sits that many rows higher.
  void set_strip_threads(uint32_t t) { strip_threads_ = t; }
  // Opt-in estimator changes (MIRT_RENDER_EVEN_LIGHTS | MIRT_RENDER_DROP_NONFINITE | MIRT_RENDER_STRATA(n)); the
  // reference has none of them, and with 0 (the default) the render is the reference's estimator.
  void set_estimator(uint32_t flags) {
    estimator_ = flags & (MIRT_RENDER_EVEN_LIGHTS | MIRT_RENDER_DROP_NONFINITE | MIRT_RENDER_STRATA_MASK);
  }

 private:
  Renderer() = default;
  uint32_t screen_width_ = 0, screen_height_ = 0, samples_per_pixel_ = 0, max_depth_ = 0;
  Color background_;
  uint32_t pitch_ = 0;
  std::vector<uint8_t> buffer_;
  mirt_stats stats_{};
  uint64_t seed_ = 7;
  uint32_t estimator_ = 0;
  uint32_t strip_threads_ = 0;
  std::vector<int> devices_{0};
  uint32_t row_bands_ = 0;
};

}  // namespace rt
