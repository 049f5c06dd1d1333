// gfx950 kernels of the wavefront path tracer.
//
//   k_render  : the whole render in one launch.  Every block owns a region of the path pool and goes round
//                 start  new camera paths in the region's free slots (renderer.rs:215-218, camera.rs:57-68)
//                 walk   every live ray to its closest surface (world.rs:75-94 / bvh.rs:148-152) and file the
//                        path under the material it hit
//                 shade  material by material: media, hit record, emit / scatter / pdfs (renderer.rs:150-189);
//                        finished paths go into the image sums, survivors into the region's other buffer
//               until the frame is done.  Tracing caller-supplied rays is the same kernel with its first and
//               last phase swapped for a load and a store.
//   k_resolve : renderer.rs:232-259
//
// Built with -ffp-contract=off: the intersection arithmetic has to round exactly like the scalar f32 code of
// the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "scene_device.h"

namespace mirt {

namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kPi = 3.14159265358979323846264338327950288f;
constexpr float kTMin = 0.001f;  // renderer.rs:150

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vec3.rs:119-121
__device__ __forceinline__ float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }       // vec3.rs:94-96
__device__ __forceinline__ float len(V3 a) { return sqrtf(len2(a)); }
__device__ __forceinline__ V3 unit(V3 a) {  // vec3.rs:103-106
  const float l = len(a);
  return v3(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ V3 cross(V3 a, V3 b) {  // vec3.rs:123-129
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 xyz(float4 f) { return v3(f.x, f.y, f.z); }

struct Ray {
  V3 o, d;
  float time;
};
__device__ __forceinline__ V3 ray_at(const Ray& r, float t) { return r.o + r.d * t; }  // ray.rs:19-21

// ---- Philox4x32-10, keyed (pixel, sample, bounce*4+stage, block) ------------------
constexpr uint32_t STAGE_CAMERA = 0, STAGE_MEDIA = 1, STAGE_SHADE = 2;

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 multiply gives both halves of a product (v_mad_u64_u32)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

struct Rng {
  uint32_t c0, c1, c2, k0, k1;
  uint32_t next;
  uint32_t block;
  uint4 v;
};
__device__ __forceinline__ Rng rng_begin(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stage,
                                         uint32_t k0, uint32_t k1) {
  Rng r;
  r.c0 = pixel; r.c1 = sample; r.c2 = bounce * 4u + stage; r.k0 = k0; r.k1 = k1;
  r.next = 0; r.block = 0xFFFFFFFFu; r.v = make_uint4(0, 0, 0, 0);
  return r;
}
__device__ __forceinline__ float rng_at(Rng& r, uint32_t index) {  // uniform in [0,1), 24 bits
  const uint32_t b = index >> 2;
  if (b != r.block) {
    r.v = philox4x32_10(r.c0, r.c1, r.c2, b, r.k0, r.k1);
    r.block = b;
  }
  const uint32_t l = index & 3u;
  const uint32_t x = l == 0 ? r.v.x : (l == 1 ? r.v.y : (l == 2 ? r.v.z : r.v.w));
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float rng_next(Rng& r) { return rng_at(r, r.next++); }
__device__ __forceinline__ float rng_range(Rng& r, float lo, float hi) { return lo + (hi - lo) * rng_next(r); }

// Where the traversal reads nodes, primitive records and wrapper chains from: the block's LDS copies
// or global memory.  PSTRIDE (template parameter of the users) is the record stride in float4s.
struct SceneTables {
  const float4* nodes;
  const float4* prims;
  const float4* chains;  // two quads per chain, see ChainHead
  const float4* chain_ops;
  float min_radius;  // smallest sphere radius of the scene (four-child walk: the box padding depends on it)
};
// What shading reads besides the geometry: materials, textures and their tables, the lights to sample.
struct ShadeTables {
  const DMaterial* materials;
  const DTexture* textures;
  const float4* perlin_vec;     // 256 per table (xyz, -)
  const uint32_t* perlin_perm;  // 768 per table: perm_x, perm_y, perm_z
  uint32_t perlin_lds;          // LDS address of the tables' copies (PERLIN_LDS_BYTES each: vectors, permutations as bytes), 0 = none
  const DImage* images;
  const uint8_t* image_data;
  const DLight* lights;
  const uint2* light_groups;  // (first_light, n_lights)
  uint32_t lights_mode, n_lights, n_light_groups;
};
// Both from a scene description in any address space (kernel argument or constant memory).
#define MIRT_GLOBAL_TABLES(sc) SceneTables{(sc).nodes, (sc).prims, (sc).chains, (sc).chain_ops, (sc).min_radius}
#define MIRT_SHADE_TABLES(sc)                                                                                   \
  ShadeTables{(sc).materials, (sc).textures, (sc).perlin_vec, (sc).perlin_perm, 0u, (sc).images, (sc).image_data, \
              (sc).lights,    (sc).light_groups, (sc).lights_mode, (sc).n_lights, (sc).n_light_groups}

// ---- wrappers (transforms.rs) --------------------------------------------------
// A chain's head says what the chain does when it is one of the common shapes -- an optional translate, an
// optional rotation, an optional translate, in this order, by the offset and angle the head carries -- so
// that applying it is straight-line code; any other chain (CHAIN_GENERIC) is a walk through its operations.
constexpr uint32_t CHAIN_T_BEFORE = 1, CHAIN_ROTATE = 2, CHAIN_T_AFTER = 4, CHAIN_GENERIC = 8;
struct ChainHead {
  V3 offset;
  float sin_t, cos_t;
  uint32_t shape, first_op, n_ops;
};
__device__ __forceinline__ ChainHead chain_head(const SceneTables& tb, uint32_t chain) {
  const float4 a = tb.chains[2 * (size_t)chain], b = tb.chains[2 * (size_t)chain + 1];
  const uint32_t w = __float_as_uint(a.w);
  return ChainHead{v3(a.x, a.y, a.z), b.x, b.y, w & 15u, w >> 12, (w >> 8) & 15u};
}
__device__ __forceinline__ void rotate_y(float s, float co, Ray& r) {  // transforms.rs:82-91
  const float ox = co * r.o.x - s * r.o.z;
  const float oz = s * r.o.x + co * r.o.z;
  const float dx = co * r.d.x - s * r.d.z;
  const float dz = s * r.d.x + co * r.d.z;
  r.o.x = ox; r.o.z = oz; r.d.x = dx; r.d.z = dz;
}
// Applies the wrappers of a chain to a ray, outermost first (transforms.rs:16, 82-91).
__device__ __forceinline__ Ray chain_forward(const SceneTables& tb, uint32_t chain, Ray r) {
  const ChainHead h = chain_head(tb, chain);
  if (h.shape & CHAIN_GENERIC) {
    for (uint32_t k = 0; k < h.n_ops; ++k) {
      const float4 op = tb.chain_ops[h.first_op + k];
      const uint32_t kind = __float_as_uint(op.x);
      if (kind == MIRT_OP_TRANSLATE) r.o = r.o - v3(op.y, op.z, op.w);
      else if (kind == MIRT_OP_ROTATE_Y) rotate_y(op.y, op.z, r);
    }
  } else {
    if (h.shape & CHAIN_T_BEFORE) r.o = r.o - h.offset;
    if (h.shape & CHAIN_ROTATE) rotate_y(h.sin_t, h.cos_t, r);
    if (h.shape & CHAIN_T_AFTER) r.o = r.o - h.offset;
  }
  return r;
}

// ---- primitive tests: t only (objects.rs) ---------------------------------------
// All keep the comparisons of the reference (`root < t_min || root > t_max`)
// except that a NaN root counts as a miss here.
__device__ __forceinline__ bool in_range(float t, float t_min, float t_max) { return t >= t_min && t <= t_max; }

__device__ __forceinline__ bool sphere_t(V3 center, float radius, const Ray& r, float t_min, float t_max, float& t) {
  const V3 os = r.o - center;  // objects.rs:42-59
  const float a = len2(r.d);
  const float half_b = dot(os, r.d);
  const float c = len2(os) - radius * radius;
  const float disc = half_b * half_b - a * c;
  if (disc < 0.0f) return false;
  const float sq = sqrtf(disc);
  float root = (-half_b - sq) / a;
  if (!in_range(root, t_min, t_max)) {
    root = (-half_b + sq) / a;
    if (!in_range(root, t_min, t_max)) return false;
  }
  t = root;
  return true;
}

__device__ __forceinline__ V3 moving_center(float4 qa, float4 qb, float time1, float time) {  // objects.rs:132-135
  const V3 c0 = xyz(qa), c1 = xyz(qb);
  const float time0 = qb.w;
  return c0 + ((time - time0) / (time1 - time0)) * (c1 - c0);
}

// Rect through plane `k` of axis n, spanning [a0,a1] x [b0,b1] on axes a, b.
// axis 0: XY (n=z,a=x,b=y)  1: XZ (n=y,a=x,b=z)  2: YZ (n=x,a=y,b=z)   objects.rs:206-228,261-283,335-357
__device__ __forceinline__ bool rect_core(float on, float dn, float oa, float da, float ob, float db, float a0,
                                          float a1, float b0, float b1, float k, float t_min, float t_max, float& t,
                                          float& pa, float& pb) {
  const float tt = (k - on) / dn;
  if (tt < t_min || tt > t_max) return false;
  const float a = oa + tt * da;
  const float b = ob + tt * db;
  if (a < a0 || a > a1 || b < b0 || b > b1) return false;
  t = tt; pa = a; pb = b;
  return true;
}
// (one call per axis: selecting ray components by a run-time axis makes the
// compiler index a private-memory copy of the ray)
__device__ __forceinline__ bool rect_t(uint32_t axis, float a0, float a1, float b0, float b1, float k, const Ray& r,
                                       float t_min, float t_max, float& t, float& pa, float& pb) {
  if (axis == 0) return rect_core(r.o.z, r.d.z, r.o.x, r.d.x, r.o.y, r.d.y, a0, a1, b0, b1, k, t_min, t_max, t, pa, pb);
  if (axis == 1) return rect_core(r.o.y, r.d.y, r.o.x, r.d.x, r.o.z, r.d.z, a0, a1, b0, b1, k, t_min, t_max, t, pa, pb);
  return rect_core(r.o.x, r.d.x, r.o.y, r.d.y, r.o.z, r.d.z, a0, a1, b0, b1, k, t_min, t_max, t, pa, pb);
}

// Box3d::hit: six rects in the reference's order with a shrinking t_max (objects.rs:396-436).
// face: 0 xy@max.z, 1 xy@min.z, 2 xz@min.y, 3 xz@max.y, 4 yz@min.x, 5 yz@max.x
__device__ __forceinline__ bool box_t(V3 mn, V3 mx, const Ray& r, float t_min, float t_max, float& t, int& face,
                                      float& pa, float& pb) {
  bool any = false;
  float closest = t_max;
  float tt, a, b;
  if (rect_t(0, mn.x, mx.x, mn.y, mx.y, mx.z, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 0; pa = a; pb = b; }
  if (rect_t(0, mn.x, mx.x, mn.y, mx.y, mn.z, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 1; pa = a; pb = b; }
  if (rect_t(1, mn.x, mx.x, mn.z, mx.z, mn.y, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 2; pa = a; pb = b; }
  if (rect_t(1, mn.x, mx.x, mn.z, mx.z, mx.y, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 3; pa = a; pb = b; }
  if (rect_t(2, mn.y, mx.y, mn.z, mx.z, mn.x, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 4; pa = a; pb = b; }
  if (rect_t(2, mn.y, mx.y, mn.z, mx.z, mx.x, r, t_min, closest, tt, a, b) && !(tt != tt)) { any = true; closest = tt; face = 5; pa = a; pb = b; }
  t = closest;
  return any;
}

// Tests the primitive whose record is number `index` against the world-space ray r: the ray goes
// through the record's wrapper chain, then through the kind's own test.  Outputs the kind and, for
// the caller's bookkeeping, the record's order rank and material word; `face` is the side hit of a box.
// (qa, qb, qc: quads 0 to 2 of the primitive's record, read by the caller -- possibly some steps ahead of the test;
// a fixed sphere has nothing in qb)
__device__ __forceinline__ bool record_quads_t(const SceneTables& tb, float4 qa, float4 qb, float4 qc, const Ray& r,
                                               float t_min, float t_max, float& t, uint32_t& tag, uint32_t& order,
                                               uint32_t& material, uint32_t& face) {
  const uint32_t chain_tag = __float_as_uint(qc.x);
  tag = chain_tag & 3u;
  order = __float_as_uint(qc.y);
  material = __float_as_uint(qc.z);
  const uint32_t chain = chain_tag >> 2;
  const Ray lr = chain != 0 ? chain_forward(tb, chain, r) : r;
  if (tag <= TAG_MOVING_SPHERE) {
    // one copy of the sphere test serves fixed and moving spheres, wrapped or not: the lanes of a wave that
    // hold any of them run it together
    V3 center = xyz(qa);
    if (tag == TAG_MOVING_SPHERE) center = moving_center(qa, qb, qc.w, lr.time);
    return sphere_t(center, qa.w, lr, t_min, t_max, t);
  } else if (tag == TAG_RECT) {
    float pa, pb;
    const bool ok = rect_t(__float_as_uint(qb.y), qa.x, qa.y, qa.z, qa.w, qb.x, lr, t_min, t_max, t, pa, pb);
    return ok && !(t != t);
  } else {
    int side = 0;
    float pa, pb;
    const bool ok = box_t(xyz(qa), xyz(qb), lr, t_min, t_max, t, side, pa, pb);
    face = (uint32_t)side;
    return ok;
  }
}

template <int PSTRIDE>
__device__ __forceinline__ bool record_t(const SceneTables& tb, uint32_t index, const Ray& r, float t_min, float t_max,
                                         float& t, uint32_t& tag, uint32_t& order, uint32_t& material,
                                         uint32_t& face) {
  const float4* p = tb.prims + PSTRIDE * (size_t)index;
  // (all three quads at once: the second one, which only a fixed sphere does without, sits between the other two, and
  // asking for it once the kind is known would be a second wait -- LDS or memory; +1.4 % on final_scene.  Asking for
  // the stack's top entry ahead of a step that may need it, with the node or before the test, did not pay.)
  const float4 qa = p[0], qb = p[1], qc = p[2];
  return record_quads_t(tb, qa, qb, qc, r, t_min, t_max, t, tag, order, material, face);
}

// ---- conservative box test ------------------------------------------------------
// Entry distance of the box grown by L * (L * growth + linear), L the largest |coordinate| of the box relative to the
// ray origin, or miss (the rule and its constants: scene_device.h).  inv = 1/d per component.
__device__ __forceinline__ bool slab(V3 bmin, V3 bmax, V3 o, V3 inv, float t_min, float t_max, float growth, float linear,
                                     float& t_near) {
  const V3 lo = bmin - o, hi = bmax - o;
  const float l = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))),
                        fmaxf(fabsf(lo.z), fabsf(hi.z)));
  const float pad = l * __builtin_fmaf(l, growth, linear);
  // (lo - pad) * inv and (hi + pad) * inv, the padding folded into one fused multiply-add per plane (this
  // test only has to be conservative; its rounding is part of the padding)
  const float ex = pad * inv.x, ey = pad * inv.y, ez = pad * inv.z;
  const float x0 = __builtin_fmaf(lo.x, inv.x, -ex), x1 = __builtin_fmaf(hi.x, inv.x, ex);
  const float y0 = __builtin_fmaf(lo.y, inv.y, -ey), y1 = __builtin_fmaf(hi.y, inv.y, ey);
  const float z0 = __builtin_fmaf(lo.z, inv.z, -ez), z1 = __builtin_fmaf(hi.z, inv.z, ez);
  // fminf/fmaxf drop a NaN operand (0 * inf on a slab face): the axis then does not constrain
  const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
  t_near = tn;
  return tn <= tf;
}

// Path records go past the caches (the non-temporal hint): a record is written once and read once, a phase or a round
// later, and the pool is far larger than any cache; what the caches hold is then the tree's nodes and the primitive
// records of a scene in global memory, the textures, and the image's sums.  (+1.4 % on final_scene, +3 % on
// cornel_box, +4 % on 10 M spheres; the same hint on the primitive records, which rays share, cost 20 %.)
__device__ __forceinline__ float4 load_quad(const float4* p) {
  typedef float Quad __attribute__((ext_vector_type(4)));
  const Quad v = __builtin_nontemporal_load(reinterpret_cast<const Quad*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_quad(float4* p, float4 v) {
  typedef float Quad __attribute__((ext_vector_type(4)));
  const Quad q = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(q, reinterpret_cast<Quad*>(p));
}

struct Work {
  uint32_t nodes, prims;            // steps down the tree and primitive tests, per lane
  uint32_t wave_nodes, wave_prims;  // the same counted once per wave instruction stream (lane 0 only)
  long long cycles[MIRT_PHASE_COUNT];
};

// How tree references are coded while walking.  The LDS copy of a small scene recodes them into 16 bits
// (leaf bit 15, count - 1 in bits 12..14, first record in bits 0..11) so a stack entry is half the size.
template <bool COMPACT>
struct RefCode {
  static constexpr uint32_t LEAF = COMPACT ? 0x8000u : REF_LEAF;
  static constexpr uint32_t SHIFT = COMPACT ? 12u : 28u;
  static constexpr uint32_t MASK = COMPACT ? 0x0FFFu : 0x0FFFFFFFu;
  static constexpr uint32_t NONE = COMPACT ? 0xFFFFu : REF_NONE;
  static __device__ __forceinline__ uint32_t encode(uint32_t ref) {
    if (!COMPACT || !(ref & REF_LEAF)) return ref;
    return ref == REF_NONE ? NONE : (LEAF | (((ref >> 28) & 7u) << SHIFT) | (ref & MASK));
  }
};
constexpr uint32_t COMPACT_MAX_PRIMS = 4096, COMPACT_MAX_NODES = 32768;
constexpr uint32_t SCAN_MAX_PRIMS = 12;  // scenes up to this size are scanned without the tree

// State of one ray's walk through the tree: the ray, the closest hit so far (smallest t, and of equal t
// the primitive with the higher `order`, which is what the reference's exhaustive scan returns) and the
// reference in hand -- an inner node, a leaf, or NONE once nothing is left to visit.
struct Walk {
  Ray r;
  V3 inv;
  float best_t;
  uint32_t best_ref, best_order, best_material, best_face, cur, sp;
  uint32_t steps;  // tree steps and primitive tests this ray has taken in the current round
  uint32_t node_base;  // first quad of the copy of the node array fitted to the ray's time
};

// The copy of the node array a ray of time `time` walks, as an offset in quads.
__device__ __forceinline__ uint32_t time_slice_base(float time, uint32_t n_slices, float time0, float scale,
                                                    uint32_t n_nodes) {
  if (n_slices <= 1u) return 0u;
  const float f = (time - time0) * scale;
  uint32_t s = f > 0.0f ? (uint32_t)f : 0u;  // (NaN: slice 0; ray times are checked against the scene's range)
  if (s >= n_slices) s = n_slices - 1u;
  return s * n_nodes * 4u;
}

__device__ __forceinline__ void walk_begin(Walk& w, const Ray& r, uint32_t root, uint32_t node_base) {
  w.r = r;
  w.node_base = node_base;
  // the reciprocals only steer the conservative box test: the hardware approximation lies well inside its padding
  w.inv = v3(__builtin_amdgcn_rcpf(r.d.x), __builtin_amdgcn_rcpf(r.d.y), __builtin_amdgcn_rcpf(r.d.z));
  w.best_t = kInf;
  w.best_ref = REF_NONE;
  w.best_order = 0;
  w.best_material = 0;
  w.best_face = 0;
  w.cur = root;
  w.sp = 0;
  w.steps = 0;
}

// A lane's traversal stack: a column of an LDS array whose rows are `stride` entries apart.  Scenes walked from
// global memory have four-child nodes and deeper stacks: the levels beyond the LDS rows live in global memory
// (`over`, rows RENDER_THREADS apart), which only a very unbalanced tree ever reaches.
template <bool COMPACT>
struct WalkStack {
  using Slot = typename std::conditional<COMPACT, uint16_t, uint32_t>::type;
  Slot* lds;
  uint32_t* over;  // this lane's column of the block's overflow rows (unused when COMPACT)
  __device__ __forceinline__ void set(uint32_t k, uint32_t v) const {
    if (COMPACT || k < (uint32_t)WIDE_LDS_STACK) lds[k * RENDER_THREADS] = (Slot)v;
    else over[(size_t)(k - WIDE_LDS_STACK) * RENDER_THREADS] = v;
  }
  __device__ __forceinline__ uint32_t get(uint32_t k) const {
    if (COMPACT || k < (uint32_t)WIDE_LDS_STACK) return lds[k * RENDER_THREADS];
    return over[(size_t)(k - WIDE_LDS_STACK) * RENDER_THREADS];
  }
};

// One step down a two-child tree (w.cur must be an inner node): both child boxes are tested, the nearer one that
// is hit becomes the reference in hand, the other one waits on the stack.
template <bool COMPACT>
__device__ __forceinline__ void node_step(Walk& w, const SceneTables& tb, float t_min, const WalkStack<COMPACT>& st) {
  using RC = RefCode<COMPACT>;
  const float4* n = tb.nodes + (w.node_base + 4 * (size_t)w.cur);
  const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
  float tl, tr;
  const bool hl = slab(xyz(n0), xyz(n1), w.r.o, w.inv, t_min, w.best_t, n2.w, BOX_GROW_LINEAR, tl);
  const bool hr = slab(xyz(n2), xyz(n3), w.r.o, w.inv, t_min, w.best_t, n3.w, BOX_GROW_LINEAR, tr);
  const uint32_t left = __float_as_uint(n0.w), right = __float_as_uint(n1.w);
  if (hl && hr) {
    const bool left_first = tl <= tr;
    st.set(w.sp++, left_first ? right : left);
    w.cur = left_first ? left : right;
  } else if (hl) {
    w.cur = left;
  } else if (hr) {
    w.cur = right;
  } else {
    w.cur = w.sp ? st.get(--w.sp) : RC::NONE;
  }
}

// One step down a four-child tree (layout: QuantNode in bvh_build.h, one 64-byte line): the four boxes are
// decoded and tested, the children that are hit are put in order of their entry distance, the nearest becomes the
// reference in hand and the others wait on the stack, nearest on top.
__device__ __forceinline__ float byte_of(uint32_t word, int k) { return (float)((word >> (8 * k)) & 255u); }  // v_cvt_f32_ubyteK
__device__ __forceinline__ void wide_step(Walk& w, const SceneTables& tb, float t_min, const WalkStack<false>& st) {
  // (four-child nodes are only ever walked from global memory: said outright, the loads stay global loads wherever
  // the address came from)
  typedef const __attribute__((address_space(1))) float4* GlobalQuads;
  const GlobalQuads n = (GlobalQuads)tb.nodes + 4 * (size_t)w.cur;
  auto quad = [&](int k) { return make_float4(n[k].x, n[k].y, n[k].z, n[k].w); };
  const float4 q0 = quad(0), q1 = quad(1), q2 = quad(2), rq = quad(3);
  uint32_t r0 = __float_as_uint(rq.x), r1 = __float_as_uint(rq.y), r2 = __float_as_uint(rq.z), r3 = __float_as_uint(rq.w);
  const float sx = q0.w, sy = q2.z, sz = q2.w;  // the grid's spacing per axis
  // Per axis the plane the ray meets first is the lower one when it travels upwards and the upper one otherwise:
  // picked once per node, on the packed bytes of the four children, this saves the per-box min / max of the two.
  const bool up_x = !(w.inv.x < 0.0f), up_y = !(w.inv.y < 0.0f), up_z = !(w.inv.z < 0.0f);
  const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z),
                 hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
  const uint32_t nx = up_x ? lox : hix, ny = up_y ? loy : hiy, nz = up_z ? loz : hiz;
  const uint32_t fx = up_x ? hix : lox, fy = up_y ? hiy : loy, fz = up_z ? hiz : loz;
  // The builder checked every plane as P = fma(byte, s, corner), rounded once; the test wants (P - o) / d -+ pad / |d|.
  // It is taken per plane in one fused multiply-add, byte * (s / d) + ((corner - o) / d -+ pad / |d|): the second
  // term once per node and axis, one padding for the node's four children.
  //  * The padding is taken with F over the node's whole grid, corner - o to corner - o + 255 s per axis, which
  //    holds every child's planes (rounding is monotonic) and is the node's own extent to a few roundings: at least
  //    each child's own F.
  //  * Rounding, with u = 2^-24, s the spacing, and in units of length (times |1/d|): s / d rounds by u s, 255 u s
  //    at the farthest plane; rel = corner - o rounds by u |rel|; the node's term by u (|rel| + pad); the sum by
  //    u |P - o|; and the builder's own rounding of P, u |P| <= u (|corner| + 255 s), is not repeated this way.  A
  //    plane tested here lies within 2 u |rel| + 255 u s + u |P| + u pad of the one the builder checked, besides the
  //    u |P - o| any way of testing it has.  3 u of |rel| + |corner| + 255 s, taken over the axes and with 5 % to
  //    spare for u pad and what it moves the padding's own length by, is added to the padding (`slack`): the planes
  //    then lie at least as far out as the ones the builder checked.
  //  * A ray that runs in a plane's direction (d = 0 on an axis) has 1/d infinite: every product and sum on that axis
  //    is then infinite with the sign of "does not constrain" or a NaN, which fminf / fmaxf drop -- no box is missed.
  const V3 rel = xyz(q0) - w.r.o;
  const V3 end = v3(__builtin_fmaf(255.0f, sx, rel.x), __builtin_fmaf(255.0f, sy, rel.y), __builtin_fmaf(255.0f, sz, rel.z));
  const float mx = fmaxf(fabsf(rel.x), fabsf(end.x)), my = fmaxf(fabsf(rel.y), fabsf(end.y)),
              mz = fmaxf(fabsf(rel.z), fabsf(end.z));
  const float slack = BOX_DECODE_SLACK * __builtin_fmaf(255.0f, fmaxf(fmaxf(sx, sy), sz),
                                                       fmaxf(fmaxf(fabsf(rel.x), fabsf(rel.y)), fabsf(rel.z)) +
                                                           fmaxf(fmaxf(fabsf(q0.x), fabsf(q0.y)), fabsf(q0.z)));
  // How far off a ball of radius >= r the f32 sphere test may still report a hit, F away: sqrt(r^2 + A F^2 + B F r) - r
  // (scene_device.h), written x / (sqrt(r^2 + x) + r) so that nothing cancels.  v_sqrt_f32 and v_rcp_f32 are good
  // to a unit in the last place, part of the 5 % margin.  (The sum of the three farthest coordinates instead of
  // their length was 8 % more tree steps and a third more primitive tests at 10 M spheres.)
  const float reach = __builtin_amdgcn_sqrtf(__builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx)));
  const float r_min = tb.min_radius;
  const float x = reach * __builtin_fmaf(reach, HALO_QUADRATIC, HALO_LINEAR * r_min);
  const float halo = x * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(__builtin_fmaf(r_min, r_min, x)) + r_min);
  const float pad = __builtin_fmaf(HALO_MARGIN, halo, __builtin_fmaf(BOX_GROW_FAR_LINEAR, reach, slack));
  const float ex = pad * fabsf(w.inv.x), ey = pad * fabsf(w.inv.y), ez = pad * fabsf(w.inv.z);
  const V3 step = v3(sx * w.inv.x, sy * w.inv.y, sz * w.inv.z);
  const V3 from = v3(__builtin_fmaf(rel.x, w.inv.x, -ex), __builtin_fmaf(rel.y, w.inv.y, -ey), __builtin_fmaf(rel.z, w.inv.z, -ez));
  const V3 till = v3(__builtin_fmaf(rel.x, w.inv.x, ex), __builtin_fmaf(rel.y, w.inv.y, ey), __builtin_fmaf(rel.z, w.inv.z, ez));
  auto child = [&](int k, float& t_near) {
    const float tn = fmaxf(fmaxf(__builtin_fmaf(byte_of(nx, k), step.x, from.x), __builtin_fmaf(byte_of(ny, k), step.y, from.y)),
                           fmaxf(__builtin_fmaf(byte_of(nz, k), step.z, from.z), t_min));
    const float tf = fminf(fminf(__builtin_fmaf(byte_of(fx, k), step.x, till.x), __builtin_fmaf(byte_of(fy, k), step.y, till.y)),
                           fminf(__builtin_fmaf(byte_of(fz, k), step.z, till.z), w.best_t));
    t_near = tn;
    return tn <= tf;
  };
  float t0, t1, t2, t3;
  const bool h0 = child(0, t0) && r0 != REF_NONE;
  const bool h1 = child(1, t1) && r1 != REF_NONE;
  const bool h2 = child(2, t2) && r2 != REF_NONE;
  const bool h3 = child(3, t3) && r3 != REF_NONE;
  // entry distances are >= t_min > 0, so their bit patterns order like the numbers; a miss sorts behind every hit
  uint32_t k0 = h0 ? __float_as_uint(t0) : 0xFFFFFFFFu, k1 = h1 ? __float_as_uint(t1) : 0xFFFFFFFFu;
  uint32_t k2 = h2 ? __float_as_uint(t2) : 0xFFFFFFFFu, k3 = h3 ? __float_as_uint(t3) : 0xFFFFFFFFu;
  auto order = [](uint32_t& ka, uint32_t& ra, uint32_t& kb, uint32_t& rb) {
    const bool swap = kb < ka;
    const uint32_t k = swap ? kb : ka, r = swap ? rb : ra;
    kb = swap ? ka : kb; rb = swap ? ra : rb;
    ka = k; ra = r;
  };
  order(k0, r0, k1, r1); order(k2, r2, k3, r3); order(k0, r0, k2, r2); order(k1, r1, k3, r3); order(k1, r1, k2, r2);
  const uint32_t n_hit = (uint32_t)h0 + (uint32_t)h1 + (uint32_t)h2 + (uint32_t)h3;
  if (n_hit == 0u) {
    w.cur = w.sp ? st.get(--w.sp) : REF_NONE;
  } else {
    if (n_hit > 3u) st.set(w.sp++, r3);
    if (n_hit > 2u) st.set(w.sp++, r2);
    if (n_hit > 1u) st.set(w.sp++, r1);
    w.cur = r0;
  }
}

// Tests the first primitive of the leaf in hand (w.cur must be a leaf other than NONE); the rest of the
// leaf stays in hand, and after its last primitive the next reference comes off the stack.
template <int PSTRIDE, bool COMPACT>
__device__ __forceinline__ void prim_step(Walk& w, const SceneTables& tb, float t_min, const WalkStack<COMPACT>& st) {
  using RC = RefCode<COMPACT>;
  const uint32_t index = w.cur & RC::MASK;
  float t;
  uint32_t tag, order, material, face = 0;
  if (record_t<PSTRIDE>(tb, index, w.r, t_min, w.best_t, t, tag, order, material, face)) {
    // t == best_t: the later one in the reference's visiting order wins
    if (t < w.best_t || w.best_ref == REF_NONE || order > w.best_order) {
      w.best_t = t; w.best_ref = make_leaf(tag, index); w.best_material = material; w.best_order = order;
      w.best_face = face;
    }
  }
  if ((w.cur >> RC::SHIFT) & 7u) w.cur += 1u - (1u << RC::SHIFT);  // one fewer left, starting one further
  else w.cur = w.sp ? st.get(--w.sp) : RC::NONE;
}

// Scenes in global memory: the first two primitives of the leaf in hand in one turn (one, if the leaf holds one).
// The same two tests in the same order as two calls of prim_step -- the second sees the closest hit the first
// left -- but both records are sent for before either is used, so the lane waits for memory once, not four times
// (record and second quad, twice).  `pairs` false: one primitive per turn.  Returns the number of primitives tested.
// (Reading a leaf's first record ahead, when the leaf comes to hand and other lanes still take tree steps, was
// measured on top of this: 4 % slower -- eight more registers held through the tree steps.)
template <int PSTRIDE>
__device__ __forceinline__ uint32_t prim_pair_step(Walk& w, const SceneTables& tb, float t_min, const WalkStack<false>& st,
                                                   bool pairs) {
  const uint32_t index = ref_index(w.cur);
  const uint32_t left = (w.cur >> 28) & 7u;  // primitives after the first
  const bool two = pairs && left != 0u;
  const float4* const p = tb.prims + PSTRIDE * (size_t)index;
  const float4* const p2 = p + (two ? PSTRIDE : 0);  // (a leaf of one: its only record again, read and not tested)
  const float4 a1 = p[0], b1 = p[1], c1 = p[2], a2 = p2[0], b2 = p2[1], c2 = p2[2];
  auto test = [&](float4 qa, float4 qb, float4 qc, uint32_t at) {
    float t;
    uint32_t tag, order, material, face = 0;
    if (record_quads_t(tb, qa, qb, qc, w.r, t_min, w.best_t, t, tag, order, material, face)) {
      if (t < w.best_t || w.best_ref == REF_NONE || order > w.best_order) {
        w.best_t = t; w.best_ref = make_leaf(tag, at); w.best_material = material; w.best_order = order;
        w.best_face = face;
      }
    }
  };
  test(a1, b1, c1, index);
  if (two) test(a2, b2, c2, index + 1u);
  const uint32_t n = two ? 2u : 1u;
  if (left >= n) w.cur += n - (n << 28);  // so many fewer left, starting so many further
  else w.cur = w.sp ? st.get(--w.sp) : REF_NONE;
  return n;
}

// Both roots of the sphere quadratic (objects.rs:42-53), or false when the discriminant is negative.
__device__ __forceinline__ bool sphere_roots(V3 center, float radius, const Ray& r, float& r1, float& r2) {
  const V3 os = r.o - center;
  const float a = len2(r.d);
  const float half_b = dot(os, r.d);
  const float c = len2(os) - radius * radius;
  const float disc = half_b * half_b - a * c;
  if (disc < 0.0f) return false;
  const float sq = sqrtf(disc);
  r1 = (-half_b - sq) / a;
  r2 = (-half_b + sq) / a;
  return true;
}

// What closest_media needs of one constant medium, staged once per block in LDS.
struct MediumEntry {
  float4 boundary;        // sphere boundary: center, radius
  uint32_t boundary_ref;
  uint32_t boundary_chain;
  uint32_t chain;         // wrappers around the medium itself
  uint32_t material_word;
  float neg_inv_density;
  uint32_t pad[3];
};

__device__ __forceinline__ void stage_media(uint32_t n_media, const DMedium* media, const float4* prims,
                                            MediumEntry* table) {
  if (threadIdx.x < n_media) {
    const DMedium med = media[threadIdx.x];
    MediumEntry e;
    e.boundary = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    e.boundary_ref = med.boundary_ref;
    e.boundary_chain = 0;
    if (ref_tag(med.boundary_ref) == TAG_SPHERE) {
      const float4* p = prims + PRIM_QUADS * (size_t)ref_index(med.boundary_ref);
      e.boundary = p[0];
      e.boundary_chain = __float_as_uint(p[2].x) >> 2;
    }
    e.chain = med.chain;
    e.material_word = med.material_word;
    e.neg_inv_density = med.neg_inv_density;
    e.pad[0] = e.pad[1] = e.pad[2] = 0;
    table[threadIdx.x] = e;
  }
}

// Constant media after the closest surface is known (transforms.rs:134-180).
// Medium m reads draw m of the STAGE_MEDIA stream, so the order media are
// visited in does not matter; an accepted medium hit replaces the current
// closest hit exactly as World::hit does with its shrinking t_max.
// `tb` serves the wrapper chains (possibly from LDS), `gt` the records in global memory.
__device__ __forceinline__ void closest_media(uint32_t n_media, const SceneTables& tb, const SceneTables& gt,
                                              const MediumEntry* table, const Ray& r, float t_min, Rng& rng,
                                              uint32_t& best_ref, float& best_t, uint32_t& best_material) {
  for (uint32_t m = 0; m < n_media; ++m) {
    const MediumEntry med = table[m];
    float h1, h2;
    if (ref_tag(med.boundary_ref) == TAG_SPHERE) {
      // The reference intersects the boundary twice, over (-inf, inf) and over (h1 + 0.0001, inf); both
      // calls solve the same quadratic, so the roots are computed once and picked as Sphere::hit would.
      const Ray lr = med.boundary_chain != 0 ? chain_forward(tb, med.boundary_chain, r) : r;
      float r1, r2;
      if (!sphere_roots(xyz(med.boundary), med.boundary.w, lr, r1, r2)) continue;
      if (in_range(r1, -kInf, kInf)) h1 = r1;
      else if (in_range(r2, -kInf, kInf)) h1 = r2;
      else continue;
      const float t2_min = h1 + 0.0001f;
      if (in_range(r1, t2_min, kInf)) h2 = r1;
      else if (in_range(r2, t2_min, kInf)) h2 = r2;
      else continue;
    } else {
      // boundary-only records are not staged in LDS: read them from global memory
      uint32_t tag, order, material, face;
      if (!record_t<PRIM_QUADS>(gt, ref_index(med.boundary_ref), r, -kInf, kInf, h1, tag, order, material, face)) continue;
      if (!record_t<PRIM_QUADS>(gt, ref_index(med.boundary_ref), r, h1 + 0.0001f, kInf, h2, tag, order, material, face)) continue;
    }
    if (h1 < t_min) h1 = t_min;
    if (h2 > best_t) h2 = best_t;
    if (h1 >= h2) continue;
    if (h1 < 0.0f) h1 = 0.0f;
    const Ray mr = med.chain != 0 ? chain_forward(tb, med.chain, r) : r;
    const float ray_length = len(mr.d);
    const float dist_inside = (h2 - h1) * ray_length;
    const float hit_distance = med.neg_inv_density * log2f(rng_at(rng, m));  // log2, as the reference
    if (hit_distance > dist_inside) continue;
    best_t = h1 + hit_distance / ray_length;
    best_ref = make_leaf(TAG_MEDIUM, m);
    best_material = med.material_word;
  }
}

// ---- full hit record (hittable.rs:8-42) ------------------------------------------
struct Hit {
  V3 p, n;
  float t, u, v;
  V3 on_sphere;    // sphere hits: the outward unit normal, from which (u, v) are worked out when a texture asks
  bool sphere_uv;  // (u, v) still to be worked out from on_sphere
  bool front;
  uint32_t material_word;
};

__device__ __forceinline__ void set_face(Hit& h, const Ray& r, V3 outward) {  // hittable.rs:28-35
  h.front = dot(r.d, outward) < 0.0f;
  h.n = h.front ? outward : -outward;
}

__device__ __forceinline__ void sphere_uv(V3 p, float& u, float& v) {  // objects.rs:17-28
  const float theta = acosf(-p.y);
  const float phi = -atan2f(p.z, p.x) + kPi;
  u = phi / (2.0f * kPi);
  v = theta / kPi;
}

__device__ __forceinline__ void sphere_record(V3 center, float radius, const Ray& r, float t, Hit& h) {
  h.t = t;
  h.p = ray_at(r, t);
  const V3 outward = (h.p - center) / radius;  // objects.rs:61-63
  h.on_sphere = outward;
  h.sphere_uv = true;
  set_face(h, r, outward);
}

__device__ __forceinline__ void rect_record(uint32_t axis, float a0, float a1, float b0, float b1, const Ray& r,
                                            float t, float pa, float pb, Hit& h) {
  h.t = t;
  h.u = (pa - a0) / (a1 - a0);
  h.v = (pb - b0) / (b1 - b0);
  h.p = ray_at(r, t);
  set_face(h, r, axis == 0 ? v3(0.0f, 0.0f, 1.0f) : (axis == 1 ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f)));
}

// Rebuilds the record of the hit that the walk / closest_media found
// at (ref, t), passing it through the wrappers inside-out as the nested
// Hittable::hit calls of the reference do.
// `face` is the side the walk found for a box hit.
template <int PSTRIDE>
__device__ __forceinline__ Hit hit_record(const SceneTables& tb, const MediumEntry* media, uint32_t ref,
                                          const Ray& world_ray, float t_found, uint32_t face) {
  Hit h;
  h.sphere_uv = false;
  h.on_sphere = v3(0.0f, 0.0f, 0.0f);
  h.u = 0.0f; h.v = 0.0f;
  const uint32_t tag = ref_tag(ref), idx = ref_index(ref);
  uint32_t chain;
  const float4* rec = tb.prims + PSTRIDE * (size_t)idx;  // not read for a medium
  if (tag == TAG_MEDIUM) {
    chain = media[idx].chain;
    h.material_word = media[idx].material_word;
  } else {
    const float4 qc = rec[2];
    chain = __float_as_uint(qc.x) >> 2;
    h.material_word = __float_as_uint(qc.z);
  }
  // the ray as the primitive sees it, and the direction each wrapper saw (after it transformed the ray)
  V3 dir_at[MIRT_MAX_CHAIN_OPS];
  Ray r = world_ray;
  ChainHead ch{};
  if (chain != 0) {
    ch = chain_head(tb, chain);
    if (ch.shape & CHAIN_GENERIC) {
      for (uint32_t k = 0; k < ch.n_ops && k < MIRT_MAX_CHAIN_OPS; ++k) {
        const float4 op = tb.chain_ops[ch.first_op + k];
        const uint32_t kind = __float_as_uint(op.x);
        if (kind == MIRT_OP_TRANSLATE) r.o = r.o - v3(op.y, op.z, op.w);
        else if (kind == MIRT_OP_ROTATE_Y) rotate_y(op.y, op.z, r);
        dir_at[k] = r.d;
      }
    } else {
      if (ch.shape & CHAIN_T_BEFORE) r.o = r.o - ch.offset;
      if (ch.shape & CHAIN_ROTATE) rotate_y(ch.sin_t, ch.cos_t, r);
      if (ch.shape & CHAIN_T_AFTER) r.o = r.o - ch.offset;
    }
  }
  if (tag == TAG_MEDIUM) {  // transforms.rs:160-172
    h.t = t_found;
    h.p = ray_at(r, t_found);
    h.n = v3(1.0f, 0.0f, 0.0f);
    h.u = 0.0f; h.v = 0.0f;
    h.front = true;
  } else if (tag <= TAG_MOVING_SPHERE) {
    const float4 g = rec[0];
    V3 center = xyz(g);
    if (tag == TAG_MOVING_SPHERE) center = moving_center(g, rec[1], rec[2].w, r.time);
    sphere_record(center, g.w, r, t_found, h);
  } else {
    // a rect, or the side of a box the walk found (one of its six rects, objects.rs:396-436): one copy of the code
    const float4 g0 = rec[0], g1 = rec[1];
    const bool is_rect = tag == TAG_RECT;
    const uint32_t axis = is_rect ? __float_as_uint(g1.y) : face >> 1;
    const float a0 = is_rect ? g0.x : (axis == 2 ? g0.y : g0.x), a1 = is_rect ? g0.y : (axis == 2 ? g1.y : g1.x);
    const float b0 = is_rect ? g0.z : (axis == 0 ? g0.y : g0.z), b1 = is_rect ? g0.w : (axis == 0 ? g1.y : g1.z);
    // where the ray meets the rect's plane, as the hit test worked it out (objects.rs:212-213)
    const float pa = axis == 2 ? r.o.y + t_found * r.d.y : r.o.x + t_found * r.d.x;
    const float pb = axis == 0 ? r.o.y + t_found * r.d.y : r.o.z + t_found * r.d.z;
    rect_record(axis, a0, a1, b0, b1, r, t_found, pa, pb, h);
  }
  // back out through the wrappers, innermost first
  auto undo_translate = [&](V3 offset, V3 d) {  // transforms.rs:18-19
    h.p = h.p + offset;
    h.front = dot(d, h.n) < 0.0f;
  };
  auto undo_rotate = [&](float s, float co, V3 d) {  // transforms.rs:94-105
    V3 p = h.p, n = h.n;
    p.x = co * h.p.x + s * h.p.z;
    p.z = -s * h.p.x + co * h.p.z;
    n.x = co * h.n.x + s * h.n.z;
    n.z = -s * h.n.x + co * h.n.z;
    h.p = p;
    h.front = dot(d, n) < 0.0f;  // object-space direction, world-space normal
    h.n = h.front ? n : -n;
  };
  if (chain != 0) {
    if (ch.shape & CHAIN_GENERIC) {
      for (int k = (int)ch.n_ops - 1; k >= 0; --k) {
        const float4 op = tb.chain_ops[ch.first_op + k];
        const uint32_t kind = __float_as_uint(op.x);
        const V3 d = dir_at[k < MIRT_MAX_CHAIN_OPS ? k : 0];
        if (kind == MIRT_OP_TRANSLATE) undo_translate(v3(op.y, op.z, op.w), d);
        else if (kind == MIRT_OP_ROTATE_Y) undo_rotate(op.y, op.z, d);
        else h.front = !h.front;  // transforms.rs:199-201
      }
    } else {
      // r.d is the direction after the rotation (if any), world_ray.d the one before it
      if (ch.shape & CHAIN_T_AFTER) undo_translate(ch.offset, r.d);
      if (ch.shape & CHAIN_ROTATE) undo_rotate(ch.sin_t, ch.cos_t, r.d);
      if (ch.shape & CHAIN_T_BEFORE) undo_translate(ch.offset, world_ray.d);
    }
  }
  return h;
}

// ---- textures (texture.rs, perlin.rs) ----------------------------------------------
template <typename Perm, typename Vec>
__device__ __forceinline__ float perlin_sum(Perm perm, Vec vec, V3 p) {  // perlin.rs:26-48,75-93
  const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
  const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
  const int i = (int)fx, j = (int)fy, k = (int)fz;
  const float uu = u * u * (3.0f - 2.0f * u);
  const float vv = v * v * (3.0f - 2.0f * v);
  const float ww = w * w * (3.0f - 2.0f * w);
  float accum = 0.0f;
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        const uint32_t h = (uint32_t)perm[(i + di) & 255] ^ (uint32_t)perm[256 + ((j + dj) & 255)] ^ (uint32_t)perm[512 + ((k + dk) & 255)];
        const V3 c = v3(vec[h].x, vec[h].y, vec[h].z);
        const V3 weight = v3(u - (float)di, v - (float)dj, w - (float)dk);
        accum += ((float)di * uu + (1.0f - (float)di) * (1.0f - uu)) *
                 ((float)dj * vv + (1.0f - (float)dj) * (1.0f - vv)) *
                 ((float)dk * ww + (1.0f - (float)dk) * (1.0f - ww)) * dot(c, weight);
      }
  return accum;
}
__device__ __forceinline__ float perlin_noise(const ShadeTables& sc, uint32_t table, V3 p) {
  if (sc.perlin_lds != 0u) {  // the block's copies: the same numbers, the permutations as bytes
    typedef const __attribute__((address_space(3))) float4* LdsVec;
    typedef const __attribute__((address_space(3))) uint8_t* LdsPerm;
    const uint32_t at = sc.perlin_lds + table * PERLIN_LDS_BYTES;
    return perlin_sum((LdsPerm)(uintptr_t)(at + 256u * 16u), (LdsVec)(uintptr_t)at, p);
  }
  return perlin_sum(sc.perlin_perm + 768 * (size_t)table, sc.perlin_vec + 256 * (size_t)table, p);
}

__device__ __forceinline__ float perlin_turb2(const ShadeTables& sc, uint32_t table, V3 p) {  // perlin.rs:50-60, depth 2
  float accum = 0.0f;
  accum += 1.0f * perlin_noise(sc, table, p);
  accum += 0.5f * perlin_noise(sc, table, p * 2.0f);
  return fabsf(accum);
}

__device__ __forceinline__ uint32_t f32_to_u32_sat(float v) {  // Rust `as u32`
  if (!(v == v) || v <= 0.0f) return 0;
  if (v >= 4294967040.0f) return 0xFFFFFFFFu;
  return (uint32_t)v;
}

__device__ __forceinline__ V3 texture_color(const ShadeTables& sc, uint32_t tex, const Hit& h) {
  const V3 p = h.p;
  DTexture t = sc.textures[tex];
  // a checker has no colour of its own: it picks one of two textures by position (texture.rs:51-60)
#pragma nounroll
  for (int depth = 0; depth < 7 && t.kind == MIRT_TEX_CHECKER; ++depth) {
    const float sines = sinf(10.0f * p.x) * sinf(10.0f * p.y) * sinf(10.0f * p.z);
    t = sc.textures[sines < 0.0f ? t.odd : t.even];
  }
  if (t.kind == MIRT_TEX_SOLID) {
    return v3(t.r, t.g, t.b);
  } else if (t.kind == MIRT_TEX_NOISE) {  // texture.rs:76-82
    const float s = 1.0f + sinf(t.scale * p.z + 10.0f * perlin_turb2(sc, t.index, p));
    return v3(1.0f, 1.0f, 1.0f) * 0.5f * s;
  } else if (t.kind == MIRT_TEX_IMAGE) {  // texture.rs:101-122
    const DImage im = sc.images[t.index];
    float u = h.u, v = h.v;
    if (h.sphere_uv) sphere_uv(h.on_sphere, u, v);
    const float cu = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
    const float cv = 1.0f - (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
    uint32_t i = f32_to_u32_sat(cu * (float)im.width);
    uint32_t j = f32_to_u32_sat(cv * (float)im.height);
    if (i >= im.width) i = im.width - 1;
    if (j >= im.height) j = im.height - 1;
    const uint8_t* px = sc.image_data + im.offset + ((size_t)j * im.width + i) * 3;
    const float scale = 1.0f / 255.0f;
    return v3(scale * (float)px[0], scale * (float)px[1], scale * (float)px[2]);
  }
  return v3(0.0f, 0.0f, 0.0f);  // checkers nested deeper than the loop follows
}

// ---- sampling helpers (vec3.rs, onb.rs, pdf.rs) ------------------------------------
struct Onb {
  V3 u, v, w;
};
__device__ __forceinline__ Onb onb_from_w(V3 n) {  // onb.rs:10-20
  Onb o;
  o.w = unit(n);
  const V3 a = fabsf(o.w.x) > 0.9f ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
  o.v = unit(cross(o.w, a));
  o.u = cross(o.w, o.v);
  return o;
}
__device__ __forceinline__ V3 onb_local(const Onb& o, V3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }  // onb.rs:26-28

__device__ __forceinline__ V3 random_cosine_direction(Rng& rng) {  // vec3.rs:62-74
  const float r1 = rng_next(rng), r2 = rng_next(rng);
  const float z = sqrtf(1.0f - r2);
  const float phi = 2.0f * kPi * r1;
  const float sr = sqrtf(r2);
  float sn, cs;
  sincosf(phi, &sn, &cs);  // one argument reduction for both
  return v3(cs * sr, sn * sr, z);
}
__device__ __forceinline__ V3 random_to_sphere(Rng& rng, float radius, float dist2) {  // vec3.rs:76-88
  const float r1 = rng_next(rng), r2 = rng_next(rng);
  const float z = 1.0f + r2 * (sqrtf(1.0f - radius * radius / dist2) - 1.0f);
  const float phi = 2.0f * kPi * r1;
  const float s = sqrtf(1.0f - z * z);
  float sn, cs;
  sincosf(phi, &sn, &cs);
  return v3(cs * s, sn * s, z);
}
__device__ __forceinline__ V3 random_in_unit_sphere(Rng& rng) {  // vec3.rs:44-51
  for (;;) {
    const float x = rng_range(rng, -1.0f, 1.0f);
    const float y = rng_range(rng, -1.0f, 1.0f);
    const float z = rng_range(rng, -1.0f, 1.0f);
    const V3 p = v3(x, y, z);
    if (len2(p) < 1.0f) return p;
  }
}
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }  // vec3.rs:108-110
__device__ __forceinline__ V3 refract(V3 v, V3 n, float ratio) {                          // vec3.rs:112-117
  const float cos_theta = fminf(dot(-v, n), 1.0f);
  const V3 perp = ratio * (v + cos_theta * n);
  const V3 par = (-sqrtf(fabsf(1.0f - len2(perp)))) * n;
  return perp + par;
}

// pdf_value / random of one light object (objects.rs:82-100, 292-309, hittable.rs:52-57)
__device__ __forceinline__ float light_pdf_value(const DLight& l, V3 origin, V3 dir) {
  if (l.kind == MIRT_LIGHT_XZ_RECT) {
    const float x0 = l.p[0], x1 = l.p[1], z0 = l.p[2], z1 = l.p[3], k = l.p[4];
    // XZRect::hit with (0.0001, inf); comparisons written as in the reference so NaN flows the same way
    const float t = (k - origin.y) / dir.y;
    if (t < 0.0001f || t > kInf) return 0.0f;
    const float x = origin.x + t * dir.x;
    const float z = origin.z + t * dir.z;
    if (x < x0 || x > x1 || z < z0 || z > z1) return 0.0f;
    const float area = (x1 - x0) * (z1 - z0);
    const float dist = t * t * len2(dir);
    // hit.normal is +-(0,1,0): |dot(dir, normal) / |dir||
    const bool front = dir.y < 0.0f;
    const float dn = front ? (dir.x * 0.0f + dir.y * 1.0f + dir.z * 0.0f) : (dir.x * -0.0f + dir.y * -1.0f + dir.z * -0.0f);
    const float cosine = fabsf(dn / len(dir));
    return dist / (cosine * area);
  } else if (l.kind == MIRT_LIGHT_SPHERE) {
    const V3 c = v3(l.p[0], l.p[1], l.p[2]);
    const float radius = l.p[3];
    // Sphere::hit(.., 0.001, inf).is_some() with the reference's comparisons
    const V3 os = origin - c;
    const float a = len2(dir);
    const float half_b = dot(os, dir);
    const float cc = len2(os) - radius * radius;
    const float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return 0.0f;
    const float sq = sqrtf(disc);
    float root = (-half_b - sq) / a;
    if (root < 0.001f || root > kInf) {
      root = (-half_b + sq) / a;
      if (root < 0.001f || root > kInf) return 0.0f;
    }
    const float cos_theta_max = sqrtf(1.0f - radius * radius / len2(c - origin));
    const float solid_angle = 2.0f * kPi * (1.0f - cos_theta_max);
    return 1.0f / solid_angle;
  }
  return 0.0f;
}
__device__ __forceinline__ V3 light_random(const DLight& l, V3 origin, Rng& rng) {
  if (l.kind == MIRT_LIGHT_XZ_RECT) {
    const float rx = rng_range(rng, l.p[0], l.p[1]);
    const float rz = rng_range(rng, l.p[2], l.p[3]);
    return v3(rx, l.p[4], rz) - origin;
  } else if (l.kind == MIRT_LIGHT_SPHERE) {
    const V3 dir = v3(l.p[0], l.p[1], l.p[2]) - origin;
    const Onb uvw = onb_from_w(dir);
    return onb_local(uvw, random_to_sphere(rng, l.p[3], len2(dir)));
  }
  return v3(1.0f, 0.0f, 0.0f);
}
// HittablePdf over the `lights` argument (pdf.rs:56-67, world.rs:100-117)
struct LightList {
  const DLight* lights;
  const uint2* groups;  // (first_light, n_lights)
  uint32_t mode, n_lights, n_groups;
};
__device__ __forceinline__ float lights_pdf_value(const LightList& ll, V3 origin, V3 dir) {
  if (ll.mode == MIRT_LIGHTS_OBJECT) return light_pdf_value(ll.lights[0], origin, dir);
  const float weight = 1.0f / (float)ll.n_lights;
  float sum = 0.0f;
  for (uint32_t i = 0; i < ll.n_lights; ++i) sum += weight * light_pdf_value(ll.lights[i], origin, dir);
  return sum;
}
__device__ __forceinline__ V3 lights_random(const LightList& ll, V3 origin, Rng& rng, bool even) {
  if (ll.mode == MIRT_LIGHTS_OBJECT) return light_random(ll.lights[0], origin, rng);
  if (even) {  // opt-in: uniform over the objects, as lights_pdf_value weighs them
    uint32_t i = (uint32_t)(rng_next(rng) * (float)ll.n_lights);
    if (i >= ll.n_lights) i = ll.n_lights - 1;
    return light_random(ll.lights[i], origin, rng);
  }
  uint32_t g = (uint32_t)(rng_next(rng) * (float)ll.n_groups);
  if (g >= ll.n_groups) g = ll.n_groups - 1;
  const uint2 grp = ll.groups[g];
  uint32_t i = (uint32_t)(rng_next(rng) * (float)grp.y);
  if (i >= grp.y) i = grp.y - 1;
  return light_random(ll.lights[grp.x + i], origin, rng);
}
// One draw of the mixture of the light list and the cosine lobe around uvw.w, and its density (pdf.rs:81-91,
// renderer.rs:166-173).
__device__ __forceinline__ V3 mixture_sample(const LightList& ll, const Onb& uvw, V3 p, Rng& rng, bool even, float& pdf) {
  V3 dir;
  if (rng_next(rng) < 0.5f) dir = lights_random(ll, p, rng, even);  // pdf.rs:85-91
  else dir = onb_local(uvw, random_cosine_direction(rng));
  const float c1 = dot(unit(dir), uvw.w);  // pdf.rs:25-32
  const float cos_pdf = c1 <= 0.0f ? 0.0f : c1 / kPi;
  pdf = 0.5f * lights_pdf_value(ll, p, dir) + 0.5f * cos_pdf;  // pdf.rs:81-83
  return dir;
}

__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {  // material.rs:105-108
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  const float m = 1.0f - cosine;
  const float m2 = m * m;
  return r0 + (1.0f - r0) * (m2 * m2 * m);
}

// ---- wave helpers ---------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Diagnostic build (-DMIRT_CHECKED, `make CHECKED=1`): every index the render kernel computes into global memory is
// tested before use.  The first violation is recorded in WorkCounters::check (site, three values, region) and the
// access is skipped, so a bad index shows up as a message from the host (run_job) instead of a memory fault.
// In the normal build the test is the constant `true` and leaves no code behind.
#ifdef MIRT_CHECKED
__device__ __noinline__ void check_failed(WorkCounters* wc, uint32_t site, uint32_t a, uint32_t b, uint32_t c) {
  if (atomicCAS(&wc->check[0], 0ull, (unsigned long long)site) == 0ull) {
    wc->check[1] = a; wc->check[2] = b; wc->check[3] = c;
    wc->check[4] = blockIdx.x; wc->check[5] = threadIdx.x;
  }
  atomicAdd(&wc->check[6], 1ull);
}
#define MIRT_CHECK(ok, wc, site, a, b, c) ((ok) ? true : (check_failed((wc), (site), (uint32_t)(a), (uint32_t)(b), (uint32_t)(c)), false))
#else
#define MIRT_CHECK(ok, wc, site, a, b, c) true
#endif

// Number of lanes below this one whose bit is set in `mask` (a ballot): two v_mbcnt instructions.
// (Written as popcount(mask & ((1 << lane) - 1)) the compiler forms the lane mask with a 64-bit shift by the lane
// id, v_lshlrev_b64.  On gfx950 the 64-bit shifts read their 32-bit shift amount as a register pair: when the
// register allocator puts the amount in the last register of the wave's allocation, every wave but the first on
// its SIMD reads 0 for it -- tools/shift_hazard_probe.hip -- and a build whose scheduling led to exactly that
// allocation packed all survivors of a batch into one slot.  No variable 64-bit shift is left in this file;
// tools/check_isa.py holds every build to it.)
__device__ __forceinline__ uint32_t lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Reserves one slot per lane that passes `pred` with a single atomic per wave.
__device__ __forceinline__ uint32_t wave_reserve(bool pred, uint32_t* counter) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0ull) return 0;
  const uint32_t lane = lane_id();
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  return base + lanes_below(mask);
}

__device__ __forceinline__ int queue_of(uint32_t material_word) { return (int)(material_word >> 28); }

__device__ __forceinline__ uint32_t fast_div(uint32_t n, FastDiv f) {
  const uint32_t t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
// Pixel of place `idx` within one sample pass: 8-row bands walked column
// by column, so 64 consecutive places cover an 8x8 pixel tile.
__device__ __forceinline__ void pixel_of(uint32_t idx, uint32_t width, uint32_t height, FastDiv by_band, uint32_t& x,
                                         uint32_t& y) {
  const uint32_t band_size = width * 8u;
  const uint32_t band = fast_div(idx, by_band);
  const uint32_t in_band = idx - band * band_size;
  const uint32_t rows = min(8u, height - band * 8u);
  x = rows == 8u ? in_band >> 3 : in_band / rows;  // only the frame's last band can be lower
  y = band * 8u + (in_band - x * rows);
}

// ---- render -----------------------------------------------------------------------
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

// Bookkeeping of one region's loop.  A region belongs to one wave, so nothing here is touched by two waves: it lives
// in LDS only because the phases of a round are separate functions.  Written by the wave's lane 0 between phases.
struct WaveShared {
  uint32_t n_fresh;                     // camera paths started this round, in the slots after the survivors
  uint32_t first_new_lo, first_new_hi;  // number of the first of them among the render's paths
  uint32_t queue_n[N_QUEUES];           // entries in the region's material queues
  uint32_t n_next;                      // survivors written to the other buffer so far
  uint32_t late_n[N_QUEUES];            // entries appended to each queue for the second shading pass
  // Survivors are kept in two streams, split by where they head (up or down): one fills the region's buffer from slot
  // 0 upwards (n_next counts it), the other from the last slot downwards (n_high); the next walk reads the one, then
  // the other, so its waves hold rays of a kind.  n_low: how many of the stored rays the walk is about to read sit
  // at the low end (the rest at the high end); split: whether shading splits at all.
  uint32_t n_high, n_low, split;
  uint32_t pad[9];
};
static_assert(sizeof(WaveShared) == 128, "one LDS row half per wave");
struct RenderShared {
  WaveShared wave[RENDER_THREADS / 64];
  MediumEntry media[MIRT_MAX_MEDIA];
};
constexpr size_t SHARED_BYTES = (sizeof(RenderShared) + 255) / 256 * 256;  // the node array starts on an LDS row

// The launch arguments live in device memory.  Each phase of the kernel reads what it needs through this:
// the empty asm hides where the address came from, so the compiler cannot move those loads up to the start of
// the kernel and hold dozens of scalar registers across phases that never use them; the constant address
// space tells it the memory does not change, so the reads go through the scalar unit.
typedef const __attribute__((address_space(4))) RenderArgs* ConstArgs;
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ ConstArgs pinned(const RenderArgs* p) {
  // (a function's arguments arrive in vector registers even when every lane holds the same value)
  unsigned long long v = ((unsigned long long)uniform((uint32_t)((unsigned long long)p >> 32)) << 32) |
                         uniform((uint32_t)(unsigned long long)p);
  asm volatile("" : "+s"(v));
  return (ConstArgs)v;
}
// The block's LDS as the phases of the kernel get it: the kernel passes the address down.  A function that
// names the dynamic LDS array itself has to look its offset up in a per-kernel table in memory, and the compiler
// repeats that scalar load wherever it needs the address again (inside the tree-step loop, for one).
// (It travels as a plain number taken through an empty asm in the kernel, or the compiler sees that every caller
// passes the same array and puts the lookup back.)
typedef __attribute__((address_space(3))) float4* LdsQuads;
__device__ __forceinline__ float4* lds_base(uint32_t address) {
  return (float4*)(LdsQuads)(uintptr_t)uniform(address);
}

// A wave owns a region of the path pool and takes it through the rounds on its own: the waves of a block share
// the staged scene and nothing else, and after staging no wave ever waits for another one.
__device__ __forceinline__ uint32_t wave_in_block() { return uniform(threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t region_of_wave(ConstArgs A) { return blockIdx.x * A->waves_per_block + wave_in_block(); }
// Between the phases of a round: what the wave's lanes wrote (path records in global memory, counters in LDS) is
// read by other lanes of the same wave.  The waves of a block run on one CU and share its vector cache, so
// waiting for the writes to complete is all it takes; no other wave is involved.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Dynamic shared memory of the render kernel: [RenderShared][nodes][48-byte primitive records][chain ops]
// [chains][materials][textures][lights][light groups][stack] when the scene is staged (LDS_SCENE), else
// [RenderShared][stack].
template <bool LDS_SCENE>
struct RenderLds {
  using Slot = typename std::conditional<LDS_SCENE, uint16_t, uint32_t>::type;
  float4* nodes;
  float4* prims;
  float4* ops;
  float4* chains;     // two quads per chain
  float4* materials;  // one quad per material (DMaterial)
  float4* textures;   // two quads per texture (DTexture)
  float4* lights;     // two quads per light (DLight)
  float4* groups;     // light groups, two to a quad
  Slot* stack;  // this thread's column; rows are RENDER_THREADS entries apart
  __device__ __forceinline__ RenderLds(float4* smem, ConstArgs A) {
    float4* after = smem + SHARED_BYTES / 16;
    nodes = after;
    prims = nodes + (LDS_SCENE ? A->sc.n_nodes * A->sc.n_slices * 4u : 0u);
    ops = prims + (LDS_SCENE ? A->sc.n_prims * PRIM_QUADS_LDS : 0u);
    chains = ops + (LDS_SCENE ? A->sc.n_chain_ops : 0u);
    materials = chains + (LDS_SCENE ? A->sc.n_chains * 2u : 0u);
    textures = materials + (LDS_SCENE ? A->sc.n_materials : 0u);
    lights = textures + (LDS_SCENE ? A->sc.n_textures * 2u : 0u);
    groups = lights + (LDS_SCENE ? A->sc.n_lights * 2u : 0u);
    // the stacks start on an LDS row: a wave's 64 entries of one level then lie in one row
    const uint32_t tables_end = (uint32_t)((groups + (LDS_SCENE ? (A->sc.n_light_groups + 1u) / 2u : 0u)) - smem);  // in quads
    stack = reinterpret_cast<Slot*>(smem + ((tables_end + 15u) & ~15u)) + threadIdx.x;
  }
  // the tables the walk and the shading read the scene's geometry from
  __device__ __forceinline__ SceneTables tables(ConstArgs A) const {
    if (LDS_SCENE) return SceneTables{nodes, prims, chains, ops, 0.0f};
    return MIRT_GLOBAL_TABLES(A->sc);
  }
};

// The three phases of a round are separate functions so that each gets registers of its own: the walk needs
// few and must not spill, shading needs many.

// Camera paths (renderer.rs:215-218, camera.rs:57-68).  A round's new paths are numbered first_new, first_new + 1, ...
// among all paths of the render; path number = pass * pixels + place in the pass.
struct CameraPaths {
  ConstArgs A;
  uint32_t first_pass, first_place;
  __device__ __forceinline__ CameraPaths(ConstArgs args, uint64_t first_new) : A(args) {
    const uint32_t pixels = A->rc.width * A->rc.band_rows;
    // the one long division is done once for the round
    first_pass = uniform((uint32_t)(first_new / pixels));
    first_place = uniform((uint32_t)(first_new - (uint64_t)first_pass * pixels));
  }
  // The k-th of the round's new paths as the first three quads of a path record.
  __device__ __forceinline__ void start(uint32_t k, float4& r0, float4& r1, float4& r2) const {
    const uint32_t width = A->rc.width, height = A->rc.height;
    // the rows this call renders (all of them unless the caller asked for a band), y counted upwards
    const uint32_t band_rows = A->rc.band_rows;
    const uint32_t pixels = width * band_rows;
    const uint32_t n_groups = A->rc.n_groups;
    const FastDiv by_pixels{A->rc.by_pixels.m, A->rc.by_pixels.sh1, A->rc.by_pixels.sh2};
    const FastDiv by_groups{A->rc.by_groups.m, A->rc.by_groups.sh1, A->rc.by_groups.sh2};
    const FastDiv by_band{A->rc.by_band.m, A->rc.by_band.sh1, A->rc.by_band.sh2};
    const uint32_t place = first_place + k;  // < 2^31 + 2^27
    const uint32_t more_passes = fast_div(place, by_pixels);
    const uint32_t pass = first_pass + more_passes;
    uint32_t in_pass = place - more_passes * pixels;
    // Consecutive groups of 64 paths start far apart in the image, so that every region's new paths are
    // a sample of the whole frame and the regions cost about the same to trace.
    const uint32_t group = in_pass >> 6;
    if (group < n_groups) {
      const uint32_t scaled = group * A->rc.group_stride;
      in_pass = ((scaled - fast_div(scaled, by_groups) * n_groups) << 6) | (in_pass & 63u);
    }
    uint32_t x, y;
    pixel_of(in_pass, width, band_rows, by_band, x, y);
    y += A->rc.band_y0;
    const uint32_t pixel = y * width + x;
    const uint32_t sample = A->rc.first_sample + pass;
    Rng rng = rng_begin(pixel, sample, 0, STAGE_CAMERA, A->rc.key0, A->rc.key1);
    float jx = rng_next(rng), jy = rng_next(rng);
    const uint32_t strata = A->rc.strata;
    if (strata != 0) {  // opt-in: sample k of a pixel stays inside cell k mod strata^2 of the pixel
      const FastDiv by_strata{A->rc.by_strata.m, A->rc.by_strata.sh1, A->rc.by_strata.sh2};
      const FastDiv by_cells{A->rc.by_cells.m, A->rc.by_cells.sh1, A->rc.by_cells.sh2};
      const uint32_t cell = sample - fast_div(sample, by_cells) * (strata * strata);
      const uint32_t cy = fast_div(cell, by_strata), cx = cell - cy * strata;
      jx = ((float)cx + jx) / (float)strata;
      jy = ((float)cy + jy) / (float)strata;
    }
    const float u = ((float)x + jx) / (float)(width - 1);    // renderer.rs:216
    const float v = ((float)y + jy) / (float)(height - 1);   // renderer.rs:217
    // The shutter time is drawn ahead of the lens sample (camera.rs:58,66 draws it after; the draws are
    // independent): a pinhole camera, whose lens sample is multiplied by zero, then needs neither the
    // rejection loop nor a second block of random numbers.
    const float time = rng_range(rng, A->rc.cam.time0, A->rc.cam.time1);
    const float lens_radius = A->rc.cam.lens_radius;
    V3 disk = v3(0.0f, 0.0f, 0.0f);
    if (lens_radius != 0.0f) {
      for (;;) {  // vec3.rs:53-60
        const float dx = rng_range(rng, -1.0f, 1.0f);
        const float dy = rng_range(rng, -1.0f, 1.0f);
        disk = v3(dx, dy, 0.0f);
        if (len2(disk) < 1.0f) break;
      }
    }
    const V3 cu = v3(A->rc.cam.u[0], A->rc.cam.u[1], A->rc.cam.u[2]);
    const V3 cv = v3(A->rc.cam.v[0], A->rc.cam.v[1], A->rc.cam.v[2]);
    const V3 origin = v3(A->rc.cam.origin[0], A->rc.cam.origin[1], A->rc.cam.origin[2]);
    const V3 llc = v3(A->rc.cam.lower_left_corner[0], A->rc.cam.lower_left_corner[1], A->rc.cam.lower_left_corner[2]);
    const V3 hor = v3(A->rc.cam.horizontal[0], A->rc.cam.horizontal[1], A->rc.cam.horizontal[2]);
    const V3 ver = v3(A->rc.cam.vertical[0], A->rc.cam.vertical[1], A->rc.cam.vertical[2]);
    const V3 rd = lens_radius * disk;
    const V3 offset = cu * rd.x + cv * rd.y;
    const V3 d = llc + u * hor + v * ver - origin - offset;
    const V3 o = origin + offset;
    r0 = make_float4(o.x, o.y, o.z, time);
    r1 = make_float4(d.x, d.y, d.z, __uint_as_float(pixel));
    r2 = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(sample << 8));
  }
};

// Finds the closest surface of every live ray of the region and puts each one into the queue of the material
// hit.  The region's rays are numbered [0, total): the first n_stored of them are path records in the region's
// buffer (the survivors of the last round, or the caller's rays of a trace), the others are camera paths started
// here, in registers, at the moment a lane takes one (they never exist in memory as rays).  Lanes whose ray has
// finished pull the region's next ray instead of waiting for the slowest lane of the wave.
template <bool COUNT, bool LDS_SCENE>
__device__ __noinline__ void walk_rays(const RenderArgs* args, uint32_t lds_arg, bool odd, uint32_t n_stored, uint32_t total,
                                       uint64_t first_new, Work* work_out, uint32_t* segments_out) {
  constexpr int THREADS = RENDER_THREADS;
  float4* const smem = lds_base(lds_arg);
  WaveShared& sh = reinterpret_cast<RenderShared*>(smem)->wave[wave_in_block()];
  using RC = RefCode<LDS_SCENE>;
  constexpr int PSTRIDE = LDS_SCENE ? (int)PRIM_QUADS_LDS : (int)PRIM_QUADS;
  ConstArgs A = pinned(args);
  odd = uniform(odd) != 0;
  total = uniform(total);
  n_stored = uniform(n_stored);
  const CameraPaths camera(A, ((uint64_t)uniform((uint32_t)(first_new >> 32)) << 32) | uniform((uint32_t)first_new));
  const RenderLds<LDS_SCENE> lds(smem, A);
  SceneTables tb = lds.tables(A);
  if (!LDS_SCENE) {
    // (the node array's address in a pair of vector registers: left to itself the compiler, short of scalar
    // registers, reads it back from the launch arguments at the head of every tree step, a scalar load the step's
    // own fetch then waits behind)
    unsigned long long nodes_at = (unsigned long long)tb.nodes;
    asm volatile("" : "+v"(nodes_at));
    tb.nodes = (const float4*)nodes_at;
  }
  // (the overflow rows of the block: [level][thread])
  const WalkStack<LDS_SCENE> st{lds.stack, LDS_SCENE || A->stack_overflow == nullptr ? nullptr
                                               : A->stack_overflow + (size_t)blockIdx.x * A->overflow_levels * THREADS + threadIdx.x};
  const uint32_t region = region_of_wave(A);
  const uint32_t lane = lane_id();
  const uint32_t cap = A->rc.region_cap;
  float4* const cur = (odd ? A->paths_b : A->paths_a) + 4 * (size_t)region * cap;
  // queue q of the region: 64-byte records at [q * cap, (q + 1) * cap)
  float4* const region_queues = A->queues + 4 * (size_t)region * N_QUEUES * cap;
  const uint32_t root = RC::encode(A->sc.root_ref);
  const uint32_t n_slices = A->sc.n_slices, n_nodes = A->sc.n_nodes;
  const float slice_time0 = A->sc.slice_time0, slice_scale = A->sc.slice_scale;
  const uint32_t refill_min = A->refill_min, node_min = A->node_min, prims_per_turn = A->prims_per_turn;
  // Scenes walked from global memory: a ray that has taken step_budget steps in this round is put aside -- its
  // place in the tree (reference in hand, stack, closest hit so far) goes into the region's save area, the ray
  // into QUEUE_CONTINUE -- and goes on from there in the next round, so that the few rays that graze a large scene
  // for thousands of steps do not hold up the round for everyone.  The save area has two halves: the walks put
  // aside in the last round are read from one while this round's are written to the other.
  const uint32_t step_budget = LDS_SCENE ? 0u : A->step_budget;
  const uint32_t save_cap = A->save_cap;
  uint4* const saves_out = A->saves + (size_t)SAVE_QUADS * save_cap * (2u * region + (odd ? 1u : 0u));
  const uint4* const saves_in = A->saves + (size_t)SAVE_QUADS * save_cap * (2u * region + (odd ? 0u : 1u));
  Work work{};
  uint32_t segments = 0;
  long long ts = 0;
  Walk w;
  w.cur = RC::NONE;
  uint32_t slot = NO_SLOT;  // slot of the ray this lane is walking
  float4 state = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // what the record holds besides the ray: throughput, sample | bounce
  float pixel_bits = 0.0f;
  bool more = true;         // rays left to hand out
  // The region is this wave's alone, so its counters are registers: lane k counts the entries of queue k,
  // the rest is the same number in every lane.
  uint32_t queue_fill = 0;
  uint32_t n_saved = 0;     // walks put aside this round (entries of the region's save area in use)
  uint32_t next_entry = 0;  // next slot to hand out
  // Scenes in LDS: the walk waits for nothing but the rays themselves, so those are read ahead of their use.
  // Lane l of `ahead` holds slot next_entry + l, loaded right after the last exchange (the loads are in flight
  // while the wave walks); a lane that needs a ray takes it from the lane that holds it (a register exchange, no
  // memory), and the 64 slots after the ones handed out are loaded for the next exchange.
  // (Every lane loads, past the end of the stored rays the last one again, and after every exchange whether or not
  // the buffer was used up: a load under a condition would have to be merged with the registers' old contents,
  // and that merge would wait for it on the spot.)
  constexpr bool AHEAD = LDS_SCENE;
  float4 ahead[3];
  for (int k = 0; k < 3; ++k) ahead[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // stored ray number e of the round sits in slot e (the stream at the low end) or in the high end's stream
  const uint32_t n_low = uniform(sh.n_low), high_shift = n_stored > n_low ? cap - n_stored : 0u;
  auto place_of = [&](uint32_t e) { return e < n_low ? e : e + high_shift; };
  auto read_ahead = [&](uint32_t first) {
    uint32_t e = min(first + lane, max(n_stored, 1u) - 1u);
    e = place_of(e);
    if (!MIRT_CHECK(e < cap, A->work, 1, e, cap, n_stored)) return;
    ahead[0] = load_quad(cur + e); ahead[1] = load_quad(cur + (size_t)cap + e);
    ahead[2] = load_quad(cur + 2 * (size_t)cap + e);
  };
  if (AHEAD) read_ahead(0u);
  auto from_lane = [](const float4& v, uint32_t byte_address) {
    return make_float4(__int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_address, __float_as_int(v.x))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_address, __float_as_int(v.y))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_address, __float_as_int(v.z))),
                       __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_address, __float_as_int(v.w))));
  };
  // The wave alternates between two things.  Exchange: rays that are done are handed in and the lanes without
  // work get new rays.  Walk: step down the tree until (almost) every lane holds a leaf, then one primitive test in
  // every lane that holds a leaf, and so on until enough lanes are without work for another exchange.
  // done: the closest hit is known (finished), or the round's budget is used up and the walk is put aside (spent).
  bool finished = false, spent = false;
  for (;;) {
    {
      const bool done = finished || spent;
      if (COUNT) ts = clock64();
      const uint32_t base = next_entry;
      float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
      // The lanes that will be without work after the hand-in.  With the scene in LDS every ray that is done
      // leaves, and the lanes' ranks are wanted now, ahead of the hand-in's stores; with a step budget a spent walk
      // may have to stay (no room in the save area), so the lanes are counted after that is settled.
      const unsigned long long m_spent = LDS_SCENE ? 0ull : __ballot(spent);
      unsigned long long m_free = __ballot(done) | __ballot(slot == NO_SLOT);
      uint32_t n_free = (uint32_t)__popcll(m_free);
      uint32_t rank = lanes_below(m_free);
      if (AHEAD && more && base < n_stored) {
        // the rays the free lanes will take, out of the lanes that hold them, and the loads for the next exchange
        r0 = from_lane(ahead[0], rank << 2);
        r1 = from_lane(ahead[1], rank << 2);
        r2 = from_lane(ahead[2], rank << 2);
        read_ahead(base + n_free);
      }
      // Rays that are done leave their hit in the record and their slot in the queue of the material
      // hit.  Lane q keeps the fill of queue q and reserves its entries for the whole wave.
      int q = !done ? -1 : (w.best_ref == REF_NONE ? QUEUE_MISS : queue_of(w.best_material));
      if (q == QUEUE_LAMBERTIAN && ref_tag(w.best_ref) >= TAG_RECT) q = QUEUE_LAMBERTIAN_FLAT;
      uint32_t saved_at = 0;  // 1 + entry of the save area
      if (!LDS_SCENE && spent) {
        const uint32_t at = n_saved + lanes_below(m_spent);
        if (at < save_cap) {
          saved_at = at + 1u;
          q = QUEUE_CONTINUE;
          uint4* const sv = saves_out + (size_t)SAVE_QUADS * at;
          sv[0] = make_uint4(w.cur, w.sp, w.best_ref, w.best_order);
          sv[1] = make_uint4(__float_as_uint(w.best_t), w.best_material, w.best_face, 0u);
          for (uint32_t k = 0; k < w.sp; k += 4u)
            sv[2u + (k >> 2)] = make_uint4(st.get(k), k + 1u < w.sp ? st.get(k + 1u) : 0u, k + 2u < w.sp ? st.get(k + 2u) : 0u,
                                           k + 3u < w.sp ? st.get(k + 3u) : 0u);
        } else {
          q = -1;
          w.steps = 0;  // walks another budget's worth before it asks again
        }
      }
      if (!LDS_SCENE) n_saved += (uint32_t)__popcll(m_spent);
      const bool leaving = q >= 0;
      if (!LDS_SCENE) {
        m_free = __ballot(leaving) | __ballot(slot == NO_SLOT);
        n_free = (uint32_t)__popcll(m_free);
        rank = lanes_below(m_free);
      }
      unsigned long long mq = 0ull;  // the lanes handing in to the same queue as this one
      uint32_t cnt = 0;
#pragma unroll
      for (int k = 0; k < N_QUEUES; ++k) {
        const unsigned long long m = __ballot(q == k);
        if (q == k) mq = m;
        if (lane == (uint32_t)k) cnt = (uint32_t)__popcll(m);
      }
      const uint32_t got = queue_fill;
      queue_fill += cnt;
      const uint32_t bq = __shfl(got, q < 0 ? 0 : q);  // every lane takes part: the lanes that keep the fills may not be done
      if (leaving) {
        const uint32_t pos = bq + lanes_below(mq);
        // the whole path record moves into the queue, hit included: shading reads its queue front to back
        float4* const out = region_queues + 4 * (size_t)q * cap + pos;
        if (MIRT_CHECK(pos < cap && q < N_QUEUES, A->work, 2, pos, cap, q)) {
        store_quad(out, make_float4(w.r.o.x, w.r.o.y, w.r.o.z, w.r.time));
        store_quad(out + cap, make_float4(w.r.d.x, w.r.d.y, w.r.d.z, pixel_bits));
        store_quad(out + 2 * (size_t)cap, state);
        store_quad(out + 3 * (size_t)cap, make_float4(w.best_t, __uint_as_float(w.best_ref), __uint_as_float(w.best_face), __uint_as_float(saved_at)));
        }
        slot = NO_SLOT;
        w.cur = RC::NONE;
        if (LDS_SCENE || q != QUEUE_CONTINUE) ++segments;
      }
      if (COUNT) { const long long now = clock64(); work.cycles[MIRT_PHASE_WRITE] += now - ts; ts = now; }
      // (the lanes refilled together get consecutive slots: new camera paths of one pixel tile, which walk
      // the tree together)
      if (more) {
        more = base + n_free < total;
        next_entry = base + n_free;
        const uint32_t entry = base + rank;
        if (slot == NO_SLOT && entry < total) {
          slot = entry;
          if (entry >= n_stored) camera.start(entry - n_stored, r0, r1, r2);
          else if (!AHEAD && MIRT_CHECK(place_of(slot) < cap, A->work, 3, slot, cap, n_stored)) { const uint32_t at = place_of(slot); r0 = load_quad(cur + at); r1 = load_quad(cur + (size_t)cap + at); r2 = load_quad(cur + 2 * (size_t)cap + at); }
          state = r2;
          pixel_bits = r1.w;
          Ray r;
          r.o = xyz(r0); r.time = r0.w; r.d = xyz(r1);
          walk_begin(w, r, root, time_slice_base(r.time, n_slices, slice_time0, slice_scale, n_nodes));
          if (!LDS_SCENE && step_budget != 0u && entry < n_stored) {
            const uint32_t saved_in = __float_as_uint(load_quad(cur + 3 * (size_t)cap + place_of(slot)).x);
            if (saved_in != 0u) {  // a walk put aside in the last round: on from where it stopped
              const uint4* const sv = saves_in + (size_t)SAVE_QUADS * (saved_in - 1u);
              const uint4 a = sv[0], b = sv[1];
              w.cur = a.x; w.sp = a.y; w.best_ref = a.z; w.best_order = a.w;
              w.best_t = __uint_as_float(b.x); w.best_material = b.y; w.best_face = b.z;
              for (uint32_t k = 0; k < w.sp; k += 4u) {
                const uint4 e = sv[2u + (k >> 2)];
                st.set(k, e.x);
                if (k + 1u < w.sp) st.set(k + 1u, e.y);
                if (k + 2u < w.sp) st.set(k + 2u, e.z);
                if (k + 3u < w.sp) st.set(k + 3u, e.w);
              }
            }
          }
        }
      }
      if (COUNT) work.cycles[MIRT_PHASE_LOAD] += clock64() - ts;
    }
    bool all_out = false;
    for (;;) {
      if (COUNT) ts = clock64();
      while (!(w.cur & RC::LEAF)) {
        if (COUNT) { ++work.nodes; if (lane == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) ++work.wave_nodes; }
        if constexpr (LDS_SCENE) node_step<true>(w, tb, kTMin, st);
        else wide_step(w, tb, kTMin, st);
        if (!LDS_SCENE) ++w.steps;
        // the last few lanes still on their way down do not keep the others waiting
        if ((uint32_t)__popcll(__ballot(!(w.cur & RC::LEAF))) < node_min) break;
      }
      if (COUNT) { const long long now = clock64(); work.cycles[MIRT_PHASE_NODES] += now - ts; ts = now; }
      if constexpr (LDS_SCENE) {
        for (uint32_t k = 0; k < prims_per_turn; ++k) {
          if ((w.cur & RC::LEAF) && w.cur != RC::NONE) {
            if (COUNT) { ++work.prims; if (lane == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) ++work.wave_prims; }
            prim_step<PSTRIDE, true>(w, tb, kTMin, st);
          }
        }
      } else {
        if ((w.cur & RC::LEAF) && w.cur != RC::NONE) {
          const uint32_t tested = prim_pair_step<PSTRIDE>(w, tb, kTMin, st, prims_per_turn >= 2u);
          if (COUNT) { work.prims += tested; if (lane == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) ++work.wave_prims; }
          w.steps += tested;
        }
      }
      if (COUNT) work.cycles[MIRT_PHASE_PRIMS] += clock64() - ts;
      finished = slot != NO_SLOT && w.cur == RC::NONE;
      spent = !LDS_SCENE && step_budget != 0u && slot != NO_SLOT && !finished && w.steps >= step_budget &&
              w.sp <= SAVE_STACK;  // (a stack deeper than a save entry holds walks on)
      const unsigned long long m_done = __ballot(finished || spent);
      const uint32_t n_out = (uint32_t)__popcll(m_done | __ballot(slot == NO_SLOT));
      all_out = n_out == 64u && m_done == 0ull && !more;
      if (all_out) break;
      if ((m_done != 0ull || more) && (n_out >= refill_min || n_out == 64u)) break;
    }
    if (all_out) break;
  }
  if (lane < (uint32_t)N_QUEUES) sh.queue_n[lane] = queue_fill;  // for shading
  if (COUNT) {
    work_out->nodes += work.nodes; work_out->prims += work.prims;
    work_out->wave_nodes += work.wave_nodes; work_out->wave_prims += work.wave_prims;
    for (int k = 0; k < MIRT_PHASE_COUNT; ++k) work_out->cycles[k] += work.cycles[k];
    *segments_out += segments;
  }
}

// The walk of a scene of a handful of primitives: no tree.  A wave takes 64 rays and tests them against one
// primitive after the other, every lane the same primitive at the same time, so a wave never runs the code of
// two kinds of primitive at once (with a tree over 8 primitives a ray tests 5 of them anyway, each lane another).
// Closest hit, equal t to the higher order rank: the same answer as the tree walk.
template <bool COUNT>
__device__ __noinline__ void scan_rays(const RenderArgs* args, uint32_t lds_arg, bool odd, uint32_t n_stored, uint32_t total,
                                       uint64_t first_new, Work* work_out, uint32_t* segments_out) {
  float4* const smem = lds_base(lds_arg);
  WaveShared& sh = reinterpret_cast<RenderShared*>(smem)->wave[wave_in_block()];
  ConstArgs A = pinned(args);
  odd = uniform(odd) != 0;
  total = uniform(total);
  n_stored = uniform(n_stored);
  const CameraPaths camera(A, ((uint64_t)uniform((uint32_t)(first_new >> 32)) << 32) | uniform((uint32_t)first_new));
  const RenderLds<true> lds(smem, A);
  const SceneTables tb = lds.tables(A);
  const uint32_t region = region_of_wave(A);
  const uint32_t lane = lane_id();
  const uint32_t cap = A->rc.region_cap;
  const uint32_t n_prims = A->sc.n_prims;
  float4* const cur = (odd ? A->paths_b : A->paths_a) + 4 * (size_t)region * cap;
  float4* const region_queues = A->queues + 4 * (size_t)region * N_QUEUES * cap;
  uint32_t segments = 0;
  long long ts = 0;
  Work work{};
  for (uint32_t base = 0;; base += 64u) {
    if (COUNT) ts = clock64();
    if (base >= total) break;
    const uint32_t entry = base + lane;
    const bool live = entry < total;
    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, state = r0;
    if (live) {
      if (entry >= n_stored) camera.start(entry - n_stored, r0, r1, state);
      else { r0 = load_quad(cur + entry); r1 = load_quad(cur + (size_t)cap + entry); state = load_quad(cur + 2 * (size_t)cap + entry); }
    }
    Ray r;
    r.o = xyz(r0); r.time = r0.w; r.d = xyz(r1);
    float best_t = kInf;
    uint32_t best_ref = REF_NONE, best_order = 0, best_material = 0, best_face = 0;
    if (COUNT) { const long long now = clock64(); work.cycles[MIRT_PHASE_LOAD] += now - ts; ts = now; }
    for (uint32_t k = 0; k < n_prims; ++k) {
      float t;
      uint32_t tag, order, material, face = 0;
      if (live && record_t<(int)PRIM_QUADS_LDS>(tb, k, r, kTMin, best_t, t, tag, order, material, face)) {
        if (t < best_t || best_ref == REF_NONE || order > best_order) {  // equal t: the later one in the reference's order
          best_t = t; best_ref = make_leaf(tag, k); best_material = material; best_order = order; best_face = face;
        }
      }
    }
    if (COUNT) {
      const long long now = clock64();
      work.cycles[MIRT_PHASE_PRIMS] += now - ts; ts = now;
      if (live) work.prims += n_prims;
      if (lane == 0) work.wave_prims += n_prims;
    }
    int q = !live ? -1 : (best_ref == REF_NONE ? QUEUE_MISS : queue_of(best_material));
    if (q == QUEUE_LAMBERTIAN && ref_tag(best_ref) >= TAG_RECT) q = QUEUE_LAMBERTIAN_FLAT;
    unsigned long long mq = 0ull;
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < N_QUEUES; ++k) {
      const unsigned long long m = __ballot(q == k);
      if (q == k) mq = m;
      if (lane == (uint32_t)k) cnt = (uint32_t)__popcll(m);
    }
    uint32_t got = 0;
    if (lane < (uint32_t)N_QUEUES && cnt) got = atomicAdd(&sh.queue_n[lane], cnt);
    const uint32_t bq = __shfl(got, q < 0 ? 0 : q);
    if (live) {
      const uint32_t pos = bq + lanes_below(mq);
      float4* const out = region_queues + 4 * (size_t)q * cap + pos;
      store_quad(out, r0);
      store_quad(out + cap, r1);
      store_quad(out + 2 * (size_t)cap, state);
      store_quad(out + 3 * (size_t)cap, make_float4(best_t, __uint_as_float(best_ref), __uint_as_float(best_face), 0.0f));
      ++segments;
    }
    if (COUNT) work.cycles[MIRT_PHASE_WRITE] += clock64() - ts;
  }
  if (COUNT) {
    work_out->prims += work.prims;
    work_out->wave_prims += work.wave_prims;
    for (int k = 0; k < MIRT_PHASE_COUNT; ++k) work_out->cycles[k] += work.cycles[k];
    *segments_out += segments;
  }
}

// Shades the region's queues (renderer.rs:141-212): constant media get their chance, the material
// scatters or ends the path, survivors are packed into the first slots of the region's other buffer.
template <bool COUNT, bool LDS_SCENE>
__device__ __noinline__ void shade_queues(const RenderArgs* args, uint32_t lds_arg, bool odd) {
  float4* const smem = lds_base(lds_arg);
  RenderShared& block = *reinterpret_cast<RenderShared*>(smem);
  WaveShared& sh = block.wave[wave_in_block()];
  constexpr int PSTRIDE = LDS_SCENE ? (int)PRIM_QUADS_LDS : (int)PRIM_QUADS;
  ConstArgs A = pinned(args);
  odd = uniform(odd) != 0;
  const RenderLds<LDS_SCENE> lds(smem, A);
  const SceneTables tb = lds.tables(A);
  const uint32_t region = region_of_wave(A);
  const uint32_t lane = lane_id();
  ShadeTables sc = MIRT_SHADE_TABLES(A->sc);
  if (LDS_SCENE) {  // the block's own copies
    sc.materials = reinterpret_cast<const DMaterial*>(lds.materials);
    sc.textures = reinterpret_cast<const DTexture*>(lds.textures);
    sc.lights = reinterpret_cast<const DLight*>(lds.lights);
    sc.light_groups = reinterpret_cast<const uint2*>(lds.groups);
    if (A->lds_perlin_offset != 0u) sc.perlin_lds = uniform(lds_arg) + A->lds_perlin_offset;
  }
  const LightList ll{sc.lights, sc.light_groups, sc.lights_mode, sc.n_lights, sc.n_light_groups};
  const SceneTables gt = MIRT_GLOBAL_TABLES(A->sc);
  const uint32_t n_media = A->sc.n_media;
  struct { uint32_t width, height, max_depth, key0, key1; float background[3]; FastDiv by_width; }
      rc{A->rc.width, A->rc.height, A->rc.max_depth, A->rc.key0, A->rc.key1,
         {A->rc.background[0], A->rc.background[1], A->rc.background[2]},
         {A->rc.by_width.m, A->rc.by_width.sh1, A->rc.by_width.sh2}};
  const uint32_t cap = A->rc.region_cap;
  const uint32_t estimator = A->rc.estimator;
  const bool budgeted = !LDS_SCENE && A->step_budget != 0u;
  const bool split = uniform(sh.split) != 0u;
  float4* const next = (odd ? A->paths_a : A->paths_b) + 4 * (size_t)region * cap;
  float4* const region_queues = A->queues + 4 * (size_t)region * N_QUEUES * cap;
  float* const rgb_sum = A->rgb_sum;
  // The walk queued every ray under the material of the surface it found.  A constant medium in front of that
  // surface changes the material, so shading takes two passes: the first gives the media their chance, shades
  // what stays in its queue and appends the rest (written back with the medium hit) to the queue of the
  // medium's material; the second shades those.  Every wave then runs one material's code with full lanes.
  for (uint32_t pass = 0; pass < 2u; ++pass) {
    // The entries of the queues are cut into batches of 64, numbered queue after queue.
    uint32_t first[N_QUEUES], count[N_QUEUES], last_batch[N_QUEUES];
    uint32_t n_batches = 0;
#pragma unroll
    for (int k = 0; k < N_QUEUES; ++k) {
      const uint32_t walked = uniform(sh.queue_n[k]);
      first[k] = pass == 0 ? 0u : walked;
      count[k] = pass == 0 ? walked : uniform(sh.late_n[k]);
      n_batches += (count[k] + 63u) >> 6;
      last_batch[k] = n_batches;
    }
    for (uint32_t batch = 0; batch < n_batches; ++batch) {
      uint32_t q = 0, n_q = count[0], e0 = batch << 6, from = first[0];
#pragma unroll
      for (int k = 1; k < N_QUEUES; ++k)
        if (batch >= last_batch[k - 1]) { q = k; n_q = count[k]; e0 = (batch - last_batch[k - 1]) << 6; from = first[k]; }
      float4* const entries = region_queues + 4 * (size_t)q * cap + from;
      const uint32_t e = e0 + lane;
      bool alive = false;
      Ray nr;
      V3 beta = v3(0.0f, 0.0f, 0.0f);
      uint32_t pixel = 0, sample = 0, bounce = 0;
      uint32_t saved_walk = 0;  // 1 + entry of the save area for a ray whose walk was put aside
      if (e < n_q && MIRT_CHECK(from + e < cap, A->work, 5, e, from, n_q)) {
        const float4* const rec = entries + e;
        const float4 r0 = load_quad(rec), r1 = load_quad(rec + cap), r2 = load_quad(rec + 2 * (size_t)cap),
                     r3 = load_quad(rec + 3 * (size_t)cap);
        Ray r;
        r.o = xyz(r0); r.time = r0.w; r.d = xyz(r1);
        pixel = __float_as_uint(r1.w);
        if (!MIRT_CHECK(pixel < rc.width * rc.height, A->work, 7, pixel, q, e)) pixel = 0;
        beta = xyz(r2);
        sample = __float_as_uint(r2.w) >> 8;
        bounce = __float_as_uint(r2.w) & 255u;
        float t = r3.x;
        uint32_t ref = __float_as_uint(r3.y);
        bool deferred = false;
        const bool goes_on = !LDS_SCENE && q == (uint32_t)QUEUE_CONTINUE;  // its walk goes on in the next round
        if (n_media != 0 && pass == 0 && !goes_on) {  // the walk found the closest surface; the media in front of it get their chance now
          uint32_t material = 0;
          const uint32_t surface = ref;
          Rng mrng = rng_begin(pixel, sample, bounce, STAGE_MEDIA, rc.key0, rc.key1);
          closest_media(n_media, tb, gt, block.media, r, kTMin, mrng, ref, t, material);
          if (ref != surface) {
            const uint32_t to_q = (uint32_t)queue_of(material);
            if (to_q != q) {  // another material's code: goes to the end of that queue, for the second pass
              deferred = true;
              const uint32_t at = sh.queue_n[to_q] + atomicAdd(&sh.late_n[to_q], 1u);
              float4* const out = region_queues + 4 * (size_t)to_q * cap + at;
              if (MIRT_CHECK(at < cap && to_q < (uint32_t)N_QUEUES, A->work, 6, at, cap, to_q)) {
              // (plain stores: these few records are read again within this shading pass -- streamed past the
              // caches like the rest they cost final_scene 3.7 %)
              out[0] = r0; out[cap] = r1; out[2 * (size_t)cap] = r2;
              out[3 * (size_t)cap] = make_float4(t, __uint_as_float(ref), r3.z, r3.w);
              }
            }
          }
        }

        V3 term = v3(0.0f, 0.0f, 0.0f);  // radiance that ends the path
        if (goes_on) {  // back into the pool as it is
          nr = r;
          saved_walk = __float_as_uint(r3.w);
          bounce -= 1u;  // (the survivor's record counts one bounce more than `bounce`)
          alive = true;
        } else if (deferred) {
          beta = v3(0.0f, 0.0f, 0.0f);  // nothing to add to the image from here
        } else if (ref == REF_NONE) {
          term = v3(rc.background[0], rc.background[1], rc.background[2]);  // renderer.rs:187-189
        } else if (!MIRT_CHECK((ref & REF_LEAF) && (ref_tag(ref) == TAG_MEDIUM ? ref_index(ref) < n_media
                                                                              : ref_tag(ref) <= TAG_BOX && ref_index(ref) < A->sc.n_prims),
                               A->work, 9, ref, q, e)) {
          beta = v3(0.0f, 0.0f, 0.0f);
        } else {
          const Hit h = hit_record<PSTRIDE>(tb, block.media, ref, r, t, __float_as_uint(r3.z));
          uint32_t material_index = h.material_word & 0x0FFFFFFFu;
          if (!MIRT_CHECK(material_index < A->sc.n_materials, A->work, 10, h.material_word, ref, q)) material_index = 0;
          const DMaterial m = sc.materials[material_index];
          Rng rng = rng_begin(pixel, sample, bounce, STAGE_SHADE, rc.key0, rc.key1);
          // the one texture lookup of the materials that have one (a light's only shows on its front side)
          V3 att = v3(0.0f, 0.0f, 0.0f);
          if (m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_METAL ||
              (m.kind == MIRT_MAT_DIFFUSE_LIGHT && h.front))
            att = texture_color(sc, m.texture, h);
          if (m.kind == MIRT_MAT_LAMBERTIAN) {  // material.rs:39-58, renderer.rs:166-183
            const Onb uvw = onb_from_w(h.n);
            V3 dir;
            float pdf;
            if (sc.lights_mode != MIRT_LIGHTS_NONE) {
              dir = mixture_sample(ll, uvw, h.p, rng, (estimator & MIRT_RENDER_EVEN_LIGHTS) != 0, pdf);
            } else {
              dir = onb_local(uvw, random_cosine_direction(rng));
              const float c1 = dot(unit(dir), uvw.w);
              pdf = c1 <= 0.0f ? 0.0f : c1 / kPi;
            }
            const float c2 = dot(h.n, unit(dir));  // material.rs:50-57
            const float spdf = c2 < 0.0f ? 0.0f : c2 / kPi;
            beta = ((beta * att) * spdf) / pdf;
            nr.o = h.p; nr.d = dir; nr.time = r.time;
            alive = true;
          } else if (m.kind == MIRT_MAT_METAL) {  // material.rs:75-91
            const V3 reflected = reflect(unit(r.d), h.n);
            const V3 fuzz = m.param * random_in_unit_sphere(rng);
            beta = beta * att;
            nr.o = h.p; nr.d = reflected + fuzz; nr.time = r.time;
            alive = true;
          } else if (m.kind == MIRT_MAT_DIELECTRIC) {  // material.rs:111-137
            const float ratio = h.front ? 1.0f / m.param : m.param;
            const V3 ud = unit(r.d);
            const float cos_theta = fminf(dot(-ud, h.n), 1.0f);
            const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool do_reflect = ratio * sin_theta > 1.0f;
            if (!do_reflect) do_reflect = reflectance(cos_theta, ratio) > rng_next(rng);
            nr.d = do_reflect ? reflect(ud, h.n) : refract(ud, h.n, ratio);
            nr.o = h.p; nr.time = r.time;
            alive = true;
          } else if (m.kind == MIRT_MAT_DIFFUSE_LIGHT) {  // material.rs:151-158
            if (h.front) term = att;
          }
          // MIRT_MAT_ISOTROPIC: no scatter, no emission
        }
        // The child call at depth 0 returns black without tracing (renderer.rs:147-149).
        if (alive && !goes_on && bounce + 1 >= rc.max_depth) alive = false;
        if (!alive) {
          // beta * term is the path's whole contribution; it is added even when
          // term is zero if beta is NaN/inf so a poisoned sample blackens its
          // pixel as in the reference (renderer.rs:241-252: NaN as u8 == 0)
          const V3 cval = beta * term;
          const bool dropped = (estimator & MIRT_RENDER_DROP_NONFINITE) != 0 &&
                               !(fabsf(cval.x) < kInf && fabsf(cval.y) < kInf && fabsf(cval.z) < kInf);  // opt-in
          if (!(cval.x == 0.0f && cval.y == 0.0f && cval.z == 0.0f) && !dropped) {
            const uint32_t y = fast_div(pixel, rc.by_width), x = pixel - y * rc.width;
            float* dst = rgb_sum + ((size_t)(rc.height - 1 - y) * rc.width + x) * 3;
            atomicAdd(dst, cval.x);
            atomicAdd(dst + 1, cval.y);
            atomicAdd(dst + 2, cval.z);
          }
        }
      }
      // survivors are packed into the first slots of the region's other buffer
      // (two streams: rays that head down are kept apart from rays that head up -- from the floor they meet
      // different things, and the next walk's waves then hold rays of a kind: primitive tests 5 % cheaper, +1 % a
      // frame on final_scene; which stream a ray is in changes nothing about the ray)
      size_t to;
      if (split) {
        const bool high = alive && nr.d.y < 0.0f;
        const uint32_t up = wave_reserve(alive && !high, &sh.n_next), down = wave_reserve(high, &sh.n_high);
        to = high ? (size_t)cap - 1u - down : (size_t)up;
      } else {
        to = wave_reserve(alive, &sh.n_next);
      }
      if (alive && MIRT_CHECK(to < cap, A->work, 8, to, cap, q)) {
        store_quad(next + to, make_float4(nr.o.x, nr.o.y, nr.o.z, nr.time));
        store_quad(next + cap + to, make_float4(nr.d.x, nr.d.y, nr.d.z, __uint_as_float(pixel)));
        store_quad(next + 2 * (size_t)cap + to, make_float4(beta.x, beta.y, beta.z, __uint_as_float((sample << 8) | (bounce + 1))));
        if (budgeted) store_quad(next + 3 * (size_t)cap + to, make_float4(__uint_as_float(saved_walk), 0.0f, 0.0f, 0.0f));
      }
    }
    if (n_media == 0 || pass == 1u) break;
    wave_sync();  // the appended entries and their records are visible to every lane
  }
}

// Tracing caller-supplied rays uses the same kernel: these two stand in for start_paths and shade_queues.
// Ray number g of the call goes into a free slot with g in the record's pixel field ...
__device__ __noinline__ void start_rays(const RenderArgs* args, bool odd, uint32_t n_fresh, uint64_t first_new) {
  ConstArgs A = pinned(args);
  odd = uniform(odd) != 0;
  n_fresh = uniform(n_fresh);
  first_new = ((uint64_t)uniform((uint32_t)(first_new >> 32)) << 32) | uniform((uint32_t)first_new);
  const size_t cap = A->rc.region_cap;
  float4* const cur = (odd ? A->paths_b : A->paths_a) + 4 * (size_t)region_of_wave(A) * cap;
  const mirt_ray* const rays = A->rays;
  for (uint32_t k = lane_id(); k < n_fresh; k += 64u) {
    const uint64_t g = first_new + k;
    const mirt_ray in = rays[g];
    store_quad(cur + k, make_float4(in.origin[0], in.origin[1], in.origin[2], in.time));
    store_quad(cur + cap + k, make_float4(in.direction[0], in.direction[1], in.direction[2], __uint_as_float((uint32_t)g)));
    store_quad(cur + 2 * cap + k, make_float4(1.0f, 1.0f, 1.0f, 0.0f));
  }
}

// ... and once the walk has found its closest surface (and the media had their chance, with the draws of
// path (g, sample 0, bounce 0)), what it hit is written to the caller's arrays; no ray goes on.
template <bool LDS_SCENE>
__device__ __noinline__ void report_hits(const RenderArgs* args, uint32_t lds_arg, bool odd) {
  float4* const smem = lds_base(lds_arg);
  RenderShared& block = *reinterpret_cast<RenderShared*>(smem);
  WaveShared& sh = block.wave[wave_in_block()];
  ConstArgs A = pinned(args);
  odd = uniform(odd) != 0;
  const RenderLds<LDS_SCENE> lds(smem, A);
  const SceneTables tb = lds.tables(A);
  const SceneTables gt = MIRT_GLOBAL_TABLES(A->sc);
  const uint32_t n_media = A->sc.n_media;
  const DMedium* const media = A->sc.media;
  const uint2* const prim_ids = A->sc.prim_ids;
  const uint32_t key0 = A->rc.key0, key1 = A->rc.key1;
  const uint32_t cap = A->rc.region_cap;
  const float4* const region_queues = A->queues + 4 * (size_t)region_of_wave(A) * N_QUEUES * cap;
  uint32_t* const hit_type = A->hit_type;
  uint32_t* const hit_object = A->hit_object;
  float* const hit_t = A->hit_t;
  float4* const hit_records = A->hit_records;  // optional: the whole hit record of every ray
  constexpr int PSTRIDE = LDS_SCENE ? (int)PRIM_QUADS_LDS : (int)PRIM_QUADS;
  ShadeTables sc = MIRT_SHADE_TABLES(A->sc);
  if (LDS_SCENE) {
    sc.materials = reinterpret_cast<const DMaterial*>(lds.materials);
    sc.textures = reinterpret_cast<const DTexture*>(lds.textures);
  }
  for (uint32_t q = 0; q < N_QUEUES; ++q) {
    const uint32_t n_q = uniform(sh.queue_n[q]);
    const float4* entries = region_queues + 4 * (size_t)q * cap;
    for (uint32_t e = lane_id(); e < n_q; e += 64u) {
      const float4* const rec = entries + e;
      const float4 r0 = load_quad(rec), r1 = load_quad(rec + cap), r3 = load_quad(rec + 3 * (size_t)cap);
      Ray r;
      r.o = xyz(r0); r.time = r0.w; r.d = xyz(r1);
      const uint32_t g = __float_as_uint(r1.w);
      float t = r3.x;
      uint32_t ref = __float_as_uint(r3.y);
      if (n_media != 0) {
        uint32_t material = 0;
        Rng rng = rng_begin(g, 0, 0, STAGE_MEDIA, key0, key1);
        closest_media(n_media, tb, gt, block.media, r, kTMin, rng, ref, t, material);
      }
      uint32_t ti = 0xFFFFFFFFu, oi = 0xFFFFFFFFu;
      if (ref != REF_NONE) {
        if (ref_tag(ref) == TAG_MEDIUM) {
          ti = media[ref_index(ref)].type_index;
          oi = media[ref_index(ref)].object_index;
        } else {
          const uint2 id = prim_ids[ref_index(ref)];
          ti = id.x; oi = id.y;
        }
      }
      hit_type[g] = ti;
      hit_object[g] = oi;
      hit_t[g] = ref != REF_NONE ? t : kInf;
      if (hit_records) {
        // the record shading would work with (hittable.rs:8-42 after every wrapper), and what the material makes
        // of it before any random draw: the albedo a Lambertian or Metal scatters with (material.rs:40-49,66-90),
        // white for a Dielectric (:111-137), what a DiffuseLight emits (:151-158), nothing for an Isotropic
        float4 o0 = make_float4(0.0f, 0.0f, 0.0f, kInf), o1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o2 = o1, o3 = o1;
        o2.z = __uint_as_float(ti); o2.w = __uint_as_float(oi);
        if (ref != REF_NONE) {
          Hit h = hit_record<PSTRIDE>(tb, block.media, ref, r, t, __float_as_uint(r3.z));
          if (h.sphere_uv) sphere_uv(h.on_sphere, h.u, h.v);
          const DMaterial m = sc.materials[h.material_word & 0x0FFFFFFFu];
          V3 color = v3(0.0f, 0.0f, 0.0f);
          if (m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_METAL || (m.kind == MIRT_MAT_DIFFUSE_LIGHT && h.front))
            color = texture_color(sc, m.texture, h);
          else if (m.kind == MIRT_MAT_DIELECTRIC)
            color = v3(1.0f, 1.0f, 1.0f);
          o0 = make_float4(h.p.x, h.p.y, h.p.z, h.t);
          o1 = make_float4(h.n.x, h.n.y, h.n.z, __uint_as_float(h.front ? 1u : 0u));
          o2.x = h.u; o2.y = h.v;
          o3 = make_float4(color.x, color.y, color.z, __uint_as_float(m.kind));
        }
        hit_records[4 * (size_t)g] = o0;
        hit_records[4 * (size_t)g + 1] = o1;
        hit_records[4 * (size_t)g + 2] = o2;
        hit_records[4 * (size_t)g + 3] = o3;
      }
    }
  }
}

// The whole render in one launch.  Every wave owns a region of the path pool and stays resident until the
// render is done.  The block stages the scene in LDS once; from then on each of its waves goes round on its own
//   start   camera paths in the region's free slots, taking their numbers from a counter all waves share,
//   walk    every live ray to its closest surface and into the region's queue for the material hit,
//   shade   queue by queue; survivors go to the first slots of the region's other buffer,
// until the region has no live path and the counter has run out.  No wave waits for another one, so the waves of
// a CU drift apart and at any time some of them walk (LDS and arithmetic) while others start or shade (global memory).
//
// LDS_SCENE: nodes, primitive records and wrapper chains are staged in LDS, so neither walk nor shade waits
// on global memory for the scene; otherwise those tables are read from global memory.
template <bool COUNT, bool LDS_SCENE>
__global__ __launch_bounds__(RENDER_THREADS) void k_render(const RenderArgs* __restrict__ args) {
  extern __shared__ float4 smem[];
  RenderShared& block = *reinterpret_cast<RenderShared*>(smem);
  uint32_t lds_address = (uint32_t)(uintptr_t)(LdsQuads)smem;
  asm volatile("" : "+s"(lds_address));
  {
    ConstArgs A = pinned(args);
    const uint32_t threads = A->waves_per_block * 64u;
    stage_media(A->sc.n_media, A->sc.media, A->sc.prims, block.media);
    if (LDS_SCENE) {
      const RenderLds<true> lds(smem, A);
      const float4 *nodes = A->sc.nodes, *prims = A->sc.prims, *chain_ops = A->sc.chain_ops;
      const float4* chains = A->sc.chains;
      const uint32_t n_node_quads = A->sc.n_nodes * A->sc.n_slices * 4u, n_prim_quads = A->sc.n_prims * PRIM_QUADS_LDS;
      const uint32_t n_chain_ops = A->sc.n_chain_ops, n_chains = A->sc.n_chains;
      for (uint32_t k = threadIdx.x; k < n_node_quads; k += threads) {
        float4 q = nodes[k];
        if ((k & 3u) < 2u) q.w = __uint_as_float(RefCode<true>::encode(__float_as_uint(q.w)));  // child refs
        lds.nodes[k] = q;
      }
      for (uint32_t k = threadIdx.x; k < n_prim_quads; k += threads)
        lds.prims[k] = prims[(k / PRIM_QUADS_LDS) * PRIM_QUADS + (k % PRIM_QUADS_LDS)];
      for (uint32_t k = threadIdx.x; k < n_chain_ops; k += threads) lds.ops[k] = chain_ops[k];
      for (uint32_t k = threadIdx.x; k < n_chains * 2u; k += threads) lds.chains[k] = chains[k];
      const float4* materials = reinterpret_cast<const float4*>(A->sc.materials);
      const float4* textures = reinterpret_cast<const float4*>(A->sc.textures);
      const float4* lights = reinterpret_cast<const float4*>(A->sc.lights);
      const uint2* groups = A->sc.light_groups;
      const uint32_t n_materials = A->sc.n_materials, n_texture_quads = A->sc.n_textures * 2u;
      const uint32_t n_light_quads = A->sc.n_lights * 2u, n_groups = A->sc.n_light_groups;
      for (uint32_t k = threadIdx.x; k < n_materials; k += threads) lds.materials[k] = materials[k];
      for (uint32_t k = threadIdx.x; k < n_texture_quads; k += threads) lds.textures[k] = textures[k];
      for (uint32_t k = threadIdx.x; k < n_light_quads; k += threads) lds.lights[k] = lights[k];
      for (uint32_t k = threadIdx.x; k < n_groups; k += threads) reinterpret_cast<uint2*>(lds.groups)[k] = groups[k];
      if (A->lds_perlin_offset != 0u) {
        unsigned char* const to = reinterpret_cast<unsigned char*>(smem) + A->lds_perlin_offset;
        const float4* const vec = A->sc.perlin_vec;
        const uint32_t* const perm = A->sc.perlin_perm;
        for (uint32_t t = 0; t < A->sc.n_perlins; ++t) {
          for (uint32_t k = threadIdx.x; k < 256u; k += threads) reinterpret_cast<float4*>(to + t * PERLIN_LDS_BYTES)[k] = vec[256u * t + k];
          for (uint32_t k = threadIdx.x; k < 768u; k += threads) to[t * PERLIN_LDS_BYTES + 4096u + k] = (unsigned char)perm[768u * t + k];
        }
      }
    }
  }
  __syncthreads();  // the staged scene is complete: the one time the waves of a block wait for each other

  WaveShared& sh = block.wave[wave_in_block()];
  const bool tracing = pinned(args)->rays != nullptr;  // caller-supplied rays instead of camera paths
  const bool small_scene = pinned(args)->sc.n_prims <= SCAN_MAX_PRIMS;  // tested one primitive after the other
  Work work{};
  long long k_begin = 0, ts = 0;
  if (COUNT) k_begin = clock64();
  uint32_t segments = 0, rounds = 0;
  uint32_t n_surv = 0;      // survivors the region carries: n_low of them in its slots [0, n_low), the rest at the high end
  uint32_t n_low = 0;
  bool exhausted = false;   // the shared counter has handed out the render's last path (lane 0 only)
  for (;;) {
    if (lane_id() == 0) {
      ConstArgs A = pinned(args);
      uint32_t fresh = 0;
      const uint32_t n_free = A->rc.region_cap - n_surv;
      const unsigned long long total_paths = A->rc.total_paths;
      if (n_free != 0 && !exhausted) {
        // Towards the end of the render the waves take smaller helpings -- an equal share of what the counter had
        // left when last looked at, rounded up to a wave's worth -- so that they all run dry together instead of
        // some sitting idle while others work through a last full region.
        unsigned long long take = n_free;
        const unsigned long long seen = __hip_atomic_load(&A->state->issued, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen < total_paths) {
          const unsigned long long share = ((total_paths - seen) / (gridDim.x * A->waves_per_block) + 64ull) & ~63ull;
          if (share < take) take = share;
        }
        const unsigned long long first = atomicAdd(&A->state->issued, take);
        if (first < total_paths) {
          const unsigned long long left = total_paths - first;
          fresh = left < take ? (uint32_t)left : (uint32_t)take;
        }
        exhausted = first + take >= total_paths;
        sh.first_new_lo = (uint32_t)first;
        sh.first_new_hi = (uint32_t)(first >> 32);
      }
      sh.n_fresh = fresh;
      for (int q = 0; q < N_QUEUES; ++q) { sh.queue_n[q] = 0; sh.late_n[q] = 0; }
      sh.n_low = tracing ? n_surv + fresh : n_low;
      sh.split = !tracing && !(LDS_SCENE && small_scene) && A->split_survivors != 0u ? 1u : 0u;
      sh.n_next = 0;
      sh.n_high = 0;
    }
    wave_sync();
    const uint32_t n_fresh = uniform(sh.n_fresh);
    const uint32_t total = n_surv + n_fresh;
    if (total == 0) break;  // nothing alive and nothing left to start: the region is done
    if (!MIRT_CHECK(total <= pinned(args)->rc.region_cap, pinned(args)->work, 11, n_surv, n_fresh, rounds)) break;
    const bool odd = (rounds & 1u) != 0;  // the region's two buffers swap roles every round
    ++rounds;

    if (COUNT) ts = clock64();
    const uint64_t first_new = ((uint64_t)uniform(sh.first_new_hi) << 32) | uniform(sh.first_new_lo);
    // (a render's new paths are started by the walk itself; the rays of a trace come from the caller's array)
    if (tracing) {
      start_rays(args, odd, n_fresh, first_new);
      wave_sync();  // the new rays are visible to every lane
    }
    if (COUNT) work.cycles[MIRT_PHASE_LOAD] += clock64() - ts;
    const uint32_t n_stored = tracing ? total : n_surv;
    if (LDS_SCENE && small_scene) scan_rays<COUNT>(args, lds_address, odd, n_stored, total, first_new, &work, &segments);
    else walk_rays<COUNT, LDS_SCENE>(args, lds_address, odd, n_stored, total, first_new, &work, &segments);
    wave_sync();  // every walked ray's record, surface hit included, sits in one of the region's queues
    if (COUNT) ts = clock64();
    if (tracing) report_hits<LDS_SCENE>(args, lds_address, odd);
    else shade_queues<COUNT, LDS_SCENE>(args, lds_address, odd);
    if (COUNT) work.cycles[MIRT_PHASE_SHADE] += clock64() - ts;
    wave_sync();
    n_low = uniform(sh.n_next);
    n_surv = n_low + uniform(sh.n_high);
#ifdef MIRT_SURVIVOR_AUDIT
    // Diagnostic build: what shading left in the region's other buffer, looked at from outside the shading code (whose
    // code generation this leaves untouched).  Per wave of the block: survivors counted by shading, and records among
    // them whose bounce field is not this round's number -- a slot no lane wrote.
    if (!tracing) {
      ConstArgs A = pinned(args);
      const uint32_t cap = A->rc.region_cap;
      const float4* const next = (odd ? A->paths_a : A->paths_b) + 4 * (size_t)region_of_wave(A) * cap;
      uint32_t stale = 0;
      const uint32_t shift = n_surv > n_low ? cap - n_surv : 0u;  // (the stream at the high end)
      for (uint32_t k = lane_id(); k < n_surv; k += 64u)
        if ((__float_as_uint(next[2 * (size_t)cap + (k < n_low ? k : k + shift)].w) & 255u) != rounds) ++stale;
      for (int off = 32; off > 0; off >>= 1) stale += __shfl_down(stale, off);
      stale = uniform(stale);
      if (lane_id() == 0) {
        atomicAdd(&A->work->audit[wave_in_block()], (unsigned long long)n_surv);
        atomicAdd(&A->work->audit[16 + wave_in_block()], (unsigned long long)stale);
      }
      // the first region found with such slots: which ones (a bit per slot, 64 slots per word), and what the round's
      // queues held
      if (stale != 0) {
        unsigned long long first = 1ull;
        if (lane_id() == 0) first = atomicCAS(&A->work->check[7], 0ull, 1ull);
        first = ((unsigned long long)uniform((uint32_t)(first >> 32)) << 32) | uniform((uint32_t)first);
        if (first == 0ull) {
          for (uint32_t w = 0; w < 6u; ++w) {
            const uint32_t k = w * 64u + lane_id();
            const bool bad = k < n_surv && (__float_as_uint(next[2 * (size_t)cap + (k < n_low ? k : k + shift)].w) & 255u) != rounds;
            const unsigned long long m = __ballot(bad);
            if (lane_id() == 0) A->work->check[w] = m;
          }
          if (lane_id() == 0) {
            A->work->check[6] = ((unsigned long long)n_surv << 32) | (wave_in_block() << 16) | rounds;
            for (int q = 0; q < N_QUEUES; ++q) A->work->audit[q] |= (unsigned long long)sh.queue_n[q] << 40;
          }
        }
      }
    }
#endif
  }

  WorkCounters* const work_out = pinned(args)->work;
  if (lane_id() == 0) atomicMax(&work_out->rounds, rounds);
  if (COUNT) {
    work.cycles[MIRT_PHASE_TOTAL] = clock64() - k_begin;
    // one atomic per wave; a lane's phase clocks stop when it leaves a loop early, so the wave's time is the max
    for (int off = 32; off > 0; off >>= 1) {
      work.nodes += __shfl_down(work.nodes, off);
      work.prims += __shfl_down(work.prims, off);
      segments += __shfl_down(segments, off);
      work.wave_nodes += __shfl_down(work.wave_nodes, off);
      work.wave_prims += __shfl_down(work.wave_prims, off);
      for (int k = 0; k < MIRT_PHASE_COUNT; ++k) {
        const long long other = __shfl_down(work.cycles[k], off);
        work.cycles[k] = other > work.cycles[k] ? other : work.cycles[k];
      }
    }
    if (lane_id() == 0) {
      atomicAdd(&work_out->nodes, (unsigned long long)work.nodes);
      atomicAdd(&work_out->prims, (unsigned long long)work.prims);
      atomicAdd(&work_out->wave_nodes, (unsigned long long)work.wave_nodes);
      atomicAdd(&work_out->wave_prims, (unsigned long long)work.wave_prims);
      atomicAdd(&work_out->segments, (unsigned long long)segments);
      for (int k = 0; k < MIRT_PHASE_COUNT; ++k) atomicAdd(&work_out->cycles[k], (unsigned long long)work.cycles[k]);
    }
  }
}

// ---- resolve (renderer.rs:232-259) -------------------------------------------------
__global__ void k_resolve(const float* __restrict__ rgb_sum, uint32_t n_values, uint32_t spp,
                          uint8_t* __restrict__ rgb24) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_values) return;
  float v = sqrtf(rgb_sum[i] / (float)spp);
  v = v < 0.0f ? 0.0f : (v > 0.999f ? 0.999f : v);  // f32::clamp keeps NaN
  v = 255.999f * v;
  uint8_t out;
  if (!(v == v) || v <= 0.0f) out = 0;  // `as u8`: NaN -> 0, saturating
  else if (v >= 255.0f) out = 255;
  else out = (uint8_t)v;
  rgb24[i] = out;
}

// dst += src over n_values floats (the partial images of the GPUs of one context, summed on the first one)
__global__ void k_add(float* __restrict__ dst, const float* __restrict__ src, size_t n_values) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t quads = n_values / 4;
  const bool aligned = (((uintptr_t)dst | (uintptr_t)src) & 15u) == 0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (aligned) {
    for (size_t q = i; q < quads; q += stride) {
      float4 a = reinterpret_cast<float4*>(dst)[q];
      const float4 b = reinterpret_cast<const float4*>(src)[q];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      reinterpret_cast<float4*>(dst)[q] = a;
    }
    for (size_t k = quads * 4 + i; k < n_values; k += stride) dst[k] += src[k];
  } else {
    for (size_t k = i; k < n_values; k += stride) dst[k] += src[k];
  }
}

}  // namespace

// ---- the fixed workspace ------------------------------------------------------------------
// Every GPU the library is loaded on holds its own copy of this array, at its own address, until the process
// ends.  In host code the array's name only stands for the symbol: the address a kernel can use comes from
// hipGetSymbolAddress, asked once per device with that device current, and is what the host passes around.
static __device__ __align__(256) unsigned char scratch[STATIC_WORKSPACE_BYTES];

hipError_t static_workspace_address(int device, void** out) {
  constexpr int MAX_DEVICES = 64;
  static std::mutex guard;
  static void* address[MAX_DEVICES] = {};
  *out = nullptr;
  if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(guard);
  if (!address[device]) {
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice;
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    address[device] = p;
  }
  *out = address[device];
  return hipSuccess;
}

// ---- launchers ------------------------------------------------------------------------
RenderConfig plan_render(const DeviceScene& sc) {
  RenderConfig cfg{};
  cfg.block = RENDER_THREADS;
  cfg.grid = RENDER_BLOCKS;
  // a stack entry per inner node on the path plus the pending sibling of the last one
  const int depth = (int)sc.bvh_height + 2;
  // nodes, 48-byte primitive records, wrapper chains; then what shading looks up: materials, textures, lights
  const size_t scene_bytes = (size_t)sc.n_nodes * std::max(1u, sc.n_slices) * 64 + (size_t)sc.n_prims * PRIM_QUADS_LDS * 16 +
                             (size_t)sc.n_chain_ops * 16 + (size_t)sc.n_chains * 32 +
                             (size_t)sc.n_materials * sizeof(DMaterial) + (size_t)sc.n_textures * sizeof(DTexture) +
                             (size_t)sc.n_lights * sizeof(DLight) + ((size_t)sc.n_light_groups + 1) / 2 * 16;
  // staged scenes are small enough for 16-bit tree references, so a stack entry is 2 bytes;
  // bookkeeping, tables, then the stacks on a row of their own
  const size_t lds_total = SHARED_BYTES + (scene_bytes + 255) / 256 * 256 + (size_t)depth * RENDER_THREADS * 2;
  // (sc.keep_out: the test aid MIRT_NO_LDS, read once when the scene is uploaded -- the node layout depends on it)
  if (!sc.keep_out && !sc.wide && lds_total <= LDS_BYTES_PER_CU && sc.n_prims <= COMPACT_MAX_PRIMS && sc.n_nodes < COMPACT_MAX_NODES) {
    cfg.lds_scene = true;
    cfg.stack_depth = depth;
    cfg.lds_bytes = lds_total;
    // (noise textures gather from their tables sixteen times a lookup: from LDS when there is room behind the stacks)
    const size_t perlin_bytes = (size_t)sc.n_perlins * PERLIN_LDS_BYTES;
    if (sc.n_perlins != 0 && lds_total + perlin_bytes <= LDS_BYTES_PER_CU && !std::getenv("MIRT_NO_LDS_PERLIN")) {
      cfg.lds_perlin_offset = (uint32_t)lds_total;
      cfg.lds_bytes = lds_total + perlin_bytes;
    }
  } else {
    cfg.lds_scene = false;
    cfg.stack_depth = WIDE_LDS_STACK;
    cfg.lds_bytes = SHARED_BYTES + (size_t)WIDE_LDS_STACK * RENDER_THREADS * 4;
  }
  // (scenes in global memory: a four-child step costs three times a two-child step in LDS, so fuller tree steps are
  // worth more there than early primitive tests: +2 % from 6.5 k to 10 M spheres over 32 / 16)
  cfg.refill_min = cfg.lds_scene ? 32 : 48;
  cfg.node_min = cfg.lds_scene ? 16 : 32;
  // (from a quarter of a million primitives up a second test in the same turn pays: the record sits next to the
  // first one's, and the wave waits for memory once; +3 % at 10 M spheres, +2 % at 1 M, -2 % at 6.5 k and in LDS)
  cfg.prims_per_turn = !cfg.lds_scene && sc.n_prims >= (1u << 18) ? 2 : 1;
  if (const char* e = std::getenv("MIRT_PRIMS_PER_TURN")) {  // tuning aid
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 1 && v <= 8) cfg.prims_per_turn = (uint32_t)v;
  }
  if (const char* e = std::getenv("MIRT_REFILL_MIN")) {  // tuning aids
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 1 && v <= 64) cfg.refill_min = (uint32_t)v;
  }
  if (const char* e = std::getenv("MIRT_NODE_MIN")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 1 && v <= 64) cfg.node_min = (uint32_t)v;
  }
  return cfg;
}

template <bool COUNT, bool LDS_SCENE>
static hipError_t launch_render_as(const RenderArgs* args, const RenderConfig& cfg, hipStream_t stream) {
  // the larger dynamic-LDS limit is a property of the function on one device: set once for each device used
  // (a context serves one call at a time, but contexts on several devices may call from several threads)
  static std::atomic<unsigned long long> devices_set{0};
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (device & 63);
  if (device >= 64 || !(devices_set.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render<COUNT, LDS_SCENE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_PER_CU);
    if (e != hipSuccess) return e;
    if (device < 64) devices_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((k_render<COUNT, LDS_SCENE>), dim3(cfg.grid), dim3(cfg.block), cfg.lds_bytes, stream, args);
  return hipGetLastError();
}

hipError_t launch_render(const RenderArgs* args, bool count_work, const RenderConfig& cfg, hipStream_t stream) {
  if (cfg.lds_scene) {
    return count_work ? launch_render_as<true, true>(args, cfg, stream) : launch_render_as<false, true>(args, cfg, stream);
  }
  return count_work ? launch_render_as<true, false>(args, cfg, stream) : launch_render_as<false, false>(args, cfg, stream);
}

void launch_add(float* dst, const float* src, size_t n_values, hipStream_t stream) {
  const uint32_t block = 256;
  const size_t quads = n_values / 4;
  const uint32_t grid = (uint32_t)std::min<size_t>((quads + block - 1) / block + 1, 1u << 16);
  hipLaunchKernelGGL(k_add, dim3(grid), dim3(block), 0, stream, dst, src, n_values);
}

void launch_resolve(const float* rgb_sum, uint32_t n_values, uint32_t spp, uint8_t* rgb24, hipStream_t stream) {
  const uint32_t block = 256;
  hipLaunchKernelGGL(k_resolve, dim3((n_values + block - 1) / block), dim3(block), 0, stream, rgb_sum, n_values, spp,
                     rgb24);
}

}  // namespace mirt
