// C ABI of the MI355X path-tracing backend (include/mirt.h): context, scene
// upload (lowered arrays -> device buffers + BVH) and the launch of the render
// kernel for a frame or for a batch of caller-supplied rays.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/mirt.h"
#include "bvh_build.h"
#include "bvh_gpu.h"
#include "scene_device.h"

using namespace mirt;

namespace {

thread_local std::string g_create_error;

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t bytes = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes && ptr) return hipSuccess;
    release();
    if (n == 0) n = 16;
    const hipError_t e = hipMalloc(&ptr, n);
    if (e == hipSuccess) bytes = n;
    else ptr = nullptr;
    return e;
  }
  template <class T>
  T* as() const { return static_cast<T*>(ptr); }
};

}  // namespace

struct mirt_context {
  int device = 0;
  int num_cu = 0;
  std::string error;
  hipStream_t stream = nullptr;  // used when the caller passes no stream
  // render workspace, grown on demand and kept between calls
  DeviceBuffer paths[2];
  DeviceBuffer queues;
  DeviceBuffer saves;
  DeviceBuffer overflow;
  DeviceBuffer state;
  DeviceBuffer args;
  DeviceBuffer work;
  DeviceBuffer image_sum;
  DeviceBuffer image_rgb;
  void* static_workspace = nullptr;  // this device's copy of the library's fixed workspace (kernels.hip)
  // A context over several GPUs: this one is the first device's; the others hang off it.  mirt_render gives every
  // device a share of the frame and sums the partial images here, in one staging buffer per other device.
  std::vector<mirt_context*> others;
  std::vector<DeviceBuffer*> staging;
  uint32_t row_bands = 1;
  ~mirt_context() { for (DeviceBuffer* b : staging) delete b; }
};

struct mirt_scene {
  DeviceBuffer nodes, prims, prim_ids;
  DeviceBuffer chains, chain_ops, materials, textures, perlin_vec, perlin_perm, images, image_data, lights,
      light_groups, media;
  DeviceScene view{};
  float time0 = 0.0f, time1 = 0.0f;
  bool has_moving = false;
  uint32_t bvh_height = 0;
  size_t n_prims = 0;
  std::vector<mirt_scene*> copies;  // the scene on the other devices of a context over several GPUs, in their order
};

namespace {

int fail(mirt_context* ctx, int code, const std::string& msg) {
  if (ctx) ctx->error = msg;
  else g_create_error = msg;
  return code;
}

// No exception leaves the library (the callers are C and Rust): every entry point that allocates host memory or
// starts threads is a function-try-block that ends here.
int caught(mirt_context* ctx) noexcept {
  int code = MIRT_ERR_DEVICE;
  const char* what = "unexpected exception";
  std::string text;
  try {
    throw;
  } catch (const std::bad_alloc&) {
    code = MIRT_ERR_OUT_OF_MEMORY;
    what = "out of host memory";
  } catch (const std::exception& e) {
    try { text = e.what(); what = text.c_str(); } catch (...) {}
  } catch (...) {
  }
  try { (void)fail(ctx, code, what); } catch (...) {}
  return code;
}

// job(0) on the calling thread, job(1) .. job(n - 1) on a thread each; returns when all of them have.  A job that
// throws leaves an error code in status[k].  False when a thread could not be started: the jobs already started
// have still run to their end, job(0) has not.
template <class Job>
bool side_by_side(size_t n, const Job& job, std::vector<int>& status) {
  auto guarded = [&](size_t k) {
    try {
      job(k);
    } catch (const std::bad_alloc&) {
      status[k] = MIRT_ERR_OUT_OF_MEMORY;
    } catch (...) {
      status[k] = MIRT_ERR_DEVICE;
    }
  };
  std::vector<std::thread> workers;
  workers.reserve(n);  // (if this throws, no thread exists yet)
  bool started = true;
  for (size_t k = 1; k < n && started; ++k) {
    try {
      workers.emplace_back(guarded, k);
    } catch (const std::system_error&) {
      started = false;
    }
  }
  if (started) guarded(0);
  for (std::thread& w : workers) w.join();
  return started;
}

#define MIRT_HIP(ctx, call)                                                                      \
  do {                                                                                           \
    const hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                        \
      return fail((ctx), e_ == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_DEVICE,   \
                  std::string(#call) + ": " + hipGetErrorString(e_));                            \
  } while (0)

template <class T>
hipError_t upload(DeviceBuffer& buf, const std::vector<T>& host) {
  hipError_t e = buf.ensure(host.size() * sizeof(T));
  if (e != hipSuccess) return e;
  if (!host.empty()) e = hipMemcpy(buf.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
  return e;
}

struct P3 {
  float x, y, z;
};

// object space -> world space through a chain (innermost wrapper first)
P3 chain_to_world(const mirt_scene_desc& d, uint32_t chain, P3 p) {
  const mirt_chain& c = d.chains[chain];
  for (int k = (int)c.n_ops - 1; k >= 0; --k) {
    const mirt_chain_op& op = d.chain_ops[c.first_op + k];
    if (op.kind == MIRT_OP_TRANSLATE) {
      p.x += op.p[0]; p.y += op.p[1]; p.z += op.p[2];
    } else if (op.kind == MIRT_OP_ROTATE_Y) {
      const float s = op.p[0], co = op.p[1];
      const float x = co * p.x + s * p.z;
      const float z = -s * p.x + co * p.z;
      p.x = x; p.z = z;
    }
  }
  return p;
}

struct Bounds {
  float mn[3], mx[3];
  Bounds() {
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::numeric_limits<float>::infinity();
      mx[a] = -std::numeric_limits<float>::infinity();
    }
  }
  void add(P3 p, float r = 0.0f) {
    const float v[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], v[a] - r);
      mx[a] = std::max(mx[a], v[a] + r);
    }
  }
  // room for the rounding of the wrapper transforms themselves
  void pad() {
    for (int a = 0; a < 3; ++a) {
      const float m = std::max(std::fabs(mn[a]), std::fabs(mx[a]));
      const float e = m * 1e-5f + 1e-6f;
      mn[a] -= e;
      mx[a] += e;
    }
  }
  bool finite() const {
    for (int a = 0; a < 3; ++a)
      if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) return false;
    return true;
  }
};

void add_corners(const mirt_scene_desc& d, uint32_t chain, const float mn[3], const float mx[3], Bounds& b) {
  for (int i = 0; i < 8; ++i) {
    const P3 p{(i & 1) ? mx[0] : mn[0], (i & 2) ? mx[1] : mn[1], (i & 4) ? mx[2] : mn[2]};
    b.add(chain_to_world(d, chain, p));
  }
}

bool check_header(const mirt_scene_desc& d, const mirt_prim_header& h, std::string& why) {
  if (h.material >= d.n_materials) { why = "primitive material index out of range"; return false; }
  if (h.chain >= d.n_chains) { why = "primitive chain index out of range"; return false; }
  return true;
}

// The queue hits on material `index` are shaded in (indices are validated before this is called).
uint32_t queue_of_material(const mirt_scene_desc& d, uint32_t index) {
  const mirt_material& m = d.materials[index];
  switch (m.kind) {
    case MIRT_MAT_LAMBERTIAN:
      return d.textures[m.texture].kind == MIRT_TEX_SOLID ? QUEUE_LAMBERTIAN : QUEUE_LAMBERTIAN_TEXTURED;
    case MIRT_MAT_METAL: return QUEUE_METAL;
    case MIRT_MAT_DIELECTRIC: return QUEUE_DIELECTRIC;
    default: return QUEUE_TERMINATE;
  }
}

int validate_desc(const mirt_scene_desc& d, std::string& why) {
  if (d.n_chains == 0 || !d.chains || d.chains[0].n_ops != 0) { why = "chains[0] must be the empty chain"; return MIRT_ERR_INVALID_ARGUMENT; }
  if (d.n_chain_ops > (1u << 20)) { why = "more than 2^20 chain operations"; return MIRT_ERR_UNSUPPORTED; }
  for (uint32_t i = 0; i < d.n_chains; ++i) {
    if (d.chains[i].n_ops > MIRT_MAX_CHAIN_OPS) { why = "chain longer than MIRT_MAX_CHAIN_OPS"; return MIRT_ERR_UNSUPPORTED; }
    if ((uint64_t)d.chains[i].first_op + d.chains[i].n_ops > d.n_chain_ops) { why = "chain ops out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
    for (uint32_t k = 0; k < d.chains[i].n_ops; ++k)
      if (d.chain_ops[d.chains[i].first_op + k].kind > MIRT_OP_FLIP_FACE) { why = "unknown chain op"; return MIRT_ERR_INVALID_ARGUMENT; }
  }
  if (d.n_media > MIRT_MAX_MEDIA) { why = "more than MIRT_MAX_MEDIA constant media"; return MIRT_ERR_UNSUPPORTED; }
  for (uint32_t i = 0; i < d.n_spheres; ++i) if (!check_header(d, d.spheres[i].h, why)) return MIRT_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < d.n_moving_spheres; ++i) if (!check_header(d, d.moving_spheres[i].h, why)) return MIRT_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < d.n_rects; ++i) {
    if (!check_header(d, d.rects[i].h, why)) return MIRT_ERR_INVALID_ARGUMENT;
    if (d.rects[i].axis > MIRT_RECT_YZ) { why = "rect axis out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
  }
  for (uint32_t i = 0; i < d.n_boxes; ++i) if (!check_header(d, d.boxes[i].h, why)) return MIRT_ERR_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < d.n_materials; ++i) {
    const mirt_material& m = d.materials[i];
    if (m.kind > MIRT_MAT_ISOTROPIC) { why = "unknown material kind"; return MIRT_ERR_INVALID_ARGUMENT; }
    if (m.kind != MIRT_MAT_DIELECTRIC && m.texture >= d.n_textures) { why = "material texture index out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
  }
  for (uint32_t i = 0; i < d.n_textures; ++i) {
    const mirt_texture& t = d.textures[i];
    if (t.kind > MIRT_TEX_IMAGE) { why = "unknown texture kind"; return MIRT_ERR_INVALID_ARGUMENT; }
    if (t.kind == MIRT_TEX_CHECKER && (t.odd >= d.n_textures || t.even >= d.n_textures)) { why = "checker texture index out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
    if (t.kind == MIRT_TEX_NOISE && t.index >= d.n_perlins) { why = "noise texture perlin index out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
    if (t.kind == MIRT_TEX_IMAGE) {
      if (t.index >= d.n_images) { why = "image texture index out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
      const mirt_image& im = d.images[t.index];
      if (!im.rgb8 || im.width == 0 || im.height == 0) { why = "empty image"; return MIRT_ERR_INVALID_ARGUMENT; }
    }
  }
  for (uint32_t i = 0; i < d.n_media; ++i) {
    const mirt_medium& m = d.media[i];
    if (m.material >= d.n_materials || m.chain >= d.n_chains) { why = "medium index out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
    const uint32_t lim = m.boundary_tag == MIRT_PRIM_SPHERE          ? d.n_spheres
                         : m.boundary_tag == MIRT_PRIM_MOVING_SPHERE ? d.n_moving_spheres
                         : m.boundary_tag == MIRT_PRIM_RECT          ? d.n_rects
                         : m.boundary_tag == MIRT_PRIM_BOX           ? d.n_boxes
                                                                     : 0;
    if (m.boundary_index >= lim) { why = "medium boundary out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
  }
  if (d.lights_mode > MIRT_LIGHTS_WORLD) { why = "unknown lights_mode"; return MIRT_ERR_INVALID_ARGUMENT; }
  if (d.lights_mode != MIRT_LIGHTS_NONE && d.n_lights == 0) { why = "lights_mode set but no lights"; return MIRT_ERR_INVALID_ARGUMENT; }
  if (d.lights_mode == MIRT_LIGHTS_WORLD) {
    if (d.n_light_groups == 0) { why = "a World of lights needs light groups"; return MIRT_ERR_INVALID_ARGUMENT; }
    for (uint32_t g = 0; g < d.n_light_groups; ++g)
      if (d.light_groups[g].n_lights == 0 || (uint64_t)d.light_groups[g].first_light + d.light_groups[g].n_lights > d.n_lights) { why = "light group out of range"; return MIRT_ERR_INVALID_ARGUMENT; }
  }
  if (!(d.time0 <= d.time1)) { why = "time0 must be <= time1"; return MIRT_ERR_INVALID_ARGUMENT; }
  return MIRT_OK;
}

}  // namespace

extern "C" {

int mirt_version(void) { return 200; }  // 0.2.0: contexts over several GPUs, frames, hit records, the fixed workspace

const char* mirt_last_error(const mirt_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mirt_context_create_multi(const int* device_ids, int n_devices, uint32_t row_bands, mirt_context** out) try {
  if (!out) return fail(nullptr, MIRT_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  if (!device_ids || n_devices < 1 || n_devices > 64)
    return fail(nullptr, MIRT_ERR_INVALID_ARGUMENT, "between 1 and 64 devices, please");
  if (row_bands == 0) row_bands = 1;
  if ((uint32_t)n_devices % row_bands != 0)
    return fail(nullptr, MIRT_ERR_INVALID_ARGUMENT, "row_bands must divide the number of devices");
  mirt_context* first = nullptr;
  int rc = mirt_context_create(device_ids[0], &first);
  if (rc != MIRT_OK) return rc;
  first->row_bands = row_bands;
  struct Owner {  // the contexts made so far are this function's until the whole set is handed out
    mirt_context* ctx;
    ~Owner() { if (ctx) mirt_context_destroy(ctx); }
  } owner{first};
  first->others.reserve((size_t)n_devices);
  first->staging.reserve((size_t)n_devices);
  for (int k = 1; k < n_devices; ++k) {
    first->staging.push_back(new DeviceBuffer());  // (room reserved: only the allocation can throw, and nothing is half-owned then)
    mirt_context* other = nullptr;
    rc = mirt_context_create(device_ids[k], &other);
    if (rc != MIRT_OK) {
      const std::string why = g_create_error;
      return fail(nullptr, rc, "device " + std::to_string(device_ids[k]) + ": " + why);
    }
    first->others.push_back(other);
    if (other->device != first->device) {
      // direct copies over the GPUs' links where the platform allows them (already enabled is fine; without peer
      // access hipMemcpyPeer still works, through the host)
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, first->device, other->device) == hipSuccess && can) {
        // (hipErrorPeerAccessAlreadyEnabled is as good as success; any other failure only means the copies go
        // through the host, which MIRT_VERBOSE reports)
        auto enable = [](int on, int peer) {
          (void)hipSetDevice(on);
          const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
          (void)hipGetLastError();
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && std::getenv("MIRT_VERBOSE"))
            std::fprintf(stderr, "mirt: no peer access from device %d to %d (%s): copies go through the host\n", on, peer,
                         hipGetErrorString(e));
        };
        enable(first->device, other->device);
        enable(other->device, first->device);
      }
    }
  }
  (void)hipSetDevice(first->device);
  owner.ctx = nullptr;
  *out = first;
  return MIRT_OK;
} catch (...) {
  return caught(nullptr);
}

int mirt_context_device_count(const mirt_context* ctx) { return ctx ? 1 + (int)ctx->others.size() : 0; }

int mirt_context_create(int device_id, mirt_context** out) try {
  if (!out) return fail(nullptr, MIRT_ERR_INVALID_ARGUMENT, "out is null");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(nullptr, MIRT_ERR_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device_id < 0 || device_id >= count) return fail(nullptr, MIRT_ERR_INVALID_ARGUMENT, "device_id out of range");
  e = hipSetDevice(device_id);
  if (e != hipSuccess) return fail(nullptr, MIRT_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_id);
  if (e != hipSuccess) return fail(nullptr, MIRT_ERR_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  mirt_context* ctx = new (std::nothrow) mirt_context();
  if (!ctx) return fail(nullptr, MIRT_ERR_OUT_OF_MEMORY, "out of host memory");
  ctx->device = device_id;
  ctx->num_cu = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete ctx;
    return fail(nullptr, MIRT_ERR_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  // (device_id is current: every GPU holds its own copy of the fixed workspace at its own address)
  e = static_workspace_address(device_id, &ctx->static_workspace);
  if (e != hipSuccess) {
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return fail(nullptr, MIRT_ERR_DEVICE, std::string("hipGetSymbolAddress(static workspace): ") + hipGetErrorString(e));
  }
  *out = ctx;
  return MIRT_OK;
} catch (...) {
  return caught(nullptr);
}

void mirt_context_destroy(mirt_context* ctx) {
  if (!ctx) return;
  for (mirt_context* other : ctx->others) mirt_context_destroy(other);
  ctx->others.clear();
  (void)hipSetDevice(ctx->device);
  for (DeviceBuffer* b : ctx->staging) b->release();
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void mirt_scene_destroy(mirt_context* ctx, mirt_scene* scene) {
  if (!scene) return;
  for (size_t k = 0; k < scene->copies.size(); ++k) {
    if (ctx && k < ctx->others.size()) (void)hipSetDevice(ctx->others[k]->device);
    delete scene->copies[k];
  }
  scene->copies.clear();
  if (ctx) (void)hipSetDevice(ctx->device);
  delete scene;
}

static int upload_one(mirt_context* ctx, const mirt_scene_desc* desc, mirt_scene** out);

int mirt_scene_upload(mirt_context* ctx, const mirt_scene_desc* desc, mirt_scene** out) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!desc || !out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  // every device of the context gets its own copy, built side by side (one host thread per device)
  const size_t n_others = ctx->others.size();
  std::vector<mirt_scene*> copies(n_others, nullptr);
  std::vector<int> status(1 + n_others, MIRT_OK);  // [0]: this context's own device
  mirt_scene* first = nullptr;
  const bool started = side_by_side(1 + n_others, [&](size_t k) {
    status[k] = k == 0 ? upload_one(ctx, desc, &first) : upload_one(ctx->others[k - 1], desc, &copies[k - 1]);
  }, status);
  int rc = started ? status[0] : fail(ctx, MIRT_ERR_DEVICE, "cannot start a host thread per device");
  for (size_t k = 0; k < n_others && rc == MIRT_OK; ++k)
    if (status[k + 1] != MIRT_OK)
      rc = fail(ctx, status[k + 1], "device " + std::to_string(ctx->others[k]->device) + ": " + ctx->others[k]->error);
  if (rc != MIRT_OK) {
    for (size_t k = 0; k < n_others; ++k) {
      (void)hipSetDevice(ctx->others[k]->device);
      delete copies[k];
    }
    (void)hipSetDevice(ctx->device);
    delete first;
    return rc;
  }
  first->copies = std::move(copies);
  (void)hipSetDevice(ctx->device);
  *out = first;
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

static int upload_one(mirt_context* ctx, const mirt_scene_desc* desc, mirt_scene** out) {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!desc || !out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  const mirt_scene_desc& d = *desc;
  std::string why;
  const int vr = validate_desc(d, why);
  if (vr != MIRT_OK) return fail(ctx, vr, why);
  MIRT_HIP(ctx, hipSetDevice(ctx->device));

  mirt_scene* sp = new (std::nothrow) mirt_scene();
  if (!sp) return fail(ctx, MIRT_ERR_OUT_OF_MEMORY, "out of host memory");
  struct Guard {
    mirt_scene* s;
    ~Guard() { delete s; }
  } guard{sp};
  mirt_scene& s = *sp;
  s.time0 = d.time0;
  s.time1 = d.time1;
  s.has_moving = d.n_moving_spheres != 0;

  // ---- geometry + bounds ------------------------------------------------------
  // Every primitive gets a 4-quad record (layout in scene_device.h) and world-space bounds.  `src` numbers
  // them kind after kind; the tree build then decides the order the records are stored in.
  const size_t n_all = (size_t)d.n_spheres + d.n_moving_spheres + d.n_rects + d.n_boxes;
  if (n_all >= MAX_PRIMS) return fail(ctx, MIRT_ERR_UNSUPPORTED, "too many primitives");
  const uint32_t first_of_tag[N_PRIM_TAGS] = {0, d.n_spheres, d.n_spheres + d.n_moving_spheres,
                                              d.n_spheres + d.n_moving_spheres + d.n_rects};
  std::vector<float4> src(n_all * PRIM_QUADS, make_float4(0, 0, 0, 0));
  std::vector<uint2> src_ids(n_all);
  std::vector<uint32_t> boundary_only;
  std::vector<BuildPrim> build;
  build.reserve(n_all);
  auto bits = [](uint32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
  auto finish = [&](Bounds& b, uint32_t tag, uint32_t g, const mirt_prim_header& h, float time1) -> bool {
    src[PRIM_QUADS * (size_t)g + 2] = make_float4(bits((h.chain << 2) | tag), bits(h.order),
                                                  bits(make_material_word(queue_of_material(d, h.material), h.material)), time1);
    src[PRIM_QUADS * (size_t)g + 3] = make_float4(bits(h.flags), 0, 0, 0);
    src_ids[g] = make_uint2(h.type_index, h.object_index);
    b.pad();
    if (!b.finite()) return false;
    if (h.flags & MIRT_PRIM_FLAG_BOUNDARY) { boundary_only.push_back(g); return true; }
    BuildPrim p;
    for (int a = 0; a < 3; ++a) { p.mn[a] = b.mn[a]; p.mx[a] = b.mx[a]; }
    p.id = g;
    build.push_back(p);
    return true;
  };
  for (uint32_t i = 0; i < d.n_spheres; ++i) {
    const mirt_sphere& p = d.spheres[i];
    const uint32_t g = first_of_tag[TAG_SPHERE] + i;
    src[PRIM_QUADS * (size_t)g] = make_float4(p.center[0], p.center[1], p.center[2], p.radius);
    Bounds b;
    b.add(chain_to_world(d, p.h.chain, P3{p.center[0], p.center[1], p.center[2]}), std::fabs(p.radius));
    if (!finish(b, TAG_SPHERE, g, p.h, 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "sphere with non-finite bounds");
  }
  // the centre of a moving sphere is linear in time (objects.rs:132-135): the two ends of an interval bound it
  auto moving_bounds = [&](const mirt_moving_sphere& p, float from, float to) {
    Bounds b;
    for (int e = 0; e < 2; ++e) {
      const float time = e == 0 ? from : to;
      const float f = (time - p.time0) / (p.time1 - p.time0);
      const P3 c{p.center0[0] + f * (p.center1[0] - p.center0[0]), p.center0[1] + f * (p.center1[1] - p.center0[1]),
                 p.center0[2] + f * (p.center1[2] - p.center0[2])};
      b.add(chain_to_world(d, p.h.chain, c), std::fabs(p.radius));
    }
    return b;
  };
  for (uint32_t i = 0; i < d.n_moving_spheres; ++i) {
    const mirt_moving_sphere& p = d.moving_spheres[i];
    const uint32_t g = first_of_tag[TAG_MOVING_SPHERE] + i;
    src[PRIM_QUADS * (size_t)g] = make_float4(p.center0[0], p.center0[1], p.center0[2], p.radius);
    src[PRIM_QUADS * (size_t)g + 1] = make_float4(p.center1[0], p.center1[1], p.center1[2], p.time0);
    Bounds b = moving_bounds(p, d.time0, d.time1);
    if (!finish(b, TAG_MOVING_SPHERE, g, p.h, p.time1)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "moving sphere with non-finite bounds");
  }
  for (uint32_t i = 0; i < d.n_rects; ++i) {
    const mirt_rect& p = d.rects[i];
    const uint32_t g = first_of_tag[TAG_RECT] + i;
    src[PRIM_QUADS * (size_t)g] = make_float4(p.a0, p.a1, p.b0, p.b1);
    src[PRIM_QUADS * (size_t)g + 1] = make_float4(p.k, bits(p.axis), 0, 0);
    float mn[3], mx[3];
    if (p.axis == MIRT_RECT_XY) { mn[0] = p.a0; mx[0] = p.a1; mn[1] = p.b0; mx[1] = p.b1; mn[2] = mx[2] = p.k; }
    else if (p.axis == MIRT_RECT_XZ) { mn[0] = p.a0; mx[0] = p.a1; mn[2] = p.b0; mx[2] = p.b1; mn[1] = mx[1] = p.k; }
    else { mn[1] = p.a0; mx[1] = p.a1; mn[2] = p.b0; mx[2] = p.b1; mn[0] = mx[0] = p.k; }
    Bounds b;
    add_corners(d, p.h.chain, mn, mx, b);
    if (!finish(b, TAG_RECT, g, p.h, 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "rect with non-finite bounds");
  }
  for (uint32_t i = 0; i < d.n_boxes; ++i) {
    const mirt_box& p = d.boxes[i];
    const uint32_t g = first_of_tag[TAG_BOX] + i;
    src[PRIM_QUADS * (size_t)g] = make_float4(p.min[0], p.min[1], p.min[2], 0);
    src[PRIM_QUADS * (size_t)g + 1] = make_float4(p.max[0], p.max[1], p.max[2], 0);
    Bounds b;
    add_corners(d, p.h.chain, p.min, p.max, b);
    if (!finish(b, TAG_BOX, g, p.h, 0.0f)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "box with non-finite bounds");
  }

  // ---- acceleration structure ----------------------------------------------------
  s.n_prims = build.size();
  for (int a = 0; a < 3; ++a) { s.view.root_lo[a] = 0.0f; s.view.root_hi[a] = 0.0f; }
  for (size_t k = 0; k < build.size(); ++k)
    for (int a = 0; a < 3; ++a) {
      if (k == 0 || build[k].mn[a] < s.view.root_lo[a]) s.view.root_lo[a] = build[k].mn[a];
      if (k == 0 || build[k].mx[a] > s.view.root_hi[a]) s.view.root_hi[a] = build[k].mx[a];
    }
  uint32_t leaf_max = LEAF_MAX_PRIMS;
  if (const char* e = std::getenv("MIRT_LEAF_MAX")) {  // tuning aid
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 1 && v <= 8) leaf_max = (uint32_t)v;
  }
  // Which builder: the host's surface-area build gives the faster tree, the device's Morton-order build takes
  // milliseconds where the host takes seconds.  Left to the library, the device builds from 2^18 primitives up.
  uint32_t builder = d.build_flags & (MIRT_BUILD_HOST_SAH | MIRT_BUILD_DEVICE_MORTON);
  if (const char* e = std::getenv("MIRT_BVH_BUILDER")) {  // overrides the description: "host" or "device"
    if (!std::strcmp(e, "host")) builder = MIRT_BUILD_HOST_SAH;
    else if (!std::strcmp(e, "device")) builder = MIRT_BUILD_DEVICE_MORTON;
  }
  if (builder != MIRT_BUILD_HOST_SAH && builder != MIRT_BUILD_DEVICE_MORTON)
    builder = build.size() >= (1u << 18) ? MIRT_BUILD_DEVICE_MORTON : MIRT_BUILD_HOST_SAH;
  // Moving spheres are bounded over the whole of [time0, time1] (objects.rs:173-180 does the same).  A ray has one
  // time, though: the tree can be kept in several copies, copy k with every box fitted to where the spheres are
  // during the k-th part of the interval, a ray walking the copy of its time (same tree, same leaves, same hits).
  // Measured on the random-spheres scenes (80 % moving spheres, 485 to 10 M of them) with 2, 4 and 8 copies: 3-7 %
  // fewer tree steps, 4-10 % fewer primitive tests, and 1-11 % fewer samples per second -- the copies share neither
  // cache lines nor LDS -- so one copy is the default and more are an experiment (MIRT_TIME_SLICES).
  uint32_t n_slices = 1;
  // test aid, read here once for the scene's lifetime: keeps a small scene out of LDS (four-child nodes, global tables)
  const uint32_t keep_out = std::getenv("MIRT_NO_LDS") != nullptr ? 1u : 0u;
  if (s.has_moving && d.time1 > d.time0) {
    if (const char* e = std::getenv("MIRT_TIME_SLICES")) {
      const long v = std::strtol(e, nullptr, 10);
      if (v >= 1 && v <= 64) n_slices = (uint32_t)v;
    }
  }
  const auto t_build = std::chrono::steady_clock::now();
  uint32_t root_ref = REF_NONE, n_nodes = 0;
  std::vector<HostNode> host_nodes;  // one copy of the tree, for the fitting to the slices
  if (builder == MIRT_BUILD_DEVICE_MORTON) {
    GpuBvh g;
    MIRT_HIP(ctx, build_bvh_gpu(build, std::min(leaf_max, 2u), STACK_DEPTH - 2, ctx->stream, g));
    s.nodes.release();
    s.nodes.ptr = g.nodes;
    s.nodes.bytes = std::max<size_t>(16, (size_t)g.n_nodes * 64);
    std::vector<BuildPrim> sorted(build.size());
    for (size_t k = 0; k < build.size(); ++k) sorted[k] = build[g.order[k]];
    build.swap(sorted);
    root_ref = g.root_ref;
    n_nodes = g.n_nodes;
    s.bvh_height = g.height;
  } else {
    BuiltBvh bvh = build_bvh(build, STACK_DEPTH - 2, leaf_max);
    // A scene that just misses the render kernel's LDS budget would be walked from global memory at three quarters
    // of the speed: larger leaves (up to 8, a dearer tree step in the builder's cost) make a smaller tree, which is
    // the better trade then.
    const auto fits_lds = [&](const BuiltBvh& b) {
      DeviceScene probe{};
      probe.n_nodes = (uint32_t)b.nodes.size();
      probe.n_slices = n_slices;
      probe.bvh_height = b.height;
      probe.n_prims = (uint32_t)build.size();
      probe.n_chain_ops = d.n_chain_ops;
      probe.n_chains = d.n_chains;
      probe.n_materials = d.n_materials;
      probe.n_textures = d.n_textures;
      probe.n_lights = d.n_lights;
      probe.n_light_groups = d.n_light_groups;
      probe.keep_out = keep_out;
      return plan_render(probe).lds_scene;
    };
    const bool retry = !std::getenv("MIRT_NO_LDS_RETRY");  // measurement aid
    for (float step_cost = 2.0f; retry && step_cost <= 4.0f && build.size() <= 4096 && !fits_lds(bvh); step_cost *= 2.0f) {
      std::vector<BuildPrim> order = build;  // (the builder leaves the primitives in the order of its tree)
      BuiltBvh smaller = build_bvh(order, STACK_DEPTH - 2, 8, step_cost);
      if (fits_lds(smaller)) { bvh = std::move(smaller); build.swap(order); }  // (a much coarser tree would not pay)
    }
    // (a scene whose copies do not fit beside each other in LDS is better off with one copy there)
    while (n_slices > 1 && !fits_lds(bvh) && build.size() <= 4096) {
      n_slices /= 2;
      if (fits_lds(bvh)) break;
    }
    if (n_slices > 1 && !fits_lds(bvh) && build.size() <= 4096) n_slices = 1;
    MIRT_HIP(ctx, upload(s.nodes, bvh.nodes));
    root_ref = bvh.root_ref;
    n_nodes = (uint32_t)bvh.nodes.size();
    s.bvh_height = bvh.height;
    if (n_slices > 1) host_nodes = std::move(bvh.nodes);
  }
  if (n_slices > 1 && n_nodes != 0) {
    if (host_nodes.empty()) {  // built on the device
      host_nodes.resize(n_nodes);
      MIRT_HIP(ctx, hipMemcpy(host_nodes.data(), s.nodes.ptr, (size_t)n_nodes * sizeof(HostNode), hipMemcpyDeviceToHost));
    }
    std::vector<HostNode> all((size_t)n_nodes * n_slices);
    std::vector<BuildPrim> fitted = build;  // tree order
    const float span = (d.time1 - d.time0) / (float)n_slices;
    for (uint32_t k = 0; k < n_slices; ++k) {
      // (a little over the slice's ends: the slice of a ray is worked out in f32 on the device)
      const float from = std::max(d.time0, d.time0 + span * ((float)k - 0.001f));
      const float to = std::min(d.time1, d.time0 + span * ((float)k + 1.001f));
      for (size_t j = 0; j < build.size(); ++j) {
        const uint32_t g = build[j].id;
        if (g < first_of_tag[TAG_MOVING_SPHERE] || g >= first_of_tag[TAG_RECT]) continue;
        Bounds b = moving_bounds(d.moving_spheres[g - first_of_tag[TAG_MOVING_SPHERE]], from, to);
        b.pad();
        for (int a = 0; a < 3; ++a) { fitted[j].mn[a] = b.mn[a]; fitted[j].mx[a] = b.mx[a]; }
      }
      std::vector<HostNode> copy = host_nodes;
      refit_bvh(copy, root_ref, fitted);
      std::copy(copy.begin(), copy.end(), all.begin() + (size_t)k * n_nodes);
    }
    s.nodes.release();
    MIRT_HIP(ctx, upload(s.nodes, all));
  } else {
    n_slices = 1;
  }
  if (s.bvh_height > (uint32_t)STACK_DEPTH - 2) return fail(ctx, MIRT_ERR_UNSUPPORTED, "BVH taller than the traversal stack");
  // A scene that will be walked from global memory gets four-child nodes: every step is a dependent fetch there,
  // and a ray reaches its leaves in fewer than half as many.
  bool wide = false;
  {
    DeviceScene probe{};
    probe.n_nodes = n_nodes;
    probe.n_slices = n_slices;
    probe.bvh_height = s.bvh_height;
    probe.n_prims = (uint32_t)build.size();
    probe.n_chain_ops = d.n_chain_ops;
    probe.n_chains = d.n_chains;
    probe.n_materials = d.n_materials;
    probe.n_textures = d.n_textures;
    probe.n_lights = d.n_lights;
    probe.n_light_groups = d.n_light_groups;
    probe.keep_out = keep_out;
    wide = !plan_render(probe).lds_scene && n_nodes != 0;
  }
  if (wide) {
    if (host_nodes.empty()) {  // the slices above did not need them; built on the device or already uploaded
      host_nodes.resize(n_nodes);
      MIRT_HIP(ctx, hipMemcpy(host_nodes.data(), s.nodes.ptr, (size_t)n_nodes * sizeof(HostNode), hipMemcpyDeviceToHost));
    }
    const WideBvh w = widen_bvh(host_nodes, root_ref);
    host_nodes.clear();
    host_nodes.shrink_to_fit();
    if (3 * w.height + 1 > (uint32_t)WIDE_MAX_STACK) return fail(ctx, MIRT_ERR_UNSUPPORTED, "BVH taller than the traversal stack");
    s.nodes.release();
    MIRT_HIP(ctx, upload(s.nodes, quantize_wide(w)));
    root_ref = w.root_ref;
    n_nodes = (uint32_t)w.nodes.size();
    s.bvh_height = w.height;
    n_slices = 1;
    if (std::getenv("MIRT_VERBOSE"))
      std::fprintf(stderr, "mirt: folded into %u four-child nodes, height %u\n", n_nodes, s.bvh_height);
  }
  if (!wide && n_nodes != 0) {
    // Two-child trees (scenes the render kernel stages in LDS): how much each child box has to grow with the
    // distance to the ray origin depends on the smallest sphere below it (scene_device.h, BOX_GROW_SPHERE).
    std::vector<float> growth(build.size(), 0.0f);
    for (size_t j = 0; j < build.size(); ++j) {
      const uint32_t g = build[j].id;
      if (g >= first_of_tag[TAG_RECT]) continue;  // rects and boxes: the linear part is enough
      const float r = std::fabs(src[PRIM_QUADS * (size_t)g].w);
      growth[j] = std::min(BOX_GROW_SPHERE / std::max(r, 1e-30f), 1e30f);
    }
    std::vector<HostNode> all((size_t)n_nodes * n_slices);
    MIRT_HIP(ctx, hipMemcpy(all.data(), s.nodes.ptr, all.size() * sizeof(HostNode), hipMemcpyDeviceToHost));
    set_box_growth(all, n_nodes, n_slices, root_ref, growth);
    MIRT_HIP(ctx, hipMemcpy(s.nodes.ptr, all.data(), all.size() * sizeof(HostNode), hipMemcpyHostToDevice));
  }
  if (std::getenv("MIRT_VERBOSE"))
    std::fprintf(stderr, "mirt: %s BVH build of %zu primitives: %.1f ms, %u nodes, height %u, %u time slice(s)\n",
                 builder == MIRT_BUILD_DEVICE_MORTON ? "device (Morton order)" : "host (surface-area)", build.size(),
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count(), n_nodes,
                 s.bvh_height, n_slices);
  // records in tree order, boundary-only records behind them
  std::vector<float4> rec(n_all * PRIM_QUADS);
  std::vector<uint2> ids(n_all);
  std::vector<uint32_t> position(n_all, 0);  // src number -> record index
  {
    size_t pos = 0;
    auto place = [&](uint32_t g) {
      for (uint32_t k = 0; k < PRIM_QUADS; ++k) rec[PRIM_QUADS * pos + k] = src[PRIM_QUADS * (size_t)g + k];
      ids[pos] = src_ids[g];
      position[g] = (uint32_t)pos++;
    };
    for (const BuildPrim& bp : build) place(bp.id);
    for (uint32_t g : boundary_only) place(g);
  }
  MIRT_HIP(ctx, upload(s.prims, rec));
  MIRT_HIP(ctx, upload(s.prim_ids, ids));

  // ---- wrappers, media, materials, textures, lights ------------------------------------
  // Chain heads: a chain that is [translate][rotate][translate] with any of the three left out (and one offset)
  // carries its numbers in the head, shape bits 1 | 2 | 4; anything else has shape 8 and is walked op by op.
  std::vector<float4> chains(2 * (size_t)d.n_chains);
  for (uint32_t i = 0; i < d.n_chains; ++i) {
    const mirt_chain& c = d.chains[i];
    uint32_t shape = 0;
    float off[3] = {0, 0, 0}, sn = 0, cs = 0;
    bool simple = true;
    int stage = 0;  // 0 nothing yet, 1 after the first translate, 2 after the rotation, 3 after the last translate
    for (uint32_t k = 0; k < c.n_ops && simple; ++k) {
      const mirt_chain_op& op = d.chain_ops[c.first_op + k];
      if (op.kind == MIRT_OP_TRANSLATE && stage == 0 ) {
        shape |= 1; stage = 1; off[0] = op.p[0]; off[1] = op.p[1]; off[2] = op.p[2];
      } else if (op.kind == MIRT_OP_TRANSLATE && stage == 2 && !(shape & 1)) {
        shape |= 4; stage = 3; off[0] = op.p[0]; off[1] = op.p[1]; off[2] = op.p[2];
      } else if (op.kind == MIRT_OP_ROTATE_Y && stage <= 1) {
        shape |= 2; stage = 2; sn = op.p[0]; cs = op.p[1];
      } else {
        simple = false;
      }
    }
    if (!simple) shape = 8;
    const uint32_t word = shape | (c.n_ops << 8) | (c.first_op << 12);
    float word_bits;
    std::memcpy(&word_bits, &word, 4);
    chains[2 * (size_t)i] = make_float4(off[0], off[1], off[2], word_bits);
    chains[2 * (size_t)i + 1] = make_float4(sn, cs, 0.0f, 0.0f);
  }
  std::vector<float4> ops(d.n_chain_ops);
  for (uint32_t i = 0; i < d.n_chain_ops; ++i) {
    float kind_bits;
    std::memcpy(&kind_bits, &d.chain_ops[i].kind, 4);
    ops[i] = make_float4(kind_bits, d.chain_ops[i].p[0], d.chain_ops[i].p[1], d.chain_ops[i].p[2]);
  }
  MIRT_HIP(ctx, upload(s.chains, chains));
  MIRT_HIP(ctx, upload(s.chain_ops, ops));

  std::vector<DMaterial> mats(d.n_materials);
  for (uint32_t i = 0; i < d.n_materials; ++i) mats[i] = DMaterial{d.materials[i].kind, d.materials[i].texture, d.materials[i].param, 0};
  MIRT_HIP(ctx, upload(s.materials, mats));

  std::vector<DTexture> texs(d.n_textures);
  for (uint32_t i = 0; i < d.n_textures; ++i) {
    const mirt_texture& t = d.textures[i];
    texs[i] = DTexture{t.kind, t.odd, t.even, t.index, t.rgb[0], t.rgb[1], t.rgb[2], t.scale};
  }
  MIRT_HIP(ctx, upload(s.textures, texs));

  std::vector<float4> pvec((size_t)d.n_perlins * 256);
  std::vector<uint32_t> pperm((size_t)d.n_perlins * 768);
  for (uint32_t i = 0; i < d.n_perlins; ++i) {
    const mirt_perlin& p = d.perlins[i];
    for (int k = 0; k < 256; ++k) {
      pvec[(size_t)i * 256 + k] = make_float4(p.random_vec[k][0], p.random_vec[k][1], p.random_vec[k][2], 0);
      pperm[(size_t)i * 768 + k] = p.perm_x[k] & 255u;
      pperm[(size_t)i * 768 + 256 + k] = p.perm_y[k] & 255u;
      pperm[(size_t)i * 768 + 512 + k] = p.perm_z[k] & 255u;
    }
  }
  MIRT_HIP(ctx, upload(s.perlin_vec, pvec));
  MIRT_HIP(ctx, upload(s.perlin_perm, pperm));

  std::vector<DImage> images(d.n_images);
  std::vector<uint8_t> image_data;
  for (uint32_t i = 0; i < d.n_images; ++i) {
    const mirt_image& im = d.images[i];
    const size_t bytes = (size_t)im.width * im.height * 3;
    images[i] = DImage{(uint64_t)image_data.size(), im.width, im.height};
    if (im.rgb8) image_data.insert(image_data.end(), im.rgb8, im.rgb8 + bytes);
  }
  MIRT_HIP(ctx, upload(s.images, images));
  MIRT_HIP(ctx, upload(s.image_data, image_data));

  std::vector<DLight> lights(d.n_lights);
  for (uint32_t i = 0; i < d.n_lights; ++i) {
    lights[i].kind = d.lights[i].kind;
    for (int k = 0; k < 5; ++k) lights[i].p[k] = d.lights[i].p[k];
    lights[i].pad[0] = lights[i].pad[1] = 0;
  }
  std::vector<uint2> groups(d.n_light_groups);
  for (uint32_t i = 0; i < d.n_light_groups; ++i) groups[i] = make_uint2(d.light_groups[i].first_light, d.light_groups[i].n_lights);
  MIRT_HIP(ctx, upload(s.lights, lights));
  MIRT_HIP(ctx, upload(s.light_groups, groups));

  DeviceScene& v = s.view;
  v.nodes = s.nodes.as<float4>();
  v.root_ref = root_ref;
  v.n_nodes = n_nodes;
  v.n_slices = n_slices;
  v.slice_time0 = d.time0;
  v.slice_scale = n_slices > 1 ? (float)n_slices / (d.time1 - d.time0) : 0.0f;
  v.wide = wide ? 1u : 0u;
  v.keep_out = keep_out;
  {  // the four-child walk's box padding takes the scene's smallest sphere radius (scene_device.h)
    float r_min = 1e15f;
    for (size_t j = 0; j < build.size(); ++j) {
      const uint32_t g = build[j].id;
      if (g >= first_of_tag[TAG_RECT]) continue;
      r_min = std::min(r_min, std::fabs(src[PRIM_QUADS * (size_t)g].w));
    }
    v.min_radius = std::max(r_min, 0.0f);
  }
  v.bvh_height = s.bvh_height;
  v.n_chains = d.n_chains;
  v.prims = s.prims.as<float4>();
  v.prim_ids = s.prim_ids.as<uint2>();
  v.n_prims = (uint32_t)build.size();
  v.n_chain_ops = d.n_chain_ops;
  v.chains = s.chains.as<float4>();
  v.chain_ops = s.chain_ops.as<float4>();
  v.n_media = d.n_media;
  v.n_perlins = d.n_perlins;
  std::vector<DMedium> media(d.n_media);
  for (uint32_t i = 0; i < d.n_media; ++i) {
    const mirt_medium& m = d.media[i];
    DMedium& dm = media[i];
    dm.boundary_ref = make_leaf(m.boundary_tag, position[first_of_tag[m.boundary_tag] + m.boundary_index]);
    dm.neg_inv_density = m.neg_inv_density;
    dm.material_word = make_material_word(queue_of_material(d, m.material), m.material);
    dm.chain = m.chain;
    dm.order = m.order;
    dm.type_index = m.type_index;
    dm.object_index = m.object_index;
    dm.pad = 0;
  }
  MIRT_HIP(ctx, upload(s.media, media));
  v.media = s.media.as<DMedium>();
  v.n_materials = d.n_materials;
  v.n_textures = d.n_textures;
  v.materials = s.materials.as<DMaterial>();
  v.textures = s.textures.as<DTexture>();
  v.perlin_vec = s.perlin_vec.as<float4>();
  v.perlin_perm = s.perlin_perm.as<uint32_t>();
  v.images = s.images.as<DImage>();
  v.image_data = s.image_data.as<uint8_t>();
  v.lights_mode = d.lights_mode;
  v.n_lights = d.n_lights;
  v.n_light_groups = d.n_light_groups;
  v.lights = s.lights.as<DLight>();
  v.light_groups = s.light_groups.as<uint2>();

  guard.s = nullptr;
  *out = sp;
  return MIRT_OK;
}

}  // extern "C"

namespace {

// What differs between a render and a trace of caller-supplied rays.
struct Job {
  RenderConst rc{};               // pool and region_cap are filled in by run_job
  float* rgb_sum = nullptr;       // render
  const mirt_ray* rays = nullptr; // trace
  uint32_t* hit_type = nullptr;
  uint32_t* hit_object = nullptr;
  float* hit_t = nullptr;
  float4* hit_records = nullptr;
  uint64_t want_pool = 0;
  uint32_t flags = 0;
  bool fixed_workspace = false;  // work in the library's fixed workspace: no allocation of any kind
};

// Device memory a job works in besides its inputs and outputs.
struct Workspace {
  float4* paths[2];
  float4* queues;
  RenderState* state;
  WorkCounters* work;
  RenderArgs* args;
  uint4* saves;       // walks put aside: 2 * regions * save_cap entries of SAVE_QUADS quads
  uint32_t save_cap;
};

// The fixed workspace, cut up: [control page: state, counters, arguments][paths a][paths b][queues].
Workspace carve_static_workspace(void* base) {
  static_assert(sizeof(RenderState) <= 256 && sizeof(WorkCounters) <= 512 && sizeof(RenderArgs) <= STATIC_CONTROL_BYTES - 768,
                "the control page of the fixed workspace is too small");
  unsigned char* p = static_cast<unsigned char*>(base);
  Workspace w;
  w.state = reinterpret_cast<RenderState*>(p);
  w.work = reinterpret_cast<WorkCounters*>(p + 256);
  w.args = reinterpret_cast<RenderArgs*>(p + 768);
  p += STATIC_CONTROL_BYTES;
  w.paths[0] = reinterpret_cast<float4*>(p);
  w.paths[1] = reinterpret_cast<float4*>(p + (size_t)STATIC_POOL_PATHS * 64);
  w.queues = reinterpret_cast<float4*>(p + (size_t)STATIC_POOL_PATHS * 128);
  w.saves = reinterpret_cast<uint4*>(p + (size_t)STATIC_POOL_PATHS * 64 * (2 + N_QUEUES));
  w.save_cap = STATIC_SAVE_CAP;
  return w;
}

// Sizes the path pool, launches the render kernel once and waits for it.
int run_job(mirt_context* ctx, const mirt_scene* scene, Job& job, hipStream_t stream, mirt_stats* stats) {
  const uint64_t total = job.rc.total_paths;
  if (job.fixed_workspace && job.want_pool > STATIC_POOL_PATHS)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "path_pool beyond the " + std::to_string(STATIC_POOL_PATHS) +
                                                    " paths the fixed workspace holds");
  RenderConfig cfg = plan_render(scene->view);
  if (job.fixed_workspace) cfg.block = (int)STATIC_WAVES_PER_BLOCK * 64;  // fewer waves, so a region still holds a wave's worth
  const uint32_t regions = cfg.regions();
  // paths in flight: every wave's region a whole number of 64 slots, no more than the job has paths
  const uint32_t granule = regions * 64u;
  // (left to the library: a sixteenth of the job's paths, between 8 Mi and 32 Mi -- 5 to 21 GB of records; on a 1080p
  // frame 32 Mi is worth 3.5 % over 8 Mi at 1024 spp, 16 Mi 1.4 % at 128 spp; 4 Mi would cost 15 %, 2 Mi half)
  // (scenes walked from global memory: an eighth -- their rounds are longer and their regions' tails cost more; on the
  // 10 M-sphere frame at 64 spp 16 Mi is worth 4 % over 8 Mi, 32 Mi no more)
  uint64_t want = job.want_pool ? job.want_pool
                  : job.fixed_workspace ? STATIC_POOL_PATHS
                                        : std::min<uint64_t>(std::max<uint64_t>(total / (cfg.lds_scene ? 16 : 8), 1u << 23), 1u << 25);
  want = std::min<uint64_t>(want, total);
  want = std::min<uint64_t>(std::max<uint64_t>(want, granule), 1u << 27);

  // Large scenes (from 2^18 primitives): a ray takes at most `budget` steps per round; longer walks are put aside
  // and go on in the next round, so the few rays that graze such a scene for thousands of steps no longer hold up
  // every round.  Smaller scenes have no such rays and would only pay for the bookkeeping.
  uint32_t step_budget = 0;
  if (!cfg.lds_scene && !job.rays && (scene->n_prims >= (1u << 18) || std::getenv("MIRT_STEP_BUDGET"))) {
    // (a step is one four-child node or one primitive test.  1 M spheres: 128 to 256 steps is best, +22 % over
    // walking a thirty-second of the scene's diagonal per round and +85 % over whole rays; 10 M: 256 to 512)
    long budget = 256;
    if (const char* e = std::getenv("MIRT_STEP_BUDGET")) budget = std::strtol(e, nullptr, 10);  // tuning aid; 0 = whole rays
    if (budget > 0) step_budget = (uint32_t)std::min<long>(budget, 1 << 30);
  }

  Workspace ws;
  uint32_t pool = 0;
  if (job.fixed_workspace) {
    pool = (uint32_t)((want + granule - 1) / granule * granule);
    if (pool > STATIC_POOL_PATHS) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "pool beyond the fixed workspace");
    if (!ctx->static_workspace) return fail(ctx, MIRT_ERR_DEVICE, "the fixed workspace of this device was not found");
    ws = carve_static_workspace(ctx->static_workspace);
  } else {
    MIRT_HIP(ctx, ctx->state.ensure(sizeof(RenderState)));
    MIRT_HIP(ctx, ctx->work.ensure(sizeof(WorkCounters)));
    MIRT_HIP(ctx, ctx->args.ensure(sizeof(RenderArgs)));
    // The pool is 640 bytes a path (5 to 21 GB at the default sizes) plus, with a step budget, 96 bytes a path of
    // save area.  When the device cannot give that much -- other contexts, other processes -- the pool is halved,
    // down to one wave's worth per region, before the call gives up: a smaller pool is slower, not wrong.
    for (;;) {
      pool = (uint32_t)((want + granule - 1) / granule * granule);
      // (the save area is only there for walks with a step budget; sized from the pool actually asked for)
      const uint32_t save_cap = step_budget ? std::max(64u, pool / regions / 4) : 0u;
      hipError_t e = ctx->paths[0].ensure((size_t)pool * 64);
      if (e == hipSuccess) e = ctx->paths[1].ensure((size_t)pool * 64);
      if (e == hipSuccess) e = ctx->queues.ensure((size_t)pool * 64 * N_QUEUES);
      if (e == hipSuccess) e = ctx->saves.ensure(std::max<size_t>(256, (size_t)2 * regions * save_cap * SAVE_QUADS * 16));
      if (e == hipSuccess) {
        ws = Workspace{{ctx->paths[0].as<float4>(), ctx->paths[1].as<float4>()}, ctx->queues.as<float4>(),
                       ctx->state.as<RenderState>(), ctx->work.as<WorkCounters>(), ctx->args.as<RenderArgs>(),
                       ctx->saves.as<uint4>(), save_cap};
        break;
      }
      (void)hipGetLastError();  // (the failed allocation's sticky error)
      ctx->paths[0].release(); ctx->paths[1].release(); ctx->queues.release(); ctx->saves.release();
      if (e != hipErrorOutOfMemory || pool <= granule || job.want_pool != 0)
        return fail(ctx, e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_DEVICE,
                    std::string("path pool of ") + std::to_string(pool) + " paths: " + hipGetErrorString(e));
      want = std::max<uint64_t>(granule, pool / 2);
      if (std::getenv("MIRT_VERBOSE")) std::fprintf(stderr, "mirt: out of device memory at %u paths, trying %llu\n", pool, (unsigned long long)want);
    }
  }
  job.rc.pool = pool;
  job.rc.region_cap = pool / regions;
  // (the workspace keeps the last call's contents: the counters must start at zero, the rest is written before read)
  MIRT_HIP(ctx, hipMemsetAsync(ws.state, 0, sizeof(RenderState), stream));
  MIRT_HIP(ctx, hipMemsetAsync(ws.work, 0, sizeof(WorkCounters), stream));

  const bool count_work = (job.flags & MIRT_RENDER_COUNT_WORK) != 0;
  const bool time_kernels = (job.flags & MIRT_RENDER_TIME_KERNELS) != 0 && !job.fixed_workspace;
  if (std::getenv("MIRT_VERBOSE"))
    std::fprintf(stderr, "mirt: %zu prims, %u nodes, height %u; %s, %d x %d threads, %zu B LDS, stack %d; pool %u\n",
                 scene->n_prims, scene->view.n_nodes, scene->view.bvh_height,
                 cfg.lds_scene ? "scene in LDS" : "scene in global memory", cfg.grid, cfg.block, cfg.lds_bytes,
                 cfg.stack_depth, pool);
  if (std::getenv("MIRT_VERBOSE"))
    std::fprintf(stderr, "mirt: paths %p %p (%zu B each), queues %p (%zu B), saves %p, state %p, work %p, args %p, image %p\n",
                 (void*)ws.paths[0], (void*)ws.paths[1], (size_t)pool * 64, (void*)ws.queues, (size_t)pool * 64 * N_QUEUES,
                 (void*)ws.saves, (void*)ws.state, (void*)ws.work, (void*)ws.args, (void*)job.rgb_sum);

  hipEvent_t ev[2] = {nullptr, nullptr};
  struct EventGuard {
    hipEvent_t* ev;
    ~EventGuard() { for (int k = 0; k < 2; ++k) if (ev[k]) (void)hipEventDestroy(ev[k]); }
  } event_guard{ev};
  if (time_kernels) {
    MIRT_HIP(ctx, hipEventCreate(&ev[0]));
    MIRT_HIP(ctx, hipEventCreate(&ev[1]));
    MIRT_HIP(ctx, hipEventRecord(ev[0], stream));
  }
  const auto t_begin = std::chrono::steady_clock::now();
  RenderArgs args{};
  args.sc = scene->view;
  args.rc = job.rc;
  args.paths_a = ws.paths[0];
  args.paths_b = ws.paths[1];
  args.queues = ws.queues;
  args.state = ws.state;
  args.work = ws.work;
  args.rgb_sum = job.rgb_sum;
  args.refill_min = cfg.refill_min;
  args.waves_per_block = (uint32_t)cfg.block / 64u;
  args.stack_overflow = nullptr;
  args.overflow_levels = 0;
  if (scene->view.wide && 3 * scene->view.bvh_height + 1 > (uint32_t)WIDE_LDS_STACK) {
    // (a very unbalanced tree: the deep end of the stacks lives in global memory)
    if (job.fixed_workspace) return fail(ctx, MIRT_ERR_UNSUPPORTED, "this scene's tree needs stack rows the fixed workspace does not hold");
    args.overflow_levels = 3 * scene->view.bvh_height + 1 - (uint32_t)WIDE_LDS_STACK;
    MIRT_HIP(ctx, ctx->overflow.ensure((size_t)cfg.grid * args.overflow_levels * RENDER_THREADS * 4));
    args.stack_overflow = ctx->overflow.as<uint32_t>();
  }
  args.step_budget = step_budget;
  args.save_cap = ws.save_cap;
  args.saves = ws.saves;
  args.node_min = cfg.node_min;
  args.prims_per_turn = cfg.prims_per_turn;
  args.lds_perlin_offset = cfg.lds_perlin_offset;
  // (scenes staged in LDS: +1 % on final_scene; the 10 M-sphere field walked from global memory lost 5 % with it --
  // its rays that head up leave at once, and the waves left with the others alone run emptier)
  args.split_survivors = cfg.lds_scene && !std::getenv("MIRT_ONE_SURVIVOR_STREAM") ? 1u : 0u;
  args.rays = job.rays;
  args.hit_type = job.hit_type;
  args.hit_object = job.hit_object;
  args.hit_t = job.hit_t;
  args.hit_records = job.hit_records;
  // (pageable source: the copy has left `args` by the time the call returns)
  MIRT_HIP(ctx, hipMemcpyAsync(ws.args, &args, sizeof(args), hipMemcpyHostToDevice, stream));
  MIRT_HIP(ctx, launch_render(ws.args, count_work, cfg, stream));
  if (time_kernels) MIRT_HIP(ctx, hipEventRecord(ev[1], stream));
  MIRT_HIP(ctx, hipStreamSynchronize(stream));
  const auto t_end = std::chrono::steady_clock::now();
#ifdef MIRT_SURVIVOR_AUDIT
  {
    WorkCounters w{};
    MIRT_HIP(ctx, hipMemcpy(&w, ws.work, sizeof(w), hipMemcpyDeviceToHost));
    std::fprintf(stderr, "mirt audit: survivors / unwritten slots per wave of a block:");
    for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %llu/%llu", w.audit[k] & 0xFFFFFFFFFFull, w.audit[16 + k]);
    std::fprintf(stderr, "\n");
    if (w.check[7]) {
      std::fprintf(stderr, "mirt audit: first region with unwritten slots: %llu survivors, wave %llu of its block, round %llu; queues of that round:",
                   w.check[6] >> 32, (w.check[6] >> 16) & 0xFFFF, w.check[6] & 0xFFFF);
      for (int q = 0; q < N_QUEUES; ++q) std::fprintf(stderr, " %llu", w.audit[q] >> 40);
      std::fprintf(stderr, "\n   unwritten slots, 64 per word from slot 0:");
      for (int k = 0; k < 6; ++k) std::fprintf(stderr, " %016llx", w.check[k]);
      std::fprintf(stderr, "\n");
    }
  }
#endif
#ifdef MIRT_CHECKED
  {  // diagnostic build: an index the kernel found out of range ends the call with what it recorded
    WorkCounters w{};
    MIRT_HIP(ctx, hipMemcpy(&w, ws.work, sizeof(w), hipMemcpyDeviceToHost));
    if (w.check[0] != 0) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "index check %llu failed (%llu times): values %llu %llu %llu, block %llu thread %llu",
                    w.check[0], w.check[6], w.check[1], w.check[2], w.check[3], w.check[4], w.check[5]);
      std::fprintf(stderr, "mirt: %s\n", msg);
      return fail(ctx, MIRT_ERR_DEVICE, msg);
    }
  }
#endif
  if (stats) {
    stats->render_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
    stats->path_pool = pool;
    stats->scene_in_lds = cfg.lds_scene ? 1u : 0u;
    stats->lds_bytes = (uint32_t)cfg.lds_bytes;
    WorkCounters w{};
    MIRT_HIP(ctx, hipMemcpy(&w, ws.work, sizeof(w), hipMemcpyDeviceToHost));
    stats->iterations = w.rounds;
    if (count_work) {
      stats->segments = w.segments;
      stats->nodes_visited = w.nodes;
      stats->prims_tested = w.prims;
      stats->node_wave_steps = w.wave_nodes;
      stats->prim_wave_steps = w.wave_prims;
      for (int k = 0; k < MIRT_PHASE_COUNT; ++k) stats->phase_cycles[k] = w.cycles[k];
    }
    if (time_kernels) {
      float ms = 0.0f;
      MIRT_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
      stats->kernel_ms[MIRT_KERNEL_RENDER] = ms;
      stats->kernel_launches[MIRT_KERNEL_RENDER] = 1;
    }
  }
  return MIRT_OK;
}

int render_impl(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera, const mirt_render_params* params,
                float* rgb_sum_device, hipStream_t stream, mirt_stats* stats, bool fixed_workspace = false) {
  if (!scene || !camera || !params || !rgb_sum_device) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  const mirt_render_params& p = *params;
  if (p.width < 2 || p.height < 2) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "width and height must be at least 2");
  if ((uint64_t)p.width * p.height >= (1ull << 31)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "image too large");
  if (p.samples_per_pixel == 0) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be positive");
  if ((uint64_t)p.first_sample + p.samples_per_pixel - 1 > MAX_SAMPLE_INDEX)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "sample index beyond 2^24 - 1");
  if (p.max_depth > MAX_DEPTH_LIMIT) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "max_depth above 255");
  const uint32_t n_rows = p.n_rows ? p.n_rows : p.height;
  if ((p.n_rows == 0 && p.first_row != 0) || (uint64_t)p.first_row + n_rows > p.height)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "rows [first_row, first_row + n_rows) outside the image");
  if (!(camera->time0 <= camera->time1)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "camera time0 must be <= time1");
  if (scene->has_moving && (camera->time0 < scene->time0 || camera->time1 > scene->time1))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "camera shutter interval outside the scene's time range");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const uint64_t total = (uint64_t)p.width * n_rows * p.samples_per_pixel;
  if (stats) stats->samples = total;
  if (p.max_depth == 0) return MIRT_OK;  // renderer.rs:147-149: every sample is black

  Job job;
  RenderConst& rc = job.rc;
  rc.cam = *camera;
  rc.width = p.width;
  rc.height = p.height;
  rc.band_rows = n_rows;
  rc.band_y0 = p.height - p.first_row - n_rows;  // the kernel counts y upwards from the bottom row (renderer.rs:216-217)
  rc.first_sample = p.first_sample;
  rc.max_depth = p.max_depth;
  rc.background[0] = p.background[0]; rc.background[1] = p.background[1]; rc.background[2] = p.background[2];
  rc.key0 = (uint32_t)p.seed;
  rc.key1 = (uint32_t)(p.seed >> 32);
  rc.total_paths = total;
  rc.n_groups = (uint32_t)(((uint64_t)p.width * n_rows) / 64);
  rc.group_stride = 1;
  if (rc.n_groups > 1) {
    // about 61 strides across the frame, the nearest one that is coprime to the group count
    uint64_t stride = std::max<uint64_t>(1, rc.n_groups / 61);
    stride = std::min<uint64_t>(stride, 0xFFFFFFFFull / rc.n_groups);
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    while (stride > 1 && gcd(stride, rc.n_groups) != 1) --stride;
    rc.group_stride = (uint32_t)stride;
  }
  rc.by_pixels = make_fastdiv(p.width * n_rows);
  rc.by_groups = make_fastdiv(std::max(1u, rc.n_groups));
  rc.by_band = make_fastdiv(p.width * 8u);
  rc.by_width = make_fastdiv(p.width);
  rc.estimator = p.flags & (MIRT_RENDER_EVEN_LIGHTS | MIRT_RENDER_DROP_NONFINITE);
  rc.strata = (p.flags & MIRT_RENDER_STRATA_MASK) >> MIRT_RENDER_STRATA_SHIFT;
  rc.by_strata = make_fastdiv(std::max(1u, rc.strata));
  rc.by_cells = make_fastdiv(std::max(1u, rc.strata * rc.strata));
  job.rgb_sum = rgb_sum_device;
  job.want_pool = p.path_pool;
  job.flags = p.flags;
  job.fixed_workspace = fixed_workspace;
  if (!fixed_workspace) return run_job(ctx, scene, job, stream, stats);
  // Every call on a GPU works in that GPU's one fixed workspace: a call made while another one is still running
  // there (from another thread; the calls block) is turned away instead of racing with it.
  static std::atomic<bool> busy[64];
  std::atomic<bool>& flag = busy[(unsigned)ctx->device & 63u];
  if (flag.exchange(true, std::memory_order_acquire))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "the fixed workspace of this GPU is in use by another call");
  struct Release {
    std::atomic<bool>& f;
    ~Release() { f.store(false, std::memory_order_release); }
  } release{flag};
  return run_job(ctx, scene, job, stream, stats);
}

// One share of a frame rendered by a context over several GPUs.
struct Share {
  uint32_t first_sample = 0, n_samples = 0;  // within the call's samples
  uint32_t first_row = 0, n_rows = 0;        // within the call's rows
};

// The frame of `params` cut into one share per device: row_bands bands of rows times n / row_bands sample
// ranges, device k taking band k / ranges, range k % ranges.  A share may be empty (fewer samples than ranges).
std::vector<Share> cut_frame(const mirt_render_params& p, size_t n_devices, uint32_t row_bands) {
  const uint32_t rows = p.n_rows ? p.n_rows : p.height;
  uint32_t bands = std::max(1u, std::min(row_bands, rows));
  const uint32_t ranges = (uint32_t)n_devices / bands;
  std::vector<Share> shares(n_devices);
  for (size_t k = 0; k < n_devices; ++k) {
    const uint32_t band = (uint32_t)k / ranges, range = (uint32_t)k % ranges;
    Share& sh = shares[k];
    if (band >= bands) continue;  // (row_bands did not divide: the leftover devices idle)
    // whole 8-row groups per band where there are enough rows (the kernel walks the image in 8-row bands)
    const uint32_t groups = (rows + 7) / 8;
    const uint32_t g0 = (uint32_t)((uint64_t)groups * band / bands), g1 = (uint32_t)((uint64_t)groups * (band + 1) / bands);
    const uint32_t r0 = std::min(rows, g0 * 8), r1 = std::min(rows, g1 * 8);
    sh.first_row = p.first_row + r0;
    sh.n_rows = r1 - r0;
    const uint32_t base = p.samples_per_pixel / ranges, extra = p.samples_per_pixel % ranges;
    sh.first_sample = p.first_sample + range * base + std::min(range, extra);
    sh.n_samples = base + (range < extra ? 1u : 0u);
  }
  return shares;
}

// mirt_render on a context over several GPUs: every device renders its share of the frame into its own sum
// image (one host thread per device), the other devices' rows are copied to the first device over the GPUs'
// links and added there.  ctx->image_sum holds the whole frame afterwards.
int render_shared(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera, const mirt_render_params* params,
                  mirt_stats* stats) {
  if (!scene || !camera) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  const size_t n = 1 + ctx->others.size();
  if (scene->copies.size() != ctx->others.size())
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "the scene was not uploaded through this context");
  const mirt_render_params& p = *params;
  if (p.samples_per_pixel == 0) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be positive");
  const uint32_t rows = p.n_rows ? p.n_rows : p.height;
  if ((p.n_rows == 0 && p.first_row != 0) || (uint64_t)p.first_row + rows > p.height)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "rows [first_row, first_row + n_rows) outside the image");
  const std::vector<Share> shares = cut_frame(p, n, ctx->row_bands);
  const size_t row_values = (size_t)p.width * 3;
  const size_t n_values = row_values * p.height;
  std::vector<int> status(n, MIRT_OK);
  std::vector<mirt_stats> share_stats(n);
  std::vector<char> copied(n, 0);
  auto work = [&](size_t k) {
    mirt_context* c = k == 0 ? ctx : ctx->others[k - 1];
    const mirt_scene* sc = k == 0 ? scene : scene->copies[k - 1];
    const Share& sh = shares[k];
    std::memset(&share_stats[k], 0, sizeof(mirt_stats));
    if (sh.n_samples == 0 || sh.n_rows == 0) return;
    mirt_render_params q = p;
    q.first_sample = sh.first_sample;
    q.samples_per_pixel = sh.n_samples;
    q.first_row = sh.first_row;
    q.n_rows = sh.n_rows;
    auto hip_ok = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return true;
      status[k] = fail(c, e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_DEVICE,
                       std::string(what) + ": " + hipGetErrorString(e));
      return false;
    };
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return;
    float* sums = ctx->image_sum.as<float>();
    if (k != 0) {
      if (!hip_ok(c->image_sum.ensure(n_values * sizeof(float)), "hipMalloc")) return;
      sums = c->image_sum.as<float>();
      // only the share's rows are touched
      if (!hip_ok(hipMemsetAsync(sums + row_values * sh.first_row, 0, row_values * sh.n_rows * sizeof(float), c->stream),
                  "hipMemsetAsync")) return;
    }
    status[k] = render_impl(c, sc, camera, &q, sums, c->stream, &share_stats[k]);
    if (status[k] != MIRT_OK || k == 0) return;
    // the share's rows, to the first device (the staging buffer was sized before the threads started)
    const size_t bytes = row_values * sh.n_rows * sizeof(float);
    // (on this device's own stream, behind its render, and waited for here: the add on the first device starts only
    // after every worker thread has returned, so no ordering between streams of different devices is relied on)
    if (!hip_ok(hipMemcpyPeerAsync(ctx->staging[k - 1]->ptr, ctx->device, sums + row_values * sh.first_row, c->device, bytes,
                                   c->stream), "hipMemcpyPeerAsync")) return;
    if (!hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize after the peer copy")) return;
    copied[k] = 1;
  };
  for (size_t k = 1; k < n; ++k)
    MIRT_HIP(ctx, ctx->staging[k - 1]->ensure(std::max<size_t>(16, row_values * shares[k].n_rows * sizeof(float))));
  const bool started = side_by_side(n, work, status);
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  if (!started) return fail(ctx, MIRT_ERR_DEVICE, "cannot start a host thread per device");
  for (size_t k = 0; k < n; ++k)
    if (status[k] != MIRT_OK) {
      if (k != 0) ctx->error = "device " + std::to_string(ctx->others[k - 1]->device) + ": " + ctx->others[k - 1]->error;
      return status[k];
    }
  for (size_t k = 1; k < n; ++k)
    if (copied[k]) {
      launch_add(ctx->image_sum.as<float>() + row_values * shares[k].first_row, ctx->staging[k - 1]->as<float>(),
                 row_values * shares[k].n_rows, ctx->stream);
      MIRT_HIP(ctx, hipGetLastError());
    }
  if (stats) {
    // the whole job: work summed over the devices, times of the slowest one
    *stats = share_stats[0];
    for (size_t k = 1; k < n; ++k) {
      const mirt_stats& o = share_stats[k];
      stats->samples += o.samples; stats->segments += o.segments;
      stats->nodes_visited += o.nodes_visited; stats->prims_tested += o.prims_tested;
      stats->node_wave_steps += o.node_wave_steps; stats->prim_wave_steps += o.prim_wave_steps;
      stats->render_ms = std::max(stats->render_ms, o.render_ms);
      stats->iterations = std::max(stats->iterations, o.iterations);
      for (int j = 0; j < MIRT_KERNEL_COUNT; ++j) {
        stats->kernel_ms[j] = std::max(stats->kernel_ms[j], o.kernel_ms[j]);
        stats->kernel_launches[j] += o.kernel_launches[j];
      }
      for (int j = 0; j < MIRT_PHASE_COUNT; ++j) stats->phase_cycles[j] += o.phase_cycles[j];
    }
  }
  return MIRT_OK;
}

}  // namespace

extern "C" {

int mirt_render_accumulate(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                           const mirt_render_params* params, float* rgb_sum_device, void* stream, mirt_stats* stats) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  return render_impl(ctx, scene, camera, params, rgb_sum_device, (hipStream_t)stream /* NULL: the null stream */, stats);
} catch (...) {
  return caught(ctx);
}

int mirt_render_accumulate_static(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                                  const mirt_render_params* params, float* rgb_sum_device, void* stream,
                                  mirt_stats* stats) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (params && (params->flags & MIRT_RENDER_TIME_KERNELS))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "MIRT_RENDER_TIME_KERNELS creates events: not available without allocation");
  return render_impl(ctx, scene, camera, params, rgb_sum_device, (hipStream_t)stream /* NULL: the null stream */, stats, true);
} catch (...) {
  return caught(ctx);
}

int mirt_static_workspace_paths(void) { return (int)STATIC_POOL_PATHS; }

int mirt_resolve(mirt_context* ctx, const float* rgb_sum_device, uint32_t width, uint32_t height,
                 uint32_t samples_per_pixel, uint8_t* rgb24_device, void* stream) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!rgb_sum_device || !rgb24_device || width == 0 || height == 0 || samples_per_pixel == 0)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "bad resolve argument");
  if ((uint64_t)width * height * 3 >= (1ull << 32)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "image too large");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = (hipStream_t)stream /* NULL: the null stream */;
  launch_resolve(rgb_sum_device, width * height * 3, samples_per_pixel, rgb24_device, s);
  MIRT_HIP(ctx, hipGetLastError());
  MIRT_HIP(ctx, hipStreamSynchronize(s));
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

int mirt_render(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                const mirt_render_params* params, uint8_t* rgb24_out, float* rgb_sum_out, mirt_stats* stats) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!params) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  if ((uint64_t)params->width * params->height >= (1ull << 30)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "image too large");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n_values = (size_t)params->width * params->height * 3;
  MIRT_HIP(ctx, ctx->image_sum.ensure(n_values * sizeof(float)));
  MIRT_HIP(ctx, ctx->image_rgb.ensure(n_values));
  MIRT_HIP(ctx, hipMemsetAsync(ctx->image_sum.ptr, 0, n_values * sizeof(float), ctx->stream));
  int rc;
  if (ctx->others.empty()) rc = render_impl(ctx, scene, camera, params, ctx->image_sum.as<float>(), ctx->stream, stats);
  else rc = render_shared(ctx, scene, camera, params, stats);
  if (rc != MIRT_OK) return rc;
  if (rgb24_out) {
    launch_resolve(ctx->image_sum.as<float>(), (uint32_t)n_values, params->samples_per_pixel,
                   ctx->image_rgb.as<uint8_t>(), ctx->stream);
    MIRT_HIP(ctx, hipGetLastError());
    MIRT_HIP(ctx, hipMemcpyAsync(rgb24_out, ctx->image_rgb.ptr, n_values, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (rgb_sum_out)
    MIRT_HIP(ctx, hipMemcpyAsync(rgb_sum_out, ctx->image_sum.ptr, n_values * sizeof(float), hipMemcpyDeviceToHost,
                                 ctx->stream));
  MIRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

}  // extern "C"

struct mirt_frame {
  DeviceBuffer sums;
  uint32_t width = 0, height = 0;
  uint32_t samples = 0;
  bool started = false;  // seed, depth and flags are fixed from the first pass on
  uint64_t seed = 0;
  uint32_t max_depth = 0, flags = 0;
  size_t values() const { return (size_t)width * height * 3; }
};

extern "C" {

int mirt_frame_create(mirt_context* ctx, uint32_t width, uint32_t height, mirt_frame** out) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (width < 2 || height < 2 || (uint64_t)width * height >= (1ull << 30))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "bad frame size");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  mirt_frame* f = new (std::nothrow) mirt_frame();
  if (!f) return fail(ctx, MIRT_ERR_OUT_OF_MEMORY, "out of host memory");
  f->width = width;
  f->height = height;
  hipError_t e = f->sums.ensure(f->values() * sizeof(float));
  if (e == hipSuccess) e = hipMemset(f->sums.ptr, 0, f->values() * sizeof(float));
  if (e != hipSuccess) {
    delete f;
    return fail(ctx, e == hipErrorOutOfMemory ? MIRT_ERR_OUT_OF_MEMORY : MIRT_ERR_DEVICE, hipGetErrorString(e));
  }
  *out = f;
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

void mirt_frame_destroy(mirt_context* ctx, mirt_frame* frame) {
  if (ctx) (void)hipSetDevice(ctx->device);
  delete frame;
}

uint32_t mirt_frame_samples(const mirt_frame* frame) { return frame ? frame->samples : 0; }

int mirt_frame_render(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                      const mirt_render_params* params, mirt_frame* frame, mirt_stats* stats) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!scene || !camera || !params || !frame) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  if (params->width != frame->width || params->height != frame->height)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "the pass is not of the frame's size");
  if (params->n_rows != 0 && (params->first_row != 0 || params->n_rows != params->height))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "a frame is rendered whole: no row window");
  const uint32_t estimator = params->flags & (MIRT_RENDER_EVEN_LIGHTS | MIRT_RENDER_DROP_NONFINITE | MIRT_RENDER_STRATA_MASK);
  if (frame->started && (params->seed != frame->seed || params->max_depth != frame->max_depth || estimator != frame->flags))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "seed, max_depth or estimator flags differ from the frame's earlier passes");
  if ((uint64_t)frame->samples + params->samples_per_pixel - 1 > MAX_SAMPLE_INDEX)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "sample index beyond 2^24 - 1");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  mirt_render_params p = *params;
  p.first_sample = frame->samples;
  p.first_row = 0;
  p.n_rows = 0;
  int rc;
  if (ctx->others.empty()) {
    // straight into the frame's sums
    rc = render_impl(ctx, scene, camera, &p, frame->sums.as<float>(), ctx->stream, stats);
  } else {
    // the devices' shares are gathered in the context's own image, which is then added to the frame
    MIRT_HIP(ctx, ctx->image_sum.ensure(frame->values() * sizeof(float)));
    MIRT_HIP(ctx, hipMemsetAsync(ctx->image_sum.ptr, 0, frame->values() * sizeof(float), ctx->stream));
    rc = render_shared(ctx, scene, camera, &p, stats);
    if (rc == MIRT_OK) {
      launch_add(frame->sums.as<float>(), ctx->image_sum.as<float>(), frame->values(), ctx->stream);
      MIRT_HIP(ctx, hipGetLastError());
      MIRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  if (rc != MIRT_OK) return rc;
  frame->started = true;
  frame->seed = params->seed;
  frame->max_depth = params->max_depth;
  frame->flags = estimator;
  frame->samples += params->samples_per_pixel;
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

int mirt_frame_read(mirt_context* ctx, const mirt_frame* frame, uint8_t* rgb24_out, float* rgb_sum_out) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!frame) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = frame->values();
  if (rgb24_out) {
    if (frame->samples == 0) {
      std::memset(rgb24_out, 0, n);
    } else {
      MIRT_HIP(ctx, ctx->image_rgb.ensure(n));
      launch_resolve(frame->sums.as<float>(), (uint32_t)n, frame->samples, ctx->image_rgb.as<uint8_t>(), ctx->stream);
      MIRT_HIP(ctx, hipGetLastError());
      MIRT_HIP(ctx, hipMemcpyAsync(rgb24_out, ctx->image_rgb.ptr, n, hipMemcpyDeviceToHost, ctx->stream));
    }
  }
  if (rgb_sum_out)
    MIRT_HIP(ctx, hipMemcpyAsync(rgb_sum_out, frame->sums.ptr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  MIRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

int mirt_frame_save(mirt_context* ctx, const mirt_frame* frame, const char* path) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!frame || !path) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<float> host(frame->values());
  const int rc = mirt_frame_read(ctx, frame, nullptr, host.data());
  if (rc != MIRT_OK) return rc;
  char head[256];
  const int len = std::snprintf(head, sizeof(head),
                                "{\"width\": %u, \"height\": %u, \"samples\": %u, \"seed\": %llu, \"depth\": %u, "
                                "\"flags\": %u, \"started\": %u}",
                                frame->width, frame->height, frame->samples, (unsigned long long)frame->seed,
                                frame->max_depth, frame->flags, frame->started ? 1u : 0u);
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string("cannot write ") + path);
  const uint32_t len32 = (uint32_t)len;  // (the file is little endian, as is every host this library runs on)
  bool ok = std::fwrite("MIRTACC1", 1, 8, f) == 8 && std::fwrite(&len32, 4, 1, f) == 1 &&
            std::fwrite(head, 1, (size_t)len, f) == (size_t)len &&
            std::fwrite(host.data(), sizeof(float), host.size(), f) == host.size();
  ok = (std::fclose(f) == 0) && ok;
  return ok ? MIRT_OK : fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string("short write on ") + path);
} catch (...) {
  return caught(ctx);
}

int mirt_frame_load(mirt_context* ctx, const char* path, mirt_frame** out) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!path || !out) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string("cannot read ") + path);
  struct Closer { FILE* f; ~Closer() { std::fclose(f); } } closer{f};
  char magic[8];
  uint32_t len = 0;
  if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "MIRTACC1", 8) != 0 || std::fread(&len, 4, 1, f) != 1 || len >= 4096)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string(path) + ": not a checkpoint of this library");
  std::string head(len, '\0');
  if (std::fread(&head[0], 1, len, f) != len) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string(path) + ": truncated checkpoint");
  // the header's numbers, by key (keys this version does not know are skipped; absent ones read as 0)
  auto number = [&](const char* key, bool& found) -> unsigned long long {
    const std::string k = std::string("\"") + key + "\":";
    const size_t at = head.find(k);
    found = at != std::string::npos;
    return found ? std::strtoull(head.c_str() + at + k.size(), nullptr, 10) : 0ull;
  };
  bool has_w, has_h, has_s, has_seed, has_depth, has_flags, has_started;
  const unsigned long long w = number("width", has_w), h = number("height", has_h), n = number("samples", has_s),
                           seed = number("seed", has_seed), depth = number("depth", has_depth),
                           flags = number("flags", has_flags), started = number("started", has_started);
  if (!has_w || !has_h || !has_s || !has_seed || !has_depth || n > MAX_SAMPLE_INDEX + 1ull || w > 0xFFFFFFFFull || h > 0xFFFFFFFFull)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string(path) + ": checkpoint header incomplete");
  mirt_frame* frame = nullptr;
  const int rc = mirt_frame_create(ctx, (uint32_t)w, (uint32_t)h, &frame);
  if (rc != MIRT_OK) return rc;
  struct Owner {  // the frame is this function's until it is handed out
    mirt_context* ctx;
    mirt_frame* frame;
    ~Owner() { if (frame) mirt_frame_destroy(ctx, frame); }
  } owner{ctx, frame};
  std::vector<float> host(frame->values());
  char extra;
  if (std::fread(host.data(), sizeof(float), host.size(), f) != host.size() || std::fread(&extra, 1, 1, f) != 0)
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, std::string(path) + ": truncated checkpoint");
  const hipError_t e = hipMemcpy(frame->sums.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(ctx, MIRT_ERR_DEVICE, hipGetErrorString(e));
  frame->samples = (uint32_t)n;
  frame->seed = seed;
  frame->max_depth = (uint32_t)depth;
  frame->flags = (uint32_t)flags;
  frame->started = has_started ? started != 0 : n != 0;
  owner.frame = nullptr;
  *out = frame;
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

static int trace_device_impl(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays_device, size_t n,
                             uint64_t seed, uint32_t* type_index_device, uint32_t* object_index_device,
                             float* t_device, float4* records_device, void* stream) {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!scene || (n && (!rays_device || !type_index_device || !object_index_device || !t_device)))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  if (n >= (1ull << 32)) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "too many rays");
  if (n == 0) return MIRT_OK;
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  // the rays go through the render kernel's own walk, one round of it
  Job job;
  job.rc.key0 = (uint32_t)seed;
  job.rc.key1 = (uint32_t)(seed >> 32);
  job.rc.total_paths = n;
  job.rc.max_depth = 1;
  job.rays = rays_device;
  job.hit_type = type_index_device;
  job.hit_object = object_index_device;
  job.hit_t = t_device;
  job.hit_records = records_device;
  return run_job(ctx, scene, job, (hipStream_t)stream /* NULL: the null stream */, nullptr);
}

int mirt_trace_device(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays_device, size_t n, uint64_t seed,
                      uint32_t* type_index_device, uint32_t* object_index_device, float* t_device, void* stream) try {
  return trace_device_impl(ctx, scene, rays_device, n, seed, type_index_device, object_index_device, t_device, nullptr,
                           stream);
} catch (...) {
  return caught(ctx);
}

static int check_ray_times(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays, size_t n) {
  if (scene->has_moving)
    for (size_t i = 0; i < n; ++i)
      if (!(rays[i].time >= scene->time0 && rays[i].time <= scene->time1))
        return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "ray time outside the scene's time range");
  return MIRT_OK;
}

int mirt_trace_records(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays, size_t n, uint64_t seed,
                       mirt_hit_record* records) try {
  static_assert(sizeof(mirt_hit_record) == 64, "a hit record is four quads");
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!scene || (n && (!rays || !records))) return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  if (n == 0) return MIRT_OK;
  const int tr = check_ray_times(ctx, scene, rays, n);
  if (tr != MIRT_OK) return tr;
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  DeviceBuffer d_rays, d_ti, d_oi, d_t, d_rec;
  MIRT_HIP(ctx, d_rays.ensure(n * sizeof(mirt_ray)));
  MIRT_HIP(ctx, d_ti.ensure(n * 4));
  MIRT_HIP(ctx, d_oi.ensure(n * 4));
  MIRT_HIP(ctx, d_t.ensure(n * 4));
  MIRT_HIP(ctx, d_rec.ensure(n * sizeof(mirt_hit_record)));
  MIRT_HIP(ctx, hipMemcpy(d_rays.ptr, rays, n * sizeof(mirt_ray), hipMemcpyHostToDevice));
  const int rc = trace_device_impl(ctx, scene, d_rays.as<mirt_ray>(), n, seed, d_ti.as<uint32_t>(), d_oi.as<uint32_t>(),
                                   d_t.as<float>(), d_rec.as<float4>(), nullptr);
  if (rc != MIRT_OK) return rc;
  MIRT_HIP(ctx, hipMemcpy(records, d_rec.ptr, n * sizeof(mirt_hit_record), hipMemcpyDeviceToHost));
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

int mirt_trace_primary(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays, size_t n, uint64_t seed,
                       uint32_t* type_index, uint32_t* object_index, float* t) try {
  if (!ctx) return MIRT_ERR_INVALID_ARGUMENT;
  if (!scene || (n && (!rays || !type_index || !object_index || !t)))
    return fail(ctx, MIRT_ERR_INVALID_ARGUMENT, "null argument");
  if (n == 0) return MIRT_OK;
  const int tr = check_ray_times(ctx, scene, rays, n);
  if (tr != MIRT_OK) return tr;
  MIRT_HIP(ctx, hipSetDevice(ctx->device));
  DeviceBuffer d_rays, d_ti, d_oi, d_t;
  MIRT_HIP(ctx, d_rays.ensure(n * sizeof(mirt_ray)));
  MIRT_HIP(ctx, d_ti.ensure(n * 4));
  MIRT_HIP(ctx, d_oi.ensure(n * 4));
  MIRT_HIP(ctx, d_t.ensure(n * 4));
  MIRT_HIP(ctx, hipMemcpy(d_rays.ptr, rays, n * sizeof(mirt_ray), hipMemcpyHostToDevice));
  const int rc = mirt_trace_device(ctx, scene, d_rays.as<mirt_ray>(), n, seed, d_ti.as<uint32_t>(), d_oi.as<uint32_t>(),
                                   d_t.as<float>(), nullptr);
  if (rc != MIRT_OK) return rc;
  MIRT_HIP(ctx, hipMemcpy(type_index, d_ti.ptr, n * 4, hipMemcpyDeviceToHost));
  MIRT_HIP(ctx, hipMemcpy(object_index, d_oi.ptr, n * 4, hipMemcpyDeviceToHost));
  MIRT_HIP(ctx, hipMemcpy(t, d_t.ptr, n * 4, hipMemcpyDeviceToHost));
  return MIRT_OK;
} catch (...) {
  return caught(ctx);
}

}  // extern "C"
