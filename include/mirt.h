/*
 * mirt.h -- C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This is the boundary a host program crosses to render a scene that was built
 * with the reference's World / Camera / Renderer API.  The reference
 * (ShadowCurse/raytracing, Rust) has no FFI of its own: its public surface is
 *   Renderer::new / Renderer::render / Renderer::present   src/renderer.rs:25-138
 *   Camera::new (all fields pub)                           src/camera.rs:4-55
 *   World::add / BVH::from_world                           src/world.rs:27-40, src/bvh.rs:24-29
 * and the body of Renderer::render (src/renderer.rs:55-87, the strip threads
 * running render_tile/ray_color) is what this library replaces.  A binding
 * lowers the object tree to the flat arrays below, calls mirt_render, and
 * copies the RGB24 image into Renderer::buffer.
 *
 * Conventions: every struct is plain old data; the caller owns every host
 * pointer for the duration of a call only; the library owns all device memory
 * behind the opaque handles; every entry point returns a status (0 = ok) and
 * never throws; a message for the last failure of a context is available from
 * mirt_last_error.  Calls block until the work is done.  One context must not
 * be used from two threads at once; different contexts are independent.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mirt_context mirt_context;
typedef struct mirt_scene mirt_scene;

enum {
  MIRT_OK = 0,
  MIRT_ERR_INVALID_ARGUMENT = 1,
  MIRT_ERR_DEVICE = 2,       /* a HIP call failed, or the host side could not go on (a thread per device could not be
                              * started, an exception other than out-of-memory was caught); see mirt_last_error */
  MIRT_ERR_UNSUPPORTED = 3,  /* scene uses a construct the device path does not lower */
  MIRT_ERR_OUT_OF_MEMORY = 4 /* device memory, or host memory (std::bad_alloc caught at the entry point) */
};

/* ---- primitives (object-space parameters) -------------------------------- */

enum {
  MIRT_PRIM_SPHERE = 0,        /* objects.rs:11-101  */
  MIRT_PRIM_MOVING_SPHERE = 1, /* objects.rs:103-181 */
  MIRT_PRIM_RECT = 2,          /* objects.rs:183-365 */
  MIRT_PRIM_BOX = 3            /* objects.rs:367-441 */
};

enum {
  MIRT_PRIM_FLAG_BOUNDARY = 1u /* only the boundary of a constant medium, not a surface */
};

/* Fields every primitive carries. */
typedef struct mirt_prim_header {
  uint32_t material;     /* index into mirt_scene_desc.materials */
  uint32_t chain;        /* index into mirt_scene_desc.chains; 0 = not wrapped */
  uint32_t order;        /* tie-break rank: of two hits with equal t the higher order wins.
                            World scan (world.rs:75-94): visiting order.  BVH
                            (bvh.rs:92-145): in-order position of the leaf. */
  uint32_t type_index;   /* WorldIndex (world.rs:13-17) of the top-level object */
  uint32_t object_index; /*   this primitive belongs to */
  uint32_t flags;
} mirt_prim_header;

typedef struct mirt_sphere {
  mirt_prim_header h;
  float center[3];
  float radius;
} mirt_sphere;

typedef struct mirt_moving_sphere {
  mirt_prim_header h;
  float center0[3];
  float center1[3];
  float time0, time1;
  float radius;
} mirt_moving_sphere;

enum { MIRT_RECT_XY = 0, MIRT_RECT_XZ = 1, MIRT_RECT_YZ = 2 };

typedef struct mirt_rect {
  mirt_prim_header h;
  uint32_t axis;        /* MIRT_RECT_* */
  float a0, a1, b0, b1; /* XY: x0,x1,y0,y1   XZ: x0,x1,z0,z1   YZ: y0,y1,z0,z1 */
  float k;
} mirt_rect;

typedef struct mirt_box {
  mirt_prim_header h;
  float min[3];
  float max[3];
} mirt_box;

/* ---- wrappers: Translate / Rotate / FlipFace (transforms.rs) --------------- */

enum {
  MIRT_OP_TRANSLATE = 0, /* p = offset                 transforms.rs:3-30    */
  MIRT_OP_ROTATE_Y = 1,  /* p[0] = sin, p[1] = cos     transforms.rs:32-115  */
  MIRT_OP_FLIP_FACE = 2  /*                            transforms.rs:187-210 */
};

typedef struct mirt_chain_op {
  uint32_t kind;
  float p[3];
} mirt_chain_op;

#define MIRT_MAX_CHAIN_OPS 4

/* A chain lists the wrappers around a primitive, outermost first.
 * chains[0] must be the empty chain. */
typedef struct mirt_chain {
  uint32_t first_op;
  uint32_t n_ops; /* <= MIRT_MAX_CHAIN_OPS */
} mirt_chain;

/* ---- constant media (transforms.rs:117-185) -------------------------------- */

#define MIRT_MAX_MEDIA 8

typedef struct mirt_medium {
  uint32_t boundary_tag;   /* MIRT_PRIM_* of the boundary primitive */
  uint32_t boundary_index; /*   (flagged MIRT_PRIM_FLAG_BOUNDARY; its chain is the full chain) */
  float neg_inv_density;
  uint32_t material;       /* phase function */
  uint32_t chain;          /* wrappers around the medium itself */
  uint32_t order;
  uint32_t type_index;
  uint32_t object_index;
} mirt_medium;

/* ---- materials and textures (material.rs, texture.rs, perlin.rs) ----------- */

enum {
  MIRT_MAT_LAMBERTIAN = 0,
  MIRT_MAT_METAL = 1,
  MIRT_MAT_DIELECTRIC = 2,
  MIRT_MAT_DIFFUSE_LIGHT = 3,
  MIRT_MAT_ISOTROPIC = 4 /* no scatter in the reference (material.rs:172-180): absorbs */
};

typedef struct mirt_material {
  uint32_t kind;
  uint32_t texture; /* albedo / emit texture; unused for dielectric */
  float param;      /* metal: fuzz (already clamped to <= 1); dielectric: index of refraction */
} mirt_material;

enum { MIRT_TEX_SOLID = 0, MIRT_TEX_CHECKER = 1, MIRT_TEX_NOISE = 2, MIRT_TEX_IMAGE = 3 };

typedef struct mirt_texture {
  uint32_t kind;
  float rgb[3];   /* solid */
  uint32_t odd;   /* checker: texture indices */
  uint32_t even;
  float scale;    /* noise */
  uint32_t index; /* noise: index into perlins; image: index into images */
} mirt_texture;

typedef struct mirt_perlin {
  float random_vec[256][3];
  uint32_t perm_x[256];
  uint32_t perm_y[256];
  uint32_t perm_z[256];
} mirt_perlin;

typedef struct mirt_image {
  const uint8_t* rgb8; /* width*height*3 bytes, row 0 = top */
  uint32_t width;
  uint32_t height;
} mirt_image;

/* ---- lights: the `lights` argument of Renderer::render (renderer.rs:47) ---- */

enum {
  MIRT_LIGHT_XZ_RECT = 0, /* p = x0,x1,z0,z1,k   objects.rs:292-309 */
  MIRT_LIGHT_SPHERE = 1,  /* p = cx,cy,cz,r      objects.rs:82-100  */
  MIRT_LIGHT_OTHER = 2    /* trait defaults: pdf 0, direction (1,0,0)  hittable.rs:52-57 */
};

typedef struct mirt_light {
  uint32_t kind;
  float p[5];
} mirt_light;

/* One group per concrete type of a World of lights: World::random picks a
 * type, then an object of it (world.rs:112-117). */
typedef struct mirt_light_group {
  uint32_t first_light;
  uint32_t n_lights;
} mirt_light_group;

enum {
  MIRT_LIGHTS_NONE = 0,   /* render(.., None)                          */
  MIRT_LIGHTS_OBJECT = 1, /* a bare object: lights[0]                  */
  MIRT_LIGHTS_WORLD = 2   /* a World: groups + lights (world.rs:100-117) */
};

/* ---- the lowered scene ------------------------------------------------------ */

/* BVH builders (mirt_scene_desc.build_flags).  The reference builds on the host, single-threaded
 * (src/bvh.rs:24-81).  The closest hit does not depend on the choice, only the time to build and to find it. */
enum {
  MIRT_BUILD_AUTO = 0,         /* host below 2^18 primitives, device from there up */
  MIRT_BUILD_HOST_SAH = 1,     /* binned surface-area build on the host: best tree, seconds for millions */
  MIRT_BUILD_DEVICE_MORTON = 2 /* Morton-order build on the GPU: milliseconds, looser tree */
};

typedef struct mirt_scene_desc {
  const mirt_sphere* spheres;
  uint32_t n_spheres;
  const mirt_moving_sphere* moving_spheres;
  uint32_t n_moving_spheres;
  const mirt_rect* rects;
  uint32_t n_rects;
  const mirt_box* boxes;
  uint32_t n_boxes;

  const mirt_chain* chains;
  uint32_t n_chains;
  const mirt_chain_op* chain_ops;
  uint32_t n_chain_ops;

  const mirt_medium* media;
  uint32_t n_media; /* <= MIRT_MAX_MEDIA */

  const mirt_material* materials;
  uint32_t n_materials;
  const mirt_texture* textures;
  uint32_t n_textures;
  const mirt_perlin* perlins;
  uint32_t n_perlins;
  const mirt_image* images;
  uint32_t n_images;

  uint32_t lights_mode; /* MIRT_LIGHTS_* */
  const mirt_light* lights;
  uint32_t n_lights;
  const mirt_light_group* light_groups;
  uint32_t n_light_groups;

  /* Ray times the acceleration structure must be valid for (moving spheres
   * are bounded over this interval).  Cameras and rays outside it are
   * rejected. */
  float time0, time1;
  uint32_t build_flags; /* MIRT_BUILD_*: who builds the BVH; 0 leaves it to the library */
} mirt_scene_desc;

/* The ten fields of camera.rs:4-15. */
typedef struct mirt_camera {
  float origin[3];
  float lower_left_corner[3];
  float horizontal[3];
  float vertical[3];
  float u[3];
  float v[3];
  float w[3];
  float lens_radius;
  float time0;
  float time1;
} mirt_camera;

enum {
  MIRT_RENDER_COUNT_WORK = 1u,   /* fill mirt_stats.nodes_visited / prims_tested (slower) */
  MIRT_RENDER_TIME_KERNELS = 2u, /* fill mirt_stats.kernel_ms / kernel_launches with HIP event times */
  /* Opt-in changes to the estimator.  None of them has a counterpart in the reference: with all of them off
   * (the default) a render is the reference's estimator, sample for sample. */
  MIRT_RENDER_EVEN_LIGHTS = 4u,    /* a World of lights is sampled uniformly over its objects, the distribution its
                                      pdf_value (world.rs:100-110) already assumes; World::random (world.rs:112-117)
                                      picks a type first, which only agrees when every type has equally many */
  MIRT_RENDER_DROP_NONFINITE = 8u, /* a sample whose radiance is NaN or infinite adds nothing, instead of turning
                                      its pixel black (renderer.rs:241-252 encodes a NaN sum as 0) */
  MIRT_RENDER_STRATA_SHIFT = 8,    /* bits 8..15: n > 0 stratifies the pixel jitter of renderer.rs:216-217 over
                                      n x n cells, sample k in cell k mod n*n (whole multiples of n*n samples per
                                      pixel cover the pixel evenly) */
  MIRT_RENDER_STRATA_MASK = 0xFF00u
};
#define MIRT_RENDER_STRATA(n) (((uint32_t)(n) << MIRT_RENDER_STRATA_SHIFT) & MIRT_RENDER_STRATA_MASK)

typedef struct mirt_render_params {
  uint32_t width;             /* Renderer::new(screen_width, ..)  renderer.rs:25-41 */
  uint32_t height;
  uint32_t samples_per_pixel; /* samples this call renders for every pixel */
  uint32_t first_sample;      /* index of the first of them (sample-range sharding); 0 for a whole render */
  uint32_t max_depth;
  float background[3];
  uint64_t seed;              /* Philox key */
  uint32_t path_pool;         /* paths in flight (640 bytes of device memory each); 0 = the library sizes the pool to the
                               * call: a sixteenth of its paths, at least 8 Mi and at most 32 Mi */
  uint32_t flags;             /* MIRT_RENDER_* */
  uint32_t first_row;         /* render only rows [first_row, first_row + n_rows) of the image, row 0 = top (the */
  uint32_t n_rows;            /*   strips of renderer.rs:57-71); n_rows = 0: every row.  Buffers keep the full size. */
} mirt_render_params;

/* MIRT_KERNEL_RENDER: the one kernel that traces a whole mirt_render_accumulate call */
enum { MIRT_KERNEL_RENDER = 0, MIRT_KERNEL_RESOLVE = 1, MIRT_KERNEL_COUNT = 4 };
/* phases of the render kernel timed by MIRT_RENDER_COUNT_WORK: starting camera paths and fetching rays,
 * steps down the BVH, primitive tests, shading, handing finished rays to the material queues, whole kernel */
enum { MIRT_PHASE_LOAD = 0, MIRT_PHASE_NODES = 1, MIRT_PHASE_PRIMS = 2, MIRT_PHASE_SHADE = 3, MIRT_PHASE_WRITE = 4,
       MIRT_PHASE_TOTAL = 5, MIRT_PHASE_COUNT = 6 };

typedef struct mirt_stats {
  double render_ms;       /* wall time of the render call's device work */
  uint64_t samples;       /* camera paths traced */
  uint64_t segments;      /* path segments traced (closest-hit queries) */
  uint64_t nodes_visited; /* BVH nodes fetched   (MIRT_RENDER_COUNT_WORK) */
  uint64_t prims_tested;  /* primitive tests     (MIRT_RENDER_COUNT_WORK) */
  uint32_t iterations;    /* most start/walk/shade rounds any one region of the pool went through */
  uint32_t path_pool;
  double kernel_ms[MIRT_KERNEL_COUNT];         /* summed device time per kernel (MIRT_RENDER_TIME_KERNELS) */
  uint32_t kernel_launches[MIRT_KERNEL_COUNT];
  uint64_t phase_cycles[MIRT_PHASE_COUNT]; /* shader-clock cycles summed over waves (MIRT_RENDER_COUNT_WORK) */
  uint64_t node_wave_steps; /* wave-wide turns spent stepping down the tree: nodes_visited / (64 * this) is the
                               share of lanes doing useful work in them   (MIRT_RENDER_COUNT_WORK) */
  uint64_t prim_wave_steps; /* the same for primitive tests */
  uint32_t scene_in_lds;    /* 1: nodes and primitive records were served from the CUs' LDS copies of the scene,
                               0: from global memory (HBM / L2) */
  uint32_t lds_bytes;       /* LDS the render kernel's blocks were launched with */
} mirt_stats;

typedef struct mirt_ray {
  float origin[3];
  float direction[3];
  float time;
} mirt_ray;

/* What Hittable::hit hands to the integrator for one ray (HitRecord, hittable.rs:8-16, after every wrapper the
 * object sits in) plus what its material makes of it before any random draw. */
typedef struct mirt_hit_record {
  float point[3];
  float t;               /* +inf for a miss; every other field of a miss is zero, the indices 0xFFFFFFFF */
  float normal[3];       /* as stored in the record: facing the ray, except where a wrapper left it (transforms.rs) */
  uint32_t front_face;
  float u, v;
  uint32_t type_index;   /* WorldIndex of the top-level object (world.rs:13-17) */
  uint32_t object_index;
  float color[3];        /* Lambertian / Metal: albedo.color(u, v, point) (material.rs:40-49, 66-90); Dielectric: 1,1,1;
                            DiffuseLight: what it emits towards the ray (material.rs:151-158); Isotropic: 0 */
  uint32_t material_kind; /* MIRT_MAT_* */
} mirt_hit_record;

/* ---- entry points ----------------------------------------------------------- */

/* Creates a context on one GPU (`device_id`, a HIP device ordinal).  There is
 * no reference counterpart: the reference has no device. */
int mirt_context_create(int device_id, mirt_context** out);
void mirt_context_destroy(mirt_context* ctx);

/* A context over several GPUs of one node: what Renderer::render's strip threads are on the CPU
 * (renderer.rs:55-87).  mirt_scene_upload then keeps a copy of the scene on every device; mirt_render gives every
 * device a share of the frame -- `row_bands` horizontal bands of rows times n_devices / row_bands sample ranges
 * (row_bands must divide n_devices; 0 or 1 = sample ranges only, which balances best) -- with one host thread per
 * device, sums the partial images on device_ids[0] over the GPUs' links and encodes there.  Images do not depend on
 * the split (same Philox streams; only the order of the f32 sums differs).  The entry points that take device
 * pointers (mirt_render_accumulate, mirt_resolve, mirt_trace_*) run on device_ids[0].  n_devices = 1 is
 * mirt_context_create.  A device may be listed more than once (its shares then run side by side on it). */
int mirt_context_create_multi(const int* device_ids, int n_devices, uint32_t row_bands, mirt_context** out);
int mirt_context_device_count(const mirt_context* ctx);

/* Message of the last failed call on `ctx` (or of the last failed
 * mirt_context_create when ctx is NULL).  Valid until the next call. */
const char* mirt_last_error(const mirt_context* ctx);

/* Copies a lowered scene to the device and builds the acceleration structure
 * that replaces bvh.rs:24-81 / the linear scan of world.rs:75-94. */
int mirt_scene_upload(mirt_context* ctx, const mirt_scene_desc* desc, mirt_scene** out);
void mirt_scene_destroy(mirt_context* ctx, mirt_scene* scene);

/* Replaces the body of Renderer::render (renderer.rs:53-91).
 * rgb24_out (optional): width*height*3 bytes, row 0 = image top, encoded as
 *   renderer.rs:232-259 does (gamma 2, NaN -> 0).
 * rgb_sum_out (optional): width*height*3 f32 per-pixel radiance sums, same layout.
 * stats (optional). */
int mirt_render(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                const mirt_render_params* params, uint8_t* rgb24_out, float* rgb_sum_out, mirt_stats* stats);

/* Same render, adding into a caller-owned DEVICE buffer of width*height*3 f32
 * (zeroed by the caller) on `stream`, a hipStream_t; NULL is the null stream, which runs behind everything the
 * caller queued on the null stream and on blocking streams before the call (the same holds for mirt_resolve and
 * mirt_trace_device; work queued on a non-blocking stream has to be waited for, or that stream passed).  Used when each rank of a multi-GPU job renders a sample range and the sums
 * are then reduced across GPUs. */
int mirt_render_accumulate(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                           const mirt_render_params* params, float* rgb_sum_device, void* stream,
                           mirt_stats* stats);

/* mirt_render_accumulate for callers that may not allocate device memory (tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  It is a blocking call like the others (it copies its
 * launch arguments to the device and waits for the stream), so it cannot be recorded into a HIP graph.  The path pool, the counters
 * and the launch arguments live in a fixed workspace the library holds on every GPU (46 MiB: 65 536 paths of 640
 * bytes, a control page, and room for walks put aside); the call
 * makes no hipMalloc / hipFree and creates no events, so MIRT_RENDER_TIME_KERNELS is rejected.  The workspace holds
 * mirt_static_workspace_paths() paths in flight: params->path_pool = 0 uses all of them, a larger path_pool is
 * rejected with MIRT_ERR_INVALID_ARGUMENT (images of any size are rendered through the pool; the small pool makes
 * this entry point slower than mirt_render_accumulate).  All calls on one GPU share its workspace, whatever
 * their contexts and streams: a call made while another one is still running on the same GPU (from another thread)
 * is rejected with MIRT_ERR_INVALID_ARGUMENT, and calls from several processes on one GPU each have a workspace of
 * their own. */
int mirt_render_accumulate_static(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                                  const mirt_render_params* params, float* rgb_sum_device, void* stream,
                                  mirt_stats* stats);
int mirt_static_workspace_paths(void);

/* ---- a frame that is rendered in passes, shown while it converges, stopped and continued ---------------------
 * The reference renders a frame in one go and shows it when it is done (renderer.rs:43-138).  A mirt_frame keeps
 * the f32 sums of a frame on the device (the first device of a context over several) together with the number of
 * samples in them; every mirt_frame_render adds params->samples_per_pixel more samples per pixel, numbered on from
 * the frame's count (params->first_sample is ignored), so that any sequence of passes, with any number of saves and
 * loads in between, ends up as the image of one uninterrupted render, up to the order of the f32 sums.  Width,
 * height, seed, max_depth and the estimator flags are fixed by the first pass (or by the checkpoint) and later
 * passes that disagree are rejected. */
typedef struct mirt_frame mirt_frame;
int mirt_frame_create(mirt_context* ctx, uint32_t width, uint32_t height, mirt_frame** out);
void mirt_frame_destroy(mirt_context* ctx, mirt_frame* frame);
int mirt_frame_render(mirt_context* ctx, const mirt_scene* scene, const mirt_camera* camera,
                      const mirt_render_params* params, mirt_frame* frame, mirt_stats* stats);
/* samples per pixel in the frame so far */
uint32_t mirt_frame_samples(const mirt_frame* frame);
/* The frame as it stands, encoded as renderer.rs:232-259 for the samples it holds: what Renderer::present
 * (renderer.rs:94-138) would show.  Either output may be NULL; both are host arrays of width*height*3. */
int mirt_frame_read(mirt_context* ctx, const mirt_frame* frame, uint8_t* rgb24_out, float* rgb_sum_out);
/* Checkpoint: "MIRTACC1", u32 header length, a JSON header (width, height, samples, seed, depth, flags), then the
 * f32 sums, little endian, row 0 = image top.  mirt_frame_load makes a new frame from such a file. */
int mirt_frame_save(mirt_context* ctx, const mirt_frame* frame, const char* path);
int mirt_frame_load(mirt_context* ctx, const char* path, mirt_frame** out);

/* renderer.rs:232-259 on the device: sums of `samples_per_pixel` samples ->
 * RGB24.  Both pointers are device pointers. */
int mirt_resolve(mirt_context* ctx, const float* rgb_sum_device, uint32_t width, uint32_t height,
                 uint32_t samples_per_pixel, uint8_t* rgb24_device, void* stream);

/* Closest hit of n rays over (0.001, inf): what Hittable::hit of the scene's
 * container returns (bvh.rs:148-152 / world.rs:75-94).  Outputs are host
 * arrays of n; a miss writes 0xFFFFFFFF indices and t = +inf.  Constant media
 * draw from the Philox stream of (pixel = ray index, sample 0, bounce 0). */
int mirt_trace_primary(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays, size_t n, uint64_t seed,
                       uint32_t* type_index, uint32_t* object_index, float* t);

/* mirt_trace_primary with the whole record of every hit: what shading works with.  `records` is a host array
 * of n. */
int mirt_trace_records(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays, size_t n, uint64_t seed,
                       mirt_hit_record* records);

/* Device-resident variant used for timing the traversal alone: rays is a
 * device array of n mirt_ray, outputs are device arrays of n.  The rays' times are not checked here (the host
 * variants reject a time outside the scene's [time0, time1]): such a ray may miss a moving sphere, whose box only
 * covers that interval. */
int mirt_trace_device(mirt_context* ctx, const mirt_scene* scene, const mirt_ray* rays_device, size_t n,
                      uint64_t seed, uint32_t* type_index_device, uint32_t* object_index_device, float* t_device,
                      void* stream);

/* Library version: major*10000 + minor*100 + patch.  The structs of this header carry no size field: a caller
 * must be built against the header of the library it loads and should check mirt_version() >= MIRT_HEADER_VERSION
 * once after loading.  Changes that a caller built against an older header would not survive:
 *   0.2.0  mirt_render_params grew by first_row / n_rows, mirt_stats by scene_in_lds / lds_bytes; a NULL `stream`
 *          in mirt_render_accumulate, mirt_resolve and mirt_trace_device means the null stream (it was the
 *          context's own stream before), which does not order against the caller's non-blocking streams. */
#define MIRT_HEADER_VERSION 200
int mirt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
