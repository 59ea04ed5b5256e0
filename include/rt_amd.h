/* rt_amd.h -- C ABI of the MI355X render path (librt_amd.so).
 *
 * This is the drop-in boundary for the per-pixel render path of
 * masotrix/raytracer.  The reference has no plugin API: its path sits behind
 * the `run -s <scene>` command line + scene files on one side and the single
 * kernel launch `film<<<blocks,128>>>(img, DCamera*, DLights*, DSpheres*,
 * DOctTree*)` (main.cu:48-49, camera.cuh:52-53) on the other.  Each entry point
 * below names the reference code it replaces.
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status;
 * rt_strerror() gives the text and rt_last_error() the detail of the last
 * failure on the calling thread.  No C++ exceptions cross this boundary.
 * Handles are thread-compatible (one thread at a time per handle).
 * Image layout is the reference's: W*H*3 bytes, RGB8, row 0 = top row
 * (camera.cu:118-122).
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_ARG = -1,      /* bad argument (NULL handle, rows outside the image, ...) */
  RT_ERR_IO = -2,       /* file missing / unreadable */
  RT_ERR_PARSE = -3,    /* malformed JSON / OBJ, unknown material name */
  RT_ERR_LIMIT = -4,    /* scene exceeds a fixed limit (octree deeper than 20, ray depth > 9, ...) */
  RT_ERR_DEVICE = -5,   /* HIP call failed or no MI355X-class device */
  RT_ERR_NOMEM = -6
} rt_status;

typedef struct rt_scene rt_scene;     /* host scene: parsed, transformed, octree built and flattened */
typedef struct rt_dscene rt_dscene;   /* the same scene resident in one GPU's HBM */

/* Primitive id written per pixel: triangle = global triangle index (meshes in
 * scene order, faces in file order: mesh.cu:116-134), sphere s = -2 - s
 * (index in scene order: object.cu:16-20), miss = -1. */
#define RT_ID_MISS (-1)
#define RT_ID_SPHERE(s) (-2 - (s))

typedef struct rt_scene_info {
  int32_t width, height, max_depth;
  int32_t n_triangles, n_spheres, n_lights, n_materials;
  int32_t n_nodes, n_leaves, n_empty_leaves, n_leaf_refs, tree_depth, max_leaf_size;
  int32_t has_bounce_materials;  /* any reflective / dielectric material in use */
  int32_t shadows_enabled;
  double build_seconds;          /* octree construction + flattening */
  double parse_seconds;          /* JSON + OBJ ingest + transforms */
} rt_scene_info;

/* Work counters of one render call.  rays = calls of the closest-hit routine
 * (primary + bounce + shadow) the reference performs for these pixels, the unit
 * of the Mrays/s metric.  The four *_tests counters are only filled when
 * RT_RENDER_COUNT_TESTS is set (instrumented kernel variant, which also walks
 * every shadow ray to the end as the reference does). */
typedef struct rt_stats {
  uint64_t rays, shadow_rays;
  uint64_t node_tests, leaf_visits, tri_tests, sphere_tests;
  uint64_t ub_leave_level0, ub_level_overflow, ub_stack_overflow;
  uint64_t rays_traced;   /* traversals actually run: one shadow ray serves every material of a hit point */
  /* instrumented production kernel only, counted per WAVE: traversals, node iterations, of which exact-test,
   * of which boundary compare form, triangle iterations, reaching stage 2, stage 3, the quotient stage */
  uint64_t wave_paths[8];
  float kernel_ms;   /* HIP-event time of the render kernels of this call (0 unless RT_RENDER_TIME) */
  float copy_ms;     /* rt_render_rows_host with RT_RENDER_TIME: the copy of the rows to host memory (main.cu:56-57) */
  uint32_t from_queues;     /* 1: the bounce rays of this call went through queues, generation by generation; 2: and twice - the
                             * launches, sized by the longest generation of the context's frame before, were too short for this one */
  uint32_t queue_fallback;  /* 1: a queue or the term-slot pool was too small; the call was rendered on the stacks instead */
} rt_stats;

enum {
  RT_RENDER_COUNT_TESTS = 1,  /* run the instrumented variant that counts node / leaf / triangle / sphere tests */
  RT_RENDER_TIME = 2,         /* bracket the kernels with HIP events on `stream` and synchronise */
  RT_RENDER_LITERAL = 4,      /* cross-check kernel: one private stack per pixel, the reference's solves evaluated as written */
  /* Bounce rays are either kept on each pixel's own 10-entry stack inside one kernel launch (ray/ray.cuh:10-33), or
   * appended to queues and rendered generation by generation, 64 per wave, their contributions summed per pixel in the
   * stack's order afterwards.  Same bytes either way; the library picks (queues for glass meshes).  These force one. */
  RT_RENDER_STACKED = 8,
  RT_RENDER_QUEUED = 16,
  /* Deal the tiles of the launch to the chip's dies in strips whatever their number (the library does so by itself from
   * 4 000 tiles on).  For callers that keep several frames or bands in flight on different streams: together those fill
   * the chip, and each die's L2 then keeps meeting its own part of the scene.  A single small launch is slower this way. */
  RT_RENDER_STRIPS = 32,
  /* Test every triangle of every leaf a ray enters, as octtree.cu:264-274 does.  Without this flag a wave first rules out
   * the triangles of a leaf that none of its rays can be accepted by (same bytes: a test that rejects changes nothing);
   * the flag is there for cross-checks and measurements. */
  RT_RENDER_EVERY_TRIANGLE = 64,
  /* rt_frame_enqueue only: leave the finished frame in device memory (rt_frame_device_image) and skip the copy to the
   * host, for callers that go on working with it on the GPU (display interop, an encoder): a 3840x2160 frame is 24.9 MB,
   * and over PCIe its copy can take longer than its render. */
  RT_FRAME_ON_DEVICE = 128
};

/* Parse resources + scene, load and transform meshes, build the octree.
 * Replaces parse() (parsing/parsing.cu:145-473) and main.cu:20-32.
 * Mesh paths inside the scene are resolved against `base_dir` (NULL or "" =
 * the process working directory, which is what the reference does). */
int rt_scene_load(const char* scene_json, const char* resources_json, const char* base_dir,
                  int width, int height, int max_depth, rt_scene** out);
void rt_scene_destroy(rt_scene* s);
int rt_scene_get_info(const rt_scene* s, rt_scene_info* out);

/* ---- a scene from arrays in memory ----------------------------------------------------------------------------------
 * For a caller that already holds the scene: the reference's own parse() (parsing/parsing.cu:145-473) and host model,
 * say, whose main() hands camera, lights, objects and meshes to its *GPUAdapter classes (main.cu:18-23).  The structs
 * below are the reference's device PODs with indices in place of pointers; the caller keeps its arrays, the library
 * copies what it needs and builds the octree itself (replaces main.cu:20-32, mesh.cu:65-97,116-134, object.cu:16-20,
 * lights.cu:232-258, camera.cu:45-74's upload).  Floats are taken as they are: whoever fills the structs does the
 * reference's host arithmetic (camera basis, mesh transforms, normals), the library does none of it again. */
typedef struct rt_camera_desc {   /* DCamera (camera/camera.cuh:10-17) as Camera::buildDCamera fills it */
  float u[3], v[3], w[3], pos[3];
  float left, right, bottom, top;
  float scene_refr, bg_col[3];
  int32_t width, height, ray_depth, shadows;
} rt_camera_desc;
typedef struct rt_light_desc {    /* DLight (lights/lights.cuh:10-13); type in DLigType's order: 0 PUNCTUAL 1 SPOT 2 AMBIENT 3 DIRECTIONAL */
  float col[3], pos[3], dir[3], angle;
  int32_t type;
} rt_light_desc;
typedef struct rt_material_desc { /* DMaterial (material/material.cuh:12-15); type in DMatType's order: 0 LAMBERT 1 PHONG 2 REFLECTIVE 3 DIELECTRIC */
  float col[3], att[3], param;    /* param: shininess (PHONG) or refraction index (DIELECTRIC) */
  int32_t type;                   /* 4 (TEXTURE) is refused: the reference has no behaviour for it; texture materials come in through rt_scene_load */
} rt_material_desc;
typedef struct rt_sphere_desc {   /* DSphere (spheres/spheres.cuh:7-11); its materials are mat_index[mat_first .. mat_first + mat_count) */
  float c[3], r;
  int32_t mat_first, mat_count;
} rt_sphere_desc;
typedef struct rt_scene_desc {
  rt_camera_desc camera;
  int32_t n_lights, n_materials, n_mat_index, n_spheres, n_meshes, n_triangles;
  const rt_light_desc* lights;
  const rt_material_desc* materials;
  const int32_t* mat_index;        /* concatenated material lists of spheres and meshes: indices into materials[] */
  const rt_sphere_desc* spheres;   /* in allObjects order (object.cu:16-20): sphere s has primitive id -2 - s */
  const int32_t* mesh_mat_first;   /* per mesh: its material list in mat_index (mesh.cu:50-63) */
  const int32_t* mesh_mat_count;
  /* triangles in MeshGPUAdapter's order (meshes in scene order, faces in file order: mesh.cu:116-134); triangle i has primitive id i */
  const float* tri_verts;          /* n_triangles * 9: DTriangle::v */
  const float* tri_normals;        /* n_triangles * 9: DTriangle::n */
  const float* tri_uv;             /* n_triangles * 6: DTriangle::uv, or NULL (the path never reads them without texture materials) */
  const int32_t* tri_mesh;         /* n_triangles: index of the triangle's mesh */
} rt_scene_desc;
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
/* The description of a scene made either way, pointing into memory the rt_scene owns (valid until rt_scene_destroy):
 * rt_scene_create of it gives a scene that renders the same bytes.  Scenes with texture materials: RT_ERR_LIMIT. */
int rt_scene_describe(rt_scene* s, rt_scene_desc* out);

/* Octree in depth-first preorder (children 0..7), the order
 * OctBranch::buildDNodes walks (octtree/octtree.cu:64-87).  kind[i] = -1 for a
 * branch else the leaf's triangle count; bounds[i*6..] = ini xyz, end xyz;
 * refs = concatenated leaf contents in that order.  Buffers are sized from
 * rt_scene_info (n_nodes, n_nodes*6, n_leaf_refs). */
int rt_scene_dump_tree(const rt_scene* s, int32_t* kind, float* bounds, int32_t* refs);
/* Triangles as uploaded by Mesh::buildDTriangles (mesh/mesh.cu:65-97):
 * verts n*9, normals n*9, mesh index n. */
int rt_scene_dump_triangles(const rt_scene* s, float* verts, float* normals, int32_t* mesh);
/* DCamera fields (camera/camera.cu:45-74): u v w pos (12 floats), left right bottom top, bg rgb. */
int rt_scene_dump_camera(const rt_scene* s, float* out19);

/* Optional: bring up the HIP runtime on `device` and load the library's kernels there, so that the first rt_scene_upload
 * and the first render do not pay for it.  Thread-safe; meant to run on a helper thread beside rt_scene_load (the `run`
 * command does: runtime start-up takes about as long as parsing and building the meshDragon scene).  The reference
 * initialises CUDA implicitly with its first cudaMalloc (main.cu:20). */
int rt_device_prepare(int device);

/* Copy the flattened scene into HBM of HIP device `device`.
 * Replaces the *GPUAdapter constructors (main.cu:20-23,32; one flat upload
 * instead of one cudaMalloc per object). */
int rt_scene_upload(const rt_scene* s, int device, rt_dscene** out);
void rt_dscene_destroy(rt_dscene* d);
/* Another render context on the scene arrays `d` already holds in HBM: its own counters, queue buffers and host-path
 * buffers, nothing uploaded again.  For callers that keep several frames (or bands) in flight on different streams;
 * the scene arrays are freed when the last handle on them is destroyed, in any order.  The reference renders one
 * frame per process (main.cu:14-70) and has nothing of the kind. */
int rt_dscene_fork(const rt_dscene* d, rt_dscene** out);

/* Render image rows [row_begin,row_end) (row 0 = top).  Replaces the film
 * launch + cudaDeviceSynchronize (main.cu:38-54).
 *   d_rgb : DEVICE pointer to a full W*H*3 image; only the rows asked for are written.
 *   d_ids : optional DEVICE pointer, W*H int32 primary-hit primitive ids (NULL to skip).
 *   d_t   : optional DEVICE pointer, W*H float primary-hit distance (64 on a miss).
 *   stream: hipStream_t (NULL = default stream).  The call is asynchronous
 *           unless RT_RENDER_TIME is set, `stats` is non-NULL (stats are
 *           read back, which synchronises the stream) or the scene's bounce
 *           rays go through queues (glass meshes, RT_RENDER_QUEUED: one read-back at the end of the frame tells
 *           whether a queue overflowed; the frame is then rendered on the stacks instead).
 * An rt_dscene owns one set of counters and queue buffers: calls on the same
 * rt_dscene must not overlap (rt_dscene_fork gives each concurrent stream a handle of its own).
 * Every entry point leaves the calling thread's current HIP device as it found it. */
int rt_render_rows(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                   void* stream, unsigned flags, rt_stats* stats);

/* rt_render_rows for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same bytes, but every
 * scratch buffer of the render (work counters; the queues, term slots and sort areas of scenes rendered from queues)
 * lives in a 64 MiB array inside the library's code object instead of memory the rt_dscene allocates and grows: the
 * call contains no hipMalloc / hipFree and nothing that allocates, and read-backs land in page-locked words set up
 * by rt_scene_upload.  Limits: RT_RENDER_TIME is refused (RT_ERR_ARG; it creates events); a scene rendered from queues
 * (glass meshes, or RT_RENDER_QUEUED) needs 8 MiB of sort bins plus, at depth 3, about 516 bytes of scratch per pixel
 * of the call (more at greater depths: one 16-byte term slot per node of a pixel's ray tree), so calls beyond about
 * 113 000 pixels of such a scene are refused with RT_ERR_LIMIT - render fewer rows per call or pass
 * RT_RENDER_STACKED, which needs the counters only; every other scene is accepted at any size.  Each GPU has ONE such array, shared by all
 * rt_dscene handles of the process: calls that overlap on one GPU race on it (the counters and queue fills are cleared
 * by each call that uses them, so calls in sequence are independent).  Without `stats`, and for scenes not rendered
 * from queues, the call only enqueues kernels and memsets on `stream`.
 * What the library does ahead of the call so that nothing is left to set up inside it: rt_scene_upload (and
 * rt_device_prepare) load the code objects of every kernel file on the device and take the array's address.  What it
 * cannot do ahead: the HIP runtime sizes the private (spill) memory of a stream's hardware queue when a kernel that needs
 * some is first launched on that stream; a caller for whom even that matters launches one throw-away render on the
 * stream first. */
int rt_render_rows_static(rt_dscene* d, int row_begin, int row_end, uint8_t* d_rgb, int32_t* d_ids, float* d_t,
                          void* stream, unsigned flags, rt_stats* stats);

/* Row-interleaved partition for rendering one image on several GPUs.  The image is cut into bands of
 * `band_rows` rows (a multiple of 16); rank r of nranks owns bands r, r+nranks, ...  rt_band_count
 * returns how many bands that is.  rt_render_bands renders them PACKED: band k of this rank starts at row
 * k*band_rows of d_rgb_local (W*band_rows*rt_band_count*3 bytes), ready for one gather of equal-sized
 * tiles; rows past the bottom of the image are left untouched.  The reference has no multi-GPU path
 * (it is listed as future work, doc/GPU Raytracing.pdf p.26); pixels are independent (camera.cu:91-94),
 * which is what this partition relies on. */
int rt_band_count(const rt_dscene* d, int band_rows, int rank, int nranks);
int rt_render_bands(rt_dscene* d, int band_rows, int rank, int nranks, uint8_t* d_rgb_local, int32_t* d_ids_local,
                    float* d_t_local, void* stream, unsigned flags, rt_stats* stats);

/* ---- several GPUs: the one exchange step (RCCL over xGMI) ----------------------------------------------------------
 * The reference has no multi-GPU path; its single device->host step is the cudaMemcpy of main.cu:56-57.  Here rank r of
 * n renders its bands packed (rt_render_bands) and ONE gather of the equal-sized tiles puts them on rank 0, where
 * rt_unpack_bands restores image order.  RCCL is loaded on first use (librccl.so.1), never by single-GPU callers.
 * An rt_comm is one rank of a communicator: either one process per GPU (rank 0 makes an id with rt_comm_unique_id,
 * hands its 128 bytes to the others by any means, everybody calls rt_comm_init_rank), or one process driving n devices
 * (rt_comm_init_all fills out[0..n-1]; calls for several ranks from one thread go between rt_comm_group_begin / _end). */
typedef struct rt_comm rt_comm;
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void* id128);
int rt_comm_init_rank(const void* id128, int nranks, int rank, int device, rt_comm** out);
int rt_comm_init_all(int ndev, const int* devices, rt_comm** out);
void rt_comm_destroy(rt_comm* c);
int rt_comm_group_begin(void);
int rt_comm_group_end(void);
/* d_local: this rank's tile (bytes_per_rank bytes, the same on every rank: rank 0 owns the most bands);
 * d_gathered: on rank 0, nranks * bytes_per_rank bytes, tile k at k * bytes_per_rank (ignored elsewhere).
 * Asynchronous on `stream`, ordered behind the render that was enqueued on it. */
int rt_gather_bands(rt_comm* c, const uint8_t* d_local, size_t bytes_per_rank, uint8_t* d_gathered, void* stream);
/* Rank 0: the gathered tiles back into image order (width * height * bytes_per_pixel bytes; 3 for RGB, 4 for ids or t). */
int rt_unpack_bands(const uint8_t* d_gathered, size_t bytes_per_rank, int nranks, int band_rows, int width, int height,
                    int bytes_per_pixel, uint8_t* d_image, void* stream);
/* One frame on the n devices of one process (`run --gpus n`): scenes[r] / comms[r] = rank r, the same scene uploaded to
 * each device.  Renders, gathers, unpacks on scenes[0]'s device and copies the frame to h_rgb.  stats (optional): counters
 * summed over the devices; kernel_ms = first launch to last tile in place on the first device, copy_ms = the copy out. */
int rt_render_frame_gathered(rt_dscene* const* scenes, rt_comm* const* comms, int n, int band_rows, uint8_t* h_rgb,
                             unsigned flags, rt_stats* stats);
/* Frames in flight: a pipeline set up once - per slot (a frame in flight) and device a render context on the uploaded
 * scene, a stream and a band tile; on scenes[0]'s device the gathered tiles, the frame and a page-locked host copy of it -
 * so that rendering a frame creates and allocates nothing: rt_frame_enqueue renders the bands of every device, gathers,
 * unpacks and starts the copy to the host, all asynchronously (scenes whose bounce rays go through queues read one word
 * back per device and frame, see rt_render_rows), rt_frame_wait returns when the slot's frame is in host memory.  With
 * 2-3 slots the next frames fill the tail of long waves a short frame ends on (what one GPU of eight renders is short):
 * pass RT_RENDER_STRIPS then.  comms == NULL with n == 1: one GPU, no exchange, RCCL is not loaded.  The reference renders one
 * frame per process and copies it back synchronously (main.cu:44-57); this replaces that for callers with more to render.
 * One thread at a time per rt_frame_ctx; the scenes and communicators must outlive it. */
typedef struct rt_frame_ctx rt_frame_ctx;
int rt_frame_ctx_create(rt_dscene* const* scenes, rt_comm* const* comms, int n, int band_rows, int slots, rt_frame_ctx** out);
void rt_frame_ctx_destroy(rt_frame_ctx* f);
int rt_frame_enqueue(rt_frame_ctx* f, int slot, unsigned flags);
/* *h_rgb: the slot's page-locked frame (W*H*3 bytes, valid until the slot is enqueued again); *frame_ms: first launch to
 * the frame in host memory, from events on the first device's stream.  Either may be NULL. */
int rt_frame_wait(rt_frame_ctx* f, int slot, const uint8_t** h_rgb, float* frame_ms);
/* The slot's frame in the memory of scenes[0]'s device (W*H*3 bytes; complete after rt_frame_wait, valid until the slot is
 * enqueued again).  With RT_FRAME_ON_DEVICE it is the only copy, and rt_frame_wait's *h_rgb is NULL. */
int rt_frame_device_image(rt_frame_ctx* f, int slot, uint8_t** d_rgb);

/* Decode a texture image (8-bit non-interlaced PNG, baseline JPEG) to RGB8, row 0 on top: the loader behind the
 * `textures` of resources.json.  The reference's own loader is an unfinished stub (material/material.cu:276-289).
 * rgb == NULL: only width and height are returned; else `capacity` must be at least width * height * 3. */
int rt_image_load(const char* path, int* width, int* height, uint8_t* rgb, size_t capacity);
/* Size, counts and device of an uploaded scene (device may be NULL). */
int rt_dscene_get_info(const rt_dscene* d, rt_scene_info* out, int* device);

/* Same as rt_render_rows, into HOST memory (adds the device->host copy of main.cu:56-57).  The device frame buffers
 * and a pinned staging buffer are kept by the rt_dscene from call to call.  Destinations in pinned memory
 * (rt_host_alloc) are written by the copy engine directly; pageable ones are filled in slices through the staging
 * buffer.  Synchronous. */
int rt_render_rows_host(rt_dscene* d, int row_begin, int row_end, uint8_t* h_rgb, int32_t* h_ids, float* h_t,
                        unsigned flags, rt_stats* stats);
/* Page-locked host memory for the outputs of rt_render_rows_host (replaces the malloc of main.cu:41-42). */
void* rt_host_alloc(size_t bytes);
void rt_host_free(void* p);

/* 8-bit RGB PNG writer.  Replaces write_image (display/display.cpp:7-16). */
int rt_write_png(const char* path, int width, int height, const uint8_t* rgb);

const char* rt_strerror(int status);
const char* rt_last_error(void);
/* Library / kernel variant description, e.g. "rt_amd 0.1 gfx950". */
const char* rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
