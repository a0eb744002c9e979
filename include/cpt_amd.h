/* C ABI of libcpt_amd.so: the MI355X (gfx950) implementation of the path tracer's data-parallel
 * core -- LBVH build, BVH traversal / ray-triangle intersection, path-tracing bounce loop.
 *
 * The reference (PepcyCh/cuda-path-tracer) has no FFI of its own; each entry point below names
 * the C++ entry point of the reference it stands for.  Conventions:
 *   - every function returns 0 on success and a negative code on failure (or a pointer / count
 *     where stated); cpt_last_error() returns a message for the calling thread's last failure;
 *   - pointers named d_* are DEVICE pointers on the current device, all others are host pointers;
 *   - a ray is 8 floats: origin.xyz, tmin, direction.xyz, tmax       (src/kernels/basic/ray.cuh:7-20)
 *   - a box is 8 floats: pmin.xyzw, pmax.xyzw                        (src/kernels/accel/accel.cuh:15-18)
 *   - a node is 2 uint32: lc_or_id, rc; rc == 0xffffffff marks a leaf (src/kernels/accel/accel.cuh:20-23)
 *   - matrices are 16 floats, column major (glm::mat4)
 *   - a material is 16 floats: emission[3] diffuse[3] specular[3] transmittance[3] shininess ior
 *     opacity bsdf_type (0 lambert, 1 phong, 2 blinn-phong, 3 microfacet)  (src/scene/material.hpp:25-32)
 *   - images are W*H*4 floats (RGBA), row 0 = top of the image        (src/kernels/integrator/path.cu:90)
 * Handles are not thread safe; use one tracer per host thread / device.
 */
#ifndef CPT_AMD_H
#define CPT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPT_OK 0
#define CPT_ERR_INVALID (-1)   /* bad argument or state */
#define CPT_ERR_DEVICE (-2)    /* HIP runtime failure (no device, out of memory, launch failure) */
#define CPT_ERR_IO (-3)        /* file missing or malformed */

typedef struct cpt_scene cpt_scene;    /* host scene: objects with mesh + material, one camera   */
typedef struct cpt_tracer cpt_tracer;  /* PathTracer + Film bound to one scene on one device     */

const char *cpt_last_error(void);
int cpt_device_count(void);            /* number of HIP devices, 0 when there is none */
int cpt_set_device(int device);

/* ---- device memory helpers, so that a C caller needs no HIP headers -------------------------- */
void *cpt_malloc(size_t bytes);        /* NULL on failure */
int cpt_free(void *d_ptr);
int cpt_memcpy_h2d(void *d_dst, const void *src, size_t bytes);
int cpt_memcpy_d2h(void *dst, const void *d_src, size_t bytes);
int cpt_memset(void *d_dst, int value, size_t bytes);
int cpt_synchronize(void);
int cpt_memory_info(size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo of the current device */

/* ---- (1) LBVH builder ------------------------------------------------------------------------
 * Replaces: void kernel::BuildAccel(AccelNode *nodes, Bbox *bboxes, Bbox merged_bbox,
 *                                   uint32_t num_primitives)   src/kernels/accel/accel_build.cuh:7
 * d_nodes / d_bboxes hold 2n-1 entries; the n leaf boxes are pre-filled at [n-1, 2n-1) and are
 * reordered in place; all nodes and the n-1 internal boxes are written.  Scratch memory is
 * allocated and freed inside the call, which returns after the build has completed. */
int cpt_build_accel(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives);

/* Same build with the intermediate arrays exposed for parity checks and profiling (all optional):
 *   d_keys_out    n sorted 64-bit keys, (morton30 << 32) | primitive index
 *   d_parents_out 2n-1 parent indices (entry 0 is never written, as in the reference)
 *   stage_ms      host array of 5 milliseconds: morton, sort, hierarchy pass one (leaf gather, topology and boxes
 *                 inside each workgroup's leaves), hierarchy pass two (nodes spanning workgroups), 0 */
int cpt_build_accel_ex(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives, uint64_t *d_keys_out,
                       uint32_t *d_parents_out, float *stage_ms);

/* cpt_build_accel for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
 * must not disturb the allocator's state).  It enqueues on the null stream and waits for the build, so it is not for
 * use under stream capture.  No counterpart in the reference, whose BuildAccel allocates and frees two temporaries per
 * call (src/kernels/accel/accel_build.cu:187-199).  Same arguments, same in-place result, returns after the build has
 * completed; all scratch (keys, the copy of the leaf boxes, exchange words, digit counts) lives in one device array of
 * 60 MiB inside the library, which every GPU the library runs on holds until the process ends.  Calls no hipMalloc /
 * hipFree.  At most cpt_build_accel_noalloc_max_primitives() = 1,048,576 primitives: CPT_ERR_INVALID beyond.  All
 * calls on one GPU share that array, so two calls that overlap on the same GPU race; calls on different GPUs do not. */
int cpt_build_accel_noalloc(void *d_nodes, void *d_bboxes, const float merged_bbox[8], uint32_t num_primitives);
uint32_t cpt_build_accel_noalloc_max_primitives(void);

/* Reusable-scratch variant for timing loops: the scratch lives in the handle, no allocation
 * happens after the first call of a given size, and the call only enqueues work (synchronise with
 * cpt_synchronize).  elapsed_ms (optional) receives the device time of this build and forces a sync. */
typedef struct cpt_builder cpt_builder;
cpt_builder *cpt_builder_create(void);
void cpt_builder_destroy(cpt_builder *b);
/* Radix sort used by this builder: 0 = per pass a tile histogram, a scan and a scatter kernel (default);
 * 1 = onesweep: digit histograms computed with the keys, one kernel per pass with decoupled look-back.
 * Same result; on MI355X variant 0 is the faster one (look-back steps cross the per-XCD L2s). */
int cpt_builder_set_sort(cpt_builder *b, int variant);
/* How many nodes per workgroup the first hierarchy pass may hand to the second one (default and maximum 64);
 * the others finish their climb inside the first pass.  Same result for every value; smaller values are for
 * testing that path. */
int cpt_builder_set_carrier_limit(cpt_builder *b, uint32_t limit);
int cpt_builder_build(cpt_builder *b, void *d_nodes, void *d_bboxes, const void *d_leaf_boxes, const float merged_bbox[8],
                      uint32_t num_primitives, float *elapsed_ms, float *stage_ms);

/* cpt_builder_build and, in the same kernels, the traversal records of the tree (no counterpart in the reference, whose
 * traversal reads AccelNode and Bbox, accel.cuh:68-90): num_primitives - 1 binary records of 64 bytes (left child's box
 * min.xyz max.xyz, right child's, left reference, right reference, 8 spare bytes) and / or as many two-level records of
 * 128 bytes (two halves in that layout, four slots: a cut through the subtree below the node -- its two children, of which
 * twice the internal one with the largest box surface, fp32 dx*dy + dy*dz + dz*dx, the earlier one on equality, is replaced
 * by its own two children -- in the order the reference's traversal reaches them, right child first at every level; unused
 * slots come last with reference 0xffffffff and a zero box; the first half's spare bytes repeat the second half's two
 * references, so that one 16-byte word holds all four) -- either pointer may
 * be null, not both -- and a root record of 40 bytes (box min.xyz max.xyz, reference, primitive count, flags, spare).
 * A reference with bit 31 set is a leaf: its sorted position, or with top_level != 0 the primitive id.  Only enqueues
 * work. */
int cpt_builder_build_records(cpt_builder *b, void *d_nodes, void *d_bboxes, const void *d_leaf_boxes, const float merged_bbox[8], uint32_t num_primitives,
                              int top_level, void *d_binary_records, void *d_two_level_records, void *d_root_record);

/* ---- scene description -----------------------------------------------------------------------
 * Replaces: Scene, SceneObject, Mesh, Material, CameraComponent   src/scene/core.hpp:116-240,
 *           src/scene/mesh.hpp, src/scene/material.hpp, src/scene/camera.hpp
 *           LoadObjScene / LoadJsonScene / LoadXmlScene            src/scene/loader.hpp:7-11 */
cpt_scene *cpt_scene_create(void);
void cpt_scene_destroy(cpt_scene *s);
int cpt_scene_load_obj(cpt_scene *s, const char *path, int bsdf_type);
/* The same scene through <path>.cptcache, a binary copy of the parsed .obj that is read when it was written
 * from the .obj and material libraries as they are now (sizes, modification times) and for this bsdf_type,
 * and (re)written otherwise; material maps always come from their image files.  *from_cache (may be NULL)
 * receives 1 or 0.  No counterpart in the reference (src/scene/loader.cpp:144-260 parses the text every run). */
int cpt_scene_load_obj_cached(cpt_scene *s, const char *path, int bsdf_type, int *from_cache);
int cpt_scene_load_json(cpt_scene *s, const char *path);
int cpt_scene_load_xml(cpt_scene *s, const char *path);
/* adds one object (mesh + material); transform16 may be NULL (identity) */
int cpt_scene_add_object(cpt_scene *s, const char *name, const float *positions, const float *normals, const float *texcoords,
                         uint32_t num_vertices, const uint32_t *indices, uint32_t num_triangles, const float material16[16],
                         const float *transform16);
int cpt_scene_set_camera(cpt_scene *s, const float pos[3], const float look_at[3], const float up[3], float fov_degrees,
                         uint32_t film_width, uint32_t film_height);
/* SceneObject::translate / rotate (degrees, applied as Rz * Ry * Rx) / scale of mesh object `index`
 * (src/scene/core.hpp:232-234); the transform is T * Rz * Ry * Rx * S (src/scene/core.cpp:109-117).
 * Takes effect at the next cpt_tracer_build_buffers. */
int cpt_scene_object_set_trs(cpt_scene *s, uint32_t index, const float translate[3], const float rotate_degrees[3], const float scale[3]);
/* Edits the material of mesh object `index` (shared by every object that uses it) the way the reference's material
 * panel does (src/scene/material.cpp:60-100: new values, then SetChanged).  A tracer built on the scene notices the
 * edit at its next update, re-uploads the material's parameter block and restarts its accumulation; a moved camera
 * (cpt_scene_set_camera, src/scene/camera.cpp:11-25) is picked up the same way.  Which objects are lights is decided
 * by cpt_tracer_build_buffers, not by an edit. */
int cpt_scene_object_set_material(cpt_scene *s, uint32_t index, const float material16[16]);
/* the object's transform and the inverse the path tracer stores with it (src/pathtracer/pathtracer.cpp:110-114),
 * column major; either pointer may be NULL */
int cpt_scene_object_transforms(const cpt_scene *s, uint32_t index, float transform16[16], float inverse16[16]);
/* introspection (used by the parity tests to feed the same arrays to the CPU oracle) */
uint32_t cpt_scene_num_objects(const cpt_scene *s);                 /* objects that carry a mesh */
int cpt_scene_object_info(const cpt_scene *s, uint32_t index, char *name_buf, size_t name_buf_size, uint32_t *num_vertices,
                          uint32_t *num_triangles, float material16[16], float transform16[16]);
int cpt_scene_object_copy(const cpt_scene *s, uint32_t index, float *positions, float *normals, float *texcoords, uint32_t *indices);
/* cam10: pos[3] look_at[3] up[3] fov; res2: film width, height.  CPT_ERR_INVALID without a camera */
int cpt_scene_camera(const cpt_scene *s, float cam10[10], uint32_t res2[2]);

/* ---- (2)+(3) traversal and bounce loop -------------------------------------------------------
 * Replaces: PathTracer(Scene&, Film&), BuildBuffers, Update, ResetAccumelation, SetMaxDepth
 *           src/pathtracer/pathtracer.hpp:7-50, and Film  src/pathtracer/film.hpp:6-30,
 *           which drive kernel::PathTracer::Render          src/kernels/integrator/path.cuh:22 */
cpt_tracer *cpt_tracer_create(cpt_scene *s, uint32_t film_width, uint32_t film_height);
void cpt_tracer_destroy(cpt_tracer *t);
int cpt_tracer_set_max_depth(cpt_tracer *t, int max_depth);         /* -1 = unbounded, Russian roulette only */
int cpt_tracer_set_channel(cpt_tracer *t, int channel);             /* 0 colour, 1 shading normal */
/* Bounce-loop kernel: 0 = persistent megakernel whose lanes pick up a new pixel as soon as their
 * path ends (default); 1 = one lane per pixel for the whole launch, the structure of the
 * reference's RenderKernel (src/kernels/integrator/path.cu:85-105); 2 = wavefront: path state and
 * ray queues in device memory, extend / shade / shadow kernels per bounce, traversal kernels that
 * refill idle lanes from the queue; 4 = persistent megakernel that keeps a lane's traversal state
 * across the shading step, so a wave shades the lanes whose ray is finished as soon as they are
 * five eighths of its live lanes and shadow rays share the loop with extension rays; 5 = persistent
 * megakernel that shades whole waves like 0 but lists the shadow rays per wave, so lanes whose own
 * extension ray is finished walk them instead of idling; 3 = 5 or 4 by scene size (4 from 65,536
 * triangles on; each is the fastest of all on MI355X on its side).  Same image in all of them. */
int cpt_tracer_set_integrator(cpt_tracer *t, int mode);
/* Traversal: 0 = exact (default): the visit order and arithmetic of AccelTop/AccelBottom::Intersect and BboxIntersect
 * (src/kernels/accel/accel.cuh:33-52, 68-90, 124-150: right child first, six correctly rounded quotients per box), so
 * first-hit ids, distances and images are the reference's bit for bit.  1 = fast: of two children that are hit the
 * nearer one is walked first and the other waits with its entry distance; slab distances by multiplication with the
 * reciprocal direction, the exit distance widened by 4 ulp.  It departs from accel.cuh:33-52 (rounding of the slab
 * distances) and accel.cuh:84-85 (visit order); the triangle test is unchanged, so a reported hit has the same
 * distance and barycentrics.  What may differ is which of several triangles at the same or nearly the same distance is
 * reported (the nearest; at exactly equal distance the one later in instance and sorted-leaf order).  Applies to
 * cpt_tracer_first_hit / cpt_tracer_occlude and to integrators 3, 4 and 5; integrators 0, 1 and 2 always run exact. */
int cpt_tracer_set_traversal(cpt_tracer *t, int mode);
/* Node records of the traversal (no counterpart in the reference, which walks one AccelNode and two Bbox per step,
 * accel.cuh:68-90).  Binary records: 64 bytes per internal node, both children's boxes and references.  Two-level
 * records: 128 bytes per internal node, the boxes and references of its (up to four) grandchildren in the order
 * accel.cuh:84-91 reaches them.  On those a ray fetches half as many dependent records in half as many traversal steps;
 * because a child's box lies inside its parent's and every operation of BboxIntersect is monotonic in the box
 * coordinates, testing the grandchildren directly walks exactly the leaves the reference walks, in its order: results
 * stay bit-exact.  mode 1 (default): the build writes two-level records, every kernel walks them in exact traversal
 * and the bounce loops in ordered traversal too; the ordered batch queries walk binary records, which are then made
 * when the first such query arrives.  mode 0: binary records only, walked by everything.  Takes effect at the next
 * cpt_tracer_build_buffers. */
int cpt_tracer_set_node_records(cpt_tracer *t, int mode);
/* Wavefront integrator only: after `bounces` queue bounces the paths still alive are finished by
 * one persistent launch that keeps each path in registers (the long, thin tail of Russian-roulette
 * paths costs a kernel sequence per bounce otherwise).  Negative = never, queues to the end
 * (with unbounded depth the host then reads a live marker back every few bounces).  Default 3. */
int cpt_tracer_set_wavefront_tail(cpt_tracer *t, int bounces);
int cpt_tracer_set_profile_build(cpt_tracer *t, int on);
int cpt_tracer_build_buffers(cpt_tracer *t);                        /* PathTracer::BuildBuffers */
int cpt_tracer_update(cpt_tracer *t);                               /* PathTracer::Update: one more sample per pixel (asynchronous) */
/* `count` more samples per pixel: the film ends up bit for bit as after `count` calls of cpt_tracer_update, but the
 * persistent kernels of integrator modes 4 and 5 render up to cpt_tracer_set_samples_per_launch (1 to 4096, default 32, 1 =
 * one launch per sample) consecutive samples in one launch, a pixel's samples still arriving in sample order.
 * The reference launches RenderKernel once per sample (src/pathtracer/pathtracer.cpp:26-61); at 1024 x 1024 the
 * start of a launch and the tail of its longest paths cost about a third of the launch. */
int cpt_tracer_update_many(cpt_tracer *t, uint32_t count);
int cpt_tracer_set_samples_per_launch(cpt_tracer *t, uint32_t samples);
/* `count` Update calls bracketed by HIP events on the tracer's own stream; elapsed_ms receives the
 * device time of the whole batch.  Synchronises. */
int cpt_tracer_update_timed(cpt_tracer *t, uint32_t count, float *elapsed_ms);
/* HIP-event stopwatch on the tracer's stream: begin records an event, end records another, waits
 * for it and returns the device milliseconds between the two. */
int cpt_tracer_timer_begin(cpt_tracer *t);
int cpt_tracer_timer_end(cpt_tracer *t, float *elapsed_ms);
int cpt_tracer_reset_accumulation(cpt_tracer *t);                   /* PathTracer::ResetAccumelation + cleared film */
int cpt_tracer_synchronize(cpt_tracer *t);
uint32_t cpt_tracer_current_spp(const cpt_tracer *t);
uint32_t cpt_tracer_num_lights(const cpt_tracer *t);
/* Sharded rendering: sample indices [spp_begin, spp_begin + spp_count), 1-based like the
 * reference's spp counter, each an independent random stream (seeded by pixel and sample index,
 * src/kernels/integrator/path.cu:92).  sum_mode = 1 accumulates plain sums (alpha counts samples)
 * so that buffers of different devices can be added; sum_mode = 0 keeps the reference's running
 * mean.  d_output (W*H*4 floats) may be NULL to use the tracer's film.  Asynchronous.
 * Launches of several samples keep a pixel's samples in sample order.  Into the tracer's own film, while the tracer
 * knows what the film holds (from cpt_tracer_reset_accumulation on: nothing but its own sums, or nothing but its own
 * running means), the pixel's alpha carries the turn; after cpt_tracer_film_device_ptr has handed the memory out, or
 * into any other buffer, a word per pixel kept by the tracer does.  Same film bit for bit either way. */
int cpt_tracer_render_samples(cpt_tracer *t, uint32_t spp_begin, uint32_t spp_count, int sum_mode, float *d_output, uint32_t width,
                              uint32_t height);
/* Replaces: static void kernel::PathTracer::Render(const Params &)   src/kernels/integrator/path.cuh:8-22, path.cu:109-115,
 * at its own level: a parameter block by value-like struct, exactly ONE sample per pixel into `output` (running mean with the
 * sample number `spp`, as RenderKernel accumulates, path.cu:99-104), and the call returns when the sample is in the buffer
 * (the reference synchronises the device and asserts).  `scene` stands for Params::scene: the device-side camera, light table,
 * instances and top-level BVH that cpt_tracer_build_buffers made (the reference fills Params::scene from the same
 * BuildBuffers, src/pathtracer/pathtracer.cpp:37-50).  The tracer's own settings, film and sample counter are not touched. */
typedef struct cpt_render_params {
    cpt_tracer *scene;
    float *output;               /* device memory, screen_width * screen_height * 4 floats, row 0 = top of the image */
    uint32_t screen_width;
    uint32_t screen_height;
    uint32_t spp;                /* 1-based number of this sample */
    uint32_t max_depth;          /* 0xffffffff = unbounded, what PathTracer::Update passes for -1 (pathtracer.cpp:56) */
    int channel;                 /* 0 = eColor, 1 = eNormal */
} cpt_render_params;
int cpt_path_tracer_render(const cpt_render_params *params);

float *cpt_tracer_film_device_ptr(cpt_tracer *t);                   /* Film::CudaMap */
int cpt_tracer_film_read(cpt_tracer *t, float *rgba);               /* synchronises */
int cpt_tracer_film_save(cpt_tracer *t, const char *path);          /* Film::SaveTo: fp32 RGBA OpenEXR */

/* Batch ray queries against the built scene (AccelTop::Intersect / AccelTop::Occlude,
 * src/kernels/accel/accel.cuh:124-173).  Outputs: instance id and ORIGINAL primitive id
 * (0xffffffff on a miss), hit distance (input tmax on a miss), barycentrics (v, w).  d_t and d_uv
 * may be NULL.  Asynchronous on the tracer's stream. */
int cpt_tracer_first_hit(cpt_tracer *t, const float *d_rays, uint64_t num_rays, uint32_t *d_instance, uint32_t *d_primitive, float *d_t,
                         float *d_uv);
int cpt_tracer_occlude(cpt_tracer *t, const float *d_rays, uint64_t num_rays, uint8_t *d_occluded);
/* Kernel of cpt_tracer_first_hit (no counterpart in the reference, which has no batch query): 0 (default) = one ray per lane
 * for the whole launch; 1 = batches of 65,536 rays and more, in exact traversal, run a persistent kernel whose lanes take the
 * next ray of the batch as soon as theirs is finished -- a quarter to a third faster for rays that diverge (secondary rays),
 * slower for coherent ones (camera rays).  Same results. */
int cpt_tracer_set_query_kernel(cpt_tracer *t, int mode);
/* The bounce loop for small scenes (cpt_tracer_set_integrator 5, and 3 below 65,536 triangles) reads node and triangle records
 * from a copy that every workgroup keeps in LDS when the scene's records are 6 KB or less (no counterpart in the reference):
 * on != 0 (default) uses that copy, 0 reads memory like every other kernel.  Same film bit for bit.  Returns 1 when the
 * built scene has such a copy and it is in use, 0 otherwise, negative on error. */
int cpt_tracer_set_scene_in_lds(cpt_tracer *t, int on);
/* camera rays of sample index spp for a width*height image, in pixel-index order */
int cpt_tracer_camera_rays(cpt_tracer *t, uint32_t width, uint32_t height, uint32_t spp, float *d_rays);

/* Finished build arrays in the reference layout, copied to the host: mesh < 0 selects the
 * top-level structure, otherwise the bottom-level structure of object `mesh`.  nodes: 2*(2n-1)
 * uint32, bboxes: 8*(2n-1) floats, keys: n uint64; any of them may be NULL.  Returns n. */
int64_t cpt_tracer_get_accel(cpt_tracer *t, int mesh, uint32_t *nodes, float *bboxes, uint64_t *keys);
/* counters since the last reset: rays (closest + any hit queries), 64-byte node records
 * fetched, triangle records fetched, camera paths started, shading surfaces evaluated */
int cpt_tracer_counters(cpt_tracer *t, uint64_t out5[5]);
/* Traversal stack entries dropped since the last reset.  A lane keeps up to 56-64 pending subtrees (8-16 in LDS,
 * 48 in private memory; the reference keeps 32 and has undefined behaviour beyond, src/kernels/accel/accel.cuh:69);
 * a subtree that does not fit is skipped and counted here.  0 means every traversal so far was complete. */
int cpt_tracer_stack_overflows(cpt_tracer *t, uint64_t *count);
int cpt_tracer_reset_counters(cpt_tracer *t);
int cpt_tracer_build_stage_ms(cpt_tracer *t, float out5[5]);
/* With cpt_tracer_set_profile_build: device milliseconds of the kernels that frame the largest mesh's build before a ray can
 * be traced: out2[0] the per-triangle leaf boxes (src/scene/mesh.cpp:88-98 computes them on the host), out2[1] the triangle
 * records (the node records are written by the build's own hierarchy kernels and are part of its stage times).  Build +
 * these two = time to a traversable BVH. */
int cpt_tracer_build_extra_ms(cpt_tracer *t, float out2[2]);

/* ---- several GPUs of one node -----------------------------------------------------------------
 * No counterpart in the reference, which renders on one GPU (nothing under src/ selects a device).  The sampler is
 * seeded from the pixel index and the 1-based sample number (src/kernels/integrator/path.cu:92), so a render's
 * samples split into disjoint blocks; every GPU holds the whole scene and its BVH, renders its block as sums and one
 * sum-reduce of the W*H*4 floats over RCCL (xGMI) to the root, followed by a division by the sample count, gives the
 * image.  librccl is opened at the first call of this section, not when the library is loaded.
 *
 * cpt_shard_samples: the block [*begin, *begin + *count) of `total` samples starting at the 1-based index `first`
 * that shard `rank` of `world` renders (contiguous, in rank order, remainder to the lowest ranks). */
int cpt_shard_samples(uint32_t total, uint32_t rank, uint32_t world, uint32_t first, uint32_t *begin, uint32_t *count);

/* One process per GPU (a launcher such as torchrun): rank 0 calls cpt_comm_unique_id and hands the 128 bytes to every
 * process by its own means; each process makes its rank with its GPU current (cpt_set_device), renders its block with
 * cpt_tracer_render_samples(t, begin, count, 1, NULL, 0, 0) into a cleared film (cpt_tracer_reset_accumulation) and
 * calls cpt_tracer_reduce_film: in-place ncclReduce(sum) of the film to `root` on the tracer's stream and, with
 * `finish`, division by the sample count on the root (alpha becomes 1).  Asynchronous; cpt_tracer_synchronize waits. */
typedef struct cpt_comm cpt_comm;
int cpt_comm_unique_id(uint8_t id[128]);
cpt_comm *cpt_comm_create(const uint8_t id[128], int rank, int world);       /* collective over all ranks; NULL on failure */
void cpt_comm_destroy(cpt_comm *c);
/* What RCCL reports about the communicator (ncclCommCount, ncclCommUserRank, ncclCommCuDevice): a measurement can state
 * that the reduce ran over that many ranks instead of assuming it.  Any pointer may be NULL. */
int cpt_comm_census(const cpt_comm *c, int *ranks, int *rank, int *device);
int cpt_tracer_reduce_film(cpt_tracer *t, cpt_comm *c, int root, int finish);

/* One process, one host thread per shard (what `cuda-pathtracer --gpus N` runs): shard i renders on HIP device
 * shard_devices[i] with its own stream, film and replicated build.  A device may be named more than once; its shards
 * are summed on the device before the reduce between devices.  cpt_multi_render renders samples 1 .. total_spp split
 * over the shards, reduces to shard 0 and leaves the mean (alpha 1) in the film; elapsed_ms (optional) receives the
 * wall time of the call, reduce included.  Counters are summed over shards: rays, node records, triangle records,
 * paths, shading surfaces, dropped stack entries. */
typedef struct cpt_multi cpt_multi;
cpt_multi *cpt_multi_create(cpt_scene *s, uint32_t film_width, uint32_t film_height, const int *shard_devices, uint32_t num_shards);
void cpt_multi_destroy(cpt_multi *m);
int cpt_multi_set_max_depth(cpt_multi *m, int max_depth);
int cpt_multi_set_integrator(cpt_multi *m, int mode);
int cpt_multi_set_traversal(cpt_multi *m, int mode);                          /* as cpt_tracer_set_traversal, for every shard */
int cpt_multi_set_node_records(cpt_multi *m, int mode);                       /* as cpt_tracer_set_node_records, for every shard */
int cpt_multi_set_samples_per_launch(cpt_multi *m, uint32_t samples);
int cpt_multi_build_buffers(cpt_multi *m);
int cpt_multi_render(cpt_multi *m, uint32_t total_spp, float *elapsed_ms);
int cpt_multi_film_read(cpt_multi *m, float *rgba);
int cpt_multi_film_save(cpt_multi *m, const char *path);
int cpt_multi_counters(cpt_multi *m, uint64_t out6[6]);
int cpt_multi_reset_counters(cpt_multi *m);
uint32_t cpt_multi_num_devices(const cpt_multi *m);                           /* distinct GPUs = ranks of the reduce */

/* Device self-test: the traversal divides by a ray direction through a reciprocal-based sequence
 * that must equal the correctly rounded quotient bit for bit; this runs at least `pairs` operand
 * pairs from its admissible range through both and reports how many differ (expected: 0). */
int cpt_selftest_division(uint64_t pairs, uint32_t seed, uint64_t *mismatches);

/* Device measurement with no counterpart in the reference: copies `bytes` of device memory `repeats`
 * times with a plain streaming kernel and reports (bytes read + bytes written) per second, the
 * rate to set next to the nominal HBM peak when judging the build and traversal kernels. */
int cpt_stream_copy_rate(uint64_t bytes, int repeats, double *gb_per_s);

/* Device measurement with no counterpart in the reference: the rate at which the device issues vector instructions.
 * Every SIMD of every CU runs `waves_per_simd` (1 to 8) waves of independent wave64 instructions that touch no memory:
 * mix 0 = v_fma_f32 only, mix 1 = the kinds a box test is made of (subtract, multiply, compare + select, max, min),
 * mix 2 = v_pk_fma_f32 only (packed: two fp32 operations per lane and instruction).
 * ginst_per_s receives 1e9 wave64 instructions per second over the whole device -- the ceiling the instruction counts of
 * the traversal kernels are set against; cycles_per_inst (optional) the shader-clock cycles one wave spent per instruction
 * (waves_per_simd of them share a SIMD, so the SIMD issues one every cycles_per_inst / waves_per_simd); shader_mhz
 * (optional) the clock the device held meanwhile. */
int cpt_valu_issue_rate(int mix, int waves_per_simd, double *ginst_per_s, double *cycles_per_inst, double *shader_mhz);

/* Standalone OpenEXR writer used by Film::SaveTo (src/pathtracer/film.cpp:61-68). */
int cpt_write_exr(const char *path, const float *rgba, uint32_t width, uint32_t height);

/* ---- material maps -----------------------------------------------------------------------------
 * Replaces: LoadTexture (stb_image, 4 channels) + CuTexture      src/scene/loader.cpp:71-97,
 *           src/cuda_helpers/texture.cpp:5-27; MaterialValue::At  src/kernels/material/common.cuh:15-27.
 * Decodes a PNG or PPM/PGM file to RGBA8 (row 0 first, red in the low byte of each word).  With
 * rgba == NULL only the size is returned; capacity is in pixels. */
int cpt_image_load(const char *path, uint32_t *width, uint32_t *height, uint32_t *rgba, size_t capacity);
int cpt_image_save_png(const char *path, const uint32_t *rgba, uint32_t width, uint32_t height);
/* size of the map of mesh object `index`'s material: which = 0 diffuse, 1 specular, 2 emission;
 * width = height = 0 when the material has no such map; rgba may be NULL */
int cpt_scene_object_map(const cpt_scene *s, uint32_t index, int which, uint32_t *width, uint32_t *height, uint32_t *rgba, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* CPT_AMD_H */
