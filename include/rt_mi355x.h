/*
 * rt_mi355x.h — C ABI of librt_mi355x.so, the MI355X (gfx950) implementation of the path
 * tracer's per-frame compute dispatch.
 *
 * Each entry point cites the piece of the reference renderer it stands in for
 * (paths relative to the reference repository).  All functions return 0 on success and a
 * non-zero code on failure; rt_last_error() then describes the failure.  A context is bound
 * to one GPU and is not thread-safe; use one context per device.
 *
 * Buffer layouts are the reference's std430 layouts, byte for byte:
 *   binding  5  spheres   32 B/elem   (shaders/compute.glsl:14-19,  src/primitive/sphere/sphere.py:38-51)
 *   binding  6  planes    32 B/elem   (compute.glsl:22-26,          src/primitive/plane/plane.py:37-45)
 *   binding  7  boxes     32 B/elem   (compute.glsl:37-41,          src/primitive/box/box.py:39-47)
 *   binding  9  materials 12 B/elem   (compute.glsl:43-47,          src/material/material.py:48-55)
 *   binding 11  BVH nodes 64 B/elem   (compute.glsl:6-11,           src/bvh/bvh_node.py:337-344)
 *   binding 12  triangles 64 B/elem   (compute.glsl:29-35,          src/primitive/triangle/triangle.py:48-55)
 * Images are RGBA32F, row 0 = bottom row (the GL image convention the shader writes in).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum {
    RT_BINDING_SPHERES = 5,
    RT_BINDING_PLANES = 6,
    RT_BINDING_BOXES = 7,
    RT_BINDING_MATERIALS = 9,
    RT_BINDING_BVH = 11,
    RT_BINDING_TRIANGLES = 12
};

/* rt_render flags */
enum {
    RT_FLAG_AUX = 1u,        /* also record the 16-word per-pixel parity record (rt_read_aux)   */
    RT_FLAG_STATS = 2u       /* also count node visits and primitive tests (slower)              */
};

typedef struct {
    uint64_t rays;          /* traverse calls (one per loop iteration of compute.glsl:498-500)   */
    uint64_t pixel_samples; /* pixel-samples rendered                                            */
    uint64_t nodes;         /* node visits           (RT_FLAG_STATS only)                        */
    uint64_t tri_tests;     /* triangle tests        (RT_FLAG_STATS only)                        */
    uint64_t sphere_tests;  /* sphere tests          (RT_FLAG_STATS only)                        */
    uint64_t box_tests;     /* box tests             (RT_FLAG_STATS only)                        */
    uint64_t max_stack;     /* deepest traversal stack seen (RT_FLAG_STATS only)                 */
    uint64_t stack_overflows; /* pushes dropped because the stack was full (always counted)      */
    uint64_t hits;          /* rays that hit a surface (RT_FLAG_STATS only)                      */
    double last_render_ms;  /* last rt_render / rt_render_sum call: device time from its first to its
                               last piece of work, every stream included (valid after rt_sync)   */
    double last_trace_ms;   /* sum of the durations of that call's trace launches (first 256);
                               launches of one call overlap, so this can exceed last_render_ms    */
} rt_stats;

/* Stands in for Renderer.__init__ (src/renderer/renderer.py:64-154): creates the device
 * objects.  `device` is the HIP device ordinal.  Default size 1280x720 (renderer.py:75-76). */
int rt_create(int device, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);

/* Message for the most recent failure on the calling thread. */
const char *rt_last_error(void);

/* Renderer.update_size (renderer.py:347-382): reallocates the three images, contents zeroed.
 * As in the reference, the accumulated-frame count is left alone: with N frames accumulated, the
 * next frame is blended into the fresh, black accumulation image with weight 1/(N+1).  A host
 * that wants a clean start after a resize calls rt_set_rendered_frames(ctx, 0) (or
 * rt_reset_accumulation after its next frame). */
int rt_resize(rt_ctx *ctx, int width, int height);

/* Renderer.rendered_frames is a plain attribute the host may assign (renderer.py:106); render()
 * sends it as the `frames` uniform of the accumulate pass (renderer.py:255-257).  This sets the
 * count the next rt_render starts from. */
int rt_set_rendered_frames(rt_ctx *ctx, uint32_t frames);

/* Renderer.update_ssbo (renderer.py:308-324): copies `nbytes` of host data into the buffer
 * for `binding`, replacing it wholesale.  The caller keeps ownership of `data`.  `nbytes` must be a
 * multiple of the binding's element size, and the element count must fit the 30-bit index of a
 * BVH child word (below 2^30; below 2^30 - 1 for the BVH itself).  Child words and material
 * indices that point beyond their arrays are legal and read zero words, as GL robust buffer
 * access delivers them to the reference shader.  If the call fails, the binding is left empty. */
int rt_upload(rt_ctx *ctx, int binding, const void *data, size_t nbytes);

/* renderer.scene_min_coords / renderer.scene_extent (src/scene/scene.py:70-71), sent as the
 * uniforms scene_min / scene_extent (renderer.py:199-210). */
int rt_set_scene_bounds(rt_ctx *ctx, const float scene_min[3], const float scene_extent[3]);

/* Uniforms inverse_view_projection (16 floats in memory order, sent untransposed) and eye
 * (renderer.py:186-198). */
int rt_set_camera(rt_ctx *ctx, const float inv_view_proj[16], const float eye[3]);

/* One Renderer.render() per frame (renderer.py:164-306), `nframes` times: the compute dispatch
 * with time = first_frame + i and the given `bounces` uniform (renderer.py:211-246), then the
 * accumulate pass (shaders/screen_fragment.glsl:94-108, renderer.py:249-288) which folds the
 * sample into the running mean and increments the accumulated-frame count.  Asynchronous;
 * rt_sync or any rt_read_* waits for it.  Consecutive frames overlap on the device (the compute
 * dispatch of frame k+1 may start while frame k drains; the accumulate passes run in frame order),
 * which changes no result; RT_OVERLAP=0 in the environment at rt_create keeps frames strictly
 * one after another, and frames with RT_FLAG_AUX always are.  `bounces` is the float the uniform
 * is (the shader traces uint(bounces) bounces, at most 64); a negative value or NaN, whose
 * conversion GLSL leaves undefined and the reference's property setter never lets through
 * (renderer.py:395-397), is an error. */
int rt_render(rt_ctx *ctx, uint32_t first_frame, uint32_t nframes, float bounces, uint32_t flags);

/* Renderer.reset_accumulation (renderer.py:326-345): the accumulation image becomes a copy of the
 * last sample image and the accumulated-frame count returns to 0. */
int rt_reset_accumulation(rt_ctx *ctx);

/* Number of frames folded into the accumulation image (Renderer.rendered_frames). */
int rt_rendered_frames(rt_ctx *ctx, uint32_t *out);

/* The image the compute shader stores (compute.glsl:673): sRGB-encoded sample of the last frame. */
int rt_read_sample(rt_ctx *ctx, float *rgba);
/* The accumulation image (`_last_frame` / `_color_tex`, renderer.py:274-286), sRGB-encoded. */
int rt_read_accumulation(rt_ctx *ctx, float *rgba);
/* The same two images without a copy into caller memory: the library reads the image back into a
 * page-locked host buffer of its own (allocated at the first call, width*height RGBA32F) and
 * hands out its address.  The transfer runs at the PCIe rate (a copy into pageable caller memory
 * runs at a fraction of it), which is what an interactive loop that presents every frame wants
 * (renderer.py:294-300 samples the texture directly; a host presents it with glTexSubImage2D).
 * `which`: RT_IMAGE_SAMPLE or RT_IMAGE_ACCUMULATION.  *rgba stays valid, and its contents
 * unchanged, until the next rt_map_image, rt_resize or rt_destroy on this context. */
enum { RT_IMAGE_SAMPLE = 0, RT_IMAGE_ACCUMULATION = 1 };
int rt_map_image(rt_ctx *ctx, int which, const float **rgba);
/* Parity record of the last frame rendered with RT_FLAG_AUX: 16 u32 per pixel,
 *   [0] type<<30|index of the primitive the first ray hit (0xFFFFFFFF: none), [1] bits of that
 *   ray's hit distance, [2] hit exists, [3] node visits (RT_FLAG_STATS), [4..6] ray direction,
 *   [7..9] ray origin, [10..12] throughput, [13] RNG state, [14] rays traced, [15] ray ior —
 *   [4..15] as they stand when the bounce loop ends. */
int rt_read_aux(rt_ctx *ctx, uint32_t *aux16);

int rt_get_stats(rt_ctx *ctx, rt_stats *out);
int rt_reset_stats(rt_ctx *ctx);
int rt_sync(rt_ctx *ctx);

/*
 * Sharded rendering for multi-GPU runs (no counterpart in the reference, which renders every
 * pixel of every frame on one device).  Renders frames [first_frame, first_frame+nframes) for the
 * image tiles owned by shard `shard` of `nshards` (tiles of 64x64 pixels dealt round-robin in
 * Morton order) and ADDS each sample's linear radiance into `sum_rgba` (device pointer, W*H*4 f32,
 * caller-zeroed; NULL selects the context's own buffer).  Frames may also be strided:
 * only frames f with (f - first_frame) % frame_stride == frame_phase are rendered.
 * After the per-GPU buffers are summed (RCCL reduce), rt_finalize_sum turns the sum into the
 * sRGB-encoded mean image.
 * Two launches are kept in flight: launches alternate, starting after every rt_clear_sum, between
 * adding into `sum_rgba` and adding into a second image inside the context, which is added to
 * `sum_rgba` and zeroed by the next rt_sync, rt_finalize_sum, rt_clear_sum, rt_render, rt_get_stats
 * or a call that names another `sum_rgba`.  `sum_rgba` is therefore complete only after one of those calls
 * (call rt_sync before handing the buffer to a collective), and holds the fp32 sum
 * (even launches, in order) + (odd launches, in order): deterministic, though not the same
 * rounding as one running sum.  RT_OVERLAP=0 gives the single running sum.
 */
int rt_render_sum(rt_ctx *ctx, uint32_t first_frame, uint32_t nframes, float bounces, uint32_t shard,
                  uint32_t nshards, uint32_t frame_stride, uint32_t frame_phase, float *sum_rgba);
int rt_sum_buffer(rt_ctx *ctx, float **device_ptr, size_t *nbytes);
int rt_clear_sum(rt_ctx *ctx, float *sum_rgba);
/* accumulation image = srgb(sum / spp), alpha 1 */
int rt_finalize_sum(rt_ctx *ctx, const float *sum_rgba, uint32_t spp);

/*
 * Fast mode: opt-in, default off, NOT the reference's data and never what parity is claimed for.
 * The reference's tree (src/bvh/bvh_node.py:103-230: SAH on the axis depth % 3; :476-480: child
 * boxes rounded to the nearest 16-bit code, so not conservative) decides which hits a ray finds;
 * the default mode consumes exactly that tree (binding 11).  With fast mode on, rt_render /
 * rt_render_sum traverse the library's own tree instead -- all-axis SAH, conservative 8-bit child
 * boxes in 64-byte nodes -- built on the host from the uploaded triangles, spheres and boxes
 * (binding 11 is ignored).  The traversal rules (compute.glsl:225-330), ray generation,
 * intersection routines, materials and RNG are unchanged, so images differ from the default mode
 * only where the reference's tree decides the hit.  The wavefront pipeline (RT_PIPELINE=wave)
 * does not apply to it.  rt_fast_bvh_size / rt_fast_bvh_export hand out the built tree (64-byte
 * node records, layout in csrc/rt_fastbvh.cpp; the export holds n_nodes + 1 of them, the last an
 * all-empty sentinel) so that a checker can traverse exactly the same data.
 */
int rt_set_fast_mode(rt_ctx *ctx, int enable);
int rt_fast_bvh_size(rt_ctx *ctx, uint32_t *n_nodes);
int rt_fast_bvh_export(rt_ctx *ctx, void *nodes, size_t nodes_bytes);

/*
 * No counterpart in the reference (a scheduling detail of this library; results do not depend on
 * it).  The default pipeline has two traversal variants -- one kind of step for nodes and leaves,
 * or leaf intersections in steps of their own -- and neither is faster on every scene.  Once a
 * scene has seen 64 plain frames (no RT_FLAG_*), or before a call of 64 or more, rt_render /
 * rt_render_sum therefore time both with four extra launches that write no image and touch no
 * counter the caller sees, and keep the faster.  That costs about four frame times (less on
 * images beyond 1080p, which are timed on every third or fourth 64x64 tile), which the better
 * variant's 6 to 18 % per frame earn back within a hundred frames, and happens once per scene: a
 * scene uploaded again at about the same size, within a quarter in records and pixels, keeps the
 * choice.  Until then the size of the scene decides.  *leaf_steps = the variant in use (1 = leaf
 * steps of their own), *decided = 0 until the timing has happened.  RT_LEAF_STEPS=0 / 1 in the
 * environment at rt_create fixes the variant and skips the timing.
 */
int rt_traversal_variant(rt_ctx *ctx, int *leaf_steps, int *decided);
/* The same choice per context instead of through the environment: variant 0 = the unified step,
 * 1 = leaf steps of their own, either of which also switches the four timing launches off (an
 * interactive host that must not see a one-off spike of about four frame times at its 65th frame
 * calls this once); -1 = back to the automatic choice. */
int rt_set_traversal_variant(rt_ctx *ctx, int variant);

/* Diagnostics with no counterpart in the reference.
 * rt_device_info: the HIP device ordinal the context is bound to and its PCI bus id
 * ("0000:05:00.0"; `pci_bus_id` must hold at least 16 bytes).
 * rt_shader_clock: the shader clock in GHz the megakernel's wavefronts ran at since the last
 * rt_reset_stats, measured by the kernel itself (a few wavefronts per launch compare the shader
 * cycle counter with the constant 100 MHz counter over their lifetime); 0 if nothing was sampled.
 * rt_selftest: checks on the device the one place where the kernels replace an IEEE operation of
 * the reference by a shorter sequence claimed to give the same bits: the reciprocal of the triangle
 * test's determinant (compute.glsl:887, `1.0 / det`).  Runs over every float (about 0.1 s) and
 * returns the number of values for which the two differ; anything but 0 means this GPU must not
 * be trusted to reproduce the reference. */
int rt_device_info(rt_ctx *ctx, int *device, char *pci_bus_id, size_t pci_bus_id_bytes);
int rt_shader_clock(rt_ctx *ctx, double *ghz);
int rt_selftest(rt_ctx *ctx, unsigned long long *reciprocal_mismatches);

/*
 * Collectives for multi-GPU runs: one process per GPU, RCCL over xGMI, no framework needed (no
 * counterpart in the reference).  Rank 0 draws an id with rt_comm_unique_id and hands its
 * RT_COMM_ID_BYTES bytes to the other ranks by any means (a file, an environment variable, a
 * socket); every rank then calls rt_comm_init on its context.  rt_comm_reduce_sum adds the ranks'
 * linear sum images (`sum_rgba`: device pointer, W*H*4 f32, NULL = the context's own buffer; the
 * same choice and image size on every rank) into rank `root`'s image; it first waits for the frames
 * the context has in flight, and returns when the result is in place.  rt_comm_all_reduce_f64
 * combines up to RT_COMM_MAX_SCALARS host doubles across ranks in place (timings, ray counts);
 * rt_comm_barrier returns once every rank has called it.  librccl is opened at the first of these
 * calls.  The communicator is released by rt_destroy.
 */
#define RT_COMM_ID_BYTES 128
#define RT_COMM_MAX_SCALARS 64
enum { RT_COMM_SUM = 0, RT_COMM_MAX = 1 };
int rt_comm_unique_id(char id[RT_COMM_ID_BYTES]);
int rt_comm_init(rt_ctx *ctx, int rank, int world, const char id[RT_COMM_ID_BYTES]);
int rt_comm_reduce_sum(rt_ctx *ctx, float *sum_rgba, int root);
int rt_comm_all_reduce_f64(rt_ctx *ctx, double *values, int count, int op);
int rt_comm_barrier(rt_ctx *ctx);
/* what the communicator itself reports: this rank's index in it and the number of ranks it spans */
int rt_comm_info(rt_ctx *ctx, int *rank, int *nranks);

/*
 * One frame without any allocation.  No counterpart in the reference; it is for callers that
 * cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  A variant of rt_render (same kernels, same results bit for bit) that takes no
 * context: every input is a DEVICE pointer to a buffer in the reference's layout (the bindings
 * above), the outputs are caller-owned device images of width*height RGBA32F, and the work is
 * queued on `stream` (a hipStream_t, NULL = the default stream) without waiting for it.
 * `sample_rgba` receives what the compute pass stores for time = frame (pixels outside the
 * dispatch's 8x4 groups are left untouched); if `acc_rgba` is not NULL the accumulate pass then
 * folds the sample into it with weight 1/(accumulated_frames+1), as rt_render does.
 * The device records the kernels read (what rt_upload + rt_render keep in hipMalloc'ed buffers) and
 * the ray counters live in a fixed array inside the library, about 8.7 MB per GPU, held until the
 * process ends; scenes beyond the RT_NOALLOC_MAX_* element counts are rejected with an error.
 * Calls on one GPU share that array: calls that overlap in time on the same GPU race; calls
 * queued on one stream (or ordered by events) do not.  Calls no hipMalloc / hipFree.
 */
#define RT_NOALLOC_MAX_NODES 32768u
#define RT_NOALLOC_MAX_TRIANGLES 65536u
#define RT_NOALLOC_MAX_SPHERES 4096u
#define RT_NOALLOC_MAX_BOXES 4096u
#define RT_NOALLOC_MAX_PLANES 256u
#define RT_NOALLOC_MAX_MATERIALS 4096u
typedef struct {
    const void *spheres;   uint32_t n_spheres;    /* binding 5,  32 B/elem */
    const void *planes;    uint32_t n_planes;     /* binding 6,  32 B/elem */
    const void *boxes;     uint32_t n_boxes;      /* binding 7,  32 B/elem */
    const void *materials; uint32_t n_materials;  /* binding 9,  12 B/elem */
    const void *bvh;       uint32_t n_nodes;      /* binding 11, 64 B/elem */
    const void *triangles; uint32_t n_triangles;  /* binding 12, 64 B/elem */
    float scene_min[3], scene_extent[3];          /* host values of the uniforms */
    float inv_view_proj[16], eye[3];
} rt_scene_view;
int rt_render_noalloc(int device, const rt_scene_view *scene, int width, int height, uint32_t frame, float bounces,
                      float *sample_rgba, float *acc_rgba, uint32_t accumulated_frames, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
