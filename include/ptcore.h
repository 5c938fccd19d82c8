/*
 * ptcore.h — C ABI of the MI355X path-tracing core (libptcore.so).
 *
 * The reference renderer has no FFI or plugin interface; the seam this library sits
 * behind is the Rust method
 *
 *     Camera::render(&self, objects: &PrimitiveBuffer, lights: &LightBuffer,
 *                    buffer: Arc<Mutex<RgbImage>>)            src/camera.rs:162-167
 *
 * called from main() on a dedicated thread (src/main.rs:90-95).  Everything that method
 * reads is handed over here as plain arrays: the four typed primitive buffers of every
 * PrimitiveBuffer in post-build order (src/objects.rs:130-136), their flat BVH node
 * arrays exactly as BVHBuilder::build leaves them (src/bvh.rs:295-311, :667-703), the
 * instance matrices (src/objects.rs:281-286), the light list (src/objects.rs:208-210),
 * the materials and textures reachable through Arc<dyn Material>/Arc<dyn Texture>
 * (src/materials.rs:80-123, src/textures.rs:10-12) and the private fields of Camera
 * (src/camera.rs:34-61).
 *
 * All functions return 0 on success or a PT_ERR_* code; pt_last_error() gives the text.
 * No function throws, aborts or panics across this boundary.  Host pointers inside the
 * descriptors are read only during the call that receives them.  A context belongs to
 * one device and is not re-entrant; use one context per GPU and one process per GPU.
 * There is no CPU fallback: without a usable HIP device pt_create fails.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#include "ptscene_src.h" /* pts_texture, pts_image, PTS_MAT_*, PTS_TEX_* */

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PT_OK = 0,
    PT_ERR_INVALID = 1,     /* bad argument / descriptor fails validation */
    PT_ERR_HIP = 2,         /* a HIP runtime call failed */
    PT_ERR_NOMEM = 3,       /* device allocation failed */
    PT_ERR_UNSUPPORTED = 4, /* e.g. instances nested inside instances, non-affine matrix */
    PT_ERR_STATE = 5        /* call order: render before upload_scene / set_camera */
};

enum { PT_PRIM_TRI = 0, PT_PRIM_SPHERE = 1, PT_PRIM_QUAD = 2, PT_PRIM_INSTANCE = 3,
       PT_PRIM_MOVING_SPHERE = 4 /* only in pt_hit::prim_type, as pt_trace_first_hit_at reports it */ };

/* One node of InnerPrimitiveBuffer::bvh (FlatBVHNode, src/bvh.rs:667-679) in 32 bytes.
 * The left child of an interior node is the next node of the array (src/bvh.rs:495). */
typedef struct pt_node {
    float bmin[3];
    float bmax[3];
    uint32_t a; /* interior: right_child_offset (index within this node array); leaf: offset (first primitive, within its buffer) */
    uint32_t b; /* interior: axis 0..2;                                        leaf: 0x80000000 | size                          */
} pt_node;
#define PT_NODE_LEAF 0x80000000u

/* One InnerPrimitiveBuffer<T> (src/objects.rs:105-109): a slice of the scene-wide
 * primitive array of its type and, if build_bvh ran, a slice of the node array. */
typedef struct pt_buffer_desc {
    uint32_t prim_first, prim_count;
    uint32_t node_first, node_count; /* node_count == 0: no BVH, linear scan (src/bvh.rs:726-737) */
} pt_buffer_desc;

/* One PrimitiveBuffer (src/objects.rs:130-136). */
typedef struct pt_primset_desc {
    pt_buffer_desc tris, spheres, quads, instances;
} pt_primset_desc;

/* One Instance (src/objects.rs:281-286). Matrices are row-major 4x4 and must be affine. */
typedef struct pt_instance_desc {
    float transform[16];
    float inverse[16];
    uint32_t child_set; /* index into sets[]; the child may not contain instances */
    /* An instance that sat inside another instance's set (src/objects.rs:281-286: an Instance holds a whole
     * PrimitiveBuffer, instances included) is listed in the world's set with the OUTERMOST instance's matrices in
     * this record, child_set = the set of the innermost one, and the levels further in as chain_count records
     * instances[chain_first ..], outer to inner, which belong to no set and have no chain of their own.  Rays go
     * through the inverses in that order and hits come back through the transforms in the opposite order, one
     * level at a time as the reference's nested Instance::hit does (:385-397), so the numbers match it bit for bit.
     * 0 / 0 for an instance that was not nested. */
    uint32_t chain_first, chain_count;
    uint32_t reserved;
} pt_instance_desc;

/* One Arc<dyn Material>. type/texture/p as in pts_material (constructor arguments after
 * the constructor's own clamping: Metal fuzz in [0,1], Disney specular_transmission >= 1e-4);
 * derived[] holds what the constructors precompute:
 *   [0..3] a_x, a_z, inv_a_x, inv_a_z   (DisneyMetal/DisneyGlass::new, src/materials/disney.rs:537-553)
 *   [4..5] a2, a2_invlog2               (DisneyClearcoat::new, :403-413) */
typedef struct pt_material_desc {
    uint32_t type;
    int32_t texture;
    float p[14];
    float derived[8];
} pt_material_desc;

typedef struct pt_scene_desc {
    uint32_t n_tris;
    const float *tri_data;  /* [n][8][3]: Triangle.data, row = px,py,pz,nx,ny,nz,tu,tv; column = vertex (src/objects/tri_mesh.rs:47-72) */
    const int32_t *tri_mat; /* [n] material index, -1 = None */
    uint32_t n_spheres;
    const float *spheres; /* [n][4] center, radius (src/objects/sphere.rs:11-16) */
    const int32_t *sphere_mat;
    uint32_t n_quads;
    const float *quads; /* [n][17] origin, u, v, normal, w, d, area (src/objects/quad_mesh.rs:12-22) */
    const int32_t *quad_mat;
    uint32_t n_instances;
    const pt_instance_desc *instances;
    uint32_t n_nodes;
    const pt_node *nodes;
    uint32_t n_sets;
    const pt_primset_desc *sets; /* [0] = objects, [1] = LightBuffer.objects (never has BVHs), [2..] = instance children */
    uint32_t n_materials;
    const pt_material_desc *materials;
    uint32_t n_textures;
    const pts_texture *textures;
    uint32_t n_images;
    const pts_image *images;
    uint64_t n_texel_bytes;
    const uint8_t *texels;
    uint32_t n_media; /* participating media, not in the reference: see pts_medium; at most PTS_MAX_MEDIA */
    const pts_medium *media;
    uint32_t n_moving;   /* moving spheres, not in the reference: see pts_moving_sphere.  They are the last n_moving */
    const float *moving; /* entries of spheres[] (centre at time 0) and sphere_mat[], outside every set;
                            moving[n][3] = centre at time 1 minus centre at time 0 */
} pt_scene_desc;

/* The fields of Camera that render/get_ray/ray_color read (src/camera.rs:34-61). */
typedef struct pt_camera_desc {
    uint32_t image_width, image_height;
    float center[3];
    float pixel00[3];
    float pixel_du[3];
    float pixel_dv[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float defocus_angle;
    uint32_t sqrt_spp;
    float inv_sqrt_spp;
    uint32_t max_depth;
    float background[3];
} pt_camera_desc;

/* Closest hit of one ray: what PrimitiveBuffer::hit returns (src/objects.rs:191-197),
 * reduced to the identity of the primitive and the ray parameter. */
typedef struct pt_hit {
    int32_t prim_type; /* PT_PRIM_TRI/SPHERE/QUAD, -1 = miss */
    int32_t prim_id;   /* index in its buffer after the BVH build permuted it */
    int32_t inst_id;   /* index in objects.instances after build, -1 = not inside an instance */
    float t;           /* ray.t_max on return */
    float u, v;        /* triangle barycentrics / quad alpha,beta; 0 for spheres */
} pt_hit;

typedef struct pt_stats {
    uint64_t rays;            /* rays traced by pt_render since the last reset (all bounces) */
    uint64_t paths;           /* camera paths started */
    uint64_t extend_launches; /* launches of the closest-hit kernel that had work */
    double extend_ms;         /* their summed device time (only with PT_OPT_PROFILE); launches of different
                                 batches overlap, so this can exceed the wall time */
    double extend_busy_ms;    /* time during which at least one of them was running (union of their intervals) */
    double shade_ms;
    double other_ms;
    double render_ms;         /* host wall time inside pt_render */
    double host_wait_ms;      /* the part of it spent waiting for the device (pt_render runs at most two batches ahead per pipe) */
    /* traversal work, only counted with PT_OPT_COUNTERS: */
    uint64_t node_tests, tri_tests, sphere_tests, quad_tests;
    uint64_t device_bytes;    /* device memory held by the context */
    uint64_t bad_refs;        /* out-of-range scene references caught by a library built with PT_BOUNDS_CHECK (else 0) */
    /* the two closest-hit kernels of a step, each bracketed by its own events (only with PT_OPT_PROFILE; they run
       side by side on two streams, so the two do not add up to extend_ms): */
    double walk_ms;           /* the ray-drawing kernel: the rays queued by earlier steps (all rays where camera rays
                                 have no kernel of their own) */
    double packet_ms;         /* the camera-ray kernel: the paths that start with the step, a wave at a time */
    uint64_t walk_launches, packet_launches; /* launches those times are summed over */
    uint64_t shadow_rays;     /* of `rays`: the shadow rays of PT_OPT_INTEGRATOR = 1 (traced by a kernel of their own) */
} pt_stats;

enum {
    PT_OPT_BSDF_SAMPLE_PROBABILITY = 0, /* BSDF_SAMPLE_PROBABILITY of src/camera.rs:317; default 1.0 */
    PT_OPT_MAX_PATHS = 1,               /* upper bound on paths in flight per batch; default 32000000 */
    PT_OPT_PROFILE = 2,                 /* 1: bracket kernels with HIP events, fill *_ms */
    PT_OPT_COUNTERS = 3,                /* 1: count AABB / primitive tests (implies the reference-order traversal) */
    PT_OPT_TRAVERSAL = 4,               /* 0 (default): 64-byte two-box nodes; 1: the reference's loop over 32-byte nodes;
                                           2: as 0, and pt_trace_first_hit / _at search for 64 consecutive rays at a time the
                                           way PT_OPT_PACKET does for camera rays (scenes without device-built BVHs), so
                                           that search can be checked on any rays */
    PT_OPT_PIPES = 5,                   /* streams the batches are spread over; 1..8.  Default without the rolling schedule: 4.
                                           With it (PT_OPT_ROLLING), until this option is set: 1 where a launch starts more
                                           than 12 M paths, 2 where it starts fewer (image and stratum ranges so small that a
                                           launch's last rays leave most of the chip idle; twice the path state) */
    PT_OPT_REFILL_BELOW = 6,            /* a traversing wave with fewer busy lanes than this fetches new rays; 1..64, default 32 */
    PT_OPT_LEAF_BATCH = 7,              /* lanes that wait at a leaf before the wave runs a primitive test; 1..64, default 16 */
    PT_OPT_BLOCKS_PER_CU = 8,           /* resident workgroups per CU of the traversal kernels; 1..5, default 5 */
    PT_OPT_ROLLING = 10,                /* paths started per launch of the rolling schedule, in which every launch carries
                                           bounce 0 of a new batch together with the later bounces of the batches before it;
                                           0 = one batch at a time per pipe; default 32000000 */
    PT_OPT_DEVICE_BVH = 11,             /* a triangle buffer that is uploaded without nodes and has at least this many
                                           triangles gets its BVH built on the device during pt_upload_scene (a different
                                           tree from the host builder's: same closest distances, the winner among exact
                                           ties may differ; triangles are reordered, see pt_read_triangle_order);
                                           0 = never (default): such buffers are searched in order */
    PT_OPT_INTEGRATOR = 12,             /* 0 (default): the reference's estimator, Camera::ray_color (src/camera.rs:262-373);
                                           1: next-event estimation, not in the reference: one LightBuffer::sample
                                           direction per non-specular vertex tested with a shadow ray, the path continued
                                           by BSDF sampling, balance heuristic between the two; BSDF_SAMPLE_PROBABILITY
                                           is not used */
    PT_OPT_TAIL_BELOW = 9,              /* accepted and ignored (it selected a launch that no longer exists) */
    PT_OPT_PACKET = 14,                 /* 1 (default): the paths that start with a launch (camera rays: 64 consecutive ones are
                                           a few neighbouring pixels with all their strata) are searched a wave at a time,
                                           the wave walking the reference's loop once with a lane mask, nodes and triangles
                                           read once per wave (scenes without device-built BVHs).  0: every ray on
                                           its own, as the rays of later bounces always are.  Same results either way */
    PT_OPT_BATCH_TICKS = 15,            /* rolling schedule: launches a batch takes part in before the few paths it still has are
                                           finished in one go; its per-path sums are held that long, so this sets the
                                           memory (16 B x paths per launch x this).  0 (default): max_depth at first, then
                                           what the first drain of the pipeline shows about how fast this scene's paths
                                           end (a handful of paths left, at least 8 launches); 1: always
                                           max_depth; >= 2: that many.  Same results for any value */
    PT_OPT_SHADE_CLAIM = 16,            /* how the shading kernels share out their queues: 0: wave w takes the pieces w, w + waves, ...;
                                           n in 1..64: waves draw blocks of n 256-entry chunks as they get to them, from a
                                           counter per XCD; -1 (default): chosen step by step, 0 where most of a step's
                                           vertices are first hits (queued in path order already), for queues of fewer
                                           than 2048 chunks and in scenes with more than four material types, 4 elsewhere.  Same results for any value */
    PT_OPT_LIVE_FACTOR = 13             /* rolling schedule: rays in flight a pipe has room for, as a multiple of the paths
                                           started per launch (at least 2, at most max_depth; default 4).  A batch starts
                                           only when its paths fit beside the rays in flight, so a scene whose paths live
                                           longer than this allows for renders correctly, with batches starting less often */
};

typedef struct pt_context pt_context;

/* Creates a context on HIP device `device`. */
int pt_create(int device, pt_context **out);
void pt_destroy(pt_context *ctx);
const char *pt_last_error(const pt_context *ctx); /* ctx may be NULL: error of the last failed pt_create */

int pt_set_option(pt_context *ctx, int key, double value);

/* Copies the scene to the device (replaces any earlier scene). */
int pt_upload_scene(pt_context *ctx, const pt_scene_desc *scene);
/* Sets the camera and (re)allocates + zeroes the accumulator for its resolution. */
int pt_set_camera(pt_context *ctx, const pt_camera_desc *camera);

/* Camera::render for the strata [stratum_begin, stratum_end) of the s_u-major sample
 * list (src/camera.rs:176-181): adds into the accumulator and the pass count
 * (src/camera.rs:199, :212-217).  The generator is keyed by (seed, pixel, stratum).
 * The work is enqueued on the device and the call returns (it waits only so as not to run
 * more than two batches ahead of each pipe); successive calls overlap on the device, and
 * every pixel still receives its samples in stratum order of the calls.  Every entry point
 * that reads results (pt_read_*, pt_copy_accum_to_device, pt_stats_get, ...) waits for all
 * rendering enqueued before it; pt_sync waits without reading anything. */
int pt_render(pt_context *ctx, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed);
/* No counterpart in the reference (its render() blocks): waits for the rendering enqueued
 * so far and reports an asynchronous device error, if any. */
int pt_sync(pt_context *ctx);

/* As pt_render, but instead of accumulating returns every sample's radiance before the
 * NaN filter: out[(stratum - stratum_begin) * W*H + pixel][3] (host memory). */
int pt_render_samples(pt_context *ctx, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed, float *out);

/* Camera::get_ray for every pixel of one stratum (src/camera.rs:238-260):
 * rays[W*H][8] = origin, direction, t_min, t_max (host memory). */
int pt_generate_rays(pt_context *ctx, uint32_t stratum, uint64_t seed, float *rays);

/* PrimitiveBuffer::hit for n caller-supplied rays: rays[n][8] as above, out[n] (host memory). */
int pt_trace_first_hit(pt_context *ctx, const float *rays, uint64_t n, pt_hit *out);

/* As pt_trace_first_hit, in a scene with moving spheres (pts_moving_sphere, not in the reference), which
 * pt_trace_first_hit does not see: times[n] = the time in [0, 1] at which each ray is traced (NULL: all 0).  The
 * moving spheres are tested after everything else, in index order, where they are at that time, as the renderer
 * does for a path of that time; a hit on moving sphere k has prim_type PT_PRIM_MOVING_SPHERE, prim_id k. */
int pt_trace_first_hit_at(pt_context *ctx, const float *rays, const float *times, uint64_t n, pt_hit *out);

/* No counterpart in the reference.  pt_trace_first_hit for callers that cannot allocate device memory
 * (tight serving loops, harnesses that forbid hipMalloc, code that runs between the captures of a HIP graph):
 * the same kernels, with everything they need beyond the rays and the hits (path state, the ray queue, the
 * spilled part of the traversal stacks) in a fixed array that the library holds on every GPU it runs on (about
 * 45 MB per GPU, until the process ends).  The call allocates and frees nothing.  rays_device[n][8] and
 * out_device[n] are DEVICE memory of ctx's GPU.  The call runs on the context's own stream and waits for it, so
 * it cannot itself be captured into a graph, and it is not ordered against work the caller has queued on other
 * streams: rays_device must be complete before the call, and out_device is complete when it returns.  At most
 * PT_STATIC_TRACE_MAX_RAYS rays per call (PT_ERR_INVALID beyond that).  All contexts of one GPU share the array;
 * the library lets one such call at a time run per GPU (a host mutex), so concurrent callers on different
 * threads are safe and take turns. */
#define PT_STATIC_TRACE_MAX_RAYS 262144u
int pt_trace_first_hit_static(pt_context *ctx, const float *rays_device, uint64_t n, pt_hit *out_device);

/* LightBuffer::pdf (src/objects.rs:259-278) for n caller-supplied rays, of which only origin and
 * direction count: the sum of every light's pdf for the ray divided by the number of lights, out[n]
 * (host memory).  tris_by_solid_angle = 0 is the reference's Triangle::pdf, 1 / area
 * (src/objects/tri_mesh.rs:252-261); 1 converts it to a density over directions as
 * PT_OPT_INTEGRATOR = 1 uses it. */
int pt_light_pdf(pt_context *ctx, const float *rays, uint64_t n, int tris_by_solid_angle, float *out);

/* The Material trait of one material at caller-supplied shading points (src/materials.rs:80-123), for checking
 * the device's shading arithmetic call by call: scatter with the generator keyed by (seed, pixel, stratum) as
 * pt_render keys it, starting at its first draw; bsdf_evaluate and scattering_pdf at the scattered direction
 * and at the caller's w_in; emitted.  Vectors are in the local shading frame (y = surface normal); u, v, p and
 * front_face stand for the HitRecord fields the materials read (src/objects.rs:14-27).  in[n], out[n] are host
 * memory.  No counterpart in the reference, whose materials are reached through Arc<dyn Material>. */
typedef struct pt_material_probe_in {
    float w_out[3];
    float u, v;
    float p[3];
    uint32_t front_face;
    uint32_t pixel, stratum;
    float w_in[3];
} pt_material_probe_in;
typedef struct pt_material_probe_out {
    float w_in[3];       /* Material::scatter */
    uint32_t draws;      /* random numbers it consumed */
    float bsdf[3];       /* bsdf_evaluate(w_out, scattered w_in) */
    float pdf;           /* scattering_pdf(w_out, scattered w_in) */
    float bsdf_at[3];    /* bsdf_evaluate(w_out, in.w_in) */
    float pdf_at;        /* scattering_pdf(w_out, in.w_in) */
    float emitted[3];    /* Material::emitted */
    uint32_t reserved;
} pt_material_probe_out;
int pt_material_probe(pt_context *ctx, int32_t material, uint64_t seed, const pt_material_probe_in *in, uint64_t n, pt_material_probe_out *out);

/* Accumulator: W*H*3 f32 sums, row-major (i = u + v*W, src/camera.rs:202-205), and the pass count. */
int pt_read_accum(pt_context *ctx, float *rgb_sum, uint32_t *n_samples);
/* (sum / n_samples).to_rgb() per pixel (src/camera.rs:221-226, src/types/color.rs:68-78). */
int pt_read_rgb8(pt_context *ctx, uint8_t *rgb);
int pt_reset_accum(pt_context *ctx);
/* order[i] = index in the uploaded triangle arrays of the triangle that pt_hit::prim_id == i names
 * (the identity unless PT_OPT_DEVICE_BVH reordered a buffer); n must be the scene's triangle count.
 * No counterpart in the reference, whose builder permutes its Vec in place (src/bvh.rs:295-311). */
int pt_read_triangle_order(pt_context *ctx, uint32_t *order, uint32_t n);
/* Device-to-device copies of the accumulator, for a collective on device memory
 * (W*H*3 f32).  `stream_sync` semantics: both block until the copy is complete. */
int pt_copy_accum_to_device(pt_context *ctx, void *dst_device);
int pt_set_accum_from_device(pt_context *ctx, const void *src_device, uint32_t n_samples);

int pt_stats_get(pt_context *ctx, pt_stats *out);
int pt_stats_reset(pt_context *ctx);

/* ---- several GPUs of one node behind one handle ------------------------------------------------------------
 * Camera::render (src/camera.rs:162-167) is one call; a group keeps it one call when the work is spread over
 * GPUs.  Every member (one pt_context per listed device) holds the whole scene; pt_group_render cuts its
 * stratum range into contiguous pieces, one per member (pt_group_split), and each member enqueues its piece
 * from a host thread of its own; the passes are independent given the (seed, pixel, stratum) generator
 * (src/camera.rs:185-198).  Readers sum the members' accumulators (src/camera.rs:172-173, :216) onto the first
 * device with one RCCL ncclReduce (sum, f32, W*H*3 values) and add up the pass counts (src/camera.rs:199); f32
 * sums are re-associated, so a group's image equals one context's to rounding, not bits.  Rendering may go on
 * after a read: the members' own accumulators are left as they are.  A group that lists a device more than once
 * is summed with copies instead of RCCL (for rehearsing the split on one GPU), and a group of one device needs no
 * collective at all.  librccl.so is opened when a group of several distinct devices is created, not when libptcore
 * is loaded, and on a thread of its own: opening it and making the communicators takes seconds, which the first
 * read waits for and uploads and rendering do not (a failure is reported by that read).  One group is not re-entrant. */
typedef struct pt_group pt_group;
int pt_group_create(const int *devices, int n, pt_group **out);
void pt_group_destroy(pt_group *g);
const char *pt_group_last_error(const pt_group *g); /* g may be NULL: error of the last failed pt_group_create */
int pt_group_size(const pt_group *g);
pt_context *pt_group_member(pt_group *g, int i);    /* for per-GPU queries (pt_stats_get, pt_trace_first_hit, ...) */
/* The piece [*begin_i, *end_i) of [stratum_begin, stratum_end) that member i of n renders: contiguous, sizes
 * differ by at most one, larger pieces first.  A pure function (no GPU needed). */
void pt_group_split(uint32_t stratum_begin, uint32_t stratum_end, int n, int i, uint32_t *begin_i, uint32_t *end_i);
int pt_group_set_option(pt_group *g, int key, double value);
int pt_group_upload_scene(pt_group *g, const pt_scene_desc *scene);
int pt_group_set_camera(pt_group *g, const pt_camera_desc *camera);
int pt_group_render(pt_group *g, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed);
int pt_group_sync(pt_group *g);
int pt_group_read_accum(pt_group *g, float *rgb_sum, uint32_t *n_samples);
int pt_group_read_rgb8(pt_group *g, uint8_t *rgb);
int pt_group_reset_accum(pt_group *g);
int pt_group_stats_get(pt_group *g, pt_stats *out); /* counts summed over the members, times of the slowest */
int pt_group_stats_reset(pt_group *g);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
