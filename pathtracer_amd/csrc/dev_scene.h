// Device-resident scene layout (HBM) and the per-path state arrays.
//
//   nodes      2 x float4 per BVH node (32 B): (min.xyz, max.x) (max.yz, a, b)      — see pt_node
//   wide       4 x float4 per interior node (64 B): left box, right box, (left ref, right ref, axis, -)
//   tri_hot    3 x float4 per triangle (48 B, 36 used): v0, e1 = v1-v0, e2 = v2-v0   — all the intersector reads
//   tri_cold   6 x float4 per triangle (96 B): the 8x3 attribute matrix, fetched once per path vertex
//   spheres    1 x float4 (center, radius)
//   quads      4 x float4 (64 B): (origin, u.x) (u.yz, v.xy) (v.z, normal) (w, d)
//   insts      7 x float4 (112 B): inverse 3x4, transform 3x4, (child_set, chain_first, chain_count, -)
// Primitive indices stored in hit records are scene-wide ("global") indices into these arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ptcore.h"
#include "dev_math.h"

namespace ptd {

// One InnerPrimitiveBuffer on the device.
struct DevBuffer {
    uint32_t prim_first, prim_count;  // slice of the scene-wide primitive array of its type
    uint32_t node_first, node_count;  // slice of `nodes` (reference order); node_count 0 = no BVH
    uint32_t wide_first;              // first 64-byte node of this buffer in `wide`
    uint32_t root_ref;                // reference to the root (a wide-node index or a leaf reference)
    float rmin[3], rmax[3];           // the root's own box
    uint32_t tree;                    // 0: no BVH, primitives are tested in order; 1: `nodes` and `wide`;
                                      // 2: `wide` only (built on the device, bvh_build.h)
};
// Child references inside wide nodes: an index into the buffer's wide nodes, or
// WIDE_LEAF | count << 24 | offset (count <= 127, offset < 2^24) for a leaf.
#define WIDE_LEAF 0x80000000u
struct DevSet {
    DevBuffer tris, spheres, quads, insts;
};

struct DevMaterial {  // 96 B, same field order as pt_material_desc
    uint32_t type;
    int32_t texture;
    float p[14];
    float derived[8];
};

struct DevScene {
    const float4 *nodes;
    const float4 *wide;
    const float4 *tri_hot;
    const float4 *tri_cold;
    const int32_t *tri_mat;
    const float4 *spheres;
    const int32_t *sphere_mat;
    const float4 *quads;
    const float *quad_area;
    const int32_t *quad_mat;
    const float4 *insts;
    const DevSet *sets;
    const DevMaterial *mats;
    const uint8_t *mat_queue;  // material index -> shade queue slot
    const pts_texture *textures;
    const pts_image *images;
    const uint8_t *texels;
    const pts_medium *media;    // participating media (pts_medium), n_media of them
    const float4 *moving_delta; // moving spheres (pts_moving_sphere): centre at time 1 minus centre at time 0, for the
                                // last n_moving entries of `spheres`, which no set and no BVH refers to
    const uint32_t *tri_order;  // [n_tris] position at upload of the triangle now at each position
    const float4 *light_boxes;  // [2 * (light triangles + spheres + quads)] lower, upper corner of a box around each
                                // light in LightBuffer order, already widened by the light's own share of light_pad
    const float4 *light_group_boxes;  // the same around every 32 consecutive lights of a kind: the groups of the
                                      // triangles, then those of the spheres, then those of the quads
    DevSet world;   // sets[0]
    DevSet lights;  // sets[1]
    uint32_t n_materials;
    uint32_t n_light_total;  // lights.len() including instances
    uint32_t boxes_finite;   // every coordinate of every BVH box is a finite number
    uint32_t n_media;
    uint32_t n_moving, n_static_spheres;
    uint32_t medium_queue;   // with media: the shade queue that rays without a surface to shade go to
    float light_pad_flat, light_pad_sphere;  // light_boxes are widened by this * (largest |coordinate| of the ray origin)
                                             // when tested: triangles and quads, spheres
    uint32_t n_wide, n_tris, n_spheres, n_quads, n_insts, n_sets;  // array lengths (bounds-checking build)
};

struct DevCamera {
    uint32_t width, height;
    f3 center, pixel00, pixel_du, pixel_dv, disk_u, disk_v;
    float defocus_angle;
    uint32_t sqrt_spp;
    float inv_sqrt_spp;
    uint32_t max_depth;
    f3 background;
};

// hit2 encoding: bits 0..1 primitive type, bits 2..31 instance index + 1 (0 = not in an instance)
#define PTD_HIT_MISS 0xFFFFFFFFu

// One shade queue per material type, and one more for rays that end without a surface to shade
// when the scene has participating media (they may still scatter on their way out).
constexpr int PTD_MAT_MEDIUM = PTS_MAT_COUNT;
constexpr int MAX_QUEUES = PTS_MAT_COUNT + 1;

// Work queues of path slots, in chunks of QUEUE_CHUNK entries.  A producing wave owns one
// open chunk per queue it writes and appends to it without synchronisation; it takes a new
// chunk (one atomic on the queue's chunk count) when the open one cannot take the next
// push, and pads what it leaves unused with QUEUE_EMPTY.  A consumer walks
// chunks * QUEUE_CHUNK entries and skips the padding.  One atomic per ~QUEUE_CHUNK entries
// instead of one per wave push matters: atomics on one 64-byte line run at about 11 ns
// each on this part whatever the number of waves, which at one per 64 entries was the
// longest pole of the first bounces.
constexpr uint32_t QUEUE_CHUNK = 256;
constexpr uint32_t QUEUE_EMPTY = 0xFFFFFFFFu;

// Control words live one per 256-byte line (atomics on neighbouring lines do not serialise).
// Words with two copies are used in ping-pong by bounce parity so that a kernel can clear
// the copy the next kernel will fill.
constexpr uint32_t CTL_STRIDE = 64;  // in 32-bit words
constexpr uint32_t CTL_RAY_CHUNKS = 0;    // +parity: chunks in ray queue [parity]
constexpr uint32_t CTL_FINISH_COUNT = 2;  // +k: paths the k-th (mod 2) finishing scan of this path state has found (k_finish_scan)
constexpr uint32_t CTL_SHADE_CHUNKS = 4;  // +2*queue+parity: chunks in material queue [queue]
constexpr uint32_t BOUNCE_PROFILE_LEN = 1024;
constexpr uint32_t CTL_SHADOW_CHUNKS = 4 + 2 * MAX_QUEUES;  // +parity: chunks in the shadow-ray queue filled during a step of that parity
constexpr uint32_t CTL_SHADOW_CURSOR = 6 + 2 * MAX_QUEUES;  // +parity: unused (CTL_SHADOW_WALK_CURSOR)
constexpr uint32_t WORK_LISTS = 8;  // lists of work blocks, one per XCD: atomics of different XCDs go to different lines
constexpr uint32_t CTL_PACKET_CURSOR = 8 + 2 * MAX_QUEUES;  // +8*parity+list: blocks of list `list` of the step's new paths handed out so far
constexpr uint32_t CTL_SHADE_CURSOR = CTL_PACKET_CURSOR + 2 * WORK_LISTS;  // +8*(2*queue+parity)+list: chunk blocks of that list of material queue [queue] handed out
constexpr uint32_t CTL_WALK_CURSOR = CTL_SHADE_CURSOR + 2 * WORK_LISTS * MAX_QUEUES;  // +8*parity+list: blocks of that list of ray queue [parity] handed out so far
constexpr uint32_t CTL_SHADOW_WALK_CURSOR = CTL_WALK_CURSOR + 2 * WORK_LISTS;           // +8*parity+list: the same for the shadow-ray queue
constexpr uint32_t CTL_WORDS = (8 + 2 * MAX_QUEUES + 2 * WORK_LISTS + 2 * WORK_LISTS * MAX_QUEUES + 4 * WORK_LISTS) * CTL_STRIDE;

// A launch may carry paths of several batches at once (each at a different bounce).  The slot
// arrays are then cut into regions of region_cap slots, one batch per region, and a path's
// batch is found from its slot.
struct RegionParams {  // 32 B
    uint64_t seed_mixed;
    uint32_t stratum_first, n_strata;
    uint32_t unused[4];
};

// Path state.  A path is named by its slot: first slot of its batch's region + pixel * strata_in_batch + (stratum -
// batch_first_stratum); only its radiance is stored by slot.  What a path carries from one vertex to the next
// (ray, draw counter, depth, throughput) travels with its ray: it is stored by the POSITION of the ray in the ray
// queue of the step that traces it, two sets used in ping-pong by step parity, so the arrays are as long as the
// largest number of rays in flight, not as the number of paths started, and a step reads them nearly in order.
// Paths on their camera ray have no stored state at all (dev_kernels.h, FreshPaths).  The results of a step's
// closest-hit search are stored by position in the step's work list: the ray queue's positions first, then the
// step's fresh paths.
struct PathState {
    float4 *ray_o[2];   // [parity][position]: origin.xyz, rng draw counter (bits)
    float4 *ray_d[2];   // direction.xyz, remaining depth (bits)
    float4 *beta[2];    // path throughput so far; w: density with which the current ray was drawn (next-event
                        // estimation), else the path's time (scenes with moving spheres)
    uint32_t *ray_queue[2];             // [parity][position]: slot of the path; max_ray_chunks * QUEUE_CHUNK entries each
    float4 *hit;      // [work-list position]: t, u, v, global primitive index (bits)
    uint32_t *hit2;   // type | (instance+1) << 2, or PTD_HIT_MISS
    uint8_t *fresh_draws;  // [index among the step's fresh paths]: draws the camera ray consumed (0: not recorded), so that
                           // whoever needs the ray again can go straight to the draws that made it
    float4 *contrib;  // [slot]: radiance of the finished path (xyz); zero between uses
    uint32_t *shade_queue[MAX_QUEUES];  // work-list positions by material type; only the queues of types in use exist
    uint2 *finish_list;                 // (position, slot) of the paths a finishing scan has taken out of the ray queue; finish_cap entries
    uint32_t finish_cap;
    uint32_t *ctl;                      // CTL_WORDS control words
    uint32_t *bounce_rays;              // [BOUNCE_PROFILE_LEN] rays traced at each bounce of the batch ([0] unused); may be null
    uint32_t *next_rays;                // rays the shading of this step has queued for the next one so far; may be null
    uint32_t max_chunks;      // capacity of each material queue, in chunks
    uint32_t max_ray_chunks;  // capacity of each ray queue and of the shadow-ray queue (several launches fill them)
    unsigned long long *queue_overflow;  // entries dropped because a queue was full (never, unless the sizing is wrong)
    uint32_t n_queues;
    uint2 *overflow;           // traversal stack levels beyond LDS: [level][overflow_stride]
    uint32_t overflow_stride;  // lanes of the widest traversal launch
    uint32_t n_slots;
    int refill_below;          // a traversing wave with fewer busy lanes than this goes back for more rays
    int leaf_batch;            // lanes that must be waiting at a leaf before the wave runs a primitive test
    uint32_t shade_claim;      // shading kernels: 0 = wave w takes pieces w, w + waves, ...; n = waves draw blocks of n chunks as they get to them
    uint32_t shade_claim_from; // ... where the queue has at least this many chunks (a short queue is not worth eight counters)
    RegionParams *regions;     // [slots / region_cap] the batches the regions hold
    uint32_t region_cap;       // slots per region
    // next-event estimation (PT_OPT_INTEGRATOR = 1): a vertex may leave a shadow ray behind; if it reaches an
    // emitter, shadow_c times the emitted radiance is added to the path's contrib
    uint32_t nee;
    uint32_t moving;  // the scene has moving spheres: beta.w holds the path's time
    uint32_t *shadow_queue;                    // slot of the path each shadow ray belongs to
    float4 *shadow_o, *shadow_d, *shadow_c;    // [position in the shadow queue]: origin, direction, factor
};

}  // namespace ptd
