// Device-side construction of the two-box BVH of one triangle buffer (bvh_build.hip).
//
// The host layer's binned-SAH builder (pathtracer_amd/host/pthost.cpp, after src/bvh.rs:295-516 of
// the reference) stays the one whose trees are compared with the reference's; this builder exists
// for meshes where those seconds of single-threaded host work are the longest step before the
// first sample.  It produces a different, valid tree over the same triangle boxes, so closest-hit
// distances are the same and only the winner among primitives hit at exactly the same distance
// may differ.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace ptd {

struct TriBvhResult {
    uint32_t root_ref;  // reference to the root: node 0, or a leaf reference for a handful of triangles
    uint32_t n_nodes;   // 64-byte nodes written (n - 1, or 0)
    uint32_t height;    // nodes on the longest root-to-leaf path, leaves included
    float rmin[3], rmax[3];
};

// tri_hot / tri_cold / tri_mat: the slices of the scene arrays that belong to the buffer (n
// triangles); they are reordered in place.  wide: room for 4 * (n - 1) float4.  order (device, n
// entries, may be null): what the caller calls each triangle (e.g. its index in the scene-wide arrays);
// permuted like the triangles, so order[i] afterwards names the triangle now at position i.
// Everything runs on `stream`; the call returns after the stream has finished.
int build_tri_bvh_device(hipStream_t stream, float4 *tri_hot, float4 *tri_cold, int32_t *tri_mat, uint32_t n, float4 *wide, uint32_t *order,
                         TriBvhResult *out, std::string &err);

}  // namespace ptd
