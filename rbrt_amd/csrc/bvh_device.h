// bvh_device.h — GPU-side BVH construction (bvh_device.hip). Same output contract as the host builder (bvh.h): 4-wide
// nodes, <= kLeafMax-triangle leaves over exactly the triangles the reference's scan can return, per-child max |e1|*|e2|.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "device_types.h"

namespace rbrt {

struct DeviceMeshSoa {  // DEVICE pointers to the mesh's SoA arrays (n_total entries each)
    const float *v0x, *v0y, *v0z, *e1x, *e1y, *e1z, *e2x, *e2y, *e2z;
    const uint8_t* is_padding;
};

struct DeviceBvhResult {
    bool ok = false;            // false: nothing was built (tiny mesh, depth or node budget exceeded): use the host builder
    BvhNode4* d_nodes = nullptr;  // hipMalloc'ed, owned by the caller when ok
    uint32_t n_nodes = 0;
    uint32_t n_valid = 0;       // indexed triangles = records written to d_tris_out[0 .. n_valid)
    uint32_t max_depth = 0;     // deepest 4-wide node (root = 0)
    float max_e12 = 0.0f;
};

// Builds the BVH of one mesh on `stream` and writes its triangle records, in leaf order, to d_tris_out (room for
// 8*floor(n_total/8) records). Leaf links are absolute positions in the scene's triangle array: tri_base is the
// position of d_tris_out[0] in it. Synchronises the stream (the sizes of the later stages depend on counts).
// algo: 0 = parallel locally-ordered clustering (PLOC), 1 = binary radix tree over the Morton codes (LBVH).
hipError_t build_bvh_device(const DeviceMeshSoa& soa, uint32_t n_total, BvhTri* d_tris_out, uint32_t tri_base,
                            DeviceBvhResult* res, hipStream_t stream, int algo = 0);

// normals[i] = (nx[i], ny[i], nz[i], 0) on the device.
hipError_t device_normals(const float* d_nx, const float* d_ny, const float* d_nz, uint32_t n, Normal4* d_out, hipStream_t stream);

}  // namespace rbrt
