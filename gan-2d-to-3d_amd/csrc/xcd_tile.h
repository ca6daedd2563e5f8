// xcd_tile.h — XCD-aware workgroup -> logical tile mapping shared by the MFMA convolution kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace g2s {

// Workgroups are dealt round-robin to the 8 XCDs by their flattened id, and each XCD has its own L2.
// The M-tiles of one pixel tile all read the same im2col operand: give every XCD a CONTIGUOUS range
// of logical tiles (M-tile fastest), so that those workgroups share one L2 instead of fetching the
// operand once per XCD.  Returns the logical tile of this workgroup within its (y, z) grid row.
// n = workgroups per grid row, bx = index within the row, row = index of the row in the launch
// (rows follow each other in the flattened workgroup order the hardware deals to the XCDs).
__device__ __forceinline__ int xcd_logical_tile(const int n, const int bx, const int row) {
    const int off = (n * row) & 7;           // XCD of block 0 of this row
    const int x = (bx + off) & 7;            // XCD this workgroup runs on
    int start = 0;
    for (int xx = 0; xx < x; xx++) {         // tiles of this row owned by the XCDs before x
        const int f = (xx - off) & 7;
        start += f < n ? (n - f + 7) >> 3 : 0;
    }
    const int f = (x - off) & 7;             // first block of this row on XCD x
    return start + ((bx - f) >> 3);
}

__device__ __forceinline__ int xcd_logical_tile() {
    return xcd_logical_tile(gridDim.x, blockIdx.x, blockIdx.y + gridDim.y * blockIdx.z);
}

}  // namespace g2s
