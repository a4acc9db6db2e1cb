// wave_ops.h — cross-lane primitives of a 64-lane wavefront built on DPP (no LDS traffic).
#pragma once
#include <hip/hip_runtime.h>

namespace bwams {

constexpr int NEG = -(1 << 28);

template <int CTRL, int RM, int BM>
__device__ __forceinline__ int dppi(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, BM, false);
}
// inclusive prefix max over the 64 lanes (6 DPP steps)
__device__ __forceinline__ int scan_max(int v) {
    // lanes without a valid source keep `old` = their own value, so no identity constant is needed
    v = max(v, dppi<0x111, 0xF, 0xF>(v, v));     // row_shr:1
    v = max(v, dppi<0x112, 0xF, 0xF>(v, v));     // row_shr:2
    v = max(v, dppi<0x114, 0xF, 0xF>(v, v));     // row_shr:4
    v = max(v, dppi<0x118, 0xF, 0xF>(v, v));     // row_shr:8
    v = max(v, dppi<0x142, 0xA, 0xF>(v, v));     // row_bcast:15 into rows 1 and 3
    v = max(v, dppi<0x143, 0xC, 0xF>(v, v));     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int lane_shr1(int v, int fill) { return dppi<0x138, 0xF, 0xF>(fill, v); }   // wave_shr:1


}  // namespace bwams
