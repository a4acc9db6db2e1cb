// wave_ops.h — cross-lane primitives of a 64-lane wavefront built on DPP (no LDS traffic).
#pragma once
#include <hip/hip_runtime.h>

namespace bwams {

constexpr int NEG = -(1 << 28);

template <int CTRL, int RM, int BM>
__device__ __forceinline__ int dppi(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, BM, false);
}
// inclusive prefix max over the 64 lanes: six v_max_i32 with a DPP-permuted first operand, in place.  Written as
// assembly because the compiler expands `max(v, update_dpp(v, v, ..))` into v_mov + v_mov_dpp + v_max (three VALU
// instructions and a wait state per step) instead of the fused form; lanes without a valid source (or outside
// row_mask) are not written, i.e. keep their own value, so no identity constant is needed.  A VALU write followed
// by a DPP read of the same register needs two wait states: the s_nop 1 in front of every step.
__device__ __forceinline__ int scan_max(int v) {
    asm("s_nop 4\n\t"         // also covers "VALU writes EXEC, then a DPP instruction" (5 wait states): the compiler cannot see into the asm
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int lane_shr1(int v, int fill) { return dppi<0x138, 0xF, 0xF>(fill, v); }   // wave_shr:1


// One value per wave from a global cursor: lane 0 advances it by `step`, every lane gets the old value.
// Deliberately out of line.  Inlined into a `for (;;)` with a `continue`, the lane-0 branch around the
// atomic was threaded through the loop's back edge (the other lanes "know" their ticket is the constant
// 0), which split the wave: lanes 1..63 re-ran the loop body with ticket 0 while lane 0 fetched the next
// one.  A call keeps the branch and the broadcast together.
__device__ __noinline__ inline unsigned long long wave_ticket(unsigned long long *cursor, unsigned long long step) {
    unsigned long long t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(cursor, step);
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) |
           (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)t);
}

}  // namespace bwams
