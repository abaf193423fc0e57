// sc_wave.h -- wave64 neighbour exchange as a single VALU instruction.
//
// A lane's float4 holds 4 consecutive x; the stencil needs the last component of the lane to the
// left and the first of the lane to the right.  __shfl_up/down compile to ds_bpermute_b32 (an LDS
// pipeline instruction, ~50+ cycles of latency on the dependent chain of a red-black update);
// gfx9-family DPP has full-wave shifts (wave_shr:1 / wave_shl:1) that do the same move in the VALU
// with no LDS round trip.  bound_ctrl makes the lane that has no neighbour receive 0.
#pragma once
#include <hip/hip_runtime.h>

namespace sc {

__device__ __forceinline__ float wave_from_left(float v)   // lane i <- lane i-1 ; lane 0 <- 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}

__device__ __forceinline__ float wave_from_right(float v)  // lane i <- lane i+1 ; lane 63 <- 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

} // namespace sc
