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

// Workgroups of a launch are dealt round-robin over the 8 XCDs (observed on MI355X, not promised by
// HIP: used for speed only), so linear workgroup i runs on the XCD that also runs i + 8, i + 16, ...
// Each XCD has its own 4 MiB L2.  xcd_tile() turns the linear id into a logical tile number such that
// every XCD works through ONE contiguous run of tiles: tiles that share halo rows / columns then share
// an L2, and the halo is fetched from the fabric once instead of once per neighbour.
__device__ __forceinline__ int xcd_tile(int i, int n)
{
    const int xcd = i & 7, local = i >> 3, per = n >> 3, rem = n & 7;
    return xcd * per + min(xcd, rem) + local;   // XCD k owns tiles [k*per + min(k,rem), ...) : per + (k < rem) of them
}

} // namespace sc
