// sc_mg_device.h -- device helpers shared by the multigrid kernels: 1-D restriction weights and
// 1-D interpolation stencils for a level pair described by MGDim (sc_common.h).
#pragma once
#include "sc_common.h"

namespace sc {

// Coarse point I gathers fine points 2I-1, 2I, 2I+1 with weights 1/2, 1, 1/2; the last coarse
// point takes the (up to two) tail points with their interpolation weights instead.  `inv` is
// the reciprocal row sum (the restriction is the row-normalised transpose of the interpolation).
__device__ __forceinline__ void restrict_weights(const MGDim &d, int I, float w[4], float &inv)
{
    w[0] = 0.5f; w[1] = 1.0f;
    if (I < d.nc) { w[2] = 0.5f; w[3] = 0.0f; inv = 0.5f; }
    else          { w[2] = d.tw1; w[3] = d.tw2; inv = d.inv_last; }
}


// Interpolation stencil of fine point i: coarse indices I0, I1 and weights w0, w1.
__device__ __forceinline__ void interp_1d(const MGDim &d, int i, int &I0, int &I1, float &w0, float &w1)
{
    if (i <= 2 * d.nc) {
        if ((i & 1) == 0) { I0 = i >> 1; I1 = I0; w0 = 1.0f; w1 = 0.0f; }
        else { I0 = (i - 1) >> 1; I1 = I0 + 1; w0 = 0.5f; w1 = 0.5f; }
    } else {
        I0 = d.nc; I1 = d.nc; w1 = 0.0f;
        w0 = (i - 2 * d.nc == 1) ? d.tw1 : d.tw2;
    }
}


} // namespace sc
