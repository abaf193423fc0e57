// sc_mg_kernels.hip -- multigrid kernels for gfx950: general-coefficient red-black smoother
// for the coarse levels, residual field (double arithmetic), normalised-transpose restriction
// and bilinear prolongation with correction.  See MGDim in sc_common.h for the level geometry
// (uniform spacing except the last interval in each direction).
//
// No reference counterpart: the reference solves the Poisson system directly with a DST
// (seamlessClone_imp.cpp:1814-1896).  The multigrid converges to that system's solution for
// any ROI size; its components are checked against oracle/mg_np.py.
#include "sc_common.h"
#include "sc_wave.h"
#include "sc_mg_device.h"
#include "sc_fd_closed.h"

namespace sc {

// ---- per-size state of a new hierarchy, built on the device (the reference does the same with its tables: initDSTMatrix_kernel,
//      seamlessClone_imp.cpp:569-603) --------------------------------------------------------------------------------------
// All planes of the levels >= 1 zeroed by ONE launch (their rings and pads must be zero; a buffer reused from another ROI size
// holds that size's data): blockIdx.y = buffer, grid-stride 16-byte stores.
__global__ __launch_bounds__(256) void k_zero_multi(ZeroJobs z)
{
    float4 *__restrict__ p = reinterpret_cast<float4 *>(z.p[blockIdx.y]);
    const size_t n = z.n16[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_zero_multi(const ZeroJobs &z, hipStream_t s)
{
    if (z.count <= 0) return;
    hipLaunchKernelGGL(k_zero_multi, dim3(128, z.count), dim3(256), 0, s, z);
}

// The five matrices of the bottom kernel's direct solve (MGBottomArgs: Mx1 | My1T | My2T | Mx2 | Dinv, zero padded to nxp / nyp)
// from the closed-form eigenpairs of the two 1-D level operators (sc_fd_closed.h): one thread per eigenvalue (50 bisection
// steps), then every thread fills its share of the ~5 n^2 entries with one double sine each.  ~15 us on one CU, once per new
// ROI size, on the instance's second stream beside the first launches of the clone that needs it.
// mm != nullptr: additionally the same four matrices as the operands of k_mg_bottom_mm (below), each row-major float
// [NP][NP] with NP = 32, 64 or 96 (zero padded), in the orientation the product that uses it reads row by row:
//     AX1[i][x] = Vx^-1[i][x]   AX2[x][i] = Vx[x][i]   AY1[j][y] = Vy^-1[j][y]   AY2[y][j] = Vy[y][j]   Dinv[j][i].
// part / parts: this workgroup's share of the entries: four 256-thread workgroups per build (a 1024-thread one is held to 128 VGPRs
// and spills; it also needs a whole idle CU and waits behind the level-0 launches of both streams)
__device__ __forceinline__ void fd_build_block(float *__restrict__ m, int nx, int ny, int nxp, int nyp, float cwx, float dx, float cwy, float dy,
                                               unsigned char *__restrict__ mm, int NPX, int NPY, int part = 0, int parts = 1)
{
    __shared__ FdPair px[128], py[128];
    __shared__ double hx[128], hy[128];        // the components of the hyperbolic pair's vector, if the direction has one (fd_pair)
    const int t0 = threadIdx.x, stride = (int)blockDim.x * parts, t = part * (int)blockDim.x + t0;
    for (int q = t0; q < 256; q += (int)blockDim.x) {          // every workgroup needs all pairs: one thread per eigenvalue
        if (q < nx) px[q] = fd_pair(q, nx, (double)cwx, (double)dx, hx);
        if (q >= 128 && q - 128 < ny) py[q - 128] = fd_pair(q - 128, ny, (double)cwy, (double)dy, hy);
    }
    __syncthreads();
    // component i (1-based) of pair p's unnormalised vector: one sine for every lane, the hyperbolic pair's from LDS
    auto fd_component = [](const FdPair &p, const double *h, int i) -> double {
        if (p.kind == 2) return h[i - 1];
        const double v = fd_sinpi(i * p.ang);
        return (p.kind == 1 && !(i & 1)) ? -v : v;
    };
    const int nxx = nxp * nxp, nyy = nyp * nyp, total = m ? 2 * nxx + 2 * nyy + nxp * nyp : 0;      // m == nullptr: the matrix-core operands only
    for (int e = t; e < total; e += stride) {
        float v = 0.f;
        if (e < nxx) {                                   // Mx1[x][i] = Vx^-1[i][x] = q_i(x) ee_x
            const int x = e / nxp, i = e - x * nxp;
            if (x < nx && i < nx) v = (float)(fd_component(px[i], hx, x + 1) * px[i].inv_norm * (x == nx - 1 ? 1.0 / (double)cwx : 1.0));
        } else if (e < nxx + nyy) {                      // My1T[y][j] = Vy^-1[j][y]
            const int r = e - nxx, y = r / nyp, j = r - y * nyp;
            if (y < ny && j < ny) v = (float)(fd_component(py[j], hy, y + 1) * py[j].inv_norm * (y == ny - 1 ? 1.0 / (double)cwy : 1.0));
        } else if (e < nxx + 2 * nyy) {                  // My2T[j][y] = Vy[y][j] = q_j(y) / ee_y
            const int r = e - nxx - nyy, j = r / nyp, y = r - j * nyp;
            if (y < ny && j < ny) v = (float)(fd_component(py[j], hy, y + 1) * py[j].inv_norm);
        } else if (e < 2 * nxx + 2 * nyy) {              // Mx2[i][x] = Vx[x][i]
            const int r = e - nxx - 2 * nyy, i = r / nxp, x = r - i * nxp;
            if (x < nx && i < nx) v = (float)(fd_component(px[i], hx, x + 1) * px[i].inv_norm);
        } else {                                         // Dinv[j][i] = 1 / (ly_j + lx_i)
            const int r = e - 2 * nxx - 2 * nyy, j = r / nxp, i = r - j * nxp;
            if (j < ny && i < nx) v = (float)(1.0 / (py[j].lam + px[i].lam));
        }
        m[e] = v;
    }
    if (!mm) return;
    const int ex = NPX * NPX, ey = NPY * NPY;
    float *ax1 = reinterpret_cast<float *>(mm), *ax2 = ax1 + ex, *ay1 = ax2 + ex, *ay2 = ay1 + ey, *dinv = ay2 + ey;
    for (int e = t; e < 2 * ex + 2 * ey + NPX * NPY; e += stride) {
        if (e < 2 * ex) {                                // AX1[i][x] (e < ex) and AX2[x][i]
            const bool second = e >= ex;
            const int r = second ? e - ex : e, row = r / NPX, k = r - row * NPX;
            const int i = second ? k : row, x = second ? row : k;
            float v = 0.f;
            if (x < nx && i < nx) v = (float)(fd_component(px[i], hx, x + 1) * px[i].inv_norm * ((!second && x == nx - 1) ? 1.0 / (double)cwx : 1.0));
            (second ? ax2 : ax1)[r] = v;
        } else if (e < 2 * ex + 2 * ey) {                // AY1[j][y] and AY2[y][j]
            const int q = e - 2 * ex;
            const bool second = q >= ey;
            const int r = second ? q - ey : q, row = r / NPY, k = r - row * NPY;
            const int j = second ? k : row, y = second ? row : k;
            float v = 0.f;
            if (y < ny && j < ny) v = (float)(fd_component(py[j], hy, y + 1) * py[j].inv_norm * ((!second && y == ny - 1) ? 1.0 / (double)cwy : 1.0));
            (second ? ay2 : ay1)[r] = v;
        } else {
            const int r = e - 2 * ex - 2 * ey, j = r / NPX, i = r - j * NPX;
            dinv[r] = (j < ny && i < nx) ? (float)(1.0 / (py[j].lam + px[i].lam)) : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void k_fd_build(float *__restrict__ m, int nx, int ny, int nxp, int nyp, float cwx, float dx, float cwy, float dy,
                                                  unsigned char *__restrict__ mm, int NPX, int NPY)
{
    fd_build_block(m, nx, ny, nxp, nyp, cwx, dx, cwy, dy, mm, NPX, NPY, (int)blockIdx.x, (int)gridDim.x);
}

// the same for every member of a size class in ONE launch (blockIdx.x = member): the matrix-core operands only, each member's
// where its table entry says (RagMember::mm), from the geometry of its own level `lev` (the level solved directly)
__global__ __launch_bounds__(256) void k_fd_build_rag(const RagMember *__restrict__ rag, int lev)
{
    const RagMember &m = rag[blockIdx.x];
    const MGGeom g = m.g[lev];
    fd_build_block(nullptr, g.x.n, g.y.n, 0, 0, g.x.cw_last, g.x.d_last, g.y.cw_last, g.y.d_last, const_cast<unsigned char *>(m.mm), m.npx, m.npy,
                   (int)blockIdx.y, (int)gridDim.y);
}

void launch_fd_build_rag(const RagMember *rag, int members, int lev, hipStream_t s)
{
    hipLaunchKernelGGL(k_fd_build_rag, dim3(members, 4), dim3(256), 0, s, rag, lev);      // (16 workgroups per member: no faster -- the time was one thread's, sc_fd_closed.h -- and they crowd the level-1 launch beside them)
}

void launch_fd_build(float *mats, const MGGeom &g, int nxp, int nyp, hipStream_t s, unsigned char *mm, int NPX, int NPY)
{
    // four 256-thread workgroups (each finds all eigenpairs, then fills its quarter of the entries): no spills at 155 VGPRs, no scratch
    hipLaunchKernelGGL(k_fd_build, dim3(4), dim3(256), 0, s, mats, g.x.n, g.y.n, nxp, nyp, g.x.cw_last, g.x.d_last, g.y.cw_last, g.y.d_last, mm, NPX, NPY);
}

// ---- general red-black half sweep (levels >= 1; ring = 0) ---------------------------------
template <bool SOR>
__global__ __launch_bounds__(256) void k_rb_half_gen(Field U, Field F, int color, float omega, MGGeom g)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int W = U.W, H = U.H, P = U.pitch;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int y = blockIdx.y * 4 + wv;
    if (y < 1 || y > H - 2 || x >= P) return;
    float *__restrict__ row = U.at(c) + (size_t)y * P;
    const float4 c4 = *reinterpret_cast<const float4 *>(row + x);
    const float4 u4 = *reinterpret_cast<const float4 *>(row - P + x);
    const float4 d4 = *reinterpret_cast<const float4 *>(row + P + x);
    const float4 f4 = *reinterpret_cast<const float4 *>(F.at(c) + (size_t)y * P + x);
    float l = wave_from_left(c4.w), r = wave_from_right(c4.x);
    if (lane == 0) l = (x > 0) ? row[x - 1] : 0.f;
    if (lane == 63) r = (x + 4 < P) ? row[x + 4] : 0.f;
    const float cn = (y == g.y.n) ? g.y.cw_last : 1.0f;
    const float dy = (y == g.y.n) ? g.y.d_last : 2.0f;
    float4 o = c4;
    const int par = (x + y + color) & 1;
#define SC_GEN_UPD(dst, XI, L, R, UU, DD, FF)                                     \
    if ((XI) >= 1 && (XI) <= W - 2) {                                             \
        const float cw = ((XI) == g.x.n) ? g.x.cw_last : 1.0f;                    \
        const float dx = ((XI) == g.x.n) ? g.x.d_last : 2.0f;                     \
        const float gs = (((cw * (L) + (R)) + (cn * (UU) + (DD))) - (FF)) / (dx + dy); \
        dst = SOR ? (dst + omega * (gs - dst)) : gs;                              \
    }
    if (par == 0) {
        SC_GEN_UPD(o.x, x + 0, l, c4.y, u4.x, d4.x, f4.x)
        SC_GEN_UPD(o.z, x + 2, c4.y, c4.w, u4.z, d4.z, f4.z)
    } else {
        SC_GEN_UPD(o.y, x + 1, c4.x, c4.z, u4.y, d4.y, f4.y)
        SC_GEN_UPD(o.w, x + 3, c4.z, r, u4.w, d4.w, f4.w)
    }
#undef SC_GEN_UPD
    *reinterpret_cast<float4 *>(row + x) = o;
}

void launch_rb_half_gen(Field U, Field F, int color, float omega, MGGeom g, hipStream_t s)
{
    dim3 grid((U.W + 255) / 256, (U.H + 3) / 4, U.C);
    if (omega == 1.0f)
        hipLaunchKernelGGL(k_rb_half_gen<false>, grid, dim3(256), 0, s, U, F, color, omega, g);
    else
        hipLaunchKernelGGL(k_rb_half_gen<true>, grid, dim3(256), 0, s, U, F, color, omega, g);
}

// ---- fused residual + restriction (one pass over U and F, no residual field in HBM) ---------
// Fc = 4 * (row-normalised transpose of the interpolation) applied to R = F - A U.
// The residual is evaluated in double from the float32 values: float32 evaluation cancels
// catastrophically on smooth error (at 2048^2 it leaves ~1 grey level of smooth error invisible),
// the double evaluation is exact for float inputs.
// Coarse tile 64 x 8 per 256-thread block.  Phase 1 stages the fine U tile (+1 halo) in LDS with
// aligned float4 loads; phase 2 forms the fine residuals into a second LDS tile, reading F
// straight from HBM exactly once; phase 3 applies the normalised transposed interpolation.
constexpr int RR_CW = 64, RR_CH = 8;
constexpr int RR_FW = 2 * RR_CW + 8, RR_FH = 2 * RR_CH + 4;   // 136 x 20 fine values

__global__ __launch_bounds__(256) void k_residual_restrict(Field U, Field F, Field Fc, MGGeom g)
{
    __shared__ __attribute__((aligned(16))) float su[RR_FH][RR_FW];
    __shared__ float sr[RR_FH][RR_FW];
    const int c = blockIdx.z;
    const int I0 = blockIdx.x * RR_CW, J0 = blockIdx.y * RR_CH;   // coarse points I0+1.., J0+1..
    const int XS = 2 * I0, YS = 2 * J0;                             // fine origin of the tiles
    const int H = U.H, P = U.pitch;
    const float *__restrict__ u = U.at(c);
    for (int i = threadIdx.x; i < RR_FH * (RR_FW / 4); i += 256) {
        const int ry = i / (RR_FW / 4), q = i - ry * (RR_FW / 4);
        const int y = YS + ry, x = XS + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < H && x < P) v = *reinterpret_cast<const float4 *>(u + (size_t)y * P + x);
        *reinterpret_cast<float4 *>(&su[ry][4 * q]) = v;
    }
    __syncthreads();
    const float *__restrict__ f = F.at(c);
    for (int i = threadIdx.x; i < (RR_FH - 2) * (RR_FW - 6); i += 256) {
        const int ry = 1 + i / (RR_FW - 6), cx = 1 + (i - (ry - 1) * (RR_FW - 6));
        const int y = YS + ry, x = XS + cx;
        float res = 0.f;
        if (x <= g.x.n && y <= g.y.n) {
            const double cw = (x == g.x.n) ? (double)g.x.cw_last : 1.0, dx = (x == g.x.n) ? (double)g.x.d_last : 2.0;
            const double cn = (y == g.y.n) ? (double)g.y.cw_last : 1.0, dy = (y == g.y.n) ? (double)g.y.d_last : 2.0;
            const double s = ((cw * (double)su[ry][cx - 1] + (double)su[ry][cx + 1]) +
                              (cn * (double)su[ry - 1][cx] + (double)su[ry + 1][cx])) - (dx + dy) * (double)su[ry][cx];
            res = (float)((double)f[(size_t)y * P + x] - s);
        }
        sr[ry][cx] = res;
    }
    __syncthreads();
    const int li = threadIdx.x & 63, lj = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int Jl = lj + 4 * k;
        const int I = I0 + 1 + li, J = J0 + 1 + Jl;
        if (I > g.x.nc || J > g.y.nc) continue;
        float wx[4], wy[4], ix, iy;
        restrict_weights(g.x, I, wx, ix);
        restrict_weights(g.y, J, wy, iy);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float rowacc = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) rowacc += wx[b] * sr[2 * Jl + 1 + a][2 * li + 1 + b];
            acc += wy[a] * rowacc;
        }
        Fc.at(c)[(size_t)J * Fc.pitch + I] = 4.0f * (acc * (ix * iy));
    }
}

void launch_residual_restrict(Field U, Field F, Field Fc, MGGeom g, hipStream_t s)
{
    dim3 grid((g.x.nc + RR_CW - 1) / RR_CW, (g.y.nc + RR_CH - 1) / RR_CH, U.C);
    hipLaunchKernelGGL(k_residual_restrict, grid, dim3(256), 0, s, U, F, Fc, g);
}

// ---- prolongation + correction: Uf += P Uc on the fine interior ------------------------------
// Thread = 4 consecutive fine points of one row (float4 read-modify-write of Uf).  With MAXC the
// largest |correction| of the launch is reduced wave64 shuffle -> LDS -> one plain store per
// block into `partial`, folded by k_max_final (a single atomic word would serialise ~10^5 waves).
template <bool MAXC>
__global__ __launch_bounds__(256) void k_prolong_add(Field Uc, Field Uf, MGGeom g, float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 256 + 4 * lane;
    const int y = blockIdx.y * 4 + wv + 1;
    const int c = blockIdx.z;
    float m = 0.f;
    if (y <= g.y.n && x <= g.x.n) {
        int J0, J1;
        float wy0, wy1;
        interp_1d(g.y, y, J0, J1, wy0, wy1);
        const float *__restrict__ e0 = Uc.at(c) + (size_t)J0 * Uc.pitch;
        const float *__restrict__ e1 = Uc.at(c) + (size_t)J1 * Uc.pitch;
        float *up = Uf.at(c) + (size_t)y * Uf.pitch + x;
        float4 u4 = *reinterpret_cast<float4 *>(up);
        float cr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xi = x + k;
            cr[k] = 0.f;
            if (xi >= 1 && xi <= g.x.n) {
                int I0, I1;
                float wx0, wx1;
                interp_1d(g.x, xi, I0, I1, wx0, wx1);
                const float top = wx0 * e0[I0] + wx1 * e0[I1];
                const float bot = wx0 * e1[I0] + wx1 * e1[I1];
                cr[k] = wy0 * top + wy1 * bot;
            }
        }
        if (x + 0 >= 1 && x + 0 <= g.x.n) u4.x = u4.x + cr[0];
        if (x + 1 <= g.x.n) u4.y = u4.y + cr[1];
        if (x + 2 <= g.x.n) u4.z = u4.z + cr[2];
        if (x + 3 <= g.x.n) u4.w = u4.w + cr[3];
        *reinterpret_cast<float4 *>(up) = u4;
        if (MAXC) m = fmaxf(fmaxf(fabsf(cr[0]), fabsf(cr[1])), fmaxf(fabsf(cr[2]), fabsf(cr[3])));
    }
    if (MAXC) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __shared__ float red[4];
        if (lane == 0) red[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] =
                fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
}

__global__ __launch_bounds__(256) void k_max_final(const float *__restrict__ partial, int n, unsigned *__restrict__ out)
{
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *out = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// two lists in one launch (this cycle's maxima and the previous cycle's): block b folds list b into out[b]; an empty list
// gives -1 ("unknown")
__global__ __launch_bounds__(256) void k_max_final2(const float *__restrict__ pa, int na, const float *__restrict__ pb, int nb,
                                                    unsigned *__restrict__ out, const unsigned *__restrict__ flag)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = flag ? *flag : 0u;      // the solve's "a 16-bit store saturated" word rides along (AbortFlag)
    const float *__restrict__ partial = blockIdx.x ? pb : pa;
    const int n = blockIdx.x ? nb : na;
    float m = n > 0 ? 0.f : -1.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

void launch_max_final2(const float *d_a, int na, const float *d_b, int nb, unsigned *d_out2, hipStream_t s, const unsigned *flag)
{
    hipLaunchKernelGGL(k_max_final2, dim3(2), dim3(256), 0, s, d_a, na, d_b, nb, d_out2, flag);
}

void launch_max_final(const float *d_partial, int n, unsigned *d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, s, d_partial, n, d_out);
}

int prolong_blocks(int nx, int ny, int C) { return ((nx + 1 + 255) / 256) * ((ny + 3) / 4) * C; }

void launch_prolong_add(Field Uc, Field Uf, MGGeom g, float *d_partial, unsigned *d_maxcorr, hipStream_t s)
{
    dim3 grid((g.x.n + 1 + 255) / 256, (g.y.n + 3) / 4, Uf.C);
    if (d_partial && d_maxcorr) {
        hipLaunchKernelGGL(k_prolong_add<true>, grid, dim3(256), 0, s, Uc, Uf, g, d_partial);
        hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, s, d_partial, (int)(grid.x * grid.y * grid.z), d_maxcorr);
    } else {
        hipLaunchKernelGGL(k_prolong_add<false>, grid, dim3(256), 0, s, Uc, Uf, g, (float *)nullptr);
    }
}

// ---- bottom of the V-cycle in ONE launch, LDS resident ----------------------------------------
// Every level from `first` down whose planes fit the CU's LDS together (U and F per level; the
// residual is formed on the fly inside the restriction) is processed by one 1024-thread
// workgroup per channel: the top RHS is read from HBM once, all smoothing / restriction /
// coarsest solve / prolongation runs on LDS with one s_barrier per phase, and only the top
// correction is written back.  This collapses the ~100 tiny launches those levels would need.
// Index mapping inside the bottom kernel: a linear index over rows of power-of-two padded width
// (shift/mask instead of an integer division per point), and the four possible diagonals (regular / last column / last row /
// corner) are inverted once per call instead of dividing per point.
struct BtInv { float rr, lr, rl, ll; };   // 1/(dx+dy): regular, last col, last row, both
__device__ __forceinline__ BtInv bt_inv(const MGGeom &g)
{
    BtInv v;
    v.rr = 0.25f;
    v.lr = 1.0f / (g.x.d_last + 2.0f);
    v.rl = 1.0f / (2.0f + g.y.d_last);
    v.ll = 1.0f / (g.x.d_last + g.y.d_last);
    return v;
}

__device__ __forceinline__ int pow2_shift(int n) { return n <= 1 ? 0 : 32 - __clz(n - 1); }   // 2^shift >= n

// Phase boundary.  WAVE = the phase is executed by ONE wavefront (the tiny levels): DS operations
// of a wave are processed in order, so draining lgkmcnt and stopping compiler reordering is
// enough -- no s_barrier, which is what the tiny levels' ~45 phases would otherwise wait on.
template <bool WAVE>
__device__ __forceinline__ void bt_sync()
{
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

template <bool WAVE>
__device__ __forceinline__ void lds_rb_half(float *u, const float *f, int P, const MGGeom &g, int color, float omega,
                                            bool sor)
{
    const BtInv iv = bt_inv(g);
    const int hx = (g.x.n + 1) >> 1, sh = pow2_shift(hx), total = g.y.n << sh;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nthr = WAVE ? 64 : (int)blockDim.x;
    for (int i = tid; i < total; i += nthr) {
        const int y = 1 + (i >> sh);
        const int x = 1 + ((1 + y + color) & 1) + 2 * (i & ((1 << sh) - 1));
        if (x > g.x.n) continue;
        const bool ylast = (y == g.y.n), xlast = (x == g.x.n);
        const float cn = ylast ? g.y.cw_last : 1.0f, cw = xlast ? g.x.cw_last : 1.0f;
        const float inv = xlast ? (ylast ? iv.ll : iv.lr) : (ylast ? iv.rl : iv.rr);
        float *p = u + y * P + x;
        const float gs = (((cw * p[-1] + p[1]) + (cn * p[-P] + p[P])) - f[y * P + x]) * inv;
        *p = sor ? (*p + omega * (gs - *p)) : gs;
    }
    bt_sync<WAVE>();
}

// residual at one fine point (float: the levels in here are small and well conditioned; the
// double evaluation matters on the fine levels handled by k_cycle0 / k_residual_restrict)
__device__ __forceinline__ float lds_res(const float *u, const float *f, int P, int x, int y, const MGGeom &g)
{
    const float cw = (x == g.x.n) ? g.x.cw_last : 1.0f, dx = (x == g.x.n) ? g.x.d_last : 2.0f;
    const float cn = (y == g.y.n) ? g.y.cw_last : 1.0f, dy = (y == g.y.n) ? g.y.d_last : 2.0f;
    const float *p = u + y * P + x;
    const float s = ((cw * p[-1] + p[1]) + (cn * p[-P] + p[P])) - (dx + dy) * p[0];
    return f[y * P + x] - s;
}

template <bool WAVE>
__device__ __forceinline__ void lds_restrict(const float *u, const float *f, int P, float *fc, int Pc, const MGGeom &g)
{
    const int sh = pow2_shift(g.x.nc), total = g.y.nc << sh;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nthr = WAVE ? 64 : (int)blockDim.x;
    for (int i = tid; i < total; i += nthr) {
        const int J = 1 + (i >> sh), I = 1 + (i & ((1 << sh) - 1));
        if (I > g.x.nc) continue;
        {
            float wy[4], iy, wx[4], ix;
            restrict_weights(g.y, J, wy, iy);
            restrict_weights(g.x, I, wx, ix);
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int y = 2 * J - 1 + a;
                if (wy[a] == 0.f || y > g.y.n) continue;
                float rowacc = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int x = 2 * I - 1 + b;
                    if (wx[b] == 0.f || x > g.x.n) continue;
                    rowacc += wx[b] * lds_res(u, f, P, x, y, g);
                }
                acc += wy[a] * rowacc;
            }
            fc[J * Pc + I] = 4.0f * (acc * (ix * iy));
        }
    }
    bt_sync<WAVE>();
}

template <bool WAVE>
__device__ __forceinline__ void lds_prolong(const float *e, int Pc, float *u, int P, const MGGeom &g)
{
    const int sh = pow2_shift(g.x.n), total = g.y.n << sh;
    const int tid = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, nthr = WAVE ? 64 : (int)blockDim.x;
    for (int i = tid; i < total; i += nthr) {
        const int y = 1 + (i >> sh), x = 1 + (i & ((1 << sh) - 1));
        if (x > g.x.n) continue;
        {
            int J0, J1, I0, I1;
            float wy0, wy1, wx0, wx1;
            interp_1d(g.y, y, J0, J1, wy0, wy1);
            interp_1d(g.x, x, I0, I1, wx0, wx1);
            const float top = wx0 * e[J0 * Pc + I0] + wx1 * e[J0 * Pc + I1];
            const float bot = wx0 * e[J1 * Pc + I0] + wx1 * e[J1 * Pc + I1];
            u[y * P + x] += wy0 * top + wy1 * bot;
        }
    }
    bt_sync<WAVE>();
}

// levels [l0, L) of the V-cycle: descent, coarsest solve, ascent (level l0's own prolongation is the caller's)
// `lv` points at the level descriptors staged in LDS (indexing the kernel-argument array with a
// run-time level number costs a dependent scalar-memory round trip per field)
template <bool WAVE>
__device__ __forceinline__ void bt_subcycle(float *lds, const MGBottomLevel *lv, int L, int pre, int l0, int l1)
{
    for (int l = l0; l < l1 && l + 1 < L; ++l) {
        const MGBottomLevel v = lv[l];
        const MGBottomLevel w = lv[l + 1];
        for (int s = 0; s < pre; ++s) {
            lds_rb_half<WAVE>(lds + v.offU, lds + v.offF, v.pitch, v.g, 0, 1.0f, false);
            lds_rb_half<WAVE>(lds + v.offU, lds + v.offF, v.pitch, v.g, 1, 1.0f, false);
        }
        lds_restrict<WAVE>(lds + v.offU, lds + v.offF, v.pitch, lds + w.offF, w.pitch, v.g);
    }
}

template <bool WAVE>
__device__ __forceinline__ void bt_ascent(float *lds, const MGBottomLevel *lv, int L, int post, int l0, int l1)
{
    for (int l = l1 - 1; l >= l0; --l) {
        if (l + 1 >= L) continue;
        const MGBottomLevel v = lv[l];
        const MGBottomLevel w = lv[l + 1];
        lds_prolong<WAVE>(lds + w.offU, w.pitch, lds + v.offU, v.pitch, v.g);
        for (int s = 0; s < post; ++s) {
            lds_rb_half<WAVE>(lds + v.offU, lds + v.offF, v.pitch, v.g, 0, 1.0f, false);
            lds_rb_half<WAVE>(lds + v.offU, lds + v.offF, v.pitch, v.g, 1, 1.0f, false);
        }
    }
}

// ---- direct solve of one bottom level by fast diagonalisation (MGBottomArgs, sc_multigrid.cpp) ----
// out[r][c] = sum_k LT[k][r] * R[k][c]  for r < rows, c < cols (both multiples of 4, zero padded), every
// operand in LDS with both reads contiguous (the left operand is kept transposed).  A thread owns a
// 2 x 4 tile: per k one ds_read_b64 + one ds_read_b128 and 8 FMAs.  TRANS writes the tile transposed
// (the next product needs this result as ITS left operand); SCALE multiplies by scale[r][c].
template <bool TRANS, bool SCALE>
__device__ __forceinline__ void fd_product(const float *LT, int pL, const float *R, int pR, int K, int rows, int cols,
                                           float *out, int pOut, const float *scale)
{
    const int tc = cols >> 2, ntiles = (rows >> 1) * tc;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const int r0 = (t / tc) << 1, c0 = (t % tc) << 2;
        // packed FMAs (v_pk_fma_f32: two lanes of a float2 per instruction): 4 instead of 8 per k
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p00 = { 0.f, 0.f }, p02 = { 0.f, 0.f }, p10 = { 0.f, 0.f }, p12 = { 0.f, 0.f };
        const float *lp = LT + r0, *rp = R + c0;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const float2 l = *reinterpret_cast<const float2 *>(lp + k * pL);
            const float4 r = *reinterpret_cast<const float4 *>(rp + k * pR);
            const f2 r01 = { r.x, r.y }, r23 = { r.z, r.w }, lx = { l.x, l.x }, ly = { l.y, l.y };
            p00 = __builtin_elementwise_fma(lx, r01, p00); p02 = __builtin_elementwise_fma(lx, r23, p02);
            p10 = __builtin_elementwise_fma(ly, r01, p10); p12 = __builtin_elementwise_fma(ly, r23, p12);
        }
        float a00 = p00.x, a01 = p00.y, a02 = p02.x, a03 = p02.y, a10 = p10.x, a11 = p10.y, a12 = p12.x, a13 = p12.y;
        if (SCALE) {
            const float4 s0 = *reinterpret_cast<const float4 *>(scale + r0 * cols + c0);
            const float4 s1 = *reinterpret_cast<const float4 *>(scale + (r0 + 1) * cols + c0);
            a00 *= s0.x; a01 *= s0.y; a02 *= s0.z; a03 *= s0.w;
            a10 *= s1.x; a11 *= s1.y; a12 *= s1.z; a13 *= s1.w;
        }
        if (TRANS) {
            *reinterpret_cast<float2 *>(out + (c0 + 0) * pOut + r0) = make_float2(a00, a10);
            *reinterpret_cast<float2 *>(out + (c0 + 1) * pOut + r0) = make_float2(a01, a11);
            *reinterpret_cast<float2 *>(out + (c0 + 2) * pOut + r0) = make_float2(a02, a12);
            *reinterpret_cast<float2 *>(out + (c0 + 3) * pOut + r0) = make_float2(a03, a13);
        } else {
            *reinterpret_cast<float4 *>(out + r0 * pOut + c0) = make_float4(a00, a01, a02, a03);
            *reinterpret_cast<float4 *>(out + (r0 + 1) * pOut + c0) = make_float4(a10, a11, a12, a13);
        }
    }
    __syncthreads();
}

// Solves level `v` (RHS plane lds + v.offF, result into lds + v.offU) exactly.  mats: the five matrices
// already staged in LDS at fd; P0, P1: two nxp x nyp scratch planes behind them.
// GLOBAL (the solved level is the bottom's first): the right-hand side comes straight from its HBM plane `fg` (pitch gp) and
// the solution goes straight to `ug` -- the level has no LDS planes at all and the caller has already put F^T into P0.
template <bool GLOBAL>
__device__ __forceinline__ void fd_solve(float *lds, const MGBottomLevel &v, const float *fd, int nxp, int nyp,
                                         float *__restrict__ ug, int gp)
{
    const int nx = v.g.x.n, ny = v.g.y.n;
    const float *Mx1 = fd, *My1T = Mx1 + nxp * nxp, *My2T = My1T + nyp * nyp, *Mx2 = My2T + nyp * nyp, *Dinv = Mx2 + nxp * nxp;
    float *P0 = const_cast<float *>(Dinv) + nyp * nxp, *P1 = P0 + nxp * nyp;
    if (!GLOBAL) {
        const float *f = lds + v.offF;
        for (int i = threadIdx.x; i < nxp * nyp; i += blockDim.x) {          // P0 = F^T, zero padded
            const int x = i / nyp, y = i - x * nyp;
            P0[i] = (x < nx && y < ny) ? f[(y + 1) * v.pitch + x + 1] : 0.f;
        }
        __syncthreads();
    }
    fd_product<false, false>(P0, nyp, Mx1, nxp, nxp, nyp, nxp, P1, nxp, nullptr);      // G1 = F Vx^-T          [y][i]
    fd_product<false, true>(My1T, nyp, P1, nxp, nyp, nyp, nxp, P0, nxp, Dinv);         // G2 = (Vy^-1 G1) / (ly + lx)
    fd_product<true, false>(My2T, nyp, P0, nxp, nyp, nyp, nxp, P1, nyp, nullptr);      // G3^T = (Vy G2)^T      [i][y]
    fd_product<false, false>(P1, nyp, Mx2, nxp, nxp, nyp, nxp, P0, nxp, nullptr);      // U = G3 Vx^T           [y][x]
    if (GLOBAL) {
        for (int i = threadIdx.x; i < nxp * ny; i += blockDim.x) {          // rows of nxp: consecutive lanes, consecutive addresses
            const int y = i / nxp, x = i - y * nxp;
            if (x < nx) ug[(size_t)(y + 1) * gp + x + 1] = P0[i];
        }
        return;
    }
    float *u = lds + v.offU;
    for (int i = threadIdx.x; i < nx * ny; i += blockDim.x) {
        const int y = i / nx, x = i - y * nx;
        u[(y + 1) * v.pitch + x + 1] = P0[y * nxp + x];
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_mg_bottom(MGBottomArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ MGBottomLevel lv[MG_BOTTOM_MAX_LEVELS];
    const int c = blockIdx.x;
    const int L = a.nlevels;
    if (threadIdx.x < MG_BOTTOM_MAX_LEVELS) lv[threadIdx.x] = a.lv[threadIdx.x];
    const bool fd = a.fd_level >= 0;
    if (fd && a.fd_level == 0) {
        // The usual case: the bottom's first level is the one solved directly.  No level planes in LDS, nothing to zero:
        // matrices and the transposed, zero-padded right-hand side are fetched together, the solution goes straight to HBM.
        const int nxp = a.fd_nxp, nyp = a.fd_nyp;
        const int n4 = (int)(fd_mat_floats(nxp, nyp) >> 2);
        const float4 *__restrict__ src = reinterpret_cast<const float4 *>(a.fd_mats);
        float4 *dst = reinterpret_cast<float4 *>(lds + a.fd_off);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
        const MGBottomLevel &v = a.lv[0];
        const float *__restrict__ fg = a.Ftop.at(c);
        float *P0 = lds + a.fd_off + fd_mat_floats(nxp, nyp);
        for (int i = threadIdx.x; i < nxp * nyp; i += blockDim.x) {       // y fastest: P0 = F^T; reads stride the plane's rows (L2 resident)
            const int x = i / nyp, y = i - x * nyp;
            P0[i] = (x < v.g.x.n && y < v.g.y.n) ? fg[(size_t)(y + 1) * a.Ftop.pitch + x + 1] : 0.f;
        }
        __syncthreads();
        fd_solve<true>(lds, v, lds + a.fd_off, nxp, nyp, a.Utop.at(c), a.Utop.pitch);
        return;
    }
    const int nzero = fd ? a.fd_off : a.lds_floats;
    for (int i = threadIdx.x; i < nzero; i += blockDim.x) lds[i] = 0.f;   // zero corrections, rings, pads
    if (fd) {   // stage the direct solver's matrices (16-byte loads; the region is 16-byte aligned and a multiple of 4 floats)
        const int n4 = (int)(fd_mat_floats(a.fd_nxp, a.fd_nyp) >> 2);
        const float4 *__restrict__ src = reinterpret_cast<const float4 *>(a.fd_mats);
        float4 *dst = reinterpret_cast<float4 *>(lds + a.fd_off);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const MGBottomLevel t = lv[0];
    {   // top RHS: HBM -> LDS
        const float *__restrict__ fg = a.Ftop.at(c);
        float *f = lds + t.offF;
        const int sh = pow2_shift(t.g.x.n), total = t.g.y.n << sh;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int y = 1 + (i >> sh), x = 1 + (i & ((1 << sh) - 1));
            if (x <= t.g.x.n) f[y * t.pitch + x] = fg[(size_t)y * a.Ftop.pitch + x];
        }
        __syncthreads();
    }
    // levels [0, lb): all 16 waves with block barriers; levels [ls, L): wave 0 alone (<= 4 colour
    // points per lane), the other waves wait at ONE barrier for the whole sub-cycle
    if (fd) {
        // V-cycle levels [0, fd_level) with all waves, level fd_level solved exactly, nothing below it
        const int lf = a.fd_level;
        bt_subcycle<false>(lds, lv, lf + 1, a.pre, 0, lf);
        fd_solve<false>(lds, lv[lf], lds + a.fd_off, a.fd_nxp, a.fd_nyp, nullptr, 0);
        bt_ascent<false>(lds, lv, lf + 1, a.post, 0, lf);
    } else {
    int ls = L;
    while (ls > 0 && ((lv[ls - 1].g.x.n + 1) >> 1) * lv[ls - 1].g.y.n <= 256) --ls;
    const int lb = ls < L - 1 ? ls : L - 1;            // block-mode levels are [0, lb)
    bt_subcycle<false>(lds, lv, L, a.pre, 0, lb);
    const MGBottomLevel cv = lv[L - 1];
    if (ls < L) {
        if ((threadIdx.x >> 6) == 0) {
            bt_subcycle<true>(lds, lv, L, a.pre, lb, L - 1);
            for (int s = 0; s < a.coarse_sweeps; ++s) {
                lds_rb_half<true>(lds + cv.offU, lds + cv.offF, cv.pitch, cv.g, 0, cv.omega, true);
                lds_rb_half<true>(lds + cv.offU, lds + cv.offF, cv.pitch, cv.g, 1, cv.omega, true);
            }
            bt_ascent<true>(lds, lv, L, a.post, lb, L - 1);
        }
        __syncthreads();
    } else {                                            // even the coarsest level is large (thin ROIs)
        for (int s = 0; s < a.coarse_sweeps; ++s) {
            lds_rb_half<false>(lds + cv.offU, lds + cv.offF, cv.pitch, cv.g, 0, cv.omega, true);
            lds_rb_half<false>(lds + cv.offU, lds + cv.offF, cv.pitch, cv.g, 1, cv.omega, true);
        }
    }
    bt_ascent<false>(lds, lv, L, a.post, 0, lb);
    }
    {   // top correction: LDS -> HBM (interior; ring and pads of the global plane stay zero)
        float *__restrict__ ug = a.Utop.at(c);
        const float *u = lds + t.offU;
        const int sh = pow2_shift(t.g.x.n), total = t.g.y.n << sh;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int y = 1 + (i >> sh), x = 1 + (i & ((1 << sh) - 1));
            if (x <= t.g.x.n) ug[(size_t)y * a.Utop.pitch + x] = u[y * t.pitch + x];
        }
    }
}

// ---- the bottom's direct solve on the matrix cores (round 4) -------------------------------------------------------------------
// The usual case -- the bottom's FIRST level is the one solved directly (fd_level == 0), both sides at most 96 unknowns -- as four
// products on v_mfma_f32_32x32x2_f32 (float32 in, float32 accumulate: the arithmetic of the SIMD form, to rounding order):
//     G1 = F Vx^-T      G2 = (Vy^-1 G1) (.) Dinv      G3 = Vy G2      U = G3 Vx^T
// k_mg_bottom's form of the same products (every operand staged in LDS, 2 x 4 register tiles on 1024 threads) took 16-17 us per
// launch for one 63 x 63 level: 8 us of staging for 80 KB of matrices, then 4 x 2.5 us of LDS-bound inner products (DESIGN
// appendix A, round 3; its float32-MFMA attempt kept that staging and layout and gained nothing).  Here the matrices never touch
// LDS: k_fd_build wrote them row-major in the orientation each product reads, one wave owns one 32 x 32 output tile (NPY / 32 x
// NPX / 32 waves, <= 9) and loads ITS rows' halves straight into registers -- lane half h takes k in [h K/2, (h+1) K/2), which
// makes a lane's operands of all K/2 MFMAs one contiguous run --, everything requested at kernel entry together with the
// right-hand side.  Between products a tile passes through LDS once, written in the layout the next product reads row-wise
// (16-byte stores of four consecutive k as the next B operand; 4-byte stores as the last product's A operand), rows padded by
// 16 bytes against bank conflicts.  (A split-bf16 form -- three MFMAs per product at 16 times the rate -- was measured first:
// 2 us faster per launch, but its 2^-17 operand rounding times the 1 / lambda_min ~ 400 amplification of the low modes left a
// relative 5e-4 in the correction: outside what the cycle tests allow against the numpy specification.  Not kept.)
typedef float mm_f16 __attribute__((ext_vector_type(16)));
// C / D layout of the 32 x 32 tile: register q of lane (r, h) is row (q & 3) + 8 (q >> 2) + 4 h, column r

// NK floats: half h of row `row` of a row-major [.][2 NK] matrix
template <int NK>
__device__ __forceinline__ void mm_row_half(const float *__restrict__ m, int row, int h, float (&v)[NK])
{
    const float4 *p = reinterpret_cast<const float4 *>(m + (size_t)row * (2 * NK) + h * NK);
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) { const float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
template <int NK>
__device__ __forceinline__ void mm_lds_half(const unsigned char *buf, int idx, int h, float (&v)[NK])
{
    constexpr int RS = 2 * NK * 4 + 16;
    const float4 *p = reinterpret_cast<const float4 *>(buf + (size_t)idx * RS + h * NK * 4);
#pragma unroll
    for (int q = 0; q < NK / 4; ++q) { const float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
template <int NK>
__device__ __forceinline__ mm_f16 mm_product(const float (&a)[NK], const float (&b)[NK])
{
    mm_f16 c = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int s = 0; s < NK; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
    return c;
}
// the tile (tm, tn) as the B operand of a product that sums over its ROW index: [column][k = row], K = 2 NK
template <int NK>
__device__ __forceinline__ void mm_store_b(unsigned char *buf, const mm_f16 &d, int tm, int tn, int r, int h)
{
    constexpr int RS = 2 * NK * 4 + 16;
    unsigned char *col = buf + (size_t)(32 * tn + r) * RS;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(col + (32 * tm + 8 * g + 4 * h) * 4) = make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
}
// ... as the A operand of a product that sums over its COLUMN index: [row][k = column]
template <int NK>
__device__ __forceinline__ void mm_store_a(unsigned char *buf, const mm_f16 &d, int tm, int tn, int r, int h)
{
    constexpr int RS = 2 * NK * 4 + 16;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<float *>(buf + (size_t)(32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h) * RS + (32 * tn + r) * 4) = d[q];
}

// SKX, SKY: a side's padded size in units of 16 (2, 4 or 6: 32, 64 or 96 unknowns).  PRE: every matrix operand is requested at
// kernel entry (4 waves, a SIMD each: registers to spare); else each product's matrix right before it (9 waves share the file).
template <int SKX, int SKY>
__global__ __launch_bounds__((SKX / 2) * (SKY / 2) * 64) void k_mg_bottom_mm(MGBottomMM a)
{
    constexpr int NPX = 16 * SKX, NPY = 16 * SKY, TX = NPX / 32, KX = NPX / 2, KY = NPY / 2;
    constexpr bool PRE = SKX <= 4 && SKY <= 4;
    constexpr int RSX = NPX * 4 + 16, RSY = NPY * 4 + 16;
    constexpr int BUF = (NPX * RSY > NPY * RSX) ? NPX * RSY : NPY * RSX;
    __shared__ __attribute__((aligned(16))) unsigned char buf0[BUF], buf1[BUF];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tm = w / TX, tn = w % TX;
    const int nx = a.nx, ny = a.ny;
    const float *ax1 = reinterpret_cast<const float *>(a.mm), *ax2 = ax1 + NPX * NPX, *ay1 = ax2 + NPX * NPX, *ay2 = ay1 + NPY * NPY;
    const float *dinv = ay2 + NPY * NPY;
    float b1[KX], a2[KY], a3[KY], b4[KX];
    mm_row_half<KX>(ax1, 32 * tn + r, h, b1);
    if (PRE) { mm_row_half<KY>(ay1, 32 * tm + r, h, a2); mm_row_half<KY>(ay2, 32 * tm + r, h, a3); mm_row_half<KX>(ax2, 32 * tn + r, h, b4); }
    // right-hand side: row y of this wave's row tile, the lane half's columns
    const float *__restrict__ fg = a.Ftop.at(c);
    const int y = 32 * tm + r;
    float fv[KX];
#pragma unroll
    for (int s = 0; s < KX; ++s) {
        const int x = h * KX + s;
        const bool in = y < ny && x < nx;
        const float v = fg[(size_t)((in ? y : 0) + 1) * a.Ftop.pitch + (in ? x : 0) + 1];
        fv[s] = in ? v : 0.f;
    }
    float dv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) dv[q] = dinv[(size_t)(32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h) * NPX + 32 * tn + r];
    // ---- G1 = F Vx^-T
    mm_f16 acc = mm_product<KX>(fv, b1);
    if (!PRE) mm_row_half<KY>(ay1, 32 * tm + r, h, a2);
    mm_store_b<KY>(buf0, acc, tm, tn, r, h);
    __syncthreads();
    // ---- G2 = (Vy^-1 G1) (.) Dinv
    {
        float g[KY];
        mm_lds_half<KY>(buf0, 32 * tn + r, h, g);
        acc = mm_product<KY>(a2, g);
    }
    if (!PRE) mm_row_half<KY>(ay2, 32 * tm + r, h, a3);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= dv[q];
    mm_store_b<KY>(buf1, acc, tm, tn, r, h);
    __syncthreads();
    // ---- G3 = Vy G2
    {
        float g[KY];
        mm_lds_half<KY>(buf1, 32 * tn + r, h, g);
        acc = mm_product<KY>(a3, g);
    }
    if (!PRE) mm_row_half<KX>(ax2, 32 * tn + r, h, b4);
    mm_store_a<KX>(buf0, acc, tm, tn, r, h);
    __syncthreads();
    // ---- U = G3 Vx^T
    {
        float g[KX];
        mm_lds_half<KX>(buf0, 32 * tm + r, h, g);
        acc = mm_product<KX>(g, b4);
    }
    float *__restrict__ ug = a.Utop.at(c);
    const int x = 32 * tn + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int yy = 32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (yy < ny && x < nx) ug[(size_t)(yy + 1) * a.Utop.pitch + x + 1] = acc[q];
    }
}

// NP = 32, 64 or 96 per direction; false: not a shape this path serves
bool launch_mg_bottom_mm(const MGBottomMM &a, int NPX, int NPY, int C, hipStream_t s)
{
#define SC_MM(SX, SY) if (NPX == 16 * SX && NPY == 16 * SY) { hipLaunchKernelGGL((k_mg_bottom_mm<SX, SY>), dim3(C), dim3((SX / 2) * (SY / 2) * 64), 0, s, a); return true; }
    SC_MM(2, 2) SC_MM(2, 4) SC_MM(2, 6) SC_MM(4, 2) SC_MM(4, 4) SC_MM(4, 6) SC_MM(6, 2) SC_MM(6, 4) SC_MM(6, 6)
#undef SC_MM
    return false;
}

// ---- the level above the bottom AND the bottom in ONE launch (round 4) -----------------------------------------------------------
// A single clone's cycle spends three dependent launches on 2 % of a percent of its unknowns: pre-smoothing + residual +
// restriction of the level above the bottom ("A", 127 x 127 at a 2048^2 ROI: 5.4 us), the bottom's direct solve ("B", 63 x 63:
// 12 us) and A's prolongation + post-smoothing (5.9 us) -- each mostly dispatch, a round trip to memory for a right-hand side the
// previous launch has just written, and the end-of-kernel release.  Here one 512-thread workgroup per channel does all three:
//   * level A lives in REGISTERS: lane l of wave w owns columns 2l, 2l+1 of rows 16w .. 16w+15 (field coordinates, ring included:
//     A has at most 127 unknowns per side), left / right neighbours by full-wave DPP shifts, the rows above / below a wave's band
//     through 1 KB of LDS per wave and one barrier per half sweep;
//   * residual and restriction stay in registers too (coarse point (I, J) = (l, 8w + j) belongs to the lane that owns its fine
//     centre), B's right-hand side goes to LDS in the layout the first product reads;
//   * B is solved by k_mg_bottom_mm's four float32 MFMA products on the first (NPX / 32)(NPY / 32) <= 4 waves, whose matrix operands
//     were requested at kernel entry and arrive while A is being smoothed;
//   * B's solution is interpolated from LDS (ghost rule of sc_mg_device.h for the irregular last interval), A is post-smoothed
//     and written out.
// Same arithmetic per point as the launches it replaces (k_cycle0<.., GEN, ZEROIN>, k_mg_bottom_mm, k_rb_tb<.., PROLONG>).
// One red-black half sweep of the lane's 16 rows x 2 columns.  GEN = false: a wave all of whose rows are regular unknowns (1 <= y < ny;
// `top`: its first row is the ring row y = 0 and is left alone) -- six instructions per point, no per-row state.  GEN = true: the wave
// that holds the level's last row (coefficients cn, 1 / (dx + d_last)) and the rows beyond it: per row two selects on scalar masks
// (an inactive row's factor is 0: it stays 0), no branches either.
template <int COLOR, bool GEN>
__device__ __forceinline__ void tail_half(float2 (&u)[16], const float2 (&f)[16], const float2 hup, const float2 hdn, bool top, int y0, int ny,
                                          float cny_last, float cw0, float cw1, float inR0, float inR1, float inL0, float inL1)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int y = y0 + i;                       // wave-uniform
        const float2 up = (i == 0) ? hup : u[i - 1], dn = (i == 15) ? hdn : u[i + 1];
        if (GEN) {
            const bool act = y >= 1 && y <= ny, ylast = y == ny;
            const float cn = ylast ? cny_last : 1.0f;
            if (((i + COLOR) & 1) == 0) {           // (x + y + colour) even: the even column
                const float l = wave_from_left(u[i].y);
                const float inv = act ? (ylast ? inL0 : inR0) : 0.f;
                u[i].x = ((__builtin_fmaf(cw0, l, u[i].y) + __builtin_fmaf(cn, up.x, dn.x)) - f[i].x) * inv;
            } else {
                const float r = wave_from_right(u[i].x);
                const float inv = act ? (ylast ? inL1 : inR1) : 0.f;
                u[i].y = ((__builtin_fmaf(cw1, u[i].x, r) + __builtin_fmaf(cn, up.y, dn.y)) - f[i].y) * inv;
            }
        } else if (((i + COLOR) & 1) == 0) {
            const float l = wave_from_left(u[i].y);
            const float v = ((__builtin_fmaf(cw0, l, u[i].y) + (up.x + dn.x)) - f[i].x) * inR0;
            if (i > 0 || !top) u[i].x = v;
        } else {
            const float r = wave_from_right(u[i].x);
            const float v = ((__builtin_fmaf(cw1, u[i].x, r) + (up.y + dn.y)) - f[i].y) * inR1;
            if (i > 0 || !top) u[i].y = v;
        }
    }
}

// hx[i] = the residual of row y0 + i restricted along x into the lane's coarse column (fine centre 2 lane): weights 1/2, 1, wxa, wxb.
// The half sweep before this was the second colour's: its points ((x + y) odd) satisfy their equations to rounding and their
// residuals are taken as the zeros they are -- only the first colour's point of a row (column 2 lane in even rows, 2 lane + 1 in odd
// ones) is evaluated.
template <bool GEN>
__device__ __forceinline__ void tail_residual(const float2 (&u)[16], const float2 (&f)[16], const float2 hup, const float2 hdn, bool top, int y0, int ny,
                                              float cny_last, float dy_last, float cw0, float cw1, float dx0, float dx1, bool v0, bool v1,
                                              float wxa, float wxb, float (&hx)[16])
{
    const float ddR0 = dx0 + 2.0f, ddR1 = dx1 + 2.0f, ddL0 = dx0 + dy_last, ddL1 = dx1 + dy_last;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int y = y0 + i;
        const float2 up = (i == 0) ? hup : u[i - 1], dn = (i == 15) ? hdn : u[i + 1];
        const bool act = GEN ? (y >= 1 && y <= ny) : (i > 0 || !top), ylast = GEN && y == ny;
        const float cn = ylast ? cny_last : 1.0f;
        if ((i & 1) == 0) {
            const float l = wave_from_left(u[i].y);
            float r0 = f[i].x - __builtin_fmaf(-(ylast ? ddL0 : ddR0), u[i].x, __builtin_fmaf(cw0, l, u[i].y) + (GEN ? __builtin_fmaf(cn, up.x, dn.x) : up.x + dn.x));
            r0 = (act && v0) ? r0 : 0.f;
            hx[i] = __builtin_fmaf(wxb, wave_from_right(r0), r0);
        } else {
            const float rr = wave_from_right(u[i].x);
            float r1 = f[i].y - __builtin_fmaf(-(ylast ? ddL1 : ddR1), u[i].y, __builtin_fmaf(cw1, u[i].x, rr) + (GEN ? __builtin_fmaf(cn, up.y, dn.y) : up.y + dn.y));
            r1 = (act && v1) ? r1 : 0.f;
            hx[i] = __builtin_fmaf(wxa, r1, 0.5f * wave_from_left(r1));
        }
    }
}

// LDS of the launch for operand paddings NPX x NPY: buf0 | bufA (bufA: B's right-hand side, then the second product's result, then B's solution)
template <int SKX, int SKY>
struct TailLds {
    static constexpr int NPX = 16 * SKX, NPY = 16 * SKY, RSX = NPX * 4 + 16, RSY = NPY * 4 + 16;
    static constexpr int BUF = (NPX * RSY > NPY * RSX) ? NPX * RSY : NPY * RSX;
    static constexpr int PB = 65;                                           // pitch of B's solution plane (coarse ring coordinates 0 .. 64)
    static constexpr int BUFA = (BUF > 65 * PB * 4) ? BUF : 65 * PB * 4;
};

template <int SKX, int SKY, bool RAG>
__device__ __forceinline__ void tail_body(const MGTail &a, unsigned char *__restrict__ buf0, unsigned char *__restrict__ bufA,
                                          float2 (*edge)[8][2][64], float (*hedge)[2][64])
{
    constexpr int NPX = 16 * SKX, NPY = 16 * SKY, TX = NPX / 32, TY = NPY / 32, KX = NPX / 2, KY = NPY / 2;
    constexpr int RSX = NPX * 4 + 16, RSY = NPY * 4 + 16;
    constexpr int PB = TailLds<SKX, SKY>::PB;
    (void)RSY;
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tm = w / TX, tn = w % TX;
    const bool mmw = w < TX * TY;
    // measurement (sc_hip_time_tail_phases): the first thread of channel 0 leaves the shader clock at every phase boundary
#define SC_TAIL_STAMP(K) do { if (a.stamps && threadIdx.x == 0 && c == 0) a.stamps[K] = __builtin_readcyclecounter(); } while (0)
    SC_TAIL_STAMP(0);
    // RAG: a size class (RagMember, sc_common.h) -- this channel's member has its own level geometry, its own matrices (padded like
    // everybody's in the class) and its own row count of the right-hand side plane; strides are the class's
    MGGeom gl = a.g;
    const unsigned char *mmp = a.mm;
    int FH = a.F.H;
    if constexpr (RAG) {
        const RagMember &m = a.rag[c / 3];
        gl = m.g[a.lev]; mmp = m.mm; FH = m.lh[a.lev];
    }
    const MGGeom &g = gl;
    const int nx = g.x.n, ny = g.y.n, ncx = g.x.nc, ncy = g.y.nc;
    const float *ax1 = reinterpret_cast<const float *>(mmp), *ax2 = ax1 + NPX * NPX, *ay1 = ax2 + NPX * NPX, *ay2 = ay1 + NPY * NPY;
    const float *dinv = ay2 + NPY * NPY;
    float b1[KX], a2[KY], a3[KY], b4[KX], dv[16];
    // ---- level A: the lane's columns and their coefficients
    const int x0 = 2 * lane, y0 = 16 * w, P = a.F.pitch;
    const bool v0 = x0 >= 1 && x0 <= nx, v1 = x0 + 1 <= nx;
    const float cw0 = (x0 == nx) ? g.x.cw_last : 1.0f, cw1 = (x0 + 1 == nx) ? g.x.cw_last : 1.0f;
    const float dx0 = (x0 == nx) ? g.x.d_last : 2.0f, dx1 = (x0 + 1 == nx) ? g.x.d_last : 2.0f;
    const float inR0 = v0 ? 1.0f / (dx0 + 2.0f) : 0.f, inR1 = v1 ? 1.0f / (dx1 + 2.0f) : 0.f;             // an unknown that does not exist stays 0
    const float inL0 = v0 ? 1.0f / (dx0 + g.y.d_last) : 0.f, inL1 = v1 ? 1.0f / (dx1 + g.y.d_last) : 0.f;
    float2 u[16], f[16];
    float2 traw[16];                 // requested before the matrices: the first half sweep waits for these only
    {
        const float *__restrict__ fg = a.F.at(c);
        const int xc = min(x0, P - 2);
#pragma unroll
        for (int i = 0; i < 16; ++i) traw[i] = *reinterpret_cast<const float2 *>(fg + (size_t)min(y0 + i, FH - 1) * P + xc);
    }
    asm volatile("" ::: "memory");
    if (mmw) {
        mm_row_half<KX>(ax1, 32 * tn + r, h, b1);
        mm_row_half<KY>(ay1, 32 * tm + r, h, a2);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool yok = y0 + i >= 1 && y0 + i <= ny;
        f[i] = make_float2((yok && v0) ? traw[i].x : 0.f, (yok && v1) ? traw[i].y : 0.f);
        u[i] = make_float2(0.f, 0.f);
    }
    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }          // measurement only: the right-hand side (and the first matrices) have arrived
    SC_TAIL_STAMP(1);
    float2 hup = make_float2(0.f, 0.f), hdn = hup;
    int eb = 0;
    // a wave's rows: all regular unknowns (rows 1 .. ny - 1; wave 0's ring row aside), or with the last row / rows beyond it, or none
    const bool regular = y0 + 15 < ny, dead = y0 > ny, top = w == 0;
#define SC_TAIL_HALF(COL)                                                                                                             \
    if (regular) tail_half<COL, false>(u, f, hup, hdn, top, y0, ny, g.y.cw_last, cw0, cw1, inR0, inR1, inL0, inL1);                       \
    else if (!dead) tail_half<COL, true>(u, f, hup, hdn, top, y0, ny, g.y.cw_last, cw0, cw1, inR0, inR1, inL0, inL1)
    auto exchange = [&]() {
        edge[eb][w][0][lane] = u[0];
        edge[eb][w][1][lane] = u[15];
        __syncthreads();
        hup = (w > 0) ? edge[eb][w - 1][1][lane] : make_float2(0.f, 0.f);
        hdn = (w < 7) ? edge[eb][w + 1][0][lane] : make_float2(0.f, 0.f);
        eb ^= 1;
    };
    for (int s = 0; s < a.pre; ++s) {
        if (s == 0) {                  // from a zero correction the first half sweep is u = -f / (dx + dy) on its colour
            if (!dead) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int y = y0 + i;
                    if (i & 1) u[i].y = -f[i].y * ((y == ny) ? inL1 : inR1);      // f is 0 where the row or the column does not exist
                    else u[i].x = -f[i].x * ((y == ny) ? inL0 : inR0);
                }
            }
        } else {
            SC_TAIL_HALF(0);
        }
        exchange();
        SC_TAIL_HALF(1);
        exchange();
    }
    SC_TAIL_STAMP(2);
    // ---- residual, restricted along x into the lane's coarse column I = lane (fine centre x0), then along y
    {
        float hx[16];
        const float wxa = (lane == ncx) ? g.x.tw1 : 0.5f, wxb = (lane == ncx) ? g.x.tw2 : 0.0f;
        if (dead) {
#pragma unroll
            for (int i = 0; i < 16; ++i) hx[i] = 0.f;
        } else if (regular) {
            tail_residual<false>(u, f, hup, hdn, top, y0, ny, g.y.cw_last, g.y.d_last, cw0, cw1, dx0, dx1, v0, v1, wxa, wxb, hx);
        } else {
            tail_residual<true>(u, f, hup, hdn, top, y0, ny, g.y.cw_last, g.y.d_last, cw0, cw1, dx0, dx1, v0, v1, wxa, wxb, hx);
        }
        hedge[w][0][lane] = hx[0];
        hedge[w][1][lane] = hx[15];
        __syncthreads();
        const float hu = (w > 0) ? hedge[w - 1][1][lane] : 0.f, hd = (w < 7) ? hedge[w + 1][0][lane] : 0.f;
        const float fx = (lane == ncx) ? 2.0f * g.x.inv_last : 1.0f;
        const int colw = (lane == 0) ? 63 : lane - 1;              // B's unknown I - 1; the lane without a coarse point zeroes the padding column
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int J = 8 * w + j;
            const float wya = (J == ncy) ? g.y.tw1 : 0.5f, wyb = (J == ncy) ? g.y.tw2 : 0.0f, fy = (J == ncy) ? 2.0f * g.y.inv_last : 1.0f;
            const float m = (j == 0) ? hu : hx[j == 0 ? 0 : 2 * j - 1], pn = (j == 7) ? hd : hx[j == 7 ? 0 : 2 * j + 2];
            const float v = ((0.5f * m + hx[2 * j]) + wya * hx[2 * j + 1]) + wyb * pn;
            const bool in = lane >= 1 && lane <= ncx && J >= 1 && J <= ncy;
            const int roww = (J == 0) ? 63 : J - 1;
            if (roww < NPY && colw < NPX) *reinterpret_cast<float *>(bufA + (size_t)roww * RSX + colw * 4) = in ? v * (fx * fy) : 0.f;
        }
    }
    __syncthreads();
    SC_TAIL_STAMP(3);
    // ---- level B on the matrix cores (k_mg_bottom_mm's products; right-hand side from LDS, solution to LDS)
    mm_f16 acc;
    if (mmw) {
        float fv[KX];
        mm_lds_half<KX>(bufA, 32 * tm + r, h, fv);
        mm_row_half<KY>(ay2, 32 * tm + r, h, a3);                   // two products ahead: arrives behind the first product's MFMAs
        acc = mm_product<KX>(fv, b1);                               // G1 = F Vx^-T
#pragma unroll
        for (int q = 0; q < 16; ++q) dv[q] = dinv[(size_t)(32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h) * NPX + 32 * tn + r];
        mm_store_b<KY>(buf0, acc, tm, tn, r, h);
    }
    __syncthreads();
    SC_TAIL_STAMP(4);
    if (mmw) {
        float gq[KY];
        mm_lds_half<KY>(buf0, 32 * tn + r, h, gq);
        mm_row_half<KX>(ax2, 32 * tn + r, h, b4);
        acc = mm_product<KY>(a2, gq);                               // G2 = (Vy^-1 G1) (.) Dinv
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] *= dv[q];
        mm_store_b<KY>(bufA, acc, tm, tn, r, h);
    }
    __syncthreads();
    SC_TAIL_STAMP(5);
    if (mmw) {
        float gq[KY];
        mm_lds_half<KY>(bufA, 32 * tn + r, h, gq);
        acc = mm_product<KY>(a3, gq);                               // G3 = Vy G2
        mm_store_a<KX>(buf0, acc, tm, tn, r, h);
    }
    __syncthreads();
    SC_TAIL_STAMP(6);
    if (mmw) {
        float gq[KX];
        mm_lds_half<KX>(buf0, 32 * tm + r, h, gq);
        acc = mm_product<KX>(gq, b4);                               // U = G3 Vx^T
        float *ub = reinterpret_cast<float *>(bufA);
#pragma unroll
        for (int q = 0; q < 16; ++q) ub[(32 * tm + (q & 3) + 8 * (q >> 2) + 4 * h + 1) * PB + 32 * tn + r + 1] = acc[q];
    }
    __syncthreads();
    SC_TAIL_STAMP(7);
    // ---- A += P B (coarse rows 8w .. 8w + 8, columns lane and lane + 1), post-smoothing
    {
        const float *ub = reinterpret_cast<const float *>(bufA);
        const float gx = 2.0f * g.x.tw1 - 1.0f, gy = 2.0f * g.y.tw1 - 1.0f;
        float2 e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int J = 8 * w + j, Jr = min(max(J, 1), max(ncy, 1));
            const float E0 = ub[Jr * PB + lane], E1 = ub[Jr * PB + lane + 1], Em = ub[Jr * PB + max(lane - 1, 0)];
            const float e0 = (lane >= 1 && lane <= ncx) ? E0 : ((lane == ncx + 1 && lane >= 2) ? gx * Em : 0.f);   // second tail point: the ghost column itself
            const float e1 = (lane + 1 <= ncx) ? E1 : (lane == ncx ? gx * e0 : 0.f);
            const float rs = (J >= 1 && J <= ncy) ? 1.0f : (J == ncy + 1 ? gy : 0.0f);
            e[j] = make_float2(e0 * rs, (0.5f * e0 + 0.5f * e1) * rs);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int y = y0 + i;
            const bool act = y >= 1 && y <= ny;
            float2 cr = e[i >> 1];
            if (i & 1) cr = make_float2(0.5f * cr.x + 0.5f * e[(i >> 1) + 1].x, 0.5f * cr.y + 0.5f * e[(i >> 1) + 1].y);
            if (act && v0) u[i].x += cr.x;
            if (act && v1) u[i].y += cr.y;
        }
    }
    exchange();
    SC_TAIL_STAMP(8);
    for (int s = 0; s < a.post; ++s) {
        SC_TAIL_HALF(0);
        exchange();
        SC_TAIL_HALF(1);
        if (s + 1 < a.post) exchange();
    }
    SC_TAIL_STAMP(9);
    float *__restrict__ ug = a.U.at(c);
    if (x0 < a.U.pitch) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int y = y0 + i;
            if (y >= 1 && y <= ny) *reinterpret_cast<float2 *>(ug + (size_t)y * a.U.pitch + x0) = u[i];
        }
    }
    SC_TAIL_STAMP(10);
#undef SC_TAIL_STAMP
#undef SC_TAIL_HALF
}

template <int SKX, int SKY, bool RAG = false>
__global__ __launch_bounds__(512) void k_mg_tail(MGTail a)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf0[TailLds<SKX, SKY>::BUF], bufA[TailLds<SKX, SKY>::BUFA];
    __shared__ float2 edge[2][8][2][64];
    __shared__ float hedge[8][2][64];
    tail_body<SKX, SKY, RAG>(a, buf0, bufA, edge, hedge);
}

// A size class whose members' directly solved levels need DIFFERENT operand paddings (31 unknowns: 32, 33: 64 -- the boundary sits at ROI
// sizes around 1075 and 2110 per side): one launch, every channel takes the body of ITS member's paddings -- the instantiation its solo
// run launches, so the order of the matrix-core sums (a lane half takes k in [h K/2, (h + 1) K/2)) is the solo run's.  LDS for the largest.
__global__ __launch_bounds__(512) void k_mg_tail_any(MGTail a)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf0[TailLds<4, 4>::BUF], bufA[TailLds<4, 4>::BUFA];
    __shared__ float2 edge[2][8][2][64];
    __shared__ float hedge[8][2][64];
    const RagMember &m = a.rag[blockIdx.x / 3];
    const int px = m.npx, py = m.npy;           // block-uniform
    if (px == 32 && py == 32) tail_body<2, 2, true>(a, buf0, bufA, edge, hedge);
    else if (px == 32) tail_body<2, 4, true>(a, buf0, bufA, edge, hedge);
    else if (py == 32) tail_body<4, 2, true>(a, buf0, bufA, edge, hedge);
    else tail_body<4, 4, true>(a, buf0, bufA, edge, hedge);
}

// level A: at most 127 unknowns per side (g = its geometry, g.*.nc = level B's sizes <= 63); NPX / NPY: level B padded to 32 or 64
bool launch_mg_tail(const MGTail &a, int NPX, int NPY, int C, hipStream_t s)
{
    if (a.g.x.n > 127 || a.g.y.n > 127 || a.g.x.nc > 63 || a.g.y.nc > 63 || a.g.x.nc > NPX || a.g.y.nc > NPY) return false;      // (a size class: a.g holds the class's maxima)
    // a size class whose members' operand paddings differ: the four bodies in one kernel (244 VGPRs: 20 us where the lean form takes 16);
    // a class whose members all share one padding (most: the padding flips at ROI ~1075 and ~2110 per side) takes that body alone
    if (a.rag && !a.rag_uniform) { hipLaunchKernelGGL(k_mg_tail_any, dim3(C), dim3(512), 0, s, a); return true; }
#define SC_TL(SX, SY) if (NPX == 16 * SX && NPY == 16 * SY) { \
        if (a.rag) hipLaunchKernelGGL((k_mg_tail<SX, SY, true>), dim3(C), dim3(512), 0, s, a); \
        else hipLaunchKernelGGL((k_mg_tail<SX, SY, false>), dim3(C), dim3(512), 0, s, a); \
        return true; }
    SC_TL(2, 2) SC_TL(2, 4) SC_TL(4, 2) SC_TL(4, 4)
#undef SC_TL
    return false;
}

hipError_t mg_bottom_prepare()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_mg_bottom), hipFuncAttributeMaxDynamicSharedMemorySize,
                               MG_BOTTOM_LDS_BYTES);
}

void launch_mg_bottom(const MGBottomArgs &a, int C, hipStream_t s)
{
    hipLaunchKernelGGL(k_mg_bottom, dim3(C), dim3(1024), (size_t)a.lds_floats * sizeof(float), s, a);
}

} // namespace sc
