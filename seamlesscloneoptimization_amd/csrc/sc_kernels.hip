// sc_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the seamless-clone hot path:
// mask stage, fused pre-process, single-sweep Jacobi / red-black smoothers, residual norm and
// the fused post-process.  wave = 64 lanes throughout.  (Register-blocked multi-sweep kernels:
// sc_sweep_tb.hip; multigrid cycle kernels: sc_cycle0.hip, sc_mg_kernels.hip.)
//
// Reference behaviour (what, not how): seamlessClone-CUDA/seamlessClone_imp.cpp
//   mask stage   :892-1071     pre-process :1920-2018     post-process :2078-2103
// The iterative smoothers have no counterpart in the reference (it solves directly with a
// DST); their specification is SURVEY.md Appendix A.5 and the CPU oracle (oracle/sc_oracle.c).
//
// Compiled with -ffp-contract=off: every float expression below is evaluated exactly as
// written so the oracle can check the sweeps bit for bit.
#include "sc_common.h"
#include <hip/hip_fp16.h>
#include "sc_wave.h"
#include <limits.h>
#include <algorithm>
#include <stdlib.h>

namespace sc {

// ------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v; // valid in lane 0
}

// ------------------------------------------------------------------------------------------
// mask stage
// ------------------------------------------------------------------------------------------
// Bounding box of mask != 0 with the 1-px border treated as zero (the reference first zeroes
// the border of its device copy, seamlessClone_imp.cpp:967-976,989, then reduces with LDS +
// global atomics, :927-963).  Here: per-lane scan of BB_ROWS rows, wave64 shuffle reduction,
// LDS across the 4 waves, one set of 4 global atomics per block that saw a set pixel.
// rect = {x_min, x_max, y_min, y_max}, host-seeded with {mw-1, 0, mh-1, 0} (:1006).
// Each lane scans 16-byte chunks (uint4 loads from the 16-B aligned address at or below the
// row start; bytes outside [1, mw-2] are masked off) of BB_ROWS rows, so a 2050^2 mask is read
// by ~400 workgroups with 4 independent 16-B loads in flight per lane.
constexpr int BB_ROWS = 4;

__device__ __forceinline__ unsigned nonzero_bytes(unsigned w)
{
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u; // bit 7 of every non-zero byte
}

// part != nullptr: the workgroup leaves its four extrema there with a plain store (neutral values if it saw nothing) and touches
// no atomic; a second, tiny launch folds them (k_mask_bbox_fold_group).  Used for groups of masks: thousands of workgroups'
// atomics on a handful of words cost more than the scan (74 us for sixteen 2050^2 masks, 0.13 of the HBM peak) -- L2 caches
// of different XCDs do not see one another's updates, so a "look first" test rarely spares one.
// fold.parts != nullptr (a single mask, round 4): the same parts, and the LAST workgroup to finish folds them -- no second launch
// and ONE atomic per workgroup (its arrival ticket) instead of four on the same line: ~400 workgroups x 4 atomics x ~12 ns were
// 18 of the 27 us this launch took at 2050^2.  Parts are written through (agent-scope stores) and drained before the ticket,
// the last arriver reads them with agent-scope loads after its add has returned (MI355X_MICROARCH.md, valid forms: one lane
// signals for its workgroup's stores, the workgroup whose add came last reads).  It writes the rectangle to device memory (the
// output launches' RectGuard reads it there) AND straight into the host's pinned mailbox: no copy command in the stream.
__device__ __forceinline__ void mask_bbox_block(const uint8_t *__restrict__ mask, int mw, int mh, int mstep,
                                                const BboxFold &fold, int bx, int by, int *__restrict__ part = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = bx * 64 + lane;
    const int yb = (by * 4 + wave) * BB_ROWS;
    int minx = INT_MAX, maxx = -1, miny = INT_MAX, maxy = -1;
    // all BB_ROWS loads are issued before the first one is used: row and chunk indices are clamped to a
    // chunk that is valid to read (chunk 0 of row 1) and the result of a clamped load is masked off, instead
    // of skipping the load with a branch (which serialises the loads behind one another's latency)
    uint4 v[BB_ROWS];
    int xbs[BB_ROWS];
    bool ok[BB_ROWS];
#pragma unroll
    for (int r = 0; r < BB_ROWS; ++r) {
        const int y = yb + r;
        const bool yin = (y >= 1) && (y < mh - 1);
        const uint8_t *row = mask + (size_t)(yin ? y : min(1, mh - 1)) * mstep;
        const int a0 = (int)((uintptr_t)row & 15);
        const int xb = chunk * 16 - a0;               // x of byte 0 of this chunk
        ok[r] = yin && (xb < mw - 1) && (xb + 15 >= 1);
        xbs[r] = xb;
        v[r] = *reinterpret_cast<const uint4 *>(row + (ok[r] ? xb : -a0));
    }
#pragma unroll
    for (int r = 0; r < BB_ROWS; ++r) {
        if (!ok[r]) continue;
        const int y = yb + r, xb = xbs[r];
        const unsigned w[4] = { v[r].x, v[r].y, v[r].z, v[r].w };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned m = nonzero_bytes(w[k]);
            const int x0 = xb + 4 * k;
            if (x0 < 1 || x0 + 3 > mw - 2) {          // partial dword at the row ends: drop invalid bytes
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (x0 + b < 1 || x0 + b > mw - 2) m &= ~(0x80u << (8 * b));
            }
            if (m) {
                const int lo = x0 + ((__ffs((int)m) - 1) >> 3), hi = x0 + ((31 - __clz((int)m)) >> 3);
                minx = min(minx, lo); maxx = max(maxx, hi);
                miny = min(miny, y); maxy = max(maxy, y);
            }
        }
    }
    minx = wave_min_i(minx); maxx = wave_max_i(maxx);
    miny = wave_min_i(miny); maxy = wave_max_i(maxy);
    __shared__ int red[4][4];
    if (lane == 0) { red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            minx = min(minx, red[w][0]); maxx = max(maxx, red[w][1]);
            miny = min(miny, red[w][2]); maxy = max(maxy, red[w][3]);
        }
        if (part) {
            part[0] = minx; part[1] = maxx; part[2] = miny; part[3] = maxy;
        } else {
            int *mine = fold.parts + 4 * (by * fold.nbx + bx);
            __hip_atomic_store(&mine[0], minx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[1], maxx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[2], miny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&mine[3], maxy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The parts above are agent-scope stores (written through: they do not linger in this XCD's L2) and are DRAINED before the
            // ticket; the last arriver reads them with agent-scope loads once its add has returned (MI355X_MICROARCH.md: one lane
            // signals for its own stores, the workgroup whose add came last reads -- a valid form).  The language-level spelling -- a
            // RELEASE on the add, an ACQUIRE fence in the folding wave (rounds 5's advisory) -- compiles to buffer_wbl2 in every one of
            // the ~400 scan workgroups and buffer_inv in the folder: a write-back of the XCD's L2 while the same launch's tiles are
            // writing the fields through it, +7.8 us on the pre-process launch of a 2048^2 clone (29.1 -> 36.9 us, measured with both
            // trees on one box) for an ordering these eight words already have.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ticket = __hip_atomic_fetch_add(fold.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0][0] = (ticket == (unsigned)fold.nblocks - 1u) ? 1 : 0;       // (the waves' extrema in red[][] have been consumed above)
        }
    }
    if (part) return;
    __syncthreads();
    if (red[0][0] == 0 || wave != 0) return;
    // the last workgroup to arrive: every part is in memory (agent-scope loads below read past this XCD's L2)
    minx = INT_MAX; maxx = -1; miny = INT_MAX; maxy = -1;
    for (int i = lane; i < fold.nblocks; i += 64) {
        const int *p = fold.parts + 4 * i;
        minx = min(minx, __hip_atomic_load(&p[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        maxx = max(maxx, __hip_atomic_load(&p[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        miny = min(miny, __hip_atomic_load(&p[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        maxy = max(maxy, __hip_atomic_load(&p[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    minx = wave_min_i(minx); maxx = wave_max_i(maxx);
    miny = wave_min_i(miny); maxy = wave_max_i(maxy);
    if (lane == 0) {
        // the reference seeds its rectangle with {W-1, 0, H-1, 0} (seamlessClone_imp.cpp:1006) and reduces into it
        int r0 = mw - 1, r1 = 0, r2 = mh - 1, r3 = 0;
        if (maxx >= 0) { r0 = min(r0, minx); r1 = max(r1, maxx); r2 = min(r2, miny); r3 = max(r3, maxy); }
        fold.rect_dev[0] = r0; fold.rect_dev[1] = r1; fold.rect_dev[2] = r2; fold.rect_dev[3] = r3;
        fold.rect_host[0] = r0; fold.rect_host[1] = r1; fold.rect_host[2] = r2; fold.rect_host[3] = r3;
        *fold.counter = 0u;                        // ready for the next launch (ordered behind this one)
    }
}

__global__ __launch_bounds__(256) void k_mask_bbox(const uint8_t *__restrict__ mask, int mw, int mh, int mstep, BboxFold fold)
{
    mask_bbox_block(mask, mw, mh, mstep, fold, blockIdx.x, blockIdx.y);
}

int mask_bbox_blocks(int mw, int mh)
{
    const int chunks = (mw + 15 + 15) / 16;   // +15: a row may start up to 15 bytes into its first chunk
    return ((chunks + 63) / 64) * ((mh + 4 * BB_ROWS - 1) / (4 * BB_ROWS));
}

void launch_mask_bbox(const uint8_t *mask, int mw, int mh, int mstep, BboxFold fold, hipStream_t s)
{
    const int chunks = (mw + 15 + 15) / 16;
    dim3 grid((chunks + 63) / 64, (mh + 4 * BB_ROWS - 1) / (4 * BB_ROWS));
    fold.nbx = (int)grid.x; fold.nblocks = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(k_mask_bbox, grid, dim3(256), 0, s, mask, mw, mh, mstep, fold);
}

// Crop to the bounding box + three 3x3 erodes (seamlessClone_imp.cpp:1052-1062, kernel
// :892-925) fused into one pass.  Each reference pass outputs 255 iff all nine inputs are 255
// and forces the ROI frame to 0, so three passes equal: "ring(x,y) >= 3 and the 7x7 window is
// all 255".  Outputs with ring < 3 are 0 whatever the window holds, and for ring >= 3 the window
// lies inside the ROI, so nothing outside the ROI ever matters.
//
// Word formulation: a lane owns one 32-bit word (4 pixels) of the output column-wise and slides
// down ER_STRIP rows.  Per source row it fetches the 12 bytes x-3 .. x+8 (four aligned dwords +
// v_alignbyte, the ROI origin has arbitrary byte alignment), turns them into "== 255" flag bytes,
// ANDs the seven byte-shifted views (horizontal 7-window for all four pixels at once) and keeps the
// last seven such words in registers for the vertical AND.
constexpr int ER_STRIP = 8;        // rows per lane (+ 6 halo rows) of a SINGLE clone's erode: 16 made a 2048^2 erode 264 workgroups of 22 dependent-ish row loads each, 12.9 us against 9.9
constexpr int ER_STRIP_GROUP = 16; // ... of a group's (throughput, not latency: 22 / 16 source rows per output row instead of 14 / 8; 8 against 16 in-step: no difference beyond noise)

__device__ __forceinline__ unsigned is255_flags(unsigned w)   // 0x80 in every byte that equals 255
{
    const unsigned t = ~w;                                     // zero byte <=> source byte == 255
    return ~((((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t)) & 0x80808080u;
}

template <int STRIP>
__device__ __forceinline__ void mask_erode3_block(const uint8_t *__restrict__ mask, int mstep, size_t mask_bytes,
                                                  const Geo &g, uint8_t *__restrict__ M, int mpitch, int bx, int by)
{
    // aligned dwords that overlap the mask buffer [mask, mask + mask_bytes): nothing outside is touched
    const uintptr_t lo = (uintptr_t)mask & ~(uintptr_t)3, hi = ((uintptr_t)mask + mask_bytes - 1) & ~(uintptr_t)3;
    const int xw = bx * 64 + (threadIdx.x & 63);       // word column
    const int x = 4 * xw;
    const int ys = (by * 4 + (threadIdx.x >> 6)) * STRIP;
    if (x >= g.W || ys >= g.H) return;
    // per-pixel ring-in-x mask for the four bytes of this word
    unsigned xring = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (x + j >= 3 && x + j <= g.W - 4) xring |= 0x80u << (8 * j);
    auto hrow = [&](int y) -> unsigned {
        if (y < 0 || y >= g.H) return 0u;
        const uint8_t *s = mask + (size_t)(y + g.y0) * mstep + (g.x0 + x - 3);
        const int a = (int)((uintptr_t)s & 3);
        const unsigned *p = reinterpret_cast<const unsigned *>(s - a);
        unsigned d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uintptr_t q = (uintptr_t)(p + k);
            d[k] = (q >= lo && q <= hi) ? p[k] : 0u;
        }
        const unsigned d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
        const unsigned fa = is255_flags(__builtin_amdgcn_alignbyte(d1, d0, a));    // bytes x-3 .. x
        const unsigned fb = is255_flags(__builtin_amdgcn_alignbyte(d2, d1, a));    // bytes x+1 .. x+4
        const unsigned fc = is255_flags(__builtin_amdgcn_alignbyte(d3, d2, a));    // bytes x+5 .. x+8
        // byte j of the result = AND of flag bytes j .. j+6 of the 12-byte window
        return fa & __builtin_amdgcn_alignbyte(fb, fa, 1) & __builtin_amdgcn_alignbyte(fb, fa, 2) &
               __builtin_amdgcn_alignbyte(fb, fa, 3) & fb & __builtin_amdgcn_alignbyte(fc, fb, 1) &
               __builtin_amdgcn_alignbyte(fc, fb, 2);
    };
    // all STRIP + 6 horizontal words first (independent loads in flight together), then the
    // vertical 7-row ANDs
    unsigned hh[STRIP + 6];
#pragma unroll
    for (int k = 0; k < STRIP + 6; ++k) hh[k] = hrow(ys - 3 + k);
#pragma unroll
    for (int r = 0; r < STRIP; ++r) {
        const int y = ys + r;
        unsigned v = hh[r] & hh[r + 1] & hh[r + 2] & hh[r + 3] & hh[r + 4] & hh[r + 5] & hh[r + 6] & xring;
        if (y < 3 || y > g.H - 4) v = 0;
        v = (v >> 7) * 255u;                                    // 0x80 flags -> 0xff bytes
        if (y < g.H) *reinterpret_cast<unsigned *>(M + (size_t)y * mpitch + x) = v;
    }
}

__global__ __launch_bounds__(256) void k_mask_erode3(const uint8_t *__restrict__ mask, int mstep, size_t mask_bytes,
                                                     Geo g, uint8_t *__restrict__ M, int mpitch)
{
    mask_erode3_block<ER_STRIP>(mask, mstep, mask_bytes, g, M, mpitch, blockIdx.x, blockIdx.y);
}

void launch_mask_erode3(const uint8_t *mask, int mstep, int mask_rows, Geo g, uint8_t *M, int mpitch, hipStream_t s)
{
    dim3 grid(((g.W + 3) / 4 + 63) / 64, (g.H + 4 * ER_STRIP - 1) / (4 * ER_STRIP));
    // last row may be shorter than the step: count only what the caller guarantees
    const size_t bytes = (size_t)mstep * (mask_rows - 1) + (size_t)(g.x0 + g.W + 1);
    hipLaunchKernelGGL(k_mask_erode3, grid, dim3(256), 0, s, mask, mstep, bytes, g, M, mpitch);
}

// OpenCV's erode for masks that are not 0 / 255 (SC_FLAG_OPENCV_GREY_MASK): cv::seamlessClone erodes the ROI view of the
// zero-bordered mask with a 3 x 3 rectangle, 3 iterations (OpenCV 3.4.5, seamless_cloning_impl.cpp, computeDerivatives) -- a 7 x 7
// MINIMUM filter that reads zeros outside the bounding box.  The reference thresholds instead (:917, sum == 255 * 9): the
// two agree on 0 / 255 masks only.  Separable: 64 x 16 outputs per workgroup, the 70 x 22 source bytes staged in LDS (zero
// outside the ROI), horizontal minima, then vertical.  Not a hot kernel (one byte per pixel, a non-default path).
__global__ __launch_bounds__(256) void k_mask_erode_min7(const uint8_t *__restrict__ mask, int mstep, Geo g, uint8_t *__restrict__ M, int mpitch)
{
    __shared__ uint8_t in[22][72];
    __shared__ uint8_t hm[22][64];
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 22 * 70; i += 256) {
        const int ry = i / 70, rx = i - ry * 70;
        const int y = y0 - 3 + ry, x = x0 - 3 + rx;
        in[ry][rx] = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? mask[(size_t)(y + g.y0) * mstep + (g.x0 + x)] : (uint8_t)0;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (int r = ly; r < 22; r += 4) {
        unsigned m = 255u;
#pragma unroll
        for (int d = 0; d < 7; ++d) m = min(m, (unsigned)in[r][lx + d]);
        hm[r][lx] = (uint8_t)m;
    }
    __syncthreads();
    for (int r = ly; r < 16; r += 4) {
        unsigned m = 255u;
#pragma unroll
        for (int d = 0; d < 7; ++d) m = min(m, (unsigned)hm[r + d][lx]);
        const int y = y0 + r, x = x0 + lx;
        if (y < g.H && x < g.W) M[(size_t)y * mpitch + x] = (uint8_t)m;
    }
}

void launch_mask_erode_min7(const uint8_t *mask, int mstep, Geo g, uint8_t *M, int mpitch, hipStream_t s)
{
    hipLaunchKernelGGL(k_mask_erode_min7, dim3((g.W + 63) / 64, (g.H + 15) / 16), dim3(256), 0, s, mask, mstep, g, M, mpitch);
}

// The mask stage of a GROUP of clones (sc_hip_run_device_batch): both kernels are latency bound (a few hundred
// workgroups each), so the group's scans, and after the read-back its erodes, go out as one launch each; blockIdx.z
// picks the clone, whose parameters travel in the kernel arguments.
__global__ __launch_bounds__(256) void k_mask_bbox_group(MaskJobs t, int *__restrict__ parts)
{
    const MaskJob &j = t.j[blockIdx.z];
    const int chunks = (j.mw + 15 + 15) / 16;
    int *part = parts + 4 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    if ((int)blockIdx.x >= (chunks + 63) / 64 || (int)blockIdx.y >= (j.mh + 4 * BB_ROWS - 1) / (4 * BB_ROWS)) {   // block-uniform
        if (threadIdx.x == 0) { part[0] = INT_MAX; part[1] = -1; part[2] = INT_MAX; part[3] = -1; }
        return;
    }
    mask_bbox_block(j.mask, j.mw, j.mh, j.mstep, BboxFold(), blockIdx.x, blockIdx.y, part);
}

// one workgroup per mask: extrema of its `per` workgroup parts -- an empty mask gives the "empty" rectangle the host's seeds used to be
// (mw - 1, 0, mh - 1, 0: round 5 dropped their upload) --, to device memory (the splices' guards read it there) and straight into the
// host's pinned mailbox (no read-back copy: two copy commands less in front of the group's erode)
__global__ __launch_bounds__(256) void k_mask_bbox_fold_group(MaskJobs t, const int *__restrict__ parts, int per)
{
    const int *p = parts + 4 * (size_t)blockIdx.x * per;
    int minx = INT_MAX, maxx = -1, miny = INT_MAX, maxy = -1;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int4 v = *reinterpret_cast<const int4 *>(p + 4 * i);
        minx = min(minx, v.x); maxx = max(maxx, v.y); miny = min(miny, v.z); maxy = max(maxy, v.w);
    }
    minx = wave_min_i(minx); maxx = wave_max_i(maxx); miny = wave_min_i(miny); maxy = wave_max_i(maxy);
    __shared__ int red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = minx; red[wave][1] = maxx; red[wave][2] = miny; red[wave][3] = maxy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            minx = min(minx, red[w][0]); maxx = max(maxx, red[w][1]);
            miny = min(miny, red[w][2]); maxy = max(maxy, red[w][3]);
        }
        const MaskJob &j = t.j[blockIdx.x];
        if (maxx < 0) { minx = j.mw - 1; maxx = 0; miny = j.mh - 1; maxy = 0; }
        j.rect[0] = minx; j.rect[1] = maxx; j.rect[2] = miny; j.rect[3] = maxy;
        if (j.rect_host) { j.rect_host[0] = minx; j.rect_host[1] = maxx; j.rect_host[2] = miny; j.rect_host[3] = maxy; }
    }
}

template <int STRIP>
__global__ __launch_bounds__(256) void k_mask_erode3_group(MaskJobs t)
{
    const MaskJob &j = t.j[blockIdx.z];
    if ((int)blockIdx.x >= ((j.g.W + 3) / 4 + 63) / 64 || (int)blockIdx.y >= (j.g.H + 4 * STRIP - 1) / (4 * STRIP)) return;
    mask_erode3_block<STRIP>(j.mask, j.mstep, j.mask_bytes, j.g, j.M, j.mpitch, blockIdx.x, blockIdx.y);
}

static void mask_bbox_group_grid(const MaskJob *jobs, int cnt, int &gx, int &gy)
{
    gx = 1; gy = 1;
    for (int i = 0; i < cnt; ++i) {
        gx = std::max(gx, ((jobs[i].mw + 15 + 15) / 16 + 63) / 64);
        gy = std::max(gy, (jobs[i].mh + 4 * BB_ROWS - 1) / (4 * BB_ROWS));
    }
}

// ints of scratch launch_mask_bbox_group needs for the workgroup parts of n masks
size_t mask_bbox_group_parts(const MaskJob *jobs, int n)
{
    size_t tot = 0;
    for (int i0 = 0; i0 < n; i0 += MaskJobs::MAX) {
        const int cnt = std::min(n - i0, (int)MaskJobs::MAX);
        int gx, gy;
        mask_bbox_group_grid(jobs + i0, cnt, gx, gy);
        tot += 4 * (size_t)gx * gy * cnt;
    }
    return tot;
}

void launch_mask_bbox_group(const MaskJob *jobs, int n, hipStream_t s, int *parts)
{
    for (int i0 = 0; i0 < n; i0 += MaskJobs::MAX) {
        MaskJobs t{};
        const int cnt = std::min(n - i0, (int)MaskJobs::MAX);
        int gx, gy;
        mask_bbox_group_grid(jobs + i0, cnt, gx, gy);
        for (int i = 0; i < cnt; ++i) t.j[i] = jobs[i0 + i];
        hipLaunchKernelGGL(k_mask_bbox_group, dim3(gx, gy, cnt), dim3(256), 0, s, t, parts);
        hipLaunchKernelGGL(k_mask_bbox_fold_group, dim3(cnt), dim3(256), 0, s, t, (const int *)parts, gx * gy);
        parts += 4 * (size_t)gx * gy * cnt;
    }
}

void launch_mask_erode3_group(const MaskJob *jobs, int n, hipStream_t s)
{
    for (int i0 = 0; i0 < n; i0 += MaskJobs::MAX) {
        MaskJobs t{};
        const int cnt = std::min(n - i0, (int)MaskJobs::MAX);
        int gx = 1, gy = 1;
        for (int i = 0; i < cnt; ++i) {
            t.j[i] = jobs[i0 + i];
            // last row may be shorter than the step: count only what the caller guarantees
            t.j[i].mask_bytes = (size_t)t.j[i].mstep * (t.j[i].mh - 1) + (size_t)(t.j[i].g.x0 + t.j[i].g.W + 1);
            gx = std::max(gx, ((t.j[i].g.W + 3) / 4 + 63) / 64);
            gy = std::max(gy, (t.j[i].g.H + 4 * ER_STRIP_GROUP - 1) / (4 * ER_STRIP_GROUP));
        }
        hipLaunchKernelGGL(k_mask_erode3_group<ER_STRIP_GROUP>, dim3(gx, gy, cnt), dim3(256), 0, s, t);
    }
}

// ------------------------------------------------------------------------------------------
// fused pre-process: ROI crop + u8->f32 + forward-difference gradients of dst ROI and patch +
// mask blend + backward-difference divergence  (seamlessClone_imp.cpp:1920-2018 in one pass,
// no gdX/gdY round trip).  Writes the Dirichlet/initial field U0 = dst ROI and the
// un-folded stencil RHS F = lap (0 on the ring).  The reflect-101 branches of the reference
// (:1937,:1940,:1944,:1947) only feed gdX/gdY at the last column/row, which no interior
// divergence reads, so they vanish here.
// ------------------------------------------------------------------------------------------
// Tile = 128 x 8 pixels per 256-thread block (lane = 4 consecutive pixels of one row).  The interleaved u8 rows of both
// images (tile + 1-px halo) are staged in LDS with aligned dword loads -- the ROI origin has arbitrary byte alignment
// -- together with each row's byte offset of the first pixel.  The staged rows are read back
// as aligned dwords (6 per image for the centre row, 5 for the rows above / below) and re-aligned with v_alignbyte --
// the shift is the same for every lane of a row -- instead of one ds_read_u8 per byte (28 LDS reads per 4 pixels
// instead of 120); bytes become floats with v_cvt_f32_ubyteN.  The eroded mask is 0 / 255, so the blend
// (1 - m) a + m b is a select (bit-identical: m is exactly 0 or 1).  The fields are written as one 8- or 16-byte store
// per lane, channel and field.
constexpr int P4_TW = 128, P4_RPT = 2, P4_TH = 8 * P4_RPT;     // tile: 32 lanes x 4 pixels wide, 8 thread rows x P4_RPT rows high
constexpr int P4_ROWQ = (3 * (P4_TW + 2) + 15 + 4 + 15) / 16;   // 16-byte pieces per staged row: the row may start up to 15 bytes into its first piece, the re-alignment reads one word ahead
constexpr int P4_ROWD = 4 * P4_ROWQ;                            // the same in dwords

// Rows ty0-1 .. ty0+P4_TH of an interleaved 8-bit image, columns tx0-1 .. tx0+P4_TW, staged as 16-byte pieces starting at
// the 16-byte boundary at or below the first byte (org[ry] = offset of that byte in its row of LDS).  A piece that lies
// inside the ROI row is one 16-byte load; the pieces at the row's ends are fetched word by word under the same bounds
// test as before (nothing is read further than 3 bytes outside the ROI's row).
__device__ __forceinline__ void p4_stage(const uint8_t *__restrict__ img, int step, int W, int H, int tx0, int ty0,
                                         unsigned (*sm)[P4_ROWD], int *org)
{
    for (int i = threadIdx.x; i < (P4_TH + 2) * P4_ROWQ; i += 256) {
        const int ry = i / P4_ROWQ, q = i - ry * P4_ROWQ;
        const int y = ty0 - 1 + ry;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        int a = 0;
        if (y >= 0 && y < H) {
            const uint8_t *row = img + (size_t)y * step;
            const uint8_t *first = row + 3 * (tx0 - 1);
            a = (int)((uintptr_t)first & 15);
            const uint8_t *p = first - a + 16 * q;              // aligned piece q of this row
            if (p >= row && p + 16 <= row + 3 * W) {
                v = *reinterpret_cast<const uint4 *>(p);
            } else {
                unsigned w[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint8_t *pk = p + 4 * k;
                    if (pk + 3 >= row && pk < row + 3 * W) w[k] = *reinterpret_cast<const unsigned *>(pk);
                }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        if (q == 0) org[ry] = a;
        *reinterpret_cast<uint4 *>(&sm[ry][4 * q]) = v;
    }
}

// bytes of pixels x-1 .. x+4 (N = 5 words) or x-1 .. x+3 and a bit (N = 4 words) of a staged row, as re-aligned words
template <int N>
__device__ __forceinline__ void p4_window(const unsigned *row, int word0, int shift, unsigned (&w)[N])
{
    unsigned d[N + 1];
#pragma unroll
    for (int k = 0; k <= N; ++k) d[k] = row[word0 + k];
#pragma unroll
    for (int k = 0; k < N; ++k) w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], shift);
}

#define P4_BYTE(w, b) ((float)(((w)[(b) >> 2] >> (8 * ((b) & 3))) & 0xffu))     // -> v_cvt_f32_ubyteN

// HF / HU: the right-hand side (an integer in [-1020, 1020]) / the initial field (8-bit values) are stored as float16,
// exactly, at the same element pitch / plane size inside their buffers; the fused multigrid path reads them that way
// (sc_cycle0.hip: every launch reads F, the first one U0).
// GREY (SC_FLAG_OPENCV_GREY_MASK): the eroded mask is a grey value and the blend is OpenCV's for such masks -- patch gradient
// times M (1/255f) plus destination gradient times (255 - M)(1/255f) (Cloning::normalClone / evaluate, OpenCV 3.4.5) -- instead
// of the select; bit-identical to the select for M in {0, 255}.  Fields are float then (the right-hand side is no integer).
// ER (round 4): the tile forms the eroded mask itself -- k_mask_erode3's arithmetic (horizontal 7-windows of "== 255" flags per source
// row, vertical AND of seven of them, ring of three) on the tile's 16 + 1 rows and 32 + 1 words, through LDS -- writes it to M (a
// repeated pass or a retry reads it there) and takes its own words from LDS.  One launch less in front of a single clone's solve
// (the erode alone: 8.8 us at 2048^2, 10 at 1000^2).
template <bool HF, bool HU, bool GREY = false, bool ER = false>
__device__ __forceinline__ void preprocess_block(const uint8_t *__restrict__ body, int bstep,
                                                 const uint8_t *__restrict__ face, int fstep,
                                                 const uint8_t *__restrict__ M, int mpitch,
                                                 const Field &U0, const Field &F, int c0, int by, const BboxTask *er = nullptr)
{
    __shared__ __attribute__((aligned(16))) unsigned sb[P4_TH + 2][P4_ROWD], sp[P4_TH + 2][P4_ROWD];
    __shared__ int ob[P4_TH + 2], op[P4_TH + 2];
    __shared__ unsigned hw[ER ? P4_TH + 7 : 1][ER ? 34 : 1], ew[ER ? P4_TH + 1 : 1][ER ? 34 : 1];
    const int W = U0.W, H = U0.H;
    const int tx0 = blockIdx.x * P4_TW, ty0 = by * P4_TH;
    p4_stage(body, bstep, W, H, tx0, ty0, sb, ob);
    p4_stage(face, fstep, W, H, tx0, ty0, sp, op);
    if (ER) {
        // horizontal words of source rows ty0 - 4 .. ty0 + P4_TH + 2, word columns tx0 - 4 .. tx0 + P4_TW - 4 (the word left of the tile: mlb)
        const uint8_t *mask = er->mask;
        const int mstep = er->mstep, gx0 = er->g.x0, gy0 = er->g.y0;
        const uintptr_t lo = (uintptr_t)mask & ~(uintptr_t)3, hi = ((uintptr_t)mask + er->mask_bytes - 1) & ~(uintptr_t)3;
        for (int i = threadIdx.x; i < (P4_TH + 7) * 33; i += 256) {
            const int r = i / 33, wq = i - r * 33;
            const int y = ty0 - 4 + r, x = tx0 - 4 + 4 * wq;
            unsigned h = 0u;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const uint8_t *s = mask + (size_t)(y + gy0) * mstep + (gx0 + x - 3);
                const int a = (int)((uintptr_t)s & 3);
                const unsigned *p = reinterpret_cast<const unsigned *>(s - a);
                unsigned d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uintptr_t q = (uintptr_t)(p + k);
                    d[k] = (q >= lo && q <= hi) ? p[k] : 0u;
                }
                const unsigned fa = is255_flags(__builtin_amdgcn_alignbyte(d[1], d[0], a));
                const unsigned fb = is255_flags(__builtin_amdgcn_alignbyte(d[2], d[1], a));
                const unsigned fc = is255_flags(__builtin_amdgcn_alignbyte(d[3], d[2], a));
                h = fa & __builtin_amdgcn_alignbyte(fb, fa, 1) & __builtin_amdgcn_alignbyte(fb, fa, 2) &
                    __builtin_amdgcn_alignbyte(fb, fa, 3) & fb & __builtin_amdgcn_alignbyte(fc, fb, 1) &
                    __builtin_amdgcn_alignbyte(fc, fb, 2);
            }
            hw[r][wq] = h;
        }
        __syncthreads();
        // eroded words of rows ty0 - 1 .. ty0 + P4_TH - 1
        for (int i = threadIdx.x; i < (P4_TH + 1) * 33; i += 256) {
            const int r = i / 33, wq = i - r * 33;
            const int y = ty0 - 1 + r, x = tx0 - 4 + 4 * wq;
            unsigned v = hw[r][wq] & hw[r + 1][wq] & hw[r + 2][wq] & hw[r + 3][wq] & hw[r + 4][wq] & hw[r + 5][wq] & hw[r + 6][wq];
            unsigned xring = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x + j >= 3 && x + j <= W - 4) xring |= 0x80u << (8 * j);
            v &= xring;
            if (y < 3 || y > H - 4 || x < 0 || x >= W) v = 0u;
            v = (v >> 7) * 255u;
            ew[r][wq] = v;
            if (r >= 1 && wq >= 1 && y < H && x < W) *reinterpret_cast<unsigned *>(er->M_out + (size_t)y * mpitch + x) = v;   // the tile's own rows and words
        }
    }
    __syncthreads();
    const int lx = threadIdx.x & 31;
    const int x = tx0 + 4 * lx;
    if (x >= W) return;
#pragma unroll 1
    for (int rr = 0; rr < P4_RPT; ++rr) {
    const int ly = (threadIdx.x >> 5) + 8 * rr;
    const int y = ty0 + ly;
    if (y >= H) break;
    const int ry = ly + 1;
    // windows: staged column 0 is pixel tx0-1, so pixel x-1 sits at byte org + 12 lx of the row
    unsigned bc[5], bu[4], bd[4], pc[5], pu[4], pd[4];
    p4_window<5>(sb[ry], (ob[ry] >> 2) + 3 * lx, ob[ry] & 3, bc);
    p4_window<4>(sb[ry - 1], (ob[ry - 1] >> 2) + 3 * lx, ob[ry - 1] & 3, bu);
    p4_window<4>(sb[ry + 1], (ob[ry + 1] >> 2) + 3 * lx, ob[ry + 1] & 3, bd);
    p4_window<5>(sp[ry], (op[ry] >> 2) + 3 * lx, op[ry] & 3, pc);
    p4_window<4>(sp[ry - 1], (op[ry - 1] >> 2) + 3 * lx, op[ry - 1] & 3, pu);
    p4_window<4>(sp[ry + 1], (op[ry + 1] >> 2) + 3 * lx, op[ry + 1] & 3, pd);
    // eroded mask: pixels x .. x+3 of rows y and y-1 (aligned words: x % 4 == 0) and pixel x-1 of row y
    const bool yin = (y >= 1) && (y <= H - 2);
    unsigned mw = 0, muw = 0, mlb = 0;
    if (ER) {
        if (yin) { mw = ew[ly + 1][lx + 1]; muw = ew[ly][lx + 1]; mlb = ew[ly + 1][lx] >> 24; }
    } else if (yin) {
        mw = *reinterpret_cast<const unsigned *>(M + (size_t)y * mpitch + x);
        muw = *reinterpret_cast<const unsigned *>(M + (size_t)(y - 1) * mpitch + x);
        mlb = x > 0 ? M[(size_t)y * mpitch + x - 1] : 0u;
    }
    const size_t o = (size_t)y * U0.pitch + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float uv[4], lv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {              // pixel x + j = window pixel j + 1
            const int b = 3 * (j + 1) + c;
            const float bcc = P4_BYTE(bc, b);
            uv[j] = (x + j < W) ? bcc : 0.f;
            float lap = 0.0f;
            if (yin && x + j >= 1 && x + j <= W - 2) {
                const float bl = P4_BYTE(bc, b - 3), br = P4_BYTE(bc, b + 3), bup = P4_BYTE(bu, b), bdn = P4_BYTE(bd, b);
                const float pcc = P4_BYTE(pc, b), pl = P4_BYTE(pc, b - 3), pr = P4_BYTE(pc, b + 3), pup = P4_BYTE(pu, b), pdn = P4_BYTE(pd, b);
                const unsigned mb = (mw >> (8 * j)) & 0xffu, mlbyte = (j == 0 ? mlb : ((mw >> (8 * (j - 1))) & 0xffu)), mub = (muw >> (8 * j)) & 0xffu;
                if (GREY) {
                    const float k = 1.0f / 255.0f;
                    const float wm = (float)mb * k, wi = (float)(255u - mb) * k, wlm = (float)mlbyte * k, wli = (float)(255u - mlbyte) * k;
                    const float wum = (float)mub * k, wui = (float)(255u - mub) * k;
                    const float gx = (br - bcc) * wi + (pr - pcc) * wm;
                    const float gxl = (bcc - bl) * wli + (pcc - pl) * wlm;
                    const float gy = (bdn - bcc) * wi + (pdn - pcc) * wm;
                    const float gyu = (bcc - bup) * wui + (pcc - pup) * wum;
                    lap = (gx - gxl) + (gy - gyu);
                } else {
                const bool m = mb != 0u;
                const bool ml = mlbyte != 0u;
                const bool mu = mub != 0u;
                const float gx = m ? (pr - pcc) : (br - bcc);
                const float gxl = ml ? (pcc - pl) : (bcc - bl);
                const float gy = m ? (pdn - pcc) : (bdn - bcc);
                const float gyu = mu ? (pcc - pup) : (bcc - bup);
                lap = (gx - gxl) + (gy - gyu);
                }
            }
            lv[j] = lap;
        }
        if (HU) {
            const __half2 a = __floats2half2_rn(uv[0], uv[1]), b2 = __floats2half2_rn(uv[2], uv[3]);
            uint2 pk; pk.x = *reinterpret_cast<const unsigned *>(&a); pk.y = *reinterpret_cast<const unsigned *>(&b2);
            *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(U0.p) + (size_t)(c0 + c) * U0.plane + o) = pk;
        } else {
            *reinterpret_cast<float4 *>(U0.at(c0 + c) + o) = make_float4(uv[0], uv[1], uv[2], uv[3]);
        }
        if (HF) {
            const __half2 a = __floats2half2_rn(lv[0], lv[1]), b2 = __floats2half2_rn(lv[2], lv[3]);
            uint2 pk; pk.x = *reinterpret_cast<const unsigned *>(&a); pk.y = *reinterpret_cast<const unsigned *>(&b2);
            *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(F.p) + (size_t)(c0 + c) * F.plane + o) = pk;
        } else {
            *reinterpret_cast<float4 *>(F.at(c0 + c) + o) = make_float4(lv[0], lv[1], lv[2], lv[3]);
        }
    }
    }
}
#undef P4_BYTE

// BB: the launch also carries the bounding-box scan of the clone's mask, as extra rows of workgroups in front of the tile rows (round
// 4).  A clone launched on a PREDICTED box needs the scan's answer only at its very end (the output launches' RectGuard, the
// host's comparison), so the scan has no business on the critical path in front of the erode: it rides in this launch, which
// depends on nothing the scan produces.  mask_bbox_block's last workgroup folds the parts as before.
template <bool HF, bool HU, bool GREY = false, bool BB = false>
__global__ __launch_bounds__(256) void k_preprocess(const uint8_t *__restrict__ body, int bstep,
                                                     const uint8_t *__restrict__ face, int fstep,
                                                     const uint8_t *__restrict__ M, int mpitch,
                                                     Field U0, Field U1, Field F, BboxTask bb)
{
    if (BB && (int)blockIdx.y < bb.scan_rows) {          // block-uniform; the scan's workgroups come FIRST: its ticket / fold chain is over long before the tiles are
        const int b = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (b < bb.fold.nblocks) mask_bbox_block(bb.mask, bb.mw, bb.mh, bb.mstep, bb.fold, b % bb.fold.nbx, b / bb.fold.nbx);
        return;
    }
    // (a launch that carries the scan is a clone on a predicted box: its tiles erode the mask themselves, bb.g / bb.mask_bytes)
    if constexpr (BB) preprocess_block<HF, HU, GREY, true>(body, bstep, face, fstep, M, mpitch, U0, F, 0, (int)blockIdx.y - bb.scan_rows, &bb);
    else preprocess_block<HF, HU, GREY>(body, bstep, face, fstep, M, mpitch, U0, F, 0, (int)blockIdx.y);
}

// a group of clones in one launch: blockIdx.z = member, which owns channels 3z..3z+2 of the group's fields
template <bool HF, bool HU>
__global__ __launch_bounds__(256) void k_preprocess_group(ImageJobs t, int mpitch, Field U0, Field F)
{
    const ImageJob &j = t.j[blockIdx.z];
    if (j.W > 0) {      // a member of a size class: its own ROI inside the class's strides; tiles beyond it have nothing to do (block-uniform)
        if ((int)blockIdx.x * P4_TW >= j.W || (int)blockIdx.y * P4_TH >= j.H) return;
        U0.W = F.W = j.W; U0.H = F.H = j.H;
    }
    preprocess_block<HF, HU>(j.body_org, j.bstep, j.face_org, j.fstep, j.M, mpitch, U0, F, 3 * blockIdx.z, (int)blockIdx.y);
}

void launch_preprocess_group(const ImageJob *jobs, int n, int mpitch, Field U0, Field F, hipStream_t s, bool f_half, bool u_half)
{
    for (int i0 = 0; i0 < n; i0 += ImageJobs::MAX) {
        ImageJobs t{};
        const int cnt = std::min(n - i0, (int)ImageJobs::MAX);
        for (int i = 0; i < cnt; ++i) t.j[i] = jobs[i0 + i];
        Field u = U0, f = F;      // this launch's first member owns channel 3 i0
        u.p = u_half ? reinterpret_cast<float *>(reinterpret_cast<uint16_t *>(U0.p) + (size_t)3 * i0 * U0.plane) : U0.p + (size_t)3 * i0 * U0.plane;
        f.p = f_half ? reinterpret_cast<float *>(reinterpret_cast<uint16_t *>(F.p) + (size_t)3 * i0 * F.plane) : F.p + (size_t)3 * i0 * F.plane;
        dim3 g4((U0.W + P4_TW - 1) / P4_TW, (U0.H + P4_TH - 1) / P4_TH, cnt);
        if (f_half && u_half) hipLaunchKernelGGL((k_preprocess_group<true, true>), g4, dim3(256), 0, s, t, mpitch, u, f);
        else if (f_half) hipLaunchKernelGGL((k_preprocess_group<true, false>), g4, dim3(256), 0, s, t, mpitch, u, f);
        else hipLaunchKernelGGL((k_preprocess_group<false, false>), g4, dim3(256), 0, s, t, mpitch, u, f);
    }
}

void launch_preprocess(const uint8_t *body_org, int bstep, const uint8_t *face_org, int fstep,
                       const uint8_t *M, int mpitch, Field U0, Field U1, Field F, hipStream_t s, bool f_half, bool u_half, bool grey,
                       const BboxTask *scan)
{
    dim3 g4((U0.W + P4_TW - 1) / P4_TW, (U0.H + P4_TH - 1) / P4_TH);
    BboxTask bb{};
    if (scan && !grey) {                       // the mask's bounding-box scan as extra rows of workgroups
        bb = *scan;
        const int chunks = (bb.mw + 15 + 15) / 16;
        bb.fold.nbx = (chunks + 63) / 64;
        bb.fold.nblocks = bb.fold.nbx * ((bb.mh + 4 * BB_ROWS - 1) / (4 * BB_ROWS));
        bb.scan_rows = (bb.fold.nblocks + (int)g4.x - 1) / (int)g4.x;
        g4.y += (unsigned)bb.scan_rows;
        if (f_half && u_half) hipLaunchKernelGGL((k_preprocess<true, true, false, true>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
        else if (f_half) hipLaunchKernelGGL((k_preprocess<true, false, false, true>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
        else hipLaunchKernelGGL((k_preprocess<false, false, false, true>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
        return;
    }
    if (grey) hipLaunchKernelGGL((k_preprocess<false, false, true>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
    else if (f_half && u_half) hipLaunchKernelGGL((k_preprocess<true, true>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
    else if (f_half) hipLaunchKernelGGL((k_preprocess<true, false>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
    else hipLaunchKernelGGL((k_preprocess<false, false>), g4, dim3(256), 0, s, body_org, bstep, face_org, fstep, M, mpitch, U0, U1, F, bb);
}

// Device-to-device refresh of the destinations of a group of clones (sc_batch_job.body_restore) in ONE launch: sixteen separate
// 16 MB copies run at 2.7 TB/s (each too short to fill the chip, 3 % of a bench step); one launch over all of them streams.
// 16-byte aligned pointers, sizes in bytes (a tail below 16 bytes is copied bytewise).
__global__ __launch_bounds__(256) void k_copy_group(CopyJobs t)
{
    const int m = blockIdx.y;
    const uint4 *__restrict__ s4 = reinterpret_cast<const uint4 *>(t.src[m]);
    uint4 *__restrict__ d4 = reinterpret_cast<uint4 *>(t.dst[m]);
    const size_t n16 = t.bytes[m] >> 4;
    const size_t T = (size_t)gridDim.x * 256;              // four fully coalesced 16-byte loads in flight per lane
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * T) {
        const size_t i1 = i + T, i2 = i + 2 * T, i3 = i + 3 * T;
        const uint4 a = s4[i], b = s4[i1 < n16 ? i1 : i], c = s4[i2 < n16 ? i2 : i], d = s4[i3 < n16 ? i3 : i];
        d4[i] = a;
        if (i1 < n16) d4[i1] = b;
        if (i2 < n16) d4[i2] = c;
        if (i3 < n16) d4[i3] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (t.bytes[m] & 15)) {
        const size_t o = (n16 << 4) + threadIdx.x;
        reinterpret_cast<uint8_t *>(t.dst[m])[o] = reinterpret_cast<const uint8_t *>(t.src[m])[o];
    }
}

void launch_copy_group(const CopyJobs &t, int n, hipStream_t s)
{
    size_t mx = 0;
    for (int i = 0; i < n; ++i) mx = std::max(mx, t.bytes[i]);
    const unsigned gx = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (mx / 16 + 1023) / 1024));
    hipLaunchKernelGGL(k_copy_group, dim3(gx, n), dim3(256), 0, s, t);
}

// float16 right-hand side (left by a multigrid clone) -> float, into another buffer; used only when a
// diagnostic hook wants to read F after such a clone
__global__ __launch_bounds__(256) void k_half_to_float(const __half *__restrict__ src, float *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = __half2float(src[i]);
}

void launch_half_to_float(const void *src_half, float *dst, size_t n, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(k_half_to_float, dim3(blocks), dim3(256), 0, s, (const __half *)src_half, dst, n);
}

// N output bytes of one lane (all interior pixels) to a row whose bytes are NOT word aligned (R = address mod 4, 1..3: three of
// four ROI positions): the bytes up to the first boundary, words, the bytes behind the last one.  Only the group splice takes this
// (sixteen 1050^2 clones: 46-61 us bytewise against 27 aligned); a single clone's splice is latency bound and keeps the byte path.
template <int N, int R>
__device__ __forceinline__ void store_run_at(uint8_t *__restrict__ b, const unsigned char (&px)[N])
{
    constexpr int head = 4 - R, nd = (N - head) / 4;
#pragma unroll
    for (int k = 0; k < head; ++k) b[k] = px[k];
    unsigned *d32 = reinterpret_cast<unsigned *>(b + head);
#pragma unroll
    for (int k = 0; k < nd; ++k)
        d32[k] = px[head + 4 * k] | (px[head + 4 * k + 1] << 8) | (px[head + 4 * k + 2] << 16) | ((unsigned)px[head + 4 * k + 3] << 24);
#pragma unroll
    for (int k = head + 4 * nd; k < N; ++k) b[k] = px[k];
}

// fused post-process: clamp to [0,255], truncate toward zero, interleave, splice into the
// destination at (lty+y, ltx+x) for the interior only (seamlessClone_imp.cpp:2091-2096 and
// the host splice loop :470-483).
// guard: when the host launched this clone on a PREDICTED bounding box (RectGuard, sc_common.h) the output is only
// written if the bounding box the device found is the predicted one; otherwise the destination stays untouched
// and the host repeats the clone with the true geometry.
template <bool LM>
__device__ __forceinline__ void postprocess_block(const Field &U, uint8_t *__restrict__ body, int bstep, int c0, const LmNodes &lm)
{
    // four pixels per lane: one 16-byte load per channel, twelve output bytes; a lane whose twelve bytes are all interior
    // pixels and start on a 4-byte boundary (the same for every lane of a row) writes three words, the others bytes
    const int x = 4 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x > U.W - 2 || y < 1 || y > U.H - 2) return;
    uint8_t *b = body + (size_t)y * bstep + 3 * x;
    const size_t o = (size_t)y * U.pitch + x;
    float4 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const float4 *>(U.at(c0 + c) + o);      // x < pitch, pitch % 4 == 0
    if (LM) {
        // float-table correction (sc_lowmode.hip): bilinear interpolation between the four nodes around the pixel; the
        // lane's four pixels lie in one 8-column cell (x is a multiple of 4)
#pragma unroll
        for (int c = 0; c < 3; ++c) lm_add4(lm, c0 + c, x, y, v[c]);
    }
    unsigned char px[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e[4] = { v[c].x, v[c].y, v[c].z, v[c].w };
#pragma unroll
        for (int k = 0; k < 4; ++k) px[3 * k + c] = (unsigned char)lm_byte(e[k]);
    }
    if (x >= 1 && x + 3 <= U.W - 2 && ((uintptr_t)b & 3) == 0) {
        unsigned w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
        unsigned *d32 = reinterpret_cast<unsigned *>(b);
        d32[0] = w[0]; d32[1] = w[1]; d32[2] = w[2];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x + k < 1 || x + k > U.W - 2) continue;
            b[3 * k + 0] = px[3 * k + 0]; b[3 * k + 1] = px[3 * k + 1]; b[3 * k + 2] = px[3 * k + 2];
        }
    }
}

template <bool LM>
__global__ __launch_bounds__(256) void k_postprocess(Field U, uint8_t *__restrict__ body, int bstep, RectGuard guard, LmNodes lm, AbortFlag ab)
{
    if (abort_set(ab)) return;       // a 16-bit store of this solve saturated: the host repeats the clone on float fields
    if (guard.d_rect) {
        const int *__restrict__ r = guard.d_rect;
        if (r[0] != guard.x0 || r[1] != guard.x1 || r[2] != guard.y0 || r[3] != guard.y1) return;
    }
    postprocess_block<LM>(U, body, bstep, 0, lm);
}

template <bool LM>
__global__ __launch_bounds__(256) void k_postprocess_group(Field U, ImageJobs t, LmNodes lm, AbortFlag ab)
{
    if (abort_set(ab)) return;
    const ImageJob &j = t.j[blockIdx.z];
    if (j.d_rect && (j.d_rect[0] != j.rx0 || j.d_rect[1] != j.rx1 || j.d_rect[2] != j.ry0 || j.d_rect[3] != j.ry1)) return;
    if (j.W > 0) { U.W = j.W; U.H = j.H; }          // a member of a size class (the node grid's strides stay the class's)
    postprocess_block<LM>(U, j.body_org, j.bstep, 3 * blockIdx.z, lm);
}

// The same splice for output values that already exist as bytes: the last multigrid launch of a clone wrote them planar into
// the memory of its partner field (k_cycle0, TAG bit 5: plane c at Q.p + c Q.plane bytes, rows of Q.pitch bytes).  Eight
// pixels per lane: two words per channel in, six words (or 24 bytes) out.
template <bool WORDS_IN_ODD_ROWS = false>
__device__ __forceinline__ void splice_block(const Field &Q, uint8_t *__restrict__ body, int bstep, int c0)
{
    const int x = 8 * (blockIdx.x * 64 + (threadIdx.x & 63));
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x > Q.W - 2 || y < 1 || y > Q.H - 2) return;
    const uint8_t *__restrict__ q = reinterpret_cast<const uint8_t *>(Q.p) + (size_t)y * Q.pitch + x;    // pitch % 64 == 0: x + 7 < pitch
    uint2 v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = *reinterpret_cast<const uint2 *>(q + (size_t)(c0 + c) * Q.plane);
    unsigned char px[24];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) px[3 * k + c] = (unsigned char)(((k < 4 ? v[c].x : v[c].y) >> (8 * (k & 3))) & 255u);
    uint8_t *b = body + (size_t)y * bstep + 3 * x;
    // (words where the row is word aligned, bytes otherwise: a variant that wrote the aligned middle of a misaligned run as words --
    //  one to three bytes, five words, the remaining bytes -- measured SLOWER at every alignment, 0.416 -> 0.428 ms for a 2048^2 clone:
    //  the memory system merges a wave's byte stores, the longer instruction stream costs more than they do)
    if (x >= 1 && x + 7 <= Q.W - 2 && ((uintptr_t)b & 3) == 0) {
        unsigned *d32 = reinterpret_cast<unsigned *>(b);
#pragma unroll
        for (int k = 0; k < 6; ++k) d32[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
    } else if (WORDS_IN_ODD_ROWS && x >= 1 && x + 7 <= Q.W - 2) {
        switch ((unsigned)(uintptr_t)b & 3u) {
        case 1: store_run_at<24, 1>(b, px); break;
        case 2: store_run_at<24, 2>(b, px); break;
        default: store_run_at<24, 3>(b, px); break;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (x + k < 1 || x + k > Q.W - 2) continue;
            b[3 * k + 0] = px[3 * k + 0]; b[3 * k + 1] = px[3 * k + 1]; b[3 * k + 2] = px[3 * k + 2];
        }
    }
}

__global__ __launch_bounds__(256) void k_splice_planar(Field Q, uint8_t *__restrict__ body, int bstep, RectGuard guard, AbortFlag ab)
{
    if (abort_set(ab)) return;
    if (guard.d_rect) {
        const int *__restrict__ r = guard.d_rect;
        if (r[0] != guard.x0 || r[1] != guard.x1 || r[2] != guard.y0 || r[3] != guard.y1) return;
    }
    splice_block(Q, body, bstep, 0);
}

__global__ __launch_bounds__(256) void k_splice_planar_group(Field Q, ImageJobs t, AbortFlag ab)
{
    if (abort_set(ab)) return;
    const ImageJob &j = t.j[blockIdx.z];
    if (j.d_rect && (j.d_rect[0] != j.rx0 || j.d_rect[1] != j.rx1 || j.d_rect[2] != j.ry0 || j.d_rect[3] != j.ry1)) return;
    if (j.W > 0) { Q.W = j.W; Q.H = j.H; }          // a member of a size class
    splice_block<true>(Q, j.body_org, j.bstep, 3 * blockIdx.z);
}

void launch_splice_planar(Field Q, uint8_t *body_org, int bstep, hipStream_t s, RectGuard guard, AbortFlag ab)
{
    dim3 grid(((Q.W + 7) / 8 + 63) / 64, (Q.H + 3) / 4);
    hipLaunchKernelGGL(k_splice_planar, grid, dim3(256), 0, s, Q, body_org, bstep, guard, ab);
}

void launch_splice_planar_group(Field Q, const ImageJob *jobs, int n, hipStream_t s, AbortFlag ab)
{
    for (int i0 = 0; i0 < n; i0 += ImageJobs::MAX) {
        ImageJobs t{};
        const int cnt = std::min(n - i0, (int)ImageJobs::MAX);
        for (int i = 0; i < cnt; ++i) t.j[i] = jobs[i0 + i];
        Field q = Q;
        q.p = reinterpret_cast<float *>(reinterpret_cast<uint8_t *>(Q.p) + (size_t)3 * i0 * Q.plane);      // planes are Q.plane BYTES apart here
        dim3 grid(((Q.W + 7) / 8 + 63) / 64, (Q.H + 3) / 4, cnt);
        hipLaunchKernelGGL(k_splice_planar_group, grid, dim3(256), 0, s, q, t, ab);
    }
}

void launch_postprocess_group(Field U, const ImageJob *jobs, int n, hipStream_t s, LmNodes lm, AbortFlag ab)
{
    for (int i0 = 0; i0 < n; i0 += ImageJobs::MAX) {
        ImageJobs t{};
        const int cnt = std::min(n - i0, (int)ImageJobs::MAX);
        for (int i = 0; i < cnt; ++i) t.j[i] = jobs[i0 + i];
        Field u = U;
        u.p = U.p + (size_t)3 * i0 * U.plane;
        LmNodes l = lm;
        if (l.CN) l.CN += (size_t)3 * i0 * l.ny * l.npitch;
        dim3 grid(((U.W + 3) / 4 + 63) / 64, (U.H + 3) / 4, cnt);
        if (l.CN) hipLaunchKernelGGL(k_postprocess_group<true>, grid, dim3(256), 0, s, u, t, l, ab);
        else hipLaunchKernelGGL(k_postprocess_group<false>, grid, dim3(256), 0, s, u, t, l, ab);
    }
}

void launch_postprocess(Field U, uint8_t *body_org, int bstep, hipStream_t s, RectGuard guard, LmNodes lm, AbortFlag ab)
{
    dim3 grid(((U.W + 3) / 4 + 63) / 64, (U.H + 3) / 4);
    if (lm.CN) hipLaunchKernelGGL(k_postprocess<true>, grid, dim3(256), 0, s, U, body_org, bstep, guard, lm, ab);
    else hipLaunchKernelGGL(k_postprocess<false>, grid, dim3(256), 0, s, U, body_org, bstep, guard, lm, ab);
}

// ------------------------------------------------------------------------------------------
// single-sweep smoothers (generic sizes; also the coarse-level smoothers of the multigrid)
// ------------------------------------------------------------------------------------------
// Jacobi, LDS-staged tile with 1-px halo.  Tile = 256 x JT_TH points (one wave spans a full
// 256-float row as 64 float4), LDS row = [3 pad | left halo | 256 | right halo | 3 pad] so the
// body stays 16-B aligned for ds_read_b128.  Left/right neighbours inside the row come from the
// adjacent lanes (wave shuffles); only lane 0 / 63 read the halo columns from LDS.
constexpr int JT_TW = 256, JT_LDW = JT_TW + 8;

// TAG only changes the symbol name: sc_hip_field_time_sweeps launches the TAG=1 instantiation so
// profiler statistics keep the isolated roofline launches apart from the in-clone launches.
template <int JT_TH, int TAG>
__global__ __launch_bounds__(256) void k_jacobi(Field Uin, Field Uout, Field F)
{
    __shared__ __attribute__((aligned(16))) float t[JT_TH + 2][JT_LDW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.z;
    const int tx0 = blockIdx.x * JT_TW, ty0 = blockIdx.y * JT_TH;
    const int W = Uin.W, H = Uin.H, P = Uin.pitch;
    const float *__restrict__ uin = Uin.at(c);
    const int x = tx0 + 4 * lane;
    for (int ry = wv; ry < JT_TH + 2; ry += 4) {
        const int y = ty0 + ry - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool yok = (y >= 0) && (y < H);
        if (yok && x < P) v = *reinterpret_cast<const float4 *>(uin + (size_t)y * P + x);
        *reinterpret_cast<float4 *>(&t[ry][4 + 4 * lane]) = v;
        if (lane == 0) t[ry][3] = (yok && tx0 > 0) ? uin[(size_t)y * P + tx0 - 1] : 0.f;
        if (lane == 63) t[ry][4 + JT_TW] = (yok && tx0 + JT_TW < P) ? uin[(size_t)y * P + tx0 + JT_TW] : 0.f;
    }
    // the right-hand side does not depend on the tile: fetch it before the barrier so its latency
    // overlaps the tile load instead of following it
    const float *__restrict__ f = F.at(c);
    float4 fr[JT_TH / 4];
#pragma unroll
    for (int k = 0; k < JT_TH / 4; ++k) {
        const int y = ty0 + wv + 4 * k;
        fr[k] = (y < H && x < P) ? *reinterpret_cast<const float4 *>(f + (size_t)y * P + x) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (x >= P) return;
    float *__restrict__ uout = Uout.at(c);
#pragma unroll
    for (int k = 0; k < JT_TH / 4; ++k) {
        const int ry = wv + 4 * k, y = ty0 + ry;
        if (y >= H) break;
        const float4 c4 = *reinterpret_cast<const float4 *>(&t[ry + 1][4 + 4 * lane]);
        const float4 u4 = *reinterpret_cast<const float4 *>(&t[ry][4 + 4 * lane]);
        const float4 d4 = *reinterpret_cast<const float4 *>(&t[ry + 2][4 + 4 * lane]);
        float l = wave_from_left(c4.w), r = wave_from_right(c4.x);
        if (lane == 0) l = t[ry + 1][3];
        if (lane == 63) r = t[ry + 1][4 + JT_TW];
        const float4 f4 = fr[k];
        const bool yi = (y >= 1) && (y <= H - 2);
        float4 o = c4;
        if (yi) {
            if (x + 0 >= 1 && x + 0 <= W - 2) o.x = 0.25f * (((l + c4.y) + (u4.x + d4.x)) - f4.x);
            if (x + 1 <= W - 2) o.y = 0.25f * (((c4.x + c4.z) + (u4.y + d4.y)) - f4.y);
            if (x + 2 <= W - 2) o.z = 0.25f * (((c4.y + c4.w) + (u4.z + d4.z)) - f4.z);
            if (x + 3 <= W - 2) o.w = 0.25f * (((c4.z + r) + (u4.w + d4.w)) - f4.w);
        }
        *reinterpret_cast<float4 *>(uout + (size_t)y * P + x) = o;
    }
}

// Jacobi without LDS: every wave owns a 256-column x S-row segment and walks down it with the rows
// y-1, y, y+1 in registers.  No barrier, so the S-row loop is one straight instruction stream the
// compiler can software-pipeline (all loads of the unrolled body issue before the first use).
// Left/right neighbours come from the adjacent lanes (DPP); lanes 0 and 63 fetch the one column
// outside the segment from memory (an L2 hit: the neighbouring wave streams that line anyway).
template <int S, int TAG>
__global__ __launch_bounds__(256) void k_jacobi_roll(Field Uin, Field Uout, Field F)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = Uin.W, H = Uin.H, P = Uin.pitch;
    // 1-D launch; each XCD gets a contiguous band of rows (vertical neighbours share two rows -> same L2),
    // walked row-major so that consecutive workgroups stream whole image rows (DRAM pages) when the field is HBM resident
    const int nbx = (W + 255) / 256, nby = (H + 4 * S - 1) / (4 * S);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int bx = tile % nbx, by = (tile / nbx) % nby, c = tile / (nbx * nby);
    const int x = bx * 256 + 4 * lane;
    const int ya = (by * 4 + wv) * S;
    if (x >= P || ya >= H) return;
    const float *__restrict__ uin = Uin.at(c);
    const float *__restrict__ f = F.at(c);
    float *__restrict__ uout = Uout.at(c);
    // Rows ya-1 .. ya+S, the right-hand sides and the two outside columns are all requested up front.
    // Row and column indices are clamped instead of tested: a clamped value only ever feeds a ring
    // or out-of-range output, which is passed through / not stored, and the loads stay branch-free.
    float4 u[S + 2], fr[S];
    float e[S];
#pragma unroll
    for (int k = 0; k < S + 2; ++k) {
        const int y = min(max(ya + k - 1, 0), H - 1);
        u[k] = *reinterpret_cast<const float4 *>(uin + (size_t)y * P + x);
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int y = min(ya + k, H - 1);
        fr[k] = *reinterpret_cast<const float4 *>(f + (size_t)y * P + x);
        e[k] = 0.f;
    }
    if (lane == 0 || lane == 63) {
        const int ex = lane == 0 ? max(x - 1, 0) : min(x + 4, P - 1);
#pragma unroll
        for (int k = 0; k < S; ++k) e[k] = uin[(size_t)min(ya + k, H - 1) * P + ex];
    }
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int y = ya + k;
        const float4 c4 = u[k + 1], u4 = u[k], d4 = u[k + 2], f4 = fr[k];
        float l = wave_from_left(c4.w), r = wave_from_right(c4.x);
        l = lane == 0 ? e[k] : l;
        r = lane == 63 ? e[k] : r;
        const bool yi = (y >= 1) && (y <= H - 2);
        float4 o;
        o.x = (yi && x + 0 >= 1 && x + 0 <= W - 2) ? 0.25f * (((l + c4.y) + (u4.x + d4.x)) - f4.x) : c4.x;
        o.y = (yi && x + 1 <= W - 2) ? 0.25f * (((c4.x + c4.z) + (u4.y + d4.y)) - f4.y) : c4.y;
        o.z = (yi && x + 2 <= W - 2) ? 0.25f * (((c4.y + c4.w) + (u4.z + d4.z)) - f4.z) : c4.z;
        o.w = (yi && x + 3 <= W - 2) ? 0.25f * (((c4.z + r) + (u4.w + d4.w)) - f4.w) : c4.w;
        if (y < H) *reinterpret_cast<float4 *>(uout + (size_t)y * P + x) = o;
    }
}

// Default: the register-rolling kernel with 4-row segments.  Measured on MI355X at 2048^2 / 4096^2
// (tools/tune_jacobi.py): roll S=4 6.2 / 5.75 TB/s, S=8 6.0 / 5.7, S=16 5.85 / 5.5; LDS tile of 16 rows
// 5.7 / 5.45, 32 rows 4.8 / 4.6, 64 rows 3.2 / 3.3.  A bare out = a + b kernel with the same tiling and
// a one-row halo (tools/stream_probe.hip) reaches 6.9 / 5.9 TB/s on the same data, so the sweep runs
// at 90 / 97 % of what this GPU streams; dropping the two outside-column loads would close most of the
// rest (measured 6.7 TB/s without them).  lds_th = 16, 32 or 64 (sc_solver_opts.jacobi_tile_rows) selects the
// LDS-tiled kernel.
void launch_jacobi(Field Uin, Field Uout, Field F, hipStream_t s, bool tag, int lds_th)
{
    if (lds_th) {
        const int th = lds_th == 64 ? 64 : (lds_th == 32 ? 32 : 16);
        dim3 grid((Uin.W + JT_TW - 1) / JT_TW, (Uin.H + th - 1) / th, Uin.C);
        if (th == 64) hipLaunchKernelGGL((k_jacobi<64, 0>), grid, dim3(256), 0, s, Uin, Uout, F);
        else if (th == 32) hipLaunchKernelGGL((k_jacobi<32, 0>), grid, dim3(256), 0, s, Uin, Uout, F);
        else hipLaunchKernelGGL((k_jacobi<16, 0>), grid, dim3(256), 0, s, Uin, Uout, F);
        return;
    }
    constexpr int S = 4;
    const dim3 grid(((Uin.W + 255) / 256) * ((Uin.H + 4 * S - 1) / (4 * S)) * Uin.C);
    if (tag) hipLaunchKernelGGL((k_jacobi_roll<4, 1>), grid, dim3(256), 0, s, Uin, Uout, F);
    else hipLaunchKernelGGL((k_jacobi_roll<4, 0>), grid, dim3(256), 0, s, Uin, Uout, F);
}

// One colour of a red-black Gauss-Seidel / SOR sweep, in place.  A colour-c point reads only
// colour 1-c neighbours, none of which is written by this launch, so in-place float4
// read-modify-write is race-free (unchanged components are stored back bit-identically).
template <bool SOR, int TAG>
__global__ __launch_bounds__(256) void k_rb_half(Field U, Field F, int color, float omega)
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
    float4 o = c4;
    const int par = (x + y + color) & 1; // 0: components 0,2 have colour `color`; 1: components 1,3
#define SC_RB_UPD(dst, L, R, UU, DD, FF)                                   \
    {                                                                      \
        const float gs = 0.25f * ((((L) + (R)) + ((UU) + (DD))) - (FF));   \
        dst = SOR ? (dst + omega * (gs - dst)) : gs;                       \
    }
    if (par == 0) {
        if (x + 0 >= 1 && x + 0 <= W - 2) SC_RB_UPD(o.x, l, c4.y, u4.x, d4.x, f4.x)
        if (x + 2 <= W - 2) SC_RB_UPD(o.z, c4.y, c4.w, u4.z, d4.z, f4.z)
    } else {
        if (x + 1 <= W - 2) SC_RB_UPD(o.y, c4.x, c4.z, u4.y, d4.y, f4.y)
        if (x + 3 <= W - 2) SC_RB_UPD(o.w, c4.z, r, u4.w, d4.w, f4.w)
    }
#undef SC_RB_UPD
    *reinterpret_cast<float4 *>(row + x) = o;
}

void launch_rb_half(Field U, Field F, int color, float omega, hipStream_t s, bool tag)
{
    dim3 grid((U.W + 255) / 256, (U.H + 3) / 4, U.C);
    if (omega == 1.0f) {
        if (tag) hipLaunchKernelGGL((k_rb_half<false, 1>), grid, dim3(256), 0, s, U, F, color, omega);
        else hipLaunchKernelGGL((k_rb_half<false, 0>), grid, dim3(256), 0, s, U, F, color, omega);
    } else {
        hipLaunchKernelGGL((k_rb_half<true, 0>), grid, dim3(256), 0, s, U, F, color, omega);
    }
}

// ------------------------------------------------------------------------------------------
// residual norm: sum r^2 and sum lap^2, r = lap - ((l+r)+(u+d) - 4U), float32 per point,
// double accumulation: per lane -> wave64 __shfl_down -> LDS across waves -> one partial pair
// per block; a second single-block kernel folds the partials in a fixed order (deterministic,
// no float atomics).
// ------------------------------------------------------------------------------------------
constexpr int RES_MAX_BLOCKS = 2048;
int residual_max_blocks() { return RES_MAX_BLOCKS; }

__global__ __launch_bounds__(256) void k_residual(Field U, Field F, double *__restrict__ partials, int row_groups)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int W = U.W, H = U.H, P = U.pitch;
    const int xgroups = (W + 255) / 256;
    const int total = xgroups * row_groups * U.C;
    double r2 = 0.0, f2 = 0.0;
    for (int job = blockIdx.x; job < total; job += gridDim.x) {
        const int c = job / (xgroups * row_groups);
        const int rem = job - c * (xgroups * row_groups);
        const int rg = rem / xgroups, xg = rem - rg * xgroups;
        const int x = xg * 256 + 4 * lane;
        const int y = rg * 4 + wv;
        if (y < 1 || y > H - 2 || x >= P) continue;
        const float *__restrict__ row = U.at(c) + (size_t)y * P;
        const float4 c4 = *reinterpret_cast<const float4 *>(row + x);
        const float4 u4 = *reinterpret_cast<const float4 *>(row - P + x);
        const float4 d4 = *reinterpret_cast<const float4 *>(row + P + x);
        const float4 f4 = *reinterpret_cast<const float4 *>(F.at(c) + (size_t)y * P + x);
        float l = wave_from_left(c4.w), r = wave_from_right(c4.x);
        if (lane == 0) l = (x > 0) ? row[x - 1] : 0.f;
        if (lane == 63) r = (x + 4 < P) ? row[x + 4] : 0.f;
#define SC_RES(CC, L, R, UU, DD, FF, XI)                                          \
    if ((XI) >= 1 && (XI) <= W - 2) {                                             \
        const float s = (((L) + (R)) + ((UU) + (DD))) - 4.0f * (CC);              \
        const float res = (FF) - s;                                               \
        r2 += (double)res * (double)res;                                          \
        f2 += (double)(FF) * (double)(FF);                                        \
    }
        SC_RES(c4.x, l, c4.y, u4.x, d4.x, f4.x, x + 0)
        SC_RES(c4.y, c4.x, c4.z, u4.y, d4.y, f4.y, x + 1)
        SC_RES(c4.z, c4.y, c4.w, u4.z, d4.z, f4.z, x + 2)
        SC_RES(c4.w, c4.z, r, u4.w, d4.w, f4.w, x + 3)
#undef SC_RES
    }
    r2 = wave_sum_d(r2);
    f2 = wave_sum_d(f2);
    __shared__ double red[4][2];
    if (lane == 0) { red[wv][0] = r2; red[wv][1] = f2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x + 0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partials[2 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__global__ __launch_bounds__(256) void k_residual_final(const double *__restrict__ partials, int n, double *__restrict__ out)
{
    double r2 = 0.0, f2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { r2 += partials[2 * i]; f2 += partials[2 * i + 1]; }
    r2 = wave_sum_d(r2);
    f2 = wave_sum_d(f2);
    __shared__ double red[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = r2; red[wv][1] = f2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        out[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

void launch_residual(Field U, Field F, double *d_partials, double *d_out, hipStream_t s)
{
    const int row_groups = (U.H + 3) / 4;
    const int xgroups = (U.W + 255) / 256;
    long total = (long)row_groups * xgroups * U.C;
    int blocks = (int)(total < RES_MAX_BLOCKS ? total : RES_MAX_BLOCKS);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_residual, dim3(blocks), dim3(256), 0, s, U, F, d_partials, row_groups);
    hipLaunchKernelGGL(k_residual_final, dim3(1), dim3(256), 0, s, d_partials, blocks, d_out);
}

// ------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_zero(Field U)
{
    const size_t n4 = U.plane * (size_t)U.C / 4;
    float4 *p = reinterpret_cast<float4 *>(U.p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_fill_zero(Field U, hipStream_t s)
{
    size_t n4 = U.plane * (size_t)U.C / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_fill_zero, dim3(blocks), dim3(256), 0, s, U);
}

} // namespace sc
