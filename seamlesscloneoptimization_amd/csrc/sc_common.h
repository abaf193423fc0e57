// sc_common.h -- shared host-side declarations of libseamlessclone_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/seamlessclone_hip.h"

namespace sc {

// Planar float32 field: C planes of H rows, row pitch in floats (multiple of 64 = 256 B so
// every row starts on a cache-line / float4 boundary; the ring column x=0 sits at the row
// start, so float4 groups are aligned in ROI coordinates).
struct Field {
    float *p = nullptr;
    int W = 0, H = 0, C = 0;
    int pitch = 0;      // floats per row
    size_t plane = 0;   // floats per plane (pitch * H rounded up to 64)
    __host__ __device__ float *at(int c) const { return p + (size_t)c * plane; }
    size_t bytes() const { return plane * (size_t)C * sizeof(float); }
};

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ROI geometry (seamlessClone_imp.cpp:1014-1016,1066)
struct Geo { int x0, y0, W, H, ltx, lty; };

// 8-bit interleaved image view (cv::Mat {data, cols, rows, step})
struct Img8 {
    const uint8_t *p; int cols, rows, step;
};

// mask stage of one member of a group of clones (launch_mask_*_group): the scan uses mask/mw/mh/mstep/rect, the erode
// additionally g/M/mpitch (mask_bytes is filled in by the launcher)
struct MaskJob {
    const uint8_t *mask; int mw, mh, mstep;
    int *rect;
    int *rect_host;      // (group scans) the same four words in the host's pinned mailbox, written by the fold: no copy command (may be nullptr)
    size_t mask_bytes;
    Geo g;
    uint8_t *M; int mpitch;
};
struct MaskJobs { enum { MAX = 16 }; MaskJob j[MAX]; };     // by value in the kernel arguments
// pre- / post-process of the members of a group: ROI origins in the images and the member's eroded mask; member i owns
// channels 3i..3i+2 of the fields (blockIdx.z = i)
struct ImageJob { const uint8_t *face_org; int fstep; uint8_t *body_org; int bstep; const uint8_t *M;
                  const int *d_rect; int rx0, rx1, ry0, ry1;
                  int W, H; };      // W > 0: the member's own ROI size inside fields laid out for a larger one (a size class, RagMember); 0: the fields' size   // d_rect != nullptr: the member ran on a PREDICTED bounding box and is spliced only if the device found exactly that box (RectGuard semantics)
struct ImageJobs { enum { MAX = 16 }; ImageJob j[MAX]; };

// ---------------------------------------------------------------- kernel launchers (sc_kernels.hip)
// single-mask bounding box: per-workgroup parts folded by the last workgroup to arrive (sc_kernels.hip, mask_bbox_block);
// nbx / nblocks are filled in by the launchers
struct BboxFold { int *parts = nullptr; unsigned *counter = nullptr; int *rect_dev = nullptr; int *rect_host = nullptr; int nbx = 0, nblocks = 0; };
int  mask_bbox_blocks(int mw, int mh);                               // workgroups of the scan = parts (4 ints each) it needs
void launch_mask_bbox(const uint8_t *mask, int mw, int mh, int mstep, BboxFold fold, hipStream_t s);
void launch_mask_erode3(const uint8_t *mask, int mstep, int mask_rows, Geo g, uint8_t *M, int mpitch, hipStream_t s);
void launch_mask_erode_min7(const uint8_t *mask, int mstep, Geo g, uint8_t *M, int mpitch, hipStream_t s);   // OpenCV's grey-mask erode (SC_FLAG_OPENCV_GREY_MASK)
size_t mask_bbox_group_parts(const MaskJob *jobs, int n);            // ints of scratch the group scan needs (one set of extrema per workgroup)
void launch_mask_bbox_group(const MaskJob *jobs, int n, hipStream_t s, int *parts);
void launch_mask_erode3_group(const MaskJob *jobs, int n, hipStream_t s);
// body_org: pointer to the pixel that corresponds to ROI (0,0); face_org likewise (patch + offset)
// scan: the clone's bounding-box scan rides in this launch as extra workgroups (fold.nbx / nblocks / scan_rows are filled in here)
// g, mask_bytes: the launch's tiles also form the eroded mask of the (predicted) ROI `g` themselves, from the mask bytes (mask_bytes of them are
// the caller's), and leave it in M as k_mask_erode3 would -- no erode launch in front of the pre-process
struct BboxTask { const uint8_t *mask = nullptr; int mw = 0, mh = 0, mstep = 0; BboxFold fold; int scan_rows = 0; Geo g{}; size_t mask_bytes = 0; uint8_t *M_out = nullptr; };     // M_out: where the tiles leave the eroded mask (the buffer the launch's M argument names)
void launch_preprocess(const uint8_t *body_org, int bstep, const uint8_t *face_org, int fstep,
                       const uint8_t *M, int mpitch, Field U0, Field U1, Field F, hipStream_t s, bool f_half = false, bool u_half = false, bool grey = false,
                       const BboxTask *scan = nullptr);
// bounding box the host assumed when it launched a clone before the device's answer was back (d_rect == nullptr: none)
struct RectGuard { const int *d_rect = nullptr; int x0 = 0, x1 = 0, y0 = 0, y1 = 0; };
// "Do not write": a device word the launches of one solve may set to that solve's generation number (a 16-bit fixed-point store
// of the field saturated, k_cycle0 TAG bit 9); the output launches of the same solve then write nothing and the host repeats
// the clone on float fields.  A generation instead of a flag: nothing has to be reset between solves.  p == nullptr: none.
// `host`: a second copy of the word in pinned host memory, for the host to read without a copy command (the device copy is the
// one the output launches test: a million threads polling a word across PCIe made the splice 170 times slower).
struct AbortFlag { unsigned *p = nullptr; unsigned gen = 0; unsigned *host = nullptr; };
#if defined(__HIPCC__)
__device__ __forceinline__ bool abort_set(const AbortFlag &a) { return a.p && *reinterpret_cast<const volatile unsigned *>(a.p) == a.gen; }
#endif
// float-table correction at the nodes = every 8th field row and column (sc_lowmode.hip): CN[c][Y][X], ny rows of npitch
// floats per channel; the post-process adds the bilinear interpolation between the four nodes around a pixel.
// CN == nullptr: none.
struct LmNodes { const float *CN = nullptr; int ny = 0, npitch = 0; };
#if defined(__HIPCC__)
// the correction at the four pixels x .. x + 3 (x a multiple of 4: one 8-column cell) of row y, added to v; and the output
// byte of a value: clamp to [0, 255], truncate (seamlessClone_imp.cpp:2091-2096).  One definition for the post-process and for
// the multigrid launch that writes output bytes itself: the same operations in the same order.
// (p00, p01: the two nodes of the cell's upper node row, p10, p11: of its lower one)
__device__ __forceinline__ void lm_add4_nodes(float p00, float p01, float p10, float p11, int x, int y, float4 &v)
{
    const float ty = 0.125f * (float)(y & 7);
    const float l = __builtin_fmaf(ty, p10 - p00, p00), r = __builtin_fmaf(ty, p11 - p01, p01);
    const float dx = 0.125f * (r - l), a0 = __builtin_fmaf((float)(x & 7), dx, l);
    v.x += a0;
    v.y += a0 + dx;
    v.z += __builtin_fmaf(2.0f, dx, a0);
    v.w += __builtin_fmaf(3.0f, dx, a0);
}
__device__ __forceinline__ void lm_add4(const LmNodes &lm, int c, int x, int y, float4 &v)
{
    const float *__restrict__ p = lm.CN + ((size_t)c * lm.ny + (y >> 3)) * lm.npitch + (x >> 3);
    lm_add4_nodes(p[0], p[1], p[lm.npitch], p[lm.npitch + 1], x, y, v);
}
__device__ __forceinline__ unsigned lm_byte(float d)
{
    d = d > 255.0f ? 255.0f : d;
    d = d < 0.0f ? 0.0f : d;
    return (unsigned)(unsigned char)d;
}
__device__ __forceinline__ float lm_bilinear(const LmNodes &lm, int c, int x, int y)
{
    const float *__restrict__ p = lm.CN + ((size_t)c * lm.ny + (y >> 3)) * lm.npitch + (x >> 3);
    const float tx = 0.125f * (float)(x & 7), ty = 0.125f * (float)(y & 7);
    const float top = __builtin_fmaf(tx, p[1] - p[0], p[0]), bot = __builtin_fmaf(tx, p[lm.npitch + 1] - p[lm.npitch], p[lm.npitch]);
    return __builtin_fmaf(ty, bot - top, top);
}
#endif
void launch_postprocess(Field U, uint8_t *body_org, int bstep, hipStream_t s, RectGuard guard = RectGuard(), LmNodes lm = LmNodes(), AbortFlag ab = AbortFlag());
// the same for a group (fields of 3n channels), one launch per 16 members
void launch_preprocess_group(const ImageJob *jobs, int n, int mpitch, Field U0, Field F, hipStream_t s, bool f_half, bool u_half);
void launch_postprocess_group(Field U, const ImageJob *jobs, int n, hipStream_t s, LmNodes lm = LmNodes(), AbortFlag ab = AbortFlag());
// splice of output bytes a multigrid launch left planar in Q's memory (launch_cycle0_out): interleave into the destination
void launch_splice_planar(Field Q, uint8_t *body_org, int bstep, hipStream_t s, RectGuard guard = RectGuard(), AbortFlag ab = AbortFlag());
void launch_splice_planar_group(Field Q, const ImageJob *jobs, int n, hipStream_t s, AbortFlag ab = AbortFlag());
void launch_half_to_float(const void *src_half, float *dst, size_t n, hipStream_t s);
// up to 16 device-to-device copies in one launch (16-byte aligned pointers)
struct CopyJobs { enum { MAX = 16 }; void *dst[MAX]; const void *src[MAX]; size_t bytes[MAX]; };
void launch_copy_group(const CopyJobs &t, int n, hipStream_t s);

void launch_jacobi(Field Uin, Field Uout, Field F, hipStream_t s, bool tag = false, int lds_tile_rows = 0);
void launch_rb_half(Field U, Field F, int color, float omega, hipStream_t s, bool tag = false);
// fused temporally-blocked kernels (sc_sweep_tb.hip); return false when the shape is unsupported
bool launch_jacobi_tb(Field Uin, Field Uout, Field F, int sweeps, hipStream_t s, bool tag = false);
bool launch_rb_tb(Field Uin, Field Uout, Field F, int sweeps, float omega, hipStream_t s, bool tag = false);
int  tb_max_depth(int method);
int  tb_hard_max_depth(int method);
long tb_big_side();
int  tb_gen_rows(int W, int H, int C, int hx, int hy);   // band height (rows per lane) of a coarse-level launch
int  tb_gen_rows_deep(int W, int H, int C, int hx, int hy);   // the same for launches of depth 3 or 4: 4 or 6
struct MGGeom;
struct ComposeArgs;
struct RagMember;
constexpr int TBM_PLAIN = 0, TBM_PROLONG = 1, TBM_ZEROIN = 4;   // mode of launch_rb_tb_gen
bool launch_rb_tb_gen(Field Uin, Field Uout, Field F, int sweeps, const MGGeom &g, int mode, Field E, hipStream_t s, const RagMember *rag = nullptr, int lev = 0);
int  launch_rb_tb_prolong0(Field Uin, Field Uout, Field F, int sweeps, const MGGeom &g, Field E, float *partial, hipStream_t s);
int  tb_blocks_level0(int W, int H, int C, int sweeps);
void launch_max_final(const float *d_partial, int n, unsigned *d_out, hipStream_t s);
void launch_max_final2(const float *d_a, int na, const float *d_b, int nb, unsigned *d_out2, hipStream_t s, const unsigned *flag = nullptr);   // out[0] = max a, out[1] = max b (-1: b empty), out[2] = *flag (0 without one)
// whole level-0 part of a V-cycle in one launch (sc_cycle0.hip): [prolong E] + `sweeps` RBGS sweeps +
// residual + restriction into Fc.  Returns #partials written, 0 without prolong, -1 if unsupported.
int  launch_cycle0(Field Uin, Field Uout, Field F, Field Fc, Field E, const MGGeom &g, int sweeps, bool prolong,
                   float *partial, hipStream_t s, bool tag = false, bool f_half = false, bool u_half = false,
                   bool final_cycle = false, float4 *bands = nullptr, bool l1_half = false, bool q16_out = false, AbortFlag sat = AbortFlag(),
                   const RagMember *rag = nullptr);      // rag (here and below): the launch serves a size class, see RagMember
// bands (final form, or 4 sweeps with prolongation): receives the cell shares of the float-table correction of the field the
// launch writes, two float4 per (channel, tile row, wave, 8-column cell) -- see k_cycle0 and sc_lowmode.hip
void cycle0_row_geometry(int H, int sweeps, int &nby, int &step, int &hy);
int  cycle0_blocks(int W, int H, int C, int sweeps);
// the same launch with its prolongation source composed on the fly from level 1 (before post-smoothing) and level 2
// (sc_cycle0.hip, ComposeArgs); -1: combination not instantiated
int  launch_cycle0_composed(Field Uin, Field Uout, Field F, Field Fc, Field U1, const MGGeom &g, int sweeps, float *partial,
                            hipStream_t s, bool tag, bool f_half, bool final_cycle, Field E2, const MGGeom &g1, float4 *bands = nullptr, bool l1_half = false,
                            int u_q16 = 0, AbortFlag sat = AbortFlag(), const RagMember *rag = nullptr);     // sat: where a saturating 16-bit store reports itself
// tagged twins of the other launches of a fast-path solve, for isolated timing (sc_cycle0.hip)
int  launch_cycle0_twin(int form, Field Uin, Field Uout, Field F, Field Fc, Field U1, const MGGeom &g, float *partial, hipStream_t s,
                        Field E2, const MGGeom &g1, float4 *bands, const LmNodes &lm);
// the last cycle with its result leaving as output bytes (planar, in Q's memory) instead of as a field; see sc_cycle0.hip
int  launch_cycle0_out(Field Uin, Field Q, Field F, Field Fc, Field E, const MGGeom &g, float *partial, hipStream_t s, bool f_half,
                       bool composed, Field E2, const MGGeom &g1, const LmNodes &lm, bool l1_half = false, const RagMember *rag = nullptr);
// coarse level: zero-guess pre-smoothing + residual + restriction fused (Uout = smoothed correction, Fc = next RHS)
bool launch_cycle_coarse(Field Uout, Field F, Field Fc, const MGGeom &g, int sweeps, hipStream_t s, bool half_io = false,
                         const RagMember *rag = nullptr, int lev = 0);

// residual: d_out[0] = sum r^2, d_out[1] = sum lap^2 (double); d_partials holds >= 2*max_blocks doubles
int  residual_max_blocks();
void launch_residual(Field U, Field F, double *d_partials, double *d_out, hipStream_t s);

// ---------------------------------------------------------------- multigrid (sc_mg_kernels.hip)
// One grid direction of a level pair.  Level l has n interior points at unit spacing (in
// level units) except the LAST gap, from point n to the Dirichlet boundary, which is alpha
// (0.5 <= alpha <= 1.5).  That one irregular interval lets any ROI size coarsen without
// moving the boundary: coarse points sit on the even fine points 2,4,..,2*nc.
struct MGDim {
    int n, nc;        // interior points on this level / on the next coarser one
    float alpha;      // last gap of this level
    float cw_last;    // stencil weight of the inner neighbour at the last point: 2/(1+alpha)
    float d_last;     // diagonal contribution at the last point: 2/alpha (regular: 2)
    float tw1, tw2;   // interpolation weights of the fine tail points 2nc+1, 2nc+2 (0 if absent)
    float inv_last;   // 1 / (row sum of the transposed interpolation at coarse point nc)
};
struct MGGeom { MGDim x, y; };
// Composed prolongation source (sc_mg_device.h): the level interpolated from ran without post-smoothing and without a
// prolongation launch of its own; E2 = finished correction of the level below it, g1 = its geometry (transfer to that level).
struct ComposeArgs { Field E2; MGGeom g1; };

// ---- size classes ("ragged groups", round 5): clones of DIFFERENT ROI sizes through one set of solver launches -----------------
// The members of a class share the fields' strides (pitch and plane of the class's largest width / height on every level), the
// launch grids (sized for the class) and every compile-time choice (hierarchy depth, the bottom solve's operand padding, the
// correction's mode-block counts); everything that depends on a member's own W x H is read from this table by member index
// (channel / 3: a wave-uniform scalar load) at kernel entry: its field size on every level, the level geometries, its bottom
// matrices, its float-table correction tables.  Tiles beyond a member's extent exit.  Every member's arithmetic is, operation for
// operation, that of its solo run: same bytes alone, in a same-size group or in a class.
constexpr int RAG_MAX_LEVELS = 12;
struct RagMember {
    int W, H;                                  // level-0 field, ring included
    int lw[RAG_MAX_LEVELS], lh[RAG_MAX_LEVELS]; // level-l field, ring included (lw[0] = W): clamp bounds of a member's loads
    MGGeom g[RAG_MAX_LEVELS];                  // level l and its transfer to l + 1
    const unsigned char *mm;                   // matrix-core operands of the member's directly solved level (k_mg_tail)
    int npx, npy;                              // ... padded to this many per side (32 or 64): the member's own, as in its solo run
    // float-table correction (sc_lowmode.hip): the member's node grid, its split of the projection and its tables
    int lm_nx, lm_ny, lm_cells_y, lm_nxt, lm_nrs, lm_nparts, lm_Kx, lm_Ky;     // nodes, cell rows, column tiles / row splits / parts of the projection, modes kept
    const float *lm_Sx, *lm_Sy, *lm_R;
    const int *lm_map[2];                      // parts of each cell row for the 2- and the 4-sweep tiling of a level-0 launch
};

void launch_rb_half_gen(Field U, Field F, int color, float omega, MGGeom g, hipStream_t s);
void launch_residual_restrict(Field U, Field F, Field Fc, MGGeom g, hipStream_t s); // Fc = 4 * normalised P^T (F - A U)
int  prolong_blocks(int nx, int ny, int C);
// Uf += P Uc; with d_partial (>= prolong_blocks floats) also *d_maxcorr = bits of max |P Uc|
void launch_prolong_add(Field Uc, Field Uf, MGGeom g, float *d_partial, unsigned *d_maxcorr, hipStream_t s);
void launch_fill_zero(Field U, hipStream_t s);

// bottom of the V-cycle fused into one launch (one workgroup per channel, levels LDS resident)
constexpr int MG_BOTTOM_MAX_LEVELS = 12;
constexpr int MG_BOTTOM_LDS_BYTES = 152 * 1024;   // of the CU's 160 KiB
struct MGBottomLevel { MGGeom g; float omega; int offU, offF, pitch; };   // LDS offsets / row pitch in floats
// Direct solve of one bottom level by fast diagonalisation (sc_multigrid.cpp builds the matrices):
// the level's operator is a tensor sum Tx (x) I + I (x) Ty of two tridiagonal 1-D operators, so with
// Tx = Vx Lx Vx^-1, Ty = Vy Ly Vy^-1 the solution of A U = F is
//     U = Vy [ (Vy^-1 F Vx^-T) / (ly_j + lx_i) ] Vx^T      -- four small dense products in LDS.
// fd_mats (HBM, padded to multiples of 4, zero padded): Mx1[nxp][nxp], My1T[nyp][nyp], My2T[nyp][nyp],
// Mx2[nxp][nxp], Dinv[nyp][nxp].  fd_level < 0: no direct solve (V-cycle down to the coarsest level).
struct MGBottomArgs {
    int nlevels, pre, post, coarse_sweeps, lds_floats;
    int fd_level, fd_nxp, fd_nyp, fd_off;   // level index inside the bottom, padded sizes, LDS offset of the FD region
    const float *fd_mats;
    Field Ftop, Utop;       // HBM planes of the first bottom level: RHS in, correction out
    MGBottomLevel lv[MG_BOTTOM_MAX_LEVELS];
};
__host__ __device__ static inline long fd_mat_floats(int nxp, int nyp) { return 2L * nxp * nxp + 2L * nyp * nyp + (long)nxp * nyp; }
__host__ __device__ static inline long fd_lds_floats(int nxp, int nyp) { return fd_mat_floats(nxp, nyp) + 2L * nxp * nyp; }
hipError_t mg_bottom_prepare();
// per-size state built on the device (sc_mg_kernels.hip): the direct solve's matrices from the closed-form eigenpairs of the
// level's two 1-D operators (nx, ny <= 128), and the zeroing of every plane of the levels >= 1 in one launch
// mm != nullptr: also the operands of the matrix-core form (k_mg_bottom_mm), padded to NPX / NPY (32, 64 or 96)
void launch_fd_build(float *mats, const MGGeom &g, int nxp, int nyp, hipStream_t s, unsigned char *mm = nullptr, int NPX = 0, int NPY = 0);
void launch_fd_build_rag(const RagMember *rag, int members, int lev, hipStream_t s);   // the operands of every member of a size class (each at its own padding), one launch
__host__ __device__ static inline long fd_mm_bytes(int NPX, int NPY) { return 4L * (2L * NPX * NPX + 2L * NPY * NPY + (long)NPX * NPY); }   // AX1 | AX2 | AY1 | AY2 | Dinv, float
// the bottom's first level solved directly on the matrix cores: right-hand side in (Ftop), correction out (Utop), nx x ny unknowns
struct MGBottomMM { const unsigned char *mm; Field Ftop, Utop; int nx, ny; };
bool launch_mg_bottom_mm(const MGBottomMM &a, int NPX, int NPY, int C, hipStream_t s);
// the level above the bottom (F: its right-hand side, U: receives its finished correction) and the bottom in one launch
// (sc_mg_kernels.hip, k_mg_tail); false: not a shape this path serves (the caller launches the three kernels it replaces)
struct MGTail { const unsigned char *mm; Field F, U; MGGeom g; int pre, post; unsigned long long *stamps;
                const RagMember *rag; int lev; bool rag_uniform; };      // (rag_uniform: every member of the class has the class's operand padding)      // rag != nullptr: a size class -- g, mm and F's row count are member (channel / 3)'s own, at its level `lev`   // stamps: measurement only (11 shader-clock values of channel 0), else nullptr
bool launch_mg_tail(const MGTail &a, int NPX, int NPY, int C, hipStream_t s);
struct ZeroJobs { enum { MAX = 48 }; void *p[MAX]; size_t n16[MAX]; int count; };     // n16: 16-byte units
void launch_zero_multi(const ZeroJobs &z, hipStream_t s);
void launch_mg_bottom(const MGBottomArgs &a, int C, hipStream_t s);

} // namespace sc
