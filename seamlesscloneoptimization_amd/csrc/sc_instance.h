// sc_instance.h -- the per-GPU instance behind the C ABI: one HIP stream, a grow-only device
// arena (the role SCImage plays in the reference, seamlessClone_imp.h:90-457: capacity only
// ever grows, steady state does no hipMalloc), pinned mailboxes for the two small read-backs
// (bounding box, residual norm) and hipEvents for per-stage timing.
#pragma once
#include "sc_common.h"
#include "sc_hostcopy.h"
#include <memory>
#include <cmath>

namespace sc {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool own = true;       // false: a piece of one of the instance's slabs (Instance::slabs) -- not a hipMalloc block of its own, never hipFree'd
};
// releases a device buffer on instance teardown: its own hipMalloc block, or nothing for a piece of a slab
inline void dev_release(DevBuf &b) { if (b.p && b.own) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

// eigen-decomposition of one 1-D level operator (sc_multigrid.cpp, fast-diagonalisation bottom solve)
struct FD1 {
    int n = 0;
    float cw_last = 0.f, d_last = 0.f;     // together with n: the key
    std::vector<double> lam, q, ee;          // eigenvalues; q[k*n + i] = component i of eigenvector k; diagonal of E
};

// float-table correction (sc_lowmode.hip): sine tables and coefficient buffers for the current ROI size
struct LowMode {
    int w = 0, h = 0, C = 0;               // interior size the tables were built for; channels the buffers hold
    int Kx = 0, Ky = 0, Kxp = 0, Kyp = 0;  // modes per direction, padded to the register block
    int nx = 0, ny = 0, npitch = 0;        // nodes (every 8th field column / row) and the row pitch of CN
    bool singular = false;                 // the reference's float denominator of the lowest mode is zero at this size: no correction
    double max_ratio = 0.0;                // max |den_exact / den_float - 1| over the corrected modes
    DevBuf B;                              // cell-share parts left by a level-0 launch [C][band rows][cells_x][2] float4
    struct PartMap { DevBuf d, h; int H = 0, sweeps = 0; } maps[2];      // parts of each cell row (device, pinned) for the 2- and the 4-sweep tiling
    PartMap *map_used = nullptr;
    int rag_tiling = 0;                    // a size class: which of the members' two maps the last bands buffer belongs to (0: 2 sweeps, 1: 4)
    int band_rows = 0;
    const float *bands_of = nullptr;       // the field whose parts B holds (nullptr: none); cleared whenever a solve starts or moves on
    DevBuf Sx, Sy, R, P, E, CN;            // Sx[nx][Kxp], Sy[ny][Kyp] (sines at the nodes), R[Kyp][Kxp] -- views of the current entry of `tables` --, P = cell shares float4[C][cells_y][cells_x], E = partial products of the coarse projection [parts][C][Kyp][Kxp], CN[C][ny][npitch]
    // the per-size tables, kept in a small LRU (a caller alternating between a few ROI sizes rebuilds nothing): Sx, Sy built on
    // the device, R on the host into the entry's own pinned staging (no stream synchronisation; the upload event is waited for
    // only when the entry is reused for another size)
    struct Tables { int w = 0, h = 0; bool singular = false; double max_ratio = 0.0; DevBuf Sx, Sy, R, hR; hipEvent_t ev = nullptr; unsigned long long used = 0; };
    enum { TABLES = 4 };
    Tables tables[TABLES];
    unsigned long long tick = 0;
};

// direct DST solve (sc_dst.hip): DST matrices, the reference's float tables and the double work planes
struct DstState {
    int w = 0, h = 0, wp = 0, hp = 0;      // interior size the tables were built for, padded to the 128 x 128 tiles
    bool singular = false;                 // the reference's float tables are singular at this size: exact denominators instead
    DevBuf Sw, Sh, fxy, G, T1, T2;         // Sw[wp][wp], Sh[hp][hp] double; fx[wp] + fy[hp] float; G, T1, T2: [C][hp][wp] double
    DevBuf hfxy;                           // pinned staging of the float tables
};

// FFT direct solve (sc_fft.hip): per-direction chirp / transform tables and the two work planes.  The tables of a transform
// length are built ON THE DEVICE (k_fft_build: both directions of a solve in one launch) and kept in a small LRU: a caller whose mask changes with every
// frame meets new ROI sizes all the time, and alternating between a few sizes costs nothing.
struct FftDim { int n = 0, logM = 0, r = 1; bool dbl = false; DevBuf chirp; unsigned long long used = 0; };   // M = r 2^logM (r = 1, 3, 5); chirp: chirp[n+1] | bhat[M] | tw[M] | tw2[2^logM] (complex float or double)
struct FftFxy { int w = 0, h = 0; bool singular = false; DevBuf d, hst; hipEvent_t ev = nullptr; unsigned long long used = 0; };   // the reference's float tables fx[w] + fy[h]: device, pinned staging of its own, upload event
struct FftState {
    enum { DIMS = 8, FXY = 4 };
    FftDim dims[DIMS];
    FftFxy fxy[FXY];
    unsigned long long tick = 0;
    DevBuf A, B;                           // work planes [C][h][w]
    DevBuf hst_all;                        // pinned staging of all FXY eigenvalue-table entries (one block; FftFxy::hst is unused since round 5)
    DevBuf tw64;                           // double twiddles of the build's own transform (float tables are built through a double FFT)
    hipEvent_t ev_fork = nullptr, ev_built = nullptr;   // the build runs on the instance's second stream
    bool pending = false;                  // ... and `stream` has not waited for ev_built yet
    FftDim *req[2] = { nullptr, nullptr }; // tables queued for a build by the current solve (its two directions), launched together
    int nreq = 0;
    bool forked = false;                   // this solve has already put the second stream behind the main one (one fork per solve: both directions' tables and the eigenvalue tables ride on it)
};

struct MGLevel {
    Field U, F, T;   // correction, RHS, scratch (residual field); level 0 aliases the instance fields
    MGGeom g;        // geometry of this level and of its transfer to the next coarser one
    float omega = 1.f; // SOR factor used when this is the coarsest level
};

// ---- size classes (sc_ragged.cpp; RagMember in sc_common.h) -------------------------------------------------------------------
// What one ROI size needs in order to share a set of launches with OTHER sizes: host arithmetic only.
struct SizePlan {
    int W = 0, H = 0;
    bool ok = false;                    // the default fast path serves this size inside a class (else: same-size groups, or alone)
    bool conditional = false;           // the float-table correction's a-priori bound does not hold at this size (plan_size): classes of such members only
    bool solo_differs = false;          // ... on another hierarchy than its solo run's (small ROIs whose level 1 a solo clone solves directly): within one grey level of it, not the same bytes
    int nl = 0, tail = 0;               // levels of its hierarchy; the level k_mg_tail holds (the one below it is solved directly)
    int npx = 0, npy = 0;               // padding of the directly solved level's operands (32 or 64 per side)
    int Kx = 0, Ky = 0, Kxp = 0, Kyp = 0, nx = 0, ny = 0, cells_y = 0, nxt = 0, nrs = 0;     // float-table correction
    double max_ratio = 0.0;
    // the heavy part, shared between copies (plans are memoised per size: sc_ragged.cpp): level geometries, the correction's ratio
    // table R[Kyp][Kxp] and the part maps of the 2- and the 4-sweep tiling
    struct Tables { std::vector<MGGeom> g; };
    std::shared_ptr<const Tables> t;
    // (what only the per-call setup needs -- the correction's ratio table, its part maps -- is computed into the staging by every call: rag_begin_table)
    // same compile-time choices and launch shapes (the spread of a group's sizes is plan_groups' business)
    bool same_class(const SizePlan &o) const
    {
        // (levels below the directly solved one are never visited: their number is free; operand paddings are per member:
        //  k_mg_tail_any; the correction's mode-block padding is the class's largest: a member's extra blocks are exact zeros)
        return ok && o.ok && tail == o.tail && conditional == o.conditional;
    }
};
bool plan_size(const sc_solver_opts &o, int W, int H, SizePlan &p);     // fills p; returns p.ok
// members (any order) -> groups that can each share one set of launches: a size class (two or more DIFFERENT sizes), a same-size
// group, or a single; `cap` = most members per group.  groups[k] lists indices into `plans`.
void plan_cache_clear();                                         // forgets every memoised plan and table (tests, measurements)
void pool_group_caps(int group, int n, int streams, int &cap, int &cap_max, long &budget_px);      // the pool's group policy (group 0: automatic)
void plan_groups(std::vector<SizePlan> &plans, int cap, std::vector<std::vector<int>> &groups, int cap_max = 0, long budget_px = 0);      // (may rewrite a member's plan: sc_ragged.cpp)

struct RagState {
    const RagMember *dev = nullptr;     // the members' table on the device WHILE a size class is being processed, else nullptr
    std::vector<RagMember> host;        // the same table (device pointers filled in)
    int n = 0;
    int nl = 0, tail = 0, npx = 0, npy = 0, Kxp = 0, Kyp = 0;     // the class's constants
    int max_nx = 0, max_ny = 0, max_nxt = 0, max_nrs = 0, max_cells_y = 0;
    double max_ratio = 0.0;
    DevBuf d_aux, h_stage;              // RagMember[n], R tables, part maps | Sx, Sy, bottom operands; pinned staging of what the host writes (the head of d_aux)
    bool pad_uniform = false;           // every member's operand padding of the directly solved level is the class's (k_mg_tail<SKX, SKY, true> instead of k_mg_tail_any)
    bool levels_built = false;          // the class's hierarchy is in I->mg (mg_build_levels_rag, called from rag_begin_builds)
    hipEvent_t ev = nullptr;            // behind the upload out of h_stage
    hipEvent_t ev_ready = nullptr;      // second stream: the level planes are zeroed and the correction's tables built (the matrices follow: Instance::ev_fd)
    bool ready_pending = false;         // ... and the main stream has not waited for that yet
};

struct Instance {
    uint32_t magic = 0x5C10E001u;
    int gpu = 0;
    hipStream_t stream = nullptr;
    // second stream of the instance: the float-table node correction of the next-to-last iterate (three latency-bound launches)
    // runs here beside the coarse levels of the last cycle; forked and joined with events, see mg_solve
    hipStream_t aux = nullptr;
    hipStream_t aux2 = nullptr;            // a size class's matrix build (rag_begin_builds), beside aux's zeroing and tables; created on first use
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool aux_pending = false;              // work on aux that `stream` has not waited for yet
    sc_solver_opts opts{};
    sc_run_info info{};
    std::string err;

    // staging copies for host-pointer runs
    DevBuf d_in, h_in;                             // a SMALL host-image call: mask | patch ROI | destination ROI packed in one pinned block, one copy, one device block
    DevBuf d_face, d_body_roi, d_mask, d_out;      // d_out: the host path's compact output buffer (ROI bytes that go back across PCIe)
    // page-locked host staging (grow-only): pageable caller images are packed here row by row so
    // each image crosses PCIe as ONE DMA instead of one slow pageable 2-D copy
    DevBuf h_face, h_body, h_mask, h_out;
    std::unique_ptr<RowCopier> copier;   // helper threads for the packing / splicing copies (created on first use)
    hipEvent_t ev_chunk[8]{};            // D2H chunk completions (host path): splice chunk k while k+1 is in flight
    // ROI mask after 3x erode
    DevBuf d_M;
    int mpitch = 0;
    // fields
    DevBuf d_U0, d_U1, d_F;
    Field U0, U1, F;      // current views into the buffers above
    bool result_in_U1 = false;
    bool out_direct = false;               // the last solve wrote output bytes from its last cycle: result(I) is the iterate before it, not the solution
    bool f_half = false;      // F currently holds float16 values (written by the pre-process for the fused multigrid path)
    bool u_half = false;      // ... and so does the initial field U0 until the first cycle has consumed it
    bool u_q16 = false;       // multigrid fast path, during a solve: the current field is 16-bit fixed point (sc_cycle0.hip, TAG bits 8, 9)
    bool mg_q16_last = false; // ... the last solve kept its field so (sc_hip_time_cycle0 times the same form)
    // 16-bit stores check their range: `sat` is the current solve's report word + generation (sc_common.h AbortFlag; p == nullptr:
    // the solve stores no 16-bit field); a solve whose word was set returns SC_RETRY_FLOAT_FIELD and the caller repeats the
    // clone with force_float_field set
    AbortFlag sat;
    unsigned sat_counter = 0;
    bool force_float_field = false;
    // Speculative epilogue: the multigrid driver enqueues the post-process right behind the cycle whose
    // convergence check it is about to wait for, so the host round trip of the check overlaps useful work.
    // If the check then fails the solve simply continues and the post-process runs again at the end.
    struct { uint8_t *body_org = nullptr; int bstep = 0; hipEvent_t ev_solved = nullptr; bool armed = false, done = false;
             std::vector<ImageJob> group; } spec_post;     // group: one destination per member (channels 3i..3i+2) of a group of clones
    // Speculative geometry: a clone may be launched on a predicted bounding box (the previous one for the same
    // mask size, else the whole mask interior) while the bbox kernel's answer is still in flight; `guard` makes
    // the post-process a no-op on a wrong guess and the host then repeats the clone (sc_api.cpp).
    RectGuard guard;
    int last_mc = -1, last_mr = -1, last_rect[4] = { 0, 0, 0, 0 };
    int spec_cooldown = 0;
    bool erode_done = false;   // the ROI device_clone is about to process has been eroded already (a clone launched on a predicted box)
    BboxTask pending_scan;     // ... and its bounding-box scan has not gone out yet: it rides in the pre-process launch
    bool scan_pending = false;
    bool scan_counter_dirty = false;   // a call failed with a HIP error: zero the scan's arrival counter before the next scan (hip_fail)
    // completion fence of a scan that rode in a pre-process launch: an event recorded right behind that launch, whatever the
    // stage marks do (a stage mark is not a fence: SC_FLAG_NO_STAGE_MARKS records none there); nullptr: no scan went out
    hipEvent_t ev_scan = nullptr, scan_fence = nullptr;
    bool bench_tag = false;   // sc_hip_field_time_sweeps: launch the second-symbol instantiations
    // multigrid hierarchy (level 0 aliases U0/U1/F)
    std::vector<DevBuf> mg_bufs;
    DevBuf mg_partial;    // per-block maxima of the level-0 correction
    DevBuf h_partial;     // pinned copy of the same (small grids are folded on the host: no reduction launch)
    std::vector<MGLevel> mg;
    size_t mg_bottom = 0;  // first level run by the fused bottom kernel (== mg.size(): none)
    bool mg_l1_half = false;   // level 1's planes currently hold float16 values (sc_multigrid.cpp: mg_level1_half; re-zeroed when the mode flips)
    // direct (fast-diagonalisation) solve inside the bottom kernel: level index relative to mg_bottom, or -1
    int fd_level = -1, fd_nxp = 0, fd_nyp = 0;
    DevBuf mg_fd;          // its matrices (built on the device, k_fd_build)
    bool fd_mm = false;    // ... the matrix-core form serves it (k_mg_bottom_mm): operands at float offset fd_mm_off, padded to fd_npx x fd_npy
    int fd_npx = 0, fd_npy = 0;
    size_t fd_mm_off = 0;
    hipEvent_t ev_fd_fork = nullptr, ev_fd = nullptr;   // the build runs on `aux`: started behind ev_fd_fork, finished at ev_fd
    bool fd_pending = false;                            // ... and `stream` has not waited for ev_fd yet
    LowMode lm;
    RagState rag;
    DstState dst;
    FftState fft;
    bool fft_lds_float = false, fft_lds_double = false;   // this instance's device has the FFT kernels opted in to > 64 KB of LDS (sc_fft.hip)
    // reductions / mailboxes
    DevBuf d_rects, h_rects;     // bounding boxes of a group of clones (sc_hip_run_device_batch): device, pinned
    DevBuf d_bbox_parts;         // per-workgroup extrema of the group's scans (folded by a second launch instead of atomics)
    hipEvent_t ev_rects = nullptr;   // behind the read-back of a group's bounding boxes
    int group_spec_cooldown = 0;     // calls left without a predicted bounding box after a wrong guess in a group
    int *d_rect = nullptr;
    int *h_rect = nullptr;       // pinned
    double *d_partials = nullptr;
    double *d_red = nullptr;
    double *h_red = nullptr;     // pinned
    unsigned *d_maxcorr = nullptr;     // max correction of the last cycle and of the one before (bits of a float), the solve's saturation word (four words allocated)
    unsigned *h_maxcorr = nullptr;     // pinned, the same
    hipEvent_t ev[8]{};
    bool stage_marks = true;   // record the stage marks (synchronous calls); tmark() in sc_api.cpp
    bool marks_ends_only = false;   // ... but only the first and the last one (SC_FLAG_NO_STAGE_MARKS)
    hipEvent_t tm[8]{};   // stage marks of the current run: ev[k], or the previous mark where a stage is empty (no record call)
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
    size_t arena_bytes = 0;
    std::vector<DevBuf> retired;           // device blocks that growth replaced: freed when the instance goes, or once they add up to 1 GB (ensure, sc_api.cpp)
    size_t retired_bytes = 0;
    // Small device buffers are pieces of a few SLABS (16 MB, then doubling) instead of hipMalloc blocks of their own: an instance owns
    // ~40 grow-only buffers, and a caller whose ROI sizes wander re-sizes several of them in one call -- each a hipMalloc of 30-100 us
    struct Slab { uint8_t *base = nullptr; size_t cap = 0, used = 0; };
    std::vector<Slab> slabs;

    bool ok() const { return magic == 0x5C10E001u; }
};

// a superseded launch form asked for (SC_FLAG_LEGACY_PATHS + sc_solver_opts.legacy_paths)
inline bool legacy_path(const sc_solver_opts &o, int which) { return (o.flags & SC_FLAG_LEGACY_PATHS) && (o.legacy_paths & which); }

// internal return code of a solve (never crosses the C ABI): a 16-bit fixed-point store saturated, no output was written;
// repeat the pre-process and the solve with Instance::force_float_field set
constexpr int SC_RETRY_FLOAT_FIELD = 1;

// error helper: records the message, returns SC_ERR_HIP
int hip_fail(Instance *I, hipError_t e, const char *what);
#define SC_HIP(I, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return hip_fail((I), e_, #call);            \
    } while (0)

int ensure(Instance *I, DevBuf &b, size_t bytes, bool zero = true);
int ensure_pinned(Instance *I, DevBuf &b, size_t bytes);
double fd_selftest_error();   // sc_multigrid.cpp
double fd_closed_selftest_error();
int setup_fields(Instance *I, int W, int H, int C);

// solver drivers (sc_solver.cpp) -- operate on I->U0/U1/F, leave the answer in result(I)
int solve(Instance *I);
bool mg_reads_half_rhs(const Instance *I);
bool mg_level1_half(const Instance *I);      // sc_multigrid.cpp: level 1's right-hand side and correction are stored as float16 in the solve configured in I
int mg_time_coarse_chain(Instance *I, int reps, float *ms_eager, float *ms_graph, int *launches);   // sc_multigrid.cpp
int mg_time_tail_phases(Instance *I, unsigned long long *out11);                                       // sc_multigrid.cpp
bool mg_composes_level1(const Instance *I);   // sc_multigrid.cpp   // sc_multigrid.cpp: would the solve configured in I->opts read a float16 F?
int lowmode_correct(Instance *I, const Field &U, const Field &Out);   // sc_lowmode.hip: Out = U + float-table correction
int lowmode_nodes(Instance *I, const Field &U, LmNodes &lm, hipStream_t on = nullptr);      // on: another stream than the instance's          // the correction of U at the node rows (what the post-process adds)
float4 *lowmode_bands_buffer(Instance *I, int sweeps);               // where a final level-0 launch leaves the correction's cell shares (nullptr: not wanted)
inline void field_moved(Instance *I) { I->lm.bands_of = nullptr; }   // anything that writes the solution field outside the judged multigrid launch calls this
int lowmode_part_map_selftest();                                     // host-only check of the parts-per-cell-row map against the launch geometry
int lowmode_early_kind(Instance *I, float update_tol);                // see sc_lowmode.hip
void lowmode_bands_written(Instance *I, const float *field);       // the launch went in: B describes `field` (nullptr: nothing)
int lowmode_count(int n);
bool lowmode_ratio(int w, int h, int Kx, int Ky, int Kxp, float *R, double &max_ratio);   // the correction's ratio table (R may be nullptr: statistics only); false: singular
bool lowmode_part_map(int H, int sweeps, std::vector<int> &m, int &band_rows);
bool lowmode_part_map(int H, int sweeps, int *m, int &band_rows);      // ... into 4 ints per cell row, all -1 on entry
int lowmode_projection_splits(int nxt, int nkb);                      // row splits of the coarse projection for a ROI with nxt column tiles
void launch_lm_tables_rag(const RagMember *rag, int members, int max_rows, int Kxp, int Kyp, hipStream_t s);
void mg_plan_levels(int W, int H, std::vector<MGGeom> &g);           // sc_multigrid.cpp
size_t mg_default_tail_level(const std::vector<MGGeom> &g);
int mg_build_levels_rag(Instance *I, hipStream_t zero_on);           // sc_multigrid.cpp: the class's level planes, zeroed on the given stream
int rag_begin_table(Instance *I, const std::vector<SizePlan> &members);      // the members' table and host tables, uploaded on the instance's stream (sets I->rag.dev)
int rag_begin_builds(Instance *I);                                           // ... then everything the device builds per call, on two more streams
void rag_end(Instance *I);
int dst_solve(Instance *I);                                           // sc_dst.hip: SC_METHOD_DST
int fft_solve(Instance *I, bool fp64);                                // sc_fft.hip: SC_METHOD_FFT (fp64: SC_FLAG_FFT_FP64)
bool fft_supported(int w, int h, bool fp64);
bool wants_float_tables(const Instance *I);
int effective_method(const Instance *I);                              // sc_solver.cpp: what SC_METHOD_AUTO resolves to for the fields bound to I
int output_nodes(Instance *I, LmNodes &lm);  // sc_solver.cpp: the float-table correction the post-process of result(I) has to add (none: lm.CN == nullptr)
int run_sweeps(Instance *I, int method, int sweeps, float omega, int sweeps_per_launch);
int fused_depth(int method, int sweeps_per_launch); // 0 = plain kernels
int eval_residual(Instance *I, double out[2]);
Field &result(Instance *I);
float optimal_omega(int W, int H);

} // namespace sc
