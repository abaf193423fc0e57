/*
 * seamlessclone_hip.h -- C ABI of libseamlessclone_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the seamless-clone (Poisson image editing, NORMAL_CLONE) hot path of
 * wujinzhong/seamlessCloneOptimization.  The four my_seamlessclone_api_imp_* symbols keep the
 * names the reference exports from seamlessClone-CUDA/seamlessclone_cuda.h:4-63 (thin
 * wrappers over seamlessClone_imp.cu:239,265,354,365) and that its Python binding re-declares
 * at seamlessClone-python-binding/SeamlessClone.cpp:37-44.  The reference passes cv::Mat* as
 * void* and returns a cv::Mat by value; a C ABI cannot, so each cv::Mat becomes the
 * {data, cols, rows, step} quadruple it wraps and the result is written in place into `body`
 * (the reference's returned Mat aliases the caller's dest buffer, seamlessClone_imp.cpp:470).
 *
 * Images are 8-bit, BGR interleaved (CV_8UC3) for face/body and single channel (CV_8UC1)
 * for mask, `step` = bytes per row.  All sc_hip_* entry points are additions: solver
 * options, a device-resident run for callers whose images already live in HBM, run
 * statistics, and stage-level hooks used by the parity tests.
 *
 * Threading: one instance <-> one HIP stream <-> one host thread at a time (reference:
 * one instance per stream, not re-entrant).  No process-global state.
 */
#ifndef SEAMLESSCLONE_HIP_H
#define SEAMLESSCLONE_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define SC_API __attribute__((visibility("default")))
#else
#define SC_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes (the reference aborts / asserts instead: seamlessClone_imp.cu:272-275,
 *      seamlessClone_imp.cpp:432-436,1013) */
#define SC_OK                 0
#define SC_ERR_BAD_ARG       -1   /* null pointer, bad instance, bad option value          */
#define SC_ERR_BAD_SIZE      -2   /* face/mask size mismatch, step too small, empty image  */
#define SC_ERR_EMPTY_MASK    -3   /* bbox of mask!=0 degenerate (reference assert :1013)   */
#define SC_ERR_ROI_OOB       -4   /* ROI leaves body (unchecked in the reference)          */
#define SC_ERR_HIP           -5   /* HIP runtime error; see sc_hip_last_error()            */
#define SC_ERR_NOT_CONVERGED -6   /* tol not reached within max_sweeps; result still written */

/* OpenCV clone flags (only NORMAL_CLONE exists in the reference: seamlessClone_imp.cu:301) */
#define SC_NORMAL_CLONE 1

/* ---- solver selection */
enum sc_method {
    SC_METHOD_JACOBI = 0,    /* U' = 1/4 (l+r+u+d - lap), ping-pong                         */
    SC_METHOD_RBGS   = 1,    /* red-black Gauss-Seidel (omega = 1)                          */
    SC_METHOD_SOR    = 2,    /* red-black SOR, omega from opts (<=0: optimal for the ROI)   */
    SC_METHOD_MULTIGRID = 3, /* V-cycles with red-black GS smoothing (converges at any ROI)  */
    SC_METHOD_DST    = 4,    /* the reference's direct solve (seamlessClone_imp.cpp:1814-1896; matrix form :1266-1334):
                                u = S_h ((S_h g S_w) / den) S_w with den = filter_X + filter_Y - 4 from the float
                                tables of :596-599, four double-precision products on the matrix cores.  O(n^3):
                                milliseconds at 2048^2; the non-iterative cross-check of the default path          */
    SC_METHOD_AUTO   = 5,    /* DEFAULT.  A direct solve -- SC_METHOD_FFT with double transforms -- for ROIs of at most
                                SC_AUTO_DIRECT_MAX unknowns per side (there it is faster than the cycles and has no iteration
                                error: diff sum against the float-table CPU port 2 instead of 129 at the reference's 300x194
                                patch, whose own published deviation from OpenCV is 44, PDF p3) and for thin ROIs
                                (SC_AUTO_THIN_MAX) -- single clones only: a group of clones (sc_hip_run_device_batch) is about
                                throughput, where the cycles win at every size --, SC_METHOD_MULTIGRID above and whenever
                                tol > 0 asks for a residual-based stop.  sc_run_info.method says which one ran.                                      */
    SC_METHOD_FFT    = 6     /* the reference's DEFAULT direct back-end (poissonSolver2D_FFT, seamlessClone_imp.cpp:1694-1918):
                                the same u = S_h ((S_h g S_w) / den) S_w in float32 with FFT-based transforms, O(n^2 log n).
                                The DST-I of each row is a chirp-z transform over a power-of-two FFT held in LDS (sc_fft.hip);
                                at most 8192 unknowns per side (SC_ERR_BAD_SIZE beyond).  float32 like cuFFT / OpenCV's dft:
                                within +-1 of the float-table port, diff sums of the size the reference publishes for its
                                own cuFFT path against OpenCV (PDF p3)                                           */
};
#define SC_AUTO_DIRECT_MAX 720        /* round 4 (900 in round 3): the cycles got faster (0.222 against 0.231 ms at 750^2, 0.228 against 0.268 at 900^2) */
#define SC_AUTO_DIRECT_AREA 450000   /* round 4: ... also where the ROI has at most this many unknowns in all, whatever its shape (an elongated ROI's
                                       transforms are short in one direction: 900 x 100 takes 0.10 ms directly, 0.19 in cycles), ... */
#define SC_AUTO_NARROW_MAX 140      /* ... and where it is at most this many unknowns across -- both up to SC_AUTO_THIN_LONG_MAX along (the double
                                       transform's limit): 4000 x 130 takes 0.37 ms directly, 0.45 in cycles                          */
#define SC_AUTO_THIN_MAX 4          /* ... and for thin ROIs (at most this many unknowns across, up to SC_AUTO_THIN_LONG_MAX along) */
#define SC_AUTO_THIN_LONG_MAX 4096

typedef struct sc_solver_opts {
    int   method;            /* enum sc_method                                              */
    int   max_sweeps;        /* sweeps (JACOBI/RBGS/SOR) or V-cycles (MULTIGRID) budget; MULTIGRID
                                reports SC_ERR_NOT_CONVERGED when the budget ends first         */
    float tol;               /* stop when ||lap - A u||_2 / ||lap||_2 <= tol; <=0: run
                                exactly max_sweeps                                         */
    int   check_every;       /* sweeps between residual checks (tol>0)                      */
    float omega;             /* SOR relaxation; <=0 -> optimum for the rectangle,
                                2/(1+sqrt(1-rho^2)), rho = (cos(pi/(w+1))+cos(pi/(h+1)))/2            */
    int   sweeps_per_launch; /* 0: library default (register-blocked fused kernels, deepest depth);
                                1: one sweep per launch with the plain kernels; -1: fused kernel, depth 1;
                                >=2: fused kernel at that depth (Jacobi 1-4, 6, 8; red-black 1-2).  All variants
                                give bit-identical fields.                                          */
    int   reference_warmup;  /* 1: clone twice in place, as the reference's run() does
                                (warm-up + 1, seamlessClone_imp.cu:303-318)                */
    int   mg_pre, mg_post;   /* multigrid smoothing sweeps per level (0 = default 2/2)      */
    float update_tol;        /* MULTIGRID stop rule, in grey levels.  The error a V-cycle leaves is
                                ~rho/(1-rho) times the largest coarse-grid correction it applied to
                                the ROI (rho = contraction per cycle, 0.05-0.1).  From the third cycle
                                on the driver measures rho from two successive corrections and stops
                                once that predicted error is <= 0.1 x update_tol; with only one
                                correction known (max_sweeps = 1) it stops once the correction itself
                                is <= update_tol.  Default 0.25 (error <= 0.025): normally 3 cycles,
                                max |delta| 1 on 0.004-0.1 % of channels vs the exact solution -- below
                                the reference's own float32 deviation from OpenCV (0.16 % at 2400x1552,
                                PDF p3).  0.02 costs one or two more cycles.
                                The float32 residual norm stalls earlier and is only reported. */
    int   flags;             /* SC_FLAG_* bits below; 0 = the default paths                  */
    int   jacobi_tile_rows;  /* single-sweep Jacobi launches (sweeps_per_launch = 1): 0 = rows rolling through
                                registers (k_jacobi_roll, default); 16, 32 or 64 = the LDS-staged 256 x rows tile
                                with a 1-pixel halo (k_jacobi<rows>).  Bit-identical fields.              */
    int   mg_level1_sweeps;  /* multigrid, default schedule (level 1 without post-smoothing, SC_FLAG_NO_COMPOSE_L1 clear): sweeps
                                level 1 does before its restriction; 0 = default (4), 2..4                        */
    int   mg_direct_max;     /* multigrid bottom kernel: the first level whose sides are both at most this many unknowns is solved
                                directly (fast diagonalisation); the LDS-resident levels above it cycle.  0 = default
                                (SC_MG_DIRECT_MAX_DEFAULT); at most 128.  Same fixed point, slightly different iterates        */
    int   legacy_paths;      /* read only with SC_FLAG_LEGACY_PATHS: SC_LEGACY_* bits, the superseded launch forms to run instead of
                                the defaults (A/B measurements and cross-checks of kernels whose decision is made)              */
} sc_solver_opts;
#define SC_MG_DIRECT_MAX_DEFAULT 128

/* ---- sc_solver_opts.flags: non-default variants of the same path, selectable per instance so one process (one
 *      test run) can drive every variant.  Unless noted a variant gives the default path's result bit for bit. */
#define SC_FLAG_NO_SPECULATE   (1 << 0)  /* wait for the device's bounding box before launching the clone (the
                                            reference's order, seamlessClone_imp.cpp:1012) instead of launching on
                                            a predicted box                                                      */
#define SC_FLAG_FLOAT_RHS      (1 << 1)  /* multigrid: keep the right-hand side as float32 (default: float16,
                                            exact -- it is an integer in [-1020, 1020]) AND level 1's right-hand side and
                                            correction as float32 (default: float16 -- a correction scheme tolerates it, the
                                            fixed point is level 0's).  Same fixed point, iterates a relative 5e-4 of a
                                            correction apart: results within one grey level of the default's, not bit-identical */
#define SC_FLAG_FLOAT_U0       (1 << 2)  /* multigrid: initial field as float32 (default: float16, exact -- 8-bit
                                            values)                                                             */
#define SC_FLAG_NO_COMPOSE_L1  (1 << 3)  /* multigrid: level 1 gets its own post-smoothing launch (textbook V(2,2));
                                            default: the level-0 launch composes its prolongation source from
                                            levels 1 and 2.  Same fixed point, slightly different iterates      */
#define SC_FLAG_VCYCLE_BOTTOM  (1 << 4)  /* multigrid: cycle down to the coarsest level inside the bottom kernel
                                            instead of solving its first level directly (fast diagonalisation).
                                            Same fixed point, slightly different iterates                       */
#define SC_FLAG_EXACT_TABLES   (1 << 5)  /* MULTIGRID / DST: return the exact solution of the 5-point
                                            system (DST: double denominators 2cos + 2cos - 4 instead of the float tables).  Default: the answer OpenCV and the reference compute, whose
                                            eigenvalue tables are stored and combined in float32
                                            (seamlessClone_imp.cpp:596-599, :1651-1653) -- see DESIGN.md sec. 5,
                                            "float-table correction"                                            */
#define SC_FLAG_LEGACY_PATHS   (1 << 6)  /* run the superseded launch forms named in sc_solver_opts.legacy_paths (round 5: one switch for the
                                            A/B scaffolding of decisions that are made; rounds 2-4 had a public flag for each):           */
#define SC_LEGACY_SEPARATE_RESTRICT 1    /*   float-table correction: the hat-weighted cell sums of the field come from a pass of their own
                                            over it (k_lm_restrict); default: the level-0 multigrid launch that writes the field leaves them
                                            behind.  Same cells, another order of the additions (differences at float rounding level)     */
#define SC_LEGACY_BOTTOM_F32   2         /*   multigrid: the bottom kernel's direct solve as float32 SIMD inner products with every operand
                                            staged in LDS (k_mg_bottom, rounds 1-3) on the hierarchy of those rounds.  Default since round 4
                                            where the bottom's first level has at most 96 unknowns per side: four products on the matrix
                                            cores in float32 (v_mfma_f32_32x32x2_f32: k_mg_bottom_mm).  Same arithmetic up to the order of
                                            the additions                                                                               */
#define SC_LEGACY_SEPARATE_TAIL 4        /*   multigrid: the level above the bottom and the bottom as the three launches of rounds 1-3
                                            (pre-smoothing + residual + restriction, direct solve, prolongation + post-smoothing).  Default
                                            since round 4 where that level has at most 127 unknowns per side: ONE launch, the level in
                                            registers (k_mg_tail).  Same arithmetic per point                                            */
#define SC_FLAG_KEEP_FIELD     (1 << 7)  /* sc_hip_run*: keep the solution field on the device (sc_hip_field_store,
                                            _residual, _finish after a run): the last multigrid cycle writes the field
                                            and a post-process launch reads it.  Default: that cycle writes the output
                                            bytes itself and no final field exists (those hooks then fail with
                                            SC_ERR_BAD_ARG); the float-table node correction it adds is the one of the
                                            iterate one cycle earlier (difference at most 0.05 grey levels in the worst case the stop rule admits,
                                            0.001-0.003 measured; ROIs where
                                            that bound does not hold take this flag's path by themselves)        */

#define SC_FLAG_FFT_FP64       (1 << 8)  /* SC_METHOD_FFT: the transforms in double instead of float32 (tables, LDS and the planes
                                            between the launches); at most 4096 unknowns per side.  No transform rounding is left:
                                            the result agrees with SC_METHOD_DST (both are the reference's float-table arithmetic
                                            with exact transforms)                                                          */

#define SC_FLAG_OPENCV_GREY_MASK (1 << 9) /* masks that are not 0 / 255: OpenCV's semantics -- the three erodes are minimum filters (a grey
                                            eroded mask) and the gradients are blended with the fractional weights M/255 and
                                            (255 - M)/255 (OpenCV 3.4.5 modules/photo, Cloning::computeDerivatives / normalClone).
                                            Default: the reference's -- it thresholds (seamlessClone_imp.cpp:917, sum == 255 * 9:
                                            every value below 255 erodes to 0) and so only ever blends with 0 / 1.  On 0 / 255 masks
                                            the two are bit-identical.  The BOUNDING BOX stays the reference's under this flag: all non-zero
                                            pixels (seamlessClone_imp.cpp:943, `mask != 0`), i.e. OpenCV's erode and blend weights on the
                                            reference's ROI; cv::seamlessClone itself takes the box of the pixels equal to 255 only, so for
                                            a feathered mask its ROI (and Dirichlet ring) is smaller than the one used here.
                                            PARITY UNPINNED: OpenCV's source is not part of the
                                            reference and none of its fixtures holds a grey mask; checked against a restatement
                                            of the published algorithm (oracle/).  Groups run one clone at a time with it.   */

#define SC_FLAG_FLOAT_L1       (1 << 10) /* multigrid: level 1's right-hand side and correction as float32 (default on the fast path:
                                            float16, see SC_FLAG_FLOAT_RHS).  Same fixed point, slightly different iterates.
                                            Implies SC_FLAG_FLOAT_FIELD                                                        */
#define SC_FLAG_FLOAT_FIELD    (1 << 11) /* multigrid: the field between the level-0 launches always as float32.  Default on the fast
                                            path (float16 right-hand side and level 1, output bytes from the last cycle): the first
                                            stores of a solve -- all but the one the judged cycle reads -- are 16-bit fixed point,
                                            code = trunc(64 u + 16384.5) in [0, 65535], i.e. [-256, 768) in steps of 1/64.  A clone's
                                            solution lies in [-255, 510], its 8-bit boundary values are exact, and a rounding of
                                            <= 1/128 two cycles before the output is gone by then (each cycle removes 90 % of any
                                            error).  Takes 2 bytes per unknown off four of the six field transfers between a
                                            solve's level-0 launches                                                           */

#define SC_FLAG_NO_STAGE_MARKS  (1 << 12) /* sc_hip_run_device(..., bSync = true) and the host-image call: record only the first and the last stage mark.  Every
                                            mark is an event in the stream with a ~5 us bubble behind it, so the per-stage timeline
                                            (ms_mask, ms_pre, ms_solve) costs a 2048^2 clone ~15 us; with this flag ms_device_total is the
                                            un-instrumented device time of the clone and the per-stage figures read 0 (all of it is booked
                                            under ms_post).  Results are unaffected                                              */

#define SC_FLAG_ROWS_RETURN    (1 << 13) /* host-image call, OPT-IN (round 5; the default of late round 4): a destination without row padding whose
                                            ROI covers most of its rows gets those ROWS back as one linear copy straight into the caller's image
                                            (0.05-0.1 ms faster at 2048^2) -- the pixels of those rows OUTSIDE the ROI are rewritten with the
                                            values they had when the call started, so nothing else may write them during the call (another
                                            thread cloning into a disjoint ROI of the same image would lose its result).  Default: the result
                                            comes back as the compact ROI through pinned staging and is spliced into the caller's rows -- only
                                            ROI bytes of the caller's image are ever written, as in the reference (seamlessClone_imp.cpp:470-483) */

#define SC_FLAG_POISON_ARENA   (1 << 14) /* testing: every device block the arena hands out -- or hands out AGAIN -- WITHOUT zeroing it (fields,
                                            level planes, image staging: "written before they are read") is filled with 0xFF bytes first -- NaN as float32 and
                                            float16 -- which is what RECYCLED device memory may hold (fresh memory reads as zero and hides a
                                            read of something never written).  Results must not change (tests/test_gpu_round5.py)            */

/* ---- statistics of the last run */
typedef struct sc_run_info {
    int    x0, y0, W, H, ltx, lty;  /* patch offset, ROI size (ring included), ROI origin in body */
    int    sweeps;                  /* sweeps (or V-cycles) executed                        */
    int    converged;               /* 1 when tol reached (or tol<=0)                       */
    double rel_residual;            /* last evaluated ||r||/||lap|| (NaN if never evaluated) */
    float  ms_h2d, ms_mask, ms_pre, ms_solve, ms_post, ms_d2h; /* hipEvent times on the instance stream; the multigrid
                                   driver enqueues the post-process directly behind its last cycle, without a mark between
                                   them: ms_post is then 0 and ms_solve includes it.  sc_hip_run_device with
                                   bSync = false records no marks (each is an event in the stream) and reports 0 */
    float  ms_device_total;         /* mask + pre + solve + post                            */
    int    sweep_launches;          /* launches of the dominant sweep kernel in the last run */
    float  last_update;             /* MULTIGRID: max |coarse-grid correction| of the last checked cycle (grey levels) */
    size_t device_bytes;            /* arena bytes owned by the instance                    */
    int    method;                  /* enum sc_method that ran (what SC_METHOD_AUTO resolved to) */
    int    device;                  /* HIP device the instance runs on (create_instance's gpu_id)  */
    float  ms_call;                 /* my_seamlessclone_api_imp_run: hipEvent time of the whole call's stream work, first upload
                                       to last download (what the reference's bSync timing brackets, seamlessClone_imp.cu:310-344) */
    int    field_retry;             /* 1: the 16-bit fixed-point field of the multigrid fast path saturated during this clone (an
                                       iterate left [-256, 768): possible when the mask mixes patch and destination gradients into a
                                       non-conservative field) -- nothing was written, the clone was repeated on float32 fields */
    int    new_size;                /* 1: this run built per-size state (multigrid hierarchy, transform or correction tables) */
    int    group_members;           /* sc_hip_run_device_batch: members of the last set of launches the call shared (0: none were shared) */
    int    group_ragged;            /* ... 1: those members had DIFFERENT ROI sizes (a size class, round 5): W, H above are the class's largest */
} sc_run_info;

/* ---- the reference's four entry points ------------------------------------------------- */

/* seamlessclone_cuda.h:23-38 / seamlessClone_imp.cu:239-263.  Selects `gpu_id` (the
 * reference only prints its properties), creates the stream and the grow-only arena.
 * Returns NULL on failure. */
SC_API void *my_seamlessclone_api_imp_create_instance(int gpu_id);

/* seamlessclone_cuda.h:6-21 / seamlessClone_imp.cu:265-352.
 * face = patch (CV_8UC3), body = destination (CV_8UC3, modified in place), mask (CV_8UC1,
 * same size as face).  Host pointers (pageable or page-locked).  The call completes before it
 * returns whatever bSync says, because the result has to land in caller memory (the reference
 * is synchronous here as well: D2H + host splice, seamlessClone_imp.cpp:471-483).  bSync = true
 * does what the reference's does (seamlessClone_imp.cu:310-349): the call is timed with events on
 * the instance's stream and prints the reference's two lines on stdout,
 *     "Compute stage performance time= %.3f msec, patch size=%dx%d" and "total device memory used: %d";
 * the reference's Python binding passes false (SeamlessClone.cpp:63), its CLI true (seamlessClone_main.cu:91).
 * The same time is in sc_run_info.ms_call either way.  Returns SC_OK or a negative SC_ERR_*. */
SC_API int my_seamlessclone_api_imp_run(void *instance,
                                 const uint8_t *face, int face_cols, int face_rows, int face_step,
                                 uint8_t *body, int body_cols, int body_rows, int body_step,
                                 const uint8_t *mask, int mask_cols, int mask_rows, int mask_step,
                                 int centerX, int centerY, int gpu_id, bool bSync);

/* seamlessclone_cuda.h:40-55 / seamlessClone_imp.cu:354-363 */
SC_API void my_seamlessclone_api_imp_destroy(void *instance);

/* seamlessclone_cuda.h:57-61 / seamlessClone_imp.cu:365-370 */
SC_API void my_seamlessclone_api_imp_sync(void *instance);

/* ---- additions ------------------------------------------------------------------------- */

SC_API void sc_hip_default_opts(sc_solver_opts *opts);
SC_API int  sc_hip_set_solver(void *instance, const sc_solver_opts *opts);
SC_API int  sc_hip_get_solver(void *instance, sc_solver_opts *opts);
SC_API int  sc_hip_get_info(void *instance, sc_run_info *info);
SC_API const char *sc_hip_last_error(void *instance);

/* Same as run(), but face/body/mask are DEVICE pointers on the instance's GPU (inputs
 * resident in HBM); body is updated in place on the device.  Asynchronous on the instance
 * stream unless bSync. */
SC_API int sc_hip_run_device(void *instance,
                      const uint8_t *d_face, int face_cols, int face_rows, int face_step,
                      uint8_t *d_body, int body_cols, int body_rows, int body_step,
                      const uint8_t *d_mask, int mask_cols, int mask_rows, int mask_step,
                      int centerX, int centerY, bool bSync);

/* plain device-memory helpers so non-HIP hosts (ctypes, cgo, JNI) can stage images */
SC_API void *sc_hip_malloc(void *instance, size_t bytes);
SC_API void  sc_hip_free(void *instance, void *dptr);
SC_API int   sc_hip_memcpy_h2d(void *instance, void *dptr, const void *hptr, size_t bytes);
SC_API int   sc_hip_memcpy_d2h(void *instance, void *hptr, const void *dptr, size_t bytes);
SC_API int   sc_hip_memcpy_d2d_async(void *instance, void *dst, const void *src, size_t bytes); /* on the instance stream */
SC_API int   sc_hip_device_count(void);
/* PCI address of HIP device `gpu_id` as "dddd:bb:dd.f" (what /sys/bus/pci/devices/ is keyed by: a host that pins its threads to
 * the cores local to a GPU reads <that directory>/local_cpulist).  Returns SC_OK, or SC_ERR_BAD_ARG / SC_ERR_HIP with buf[0] = 0. */
SC_API int   sc_hip_device_pci_bus_id(int gpu_id, char *buf, int len);
/* page-locked host memory for callers that want their images DMA-able in place (run() copies a page-locked image
 * whose row step equals the library's device pitch without staging; every other host image is packed first) */
SC_API void *sc_hip_host_alloc(void *instance, size_t bytes);
SC_API void  sc_hip_host_free(void *instance, void *hptr);

/* ---- stage-level hooks (parity tests drive each kernel through these) ------------------- */

/* mask stage only (seamlessClone_imp.cpp:978-1071): geo = {x0,y0,W,H,ltx,lty}; M_out
 * receives the 3x eroded ROI mask, dense W*H bytes (may be NULL). */
SC_API int sc_hip_mask_stage(void *instance, const uint8_t *mask, int mask_cols, int mask_rows, int mask_step,
                      int centerX, int centerY, int geo[6], uint8_t *M_out, size_t M_capacity);

/* mask stage + fused pre-process (seamlessClone_imp.cpp:1920-2018): downloads the dst-ROI
 * field B and the un-folded RHS lap, planar [3][H][W] float32, channel = BGR index. */
SC_API int sc_hip_build_rhs(void *instance,
                     const uint8_t *face, int face_cols, int face_rows, int face_step,
                     const uint8_t *body, int body_cols, int body_rows, int body_step,
                     const uint8_t *mask, int mask_cols, int mask_rows, int mask_step,
                     int centerX, int centerY, int geo[6], float *B_out, float *lap_out, size_t plane_capacity);

/* solver-only hooks on caller-supplied fields, planar [C][H][W] float32 (ring included). */
SC_API int sc_hip_field_load(void *instance, int W, int H, int C, const float *U, const float *lap);
SC_API int sc_hip_field_sweep(void *instance, int method, int sweeps, float omega, int sweeps_per_launch);
SC_API int sc_hip_field_residual(void *instance, double out[2] /* sum r^2, sum lap^2 */);
SC_API int sc_hip_field_solve(void *instance);                      /* run the configured solver on the loaded field */
SC_API int sc_hip_field_shape(void *instance, int whc[3]);   /* W, H, C of the fields currently on the device */
SC_API int sc_hip_field_store(void *instance, float *U_out, size_t capacity_floats);

/* post-process alone (seamlessClone_imp.cpp:2078-2103 and the host splice :470-483) on the field currently on the
 * device (sc_hip_field_load, or what a solve left): clamp to [0,255], truncate, interleave the interior of the
 * 3-channel field into the host image `body` with the ROI origin at (ltx, lty). */
SC_API int sc_hip_field_finish(void *instance, uint8_t *body, int body_cols, int body_rows, int body_step, int ltx, int lty);
/* float-table correction alone on the field currently on the device: the result becomes
 * result + correction, i.e. the exact solution of the 5-point system turns into what the reference's float32
 * eigenvalue tables give (seamlessClone_imp.cpp:596-599, :1651-1653). */
SC_API int sc_hip_field_lowmode(void *instance);

/* microbenchmark hook used by bench.py: runs `launches` launches of the sweep kernel
 * (method, sweeps_per_launch) on the loaded field and returns the mean launch time measured
 * with hipEvents on the instance stream. */
SC_API int sc_hip_field_time_sweeps(void *instance, int method, int launches, int sweeps_per_launch, float omega,
                             float *ms_per_launch);

/* ---- native batch driver: K instances (HIP streams) on one GPU, one host thread each ----------
 * Clones are independent; several in flight hide one another's latency-bound phases.  Jobs are
 * pulled from a shared counter, each runs exactly once, sc_hip_pool_run returns when all are done. */
typedef struct sc_batch_job {
    const uint8_t *face; int face_cols, face_rows, face_step;
    uint8_t *body;       int body_cols, body_rows, body_step;
    const uint8_t *mask; int mask_cols, mask_rows, mask_step;
    int centerX, centerY;
    const uint8_t *body_restore;   /* device-resident batches only: if non-NULL, body is refreshed from
                                      this device image (body_step * body_rows bytes) before the clone */
    int rc;                        /* out: SC_OK or SC_ERR_* of this job */
} sc_batch_job;
/* n device-resident clones on ONE instance.  Members whose ROIs have the same size (W x H; masks, positions and
 * images are free) are solved as one field of 3n channels: every solver launch is n times larger and there is one
 * set of launches for the group, which is what fills a 256-CU GPU with small and medium ROIs.  Results are the ones the
 * clones get one by one (channels never interact), except that the stop rule sees the group's largest correction, so
 * every member gets the cycle count of the slowest.  The group is launched on predicted bounding boxes (the masks'
 * interiors); a member whose box turns out different is left untouched by the group and repeated alone.
 * Round 5: the members of a call are PARTITIONED -- same-size members share launches as above; members of one SIZE CLASS
 * (different sizes whose solves are the same program: same hierarchy depth and bottom solve, widths and heights within 2x of
 * each other; default solver options) share them through a per-member geometry table the kernels read (csrc/sc_ragged.cpp),
 * each member with the bytes of its solo run; what fits neither (and a failing member) runs alone.  jobs[i].rc receives each
 * clone's code; the call is asynchronous like sc_hip_run_device(..., false): sync the instance before reading bodies. */
SC_API int   sc_hip_run_device_batch(void *instance, sc_batch_job *jobs, int n);
SC_API void *sc_hip_pool_create(int gpu_id, int streams);
SC_API void  sc_hip_pool_destroy(void *pool);
SC_API int   sc_hip_pool_size(void *pool);
SC_API void *sc_hip_pool_instance(void *pool, int k);          /* instance k, e.g. for sc_hip_get_info */
SC_API int   sc_hip_pool_set_solver(void *pool, const sc_solver_opts *opts);
SC_API int   sc_hip_pool_run(void *pool, sc_batch_job *jobs, int n, int device_resident);
/* device-resident batches: every worker takes up to `group` jobs at a time -- the batch's jobs bucketed by ROI size: same-size
 * jobs and jobs of one size class -- and runs them through sc_hip_run_device_batch (default 1 = one clone per set of launches; at most 64).
 * SC_POOL_GROUP_AUTO: sixteen at least where the batch has them, more for small ROIs -- up to half the batch (two groups at a time
 * are what pays: more streams launching small kernels at once only contend) and 64, while a group's fields stay within what sixteen 2048 x 2048 members occupy (small clones are latency bound:
 * 64 clones of 120..190 pixels take 0.43 ms in two groups of 32, 0.64 in four of 16) */
#define SC_POOL_GROUP_AUTO 0
SC_API int   sc_hip_pool_set_group(void *pool, int group);

/* isolated timing of the fused level-0 multigrid cycle kernel on the state left by the last
 * MULTIGRID run (values are discarded; bench.py roofline) */
SC_API int sc_hip_time_cycle0(void *instance, int launches, float *ms_per_launch);
/* ... and of the other three level-0 launches a fast-path solve is made of, each under a second symbol of its own:
 * form 0 = the full cycle (as sc_hip_time_cycle0), 1 = the full cycle before the judged one (16-bit field in, float out, leaves the
 * float-table correction's cell shares), 2 = the judged cycle (two sweeps, output bytes), 3 = the first launch of a solve (two
 * sweeps from the float16 initial field, no prolongation).  SC_ERR_BAD_ARG unless the last run was a default multigrid solve. */
SC_API int sc_hip_time_cycle0_form(void *instance, int form, int launches, float *ms_per_launch);
/* measurement: the launch-bound part of a multigrid cycle (levels 2 .. bottom .. 2 of the hierarchy the last multigrid run left,
 * `*launches` dependent launches) `reps` times as plain launches and as replays of ONE captured HIP graph: ms per pass of each */
SC_API int sc_hip_time_coarse_chain(void *instance, int reps, float *ms_eager, float *ms_graph, int *launches);
/* measurement: the shader clock at the eleven phase boundaries of ONE k_mg_tail launch (the level above the bottom and the bottom in one
 * launch, SC_FLAG_SEPARATE_TAIL) on the hierarchy the last multigrid run left: entry | right-hand side loaded | pre-smoothing | residual +
 * restriction | the four products of the direct solve | prolongation | post-smoothing | stores issued.  SC_ERR_BAD_ARG unless that
 * hierarchy runs its bottom this way. */
SC_API int sc_hip_time_tail_phases(void *instance, unsigned long long *cycles11);

/* Host-only (needs no GPU): how sc_hip_run_device_batch / the pool would partition a batch whose members have these ROI sizes
 * (wh[2i], wh[2i+1]: width and height, ring included) under `opts` (NULL: the defaults), at most `cap` members per group (<= 0: no
 * limit): group_of[i] = the member's group, kind_of[i] (may be NULL) = 0 alone, 1 a same-size group, 2 a size class (different
 * sizes, the same solve: csrc/sc_ragged.cpp; the member's bytes are those of its solo run), 3 a size class on another hierarchy than
 * the member's solo run takes (a small ROI, or the leftover of a class moved onto the next deeper one: within one grey level of the
 * solo run).  Returns the number of groups, or SC_ERR_BAD_ARG. */
/* Host-only: what decides a ROI size's class: out = { eligible, levels, level held by k_mg_tail (THE class key, beside the 2x spread),
 * operand padding x, y of the level solved directly, mode-block padding x, y of the correction, its column tiles, its row splits,
 * 1000 * nx + ny of the level solved directly, solo_differs (1: a small ROI whose level 1 a solo clone solves directly -- inside a
 * class it runs the general hierarchy and comes out within one grey level of its solo run instead of with its bytes), conditional
 * (1: the float tables' lowest modes are off by more than 4 % at this size; such members form classes of their own, in which the
 * judged cycle's measured update decides the output's form for the whole group) } */
/* ... and how sc_hip_pool_run would: a pool of `streams` workers with group size `group` (SC_POOL_GROUP_AUTO allowed), jobs handed
 * to the planner largest first */
SC_API int sc_hip_plan_groups_pool(const int *wh, int n, int group, int streams, const sc_solver_opts *opts, int *group_of, int *kind_of);
SC_API int sc_hip_plan_size(int W, int H, const sc_solver_opts *opts, int out[12]);
/* Host-only: plans and per-size host tables (the float-table correction's ratio table, its part maps) are pure functions of the ROI
 * size and memoised process-wide on first use -- 5-20 us of host arithmetic per NEW size, paid inside the first batch call that meets
 * it.  A caller that knows its sizes ahead (a set of patch templates, the boxes of the previous frame) moves that out of its latency
 * path: prepares wh[2i], wh[2i+1] (ring included) under `opts` (NULL: the defaults); returns how many of them can join a size class.
 * sc_hip_plan_cache_clear forgets everything memoised (tests and measurements). */
SC_API int sc_hip_plan_prepare(const int *wh, int n, const sc_solver_opts *opts);
SC_API void sc_hip_plan_cache_clear(void);
SC_API int sc_hip_plan_groups(const int *wh, int n, int cap, const sc_solver_opts *opts, int *group_of, int *kind_of);

/* Host-only (needs no GPU): 1 when the reference's float32 eigenvalue tables are singular for an ROI of w x h unknowns --
 * (float)(2 cos(PI/(n+1))) is exactly 2.0f in both directions (n >= ~12 870), so the reference's denominator
 * filter_X[0] + filter_Y[0] - 4 (seamlessClone_imp.cpp:1651-1653) is zero and its result undefined.  For such ROIs the
 * default path and SC_METHOD_DST return the exact system's solution (as SC_FLAG_EXACT_TABLES does). */
SC_API int sc_hip_reference_tables_singular(int w, int h);

/* host-only self test (needs no GPU): the parked-thread row copier of the host path and the tridiagonal
 * eigen-solver behind the direct bottom solve (residual of T V = V L for level operators with an irregular
 * last interval).  Returns 0, or the number of the check that failed. */
SC_API int sc_hip_selftest_host(void);

#ifdef __cplusplus
}
#endif
#endif /* SEAMLESSCLONE_HIP_H */
