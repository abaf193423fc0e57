// sc_api.cpp -- the extern "C" boundary of libseamlessclone_hip.so (include/seamlessclone_hip.h).
//
// Host orchestration of one clone (reference call stack: seamlessClone_imp.cu:265-352 ->
// seamlessClone_imp.cpp:430-486 seamlessCloneGPU -> :2105-2135 run()):
//   H2D mask -> bbox kernel -> 16-byte read-back (the one mid-pipeline sync the reference also
//   has, :1012) -> H2D of the face/body ROI only -> fused erode -> fused pre-process ->
//   iterative solve -> fused post-process into the body ROI -> D2H of the interior straight
//   into the caller's image (replaces the reference's D2H + host splice loop, :470-483).
#include "sc_instance.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

using namespace sc;

namespace sc {

int hip_fail(Instance *I, hipError_t e, const char *what)
{
    if (I) {
        I->err = std::string(what) + ": " + hipGetErrorString(e);
        I->scan_counter_dirty = true;      // a launch that never completed may have left the scan's arrival counter non-zero
    }
    return SC_ERR_HIP;
}

// Grow-only, amortised (the reference's SCImage::resize, seamlessClone_imp.h:83,119-121,137-149).  Round 5: growth stays off the
// stream's critical path -- the new block is allocated FIRST, with no wait on the stream, and the block it replaces is RETIRED, not
// freed: launches already queued keep reading and writing it, and hipFree (a device-wide synchronisation that also stalls the other
// instances of a pool) happens once, when the instance is destroyed.  Capacities double, so the retired blocks of a buffer add up to
// less than its final size.  (Rounds 1-4: stream synchronisation + hipFree + hipMalloc + a memset of the whole new capacity inside
// the call that happened to need more -- the p95 / max of the first call at a new ROI size: 1.6x / 4.0x the steady call.)
// zero: the caller reads the block before it writes it (tables with zero padding, accumulation buffers); the large blocks -- fields,
// level planes, image staging -- are written before they are read (or their unwritten parts only ever reach masked lanes) and skip it.
int ensure(Instance *I, DevBuf &b, size_t bytes, bool zero)
{
    if (bytes <= b.cap) {
        // (testing: a buffer that is RE-USED without zeroing holds what the previous call left -- here: NaN bytes, in place before any
        //  stream touches it; every stream of the instance has drained first, the previous call may still be reading)
        if (!zero && bytes && (I->opts.flags & SC_FLAG_POISON_ARENA)) {
            SC_HIP(I, hipStreamSynchronize(I->stream));
            if (I->aux) SC_HIP(I, hipStreamSynchronize(I->aux));
            if (I->aux2) SC_HIP(I, hipStreamSynchronize(I->aux2));
            SC_HIP(I, hipMemsetAsync(b.p, 0xFF, bytes, I->stream));
            SC_HIP(I, hipStreamSynchronize(I->stream));
        }
        return SC_OK;
    }
    size_t ncap = bytes > 2 * b.cap ? bytes : 2 * b.cap;
    ncap = (ncap + 4095) & ~(size_t)4095;
    void *np = nullptr;
    bool own = true;
    constexpr size_t SLAB_FIRST = (size_t)16 << 20, SLAB_PIECE_MAX = (size_t)8 << 20;
    if (ncap <= SLAB_PIECE_MAX) {          // a piece of a slab: no hipMalloc unless the slabs are used up
        if (I->slabs.empty() || I->slabs.back().cap - I->slabs.back().used < ncap) {
            Instance::Slab sl;
            sl.cap = I->slabs.empty() ? SLAB_FIRST : 2 * I->slabs.back().cap;
            SC_HIP(I, hipMalloc((void **)&sl.base, sl.cap));
            I->arena_bytes += sl.cap;
            I->slabs.push_back(sl);
        }
        Instance::Slab &sl = I->slabs.back();
        np = sl.base + sl.used;
        sl.used += ncap;                   // (ncap is a multiple of 4096: every piece is page aligned)
        own = false;
    } else {
        SC_HIP(I, hipMalloc(&np, ncap));
        I->arena_bytes += ncap;
    }
    if (zero) SC_HIP(I, hipMemsetAsync(np, 0, ncap, I->stream));
    else if (I->opts.flags & SC_FLAG_POISON_ARENA) {      // (testing: what recycled memory may hold; in place before ANY stream uses the block)
        SC_HIP(I, hipMemsetAsync(np, 0xFF, ncap, I->stream));
        SC_HIP(I, hipStreamSynchronize(I->stream));
    }
    if (b.p && b.own) {                              // (a replaced slab piece simply stays unused)
        I->retired.push_back(b);
        I->retired_bytes += b.cap;
        // ... unless the retired blocks have become large (an instance walking up through multi-gigabyte ROI sizes): then, and only
        // then, wait for the stream and give them back -- a growth step of that size is milliseconds of hipMalloc anyway
        if (I->retired_bytes > ((size_t)1 << 30)) {
            SC_HIP(I, hipStreamSynchronize(I->stream));
            if (I->aux) SC_HIP(I, hipStreamSynchronize(I->aux));
            if (I->aux2) SC_HIP(I, hipStreamSynchronize(I->aux2));
            for (DevBuf &r : I->retired) { I->arena_bytes -= r.cap; dev_release(r); }
            I->retired.clear();
            I->retired_bytes = 0;
        }
    }
    b.p = np;
    b.cap = ncap;
    b.own = own;
    return SC_OK;
}

// Page-locked host staging, grow-only.  At least 256 KB per buffer (round 5): the small ones -- eigenvalue tables, ratio tables, part
// maps, the stop rule's maxima -- used to start at a page and re-grow (stream wait + hipHostFree + hipHostMalloc: ~0.3 ms) whenever a
// caller's ROI size set a new record: the slowest first calls of the new_size leg (2.4-3.0x the steady call) were exactly those.
int ensure_pinned(Instance *I, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return SC_OK;
    size_t ncap = bytes > 2 * b.cap ? bytes : 2 * b.cap;
    ncap = std::max(ncap, (size_t)256 << 10);
    ncap = (ncap + 4095) & ~(size_t)4095;
    if (b.p) {
        SC_HIP(I, hipStreamSynchronize(I->stream));
        SC_HIP(I, hipHostFree(b.p));
        b.p = nullptr; b.cap = 0;
    }
    SC_HIP(I, hipHostMalloc(&b.p, ncap, hipHostMallocDefault));
    b.cap = ncap;
    if (I->opts.flags & SC_FLAG_POISON_ARENA) memset(b.p, 0x5A, ncap);      // (testing: what recycled host memory may hold -- the pad bytes of packed rows are never written; 0x5A: neither "inside the mask" nor "outside")
    return SC_OK;
}

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

// Row-wise host copy between a caller image and the pinned staging.  A single core moves ~14 GB/s, which
// would make the packing (not PCIe, not the GPU) the longest part of a 2048^2 call, so copies above 512 KB are
// shared between the calling thread and the instance's parked helpers (sc_hostcopy.h) in ~256 KB pieces.
// (1 MB until late in round 5: the splice of a 592^2 output, 1 037 232 bytes, ran on one core: 71 us of a 0.37-ms call, 25 shared;
//  256 KB loses: waking the helpers costs more than they save on the 0.4-MB pieces of a small call's upload)
static void copy_rows(Instance *I, uint8_t *dst, size_t dpitch, const uint8_t *src, size_t spitch, size_t row_bytes, int rows)
{
    const size_t total = row_bytes * (size_t)rows;
    auto span = [=](int y0, int y1) {
        if (dpitch == row_bytes && spitch == row_bytes) {
            memcpy(dst + (size_t)y0 * row_bytes, src + (size_t)y0 * row_bytes, row_bytes * (size_t)(y1 - y0));
            return;
        }
        for (int y = y0; y < y1; ++y) memcpy(dst + (size_t)y * dpitch, src + (size_t)y * spitch, row_bytes);
    };
    if (total < ((size_t)512 << 10)) { span(0, rows); return; }
    if (!I->copier) {
        const unsigned hw = std::thread::hardware_concurrency();
        int n = 8;                         // measured: packing saturates near 8 threads
        if (hw && (unsigned)n > hw) n = (int)hw;
        I->copier.reset(new RowCopier(n - 1));
    }
    const int rows_per = (int)std::max<size_t>(1, ((size_t)256 << 10) / std::max<size_t>(row_bytes, 1));
    const int parts = (rows + rows_per - 1) / rows_per;
    I->copier->parallel(parts, [&](int i) { span(i * rows_per, std::min(rows, (i + 1) * rows_per)); });
}

// rows x row_bytes from caller memory (pitch hpitch) to device memory (pitch dpitch).
// The library issues NO 2-D copies.  hipMemcpy2DAsync becomes one DMA per row (~6 us each, measured with
// rocprofv3: 384 copies per 298x192 clone), and under rocprofv3's copy interception the row DMAs of the SECOND of two
// back-to-back 2-D copies were released out of stream order (they landed after the kernels
// that read them had started, the last ones after teardown had freed the arena -> GPU memory-access fault).  So
// every strided host image -- pageable or caller-pinned -- is packed into the instance's pinned staging AT THE
// DEVICE PITCH and crosses PCIe as linear copies; only a caller-pinned image that already has the device pitch is
// copied in place.  The caller must not reuse `stage` before the stream has passed these copies.
static int upload_rows(Instance *I, DevBuf &stage, void *d, size_t dpitch, const uint8_t *h, size_t hpitch,
                       size_t row_bytes, int rows)
{
    if (rows <= 0 || row_bytes == 0) return SC_OK;
    if (hpitch == dpitch && is_pinned(h)) {
        SC_HIP(I, hipMemcpyAsync(d, h, dpitch * (size_t)(rows - 1) + row_bytes, hipMemcpyHostToDevice, I->stream));
        return SC_OK;
    }
    int rc = ensure_pinned(I, stage, dpitch * (size_t)rows);
    if (rc) return rc;
    uint8_t *s = (uint8_t *)stage.p;
    // Pieces: the DMA of piece k runs while piece k+1 is being packed.  The first piece is small (1 MB: the link starts moving
    // early), the following ones grow to 8 MB.  (Measured against equal 4 MB pieces on one box: no difference beyond noise --
    // the packing itself, 33-45 GB/s with eight threads, is what paces this path, not the DMA commands.)
    size_t piece = (size_t)1 << 20;
    for (int y0 = 0; y0 < rows;) {
        const int n = std::min((int)std::max<size_t>(1, piece / dpitch), rows - y0);
        copy_rows(I, s + (size_t)y0 * dpitch, dpitch, h + (size_t)y0 * hpitch, hpitch, row_bytes, n);
        SC_HIP(I, hipMemcpyAsync((uint8_t *)d + (size_t)y0 * dpitch, s + (size_t)y0 * dpitch, dpitch * (size_t)(n - 1) + row_bytes,
                                 hipMemcpyHostToDevice, I->stream));
        y0 += n;
        piece = std::min(piece * 4, (size_t)8 << 20);
    }
    return SC_OK;
}

// rows x row_bytes from device memory (pitch dpitch) into caller memory (pitch hpitch): ONE linear device-to-host copy
// into the pinned staging, a wait, then the rows are spliced on the host (no 2-D copy, see upload_rows).  Synchronous.
static int download_rows(Instance *I, DevBuf &stage, uint8_t *h, size_t hpitch, const void *d, size_t dpitch,
                         size_t row_bytes, int rows)
{
    if (rows <= 0 || row_bytes == 0) return SC_OK;
    const size_t total = dpitch * (size_t)(rows - 1) + row_bytes;
    int rc = ensure_pinned(I, stage, total);
    if (rc) return rc;
    SC_HIP(I, hipMemcpyAsync(stage.p, d, total, hipMemcpyDeviceToHost, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    copy_rows(I, h, hpitch, (const uint8_t *)stage.p, dpitch, row_bytes, rows);
    return SC_OK;
}

static Field make_field(void *p, int W, int H, int C)
{
    Field f;
    f.p = (float *)p; f.W = W; f.H = H; f.C = C;
    f.pitch = round_up(W, 64);
    f.plane = (size_t)f.pitch * H;
    return f;
}

int setup_fields(Instance *I, int W, int H, int C)
{
    field_moved(I);
    I->out_direct = false;             // new fields are about to be built (a clone's pre-process, sc_hip_build_rhs, sc_hip_field_load)
    Field proto = make_field(nullptr, W, H, C);
    const size_t bytes = proto.bytes() + 4096;
    int rc;
    if ((rc = ensure(I, I->d_U0, bytes, false))) return rc;
    if ((rc = ensure(I, I->d_U1, bytes, false))) return rc;
    if ((rc = ensure(I, I->d_F, bytes, false))) return rc;
    const bool same = I->F.p == I->d_F.p && I->U0.p == I->d_U0.p && I->U1.p == I->d_U1.p && I->F.W == W &&
                      I->F.H == H && I->F.C == C;
    I->U0 = make_field(I->d_U0.p, W, H, C);
    I->U1 = make_field(I->d_U1.p, W, H, C);
    I->F = make_field(I->d_F.p, W, H, C);
    I->result_in_U1 = false;
    I->f_half = false;        // whoever fills F next says what it holds
    I->u_half = false;
    if (!same) I->mg.clear(); // the multigrid hierarchy is rebuilt only when the ROI shape changes
    return SC_OK;
}

} // namespace sc

// A multigrid clone leaves its right-hand side as float16 inside F's buffer; the diagnostic hooks below read
// float.  Expand through the field that does not hold the result and copy back (not on any hot path).
static int float_rhs(Instance *I)
{
    if (!I->f_half) return SC_OK;
    SC_HIP(I, hipSetDevice(I->gpu));
    Field scratch = I->result_in_U1 ? I->U0 : I->U1;
    const size_t n = I->F.plane * (size_t)I->F.C;
    launch_half_to_float(I->F.p, scratch.p, n, I->stream);
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipMemcpyAsync(I->F.p, scratch.p, n * sizeof(float), hipMemcpyDeviceToDevice, I->stream));
    I->f_half = false;
    return SC_OK;
}

static Instance *get(void *p)
{
    Instance *I = (Instance *)p;
    if (!I || !I->ok()) return nullptr;
    return I;
}

static int validate_images(Instance *I, const void *face, int fc, int fr, int fs, const void *body, int bc, int br,
                           int bs, const void *mask, int mc, int mr, int ms)
{
    if (!face || !body || !mask) { I->err = "null image pointer"; return SC_ERR_BAD_ARG; }
    if (fc <= 0 || fr <= 0 || bc <= 0 || br <= 0 || mc <= 0 || mr <= 0) { I->err = "empty image"; return SC_ERR_BAD_SIZE; }
    if (fc != mc || fr != mr) { I->err = "face and mask sizes differ"; return SC_ERR_BAD_SIZE; }
    if (fs < 3 * fc || bs < 3 * bc || ms < mc) { I->err = "row step smaller than the row"; return SC_ERR_BAD_SIZE; }
    return SC_OK;
}

// bbox kernel + read-back of the rectangle into h_rect[4..7] (enqueue only).  mask is a device pointer.
// Stage mark k of the run's timeline (finish_timing).  An empty stage reuses the previous mark instead of recording an
// event: every hipEventRecord is a few microseconds of host time in front of the next launch.
static int tmark(Instance *I, int k, bool empty_stage = false)
{
    if (!I->stage_marks) { I->tm[k] = nullptr; return SC_OK; }    // asynchronous device call: nobody reads the timeline
    if (I->marks_ends_only && k != 0 && k != 7) empty_stage = true;   // SC_FLAG_NO_STAGE_MARKS: first and last mark only
    if (empty_stage && k > 0) { I->tm[k] = I->tm[k - 1]; return SC_OK; }
    I->tm[k] = I->ev[k];
    SC_HIP(I, hipEventRecord(I->ev[k], I->stream));
    return SC_OK;
}

// With `predicted` the same launch also erodes that ROI (the whole mask stage in one kernel: the erode of a predicted box
// does not depend on the box being computed); device_clone then skips its own erode launch.
static int bbox_enqueue(Instance *I, const uint8_t *d_mask, int mc, int mr, int ms, const Geo *predicted = nullptr)
{
    // The scan's workgroups leave their extrema as parts, the last one to finish folds them into the rectangle (seeded like the
    // reference's, seamlessClone_imp.cpp:1006) and stores it in device memory AND in the pinned mailbox h_rect[4..7]: no seed
    // upload and no read-back copy in the stream (each was a command of its own in front of / behind the launch).
    int rc = ensure(I, I->d_bbox_parts, sizeof(int) * 4 * (size_t)mask_bbox_blocks(mc, mr));
    if (rc) return rc;
    BboxFold fold;
    fold.parts = (int *)I->d_bbox_parts.p;
    fold.counter = (unsigned *)(I->d_rect + 32);      // a line of its own; zero between launches (the folding workgroup resets it)
    fold.rect_dev = I->d_rect;
    fold.rect_host = I->h_rect + 4;
    I->erode_done = false;
    I->scan_pending = false;
    I->scan_fence = nullptr;
    if (I->scan_counter_dirty) {      // after a HIP error: the folding workgroup is the only one that resets the counter, and it may never have run
        SC_HIP(I, hipMemsetAsync(fold.counter, 0, sizeof(unsigned), I->stream));
        I->scan_counter_dirty = false;
    }
    if (predicted && !(I->opts.flags & SC_FLAG_OPENCV_GREY_MASK)) {
        // A clone launched on a predicted box needs the scan's answer only at its end: the erode of the predicted ROI goes out
        // alone and the scan rides in the pre-process launch behind it (device_clone, launch_preprocess) -- off the critical path.
        I->mpitch = round_up(predicted->W, 64);
        rc = ensure(I, I->d_M, (size_t)I->mpitch * predicted->H, false);
        if (rc) return rc;
        // (round 4, late: no erode launch either -- the pre-process tiles form the eroded mask themselves and leave it in d_M)
        I->erode_done = true;
        I->pending_scan = BboxTask();
        I->pending_scan.mask = d_mask; I->pending_scan.mw = mc; I->pending_scan.mh = mr; I->pending_scan.mstep = ms;
        I->pending_scan.fold = fold;
        I->pending_scan.g = *predicted;
        I->pending_scan.mask_bytes = (size_t)ms * (mr - 1) + (size_t)(predicted->x0 + predicted->W + 1);      // as launch_mask_erode3 counts them
        I->scan_pending = true;
    } else
    launch_mask_bbox(d_mask, mc, mr, ms, fold, I->stream);
    SC_HIP(I, hipGetLastError());
    return tmark(I, 2);
}

static int geo_from_rect(Instance *I, const int r[4], int cx, int cy, Geo &g)
{
    const int x0 = r[0], x1 = r[1], y0 = r[2], y1 = r[3];
    if (!((x1 - x0) > 0 && (y1 - y0) > 0)) { I->err = "mask has no usable non-zero region"; return SC_ERR_EMPTY_MASK; }
    g.x0 = x0; g.y0 = y0; g.W = x1 - x0 + 1; g.H = y1 - y0 + 1;
    g.ltx = cx - (g.W >> 1); // seamlessClone_imp.cpp:1066
    g.lty = cy - (g.H >> 1);
    return SC_OK;
}

// synchronous form: waits for the device's rectangle (the reference does the same, seamlessClone_imp.cpp:1012)
static int device_bbox(Instance *I, const uint8_t *d_mask, int mc, int mr, int ms, int cx, int cy, Geo &g)
{
    int rc = bbox_enqueue(I, d_mask, mc, mr, ms);
    if (rc) return rc;
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return geo_from_rect(I, I->h_rect + 4, cx, cy, g);
}

// Predicted bounding box for a clone launched before the device's answer is back: the previous rectangle when the
// mask has the size of the previous call's (a sequence of clones with one mask), else the whole interior (every
// mask that touches its four inner borders, all-255 masks in particular).  After a wrong guess speculation pauses
// for a few calls, so a stream of unpredictable masks pays at most one wasted clone in nine.
static bool predict_rect(Instance *I, int mc, int mr, int r[4])
{
    if ((I->opts.flags & SC_FLAG_NO_SPECULATE) || mc < 3 || mr < 3) return false;
    if (I->spec_cooldown > 0) { --I->spec_cooldown; return false; }
    if (I->last_mc == mc && I->last_mr == mr) { memcpy(r, I->last_rect, sizeof(int) * 4); return true; }
    r[0] = 1; r[1] = mc - 2; r[2] = 1; r[3] = mr - 2;
    return true;
}

static void remember_rect(Instance *I, int mc, int mr, const int r[4])
{
    I->last_mc = mc; I->last_mr = mr;
    memcpy(I->last_rect, r, sizeof(int) * 4);
}

static RectGuard make_guard(Instance *I, const int r[4])
{
    RectGuard g;
    g.d_rect = I->d_rect; g.x0 = r[0]; g.x1 = r[1]; g.y0 = r[2]; g.y1 = r[3];
    return g;
}

static int check_roi(Instance *I, const Geo &g, int bc, int br)
{
    if (g.ltx < 0 || g.lty < 0 || g.ltx + g.W > bc || g.lty + g.H > br) {
        I->err = "ROI leaves the destination image";
        return SC_ERR_ROI_OOB;
    }
    return SC_OK;
}

// erode -> pre-process -> solve -> post-process on device-resident ROI origins
// out_org / ostep: where the output bytes go (ROI origin, row step): the destination itself for device-resident images; the
// host path hands a compact buffer of its own so that what comes back across PCIe is the ROI and nothing else
static int device_clone(Instance *I, const uint8_t *d_mask, int ms, int mr, const uint8_t *face_org, int fstep,
                        uint8_t *body_org, int bstep, const Geo &g, int passes, uint8_t *out_org = nullptr, int ostep = 0,
                        bool fence_at_end = true)
{
    // fence_at_end: the event behind which the scan's rectangle may be read (scan_fence) is the clone's LAST mark, not one of its
    // own behind the pre-process launch -- every event in the stream is a ~5 us bubble, and a caller that waits for the whole clone
    // anyway (a device-resident synchronous call, a host call whose output needs no copy command) loses nothing by it.  false: the
    // host call that still has device-to-host copies to enqueue behind the clone checks the rectangle while the clone's tail runs.
    if (!out_org) { out_org = body_org; ostep = bstep; }
    bool fence_pending = false;
    int rc;
    I->mpitch = round_up(g.W, 64);
    if ((rc = ensure(I, I->d_M, (size_t)I->mpitch * g.H, false))) return rc;
    if ((rc = setup_fields(I, g.W, g.H, 3))) return rc;
    const bool eroded = I->erode_done;
    const bool grey = (I->opts.flags & SC_FLAG_OPENCV_GREY_MASK) != 0;
    if (!eroded && grey) launch_mask_erode_min7(d_mask, ms, g, (uint8_t *)I->d_M.p, I->mpitch, I->stream);
    else if (!eroded) launch_mask_erode3(d_mask, ms, mr, g, (uint8_t *)I->d_M.p, I->mpitch, I->stream);
    I->erode_done = false;
    if ((rc = tmark(I, 4, eroded))) return rc;
    int solve_rc = SC_OK;
    for (int pass = 0; pass < passes; ++pass) {
        if (solve_rc == SC_RETRY_FLOAT_FIELD) {      // the 16-bit field of the pass before saturated: nothing was written, the
            I->force_float_field = true;             // same pass again on float fields (sc_cycle0.hip, c0_q16_checked)
            I->info.field_retry = 1;
        }
        I->result_in_U1 = false;
        I->f_half = mg_reads_half_rhs(I);
        I->u_half = I->f_half && !(I->opts.flags & SC_FLAG_FLOAT_U0);
        if (I->scan_pending) I->pending_scan.M_out = (uint8_t *)I->d_M.p;      // the launch's tiles erode the mask themselves and leave it here
        const bool had_scan = I->scan_pending;
        launch_preprocess(body_org, bstep, face_org, fstep, (const uint8_t *)I->d_M.p, I->mpitch, I->U0, I->U1, I->F,
                          I->stream, I->f_half, I->u_half, grey, I->scan_pending ? &I->pending_scan : nullptr);
        I->scan_pending = false;
        if (pass == passes - 1 && (rc = tmark(I, 5))) return rc;
        if (had_scan) {
            // the host compares the scan's rectangle (pinned mailbox) with its guess once THIS point of the stream has passed: mark 5
            // when it was really recorded just now, else the clone's last mark (fence_at_end), else an event of its own
            if (pass == passes - 1 && I->stage_marks && I->tm[5] == I->ev[5]) I->scan_fence = I->ev[5];
            else if (fence_at_end && I->stage_marks) fence_pending = true;
            else { SC_HIP(I, hipEventRecord(I->ev_scan, I->stream)); I->scan_fence = I->ev_scan; }
        }
        I->info.sweep_launches = 0;
        I->spec_post.body_org = out_org; I->spec_post.bstep = ostep;
        I->spec_post.ev_solved = nullptr;
        I->spec_post.armed = true; I->spec_post.done = false;
        solve_rc = solve(I);
        I->spec_post.armed = false;
        I->force_float_field = false;
        if (solve_rc == SC_RETRY_FLOAT_FIELD) { --pass; continue; }
        if (solve_rc != SC_OK && solve_rc != SC_ERR_NOT_CONVERGED) return solve_rc;
        if (!I->spec_post.done) {          // otherwise the solver already enqueued it behind its last cycle
            if (pass == passes - 1 && (rc = tmark(I, 6))) return rc;
            LmNodes lm;
            if ((rc = output_nodes(I, lm))) return rc;
            launch_postprocess(result(I), out_org, ostep, I->stream, I->guard, lm);      // (a solve that got here stored no 16-bit field or kept it in range)
        } else if (pass == passes - 1) {
            I->tm[6] = nullptr;            // no mark between the last cycle and the post-process (an event there costs a
        }                                  // ~5 us bubble): ms_post is reported as 0 and ms_solve includes it
        SC_HIP(I, hipGetLastError());
    }
    if ((rc = tmark(I, 7))) return rc;
    if (!I->tm[6]) I->tm[6] = I->tm[7];
    if (fence_pending) {
        if (I->tm[7]) I->scan_fence = I->tm[7];
        else { SC_HIP(I, hipEventRecord(I->ev_scan, I->stream)); I->scan_fence = I->ev_scan; }
    }
    return solve_rc;
}

static void fill_info_geo(Instance *I, const Geo &g)
{
    I->info.x0 = g.x0; I->info.y0 = g.y0; I->info.W = g.W; I->info.H = g.H; I->info.ltx = g.ltx; I->info.lty = g.lty;
    I->info.device_bytes = I->arena_bytes;
    I->info.device = I->gpu;
}

static float ev_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

extern "C" {

void sc_hip_default_opts(sc_solver_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->method = SC_METHOD_AUTO;
    o->max_sweeps = 30;      // V-cycles
    o->tol = 0.f;            // MULTIGRID stops on update_tol; the sweep methods on tol
    o->check_every = 1;
    o->omega = 0.f;
    o->sweeps_per_launch = 0;
    o->reference_warmup = 0;
    o->mg_pre = 2;
    o->mg_post = 2;
    o->update_tol = 0.25f;   // residual error ~0.01 grey levels (measured: tools/mg_convergence.py)
}

int sc_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sc_hip_device_pci_bus_id(int gpu_id, char *buf, int len)
{
    if (!buf || len < 16) return SC_ERR_BAD_ARG;
    buf[0] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || gpu_id < 0 || gpu_id >= n) { (void)hipGetLastError(); return SC_ERR_BAD_ARG; }
    if (hipDeviceGetPCIBusId(buf, len, gpu_id) != hipSuccess) { (void)hipGetLastError(); buf[0] = 0; return SC_ERR_HIP; }
    return SC_OK;
}

void *my_seamlessclone_api_imp_create_instance(int gpu_id)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || gpu_id < 0 || gpu_id >= n) {
        fprintf(stderr, "seamlessclone_hip: cannot use GPU %d (%d visible)\n", gpu_id, n);
        return nullptr;
    }
    if (hipSetDevice(gpu_id) != hipSuccess) return nullptr;
    Instance *I = new (std::nothrow) Instance();
    if (!I) return nullptr;
    I->gpu = gpu_id;
    sc_hip_default_opts(&I->opts);
    bool ok = hipStreamCreateWithFlags(&I->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&I->aux, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&I->ev_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&I->ev_join, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&I->ev_fd_fork, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&I->ev_fd, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&I->ev_scan, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&I->h_rect, 8 * sizeof(int), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&I->h_red, 2 * sizeof(double), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void **)&I->d_rect, 64 * sizeof(int)) == hipSuccess;       // the rectangle; word 32: the scan's arrival counter
    ok = ok && hipMemset(I->d_rect, 0, 64 * sizeof(int)) == hipSuccess;
    ok = ok && hipMalloc((void **)&I->d_partials, 2 * sizeof(double) * residual_max_blocks()) == hipSuccess;
    ok = ok && hipMalloc((void **)&I->d_red, 2 * sizeof(double)) == hipSuccess;
    ok = ok && hipMalloc((void **)&I->d_maxcorr, 4 * sizeof(unsigned)) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&I->h_maxcorr, 4 * sizeof(unsigned), hipHostMallocDefault) == hipSuccess;
    ok = ok && mg_bottom_prepare() == hipSuccess;   // opt in to >64 KiB dynamic LDS for the bottom kernel
    for (int i = 0; ok && i < 8; ++i) ok = hipEventCreate(&I->ev[i]) == hipSuccess;
    ok = ok && hipEventCreate(&I->ev_k0) == hipSuccess && hipEventCreate(&I->ev_k1) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i) ok = hipEventCreate(&I->ev_chunk[i]) == hipSuccess;
    if (!ok) {
        fprintf(stderr, "seamlessclone_hip: instance creation failed on GPU %d: %s\n", gpu_id,
                hipGetErrorString(hipGetLastError()));
        my_seamlessclone_api_imp_destroy(I);
        return nullptr;
    }
    return I;
}

void my_seamlessclone_api_imp_destroy(void *p)
{
    Instance *I = get(p);
    if (!I) return;
    (void)hipSetDevice(I->gpu);
    if (I->stream) (void)hipStreamSynchronize(I->stream);
    if (I->aux) (void)hipStreamSynchronize(I->aux);
    if (I->aux2) (void)hipStreamSynchronize(I->aux2);
    for (DevBuf &b : I->retired) dev_release(b);                    // blocks that growth replaced (ensure)
    for (Instance::Slab &sl : I->slabs) if (sl.base) (void)hipFree(sl.base);
    DevBuf *bufs[] = { &I->d_face, &I->d_body_roi, &I->d_out, &I->d_mask, &I->d_in, &I->d_M, &I->d_U0, &I->d_U1, &I->d_F };
    for (DevBuf *b : bufs) dev_release(*b);
    for (DevBuf &b : I->mg_bufs) dev_release(b);
    dev_release(I->mg_partial);
    if (I->h_partial.p) (void)hipHostFree(I->h_partial.p);
    for (DevBuf *b : { &I->lm.P, &I->lm.E, &I->lm.CN, &I->lm.B, &I->lm.maps[0].d, &I->lm.maps[1].d }) dev_release(*b);
    for (LowMode::Tables &t : I->lm.tables) {          // (lm.Sx / Sy / R are views of one of these)
        for (DevBuf *b : { &t.Sx, &t.Sy, &t.R }) dev_release(*b);
        if (t.hR.p) (void)hipHostFree(t.hR.p);
        if (t.ev) (void)hipEventDestroy(t.ev);
    }
    for (auto &m : I->lm.maps) if (m.h.p) (void)hipHostFree(m.h.p);
    for (DevBuf *b : { &I->dst.Sw, &I->dst.Sh, &I->dst.fxy, &I->dst.G, &I->dst.T1, &I->dst.T2 }) dev_release(*b);
    if (I->dst.hfxy.p) (void)hipHostFree(I->dst.hfxy.p);
    for (DevBuf *b : { &I->fft.A, &I->fft.B, &I->fft.tw64 }) dev_release(*b);
    for (FftDim &d : I->fft.dims) dev_release(d.chirp);
    if (I->fft.hst_all.p) (void)hipHostFree(I->fft.hst_all.p);
    for (FftFxy &f : I->fft.fxy) {
        dev_release(f.d);
        if (f.hst.p) (void)hipHostFree(f.hst.p);
        if (f.ev) (void)hipEventDestroy(f.ev);
    }
    if (I->fft.ev_fork) (void)hipEventDestroy(I->fft.ev_fork);
    if (I->fft.ev_built) (void)hipEventDestroy(I->fft.ev_built);
    dev_release(I->mg_fd);
    dev_release(I->rag.d_aux);
    if (I->rag.h_stage.p) (void)hipHostFree(I->rag.h_stage.p);
    if (I->rag.ev) (void)hipEventDestroy(I->rag.ev);
    if (I->rag.ev_ready) (void)hipEventDestroy(I->rag.ev_ready);
    if (I->ev_fd_fork) (void)hipEventDestroy(I->ev_fd_fork);
    if (I->ev_fd) (void)hipEventDestroy(I->ev_fd);
    if (I->ev_scan) (void)hipEventDestroy(I->ev_scan);
    if (I->d_rect) (void)hipFree(I->d_rect);
    dev_release(I->d_rects);
    dev_release(I->d_bbox_parts);
    if (I->h_rects.p) (void)hipHostFree(I->h_rects.p);
    if (I->d_partials) (void)hipFree(I->d_partials);
    if (I->d_red) (void)hipFree(I->d_red);
    if (I->d_maxcorr) (void)hipFree(I->d_maxcorr);
    if (I->h_maxcorr) (void)hipHostFree(I->h_maxcorr);
    if (I->h_rect) (void)hipHostFree(I->h_rect);
    if (I->h_red) (void)hipHostFree(I->h_red);
    for (DevBuf *b : { &I->h_face, &I->h_body, &I->h_mask, &I->h_out, &I->h_in }) if (b->p) (void)hipHostFree(b->p);
    for (int i = 0; i < 8; ++i) if (I->ev[i]) (void)hipEventDestroy(I->ev[i]);
    for (int i = 0; i < 8; ++i) if (I->ev_chunk[i]) (void)hipEventDestroy(I->ev_chunk[i]);
    if (I->ev_k0) (void)hipEventDestroy(I->ev_k0);
    if (I->ev_k1) (void)hipEventDestroy(I->ev_k1);
    if (I->ev_rects) (void)hipEventDestroy(I->ev_rects);
    if (I->ev_fork) (void)hipEventDestroy(I->ev_fork);
    if (I->ev_join) (void)hipEventDestroy(I->ev_join);
    if (I->aux) (void)hipStreamDestroy(I->aux);
    if (I->aux2) (void)hipStreamDestroy(I->aux2);
    if (I->stream) (void)hipStreamDestroy(I->stream);
    I->magic = 0;
    delete I;
}

void my_seamlessclone_api_imp_sync(void *p)
{
    Instance *I = get(p);
    if (!I) return;
    (void)hipSetDevice(I->gpu);
    (void)hipStreamSynchronize(I->stream);
}

int sc_hip_set_solver(void *p, const sc_solver_opts *o)
{
    Instance *I = get(p);
    if (!I || !o) return SC_ERR_BAD_ARG;
    if (o->method < SC_METHOD_JACOBI || o->method > SC_METHOD_FFT || o->max_sweeps < 0) {
        I->err = "bad solver options";
        return SC_ERR_BAD_ARG;
    }
    if (o->jacobi_tile_rows != 0 && o->jacobi_tile_rows != 16 && o->jacobi_tile_rows != 32 && o->jacobi_tile_rows != 64) {
        I->err = "jacobi_tile_rows must be 0, 16, 32 or 64";
        return SC_ERR_BAD_ARG;
    }
    if (o->mg_level1_sweeps != 0 && (o->mg_level1_sweeps < 2 || o->mg_level1_sweeps > 4)) {
        I->err = "mg_level1_sweeps must be 0 or 2..4";
        return SC_ERR_BAD_ARG;
    }
    if (((o->flags ^ I->opts.flags) & SC_FLAG_VCYCLE_BOTTOM) || legacy_path(*o, SC_LEGACY_BOTTOM_F32) != legacy_path(I->opts, SC_LEGACY_BOTTOM_F32) || o->mg_direct_max != I->opts.mg_direct_max) I->mg.clear();   // the hierarchy (which bottom level is solved directly) depends on these only
    if (o->mg_direct_max < 0) { I->err = "mg_direct_max must be >= 0"; return SC_ERR_BAD_ARG; }
    I->opts = *o;
    return SC_OK;
}

int sc_hip_get_solver(void *p, sc_solver_opts *o)
{
    Instance *I = get(p);
    if (!I || !o) return SC_ERR_BAD_ARG;
    *o = I->opts;
    return SC_OK;
}

int sc_hip_get_info(void *p, sc_run_info *info)
{
    Instance *I = get(p);
    if (!I || !info) return SC_ERR_BAD_ARG;
    I->info.device_bytes = I->arena_bytes;
    I->info.device = I->gpu;
    *info = I->info;
    return SC_OK;
}

const char *sc_hip_last_error(void *p)
{
    Instance *I = get(p);
    if (!I) return "bad instance";
    return I->err.c_str();
}

void *sc_hip_malloc(void *p, size_t bytes)
{
    Instance *I = get(p);
    if (!I) return nullptr;
    (void)hipSetDevice(I->gpu);
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    return d;
}

void sc_hip_free(void *p, void *d)
{
    Instance *I = get(p);
    if (!I || !d) return;
    (void)hipSetDevice(I->gpu);
    (void)hipStreamSynchronize(I->stream);
    (void)hipFree(d);
}

void *sc_hip_host_alloc(void *p, size_t bytes)
{
    Instance *I = get(p);
    if (!I || bytes == 0) return nullptr;
    (void)hipSetDevice(I->gpu);
    void *h = nullptr;
    if (hipHostMalloc(&h, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return h;
}

void sc_hip_host_free(void *p, void *h)
{
    Instance *I = get(p);
    if (!I || !h) return;
    (void)hipSetDevice(I->gpu);
    (void)hipStreamSynchronize(I->stream);
    (void)hipHostFree(h);
}

int sc_hip_memcpy_h2d(void *p, void *d, const void *h, size_t bytes)
{
    Instance *I = get(p);
    if (!I || !d || !h) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    SC_HIP(I, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_memcpy_d2h(void *p, void *h, const void *d, size_t bytes)
{
    Instance *I = get(p);
    if (!I || !d || !h) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    SC_HIP(I, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_memcpy_d2d_async(void *p, void *dst, const void *src, size_t bytes)
{
    Instance *I = get(p);
    if (!I || !dst || !src) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    SC_HIP(I, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, I->stream));
    return SC_OK;
}

static void finish_timing(Instance *I, bool staged)
{
    // ev: 0 start | 1 mask on device | 2 bbox done | 3 ROI images on device | 4 erode done |
    //     5 pre-process done | 6 solve done | 7 post-process done ; ev_k1 = D2H done
    I->info.ms_h2d = staged ? ev_ms(I->tm[0], I->tm[1]) + ev_ms(I->tm[2], I->tm[3]) : 0.f;
    I->info.ms_mask = ev_ms(I->tm[1], I->tm[2]) + ev_ms(I->tm[3], I->tm[4]);
    I->info.ms_pre = ev_ms(I->tm[4], I->tm[5]);
    I->info.ms_solve = ev_ms(I->tm[5], I->tm[6]);
    I->info.ms_post = ev_ms(I->tm[6], I->tm[7]);
    I->info.ms_d2h = staged ? ev_ms(I->tm[7], I->ev_k1) : 0.f;
    I->info.ms_device_total = I->info.ms_mask + I->info.ms_pre + I->info.ms_solve + I->info.ms_post;
}

int my_seamlessclone_api_imp_run(void *p, const uint8_t *face, int fc, int fr, int fs, uint8_t *body, int bc, int br,
                                 int bs, const uint8_t *mask, int mc, int mr, int ms, int cx, int cy, int gpu_id,
                                 bool bSync)
{
    (void)gpu_id; // the instance already owns its device (the reference ignores it as well, seamlessClone_imp.cu:265)
    Instance *I = get(p);
    if (!I) return SC_ERR_BAD_ARG;
    I->err.clear();
    I->info.field_retry = 0; I->info.new_size = 0; I->info.group_members = 0; I->info.group_ragged = 0;
    SC_HIP(I, hipSetDevice(I->gpu));
    int rc = validate_images(I, face, fc, fr, fs, body, bc, br, bs, mask, mc, mr, ms);
    if (rc) return rc;
    // --- the predicted box (the previous one for this mask size, else the mask's interior), if any
    int guess[4];
    Geo gp{};
    bool predicted = predict_rect(I, mc, mr, guess);
    if (predicted) {
        predicted = geo_from_rect(I, guess, cx, cy, gp) == SC_OK && check_roi(I, gp, bc, br) == SC_OK;
        I->err.clear();                               // a guess that does not fit the destination is not an error
    }
    // --- SMALL calls (the reference's own patches: 154 x 100 ... 592^2 into 1600 x 898): what the clone reads -- mask, patch ROI,
    //     destination ROI -- is packed row by row into ONE pinned block and crosses PCIe as ONE copy into one device block.  Three
    //     copies out of pageable memory cost 12-40 us EACH before a byte moves (the runtime stages them itself): h2d 0.037 / 0.050 /
    //     0.131 ms at 154 x 100 / 300 x 194 / 592^2 for 0.1 / 0.4 / 2.4 MB.  Needs the box before the first copy: predicted calls only.
    struct PreIn { const uint8_t *face = nullptr; uint8_t *body = nullptr; int pitch = 0; } pre;
    constexpr size_t SMALL_CALL_MAX = (size_t)1 << 20;      // (measured: 0.138 -> 0.117 ms per call at 154 x 100, 0.177 -> 0.166 at 300 x 194; at 592^2 -- 2.4 MB -- packing on the host loses: 0.415 -> 0.425)
    const uint8_t *dmask = nullptr;                   // the mask on the device, its row step
    int dms = 0;
    I->stage_marks = true;         // a host-image call is synchronous whatever bSync says: its timeline is always read
    I->marks_ends_only = (I->opts.flags & SC_FLAG_NO_STAGE_MARKS) != 0;      // (... unless the caller gives the per-stage figures up for their ~5 us bubbles)
    if (predicted && !I->opts.reference_warmup) {
        const int sdms = round_up(mc, 256), sdfs = round_up(3 * gp.W, 256);
        const size_t bm = (size_t)sdms * mr, bi = (size_t)sdfs * gp.H, total = bm + 2 * bi;
        if (total <= SMALL_CALL_MAX) {
            if ((rc = ensure(I, I->d_in, total + 64, false))) return rc;
            if ((rc = ensure_pinned(I, I->h_in, total))) return rc;
            if ((rc = tmark(I, 0))) return rc;
            uint8_t *const hs = (uint8_t *)I->h_in.p, *const ds = (uint8_t *)I->d_in.p;
            copy_rows(I, hs, (size_t)sdms, mask, (size_t)ms, (size_t)mc, mr);
            copy_rows(I, hs + bm, (size_t)sdfs, face + (size_t)gp.y0 * fs + 3 * (size_t)gp.x0, (size_t)fs, 3 * (size_t)gp.W, gp.H);
            copy_rows(I, hs + bm + bi, (size_t)sdfs, body + (size_t)gp.lty * bs + 3 * (size_t)gp.ltx, (size_t)bs, 3 * (size_t)gp.W, gp.H);
            {   // ... by a KERNEL that reads the pinned block across PCIe (k_copy_group: sixteen bytes per lane, four loads in flight): a copy
                // command takes ~20 us before its first byte moves, whatever its size (h2d stage at 154 x 100: 24 us for 0.1 MB)
                CopyJobs cj{};
                cj.dst[0] = ds; cj.src[0] = hs; cj.bytes[0] = total;
                launch_copy_group(cj, 1, I->stream);
                SC_HIP(I, hipGetLastError());
            }
            dmask = ds; dms = sdms;
            pre.face = ds + bm; pre.body = ds + bm + bi; pre.pitch = sdfs;
            if ((rc = tmark(I, 1))) return rc;
        }
    }
    if (!dmask) {
        // --- mask to the device
        const bool whole_m = 4 * (size_t)mc >= 3 * (size_t)ms;      // (not a narrow view of a much wider image): one linear copy at the caller's step
        dms = whole_m ? ms : round_up(mc, 256);
        if ((rc = ensure(I, I->d_mask, (size_t)dms * mr + 64, false))) return rc;
        if ((rc = tmark(I, 0))) return rc;
        if (whole_m) SC_HIP(I, hipMemcpyAsync(I->d_mask.p, mask, (size_t)ms * (mr - 1) + mc, hipMemcpyHostToDevice, I->stream));
        else if ((rc = upload_rows(I, I->h_mask, I->d_mask.p, dms, mask, ms, mc, mr))) return rc;
        dmask = (const uint8_t *)I->d_mask.p;
        if ((rc = tmark(I, 1))) return rc;
    }
    // One attempt on a given geometry: ROI of face/body to the device (the reference uploads both images whole),
    // clone, [check the predicted box], result back into the caller's image.  SC_GUESS_WRONG = the device found a
    // different box than `guess`; nothing has been written anywhere the caller can see.
    constexpr int SC_GUESS_WRONG = 1;
    auto attempt = [&](const Geo &g, const int *guess, const PreIn *in) -> int {      // in: the ROIs are on the device already (the small call's one copy)
        int r;
        const int dfs = round_up(3 * g.W, 256);
        // An image whose ROI covers most of its rows (>= 3/4 of the row step: the patch always, the destination often) crosses
        // PCIe as ONE linear copy of whole rows straight from the caller's memory at the caller's row step -- no packing pass,
        // no helper threads; the kernels take any origin and step.  Packing 29 MB with eight threads ran at 33-45 GB/s on the
        // boxes measured, under the link's 55; the direct copy: h2d 0.90 -> 0.67 ms of a 2048^2 call on the slower box.  A small
        // ROI of a large image is still packed (row by row into pinned staging, one DMA per piece).  The library still issues
        // no 2-D copies (appendix B).
        // ... or the rows' bytes outside the ROI are few in all (a small ROI of a larger image -- the reference's published table:
        // patches into a 1600 x 898 destination): the packed path has ~0.1 ms of fixed cost (helper threads, staging, the splice
        // behind the last DMA), 3 MB more on the link cost 0.055 (600^2 ROI of a 1600-wide image: 0.475 -> 0.36 ms per call)
        // -- but not more than twice the ROI's own bytes: a 300 x 194 patch of a 1600-wide image (0.76 MB around 0.17) is faster packed
        // (0.20 against 0.25 ms)
        constexpr size_t WHOLE_EXTRA_MAX = (size_t)3 << 20;
        auto few_extra = [&](int step) { const size_t extra = ((size_t)step - 3 * (size_t)g.W) * g.H; return extra <= WHOLE_EXTRA_MAX && extra <= 2 * 3 * (size_t)g.W * g.H; };
        const bool whole_f = !in && (4 * 3 * (size_t)g.W >= 3 * (size_t)fs || few_extra(fs));
        const bool whole_b = !in && (4 * 3 * (size_t)g.W >= 3 * (size_t)bs || few_extra(bs));
        const int fpitch = in ? in->pitch : whole_f ? fs : dfs, bpitch = in ? in->pitch : whole_b ? bs : dfs;
        const size_t foff = whole_f ? 3 * (size_t)g.x0 : 0, boff = whole_b ? 3 * (size_t)g.ltx : 0;
        if ((r = ensure(I, I->d_out, (size_t)dfs * g.H + 64, false))) return r;
        if (!in) {
            if ((r = ensure(I, I->d_face, (size_t)fpitch * g.H + 64, false))) return r;
            if ((r = ensure(I, I->d_body_roi, (size_t)bpitch * g.H + 64, false))) return r;
            if (whole_f) SC_HIP(I, hipMemcpyAsync(I->d_face.p, face + (size_t)g.y0 * fs, (size_t)fs * (g.H - 1) + foff + 3 * (size_t)g.W, hipMemcpyHostToDevice, I->stream));
            else if ((r = upload_rows(I, I->h_face, I->d_face.p, dfs, face + (size_t)g.y0 * fs + 3 * g.x0, fs, 3 * (size_t)g.W, g.H))) return r;
            if (whole_b) SC_HIP(I, hipMemcpyAsync(I->d_body_roi.p, body + (size_t)g.lty * bs, (size_t)bs * (g.H - 1) + boff + 3 * (size_t)g.W, hipMemcpyHostToDevice, I->stream));
            else if ((r = upload_rows(I, I->h_body, I->d_body_roi.p, dfs, body + (size_t)g.lty * bs + 3 * g.ltx, bs, 3 * (size_t)g.W, g.H))) return r;
        }
        const uint8_t *const d_face_roi = in ? in->face : (const uint8_t *)I->d_face.p + foff;
        uint8_t *const d_body_roi = in ? in->body : (uint8_t *)I->d_body_roi.p + boff;
        if ((r = tmark(I, 3))) return r;
        const int passes = I->opts.reference_warmup ? 2 : 1;
        // the output bytes go to a compact buffer of their own (the interior only is written, and only that comes back); the
        // reference's warm-up pass (two applications in place) needs the first result where the second reads it: the body buffer
        // SC_FLAG_ROWS_RETURN (opt-in since round 5): a destination uploaded as whole rows takes the output bytes in place as well and the
        // rows come back as ONE linear copy straight into the caller's image -- no pinned staging, no splice on the host behind the
        // last DMA.  What it overwrites outside the ROI's columns are the caller's own bytes, uploaded a moment ago: harmless only
        // while nobody else writes those pixels during the call, which the library cannot know (two calls cloning into disjoint ROIs
        // of one image would lose each other's result) -- so the default writes ROI bytes only, as the reference does
        // (seamlessClone_imp.cpp:470-483).
        const bool inplace = whole_b && passes == 1 && (size_t)bs == 3 * (size_t)bc && (I->opts.flags & SC_FLAG_ROWS_RETURN);      // (a view into a wider array keeps the staged path: nothing beyond the view's own pixels is ever written)
        constexpr size_t DIRECT_OUT_MAX = (size_t)2 << 20;
        // a SMALL output: the output launch writes the ROI's bytes straight into the pinned staging (three words per lane, coalesced: posted
        // writes across PCIe) -- no device-to-host copy command (~10 us before its first byte moves) behind the last kernel
        const bool direct_out = passes == 1 && !inplace && (size_t)dfs * g.H <= DIRECT_OUT_MAX;
        if (direct_out && (r = ensure_pinned(I, I->h_out, (size_t)dfs * g.H + 64))) return r;
        uint8_t *const out_dev = (passes > 1 || inplace) ? d_body_roi : direct_out ? (uint8_t *)I->h_out.p : (uint8_t *)I->d_out.p;
        const int out_pitch = (passes > 1 || inplace) ? bpitch : dfs;
        r = device_clone(I, dmask, dms, mr, d_face_roi, fpitch, d_body_roi, bpitch, g, passes, out_dev, out_pitch, direct_out);
        if (r != SC_OK && r != SC_ERR_NOT_CONVERGED) return r;
        // direct_out: nothing is enqueued behind the clone -- ONE wait for the stream, then the rectangle is compared, then the rows are
        // spliced (a wrong guess: the guarded output launch wrote nothing, the staging holds nothing the caller will see)
        const bool check_after_sync = guess && direct_out;
        if (guess && !check_after_sync) {      // the scan rode in the pre-process launch and finished long ago: this wait on the event behind that launch is free
            if (I->scan_fence) SC_HIP(I, hipEventSynchronize(I->scan_fence));
            else SC_HIP(I, hipStreamSynchronize(I->stream));      // (cannot happen: a clone launched on a guess always carries its scan)
            I->scan_fence = nullptr;
            if (memcmp(guess, I->h_rect + 4, 4 * sizeof(int)) != 0) return SC_GUESS_WRONG;
        }
        // Interior back into the caller's image: linear D2H pieces of the compact ROI buffer into pinned staging
        // (a 2-D copy would be one DMA per row), each spliced into the image row by row while the next one is
        // still crossing PCIe.  The call completes before returning whatever bSync says: the result has to be in
        // caller memory (the reference is effectively synchronous too: its D2H + host splice, imp.cpp:471).
        const int orows = g.H - 2;
        const size_t ob = 3 * (size_t)(g.W - 2);
        if (orows > 0 && g.W > 2 && inplace) {
            SC_HIP(I, hipMemcpyAsync(body + (size_t)(g.lty + 1) * bs + 3 * (size_t)(g.ltx + 1), out_dev + (size_t)out_pitch + 3,
                                     (size_t)bs * (orows - 1) + ob, hipMemcpyDeviceToHost, I->stream));
            SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
        } else if (orows > 0 && g.W > 2 && direct_out) {
            SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
            SC_HIP(I, hipStreamSynchronize(I->stream));              // the kernel's stores are in host memory when it has ended
            if (check_after_sync) {
                I->scan_fence = nullptr;
                if (memcmp(guess, I->h_rect + 4, 4 * sizeof(int)) != 0) return SC_GUESS_WRONG;
            }
            copy_rows(I, body + (size_t)(g.lty + 1) * bs + 3 * (g.ltx + 1), (size_t)bs, (const uint8_t *)I->h_out.p + dfs + 3, (size_t)dfs, ob, orows);
        } else if (orows > 0 && g.W > 2) {
            const int dfs = out_pitch;            // (shadows the compact pitch: the warm-up variant returns at the body buffer's)
            if ((r = ensure_pinned(I, I->h_out, (size_t)dfs * g.H + 64))) return r;
            uint8_t *dst_org = body + (size_t)(g.lty + 1) * bs + 3 * (g.ltx + 1);
            const uint8_t *src = out_dev + dfs;                                   // ROI row 1
            uint8_t *stage = (uint8_t *)I->h_out.p + dfs;
            // pieces that SHRINK towards the end: the splice of piece k runs while piece k + 1 crosses the link, and what is left
            // after the last DMA is the splice of a small piece only (equal pieces left 1/3 of the image to splice behind the link)
            int starts[9], pieces = 0;
            {
                const size_t total = (size_t)dfs * orows;
                size_t left = total;
                int yy = 0;
                while (yy < orows && pieces < 7) {
                    size_t want = left > ((size_t)3 << 20) ? left / 2 : left;          // halves, down to ~1.5-3 MB
                    int n = (int)std::max<size_t>(1, want / dfs);
                    if (orows - yy - n < 8) n = orows - yy;
                    starts[pieces++] = yy;
                    yy += n; left = (size_t)dfs * (orows - yy);
                }
                if (yy < orows) starts[pieces++] = yy;
                starts[pieces] = orows;
            }
            for (int k = 0; k < pieces; ++k) {
                const int y0 = starts[k], n = starts[k + 1] - y0;
                SC_HIP(I, hipMemcpyAsync(stage + (size_t)y0 * dfs, src + (size_t)y0 * dfs, (size_t)dfs * (n - 1) + 3 + ob,
                                         hipMemcpyDeviceToHost, I->stream));
                SC_HIP(I, hipEventRecord(I->ev_chunk[k], I->stream));
            }
            SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
            for (int k = 0; k < pieces; ++k) {
                const int y0 = starts[k], n = starts[k + 1] - y0;
                SC_HIP(I, hipEventSynchronize(I->ev_chunk[k]));
                copy_rows(I, dst_org + (size_t)y0 * bs, (size_t)bs, stage + (size_t)y0 * dfs + 3, (size_t)dfs, ob, n);
            }
        } else {
            SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
        }
        SC_HIP(I, hipStreamSynchronize(I->stream));
        if (check_after_sync && I->scan_fence) {      // (an ROI without interior rows: nothing was spliced above)
            I->scan_fence = nullptr;
            if (memcmp(guess, I->h_rect + 4, 4 * sizeof(int)) != 0) return SC_GUESS_WRONG;
        }
        return r;
    };
    Geo g{};
    bool done = false;
    I->guard = RectGuard();
    if (predicted) {
        // launch on the predicted box; the bbox kernel's answer is checked before anything reaches the caller
        if ((rc = bbox_enqueue(I, dmask, mc, mr, dms, &gp))) return rc;
        I->guard = make_guard(I, guess);
        rc = attempt(gp, guess, pre.face ? &pre : nullptr);
        I->guard = RectGuard();
        if (rc == SC_GUESS_WRONG) {               // repeat on the true box (the mask stays where it is, the ROIs go up the ordinary way), pause speculation for a while
            I->spec_cooldown = 8;
            if ((rc = geo_from_rect(I, I->h_rect + 4, cx, cy, g))) return rc;
            fill_info_geo(I, g);
            if ((rc = check_roi(I, g, bc, br))) return rc;
            rc = attempt(g, nullptr, nullptr);
        } else {
            g = gp;
        }
        if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;
        done = true;
    }
    if (!done) {                                      // no usable prediction: wait for the device's box first
        if ((rc = device_bbox(I, dmask, mc, mr, dms, cx, cy, g))) return rc;
        fill_info_geo(I, g);
        if ((rc = check_roi(I, g, bc, br))) return rc;
        rc = attempt(g, nullptr, nullptr);
        if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;
    }
    fill_info_geo(I, g);
    remember_rect(I, mc, mr, I->h_rect + 4);
    finish_timing(I, true);
    I->info.ms_call = ev_ms(I->tm[0], I->ev_k1);
    if (bSync) {      // the reference's bSync: time the call on the stream and say so (seamlessClone_imp.cu:336-349)
        printf("Compute stage performance time= %.3f msec, patch size=%dx%d\n", I->info.ms_call, g.W, g.H);      // the reference's ucMask has the ROI's size by then (seamlessClone_imp.cpp:1024)
        printf("total device memory used: %zu\n", I->arena_bytes);
        fflush(stdout);
    }
    return rc;
}

int sc_hip_run_device(void *p, const uint8_t *d_face, int fc, int fr, int fs, uint8_t *d_body, int bc, int br, int bs,
                      const uint8_t *d_mask, int mc, int mr, int ms, int cx, int cy, bool bSync)
{
    Instance *I = get(p);
    if (!I) return SC_ERR_BAD_ARG;
    I->err.clear();
    I->info.field_retry = 0; I->info.new_size = 0; I->info.group_members = 0; I->info.group_ragged = 0;
    SC_HIP(I, hipSetDevice(I->gpu));
    int rc = validate_images(I, d_face, fc, fr, fs, d_body, bc, br, bs, d_mask, mc, mr, ms);
    if (rc) return rc;
    I->stage_marks = bSync;        // the stage timeline (sc_run_info::ms_*) is filled for synchronous calls only, like the
                                   // reference's bSync timing: each mark is an event in the stream and a ~5 us bubble behind it
    I->marks_ends_only = bSync && (I->opts.flags & SC_FLAG_NO_STAGE_MARKS);
    if ((rc = tmark(I, 0))) return rc;
    if ((rc = tmark(I, 1, true))) return rc;       // nothing to upload: images are device resident
    const int passes = I->opts.reference_warmup ? 2 : 1;
    auto attempt = [&](const Geo &g) -> int {
        int r = tmark(I, 3, true);
        if (r) return r;
        r = device_clone(I, d_mask, ms, mr, d_face + (size_t)g.y0 * fs + 3 * g.x0, fs,
                             d_body + (size_t)g.lty * bs + 3 * g.ltx, bs, g, passes);
        if (r != SC_OK && r != SC_ERR_NOT_CONVERGED) return r;
        SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
        return r;
    };
    Geo g{};
    bool done = false;
    int guess[4];
    I->guard = RectGuard();
    // a predicted box needs the final wait to be checked against the device's answer: synchronous calls only
    if (bSync && predict_rect(I, mc, mr, guess)) {
        Geo gp{};
        if (geo_from_rect(I, guess, cx, cy, gp) == SC_OK && check_roi(I, gp, bc, br) == SC_OK) {
            if ((rc = bbox_enqueue(I, d_mask, mc, mr, ms, &gp))) return rc;
            I->guard = make_guard(I, guess);
            rc = attempt(gp);
            I->guard = RectGuard();
            if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;
            SC_HIP(I, hipStreamSynchronize(I->stream));
            if (memcmp(guess, I->h_rect + 4, sizeof(guess)) == 0) {
                g = gp;
            } else {                                  // wrong guess: the destination was not touched
                I->spec_cooldown = 8;
                if ((rc = geo_from_rect(I, I->h_rect + 4, cx, cy, g))) return rc;
                fill_info_geo(I, g);
                if ((rc = check_roi(I, g, bc, br))) return rc;
                rc = attempt(g);
                if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;
                SC_HIP(I, hipStreamSynchronize(I->stream));
            }
            done = true;
        }
        I->err.clear();
    }
    if (!done) {
        if ((rc = device_bbox(I, d_mask, mc, mr, ms, cx, cy, g))) return rc;
        fill_info_geo(I, g);
        if ((rc = check_roi(I, g, bc, br))) return rc;
        rc = attempt(g);
        if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;
        if (bSync) SC_HIP(I, hipStreamSynchronize(I->stream));
    }
    fill_info_geo(I, g);
    remember_rect(I, mc, mr, I->h_rect + 4);
    if (bSync) finish_timing(I, false);
    else I->info.ms_h2d = I->info.ms_mask = I->info.ms_pre = I->info.ms_solve = I->info.ms_post = I->info.ms_d2h = I->info.ms_device_total = 0.f;
    return rc;
}

// ---- n device-resident clones through ONE set of launches ------------------------------------------------------
// The solver treats the channels of a field as independent planes, so n clones whose ROIs have the same size are one
// field of 3n channels: every multigrid launch is n times larger (the coarse levels stop being launch-latency bound,
// the level-1 grid fills whole rounds of workgroup slots) and there are 27 solver launches for the group instead of
// 27 n.  Masks, positions and images are per clone (bounding box, erode, pre- and post-process each go out as one launch
// for the group, blockIdx.z = member); the stop rule sees the largest correction of the group.
// Round 5: the members of a call are PARTITIONED by ROI size -- every sub-group of two or more same-size members shares one
// set of launches, the rest run alone -- instead of the whole call falling back to one clone at a time as soon as one member
// differs (a batch of real clones has a mask box per face / frame).  A failing member, the reference's warm-up option and
// OpenCV's grey-mask semantics still run one after the other through sc_hip_run_device.
namespace {

constexpr int GROUP_RS = 32;      // ints between the rectangles of a group's scans: one 128-byte line each (eight rectangles in one line: 162 us for the group's scan instead of 20)

struct RagScope {      // leaves the size-class mode on every way out
    Instance *I;
    ~RagScope() { rag_end(I); }
};

// members idx[0..n) of `jobs` as one field of 3n channels: all with the same ROI size (plans == nullptr), or a SIZE CLASS (sc_ragged.cpp:
// plans[k] = member idx[k]'s plan; the fields take the class's largest width and height).  guess: the predicted rectangles the
// members were launched on (nullptr: their boxes are the device's), d_r: the device rectangles of ALL members of the call
int run_group_members(Instance *I, sc_batch_job *jobs, const std::vector<int> &idx, const std::vector<Geo> &geo, const int *guess, int *d_r,
                      const std::vector<SizePlan> *plans)
{
    const int n = (int)idx.size();
    Geo g0 = geo[idx[0]];
    if (plans) for (int k = 1; k < n; ++k) { g0.W = std::max(g0.W, geo[idx[k]].W); g0.H = std::max(g0.H, geo[idx[k]].H); }
    int rc;
    I->mpitch = round_up(g0.W, 64);
    const size_t mplane = (size_t)I->mpitch * g0.H;
    if ((rc = ensure(I, I->d_M, mplane * n, false))) return rc;
    if ((rc = setup_fields(I, g0.W, g0.H, 3 * n))) return rc;
    RagScope scope{ I };
    std::vector<MaskJob> mj(n);
    std::vector<ImageJob> ij(n);
    for (int k = 0; k < n; ++k) {
        const int i = idx[k];
        const sc_batch_job &j = jobs[i];
        mj[k] = MaskJob{};
        mj[k].mask = j.mask; mj[k].mw = j.mask_cols; mj[k].mh = j.mask_rows; mj[k].mstep = j.mask_step;
        mj[k].rect = d_r + GROUP_RS * i;
        mj[k].g = geo[i]; mj[k].M = (uint8_t *)I->d_M.p + mplane * k; mj[k].mpitch = I->mpitch;
        ij[k].face_org = j.face + (size_t)geo[i].y0 * j.face_step + 3 * geo[i].x0; ij[k].fstep = j.face_step;
        ij[k].body_org = j.body + (size_t)geo[i].lty * j.body_step + 3 * geo[i].ltx; ij[k].bstep = j.body_step;
        ij[k].M = (const uint8_t *)I->d_M.p + mplane * k;
        ij[k].d_rect = guess ? d_r + GROUP_RS * i : nullptr;
        if (guess) { ij[k].rx0 = guess[4 * i]; ij[k].rx1 = guess[4 * i + 1]; ij[k].ry0 = guess[4 * i + 2]; ij[k].ry1 = guess[4 * i + 3]; }
        if (plans) { ij[k].W = geo[i].W; ij[k].H = geo[i].H; }
    }
    // (a size class: the members' table goes up FIRST -- 9 us of host time, a 3-us copy in front of the erode -- so that the other
    //  streams' builds, which wait for it, run beside the erode and the pre-process)
    if (plans && (rc = rag_begin_table(I, *plans))) return rc;
    bool builds_done = false;
    launch_mask_erode3_group(mj.data(), n, I->stream);
    I->erode_done = false;
    int solve_rc = SC_OK;
    for (;;) {
        I->result_in_U1 = false;
        I->f_half = mg_reads_half_rhs(I);
        I->u_half = I->f_half && !(I->opts.flags & SC_FLAG_FLOAT_U0);
        launch_preprocess_group(ij.data(), n, I->mpitch, I->U0, I->F, I->stream, I->f_half, I->u_half);
        SC_HIP(I, hipGetLastError());
        // a size class: the launches that build its per-call state (rag_begin_builds: 12 us of host time) go in HERE, while the device
        // erodes and pre-processes -- neither reads the table (member sizes travel in the ImageJobs); with all of rag_begin in front of
        // the erode the device idled for as long (16 x 320^2: the first level-0 launch started 119 us into the call, now ~80)
        if (plans && !builds_done) {
            if ((rc = rag_begin_builds(I))) return rc;
            builds_done = true;
        }
        // --- one solve for the group, results spliced per clone
        I->info.sweep_launches = 0;
        I->guard = RectGuard();
        I->spec_post.group = ij;
        I->spec_post.ev_solved = nullptr;
        I->spec_post.armed = true; I->spec_post.done = false;     // the solver enqueues the splices behind the cycle it expects to accept
        solve_rc = solve(I);
        I->spec_post.armed = false;
        I->force_float_field = false;
        if (solve_rc != SC_RETRY_FLOAT_FIELD) break;
        I->force_float_field = true;       // a member's 16-bit field saturated: no member was written, the group again on float fields
        I->info.field_retry = 1;
    }
    const bool spliced = I->spec_post.done;
    I->spec_post.group.clear();
    if (solve_rc != SC_OK && solve_rc != SC_ERR_NOT_CONVERGED) return solve_rc;
    if (!spliced) {
        LmNodes lm;
        if ((rc = output_nodes(I, lm))) return rc;
        launch_postprocess_group(result(I), ij.data(), n, I->stream, lm);
    }
    for (int k = 0; k < n; ++k) jobs[idx[k]].rc = solve_rc;
    SC_HIP(I, hipGetLastError());
    fill_info_geo(I, g0);
    I->info.group_members = n; I->info.group_ragged = plans ? 1 : 0;
    return solve_rc;
}

} // namespace

int sc_hip_run_device_batch(void *p, sc_batch_job *jobs, int n)
{
    Instance *I = get(p);
    if (!I || !jobs || n <= 0) return SC_ERR_BAD_ARG;
    I->err.clear();
    I->info.field_retry = 0; I->info.new_size = 0; I->info.group_members = 0; I->info.group_ragged = 0;
    SC_HIP(I, hipSetDevice(I->gpu));
    auto worse = [](int worst, int rc) { return (rc != SC_OK && (worst == SC_OK || worst == SC_ERR_NOT_CONVERGED)) ? rc : worst; };
    auto alone = [&](int i) -> int {
        sc_batch_job &j = jobs[i];
        j.rc = sc_hip_run_device(p, j.face, j.face_cols, j.face_rows, j.face_step, j.body, j.body_cols, j.body_rows, j.body_step,
                                 j.mask, j.mask_cols, j.mask_rows, j.mask_step, j.centerX, j.centerY, false);
        return j.rc;
    };
    auto one_by_one = [&]() -> int {
        int worst = SC_OK;
        for (int i = 0; i < n; ++i) worst = worse(worst, alone(i));
        return worst;
    };
    {   // refresh the destinations that ask for it: one launch per 16 (k_copy_group); odd alignments take the runtime's copy
        CopyJobs cj{};
        int cn = 0;
        auto flush = [&]() { if (cn) { launch_copy_group(cj, cn, I->stream); cn = 0; } };
        for (int i = 0; i < n; ++i) {
            const sc_batch_job &j = jobs[i];
            if (!j.body_restore) continue;
            const size_t bytes = (size_t)j.body_step * j.body_rows;
            if ((((uintptr_t)j.body | (uintptr_t)j.body_restore) & 15) != 0) {
                SC_HIP(I, hipMemcpyAsync(j.body, j.body_restore, bytes, hipMemcpyDeviceToDevice, I->stream));
                continue;
            }
            cj.dst[cn] = j.body; cj.src[cn] = j.body_restore; cj.bytes[cn] = bytes;
            if (++cn == CopyJobs::MAX) flush();
        }
        flush();
        SC_HIP(I, hipGetLastError());
    }
    if (n == 1 || I->opts.reference_warmup || (I->opts.flags & SC_FLAG_OPENCV_GREY_MASK)) return one_by_one();
    // members whose images do not even validate run alone (and report their own error); the others are candidates for a group
    std::vector<char> usable(n, 1);
    int nusable = 0;
    for (int i = 0; i < n; ++i) {
        const sc_batch_job &j = jobs[i];
        if (validate_images(I, j.face, j.face_cols, j.face_rows, j.face_step, j.body, j.body_cols, j.body_rows, j.body_step,
                            j.mask, j.mask_cols, j.mask_rows, j.mask_step) != SC_OK) usable[i] = 0;
        else ++nusable;
    }
    I->err.clear();
    if (nusable < 2) return one_by_one();
    I->stage_marks = false;
    // --- bounding boxes of all masks, one read-back
    int rc;
    constexpr int RS = GROUP_RS;
    if ((rc = ensure(I, I->d_rects, (size_t)n * RS * sizeof(int)))) return rc;
    if ((rc = ensure_pinned(I, I->h_rects, (size_t)n * 2 * RS * sizeof(int)))) return rc;
    // (the fold launch writes every usable member's rectangle to d_r AND into the pinned h_out: no seeds to upload, nothing to read back)
    int *h_out = (int *)I->h_rects.p + RS * n, *d_r = (int *)I->d_rects.p;
    {
        std::vector<MaskJob> mj;
        mj.reserve(n);
        for (int i = 0; i < n; ++i) {
            if (!usable[i]) continue;
            MaskJob m{};
            m.mask = jobs[i].mask; m.mw = jobs[i].mask_cols; m.mh = jobs[i].mask_rows; m.mstep = jobs[i].mask_step;
            m.rect = d_r + RS * i; m.rect_host = h_out + RS * i;
            mj.push_back(m);
        }
        if ((rc = ensure(I, I->d_bbox_parts, sizeof(int) * mask_bbox_group_parts(mj.data(), (int)mj.size())))) return rc;
        launch_mask_bbox_group(mj.data(), (int)mj.size(), I->stream, (int *)I->d_bbox_parts.p);
    }
    SC_HIP(I, hipGetLastError());
    if (!I->ev_rects) SC_HIP(I, hipEventCreateWithFlags(&I->ev_rects, hipEventDisableTiming));
    SC_HIP(I, hipEventRecord(I->ev_rects, I->stream));
    // Like a single clone (predict_rect), the members are launched on PREDICTED bounding boxes -- the interior of every mask,
    // which is what a mask that touches its four inner borders gives -- while the scans' answers are in flight: no host
    // wait in front of the erodes.  Every member's splice carries its guess and writes nothing unless the device found
    // that box; the host compares when the answers are in (they are by the time the solver has waited for its stop rule)
    // and repeats the members that were guessed wrong, one by one on their true boxes.
    std::vector<Geo> geo(n);
    std::vector<int> guess(4 * (size_t)n);
    bool speculative = !(I->opts.flags & SC_FLAG_NO_SPECULATE) && I->group_spec_cooldown == 0;
    if (I->group_spec_cooldown > 0) --I->group_spec_cooldown;
    std::vector<char> grouped(n, 0);          // the member's geometry is known (or predicted) and fits its destination
    if (speculative) {
        for (int i = 0; i < n; ++i) {
            if (!usable[i]) continue;
            int *r = &guess[4 * i];
            r[0] = 1; r[1] = jobs[i].mask_cols - 2; r[2] = 1; r[3] = jobs[i].mask_rows - 2;
            // (a member whose guess does not fit -- a mask narrower than three pixels, a box that leaves the destination -- runs alone on its true box)
            grouped[i] = jobs[i].mask_cols >= 3 && jobs[i].mask_rows >= 3 && geo_from_rect(I, r, jobs[i].centerX, jobs[i].centerY, geo[i]) == SC_OK &&
                         check_roi(I, geo[i], jobs[i].body_cols, jobs[i].body_rows) == SC_OK;
        }
    } else {
        SC_HIP(I, hipStreamSynchronize(I->stream));
        for (int i = 0; i < n; ++i) {
            if (!usable[i]) continue;
            grouped[i] = geo_from_rect(I, h_out + RS * i, jobs[i].centerX, jobs[i].centerY, geo[i]) == SC_OK &&
                         check_roi(I, geo[i], jobs[i].body_cols, jobs[i].body_rows) == SC_OK;
        }
    }
    I->err.clear();
    // --- partition (first-come order inside a sub-group and between them): same-size members share one set of launches as they
    //     are, members of one size class (sc_ragged.cpp: different sizes, the same solve) through the per-member table
    std::vector<int> cand;
    std::vector<SizePlan> plans;
    bool one_size = true;
    for (int i = 0; i < n; ++i) {
        if (!grouped[i]) continue;
        if (!cand.empty() && (geo[i].W != geo[cand[0]].W || geo[i].H != geo[cand[0]].H)) one_size = false;
        cand.push_back(i);
    }
    std::vector<std::vector<int>> parts;          // indices into cand / plans
    if (one_size && cand.size() >= 2) {
        // every member has the same ROI size (a benchmark's batch, a tiled image): one field of 3n channels as in rounds 2-4, and
        // nothing to plan -- sixteen memo look-ups per call and thirty-two in the pool were 0.5 % of the 2048^2 step
        parts.emplace_back(cand.size());
        for (size_t k = 0; k < cand.size(); ++k) parts[0][k] = (int)k;
        plans.resize(cand.size());
    } else {
        plans.resize(cand.size());
        for (size_t k = 0; k < cand.size(); ++k) plan_size(I->opts, geo[cand[k]].W, geo[cand[k]].H, plans[k]);
        plan_groups(plans, n, parts);
    }
    int worst = SC_OK;
    sc_run_info keep{};
    bool have_group = false;
    std::vector<int> singles;
    for (int i = 0; i < n; ++i) if (!grouped[i]) singles.push_back(i);
    for (const auto &pq : parts) {
        if (pq.size() < 2) { singles.push_back(cand[pq[0]]); grouped[cand[pq[0]]] = 0; continue; }
        std::vector<int> q(pq.size());
        std::vector<SizePlan> qp;
        bool uniform = true;
        for (size_t k = 0; k < pq.size(); ++k) {
            q[k] = cand[pq[k]];
            uniform = uniform && geo[q[k]].W == geo[q[0]].W && geo[q[k]].H == geo[q[0]].H;
        }
        if (!uniform) for (int k : pq) qp.push_back(plans[k]);
        rc = run_group_members(I, jobs, q, geo, speculative ? guess.data() : nullptr, d_r, uniform ? nullptr : &qp);
        if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) return rc;          // a HIP error: nothing more can be trusted on this stream
        worst = worse(worst, rc);
        keep = I->info; have_group = true;
    }
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    I->info.ms_h2d = I->info.ms_mask = I->info.ms_pre = I->info.ms_solve = I->info.ms_post = I->info.ms_d2h = I->info.ms_device_total = 0.f;
    if (have_group) keep = I->info;
    if (speculative && have_group) {
        SC_HIP(I, hipEventSynchronize(I->ev_rects));       // long since passed when the solver has waited for its stop rule
        for (int i = 0; i < n; ++i)
            if (grouped[i] && memcmp(&guess[4 * i], h_out + RS * i, 4 * sizeof(int)) != 0) {
                I->group_spec_cooldown = 8;
                singles.push_back(i);                          // its destination was not touched: repeat it alone on its true box
            }
    }
    for (int i : singles) worst = worse(worst, alone(i));
    if (have_group) I->info = keep;                            // the statistics of the (last) group, not of a straggler
    return worst;
}

int sc_hip_plan_size(int W, int H, const sc_solver_opts *opts, int out[12])
{
    if (!out) return SC_ERR_BAD_ARG;
    sc_solver_opts o;
    if (opts) o = *opts; else sc_hip_default_opts(&o);
    SizePlan p;
    plan_size(o, W, H, p);
    const int v[12] = { p.ok ? 1 : 0, p.nl, p.tail, p.npx, p.npy, p.Kxp, p.Kyp, p.nxt, p.nrs, (p.t && p.tail > 0) ? p.t->g[p.tail].x.nc * 1000 + p.t->g[p.tail].y.nc : 0,
                        p.solo_differs ? 1 : 0, p.conditional ? 1 : 0 };
    memcpy(out, v, sizeof(v));
    return SC_OK;
}

int sc_hip_plan_groups(const int *wh, int n, int cap, const sc_solver_opts *opts, int *group_of, int *kind_of)
{
    if (!wh || n < 1 || !group_of) return SC_ERR_BAD_ARG;
    sc_solver_opts o;
    if (opts) o = *opts; else sc_hip_default_opts(&o);
    std::vector<SizePlan> plans(n);
    for (int i = 0; i < n; ++i) plan_size(o, wh[2 * i], wh[2 * i + 1], plans[i]);
    std::vector<std::vector<int>> groups;
    plan_groups(plans, cap > 0 ? cap : n, groups);
    for (size_t g = 0; g < groups.size(); ++g) {
        bool uniform = true;
        for (int i : groups[g]) uniform = uniform && plans[i].W == plans[groups[g][0]].W && plans[i].H == plans[groups[g][0]].H;
        for (int i : groups[g]) {
            group_of[i] = (int)g;
            if (kind_of) kind_of[i] = groups[g].size() < 2 ? 0 : uniform ? 1 : plans[i].solo_differs ? 3 : 2;
        }
    }
    return (int)groups.size();
}

int sc_hip_plan_groups_pool(const int *wh, int n, int group, int streams, const sc_solver_opts *opts, int *group_of, int *kind_of)
{
    if (!wh || n < 1 || !group_of || group < 0 || group > 64 || streams < 1) return SC_ERR_BAD_ARG;
    sc_solver_opts o;
    if (opts) o = *opts; else sc_hip_default_opts(&o);
    // as sc_hip_pool_run: largest first, then the planner under the pool's caps
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return (long)(wh[2 * x] + 2) * (wh[2 * x + 1] + 2) > (long)(wh[2 * y] + 2) * (wh[2 * y + 1] + 2); });
    std::vector<SizePlan> plans(n);
    for (int i = 0; i < n; ++i) plan_size(o, wh[2 * order[i]], wh[2 * order[i] + 1], plans[i]);
    int cap, cap_max;
    long budget;
    pool_group_caps(group, n, streams, cap, cap_max, budget);
    std::vector<std::vector<int>> groups;
    plan_groups(plans, cap, groups, cap_max, budget);
    for (size_t g = 0; g < groups.size(); ++g) {
        bool uniform = true;
        for (int i : groups[g]) uniform = uniform && plans[i].W == plans[groups[g][0]].W && plans[i].H == plans[groups[g][0]].H;
        for (int i : groups[g]) {
            group_of[order[i]] = (int)g;
            if (kind_of) kind_of[order[i]] = groups[g].size() < 2 ? 0 : uniform ? 1 : plans[i].solo_differs ? 3 : 2;
        }
    }
    return (int)groups.size();
}

int sc_hip_plan_prepare(const int *wh, int n, const sc_solver_opts *opts)
{
    if (!wh || n < 1) return SC_ERR_BAD_ARG;
    sc_solver_opts o;
    if (opts) o = *opts; else sc_hip_default_opts(&o);
    int eligible = 0;
    for (int i = 0; i < n; ++i) {
        SizePlan p;
        if (plan_size(o, wh[2 * i], wh[2 * i + 1], p)) ++eligible;
    }
    return eligible;
}

void sc_hip_plan_cache_clear(void) { plan_cache_clear(); }

int sc_hip_reference_tables_singular(int w, int h)
{
    if (w < 1 || h < 1) return 0;
    const double PIf = (double)3.14159265358979323846f;           // seamlessClone_imp.h:17
    const float fx0 = (float)(2.0 * std::cos(PIf / (w + 1.0))), fy0 = (float)(2.0 * std::cos(PIf / (h + 1.0)));
    return ((fx0 + fy0) - 4.0f < 0.0f) ? 0 : 1;
}

int sc_hip_selftest_host(void)
{
    // 1: row copier -- strided copy of an awkward shape through the parked helpers, twice (reuse of the pool)
    {
        RowCopier rc(5);
        const int rows = 1237, rb = 3001, sp = 3100, dp = 3072;
        std::vector<uint8_t> src((size_t)rows * sp), dst((size_t)rows * dp, 0);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)(i * 2654435761u >> 24);
        for (int rep = 0; rep < 2; ++rep) {
            std::fill(dst.begin(), dst.end(), 0);
            const int per = 17, parts = (rows + per - 1) / per;
            rc.parallel(parts, [&](int i) {
                for (int y = i * per; y < std::min(rows, (i + 1) * per); ++y) memcpy(&dst[(size_t)y * dp], &src[(size_t)y * sp], rb);
            });
            for (int y = 0; y < rows; ++y) {
                if (memcmp(&dst[(size_t)y * dp], &src[(size_t)y * sp], rb) != 0) return 1;
                for (int x = rb; x < dp; ++x) if (dst[(size_t)y * dp + x]) return 1;
            }
        }
        int hits = 0;
        rc.parallel(1, [&](int) { ++hits; });              // single part runs inline
        rc.parallel(0, [&](int) { ++hits; });
        if (hits != 1) return 1;
    }
    // 2: eigen-decomposition of the 1-D level operators (the QL reference), 4: the closed form the device builds from against it
    if (!(sc::fd_selftest_error() < 1e-11)) return 2;
    if (!(sc::fd_closed_selftest_error() < 1e-10)) return 4;
    // 3: which parts of a level-0 launch make up each cell row of the float-table correction (sc_lowmode.hip)
    if (sc::lowmode_part_map_selftest() != 0) return 3;
    // 5: the pruned search for the correction's largest ratio (plan_size) against the full table's maximum, over sizes of every kind
    //    (square, elongated, the 2100s and 3000s where one size in ten crosses the 4 % line)
    {
        std::vector<float> R(256 * 256);
        const int ws[] = { 46, 98, 154, 300, 511, 640, 1000, 1027, 1100, 1555, 2046, 2051, 2105, 2118, 2135, 2400, 3118, 3328, 3468, 4096, 6000, 9000 };
        for (int w : ws)
            for (int dh = -7; dh <= 7; ++dh) {
                for (int h : { w + 3 * dh, w / 3 + dh + 40 }) {
                    if (h < 4) continue;
                    const int Kx = sc::lowmode_count(w), Ky = sc::lowmode_count(h), Kxp = (Kx + 31) / 32 * 32;
                    double pruned = -1.0, full = -2.0;
                    const bool a = sc::lowmode_ratio(w, h, Kx, Ky, Kxp, nullptr, pruned), b = sc::lowmode_ratio(w, h, Kx, Ky, Kxp, R.data(), full);
                    if (a != b || pruned != full) return 5;
                }
            }
    }
    return 0;
}

// ---------------------------------------------------------------- stage-level hooks

int sc_hip_mask_stage(void *p, const uint8_t *mask, int mc, int mr, int ms, int cx, int cy, int geo[6], uint8_t *M_out,
                      size_t M_capacity)
{
    Instance *I = get(p);
    if (!I || !mask || !geo) return SC_ERR_BAD_ARG;
    I->err.clear();
    SC_HIP(I, hipSetDevice(I->gpu));
    if (mc <= 0 || mr <= 0 || ms < mc) return SC_ERR_BAD_SIZE;
    int rc;
    const int dms = round_up(mc, 256);
    if ((rc = ensure(I, I->d_mask, (size_t)dms * mr))) return rc;
    if ((rc = upload_rows(I, I->h_mask, I->d_mask.p, dms, mask, ms, mc, mr))) return rc;
    Geo g;
    if ((rc = device_bbox(I, (const uint8_t *)I->d_mask.p, mc, mr, dms, cx, cy, g))) return rc;
    fill_info_geo(I, g);
    geo[0] = g.x0; geo[1] = g.y0; geo[2] = g.W; geo[3] = g.H; geo[4] = g.ltx; geo[5] = g.lty;
    I->mpitch = round_up(g.W, 64);
    if ((rc = ensure(I, I->d_M, (size_t)I->mpitch * g.H, false))) return rc;
    if (I->opts.flags & SC_FLAG_OPENCV_GREY_MASK) launch_mask_erode_min7((const uint8_t *)I->d_mask.p, dms, g, (uint8_t *)I->d_M.p, I->mpitch, I->stream);
    else launch_mask_erode3((const uint8_t *)I->d_mask.p, dms, mr, g, (uint8_t *)I->d_M.p, I->mpitch, I->stream);
    SC_HIP(I, hipGetLastError());
    if (M_out) {
        if (M_capacity < (size_t)g.W * g.H) return SC_ERR_BAD_SIZE;
        if ((rc = download_rows(I, I->h_out, M_out, g.W, I->d_M.p, I->mpitch, g.W, g.H))) return rc;
    }
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

// planar float field -> dense [C][H][W] host array.  The planes of a field are contiguous (plane = pitch * H), so the
// whole field is C * H rows at one pitch.
static int download_field(Instance *I, const Field &f, float *out)
{
    return download_rows(I, I->h_out, (uint8_t *)out, (size_t)f.W * sizeof(float), f.p, (size_t)f.pitch * sizeof(float),
                         (size_t)f.W * sizeof(float), f.C * f.H);
}

int sc_hip_build_rhs(void *p, const uint8_t *face, int fc, int fr, int fs, const uint8_t *body, int bc, int br, int bs,
                     const uint8_t *mask, int mc, int mr, int ms, int cx, int cy, int geo[6], float *B_out,
                     float *lap_out, size_t plane_capacity)
{
    Instance *I = get(p);
    if (!I || !geo) return SC_ERR_BAD_ARG;
    I->err.clear();
    SC_HIP(I, hipSetDevice(I->gpu));
    int rc = validate_images(I, face, fc, fr, fs, body, bc, br, bs, mask, mc, mr, ms);
    if (rc) return rc;
    if ((rc = sc_hip_mask_stage(p, mask, mc, mr, ms, cx, cy, geo, nullptr, 0))) return rc;
    Geo g{ geo[0], geo[1], geo[2], geo[3], geo[4], geo[5] };
    if ((rc = check_roi(I, g, bc, br))) return rc;
    if (plane_capacity < (size_t)g.W * g.H) return SC_ERR_BAD_SIZE;
    const int dfs = round_up(3 * g.W, 256);
    if ((rc = ensure(I, I->d_face, (size_t)dfs * g.H))) return rc;
    if ((rc = ensure(I, I->d_body_roi, (size_t)dfs * g.H))) return rc;
    if ((rc = upload_rows(I, I->h_face, I->d_face.p, dfs, face + (size_t)g.y0 * fs + 3 * g.x0, fs, 3 * (size_t)g.W, g.H))) return rc;
    if ((rc = upload_rows(I, I->h_body, I->d_body_roi.p, dfs, body + (size_t)g.lty * bs + 3 * g.ltx, bs, 3 * (size_t)g.W, g.H))) return rc;
    if ((rc = setup_fields(I, g.W, g.H, 3))) return rc;
    launch_preprocess((const uint8_t *)I->d_body_roi.p, dfs, (const uint8_t *)I->d_face.p, dfs,
                      (const uint8_t *)I->d_M.p, I->mpitch, I->U0, I->U1, I->F, I->stream, false, false, (I->opts.flags & SC_FLAG_OPENCV_GREY_MASK) != 0);
    SC_HIP(I, hipGetLastError());
    if (B_out && (rc = download_field(I, I->U0, B_out))) return rc;
    if (lap_out && (rc = download_field(I, I->F, lap_out))) return rc;
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_field_load(void *p, int W, int H, int C, const float *U, const float *lap)
{
    Instance *I = get(p);
    if (I) field_moved(I);
    if (I) I->out_direct = false;
    if (!I || !U || !lap) return SC_ERR_BAD_ARG;
    I->err.clear();
    SC_HIP(I, hipSetDevice(I->gpu));
    if (W < 1 || H < 1 || C < 1 || C > 16) return SC_ERR_BAD_SIZE;
    int rc;
    if ((rc = setup_fields(I, W, H, C))) return rc;
    // deterministic pads
    SC_HIP(I, hipMemsetAsync(I->d_U0.p, 0, I->U0.bytes(), I->stream));
    SC_HIP(I, hipMemsetAsync(I->d_U1.p, 0, I->U1.bytes(), I->stream));
    SC_HIP(I, hipMemsetAsync(I->d_F.p, 0, I->F.bytes(), I->stream));
    const size_t wb = (size_t)W * sizeof(float), pb = (size_t)I->U0.pitch * sizeof(float);
    // a field's planes are contiguous: C * H rows at one pitch, one packed upload each (separate staging buffers)
    if ((rc = upload_rows(I, I->h_face, I->U0.p, pb, (const uint8_t *)U, wb, wb, C * H))) return rc;
    if ((rc = upload_rows(I, I->h_body, I->F.p, pb, (const uint8_t *)lap, wb, wb, C * H))) return rc;
    SC_HIP(I, hipMemcpyAsync(I->U1.p, I->U0.p, pb * (size_t)(C * H - 1) + wb, hipMemcpyDeviceToDevice, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_field_sweep(void *p, int method, int sweeps, float omega, int spl)
{
    Instance *I = get(p);
    if (I && I->out_direct) { I->err = "the last clone kept no solution field (set SC_FLAG_KEEP_FIELD to keep it)"; return SC_ERR_BAD_ARG; }
    if (I && I->f_half) { int frc = float_rhs(I); if (frc) return frc; }
    if (!I || !I->F.p) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    I->info.sweep_launches = 0;
    int rc = run_sweeps(I, method, sweeps, omega, spl);
    if (rc) return rc;
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_field_residual(void *p, double out[2])
{
    Instance *I = get(p);
    if (I && I->out_direct) { I->err = "the last clone kept no solution field (set SC_FLAG_KEEP_FIELD to keep it)"; return SC_ERR_BAD_ARG; }
    if (I && I->f_half) { int frc = float_rhs(I); if (frc) return frc; }
    if (!I || !I->F.p || !out) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    return eval_residual(I, out);
}

int sc_hip_field_solve(void *p)
{
    Instance *I = get(p);
    if (I && I->f_half) { int frc = float_rhs(I); if (frc) return frc; }
    if (!I || !I->F.p) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    I->info.sweep_launches = 0;
    int rc = solve(I);
    hipError_t e = hipStreamSynchronize(I->stream);
    if (e != hipSuccess) return hip_fail(I, e, "hipStreamSynchronize");
    return rc;
}

int sc_hip_field_shape(void *p, int whc[3])
{
    Instance *I = get(p);
    if (!I || !I->F.p || !whc) return SC_ERR_BAD_ARG;
    whc[0] = I->F.W; whc[1] = I->F.H; whc[2] = I->F.C;
    return SC_OK;
}

int sc_hip_field_store(void *p, float *U_out, size_t capacity_floats)
{
    Instance *I = get(p);
    if (I && I->out_direct) { I->err = "the last clone kept no solution field (set SC_FLAG_KEEP_FIELD to keep it)"; return SC_ERR_BAD_ARG; }
    if (!I || !I->F.p || !U_out) return SC_ERR_BAD_ARG;
    if (capacity_floats < (size_t)I->F.W * I->F.H * I->F.C) { I->err = "field_store: buffer too small"; return SC_ERR_BAD_SIZE; }
    SC_HIP(I, hipSetDevice(I->gpu));
    int rc = download_field(I, result(I), U_out);
    if (rc) return rc;
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_field_finish(void *p, uint8_t *body, int bc, int br, int bs, int ltx, int lty)
{
    Instance *I = get(p);
    if (I && I->out_direct) { I->err = "the last clone kept no solution field (set SC_FLAG_KEEP_FIELD to keep it)"; return SC_ERR_BAD_ARG; }
    if (!I || !I->F.p || !body) return SC_ERR_BAD_ARG;
    I->err.clear();
    SC_HIP(I, hipSetDevice(I->gpu));
    const Field &U = result(I);
    if (U.C != 3) { I->err = "field_finish: needs a 3-channel field"; return SC_ERR_BAD_SIZE; }
    if (bc <= 0 || br <= 0 || bs < 3 * bc) return SC_ERR_BAD_SIZE;
    Geo g{ 0, 0, U.W, U.H, ltx, lty };
    int rc;
    if ((rc = check_roi(I, g, bc, br))) return rc;
    const int dfs = round_up(3 * g.W, 256);
    if ((rc = ensure(I, I->d_body_roi, (size_t)dfs * g.H))) return rc;
    uint8_t *roi = body + (size_t)lty * bs + 3 * ltx;
    if ((rc = upload_rows(I, I->h_body, I->d_body_roi.p, dfs, roi, bs, 3 * (size_t)g.W, g.H))) return rc;
    launch_postprocess(U, (uint8_t *)I->d_body_roi.p, dfs, I->stream);
    SC_HIP(I, hipGetLastError());
    return download_rows(I, I->h_out, roi, bs, I->d_body_roi.p, dfs, 3 * (size_t)g.W, g.H);
}

int sc_hip_field_lowmode(void *p)
{
    Instance *I = get(p);
    if (I && I->out_direct) { I->err = "the last clone kept no solution field (set SC_FLAG_KEEP_FIELD to keep it)"; return SC_ERR_BAD_ARG; }
    if (I && I->f_half) { int frc = float_rhs(I); if (frc) return frc; }
    if (!I || !I->F.p) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    Field &U = result(I), &O = I->result_in_U1 ? I->U0 : I->U1;
    // the partner buffer receives the interior; give it the ring as well so it is a complete field
    SC_HIP(I, hipMemcpyAsync(O.p, U.p, U.bytes(), hipMemcpyDeviceToDevice, I->stream));
    int rc = lowmode_correct(I, U, O);
    if (rc) return rc;
    I->result_in_U1 = !I->result_in_U1;
    SC_HIP(I, hipStreamSynchronize(I->stream));
    return SC_OK;
}

int sc_hip_field_time_sweeps(void *p, int method, int launches, int spl, float omega, float *ms_per_launch)
{
    Instance *I = get(p);
    if (I) field_moved(I);
    if (I && I->f_half) { int frc = float_rhs(I); if (frc) return frc; }
    if (!I || !I->F.p || !ms_per_launch || launches < 1) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    int d = fused_depth(method, spl);
    if (method == SC_METHOD_JACOBI && (d == 5 || d == 7)) d -= 1;   // instantiated depths: 1-4, 6, 8
    const int per = d > 0 ? d : 1;                      // sweeps one "launch group" performs
    I->bench_tag = true;                                // same code under a second symbol (see k_jacobi)
    int rc = run_sweeps(I, method, per, omega, spl);    // warm-up
    if (rc) { I->bench_tag = false; return rc; }
    I->info.sweep_launches = 0;
    SC_HIP(I, hipEventRecord(I->ev_k0, I->stream));
    rc = run_sweeps(I, method, launches * per, omega, spl);
    I->bench_tag = false;
    if (rc) return rc;
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    SC_HIP(I, hipStreamSynchronize(I->stream));
    const int n = I->info.sweep_launches > 0 ? I->info.sweep_launches : 1;
    *ms_per_launch = ev_ms(I->ev_k0, I->ev_k1) / (float)n;
    return SC_OK;
}

// Isolated timing of the level-0 cycle kernel (prolongation + 4 red-black sweeps + residual +
// restriction) on the fields and hierarchy the last MULTIGRID run left on the device.  The values
// it produces are discarded; only the launch duration matters (bench.py roofline).
int sc_hip_time_cycle0(void *p, int launches, float *ms_per_launch)
{
    Instance *I = get(p);
    if (I) field_moved(I);
    if (!I || !ms_per_launch || launches < 1) return SC_ERR_BAD_ARG;
    if (!I->F.p || I->mg.size() < 2 || !I->mg_partial.p) { I->err = "time_cycle0: run a multigrid clone first"; return SC_ERR_BAD_ARG; }
    SC_HIP(I, hipSetDevice(I->gpu));
    const bool comp = mg_composes_level1(I);          // time the form the clone itself runs
    auto once = [&]() {
        if (comp)
            launch_cycle0_composed(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U, I->mg[0].g, 4,
                                   (float *)I->mg_partial.p, I->stream, true, I->f_half, false, I->mg[2].U, I->mg[1].g, nullptr, I->mg_l1_half,
                                   (I->mg_l1_half && I->mg_q16_last) ? 3 : 0);
        else
            launch_cycle0(result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U, I->mg[0].g, 4, true,
                          (float *)I->mg_partial.p, I->stream, true, I->f_half);
        I->result_in_U1 = !I->result_in_U1;
    };
    once();
    SC_HIP(I, hipEventRecord(I->ev_k0, I->stream));
    for (int i = 0; i < launches; ++i) once();
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipStreamSynchronize(I->stream));
    *ms_per_launch = ev_ms(I->ev_k0, I->ev_k1) / (float)launches;
    return SC_OK;
}

int sc_hip_time_cycle0_form(void *p, int form, int launches, float *ms_per_launch)
{
    if (form == 0) return sc_hip_time_cycle0(p, launches, ms_per_launch);
    Instance *I = get(p);
    if (I) field_moved(I);
    if (!I || !ms_per_launch || launches < 1 || form < 1 || form > 3) return SC_ERR_BAD_ARG;
    if (!I->F.p || I->mg.size() < 3 || !I->mg_partial.p || !mg_composes_level1(I) || !I->mg_l1_half || !I->mg_q16_last || !I->f_half) {
        I->err = "time_cycle0_form: run a default multigrid clone first";
        return SC_ERR_BAD_ARG;
    }
    SC_HIP(I, hipSetDevice(I->gpu));
    float4 *bands = form == 1 ? lowmode_bands_buffer(I, 4) : nullptr;
    LmNodes lm;
    if (form == 2 && I->lm.CN.p && !I->lm.singular) { lm.CN = (const float *)I->lm.CN.p; lm.ny = I->lm.ny; lm.npitch = I->lm.npitch; }
    auto once = [&]() {
        // values are discarded: every form reads the fields in the format it expects (whatever bits they hold) and writes the partner
        launch_cycle0_twin(form, result(I), I->result_in_U1 ? I->U0 : I->U1, I->F, I->mg[1].F, I->mg[1].U, I->mg[0].g,
                           (float *)I->mg_partial.p, I->stream, I->mg[2].U, I->mg[1].g, bands, lm);
    };
    once();
    SC_HIP(I, hipEventRecord(I->ev_k0, I->stream));
    for (int i = 0; i < launches; ++i) once();
    SC_HIP(I, hipEventRecord(I->ev_k1, I->stream));
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipStreamSynchronize(I->stream));
    lowmode_bands_written(I, nullptr);
    *ms_per_launch = ev_ms(I->ev_k0, I->ev_k1) / (float)launches;
    return SC_OK;
}

int sc_hip_time_coarse_chain(void *p, int reps, float *ms_eager, float *ms_graph, int *launches)
{
    Instance *I = get(p);
    if (!I || reps < 1 || !ms_eager || !ms_graph || !launches) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    return mg_time_coarse_chain(I, reps, ms_eager, ms_graph, launches);
}

int sc_hip_time_tail_phases(void *p, unsigned long long *cycles11)
{
    Instance *I = get(p);
    if (!I || !cycles11) return SC_ERR_BAD_ARG;
    SC_HIP(I, hipSetDevice(I->gpu));
    return mg_time_tail_phases(I, cycles11);
}

} // extern "C"
