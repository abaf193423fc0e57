// sc_ragged.cpp -- size classes: clones of DIFFERENT ROI sizes through one set of solver launches (round 5).
//
// The reference has no batch mode at all (one clone per call, seamlessClone_imp.cu:239-263); its use case -- a mask box per face,
// per frame -- produces batches in which no two ROIs have the same size.  sc_hip_run_device_batch (sc_api.cpp) solves n same-size
// clones as one field of 3n channels; a SIZE CLASS extends that to members whose sizes differ but whose solves are the same
// program: the same hierarchy depth, the same level held by k_mg_tail with the level below it solved directly at the same operand
// padding, the same mode-block counts of the float-table correction.  Such members share strides (the class's largest width and
// height on every level) and launch grids; everything else is per member, read by the kernels from a table (RagMember,
// sc_common.h).  This file is the host side of that table: the planner (what a size needs, which sizes go together) and the
// per-call setup (tables on the device).  Host arithmetic and copies only; the kernels are where they always were.
#include "sc_instance.h"
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace sc {

// ROI sizes of one class may differ by this factor per direction at most (strides are the largest member's: what a smaller
// member leaves unused of its planes is never touched, but the grids are sized for the largest)
static constexpr double RAG_SPREAD = 2.0;
static constexpr int RAG_SPREAD_PIXELS = 64;
static constexpr long RAG_SPREAD_AREA = 100000;

// Plans are pure functions of (W, H) and of the few solver options below: memoised, so that a caller whose ROI sizes recur (video:
// the same faces frame after frame) plans each size once.  Bounded; shared by every instance and pool of the process.
namespace {
struct PlanKey {
    int W, H, method, flags, pre, post, spl, warm, dmax, l1s, maxs; float tol, utol;
    bool operator==(const PlanKey &o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct PlanKeyHash {
    size_t operator()(const PlanKey &k) const
    {
        size_t h = 1469598103934665603ull;
        const unsigned char *p = reinterpret_cast<const unsigned char *>(&k);
        for (size_t i = 0; i < sizeof(k); ++i) h = (h ^ p[i]) * 1099511628211ull;
        return h;
    }
};
std::mutex g_plan_mu;
std::unordered_map<PlanKey, SizePlan, PlanKeyHash> g_plan_cache;
constexpr size_t PLAN_CACHE_MAX = 2048;
bool plan_size_uncached(const sc_solver_opts &o, int W, int H, SizePlan &p);
} // namespace

bool plan_size(const sc_solver_opts &o, int W, int H, SizePlan &p)
{
    PlanKey k;
    memset(&k, 0, sizeof(k));
    k.W = W; k.H = H; k.method = o.method; k.flags = o.flags; k.pre = o.mg_pre; k.post = o.mg_post; k.spl = o.sweeps_per_launch;
    k.warm = o.reference_warmup; k.dmax = o.mg_direct_max; k.l1s = o.mg_level1_sweeps; k.maxs = o.max_sweeps; k.tol = o.tol; k.utol = o.update_tol;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_plan_cache.find(k);
        if (it != g_plan_cache.end()) { p = it->second; return p.ok; }
    }
    plan_size_uncached(o, W, H, p);
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (g_plan_cache.size() >= PLAN_CACHE_MAX) g_plan_cache.clear();
        g_plan_cache.emplace(k, p);
    }
    return p.ok;
}

namespace {
bool plan_size_uncached(const sc_solver_opts &o, int W, int H, SizePlan &p)
{
    p = SizePlan();
    p.W = W; p.H = H;
    auto T = std::make_shared<SizePlan::Tables>();
    std::vector<MGGeom> &pg = T->g;
    // --- the solve configured in `o` is the default fast path: float16 right-hand side and level 1, composed level-1 schedule,
    //     output bytes from the judged cycle (sc_multigrid.cpp: mg_reads_half_rhs, mg_level1_half, mg_composes_level1)
    const int pre = o.mg_pre > 0 ? o.mg_pre : 2, post = o.mg_post > 0 ? o.mg_post : 2;
    const int blocked = SC_FLAG_FLOAT_RHS | SC_FLAG_FLOAT_U0 | SC_FLAG_NO_COMPOSE_L1 | SC_FLAG_VCYCLE_BOTTOM | SC_FLAG_EXACT_TABLES |
                        SC_FLAG_LEGACY_PATHS | SC_FLAG_KEEP_FIELD | SC_FLAG_OPENCV_GREY_MASK | SC_FLAG_FLOAT_L1;
    if ((o.method != SC_METHOD_AUTO && o.method != SC_METHOD_MULTIGRID) || o.tol > 0.f || (o.flags & blocked) || pre != 2 || post != 2 ||
        o.sweeps_per_launch != 0 || o.reference_warmup || o.mg_direct_max != 0 || (o.mg_level1_sweeps != 0 && o.mg_level1_sweeps != 4) ||
        (o.max_sweeps > 0 && o.max_sweeps < 3))
        return false;
    if (std::min(W, H) - 2 <= 3) return false;
    // --- its hierarchy ends the default way: a level of at most 127 unknowns per side in k_mg_tail, the one below it solved there
    mg_plan_levels(W, H, pg);
    p.nl = (int)pg.size();
    size_t a = mg_default_tail_level(pg);
    if (!a) {
        // A ladder whose level 1 a solo clone (or a same-size group) solves directly -- at most 64 unknowns per side: ROIs up to ~130
        // pixels (the reference's own 154 x 100 patch runs the general hierarchy alone as well since late in round 5).  A class has no such form; it takes the general one -- level 2 in
        // k_mg_tail, level 3 solved directly -- where that exists.  Same fixed point, different iterates: such a member is within one
        // grey level of its solo run, not byte-identical to it (`solo_differs`; every larger size is).
        if (std::min(W, H) >= 48)                            // (below that the directly solved level has a handful of unknowns per side: such clones stay alone)
            for (size_t l = 2; l + 1 < pg.size() && !a; ++l)
                if (pg[l].x.n <= 127 && pg[l].y.n <= 127) a = l;
        p.solo_differs = a != 0;
    }
    if (!a || (int)a + 2 > RAG_MAX_LEVELS || p.nl < (int)a + 2) return false;
    p.tail = (int)a;
    const MGGeom &A = pg[a];
    if (A.x.n > 127 || A.y.n > 127 || A.x.nc > 63 || A.y.nc > 63 || A.x.nc < 1 || A.y.nc < 1) return false;
    p.npx = round_up(A.x.nc, 32); p.npy = round_up(A.y.nc, 32);
    // --- its float-table correction: regular tables, the node correction of the iterate one cycle earlier admissible a priori
    //     (lowmode_early_kind 1), few enough parts for the expansion to add them itself
    const int w = W - 2, h = H - 2;
    p.Kx = lowmode_count(w); p.Ky = lowmode_count(h);
    p.Kxp = round_up(p.Kx, 32); p.Kyp = round_up(p.Ky, 32);
    p.nx = (w >> 3) + 2; p.ny = (h >> 3) + 2;
    p.cells_y = (H + 7) / 8;
    p.nxt = (p.nx + 63) / 64;
    p.nrs = lowmode_projection_splits(p.nxt, (p.Kyp + 31) / 32);
    if (p.nxt * p.nrs >= 32) return false;
    const bool regular = lowmode_ratio(w, h, p.Kx, p.Ky, p.Kxp, nullptr, p.max_ratio);
    const float utol = o.update_tol > 0.f ? o.update_tol : 0.25f;
    if (!regular) return false;
    // sizes whose float tables are off by more than 4 % in their lowest modes (one in ten around 2100 a side, most beyond 3000): the
    // output may carry the earlier iterate's correction only if the judged cycle's MEASURED update allows it (lowmode_early_kind 3,
    // decided beside the stop rule in mg_solve with the class's largest ratio and the group's largest update).  Classes of their own:
    // a member whose bound holds a priori never shares that decision -- and such a member is byte-identical to its solo run when the
    // decision falls as its solo run's does, within one grey level of it otherwise (`conditional`).
    p.conditional = !(p.max_ratio * 4.9 * (double)utol <= 0.049);
    p.t = T;
    p.ok = true;
    return true;
}

} // namespace

// A member's plan one level deeper than its own: the level its solo run holds in k_mg_tail becomes an ordinary coarse level, the one
// its solo run solves directly goes into k_mg_tail and the next one is solved directly.  Same fixed point, other iterates: within
// one grey level of the solo run, not its bytes (solo_differs).  false: the ladder has no such level.
static bool plan_deeper(const SizePlan &p, SizePlan &q)
{
    if (!p.ok || !p.t) return false;
    const int a = p.tail + 1;
    if (a + 2 > RAG_MAX_LEVELS || p.nl < a + 2) return false;
    const MGGeom &A = p.t->g[a];
    if (A.x.n > 127 || A.y.n > 127 || A.x.nc > 63 || A.y.nc > 63 || A.x.nc < 1 || A.y.nc < 1) return false;
    q = p;
    q.tail = a;
    q.npx = round_up(A.x.nc, 32); q.npy = round_up(A.y.nc, 32);
    q.solo_differs = true;
    return true;
}

void plan_cache_clear()
{
    { std::lock_guard<std::mutex> lk(g_plan_mu); g_plan_cache.clear(); }
}

// Greedy, order preserving: a member joins the first open group it fits -- the same size as the group's members, or the same class
// with the group's spread staying within bounds -- else it opens a new one.  Then the LEFTOVERS of a class (a group of at most cap / 2
// members) whose hierarchy is one level shallower than that of a group with room move there, on the deeper hierarchy (plan_deeper:
// their plans are rewritten in `plans`): sizes straddling 513 / 1027 / 2050 unknowns per side -- where the ladder gains a level --
// would otherwise always split in two.
// What the pool hands plan_groups for a batch of n jobs on `streams` workers (sc_pool.cpp; sc_hip_plan_groups_pool mirrors it):
// an explicit group size is a hard cap; SC_POOL_GROUP_AUTO (0) means sixteen members at least where the batch has them, more -- up to
// n / min(streams, 2), so that two streams get a group each, and up to 64 -- while a group's fields stay within what sixteen 2048^2 members
// occupy.  Measured (64 jobs, 2 streams; groups of 16 -> 32): [120, 190]^2 0.64 -> 0.43 ms per batch, [300, 340]^2 0.91 -> 0.80,
// [500, 560]^2 1.73 -> 1.54, [1000, 1100]^2 4.45 -> 4.24; ONE group of 64 loses (the second stream idles).
void pool_group_caps(int group, int n, int streams, int &cap, int &cap_max, long &budget_px)
{
    if (group > 0) { cap = cap_max = group; budget_px = 0; return; }
    // (TWO groups at a time are what pays: 64 clones of 120..190 pixels take 0.42 ms as 2 x 32 on two streams, 0.48 as 3 groups on three,
    //  0.61 as 4 x 16 on four, 0.99 as 8 x 8 on eight -- more host threads launching small kernels at once only contend; 1000..1100:
    //  4.05 ms as 2 x 32, 4.65 as 4 x 16 -- so a pool with more than two streams still gets half the batch per group)
    const int lanes = std::max(1, std::min(streams, 2));
    const int per_stream = (n + lanes - 1) / lanes;
    cap = std::max(1, std::min(16, per_stream));             // (a batch smaller than 16 x streams is split among the streams as well: 8 x 1024^2 on two: 2 x 4 beats 1 x 8)
    cap_max = std::max(cap, std::min(64, per_stream));
    budget_px = 16L * 2048 * 2048;
}

// cap_max > cap: a group may grow beyond `cap` members while its fields stay within `budget_px` pixels per channel triple (members x
// the group's largest width x height), up to cap_max -- small ROIs are latency bound and sixteen of them do not fill a launch.
void plan_groups(std::vector<SizePlan> &plans, int cap, std::vector<std::vector<int>> &groups, int cap_max, long budget_px)
{
    struct Open { int first; int minW, maxW, minH, maxH; bool uniform; };
    std::vector<Open> open;
    groups.clear();
    // (2x per direction, or 64 pixels where that is more -- a level-0 tile is 232 x 44, small ROIs differ by less than the grid's
    //  grain --, or 100 000 pixels of area: a set of launches costs what ~3 Mpixels of level-0 work cost, whatever its size)
    auto fits = [](const Open &g, const SizePlan &p, Open &w) {
        w = g;
        w.minW = std::min(g.minW, p.W); w.maxW = std::max(g.maxW, p.W); w.minH = std::min(g.minH, p.H); w.maxH = std::max(g.maxH, p.H);
        const bool wx = (double)w.maxW <= RAG_SPREAD * w.minW || w.maxW - w.minW <= RAG_SPREAD_PIXELS;
        const bool wy = (double)w.maxH <= RAG_SPREAD * w.minH || w.maxH - w.minH <= RAG_SPREAD_PIXELS;
        return (wx && wy) || (long)w.maxW * w.maxH - (long)w.minW * w.minH <= RAG_SPREAD_AREA;
    };
    auto cap_of = [&](const Open &w) {
        if (cap_max <= cap || budget_px <= 0) return cap;
        const long by_area = budget_px / std::max(1L, (long)w.maxW * w.maxH);
        return (int)std::max<long>(cap, std::min<long>(cap_max, by_area));
    };
    for (int i = 0; i < (int)plans.size(); ++i) {
        const SizePlan &p = plans[i];
        int into = -1;
        Open w{};
        for (int k = 0; k < (int)open.size() && into < 0; ++k) {
            const Open &g = open[k];
            const SizePlan &q = plans[g.first];
            if (g.uniform && q.W == p.W && q.H == p.H) {
                if ((int)groups[k].size() >= cap_of(g)) continue;
                into = k; w = g; break;
            }
            if (!p.same_class(q)) continue;
            if (fits(g, p, w) && (int)groups[k].size() < cap_of(w)) into = k;
        }
        if (into < 0) {
            open.push_back({ i, p.W, p.W, p.H, p.H, true });
            groups.push_back(std::vector<int>(1, i));
            continue;
        }
        if (plans[open[into].first].W != p.W || plans[open[into].first].H != p.H) w.uniform = false;
        open[into] = w;
        groups[into].push_back(i);
    }
    // --- leftovers onto the deeper hierarchy of a group with room
    const int small = std::max(1, cap / 2);
    std::vector<char> received(groups.size(), 0);          // (a group that took leftovers in stays where it is)
    for (int k = 0; k < (int)groups.size(); ++k) {
        if (groups[k].empty() || (int)groups[k].size() > small || !plans[groups[k][0]].ok || received[k]) continue;
        std::vector<int> stay;
        for (int i : groups[k]) {
            SizePlan d;
            bool moved = false;
            if (plan_deeper(plans[i], d))
                for (int t = 0; t < (int)groups.size() && !moved; ++t) {
                    if (t == k || groups[t].empty()) continue;
                    const SizePlan &q = plans[open[t].first];
                    Open w{};
                    if (!d.same_class(q) || q.tail != plans[i].tail + 1 || !fits(open[t], d, w) || (int)groups[t].size() >= cap_of(w)) continue;
                    w.uniform = false;
                    open[t] = w;
                    groups[t].push_back(i);
                    plans[i] = d;
                    received[t] = 1;
                    moved = true;
                }
            if (!moved) stay.push_back(i);
        }
        groups[k].swap(stay);
    }
    size_t o = 0;
    for (size_t k = 0; k < groups.size(); ++k)
        if (!groups[k].empty()) { if (o != k) groups[o] = std::move(groups[k]); ++o; }
    groups.resize(o);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The members' table and everything it points to, on the device:
//   h_stage (pinned) = RagMember[n] | R tables | part maps   -> ONE upload into the head of d_aux, on the instance's stream
//   d_aux            = table | R tables | part maps | Sx | Sy tables | bottom operands
// and, on the instance's SECOND stream from here on -- beside the scans, the erodes, the pre-process and the first level-0 launch of
// the solve --, everything the device builds per call: the Sx / Sy tables, the members' bottom matrices, the zeroing of the coarse
// planes (mg_build_levels_rag).  The main stream waits for all of it in front of its first coarse-level launch (mg_solve).
// Sets I->rag.dev; the caller has bound the class's fields (setup_fields) before.
int rag_begin_table(Instance *I, const std::vector<SizePlan> &members)
{
    RagState &R = I->rag;
    const int n = (int)members.size();
    R.dev = nullptr;
    if (n < 1) return SC_ERR_BAD_ARG;
    const SizePlan &p0 = members[0];
    R.n = n; R.nl = p0.tail + 2; R.tail = p0.tail;      // levels 0 .. tail + 1: the one solved directly is the last one anybody visits
    R.npx = p0.npx; R.npy = p0.npy; R.Kxp = p0.Kxp; R.Kyp = p0.Kyp;      // (npx / npy: the class's largest, below)
    R.max_nx = R.max_ny = R.max_nxt = R.max_nrs = R.max_cells_y = 0;
    R.max_ratio = 0.0;
    for (const SizePlan &p : members) {
        if (!p.same_class(p0)) { I->err = "size class: members of different classes"; return SC_ERR_BAD_ARG; }
        R.max_nx = std::max(R.max_nx, p.nx); R.max_ny = std::max(R.max_ny, p.ny);
        R.max_nxt = std::max(R.max_nxt, p.nxt); R.max_nrs = std::max(R.max_nrs, p.nrs);
        R.max_cells_y = std::max(R.max_cells_y, p.cells_y);
        R.max_ratio = std::max(R.max_ratio, p.max_ratio);
        R.npx = std::max(R.npx, p.npx); R.npy = std::max(R.npy, p.npy);
        // the correction's tables take the class's largest mode-block padding as their row pitch: what a member does not have is zero
        // (k_lm_table_rag, the ratio table below), and a zero mode adds exact zeros to every sum -- same bytes as at its own padding
        R.Kxp = std::max(R.Kxp, p.Kxp); R.Kyp = std::max(R.Kyp, p.Kyp);
    }
    R.pad_uniform = true;
    for (const SizePlan &p : members) R.pad_uniform = R.pad_uniform && p.npx == R.npx && p.npy == R.npy;
    // --- layout of d_aux (256-byte aligned pieces; the per-member pieces one member after the other)
    const size_t bR = align_up(sizeof(float) * (size_t)R.Kyp * R.Kxp, 256), bMap = align_up(sizeof(int) * 4 * (size_t)R.max_cells_y, 256);
    const size_t bSx = align_up(sizeof(float) * (size_t)R.max_nx * R.Kxp, 256), bSy = align_up(sizeof(float) * (size_t)R.max_ny * R.Kyp, 256);
    const size_t bMM = align_up((size_t)fd_mm_bytes(R.npx, R.npy), 256);
    const size_t table_bytes = align_up(sizeof(RagMember) * (size_t)n, 256);
    const size_t host_part = table_bytes + (bR + 2 * bMap) * n;        // what the host writes: one upload
    const size_t aux_bytes = host_part + (bSx + bSy + bMM) * n;
    int rc;
    if (I->fd_pending) { SC_HIP(I, hipEventSynchronize(I->ev_fd)); I->fd_pending = false; }      // second-stream work nobody joined (a call that failed early): before its buffers may move
    R.ready_pending = false;
    if ((rc = ensure(I, R.d_aux, aux_bytes))) return rc;
    if (!R.ev) SC_HIP(I, hipEventCreateWithFlags(&R.ev, hipEventDisableTiming));
    else SC_HIP(I, hipEventSynchronize(R.ev));                        // the previous upload out of the staging: long complete
    if ((rc = ensure_pinned(I, R.h_stage, host_part))) return rc;
    uint8_t *const hs = (uint8_t *)R.h_stage.p, *const da = (uint8_t *)R.d_aux.p;
    R.host.assign(n, RagMember());
    for (int i = 0; i < n; ++i) {
        const SizePlan &p = members[i];
        RagMember &m = R.host[i];
        std::memset(&m, 0, sizeof(m));
        m.W = p.W; m.H = p.H;
        const SizePlan::Tables &T = *p.t;
        for (int l = 0; l < R.nl; ++l) { m.g[l] = T.g[l]; m.lw[l] = T.g[l].x.n + 2; m.lh[l] = T.g[l].y.n + 2; }
        for (int l = R.nl; l < RAG_MAX_LEVELS; ++l) { m.lw[l] = 3; m.lh[l] = 3; }
        m.lm_nx = p.nx; m.lm_ny = p.ny; m.lm_cells_y = p.cells_y; m.lm_nxt = p.nxt; m.lm_nrs = p.nrs; m.lm_nparts = p.nxt * p.nrs;
        m.lm_Kx = p.Kx; m.lm_Ky = p.Ky;
        const size_t oR = table_bytes + (bR + 2 * bMap) * i, oMap0 = oR + bR, oMap1 = oMap0 + bMap;
        m.lm_R = (const float *)(da + oR);
        m.lm_map[0] = (const int *)(da + oMap0); m.lm_map[1] = (const int *)(da + oMap1);
        m.lm_Sx = (const float *)(da + host_part + (bSx + bSy + bMM) * i);
        m.lm_Sy = (const float *)(da + host_part + (bSx + bSy + bMM) * i + bSx);
        m.mm = da + host_part + (bSx + bSy + bMM) * i + bSx + bSy;
        m.npx = p.npx; m.npy = p.npy;
        // the host-built pieces, straight into the staging: the correction's ratio table (the reference's float expressions through libm,
        // as the CPU oracle's; rows at the class's pitch, what the member does not have is zero) and the two part maps.  Functions of
        // (W, H) alone, but NOT memoised: 0.5 / 1.5 / 4.5 us per member at 320^2 / 1050^2 / 2100^2, while a memo's fresh 24 KB per new size
        // cost 20 us in first-touch page faults (rounds of never-repeating sizes: +11 % per step at [1000, 1100]^2) -- and this runs
        // while the device erodes and pre-processes
        double mr;
        std::memset(hs + oR, 0, bR);
        lowmode_ratio(p.W - 2, p.H - 2, p.Kx, p.Ky, R.Kxp, (float *)(hs + oR), mr);
        int br = 0;
        std::memset(hs + oMap0, 0xff, 2 * bMap);
        if (!lowmode_part_map(p.H, 2, (int *)(hs + oMap0), br) || !lowmode_part_map(p.H, 4, (int *)(hs + oMap1), br)) {
            I->err = "size class: more than four parts per cell row"; return SC_ERR_BAD_ARG;
        }
    }
    std::memcpy(hs, R.host.data(), sizeof(RagMember) * (size_t)n);
    SC_HIP(I, hipMemcpyAsync(da, hs, host_part, hipMemcpyHostToDevice, I->stream));
    SC_HIP(I, hipEventRecord(R.ev, I->stream));
    R.dev = (const RagMember *)da;
    I->info.new_size = 1;
    return SC_OK;
}

// ... second half: five launches and four events on two more streams (12 us of host time).  The caller enqueues the class's erode and
// pre-process between the halves, so that the device has work while the host is here.
int rag_begin_builds(Instance *I)
{
    RagState &R = I->rag;
    if (!R.dev) return SC_ERR_BAD_ARG;
    const int n = R.n;
    int rc;
    // --- the device-built pieces, on the second stream behind the upload and behind everything that read the previous call's
    //     tables, matrices and level planes
    // (the short ones first, with an event of their own: the first coarse-level launch needs the zeroed planes, only the first
    // k_mg_tail the matrices -- whose build is ~100 us of dependent double-precision arithmetic)
    SC_HIP(I, hipStreamWaitEvent(I->aux, R.ev, 0));
    if ((rc = mg_build_levels_rag(I, I->aux))) { R.dev = nullptr; return rc; }
    launch_lm_tables_rag(R.dev, n, std::max(R.max_nx, R.max_ny), R.Kxp, R.Kyp, I->aux);
    if (!R.ev_ready) SC_HIP(I, hipEventCreateWithFlags(&R.ev_ready, hipEventDisableTiming));
    SC_HIP(I, hipEventRecord(R.ev_ready, I->aux));
    R.ready_pending = true;
    // (a stream of their own: behind the zeroing and the sine tables they started 13 us later and the first k_mg_tail of a class of
    //  small ROIs waited that long for them)
    if (!I->aux2) SC_HIP(I, hipStreamCreateWithFlags(&I->aux2, hipStreamNonBlocking));
    SC_HIP(I, hipStreamWaitEvent(I->aux2, R.ev, 0));
    launch_fd_build_rag(R.dev, n, R.tail + 1, I->aux2);
    SC_HIP(I, hipGetLastError());
    SC_HIP(I, hipEventRecord(I->ev_fd, I->aux2));
    I->fd_pending = true;
    return SC_OK;
}

// the class is done (or failed): the next solve on this instance is an ordinary one and builds its own hierarchy and tables
void rag_end(Instance *I)
{
    if (!I->rag.dev) return;
    I->rag.dev = nullptr;
    I->rag.levels_built = false;
    I->mg.clear();
    I->lm.w = I->lm.h = 0;
    field_moved(I);
}

} // namespace sc
