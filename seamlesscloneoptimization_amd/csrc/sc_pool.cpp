// sc_pool.cpp -- native batch driver: K library instances (= K HIP streams) on one GPU, each driven
// by its own host thread, pulling independent clone jobs from a shared counter.
//
// The reference has no batch or multi-stream mode (one instance, one stream, seamlessClone_imp.cu:
// 239-263).  Clones are independent and a single clone leaves the GPU idle in its latency-bound
// phases (coarse multigrid levels, the bounding-box read-back), so several in flight raise
// throughput (2048^2 ROI: 5.3 -> 9 Gpix/s with four).  This is the C++ equivalent of
// seamlesscloneoptimization_amd/batch.py:StreamPool, behind the same C ABI.
#include "sc_instance.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace sc;

namespace {

struct Pool {
    uint32_t magic = 0x5C10E002u;
    int gpu = 0;
    std::vector<void *> inst;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    // current batch
    sc_batch_job *jobs = nullptr;
    int njobs = 0, device_resident = 0;
    int group = 1;                 // device-resident jobs a worker takes at a time (sc_hip_run_device_batch)
    // group > 1: the batch's jobs gathered into groups that can share launches (`sorted`: a copy in group order, `origin[k]` = its index
    // in the caller's array), one chunk of at most `group` members each; `next` counts chunks then, jobs otherwise
    std::vector<sc_batch_job> sorted;
    std::vector<int> origin;
    std::vector<std::pair<int, int>> chunks;       // first member, members
    std::atomic<int> next{ 0 };
    int generation = 0, finished_workers = 0;
    bool stop = false;
};

void run_job(void *inst, sc_batch_job &j, int device_resident)
{
    if (device_resident) {
        if (j.body_restore) {
            const int rc = sc_hip_memcpy_d2d_async(inst, j.body, j.body_restore, (size_t)j.body_step * j.body_rows);
            if (rc != SC_OK) { j.rc = rc; return; }
        }
        j.rc = sc_hip_run_device(inst, j.face, j.face_cols, j.face_rows, j.face_step, j.body, j.body_cols, j.body_rows,
                                 j.body_step, j.mask, j.mask_cols, j.mask_rows, j.mask_step, j.centerX, j.centerY, false);
    } else {
        j.rc = my_seamlessclone_api_imp_run(inst, j.face, j.face_cols, j.face_rows, j.face_step, j.body, j.body_cols,
                                            j.body_rows, j.body_step, j.mask, j.mask_cols, j.mask_rows, j.mask_step,
                                            j.centerX, j.centerY, 0, false);     // bSync = false: no timing lines on stdout (the call is synchronous either way)
    }
}

void worker(Pool *P, int k)
{
    int seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(P->mu);
            P->cv_work.wait(lk, [&] { return P->stop || P->generation != seen; });
            if (P->stop) return;
            seen = P->generation;
        }
        const bool chunked = !P->chunks.empty();
        for (;;) {
            const int i = P->next.fetch_add(1);
            if (chunked) {
                if (i >= (int)P->chunks.size()) break;
                const int first = P->chunks[i].first, cnt = P->chunks[i].second;
                sc_batch_job *js = P->sorted.data() + first;
                if (cnt > 1) {
                    // same-size clones of the chunk share one set of launches; per-job codes are filled in by the call
                    constexpr int unset = -2147483647;
                    for (int q = 0; q < cnt; ++q) js[q].rc = unset;
                    const int rc = sc_hip_run_device_batch(P->inst[k], js, cnt);
                    for (int q = 0; q < cnt; ++q)          // the call failed before it got to this member
                        if (js[q].rc == unset) js[q].rc = (rc != SC_OK) ? rc : SC_ERR_HIP;
                } else {
                    run_job(P->inst[k], js[0], P->device_resident);
                }
                for (int q = 0; q < cnt; ++q) P->jobs[P->origin[first + q]].rc = js[q].rc;
            } else {
                if (i >= P->njobs) break;
                run_job(P->inst[k], P->jobs[i], P->device_resident);
            }
        }
        my_seamlessclone_api_imp_sync(P->inst[k]);          // the batch is complete when run() returns
        {
            std::lock_guard<std::mutex> lk(P->mu);
            if (++P->finished_workers == (int)P->workers.size()) P->cv_done.notify_all();
        }
    }
}

Pool *get_pool(void *p)
{
    Pool *P = (Pool *)p;
    return (P && P->magic == 0x5C10E002u) ? P : nullptr;
}

} // namespace

extern "C" {

void *sc_hip_pool_create(int gpu_id, int streams)
{
    if (streams < 1) streams = 1;
    if (streams > 16) streams = 16;
    Pool *P = new (std::nothrow) Pool();
    if (!P) return nullptr;
    P->gpu = gpu_id;
    for (int k = 0; k < streams; ++k) {
        void *inst = my_seamlessclone_api_imp_create_instance(gpu_id);
        if (!inst) {
            for (void *i : P->inst) my_seamlessclone_api_imp_destroy(i);
            delete P;
            return nullptr;
        }
        P->inst.push_back(inst);
    }
    for (int k = 0; k < streams; ++k) P->workers.emplace_back(worker, P, k);
    return P;
}

void sc_hip_pool_destroy(void *p)
{
    Pool *P = get_pool(p);
    if (!P) return;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->stop = true;
    }
    P->cv_work.notify_all();
    for (std::thread &t : P->workers) t.join();
    for (void *i : P->inst) my_seamlessclone_api_imp_destroy(i);
    P->magic = 0;
    delete P;
}

int sc_hip_pool_size(void *p)
{
    Pool *P = get_pool(p);
    return P ? (int)P->inst.size() : 0;
}

void *sc_hip_pool_instance(void *p, int k)
{
    Pool *P = get_pool(p);
    if (!P || k < 0 || k >= (int)P->inst.size()) return nullptr;
    return P->inst[k];
}

int sc_hip_pool_set_solver(void *p, const sc_solver_opts *opts)
{
    Pool *P = get_pool(p);
    if (!P || !opts) return SC_ERR_BAD_ARG;
    // host jobs of one batch may share a destination (disjoint ROIs of one image): the whole-row return would let one job erase
    // another's result, so the pool never takes it
    sc_solver_opts o = *opts;
    o.flags &= ~SC_FLAG_ROWS_RETURN;
    opts = &o;
    for (void *i : P->inst) {
        const int rc = sc_hip_set_solver(i, opts);
        if (rc != SC_OK) return rc;
    }
    return SC_OK;
}

// Runs all jobs (any order, each exactly once) and returns when every one has completed on the GPU.
// Return value: SC_OK, or the first failing job's code (each job's own code is in jobs[i].rc).
int sc_hip_pool_set_group(void *p, int group)
{
    Pool *P = get_pool(p);
    if (!P || group < 0 || group > 64) return SC_ERR_BAD_ARG;      // 0: SC_POOL_GROUP_AUTO
    std::lock_guard<std::mutex> lk(P->mu);
    P->group = group;
    return SC_OK;
}

int sc_hip_pool_run(void *p, sc_batch_job *jobs, int n, int device_resident)
{
    Pool *P = get_pool(p);
    if (!P || (n > 0 && !jobs) || n < 0) return SC_ERR_BAD_ARG;
    if (n == 0) return SC_OK;
    {
        std::lock_guard<std::mutex> lk(P->mu);
        P->jobs = jobs; P->njobs = n; P->device_resident = device_resident;
        P->chunks.clear();
        if (device_resident && P->group != 1) {
            // Form the groups a worker takes at a time from the ROI size every job will most likely have -- the interior of its mask,
            // which is what a clone is launched on before the device's bounding box is back: same-size jobs and jobs of one size class
            // (sc_ragged.cpp: different sizes, the same solve) share one set of launches inside sc_hip_run_device_batch.  Round 4 took
            // CONSECUTIVE jobs and ran a group with one odd size one clone at a time.  Largest groups first: the stragglers at the end
            // of a batch are the cheap ones.
            int cap, cap_max;
            long budget;
            pool_group_caps(P->group, n, (int)P->workers.size(), cap, cap_max, budget);
            bool one_size = true;
            for (int i = 1; i < n && one_size; ++i) one_size = jobs[i].mask_cols == jobs[0].mask_cols && jobs[i].mask_rows == jobs[0].mask_rows;
            if (one_size) {
                // every job has the same mask size (a benchmark's batch, a tiled image): consecutive chunks, nothing to plan
                const long area = (long)std::max(1, jobs[0].mask_cols) * std::max(1, jobs[0].mask_rows);
                int per = cap;
                if (cap_max > cap && budget > 0) per = (int)std::max<long>(cap, std::min<long>(cap_max, budget / area));
                P->origin.clear();
                P->sorted.clear();
                for (int i0 = 0; i0 < n; i0 += per) {
                    const int cnt = std::min(per, n - i0);
                    P->chunks.emplace_back(i0, cnt);
                    for (int i = i0; i < i0 + cnt; ++i) { P->origin.push_back(i); P->sorted.push_back(jobs[i]); }
                }
            } else {
            sc_solver_opts o;
            sc_hip_get_solver(P->inst[0], &o);
            // (the planner is first-come; the pool is free to reorder, so it hands the jobs over largest first: the members of a group
            //  then differ as little as the batch allows, and a class's grids -- sized for its largest member -- waste the least)
            std::vector<int> order(n);
            for (int i = 0; i < n; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                return (long)jobs[x].mask_cols * jobs[x].mask_rows > (long)jobs[y].mask_cols * jobs[y].mask_rows;
            });
            std::vector<SizePlan> plans(n);
            for (int i = 0; i < n; ++i) plan_size(o, jobs[order[i]].mask_cols - 2, jobs[order[i]].mask_rows - 2, plans[i]);
            std::vector<std::vector<int>> groups;
            plan_groups(plans, cap, groups, cap_max, budget);
            {
                std::vector<SizePlan> by_job(n);
                for (int i = 0; i < n; ++i) by_job[order[i]] = plans[i];
                plans.swap(by_job);
                for (auto &g : groups) for (int &i : g) i = order[i];
            }
            std::stable_sort(groups.begin(), groups.end(), [&](const std::vector<int> &x, const std::vector<int> &y) {
                auto px = [&](const std::vector<int> &g) { long t = 0; for (int i : g) t += (long)plans[i].W * plans[i].H; return t; };
                return px(x) > px(y);
            });
            P->origin.clear();
            P->sorted.clear();
            for (const auto &g : groups) {
                P->chunks.emplace_back((int)P->sorted.size(), (int)g.size());
                for (int i : g) { P->origin.push_back(i); P->sorted.push_back(jobs[i]); }
            }
            }
        }
        P->next.store(0);
        P->finished_workers = 0;
        ++P->generation;
    }
    P->cv_work.notify_all();
    {
        std::unique_lock<std::mutex> lk(P->mu);
        P->cv_done.wait(lk, [&] { return P->finished_workers == (int)P->workers.size(); });
    }
    for (int i = 0; i < n; ++i)
        if (jobs[i].rc != SC_OK && jobs[i].rc != SC_ERR_NOT_CONVERGED) return jobs[i].rc;
    return SC_OK;
}

} // extern "C"
