// sanitize_main.cpp -- host-side sanitizer run of libseamlessclone_hip's C++ (make sanitize; tests/test_host.py).
//
// GPU AddressSanitizer is not available on the pool this library is developed on, the host side is: 3 000 lines of C++ with a
// parked-thread row copier (sc_hostcopy.cpp), pool workers pulling chunks from a shared counter (sc_pool.cpp), memoised size plans
// behind mutexes (sc_ragged.cpp), LRU bookkeeping and level geometry.  This program is the library's host code -- every source,
// compiled for the host only, with -fsanitize=address,undefined or -fsanitize=thread -- driven through the entry points that need
// no GPU:
//   1. sc_hip_selftest_host: the row copier across its helper threads, both eigen-solvers, the part maps;
//   2. the planner from several threads at once (plan cache hits, misses and evictions), against a single-threaded reference;
//   3. the pool's hand-out of jobs against stub instances (sc_pool.cpp compiled with its instance calls renamed to the stubs
//      below): groups formed, every job run exactly once, per-job codes copied back, several batches on one pool.
// Exit code 0 = clean (a sanitizer report aborts with its own).
#include "../../include/seamlessclone_hip.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// ---- stub instances for the pool (sc_pool.cpp is compiled with -Dmy_seamlessclone_api_imp_create_instance=stub_create ...)
namespace {
struct StubInst { int gpu; std::atomic<long> calls{ 0 }; sc_solver_opts opts; };
std::atomic<long> g_jobs_run{ 0 };
}
extern "C" {
void *stub_create(int gpu) { StubInst *s = new StubInst(); s->gpu = gpu; sc_hip_default_opts(&s->opts); return s; }
void stub_destroy(void *p) { delete (StubInst *)p; }
void stub_sync(void *) {}
int stub_set_solver(void *p, const sc_solver_opts *o) { ((StubInst *)p)->opts = *o; return SC_OK; }
int stub_get_solver(void *p, sc_solver_opts *o) { *o = ((StubInst *)p)->opts; return SC_OK; }
int stub_memcpy_d2d_async(void *, void *, const void *, size_t) { return SC_OK; }
int stub_run_device(void *p, const uint8_t *, int, int, int, uint8_t *body, int, int, int, const uint8_t *, int, int, int, int, int, bool)
{
    ((StubInst *)p)->calls++;
    g_jobs_run++;
    *(volatile uint8_t *)body += 1;          // every job owns its body byte: a job handed out twice shows as 2 (and as a race under TSan)
    return SC_OK;
}
int stub_run(void *p, const uint8_t *f, int fc, int fr, int fs, uint8_t *body, int bc, int br, int bs, const uint8_t *m, int mc, int mr, int ms, int cx, int cy, int, bool)
{
    return stub_run_device(p, f, fc, fr, fs, body, bc, br, bs, m, mc, mr, ms, cx, cy, false);
}
int stub_run_device_batch(void *p, sc_batch_job *jobs, int n)
{
    ((StubInst *)p)->calls++;
    for (int i = 0; i < n; ++i) { g_jobs_run++; *(volatile uint8_t *)jobs[i].body += 1; jobs[i].rc = SC_OK; }
    return SC_OK;
}
}

// the compiler's registration hooks for device code: there is none in this build
extern "C" {
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void *, void *, const char *, size_t, unsigned) {}
void __hipRegisterSurface(void **, void *, char *, char *, int, int) {}
void __hipRegisterTexture(void **, void *, char *, char *, int, int, int) {}
}

static int fail(const char *what) { fprintf(stderr, "sanitize_main: %s\n", what); return 1; }

int main()
{
    // 1
    if (sc_hip_selftest_host() != 0) return fail("sc_hip_selftest_host");
    // 2: the planner, concurrently
    {
        const int N = 96, T = 6, ROUNDS = 40;
        std::vector<int> wh(2 * N);
        unsigned s = 12345u;
        auto rnd = [&](int lo, int hi) { s = s * 1664525u + 1013904223u; return lo + (int)((s >> 8) % (unsigned)(hi - lo + 1)); };
        for (int i = 0; i < N; ++i) { wh[2 * i] = rnd(90, 2300); wh[2 * i + 1] = rnd(90, 2300); }
        for (int i = 0; i < N; i += 3) { wh[2 * i] = 1000 + i; wh[2 * i + 1] = 1040 - i; }      // a size class among them
        std::vector<int> ref_g(N), ref_k(N);
        const int ref_n = sc_hip_plan_groups(wh.data(), N, 16, nullptr, ref_g.data(), ref_k.data());
        if (ref_n < 1) return fail("plan_groups");
        std::atomic<int> bad{ 0 };
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() {
                std::vector<int> g(N), k(N), other(2 * 700);
                unsigned q = 777u * (t + 1);
                for (int r = 0; r < ROUNDS; ++r) {
                    if (sc_hip_plan_groups(wh.data(), N, 16, nullptr, g.data(), k.data()) != ref_n || g != ref_g || k != ref_k) bad++;
                    for (int i = 0; i < 700; ++i) { q = q * 1664525u + 1013904223u; other[2 * i] = 100 + (int)((q >> 9) % 3000u); other[2 * i + 1] = 100 + (int)((q >> 3) % 3000u); }
                    std::vector<int> og(700);
                    if (sc_hip_plan_groups(other.data(), 700, 0, nullptr, og.data(), nullptr) < 1) bad++;      // misses, evictions
                    int out[12];
                    if (sc_hip_plan_size(1000 + r, 1000 + t, nullptr, out) != SC_OK) bad++;
                }
            });
        for (std::thread &x : th) x.join();
        if (bad.load()) return fail("the planner gave different answers under concurrency");
    }
    // 3: the pool's hand-out
    for (int streams : { 1, 3, 8 })
        for (int group : { 1, 4, 16 }) {
            void *pool = sc_hip_pool_create(0, streams);
            if (!pool) return fail("pool_create (stub instances)");
            if (sc_hip_pool_set_group(pool, group) != SC_OK) return fail("pool_set_group");
            for (int batch = 0; batch < 5; ++batch) {
                const int n = 1 + 37 * batch;
                std::vector<uint8_t> bodies(n, 0);
                std::vector<sc_batch_job> jobs(n);
                unsigned s = 99u + batch;
                for (int i = 0; i < n; ++i) {
                    memset(&jobs[i], 0, sizeof(jobs[i]));
                    s = s * 1664525u + 1013904223u;
                    const int w = (i % 5 == 0) ? 1002 : 1000 + (int)((s >> 10) % 90u), h = 1000 + (int)((s >> 17) % 90u);
                    jobs[i].face = jobs[i].mask = &bodies[i]; jobs[i].body = &bodies[i];
                    jobs[i].face_cols = jobs[i].mask_cols = w; jobs[i].face_rows = jobs[i].mask_rows = h;
                    jobs[i].face_step = 3 * w; jobs[i].mask_step = w; jobs[i].body_cols = 4000; jobs[i].body_rows = 4000; jobs[i].body_step = 12000;
                    jobs[i].centerX = jobs[i].centerY = 2000;
                    jobs[i].rc = -12345;
                }
                const long before = g_jobs_run.load();
                for (int device_resident : { 1, 0 }) {
                    if (sc_hip_pool_run(pool, jobs.data(), n, device_resident) != SC_OK) return fail("pool_run");
                    for (int i = 0; i < n; ++i) if (jobs[i].rc != SC_OK) return fail("a job's code was not copied back");
                }
                if (g_jobs_run.load() - before != 2L * n) return fail("jobs run != jobs given");
                for (int i = 0; i < n; ++i) if (bodies[i] != 2) return fail("a job was handed out twice or never");
            }
            sc_hip_pool_destroy(pool);
        }
    printf("sanitize_main: clean\n");
    return 0;
}
