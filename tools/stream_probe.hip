// stream_probe -- what a pure streaming kernel reaches on this GPU at the solver's working-set sizes.
// out = a + b over float4 (read 8 B + write 4 B per element = the Jacobi sweep's 12 B/unknown traffic shape),
// and a copy (4 B + 4 B).  Build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_triad(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o, size_t n4)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    float4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n4) { va[u] = a[i + (size_t)u * 256]; vb[u] = b[i + (size_t)u * 256]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n4) {
        float4 r; r.x = va[u].x + vb[u].x; r.y = va[u].y + vb[u].y; r.z = va[u].z + vb[u].z; r.w = va[u].w + vb[u].w;
        o[i + (size_t)u * 256] = r;
    }
}

template <int U>
__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ a, float4 *__restrict__ o, size_t n4)
{
    size_t i = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    float4 va[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n4) va[u] = a[i + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + (size_t)u * 256 < n4) o[i + (size_t)u * 256] = va[u];
}


// the same sum with the Jacobi kernels' tiling: a wave owns 256 columns x S rows of a pitch-P plane
template <int S, int HALO>
__global__ __launch_bounds__(256) void k_triad2d(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o, int P, int H)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t plane = (size_t)P * H;
    a += blockIdx.z * plane; b += blockIdx.z * plane; o += blockIdx.z * plane;
    const int x = blockIdx.x * 256 + 4 * lane, ya = (blockIdx.y * 4 + wv) * S;
    float4 va[S + 2 * HALO], vb[S];
#pragma unroll
    for (int k = 0; k < S + 2 * HALO; ++k) { int y = ya + k - HALO; y = y < 0 ? 0 : (y >= H ? H - 1 : y); va[k] = *reinterpret_cast<const float4 *>(a + (size_t)y * P + x); }
#pragma unroll
    for (int k = 0; k < S; ++k) vb[k] = *reinterpret_cast<const float4 *>(b + (size_t)(ya + k) * P + x);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        float4 r = va[k + HALO];
        if (HALO) { r.x += va[k].x + va[k + 2].x; r.y += va[k].y + va[k + 2].y; r.z += va[k].z + va[k + 2].z; r.w += va[k].w + va[k + 2].w; }
        r.x += vb[k].x; r.y += vb[k].y; r.z += vb[k].z; r.w += vb[k].w;
        *reinterpret_cast<float4 *>(o + (size_t)(ya + k) * P + x) = r;
    }
}

__global__ void k_fill(float *p, size_t n, unsigned seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = 100.0f + 30.0f * ((h & 0xffff) / 65536.0f - 0.5f);
}

template <typename L>
static double time_us(L launch, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char **argv)
{
    int sides[] = {2048, 4096, 8192};
    for (int side : sides) {
        size_t n = (size_t)side * side * 3, n4 = n / 4;
        float4 *a, *b, *o;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&o, n * 4));
        if (argc > 1) { CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); }   // any argument: all-zero data
        else { k_fill<<<(unsigned)((n + 255) / 256), 256>>>((float *)a, n, 1u); k_fill<<<(unsigned)((n + 255) / 256), 256>>>((float *)b, n, 2u); }
        double best_t = 1e30, best_c = 1e30; int ut = 0, uc = 0;
#define TRY(U) { unsigned g = (unsigned)((n4 + 256 * U - 1) / (256 * U)); \
        double t = time_us([&] { k_triad<U><<<g, 256>>>(a, b, o, n4); }, 50); if (t < best_t) { best_t = t; ut = U; } \
        double c = time_us([&] { k_copy<U><<<g, 256>>>(a, o, n4); }, 50); if (c < best_c) { best_c = c; uc = U; } }
        TRY(1) TRY(2) TRY(4) TRY(8)
        {
            const float *fa = (const float *)a, *fb = (const float *)b; float *fo = (float *)o;
#define T2D(S, HALO) { dim3 g(side / 256, side / (4 * S), 3); double t = time_us([&] { k_triad2d<S, HALO><<<g, 256>>>(fa, fb, fo, side, side); }, 50); \
            printf("  tiled S=%d halo=%d: %.2f us  %.0f GB/s (12 B/elt)\n", S, HALO, t, 12.0 * n / t / 1e3); }
            T2D(4, 0) T2D(8, 0) T2D(16, 0) T2D(4, 1) T2D(8, 1) T2D(16, 1)
        }
        printf("{\"side\": %d, \"plane_MB\": %.1f, \"triad_us\": %.2f, \"triad_GBps\": %.0f, \"triad_unroll\": %d, "
               "\"copy_us\": %.2f, \"copy_GBps\": %.0f, \"copy_unroll\": %d}\n",
               side, n * 4 / 1e6, best_t, 12.0 * n / best_t / 1e3, ut, best_c, 8.0 * n / best_c / 1e3, uc);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(o));
    }
    return 0;
}
