/*
 * sc_oracle.c -- CPU restatement (plain C, float32) of the seamless-clone hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libseamlessclone_hip.so) never
 * links, loads or calls it.
 *
 * Reference: wujinzhong/seamlessCloneOptimization (a CUDA restatement of OpenCV 3.4.5
 * cv::seamlessClone(NORMAL_CLONE); OpenCV itself is an un-vendored dependency, absent
 * here).  Each function cites the reference lines it follows
 * (IMP.cpp = seamlessClone-CUDA/seamlessClone_imp.cpp).
 *
 * PARITY PIN: see oracle/oracle_np.py header -- pinned by the reference's input fixtures
 * and its one committed (lossy JPEG) output; bit-level parity with OpenCV is unpinned.
 *
 * Layout conventions (same as the HIP library):
 *   images   : interleaved 3-channel u8, row stride in bytes (cv::Mat data/step)
 *   fields   : planar float32 [c][y][x], C=3, dense H x W (ring included)
 *   geo[6]   : {x0, y0, W, H, ltx, lty}
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SCO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ mask stage */

/* one 3x3 erode pass: IMP.cpp:892-925 */
static void erode_pass(uint8_t *dst, const uint8_t *src, int W, int H)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int idx = y * W + x;
            if (x == 0 || y == 0 || x == W - 1 || y == H - 1) { dst[idx] = 0; continue; }
            int sum = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) sum += src[idx + dy * W + dx];
            dst[idx] = (sum == 255 * 9) ? 255 : 0;
        }
}

/* IMP.cpp:978-1071.  M must hold mw*mh bytes; on return its first W*H bytes are the
 * 3x eroded ROI mask.  Returns 0, or -3 for an empty / degenerate mask (:1013). */
SCO_API int sco_mask_stage(const uint8_t *mask, int mw, int mh, int mstride, int cx, int cy,
                           int *geo, uint8_t *M)
{
    int x0 = mw - 1, x1 = 0, y0 = mh - 1, y1 = 0; /* seeds: IMP.cpp:1006 */
    for (int y = 1; y < mh - 1; ++y)             /* border zeroed first: :989 */
        for (int x = 1; x < mw - 1; ++x)
            if (mask[(size_t)y * mstride + x] != 0) { /* :945 */
                if (x < x0) x0 = x;
                if (x > x1) x1 = x;
                if (y < y0) y0 = y;
                if (y > y1) y1 = y;
            }
    if (!((x1 - x0) > 0 && (y1 - y0) > 0)) return -3;
    int W = x1 - x0 + 1, H = y1 - y0 + 1;
    uint8_t *a = (uint8_t *)malloc((size_t)W * H), *b = (uint8_t *)malloc((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int sx = x + x0, sy = y + y0;
            int border = (sx == 0 || sy == 0 || sx == mw - 1 || sy == mh - 1);
            a[y * W + x] = border ? 0 : mask[(size_t)sy * mstride + sx];
        }
    erode_pass(b, a, W, H); /* :1060-1062 */
    erode_pass(a, b, W, H);
    erode_pass(b, a, W, H);
    memcpy(M, b, (size_t)W * H);
    free(a); free(b);
    geo[0] = x0; geo[1] = y0; geo[2] = W; geo[3] = H;
    geo[4] = cx - (W >> 1); /* :1066 */
    geo[5] = cy - (H >> 1);
    return 0;
}

/* ------------------------------------------------------------------ RHS */

/* IMP.cpp:1920-2018 fused: B = dst ROI as float (planar), lap = un-folded divergence of
 * the mask-blended forward-difference gradient field (interior; ring = 0).
 * Returns -4 when the ROI leaves the destination (unchecked in the reference). */
SCO_API int sco_build_rhs(const uint8_t *dst, int dw, int dh, int dstride,
                          const uint8_t *patch, int pw, int ph, int pstride,
                          const uint8_t *M, const int *geo, float *B, float *lap)
{
    const int x0 = geo[0], y0 = geo[1], W = geo[2], H = geo[3], ltx = geo[4], lty = geo[5];
    if (ltx < 0 || lty < 0 || ltx + W > dw || lty + H > dh) return -4;
    if (x0 + W > pw || y0 + H > ph) return -2;
    const size_t plane = (size_t)W * H;
#define BK(y, x) ((float)dst[(size_t)((y) + lty) * dstride + ((x) + ltx) * 3 + c])
#define PT(y, x) ((float)patch[(size_t)((y) + y0) * pstride + ((x) + x0) * 3 + c])
    for (int c = 0; c < 3; ++c) {
        float *gx = (float *)malloc(plane * sizeof(float));
        float *gy = (float *)malloc(plane * sizeof(float));
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float b0 = BK(y, x), p0 = PT(y, x);
                float bx = (x < W - 1) ? BK(y, x + 1) : BK(y, x - 1); /* :1937 */
                float by = (y < H - 1) ? BK(y + 1, x) : BK(y - 1, x); /* :1940 */
                float px = (x < W - 1) ? PT(y, x + 1) : PT(y, x - 1); /* :1944 */
                float py = (y < H - 1) ? PT(y + 1, x) : PT(y - 1, x); /* :1947 */
                float msk = (float)M[y * W + x] * (1.0f / 255.0f);    /* :1950 */
                gx[y * W + x] = (1.0f - msk) * (bx - b0) + msk * (px - p0); /* :1952 */
                gy[y * W + x] = (1.0f - msk) * (by - b0) + msk * (py - p0); /* :1953 */
                B[c * plane + y * W + x] = b0;
                lap[c * plane + y * W + x] = 0.0f;
            }
        for (int y = 1; y < H - 1; ++y)
            for (int x = 1; x < W - 1; ++x) {
                float dx_ = gx[y * W + x] - gx[y * W + x - 1]; /* :1987 */
                float dy_ = gy[y * W + x] - gy[(y - 1) * W + x]; /* :1988 */
                lap[c * plane + y * W + x] = dx_ + dy_;
            }
        free(gx); free(gy);
    }
#undef BK
#undef PT
    return 0;
}

/* IMP.cpp:1992-2008: compact folded RHS g[(y-1)*w + (x-1)], planar [c]. */
SCO_API void sco_fold(const float *B, const float *lap, int W, int H, float *g)
{
    const int w = W - 2, h = H - 2;
    const size_t plane = (size_t)W * H, gpl = (size_t)w * h;
    for (int c = 0; c < 3; ++c)
        for (int y = 1; y < H - 1; ++y)
            for (int x = 1; x < W - 1; ++x) {
                const float *b = B + c * plane + (size_t)y * W + x;
                float v = lap[c * plane + (size_t)y * W + x];
                if (x == 1) v -= b[-1];
                if (y == 1) v -= b[-W];
                if (x == W - 2) v -= b[1];
                if (y == H - 2) v -= b[W];
                g[c * gpl + (size_t)(y - 1) * w + (x - 1)] = v;
            }
}

/* ------------------------------------------------------------------ FFT (Bluestein over radix-2) */

/* FFT internals run in double: a float32 Bluestein loses ~3 grey levels of smooth error at
 * 2048^2 (measured against the float64 numpy oracle); inputs/outputs stay float32. */
typedef double real_t;
typedef struct { real_t re, im; } cpx;

static inline cpx cmul(cpx a, cpx b) { cpx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }

typedef struct {
    int n;        /* transform length */
    int m;        /* radix-2 length (== n when n is a power of two, else >= 2n-1) */
    int direct;   /* n is a power of two */
    cpx *tw;      /* m/2 twiddles exp(-2 pi i k / m) */
    int *rev;     /* bit reversal, m */
    cpx *chirp;   /* n: exp(-pi i k^2 / n) */
    cpx *bfft;    /* m: FFT of the conjugate-chirp filter */
} fft_plan;

static void fft_pow2(cpx *x, int m, const cpx *tw, const int *rev, int inverse)
{
    for (int i = 0; i < m; ++i) { int j = rev[i]; if (j > i) { cpx t = x[i]; x[i] = x[j]; x[j] = t; } }
    for (int len = 2; len <= m; len <<= 1) {
        int half = len >> 1, step = m / len;
        for (int i = 0; i < m; i += len)
            for (int k = 0; k < half; ++k) {
                cpx w = tw[k * step];
                if (inverse) w.im = -w.im;
                cpx a = x[i + k], b = cmul(x[i + k + half], w);
                x[i + k].re = a.re + b.re; x[i + k].im = a.im + b.im;
                x[i + k + half].re = a.re - b.re; x[i + k + half].im = a.im - b.im;
            }
    }
}

static fft_plan *plan_create(int n)
{
    fft_plan *p = (fft_plan *)calloc(1, sizeof(fft_plan));
    p->n = n;
    int pow2 = (n & (n - 1)) == 0;
    p->direct = pow2;
    int m = 1;
    if (pow2) m = n; else while (m < 2 * n - 1) m <<= 1;
    p->m = m;
    p->tw = (cpx *)malloc(sizeof(cpx) * (m / 2 + 1));
    for (int k = 0; k < m / 2; ++k) {
        double a = -2.0 * M_PI * k / m;
        p->tw[k].re = cos(a); p->tw[k].im = sin(a);
    }
    p->rev = (int *)malloc(sizeof(int) * m);
    int bits = 0; while ((1 << bits) < m) ++bits;
    for (int i = 0; i < m; ++i) {
        int r = 0; for (int b = 0; b < bits; ++b) if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        p->rev[i] = r;
    }
    if (!pow2) {
        p->chirp = (cpx *)malloc(sizeof(cpx) * n);
        for (int k = 0; k < n; ++k) {
            long long k2 = ((long long)k * k) % (2LL * n); /* keep the phase argument small */
            double a = -M_PI * (double)k2 / n;
            p->chirp[k].re = cos(a); p->chirp[k].im = sin(a);
        }
        p->bfft = (cpx *)calloc(m, sizeof(cpx));
        for (int k = 0; k < n; ++k) {
            cpx c = { p->chirp[k].re, -p->chirp[k].im };
            p->bfft[k] = c;
            if (k) p->bfft[m - k] = c;
        }
        fft_pow2(p->bfft, m, p->tw, p->rev, 0);
    }
    return p;
}

static void plan_destroy(fft_plan *p)
{
    if (!p) return;
    free(p->tw); free(p->rev); free(p->chirp); free(p->bfft); free(p);
}

/* forward complex DFT of length p->n, in place in x[0..n); work holds p->m entries */
static void fft_forward(const fft_plan *p, cpx *x, cpx *work)
{
    if (p->direct) { fft_pow2(x, p->n, p->tw, p->rev, 0); return; }
    const int n = p->n, m = p->m;
    for (int k = 0; k < n; ++k) work[k] = cmul(x[k], p->chirp[k]);
    for (int k = n; k < m; ++k) { work[k].re = 0; work[k].im = 0; }
    fft_pow2(work, m, p->tw, p->rev, 0);
    for (int k = 0; k < m; ++k) work[k] = cmul(work[k], p->bfft[k]);
    fft_pow2(work, m, p->tw, p->rev, 1);
    const real_t s = 1.0 / (real_t)m;
    for (int k = 0; k < n; ++k) {
        cpx v = { work[k].re * s, work[k].im * s };
        x[k] = cmul(v, p->chirp[k]);
    }
}

/* Unnormalised DST-I of two real rows a,b (length n) at once through ONE complex FFT of
 * the odd extension, length N = 2n+2 -- the construction OpenCV uses (quoted at
 * IMP.cpp:1342-1351): t = [0, x, 0, -reverse(x)];  DST_k = -Im(FFT(t))_{k+1} / 2. */
static void dst1_pair(const fft_plan *p, const float *a, const float *b, float *oa, float *ob,
                      int n, cpx *t, cpx *work)
{
    const int N = 2 * n + 2;
    t[0].re = t[0].im = 0; t[n + 1].re = t[n + 1].im = 0;
    for (int j = 0; j < n; ++j) {
        real_t xa = a[j], xb = b ? b[j] : 0;
        t[1 + j].re = xa; t[1 + j].im = xb;
        t[N - 1 - j].re = -xa; t[N - 1 - j].im = -xb;
    }
    fft_forward(p, t, work);
    /* FFT(ta) = i*A, FFT(tb) = i*Bv  ->  T = i*A - Bv : Im(T) = A, Re(T) = -Bv */
    for (int k = 0; k < n; ++k) {
        oa[k] = (float)(-0.5 * t[k + 1].im);
        if (ob) ob[k] = (float)(0.5 * t[k + 1].re);
    }
}

static void dst1_rows(const fft_plan *p, float *data, int rows, int n, int nthreads)
{
    (void)nthreads;
#pragma omp parallel num_threads(nthreads)
    {
        cpx *t = (cpx *)malloc(sizeof(cpx) * (2 * n + 2));
        cpx *work = (cpx *)malloc(sizeof(cpx) * p->m);
#pragma omp for schedule(static)
        for (int r = 0; r < rows; r += 2) {
            float *a = data + (size_t)r * n;
            float *b = (r + 1 < rows) ? a + n : NULL;
            dst1_pair(p, a, b, a, b, n, t, work);
        }
        free(t); free(work);
    }
}

static void transpose(const float *src, float *dst, int rows, int cols)
{
    const int T = 32;
    for (int r0 = 0; r0 < rows; r0 += T)
        for (int c0 = 0; c0 < cols; c0 += T)
            for (int r = r0; r < r0 + T && r < rows; ++r)
                for (int c = c0; c < c0 + T && c < cols; ++c) dst[(size_t)c * rows + r] = src[(size_t)r * cols + c];
}

/* A.4: u = S_h ( (S_h g S_w) / den ) S_w * 4/((w+1)(h+1)),  den = 2cos(pi(i+1)/(w+1)) +
 * 2cos(pi(j+1)/(h+1)) - 4  (IMP.cpp:1825-1832, tables :596-599 in double, stored float).
 * g,u planar [c][h][w]; may alias. */
SCO_API void sco_solve_dst2(const float *g, int w, int h, int C, float *u, int nthreads, int exact_den);
SCO_API void sco_solve_dst(const float *g, int w, int h, int C, float *u, int nthreads)
{
    sco_solve_dst2(g, w, h, C, u, nthreads, 0);
}

/* exact_den = 0: eigenvalue tables stored as float and combined in float -- what OpenCV and
 * the reference do (IMP.cpp:596-599, :1651-1653).  At 2048^2 the float cancellation in
 * (fx + fy - 4) perturbs the lowest modes by a few percent, i.e. OpenCV's own answer is up to
 * ~3 grey levels away from the exact solution of the linear system.
 * exact_den = 1: denominator formed in double (agrees with the float64 numpy oracle). */
SCO_API void sco_solve_dst2(const float *g, int w, int h, int C, float *u, int nthreads, int exact_den)
{
    if (nthreads < 1) nthreads = 1;
    fft_plan *pw = plan_create(2 * w + 2), *ph = plan_create(2 * h + 2);
    float *fx = (float *)malloc(sizeof(float) * w), *fy = (float *)malloc(sizeof(float) * h);
    double *dx = (double *)malloc(sizeof(double) * w), *dy = (double *)malloc(sizeof(double) * h);
    /* float tables exactly as the reference builds them (IMP.cpp:596-599): double cos of PI/(n+1.0)*(x+1.0) with PI the
     * FLOAT literal 3.14159265358979323846f of seamlessClone_imp.h:17, stored as float.  The exact-denominator
     * variant (not the reference's: the exact linear system) uses the true pi in double. */
    const double PIf = (double)3.14159265358979323846f;
    for (int i = 0; i < w; ++i) { dx[i] = 2.0 * cos(M_PI * (i + 1.0) / (w + 1.0)); fx[i] = (float)(2.0 * cos(PIf / (w + 1.0) * (i + 1.0))); }
    for (int j = 0; j < h; ++j) { dy[j] = 2.0 * cos(M_PI * (j + 1.0) / (h + 1.0)); fy[j] = (float)(2.0 * cos(PIf / (h + 1.0) * (j + 1.0))); }
    const size_t pl = (size_t)w * h;
    float *a = (float *)malloc(sizeof(float) * pl), *b = (float *)malloc(sizeof(float) * pl);
    const float scale = (float)(4.0 / ((double)(w + 1) * (double)(h + 1)));
    for (int c = 0; c < C; ++c) {
        memcpy(a, g + c * pl, sizeof(float) * pl);
        dst1_rows(pw, a, h, w, nthreads);
        transpose(a, b, h, w);
        dst1_rows(ph, b, w, h, nthreads); /* b[x][y] */
        for (int x = 0; x < w; ++x)
            for (int y = 0; y < h; ++y) {
                if (exact_den) b[(size_t)x * h + y] = (float)((double)b[(size_t)x * h + y] / (dx[x] + dy[y] - 4.0));
                else b[(size_t)x * h + y] /= (fx[x] + fy[y] - 4.0f);
            }
        dst1_rows(ph, b, w, h, nthreads);
        transpose(b, a, w, h);
        dst1_rows(pw, a, h, w, nthreads);
        for (size_t i = 0; i < pl; ++i) u[c * pl + i] = a[i] * scale;
    }
    free(a); free(b); free(fx); free(fy); free(dx); free(dy);
    plan_destroy(pw); plan_destroy(ph);
}

/* ------------------------------------------------------------------ stencil sweeps (A.5) */

/* Jacobi: U' = 0.25f*(((l+r)+(u+d)) - f); ring fixed; `sweeps` ping-pong passes. */
SCO_API void sco_jacobi(float *U, const float *lap, int W, int H, int C, int sweeps)
{
    const size_t plane = (size_t)W * H;
    float *tmp = (float *)malloc(sizeof(float) * plane);
    for (int c = 0; c < C; ++c) {
        float *cur = U + c * plane, *nxt = tmp;
        const float *f = lap + c * plane;
        memcpy(tmp, cur, sizeof(float) * plane);
        for (int s = 0; s < sweeps; ++s) {
            for (int y = 1; y < H - 1; ++y) {
                const float *r = cur + (size_t)y * W;
                float *o = nxt + (size_t)y * W;
                const float *fr = f + (size_t)y * W;
                for (int x = 1; x < W - 1; ++x)
                    o[x] = 0.25f * (((r[x - 1] + r[x + 1]) + (r[x - W] + r[x + W])) - fr[x]);
            }
            float *t = cur; cur = nxt; nxt = t;
        }
        if (cur != U + c * plane) memcpy(U + c * plane, cur, sizeof(float) * plane);
    }
    free(tmp);
}

/* Red-black GS / SOR, colour=(x+y)&1, colour 0 first.  omega==1 -> plain GS. */
SCO_API void sco_rbgs(float *U, const float *lap, int W, int H, int C, int sweeps, float omega)
{
    const size_t plane = (size_t)W * H;
    for (int c = 0; c < C; ++c) {
        float *u = U + c * plane;
        const float *f = lap + c * plane;
        for (int s = 0; s < sweeps; ++s)
            for (int color = 0; color < 2; ++color)
                for (int y = 1; y < H - 1; ++y) {
                    float *r = u + (size_t)y * W;
                    const float *fr = f + (size_t)y * W;
                    for (int x = 1 + ((1 + y + color) & 1); x < W - 1; x += 2) {
                        float gs = 0.25f * (((r[x - 1] + r[x + 1]) + (r[x - W] + r[x + W])) - fr[x]);
                        r[x] = (omega == 1.0f) ? gs : r[x] + omega * (gs - r[x]);
                    }
                }
    }
}

/* out[0] = sum r^2, out[1] = sum lap^2 over interior and channels; r computed in float32,
 * accumulated in double. */
SCO_API void sco_residual(const float *U, const float *lap, int W, int H, int C, double *out)
{
    const size_t plane = (size_t)W * H;
    double r2 = 0.0, f2 = 0.0;
    for (int c = 0; c < C; ++c)
        for (int y = 1; y < H - 1; ++y) {
            const float *r = U + c * plane + (size_t)y * W;
            const float *fr = lap + c * plane + (size_t)y * W;
            for (int x = 1; x < W - 1; ++x) {
                float s = ((r[x - 1] + r[x + 1]) + (r[x - W] + r[x + W])) - 4.0f * r[x];
                float res = fr[x] - s;
                r2 += (double)res * res; f2 += (double)fr[x] * fr[x];
            }
        }
    out[0] = r2; out[1] = f2;
}

/* ------------------------------------------------------------------ output (A.7) */

/* interior of planar field U (H x W, ring ignored) -> clamp, truncate, splice into dst.
 * IMP.cpp:2091-2096 and :470-483. */
SCO_API void sco_finish(uint8_t *dst, int dstride, const float *U, const int *geo)
{
    const int W = geo[2], H = geo[3], ltx = geo[4], lty = geo[5];
    const size_t plane = (size_t)W * H;
    for (int c = 0; c < 3; ++c)
        for (int y = 1; y < H - 1; ++y)
            for (int x = 1; x < W - 1; ++x) {
                float d = U[c * plane + (size_t)y * W + x];
                d = d > 255.0f ? 255.0f : d;
                d = d < 0.0f ? 0.0f : d;
                dst[(size_t)(y + lty) * dstride + (x + ltx) * 3 + c] = (uint8_t)d;
            }
}

/* Whole NORMAL_CLONE path with the direct DST solve, in place on dst (reference
 * semantics, IMP.cpp:470).  Returns 0 or a negative error. */
SCO_API int sco_seamless_clone2(const uint8_t *patch, int pw, int ph, int pstride,
                                uint8_t *dst, int dw, int dh, int dstride,
                                const uint8_t *mask, int mw, int mh, int mstride,
                                int cx, int cy, int nthreads, int exact_den)
{
    if (pw != mw || ph != mh) return -2;
    int geo[6];
    uint8_t *M = (uint8_t *)malloc((size_t)mw * mh);
    int rc = sco_mask_stage(mask, mw, mh, mstride, cx, cy, geo, M);
    if (rc) { free(M); return rc; }
    const int W = geo[2], H = geo[3], w = W - 2, h = H - 2;
    const size_t plane = (size_t)W * H;
    float *B = (float *)malloc(sizeof(float) * plane * 3);
    float *lap = (float *)malloc(sizeof(float) * plane * 3);
    rc = sco_build_rhs(dst, dw, dh, dstride, patch, pw, ph, pstride, M, geo, B, lap);
    if (rc == 0 && w > 0 && h > 0) {
        float *g = (float *)malloc(sizeof(float) * (size_t)w * h * 3);
        sco_fold(B, lap, W, H, g);
        sco_solve_dst2(g, w, h, 3, g, nthreads, exact_den);
        for (int c = 0; c < 3; ++c)
            for (int y = 1; y < H - 1; ++y)
                for (int x = 1; x < W - 1; ++x)
                    B[c * plane + (size_t)y * W + x] = g[(size_t)c * w * h + (size_t)(y - 1) * w + (x - 1)];
        sco_finish(dst, dstride, B, geo);
        free(g);
    }
    free(B); free(lap); free(M);
    return rc;
}

SCO_API int sco_seamless_clone(const uint8_t *patch, int pw, int ph, int pstride,
                               uint8_t *dst, int dw, int dh, int dstride,
                               const uint8_t *mask, int mw, int mh, int mstride,
                               int cx, int cy, int nthreads)
{
    return sco_seamless_clone2(patch, pw, ph, pstride, dst, dw, dh, dstride, mask, mw, mh, mstride,
                               cx, cy, nthreads, 0);
}

SCO_API int sco_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
