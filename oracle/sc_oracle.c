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
SCO_API int sco_mask_stage2(const uint8_t *mask, int mw, int mh, int mstride, int cx, int cy, int *geo, uint8_t *M, int grey);
SCO_API int sco_mask_stage(const uint8_t *mask, int mw, int mh, int mstride, int cx, int cy,
                           int *geo, uint8_t *M)
{
    return sco_mask_stage2(mask, mw, mh, mstride, cx, cy, geo, M, 0);
}

/* grey != 0: OpenCV's semantics for masks that are not 0/255 (PARITY UNPINNED: restatement of the published OpenCV 3.4.5
 * algorithm, modules/photo/src/seamless_cloning_impl.cpp Cloning::computeDerivatives -- erode(mask ROI view, 3x3 ones,
 * iterations 3) is one 7x7 minimum filter reading zeros outside the bounding box; see oracle_np.erode_min7).  The reference
 * thresholds instead (IMP.cpp:917: sum == 255 * 9). */
SCO_API int sco_mask_stage2(const uint8_t *mask, int mw, int mh, int mstride, int cx, int cy,
                            int *geo, uint8_t *M, int grey)
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
    if (grey) {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int mn = 255;
                for (int dy = -3; dy <= 3; ++dy)
                    for (int dx = -3; dx <= 3; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        const int v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : a[yy * W + xx];
                        if (v < mn) mn = v;
                    }
                b[y * W + x] = (uint8_t)mn;
            }
    } else {
        erode_pass(b, a, W, H); /* :1060-1062 */
        erode_pass(a, b, W, H);
        erode_pass(b, a, W, H);
    }
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
SCO_API int sco_build_rhs2(const uint8_t *dst, int dw, int dh, int dstride, const uint8_t *patch, int pw, int ph, int pstride,
                           const uint8_t *M, const int *geo, float *B, float *lap, int grey);
SCO_API int sco_build_rhs(const uint8_t *dst, int dw, int dh, int dstride,
                          const uint8_t *patch, int pw, int ph, int pstride,
                          const uint8_t *M, const int *geo, float *B, float *lap)
{
    return sco_build_rhs2(dst, dw, dh, dstride, patch, pw, ph, pstride, M, geo, B, lap, 0);
}

/* grey != 0: OpenCV's blend for a grey eroded mask (Cloning::normalClone / evaluate, OpenCV 3.4.5): patch gradient times
 * M * (1/255f), destination gradient times (255 - M) * (1/255f) (convertTo of the mask and of its bitwise_not), summed
 * destination first.  Identical to the reference's formula for M in {0, 255}. */
SCO_API int sco_build_rhs2(const uint8_t *dst, int dw, int dh, int dstride,
                           const uint8_t *patch, int pw, int ph, int pstride,
                           const uint8_t *M, const int *geo, float *B, float *lap, int grey)
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
                if (grey) {
                    const float inv = (float)(255 - M[y * W + x]) * (1.0f / 255.0f);
                    gx[y * W + x] = (bx - b0) * inv + (px - p0) * msk;
                    gy[y * W + x] = (by - b0) * inv + (py - p0) * msk;
                } else {
                    gx[y * W + x] = (1.0f - msk) * (bx - b0) + msk * (px - p0); /* :1952 */
                    gy[y * W + x] = (1.0f - msk) * (by - b0) + msk * (py - p0); /* :1953 */
                }
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

/* ------------------------------------------------------------------ float32 transform internals
 *
 * OpenCV's dft() and cuFFT (cufftExecC2C, IMP.cpp:1740,1791) run the two batched 1-D complex transforms of the DST in
 * FLOAT32; the port above transforms in double.  The variants below bound what that costs: the same odd-extension
 * construction, every buffer float, with
 *   internals 1: a mixed-radix Cooley-Tukey transform (radix 4 / 2 and a generic O(p^2) butterfly for every other prime
 *                factor, twiddle table computed in double and stored float) -- the kind of transform OpenCV's dft and
 *                cuFFT run for composite lengths; lengths with a prime factor > 127 go through Bluestein, as in cuFFT;
 *   internals 2: Bluestein's chirp-z over a power-of-two float transform for every length.
 * Neither is bit-identical to OpenCV or cuFFT (their butterfly order is not published to that level); they show the size
 * of float32 transform rounding in the 8-bit result.  The sequence of steps is the reference's (IMP.cpp:1694-1896):
 * odd extension [0, s, 0, -reverse(s)] as complex with zero imaginary part (:1342-1351), transform of length 2n+2,
 * imaginary part of bin j+1 transposed and odd-extended along the other axis (:1465-1478), second transform,
 * transposed copy of the imaginary part (:1797-1801), divide (:1646-1655), the same again with the inverse flag,
 * one final scale 1.0f/((2w+2)(2h+2)) (:1893). */
typedef struct { float re, im; } cpxf;
static inline cpxf cmulf(cpxf a, cpxf b) { cpxf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }

typedef struct fftf_plan {
    int n, inverse;
    int fac[2 * 40];       /* (radix, remaining length) pairs */
    cpxf *tw;              /* n twiddles exp(-/+ 2 pi i k / n) */
    int maxp;              /* largest generic radix */
    /* Bluestein */
    int blue, m;
    struct fftf_plan *sub_f, *sub_i;
    cpxf *chirp, *bfft;
} fftf_plan;

static void fftf_factor(int n, int *fac, int *maxp)
{
    int p = 4, i = 0;
    *maxp = 1;
    while (n > 1) {
        while (n % p) {
            if (p == 4) p = 2; else if (p == 2) p = 3; else p += 2;
            if ((long long)p * p > n) p = n;
        }
        n /= p;
        fac[i++] = p; fac[i++] = n;
        if (p != 4 && p != 2 && p > *maxp) *maxp = p;
    }
}

static int largest_prime_factor(int n)
{
    int best = 1;
    for (int p = 2; (long long)p * p <= n; ++p) while (n % p == 0) { best = p; n /= p; }
    return n > 1 ? n : best;
}

static void fftf_work(cpxf *out, const cpxf *in, int fstride, const int *fac, const fftf_plan *P, cpxf *scratch)
{
    const int p = fac[0], m = fac[1];
    if (m == 1) { for (int q = 0; q < p; ++q) out[q] = in[(size_t)q * fstride]; }
    else for (int q = 0; q < p; ++q) fftf_work(out + (size_t)q * m, in + (size_t)q * fstride, fstride * p, fac + 2, P, scratch);
    const cpxf *tw = P->tw;
    const int N = P->n;
    if (p == 2) {
        for (int k = 0; k < m; ++k) {
            cpxf t = cmulf(out[k + m], tw[(size_t)k * fstride]);
            cpxf a = out[k];
            out[k + m].re = a.re - t.re; out[k + m].im = a.im - t.im;
            out[k].re = a.re + t.re; out[k].im = a.im + t.im;
        }
    } else if (p == 4) {
        for (int k = 0; k < m; ++k) {
            cpxf a0 = out[k];
            cpxf a1 = cmulf(out[k + m], tw[(size_t)k * fstride]);
            cpxf a2 = cmulf(out[k + 2 * m], tw[(size_t)k * fstride * 2]);
            cpxf a3 = cmulf(out[k + 3 * m], tw[(size_t)k * fstride * 3]);
            cpxf s02 = { a0.re + a2.re, a0.im + a2.im }, d02 = { a0.re - a2.re, a0.im - a2.im };
            cpxf s13 = { a1.re + a3.re, a1.im + a3.im }, d13 = { a1.re - a3.re, a1.im - a3.im };
            out[k].re = s02.re + s13.re; out[k].im = s02.im + s13.im;
            out[k + 2 * m].re = s02.re - s13.re; out[k + 2 * m].im = s02.im - s13.im;
            /* forward: -i * d13 ; inverse: +i * d13 */
            cpxf r = P->inverse ? (cpxf){ -d13.im, d13.re } : (cpxf){ d13.im, -d13.re };
            out[k + m].re = d02.re + r.re; out[k + m].im = d02.im + r.im;
            out[k + 3 * m].re = d02.re - r.re; out[k + 3 * m].im = d02.im - r.im;
        }
    } else {
        for (int u = 0; u < m; ++u) {
            int k = u;
            for (int q1 = 0; q1 < p; ++q1) { scratch[q1] = out[k]; k += m; }
            k = u;
            for (int q1 = 0; q1 < p; ++q1) {
                long long twidx = 0;
                cpxf acc = scratch[0];
                for (int q = 1; q < p; ++q) {
                    twidx += (long long)fstride * k;
                    if (twidx >= N) twidx %= N;
                    cpxf t = cmulf(scratch[q], tw[twidx]);
                    acc.re += t.re; acc.im += t.im;
                }
                out[k] = acc;
                k += m;
            }
        }
    }
}

static fftf_plan *fftf_create(int n, int inverse, int force_bluestein);
static void fftf_destroy(fftf_plan *P)
{
    if (!P) return;
    free(P->tw); free(P->chirp); free(P->bfft);
    fftf_destroy(P->sub_f); fftf_destroy(P->sub_i);
    free(P);
}

static fftf_plan *fftf_create(int n, int inverse, int force_bluestein)
{
    fftf_plan *P = (fftf_plan *)calloc(1, sizeof(fftf_plan));
    P->n = n; P->inverse = inverse;
    const int pow2 = (n & (n - 1)) == 0;
    if (!pow2 && (force_bluestein || largest_prime_factor(n) > 127)) {
        P->blue = 1;
        int m = 1; while (m < 2 * n - 1) m <<= 1;
        P->m = m;
        P->sub_f = fftf_create(m, 0, 0);
        P->sub_i = fftf_create(m, 1, 0);
        P->chirp = (cpxf *)malloc(sizeof(cpxf) * n);
        const double sgn = inverse ? 1.0 : -1.0;
        for (int k = 0; k < n; ++k) {
            long long k2 = ((long long)k * k) % (2LL * n);
            double a = sgn * M_PI * (double)k2 / n;
            P->chirp[k].re = (float)cos(a); P->chirp[k].im = (float)sin(a);
        }
        cpxf *b = (cpxf *)calloc(m, sizeof(cpxf));
        for (int k = 0; k < n; ++k) {
            cpxf c = { P->chirp[k].re, -P->chirp[k].im };
            b[k] = c;
            if (k) b[m - k] = c;
        }
        P->bfft = (cpxf *)malloc(sizeof(cpxf) * m);
        cpxf *scr = (cpxf *)malloc(sizeof(cpxf) * (P->sub_f->maxp + 1));
        fftf_work(P->bfft, b, 1, P->sub_f->fac, P->sub_f, scr);
        free(scr); free(b);
        return P;
    }
    fftf_factor(n, P->fac, &P->maxp);
    P->tw = (cpxf *)malloc(sizeof(cpxf) * n);
    for (int k = 0; k < n; ++k) {
        double a = (inverse ? 2.0 : -2.0) * M_PI * (double)k / n;
        P->tw[k].re = (float)cos(a); P->tw[k].im = (float)sin(a);
    }
    return P;
}

static int fftf_work_len(const fftf_plan *P) { return P->blue ? 2 * P->m + 8 : P->maxp + 1; }

/* out[0..n) = unnormalised DFT of in[0..n) (sign per plan); work holds fftf_work_len entries */
static void fftf_exec(const fftf_plan *P, const cpxf *in, cpxf *out, cpxf *work)
{
    if (!P->blue) { fftf_work(out, in, 1, P->fac, P, work); return; }
    const int n = P->n, m = P->m;
    cpxf *a = work, *b = work + m, *scr = work + 2 * m;
    for (int k = 0; k < n; ++k) a[k] = cmulf(in[k], P->chirp[k]);
    for (int k = n; k < m; ++k) { a[k].re = 0.f; a[k].im = 0.f; }
    fftf_work(b, a, 1, P->sub_f->fac, P->sub_f, scr);
    for (int k = 0; k < m; ++k) b[k] = cmulf(b[k], P->bfft[k]);
    fftf_work(a, b, 1, P->sub_i->fac, P->sub_i, scr);
    const float s = 1.0f / (float)m;
    for (int k = 0; k < n; ++k) {
        cpxf v = { a[k].re * s, a[k].im * s };
        out[k] = cmulf(v, P->chirp[k]);
    }
}

/* one pass of the reference's dst() (IMP.cpp:1694-1811) over `rows` rows of length n held in `src` (row-major,
 * rows x n): odd extension, complex transform, imaginary part of bins 1..n written TRANSPOSED into dst (n x rows). */
static void dstf_pass(const fftf_plan *P, const float *src, float *dst, int rows, int n, int nthreads)
{
    const int N = 2 * n + 2;
#pragma omp parallel num_threads(nthreads)
    {
        cpxf *t = (cpxf *)malloc(sizeof(cpxf) * N), *o = (cpxf *)malloc(sizeof(cpxf) * N);
        cpxf *work = (cpxf *)malloc(sizeof(cpxf) * fftf_work_len(P));
#pragma omp for schedule(static)
        for (int r = 0; r < rows; ++r) {
            const float *s = src + (size_t)r * n;
            t[0].re = t[0].im = 0.f; t[n + 1].re = t[n + 1].im = 0.f;
            for (int j = 0; j < n; ++j) {
                t[1 + j].re = s[j]; t[1 + j].im = 0.f;
                t[n + 2 + j].re = -s[n - 1 - j]; t[n + 2 + j].im = 0.f;
            }
            fftf_exec(P, t, o, work);
            for (int j = 0; j < n; ++j) dst[(size_t)j * rows + r] = o[j + 1].im;
        }
        free(t); free(o); free(work);
    }
}

/* internals: 0 = double transforms (sco_solve_dst2), 1 = float32 mixed radix, 2 = float32 Bluestein */
SCO_API void sco_solve_dst3(const float *g, int w, int h, int C, float *u, int nthreads, int exact_den, int internals)
{
    if (internals == 0) { sco_solve_dst2(g, w, h, C, u, nthreads, exact_den); return; }
    if (nthreads < 1) nthreads = 1;
    const int fb = internals == 2;
    fftf_plan *pwf = fftf_create(2 * w + 2, 0, fb), *phf = fftf_create(2 * h + 2, 0, fb);
    fftf_plan *pwi = fftf_create(2 * w + 2, 1, fb), *phi = fftf_create(2 * h + 2, 1, fb);
    float *fx = (float *)malloc(sizeof(float) * w), *fy = (float *)malloc(sizeof(float) * h);
    const double PIf = (double)3.14159265358979323846f;     /* seamlessClone_imp.h:17 */
    for (int i = 0; i < w; ++i) fx[i] = (float)(2.0 * cos(PIf / (w + 1.0) * (i + 1.0)));   /* IMP.cpp:596-599 */
    for (int j = 0; j < h; ++j) fy[j] = (float)(2.0 * cos(PIf / (h + 1.0) * (j + 1.0)));
    const size_t pl = (size_t)w * h;
    float *a = (float *)malloc(sizeof(float) * pl), *b = (float *)malloc(sizeof(float) * pl);
    const float scale = 1.0f / (float)((w * 2 + 2) * (h * 2 + 2));                        /* IMP.cpp:1893 */
    for (int c = 0; c < C; ++c) {
        dstf_pass(pwf, g + c * pl, b, h, w, nthreads);        /* rows of length w -> b[x][y] */
        dstf_pass(phf, b, a, w, h, nthreads);                 /* rows of length h -> a[y][x] */
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                if (exact_den) a[(size_t)y * w + x] = (float)((double)a[(size_t)y * w + x] /
                    (2.0 * cos(M_PI * (x + 1.0) / (w + 1.0)) + 2.0 * cos(M_PI * (y + 1.0) / (h + 1.0)) - 4.0));
                else a[(size_t)y * w + x] /= (fx[x] + fy[y] - 4.0f);                      /* IMP.cpp:1651-1653 */
            }
        dstf_pass(pwi, a, b, h, w, nthreads);
        dstf_pass(phi, b, a, w, h, nthreads);
        for (size_t i = 0; i < pl; ++i) u[c * pl + i] = a[i] * scale;
    }
    free(a); free(b); free(fx); free(fy);
    fftf_destroy(pwf); fftf_destroy(phf); fftf_destroy(pwi); fftf_destroy(phi);
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
SCO_API int sco_seamless_clone4(const uint8_t *patch, int pw, int ph, int pstride, uint8_t *dst, int dw, int dh, int dstride,
                                const uint8_t *mask, int mw, int mh, int mstride, int cx, int cy, int nthreads, int exact_den,
                                int internals, int grey);
SCO_API int sco_seamless_clone3(const uint8_t *patch, int pw, int ph, int pstride,
                                uint8_t *dst, int dw, int dh, int dstride,
                                const uint8_t *mask, int mw, int mh, int mstride,
                                int cx, int cy, int nthreads, int exact_den, int internals)
{
    return sco_seamless_clone4(patch, pw, ph, pstride, dst, dw, dh, dstride, mask, mw, mh, mstride, cx, cy, nthreads, exact_den,
                               internals, 0);
}

SCO_API int sco_seamless_clone4(const uint8_t *patch, int pw, int ph, int pstride,
                                uint8_t *dst, int dw, int dh, int dstride,
                                const uint8_t *mask, int mw, int mh, int mstride,
                                int cx, int cy, int nthreads, int exact_den, int internals, int grey)
{
    if (pw != mw || ph != mh) return -2;
    int geo[6];
    uint8_t *M = (uint8_t *)malloc((size_t)mw * mh);
    int rc = sco_mask_stage2(mask, mw, mh, mstride, cx, cy, geo, M, grey);
    if (rc) { free(M); return rc; }
    const int W = geo[2], H = geo[3], w = W - 2, h = H - 2;
    const size_t plane = (size_t)W * H;
    float *B = (float *)malloc(sizeof(float) * plane * 3);
    float *lap = (float *)malloc(sizeof(float) * plane * 3);
    rc = sco_build_rhs2(dst, dw, dh, dstride, patch, pw, ph, pstride, M, geo, B, lap, grey);
    if (rc == 0 && w > 0 && h > 0) {
        float *g = (float *)malloc(sizeof(float) * (size_t)w * h * 3);
        sco_fold(B, lap, W, H, g);
        sco_solve_dst3(g, w, h, 3, g, nthreads, exact_den, internals);
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

SCO_API int sco_seamless_clone2(const uint8_t *patch, int pw, int ph, int pstride,
                                uint8_t *dst, int dw, int dh, int dstride,
                                const uint8_t *mask, int mw, int mh, int mstride,
                                int cx, int cy, int nthreads, int exact_den)
{
    return sco_seamless_clone3(patch, pw, ph, pstride, dst, dw, dh, dstride, mask, mw, mh, mstride,
                               cx, cy, nthreads, exact_den, 0);
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
