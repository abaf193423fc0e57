// sc_fd_closed.h -- eigen-decomposition of a 1-D multigrid level operator in closed form (host and device).
//
// The bottom kernel solves its first level exactly by fast diagonalisation (sc_mg_device.h, MGBottomArgs) and needs the
// eigenvectors of the two tridiagonal 1-D operators
//     T = tridiag(1, -2, 1) with the LAST row (cw, -d)          (MGDim: cw = cw_last = 2/(1+alpha), d = d_last = 2/alpha)
// for every new ROI size.  The reference rebuilds its per-size tables on the device inside every call (initDSTMatrix_kernel,
// seamlessClone_imp.cpp:569-603, launched from init_resize :636-655) precisely because building them on the host was its
// bottleneck (PDF p14: 18.8 -> 14.1 ms); round 3 of this library ran an O(n^3) implicit-QL eigen-solve on the host for every
// new size.  T is Toeplitz except for its last row, so its eigenvectors are known up to one scalar each:
//     rows 1 .. n-1 (with v_0 = 0) force   v_i = sin(i theta),        lambda = 2 cos(theta) - 2 = -4 sin^2(theta/2),
//     the last row then is the scalar equation
//         g(theta) = cw sin((n-1) theta) - (d - 2 + 2 cos(theta)) sin(n theta) = 0.
// g alternates in sign at theta = j pi / n (there sin(n theta) = 0 and g = cw (-1)^(j+1) sin(j pi / n)), is negative just right
// of 0, and T has n distinct real eigenvalues: exactly ONE root in every interval (k pi / n, (k+1) pi / n), k = 0 .. n-2, and the
// n-th either in the last interval or -- when s = cw (n-1) + (d-4) n > 0, i.e. a short last interval, alpha < 0.7071 -- below -4:
//     v_i = (-1)^(i+1) sinh(i t),  lambda = -2 - 2 cosh(t),  cw sinh((n-1) t) + (d - 2 - 2 cosh(t)) sinh(n t) = 0.
// Roots in the upper half are found in phi = pi - theta (the same function with every sign folded in), so a root next to pi
// keeps its relative accuracy.  One thread per eigenvalue: 50 bisection steps of two sincos each; no iteration over the matrix.
//
// E T E^-1 is symmetric for E = diag(1, .., 1, 1/sqrt(cw)); with q_k = E v_k / |E v_k|:  T = V L V^-1,  V = E^-1 Q,  V^-1 = Q^T E
// (the same objects sc_multigrid.cpp's QL-based FD1 holds; sc_hip_selftest_host compares the two).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace sc {

// sin / cos of pi x.  On the device: sincospi -- its argument reduction is exact and needs no table, where sin(double) carries the
// Payne-Hanek reduction's private arrays: 100 bytes of scratch per lane made every build kernel wait ~100 us for the runtime's scratch
// set-up (k_fd_build_rag: 117 us per launch for ~10 us of arithmetic, measured in round 5).  On the host (the self-test): libm.
__host__ __device__ __attribute__((always_inline)) inline void fd_sincospi(double x, double *s, double *c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sincospi(x, s, c);
#else
    sincos(3.14159265358979323846 * x, s, c);
#endif
}
__host__ __device__ __attribute__((always_inline)) inline double fd_sinpi(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return sinpi(x);
#else
    return sin(3.14159265358979323846 * x);
#endif
}

struct FdPair {          // eigenpair k of one 1-D operator
    double ang;          // theta / pi (kind 0), phi / pi = 1 - theta / pi (kind 1) or t (kind 2)
    double lam;          // eigenvalue
    double inv_norm;     // 1 / |E v|
    int kind;            // 0: v_i = sin(i theta); 1: v_i = (-1)^(i+1) sin(i phi); 2: v_i = (-1)^(i+1) sinh(i t) / sinh(n t)
};

// hyperbolic branch divided by sinh(n t): cw sinh((n-1)t)/sinh(nt) + d - 4 - 4 sinh^2(t/2)
__host__ __device__ __attribute__((always_inline)) inline double fd_g_hyp(double t, int n, double cw, double d)
{
    const double r = exp(-t) * (expm1(-2.0 * (n - 1) * t) / expm1(-2.0 * n * t));
    const double sh = sinh(0.5 * t);
    return cw * r + d - 4.0 - 4.0 * sh * sh;
}

// component i (1-based, 1 .. n) of the UNNORMALISED eigenvector of pair p
__host__ __device__ __attribute__((always_inline)) inline double fd_component(const FdPair &p, int i, int n)
{
    if (p.kind == 0) return fd_sinpi(i * p.ang);
    const double sgn = (i & 1) ? 1.0 : -1.0;
    if (p.kind == 1) return sgn * fd_sinpi(i * p.ang);
    // sinh(i t) / sinh(n t) = e^{-(n-i) t} (1 - e^{-2 i t}) / (1 - e^{-2 n t})
    return sgn * exp(-(double)(n - i) * p.ang) * (expm1(-2.0 * i * p.ang) / expm1(-2.0 * n * p.ang));
}

// g and dg/dtheta of the lower-half equation at theta = pi tau, from two sincos: with S = sin(n theta), C = cos(n theta),
// s = sin theta, c = cos theta:  sin((n-1) theta) = S c - C s,  cos((n-1) theta) = C c + S s
__host__ __device__ __attribute__((always_inline)) inline void fd_g_lo_d(double tau, int n, double cw, double d, double &g, double &dg)
{
    double S, C, s, c;
    fd_sincospi(n * tau, &S, &C);
    fd_sincospi(tau, &s, &c);
    const double sh = fd_sinpi(0.5 * tau), q = d - 4.0 * sh * sh;           // d - 2 + 2 cos theta, without the cancellation
    g = cw * (S * c - C * s) - q * S;
    dg = cw * (n - 1) * (C * c + S * s) - q * n * C + 2.0 * s * S;
}
// ... and of the upper-half equation in phi = pi - theta = pi tau
__host__ __device__ __attribute__((always_inline)) inline void fd_g_hi_d(double tau, int n, double cw, double d, double &g, double &dg)
{
    double S, C, s, c;
    fd_sincospi(n * tau, &S, &C);
    fd_sincospi(tau, &s, &c);
    const double sh = fd_sinpi(0.5 * tau), q = d - 4.0 + 4.0 * sh * sh;     // d - 2 - 2 cos phi
    g = cw * (S * c - C * s) + q * S;
    dg = cw * (n - 1) * (C * c + S * s) + q * n * C + 2.0 * s * S;
}

// The root of g in (lo, hi) -- angles in units of pi --, where g has the sign of flo at lo and the other one at hi: Newton steps kept
// inside the bracket (a step that leaves it is replaced by the midpoint, so the worst case is the bisection rounds 1-4 ran: 50 steps;
// Newton needs 5-8).  KIND 0: fd_g_lo, 1: fd_g_hi.
template <int KIND>
__host__ __device__ __attribute__((always_inline)) inline double fd_root(double lo, double hi, double flo, int n, double cw, double d)
{
    const double INV_PI = 0.31830988618379067154;
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 60; ++it) {
        double g, dg;
#if defined(FD_COUNT_ITERS)
        ++g_iters;
#endif
        if (KIND == 0) fd_g_lo_d(x, n, cw, d, g, dg); else fd_g_hi_d(x, n, cw, d, g, dg);
        const double step = INV_PI * (g / dg);              // dg is d/dtheta; the step in tau = theta / pi
        // converged when the Newton step is a few units in the last place (tested BEFORE the bracket logic: at the root the sign of g
        // is noise, and a step of that size towards the freshly moved bracket end would be "outside" and cost a bisection from afar)
        if (fabs(step) <= 9e-16 * x) return x - step;
        if ((g > 0.0) == (flo > 0.0)) lo = x; else hi = x;
        double xn = x - step;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (hi - lo <= 4.5e-16 * hi) return xn;
        x = xn;
    }
    return x;
}

// eigenpair k (0 .. n-1) of the operator (n, cw, d)
__host__ __device__ __attribute__((always_inline)) inline FdPair fd_pair(int k, int n, double cw, double d)
{
    FdPair p;
    if (n == 1) { p.ang = 0.5; p.kind = 0; p.lam = -d; p.inv_norm = sqrt(cw); return p; }   // v = (1); E v = 1/sqrt(cw)
    const double s = cw * (n - 1) + (d - 4.0) * n;      // sign of (-1)^n g just left of pi
    double lo, hi, flo;
    if (k == n - 1 && s > 0.0) {                        // the n-th eigenvalue lies below -4
        p.kind = 2;
        lo = 0.0; hi = 2.0; flo = 1.0;                  // fd_g_hyp -> s / n > 0 at 0+, < 0 at 2
        // regula falsi with the Illinois correction on the bracket (fd_g_hyp is smooth and monotone there): ~10 evaluations
        // (the false-position estimate converges from one side: the BRACKET need not shrink to a point -- the estimate is the answer,
        // and it has converged when it stops moving)
        double fl = s / n, fh = fd_g_hyp(hi, n, cw, d), est = 1.0, prev = -1.0;
        int side = 0;
        for (int it = 0; it < 100; ++it) {
            est = (lo * fh - hi * fl) / (fh - fl);
            if (!(est > lo && est < hi)) est = 0.5 * (lo + hi);
            if (fabs(est - prev) <= 4.5e-16 * est) break;
            prev = est;
            const double f = fd_g_hyp(est, n, cw, d);
            if (f == 0.0) break;
            if ((f > 0.0) == (flo > 0.0)) { lo = est; fl = f; if (side == -1) fh *= 0.5; side = -1; }
            else { hi = est; fh = f; if (side == 1) fl *= 0.5; side = 1; }
        }
        p.ang = est;
        const double sh = sinh(0.5 * p.ang);
        p.lam = -4.0 - 4.0 * sh * sh;
    } else if (2 * (k + 1) <= n) {                      // the interval lies in the lower half: theta itself
        p.kind = 0;
        lo = (k == 0) ? 1.0 / (4.0 * n) : (double)k / n;  // (units of pi) below the smallest root (~ pi / (n + alpha)); g < 0 there
        hi = (double)(k + 1) / n;
        flo = (k == 0) ? -1.0 : ((k & 1) ? 1.0 : -1.0); // g(j pi / n) = cw (-1)^(j+1) sin(j pi / n)
        p.ang = fd_root<0>(lo, hi, flo, n, cw, d);
        const double sh = fd_sinpi(0.5 * p.ang);
        p.lam = -4.0 * sh * sh;
    } else {                                            // upper half: phi in ((n-k-1) pi / n, (n-k) pi / n)
        p.kind = 1;
        const int j = n - k - 1;                        // 0 for the last interval
        lo = (double)j / n; hi = (double)(j + 1) / n;   // (units of pi)
        // fd_g_hi(j pi / n) = cw sin((n-1) j pi / n) = cw (-1)^(j+1) sin(j pi / n) for j >= 1; just right of 0 its sign is s's (< 0 here)
        flo = (j == 0) ? -1.0 : ((j & 1) ? 1.0 : -1.0);
        p.ang = fd_root<1>(lo, hi, flo, n, cw, d);
        const double sh = fd_sinpi(0.5 * p.ang);
        p.lam = -4.0 + 4.0 * sh * sh;
    }
    double ss = 0.0;                                    // |E v|^2, term by term (a closed form cancels for small angles)
    if (p.kind == 2) {
        for (int i = 1; i <= n; ++i) {
            const double v = fd_component(p, i, n);
            ss += (i == n) ? v * v / cw : v * v;
        }
    } else {
        // sin(i a) by the three-term recurrence s_{i+1} = 2 cos(a) s_i - s_{i-1} (the signs (-1)^(i+1) of kind 1 square away): n
        // multiply-adds in double instead of n sines, relative error ~ n^2 eps -- 1e-12 at n = 127, far below the float matrices
        double s1, c1;
        fd_sincospi(p.ang, &s1, &c1);
        const double two_c = 2.0 * c1;
        double sm = 0.0, sc = s1;                       // sin(0), sin(a)
        for (int i = 1; i <= n; ++i) {
            ss += (i == n) ? sc * sc / cw : sc * sc;
            const double nx = two_c * sc - sm;
            sm = sc; sc = nx;
        }
    }
    p.inv_norm = 1.0 / sqrt(ss);
    return p;
}

} // namespace sc
