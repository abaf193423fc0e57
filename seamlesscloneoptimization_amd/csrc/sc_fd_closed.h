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
// (mag: the sum of the magnitudes of g's terms -- |g| <= a few eps x mag is a residual at rounding level)
__host__ __device__ __attribute__((always_inline)) inline void fd_g_lo_d(double tau, int n, double cw, double d, double &g, double &dg, double &mag)
{
    double S, C, s, c;
    fd_sincospi(n * tau, &S, &C);
    fd_sincospi(tau, &s, &c);
    const double sh = fd_sinpi(0.5 * tau), q = d - 4.0 * sh * sh;           // d - 2 + 2 cos theta, without the cancellation
    g = cw * (S * c - C * s) - q * S;
    dg = cw * (n - 1) * (C * c + S * s) - q * n * C + 2.0 * s * S;
    mag = cw * (fabs(S * c) + fabs(C * s)) + fabs(q * S);
}
// ... and of the upper-half equation in phi = pi - theta = pi tau
__host__ __device__ __attribute__((always_inline)) inline void fd_g_hi_d(double tau, int n, double cw, double d, double &g, double &dg, double &mag)
{
    double S, C, s, c;
    fd_sincospi(n * tau, &S, &C);
    fd_sincospi(tau, &s, &c);
    const double sh = fd_sinpi(0.5 * tau), q = d - 4.0 + 4.0 * sh * sh;     // d - 2 - 2 cos phi
    g = cw * (S * c - C * s) + q * S;
    dg = cw * (n - 1) * (C * c + S * s) + q * n * C + 2.0 * s * S;
    mag = cw * (fabs(S * c) + fabs(C * s)) + fabs(q * S);
}

// either of the two by a flag, in one instruction stream: they differ in two signs
__host__ __device__ __attribute__((always_inline)) inline void fd_g_trig_d(double tau, int n, double cw, double d, bool upper, double &g, double &dg, double &mag)
{
    double S, C, s, c;
    fd_sincospi(n * tau, &S, &C);
    fd_sincospi(tau, &s, &c);
    const double sh = fd_sinpi(0.5 * tau);
    const double q = upper ? d - 4.0 + 4.0 * sh * sh : -(d - 4.0 * sh * sh);      // (+q S in the upper half, -q S in the lower)
    g = cw * (S * c - C * s) + q * S;
    dg = cw * (n - 1) * (C * c + S * s) + q * n * C + 2.0 * s * S;
    mag = cw * (fabs(S * c) + fabs(C * s)) + fabs(q * S);
}

// ... and of the hyperbolic-branch equation fd_g_hyp (divided by sinh(n t)), in t:  with r = sinh((n-1)t) / sinh(nt),
// r' = (n-1) cosh((n-1)t) / sinh(nt) - n r coth(nt), every quotient in the e^{-x} form that neither overflows nor cancels
__host__ __device__ __attribute__((always_inline)) inline void fd_g_hyp_d(double t, int n, double cw, double d, double &g, double &dg, double &mag)
{
    const double e1 = exp(-t), EA = expm1(-2.0 * (n - 1) * t), EB = expm1(-2.0 * n * t);
    const double r = e1 * (EA / EB);
    const double sh = sinh(0.5 * t), ch = sqrt(1.0 + sh * sh);
    g = cw * r + d - 4.0 - 4.0 * sh * sh;
    const double cA = e1 * ((2.0 + EA) / -EB), cB = (2.0 + EB) / -EB;
    dg = cw * ((n - 1) * cA - n * r * cB) - 4.0 * sh * ch;
    mag = cw * r + fabs(d - 4.0) + 4.0 * sh * sh;
}

// The root of g in (lo, hi) -- angles in units of pi --, where g has the sign of flo at lo and the other one at hi: Newton steps kept
// inside the bracket (a step that leaves it is replaced by the midpoint, so the worst case is the bisection rounds 1-4 ran: 50 steps;
// Newton needs 5-8).  KIND 0: fd_g_lo, 1: fd_g_hi, 2: fd_g_hyp (x0: where to start; the midpoint otherwise), 3: fd_g_lo or fd_g_hi by `upper` --
// ONE loop for the lanes of a wave whose intervals lie in either half (the device build: two loops run one after the other).
template <int KIND>
__host__ __device__ __attribute__((always_inline)) inline double fd_root(double lo, double hi, double flo, int n, double cw, double d, double x0 = -1.0, bool upper = false)
{
    const double INV_PI = 0.31830988618379067154;
    double x = (x0 > lo && x0 < hi) ? x0 : 0.5 * (lo + hi);
    for (int it = 0; it < 60; ++it) {
        double g, dg, mag;
#if defined(FD_COUNT_ITERS)
        ++g_iters;
#endif
        if (KIND == 0) fd_g_lo_d(x, n, cw, d, g, dg, mag); else if (KIND == 1) fd_g_hi_d(x, n, cw, d, g, dg, mag); else if (KIND == 2) fd_g_hyp_d(x, n, cw, d, g, dg, mag); else fd_g_trig_d(x, n, cw, d, upper, g, dg, mag);
        const double step = (KIND == 2 ? 1.0 : INV_PI) * (g / dg);      // dg is d/dtheta; the step in tau = theta / pi (KIND 2: in t itself)
        // converged when the residual is at rounding level or the Newton step is a few units in the last place (tested BEFORE the
        // bracket logic: at the root the sign of g is noise, and a step of that size towards the freshly moved bracket end would be
        // "outside" and cost a bisection from afar; where g' is small -- roots near 0 -- the step never gets below the noise of g / g',
        // and waiting for the bracket to collapse instead took up to 17 evaluations where 6 do)
        if (fabs(g) <= 8e-16 * mag || fabs(step) <= 9e-16 * x) return x - step;
        if ((g > 0.0) == (flo > 0.0)) lo = x; else hi = x;
        double xn = x - step;
        if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
        if (hi - lo <= 4.5e-16 * hi) return xn;
        x = xn;
    }
    return x;
}

// eigenpair k (0 .. n-1) of the operator (n, cw, d)
// hyp (may be nullptr): receives the n components of the hyperbolic pair's unnormalised vector (kind 2; fd_component's values by the
// recurrence) -- the device build reads them from there: fd_component's exp / expm1 branch, taken by ONE column of every matrix,
// made every wave of the fill loop execute both branches (20 of the build's 32 us)
__host__ __device__ __attribute__((always_inline)) inline FdPair fd_pair(int k, int n, double cw, double d, double *hyp = nullptr)
{
    FdPair p;
    if (n == 1) { p.ang = 0.5; p.kind = 0; p.lam = -d; p.inv_norm = sqrt(cw); return p; }   // v = (1); E v = 1/sqrt(cw)
    const double s = cw * (n - 1) + (d - 4.0) * n;      // sign of (-1)^n g just left of pi
    double lo, hi, flo;
    if (k == n - 1 && s > 0.0) {                        // the n-th eigenvalue lies below -4
        p.kind = 2;
        lo = 0.0; hi = 2.0; flo = 1.0;                  // fd_g_hyp -> s / n > 0 at 0+, < 0 at 2
        // Newton from the smaller of two estimates: for n t large r -> e^{-t} and the equation is a quadratic in z = e^t,
        // z^2 - (d - 2) z - (cw - 1) = 0 (the true root lies just below its root: r < e^{-t}); for n t small
        // g ~ s / n - K t^2 with K = cw (n-1)(2n-1) / (6n) + 1.  5-6 evaluations; the regula falsi on (0, 2) of rounds 4-5 took 11 on
        // average and up to 42 -- in ONE thread of the build, while every other one had long finished (31 of k_fd_build_rag's 46 us)
        const double b = d - 2.0, z = 0.5 * (b + sqrt(b * b + 4.0 * (cw - 1.0)));
        const double K = cw * (n - 1) * (2.0 * n - 1.0) / (6.0 * n) + 1.0, tT = sqrt(s / (n * K));
        const double t0 = (z > 1.0) ? log(z) : 2.0;
        p.ang = fd_root<2>(lo, hi, flo, n, cw, d, t0 < tT ? t0 : tT);
        const double sh = sinh(0.5 * p.ang);
        p.lam = -4.0 - 4.0 * sh * sh;
    } else {
        const bool upper = 2 * (k + 1) > n;
        if (!upper) {                                   // the interval lies in the lower half: theta itself
            p.kind = 0;
            lo = (k == 0) ? 1.0 / (4.0 * n) : (double)k / n;  // (units of pi) below the smallest root (~ pi / (n + alpha)); g < 0 there
            hi = (double)(k + 1) / n;
            flo = (k == 0) ? -1.0 : ((k & 1) ? 1.0 : -1.0); // g(j pi / n) = cw (-1)^(j+1) sin(j pi / n)
        } else {                                        // upper half: phi in ((n-k-1) pi / n, (n-k) pi / n)
            p.kind = 1;
            const int j = n - k - 1;                    // 0 for the last interval
            lo = (double)j / n; hi = (double)(j + 1) / n;   // (units of pi)
            // fd_g_hi(j pi / n) = cw sin((n-1) j pi / n) = cw (-1)^(j+1) sin(j pi / n) for j >= 1; just right of 0 its sign is s's (< 0 here)
            flo = (j == 0) ? -1.0 : ((j & 1) ? 1.0 : -1.0);
        }
        p.ang = fd_root<3>(lo, hi, flo, n, cw, d, -1.0, upper);
        const double sh = fd_sinpi(0.5 * p.ang);
        p.lam = upper ? -4.0 + 4.0 * sh * sh : -4.0 * sh * sh;
    }
    double ss = 0.0;                                    // |E v|^2, term by term (a closed form cancels for small angles)
    if (p.kind == 2) {
        // sinh(i t) / sinh(n t) by the same recurrence, s_{i+1} = 2 cosh(t) s_i - s_{i-1}, from s_0 = 0 and s_1 in the form that
        // neither overflows nor cancels: forwards the growing solution is the stable one (relative error ~ n eps).  n evaluations of
        // exp + 2 expm1 in ONE thread were 40 us of a 46-us build (every other thread had long finished): the launch a size class of small
        // ROIs waits for in front of its first k_mg_tail.
        const double two_ch = 2.0 + 4.0 * sinh(0.5 * p.ang) * sinh(0.5 * p.ang);      // 2 cosh t
        double sm = 0.0, sc = fd_component(p, 1, n);        // (signs square away)
        for (int i = 1; i <= n; ++i) {
            if (hyp) hyp[i - 1] = (i & 1) ? sc : -sc;
            ss += (i == n) ? sc * sc / cw : sc * sc;
            const double nx = two_ch * sc - sm;
            sm = sc; sc = nx;
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
