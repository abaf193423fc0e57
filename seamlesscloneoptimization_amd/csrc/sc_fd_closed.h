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

struct FdPair {          // eigenpair k of one 1-D operator
    double ang;          // theta (kind 0), phi = pi - theta (kind 1) or t (kind 2)
    double lam;          // eigenvalue
    double inv_norm;     // 1 / |E v|
    int kind;            // 0: v_i = sin(i theta); 1: v_i = (-1)^(i+1) sin(i phi); 2: v_i = (-1)^(i+1) sinh(i t) / sinh(n t)
};

// lower half: g(theta) as above, with d - 2 + 2 cos = d - 4 sin^2(theta/2)
__host__ __device__ inline double fd_g_lo(double th, int n, double cw, double d)
{
    const double sh = sin(0.5 * th);
    return cw * sin((n - 1) * th) - (d - 4.0 * sh * sh) * sin(n * th);
}
// upper half in phi = pi - theta: (-1)^n g = cw sin((n-1) phi) + (d - 4 + 4 sin^2(phi/2)) sin(n phi)
__host__ __device__ inline double fd_g_hi(double ph, int n, double cw, double d)
{
    const double sh = sin(0.5 * ph);
    return cw * sin((n - 1) * ph) + (d - 4.0 + 4.0 * sh * sh) * sin(n * ph);
}
// hyperbolic branch divided by sinh(n t): cw sinh((n-1)t)/sinh(nt) + d - 4 - 4 sinh^2(t/2)
__host__ __device__ inline double fd_g_hyp(double t, int n, double cw, double d)
{
    const double r = exp(-t) * (expm1(-2.0 * (n - 1) * t) / expm1(-2.0 * n * t));
    const double sh = sinh(0.5 * t);
    return cw * r + d - 4.0 - 4.0 * sh * sh;
}

// component i (1-based, 1 .. n) of the UNNORMALISED eigenvector of pair p
__host__ __device__ inline double fd_component(const FdPair &p, int i, int n)
{
    if (p.kind == 0) return sin(i * p.ang);
    const double sgn = (i & 1) ? 1.0 : -1.0;
    if (p.kind == 1) return sgn * sin(i * p.ang);
    // sinh(i t) / sinh(n t) = e^{-(n-i) t} (1 - e^{-2 i t}) / (1 - e^{-2 n t})
    return sgn * exp(-(double)(n - i) * p.ang) * (expm1(-2.0 * i * p.ang) / expm1(-2.0 * n * p.ang));
}

// eigenpair k (0 .. n-1) of the operator (n, cw, d)
__host__ __device__ inline FdPair fd_pair(int k, int n, double cw, double d)
{
    const double PI = 3.14159265358979323846;
    FdPair p;
    if (n == 1) { p.ang = 0.5 * PI; p.kind = 0; p.lam = -d; p.inv_norm = sqrt(cw); return p; }   // v = (1); E v = 1/sqrt(cw)
    const double s = cw * (n - 1) + (d - 4.0) * n;      // sign of (-1)^n g just left of pi
    double lo, hi, flo;
    if (k == n - 1 && s > 0.0) {                        // the n-th eigenvalue lies below -4
        p.kind = 2;
        lo = 0.0; hi = 2.0; flo = 1.0;                  // fd_g_hyp -> s / n > 0 at 0+, < 0 at 2
        for (int it = 0; it < 60; ++it) {
            const double mid = 0.5 * (lo + hi), f = fd_g_hyp(mid, n, cw, d);
            if ((f > 0.0) == (flo > 0.0)) lo = mid; else hi = mid;
        }
        p.ang = 0.5 * (lo + hi);
        const double sh = sinh(0.5 * p.ang);
        p.lam = -4.0 - 4.0 * sh * sh;
    } else if (2 * (k + 1) <= n) {                      // the interval lies in the lower half: theta itself
        p.kind = 0;
        lo = (k == 0) ? PI / (4.0 * n) : k * PI / n;    // below the smallest root (~ pi / (n + alpha)); g < 0 there
        hi = (k + 1) * PI / n;
        flo = (k == 0) ? -1.0 : ((k & 1) ? 1.0 : -1.0); // g(j pi / n) = cw (-1)^(j+1) sin(j pi / n)
        for (int it = 0; it < 50; ++it) {
            const double mid = 0.5 * (lo + hi), f = fd_g_lo(mid, n, cw, d);
            if ((f > 0.0) == (flo > 0.0)) lo = mid; else hi = mid;
        }
        p.ang = 0.5 * (lo + hi);
        const double sh = sin(0.5 * p.ang);
        p.lam = -4.0 * sh * sh;
    } else {                                            // upper half: phi in ((n-k-1) pi / n, (n-k) pi / n)
        p.kind = 1;
        const int j = n - k - 1;                        // 0 for the last interval
        lo = j * PI / n; hi = (j + 1) * PI / n;
        // fd_g_hi(j pi / n) = cw sin((n-1) j pi / n) = cw (-1)^(j+1) sin(j pi / n) for j >= 1; just right of 0 its sign is s's (< 0 here)
        flo = (j == 0) ? -1.0 : ((j & 1) ? 1.0 : -1.0);
        for (int it = 0; it < 50; ++it) {
            const double mid = 0.5 * (lo + hi), f = fd_g_hi(mid, n, cw, d);
            if ((f > 0.0) == (flo > 0.0)) lo = mid; else hi = mid;
        }
        p.ang = 0.5 * (lo + hi);
        const double sh = sin(0.5 * p.ang);
        p.lam = -4.0 + 4.0 * sh * sh;
    }
    double ss = 0.0;                                    // |E v|^2, term by term (a closed form cancels for small angles)
    for (int i = 1; i <= n; ++i) {
        const double v = fd_component(p, i, n);
        ss += (i == n) ? v * v / cw : v * v;
    }
    p.inv_norm = 1.0 / sqrt(ss);
    return p;
}

} // namespace sc
