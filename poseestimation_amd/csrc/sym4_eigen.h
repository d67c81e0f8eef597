// Eigenvector of the smallest eigenvalue of a symmetric 4 x 4 matrix by cyclic Jacobi rotations (round-robin
// order: the three pairs of disjoint planes), all in registers with compile-time indices.  Shared by the DLT and RANSAC triangulation kernels, where the matrix is
// the Gram matrix A^T A of a DLT design matrix and the eigenvector is the right singular vector of A's smallest
// singular value.
#pragma once

#include "mvx_common.h"

namespace mvx {

constexpr int kMaxJacobiSweeps = 16;

// 1/x and 1/sqrt(x) to float64 accuracy from the hardware estimates (v_rcp_f64 / v_rsq_f64, about 2^-26 relative
// error, refined by two Newton steps each): a much shorter dependent chain than the IEEE division and square
// root routines, which matters because one lane walks 3 rotation pairs x several sweeps serially.
__device__ __forceinline__ double rcp_refined(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double rsqrt_refined(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x, y * y, 1.5);
    y = y * fma(-0.5 * x, y * y, 1.5);
    return y;
}

// Parameters of the Jacobi rotation that annihilates a_pq: t = tan(angle) is the smaller root of
// t^2 + 2 t theta - 1 = 0, theta = (a_qq - a_pp) / (2 a_pq); for |theta| beyond 1e100 the square would overflow
// and t = 1 / (2 theta) (also the limit for theta = inf).  Branch-free, so that the chains of two independent
// rotations interleave: a_pq == 0 or a NaN gives the identity (t = 0, c = 1, s = 0).
struct Rotation {
    double t, c, s;
    bool on;
};
__device__ __forceinline__ Rotation rotation_for(double app, double aqq, double apq) {
    const double theta = (aqq - app) * 0.5 * rcp_refined(apq);
    const double ath = fabs(theta);
    const double x = fma(theta, theta, 1.0);
    const double t_small = rcp_refined(ath + x * rsqrt_refined(x));
    const double t_large = 0.5 * rcp_refined(ath);
    double t = ath < 1e100 ? t_small : t_large;
    t = theta >= 0.0 ? t : -t;
    Rotation r;
    r.on = (apq != 0.0) && (t == t);
    r.t = r.on ? t : 0.0;
    r.c = rsqrt_refined(fma(r.t, r.t, 1.0));
    r.s = r.t * r.c;
    return r;
}

// Applies the rotation in the (P, Q) plane; a is kept full and symmetric, v accumulates the rotations (columns
// of v are the eigenvectors).
template <int P, int Q>
__device__ __forceinline__ void apply_rotation(const Rotation& r, double (&a)[4][4], double (&v)[4][4]) {
    const double apq = a[P][Q];
    a[P][P] = fma(-r.t, apq, a[P][P]);
    a[Q][Q] = fma(r.t, apq, a[Q][Q]);
    a[P][Q] = r.on ? 0.0 : apq;
    a[Q][P] = a[P][Q];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k != P && k != Q) {
            const double akp = a[k][P], akq = a[k][Q];
            const double np_ = r.c * akp - r.s * akq;
            const double nq_ = r.s * akp + r.c * akq;
            a[k][P] = np_;
            a[P][k] = np_;
            a[k][Q] = nq_;
            a[Q][k] = nq_;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double vkp = v[k][P], vkq = v[k][Q];
        v[k][P] = r.c * vkp - r.s * vkq;
        v[k][Q] = r.s * vkp + r.c * vkq;
    }
}

// Two rotations in disjoint planes.  The first does not touch a_rr, a_ss or a_rs, so both parameter sets come
// from the same matrix and their (long, dependent) evaluations overlap; applying them one after the other is
// then exactly the sequential Jacobi step.
template <int P, int Q, int R, int S>
__device__ __forceinline__ void jacobi_pair(double (&a)[4][4], double (&v)[4][4]) {
    const Rotation r1 = rotation_for(a[P][P], a[Q][Q], a[P][Q]);
    const Rotation r2 = rotation_for(a[R][R], a[S][S], a[R][S]);
    apply_rotation<P, Q>(r1, a, v);
    apply_rotation<R, S>(r2, a, v);
}

__device__ __forceinline__ bool off_diagonal_negligible(const double (&a)[4][4]) {
    bool done = true;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = p + 1; q < 4; ++q) done = done && (a[p][q] * a[p][q] <= 1e-36 * fabs(a[p][p] * a[q][q]));
    return done;
}

// Diagonalises g (destroyed) and returns the eigenvector of its smallest eigenvalue in h.  Sweeps repeat until
// every off-diagonal entry is below 1e-18 of the geometric mean of its diagonal neighbours.
__device__ __forceinline__ void smallest_eigenvector(double (&g)[4][4], double (&h)[4]) {
    double e[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) e[p][q] = (p == q) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < kMaxJacobiSweeps; ++sweep) {
        if (off_diagonal_negligible(g)) break;
        jacobi_pair<0, 1, 2, 3>(g, e);
        jacobi_pair<0, 2, 1, 3>(g, e);
        jacobi_pair<0, 3, 1, 2>(g, e);
    }
    double best = g[0][0];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = e[k][0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const bool take = g[q][q] < best;
        best = take ? g[q][q] : best;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = take ? e[k][q] : h[k];
    }
}

}  // namespace mvx
