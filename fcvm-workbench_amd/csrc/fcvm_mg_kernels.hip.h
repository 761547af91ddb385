// p-multigrid preconditioner kernels: tet4 coarse operator, tet10 <-> tet4 transfer, Chebyshev
// smoother steps.  No reference counterpart (the reference solves with CHOLMOD, fcVM.py:1121).
#pragma once
#include "fcvm_kernels.hip.h"

namespace fcvm {

// Linear-tetrahedron stiffness on the corner nodes of the tet10 elements, same boundary-condition folding as the
// fine matrix (unit diagonal per element on fixed dofs).  One work-item per (element, row corner a).
// TANGENT: the consistent elastoplastic tangent D - pf s s^T of the element's four Gauss points (as in
// k_assemble<true>, fcVM.py:983-998) and the displaced corner coordinates -- for straight-sided elements this is the
// Galerkin operator P^T K_t P of the tangent matrix, so the coarse correction stays consistent with a softened fine level.
template <bool TANGENT, bool OWNER>
__device__ __forceinline__ void tet4_element_row(const int64_t e, const int a, const int lane, const int32_t *__restrict__ el4,
                                                       const double *__restrict__ coord1,
                                                       const uint8_t *__restrict__ fixmask1,
                                                       const uint32_t *__restrict__ emap4, MatConst mc,
                                                       double *__restrict__ vals, const int32_t *__restrict__ c2f,
                                                       const double *__restrict__ disp, const double *__restrict__ sig,
                                                       const uint8_t *__restrict__ pgp) {
    int nd[4];
    double x[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        nd[k] = el4[e * 4 + k];
#pragma unroll
        for (int c = 0; c < 3; c++) x[k][c] = coord1[3 * (int64_t)nd[k] + c];
        if (TANGENT && disp) {
            const double *u = disp + 3 * (int64_t)c2f[nd[k]];
            x[k][0] += u[0]; x[k][1] += u[1]; x[k][2] += u[2];
        }
    }
    double e1[3], e2[3], e3[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        e1[c] = x[1][c] - x[0][c];
        e2[c] = x[2][c] - x[0][c];
        e3[c] = x[3][c] - x[0][c];
    }
    double g[4][3];
    g[1][0] = e2[1] * e3[2] - e2[2] * e3[1]; g[1][1] = e2[2] * e3[0] - e2[0] * e3[2]; g[1][2] = e2[0] * e3[1] - e2[1] * e3[0];
    g[2][0] = e3[1] * e1[2] - e3[2] * e1[1]; g[2][1] = e3[2] * e1[0] - e3[0] * e1[2]; g[2][2] = e3[0] * e1[1] - e3[1] * e1[0];
    g[3][0] = e1[1] * e2[2] - e1[2] * e2[1]; g[3][1] = e1[2] * e2[0] - e1[0] * e2[2]; g[3][2] = e1[0] * e2[1] - e1[1] * e2[0];
    const double det = e1[0] * g[1][0] + e1[1] * g[1][1] + e1[2] * g[1][2];
    const double inv = 1.0 / det;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        g[1][c] *= inv; g[2][c] *= inv; g[3][c] *= inv;
        g[0][c] = -(g[1][c] + g[2][c] + g[3][c]);
    }
    const double vol = fabs(det) / 6.0;
    double ga[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (k == a) { ga[0] = g[k][0]; ga[1] = g[k][1]; ga[2] = g[k][2]; }
    double S[4][6], pf[4], ta[4][3];
    if (TANGENT) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            pf[q] = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) S[q][j] = 0.0;
            if (pgp[4 * e + q]) {
                const double *s0 = sig + 24 * e + 6 * q;
                double sd[6] = {s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
                const double p = (sd[0] + sd[1] + sd[2]) / 3.0;
                sd[0] -= p; sd[1] -= p; sd[2] -= p;
                double svm = sqrt(1.5 * (sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]) + 3.0 * (sd[3] * sd[3] + sd[4] * sd[4] + sd[5] * sd[5]));
                if (svm == 0.0) svm = 1.0;
                pf[q] = 0.25 * 3.0 * mc.G / (1.0 + mc.H / 3.0 / mc.G) / (svm * svm);   // 0.25: four equal Gauss weights
#pragma unroll
                for (int j = 0; j < 6; j++) S[q][j] = sd[j];
            }
            ta[q][0] = ga[0] * S[q][0] + ga[1] * S[q][3] + ga[2] * S[q][4];
            ta[q][1] = ga[1] * S[q][1] + ga[0] * S[q][3] + ga[2] * S[q][5];
            ta[q][2] = ga[2] * S[q][2] + ga[0] * S[q][4] + ga[1] * S[q][5];
        }
    }
    const int fa = fixmask1[nd[a]];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const double *gb = g[b];
        const double dot = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
        double sub[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (TANGENT) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double tb[3];
                tb[0] = gb[0] * S[q][0] + gb[1] * S[q][3] + gb[2] * S[q][4];
                tb[1] = gb[1] * S[q][1] + gb[0] * S[q][3] + gb[2] * S[q][5];
                tb[2] = gb[2] * S[q][2] + gb[0] * S[q][4] + gb[1] * S[q][5];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) sub[r * 3 + c] += pf[q] * ta[q][r] * tb[c];
            }
        }
        const int fb = fixmask1[nd[b]];
        int st;
        double *dst = vals + unpack_offset(emap4[(e * 4 + a) * 4 + b], lane, st);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const bool rfix = (fa >> r) & 1;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const bool cfix = (fb >> c) & 1;
                double v;
                if (r == c) v = mc.D11 * ga[r] * gb[r] + mc.D44 * (dot - ga[r] * gb[r]);
                else v = mc.D12 * ga[r] * gb[c] + mc.D44 * ga[c] * gb[r];
                if (TANGENT) v -= sub[r * 3 + c];
                double add = 0.0;
                if (!rfix && !cfix) add = vol * v;
                else if (rfix && a == b && r == c) add = 1.0;
                else continue;
                dst[(r * 3 + c) * st] += add;
            }
        }
    }
}

// one lane per row of the level-1 matrix (slot -> coarse node), walking the elements at that corner: no atomics, the
// coarse operator -- and with it the preconditioner and every PCG iterate -- is bitwise reproducible
template <bool TANGENT>
__global__ __launch_bounds__(256) void k_assemble_tet4_rows(int64_t nslots, const int32_t *__restrict__ rownode1,
                                                            const int64_t *__restrict__ nptr1, const int32_t *__restrict__ nel1,
                                                            const int32_t *__restrict__ el4, const double *__restrict__ coord1,
                                                            const uint8_t *__restrict__ fixmask1, const uint32_t *__restrict__ emap4,
                                                            MatConst mc, double *__restrict__ vals, const int32_t *__restrict__ c2f,
                                                            const double *__restrict__ disp, const double *__restrict__ sig,
                                                            const uint8_t *__restrict__ pgp) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    const int node = rownode1[slot];
    if (node < 0) return;
    for (int64_t q = nptr1[node]; q < nptr1[node + 1]; q++) {
        const int32_t packed = nel1[q];
        tet4_element_row<TANGENT, true>((int64_t)(packed >> 4), packed & 15, (int)(slot & 63), el4, coord1, fixmask1, emap4, mc, vals, c2f, disp, sig, pgp);
    }
}

// r1 = M1 P^T M0 r0 : coarse node c collects its own fine node and half of every mid-side node
// on an edge ending in c (CSR lists, fixed order -> deterministic); M = free-dof masks.
// wgt (optional, per fine dof): partition-of-unity weights of the multi-GPU cycle (1 / ranks holding the node).
__global__ __launch_bounds__(256) void k_restrict(int64_t nc, const int32_t *__restrict__ c2f,
                                                  const int64_t *__restrict__ mptr, const int32_t *__restrict__ mids,
                                                  const uint8_t *__restrict__ fixmask0, const double *__restrict__ r0,
                                                  double *__restrict__ r1, const double *__restrict__ done,
                                                  const double *__restrict__ wgt = nullptr, const int32_t *__restrict__ nlist = nullptr) {
    if (done && *done != 0.0) return;
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= nc) return;
    if (nlist) c = nlist[c];       // (multi-GPU layer: nc entries of a node list -- the shared corner nodes first, the others beside their exchange)
    const int64_t f = c2f[c];
    const int fm = fixmask0[f];
    double a0 = r0[3 * f], a1 = r0[3 * f + 1], a2 = r0[3 * f + 2];
    if (wgt) { a0 *= wgt[3 * f]; a1 *= wgt[3 * f + 1]; a2 *= wgt[3 * f + 2]; }
    for (int64_t q = mptr[c]; q < mptr[c + 1]; q++) {
        const int64_t m = mids[q];
        const int mm = fixmask0[m];
        const double w0 = wgt ? wgt[3 * m] : 1.0, w1 = wgt ? wgt[3 * m + 1] : 1.0, w2 = wgt ? wgt[3 * m + 2] : 1.0;
        if (!(mm & 1)) a0 += 0.5 * w0 * r0[3 * m];
        if (!(mm & 2)) a1 += 0.5 * w1 * r0[3 * m + 1];
        if (!(mm & 4)) a2 += 0.5 * w2 * r0[3 * m + 2];
    }
    r1[3 * c] = (fm & 1) ? 0.0 : a0;
    r1[3 * c + 1] = (fm & 2) ? 0.0 : a1;
    r1[3 * c + 2] = (fm & 4) ? 0.0 : a2;
}

// r1 = P^T (wgt * r0 - sub) with four lanes per coarse node (the partition and reduction tree of k_restrict_init below): the
// multi-GPU cycle's restriction of w0 b - K~_local x1, formed on the fly (round 4; one lane per node: 60 us at 1 M tet10, this: 30)
__global__ __launch_bounds__(256) void k_restrict_diff4(int64_t nc, const int32_t *__restrict__ c2f, const int64_t *__restrict__ mptr,
                                                        const int32_t *__restrict__ mids, const uint8_t *__restrict__ fixmask0,
                                                        const double *__restrict__ r0, const double *__restrict__ wgt,
                                                        const double *__restrict__ sub, double *__restrict__ r1) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t c = t >> 2;
    const int q = (int)(t & 3);
    const bool valid = c < nc;
    if (!valid) c = nc - 1;
    const int64_t f = c2f[c];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (q == 0) {
        a0 = wgt[3 * f] * r0[3 * f] - sub[3 * f];
        a1 = wgt[3 * f + 1] * r0[3 * f + 1] - sub[3 * f + 1];
        a2 = wgt[3 * f + 2] * r0[3 * f + 2] - sub[3 * f + 2];
    }
    for (int64_t k = mptr[c] + q; k < mptr[c + 1]; k += 4) {
        const int64_t m = mids[k];
        const int mm = fixmask0[m];
        if (!(mm & 1)) a0 += 0.5 * (wgt[3 * m] * r0[3 * m] - sub[3 * m]);
        if (!(mm & 2)) a1 += 0.5 * (wgt[3 * m + 1] * r0[3 * m + 1] - sub[3 * m + 1]);
        if (!(mm & 4)) a2 += 0.5 * (wgt[3 * m + 2] * r0[3 * m + 2] - sub[3 * m + 2]);
    }
    a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
    a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
    if (!valid || q != 0) return;
    const int fm = fixmask0[f];
    r1[3 * c] = (fm & 1) ? 0.0 : a0;
    r1[3 * c + 1] = (fm & 2) ? 0.0 : a1;
    r1[3 * c + 2] = (fm & 4) ? 0.0 : a2;
}

// The V-cycle's form of the restriction: four lanes per coarse node share the list of mid-side nodes (fixed partition,
// fixed reduction tree => deterministic), and the start of the level-1 Chebyshev sweep (k_cheb_init: x = 0, r = b,
// d = c0 Dinv b) rides on it: the restricted residual goes straight to r, x and d.
__global__ __launch_bounds__(256) void k_restrict_init(int64_t nc, const int32_t *__restrict__ c2f, const int64_t *__restrict__ mptr,
                                                       const int32_t *__restrict__ mids, const uint8_t *__restrict__ fixmask0,
                                                       const double *__restrict__ r0, const double *__restrict__ dinv1, double c0,
                                                       double *__restrict__ x1, double *__restrict__ r1, double *__restrict__ d1,
                                                       const double *__restrict__ done, float *__restrict__ d32 = nullptr) {
    if (done && *done != 0.0) return;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t c = t >> 2;
    const int q = (int)(t & 3);
    const bool valid = c < nc;
    if (!valid) c = nc - 1;
    const int64_t f = c2f[c];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (q == 0) { a0 = r0[3 * f]; a1 = r0[3 * f + 1]; a2 = r0[3 * f + 2]; }
    for (int64_t k = mptr[c] + q; k < mptr[c + 1]; k += 4) {
        const int64_t m = mids[k];
        const int mm = fixmask0[m];
        if (!(mm & 1)) a0 += 0.5 * r0[3 * m];
        if (!(mm & 2)) a1 += 0.5 * r0[3 * m + 1];
        if (!(mm & 4)) a2 += 0.5 * r0[3 * m + 2];
    }
    a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
    a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
    if (!valid || q != 0) return;
    const int fm = fixmask0[f];
    if (fm & 1) a0 = 0.0;
    if (fm & 2) a1 = 0.0;
    if (fm & 4) a2 = 0.0;
    double z0, z1, z2;
    apply_dinv(dinv1 + c * 9, a0, a1, a2, z0, z1, z2);
    x1[3 * c] = 0.0; x1[3 * c + 1] = 0.0; x1[3 * c + 2] = 0.0;
    r1[3 * c] = a0; r1[3 * c + 1] = a1; r1[3 * c + 2] = a2;
    d1[3 * c] = c0 * z0; d1[3 * c + 1] = c0 * z1; d1[3 * c + 2] = c0 * z2;
    if (d32) { d32[4 * c] = (float)(c0 * z0); d32[4 * c + 1] = (float)(c0 * z1); d32[4 * c + 2] = (float)(c0 * z2); d32[4 * c + 3] = 0.f; }   // FP32 twin for the gathers of k_spmv_l1_i16
}

// d0 = M0 P M1 e1 : fine corner nodes copy, mid-side nodes average their two edge ends
__global__ __launch_bounds__(256) void k_prolong(int64_t nn, const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                 const uint8_t *__restrict__ fixmask0, const double *__restrict__ e1,
                                                 double *__restrict__ d0, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const int64_t a = pa[i], b = pb[i];
    const int fm = fixmask0[i];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    if (a >= 0) {
        v0 = 0.5 * (e1[3 * a] + e1[3 * b]);
        v1 = 0.5 * (e1[3 * a + 1] + e1[3 * b + 1]);
        v2 = 0.5 * (e1[3 * a + 2] + e1[3 * b + 2]);
    }
    d0[3 * i] = (fm & 1) ? 0.0 : v0;
    d0[3 * i + 1] = (fm & 2) ? 0.0 : v1;
    d0[3 * i + 2] = (fm & 4) ? 0.0 : v2;
}

// z = z0 + M0 P M1 e1 (the additive cycle's second half without the inner products: multi-GPU host code)
__global__ __launch_bounds__(256) void k_prolong_add(int64_t nn, const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                     const uint8_t *__restrict__ fixmask0, const double *__restrict__ e1,
                                                     const double *__restrict__ z0, double *__restrict__ z,
                                                     const double *__restrict__ done = nullptr) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const int64_t a = pa[i], b = pb[i];
    const int fm = fixmask0[i];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double v = (a >= 0 && !((fm >> c) & 1)) ? 0.5 * (e1[3 * a + c] + e1[3 * b + c]) : 0.0;
        z[3 * i + c] = z0[3 * i + c] + v;
    }
}

// x += alpha p ; r -= alpha q ; z0 = w * Dinv r, alpha read from device memory (multi-GPU host code)
// z = Dinv r with the six distinct entries of the symmetric inverse diagonal block in FP32 (xx xy | xz yy | yz zz)
__device__ __forceinline__ void apply_dinv6f(const float *__restrict__ d6, double r0, double r1, double r2, double &a0, double &a1, double &a2) {
    const float2 *d2 = (const float2 *)d6;
    const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];
    a0 = (double)u0.x * r0 + (double)u0.y * r1 + (double)u1.x * r2;
    a1 = (double)u0.y * r0 + (double)u1.y * r1 + (double)u2.x * r2;
    a2 = (double)u1.x * r0 + (double)u2.x * r1 + (double)u2.y * r2;
}
__global__ __launch_bounds__(256) void k_update_xr_z0_ext(int64_t nn, const double *__restrict__ alpha_ptr, const double *__restrict__ p,
                                                          const double *__restrict__ q, double *__restrict__ x, double *__restrict__ r,
                                                          const double *__restrict__ dinv, double *__restrict__ z0, double w,
                                                          const float *__restrict__ dinv6 = nullptr) {
    const double alpha = *alpha_ptr;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double rn[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        x[3 * i + c] += alpha * p[3 * i + c];
        rn[c] = r[3 * i + c] - alpha * q[3 * i + c];
        r[3 * i + c] = rn[c];
    }
    double a0, a1, a2;
    if (dinv6) apply_dinv6f(dinv6 + 6 * i, rn[0], rn[1], rn[2], a0, a1, a2);
    else apply_dinv(dinv + i * 9, rn[0], rn[1], rn[2], a0, a1, a2);
    z0[3 * i] = w * a0; z0[3 * i + 1] = w * a1; z0[3 * i + 2] = w * a2;
}

// the vector half of a Chronopoulos-Gear iteration of the multi-GPU PCG in one pass (round 4): the direction updates of the PREVIOUS
// iteration, p = z + beta p, s = w + beta s (s = K p by recurrence), then x += alpha p, r -= alpha s, z0 = weight * Dinv r with the
// six-entry FP32 Dinv; alpha = S[5], beta = S[6] from k_dist_pcg_scalars (both zero once converged: nothing changes any more).
// Eleven vector passes and one launch instead of thirteen and three.
__global__ __launch_bounds__(256) void k_dist_cg_update(int64_t nn, const double *__restrict__ S, const double *__restrict__ z,
                                                        const double *__restrict__ w, double *__restrict__ p, double *__restrict__ s,
                                                        double *__restrict__ x, double *__restrict__ r, const float *__restrict__ dinv6,
                                                        double *__restrict__ z0, double weight) {
    const double alpha = S[5], beta = S[6];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double rn[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double pn = z[3 * i + c] + beta * p[3 * i + c], sn = w[3 * i + c] + beta * s[3 * i + c];
        p[3 * i + c] = pn;
        s[3 * i + c] = sn;
        x[3 * i + c] += alpha * pn;
        rn[c] = r[3 * i + c] - alpha * sn;
        r[3 * i + c] = rn[c];
    }
    double a0, a1, a2;
    apply_dinv6f(dinv6 + 6 * i, rn[0], rn[1], rn[2], a0, a1, a2);
    z0[3 * i] = weight * a0; z0[3 * i + 1] = weight * a1; z0[3 * i + 2] = weight * a2;
}

// end of the multi-GPU cycle (form 2) once the second 16-bit product t = K~ x2 has been summed over the ranks (round 4):
// z = x2 + c0 Dinv (b - t) with the six-entry FP32 form of Dinv (the S of k_jacobi6f and k_update_xr_z0_ext: pre- and
// post-smoothing must be one operator), and on the way the rank's owner-weighted shares of b.z and b.b, which the
// single-reduction CG wants next (b is its residual r): per-workgroup partials [2][gridDim.x], summed by k_sum_final2.
// Replaces a subtraction, k_cheb_init, an addition and two of the three inner-product passes.
__global__ __launch_bounds__(256) void k_dist_post_smooth(int64_t nn, const float *__restrict__ dinv6, const double *__restrict__ x2,
                                                          const double *__restrict__ b, const double *__restrict__ t, double c0,
                                                          const double *__restrict__ ow, double *__restrict__ z, double *__restrict__ part) {
    __shared__ double sh[4];
    double bz = 0.0, bb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (int64_t)gridDim.x * 256) {
        const double b0 = b[3 * i], b1 = b[3 * i + 1], b2 = b[3 * i + 2];
        double a0, a1, a2;
        apply_dinv6f(dinv6 + 6 * i, b0 - t[3 * i], b1 - t[3 * i + 1], b2 - t[3 * i + 2], a0, a1, a2);
        const double z0 = x2[3 * i] + c0 * a0, z1 = x2[3 * i + 1] + c0 * a1, z2 = x2[3 * i + 2] + c0 * a2;
        z[3 * i] = z0; z[3 * i + 1] = z1; z[3 * i + 2] = z2;
        const double w0 = ow[3 * i], w1 = ow[3 * i + 1], w2 = ow[3 * i + 2];
        bz += w0 * b0 * z0 + w1 * b1 * z1 + w2 * b2 * z2;
        bb += w0 * b0 * b0 + w1 * b1 * b1 + w2 * b2 * b2;
    }
    bz = block_sum(bz, sh);
    __syncthreads();
    bb = block_sum(bb, sh);
    if (threadIdx.x == 0) { part[blockIdx.x] = bz; part[gridDim.x + blockIdx.x] = bb; }
}
__global__ __launch_bounds__(256) void k_sum_final2(const double *__restrict__ part, int64_t n, double *__restrict__ out0, double *__restrict__ out1) {
    __shared__ double sh[4];
    const double a = sum_partials(part, n, sh);
    __syncthreads();
    const double b = sum_partials(part + n, n, sh);
    if (threadIdx.x == 0) { *out0 = a; *out1 = b; }
}

// start of a Chebyshev sweep from the zero vector: x = 0, r = b, d = c0 * Dinv b
__global__ __launch_bounds__(256) void k_cheb_init(int64_t nn, const double *__restrict__ dinv, const double *__restrict__ b,
                                                   double *__restrict__ x, double *__restrict__ r, double *__restrict__ d,
                                                   double c0, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const double r0 = b[3 * i], r1 = b[3 * i + 1], r2 = b[3 * i + 2];
    double z0, z1, z2;
    apply_dinv(dinv + i * 9, r0, r1, r2, z0, z1, z2);
    x[3 * i] = 0.0; x[3 * i + 1] = 0.0; x[3 * i + 2] = 0.0;
    if (r != b) { r[3 * i] = r0; r[3 * i + 1] = r1; r[3 * i + 2] = r2; }
    d[3 * i] = c0 * z0; d[3 * i + 1] = c0 * z1; d[3 * i + 2] = c0 * z2;
}

// x += d ; r -= t (t = A d, may be NULL: then r is left alone) ; d = cdd * d + cdr * Dinv r
__global__ __launch_bounds__(256) void k_cheb_step(int64_t nn, const double *__restrict__ dinv, const double *__restrict__ t,
                                                   double *__restrict__ x, double *__restrict__ r, double *__restrict__ d,
                                                   double cdd, double cdr, int update_d, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const double d0 = d[3 * i], d1 = d[3 * i + 1], d2 = d[3 * i + 2];
    x[3 * i] += d0; x[3 * i + 1] += d1; x[3 * i + 2] += d2;
    if (t) {
        const double r0 = r[3 * i] - t[3 * i], r1 = r[3 * i + 1] - t[3 * i + 1], r2 = r[3 * i + 2] - t[3 * i + 2];
        r[3 * i] = r0; r[3 * i + 1] = r1; r[3 * i + 2] = r2;
        if (update_d) {
            double z0, z1, z2;
            apply_dinv(dinv + i * 9, r0, r1, r2, z0, z1, z2);
            d[3 * i] = cdd * d0 + cdr * z0; d[3 * i + 1] = cdd * d1 + cdr * z1; d[3 * i + 2] = cdd * d2 + cdr * z2;
        }
    }
}

// PCG with an external preconditioner: x += alpha p ; r -= alpha q ; partial of r.r (slot 1)
__global__ __launch_bounds__(256) void k_pcg_update_xr_plain(int64_t nn, const double *__restrict__ p, const double *__restrict__ q,
                                                             double *__restrict__ x, double *__restrict__ r,
                                                             const double *__restrict__ S) {
    if (S[S_DONE] != 0.0) return;
    const double alpha = S[S_ALPHA];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        x[3 * i + c] += alpha * p[3 * i + c];
        r[3 * i + c] -= alpha * q[3 * i + c];
    }
}

// additive cycle, fused with the PCG update: x += alpha p ; r -= alpha q ; z0 = w * Dinv r (the smoother part of z)
__global__ __launch_bounds__(256) void k_pcg_update_xr_z0(int64_t nn, const double *__restrict__ p, const double *__restrict__ q,
                                                          double *__restrict__ x, double *__restrict__ r,
                                                          const double *__restrict__ dinv, double *__restrict__ z0, double w,
                                                          const double *__restrict__ S, const float *__restrict__ dinv6 = nullptr) {
    if (S[S_DONE] != 0.0) return;
    const double alpha = S[S_ALPHA];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double rn[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        x[3 * i + c] += alpha * p[3 * i + c];
        rn[c] = r[3 * i + c] - alpha * q[3 * i + c];
        r[3 * i + c] = rn[c];
    }
    double a0, a1, a2;
    if (dinv6) {   // the six distinct entries of the (symmetric) inverse diagonal block in FP32: 24 instead of 72 bytes per node
        const float2 *d2 = (const float2 *)(dinv6 + 6 * i);
        const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];   // xx xy | xz yy | yz zz
        a0 = (double)u0.x * rn[0] + (double)u0.y * rn[1] + (double)u1.x * rn[2];
        a1 = (double)u0.y * rn[0] + (double)u1.y * rn[1] + (double)u2.x * rn[2];
        a2 = (double)u1.x * rn[0] + (double)u2.x * rn[1] + (double)u2.y * rn[2];
    } else {
        apply_dinv(dinv + i * 9, rn[0], rn[1], rn[2], a0, a1, a2);
    }
    z0[3 * i] = w * a0; z0[3 * i + 1] = w * a1; z0[3 * i + 2] = w * a2;
}

// z = w * Dinv r with the FP32 six-entry form of Dinv: the smoother of the multiplicative cycle outside the PCG loop (the
// same operator as in k_pcg_update_xr_z0 and the epilogue of k_spmv_win_lo<2>: pre- and post-smoothing must be one S)
__global__ __launch_bounds__(256) void k_jacobi6f(int64_t nn, const float *__restrict__ dinv6, const double *__restrict__ r, double w,
                                                  double *__restrict__ z) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const float2 *d2 = (const float2 *)(dinv6 + 6 * i);
    const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];   // xx xy | xz yy | yz zz
    const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
    z[3 * i] = w * ((double)u0.x * r0 + (double)u0.y * r1 + (double)u1.x * r2);
    z[3 * i + 1] = w * ((double)u0.y * r0 + (double)u1.y * r1 + (double)u2.x * r2);
    z[3 * i + 2] = w * ((double)u1.x * r0 + (double)u2.x * r1 + (double)u2.y * r2);
}

// six distinct entries of every inverse diagonal block, FP32 (for k_pcg_update_xr_z0)
__global__ __launch_bounds__(256) void k_dinv_pack6f(int64_t nn, const double *__restrict__ dinv, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    const double *d = dinv + 9 * i;
    float *o = out + 6 * i;
    o[0] = (float)d[0]; o[1] = (float)(0.5 * (d[1] + d[3])); o[2] = (float)(0.5 * (d[2] + d[6]));
    o[3] = (float)d[4]; o[4] = (float)(0.5 * (d[5] + d[7])); o[5] = (float)d[8];
}

// additive cycle, second half: z = z0 + M0 P M1 e1 and the partials of r.z (slot 0) and r.r (slot 1)
__global__ __launch_bounds__(256) void k_prolong_add_dot(int64_t nn, const int32_t *__restrict__ pa, const int32_t *__restrict__ pb,
                                                         const uint8_t *__restrict__ fixmask0, const double *__restrict__ e1,
                                                         const double *__restrict__ z0, const double *__restrict__ r,
                                                         double *__restrict__ z, double *__restrict__ part, int64_t nb,
                                                         const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    __shared__ double sh[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr = 0.0;
    if (i < nn) {
        const int64_t a = pa[i], b = pb[i];
        const int fm = fixmask0[i];
        double v[3] = {0.0, 0.0, 0.0};
        if (a >= 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) v[c] = 0.5 * (e1[3 * a + c] + e1[3 * b + c]);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double zc = z0[3 * i + c] + (((fm >> c) & 1) ? 0.0 : v[c]);
            const double rc = r[3 * i + c];
            z[3 * i + c] = zc;
            rz += rc * zc;
            rr += rc * rc;
        }
    }
    rz = block_sum(rz, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = rz;
        part[nb + blockIdx.x] = rr;
    }
}

// partials of r.z (slot 0) and r.r (slot 1), layout of k_pcg_scalars
__global__ __launch_bounds__(256) void k_dot2_partial(int64_t nn, const double *__restrict__ r, const double *__restrict__ z,
                                                      double *__restrict__ part, int64_t nb, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    __shared__ double sh[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr = 0.0;
    if (i < nn) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double a = r[3 * i + c];
            rz += a * z[3 * i + c];
            rr += a * a;
        }
    }
    rz = block_sum(rz, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = rz;
        part[nb + blockIdx.x] = rr;
    }
}

// ---- level 2: aggregates of coarse nodes with (orthonormalised) rigid-body modes -------------
// r2[aoff[g]+k] = sum_{i in aggregate g} sum_c Z[i][c][k] * r1[3i+c]; one wavefront per aggregate
__global__ __launch_bounds__(256) void k_l2_restrict(int64_t nagg, const int64_t *__restrict__ aptr, const int32_t *__restrict__ anodes,
                                                     const int64_t *__restrict__ aoff, const double *__restrict__ Z,
                                                     const double *__restrict__ r1, double *__restrict__ r2, const double *__restrict__ done,
                                                     const double *__restrict__ wgt = nullptr, const double *__restrict__ sub = nullptr) {
    if (done && *done != 0.0) return;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nagg) return;
    const int lane = threadIdx.x & 63;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t q = aptr[g] + lane; q < aptr[g + 1]; q += 64) {
        const int64_t i = anodes[q];
        const double *z = Z + i * 18;
        double a0 = r1[3 * i], a1 = r1[3 * i + 1], a2 = r1[3 * i + 2];
        if (wgt) { a0 *= wgt[3 * i]; a1 *= wgt[3 * i + 1]; a2 *= wgt[3 * i + 2]; }   // (multi-GPU: owner weights, shared nodes count once)
        if (sub) { a0 -= sub[3 * i]; a1 -= sub[3 * i + 1]; a2 -= sub[3 * i + 2]; }     // (... minus the rank's unsummed last product, see DistMG._coarse)
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] += z[k] * a0 + z[6 + k] * a1 + z[12 + k] * a2;
    }
    const int64_t o = aoff[g], ncol = aoff[g + 1] - o;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double v = wave_sum(acc[k]);
        if (lane == 0 && k < ncol) r2[o + k] = v;
    }
}

// y = A x, A dense row-major n x n (FP64 or FP32 storage, FP64 accumulation); one wavefront per row
template <class T>
__global__ __launch_bounds__(256) void k_gemv(int64_t n, const T *__restrict__ A, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    const T *a = A + row * n;
    double acc = 0.0;
    for (int64_t j = lane; j < n; j += 64) acc += (double)a[j] * x[j];
    acc = wave_sum(acc);
    if (lane == 0) y[row] = acc;
}

// y = A x for the FP32-stored dense inverse (m x n, rows padded to lda = multiple of 4, zero filled): four rows per
// wavefront share the x values in registers, 16-byte loads, FP64 accumulation
__global__ __launch_bounds__(256) void k_gemv_f32x4(int64_t m, int64_t n, int64_t lda, const float *__restrict__ A,
                                                    const double *__restrict__ x, double *__restrict__ y,
                                                    const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (row0 >= m) return;
    const int lane = threadIdx.x & 63;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *a[4];
#pragma unroll
    for (int r = 0; r < 4; r++) a[r] = (const v4f *)(A + (row0 + r < m ? row0 + r : m - 1) * lda);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t j = lane * 4; j < lda; j += 256) {
        v4f v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = __builtin_nontemporal_load(a[r] + (j >> 2));
        // (every x is guarded: lda is a multiple of 32 since round 3, so j runs up to 31 entries past n -- the padding of A is zero there, but
        // 0 x whatever lies behind the vector is NaN when that happens to be a NaN or an infinity: found in round 5 as a solve that broke down
        // only late in long test runs, on recycled device memory)
        const double x0 = j < n ? x[j] : 0.0, x1 = j + 1 < n ? x[j + 1] : 0.0, x2 = j + 2 < n ? x[j + 2] : 0.0, x3 = j + 3 < n ? x[j + 3] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] += (double)v[r].x * x0 + (double)v[r].y * x1 + (double)v[r].z * x2 + (double)v[r].w * x3;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double t = wave_sum(acc[r]);
        if (lane == 0 && row0 + r < m) y[row0 + r] = t;
    }
}

// y = A x for the SYMMETRIC FP32-stored dense inverse, reading its lower triangle only (round 3: with aggregates of 72 nodes
// the 11818^2 inverse of the 1 M plate is 0.56 GB and k_gemv_f32x4 0.10 ms of every PCG iteration).  One workgroup per 128 x 128
// tile (I, J), J <= I, numbered row by row; thread (ty, tx) of a 16 x 16 grid owns an 8 x 8 block of it and forms both
// products of that block: rows with x_J (for y_I) and, off the diagonal, columns with x_I (for y_J).  The 16 partial sums per
// row / column are added in a fixed order through LDS and the tile's two result vectors go to slot J of row block I and slot
// I of row block J of `part` [nt][nt][128]; k_symv_reduce adds the nt slots of a row block in slot order: no atomics, bitwise
// reproducible.  Rows >= n are read from row n - 1 and multiplied by zero; columns n .. lda - 1 are the zero padding.
constexpr int SYMV_T = 128;
__global__ __launch_bounds__(256) void k_symv_tiles(int nt, int64_t n, int64_t lda, const float *__restrict__ A, const double *__restrict__ x,
                                                    double *__restrict__ part, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    __shared__ double sR[16][SYMV_T + 1], sC[16][SYMV_T + 1];
    const int t = blockIdx.x;
    int I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) I++;
    while (I * (I + 1) / 2 > t) I--;
    const int J = t - I * (I + 1) / 2;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int64_t r0 = (int64_t)I * SYMV_T + ty * 8, c0 = (int64_t)J * SYMV_T + tx * 8;
    typedef float v4f __attribute__((ext_vector_type(4)));
    double xJ[8], xI[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        xJ[k] = c0 + k < n ? x[c0 + k] : 0.0;
        xI[k] = r0 + k < n ? x[r0 + k] : 0.0;
    }
    v4f a[8][2];
    const bool in0 = c0 < lda, in1 = c0 + 4 < lda;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int64_t rr = r0 + r < n ? r0 + r : n - 1;
        const v4f *row = (const v4f *)(A + rr * lda + c0);
        a[r][0] = in0 ? __builtin_nontemporal_load(row) : (v4f)(0.0f);
        a[r][1] = in1 ? __builtin_nontemporal_load(row + 1) : (v4f)(0.0f);
    }
    double cv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double v[8] = {(double)a[r][0].x, (double)a[r][0].y, (double)a[r][0].z, (double)a[r][0].w,
                             (double)a[r][1].x, (double)a[r][1].y, (double)a[r][1].z, (double)a[r][1].w};
        double ru = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            ru += v[k] * xJ[k];
            cv[k] += v[k] * xI[r];
        }
        sR[tx][ty * 8 + r] = ru;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) sC[ty][tx * 8 + k] = cv[k];
    __syncthreads();
    if (threadIdx.x < SYMV_T) {
        double u = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q++) u += sR[q][threadIdx.x];
        part[((int64_t)I * nt + J) * SYMV_T + threadIdx.x] = u;
    } else if (I != J) {
        const int cc = threadIdx.x - SYMV_T;
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q++) v += sC[q][cc];
        part[((int64_t)J * nt + I) * SYMV_T + cc] = v;
    }
}
// one workgroup per row block: thread (half, tt) adds every second slot of entry tt with four loads in flight, the two halves are
// joined through LDS -- a fixed association, whatever the launch (a single chain of nt dependent loads took 25 us at nt = 93)
__global__ __launch_bounds__(256) void k_symv_reduce(int nt, int64_t n, const double *__restrict__ part, double *__restrict__ y,
                                                     const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    __shared__ double sh[SYMV_T];
    const int64_t K = blockIdx.x;
    const int tt = threadIdx.x & (SYMV_T - 1), half = threadIdx.x >> 7;
    const double *p = part + K * nt * SYMV_T + tt;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int sl = half;
    for (; sl + 6 < nt; sl += 8) {
        v0 += p[(int64_t)sl * SYMV_T];
        v1 += p[(int64_t)(sl + 2) * SYMV_T];
        v2 += p[(int64_t)(sl + 4) * SYMV_T];
        v3 += p[(int64_t)(sl + 6) * SYMV_T];
    }
    for (; sl < nt; sl += 2) v0 += p[(int64_t)sl * SYMV_T];
    const double v = (v0 + v1) + (v2 + v3);
    if (half) sh[tt] = v;
    __syncthreads();
    const int64_t i = K * SYMV_T + tt;
    if (!half && i < n) y[i] = v + sh[tt];
}

// A[a][b] = A[b][a] for a < b: potri(upper, column-major) leaves the inverse in the row-major lower triangle
__global__ __launch_bounds__(256) void k_mirror_lower_to_upper(int64_t n, double *A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * n) return;
    const int64_t a = i / n, b = i - a * n;
    if (a < b) A[i] = A[b * n + a];
}

// FP64 n x n -> FP32 n x lda (zero padded)
__global__ __launch_bounds__(256) void k_to_f32_padded(int64_t n, int64_t lda, const double *__restrict__ a, float *__restrict__ o) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * lda) return;
    const int64_t r = i / lda, c = i - r * lda;
    o[i] = c < n ? (float)a[r * n + c] : 0.0f;
}

// d1[3i+c] = sum_k Z[i][c][k] * e2[aoff[agg[i]]+k]
__global__ __launch_bounds__(256) void k_l2_prolong(int64_t nc, const int32_t *__restrict__ agg, const int64_t *__restrict__ aoff,
                                                    const double *__restrict__ Z, const double *__restrict__ e2, double *__restrict__ d1, const double *__restrict__ done,
                                                    float *__restrict__ d32 = nullptr) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nc) return;
    const int64_t g = agg[i], o = aoff[g], ncol = aoff[g + 1] - o;
    const double *z = Z + i * 18;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (int k = 0; k < ncol; k++) {
        const double e = e2[o + k];
        v0 += z[k] * e; v1 += z[6 + k] * e; v2 += z[12 + k] * e;
    }
    d1[3 * i] = v0; d1[3 * i + 1] = v1; d1[3 * i + 2] = v2;
    if (d32) { d32[4 * i] = (float)v0; d32[4 * i + 1] = (float)v1; d32[4 * i + 2] = (float)v2; d32[4 * i + 3] = 0.f; }
}

// ---- shared-node exchange of the element partition (multi-GPU): pack / fixed-order sum ----------------
// out[k][c] = v[idx[k]][c]: the values a rank sends to its neighbours, all neighbours in one buffer
__global__ __launch_bounds__(256) void k_pack_nodes(int64_t n, const int64_t *__restrict__ idx, int ncomp, const double *__restrict__ v,
                                                    double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * ncomp) return;
    const int64_t k = t / ncomp, c = t - k * ncomp;
    out[t] = v[idx[k] * ncomp + c];
}
// v[idx[k]][c] += a[k][c] for DISTINCT idx: the follower-pressure part of the external load vector, computed on the host for the loaded faces' nodes only
// (fcVM.py:858-883: pressure follows the deformed geometry in a GNLY analysis) and added on the device
__global__ __launch_bounds__(256) void k_scatter_add_nodes(int64_t n, const int64_t *__restrict__ idx, int ncomp, const double *__restrict__ a,
                                                           double *__restrict__ v) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * ncomp) return;
    const int64_t k = t / ncomp, c = t - k * ncomp;
    v[idx[k] * ncomp + c] += a[t];
}
// v[node[t]][c] = sum over the entries of interface node t, in list order, of (src < 0 ? v[node[t]][c] : recv[src][c]).
// The lists are sorted by the rank a contribution comes from, the own one standing at the place of the own rank, so that
// every rank holding the node adds the same numbers in the same order: shared values stay bit-identical on all sharers.
__global__ __launch_bounds__(256) void k_unpack_sum(int64_t nshared, const int64_t *__restrict__ node, const int64_t *__restrict__ eptr,
                                                    const int64_t *__restrict__ src, int ncomp, double *__restrict__ v,
                                                    const double *__restrict__ recv) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nshared * ncomp) return;
    const int64_t k = t / ncomp, c = t - k * ncomp;
    const int64_t nd = node[k];
    const double own = v[nd * ncomp + c];
    double acc = 0.0;
    for (int64_t e = eptr[k]; e < eptr[k + 1]; e++) {
        const int64_t sidx = src[e];
        acc += sidx < 0 ? own : recv[sidx * ncomp + c];
    }
    v[nd * ncomp + c] = acc;
}

// The shared-node sum of k_unpack_sum (3 components) followed by the Chebyshev smoother step of that node -- the rows k_spmv_l1_i16 deferred
// (ChebEpi::defer): t = sum over the ranks of their products;  r -= t;  x += d;  if (update_d) { dout = cdd d + cdr Dinv r; if (add_dout) x += dout }.
// Every sharer adds the same numbers in the same order and steps with the same arithmetic: the consistent vectors stay bit-identical.
// thread = interface node.
__global__ __launch_bounds__(256) void k_unpack_sum_cheb(int64_t nshared, const int64_t *__restrict__ node, const int64_t *__restrict__ eptr,
                                                         const int64_t *__restrict__ src, const double *__restrict__ traw,
                                                         const double *__restrict__ recv, const float *__restrict__ dinv6,
                                                         const double *__restrict__ din, double *__restrict__ dout, double *__restrict__ x,
                                                         double *__restrict__ r, double cdd, double cdr, int update_d, int add_dout) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= nshared) return;
    const int64_t nd = node[k], o = 3 * nd;
    double t[3] = {0.0, 0.0, 0.0};
    for (int64_t e = eptr[k]; e < eptr[k + 1]; e++) {
        const int64_t sidx = src[e];
        const double *p = sidx < 0 ? traw + o : recv + 3 * sidx;
        t[0] += p[0]; t[1] += p[1]; t[2] += p[2];
    }
    const double d0 = din[o], d1 = din[o + 1], d2 = din[o + 2];
    const double r0 = r[o] - t[0], r1 = r[o + 1] - t[1], r2 = r[o + 2] - t[2];
    r[o] = r0; r[o + 1] = r1; r[o + 2] = r2;
    double a0 = d0, a1 = d1, a2 = d2;
    if (update_d) {
        const float2 *q = (const float2 *)(dinv6 + 6 * nd);
        const float2 v0 = q[0], v1 = q[1], v2 = q[2];   // xx xy | xz yy | yz zz
        const double z0 = (double)v0.x * r0 + (double)v0.y * r1 + (double)v1.x * r2;
        const double z1 = (double)v0.y * r0 + (double)v1.y * r1 + (double)v2.x * r2;
        const double z2 = (double)v1.x * r0 + (double)v2.x * r1 + (double)v2.y * r2;
        const double n0 = cdd * d0 + cdr * z0, n1 = cdd * d1 + cdr * z1, n2 = cdd * d2 + cdr * z2;
        dout[o] = n0; dout[o + 1] = n1; dout[o + 2] = n2;
        if (add_dout) { a0 += n0; a1 += n1; a2 += n2; }
    }
    x[o] += a0; x[o + 1] += a1; x[o + 2] += a2;
}

__global__ __launch_bounds__(256) void k_add_if(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ o,
                                                const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

// Scalars of the multi-GPU PCG (fcvm_hip/dist.py), kept on the device so that the host never waits for them:
// S = {rz, rr, thresh, done, iterations, alpha, beta}.  phase 0: in[0] = p.q (all-reduced) -> alpha; phase 1: in = {r.z, r.r}
// (all-reduced) -> beta, rz, rr, iterations, done.  After convergence (done != 0) alpha = beta = 0: queued iterations change
// nothing.  One thread: this replaces a dozen one-element tensor operations per iteration.
__global__ void k_dist_pcg_scalars(int phase, double *__restrict__ S, const double *__restrict__ in) {
    // phases 2 / 3: the single-reduction (Chronopoulos-Gear) loop, in = {r.z, w.z, r.r} with w = K z (all-reduced).
    // 3 = start (S[2] = threshold set by the host): alpha = gamma / delta, beta = 0.  2 = after x += alpha p, r -= alpha s:
    // iteration count, convergence, then beta = gamma_new / gamma, alpha = gamma_new / (delta - beta gamma_new / alpha).
    if (phase == 3) {
        S[0] = in[0]; S[1] = in[2]; S[3] = 0.0; S[4] = 0.0;
        S[5] = in[1] > 0.0 && in[0] >= 0.0 ? in[0] / in[1] : 0.0;
        if (in[2] > S[2] && (in[1] <= 0.0 || in[0] < 0.0)) S[3] = 2.0;       // breakdown at the start (see phase 2)
        S[6] = 0.0;
        return;
    }
    if (phase == 2) {
        const double gn = in[0], delta = in[1], rrn = in[2];
        if (S[3] == 0.0) {
            S[4] += 1.0;
            S[1] = rrn;
            if (rrn <= S[2] || !isfinite(rrn)) S[3] = 1.0;
        }
        bool go = S[3] == 0.0;
        double beta = (go && S[0] != 0.0) ? gn / S[0] : 0.0;
        const double den = delta - beta * gn / (S[5] != 0.0 ? S[5] : 1.0);
        // p.Kp (= den) <= 0 or r.M^-1 r (= gn) < 0: operator or preconditioner not positive definite -- S[3] = 2 stops the loop
        // as a breakdown (the host raises) instead of iterating on
        if (go && (den <= 0.0 || gn < 0.0)) {      // (S[5], S[6] = the offending p.Kp and r.M^-1 r for the host's message; the loop is over)
            S[3] = 2.0; S[5] = den; S[6] = gn;
            return;
        }
        if (S[3] == 2.0) return;
        const double alpha = (go && S[5] != 0.0) ? gn / den : 0.0;
        if (go) S[0] = gn;
        S[5] = alpha;
        S[6] = beta;
        return;
    }
    const bool live = S[3] == 0.0;
    if (phase == 0) {
        const double pq = in[0];
        S[5] = (live && pq != 0.0) ? S[0] / pq : 0.0;
        return;
    }
    const double rz_new = in[0], rr_new = in[1];
    S[6] = (live && S[0] != 0.0) ? rz_new / S[0] : 0.0;
    if (live) {
        S[0] = rz_new;
        S[1] = rr_new;
        S[4] += 1.0;
    }
    if (S[1] <= S[2] || !isfinite(S[1])) S[3] = 1.0;
}

// Start vector of the power iterations (mg_estimate_lmax): entry i is what the host loop st = st * a + c, h[i] = f(st) gave --
// the i + 1-fold affine map by repeated squaring, so that the smoother bounds (and everything downstream) stay bitwise what the
// host-generated vector gave, without generating and uploading 34 MB per estimate.
__global__ __launch_bounds__(256) void k_lcg_vector(int64_t n, double *__restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t A = 1, C = 0, pa = 6364136223846793005ull, pc = 1442695040888963407ull;
    for (uint64_t e = (uint64_t)i + 1; e; e >>= 1) {
        if (e & 1) { A = pa * A; C = pa * C + pc; }
        pc = pa * pc + pc;
        pa = pa * pa;
    }
    const uint64_t st = A * 0x9E3779B97F4A7C15ull + C;
    v[i] = (double)((st >> 11) & 0xFFFFF) / 1048576.0 - 0.5;
}

}  // namespace fcvm
