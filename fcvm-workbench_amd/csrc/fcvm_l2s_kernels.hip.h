// Sparse aggregate level of the multigrid cycle (meshes whose aggregate count exceeds what a dense inverse can carry):
// level 2 keeps aggregates of ~176 level-1 nodes x 6 rigid-body modes with A2 = Z^T A1 Z in block-CSR form (6 x 6 blocks,
// aggregates couple to their neighbours only), smoothed by block-Jacobi Chebyshev steps; a third, dense level of
// "super-aggregates" (bins of 2 x 2 x 2 aggregates) x 6 rigid-body modes closes the cycle:  A3 = C^T A2 C, explicit inverse.
// All level-2 / level-3 vectors have a few 10^4 entries: the kernels are launch-latency sized, one thread per aggregate or
// per (aggregate, mode), fixed summation order.  No reference counterpart (fcVM.py:1121 factorises with CHOLMOD).
#pragma once
#include "fcvm_kernels.hip.h"

namespace fcvm {

// A2 blocks: as k_galerkin_rows, but the (row aggregate, slot) block goes to blocks[slot * 36 + k * 6 + k'] (every aggregate
// carries six modes on this path; dropped modes are zero columns of Z)
__global__ __launch_bounds__(256) void k_galerkin_blocks(const int64_t *__restrict__ sptr_g, const int64_t *__restrict__ eptr,
                                                         const int32_t *__restrict__ ent_i, const int32_t *__restrict__ ent_j,
                                                         const uint64_t *__restrict__ ent_loc, const double *__restrict__ vals1,
                                                         const double *__restrict__ Z, double *__restrict__ blocks) {
    const int64_t g = blockIdx.x;
    const int64_t s0 = sptr_g[g], ns = sptr_g[g + 1] - s0;
    for (int64_t a = threadIdx.x; a < ns * 36; a += 256) {
        const int64_t sl = s0 + a / 36;
        const int kk = (int)(a % 36), k = kk / 6, kp = kk - 6 * k;
        double acc = 0.0;
        for (int64_t e = eptr[sl]; e < eptr[sl + 1]; e++) {
            const double *zi = Z + (int64_t)ent_i[e] * 18, *zj = Z + (int64_t)ent_j[e] * 18;
            int st;
            const double *v = vals1 + unpack_abs(ent_loc[e], st);
            const double z0 = zj[kp], z1 = zj[6 + kp], z2 = zj[12 + kp];
            const double t0 = v[0] * z0 + v[st] * z1 + v[2 * st] * z2;
            const double t1 = v[3 * st] * z0 + v[4 * st] * z1 + v[5 * st] * z2;
            const double t2 = v[6 * st] * z0 + v[7 * st] * z1 + v[8 * st] * z2;
            acc += zi[k] * t0 + zi[6 + k] * t1 + zi[12 + k] * t2;
        }
        blocks[sl * 36 + kk] = acc;
    }
}

// y = A2 x, block-CSR with 6 x 6 blocks (FP32 storage, FP64 vectors and accumulation); thread = (aggregate, mode)
__global__ __launch_bounds__(256) void k_bsr6_spmv(int64_t nagg, const int64_t *__restrict__ sptr_g, const int32_t *__restrict__ slot_agg,
                                                   const float *__restrict__ blocks, const double *__restrict__ x, double *__restrict__ y,
                                                   const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 6 * nagg) return;
    const int64_t g = t / 6;
    const int k = (int)(t - 6 * g);
    double acc = 0.0;
    for (int64_t sl = sptr_g[g]; sl < sptr_g[g + 1]; sl++) {
        const float *b = blocks + sl * 36 + k * 6;
        const double *xh = x + 6 * (int64_t)slot_agg[sl];
        acc += (double)b[0] * xh[0] + (double)b[1] * xh[1] + (double)b[2] * xh[2] + (double)b[3] * xh[3] + (double)b[4] * xh[4] +
               (double)b[5] * xh[5];
    }
    y[t] = acc;
}

__device__ __forceinline__ void apply_dinv6(const double *__restrict__ D, const double r[6], double z[6]) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < 6; l++) s += D[k * 6 + l] * r[l];
        z[k] = s;
    }
}

// x = 0 ; r = b ; d = c0 Dinv b      (thread = aggregate)
__global__ __launch_bounds__(256) void k_l2s_init(int64_t nagg, const double *__restrict__ dinv, const double *__restrict__ b,
                                                  double *__restrict__ x, double *__restrict__ r, double *__restrict__ d, double c0,
                                                  const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nagg) return;
    double rr[6], z[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rr[k] = b[6 * g + k];
    apply_dinv6(dinv + g * 36, rr, z);
#pragma unroll
    for (int k = 0; k < 6; k++) {
        x[6 * g + k] = 0.0;
        r[6 * g + k] = rr[k];
        d[6 * g + k] = c0 * z[k];
    }
}

// one smoother step with its SpMV: t = A2 din ; x += din ; r -= t ; if (update_d) { dout = cdd * din + cdr * Dinv r ; if (add_d)
// x += dout }.  Eight lanes per aggregate (six active: one per mode) exchange the residual with shuffles; din is read by the
// neighbours while this launch runs, so the new direction goes to dout (the caller ping-pongs).
__global__ __launch_bounds__(256) void k_l2s_spmv_step(int64_t nagg, const int64_t *__restrict__ sptr_g, const int32_t *__restrict__ slot_agg,
                                                       const float *__restrict__ blocks, const double *__restrict__ dinv,
                                                       const double *__restrict__ din, double *__restrict__ dout, double *__restrict__ x,
                                                       double *__restrict__ r, double cdd, double cdr, int update_d, int add_d,
                                                       const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t g = t >> 3;
    const int k = (int)(t & 7);
    const bool active = k < 6 && g < nagg;
    const int lane = threadIdx.x & 63, base = lane & ~7;
    double acc = 0.0, dk = 0.0, rk = 0.0;
    if (active) {
        for (int64_t sl = sptr_g[g]; sl < sptr_g[g + 1]; sl++) {
            const float *b = blocks + sl * 36 + k * 6;
            const double *xh = din + 6 * (int64_t)slot_agg[sl];
            acc += (double)b[0] * xh[0] + (double)b[1] * xh[1] + (double)b[2] * xh[2] + (double)b[3] * xh[3] + (double)b[4] * xh[4] +
                   (double)b[5] * xh[5];
        }
        dk = din[6 * g + k];
        rk = r[6 * g + k] - acc;
        r[6 * g + k] = rk;
    }
    double z = 0.0;
#pragma unroll
    for (int l = 0; l < 6; l++) {
        const double rl = __shfl(rk, base + l, 64);
        if (active && update_d) z += dinv[g * 36 + k * 6 + l] * rl;
    }
    if (active) {
        double a = dk;
        if (update_d) {
            const double dn = cdd * dk + cdr * z;
            dout[6 * g + k] = dn;
            if (add_d) a += dn;
        }
        x[6 * g + k] += a;
    }
}

// r3 = C^T r2: thread = (super-aggregate, mode m); members in list order
__global__ __launch_bounds__(256) void k_l3_restrict(int64_t nsuper, const int64_t *__restrict__ sptr_s, const int32_t *__restrict__ smem,
                                                     const int64_t *__restrict__ soff, const double *__restrict__ C,
                                                     const double *__restrict__ r2, double *__restrict__ r3, const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 6 * nsuper) return;
    const int64_t S = t / 6;
    const int m = (int)(t - 6 * S);
    if (m >= soff[S + 1] - soff[S]) return;
    double acc = 0.0;
    for (int64_t q = sptr_s[S]; q < sptr_s[S + 1]; q++) {
        const int64_t g = smem[q];
        const double *cg = C + g * 36;   // [k][m]
        const double *rg = r2 + 6 * g;
#pragma unroll
        for (int k = 0; k < 6; k++) acc += cg[k * 6 + m] * rg[k];
    }
    r3[soff[S] + m] = acc;
}

// d2 = C e3 (the correction handed to the level-2 post-smoothing): thread = aggregate
__global__ __launch_bounds__(256) void k_l3_prolong(int64_t nagg, const int32_t *__restrict__ sagg, const int64_t *__restrict__ soff,
                                                    const double *__restrict__ C, const double *__restrict__ e3, double *__restrict__ d2,
                                                    const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nagg) return;
    const int64_t S = sagg[g], o = soff[S];
    const int nm = (int)(soff[S + 1] - o);
    const double *cg = C + g * 36;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double acc = 0.0;
        for (int m = 0; m < nm; m++) acc += cg[k * 6 + m] * e3[o + m];
        d2[6 * g + k] = acc;
    }
}

// z = Dinv r  (power iteration for the smoother's eigenvalue bound)
__global__ __launch_bounds__(256) void k_l2s_precond(int64_t nagg, const double *__restrict__ dinv, const double *__restrict__ r,
                                                     double *__restrict__ z) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= nagg) return;
    double rr[6], zz[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rr[k] = r[6 * g + k];
    apply_dinv6(dinv + g * 36, rr, zz);
#pragma unroll
    for (int k = 0; k < 6; k++) z[6 * g + k] = zz[k];
}

__global__ __launch_bounds__(256) void k_f64_to_f32(int64_t n, const double *__restrict__ a, float *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = (float)a[i];
}

}  // namespace fcvm
