// Device code of libfcvm_hip: tet10 element kernels, sliced-ELL 3x3-block SpMV, PCG vector
// kernels.  gfx950 only (64-wide wavefronts are assumed throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcvm {

#define FCVM_GPW 0.041666666666667  // fcVM.py:589-596 Gauss weight
#define FCVM_GPA 0.138196601125011
#define FCVM_GPB 0.585410196624968

__device__ __forceinline__ void gp_point(int g, double &xi, double &et, double &ze) {
    xi = (g == 1) ? FCVM_GPB : FCVM_GPA;
    et = (g == 2) ? FCVM_GPB : FCVM_GPA;
    ze = (g == 3) ? FCVM_GPB : FCVM_GPA;
}

// local derivatives of shape function k (fcVM.py:390-424)
__device__ __forceinline__ void dshp_k(int k, double xi, double et, double ze, double &d0, double &d1, double &d2) {
    switch (k) {
        case 0: d0 = d1 = d2 = 1.0 - 4.0 * (1.0 - xi - et - ze); break;
        case 1: d0 = 4.0 * xi - 1.0; d1 = 0.0; d2 = 0.0; break;
        case 2: d0 = 0.0; d1 = 4.0 * et - 1.0; d2 = 0.0; break;
        case 3: d0 = 0.0; d1 = 0.0; d2 = 4.0 * ze - 1.0; break;
        case 4: d0 = 4.0 * (1.0 - 2.0 * xi - et - ze); d1 = -4.0 * xi; d2 = -4.0 * xi; break;
        case 5: d0 = 4.0 * et; d1 = 4.0 * xi; d2 = 0.0; break;
        case 6: d0 = -4.0 * et; d1 = 4.0 * (1.0 - xi - 2.0 * et - ze); d2 = -4.0 * et; break;
        case 7: d0 = -4.0 * ze; d1 = -4.0 * ze; d2 = 4.0 * (1.0 - xi - et - 2.0 * ze); break;
        case 8: d0 = 4.0 * ze; d1 = 0.0; d2 = 4.0 * xi; break;
        default: d0 = 0.0; d1 = 4.0 * ze; d2 = 4.0 * et; break;
    }
}

// shape function k (fcVM.py:365-380)
__device__ __forceinline__ double shp_k(int k, double xi, double et, double ze) {
    double a = 1.0 - xi - et - ze;
    switch (k) {
        case 0: return (2.0 * a - 1.0) * a;
        case 1: return xi * (2.0 * xi - 1.0);
        case 2: return et * (2.0 * et - 1.0);
        case 3: return ze * (2.0 * ze - 1.0);
        case 4: return 4.0 * xi * a;
        case 5: return 4.0 * xi * et;
        case 6: return 4.0 * et * a;
        case 7: return 4.0 * ze * a;
        case 8: return 4.0 * xi * ze;
        default: return 4.0 * et * ze;
    }
}

// xs[i*3+j] = d x_i / d xi_j ; returns det, xsi = inverse as the reference writes it
// (fcVM.py:436-453): xsi[m*3+i]
__device__ __forceinline__ double jac_inverse(const double *xs, double *xsi) {
    double xsj = xs[0] * xs[4] * xs[8] - xs[0] * xs[5] * xs[7] + xs[2] * xs[3] * xs[7] -
                 xs[2] * xs[4] * xs[6] + xs[1] * xs[5] * xs[6] - xs[1] * xs[3] * xs[8];
    double r = 1.0 / xsj;
    xsi[0] = (xs[4] * xs[8] - xs[7] * xs[5]) * r;
    xsi[1] = (xs[2] * xs[7] - xs[1] * xs[8]) * r;
    xsi[2] = (xs[1] * xs[5] - xs[2] * xs[4]) * r;
    xsi[3] = (xs[5] * xs[6] - xs[3] * xs[8]) * r;
    xsi[4] = (xs[0] * xs[8] - xs[2] * xs[6]) * r;
    xsi[5] = (xs[3] * xs[2] - xs[0] * xs[5]) * r;
    xsi[6] = (xs[3] * xs[7] - xs[6] * xs[4]) * r;
    xsi[7] = (xs[6] * xs[1] - xs[0] * xs[7]) * r;
    xsi[8] = (xs[0] * xs[4] - xs[3] * xs[1]) * r;
    return xsj;
}

// global derivative of a shape function from its local derivative (fcVM.py:457-460)
__device__ __forceinline__ void to_global(const double *xsi, double d0, double d1, double d2, double &g0, double &g1, double &g2) {
    g0 = xsi[0] * d0 + xsi[3] * d1 + xsi[6] * d2;
    g1 = xsi[1] * d0 + xsi[4] * d1 + xsi[7] * d2;
    g2 = xsi[2] * d0 + xsi[5] * d1 + xsi[8] * d2;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// deterministic block sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; i++) r += sh[i];
    }
    return r;
}

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// packed block location (Pattern::pack_offset: half chunk << 2 | (j & 1) << 1 | single) of a block of the row in `lane`
// -> offset in doubles of component 0 of the 3x3 block; its components are `stride` doubles apart (128 in a block pair,
// 64 in the single last block of an odd-length slice)
__device__ __forceinline__ int64_t unpack_offset(uint32_t em, int lane, int &stride) {
    const bool single = em & 1u;
    stride = single ? 64 : 128;
    return (int64_t)(em >> 2) * 576 + (single ? lane : 2 * lane + (int)((em >> 1) & 1u));
}
// Element map on the device (Pattern::emap, fcvm_pattern.cpp "element map"): one lane per (element, local row node a) walks the
// column entries of that row in the sliced-ELL layout (Pattern::col_index) and notes, for each of the element's ten nodes, the
// packed location (Pattern::pack_offset) of the block whose column it is.  Columns ascend and the padding behind a row's last
// block repeats the row's own node, whose real entry comes earlier: the first hit is the block.  The rows of an element's
// neighbours are read by all their elements -- L2 hits; at 1 M elements the kernel replaces 0.07 s of host searches and a
// 400 MB upload.
__global__ __launch_bounds__(256) void k_build_emap(int64_t ne, const int32_t *__restrict__ el, const int32_t *__restrict__ slotof,
                                                    const int64_t *__restrict__ sptr, const int32_t *__restrict__ cols,
                                                    uint32_t *__restrict__ emap) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne * 10) return;
    const int64_t e = t / 10;
    int32_t nd[10];
#pragma unroll
    for (int b = 0; b < 10; b++) nd[b] = el[e * 10 + b];
    const int32_t slot = slotof[el[t]];
    const int64_t s = slot >> 6, l = slot & 63;
    const int64_t p0 = sptr[s], L = sptr[s + 1] - p0;
    const bool odd = L & 1;
    uint32_t out[10];
#pragma unroll
    for (int b = 0; b < 10; b++) out[b] = 0xffffffffu;
    int found = 0;
    for (int64_t j = 0; j < L && found != 0x3ff; j++) {
        const bool single = odd && j == L - 1;
        const int64_t unit = p0 + (single ? j : (j & ~(int64_t)1));
        const int32_t c = cols[single ? unit * 64 + l : unit * 64 + 2 * l + (j & 1)];
        const uint32_t loc = (uint32_t)(((uint64_t)unit << 2) | (single ? 1u : (uint32_t)((j & 1) << 1)));
#pragma unroll
        for (int b = 0; b < 10; b++)
            if (!((found >> b) & 1) && nd[b] == c) { out[b] = loc; found |= 1 << b; }
    }
#pragma unroll
    for (int b = 0; b < 10; b++) emap[t * 10 + b] = out[b];
}

// Element map of the assembly into the symmetric window form (k_assemble_lds<.., SYM>): one byte per element block (a, b), the
// KEPT number of the block in the row of node a (WinSym::kmap, fcvm_pattern.h) or 0xFF when its transpose is the stored one.
// From the sliced-ELL map: its entry gives the block's number in the row (both layouts keep a row's blocks in column order).
__global__ __launch_bounds__(256) void k_build_emap_sym(int64_t ne, const int32_t *__restrict__ el, const int32_t *__restrict__ slotof,
                                                        const int64_t *__restrict__ sptr, const int64_t *__restrict__ rowptr,
                                                        const uint32_t *__restrict__ emap, const uint8_t *__restrict__ kmap,
                                                        uint8_t *__restrict__ emap_sym) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ne * 10) return;
    const int32_t i = el[t];
    const int64_t jb = sptr[slotof[i] >> 6], base = rowptr[i];
    uint8_t out[10];
#pragma unroll
    for (int b = 0; b < 10; b++) {
        const uint32_t em = emap[t * 10 + b];
        const int64_t jj = (int64_t)(em >> 2) - jb + (int64_t)((em >> 1) & 1u);
        out[b] = kmap[base + jj];
    }
#pragma unroll
    for (int b = 0; b < 10; b++) emap_sym[t * 10 + b] = out[b];
}

// the 64-bit form (Pattern::abs_offset: offset << 1 | single) of tables that are read without knowing the lane
__device__ __forceinline__ int64_t unpack_abs(uint64_t loc, int &stride) {
    stride = (loc & 1ull) ? 64 : 128;
    return (int64_t)(loc >> 1);
}

__device__ __forceinline__ void atomic_fmax_f64(double *addr, double v) {
    unsigned long long *a = (unsigned long long *)addr;
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        if (__longlong_as_double((long long)assumed) >= v) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    } while (assumed != old);
}

// ---------------------------------------------------------------------------------------------
// update_stress_load (fcVM.py:2197-2464): one work-item per (element, Gauss point); the four
// Gauss points of an element sit in four adjacent lanes and combine their internal-force
// contributions with two lane shuffles.
// ---------------------------------------------------------------------------------------------
struct MatConst {
    double D11, D12, D44, G, H, Et, rho;
};

// GATHER = false: internal forces scattered into qin with f64 atomics.  GATHER = true: the element's 10 x 3 nodal
// forces go to fe[e][30] and k_gather_qin sums them per node in a fixed order (bitwise reproducible residuals).
template <bool LD, bool GATHER>
__global__ __launch_bounds__(256, 4) void k_update_stress(int64_t ne, const int32_t *__restrict__ el,
                                                       const double *__restrict__ coord,
                                                       const double *__restrict__ disp_new,
                                                       const double *__restrict__ du,
                                                       const double *__restrict__ sig_old,
                                                       double *__restrict__ sig_new,
                                                       double *__restrict__ sig_test,
                                                       const double *__restrict__ sig_yield,
                                                       uint8_t *__restrict__ pgp, double *__restrict__ qin,
                                                       MatConst mc, double *__restrict__ fe) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t e = t >> 2;
    const int g = (int)(t & 3);
    const bool valid = e < ne;
    if (!valid) e = ne - 1;
    int nd[10];
#pragma unroll
    for (int k = 0; k < 10; k++) nd[k] = el[e * 10 + k];
    double xi, et, ze;
    gp_point(g, xi, et, ze);
    double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 10; k++) {
        double d0, d1, d2;
        dshp_k(k, xi, et, ze, d0, d1, d2);
        const double *c = coord + 3 * (int64_t)nd[k];
        double c0 = c[0], c1 = c[1], c2 = c[2];
        if (LD) {
            const double *u = disp_new + 3 * (int64_t)nd[k];
            c0 += u[0]; c1 += u[1]; c2 += u[2];
        }
        xs[0] += c0 * d0; xs[1] += c0 * d1; xs[2] += c0 * d2;
        xs[3] += c1 * d0; xs[4] += c1 * d1; xs[5] += c1 * d2;
        xs[6] += c2 * d0; xs[7] += c2 * d1; xs[8] += c2 * d2;
    }
    double xsi[9];
    const double xsj = jac_inverse(xs, xsi);
    double deps[6] = {0, 0, 0, 0, 0, 0};
    double F[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int k = 0; k < 10; k++) {
        double d0, d1, d2, g0, g1, g2;
        dshp_k(k, xi, et, ze, d0, d1, d2);
        to_global(xsi, d0, d1, d2, g0, g1, g2);
        const double *u = du + 3 * (int64_t)nd[k];
        const double ux = u[0], uy = u[1], uz = u[2];
        deps[0] += g0 * ux;
        deps[1] += g1 * uy;
        deps[2] += g2 * uz;
        deps[3] += g1 * ux + g0 * uy;
        deps[4] += g2 * ux + g0 * uz;
        deps[5] += g2 * uy + g1 * uz;
        if (LD) {
            F[0] += ux * g0; F[1] += ux * g1; F[2] += ux * g2;
            F[3] += uy * g0; F[4] += uy * g1; F[5] += uy * g2;
            F[6] += uz * g0; F[7] += uz * g1; F[8] += uz * g2;
        }
    }
    const int64_t ipp = 4 * e + g, ippos = 24 * e + 6 * g;
    double sc[6];
#pragma unroll
    for (int j = 0; j < 6; j++) sc[j] = sig_old[ippos + j];
    if (LD) {  // convected stress  rr * F S F^T  (fcVM.py:2383-2429)
        double S[9] = {sc[0], sc[3], sc[4], sc[3], sc[1], sc[5], sc[4], sc[5], sc[2]};
        double rr = F[0] * F[4] * F[8] - F[0] * F[5] * F[7] + F[2] * F[3] * F[7] - F[2] * F[4] * F[6] +
                    F[1] * F[5] * F[6] - F[1] * F[3] * F[8];
        rr = 1.0 / rr;
        double FS[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int l = 0; l < 3; l++) FS[i * 3 + l] = F[i * 3] * S[l] + F[i * 3 + 1] * S[3 + l] + F[i * 3 + 2] * S[6 + l];
        double C[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) C[i * 3 + k] = FS[i * 3] * F[k * 3] + FS[i * 3 + 1] * F[k * 3 + 1] + FS[i * 3 + 2] * F[k * 3 + 2];
        sc[0] = rr * C[0]; sc[1] = rr * C[4]; sc[2] = rr * C[8];
        sc[3] = rr * C[1]; sc[4] = rr * C[2]; sc[5] = rr * C[5];
    }
    double st[6];
    st[0] = sc[0] + mc.D11 * deps[0] + mc.D12 * deps[1] + mc.D12 * deps[2];
    st[1] = sc[1] + mc.D12 * deps[0] + mc.D11 * deps[1] + mc.D12 * deps[2];
    st[2] = sc[2] + mc.D12 * deps[0] + mc.D12 * deps[1] + mc.D11 * deps[2];
    st[3] = sc[3] + mc.D44 * deps[3];
    st[4] = sc[4] + mc.D44 * deps[4];
    st[5] = sc[5] + mc.D44 * deps[5];
    // radial return (fcVM.py:2469-2492)
    const double p = (st[0] + st[1] + st[2]) / 3.0;
    const double s0 = st[0] - p, s1 = st[1] - p, s2 = st[2] - p;
    const double svm = sqrt(1.5 * (s0 * s0 + s1 * s1 + s2 * s2) + 3.0 * (st[3] * st[3] + st[4] * st[4] + st[5] * st[5]));
    const double sy = sig_yield[ipp];
    double fac = 1.0;
    uint8_t pp = 0;
    if (!(sy > svm)) {
        fac = 1.0 - (1.0 - sy / svm) * 3.0 * mc.G / (mc.H + 3 * mc.G);
        pp = 1;
    }
    double sn[6] = {fac * s0 + p, fac * s1 + p, fac * s2 + p, fac * st[3], fac * st[4], fac * st[5]};
    if (valid) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            sig_test[ippos + j] = st[j];
            sig_new[ippos + j] = sn[j];
        }
        pgp[ipp] = pp;
    }
    const double w = FCVM_GPW * fabs(xsj);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        double d0, d1, d2, g0, g1, g2;
        dshp_k(k, xi, et, ze, d0, d1, d2);
        to_global(xsi, d0, d1, d2, g0, g1, g2);
        double f0 = (g0 * sn[0] + g1 * sn[3] + g2 * sn[4]) * w;
        double f1 = (g1 * sn[1] + g0 * sn[3] + g2 * sn[5]) * w;
        double f2 = (g2 * sn[2] + g0 * sn[4] + g1 * sn[5]) * w;
        f0 += __shfl_xor(f0, 1, 64); f1 += __shfl_xor(f1, 1, 64); f2 += __shfl_xor(f2, 1, 64);
        f0 += __shfl_xor(f0, 2, 64); f1 += __shfl_xor(f1, 2, 64); f2 += __shfl_xor(f2, 2, 64);
        if (valid && g == (k & 3)) {
            if (GATHER) {
                double *q = fe + 30 * e + 3 * k;
                q[0] = f0; q[1] = f1; q[2] = f2;
            } else {
                double *q = qin + 3 * (int64_t)nd[k];
                atomic_add_f64(q, f0);
                atomic_add_f64(q + 1, f1);
                atomic_add_f64(q + 2, f2);
            }
        }
    }
}

// qin[node] = sum over the elements containing the node (ascending element order) of their nodal force fe[e][a][3]
__global__ __launch_bounds__(256) void k_gather_qin(int64_t nn, const int64_t *__restrict__ nptr, const int32_t *__restrict__ nel,
                                                    const double *__restrict__ fe, double *__restrict__ qin) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int64_t q = nptr[i]; q < nptr[i + 1]; q++) {
        const int32_t packed = nel[q];
        const double *f = fe + 30 * (int64_t)(packed >> 4) + 3 * (packed & 15);
        a0 += f[0]; a1 += f[1]; a2 += f[2];
    }
    qin[3 * i] = a0; qin[3 * i + 1] = a1; qin[3 * i + 2] = a2;
}

// ---------------------------------------------------------------------------------------------
// calcGSM / calcTSM element stiffness + assembly (fcVM.py:729-800, 940-1050): one work-item per
// (element, row node a); it integrates the ten 3x3 blocks K_ab over the four Gauss points and
// scatters them into the sliced-ELL matrix with the boundary conditions folded in.
// ---------------------------------------------------------------------------------------------
// The calling lane is the only one that ever touches row nd[a] (k_assemble_rows), so the matrix blocks are plain
// read-modify-writes in a fixed element order (bitwise reproducible) and the row's gravity / modf / volume sums stay in
// the caller's registers.  lane = the row's lane in its slice (slot & 63).
template <bool TANGENT>
__device__ __forceinline__ void assemble_element_row(int64_t e, const int a, const int lane, const int32_t *__restrict__ el,
                                                  const double *__restrict__ coord,
                                                  const double *__restrict__ disp,
                                                  const uint8_t *__restrict__ fixmask,
                                                  const double *__restrict__ fixval,
                                                  const uint32_t *__restrict__ emap,
                                                  const double *__restrict__ sig,
                                                  const uint8_t *__restrict__ pgp, MatConst mc, double gx,
                                                  double gy, double gz, double *__restrict__ vals,
                                                  double *__restrict__ xgp, int penalty_bc,
                                                  double *racc) {
    int nd[10];
    double cx[10], cy[10], cz[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        nd[k] = el[e * 10 + k];
        const double *c = coord + 3 * (int64_t)nd[k];
        cx[k] = c[0]; cy[k] = c[1]; cz[k] = c[2];
        if (disp) {
            const double *u = disp + 3 * (int64_t)nd[k];
            cx[k] += u[0]; cy[k] += u[1]; cz[k] += u[2];
        }
    }
    double xsi[4][9], w[4], ga[4][3];
    double S[4][6], pf[4], ta[4][3];
    double vol = 0.0, gam = 0.0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        double xi, et, ze;
        gp_point(g, xi, et, ze);
        double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 10; k++) {
            double d0, d1, d2;
            dshp_k(k, xi, et, ze, d0, d1, d2);
            xs[0] += cx[k] * d0; xs[1] += cx[k] * d1; xs[2] += cx[k] * d2;
            xs[3] += cy[k] * d0; xs[4] += cy[k] * d1; xs[5] += cy[k] * d2;
            xs[6] += cz[k] * d0; xs[7] += cz[k] * d1; xs[8] += cz[k] * d2;
        }
        const double xsj = jac_inverse(xs, xsi[g]);
        w[g] = FCVM_GPW * fabs(xsj);
        vol += xsj * FCVM_GPW;
        double d0, d1, d2;
        dshp_k(a, xi, et, ze, d0, d1, d2);
        to_global(xsi[g], d0, d1, d2, ga[g][0], ga[g][1], ga[g][2]);
        gam += w[g] * shp_k(a, xi, et, ze);
        if (TANGENT) {  // fcVM.py:983-998
            pf[g] = 0.0;
            if (pgp[4 * e + g]) {
                const double *s0 = sig + 24 * e + 6 * g;
                double s[6] = {s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
                const double p = (s[0] + s[1] + s[2]) / 3.0;
                s[0] -= p; s[1] -= p; s[2] -= p;
                double svm = sqrt(1.5 * (s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) + 3.0 * (s[3] * s[3] + s[4] * s[4] + s[5] * s[5]));
                if (svm == 0.0) svm = 1.0;
                pf[g] = 3.0 * mc.G / (1.0 + mc.H / 3.0 / mc.G) / (svm * svm);
#pragma unroll
                for (int j = 0; j < 6; j++) S[g][j] = s[j];
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) S[g][j] = 0.0;
            }
            ta[g][0] = ga[g][0] * S[g][0] + ga[g][1] * S[g][3] + ga[g][2] * S[g][4];
            ta[g][1] = ga[g][1] * S[g][1] + ga[g][0] * S[g][3] + ga[g][2] * S[g][5];
            ta[g][2] = ga[g][2] * S[g][2] + ga[g][0] * S[g][4] + ga[g][1] * S[g][5];
        }
    }
    // gravity load (fcVM.py:757-767) and Gauss point coordinates (fcVM.py:761)
    racc[0] += gx * mc.rho * gam; racc[1] += gy * mc.rho * gam; racc[2] += gz * mc.rho * gam;
    if (xgp && a < 4) {
        double xi, et, ze;
        gp_point(a, xi, et, ze);
        double px = 0, py = 0, pz = 0;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const double n = shp_k(k, xi, et, ze);
            px += cx[k] * n; py += cy[k] * n; pz += cz[k] * n;
        }
        double *o = xgp + 3 * (4 * e + a);
        o[0] = px; o[1] = py; o[2] = pz;
    }
    if (a == 0) racc[6] += vol;
    const int fa = fixmask[nd[a]];
    double macc[3] = {0, 0, 0};
    const uint32_t *em = emap + e * 100 + a * 10;
#pragma unroll
    for (int b = 0; b < 10; b++) {
        double K[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < 4; g++) {
            double xi, et, ze, d0, d1, d2, gb[3];
            gp_point(g, xi, et, ze);
            dshp_k(b, xi, et, ze, d0, d1, d2);
            to_global(xsi[g], d0, d1, d2, gb[0], gb[1], gb[2]);
            const double dot = ga[g][0] * gb[0] + ga[g][1] * gb[1] + ga[g][2] * gb[2];
            double tb[3];
            if (TANGENT) {
                tb[0] = gb[0] * S[g][0] + gb[1] * S[g][3] + gb[2] * S[g][4];
                tb[1] = gb[1] * S[g][1] + gb[0] * S[g][3] + gb[2] * S[g][5];
                tb[2] = gb[2] * S[g][2] + gb[0] * S[g][4] + gb[1] * S[g][5];
            }
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    double v;
                    if (r == c) v = mc.D11 * ga[g][r] * gb[r] + mc.D44 * (dot - ga[g][r] * gb[r]);
                    else v = mc.D12 * ga[g][r] * gb[c] + mc.D44 * ga[g][c] * gb[r];
                    if (TANGENT) v -= pf[g] * ta[g][r] * tb[c];
                    K[r * 3 + c] += w[g] * v;
                }
        }
        const int fb = fixmask[nd[b]];
        int st;
        double *dst = vals + unpack_offset(em[b], lane, st);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const bool rfix = (fa >> r) & 1;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const bool cfix = (fb >> c) & 1;
                double add = 0.0;
                bool has = false;
                if (penalty_bc) {  // linear-buckling matrices (fcVM.py:1051-1062): no elimination, fixed diagonal x 100
                    add = (rfix && a == b && r == c) ? 100.0 * K[r * 3 + c] : K[r * 3 + c];
                    has = true;
                } else if (!rfix && !cfix) {
                    add = K[r * 3 + c];
                    has = true;
                } else if (!rfix && cfix) {
                    macc[r] -= K[r * 3 + c] * fixval[3 * (int64_t)nd[b] + c];
                } else if (rfix && a == b && r == c) {  // fcVM.py:773-778
                    add = 1.0;
                    has = true;
                    macc[r] += fixval[3 * (int64_t)nd[a] + r];
                }
                if (has) dst[(r * 3 + c) * st] += add;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 3; r++) racc[3 + r] += macc[r];
}

// Atomic-free assembly: one lane per matrix row (slot of the sliced-ELL layout, i.e. one node), walking the elements
// that contain the node (nptr / nel: element << 4 | local index, ascending element order).  The lane owns the storage of
// its row, so contributions are added with plain loads / stores in a fixed order -- the assembled matrix, gravity vector
// and modf are bitwise reproducible -- and the 64 lanes of a wave work inside one slice (its blocks are L2 resident).
template <bool TANGENT>
__global__ __launch_bounds__(256) void k_assemble_rows(int64_t nslots, const int32_t *__restrict__ rownode,
                                                       const int64_t *__restrict__ nptr, const int32_t *__restrict__ nel,
                                                       const int32_t *__restrict__ el, const double *__restrict__ coord,
                                                       const double *__restrict__ disp, const uint8_t *__restrict__ fixmask,
                                                       const double *__restrict__ fixval, const uint32_t *__restrict__ emap,
                                                       const double *__restrict__ sig, const uint8_t *__restrict__ pgp, MatConst mc,
                                                       double gx, double gy, double gz, double *__restrict__ vals,
                                                       double *__restrict__ glv, double *__restrict__ modf, double *__restrict__ xgp,
                                                       double *__restrict__ volume, int penalty_bc) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int node = slot < nslots ? rownode[slot] : -1;
    double racc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (node >= 0) {
        for (int64_t q = nptr[node]; q < nptr[node + 1]; q++) {
            const int32_t packed = nel[q];
            assemble_element_row<TANGENT>((int64_t)(packed >> 4), packed & 15, (int)(slot & 63), el, coord, disp, fixmask, fixval, emap, sig, pgp,
                                          mc, gx, gy, gz, vals, xgp, penalty_bc, racc);
        }
        glv[3 * (int64_t)node] = racc[0]; glv[3 * (int64_t)node + 1] = racc[1]; glv[3 * (int64_t)node + 2] = racc[2];
        modf[3 * (int64_t)node] = racc[3]; modf[3 * (int64_t)node + 1] = racc[4]; modf[3 * (int64_t)node + 2] = racc[5];
    }
    if (volume) {   // one partial per wavefront (= slice), summed in a fixed order by k_sum_final
        const double v = wave_sum(racc[6]);
        if ((threadIdx.x & 63) == 0) volume[slot >> 6] = v;
    }
}

// inverse of the 3x3 diagonal blocks -> dinv[node*9]
__global__ __launch_bounds__(256) void k_diag_get(int64_t nn, const uint64_t *__restrict__ diagoff,
                                                  const double *__restrict__ vals, double *__restrict__ dblk) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    int st;
    const double *v = vals + unpack_abs(diagoff[i], st);
#pragma unroll
    for (int c = 0; c < 9; c++) dblk[i * 9 + c] = v[c * st];
}

__global__ __launch_bounds__(256) void k_diag_invert(int64_t nn, const double *__restrict__ dblk, double *__restrict__ dinv) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double m[9];
#pragma unroll
    for (int c = 0; c < 9; c++) m[c] = dblk[i * 9 + c];
    double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    double r = 1.0 / det;
    double *o = dinv + i * 9;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
}

// ---------------------------------------------------------------------------------------------
// y = K x : sliced-ELL, 3x3 blocks, one wavefront per slice of 64 node rows, one row per lane.
// Per pair of blocks a lane issues nine 16-byte loads (fully coalesced: 1 KiB per wave
// instruction), two 4-byte column loads (coalesced) and two 24-byte gathers of x.
// DOT: also emits the per-workgroup partial of sum_i x_i y_i (p.Ap of conjugate gradients).
// ---------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ double2 ld_val(const double2 *p) {
    if (NT) {
        double2 r;
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
        return r;
    }
    return *p;
}
// matrix stored in FP32 (coarse multigrid level): widened to FP64 in registers
template <bool NT>
__device__ __forceinline__ double2 ld_val(const float2 *p) {
    float2 f;
    if (NT) {
        f.x = __builtin_nontemporal_load(&p->x);
        f.y = __builtin_nontemporal_load(&p->y);
    } else {
        f = *p;
    }
    double2 r;
    r.x = (double)f.x;
    r.y = (double)f.y;
    return r;
}
template <bool NT>
__device__ __forceinline__ int32_t ld_col(const int32_t *p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// one block pair of a lane: the two column nodes (one 8-byte load), nine 16-byte value loads, two 24-byte x gathers.
// GATHER = 1: every gathered node is fetched with one 8-byte and one 16-byte load instead of three 8-byte ones (node
// numbers are 24 bytes apart: even nodes are 16-byte aligned, odd ones 8 bytes off, so the 16-byte load takes
// (x, y) of an even and (y, z) of an odd node).
template <bool NT, class V2, int GATHER>
__device__ __forceinline__ void spmv_pair(const int32_t *cp, const V2 *vp, const double *__restrict__ x,
                                          double &y0, double &y1, double &y2) {
    int2 cc;
    if (NT) {
        cc.x = __builtin_nontemporal_load(cp);
        cc.y = __builtin_nontemporal_load(cp + 1);
    } else {
        cc = *(const int2 *)cp;
    }
    const int32_t c0 = cc.x, c1 = cc.y;
    double2 v[9];
#pragma unroll
    for (int c = 0; c < 9; c++) v[c] = ld_val<NT>(vp + c * 64);
    double a0, a1, a2, b0, b1, b2;
    const double *xa = x + 3 * (int64_t)c0, *xb = x + 3 * (int64_t)c1;
    if (GATHER == 1) {
        const int oa = c0 & 1, ob = c1 & 1;
        const double2 wa = *(const double2 *)(xa + oa), wb = *(const double2 *)(xb + ob);
        const double sa = xa[oa ? 0 : 2], sb = xb[ob ? 0 : 2];
        a0 = oa ? sa : wa.x; a1 = oa ? wa.x : wa.y; a2 = oa ? wa.y : sa;
        b0 = ob ? sb : wb.x; b1 = ob ? wb.x : wb.y; b2 = ob ? wb.y : sb;
    } else {
        a0 = xa[0]; a1 = xa[1]; a2 = xa[2];
        b0 = xb[0]; b1 = xb[1]; b2 = xb[2];
    }
    y0 += v[0].x * a0 + v[1].x * a1 + v[2].x * a2 + v[0].y * b0 + v[1].y * b1 + v[2].y * b2;
    y1 += v[3].x * a0 + v[4].x * a1 + v[5].x * a2 + v[3].y * b0 + v[4].y * b1 + v[5].y * b2;
    y2 += v[6].x * a0 + v[7].x * a1 + v[8].x * a2 + v[6].y * b0 + v[7].y * b1 + v[8].y * b2;
}

// the single last block of an odd-length slice: nine 8-byte (4-byte) value loads, one column node, one gather
template <bool NT, class V2, int GATHER>
__device__ __forceinline__ void spmv_single(const int32_t *cp, const V2 *vp2, const double *__restrict__ x,
                                            double &y0, double &y1, double &y2) {
    typedef decltype(V2().x) T;                    // double or float
    const T *vp = (const T *)vp2;
    const int32_t c0 = ld_col<NT>(cp);
    double v[9];
#pragma unroll
    for (int c = 0; c < 9; c++) v[c] = (double)(NT ? __builtin_nontemporal_load(vp + c * 64) : vp[c * 64]);
    const double *xa = x + 3 * (int64_t)c0;
    double a0, a1, a2;
    if (GATHER == 1) {
        const int oa = c0 & 1;
        const double2 wa = *(const double2 *)(xa + oa);
        const double sa = xa[oa ? 0 : 2];
        a0 = oa ? sa : wa.x; a1 = oa ? wa.x : wa.y; a2 = oa ? wa.y : sa;
    } else {
        a0 = xa[0]; a1 = xa[1]; a2 = xa[2];
    }
    y0 += v[0] * a0 + v[1] * a1 + v[2] * a2;
    y1 += v[3] * a0 + v[4] * a1 + v[5] * a2;
    y2 += v[6] * a0 + v[7] * a1 + v[8] * a2;
}

// Optional epilogue of the SpMV (EPI = 1): one Chebyshev step of the multigrid smoother on the row's own node,
// with t = (A d)_node still in registers (no t vector, no second kernel):
//   x += d ; r -= t ; if (update_d) { dout = cdd * d + cdr * Dinv r ; if (add_dout) x += dout }
// d (the SpMV input, gathered by other rows while this kernel runs) is not written: dout != d.
struct ChebEpi {
    const double *dinv;
    double *x, *r, *dout;
    double cdd, cdr;
    int update_d, add_dout;
    const float *dinv6 = nullptr;   // optional: the six distinct entries of every inverse diagonal block in FP32 (24 B instead of 72 B)
    // multi-GPU layer (k_spmv_l1_i16 only): rows of nodes shared with other ranks hold a PART of the product -- their step waits for the
    // exchange: the raw product goes to t_out and the row is left alone (k_unpack_sum_cheb finishes it); t2: a second, pending product
    // (this rank's part) added to this one at every row -- the last pre-smoothing product that rides on the first post-smoothing exchange
    const uint8_t *defer = nullptr;
    double *t_out = nullptr;
    const double *t2 = nullptr;
    // round 5 (k_spmv_l1_i16 only): FP32 twin of dout, one 16-byte record (x, y, z, 0) per node -- the next pass gathers from it
    float *dout32 = nullptr;
};
__device__ __forceinline__ void apply_dinv(const double *d, double r0, double r1, double r2, double &z0, double &z1, double &z2);

template <bool DOT, bool NT, int UNROLL, class V2 = double2, int EPI = 0, int GATHER = 0>
__global__ __launch_bounds__(256, (sizeof(V2) == 16 ? 4 : 1)) void k_spmv(int64_t nslices, int64_t nblocks, const int64_t *__restrict__ sptr,
                                              const int32_t *__restrict__ cols,
                                              const V2 *__restrict__ vals,
                                              const int32_t *__restrict__ rownode,
                                              const double *__restrict__ x, double *__restrict__ y,
                                              double *__restrict__ partial, const double *__restrict__ done,
                                              ChebEpi epi) {
    __shared__ double sh[4];
    if (done && *done != 0.0) return;
    // XCD-aware remap: workgroups b, b+8, b+16.. share an XCD (and its L2); give each XCD a
    // contiguous range of slices so that the x-gathers of neighbouring slices hit the same L2.
    const int64_t b = blockIdx.x;
    const int64_t q = nblocks >> 3, r = nblocks & 7, xcd = b & 7;
    const int64_t wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int lane = threadIdx.x & 63;
    // one slice per wavefront: the slice number is wave-uniform (scalar registers, scalar loads of the slice table)
    const int64_t s = wg * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    int node = -1;
    if (s < nslices) {
        // a slice of len blocks = len half chunks of 9 x 64 scalars, starting at half chunk sptr[s]: len / 2 block pairs
        // (two half chunks each: nine double2 components) and, when len is odd, one single block
        const int64_t jb = sptr[s], je = sptr[s + 1];
        const int32_t *cp = cols + jb * 64 + 2 * lane;   // two column nodes per lane and pair, adjacent
        const V2 *vp = (const V2 *)((const decltype(V2().x) *)vals + jb * (9 * 64)) + lane;
        int64_t np = (je - jb) >> 1;
        const bool odd = (je - jb) & 1;
        if (UNROLL == 4) {
            double z0 = 0.0, z1 = 0.0, z2 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0, w0 = 0.0, w1 = 0.0, w2 = 0.0;
            for (; np >= 4; np -= 4) {
                spmv_pair<NT, V2, GATHER>(cp, vp, x, y0, y1, y2);
                spmv_pair<NT, V2, GATHER>(cp + 128, vp + 9 * 64, x, z0, z1, z2);
                spmv_pair<NT, V2, GATHER>(cp + 256, vp + 18 * 64, x, u0, u1, u2);
                spmv_pair<NT, V2, GATHER>(cp + 384, vp + 27 * 64, x, w0, w1, w2);
                cp += 512;
                vp += 36 * 64;
            }
            y0 += z0 + u0 + w0; y1 += z1 + u1 + w1; y2 += z2 + u2 + w2;
        }
        if (UNROLL >= 2) {
            double z0 = 0.0, z1 = 0.0, z2 = 0.0;
            for (; np >= 2; np -= 2) {
                spmv_pair<NT, V2, GATHER>(cp, vp, x, y0, y1, y2);
                spmv_pair<NT, V2, GATHER>(cp + 128, vp + 9 * 64, x, z0, z1, z2);
                cp += 256;
                vp += 18 * 64;
            }
            y0 += z0; y1 += z1; y2 += z2;
        }
        for (; np > 0; np--) {
            spmv_pair<NT, V2, GATHER>(cp, vp, x, y0, y1, y2);
            cp += 128;
            vp += 9 * 64;
        }
        if (odd)   // cp, vp point at the first lane of the single block: step back to this lane's scalar slot
            spmv_single<NT, V2, GATHER>(cp - lane, (const V2 *)((const decltype(V2().x) *)(vp - lane) + lane), x, y0, y1, y2);
        node = rownode[s * 64 + lane];
        if (EPI == 0) {
            if (node >= 0) {
                double *yo = y + 3 * (int64_t)node;
                yo[0] = y0; yo[1] = y1; yo[2] = y2;   // (nontemporal y stores: 0.641 against 0.617 ms, profiles/r02m_spmv_nt_y_ab.txt)
            }
        } else if (node >= 0) {
            const int64_t o = 3 * (int64_t)node;
            const double d0 = x[o], d1 = x[o + 1], d2 = x[o + 2];
            const double r0 = epi.r[o] - y0, r1 = epi.r[o + 1] - y1, r2 = epi.r[o + 2] - y2;
            epi.r[o] = r0; epi.r[o + 1] = r1; epi.r[o + 2] = r2;
            double a0 = d0, a1 = d1, a2 = d2;
            if (epi.update_d) {
                double z0, z1, z2;
                if (epi.dinv6) {
                    const float2 *q2 = (const float2 *)(epi.dinv6 + 6 * (int64_t)node);
                    const float2 u0 = q2[0], u1 = q2[1], u2 = q2[2];   // xx xy | xz yy | yz zz
                    z0 = (double)u0.x * r0 + (double)u0.y * r1 + (double)u1.x * r2;
                    z1 = (double)u0.y * r0 + (double)u1.y * r1 + (double)u2.x * r2;
                    z2 = (double)u1.x * r0 + (double)u2.x * r1 + (double)u2.y * r2;
                } else {
                    apply_dinv(epi.dinv + (int64_t)node * 9, r0, r1, r2, z0, z1, z2);
                }
                const double n0 = epi.cdd * d0 + epi.cdr * z0, n1 = epi.cdd * d1 + epi.cdr * z1, n2 = epi.cdd * d2 + epi.cdr * z2;
                epi.dout[o] = n0; epi.dout[o + 1] = n1; epi.dout[o + 2] = n2;
                if (epi.add_dout) { a0 += n0; a1 += n1; a2 += n2; }
            }
            epi.x[o] += a0; epi.x[o + 1] += a1; epi.x[o + 2] += a2;
        }
    }
    if (DOT) {
        double d = 0.0;
        if (node >= 0) {
            const double *xo = x + 3 * (int64_t)node;
            d = y0 * xo[0] + y1 * xo[1] + y2 * xo[2];
        }
        d = block_sum(d, sh);
        if (threadIdx.x == 0) partial[wg] = d;
    }
}

// ---------------------------------------------------------------------------------------------
// PCG scalars live on the device: S[0]=rz S[1]=pq S[2]=alpha S[3]=beta S[4]=rr S[5]=thresh
// S[6]=done S[7]=iterations S[8]=bb
// ---------------------------------------------------------------------------------------------
enum { S_RZ = 0, S_PQ, S_ALPHA, S_BETA, S_RR, S_THRESH, S_DONE, S_ITERS, S_BB, S_COUNT = 16 };

__device__ __forceinline__ void apply_dinv(const double *d, double r0, double r1, double r2, double &z0, double &z1, double &z2) {
    z0 = d[0] * r0 + d[1] * r1 + d[2] * r2;
    z1 = d[3] * r0 + d[4] * r1 + d[5] * r2;
    z2 = d[6] * r0 + d[7] * r1 + d[8] * r2;
}

__global__ __launch_bounds__(256) void k_precond(int64_t nn, const double *__restrict__ dinv,
                                                 const double *__restrict__ r, double *__restrict__ z) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nn) return;
    double z0, z1, z2;
    apply_dinv(dinv + i * 9, r[3 * i], r[3 * i + 1], r[3 * i + 2], z0, z1, z2);
    z[3 * i] = z0; z[3 * i + 1] = z1; z[3 * i + 2] = z2;
}

// x = 0, r = b, z = Dinv r, p = z ; partials of r.z and r.r
__global__ __launch_bounds__(256) void k_pcg_init(int64_t nn, const double *__restrict__ dinv,
                                                  const double *__restrict__ b, double *__restrict__ x,
                                                  double *__restrict__ r, double *__restrict__ z,
                                                  double *__restrict__ p, double *__restrict__ part, int64_t nb) {
    __shared__ double sh[4];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr = 0.0;
    if (i < nn) {
        double r0 = b[3 * i], r1 = b[3 * i + 1], r2 = b[3 * i + 2], z0, z1, z2;
        apply_dinv(dinv + i * 9, r0, r1, r2, z0, z1, z2);
        x[3 * i] = 0.0; x[3 * i + 1] = 0.0; x[3 * i + 2] = 0.0;
        r[3 * i] = r0; r[3 * i + 1] = r1; r[3 * i + 2] = r2;
        z[3 * i] = z0; z[3 * i + 1] = z1; z[3 * i + 2] = z2;
        p[3 * i] = z0; p[3 * i + 1] = z1; p[3 * i + 2] = z2;
        rz = r0 * z0 + r1 * z1 + r2 * z2;
        rr = r0 * r0 + r1 * r1 + r2 * r2;
    }
    rz = block_sum(rz, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = rz;
        part[nb + blockIdx.x] = rr;
    }
}

// x += alpha p ; r -= alpha q ; z = Dinv r ; partials of r.z and r.r
__global__ __launch_bounds__(256) void k_pcg_update_xr(int64_t nn, const double *__restrict__ dinv,
                                                       const double *__restrict__ p, const double *__restrict__ q,
                                                       double *__restrict__ x, double *__restrict__ r,
                                                       double *__restrict__ z, double *__restrict__ part,
                                                       int64_t nb, const double *__restrict__ S) {
    __shared__ double sh[4];
    if (S[S_DONE] != 0.0) return;
    const double alpha = S[S_ALPHA];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr = 0.0;
    if (i < nn) {
        double r0 = r[3 * i] - alpha * q[3 * i], r1 = r[3 * i + 1] - alpha * q[3 * i + 1], r2 = r[3 * i + 2] - alpha * q[3 * i + 2];
        x[3 * i] += alpha * p[3 * i]; x[3 * i + 1] += alpha * p[3 * i + 1]; x[3 * i + 2] += alpha * p[3 * i + 2];
        double z0, z1, z2;
        apply_dinv(dinv + i * 9, r0, r1, r2, z0, z1, z2);
        r[3 * i] = r0; r[3 * i + 1] = r1; r[3 * i + 2] = r2;
        z[3 * i] = z0; z[3 * i + 1] = z1; z[3 * i + 2] = z2;
        rz = r0 * z0 + r1 * z1 + r2 * z2;
        rr = r0 * r0 + r1 * r1 + r2 * r2;
    }
    rz = block_sum(rz, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = rz;
        part[nb + blockIdx.x] = rr;
    }
}

__global__ __launch_bounds__(256) void k_pcg_update_p(int64_t n, const double *__restrict__ z, double *__restrict__ p, const double *__restrict__ S) {
    if (S[S_DONE] != 0.0) return;
    const double beta = S[S_BETA];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = z[i] + beta * p[i];
}
// fixed-order sum of n partials (single workgroup); result valid in thread 0
__device__ __forceinline__ double sum_partials(const double *part, int64_t n, double *sh) {
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    return block_sum(v, sh);
}

// mode 0: after init (rz, rr, threshold); mode 1: after SpMV (pq -> alpha); mode 2: after
// update_xr (rz_new, rr -> beta, convergence)
__global__ __launch_bounds__(1024) void k_pcg_scalars(int mode, const double *__restrict__ part, int64_t nb_vec, int64_t nb_spmv,
                                                      double *__restrict__ S, double rtol) {
    __shared__ double sh[16];   // launched with 1024 threads: the SpMV leaves one partial per slice (21 k at 1 M elements)
    if (mode != 0 && S[S_DONE] != 0.0) return;
    if (mode == 1) {
        double pq = sum_partials(part, nb_spmv, sh);
        if (threadIdx.x == 0) {
            S[S_PQ] = pq;
            S[S_ALPHA] = S[S_RZ] / pq;
        }
        return;
    }
    double rz = sum_partials(part, nb_vec, sh);
    double rr = sum_partials(part + nb_vec, nb_vec, sh);
    if (threadIdx.x == 0) {
        if (mode == 0) {
            S[S_RZ] = rz; S[S_RR] = rr; S[S_BB] = rr;
            S[S_THRESH] = rtol * rtol * rr;
            S[S_ITERS] = 0.0;
            S[S_DONE] = (rr == 0.0) ? 1.0 : 0.0;
        } else {
            S[S_BETA] = rz / S[S_RZ];
            S[S_RZ] = rz; S[S_RR] = rr;
            S[S_ITERS] += 1.0;
            if (rr <= S[S_THRESH]) S[S_DONE] = 1.0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// small vector kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_axpby(int64_t n, double a, const double *__restrict__ x, double b, const double *__restrict__ y, double *__restrict__ o) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}
__global__ __launch_bounds__(256) void k_mul(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ o) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = x[i] * y[i];
}
__global__ __launch_bounds__(256) void k_dot_partial(int64_t n, const double *__restrict__ w, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ part) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        v += (w ? w[i] : 1.0) * x[i] * y[i];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_sum_final(const double *__restrict__ part, int64_t n, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = sum_partials(part, n, sh);
    if (threadIdx.x == 0) *out = v;
}
// the same with 1024 threads, for long lists (the volume partials of the assembly: one per work item, 90 k at 1 M tet10 -- 149 us
// with 256 threads in a kernel trace, on the critical path behind every assembly)
__global__ __launch_bounds__(1024) void k_sum_final_wide(const double *__restrict__ part, int64_t n, double *__restrict__ out) {
    __shared__ double sh[16];
    double v = sum_partials(part, n, sh);
    if (threadIdx.x == 0) *out = v;
}
// k <= 16 inner products x_j . y in one pass over y (the Newton driver's projection / Gram / Riks scalars: one launch and one
// host synchronisation instead of k).  Same grid, same per-thread order and the same two-stage sums as k_dot_partial /
// k_sum_final, so every result is bitwise what fcvm_vec_dot returns for that pair.
struct MdotPtrs { const double *x[16]; };
__global__ __launch_bounds__(256) void k_mdot_partial(int64_t n, int k, MdotPtrs xs, const double *__restrict__ y, double *__restrict__ part,
                                                      int64_t nb) {
    __shared__ double sh[4];
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double yi = y[i];
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < k) v[j] += 1.0 * xs.x[j][i] * yi;
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < k) {
            const double t = block_sum(v[j], sh);
            if (threadIdx.x == 0) part[j * nb + blockIdx.x] = t;
        }
}
__global__ __launch_bounds__(256) void k_msum_final(const double *__restrict__ part, int64_t nb, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = sum_partials(part + blockIdx.x * nb, nb, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_max_node_norm_partial(int64_t nnodes, const double *__restrict__ u, double *__restrict__ part) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnodes; i += (int64_t)gridDim.x * 256) {
        double a = u[3 * i], b = u[3 * i + 1], c = u[3 * i + 2];
        v = fmax(v, a * a + b * b + c * c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__global__ __launch_bounds__(256) void k_max_final(const double *__restrict__ part, int64_t n, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) v = fmax(v, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *out = sqrt(fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3])));
}

// ---------------------------------------------------------------------------------------------
// update_PEEQ_CSR (fcVM.py:2085-2137) + argmax(csr) / max(peeq) partials
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_peeq_csr(int64_t ngp, const double *__restrict__ sig_test,
                                                  const double *__restrict__ sig_new, double *__restrict__ sig_yield,
                                                  double *__restrict__ peeq, double *__restrict__ csr,
                                                  double *__restrict__ triax, double *__restrict__ pressure,
                                                  double *__restrict__ sigmises, double *__restrict__ ecr, double G,
                                                  double H, double Et, double alpha, double *__restrict__ pmaxc,
                                                  long long *__restrict__ pidx, double *__restrict__ pmaxp) {
    __shared__ double shc[256];
    __shared__ long long shi[256];
    __shared__ double shp[256];
    int64_t ip = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double mc = -1.0e300, mp = -1.0e300;
    long long mi = 0x7fffffffffffffffLL;
    if (ip < ngp) {
        const double *t = sig_test + 6 * ip, *n = sig_new + 6 * ip;
        const double pt = (t[0] + t[1] + t[2]) / 3.0, pn = (n[0] + n[1] + n[2]) / 3.0;
        const double t0 = t[0] - pt, t1 = t[1] - pt, t2 = t[2] - pt;
        const double n0 = n[0] - pn, n1 = n[1] - pn, n2 = n[2] - pn;
        const double svt = sqrt(1.5 * (t0 * t0 + t1 * t1 + t2 * t2) + 3.0 * (t[3] * t[3] + t[4] * t[4] + t[5] * t[5]));
        const double svn = sqrt(1.5 * (n0 * n0 + n1 * n1 + n2 * n2) + 3.0 * (n[3] * n[3] + n[4] * n[4] + n[5] * n[5]));
        double sy = sig_yield[ip], pq = peeq[ip], DL = 0.0;
        if (svt > sy) {
            DL = (svt - sy) / (3.0 * G + H);
            pq += DL;
            sy += Et * DL;
            peeq[ip] = pq;
            sig_yield[ip] = sy;
        }
        const double T = pn / sy;
        pressure[ip] = pn;
        sigmises[ip] = svn;
        triax[ip] = T;
        double ec = alpha * exp(-1.5 * T);
        if (ec < 1.0e-6) ec = 1.0e-6;
        ecr[ip] = ec;
        const double cs = csr[ip] + DL / ec;
        csr[ip] = cs;
        mc = cs; mi = ip; mp = pq;
    }
    shc[threadIdx.x] = mc; shi[threadIdx.x] = mi; shp[threadIdx.x] = mp;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            double oc = shc[threadIdx.x + o];
            long long oi = shi[threadIdx.x + o];
            if (oc > shc[threadIdx.x] || (oc == shc[threadIdx.x] && oi < shi[threadIdx.x])) {
                shc[threadIdx.x] = oc; shi[threadIdx.x] = oi;
            }
            shp[threadIdx.x] = fmax(shp[threadIdx.x], shp[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pmaxc[blockIdx.x] = shc[0]; pidx[blockIdx.x] = shi[0]; pmaxp[blockIdx.x] = shp[0];
    }
}

__global__ __launch_bounds__(256) void k_peeq_csr_final(int64_t nb, int64_t ngp, const double *__restrict__ pmaxc,
                                                        const long long *__restrict__ pidx, const double *__restrict__ pmaxp,
                                                        const double *__restrict__ pressure, const double *__restrict__ sigmises,
                                                        const double *__restrict__ triax, const double *__restrict__ ecr,
                                                        const double *__restrict__ peeq, double *__restrict__ out) {
    __shared__ double shc[256];
    __shared__ long long shi[256];
    __shared__ double shp[256];
    double mc = -1.0e300, mp = -1.0e300;
    long long mi = 0x7fffffffffffffffLL;
    for (int64_t i = threadIdx.x; i < nb; i += 256) {
        if (pmaxc[i] > mc || (pmaxc[i] == mc && pidx[i] < mi)) { mc = pmaxc[i]; mi = pidx[i]; }
        mp = fmax(mp, pmaxp[i]);
    }
    shc[threadIdx.x] = mc; shi[threadIdx.x] = mi; shp[threadIdx.x] = mp;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            double oc = shc[threadIdx.x + o];
            long long oi = shi[threadIdx.x + o];
            if (oc > shc[threadIdx.x] || (oc == shc[threadIdx.x] && oi < shi[threadIdx.x])) {
                shc[threadIdx.x] = oc; shi[threadIdx.x] = oi;
            }
            shp[threadIdx.x] = fmax(shp[threadIdx.x], shp[threadIdx.x + o]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        long long i = shi[0];
        out[1] = shc[0]; out[2] = shp[0];
        // a state that is NaN throughout (a diverged analysis) leaves no index: report it as such instead of reading at 2^63 - 1 (round 5)
        if (i < 0 || i >= ngp) {
            const double nan = __builtin_nan("");
            out[0] = -1.0; out[1] = nan; out[3] = nan; out[4] = nan; out[5] = nan; out[6] = nan; out[7] = nan;
        } else {
            out[0] = (double)i;
            out[3] = pressure[i]; out[4] = sigmises[i]; out[5] = triax[i]; out[6] = ecr[i]; out[7] = peeq[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// mapStresses (fcVM.py:2497-2554)
// ---------------------------------------------------------------------------------------------
// corner nodes: lane per node, gathering the Gauss-point values of the incident elements (Gauss point a of an element
// maps to its corner a, fcVM.py:2510-2540) in ascending element order -- the reference's own summation order, no atomics
__global__ __launch_bounds__(256) void k_map_corners(int64_t nn, const int64_t *__restrict__ nptr, const int32_t *__restrict__ nel,
                                                     const int32_t *__restrict__ noce, int averaged, const double *__restrict__ sig,
                                                     const double *__restrict__ peeq, const double *__restrict__ svm,
                                                     const double *__restrict__ csr, double sig_yield, double *__restrict__ t10s,
                                                     double *__restrict__ t10peeq, double *__restrict__ t10csr,
                                                     double *__restrict__ t10svm, double *__restrict__ t10triax) {
    const int64_t nd = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (nd >= nn) return;
    const double cnt = (double)noce[nd];
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double ap = 0.0, ac = 0.0, av = 0.0, at = 0.0;
    for (int64_t q = nptr[nd]; q < nptr[nd + 1]; q++) {
        const int32_t packed = nel[q];
        const int a = packed & 15;
        if (a >= 4) continue;
        const int64_t t = 4 * (int64_t)(packed >> 4) + a;
        const double *s = sig + 6 * t;
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] += s[k] / cnt;
        const double tr = (s[0] + s[1] + s[2]) / 3.0 / sig_yield;
        if (averaged) {
            ap += peeq[t] / cnt; ac += csr[t] / cnt; av += svm[t] / cnt; at += tr / cnt;
        } else {
            ap = fmax(ap, peeq[t]); ac = fmax(ac, csr[t]); av = fmax(av, svm[t]); at = fmax(at, tr);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) t10s[6 * nd + k] = acc[k];
    t10peeq[nd] = ap; t10csr[nd] = ac; t10svm[nd] = av; t10triax[nd] = at;
}

__global__ __launch_bounds__(256) void k_map_midside(int64_t ne, const int32_t *__restrict__ el, double *__restrict__ t10s,
                                                     double *__restrict__ t10peeq, double *__restrict__ t10csr,
                                                     double *__restrict__ t10svm, double *__restrict__ t10triax) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= 6 * ne) return;
    const int64_t e = t / 6;
    const int m = (int)(t - 6 * e);
    const int ia = (m == 1) ? 1 : ((m == 4) ? 1 : ((m == 5) ? 2 : 0));
    const int ib = (m == 0) ? 1 : ((m == 1 || m == 2) ? 2 : 3);
    const int64_t nd = el[e * 10 + 4 + m], a = el[e * 10 + ia], b = el[e * 10 + ib];
#pragma unroll
    for (int k = 0; k < 6; k++) t10s[6 * nd + k] = 0.5 * t10s[6 * a + k] + 0.5 * t10s[6 * b + k];
    t10peeq[nd] = 0.5 * t10peeq[a] + 0.5 * t10peeq[b];
    t10csr[nd] = 0.5 * t10csr[a] + 0.5 * t10csr[b];
    t10svm[nd] = 0.5 * t10svm[a] + 0.5 * t10svm[b];
    t10triax[nd] = 0.5 * t10triax[a] + 0.5 * t10triax[b];
}

// calculate_principal_stress (fcVM.py:2954-2994): cyclic Jacobi on the symmetric 3x3 tensor
__global__ __launch_bounds__(256) void k_principal(int64_t n, const double *__restrict__ t10s, double *__restrict__ ps, double *__restrict__ pv) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *s = t10s + 6 * i;
    double A[3][3] = {{s[0], s[3], s[4]}, {s[3], s[1], s[5]}, {s[4], s[5], s[2]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double dia = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * dia) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            if (A[p][q] == 0.0) continue;
            double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            double tt = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double akp = A[k][p], akq = A[k][q];
                A[k][p] = cs * akp - sn * akq;
                A[k][q] = sn * akp + cs * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double apk = A[p][k], aqk = A[q][k];
                A[p][k] = cs * apk - sn * aqk;
                A[q][k] = sn * apk + cs * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = cs * vkp - sn * vkq;
                V[k][q] = sn * vkp + cs * vkq;
            }
        }
    }
    double w[3] = {A[0][0], A[1][1], A[2][2]};
    int o0 = 0, o1 = 1, o2 = 2, tmp;
    if (w[o0] < w[o1]) { tmp = o0; o0 = o1; o1 = tmp; }
    if (w[o0] < w[o2]) { tmp = o0; o0 = o2; o2 = tmp; }
    if (w[o1] < w[o2]) { tmp = o1; o1 = o2; o2 = tmp; }
    const int ord[3] = {o0, o1, o2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double ev = w[ord[k]];
        ps[3 * i + k] = ev;
        for (int c = 0; c < 3; c++) pv[9 * i + 3 * k + c] = ev * V[c][ord[k]];
    }
}

// flag[0] = 1 if any byte of a is non-zero (flag zeroed by the caller; every writer stores the same value)
__global__ __launch_bounds__(256) void k_any_u8(int64_t n, const uint8_t *__restrict__ a, int *__restrict__ flag) {
    bool hit = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) hit |= a[i] != 0;
    if (__any(hit) && (threadIdx.x & 63) == 0) flag[0] = 1;
}

__global__ __launch_bounds__(256) void k_to_f32(int64_t n, const double *__restrict__ a, float *__restrict__ o) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = (float)a[i];
}

// plain streaming read (16 B per lane, nontemporal, grid-stride): the HBM rate a box delivers, for comparison with the SpMV
__global__ __launch_bounds__(256) void k_stream_read(int64_t n2, const double2 *__restrict__ a, double *__restrict__ out) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const double2 v = ld_val<true>(a + i);
        s += v.x + v.y;
    }
    if (s == 1.2345e300) out[0] = s;   // never true: keeps the loads alive
}

// the same with 8 B per lane (the load width of the window-format SpMV): calibrates the FETCH_SIZE counter for that width
__global__ __launch_bounds__(256) void k_stream_read8(int64_t n, const double *__restrict__ a, double *__restrict__ out) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += __builtin_nontemporal_load(a + i);
    if (s == 1.2345e300) out[0] = s;
}

// geometric stiffness of calcTSM(nstep = 1), fcVM.py:1002-1006: Kg_ab = (sum_gp w ga . sigma . gb) I3, scattered into a
// second value array that shares the pattern of the stiffness matrix.  One lane per matrix row walking its elements
// (as k_assemble_rows: plain read-modify-writes in a fixed order, no atomics).
__device__ __forceinline__ void geometric_element_row(const int64_t e, const int a, const int lane, const int32_t *__restrict__ el,
                                                      const double *__restrict__ coord, const uint32_t *__restrict__ emap,
                                                      const double *__restrict__ sig, double *__restrict__ vals_g) {
    int nd[10];
    double cx[10], cy[10], cz[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        nd[k] = el[e * 10 + k];
        const double *c = coord + 3 * (int64_t)nd[k];
        cx[k] = c[0]; cy[k] = c[1]; cz[k] = c[2];
    }
    double xsi[4][9], w[4], sa[4][3];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        double xi, et, ze;
        gp_point(g, xi, et, ze);
        double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 10; k++) {
            double d0, d1, d2;
            dshp_k(k, xi, et, ze, d0, d1, d2);
            xs[0] += cx[k] * d0; xs[1] += cx[k] * d1; xs[2] += cx[k] * d2;
            xs[3] += cy[k] * d0; xs[4] += cy[k] * d1; xs[5] += cy[k] * d2;
            xs[6] += cz[k] * d0; xs[7] += cz[k] * d1; xs[8] += cz[k] * d2;
        }
        const double xsj = jac_inverse(xs, xsi[g]);
        w[g] = FCVM_GPW * fabs(xsj);
        double d0, d1, d2, g0, g1, g2;
        dshp_k(a, xi, et, ze, d0, d1, d2);
        to_global(xsi[g], d0, d1, d2, g0, g1, g2);
        const double *s = sig + 24 * e + 6 * g;  // (xx, yy, zz, xy, zx, yz)
        sa[g][0] = g0 * s[0] + g1 * s[3] + g2 * s[4];
        sa[g][1] = g0 * s[3] + g1 * s[1] + g2 * s[5];
        sa[g][2] = g0 * s[4] + g1 * s[5] + g2 * s[2];
    }
    const uint32_t *em = emap + e * 100 + a * 10;
#pragma unroll
    for (int b = 0; b < 10; b++) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            double xi, et, ze, d0, d1, d2, g0, g1, g2;
            gp_point(g, xi, et, ze);
            dshp_k(b, xi, et, ze, d0, d1, d2);
            to_global(xsi[g], d0, d1, d2, g0, g1, g2);
            v += w[g] * (sa[g][0] * g0 + sa[g][1] * g1 + sa[g][2] * g2);
        }
        int st;
        double *dst = vals_g + unpack_offset(em[b], lane, st);
        dst[0] += v;
        dst[4 * st] += v;
        dst[8 * st] += v;
    }
}

__global__ __launch_bounds__(256) void k_assemble_geometric(int64_t nslots, const int32_t *__restrict__ rownode,
                                                            const int64_t *__restrict__ nptr, const int32_t *__restrict__ nel,
                                                            const int32_t *__restrict__ el, const double *__restrict__ coord,
                                                            const uint32_t *__restrict__ emap, const double *__restrict__ sig,
                                                            double *__restrict__ vals_g) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    const int node = rownode[slot];
    if (node < 0) return;
    for (int64_t q = nptr[node]; q < nptr[node + 1]; q++) {
        const int32_t packed = nel[q];
        geometric_element_row((int64_t)(packed >> 4), packed & 15, (int)(slot & 63), el, coord, emap, sig, vals_g);
    }
}

}  // namespace fcvm
