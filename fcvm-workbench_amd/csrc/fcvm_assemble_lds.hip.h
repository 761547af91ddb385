// calcGSM / calcTSM stiffness assembly with the matrix rows accumulated in LDS (fcVM.py:729-800, 940-1050).
//
// The row-owner kernel k_assemble_rows adds every 3x3 block with scattered 8-byte read-modify-writes into the
// sliced-ELL value array: 900 M of them at 1 M elements, each a different cache line for each lane of the wave, and the
// per-CU address / L1 pipeline (about 4.4 cycles per distinct line, profiles/r02*_pmc) sets its 17 ms.  Here one
// wavefront owns 16 consecutive rows of a slice (a quarter slice) and keeps their blocks in LDS (len x 72 B per row)
// until every element of every row has been visited; the finished quarter slice then leaves in one pass of coalesced
// 16-byte nontemporal stores (each value written exactly once, no memset of the 3 GB array, no atomics, fixed
// summation order => bitwise reproducible).  Four lanes share a row:
//   * lane h of the quad evaluates the Jacobian of Gauss point h of the visited element, the quad exchanges the four
//     inverse Jacobians / weights with lane shuffles (each Jacobian is computed once per visit instead of four times);
//   * lane h integrates the blocks K_ab of the local columns b = h, h + 4, h + 8 and adds them to the row's LDS image;
//   * element connectivity, node coordinates, block locations and support flags of visit v + 1 (and the
//     connectivity of visit v + 2) are in flight while visit v is integrated.
// What still goes through the scattered-access path are the inputs: 16 distinct elements per wave-instruction instead
// of 64 distinct matrix lines, 35 instead of 240 instructions per visit.
#pragma once
#include "fcvm_kernels.hip.h"

namespace fcvm {

constexpr int ASM_ROWS = 16;                       // rows per wavefront (4 lanes each)
__host__ __device__ inline int64_t asm_row_stride(int64_t len) { return len * 9 + 1; }   // doubles; odd: rows spread over the LDS banks
__host__ __device__ inline int64_t asm_lds_bytes(int64_t len) { return ASM_ROWS * asm_row_stride(len) * 8; }

__device__ __forceinline__ double quad_get(double v, int lane_in_quad) {
    return __shfl(v, (int)((threadIdx.x & 60u) | (unsigned)lane_in_quad), 64);
}

struct AsmVisit {     // what a lane needs of one (element, row node) visit
    int32_t packed;   // element << 4 | local index of the row node, -1: no visit (lanes of shorter rows)
    int nd[10];
};
struct AsmGeom {
    double cx[10], cy[10], cz[10];
    uint32_t em[3];   // packed block location of local column b = h + 4 t
    int ndb[3];       // node of that column
    int fb[3];        // its support flags
};

template <bool TANGENT>
__global__ __launch_bounds__(64) void k_assemble_lds(const int32_t *__restrict__ items, const int64_t *__restrict__ sptr,
                                                     const int32_t *__restrict__ rownode, const int64_t *__restrict__ nptr,
                                                     const int32_t *__restrict__ nel, const int32_t *__restrict__ el,
                                                     const double *__restrict__ coord, const double *__restrict__ disp,
                                                     const uint8_t *__restrict__ fixmask, const double *__restrict__ fixval,
                                                     const uint32_t *__restrict__ emap, const double *__restrict__ sig,
                                                     const uint8_t *__restrict__ pgp, MatConst mc, double gx, double gy, double gz,
                                                     double *__restrict__ vals, double *__restrict__ glv, double *__restrict__ modf,
                                                     double *__restrict__ xgp, double *__restrict__ volume, int penalty_bc) {
    extern __shared__ double rowbuf[];
    // XCD-aware remap (as k_spmv): the workgroups of one XCD take a contiguous range of work items, whose rows share
    // elements and nodes, so that their gathers meet in one L2
    const int64_t nb = gridDim.x, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    const int64_t wg = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int item = items[wg];
    const int64_t s = item >> 2;
    const int qd = item & 3;
    const int lane = threadIdx.x, rowl = lane >> 2, h = lane & 3;
    const int64_t jb = sptr[s], len = sptr[s + 1] - jb, cbase = jb >> 1;
    const int64_t S = asm_row_stride(len);
    for (int64_t i = lane; i < ASM_ROWS * S; i += 64) rowbuf[i] = 0.0;
    const int64_t slot = s * 64 + qd * ASM_ROWS + rowl;
    const int node = rownode[slot];
    const int64_t q0 = node >= 0 ? nptr[node] : 0;
    const int nvis = node >= 0 ? (int)(nptr[node + 1] - q0) : 0;
    int maxvis = nvis;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxvis = max(maxvis, __shfl_xor(maxvis, o, 64));
    const int fa = node >= 0 ? fixmask[node] : 0;
    double *myrow = rowbuf + rowl * S;
    __syncthreads();

    auto load_packed = [&](int v) -> int32_t { return v < nvis ? nel[q0 + v] : -1; };
    auto load_nodes = [&](int32_t packed, AsmVisit &V) {
        V.packed = packed;
        const int64_t e = packed >= 0 ? (int64_t)(packed >> 4) : 0;   // idle lanes read element 0 (valid memory), results unused
        const int2 *p = (const int2 *)(el + e * 10);                  // 40-byte rows: 8-byte aligned
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int2 t = p[k];
            V.nd[2 * k] = t.x;
            V.nd[2 * k + 1] = t.y;
        }
    };
    auto load_geom = [&](const AsmVisit &V, AsmGeom &G) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const double *c = coord + 3 * (int64_t)V.nd[k];
            G.cx[k] = c[0]; G.cy[k] = c[1]; G.cz[k] = c[2];
        }
        if (disp) {
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const double *u = disp + 3 * (int64_t)V.nd[k];
                G.cx[k] += u[0]; G.cy[k] += u[1]; G.cz[k] += u[2];
            }
        }
        const int64_t e = V.packed >= 0 ? (int64_t)(V.packed >> 4) : 0;
        const int a = V.packed >= 0 ? (V.packed & 15) : 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int b = min(h + 4 * t, 9);
            // column node of block t: nd[4 t + h] without a runtime-indexed register array
            int nb = V.nd[min(4 * t, 9)];
            if (h == 1) nb = V.nd[min(4 * t + 1, 9)];
            if (h == 2) nb = V.nd[min(4 * t + 2, 9)];
            if (h == 3) nb = V.nd[min(4 * t + 3, 9)];
            G.ndb[t] = nb;
            G.em[t] = emap[e * 100 + a * 10 + b];
            G.fb[t] = fixmask[nb];
        }
    };

    AsmVisit V0, V1, V2;
    AsmGeom G0, G1;
    load_nodes(load_packed(0), V0);
    load_nodes(load_packed(1), V1);
    int32_t pk2 = load_packed(2);
    load_geom(V0, G0);

    double racc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // gravity (3), modf (3), volume
    for (int v = 0; v < maxvis; v++) {
        // loads of the next stages first: they are in flight while this visit is integrated
        load_geom(V1, G1);
        load_nodes(pk2, V2);
        pk2 = load_packed(v + 3);

        const bool active = V0.packed >= 0;
        const int64_t e = active ? (int64_t)(V0.packed >> 4) : 0;
        const int a = active ? (V0.packed & 15) : 0;
        // ---- Jacobian of Gauss point h (fcVM.py:436-453), shared inside the quad ----
        double xsi[4][9], w[4], vol = 0.0;
        {
            double xi, et, ze;
            gp_point(h, xi, et, ze);
            double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 10; k++) {
                double d0, d1, d2;
                dshp_k(k, xi, et, ze, d0, d1, d2);
                xs[0] += G0.cx[k] * d0; xs[1] += G0.cx[k] * d1; xs[2] += G0.cx[k] * d2;
                xs[3] += G0.cy[k] * d0; xs[4] += G0.cy[k] * d1; xs[5] += G0.cy[k] * d2;
                xs[6] += G0.cz[k] * d0; xs[7] += G0.cz[k] * d1; xs[8] += G0.cz[k] * d2;
            }
            double mine[9];
            const double xsj = jac_inverse(xs, mine);
            const double wm = FCVM_GPW * fabs(xsj), vm = xsj * FCVM_GPW;
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
                for (int k = 0; k < 9; k++) xsi[g][k] = quad_get(mine[k], g);
                w[g] = quad_get(wm, g);
                vol += quad_get(vm, g);
            }
        }
        // Gauss point coordinates (fcVM.py:761): lane 3 of the quad (it integrates one block less)
        if (xgp && active && a < 4 && h == 3) {
            double xi, et, ze;
            gp_point(a, xi, et, ze);
            double px = 0, py = 0, pz = 0;
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const double n = shp_k(k, xi, et, ze);
                px += G0.cx[k] * n; py += G0.cy[k] * n; pz += G0.cz[k] * n;
            }
            double *o = xgp + 3 * (4 * e + a);
            o[0] = px; o[1] = py; o[2] = pz;
        }
        // ---- row node a: global gradients, gravity weight, tangent terms ----
        double ga[4][3], gam = 0.0;
        double Sg[4][6], pf[4], ta[4][3];
        if (TANGENT) {   // lane h fetches the stresses of Gauss point h (fcVM.py:983-998), the quad shares them
            double sd[6] = {0, 0, 0, 0, 0, 0}, pfm = 0.0;
            if (active && pgp[4 * e + h]) {
                const double *s0 = sig + 24 * e + 6 * h;
#pragma unroll
                for (int j = 0; j < 6; j++) sd[j] = s0[j];
                const double p = (sd[0] + sd[1] + sd[2]) / 3.0;
                sd[0] -= p; sd[1] -= p; sd[2] -= p;
                double svm = sqrt(1.5 * (sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]) + 3.0 * (sd[3] * sd[3] + sd[4] * sd[4] + sd[5] * sd[5]));
                if (svm == 0.0) svm = 1.0;
                pfm = 3.0 * mc.G / (1.0 + mc.H / 3.0 / mc.G) / (svm * svm);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
#pragma unroll
                for (int j = 0; j < 6; j++) Sg[g][j] = quad_get(sd[j], g);
                pf[g] = quad_get(pfm, g);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
            double xi, et, ze, d0, d1, d2;
            gp_point(g, xi, et, ze);
            dshp_k(a, xi, et, ze, d0, d1, d2);
            to_global(xsi[g], d0, d1, d2, ga[g][0], ga[g][1], ga[g][2]);
            gam += w[g] * shp_k(a, xi, et, ze);
            if (TANGENT) {
                ta[g][0] = ga[g][0] * Sg[g][0] + ga[g][1] * Sg[g][3] + ga[g][2] * Sg[g][4];
                ta[g][1] = ga[g][1] * Sg[g][1] + ga[g][0] * Sg[g][3] + ga[g][2] * Sg[g][5];
                ta[g][2] = ga[g][2] * Sg[g][2] + ga[g][0] * Sg[g][4] + ga[g][1] * Sg[g][5];
            }
        }
        if (active && h == 0) {   // gravity load (fcVM.py:757-767) and volume
            racc[0] += gx * mc.rho * gam; racc[1] += gy * mc.rho * gam; racc[2] += gz * mc.rho * gam;
            if (a == 0) racc[6] += vol;
        }
        // ---- blocks K_ab, b = h, h + 4, h + 8 ----
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int b = h + 4 * t;
            if (active && b < 10) {
                double K[9];
                if (!TANGENT) {
                    // constant D over the Gauss points: K = D12 M + D44 M^T + (D11 - D12 - D44) diag(M) + D44 tr(M) I with
                    // M = sum_g w_g ga_g gb_g^T (the same numbers as fcVM.py:769-771, summed in a different order)
                    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        double xi, et, ze, d0, d1, d2, gb0, gb1, gb2;
                        gp_point(g, xi, et, ze);
                        dshp_k(b, xi, et, ze, d0, d1, d2);
                        to_global(xsi[g], d0, d1, d2, gb0, gb1, gb2);
                        const double a0 = w[g] * ga[g][0], a1 = w[g] * ga[g][1], a2 = w[g] * ga[g][2];
                        M[0] += a0 * gb0; M[1] += a0 * gb1; M[2] += a0 * gb2;
                        M[3] += a1 * gb0; M[4] += a1 * gb1; M[5] += a1 * gb2;
                        M[6] += a2 * gb0; M[7] += a2 * gb1; M[8] += a2 * gb2;
                    }
                    const double tr = M[0] + M[4] + M[8];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            K[r * 3 + c] = (r == c) ? mc.D11 * M[r * 3 + r] + mc.D44 * (tr - M[r * 3 + r])
                                                    : mc.D12 * M[r * 3 + c] + mc.D44 * M[c * 3 + r];
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) K[k] = 0.0;
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        double xi, et, ze, d0, d1, d2, gb[3], tb[3];
                        gp_point(g, xi, et, ze);
                        dshp_k(b, xi, et, ze, d0, d1, d2);
                        to_global(xsi[g], d0, d1, d2, gb[0], gb[1], gb[2]);
                        const double dot = ga[g][0] * gb[0] + ga[g][1] * gb[1] + ga[g][2] * gb[2];
                        tb[0] = gb[0] * Sg[g][0] + gb[1] * Sg[g][3] + gb[2] * Sg[g][4];
                        tb[1] = gb[1] * Sg[g][1] + gb[0] * Sg[g][3] + gb[2] * Sg[g][5];
                        tb[2] = gb[2] * Sg[g][2] + gb[0] * Sg[g][4] + gb[1] * Sg[g][5];
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                double val;
                                if (r == c) val = mc.D11 * ga[g][r] * gb[r] + mc.D44 * (dot - ga[g][r] * gb[r]);
                                else val = mc.D12 * ga[g][r] * gb[c] + mc.D44 * ga[g][c] * gb[r];
                                val -= pf[g] * ta[g][r] * tb[c];
                                K[r * 3 + c] += w[g] * val;
                            }
                    }
                }
                // boundary conditions folded in (fcVM.py:773-800) and accumulation into the row's LDS image
                const uint32_t em = G0.em[t];
                const int64_t jj = 2 * ((int64_t)(em >> 7) - cbase) + (int64_t)(em & 1);
                double *dst = myrow + jj * 9;
                const int fb = G0.fb[t];
                const bool diag = (a == b);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const bool rfix = (fa >> r) & 1;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const bool cfix = (fb >> c) & 1;
                        double add = 0.0;
                        bool has = false;
                        if (penalty_bc) {   // linear-buckling matrices (fcVM.py:1051-1062): no elimination, fixed diagonal x 100
                            add = (rfix && diag && r == c) ? 100.0 * K[r * 3 + c] : K[r * 3 + c];
                            has = true;
                        } else if (!rfix && !cfix) {
                            add = K[r * 3 + c];
                            has = true;
                        } else if (!rfix && cfix) {
                            racc[3 + r] -= K[r * 3 + c] * fixval[3 * (int64_t)G0.ndb[t] + c];
                        } else if (rfix && diag && r == c) {   // fcVM.py:773-778: unit diagonal per element occurrence
                            add = 1.0;
                            has = true;
                            racc[3 + r] += fixval[3 * (int64_t)node + r];
                        }
                        if (has) dst[r * 3 + c] += add;
                    }
                }
            }
        }
        V0 = V1; V1 = V2; G0 = G1;
    }
    // ---- per-row sums: quad reduction in a fixed order, lane 0 of the quad writes ----
#pragma unroll
    for (int k = 3; k < 6; k++) {
        const double v0 = quad_get(racc[k], 0), v1 = quad_get(racc[k], 1), v2 = quad_get(racc[k], 2), v3 = quad_get(racc[k], 3);
        racc[k] = (v0 + v1) + (v2 + v3);
    }
    if (node >= 0 && h == 0) {
        glv[3 * (int64_t)node] = racc[0]; glv[3 * (int64_t)node + 1] = racc[1]; glv[3 * (int64_t)node + 2] = racc[2];
        modf[3 * (int64_t)node] = racc[3]; modf[3 * (int64_t)node + 1] = racc[4]; modf[3 * (int64_t)node + 2] = racc[5];
    }
    if (volume) {
        const double vsum = wave_sum(h == 0 ? racc[6] : 0.0);
        if (lane == 0) volume[wg] = vsum;
    }
    // ---- the finished quarter slice leaves LDS: (pair p, component c) -> 16 lanes x 16 B, four segments per instruction ----
    __syncthreads();
    {
        const int sub = lane >> 4, r16 = lane & 15;
        const double *src = rowbuf + r16 * S;
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d *out = (v2d *)vals + cbase * (9 * 64) + qd * ASM_ROWS + r16;
        const int64_t npairs = len >> 1;
        int64_t p = 0;
        int c = sub;
        while (p < npairs) {
            v2d d;
            d.x = src[(2 * p) * 9 + c];
            d.y = src[(2 * p + 1) * 9 + c];
            __builtin_nontemporal_store(d, out + (p * 9 + c) * 64);
            c += 4;
            if (c >= 9) { c -= 9; p++; }
        }
    }
}

}  // namespace fcvm
