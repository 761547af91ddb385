// calcGSM / calcTSM stiffness assembly with the matrix rows accumulated in LDS (fcVM.py:729-800, 940-1050).
//
// The row-owner kernel k_assemble_rows adds every 3x3 block with scattered 8-byte read-modify-writes into the
// sliced-ELL value array: 900 M of them at 1 M elements, each a different cache line for each lane of the wave, and the
// per-CU address / L1 pipeline (about 4.4 cycles per distinct line, profiles/r02*_pmc) sets its 17 ms.  Here one
// wavefront owns 16 consecutive rows of a slice (a quarter slice) and keeps their blocks in LDS (len x 72 B per row)
// until every element of every row has been visited; the finished quarter slice then leaves in one pass of coalesced
// 16-byte nontemporal stores (each value written exactly once, no memset of the 3 GB array, no atomics, fixed
// summation order => bitwise reproducible).  Four lanes share a row:
//   * lane h of the quad evaluates the Jacobian of Gauss point h of the visited element and leaves it in LDS for the
//     other three (each Jacobian is computed once per visit instead of four times); shape-function derivatives at
//     the Gauss points come from a 1.3 KB LDS table instead of per-lane switch statements;
//   * lane h integrates the blocks K_ab of the local columns b = h, h + 4, h + 8 and adds them to the row's LDS image;
//   * lane h also fetches the inputs of "its" nodes only (connectivity entries, coordinates with one 8-byte and one
//     16-byte load per node, block locations, support flags: 18 vector-memory instructions per visit); the quad
//     exchanges the coordinates through LDS;
//   * the inputs of visit v + 1 (and the connectivity of visit v + 2) are in flight while visit v is integrated.
// What still goes through the scattered-access path are these inputs: 18 instead of 240 instructions per visit.
#pragma once
#include "fcvm_kernels.hip.h"
#include "fcvm_spmv_win.hip.h"

namespace fcvm {

constexpr int ASM_ROWS = 16;                       // rows per wavefront (4 lanes each)
#ifndef ASM_RELOAD_DN
#define ASM_RELOAD_DN false
#endif
#ifndef ASM_RELOAD_DB
#define ASM_RELOAD_DB TANGENT
#endif
// LDS of a work item: shape-function tables | Jacobians of the visited elements (one per row and Gauss point) | rows
constexpr int ASM_TAB = 160;                       // doubles: dN/dxi at the 4 Gauss points [4][10][3], then N [4][10]
constexpr int ASM_JREC = 10;                       // doubles per (row, Gauss point): xsi[9], GPW * det J (signed)
constexpr int ASM_JAC = ASM_ROWS * 4 * ASM_JREC;   // doubles (16 rows per wavefront; `rows` rows: rows * 4 * ASM_JREC)
// len = blocks of the longest row of the slice; row stride in doubles, odd so that the 16 rows spread over the LDS banks
__host__ __device__ inline int64_t asm_row_stride(int64_t len) { return len * 9 + ((len & 1) ? 0 : 1); }
// tangent form: deviatoric stress and plastic factor of the visited elements' Gauss points, one record per (row, Gauss point)
constexpr int ASM_PREC = 8;                        // doubles: pf, s[6], unused
constexpr int ASM_PL = ASM_ROWS * 4 * ASM_PREC;    // doubles
// rows: rows per wavefront = 64 / lanes per row (16 with four lanes per row; round 4: 8 or 4 where the assembly writes the symmetric
// window form -- half / a quarter of the LDS per wavefront, so twice / four times the wavefronts on the same rows)
__host__ __device__ inline int64_t asm_lds_bytes(int64_t len, bool tangent = true, int rows = ASM_ROWS) {
    return (ASM_TAB + rows * 4 * ASM_JREC + (tangent ? rows * 4 * ASM_PREC : 0) + rows * asm_row_stride(len)) * 8;
}

// Hides where a pointer came from, so that loads through it are not hoisted out of the visit loop: every lane always
// uses the same table entries (its Gauss point, its three local columns: 66 doubles), and keeping all of them in
// registers for the whole loop costs more (spills) than re-reading them from LDS.
template <bool ON, class T>
__device__ __forceinline__ T *reload_each_time(T *p) {
    if (ON) asm volatile("" : "+v"(p));
    return p;
}

// dst += v in LDS with one ds_add_f64 (no return value).  Inside one instruction every lane has its own address (the
// four lanes of a row hold different columns), and the instructions of the single wavefront of a work item execute in
// program order: the sums are formed in a fixed order, the atomic is only the cheaper encoding of read-add-write.
__device__ __forceinline__ void lds_add(double *dst, double v) { __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int NT>
struct AsmGeomT {     // what lane h of a row's lanes fetches for one visit: its local columns b = h + LPR t, t < NT
    double c[NT][3];  // (updated) coordinates of node b
    uint32_t em[NT];  // packed block location of (row node, b)
    int ndb[NT];      // node b
    int fb[NT];       // its support flags
    uint32_t pg;      // tangent form: the element's four plastic flags (pgp[4 e .. 4 e + 3])
    double sg[6];     // ... and the stress of Gauss point h of the element
};

// three doubles at a 24-byte stride: one 8-byte and one 16-byte load (even nodes are 16-byte aligned, odd ones 8 off)
__device__ __forceinline__ void load3(const double *__restrict__ base, int node, double &x, double &y, double &z) {
    const double *p = base + 3 * (int64_t)node;
    const int o = node & 1;
    const double2 w = *(const double2 *)(p + o);
    const double s1 = p[o ? 0 : 2];
    x = o ? s1 : w.x; y = o ? w.x : w.y; z = o ? w.y : s1;
}

// SYM (round 4): the work item is 16 rows of one group of the SYMMETRIC window form (WinSym, fcvm_pattern.h) instead of
// a quarter slice of the sliced-ELL layout, the row image in LDS holds the row's KEPT blocks only (0.70 x the blocks: less LDS
// per item, more items per CU), and the finished rows leave LDS straight into the SpMV's storage -- FP64 values of the symmetric
// form, its 16-bit block-scaled copy and the diagonal blocks for the preconditioner -- so neither the sliced-ELL array nor the
// copy kernel (k_repack_sym: 3.3 ms per assembly at 1 M elements) is involved.  A cut row is assembled once and written
// segment by segment to the lanes of its segments.  The item's rows are the group's rows in the order of their kept length, 16
// at a time (every quad busy, one length class per item); their lanes are near each other in the group's waves, so the
// 16-byte stores of a block row fall into a few 128-byte lines that the L2 completes with the neighbouring items'.  All element blocks are still integrated (the boundary-condition terms need the blocks that are
// not stored too); only the accumulation into LDS is skipped for a block whose transpose is the stored one.
struct AsmSym {
    const int32_t *rownode;     // [nwaves*64] node of a lane
    const uint32_t *seg;        // [nwaves*64] low byte > 0: first segment (the lane that assembles the row)
    const uint8_t *emap;        // [ne*100] kept number of element block (a, b) in the row of node a, 0xFF: not stored
    const uint16_t *rowlen, *jfirst, *rowoff;
    const int32_t *nextseg;
    const int64_t *woff;
    const uint32_t *cols;       // [nblocks] entry: window position | target lane << 16 (copied into the 16-bit tail)
    const uint8_t *diagk;       // [nn]
    double *vals;               // FP64 values of the symmetric form
    short8 *lo8;                // 16-bit copy (null: not wanted)
    WinLoTailS *lo2;
    double *dblk;               // [nn*9] diagonal blocks
    int maxl1;                  // row stride of rowoff (longest segment + 1)
    int waves;                  // waves per group
    int nt;                     // nontemporal stores of the finished rows (FCVM_ASM_NT)
    int len;                    // kept blocks of the longest row of the launch's length class
};

// LPR = lanes per row (round 4, SYM only: 8 or 16).  The kernel is bound by the latency of the dependent chain of an element visit
// and the LDS capacity fixes how many ROWS a CU holds, so what is left is the length of that chain: with LPR lanes per row a lane
// integrates ceil(10 / LPR) blocks per visit instead of three and a wavefront holds 64 / LPR rows -- more wavefronts on the same LDS
// (measured first: forcing the short-row classes down to two wavefronts per CU made them 2.3-3.6 x slower).
template <bool TANGENT, bool SYM = false, int LPR = 4>
__global__ __launch_bounds__(64) void k_assemble_lds(const int32_t *__restrict__ items, const int64_t *__restrict__ sptr,
                                                     const int32_t *__restrict__ rownode, const int64_t *__restrict__ nptr,
                                                     const int32_t *__restrict__ nel, const int32_t *__restrict__ el,
                                                     const double *__restrict__ coord, const double *__restrict__ disp,
                                                     const uint8_t *__restrict__ fixmask, const double *__restrict__ fixval,
                                                     const uint32_t *__restrict__ emap, const double *__restrict__ sig,
                                                     const uint8_t *__restrict__ pgp, MatConst mc, double gx, double gy, double gz,
                                                     double *__restrict__ vals, double *__restrict__ glv, double *__restrict__ modf,
                                                     double *__restrict__ xgp, double *__restrict__ volume, int penalty_bc,
                                                     AsmSym sy = AsmSym{}) {
    static_assert(LPR == 4 || (SYM && (LPR == 8 || LPR == 16)), "lanes per row: 4, or 8 / 16 for the symmetric window form");
    constexpr int R = 64 / LPR;                 // rows per wavefront
    constexpr int NT = (10 + LPR - 1) / LPR;    // local columns per lane
    constexpr int JAC = R * 4 * ASM_JREC, PLD = R * 4 * ASM_PREC;
    typedef AsmGeomT<NT> AsmGeom;
    extern __shared__ double lds[];
    double *tab = lds, *jac = lds + ASM_TAB, *pl = lds + ASM_TAB + JAC, *rowbuf = lds + ASM_TAB + JAC + (TANGENT ? PLD : 0);
    // XCD-aware remap (as k_spmv): the workgroups of one XCD take a contiguous range of work items, whose rows share
    // elements and nodes, so that their gathers meet in one L2
    const int64_t nb = gridDim.x, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    const int64_t wg = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const int lane = threadIdx.x, rowl = lane / LPR, h = lane % LPR;
    // SYM: the item lists the lane (wave * 64 + lane) of each of its R rows in the window layout, -1: none
    const int item = SYM ? items[wg * R] : items[wg];
    const int64_t s = SYM ? (int64_t)(item >> 6) : (int64_t)(item >> 2);     // SYM: a wave of the item's group
    const int qd = item & 3;
    // SYM: item = wave * 4 + quarter of the symmetric form; len = the launch's length class (kept blocks)
    const int64_t jb = SYM ? 0 : sptr[s], len = SYM ? (int64_t)sy.len : sptr[s + 1] - jb;   // blocks of the slice's longest row = half chunks of the slice
    const int64_t S = asm_row_stride(len);
    for (int64_t i = lane; i < R * S; i += 64) rowbuf[i] = 0.0;
    if (lane < 40) {   // shape-function tables (fcVM.py:365-424 at the Gauss points of fcVM.py:587-596)
        const int g = lane / 10, k = lane - 10 * g;
        double xi, et, ze, d0, d1, d2;
        gp_point(g, xi, et, ze);
        dshp_k(k, xi, et, ze, d0, d1, d2);
        tab[3 * lane] = d0; tab[3 * lane + 1] = d1; tab[3 * lane + 2] = d2;
        tab[120 + lane] = shp_k(k, xi, et, ze);
    }
    const int64_t slot = SYM ? (int64_t)items[wg * R + rowl] : s * 64 + qd * ASM_ROWS + rowl;
    const int node = SYM ? (slot >= 0 ? sy.rownode[slot] : -1) : rownode[slot];
    const int64_t q0 = node >= 0 ? nptr[node] : 0;
    const int nvis = node >= 0 ? (int)(nptr[node + 1] - q0) : 0;
    int maxvis = nvis;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxvis = max(maxvis, __shfl_xor(maxvis, o, 64));
    const int fa = node >= 0 ? fixmask[node] : 0;
    double *myrow = rowbuf + rowl * S;
    const int hg = h & 3;                                    // Gauss point of this lane (lanes h >= 4 of a row idle in that stage)
    double *myjac = jac + (rowl * 4 + hg) * ASM_JREC;        // this lane's Gauss point of this row's visited element
    const double *rowjac = jac + rowl * 4 * ASM_JREC;        // all four
    const double *mydn = tab + 30 * hg;                      // dN/dxi at Gauss point h
    __syncthreads();

    // packed visit: element << 4 | local index of the row node, -1: no visit (lanes of shorter rows)
    auto load_packed = [&](int v) -> int32_t { return v < nvis ? nel[q0 + v] : -1; };
    // The four lanes of a quad visit the same element: lane h fetches the nodes b = h, h + 4, h + 8 -- the columns whose
    // blocks it integrates -- and their coordinates; the quad exchanges the coordinates through LDS (the Jacobian needs all
    // ten).  12 + 6 vector-memory instructions per visit instead of 44 with every lane loading everything.
    int nd[NT];       // nodes b of the visit whose coordinates are fetched next
    auto load_nodes = [&](int32_t packed) {
        const int64_t e = packed >= 0 ? (int64_t)(packed >> 4) : 0;   // idle lanes read element 0 (valid memory), results unused
#pragma unroll
        for (int t = 0; t < NT; t++) nd[t] = el[e * 10 + min(h + LPR * t, 9)];
    };
    auto load_geom = [&](int32_t packed, AsmGeom &G) {
        const int64_t e = packed >= 0 ? (int64_t)(packed >> 4) : 0;
        const int a = packed >= 0 ? (packed & 15) : 0;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            load3(coord, nd[t], G.c[t][0], G.c[t][1], G.c[t][2]);
            if (disp) {
                double u0, u1, u2;
                load3(disp, nd[t], u0, u1, u2);
                G.c[t][0] += u0; G.c[t][1] += u1; G.c[t][2] += u2;
            }
            G.ndb[t] = nd[t];
            G.em[t] = SYM ? (uint32_t)sy.emap[e * 100 + a * 10 + min(h + LPR * t, 9)] : emap[e * 100 + a * 10 + min(h + LPR * t, 9)];
            G.fb[t] = fixmask[nd[t]];
        }
        if (TANGENT) {   // with the other inputs of the visit, one visit ahead (round 4: these loads used to sit inside the Gauss
                         // point loop, four dependent round trips to memory per visit with one or two wavefronts per CU to hide them)
            G.pg = *(const uint32_t *)(pgp + 4 * e);
            const double2 *s2 = (const double2 *)(sig + 24 * e + 6 * hg);
            const double2 q0 = s2[0], q1 = s2[1], q2 = s2[2];
            G.sg[0] = q0.x; G.sg[1] = q0.y; G.sg[2] = q1.x; G.sg[3] = q1.y; G.sg[4] = q2.x; G.sg[5] = q2.y;
        }
    };

    double racc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // gravity (3), modf (3), volume
    auto integrate = [&](const int32_t pk, const AsmGeom &G0) {
        const bool active = pk >= 0;
        const int64_t e = active ? (int64_t)(pk >> 4) : 0;
        const int a = active ? (pk & 15) : 0;
        // ---- coordinates of the element's ten nodes: every lane contributes its three, the row's image is [node][3] ----
        double *crd = jac + rowl * 4 * ASM_JREC;   // the row's Jacobian records double as the exchange area (30 of 40 doubles)
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (h + LPR * t < 10) {
                double *o = crd + 3 * (h + LPR * t);
                o[0] = G0.c[t][0]; o[1] = G0.c[t][1]; o[2] = G0.c[t][2];
            }
        __syncthreads();
        double cx[10], cy[10], cz[10];
        {   // 30 doubles as fifteen 16-byte LDS reads (the row's area is 16-byte aligned)
            const double2 *c2 = (const double2 *)crd;
            double f[30];
#pragma unroll
            for (int k = 0; k < 15; k++) { const double2 t2 = c2[k]; f[2 * k] = t2.x; f[2 * k + 1] = t2.y; }
#pragma unroll
            for (int k = 0; k < 10; k++) { cx[k] = f[3 * k]; cy[k] = f[3 * k + 1]; cz[k] = f[3 * k + 2]; }
        }
        __syncthreads();   // all coordinates are in registers before the records overwrite them
        // ---- Jacobian of Gauss point h (fcVM.py:436-453) -> LDS record of (row, h) ----
        {
            double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            const double *dn = reload_each_time<ASM_RELOAD_DN>(mydn);
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const double d0 = dn[3 * k], d1 = dn[3 * k + 1], d2 = dn[3 * k + 2];
                xs[0] += cx[k] * d0; xs[1] += cx[k] * d1; xs[2] += cx[k] * d2;
                xs[3] += cy[k] * d0; xs[4] += cy[k] * d1; xs[5] += cy[k] * d2;
                xs[6] += cz[k] * d0; xs[7] += cz[k] * d1; xs[8] += cz[k] * d2;
            }
            double mine[9];
            const double xsj = jac_inverse(xs, mine);
            if (LPR == 4 || h < 4) {      // (more than four lanes per row: the others would write the same records)
                double2 *j2 = (double2 *)myjac;   // 10 doubles = five 16-byte LDS writes
                j2[0] = double2{mine[0], mine[1]}; j2[1] = double2{mine[2], mine[3]}; j2[2] = double2{mine[4], mine[5]};
                j2[3] = double2{mine[6], mine[7]}; j2[4] = double2{mine[8], xsj * FCVM_GPW};
            }
        }
        if (TANGENT) {   // deviatoric stress and plastic factor of Gauss point h (fcVM.py:983-998), once per quad -> LDS record of (row, h)
            double pf = 0.0, Sg[6] = {0, 0, 0, 0, 0, 0};
            if (active && ((G0.pg >> (8 * hg)) & 0xFFu)) {
#pragma unroll
                for (int j = 0; j < 6; j++) Sg[j] = G0.sg[j];
                const double p = (Sg[0] + Sg[1] + Sg[2]) / 3.0;
                Sg[0] -= p; Sg[1] -= p; Sg[2] -= p;
                double svm = sqrt(1.5 * (Sg[0] * Sg[0] + Sg[1] * Sg[1] + Sg[2] * Sg[2]) + 3.0 * (Sg[3] * Sg[3] + Sg[4] * Sg[4] + Sg[5] * Sg[5]));
                if (svm == 0.0) svm = 1.0;
                pf = 3.0 * mc.G / (1.0 + mc.H / 3.0 / mc.G) / (svm * svm);
            }
            if (LPR == 4 || h < 4) {
                double2 *p2 = (double2 *)(pl + (rowl * 4 + hg) * ASM_PREC);
                p2[0] = double2{pf, Sg[0]}; p2[1] = double2{Sg[1], Sg[2]}; p2[2] = double2{Sg[3], Sg[4]}; p2[3] = double2{Sg[5], 0.0};
            }
        }
        // Gauss point coordinates (fcVM.py:761): lane 3 of the quad (it integrates one block less)
        if (xgp && active && a < 4 && h == 3) {
            const double *nsa = tab + 120 + 10 * a;
            double px = 0, py = 0, pz = 0;
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const double n = nsa[k];
                px += cx[k] * n; py += cy[k] * n; pz += cz[k] * n;
            }
            double *o = xgp + 3 * (4 * e + a);
            o[0] = px; o[1] = py; o[2] = pz;
        }
        __syncthreads();   // one wavefront: orders the LDS records of the quad before they are read
        // ---- blocks K_ab, b = h + LPR t (h, h + 4, h + 8 with four lanes per row), Gauss points outermost ----
        double M[NT][9], Pl[TANGENT ? NT : 1][9];   // Pl: plastic part of the tangent blocks
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int k = 0; k < 9; k++) M[t][k] = 0.0;
#pragma unroll
        for (int t = 0; t < (TANGENT ? NT : 1); t++)
#pragma unroll
            for (int k = 0; k < 9; k++) Pl[t][k] = 0.0;
        double gam = 0.0, vol = 0.0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const double2 *J2 = (const double2 *)(rowjac + g * ASM_JREC);
            const double2 q0_ = J2[0], q1_ = J2[1], q2_ = J2[2], q3_ = J2[3], q4_ = J2[4];
            const double xsi[9] = {q0_.x, q0_.y, q1_.x, q1_.y, q2_.x, q2_.y, q3_.x, q3_.y, q4_.x};
            const double wg_ = fabs(q4_.y);
            vol += q4_.y;
            const double *da = tab + 30 * g + 3 * a;
            double ga[3];
            to_global(xsi, da[0], da[1], da[2], ga[0], ga[1], ga[2]);
            gam += wg_ * tab[120 + 10 * g + a];
            double pf = 0.0, Sg[6] = {0, 0, 0, 0, 0, 0}, ta[3];
            if (TANGENT) {   // the record lane g of the quad left for Gauss point g
                const double2 *P2 = (const double2 *)(pl + (rowl * 4 + g) * ASM_PREC);
                const double2 r0 = P2[0], r1 = P2[1], r2 = P2[2], r3 = P2[3];
                pf = r0.x; Sg[0] = r0.y; Sg[1] = r1.x; Sg[2] = r1.y; Sg[3] = r2.x; Sg[4] = r2.y; Sg[5] = r3.x;
                ta[0] = ga[0] * Sg[0] + ga[1] * Sg[3] + ga[2] * Sg[4];
                ta[1] = ga[1] * Sg[1] + ga[0] * Sg[3] + ga[2] * Sg[5];
                ta[2] = ga[2] * Sg[2] + ga[0] * Sg[4] + ga[1] * Sg[5];
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int b = min(h + LPR * t, 9);
                const double *db = reload_each_time<ASM_RELOAD_DB>(tab + 30 * g + 3 * b);
                double gb[3];
                to_global(xsi, db[0], db[1], db[2], gb[0], gb[1], gb[2]);
                {   // M = sum_g w_g ga_g gb_g^T; the constant elastic D is applied once after the loop
                    const double a0 = wg_ * ga[0], a1 = wg_ * ga[1], a2 = wg_ * ga[2];
                    M[t][0] += a0 * gb[0]; M[t][1] += a0 * gb[1]; M[t][2] += a0 * gb[2];
                    M[t][3] += a1 * gb[0]; M[t][4] += a1 * gb[1]; M[t][5] += a1 * gb[2];
                    M[t][6] += a2 * gb[0]; M[t][7] += a2 * gb[1]; M[t][8] += a2 * gb[2];
                }
                if (TANGENT) {
                    // D_t = D - pf s s^T per Gauss point: the elastic part went through M, the plastic part is the rank-one
                    // update (B_a^T s)(s^T B_b) -- 9 + 12 multiply-adds per Gauss point and block instead of the ~100 flops of
                    // evaluating B_a^T D_t B_b entry by entry (round 3: the tangent form's corner-node class took 7.5 ms per
                    // refresh at 1 M elements against 4.2 ms of the elastic form)
                    if (pf != 0.0) {   // (same for the whole quad: the Gauss point is the element's)
                        double tb[3];
                        tb[0] = gb[0] * Sg[0] + gb[1] * Sg[3] + gb[2] * Sg[4];
                        tb[1] = gb[1] * Sg[1] + gb[0] * Sg[3] + gb[2] * Sg[5];
                        tb[2] = gb[2] * Sg[2] + gb[0] * Sg[4] + gb[1] * Sg[5];
                        const double p0 = wg_ * pf * ta[0], p1 = wg_ * pf * ta[1], p2 = wg_ * pf * ta[2];
                        Pl[t][0] += p0 * tb[0]; Pl[t][1] += p0 * tb[1]; Pl[t][2] += p0 * tb[2];
                        Pl[t][3] += p1 * tb[0]; Pl[t][4] += p1 * tb[1]; Pl[t][5] += p1 * tb[2];
                        Pl[t][6] += p2 * tb[0]; Pl[t][7] += p2 * tb[1]; Pl[t][8] += p2 * tb[2];
                    }
                }
            }
        }
        if (active && h == 0) {   // gravity load (fcVM.py:757-767) and volume
            racc[0] += gx * mc.rho * gam; racc[1] += gy * mc.rho * gam; racc[2] += gz * mc.rho * gam;
            if (a == 0) racc[6] += vol;
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int b = h + LPR * t;
            if (active && b < 10) {
                double K[9];
                {   // K = D12 M + D44 M^T + (D11 - D12 - D44) diag(M) + D44 tr(M) I (fcVM.py:769-771 regrouped)
                    const double tr = M[t][0] + M[t][4] + M[t][8];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            K[r * 3 + c] = (r == c) ? mc.D11 * M[t][r * 3 + r] + mc.D44 * (tr - M[t][r * 3 + r])
                                                    : mc.D12 * M[t][r * 3 + c] + mc.D44 * M[t][c * 3 + r];
                }
                if (TANGENT) {   // minus the plastic part sum_g w_g pf_g (B_a^T s_g)(s_g^T B_b)  (fcVM.py:983-998)
#pragma unroll
                    for (int k = 0; k < 9; k++) K[k] -= Pl[t][k];
                }
                // boundary conditions folded in (fcVM.py:773-800) and accumulation into the row's LDS image
                const uint32_t em = G0.em[t];
                const int64_t jj = SYM ? (int64_t)em : (int64_t)(em >> 2) - jb + (int64_t)((em >> 1) & 1u);   // block index inside the row
                const bool kept = !SYM || em != 0xFFu;     // SYM: the transpose of this block is the stored one -- nothing to add here
                double *dst = myrow + (kept ? jj : 0) * 9;
                const int fb = G0.fb[t];
                const bool diag = (a == b);
                if (penalty_bc) {   // linear-buckling matrices (fcVM.py:1051-1062): no elimination, fixed diagonal x 100
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++)
                            lds_add(dst + r * 3 + c, (((fa >> r) & 1) && diag && r == c) ? 100.0 * K[r * 3 + c] : K[r * 3 + c]);
                } else if ((fa | fb) == 0) {   // no supports at either node: the whole block
                    if (kept) {
#pragma unroll
                        for (int k = 0; k < 9; k++) lds_add(dst + k, K[k]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const bool rfix = (fa >> r) & 1;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const bool cfix = (fb >> c) & 1;
                            if (!rfix && !cfix) {
                                if (kept) lds_add(dst + r * 3 + c, K[r * 3 + c]);
                            } else if (!rfix && cfix) {
                                racc[3 + r] -= K[r * 3 + c] * fixval[3 * (int64_t)G0.ndb[t] + c];
                            } else if (rfix && diag && r == c) {   // fcVM.py:773-778: unit diagonal per element occurrence
                                lds_add(dst + r * 3 + c, 1.0);
                                racc[3 + r] += fixval[3 * (int64_t)node + r];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();   // the Jacobian records are rewritten by the next visit
    };

    // software pipeline over the visits, two per trip so that the two coordinate sets swap roles without copies:
    // while visit v is integrated, the coordinates / block locations of visit v + 1 and the connectivity of visit
    // v + 2 are in flight
    AsmGeom GA, GB;
    int32_t pk0 = load_packed(0), pk1 = load_packed(1), pk2 = load_packed(2);
    load_nodes(pk0);
    load_geom(pk0, GA);
    load_nodes(pk1);
    for (int v = 0; v < maxvis; v += 2) {
        load_geom(pk1, GB);
        load_nodes(pk2);
        const int32_t pk3 = load_packed(v + 3);
        integrate(pk0, GA);
        int32_t pk4 = -1;
        if (v + 1 < maxvis) {
            load_geom(pk2, GA);
            load_nodes(pk3);
            pk4 = load_packed(v + 4);
            integrate(pk1, GB);
        }
        pk0 = pk2; pk1 = pk3; pk2 = pk4;
    }
    // ---- per-row sums: quad reduction in a fixed order, lane 0 of the quad writes ----
#pragma unroll
    for (int k = 3; k < 6; k++) {
        if (LPR == 4) {
            const int qb = lane & 60;
            const double v0 = __shfl(racc[k], qb, 64), v1 = __shfl(racc[k], qb + 1, 64), v2 = __shfl(racc[k], qb + 2, 64), v3 = __shfl(racc[k], qb + 3, 64);
            racc[k] = (v0 + v1) + (v2 + v3);
        } else {      // the same pairwise tree over the row's LPR lanes
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) racc[k] += __shfl_xor(racc[k], o, 64);
        }
    }
    if (node >= 0 && h == 0) {
        glv[3 * (int64_t)node] = racc[0]; glv[3 * (int64_t)node + 1] = racc[1]; glv[3 * (int64_t)node + 2] = racc[2];
        modf[3 * (int64_t)node] = racc[3]; modf[3 * (int64_t)node + 1] = racc[4]; modf[3 * (int64_t)node + 2] = racc[5];
    }
    if (volume) {
        const double vsum = wave_sum(h == 0 ? racc[6] : 0.0);
        if (lane == 0) volume[wg] = vsum;
    }
    __syncthreads();
    if (SYM) {
        // ---- the finished rows leave LDS in the symmetric window layout: thread (sub, r16) of row r16 stores the components
        //      (2 sub, 2 sub + 1) of every kept block as one 16-byte word; sub 0 also component 8, sub 1 the eight 16-bit values, sub 2
        //      the 8-byte tail {ninth 16-bit value, window column, block exponent, target lane}, sub 3 the diagonal block.
        //      Segment after segment of a cut row; 16 consecutive lanes of a wave write 256 contiguous bytes per instruction ----
        typedef double v2d __attribute__((ext_vector_type(2)));
        // thread (sub, r16): row r16 of the item's R rows, sub = 0 .. LPR - 1 (the names keep the four-lanes-per-row case's)
        const int sub = lane / R, r16 = lane % R;
        constexpr int SUB8 = LPR == 4 ? 0 : 4, SUBLO8 = LPR == 4 ? 1 : 5, SUBLO2 = LPR == 4 ? 2 : 6, SUBDIAG = LPR == 4 ? 3 : 7;
        const double *src = rowbuf + r16 * S;
        const int nd16 = __shfl(node, LPR * r16, 64);
        int64_t cur = nd16 >= 0 ? (int64_t)items[wg * R + r16] : -1;
        // the block-row offsets of the waves of this group (every segment of the item's rows lives in one of them) staged in the
        // free Jacobian area: a dependent global load per block row otherwise, with one wavefront per item to hide it
        const int64_t wave0 = s - s % sy.waves;
        uint16_t *ros = (uint16_t *)jac;
        int64_t *wos = (int64_t *)(jac + JAC - 16);
        const bool staged = sy.waves * sy.maxl1 <= (JAC - 16) * 4 && sy.waves <= 16;
        if (staged) {
            for (int i = lane; i < sy.waves * sy.maxl1; i += 64) ros[i] = sy.rowoff[wave0 * sy.maxl1 + i];
            if (lane < sy.waves) wos[lane] = sy.woff[wave0 + lane];
        }
        __syncthreads();
        if (sub == SUBDIAG && nd16 >= 0) {
            const double *sd = src + 9 * (int)sy.diagk[nd16];
            double *dd = sy.dblk + 9 * (int64_t)nd16;
#pragma unroll
            for (int k = 0; k < 9; k++) dd[k] = sd[k];
        }
        while (cur >= 0) {
            const int64_t wv = cur >> 6;
            const int ln = (int)(cur & 63);
            const int L = sy.rowlen[cur], j0 = sy.jfirst[cur];
            const int64_t wo = staged ? wos[wv - wave0] : sy.woff[wv];
            const uint16_t *ro = staged ? ros + (wv - wave0) * sy.maxl1 : sy.rowoff + wv * sy.maxl1;
            int o0 = ro[0];
            for (int j = 0; j < L; j++) {
                const int o1 = ro[j + 1];
                const int n = o1 - o0;
                const int64_t eoff = wo + o0;
                const double *b9 = src + 9 * (j0 + j);
                if (sub < 4) {
                    v2d d;
                    d.x = b9[2 * sub]; d.y = b9[2 * sub + 1];
                    v2d *dv = (v2d *)(sy.vals + 9 * eoff + (int64_t)2 * sub * n + 2 * ln);
                    if (sy.nt) __builtin_nontemporal_store(d, dv); else *dv = d;
                }
                if (sub == SUB8) {
                    double *d8 = sy.vals + 9 * eoff + (int64_t)8 * n + ln;
                    if (sy.nt) __builtin_nontemporal_store(b9[8], d8); else *d8 = b9[8];
                }
                if (sy.lo8 && (sub == SUBLO8 || sub == SUBLO2)) {
                    // 16-bit block-scaled copy: nine integers on the block's own power of two, the exponent stored absolutely
                    // (frexp exponent of the block maximum; the kernels subtract the matrix exponent e0 when they read it)
                    double m = 0.0;
#pragma unroll
                    for (int c = 0; c < 9; c++) m = fmax(m, fabs(b9[c]));
                    int eb = -1023;
                    if (m > 0.0) (void)frexp(m, &eb);
                    const int shq = 15 - eb;
                    if (sub == SUBLO8) {
                        short8 hq;
#pragma unroll
                        for (int c = 0; c < 8; c++) {
                            double t = rint(ldexp(b9[c], shq));
                            t = t > 32767.0 ? 32767.0 : (t < -32767.0 ? -32767.0 : t);
                            hq[c] = (short)(int)t;
                        }
                        if (sy.nt) __builtin_nontemporal_store(hq, sy.lo8 + eoff + ln); else sy.lo8[eoff + ln] = hq;
                    } else {
                        double t = rint(ldexp(b9[8], shq));
                        t = t > 32767.0 ? 32767.0 : (t < -32767.0 ? -32767.0 : t);
                        // {q8, eb}: the first four bytes of the tail; the other four (window column, target lane) belong to the
                        // pattern and were written when the copy was allocated (k_lo2_static) -- nothing is read here
                        const uint32_t w4 = (uint32_t)(uint16_t)(short)(int)t | ((uint32_t)(uint16_t)(short)eb << 16);
                        if (sy.nt) __builtin_nontemporal_store(w4, (uint32_t *)(sy.lo2 + eoff + ln)); else *(uint32_t *)(sy.lo2 + eoff + ln) = w4;
                    }
                }
                o0 = o1;
            }
            cur = sy.nextseg[cur];
        }
        return;
    }
    // ---- the finished quarter slice leaves LDS: (pair p, component c) -> 16 lanes x 16 B, four segments per instruction;
    //      the single last block of an odd-length slice as 16 lanes x 8 B ----
    {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int sub = lane >> 4, r16 = lane & 15;
        const double *src = rowbuf + r16 * S;
        double *base = vals + jb * (9 * 64);
        v2d *out = (v2d *)base + qd * ASM_ROWS + r16;
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
        if (len & 1) {
            double *o1 = base + 2 * npairs * (9 * 64) + qd * ASM_ROWS + r16;
            for (int cc = sub; cc < 9; cc += 4) __builtin_nontemporal_store(src[(len - 1) * 9 + cc], o1 + cc * 64);
        }
    }
}

// ---- level 1 of the multigrid hierarchy (linear tets on the corner nodes) with the same scheme (round 4) -------------------------
// k_assemble_tet4_rows (round 1) adds every block with read-modify-writes in global memory: PMC in round 4 (1 M plate, 186 k
// coarse nodes) 5.35 GB fetched + 2.25 GB written for a 0.1 GB matrix, 95 % of the wave cycles waiting, 1.3-1.5 ms.  Here a wavefront
// owns 16 rows of a level-1 slice, four lanes per row: lane h integrates block (a, h) of every element at the row's corner (the
// element's four gradients are cheap enough to be formed by each lane), the rows are summed in LDS and leave in one coalesced
// pass -- each value written once, no memset, every entry summed in the order of the old kernel (bitwise the same matrix).
template <bool TANGENT>
__global__ __launch_bounds__(64) void k_assemble_tet4_lds(int64_t nitems, const int64_t *__restrict__ sptr1, const int32_t *__restrict__ rownode1,
                                                          const int64_t *__restrict__ nptr1, const int32_t *__restrict__ nel1,
                                                          const int32_t *__restrict__ el4, const double *__restrict__ coord1,
                                                          const uint8_t *__restrict__ fixmask1, const uint32_t *__restrict__ emap4, MatConst mc,
                                                          double *__restrict__ vals, const int32_t *__restrict__ c2f,
                                                          const double *__restrict__ disp, const double *__restrict__ sig,
                                                          const uint8_t *__restrict__ pgp) {
    extern __shared__ double rowbuf[];
    const int64_t nb = gridDim.x, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    const int64_t wg = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    if (wg >= nitems) return;
    const int64_t s = wg >> 2;
    const int qd = (int)(wg & 3);
    const int lane = threadIdx.x, rowl = lane >> 2, h = lane & 3;
    const int64_t jb = sptr1[s], len = sptr1[s + 1] - jb;
    const int64_t S = asm_row_stride(len);
    for (int64_t i = lane; i < ASM_ROWS * S; i += 64) rowbuf[i] = 0.0;
    const int64_t slot = s * 64 + qd * ASM_ROWS + rowl;
    const int node = rownode1[slot];
    const int64_t q0 = node >= 0 ? nptr1[node] : 0;
    const int nvis = node >= 0 ? (int)(nptr1[node + 1] - q0) : 0;
    int maxvis = nvis;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxvis = max(maxvis, __shfl_xor(maxvis, o, 64));
    const int fa = node >= 0 ? fixmask1[node] : 0;
    double *myrow = rowbuf + rowl * S;
    __syncthreads();
    for (int v = 0; v < maxvis; v++) {
        const int32_t packed = v < nvis ? nel1[q0 + v] : -1;
        const bool active = packed >= 0;
        const int64_t e = active ? (int64_t)(packed >> 4) : 0;
        const int a = active ? (packed & 15) : 0;
        int nd[4];
        double x[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            nd[k] = el4[e * 4 + k];
            load3(coord1, nd[k], x[k][0], x[k][1], x[k][2]);
            if (TANGENT && disp) {
                double u0, u1, u2;
                load3(disp, c2f[nd[k]], u0, u1, u2);
                x[k][0] += u0; x[k][1] += u1; x[k][2] += u2;
            }
        }
        // gradients of the four linear shape functions (the arithmetic of tet4_element_row, fcvm_mg_kernels.hip.h)
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
        double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
        int ndb = nd[0];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k == a) { ga[0] = g[k][0]; ga[1] = g[k][1]; ga[2] = g[k][2]; }
            if (k == h) { gb[0] = g[k][0]; gb[1] = g[k][1]; gb[2] = g[k][2]; ndb = nd[k]; }
        }
        const double dot = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
        double sub[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (TANGENT) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (active && pgp[4 * e + q]) {
                    const double *s0 = sig + 24 * e + 6 * q;
                    double sd[6] = {s0[0], s0[1], s0[2], s0[3], s0[4], s0[5]};
                    const double p = (sd[0] + sd[1] + sd[2]) / 3.0;
                    sd[0] -= p; sd[1] -= p; sd[2] -= p;
                    double svm = sqrt(1.5 * (sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]) + 3.0 * (sd[3] * sd[3] + sd[4] * sd[4] + sd[5] * sd[5]));
                    if (svm == 0.0) svm = 1.0;
                    const double pf = 0.25 * 3.0 * mc.G / (1.0 + mc.H / 3.0 / mc.G) / (svm * svm);   // 0.25: four equal Gauss weights
                    double ta[3], tb[3];
                    ta[0] = ga[0] * sd[0] + ga[1] * sd[3] + ga[2] * sd[4];
                    ta[1] = ga[1] * sd[1] + ga[0] * sd[3] + ga[2] * sd[5];
                    ta[2] = ga[2] * sd[2] + ga[0] * sd[4] + ga[1] * sd[5];
                    tb[0] = gb[0] * sd[0] + gb[1] * sd[3] + gb[2] * sd[4];
                    tb[1] = gb[1] * sd[1] + gb[0] * sd[3] + gb[2] * sd[5];
                    tb[2] = gb[2] * sd[2] + gb[0] * sd[4] + gb[1] * sd[5];
#pragma unroll
                    for (int r = 0; r < 3; r++)
#pragma unroll
                        for (int c = 0; c < 3; c++) sub[r * 3 + c] += pf * ta[r] * tb[c];
                }
            }
        }
        if (active) {
            const int fb = fixmask1[ndb];
            const uint32_t em = emap4[(e * 4 + a) * 4 + h];
            const int64_t jj = (int64_t)(em >> 2) - jb + (int64_t)((em >> 1) & 1u);
            double *dst = myrow + 9 * jj;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const bool rfix = (fa >> r) & 1;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const bool cfix = (fb >> c) & 1;
                    double vv;
                    if (r == c) vv = mc.D11 * ga[r] * gb[r] + mc.D44 * (dot - ga[r] * gb[r]);
                    else vv = mc.D12 * ga[r] * gb[c] + mc.D44 * ga[c] * gb[r];
                    if (TANGENT) vv -= sub[r * 3 + c];
                    if (!rfix && !cfix) dst[r * 3 + c] += vol * vv;
                    else if (rfix && a == h && r == c) dst[r * 3 + c] += 1.0;
                }
            }
        }
    }
    __syncthreads();
    {   // the finished quarter slice: as the fine kernel's sliced-ELL write-out
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int sub4 = lane >> 4, r16 = lane & 15;
        const double *src = rowbuf + r16 * S;
        double *base = vals + jb * (9 * 64);
        v2d *out = (v2d *)base + qd * ASM_ROWS + r16;
        const int64_t npairs = len >> 1;
        int64_t p = 0;
        int c = sub4;
        while (p < npairs) {
            v2d d;
            d.x = src[(2 * p) * 9 + c];
            d.y = src[(2 * p + 1) * 9 + c];
            out[(p * 9 + c) * 64] = d;
            c += 4;
            if (c >= 9) { c -= 9; p++; }
        }
        if (len & 1) {
            double *o1 = base + 2 * npairs * (9 * 64) + qd * ASM_ROWS + r16;
            for (int cc = sub4; cc < 9; cc += 4) o1[cc * 64] = src[(len - 1) * 9 + cc];
        }
    }
}

// ---- geometric stiffness of the buckling pre-analysis (calcTSM with nstep = 1, fcVM.py:1002-1006) with rows in LDS (round 4) ------
// Kg_ab = (sum_gp w ga . sigma . gb) I3: one scalar per block.  k_assemble_geometric (round 1) adds it to three components of every
// block with read-modify-writes in global memory; here a wavefront owns 16 rows of a slice, four lanes per row, keeps ONE double per
// block in LDS (lane h forms Gauss point h's Jacobian and sigma . ga once per visit and leaves them in an LDS record for the quad,
// then sums the blocks of the local columns h, h + 4, h + 8) and writes the three diagonal components of the finished rows once
// (the array is zeroed before, as for the old kernel; every entry is summed in the old kernel's order).
constexpr int ASM_GREC = 14;      // doubles per (row, Gauss point): xsi[9], w, sigma . ga [3], unused
__host__ __device__ inline int64_t asm_geom_lds_bytes(int64_t len) { return (ASM_ROWS * 4 * ASM_GREC + ASM_ROWS * (len | 1)) * 8; }
__global__ __launch_bounds__(64) void k_assemble_geometric_lds(int64_t nitems, const int64_t *__restrict__ sptr, const int32_t *__restrict__ rownode,
                                                               const int64_t *__restrict__ nptr, const int32_t *__restrict__ nel,
                                                               const int32_t *__restrict__ el, const double *__restrict__ coord,
                                                               const uint32_t *__restrict__ emap, const double *__restrict__ sig,
                                                               double *__restrict__ vals_g) {
    extern __shared__ double glds[];
    double *rec = glds, *rowbuf = glds + ASM_ROWS * 4 * ASM_GREC;
    const int64_t nb = gridDim.x, bq = nb >> 3, br = nb & 7, xcd = blockIdx.x & 7;
    const int64_t wg = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    if (wg >= nitems) return;
    const int64_t s = wg >> 2;
    const int qd = (int)(wg & 3);
    const int lane = threadIdx.x, rowl = lane >> 2, h = lane & 3;
    const int64_t jb = sptr[s], len = sptr[s + 1] - jb;
    const int64_t S = len | 1;
    for (int64_t i = lane; i < ASM_ROWS * S; i += 64) rowbuf[i] = 0.0;
    const int64_t slot = s * 64 + qd * ASM_ROWS + rowl;
    const int node = rownode[slot];
    const int64_t q0 = node >= 0 ? nptr[node] : 0;
    const int nvis = node >= 0 ? (int)(nptr[node + 1] - q0) : 0;
    int maxvis = nvis;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxvis = max(maxvis, __shfl_xor(maxvis, o, 64));
    double *myrow = rowbuf + rowl * S;
    double *myrec = rec + (rowl * 4 + h) * ASM_GREC;
    const double *rowrec = rec + rowl * 4 * ASM_GREC;
    __syncthreads();
    for (int v = 0; v < maxvis; v++) {
        const int32_t packed = v < nvis ? nel[q0 + v] : -1;
        const bool active = packed >= 0;
        const int64_t e = active ? (int64_t)(packed >> 4) : 0;
        const int a = active ? (packed & 15) : 0;
        {   // Gauss point h: Jacobian, weight, sigma . ga  (the arithmetic of geometric_element_row)
            double xi, et, ze;
            gp_point(h, xi, et, ze);
            double xs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 10; k++) {
                const int ndk = el[e * 10 + k];
                double cx, cy, cz, d0, d1, d2;
                load3(coord, ndk, cx, cy, cz);
                dshp_k(k, xi, et, ze, d0, d1, d2);
                xs[0] += cx * d0; xs[1] += cx * d1; xs[2] += cx * d2;
                xs[3] += cy * d0; xs[4] += cy * d1; xs[5] += cy * d2;
                xs[6] += cz * d0; xs[7] += cz * d1; xs[8] += cz * d2;
            }
            double xsi[9];
            const double xsj = jac_inverse(xs, xsi);
            double d0, d1, d2, g0, g1, g2;
            dshp_k(a, xi, et, ze, d0, d1, d2);
            to_global(xsi, d0, d1, d2, g0, g1, g2);
            const double *sg = sig + 24 * e + 6 * h;  // (xx, yy, zz, xy, zx, yz)
#pragma unroll
            for (int k = 0; k < 9; k++) myrec[k] = xsi[k];
            myrec[9] = FCVM_GPW * fabs(xsj);
            myrec[10] = g0 * sg[0] + g1 * sg[3] + g2 * sg[4];
            myrec[11] = g0 * sg[3] + g1 * sg[1] + g2 * sg[5];
            myrec[12] = g0 * sg[4] + g1 * sg[5] + g2 * sg[2];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int b = h + 4 * t;
            if (active && b < 10) {
                double vsum = 0.0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const double *rg = rowrec + g * ASM_GREC;
                    double xi, et, ze, d0, d1, d2, g0, g1, g2;
                    gp_point(g, xi, et, ze);
                    dshp_k(b, xi, et, ze, d0, d1, d2);
                    to_global(rg, d0, d1, d2, g0, g1, g2);
                    vsum += rg[9] * (rg[10] * g0 + rg[11] * g1 + rg[12] * g2);
                }
                const uint32_t em = emap[e * 100 + a * 10 + b];
                const int64_t jj = (int64_t)(em >> 2) - jb + (int64_t)((em >> 1) & 1u);
                myrow[jj] += vsum;
            }
        }
        __syncthreads();   // the records are rewritten by the next visit
    }
    __syncthreads();
    {   // components 0, 4, 8 of every block of the quarter slice (pairs as 16-byte words, a single last block as 8-byte words)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const int sub4 = lane >> 4, r16 = lane & 15;
        const double *src = rowbuf + r16 * S;
        double *base = vals_g + jb * (9 * 64);
        v2d *out = (v2d *)base + qd * ASM_ROWS + r16;
        const int64_t npairs = len >> 1;
        if (sub4 < 3) {
            const int c = 4 * sub4;
            for (int64_t p = 0; p < npairs; p++) {
                v2d d;
                d.x = src[2 * p];
                d.y = src[2 * p + 1];
                out[(p * 9 + c) * 64] = d;
            }
            if (len & 1) {
                double *o1 = base + 2 * npairs * (9 * 64) + qd * ASM_ROWS + r16;
                o1[c * 64] = src[len - 1];
            }
        }
    }
}

}  // namespace fcvm
