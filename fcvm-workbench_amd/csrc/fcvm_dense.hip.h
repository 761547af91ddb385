// Dense FP64 kernels for the aggregate level of the multigrid hierarchy (no reference counterpart: the reference
// factorises the sparse matrix with CHOLMOD, fcVM.py:1121): A2 = Z^T A1 Z from the sliced-ELL level-1 matrix and the
// explicit inverse of that SPD matrix (blocked Cholesky, triangular inverse, X^T X), so that the hierarchy set-up
// needs neither host LAPACK nor a vendor solver library.  Matrices are row-major with a leading dimension that is
// a multiple of TB; the caller pads with an identity block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fcvm_kernels.hip.h"

namespace fcvm {

constexpr int TB = 64;     // tile edge
constexpr int TK = 16;     // k-slab staged through LDS
constexpr int TPAD = 4;    // LDS row padding (doubles)

// acc[4][4] += sum_k A(r, k) * B(c, k) for the 64 x 64 tile of a 16 x 16 thread grid (thread (ty, tx) owns rows
// 4 ty.., columns 4 tx..).  TA = false: A(r, k) = A[r * lda + k]; TA = true: A(r, k) = A[k * lda + r]; same for B.
// K is a multiple of TK.  sA / sB: [TK][TB + TPAD] each.
template <bool TA, bool TBt>
__device__ __forceinline__ void tile_mac(double (&acc)[4][4], const double *__restrict__ A, int64_t lda, const double *__restrict__ B,
                                         int64_t ldb, int K, double (*sA)[TB + TPAD], double (*sB)[TB + TPAD]) {
    const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
    for (int k0 = 0; k0 < K; k0 += TK) {
        __syncthreads();
        if (!TA) {   // 64 rows x 16 k, k contiguous: thread -> (row, 4 k)
            const int r = t >> 2, kq = (t & 3) * 4;
            const double *p = A + (int64_t)r * lda + k0 + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sA[kq + u][r] = p[u];
        } else {     // 16 k-rows x 64 r, r contiguous: thread -> (k, 4 r)
            const int q = t >> 4, r4 = (t & 15) * 4;
            const double *p = A + (int64_t)(k0 + q) * lda + r4;
#pragma unroll
            for (int u = 0; u < 4; u++) sA[q][r4 + u] = p[u];
        }
        if (!TBt) {
            const int r = t >> 2, kq = (t & 3) * 4;
            const double *p = B + (int64_t)r * ldb + k0 + kq;
#pragma unroll
            for (int u = 0; u < 4; u++) sB[kq + u][r] = p[u];
        } else {
            const int q = t >> 4, r4 = (t & 15) * 4;
            const double *p = B + (int64_t)(k0 + q) * ldb + r4;
#pragma unroll
            for (int u = 0; u < 4; u++) sB[q][r4 + u] = p[u];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; k++) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a[u] = sA[k][4 * ty + u]; b[u] = sB[k][4 * tx + u]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int v = 0; v < 4; v++) acc[u][v] += a[u] * b[v];
        }
    }
}

// Cholesky factor of the diagonal tile k (lower, in place, upper part of the tile zeroed) and the inverse of that factor
// (lower triangular) into Linv[k].  One workgroup.  info[0] = 1 + row of the first non-positive pivot.
// The tile kernels of a factorisation form ONE dependent chain (nt of them: 185 at 11,818 unknowns), so their latency is set-up
// time one for one.  Round 3 (178 -> see profiles): (i) the elimination works on UNSCALED columns -- step j subtracts
// T_ij T_mj / T_jj from the trailing triangle only (16 x 16 thread grid striding over it, no scan of the whole tile) and needs a
// single barrier; column j is final after step j and all columns are scaled by 1 / sqrt(pivot) in one pass at the end;
// (ii) the inverse of the factor takes four threads per column (the sum over m split four ways, combined in a fixed order).
__global__ __launch_bounds__(256) void k_chol_diag(double *__restrict__ M, int64_t ld, int k, double *__restrict__ Linv, int *__restrict__ info) {
    __shared__ double T[TB][TB + 1];
    __shared__ double X[TB][TB + 1];
    __shared__ double P[4][TB];
    const int t = threadIdx.x;
    double *blk = M + ((int64_t)k * TB) * ld + (int64_t)k * TB;
    for (int e = t; e < TB * TB; e += 256) T[e >> 6][e & 63] = blk[(int64_t)(e >> 6) * ld + (e & 63)];
    __syncthreads();
    const int ti = t >> 4, tm = t & 15;
    for (int j = 0; j < TB; j++) {
        const double d = T[j][j];
        if (!(d > 0.0)) {      // (the same value in every thread: the whole workgroup leaves)
            if (t == 0 && atomicCAS(info, 0, k * TB + j + 1) == 0) {}
            return;
        }
        const double inv = 1.0 / d;
        for (int i = j + 1 + ti; i < TB; i += 16) {
            const double lij = T[i][j] * inv;
            for (int m = j + 1 + tm; m <= i; m += 16) T[i][m] -= lij * T[m][j];
        }
        __syncthreads();
    }
    // scale: L_ij = T_ij / sqrt(T_jj), L_jj = sqrt(T_jj)
    for (int e = t; e < TB * TB; e += 256) {
        const int i = e >> 6, m = e & 63;
        if (m < i) X[i][m] = T[i][m] / sqrt(T[m][m]);       // (X as scratch: T's diagonal is still being read)
        else if (m == i) X[i][m] = sqrt(T[i][i]);
        else X[i][m] = 0.0;
    }
    __syncthreads();
    for (int e = t; e < TB * TB; e += 256) T[e >> 6][e & 63] = X[e >> 6][e & 63];
    __syncthreads();
    // inverse of the lower-triangular factor by forward substitution, column c = t / 4, the sum over m in four interleaved parts
    const int c = t >> 2, part = t & 3;
    for (int i = 0; i < TB; i++) {
        double sacc = 0.0;
        if (i > c)
            for (int m = c + part; m < i; m += 4) sacc += T[i][m] * X[m][c];
        P[part][c] = sacc;
        __syncthreads();
        if (part == 0) {
            double v = 0.0;
            if (i >= c) v = ((i == c ? 1.0 : 0.0) - (((P[0][c] + P[1][c]) + P[2][c]) + P[3][c])) / T[i][i];
            X[i][c] = v;
        }
        __syncthreads();
    }
    double *li = Linv + (int64_t)k * TB * TB;
    for (int e = t; e < TB * TB; e += 256) {
        const int i = e >> 6, m = e & 63;
        blk[(int64_t)i * ld + m] = T[i][m];
        li[e] = X[i][m];
    }
}

// panel below the diagonal tile k: L_ik = A_ik Linv_kk^T
__global__ __launch_bounds__(256) void k_chol_panel(double *__restrict__ M, int64_t ld, int k, const double *__restrict__ Linv) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    const int i = k + 1 + blockIdx.x;
    double *A = M + ((int64_t)i * TB) * ld + (int64_t)k * TB;
    double acc[4][4] = {};
    tile_mac<false, false>(acc, A, ld, Linv + (int64_t)k * TB * TB, TB, TB, sA, sB);
    __syncthreads();   // every thread has read its part of A_ik before it is overwritten
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) A[(int64_t)(4 * ty + u) * ld + 4 * tx + v] = acc[u][v];
}

// trailing update after panel k: A_ij -= L_ik L_jk^T for k < j <= i (grid: (i - k - 1, j - k - 1), upper half exits)
__global__ __launch_bounds__(256) void k_chol_trailing(double *__restrict__ M, int64_t ld, int k) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
    if (j > i) return;
    const double *Li = M + ((int64_t)i * TB) * ld + (int64_t)k * TB;
    const double *Lj = M + ((int64_t)j * TB) * ld + (int64_t)k * TB;
    double acc[4][4] = {};
    tile_mac<false, false>(acc, Li, ld, Lj, ld, TB, sA, sB);
    double *C = M + ((int64_t)i * TB) * ld + (int64_t)j * TB;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) C[(int64_t)(4 * ty + u) * ld + 4 * tx + v] -= acc[u][v];
}

// block row i of X = L^-1: X_ii = Linv_ii, X_ij = - Linv_ii sum_{j <= m < i} L_im X_mj (grid: j = 0 .. i; block i itself copies)
__global__ __launch_bounds__(256) void k_trinv_row(const double *__restrict__ M, double *__restrict__ Xm, int64_t ld, int i, const double *__restrict__ Linv) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    __shared__ double sT[TB][TB + TPAD];
    const int j = blockIdx.x;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double *out = Xm + ((int64_t)i * TB) * ld + (int64_t)j * TB;
    const double *li = Linv + (int64_t)i * TB * TB;
    if (j == i) {
        for (int e = threadIdx.x; e < TB * TB; e += 256) out[(int64_t)(e >> 6) * ld + (e & 63)] = li[e];
        return;
    }
    // S = sum_m L_im X_mj : A(r, k) = L[i-row r][m-col k] (k contiguous), B(c, k) = X[m-row k][j-col c] (c contiguous => transposed form)
    double acc[4][4] = {};
    tile_mac<false, true>(acc, M + ((int64_t)i * TB) * ld + (int64_t)j * TB, ld, Xm + ((int64_t)j * TB) * ld + (int64_t)j * TB, ld, (i - j) * TB, sA, sB);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) sT[4 * ty + u][4 * tx + v] = acc[u][v];
    __syncthreads();
    // X_ij = - Linv_ii S : rows r, columns c, sum over q: Linv[r][q] S[q][c]
    double o[4][4] = {};
    for (int q = 0; q < TB; q++) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a[u] = li[(4 * ty + u) * TB + q]; b[u] = sT[q][4 * tx + u]; }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < 4; v++) o[u][v] += a[u] * b[v];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) out[(int64_t)(4 * ty + u) * ld + 4 * tx + v] = -o[u][v];
}

// X = L^-1 by recursive doubling over the block rows (round 3; k_trinv_row above walks the block rows one launch after the
// other with i + 1 workgroups each -- 103 dependent launches at 6536 unknowns, 36 ms of a 65 ms inverse).  With the diagonal
// tiles inverted (Linv of k_chol_diag), level l = 0, 1, .. joins the inverses of two adjacent aligned runs of h = 2^l block rows,
//   [ L11  0  ]^-1   [ X11            0  ]
//   [ L21 L22 ]    = [ -X22 L21 X11  X22 ],
// in two launches for ALL pairs of the level: T = L21 X11 (k_trinv_t, written TRANSPOSED into the unused upper triangle of M:
// tile (j, i) holds T_ij^T, so that the next launch walks it with unit stride in k) and X21 = -X22 T (k_trinv_x).  The last
// pair of a level may have a shorter (or no) right half: nt need not be a power of two.  2 ceil(log2 nt) launches, every one
// a grid of independent 64 x 64 tiles with the k-loops of tile_mac.
// grid (h, h, pairs): left block jl = blockIdx.x, right block ir = blockIdx.y of pair blockIdx.z
__global__ __launch_bounds__(256) void k_trinv_t(double *__restrict__ M, const double *__restrict__ Xm, int64_t ld, int nt, int h) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    const int a = (int)blockIdx.z * 2 * h, j = a + (int)blockIdx.x, i = a + h + (int)blockIdx.y;
    if (i >= nt) return;
    // T_ij = sum_{m = j}^{a + h - 1} L_im X_mj (X11 is lower block-triangular: X_mj = 0 for m < j)
    double acc[4][4] = {};
    tile_mac<false, true>(acc, M + ((int64_t)i * TB) * ld + (int64_t)j * TB, ld, Xm + ((int64_t)j * TB) * ld + (int64_t)j * TB, ld, (a + h - j) * TB, sA, sB);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double *out = M + ((int64_t)j * TB) * ld + (int64_t)i * TB;    // tile (j, i): element (c, r) = T_ij(r, c)
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) out[(int64_t)(4 * tx + v) * ld + 4 * ty + u] = acc[u][v];
}
__global__ __launch_bounds__(256) void k_trinv_x(const double *__restrict__ M, double *__restrict__ Xm, int64_t ld, int nt, int h) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    const int a = (int)blockIdx.z * 2 * h, j = a + (int)blockIdx.x, i = a + h + (int)blockIdx.y;
    if (i >= nt) return;
    // X_ij = - sum_{m = a + h}^{i} X_im T_mj : A(r, k) = X[i-row r][column (a + h) TB + k], B(c, k) = T_mj(k', c) = M[(j TB + c) ld + (a + h) TB + k]
    double acc[4][4] = {};
    tile_mac<false, false>(acc, Xm + ((int64_t)i * TB) * ld + (int64_t)(a + h) * TB, ld, M + ((int64_t)j * TB) * ld + (int64_t)(a + h) * TB, ld,
                           (i - (a + h) + 1) * TB, sA, sB);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double *out = Xm + ((int64_t)i * TB) * ld + (int64_t)j * TB;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) out[(int64_t)(4 * ty + u) * ld + 4 * tx + v] = -acc[u][v];
}
// X_kk = Linv_k for all k (the diagonal tiles of X; everything else of X is written by k_trinv_x or is never read: k_xtx and
// the two kernels above touch the lower block triangle only)
__global__ __launch_bounds__(256) void k_trinv_diag(double *__restrict__ Xm, int64_t ld, const double *__restrict__ Linv) {
    const int k = blockIdx.x;
    double *out = Xm + ((int64_t)k * TB) * ld + (int64_t)k * TB;
    const double *li = Linv + (int64_t)k * TB * TB;
    for (int e = threadIdx.x; e < TB * TB; e += 256) out[(int64_t)(e >> 6) * ld + (e & 63)] = li[e];
}

// Ainv = X^T X for lower block-triangular X: tile (a, b), b <= a: sum over block rows m >= a of X_ma^T X_mb; both
// (a, b) and its transpose (b, a) are written
__global__ __launch_bounds__(256) void k_xtx(const double *__restrict__ Xm, double *__restrict__ Out, int64_t ld, int nt) {
    __shared__ double sA[TK][TB + TPAD], sB[TK][TB + TPAD];
    const int a = blockIdx.y, b = blockIdx.x;
    if (b > a) return;
    double acc[4][4] = {};
    // A(r, k) = X[row a*TB + k .. ][col a-tile r] -> element X[(a TB + k) ld + a TB + r]: transposed form, K = (nt - a) TB
    tile_mac<true, true>(acc, Xm + ((int64_t)a * TB) * ld + (int64_t)a * TB, ld, Xm + ((int64_t)a * TB) * ld + (int64_t)b * TB, ld, (nt - a) * TB, sA, sB);
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int64_t r = (int64_t)a * TB + 4 * ty + u, c = (int64_t)b * TB + 4 * tx + v;
            Out[r * ld + c] = acc[u][v];
            Out[c * ld + r] = acc[u][v];
        }
}

// dst (n x n, leading dimension ldd) = 0.5 (src + src^T) of the leading n x n part of src (leading dimension lds),
// plus shift on the diagonal; entries outside n x n of dst (up to np) = identity
__global__ __launch_bounds__(256) void k_sym_pad(int64_t n, int64_t np, const double *__restrict__ src, int64_t lds, double shift,
                                                 double *__restrict__ dst, int64_t ldd) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= np * np) return;
    const int64_t r = e / np, c = e - r * np;
    double v;
    if (r < n && c < n) v = 0.5 * (src[r * lds + c] + src[c * lds + r]) + (r == c ? shift : 0.0);
    else v = (r == c) ? 1.0 : 0.0;
    dst[r * ldd + c] = v;
}

// dense matrix from 6 x 6 blocks: A[6 row[sl] + k][6 col[sl] + l] = B[sl][k][l] (every block once: plain stores into a zeroed matrix)
__global__ __launch_bounds__(256) void k_scatter_blocks6(int64_t nslots, const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                                         const double *__restrict__ B, double *__restrict__ A, int64_t lda) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nslots * 36) return;
    const int64_t sl = t / 36;
    const int kk = (int)(t - sl * 36), k = kk / 6, l = kk - 6 * k;
    A[(6 * (int64_t)row[sl] + k) * lda + 6 * (int64_t)col[sl] + l] = B[t];
}

__global__ __launch_bounds__(256) void k_copy_block(int64_t n, const double *__restrict__ src, int64_t lds, double *__restrict__ dst, int64_t ldd) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int64_t r = e / n, c = e - r * n;
    dst[r * ldd + c] = src[r * lds + c];
}

// trace of the leading n x n part (single workgroup, fixed order)
__global__ __launch_bounds__(256) void k_trace(int64_t n, const double *__restrict__ A, int64_t ld, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) v += A[i * ld + i];
    v = block_sum(v, sh);
    if (threadIdx.x == 0) *out = v;
}

// A2 = Z^T A1 Z, one workgroup per aggregate g.  The (row node, column node) blocks of A1 whose row lies in g are
// listed per neighbouring aggregate ("slot") by the host; accumulator (slot, k, k') belongs to one thread, which adds
// the entries of its slot in list order: no atomics, bitwise reproducible.
__global__ __launch_bounds__(256) void k_galerkin_rows(const int64_t *__restrict__ sptr_g, const int32_t *__restrict__ slot_agg,
                                                       const int64_t *__restrict__ eptr, const int32_t *__restrict__ ent_i,
                                                       const int32_t *__restrict__ ent_j, const uint64_t *__restrict__ ent_loc,
                                                       const double *__restrict__ vals1, const double *__restrict__ Z,
                                                       const int64_t *__restrict__ aoff, double *__restrict__ A2, int64_t ld) {
    const int64_t g = blockIdx.x;
    const int64_t s0 = sptr_g[g], ns = sptr_g[g + 1] - s0;
    const int64_t og = aoff[g], ng = aoff[g + 1] - og;
    for (int64_t a = threadIdx.x; a < ns * 36; a += 256) {
        const int64_t sl = s0 + a / 36;
        const int kk = (int)(a % 36), k = kk / 6, kp = kk - 6 * k;
        const int64_t h = slot_agg[sl], oh = aoff[h], nh = aoff[h + 1] - oh;
        if (k >= ng || kp >= nh) continue;
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
        A2[(og + k) * ld + oh + kp] = acc;
    }
}

}  // namespace fcvm
