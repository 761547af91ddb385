// Vertex-patch smoother of the fine level (round 5).
//
// The fine level of the p-multigrid cycle was smoothed by one damped block-Jacobi step (3 x 3 nodal blocks) per side.  On a
// P2 -> P1 coarsening that is what limits the cycle: with an EXACT coarse solve the plate needs 37-38 PCG iterations and the
// Embankment (nu = 0.45) 54-65 (tests/tools/cycle_forms_cpu.py).  The additive Schwarz smoother over VERTEX PATCHES -- a corner
// node with the mid-side nodes of its edges, 3 (1 + degree) unknowns, ~38 on these meshes -- brings 25 and 32-40 there
// (tests/tools/patch_smoother_cpu.py), for one dense symmetric product per patch and side:
//     S = sum_c R_c^T (R_c K R_c^T)^-1 R_c,      cycle:  x1 = w S b;  x2 = x1 + P M1^-1 P^T (b - K~ x1);  x = x2 + w S (b - K~ x2)
// A mid-side node sits in the two patches of its edge's ends, a corner node in its own.  Patch matrices are cut out of the
// assembled matrix (symmetric window form), inverted in LDS (Cholesky, FP64) and kept as packed lower triangles in FP32; the
// product is formed per patch into a patch-major result vector and summed per node in a fixed order (a, then b), so the smoother
// is a fixed symmetric positive definite operator and bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcvm {

// tables of the patches (device): corner c <-> patch c; its nodes are c2f[c] followed by mids[mptr[c] .. mptr[c + 1]); node slot of
// position p of patch c = mptr[c] + c + p (the patch-major vectors have 3 doubles per slot)
struct PatchTab {
    const int64_t *mptr;
    const int32_t *mids, *c2f, *pa, *pb;
    const uint16_t *posa, *posb;     // position of a mid-side node in the patches of pa / pb (corner nodes: 0)
};

// one stored 3 x 3 block K_ij -> the LOWER triangle of every patch matrix that holds both nodes.  Every entry has exactly one writer: a
// block writes its own orientation where that is below the diagonal, and its transpose only when the symmetric form stores the pair
// once (`once`: the entry carries a transposed product; pairs of different groups are stored in both orientations, and their values
// may differ in the last bit)
__device__ __forceinline__ void patch_put(const PatchTab &t, int32_t i, int32_t j, const double (&b)[9], bool once, const int64_t *__restrict__ toff,
                                          double *__restrict__ T) {
    const int32_t ci = t.pa[i], di = t.pb[i], cj = t.pa[j], dj = t.pb[j];
    if (ci < 0 || cj < 0) return;
#pragma unroll
    for (int s = 0; s < 2; s++) {
        if (s && di == ci) break;
        const int32_t c = s ? di : ci;
        if (c != cj && c != dj) continue;
        const int pi = (ci == di) ? 0 : (int)(s ? t.posb[i] : t.posa[i]);
        const int pj = (cj == dj) ? 0 : (int)(c == cj ? t.posa[j] : t.posb[j]);
        const int64_t n = 3 * (t.mptr[c + 1] - t.mptr[c] + 1);
        double *base = T + toff[c];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int R = 3 * pi + r, C = 3 * pj + cc;
                if (R >= C) base[R * n + C] = b[3 * r + cc];
                else if (once && i != j) base[C * n + R] = b[3 * r + cc];
            }
    }
}

// patch matrices (dense, FP64, lower triangle of n_c x n_c at toff[c], zeroed before) from the FP64 values of the symmetric window form: one wavefront
// per storage wave, a lane walks the blocks of its row as the SpMV does (decode as matrix_to_coo_sym in fcvm_hip.hip)
__global__ __launch_bounds__(256) void k_patch_extract_sym(int64_t nwaves, int waves_per_group, const int64_t *__restrict__ woff,
                                                           const uint16_t *__restrict__ rowlen, const int32_t *__restrict__ rownode,
                                                           const uint32_t *__restrict__ cols, const double *__restrict__ vals,
                                                           const int64_t *__restrict__ gptr, const int32_t *__restrict__ wlist, PatchTab t,
                                                           const int64_t *__restrict__ toff, double *__restrict__ T) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const int64_t row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);      // lanes are sorted by length: lane 0 holds the longest row
    const int32_t i = rownode[row];
    const int64_t w0 = gptr[wave / waves_per_group];
    int64_t off = woff[wave];
    for (int j = 0; j < L; j++) {
        const bool p = len > j;
        const int n = __popcll(__builtin_amdgcn_ballot_w64(p));
        if (p && i >= 0) {
            const uint32_t en = cols[off + lane];
            const int32_t jn = wlist[w0 + (en & 0xFFFFu)];
            double b[9];
            const double *v = vals + 9 * off;
#pragma unroll
            for (int cmp = 0; cmp < 9; cmp++) b[cmp] = v[cmp < 8 ? (cmp >> 1) * 2 * n + 2 * lane + (cmp & 1) : 8 * n + lane];
            patch_put(t, i, jn, b, (en >> 16) != 0xFFFFu, toff, T);
        }
        off += n;
    }
}

// inverse of every patch matrix: one workgroup of 128 threads per patch, the matrix in LDS (row stride odd: columns are read as often as
// rows); Cholesky (right-looking), the factor inverted in place column by column from the last, K^-1 = X^T X written as the packed
// lower triangle in FP32 (row i at i (i + 1) / 2).  A pivot that is not positive sets *fail (the caller keeps block-Jacobi then).
__global__ __launch_bounds__(128) void k_patch_invert(const int64_t *__restrict__ mptr, const int64_t *__restrict__ toff, const double *__restrict__ T,
                                                      const int64_t *__restrict__ poff, float *__restrict__ inv, int *__restrict__ fail) {
    extern __shared__ double psh[];
    const int64_t c = blockIdx.x;
    const int n = 3 * (int)(mptr[c + 1] - mptr[c] + 1), ld = n | 1, tid = threadIdx.x;
    double *A = psh, *col = psh + (int64_t)n * ld;
    const double *src = T + toff[c];
    for (int idx = tid; idx < n * n; idx += 128) A[(idx / n) * ld + idx % n] = src[idx];
    __syncthreads();
    bool bad = false;
    for (int k = 0; k < n; k++) {
        const double d = A[k * ld + k];
        __syncthreads();
        if (!(d > 0.0) || !(d < 1e300)) bad = true;
        const double ip = bad ? 1.0 : 1.0 / sqrt(d);
        for (int i = k + tid; i < n; i += 128) A[i * ld + k] *= ip;
        __syncthreads();
        const int m = n - k - 1;
        for (int idx = tid; idx < m * m; idx += 128) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) A[i * ld + j] -= A[i * ld + k] * A[j * ld + k];
        }
        __syncthreads();
    }
    // X = L^-1 in place: column j from the columns behind it
    for (int j = n - 1; j >= 0; j--) {
        for (int k = j + tid; k < n; k += 128) col[k] = A[k * ld + j];
        __syncthreads();
        const double xjj = 1.0 / col[j];
        for (int i = j + tid; i < n; i += 128) {
            if (i == j) {
                A[j * ld + j] = xjj;
            } else {
                double s = 0.0;
                for (int k = j + 1; k <= i; k++) s += A[i * ld + k] * col[k];
                A[i * ld + j] = -s * xjj;
            }
        }
        __syncthreads();
    }
    float *dst = inv + poff[c];
    for (int idx = tid; idx < n * n; idx += 128) {
        const int i = idx / n, j = idx % n;
        if (j > i) continue;
        double s = 0.0;
        for (int k = i; k < n; k++) s += A[k * ld + i] * A[k * ld + j];
        dst[i * (i + 1) / 2 + j] = (float)s;
    }
    if (bad && tid == 0) atomicOr(fail, 1);
}

// y_c = K_c^-1 x|_c for every patch: one wavefront per patch (WPB patches per workgroup), the packed triangle and the gathered x in LDS,
// lane i forms row i in a fixed order (j ascending).  y is patch-major: 3 doubles per node slot.
template <int WPB>
__global__ __launch_bounds__(64 * WPB) void k_patch_apply(int64_t npatch, PatchTab t, const int64_t *__restrict__ poff, const float *__restrict__ inv,
                                                          const double *__restrict__ x, double *__restrict__ y, int maxn, int maxtri4,
                                                          const double *__restrict__ done) {
    extern __shared__ double psh[];
    if (done && *done != 0.0) return;
    const int wig = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * WPB + wig;
    double *xs = psh + (int64_t)wig * (maxn + maxtri4 / 2);
    float *tri = (float *)(xs + maxn);
    int n = 0;
    int64_t s0 = 0;
    if (c < npatch) {
        s0 = t.mptr[c];
        n = 3 * (int)(t.mptr[c + 1] - s0 + 1);
        const int nt4 = (n * (n + 1) / 2 + 3) >> 2;
        typedef float pv4f __attribute__((ext_vector_type(4)));
        const pv4f *src = (const pv4f *)(inv + poff[c]);
        for (int q = lane; q < nt4; q += 64) ((pv4f *)tri)[q] = __builtin_nontemporal_load(src + q);
        for (int k = lane; k < n; k += 64) {
            const int p = k / 3;
            const int32_t node = p == 0 ? t.c2f[c] : t.mids[s0 + p - 1];
            xs[k] = x[3 * (int64_t)node + (k - 3 * p)];
        }
    }
    __syncthreads();
    if (c >= npatch) return;
    double *yo = y + 3 * (s0 + c);
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        const float *ri = tri + i * (i + 1) / 2;
        for (int j = 0; j <= i; j++) acc += (double)ri[j] * xs[j];
        for (int j = i + 1; j < n; j++) acc += (double)tri[j * (j + 1) / 2 + i] * xs[j];
        yo[i] = acc;
    }
}

// per node: s = y[slot in patch a] (+ y[slot in patch b] for a mid-side node), a before b
//   MODE 0: out = w s                                          (pre-smoothing from zero)
//   MODE 1: out = x + w s, partials of r.out and r.r           (post-smoothing; layout of k_pcg_scalars: part[0..nb), part[nb..2 nb))
template <int MODE>
__global__ __launch_bounds__(256) void k_patch_combine(int64_t nn, PatchTab t, const double *__restrict__ y, double w, const double *__restrict__ x,
                                                       const double *__restrict__ r, double *__restrict__ out, double *__restrict__ part, int64_t nb,
                                                       const double *__restrict__ done) {
    if (done && *done != 0.0) return;
    __shared__ double sh[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double rz = 0.0, rr = 0.0;
    if (i < nn) {
        const int64_t a = t.pa[i], b = t.pb[i];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (a >= 0) {
            const double *ya = y + 3 * (t.mptr[a] + a + (a == b ? 0 : (int64_t)t.posa[i]));
            s0 = ya[0]; s1 = ya[1]; s2 = ya[2];
            if (a != b) {
                const double *yb = y + 3 * (t.mptr[b] + b + (int64_t)t.posb[i]);
                s0 += yb[0]; s1 += yb[1]; s2 += yb[2];
            }
        }
        if (MODE == 0) {
            out[3 * i] = w * s0; out[3 * i + 1] = w * s1; out[3 * i + 2] = w * s2;
        } else {
            const double z0 = x[3 * i] + w * s0, z1 = x[3 * i + 1] + w * s1, z2 = x[3 * i + 2] + w * s2;
            out[3 * i] = z0; out[3 * i + 1] = z1; out[3 * i + 2] = z2;
            if (part) {
                const double r0 = r[3 * i], r1 = r[3 * i + 1], r2 = r[3 * i + 2];
                rz = r0 * z0 + r1 * z1 + r2 * z2;
                rr = r0 * r0 + r1 * r1 + r2 * r2;
            }
        }
    }
    if (MODE == 1 && part) {
        rz = block_sum(rz, sh);
        rr = block_sum(rr, sh);
        if (threadIdx.x == 0) {
            part[blockIdx.x] = rz;
            part[nb + blockIdx.x] = rr;
        }
    }
}

}  // namespace fcvm
