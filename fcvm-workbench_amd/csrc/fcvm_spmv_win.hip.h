// SpMV on the "window format" (fcvm_pattern.h, WinPattern): one workgroup = one group of at most WIN_WAVES * 64 rows /
// row segments taken along a Morton curve.  The workgroup first copies x of the group's window (the union of its column
// nodes, ~1300-1900 nodes on tet10 meshes) into LDS with near-coalesced loads; afterwards every block's x comes from LDS,
// so the only vector-memory instructions of the main loop are the coalesced value and column streams.  Why: in the
// sliced-ELL kernel every x gather is a 64-address instruction of its own and those gathers kept the address pipeline ~85 %
// busy, which made the kernel follow the shader clock (0.52 <-> 0.62 ms between clock states, DESIGN.md "SpMV").  Storage
// is padding-free: for block number j only the n_j lanes of a wave that still have a block are stored (lanes sorted by
// length, longest in lane 0).
//
// No reference counterpart (the reference factorises with CHOLMOD, fcVM.py:1121); restated by oracle.py's scipy SpMV in
// the parity tests.
#pragma once
#include "fcvm_kernels.hip.h"
#include "fcvm_pattern.h"

namespace fcvm {


// position of component c of lane's block inside a block row of n lanes (9 * n doubles).  v2 = 0: nine runs of n doubles;
// v2 = 1: components (0,1), (2,3), (4,5), (6,7) as n double2's each (one 16-byte load per lane), component 8 as n doubles
__device__ __forceinline__ int win_value_at(int c, int n, int lane, int v2) {
    if (!v2) return c * n + lane;
    return c < 8 ? (c >> 1) * 2 * n + 2 * lane + (c & 1) : 8 * n + lane;
}

// values of the sliced-ELL array -> window format; thread = (wave, lane) vrow, walks its blocks.  Writes are coalesced over
// the lanes of a wave, reads follow the row's slice in the source layout.
__global__ __launch_bounds__(256) void k_repack_win(int64_t nrows, const int32_t *__restrict__ srcslot,
                                                    const uint16_t *__restrict__ rowlen, const uint16_t *__restrict__ jstart,
                                                    const int64_t *__restrict__ woff, const int64_t *__restrict__ sptr,
                                                    const double *__restrict__ src, double *__restrict__ dst, int v2) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int len = rowlen[row], j0 = jstart[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    const int64_t slot = srcslot[row], s = slot >> 6, ls = slot & 63;
    const int64_t jb = sptr[s], slen = sptr[s + 1] - jb;
    int64_t off = woff[row >> 6];
    for (int j = 0; j < L; j++) {
        const int n = __popcll(__builtin_amdgcn_ballot_w64(len > j));
        if (len > j) {
            const int64_t js = j0 + j;      // block of the row in the sliced-ELL layout
            const bool single = (slen & 1) && js == slen - 1;
            const int64_t unit = jb + (single ? js : (js & ~(int64_t)1));
            const double *sp = src + unit * 576 + (single ? ls : 2 * ls + (js & 1));
            const int cs = single ? 64 : 128;
            double *dp = dst + 9 * off;
#pragma unroll
            for (int c = 0; c < 9; c++) dp[win_value_at(c, n, lane, v2)] = sp[c * cs];
        }
        off += n;
    }
}

struct WinSlot {
    double v[9];
    int col;
};

// block j of the wave: the first n lanes have one; the others read lane n-1's entry again (same cache lines, no extra
// traffic, no divergent control flow) and multiply it by x = 0
template <bool V2>
__device__ __forceinline__ void win_load(WinSlot &t, const uint16_t *__restrict__ cols, const double *__restrict__ vals,
                                         int64_t off, int n, int lane) {
    const int l = lane < n ? lane : n - 1;
    t.col = __builtin_nontemporal_load(cols + off + l);
    if (V2) {   // four 16-byte loads (8-byte aligned) + one 8-byte load
        const double *vp = vals + 9 * off + 2 * l;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double2 w = ld_val<true>((const double2 *)(vp + (int64_t)2 * k * n));
            t.v[2 * k] = w.x;
            t.v[2 * k + 1] = w.y;
        }
        t.v[8] = __builtin_nontemporal_load(vals + 9 * off + (int64_t)8 * n + l);
    } else {
        const double *vp = vals + 9 * off + l;
#pragma unroll
        for (int c = 0; c < 9; c++) t.v[c] = __builtin_nontemporal_load(vp + (int64_t)c * n);
    }
}
__device__ __forceinline__ void win_fma(const WinSlot &t, bool active, const double *xs, double &y0, double &y1, double &y2) {
    const double *xa = xs + 3 * t.col;
    const double a0 = active ? xa[0] : 0.0, a1 = active ? xa[1] : 0.0, a2 = active ? xa[2] : 0.0;
    y0 += t.v[0] * a0 + t.v[1] * a1 + t.v[2] * a2;
    y1 += t.v[3] * a0 + t.v[4] * a1 + t.v[5] * a2;
    y2 += t.v[6] * a0 + t.v[7] * a1 + t.v[8] * a2;
}

// WAVES: waves per workgroup (= per group of the pattern).  Measured and dropped (profiles/r02r_spmv_window_ab.txt): groups of
// 4 waves (0.503 against 0.487 ms), requesting a wave's first four blocks before the window is staged (0.520 ms).
template <bool DOT, int WAVES, bool V2>
__global__ __launch_bounds__(64 * WAVES, 4) void k_spmv_win(int64_t ngroups, int64_t wmax, const int64_t *__restrict__ gptr,
                                                             const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                             const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                             const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg,
                                                             const uint16_t *__restrict__ own, const uint16_t *__restrict__ cols,
                                                             const double *__restrict__ vals, const double *__restrict__ x,
                                                             double *__restrict__ y, double *__restrict__ partial,
                                                             const double *__restrict__ done) {
    extern __shared__ double xs[];      // x of the window (3 doubles per window node), then the partial results of later segments
    __shared__ double sh[WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    if (done && *done != 0.0) return;
    // XCD-aware remap (workgroups b, b+8, .. share an XCD): every XCD walks a contiguous range of groups
    const int64_t b = blockIdx.x;
    const int64_t q = ngroups >> 3, r = ngroups & 7, xcd = b & 7;
    const int64_t g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);     // lane 0 holds the longest vrow of the wave
    int64_t off = woff[wave];
    WinSlot a, bb, cc, dd;
    bool p0 = false, p1 = false, p2 = false, p3 = false;
    int j = 0;
    {   // stage x of the window: node numbers ascend, so neighbouring threads mostly read neighbouring nodes
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            const double *xa = x + 3 * (int64_t)nd;
            const int o = nd & 1;            // 8 + 16 byte loads: even nodes are 16-byte aligned, odd ones 8 off
            const double2 wv = *(const double2 *)(xa + o);
            const double sv = xa[o ? 0 : 2];
            xs[3 * i] = o ? sv : wv.x;
            xs[3 * i + 1] = o ? wv.x : wv.y;
            xs[3 * i + 2] = o ? wv.y : sv;
        }
    }
    __syncthreads();
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    {   // four blocks in flight
        double z0 = 0.0, z1 = 0.0, z2 = 0.0, u0 = 0.0, u1 = 0.0, u2 = 0.0, w0 = 0.0, w1 = 0.0, w2 = 0.0;
        for (; j + 4 <= L; j += 4) {
            p0 = len > j; p1 = len > j + 1; p2 = len > j + 2; p3 = len > j + 3;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0)), n1 = __popcll(__builtin_amdgcn_ballot_w64(p1)),
                      n2 = __popcll(__builtin_amdgcn_ballot_w64(p2)), n3 = __popcll(__builtin_amdgcn_ballot_w64(p3));
            win_load<V2>(a, cols, vals, off, n0, lane);
            win_load<V2>(bb, cols, vals, off + n0, n1, lane);
            win_load<V2>(cc, cols, vals, off + n0 + n1, n2, lane);
            win_load<V2>(dd, cols, vals, off + n0 + n1 + n2, n3, lane);
            win_fma(a, p0, xs, y0, y1, y2);
            win_fma(bb, p1, xs, z0, z1, z2);
            win_fma(cc, p2, xs, u0, u1, u2);
            win_fma(dd, p3, xs, w0, w1, w2);
            off += n0 + n1 + n2 + n3;
        }
        y0 += z0 + u0 + w0; y1 += z1 + u1 + w1; y2 += z2 + u2 + w2;
    }
    for (; j < L; j++) {
        p0 = len > j;
        const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
        win_load<V2>(a, cols, vals, off, n0, lane);
        win_fma(a, p0, xs, y0, y1, y2);
        off += n0;
    }
    const int32_t node = rownode[row];
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    if (gextra[g] > 0) {   // split rows in this group (uniform): later segments hand their partial results to the first one
        double *yp = xs + 3 * wmax + 3 * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) { yp[0] = y0; yp[1] = y1; yp[2] = y2; }
        __syncthreads();
        for (int k = 0; k + 1 < nseg; k++) { y0 += yp[3 * k]; y1 += yp[3 * k + 1]; y2 += yp[3 * k + 2]; }   // segment order
    }
    if (node >= 0 && nseg > 0) {
        double *yo = y + 3 * (int64_t)node;
        yo[0] = y0; yo[1] = y1; yo[2] = y2;
    }
    if (DOT) {
        double d = 0.0;
        if (node >= 0 && nseg > 0) {
            const double *xo = xs + 3 * (int)own[row];
            d = y0 * xo[0] + y1 * xo[1] + y2 * xo[2];
        }
        d = wave_sum(d);
        if (lane == 0) sh[wig] = d;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; i++) t += sh[i];
            partial[g] = t;
        }
    }
}

}  // namespace fcvm
