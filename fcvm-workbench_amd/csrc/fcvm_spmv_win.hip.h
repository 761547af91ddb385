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

// Experiment of round 5 (FCVM_SPMV_TOUCH_KB): one load per `stride` bytes of an array from every workgroup -- 16 workgroups land on all eight
// XCDs, so each XCD's address translation caches hold the array's pages when the kernel that streams it starts.
__global__ __launch_bounds__(256) void k_touch_pages(const char *__restrict__ base, int64_t bytes, int64_t stride, int *__restrict__ sink) {
    int acc = 0;
    for (int64_t o = (int64_t)threadIdx.x * stride; o < bytes; o += 256 * stride) acc += __builtin_nontemporal_load((const int *)(base + o));
    if (acc == 0x7fffffff) *sink = acc;
}

// Which group of the window format a workgroup takes.  Workgroups go to the eight XCDs round-robin (blockIdx & 7); neighbouring groups share
// window nodes, so an XCD should work on neighbouring groups (its L2 then serves the second reader of a node's x).  mode 0 (rounds 2-4): every
// XCD walks one contiguous eighth of the groups -- eight streams through the value array, each an eighth of the array apart;
// mode 1: group = block (neighbours on different XCDs, one moving front); mode C >= 2: chunks of C consecutive groups per XCD, the eight XCDs
// side by side inside a super-chunk of 8 C groups -- one moving front of 8 C groups with the L2 sharing inside the chunks.  Round 5
// (tools/sessions/gmap_probe_r05.py, profiles/r05_spmv_slow_box.txt: four model instances in one process, every mode on each): C = 16 is
// 2-4 % faster than mode 0 for the FP64 SpMV and the 16-bit passes in every instance (0.3644 / 0.3418 / 0.3387 / 0.3348 against 0.3817 /
// 0.3548 / 0.3550 / 0.3418 ms back to back), identity and C = 4 / 64 in between: the default.  The results do not depend on the map (a group's
// numbers are formed by its own workgroup).  Per device: fcvm_set_group_map / FCVM_WIN_GMAP.
__device__ int g_win_gmap = 16;
__device__ __forceinline__ int64_t win_group_of_block(int64_t b, int64_t ngroups) {
    const int mode = g_win_gmap;
    if (mode == 1) return b;
    if (mode >= 2) {
        const int64_t C = mode, full = (ngroups / (8 * C)) * (8 * C);
        if (b >= full) return b;
        const int64_t xcd = b & 7, k = b >> 3;
        return (k / C) * (8 * C) + xcd * C + (k % C);
    }
    const int64_t q = ngroups >> 3, r8 = ngroups & 7, xcd = b & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
}


// position of component c of lane's block inside a block row of n lanes (9 * n doubles).  v2 = 0: nine runs of n doubles;
// v2 = 1: components (0,1), (2,3), (4,5), (6,7) as n double2's each (one 16-byte load per lane), component 8 as n doubles
__device__ __forceinline__ int win_value_at(int c, int n, int lane, int v2) {
    if (!v2) return c * n + lane;
    return c < 8 ? (c >> 1) * 2 * n + 2 * lane + (c & 1) : 8 * n + lane;
}

// values of the sliced-ELL array -> window format; thread = (wave, lane) vrow, walks its blocks.  Writes are coalesced over
// the lanes of a wave, reads follow the row's slice in the source layout.
// ---- 16-bit copy of the matrix for the smoother of the multiplicative cycle (k_spmv_win_lo) ---------------------------------
// A 3x3 block is stored as nine 16-bit integers with one power-of-two scale per block: value = q * 2^(e0 + e - 14), e0 the
// exponent of the largest diagonal entry of the matrix (global), e <= 0 the block's own (int8).  Why not FP16: the smoother
// applies K~ to smooth vectors (the coarse correction), where K x cancels to a small residual but the rounding errors of K~ do
// not; FP16's 2^-12 relative error per entry cost 14 % more PCG iterations on the 1 M-element plate than the FP64 smoother
// (53.6 -> 61.1, gpurun_out/r03d), fixed point on the block maximum is 2^-16 of it.  K_ji = K_ij^T have the same maximum, so the
// copy is exactly symmetric whenever the FP64 values are.  Entry off + lane of a block row: lo8 = components 0..7 (16 bytes),
// lo2 = {component 8, window column, block exponent, 0} (8 bytes): two loads, 24 bytes per block instead of 74.
typedef short short8 __attribute__((ext_vector_type(8)));
struct __attribute__((aligned(8))) WinLoTail {
    short q8;
    unsigned short col;
    signed char e;
    signed char pad[3];
};
static_assert(sizeof(WinLoTail) == 8, "WinLoTail must be 8 bytes");
// the same eight bytes in the symmetric form (WinSym layout): the spare bytes carry the position of the column node's row
// and the block exponent is stored ABSOLUTELY (frexp exponent of the block maximum, 16 bits; round 4): whoever writes a block
// needs nothing global -- the assembly kernel writes this layout itself, before the matrix exponent e0 is known -- and the
// kernels form e = eb - e0 when they read it
// kernels form e = eb - e0 when they read it.  The first four bytes change with every assembly, the last four are the pattern's
// (written once, k_lo2_static): the assembly kernel stores {q8, eb} as one 4-byte word without reading the entry
struct __attribute__((aligned(8))) WinLoTailS {
    short q8;
    short eb;
    unsigned short col;
    unsigned short ts;      // position in the group of the first segment of the column node's row, 0xFFFF: no transposed product
};
static_assert(sizeof(WinLoTailS) == 8, "WinLoTailS must be 8 bytes");

// out[0] = e0 (as a double): frexp exponent of max |dblk| -- every entry of an SPD matrix is bounded by its largest diagonal
// entry, so block exponents relative to e0 are <= 0.  part: block maxima of |dblk| (k_absmax_partial).
__global__ __launch_bounds__(256) void k_absmax_partial(int64_t n, const double *__restrict__ a, double *__restrict__ part) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v = fmax(v, fabs(a[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// How strongly is the mesh graded?  Per node the largest |diagonal entry| over its FREE dofs (supported dofs carry a unit-sized diagonal
// that no product of the solver touches); out[0] = max, out[1] = min (> 0) of that over the nodes.  The fixed-point sums of the symmetric
// window form take their unit from the GLOBAL maximum, so the transposed products of the region of smallest entries carry a relative
// error of 2^-54 * out[0] / out[1] (see k_spmv_win_sym): fcvm_assemble declines the form beyond FCVM_SYM_GRADING_MAX (1e4).
// k_diag_grading_partial: part[2 b], part[2 b + 1] = maximum, minimum of block b; k_diag_grading_final (one block) reduces them.
__global__ __launch_bounds__(256) void k_diag_grading_partial(int64_t nn, const double *__restrict__ dblk, const uint8_t *__restrict__ fixmask,
                                                              double *__restrict__ part) {
    __shared__ double shmax[4], shmin[4];
    double vmax = 0.0, vmin = 1.0e300;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nn; i += (int64_t)gridDim.x * 256) {
        const uint8_t fm = fixmask[i];
        double m = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (!((fm >> k) & 1)) m = fmax(m, fabs(dblk[9 * i + 4 * k]));
        if (m > 0.0) { vmax = fmax(vmax, m); vmin = fmin(vmin, m); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmax = fmax(vmax, __shfl_down(vmax, o, 64)); vmin = fmin(vmin, __shfl_down(vmin, o, 64)); }
    if ((threadIdx.x & 63) == 0) { shmax[threadIdx.x >> 6] = vmax; shmin[threadIdx.x >> 6] = vmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmax(fmax(shmax[0], shmax[1]), fmax(shmax[2], shmax[3]));
        part[2 * blockIdx.x + 1] = fmin(fmin(shmin[0], shmin[1]), fmin(shmin[2], shmin[3]));
    }
}
__global__ __launch_bounds__(256) void k_diag_grading_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    __shared__ double shmax[4], shmin[4];
    double vmax = 0.0, vmin = 1.0e300;
    for (int i = threadIdx.x; i < nb; i += 256) { vmax = fmax(vmax, part[2 * i]); vmin = fmin(vmin, part[2 * i + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { vmax = fmax(vmax, __shfl_down(vmax, o, 64)); vmin = fmin(vmin, __shfl_down(vmin, o, 64)); }
    if ((threadIdx.x & 63) == 0) { shmax[threadIdx.x >> 6] = vmax; shmin[threadIdx.x >> 6] = vmin; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fmax(fmax(shmax[0], shmax[1]), fmax(shmax[2], shmax[3]));
        out[1] = fmin(fmin(shmin[0], shmin[1]), fmin(shmin[2], shmin[3]));
    }
}
__global__ __launch_bounds__(256) void k_lo_scale(const double *__restrict__ part, int64_t n, double *__restrict__ out) {
    __shared__ double sh[4];
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) v = fmax(v, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        int e = 0;
        if (m > 0.0 && isfinite(m)) (void)frexp(m, &e);      // m = f * 2^e, f in [0.5, 1)
        out[0] = (double)e;
        out[1] = ldexp(1.0, e - 15);                          // value of one unit of a block with e = 0 ... see k_spmv_win_lo
    }
}

// lo8 / lo2 != NULL: also writes the 16-bit copy (above); cols: the window columns of the format (copied into lo2)
__global__ __launch_bounds__(256) void k_repack_win(int64_t nrows, const int32_t *__restrict__ srcslot,
                                                    const uint16_t *__restrict__ rowlen, const uint16_t *__restrict__ jstart,
                                                    const int64_t *__restrict__ woff, const int64_t *__restrict__ sptr,
                                                    const double *__restrict__ src, double *__restrict__ dst, int v2,
                                                    const double *__restrict__ lo_scale = nullptr, short8 *__restrict__ lo8 = nullptr,
                                                    WinLoTail *__restrict__ lo2 = nullptr, const uint16_t *__restrict__ cols = nullptr) {
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
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; c++) v[c] = sp[c * cs];
            if (dst) {      // (null: only the 16-bit copy is wanted -- the FP64 values go to the symmetric form, k_repack_sym)
#pragma unroll
                for (int c = 0; c < 9; c++) dp[win_value_at(c, n, lane, v2)] = v[c];
            }
            if (lo8) {
                const int e0 = (int)lo_scale[0];
                double m = 0.0;
#pragma unroll
                for (int c = 0; c < 9; c++) m = fmax(m, fabs(v[c]));
                int e = -127;                              // m = f * 2^eb, f in [0.5, 1): |v| * 2^(15 - eb) < 2^15
                if (m > 0.0) {
                    int eb;
                    (void)frexp(m, &eb);
                    e = eb - e0;
                    e = e < -127 ? -127 : (e > 127 ? 127 : e);   // (> 0 only for matrices that are not SPD, e.g. a shifted buckling operator)
                }
                const int sh = 15 - (e0 + e);
                short8 h;
                int q[9];
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    double t = rint(ldexp(v[c], sh));
                    t = t > 32767.0 ? 32767.0 : (t < -32767.0 ? -32767.0 : t);
                    q[c] = (int)t;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) h[c] = (short)q[c];
                lo8[off + lane] = h;
                WinLoTail tl;
                tl.q8 = (short)q[8];
                tl.col = cols[off + lane];
                tl.e = (signed char)e;
                tl.pad[0] = tl.pad[1] = tl.pad[2] = 0;
                lo2[off + lane] = tl;
            }
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

// ---- two right-hand sides: the matrix stream is read once for both ------------------------------------------------------------
// The PCG pairs of the GNLY Newton iteration (K ue = f and K due = r with one tangent matrix, fcVM.py:1351-1401) and of the
// buckling subspace iteration run in lockstep (fcvm_solve2); their three matrix passes per iteration use these kernels.
// LDS: x of the window for both vectors, interleaved per node [node][rhs][3]; a block's six x values are 48 contiguous bytes.
__device__ __forceinline__ void win_fma2(const WinSlot &t, bool active, const double *xs, double *y) {
    const double *xa = xs + 6 * t.col;
    const double a0 = active ? xa[0] : 0.0, a1 = active ? xa[1] : 0.0, a2 = active ? xa[2] : 0.0;
    const double b0 = active ? xa[3] : 0.0, b1 = active ? xa[4] : 0.0, b2 = active ? xa[5] : 0.0;
    y[0] += t.v[0] * a0 + t.v[1] * a1 + t.v[2] * a2;
    y[1] += t.v[3] * a0 + t.v[4] * a1 + t.v[5] * a2;
    y[2] += t.v[6] * a0 + t.v[7] * a1 + t.v[8] * a2;
    y[3] += t.v[0] * b0 + t.v[1] * b1 + t.v[2] * b2;
    y[4] += t.v[3] * b0 + t.v[4] * b1 + t.v[5] * b2;
    y[5] += t.v[6] * b0 + t.v[7] * b1 + t.v[8] * b2;
}
__device__ __forceinline__ void win_stage2(double *xs, int i, const double *__restrict__ x, int32_t nd, int rhs) {
    const double *xa = x + 3 * (int64_t)nd;
    const int o = nd & 1;            // 8 + 16 byte loads: even nodes are 16-byte aligned, odd ones 8 off
    const double2 wv = *(const double2 *)(xa + o);
    const double sv = xa[o ? 0 : 2];
    double *d = xs + 6 * i + 3 * rhs;
    d[0] = o ? sv : wv.x;
    d[1] = o ? wv.x : wv.y;
    d[2] = o ? wv.y : sv;
}

template <bool DOT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 2) void k_spmv_win2(int64_t ngroups, int64_t wmax, const int64_t *__restrict__ gptr,
                                                              const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                              const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                              const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg,
                                                              const uint16_t *__restrict__ own, const uint16_t *__restrict__ cols,
                                                              const double *__restrict__ vals, const double *__restrict__ xA,
                                                              const double *__restrict__ xB, double *__restrict__ yA, double *__restrict__ yB,
                                                              double *__restrict__ partA, double *__restrict__ partB,
                                                              const double *__restrict__ doneA, const double *__restrict__ doneB) {
    extern __shared__ double xs[];
    __shared__ double sh[2 * WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    if (doneA && doneB && *doneA != 0.0 && *doneB != 0.0) return;
    const int64_t b = blockIdx.x;
    const int64_t q = ngroups >> 3, r = ngroups & 7, xcd = b & 7;
    const int64_t g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    int64_t off = woff[wave];
    {
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            win_stage2(xs, i, xA, nd, 0);
            win_stage2(xs, i, xB, nd, 1);
        }
    }
    __syncthreads();
    double y[6] = {0, 0, 0, 0, 0, 0};
    {
        WinSlot a, bb, cc, dd;
        double z[6] = {0, 0, 0, 0, 0, 0}, u[6] = {0, 0, 0, 0, 0, 0}, w[6] = {0, 0, 0, 0, 0, 0};
        int j = 0;
        for (; j + 4 <= L; j += 4) {
            const bool p0 = len > j, p1 = len > j + 1, p2 = len > j + 2, p3 = len > j + 3;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0)), n1 = __popcll(__builtin_amdgcn_ballot_w64(p1)),
                      n2 = __popcll(__builtin_amdgcn_ballot_w64(p2)), n3 = __popcll(__builtin_amdgcn_ballot_w64(p3));
            win_load<true>(a, cols, vals, off, n0, lane);
            win_load<true>(bb, cols, vals, off + n0, n1, lane);
            win_load<true>(cc, cols, vals, off + n0 + n1, n2, lane);
            win_load<true>(dd, cols, vals, off + n0 + n1 + n2, n3, lane);
            win_fma2(a, p0, xs, y);
            win_fma2(bb, p1, xs, z);
            win_fma2(cc, p2, xs, u);
            win_fma2(dd, p3, xs, w);
            off += n0 + n1 + n2 + n3;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] += z[k] + u[k] + w[k];
        for (; j < L; j++) {
            const bool p0 = len > j;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
            win_load<true>(a, cols, vals, off, n0, lane);
            win_fma2(a, p0, xs, y);
            off += n0;
        }
    }
    const int32_t node = rownode[row];
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    if (gextra[g] > 0) {
        double *yp = xs + 6 * wmax + 6 * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) yp[k] = y[k];
        }
        __syncthreads();
        for (int s2 = 0; s2 + 1 < nseg; s2++) {
#pragma unroll
            for (int k = 0; k < 6; k++) y[k] += yp[6 * s2 + k];
        }
    }
    if (node >= 0 && nseg > 0) {
        double *ya = yA + 3 * (int64_t)node, *yb = yB + 3 * (int64_t)node;
        ya[0] = y[0]; ya[1] = y[1]; ya[2] = y[2];
        yb[0] = y[3]; yb[1] = y[4]; yb[2] = y[5];
    }
    if (DOT) {
        double dA = 0.0, dB = 0.0;
        if (node >= 0 && nseg > 0) {
            const double *xo = xs + 6 * (int)own[row];
            dA = y[0] * xo[0] + y[1] * xo[1] + y[2] * xo[2];
            dB = y[3] * xo[3] + y[4] * xo[4] + y[5] * xo[5];
        }
        dA = wave_sum(dA);
        dB = wave_sum(dB);
        if (lane == 0) { sh[wig] = dA; sh[WAVES + wig] = dB; }
        __syncthreads();
        if (tid == 0) {
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; i++) { t0 += sh[i]; t1 += sh[WAVES + i]; }
            partA[g] = t0;
            partB[g] = t1;
        }
    }
}

// ---- the same SpMV on the 16-bit copy of the values, for the smoother of the multiplicative cycle ---------------------------
// 24 instead of 74 bytes per block; x, the accumulation and everything written are FP64.  The SpMV result never leaves the
// registers: the epilogue is the cycle's step that needs it (no reference counterpart, see the header of this file).
//   EPI 0:                        out = K~ x                                               (multi-GPU layer: summed over the ranks first)
//   EPI 1 (after pre-smoothing):  out = r - K~ x                                           (residual for the coarse correction)
//   EPI 2 (post-smoothing):       out = x + c0 * Dinv (r - K~ x) ; partials of r.out and r.r (the PCG's z, r.z, r.r)
struct WinSlotLo {
    short8 a;
    WinLoTail t;
};
__device__ __forceinline__ void win_load_lo(WinSlotLo &t, const short8 *__restrict__ lo8, const WinLoTail *__restrict__ lo2,
                                            int64_t off, int n, int lane) {
    const int l = lane < n ? lane : n - 1;
    t.a = __builtin_nontemporal_load(lo8 + off + l);
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = __builtin_nontemporal_load((const u32x2 *)(lo2 + off + l));
    t.t = __builtin_bit_cast(WinLoTail, w);
}
// q . a with one fixed operation order (a product and two fused multiply-adds, written out: the compiler picks different
// contractions -- packed FP32 pairs -- for the one- and the two-vector kernels otherwise, and their results must be bitwise equal)
__device__ __forceinline__ float win_dot3(float q0, float q1, float q2, float a0, float a1, float a2) {
    return __builtin_fmaf(q2, a2, __builtin_fmaf(q1, a1, q0 * a0));
}
// FP32 inside the block products: q (16-bit integer) and x rounded to FP32 carry 2^-24 relative error against 2^-16 of the
// quantisation; the conversions and products run at the full VALU rate and x of the window takes half the LDS
__device__ __forceinline__ void win_fma_lo(const WinSlotLo &t, bool active, const float *xs, float &y0, float &y1, float &y2) {
    const float *xa = xs + 3 * (int)t.t.col;
    const int e = (int)t.t.e;
    const float sc = active ? __builtin_bit_cast(float, (127 + (e < -126 ? -126 : e)) << 23) : 0.0f;   // 2^e on x (exact)
    const float a0 = xa[0] * sc, a1 = xa[1] * sc, a2 = xa[2] * sc;
    y0 += win_dot3((float)(int)t.a[0], (float)(int)t.a[1], (float)(int)t.a[2], a0, a1, a2);
    y1 += win_dot3((float)(int)t.a[3], (float)(int)t.a[4], (float)(int)t.a[5], a0, a1, a2);
    y2 += win_dot3((float)(int)t.a[6], (float)(int)t.a[7], (float)(int)t.t.q8, a0, a1, a2);
}

#ifndef FCVM_LO_MINWAVES
#define FCVM_LO_MINWAVES 4
#endif
#ifndef FCVM_LO_INFLIGHT
#define FCVM_LO_INFLIGHT 8
#endif
template <int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES, FCVM_LO_MINWAVES) void k_spmv_win_lo(int64_t ngroups, int64_t wmax, const int64_t *__restrict__ gptr,
                                                                const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                                const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                                const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg,
                                                                const short8 *__restrict__ lo8, const WinLoTail *__restrict__ lo2,
                                                                const double *__restrict__ lo_scale, const double *__restrict__ x,
                                                                const double *__restrict__ r, double *__restrict__ out, double c0,
                                                                const float *__restrict__ dinv6, double *__restrict__ partial,
                                                                const double *__restrict__ done) {
    extern __shared__ float xsf[];      // x of the window in FP32 (3 per window node), then the partial results of later segments
    __shared__ double sh[2 * WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    if (done && *done != 0.0) return;
    const int64_t b = blockIdx.x;
    const int64_t q = ngroups >> 3, rr8 = ngroups & 7, xcd = b & 7;
    const int64_t g = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + (b >> 3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    int64_t off = woff[wave];
    int j = 0;
    {
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            const double *xa = x + 3 * (int64_t)nd;
            const int o = nd & 1;
            const double2 wv = *(const double2 *)(xa + o);
            const double sv = xa[o ? 0 : 2];
            xsf[3 * i] = (float)(o ? sv : wv.x);
            xsf[3 * i + 1] = (float)(o ? wv.x : wv.y);
            xsf[3 * i + 2] = (float)(o ? wv.y : sv);
        }
    }
    __syncthreads();
    float y0 = 0.f, y1 = 0.f, y2 = 0.f;
    {   // NB blocks in flight (24 bytes each)
        constexpr int NB = FCVM_LO_INFLIGHT;
        WinSlotLo s[NB];
        float z0 = 0.f, z1 = 0.f, z2 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
        for (; j + NB <= L; j += NB) {
            bool p[NB];
            int n[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                p[k] = len > j + k;
                n[k] = __popcll(__builtin_amdgcn_ballot_w64(p[k]));
            }
            int64_t o = off;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                win_load_lo(s[k], lo8, lo2, o, n[k], lane);
                o += n[k];
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if ((k & 3) == 0) win_fma_lo(s[k], p[k], xsf, y0, y1, y2);
                else if ((k & 3) == 1) win_fma_lo(s[k], p[k], xsf, z0, z1, z2);
                else if ((k & 3) == 2) win_fma_lo(s[k], p[k], xsf, u0, u1, u2);
                else win_fma_lo(s[k], p[k], xsf, w0, w1, w2);
            }
            off = o;
        }
        // the rest: block j always goes to accumulator set j & 3 (the chunk length is a multiple of 4), so the summation order
        // does not depend on the number of blocks in flight -- k_spmv_win_lo2 keeps the same order: bitwise the same result
        for (; j + 4 <= L; j += 4) {
            const bool p0 = len > j, p1 = len > j + 1, p2 = len > j + 2, p3 = len > j + 3;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0)), n1 = __popcll(__builtin_amdgcn_ballot_w64(p1)),
                      n2 = __popcll(__builtin_amdgcn_ballot_w64(p2)), n3 = __popcll(__builtin_amdgcn_ballot_w64(p3));
            win_load_lo(s[0], lo8, lo2, off, n0, lane);
            win_load_lo(s[1], lo8, lo2, off + n0, n1, lane);
            win_load_lo(s[2], lo8, lo2, off + n0 + n1, n2, lane);
            win_load_lo(s[3], lo8, lo2, off + n0 + n1 + n2, n3, lane);
            win_fma_lo(s[0], p0, xsf, y0, y1, y2);
            win_fma_lo(s[1], p1, xsf, z0, z1, z2);
            win_fma_lo(s[2], p2, xsf, u0, u1, u2);
            win_fma_lo(s[3], p3, xsf, w0, w1, w2);
            off += n0 + n1 + n2 + n3;
        }
        for (; j < L; j++) {
            const bool p0 = len > j;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
            win_load_lo(s[0], lo8, lo2, off, n0, lane);
            const int a = j & 3;
            if (a == 0) win_fma_lo(s[0], p0, xsf, y0, y1, y2);
            else if (a == 1) win_fma_lo(s[0], p0, xsf, z0, z1, z2);
            else if (a == 2) win_fma_lo(s[0], p0, xsf, u0, u1, u2);
            else win_fma_lo(s[0], p0, xsf, w0, w1, w2);
            off += n0;
        }
        y0 += z0 + u0 + w0; y1 += z1 + u1 + w1; y2 += z2 + u2 + w2;
    }
    const int32_t node = rownode[row];
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    if (gextra[g] > 0) {
        float *yp = xsf + 3 * wmax + 3 * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) { yp[0] = y0; yp[1] = y1; yp[2] = y2; }
        __syncthreads();
        for (int k = 0; k + 1 < nseg; k++) { y0 += yp[3 * k]; y1 += yp[3 * k + 1]; y2 += yp[3 * k + 2]; }
    }
    double rz = 0.0, rr = 0.0;
    if (EPI == 0) {     // plain product out = K~ x (the host-driven cycle of the multi-GPU layer exchanges it before using it)
        if (node >= 0 && nseg > 0) {
            const double is = lo_scale[1];
            const int64_t o = 3 * (int64_t)node;
            out[o] = is * y0; out[o + 1] = is * y1; out[o + 2] = is * y2;
        }
        return;
    }
    if (node >= 0 && nseg > 0) {
        const double is = lo_scale[1];
        const int64_t o = 3 * (int64_t)node;
        const double r0 = r[o], r1 = r[o + 1], r2 = r[o + 2];
        const double t0 = r0 - is * y0, t1 = r1 - is * y1, t2 = r2 - is * y2;
        if (EPI == 1) {
            out[o] = t0; out[o + 1] = t1; out[o + 2] = t2;
        } else {
            const float2 *d2 = (const float2 *)(dinv6 + 6 * (int64_t)node);
            const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];   // xx xy | xz yy | yz zz
            const double *xo = x + o;      // the FP64 x of the row's own node (the window in LDS holds FP32 roundings)
            const double z0 = xo[0] + c0 * ((double)u0.x * t0 + (double)u0.y * t1 + (double)u1.x * t2);
            const double z1 = xo[1] + c0 * ((double)u0.y * t0 + (double)u1.y * t1 + (double)u2.x * t2);
            const double z2 = xo[2] + c0 * ((double)u1.x * t0 + (double)u2.x * t1 + (double)u2.y * t2);
            out[o] = z0; out[o + 1] = z1; out[o + 2] = z2;
            rz = r0 * z0 + r1 * z1 + r2 * z2;
            rr = r0 * r0 + r1 * r1 + r2 * r2;
        }
    }
    if (EPI == 2) {
        rz = wave_sum(rz);
        rr = wave_sum(rr);
        if (lane == 0) { sh[wig] = rz; sh[WAVES + wig] = rr; }
        __syncthreads();
        if (tid == 0) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; i++) { a0 += sh[i]; a1 += sh[WAVES + i]; }
            partial[g] = a0;
            partial[ngroups + g] = a1;
        }
    }
}

// ---- the 16-bit smoother passes for two right-hand sides (see k_spmv_win2) ------------------------------------------------------
__device__ __forceinline__ void win_fma_lo2(const WinSlotLo &t, bool active, const float *xs, float *y) {
    const float *xa = xs + 6 * (int)t.t.col;
    const int e = (int)t.t.e;
    const float sc = active ? __builtin_bit_cast(float, (127 + (e < -126 ? -126 : e)) << 23) : 0.0f;
    const float a0 = xa[0] * sc, a1 = xa[1] * sc, a2 = xa[2] * sc, b0 = xa[3] * sc, b1 = xa[4] * sc, b2 = xa[5] * sc;
    const float q0 = (float)(int)t.a[0], q1 = (float)(int)t.a[1], q2 = (float)(int)t.a[2], q3 = (float)(int)t.a[3], q4 = (float)(int)t.a[4],
                q5 = (float)(int)t.a[5], q6 = (float)(int)t.a[6], q7 = (float)(int)t.a[7], q8 = (float)(int)t.t.q8;
    y[0] += win_dot3(q0, q1, q2, a0, a1, a2);
    y[1] += win_dot3(q3, q4, q5, a0, a1, a2);
    y[2] += win_dot3(q6, q7, q8, a0, a1, a2);
    y[3] += win_dot3(q0, q1, q2, b0, b1, b2);
    y[4] += win_dot3(q3, q4, q5, b0, b1, b2);
    y[5] += win_dot3(q6, q7, q8, b0, b1, b2);
}

// one right-hand side's share of the epilogue of k_spmv_win_lo (same arithmetic, same order)
template <int EPI>
__device__ __forceinline__ void win_lo_epilogue(int64_t o, double is, float y0, float y1, float y2, const double *__restrict__ x,
                                                const double *__restrict__ r, double *__restrict__ out, double c0,
                                                const float *__restrict__ dinv6, int64_t node, double &rz, double &rr) {
    const double r0 = r[o], r1 = r[o + 1], r2 = r[o + 2];
    const double t0 = r0 - is * y0, t1 = r1 - is * y1, t2 = r2 - is * y2;
    if (EPI == 1) {
        out[o] = t0; out[o + 1] = t1; out[o + 2] = t2;
    } else {
        const float2 *d2 = (const float2 *)(dinv6 + 6 * node);
        const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];   // xx xy | xz yy | yz zz
        const double *xo = x + o;
        const double z0 = xo[0] + c0 * ((double)u0.x * t0 + (double)u0.y * t1 + (double)u1.x * t2);
        const double z1 = xo[1] + c0 * ((double)u0.y * t0 + (double)u1.y * t1 + (double)u2.x * t2);
        const double z2 = xo[2] + c0 * ((double)u1.x * t0 + (double)u2.x * t1 + (double)u2.y * t2);
        out[o] = z0; out[o + 1] = z1; out[o + 2] = z2;
        rz = r0 * z0 + r1 * z1 + r2 * z2;
        rr = r0 * r0 + r1 * r1 + r2 * r2;
    }
}

template <int EPI, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 4) void k_spmv_win_lo2(int64_t ngroups, int64_t wmax, const int64_t *__restrict__ gptr,
                                                                 const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                                 const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                                 const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg,
                                                                 const short8 *__restrict__ lo8, const WinLoTail *__restrict__ lo2,
                                                                 const double *__restrict__ lo_scale, const double *__restrict__ xA,
                                                                 const double *__restrict__ xB, const double *__restrict__ rA,
                                                                 const double *__restrict__ rB, double *__restrict__ outA,
                                                                 double *__restrict__ outB, double c0, const float *__restrict__ dinv6,
                                                                 double *__restrict__ partA, double *__restrict__ partB,
                                                                 const double *__restrict__ doneA, const double *__restrict__ doneB) {
    extern __shared__ float xsf[];      // [window node][rhs][3] in FP32, then the partial results of later segments
    __shared__ double sh[4 * WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    const bool liveA = !(doneA && *doneA != 0.0), liveB = !(doneB && *doneB != 0.0);
    if (!liveA && !liveB) return;
    const int64_t b = blockIdx.x;
    const int64_t q = ngroups >> 3, rr8 = ngroups & 7, xcd = b & 7;
    const int64_t g = (xcd < rr8 ? xcd * (q + 1) : rr8 * (q + 1) + (xcd - rr8) * q) + (b >> 3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    int64_t off = woff[wave];
    {
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            const int o = nd & 1;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const double *xa = (h ? xB : xA) + 3 * (int64_t)nd;
                const double2 wv = *(const double2 *)(xa + o);
                const double sv = xa[o ? 0 : 2];
                xsf[6 * i + 3 * h] = (float)(o ? sv : wv.x);
                xsf[6 * i + 3 * h + 1] = (float)(o ? wv.x : wv.y);
                xsf[6 * i + 3 * h + 2] = (float)(o ? wv.y : sv);
            }
        }
    }
    __syncthreads();
    float y[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        constexpr int NB = 4;
        WinSlotLo s[NB];
        float z[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, u[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, w[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int j = 0;
        for (; j + NB <= L; j += NB) {
            bool p[NB];
            int n[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                p[k] = len > j + k;
                n[k] = __popcll(__builtin_amdgcn_ballot_w64(p[k]));
            }
            int64_t o = off;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                win_load_lo(s[k], lo8, lo2, o, n[k], lane);
                o += n[k];
            }
            win_fma_lo2(s[0], p[0], xsf, y);
            win_fma_lo2(s[1], p[1], xsf, z);
            win_fma_lo2(s[2], p[2], xsf, u);
            win_fma_lo2(s[3], p[3], xsf, w);
            off = o;
        }
        for (; j < L; j++) {    // block j -> accumulator set j & 3, as in k_spmv_win_lo
            const bool p0 = len > j;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
            win_load_lo(s[0], lo8, lo2, off, n0, lane);
            const int a = j & 3;
            if (a == 0) win_fma_lo2(s[0], p0, xsf, y);
            else if (a == 1) win_fma_lo2(s[0], p0, xsf, z);
            else if (a == 2) win_fma_lo2(s[0], p0, xsf, u);
            else win_fma_lo2(s[0], p0, xsf, w);
            off += n0;
        }
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] += z[k] + u[k] + w[k];
    }
    const int32_t node = rownode[row];
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    if (gextra[g] > 0) {
        float *yp = xsf + 6 * wmax + 6 * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) yp[k] = y[k];
        }
        __syncthreads();
        for (int s2 = 0; s2 + 1 < nseg; s2++) {
#pragma unroll
            for (int k = 0; k < 6; k++) y[k] += yp[6 * s2 + k];
        }
    }
    double rzA = 0.0, rrA = 0.0, rzB = 0.0, rrB = 0.0;
    if (node >= 0 && nseg > 0) {
        const double is = lo_scale[1];
        const int64_t o = 3 * (int64_t)node;
        if (liveA) win_lo_epilogue<EPI>(o, is, y[0], y[1], y[2], xA, rA, outA, c0, dinv6, node, rzA, rrA);
        if (liveB) win_lo_epilogue<EPI>(o, is, y[3], y[4], y[5], xB, rB, outB, c0, dinv6, node, rzB, rrB);
    }
    if (EPI == 2) {
        rzA = wave_sum(rzA); rrA = wave_sum(rrA); rzB = wave_sum(rzB); rrB = wave_sum(rrB);
        if (lane == 0) { sh[wig] = rzA; sh[WAVES + wig] = rrA; sh[2 * WAVES + wig] = rzB; sh[3 * WAVES + wig] = rrB; }
        __syncthreads();
        if (tid == 0) {
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int i = 0; i < WAVES; i++) { a0 += sh[i]; a1 += sh[WAVES + i]; b0 += sh[2 * WAVES + i]; b1 += sh[3 * WAVES + i]; }
            if (liveA) { partA[g] = a0; partA[ngroups + g] = a1; }
            if (liveB) { partB[g] = b0; partB[ngroups + g] = b1; }
        }
    }
}

// ---- level-1 matrix (sliced-ELL, tet4 on the corner nodes) in the same 16-bit block-scaled form ---------------------------------
// The V-cycle streams the level-1 matrix 2 * K1S times per PCG iteration (six at 1 M tet10): 40 B per block in FP32, 24 B here.
// One record per block PAIR and lane, three 16-byte vectors in separate arrays (coalesced over the lanes):
//   A = block 0 components 0..7;  B = {block 0 component 8, block 1 components 0..6};
//   C = {block 1 components 7, 8; exponents e0, e1 (int8); column nodes c0, c1 (int32)}
// value = q * 2^(E + e - 15), E = exponent of the largest diagonal entry of the level-1 matrix.  A slice of odd length ends in a
// record whose second block is empty (q = 0, column = the first block's).
typedef int int4v __attribute__((ext_vector_type(4)));

// thread = slot (slice, lane) of the level-1 pattern; walks the row's block pairs in the FP64 sliced-ELL array
__global__ __launch_bounds__(256) void k_l1_to_i16(int64_t nslots, const int64_t *__restrict__ sptr, const int64_t *__restrict__ rp,
                                                   const int32_t *__restrict__ cols, const double *__restrict__ vals,
                                                   const double *__restrict__ scale, int4v *__restrict__ A, int4v *__restrict__ B,
                                                   int4v *__restrict__ C) {
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    const int64_t s = slot >> 6, l = slot & 63;
    const int64_t jb = sptr[s], len = sptr[s + 1] - jb;
    const int e0 = (int)scale[0];
    const int64_t P = (len + 1) >> 1;
    for (int64_t p = 0; p < P; p++) {
        short q[18];
        int e[2], col[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int64_t j = 2 * p + h;
            double v[9];
            if (j < len) {
                const bool single = (len & 1) && j == len - 1;
                const int64_t unit = jb + (single ? j : (j & ~(int64_t)1));
                const double *sp = vals + unit * 576 + (single ? l : 2 * l + (j & 1));
                const int cs = single ? 64 : 128;
#pragma unroll
                for (int c = 0; c < 9; c++) v[c] = sp[c * cs];
                col[h] = single ? cols[(jb + j) * 64 + l] : cols[(jb + (j & ~(int64_t)1)) * 64 + 2 * l + (j & 1)];
            } else {
#pragma unroll
                for (int c = 0; c < 9; c++) v[c] = 0.0;
                col[h] = col[0];
            }
            double m = 0.0;
#pragma unroll
            for (int c = 0; c < 9; c++) m = fmax(m, fabs(v[c]));
            int eb = -127;
            if (m > 0.0) {
                int ex;
                (void)frexp(m, &ex);
                eb = ex - e0;
                eb = eb < -127 ? -127 : (eb > 127 ? 127 : eb);
            }
            e[h] = eb;
            const int sh = 15 - (e0 + eb);
#pragma unroll
            for (int c = 0; c < 9; c++) {
                double t = rint(ldexp(v[c], sh));
                t = t > 32767.0 ? 32767.0 : (t < -32767.0 ? -32767.0 : t);
                q[9 * h + c] = (short)(int)t;
            }
        }
        auto pk = [](short a, short b) { return (int)((unsigned)(unsigned short)a | ((unsigned)(unsigned short)b << 16)); };
        const int64_t at = (rp[s] + p) * 64 + l;
        A[at] = int4v{pk(q[0], q[1]), pk(q[2], q[3]), pk(q[4], q[5]), pk(q[6], q[7])};
        B[at] = int4v{pk(q[8], q[9]), pk(q[10], q[11]), pk(q[12], q[13]), pk(q[14], q[15])};
        C[at] = int4v{pk(q[16], q[17]), (int)((unsigned)(unsigned char)(signed char)e[0] | ((unsigned)(unsigned char)(signed char)e[1] << 8)), col[0], col[1]};
    }
}

struct L1Rec {
    int4v a, b, c;
};
template <bool NT>
__device__ __forceinline__ void l1_load(L1Rec &t, const int4v *__restrict__ A, const int4v *__restrict__ B, const int4v *__restrict__ C, int64_t at) {
    if (NT) {
        t.a = __builtin_nontemporal_load(A + at);
        t.b = __builtin_nontemporal_load(B + at);
        t.c = __builtin_nontemporal_load(C + at);
    } else {   // plain loads: the level-1 matrix (tens of MB) may stay in the memory-side cache between the passes of one V-cycle
        t.a = A[at]; t.b = B[at]; t.c = C[at];
    }
}
__device__ __forceinline__ float lo16(int w) { return (float)(int)(short)(w & 0xffff); }
__device__ __forceinline__ float hi16(int w) { return (float)(w >> 16); }
__device__ __forceinline__ void l1_gather(const double *__restrict__ x, int node, float sc, float &a0, float &a1, float &a2) {
    const double *xa = x + 3 * (int64_t)node;
    const int o = node & 1;
    const double2 wv = *(const double2 *)(xa + o);
    const double sv = xa[o ? 0 : 2];
    a0 = (float)(o ? sv : wv.x) * sc; a1 = (float)(o ? wv.x : wv.y) * sc; a2 = (float)(o ? wv.y : sv) * sc;
}
// the same from an FP32 twin of x with one 16-byte record per node (x, y, z, unused) -- round 5: one load and one cache line per gathered node instead of
// two loads over 24 bytes (the pass is bound by its gathers: 70 MB of matrix stream in 10.6 us, the pass took 24.5); the values are the roundings
// l1_gather forms itself, so the products are bitwise the same.  The twin is written by whoever writes the vector (k_restrict_init, k_l2_prolong, the
// epilogue of the previous pass).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void l1_gather32(const f4v *__restrict__ x32, int node, float sc, float &a0, float &a1, float &a2) {
    const f4v v = x32[node];
    a0 = v.x * sc; a1 = v.y * sc; a2 = v.z * sc;
}
__device__ __forceinline__ void l1_fma32(const L1Rec &t, const f4v *__restrict__ x32, float &y0, float &y1, float &y2) {
    const int e0 = (int)(signed char)(t.c[1] & 0xff), e1 = (int)(signed char)((t.c[1] >> 8) & 0xff);
    const float s0 = __builtin_bit_cast(float, (127 + (e0 < -126 ? -126 : e0)) << 23), s1 = __builtin_bit_cast(float, (127 + (e1 < -126 ? -126 : e1)) << 23);
    float a0, a1, a2, b0, b1, b2;
    l1_gather32(x32, t.c[2], s0, a0, a1, a2);
    l1_gather32(x32, t.c[3], s1, b0, b1, b2);
    y0 += win_dot3(lo16(t.a[0]), hi16(t.a[0]), lo16(t.a[1]), a0, a1, a2) + win_dot3(hi16(t.b[0]), lo16(t.b[1]), hi16(t.b[1]), b0, b1, b2);
    y1 += win_dot3(hi16(t.a[1]), lo16(t.a[2]), hi16(t.a[2]), a0, a1, a2) + win_dot3(lo16(t.b[2]), hi16(t.b[2]), lo16(t.b[3]), b0, b1, b2);
    y2 += win_dot3(lo16(t.a[3]), hi16(t.a[3]), lo16(t.b[0]), a0, a1, a2) + win_dot3(hi16(t.b[3]), lo16(t.c[0]), hi16(t.c[0]), b0, b1, b2);
}
__device__ __forceinline__ void l1_fma(const L1Rec &t, const double *__restrict__ x, float &y0, float &y1, float &y2) {
    const int e0 = (int)(signed char)(t.c[1] & 0xff), e1 = (int)(signed char)((t.c[1] >> 8) & 0xff);
    const float s0 = __builtin_bit_cast(float, (127 + (e0 < -126 ? -126 : e0)) << 23), s1 = __builtin_bit_cast(float, (127 + (e1 < -126 ? -126 : e1)) << 23);
    float a0, a1, a2, b0, b1, b2;
    l1_gather(x, t.c[2], s0, a0, a1, a2);
    l1_gather(x, t.c[3], s1, b0, b1, b2);
    y0 += win_dot3(lo16(t.a[0]), hi16(t.a[0]), lo16(t.a[1]), a0, a1, a2) + win_dot3(hi16(t.b[0]), lo16(t.b[1]), hi16(t.b[1]), b0, b1, b2);
    y1 += win_dot3(hi16(t.a[1]), lo16(t.a[2]), hi16(t.a[2]), a0, a1, a2) + win_dot3(lo16(t.b[2]), hi16(t.b[2]), lo16(t.b[3]), b0, b1, b2);
    y2 += win_dot3(lo16(t.a[3]), hi16(t.a[3]), lo16(t.b[0]), a0, a1, a2) + win_dot3(hi16(t.b[3]), lo16(t.c[0]), hi16(t.c[0]), b0, b1, b2);
}

// SpMV t = A1 d on the 16-bit copy with the Chebyshev smoother step of the row's node as epilogue (ChebEpi, as k_spmv<EPI = 1>);
// EPI 0: y = A1 x.  One wavefront per slice, one row per lane, four records in flight.
template <int EPI, bool NT = true, bool X32 = false>
__global__ __launch_bounds__(64) void k_spmv_l1_i16(int64_t nslices, const int64_t *__restrict__ rp, const int4v *__restrict__ A,
                                                    const int4v *__restrict__ B, const int4v *__restrict__ C,
                                                    const double *__restrict__ scale, const int32_t *__restrict__ rownode,
                                                    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ done,
                                                    ChebEpi epi, const f4v *__restrict__ x32 = nullptr, const int32_t *__restrict__ slist = nullptr) {
    if (done && *done != 0.0) return;
    const int64_t b = blockIdx.x, nb = nslices;
    const int64_t q = nb >> 3, r = nb & 7, xcd = b & 7;
    // slist (multi-GPU layer): the slices this launch covers -- those with a row of a node shared with another rank first, the rest beside the exchange
    const int64_t s = slist ? (int64_t)slist[b] : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int lane = threadIdx.x;
    const int64_t p0 = rp[s], P = rp[s + 1] - p0;
    // the epilogue's operands do not depend on the products: their loads are issued before the record loop, not after it (the
    // kernel is a chain of dependent round trips at 2900 one-wave workgroups: two fewer per pass)
    const int node = rownode[s * 64 + lane];
    const int64_t o = 3 * (int64_t)(node < 0 ? 0 : node);
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0;
    float2 q0 = {0.f, 0.f}, q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
    if (EPI != 0 && node >= 0) {
        d0 = x[o]; d1 = x[o + 1]; d2 = x[o + 2];
        e0 = epi.r[o]; e1 = epi.r[o + 1]; e2 = epi.r[o + 2];
        if (epi.update_d && epi.dinv6) {
            const float2 *q2p = (const float2 *)(epi.dinv6 + 6 * (int64_t)node);
            q0 = q2p[0]; q1 = q2p[1]; q2 = q2p[2];
        }
    }
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, z0 = 0.f, z1 = 0.f, z2 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
    int64_t at = p0 * 64 + lane, p = 0;
    L1Rec ra, rb, rc, rd;
    for (; p + 4 <= P; p += 4) {
        l1_load<NT>(ra, A, B, C, at); l1_load<NT>(rb, A, B, C, at + 64); l1_load<NT>(rc, A, B, C, at + 128); l1_load<NT>(rd, A, B, C, at + 192);
        if (X32) {
            l1_fma32(ra, x32, y0, y1, y2);
            l1_fma32(rb, x32, z0, z1, z2);
            l1_fma32(rc, x32, u0, u1, u2);
            l1_fma32(rd, x32, w0, w1, w2);
        } else {
        l1_fma(ra, x, y0, y1, y2);
        l1_fma(rb, x, z0, z1, z2);
        l1_fma(rc, x, u0, u1, u2);
        l1_fma(rd, x, w0, w1, w2);
        }
        at += 256;
    }
    for (; p < P; p++) {
        l1_load<NT>(ra, A, B, C, at);
        if (X32) l1_fma32(ra, x32, y0, y1, y2);
        else l1_fma(ra, x, y0, y1, y2);
        at += 64;
    }
    const double is = scale[1];
    double t0 = is * (double)(y0 + z0 + u0 + w0), t1 = is * (double)(y1 + z1 + u1 + w1), t2 = is * (double)(y2 + z2 + u2 + w2);
    if (node < 0) return;
    if (EPI == 0) {
        y[o] = t0; y[o + 1] = t1; y[o + 2] = t2;
        return;
    }
    if (epi.t2) { t0 += epi.t2[o]; t1 += epi.t2[o + 1]; t2 += epi.t2[o + 2]; }
    if (epi.defer && epi.defer[node]) {      // a node other ranks hold too: its step needs the sum of the ranks' products (k_unpack_sum_cheb)
        epi.t_out[o] = t0; epi.t_out[o + 1] = t1; epi.t_out[o + 2] = t2;
        return;
    }
    const double r0 = e0 - t0, r1 = e1 - t1, r2 = e2 - t2;
    epi.r[o] = r0; epi.r[o + 1] = r1; epi.r[o + 2] = r2;
    double a0 = d0, a1 = d1, a2 = d2;
    if (epi.update_d) {
        double zz0, zz1, zz2;
        if (epi.dinv6) {
            const float2 v0 = q0, v1 = q1, v2 = q2;   // xx xy | xz yy | yz zz
            zz0 = (double)v0.x * r0 + (double)v0.y * r1 + (double)v1.x * r2;
            zz1 = (double)v0.y * r0 + (double)v1.y * r1 + (double)v2.x * r2;
            zz2 = (double)v1.x * r0 + (double)v2.x * r1 + (double)v2.y * r2;
        } else {
            apply_dinv(epi.dinv + (int64_t)node * 9, r0, r1, r2, zz0, zz1, zz2);
        }
        const double n0 = epi.cdd * d0 + epi.cdr * zz0, n1 = epi.cdd * d1 + epi.cdr * zz1, n2 = epi.cdd * d2 + epi.cdr * zz2;
        epi.dout[o] = n0; epi.dout[o + 1] = n1; epi.dout[o + 2] = n2;
        if (epi.dout32) ((f4v *)epi.dout32)[node] = f4v{(float)n0, (float)n1, (float)n2, 0.f};
        if (epi.add_dout) { a0 += n0; a1 += n1; a2 += n2; }
    }
    epi.x[o] += a0; epi.x[o + 1] += a1; epi.x[o + 2] += a2;
}

// the pattern's half of the 16-bit tails of the symmetric form (window column, target lane), written once when the copy is allocated
__global__ __launch_bounds__(256) void k_lo2_static(int64_t n, const uint32_t *__restrict__ cols, WinLoTailS *__restrict__ lo2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t ct = cols[i];
    WinLoTailS tl;
    tl.q8 = 0;
    tl.eb = -1023;
    tl.col = (unsigned short)(ct & 0xFFFFu);
    tl.ts = (unsigned short)(ct >> 16);
    lo2[i] = tl;
}

// ---- symmetric form of the window format (WinSym, fcvm_pattern.h): the PCG's FP64 SpMV on 0.70 x the blocks --------------------
// values of the sliced-ELL array -> symmetric window storage (V2 component layout); thread = vrow, source block from jell
__global__ __launch_bounds__(256) void k_repack_sym(int64_t nrows, const int32_t *__restrict__ srcslot, const uint16_t *__restrict__ rowlen,
                                                    const int64_t *__restrict__ woff, const uint8_t *__restrict__ jell,
                                                    const int64_t *__restrict__ sptr, const double *__restrict__ src, double *__restrict__ dst,
                                                    const double *__restrict__ lo_scale = nullptr, short8 *__restrict__ lo8 = nullptr,
                                                    WinLoTailS *__restrict__ lo2 = nullptr, const uint32_t *__restrict__ cols = nullptr) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    const int64_t slot = srcslot[row], s = slot >> 6, ls = slot & 63;
    const int64_t jb = sptr[s], slen = sptr[s + 1] - jb;
    int64_t off = woff[row >> 6];
    // the source block numbers of the lane, four block rows ahead of the copy (a dependent load per block row otherwise)
    int jq[4] = {0, 0, 0, 0};
    {
        int64_t o2 = off;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int n = __popcll(__builtin_amdgcn_ballot_w64(len > k));
            if (len > k) jq[k] = jell[o2 + lane];
            o2 += n;
        }
    }
    int64_t offp = off;      // first entry of block row j + 4
    for (int k = 0; k < 4; k++) offp += __popcll(__builtin_amdgcn_ballot_w64(len > k));
    for (int j = 0; j < L; j++) {
        const int n = __popcll(__builtin_amdgcn_ballot_w64(len > j));
        const int np = __popcll(__builtin_amdgcn_ballot_w64(len > j + 4));
        const int64_t js_now = jq[0];
        jq[0] = jq[1]; jq[1] = jq[2]; jq[2] = jq[3];
        jq[3] = len > j + 4 ? (int)jell[offp + lane] : 0;
        offp += np;
        if (len > j) {
            const int64_t js = js_now;
            const bool single = (slen & 1) && js == slen - 1;
            const int64_t unit = jb + (single ? js : (js & ~(int64_t)1));
            const double *sp = src + unit * 576 + (single ? ls : 2 * ls + (js & 1));
            const int cs = single ? 64 : 128;
            double *dp = dst + 9 * off;
            double v[9];
#pragma unroll
            for (int c = 0; c < 9; c++) v[c] = sp[c * cs];
#pragma unroll
            for (int c = 0; c < 9; c++) dp[win_value_at(c, n, lane, 1)] = v[c];
            if (lo8) {      // 16-bit block-scaled copy, quantised exactly as in k_repack_win (the exponent stored absolutely)
                double m = 0.0;
#pragma unroll
                for (int c = 0; c < 9; c++) m = fmax(m, fabs(v[c]));
                int eb = -1023;
                if (m > 0.0) (void)frexp(m, &eb);
                const int sh = 15 - eb;
                short8 h;
                int q[9];
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    double t = rint(ldexp(v[c], sh));
                    t = t > 32767.0 ? 32767.0 : (t < -32767.0 ? -32767.0 : t);
                    q[c] = (int)t;
                }
#pragma unroll
                for (int c = 0; c < 8; c++) h[c] = (short)q[c];
                lo8[off + lane] = h;
                const uint32_t ct = cols[off + lane];
                WinLoTailS tl;
                tl.q8 = (short)q[8];
                tl.col = (unsigned short)(ct & 0xFFFFu);
                tl.eb = (short)eb;
                tl.ts = (unsigned short)(ct >> 16);
                lo2[off + lane] = tl;
            }
        }
        off += n;
    }
}

struct WinSlotS {
    double v[9];
    uint32_t ct;     // window position of the column node | position in the group of its row's first segment << 16 (0xFFFF: none)
};
__device__ __forceinline__ void win_load_sym(WinSlotS &t, const uint32_t *__restrict__ cols, const double *__restrict__ vals, int64_t off,
                                             int n, int lane) {
    const int l = lane < n ? lane : n - 1;
    t.ct = __builtin_nontemporal_load(cols + off + l);
    const double *vp = vals + 9 * off + 2 * l;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double2 w = ld_val<true>((const double2 *)(vp + (int64_t)2 * k * n));
        t.v[2 * k] = w.x;
        t.v[2 * k + 1] = w.y;
    }
    t.v[8] = __builtin_nontemporal_load(vals + 9 * off + (int64_t)8 * n + l);
}
// round(t * 2^sh) as a 64-bit two's-complement integer, |t * 2^sh| < 2^62, by shifting the mantissa: a dozen integer
// operations at the full VALU rate (the first version -- two multiplications, two roundings and two conversions in FP64, all at a
// quarter of that rate, three times per transposed product -- cost the kernel 10 % at sustained clocks).  Halves round away from zero.
__device__ __forceinline__ unsigned long long fixed62(double t, int sh) {
    const long long b = __double_as_longlong(t);
    const int e = (int)((b >> 52) & 0x7FF);
    const unsigned long long m = (unsigned long long)(b & 0xFFFFFFFFFFFFFll) | (e ? 0x10000000000000ull : 0ull);
    const int k = (e ? e : 1) - 1075 + sh;          // |t| = m * 2^((e ? e : 1) - 1075)
    unsigned long long q;
    if (k >= 0) q = m << (k > 10 ? 10 : k);          // (k <= 9 by the bound on |t * 2^sh|)
    else if (k > -64) q = (m + (1ull << (-k - 1))) >> (-k);
    else q = 0ull;
    return b < 0 ? 0ull - q : q;
}
__device__ __forceinline__ unsigned long long fixed62f(float t, int sh) {
    const int b = __float_as_int(t);
    const int e = (b >> 23) & 0xFF;
    const unsigned int m = (unsigned int)(b & 0x7FFFFF) | (e ? 0x800000u : 0u);
    const int k = (e ? e : 1) - 150 + sh;           // |t| = m * 2^((e ? e : 1) - 150)
    unsigned long long q;
    if (k >= 0) q = (unsigned long long)m << (k > 39 ? 39 : k);
    else if (k > -32) q = (unsigned long long)((m + (1u << (-k - 1))) >> (-k));
    else q = 0ull;
    return b < 0 ? 0ull - q : q;
}
// y += K_ij x_j and, where wanted, the row of the column node += K_ij^T x_i in 64-bit fixed point (unit 2^-sh), for NRHS vectors
// whose windows are interleaved per node in LDS ([node][rhs][3]).  Every product is spelled out as a chain of fused multiply-adds:
// the one- and the two-vector kernels must produce bitwise the same numbers per vector (fcvm_solve2 against fcvm_solve), and the
// compiler's own contraction of a * b + c * d + e * f differs between instantiations.
template <int NRHS>
__device__ __forceinline__ void win_fma_sym(const WinSlotS &t, bool active, const double *xs, const double (&o)[3 * NRHS],
                                            unsigned long long *yacc, const int (&sh)[NRHS], const bool (&tr_on)[NRHS],
                                            double (&y)[3 * NRHS]) {
    const double *xa = xs + 3 * NRHS * (int)(t.ct & 0xFFFFu);
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
        const double a0 = active ? xa[3 * r] : 0.0, a1 = active ? xa[3 * r + 1] : 0.0, a2 = active ? xa[3 * r + 2] : 0.0;
        y[3 * r] = __builtin_fma(t.v[2], a2, __builtin_fma(t.v[1], a1, __builtin_fma(t.v[0], a0, y[3 * r])));
        y[3 * r + 1] = __builtin_fma(t.v[5], a2, __builtin_fma(t.v[4], a1, __builtin_fma(t.v[3], a0, y[3 * r + 1])));
        y[3 * r + 2] = __builtin_fma(t.v[8], a2, __builtin_fma(t.v[7], a1, __builtin_fma(t.v[6], a0, y[3 * r + 2])));
    }
    const uint32_t ts = t.ct >> 16;
#ifdef FCVM_AB_NO_TR      // (timing experiment only: results are wrong without the transposed products)
    if (false) {
#else
    if (active && ts != 0xFFFFu) {
#endif
#pragma unroll
        for (int r = 0; r < NRHS; r++)
            if (tr_on[r]) {
                const double o0 = o[3 * r], o1 = o[3 * r + 1], o2 = o[3 * r + 2];
                const double t0 = __builtin_fma(t.v[6], o2, __builtin_fma(t.v[3], o1, t.v[0] * o0));
                const double t1 = __builtin_fma(t.v[7], o2, __builtin_fma(t.v[4], o1, t.v[1] * o0));
                const double t2 = __builtin_fma(t.v[8], o2, __builtin_fma(t.v[5], o1, t.v[2] * o0));
                unsigned long long *ya = yacc + 3 * NRHS * ts + 3 * r;
#if defined(FCVM_AB_PLAIN_LDS)      // (timing experiments only, wrong results: plain LDS stores instead of the 64-bit atomic adds ...)
                ya[0] = fixed62(t0, sh[r]); ya[1] = fixed62(t1, sh[r]); ya[2] = fixed62(t2, sh[r]);
#elif defined(FCVM_AB_NO_FIXED)     // (... the atomic adds without the fixed-point conversion in front of them)
                atomicAdd(ya, (unsigned long long)__double_as_longlong(t0));
                atomicAdd(ya + 1, (unsigned long long)__double_as_longlong(t1));
                atomicAdd(ya + 2, (unsigned long long)__double_as_longlong(t2));
#else
                atomicAdd(ya, fixed62(t0, sh[r]));
                atomicAdd(ya + 1, fixed62(t1, sh[r]));
                atomicAdd(ya + 2, fixed62(t2, sh[r]));
#endif
            }
    }
}

// vectors of a launch: NRHS right-hand sides share one pass over the matrix (fcvm_solve2: the GNLY pair, the buckling block)
template <int NRHS>
struct SymVecs {
    const double *x[NRHS];
    double *y[NRHS];
    double *partial[NRHS];
    const double *done[NRHS];
    const int32_t *glist = nullptr;   // launch over a SUBSET of the groups (grid = its length): workgroup b takes group glist[b] -- the multi-GPU layer's
                                      // boundary-groups-first split (fcvm_set_interface_nodes); null: all groups through win_group_of_block
};

// LDS: x of the window [3 NRHS wmax], partial results of later segments [3 NRHS xseg], fixed-point accumulators [3 NRHS * 64 * WAVES].
// kexp[0] = exponent e0 with |K entries| < 2^e0 (k_lo_scale).  The unit of the accumulators is chosen per group (and vector) from the largest
// |x| of its window: a row receives at most SYM_MAX_RECEIVED = 85 transposed products (counted by build_window_sym, which declines
// the form beyond that; recounted by check_window_sym) of three terms below 2^e0 * xmax each, so the sums stay below
// 3 * 85 * 2^(e0 + ex) < 2^(e0 + ex + 8) = 2^62 units of 2^(e0 + ex + 8 - 62): no wrap-around of the signed sums; one unit is
// 2^-54 of kmax * xmax, half an FP64 rounding error of the largest possible term (with four bits more margin the attainable
// residual of a solve to 1e-11 rose from 5e-11 to 6e-11 of |f|: the margin was taken out).  The unit follows the GLOBAL
// kmax: on a mesh graded by a factor s in element size the transposed products of the region of smallest entries carry a relative
// error of 2^-54 * kmax / K_local = 2^-54 * s (tet10 stiffness entries scale with h): 1e-13 at s = 1000 -- far below the PCG
// tolerances in use (1e-10 .. 1e-12); a mesh graded beyond 1e4 should run with FCVM_WIN_SYM=0.
// NRHS = 2 (round 4): both windows and both sets of accumulators in LDS -- one workgroup per CU instead of two, twice the
// registers per lane -- and every number per vector bitwise what the one-vector kernel gives (same products, same accumulator
// per block, integer sums).
#ifndef FCVM_SYM_MINWAVES
#define FCVM_SYM_MINWAVES 4
#endif
template <bool DOT, int WAVES, int NRHS>
__global__ __launch_bounds__(64 * WAVES, NRHS == 1 ? FCVM_SYM_MINWAVES : 2) void k_spmv_win_sym(int64_t ngroups, int64_t wmax, int64_t xseg, const int64_t *__restrict__ gptr,
                                                                 const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                                 const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                                 const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg,
                                                                 const uint16_t *__restrict__ own, const uint32_t *__restrict__ cols,
                                                                 const double *__restrict__ vals, const double *__restrict__ kexp,
                                                                 SymVecs<NRHS> V) {
    extern __shared__ double xs[];
    __shared__ double sh[NRHS * WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    {
        bool all_done = true;
#pragma unroll
        for (int r = 0; r < NRHS; r++) all_done = all_done && V.done[r] && *V.done[r] != 0.0;
        if (all_done) return;
    }
    const int64_t b = blockIdx.x;
    const int64_t g = V.glist ? (int64_t)V.glist[b] : win_group_of_block(b, ngroups);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    int64_t off = woff[wave];
    unsigned long long *yacc = (unsigned long long *)(xs + 3 * NRHS * (wmax + xseg));
    double xm[NRHS];
#pragma unroll
    for (int r = 0; r < NRHS; r++) xm[r] = 0.0;
    {
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            const int o = nd & 1;
#pragma unroll
            for (int r = 0; r < NRHS; r++) {
                const double *xa = V.x[r] + 3 * (int64_t)nd;
                const double2 wv = *(const double2 *)(xa + o);
                const double sv = xa[o ? 0 : 2];
                const double v0 = o ? sv : wv.x, v1 = o ? wv.x : wv.y, v2 = o ? wv.y : sv;
                double *xd = xs + 3 * NRHS * i + 3 * r;
                xd[0] = v0; xd[1] = v1; xd[2] = v2;
                xm[r] = fmax(xm[r], fmax(fabs(v0), fmax(fabs(v1), fabs(v2))));
                if (!(fabs(v0) + fabs(v1) + fabs(v2) < 1e300)) xm[r] = __builtin_huge_val();     // (fmax drops a NaN: carry it as Inf)
            }
        }
        for (int i = tid; i < 3 * NRHS * WIN_THREADS; i += WIN_THREADS) yacc[i] = 0ull;
#pragma unroll
        for (int r = 0; r < NRHS; r++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) xm[r] = fmax(xm[r], __shfl_xor(xm[r], o, 64));
            if (lane == 0) sh[r * WAVES + wig] = xm[r];
        }
    }
    __syncthreads();
    int shf[NRHS];
    bool tr_on[NRHS], poison[NRHS];
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
#pragma unroll
        for (int i = 0; i < WAVES; i++) xm[r] = fmax(xm[r], sh[r * WAVES + i]);
        int ex = 0;
        // x of the window not finite or beyond 1e300: the fixed-point unit cannot be formed.  The transposed products are not
        // silently left out (that would be the product with an unsymmetric matrix): the group's rows are written as NaN
        poison[r] = !(xm[r] < 1e300);
        tr_on[r] = xm[r] > 0.0 && !poison[r];
        if (tr_on[r]) (void)frexp(xm[r], &ex);
        shf[r] = 62 - ((int)kexp[0] + ex + 8);
    }
    const int32_t node = rownode[row];
    double o[3 * NRHS];
#pragma unroll
    for (int k = 0; k < 3 * NRHS; k++) o[k] = 0.0;
    if (node >= 0) {
        const double *xo = xs + 3 * NRHS * (int)own[row];
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) o[k] = xo[k];
    }
    __syncthreads();       // (sh is used again below)
    WinSlotS a, bb, cc;
    int j = 0;
    double y[3 * NRHS];
#pragma unroll
    for (int k = 0; k < 3 * NRHS; k++) y[k] = 0.0;
    {   // three blocks in flight, two accumulator sets (block j of a chunk of three goes to set j & 1; the rest to the first:
        // the same assignment for every NRHS, so the sums per vector do not depend on it)
        double z[3 * NRHS];
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) z[k] = 0.0;
        for (; j + 3 <= L; j += 3) {
            const bool p0 = len > j, p1 = len > j + 1, p2 = len > j + 2;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0)), n1 = __popcll(__builtin_amdgcn_ballot_w64(p1)),
                      n2 = __popcll(__builtin_amdgcn_ballot_w64(p2));
            win_load_sym(a, cols, vals, off, n0, lane);
            win_load_sym(bb, cols, vals, off + n0, n1, lane);
            win_load_sym(cc, cols, vals, off + n0 + n1, n2, lane);
            win_fma_sym<NRHS>(a, p0, xs, o, yacc, shf, tr_on, y);
            win_fma_sym<NRHS>(bb, p1, xs, o, yacc, shf, tr_on, z);
            win_fma_sym<NRHS>(cc, p2, xs, o, yacc, shf, tr_on, y);
            off += n0 + n1 + n2;
        }
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) y[k] += z[k];
    }
    for (; j < L; j++) {
        const bool p0 = len > j;
        const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
        win_load_sym(a, cols, vals, off, n0, lane);
        win_fma_sym<NRHS>(a, p0, xs, o, yacc, shf, tr_on, y);
        off += n0;
    }
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    __syncthreads();       // every transposed product of the group has arrived
    if (node >= 0 && nseg > 0) {
        const unsigned long long *ya = yacc + 3 * NRHS * (wig * 64 + lane);
#pragma unroll
        for (int r = 0; r < NRHS; r++)
            if (tr_on[r]) {
                y[3 * r] += ldexp((double)(long long)ya[3 * r], -shf[r]);
                y[3 * r + 1] += ldexp((double)(long long)ya[3 * r + 1], -shf[r]);
                y[3 * r + 2] += ldexp((double)(long long)ya[3 * r + 2], -shf[r]);
            }
    }
    if (gextra[g] > 0) {
        double *yp = xs + 3 * NRHS * wmax + 3 * NRHS * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) {
#pragma unroll
            for (int k = 0; k < 3 * NRHS; k++) yp[k] = y[k];
        }
        __syncthreads();
        for (int k = 0; k + 1 < nseg; k++) {
#pragma unroll
            for (int m = 0; m < 3 * NRHS; m++) y[m] += yp[3 * NRHS * k + m];
        }
    }
#pragma unroll
    for (int r = 0; r < NRHS; r++)
        if (poison[r]) y[3 * r] = y[3 * r + 1] = y[3 * r + 2] = __builtin_nan("");
    if (node >= 0 && nseg > 0) {
#pragma unroll
        for (int r = 0; r < NRHS; r++) {
            double *yo = V.y[r] + 3 * (int64_t)node;
            yo[0] = y[3 * r]; yo[1] = y[3 * r + 1]; yo[2] = y[3 * r + 2];
        }
    }
    if (DOT) {
        double d[NRHS];
#pragma unroll
        for (int r = 0; r < NRHS; r++) {
            d[r] = 0.0;
            if (node >= 0 && nseg > 0) d[r] = __builtin_fma(y[3 * r + 2], o[3 * r + 2], __builtin_fma(y[3 * r + 1], o[3 * r + 1], y[3 * r] * o[3 * r]));
            d[r] = wave_sum(d[r]);
            if (lane == 0) sh[r * WAVES + wig] = d[r];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int r = 0; r < NRHS; r++) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < WAVES; i++) t += sh[r * WAVES + i];
                V.partial[r][g] = t;
            }
        }
    }
}

// direct product as win_fma_lo; K~_ij^T x_i of a stored in-group block goes to row j's accumulator in 64-bit fixed point.
// NRHS vectors: windows interleaved per node ([node][rhs][3] floats), the block's nine values converted once.
template <int NRHS>
__device__ __forceinline__ void win_fma_lo_sym(const WinSlotLo &t, bool active, const float *xs, const float (&o)[3 * NRHS],
                                               unsigned long long *yacc, const int (&sh)[NRHS], const bool (&tr_on)[NRHS], int e0,
                                               float (&y)[3 * NRHS]) {
    const WinLoTailS ts8 = __builtin_bit_cast(WinLoTailS, t.t);
    const float *xa = xs + 3 * NRHS * (int)ts8.col;
    const int e = (int)ts8.eb - e0;       // <= 0 for an SPD matrix (every entry below its largest diagonal entry)
    const float sc = active ? __builtin_bit_cast(float, (127 + (e < -126 ? -126 : (e > 127 ? 127 : e))) << 23) : 0.0f;
    const float q0 = (float)(int)t.a[0], q1 = (float)(int)t.a[1], q2 = (float)(int)t.a[2], q3 = (float)(int)t.a[3], q4 = (float)(int)t.a[4],
                q5 = (float)(int)t.a[5], q6 = (float)(int)t.a[6], q7 = (float)(int)t.a[7], q8 = (float)(int)t.t.q8;
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
        const float a0 = xa[3 * r] * sc, a1 = xa[3 * r + 1] * sc, a2 = xa[3 * r + 2] * sc;
        y[3 * r] += win_dot3(q0, q1, q2, a0, a1, a2);
        y[3 * r + 1] += win_dot3(q3, q4, q5, a0, a1, a2);
        y[3 * r + 2] += win_dot3(q6, q7, q8, a0, a1, a2);
    }
    const unsigned ts = (unsigned)ts8.ts;
    if (active && ts != 0xFFFFu) {
#pragma unroll
        for (int r = 0; r < NRHS; r++)
            if (tr_on[r]) {
                const float b0 = o[3 * r] * sc, b1 = o[3 * r + 1] * sc, b2 = o[3 * r + 2] * sc;
                unsigned long long *ya = yacc + 3 * NRHS * ts + 3 * r;
                atomicAdd(ya, fixed62f(win_dot3(q0, q3, q6, b0, b1, b2), sh[r]));
                atomicAdd(ya + 1, fixed62f(win_dot3(q1, q4, q7, b0, b1, b2), sh[r]));
                atomicAdd(ya + 2, fixed62f(win_dot3(q2, q5, q8, b0, b1, b2), sh[r]));
            }
    }
}
// the prolongation of the coarse correction fused into the staging of the post-smoothing pass: x := x + P e1 (k_prolong_add's
// arithmetic, node by node in FP64, so the pass sees bitwise the vector the separate kernel would have written); e1 = null: off
struct ProlongFuse {
    const int32_t *pa, *pb;
    const uint8_t *fixmask;
    const double *e1;
};
__device__ __forceinline__ void prolong_fuse_add(const ProlongFuse &pf, int32_t nd, double &v0, double &v1, double &v2) {
    const int64_t a = pf.pa[nd];
    if (a < 0) return;
    const int64_t b = pf.pb[nd];
    const int fm = pf.fixmask[nd];
    const double *ea = pf.e1 + 3 * a, *eb = pf.e1 + 3 * b;
    v0 += (fm & 1) ? 0.0 : 0.5 * (ea[0] + eb[0]);
    v1 += (fm & 2) ? 0.0 : 0.5 * (ea[1] + eb[1]);
    v2 += (fm & 4) ? 0.0 : 0.5 * (ea[2] + eb[2]);
}
// vectors of a 16-bit pass: x (smoothed vector), r (right-hand side), out, partials of r.out and r.r (EPI 2), done flag
template <int NRHS>
struct LoSymVecs {
    const double *x[NRHS];
    const double *r[NRHS];
    double *out[NRHS];
    double *partial[NRHS];
    const double *done[NRHS];
    const int32_t *glist = nullptr;   // as SymVecs::glist
    const double *rw = nullptr;       // EPI 1 only: per-dof weights of r -- out = rw r - K~ x (multi-GPU layer: r counted once over the ranks that hold a node)
};
// one vector's share of the epilogue (one function for every NRHS: the same arithmetic in the same order)
template <int EPI>
__device__ __forceinline__ void win_lo_sym_epilogue(int64_t node, double is, double yd0, double yd1, double yd2, const double *__restrict__ x,
                                                    const double *__restrict__ r, double *__restrict__ out, double c0,
                                                    const float *__restrict__ dinv6, const ProlongFuse &pf, bool fuse, double &rz, double &rr,
                                                    const double *__restrict__ rw = nullptr) {
    const int64_t o = 3 * node;
    if (EPI == 0) {     // plain product out = K~ x (the host-driven cycle of the multi-GPU layer exchanges it before using it)
        out[o] = is * yd0; out[o + 1] = is * yd1; out[o + 2] = is * yd2;
        return;
    }
    double r0 = r[o], r1 = r[o + 1], r2 = r[o + 2];
    if (EPI == 1 && rw) { const double w = rw[o]; r0 *= w; r1 *= w; r2 *= w; }
    const double t0 = r0 - is * yd0, t1 = r1 - is * yd1, t2 = r2 - is * yd2;
    if (EPI == 1) {
        out[o] = t0; out[o + 1] = t1; out[o + 2] = t2;
    } else {
        const float2 *d2 = (const float2 *)(dinv6 + 6 * node);
        const float2 u0 = d2[0], u1 = d2[1], u2 = d2[2];   // xx xy | xz yy | yz zz
        const double *xo = x + o;      // the FP64 x of the row's own node (the window in LDS holds FP32 roundings)
        double x0 = xo[0], x1 = xo[1], x2 = xo[2];
        if (fuse) prolong_fuse_add(pf, (int32_t)node, x0, x1, x2);
        const double z0 = x0 + c0 * ((double)u0.x * t0 + (double)u0.y * t1 + (double)u1.x * t2);
        const double z1 = x1 + c0 * ((double)u0.y * t0 + (double)u1.y * t1 + (double)u2.x * t2);
        const double z2 = x2 + c0 * ((double)u1.x * t0 + (double)u2.x * t1 + (double)u2.y * t2);
        out[o] = z0; out[o + 1] = z1; out[o + 2] = z2;
        rz = r0 * z0 + r1 * z1 + r2 * z2;
        rr = r0 * r0 + r1 * r1 + r2 * r2;
    }
}
#ifndef FCVM_LOSYM_MINWAVES
#define FCVM_LOSYM_MINWAVES 8      // four workgroups per CU (31 KB of LDS, 58 registers) with two blocks in flight: 177 / 150 us; three with three: 181 / 152; two with six: 192 / 157
#endif
#ifndef FCVM_LOSYM_INFLIGHT
#define FCVM_LOSYM_INFLIGHT 2
#endif
// NRHS = 2 (round 4): both FP32 windows and both sets of accumulators in LDS (two workgroups per CU), per vector bitwise the
// one-vector kernel's numbers.  The prolongation fused into the staging (pf) exists for one vector only.
template <int EPI, int WAVES, int NRHS>
__global__ __launch_bounds__(64 * WAVES, NRHS == 1 ? FCVM_LOSYM_MINWAVES : 4) void k_spmv_win_lo_sym(int64_t ngroups, int64_t wmax, int64_t xseg, const int64_t *__restrict__ gptr,
                                                                const int32_t *__restrict__ gextra, const int32_t *__restrict__ wlist,
                                                                const int64_t *__restrict__ woff, const uint16_t *__restrict__ rowlen,
                                                                const int32_t *__restrict__ rownode, const uint32_t *__restrict__ seg, const uint16_t *__restrict__ own,
                                                                const short8 *__restrict__ lo8, const WinLoTail *__restrict__ lo2,
                                                                const double *__restrict__ lo_scale, LoSymVecs<NRHS> V, double c0,
                                                                const float *__restrict__ dinv6, ProlongFuse pf) {
    extern __shared__ float xsf[];      // x of the window in FP32 (3 NRHS per window node), then the partial results of later segments
    __shared__ double sh[2 * NRHS * WAVES];
    constexpr int WIN_THREADS = 64 * WAVES;
    {
        bool all_done = true;
#pragma unroll
        for (int r = 0; r < NRHS; r++) all_done = all_done && V.done[r] && *V.done[r] != 0.0;
        if (all_done) return;
    }
    const bool fuse = NRHS == 1 && EPI == 2 && pf.e1 != nullptr;
    const int e0 = (int)lo_scale[0];      // matrix exponent: block exponents are stored absolutely (WinLoTailS)
    const int64_t b = blockIdx.x;
    const int64_t g = V.glist ? (int64_t)V.glist[b] : win_group_of_block(b, ngroups);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wig = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wave = g * WAVES + wig, row = wave * 64 + lane;
    const int len = rowlen[row];
    const int L = __builtin_amdgcn_readfirstlane(len);
    int64_t off = woff[wave];
    int j = 0;
    unsigned long long *yacc = (unsigned long long *)(xsf + ((3 * NRHS * (wmax + xseg) + 1) & ~(int64_t)1));
    __shared__ float shx[NRHS * WAVES];
    float xm[NRHS];
#pragma unroll
    for (int r = 0; r < NRHS; r++) xm[r] = 0.f;
    {
        const int64_t w0 = gptr[g];
        const int W = (int)(gptr[g + 1] - w0);
        for (int i = tid; i < W; i += WIN_THREADS) {
            const int32_t nd = wlist[w0 + i];
            const int o = nd & 1;
#pragma unroll
            for (int r = 0; r < NRHS; r++) {
                const double *xa = V.x[r] + 3 * (int64_t)nd;
                const double2 wv = *(const double2 *)(xa + o);
                const double sv = xa[o ? 0 : 2];
                double d0 = o ? sv : wv.x, d1 = o ? wv.x : wv.y, d2 = o ? wv.y : sv;
                if (fuse) prolong_fuse_add(pf, nd, d0, d1, d2);
                const float v0 = (float)d0, v1 = (float)d1, v2 = (float)d2;
                float *xd = xsf + 3 * NRHS * i + 3 * r;
                xd[0] = v0; xd[1] = v1; xd[2] = v2;
                xm[r] = fmaxf(xm[r], fmaxf(fabsf(v0), fmaxf(fabsf(v1), fabsf(v2))));
                if (!(fabsf(v0) + fabsf(v1) + fabsf(v2) < 3e38f)) xm[r] = __builtin_huge_valf();
            }
        }
        for (int i = tid; i < 3 * NRHS * WIN_THREADS; i += WIN_THREADS) yacc[i] = 0ull;
#pragma unroll
        for (int r = 0; r < NRHS; r++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) xm[r] = fmaxf(xm[r], __shfl_xor(xm[r], o, 64));
            if (lane == 0) shx[r * WAVES + wig] = xm[r];
        }
    }
    __syncthreads();
    int shf[NRHS];
    bool tr_on[NRHS], poison[NRHS];
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
#pragma unroll
        for (int i = 0; i < WAVES; i++) xm[r] = fmaxf(xm[r], shx[r * WAVES + i]);
        int ex = 0;
        poison[r] = !(xm[r] < 3e38f);          // as in k_spmv_win_sym: NaN out instead of a product without its transposed half
        tr_on[r] = xm[r] > 0.f && !poison[r];
        if (tr_on[r]) (void)frexpf(xm[r], &ex);
        // products are in units of 2^(e0 - 15): |q 2^e| < 2^15, three terms, at most SYM_MAX_RECEIVED transposed products per row (see k_spmv_win_sym)
        shf[r] = 62 - (15 + ex + 8);
    }
    const int32_t node = rownode[row];
    float o[3 * NRHS];
#pragma unroll
    for (int k = 0; k < 3 * NRHS; k++) o[k] = 0.f;
    if (node >= 0) {
        const float *xo = xsf + 3 * NRHS * (int)own[row];
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) o[k] = xo[k];
    }
    float y[3 * NRHS];
#pragma unroll
    for (int k = 0; k < 3 * NRHS; k++) y[k] = 0.f;
    {   // NB blocks in flight (24 bytes each)
        constexpr int NB = FCVM_LOSYM_INFLIGHT;      // (eight in flight spill beside the transposed products)
        static_assert(NB >= 1 && NB <= 4, "accumulator sets: block k of a chunk goes to set k");
        WinSlotLo s[NB];
        float z[3 * NRHS], u[3 * NRHS], w[3 * NRHS];
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) z[k] = u[k] = w[k] = 0.f;
        for (; j + NB <= L; j += NB) {
            bool p[NB];
            int n[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                p[k] = len > j + k;
                n[k] = __popcll(__builtin_amdgcn_ballot_w64(p[k]));
            }
            int64_t o2 = off;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                win_load_lo(s[k], lo8, lo2, o2, n[k], lane);
                o2 += n[k];
            }
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if (k == 0) win_fma_lo_sym<NRHS>(s[k], p[k], xsf, o, yacc, shf, tr_on, e0, y);
                else if (k == 1) win_fma_lo_sym<NRHS>(s[k], p[k], xsf, o, yacc, shf, tr_on, e0, z);
                else if (k == 2) win_fma_lo_sym<NRHS>(s[k], p[k], xsf, o, yacc, shf, tr_on, e0, u);
                else win_fma_lo_sym<NRHS>(s[k], p[k], xsf, o, yacc, shf, tr_on, e0, w);
            }
            off = o2;
        }
        // the rest: block j goes to accumulator set j & 3 -- a fixed function of j and the wave's longest row, the same for every
        // NRHS, so the sums per vector do not depend on the number of vectors
        for (; j < L; j++) {
            const bool p0 = len > j;
            const int n0 = __popcll(__builtin_amdgcn_ballot_w64(p0));
            win_load_lo(s[0], lo8, lo2, off, n0, lane);
            const int a = j & 3;
            if (a == 0) win_fma_lo_sym<NRHS>(s[0], p0, xsf, o, yacc, shf, tr_on, e0, y);
            else if (a == 1) win_fma_lo_sym<NRHS>(s[0], p0, xsf, o, yacc, shf, tr_on, e0, z);
            else if (a == 2) win_fma_lo_sym<NRHS>(s[0], p0, xsf, o, yacc, shf, tr_on, e0, u);
            else win_fma_lo_sym<NRHS>(s[0], p0, xsf, o, yacc, shf, tr_on, e0, w);
            off += n0;
        }
#pragma unroll
        for (int k = 0; k < 3 * NRHS; k++) y[k] += z[k] + u[k] + w[k];
    }
    const uint32_t sc = seg[row];
    const int nseg = (int)(sc & 255u);
    __syncthreads();       // every transposed product of the group has arrived
    if (gextra[g] > 0) {
        float *yp = xsf + 3 * NRHS * wmax + 3 * NRHS * (int)(sc >> 8);
        if (node >= 0 && nseg == 0) {
#pragma unroll
            for (int k = 0; k < 3 * NRHS; k++) yp[k] = y[k];
        }
        __syncthreads();
        for (int k = 0; k + 1 < nseg; k++) {
#pragma unroll
            for (int m = 0; m < 3 * NRHS; m++) y[m] += yp[3 * NRHS * k + m];
        }
    }
    double rz[NRHS], rr[NRHS];
#pragma unroll
    for (int r = 0; r < NRHS; r++) {
        rz[r] = rr[r] = 0.0;
        // the row's sum in FP64: its own FP32 sums + the fixed-point sum of the transposed products
        double yd0 = (double)y[3 * r], yd1 = (double)y[3 * r + 1], yd2 = (double)y[3 * r + 2];
        if (poison[r]) yd0 = yd1 = yd2 = __builtin_nan("");
        if (node >= 0 && nseg > 0) {
            if (tr_on[r]) {
                const unsigned long long *ya = yacc + 3 * NRHS * (wig * 64 + lane) + 3 * r;
                yd0 += ldexp((double)(long long)ya[0], -shf[r]);
                yd1 += ldexp((double)(long long)ya[1], -shf[r]);
                yd2 += ldexp((double)(long long)ya[2], -shf[r]);
            }
            win_lo_sym_epilogue<EPI>((int64_t)node, lo_scale[1], yd0, yd1, yd2, V.x[r], V.r[r], V.out[r], c0, dinv6, pf, fuse, rz[r], rr[r], V.rw);
        }
    }
    if (EPI == 2) {
#pragma unroll
        for (int r = 0; r < NRHS; r++) {
            rz[r] = wave_sum(rz[r]);
            rr[r] = wave_sum(rr[r]);
            if (lane == 0) { sh[2 * r * WAVES + wig] = rz[r]; sh[(2 * r + 1) * WAVES + wig] = rr[r]; }
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int r = 0; r < NRHS; r++) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int i = 0; i < WAVES; i++) { a0 += sh[2 * r * WAVES + i]; a1 += sh[(2 * r + 1) * WAVES + i]; }
                V.partial[r][g] = a0;
                V.partial[r][ngroups + g] = a1;
            }
        }
    }
}

}  // namespace fcvm
