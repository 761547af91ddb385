// Host-side sparsity pattern of the tet10 stiffness matrix in the sliced-ELL 3x3-block layout
// the SpMV / assembly kernels use (see DESIGN.md "Matrix layout in HBM").
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <utility>
#include <vector>

namespace fcvm {

constexpr int SLICE = 64;           // rows (nodes) per slice = one wavefront, one row per lane
constexpr int SORT_WINDOW = 64;     // slices per length-sorting window (4096 rows); measured: 8 -> 64 = -4% bytes
constexpr int64_t HALF = 9 * SLICE;   // doubles of one block of all 64 lanes of a slice ("half chunk": 4.5 KiB)

// Storage of a slice of `len` blocks per lane (len = blocks of its longest row): the blocks 2p, 2p+1 of a lane form a
// pair whose nine components are stored as double2's, 64 lanes contiguous (one 1 KiB line per component: a 16-byte
// load per lane); if len is odd the last block stands alone, its components stored as 64 contiguous doubles each.
// A slice occupies exactly len half chunks (HALF doubles each), starting at half chunk sptr[s]: no rounding of odd
// lengths to pairs (round 2; 3.3 % of the matrix bytes on the cube meshes, whose row lengths 19 / 27 / 65 are all odd).
// std::vector whose resize() leaves new elements uninitialised: the big index arrays (element map 400 MB, columns 160 MB at
// 1 M elements) are written completely by the worker threads right after being sized, and value-initialising them first was a
// single-threaded pass over all of it (and the first touch of every page by one thread)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

// fn(begin, end, thread) over [0, n) cut into one contiguous chunk per thread (the same cut for the same n and nthreads);
// min_parallel: below this many items the loop runs on the calling thread (items of the per-node loops are tiny; the window
// format's items are whole groups of 512 rows: 3400 of them at 1 M elements, which the default threshold kept serial)
void parallel_for(int64_t n, int nthreads, const std::function<void(int64_t, int64_t, int)> &fn, int64_t min_parallel = 4096);

struct Pattern {
    int64_t ne = 0, nn = 0;
    int npe = 10;           // nodes per element (10: tet10 fine level, 4: tet4 coarse level)
    int64_t nslices = 0;
    int64_t nnzb = 0;       // true 3x3 blocks
    int64_t nslots = 0;     // stored block slots (sum over slices of 64 * slice length)
    int64_t maxlen = 0;
    std::vector<int64_t> rowptr;    // [nn+1] CSR over nodes (sorted columns, diagonal included)
    RawVec<int32_t> colidx;         // [nnzb]
    std::vector<int64_t> sptr;      // [nslices+1] cumulative slice lengths (blocks per lane) = first half chunk of a slice
    std::vector<int32_t> rownode;   // [nslices*64] node of slot, -1 = padding
    std::vector<int32_t> slotof;    // [nn] slot (slice*64+lane) of node
    RawVec<int32_t> cols;           // [sptr[nslices]*64] column nodes: see col_index
    RawVec<uint32_t> emap;          // [ne*npe*npe] packed location of element block (a,b), see pack_offset
    std::vector<uint64_t> diagoff;  // [nn] location of the diagonal block of a node, see abs_offset
    std::vector<int64_t> nptr;      // [nn+1] node -> element incidence (CSR)
    RawVec<int32_t> nel;            // [ne*npe] element << 4 | local node index, ascending elements per node

    inline int64_t slice_len(int64_t s) const { return sptr[s + 1] - sptr[s]; }
    inline bool is_single(int64_t s, int64_t j) const { const int64_t len = slice_len(s); return (len & 1) && j == len - 1; }
    // first half chunk of the storage unit (pair or single) that holds block j of slice s
    inline int64_t unit(int64_t s, int64_t j) const { return sptr[s] + (is_single(s, j) ? j : (j & ~(int64_t)1)); }
    // offset in doubles of component 0 of block j of the row living in `slot`; the components are comp_stride apart
    inline uint64_t block_offset(int64_t slot, int64_t j) const {
        const int64_t s = slot / SLICE, l = slot % SLICE;
        return (uint64_t)(unit(s, j) * HALF + (is_single(s, j) ? l : 2 * l + (j & 1)));
    }
    inline int64_t comp_stride(int64_t slot, int64_t j) const { return is_single(slot / SLICE, j) ? SLICE : 2 * SLICE; }
    // position in `cols` of the column node of block j of lane l of slice s: the two nodes of a pair adjacent per lane
    // (one 8-byte load), the node of a single block alone
    inline int64_t col_index(int64_t s, int64_t l, int64_t j) const {
        return is_single(s, j) ? (sptr[s] + j) * SLICE + l : (sptr[s] + (j & ~(int64_t)1)) * SLICE + 2 * l + (j & 1);
    }
    int64_t value_count() const { return sptr[nslices] * HALF; }
    // 32-bit form for the element maps: half chunk << 2 | (j & 1) << 1 | single.  The lane is not stored: every kernel
    // that places element blocks owns the row and knows its lane (unpack_offset in fcvm_kernels.hip.h).  2^30 half
    // chunks = 4.9 TB of values.
    inline uint32_t pack_offset(int64_t slot, int64_t j) const {
        const int64_t s = slot / SLICE;
        const bool single = is_single(s, j);
        return (uint32_t)(((uint64_t)unit(s, j) << 2) | (single ? 1u : (uint64_t)((j & 1) << 1)));
    }
    // 64-bit form for tables that are read without knowing the lane: offset of component 0 << 1 | single
    inline uint64_t abs_offset(int64_t slot, int64_t j) const { return (block_offset(slot, j) << 1) | (is_single(slot / SLICE, j) ? 1u : 0u); }
};

// elnodes0: 0-based int32 [ne*10].  Throws std::runtime_error on bad input.
// want_emap = false leaves Pattern::emap empty: fcvm_set_mesh builds the fine level's map (400 MB at 1 M elements) on the
// device from cols / sptr / slotof instead (k_build_emap; this function's map is the reference it is checked against).
void build_pattern(int64_t ne, int64_t nn, const int32_t *elnodes0, Pattern &p, int nthreads,
                   int sort_window = SORT_WINDOW, int npe = 10, const int32_t *row_order = nullptr, bool want_emap = true);

// ---------------------------------------------------------------------------------------------------------------
// Second storage of the same matrix for the SpMV alone ("window format", round 2).  The sliced-ELL layout above is what
// the assembly kernels write; its SpMV gathers x from global memory, 64 scattered addresses per instruction, and those
// gathers keep the address pipeline busy (DESIGN.md "SpMV").  Here
//   * rows are taken along a Morton curve; a row longer than WIN_SPLIT blocks is cut into equal SEGMENTS of at most
//     WIN_SPLIT blocks (the corner nodes of tet10 meshes couple to 65 nodes, the others to 19-27: without the cut a wave
//     would spend most of its time on a few long rows); rows and segments -- "vrows" -- are collected into GROUPS of at
//     most WIN_WAVES * 64 vrows (one workgroup; the segments of a row stay in one group);
//   * the union of the column nodes of a group -- its WINDOW, a compact patch of the mesh -- is listed once; the workgroup
//     copies x of the window into LDS with near-coalesced loads and every block's x comes from LDS;
//   * column entries are 16-bit positions in the window;
//   * the vrows of a group are sorted by length (longest first) and cut into waves of 64, longest in lane 0: for block
//     number j the lanes that still have a block are a PREFIX of the wave, n_j of them, and only those are stored (no
//     padding at all): entry (j, lane) of wave w is number woff[w] + sum_{i<j} n_i + lane, its nine values are vals[9 * (woff[w] + sum_{i<j} n_i) + c * n_j + lane];
//   * the partial results of the second, third .. segment of a row go through LDS to the lane of its first segment, which
//     adds them in segment order (fixed order: bitwise reproducible) and writes y.
// The values are copied from the sliced-ELL array by k_repack_win after every assembly (src: SELL slot of the row; both
// layouts keep a row's blocks in ascending column order, so block j of a segment is block jstart + j of the row).
#ifndef FCVM_WIN_WAVES
#define FCVM_WIN_WAVES 8           // (experiment builds: -DFCVM_WIN_WAVES=16, tools/sessions/gpu_r04zd.sh)
#endif
constexpr int WIN_WAVES = FCVM_WIN_WAVES;      // waves per group (default; WinPattern::waves holds the value a pattern was built with)
constexpr int WIN_SPLIT = 32;     // longest row segment (default; WinPattern::split)
struct WinPattern {
    int64_t ngroups = 0, nwaves = 0, nblocks = 0;
    int waves = WIN_WAVES;            // waves (of 64 vrows) per group
    int split = WIN_SPLIT;            // longest row segment
    int64_t wmax = 0;                 // largest window (nodes)
    int64_t xmax = 0;                 // most second-and-later segments in one group (LDS slots for their partial results)
    double wmean = 0.0;
    std::vector<int32_t> rownode;     // [nwaves*64] node of (wave, lane), -1 = no vrow
    std::vector<int32_t> srcslot;     // [nwaves*64] SELL slot (slice*64+lane) of that node
    std::vector<uint16_t> rowlen;     // [nwaves*64] blocks of the vrow (0 for empty lanes)
    std::vector<uint16_t> jstart;     // [nwaves*64] first block of the segment within its row
    std::vector<uint32_t> seg;        // [nwaves*64] LDS slot << 8 | segments: segments = 1 ordinary row; > 1: first segment of a
                                      // split row, adds the partials in slots slot .. slot + segments - 2; 0: later segment,
                                      // writes its partial to slot
    std::vector<uint16_t> own;        // [nwaves*64] window position of the row's own node
    std::vector<int64_t> woff;        // [nwaves+1] first entry of a wave
    std::vector<int64_t> gptr;        // [ngroups+1] first window entry of a group
    std::vector<int32_t> gextra;      // [ngroups] later segments in the group
    std::vector<int32_t> wlist;       // [gptr[ngroups]] window nodes, ascending per group
    std::vector<uint16_t> cols;       // [nblocks] window position of the column node of an entry
};
// false: the mesh does not suit the format (a window larger than max_window nodes)
bool build_window_format(const Pattern &p, const double *xyz, int nthreads, int64_t max_window, WinPattern &w, int waves = WIN_WAVES,
                         int split = WIN_SPLIT, const int32_t *given_order = nullptr);   // given_order: morton_order(nn, xyz) if at hand

// ---------------------------------------------------------------------------------------------------------------
// Symmetric form of the window format (round 3, the PCG's FP64 SpMV only).  K is symmetric and 2/3 of its blocks couple two
// nodes of the SAME group (measured on the plate meshes, groups of 512 rows).  Of such a pair K_ij, K_ji = K_ij^T only one is
// stored -- in row min(i, j) when i + j is even, in row max(i, j) otherwise, which keeps the row lengths balanced -- and the
// kernel forms both products of it: K_ij x_j into the row's own accumulator and K_ij^T x_i into the accumulator of row j,
// through LDS in 64-bit FIXED POINT (integer adds commute: the result does not depend on the order in which the lanes arrive,
// so runs stay bitwise reproducible).  Blocks to other groups and the diagonal blocks are stored as before.  Same groups, same
// windows (WinPattern::gptr / wlist are shared), same prefix layout; the rows are shorter, so the lanes of a group are sorted
// again and rows are cut into segments by their NEW length.  Entry of a stored block: window position of its column node in the
// low 16 bits, in the high 16 bits the position in the group (wave * 64 + lane) of the FIRST segment of the column node's row
// when the transposed product is wanted, 0xFFFF otherwise.  jell: number of the block in its sliced-ELL row (the copy kernel's
// source; rows are filtered, so "block j of the segment = block jstart + j of the row" no longer holds).
// Most transposed products a row may receive: each is three terms below kmax * xmax, the accumulators' unit is
// 2^(8 - 62) kmax * xmax (k_spmv_win_sym), so |sum| <= 3 * 85 * 2^54 < 2^62 units -- no wrap-around of the signed 64-bit sums.
// A row receives one product per in-group neighbour whose row keeps the pair, about half of its in-group neighbours: 85 means
// some 170 in-group neighbours (tet10 corner nodes have 65 neighbours in all on structured meshes, ~100 at 40 incident tets).
constexpr int SYM_MAX_RECEIVED = 85;
struct WinSym {
    int64_t ngroups = 0, nwaves = 0, nblocks = 0, dropped = 0;
    int64_t trmax = 0;                // most transposed products any row receives (<= SYM_MAX_RECEIVED, or the form is declined)
    int waves = WIN_WAVES, split = WIN_SPLIT;
    int64_t xmax = 0;                 // most second-and-later segments in one group
    std::vector<int32_t> rownode, srcslot;      // [nwaves*64]
    std::vector<uint16_t> rowlen, own;          // [nwaves*64]
    std::vector<uint32_t> seg;                  // [nwaves*64] as WinPattern::seg
    std::vector<int64_t> woff;                  // [nwaves+1]
    std::vector<int32_t> gextra;                // [ngroups]
    RawVec<uint32_t> cols;                      // [nblocks]
    RawVec<uint8_t> jell;                       // [nblocks]
    // --- what the assembly needs to write this layout itself (round 4: k_assemble_sym, no sliced-ELL detour, no copy kernel) ---
    // A row's KEPT blocks are numbered 0 .. rowtot-1 in the order its segments store them (the ones that want a transposed
    // product first, each class in column order); segment t holds the kept blocks jfirst .. jfirst + rowlen - 1.
    RawVec<uint8_t> kmap;                       // [p.nnzb] kept number of block q of the matrix (CSR order of Pattern), 0xFF: not stored
    std::vector<uint8_t> diagk;                 // [nn] kept number of the diagonal block of a node's row
    std::vector<uint16_t> jfirst;               // [nwaves*64] first kept block of the segment
    std::vector<uint8_t> rowtot;                // [nwaves*64] kept blocks of the whole row at its first segment's lane, 0 elsewhere
    std::vector<int32_t> nextseg;               // [nwaves*64] lane (wave*64+lane) of the row's next segment, -1: none
    int maxl = 0;                               // longest segment of the mesh (<= split)
    std::vector<uint16_t> rowoff;               // [nwaves*(maxl+1)] entry of block row j of a wave, relative to woff[wave]
};
// false: not applicable (a row with more than 255 blocks, or one that would receive more than SYM_MAX_RECEIVED transposed products)
bool build_window_sym(const Pattern &p, const WinPattern &w, int nthreads, WinSym &s);
// number of violated invariants (0 = every block of the matrix is covered exactly once, directly or as a transposed product)
int64_t check_window_sym(const Pattern &p, const WinPattern &w, const WinSym &s);

// nodes sorted along a Morton (Z-order) curve through their coordinates: a locality-preserving order in which
// the SpMV walks the rows when the caller's node numbering has none (unstructured meshes)
std::vector<int32_t> morton_order(int64_t nn, const double *xyz);
// mean |max - min| of the node numbers of an element, relative to nn: ~0 for band-like numberings
double numbering_spread(int64_t ne, int64_t nn, const int32_t *el, int npe);

}  // namespace fcvm
