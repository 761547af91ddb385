// Host-side sparsity pattern of the tet10 stiffness matrix in the sliced-ELL 3x3-block layout
// the SpMV / assembly kernels use (see DESIGN.md "Matrix layout in HBM").
#pragma once
#include <cstdint>
#include <vector>

namespace fcvm {

constexpr int SLICE = 64;           // rows (nodes) per slice = one wavefront, one row per lane
constexpr int SORT_WINDOW = 64;     // slices per length-sorting window (4096 rows); measured: 8 -> 64 = -4% bytes
constexpr int64_t VSTRIDE = 2 * SLICE;  // doubles between two components of one 3x3 block

struct Pattern {
    int64_t ne = 0, nn = 0;
    int npe = 10;           // nodes per element (10: tet10 fine level, 4: tet4 coarse level)
    int64_t nslices = 0;
    int64_t nnzb = 0;       // true 3x3 blocks
    int64_t nslots = 0;     // padded block slots (sum over slices of 64 * slice length)
    int64_t maxlen = 0;
    std::vector<int64_t> rowptr;    // [nn+1] CSR over nodes (sorted columns, diagonal included)
    std::vector<int32_t> colidx;    // [nnzb]
    std::vector<int64_t> sptr;      // [nslices+1] cumulative slice lengths (blocks per lane, even)
    std::vector<int32_t> rownode;   // [nslices*64] node of slot, -1 = padding
    std::vector<int32_t> slotof;    // [nn] slot (slice*64+lane) of node
    std::vector<int32_t> cols;      // [sptr[nslices]*64] column nodes, the two of a block pair adjacent per lane: see col_index
    std::vector<uint32_t> emap;     // [ne*npe*npe] packed location of component 0 of block (a,b), see pack_offset
    std::vector<uint32_t> diagoff;  // [nn] packed location of component 0 of the diagonal block
    std::vector<int64_t> nptr;      // [nn+1] node -> element incidence (CSR)
    std::vector<int32_t> nel;       // [ne*npe] element << 4 | local node index, ascending elements per node

    // offset in doubles of component 0 of block j of the row living in `slot`
    inline uint64_t block_offset(int64_t slot, int64_t j) const {
        int64_t s = slot / SLICE, l = slot % SLICE;
        return (uint64_t)(((sptr[s] / 2 + (j >> 1)) * 9 * SLICE + l) * 2 + (j & 1));
    }
    // position in `cols` of the column node of block j of lane l of slice s: one 8-byte load per lane and block pair
    inline int64_t col_index(int64_t s, int64_t l, int64_t j) const { return (((sptr[s] + j) >> 1) * SLICE + l) * 2 + (j & 1); }
    int64_t value_count() const { return sptr[nslices] * SLICE * 9; }
    // 32-bit form of block_offset for the device tables: pair row P = sptr[s]/2 + j/2 (25 bits), lane (6 bits), j & 1;
    // offset in doubles = ((P * 9 * 64) + lane) * 2 + (j & 1)  (unpack_offset in fcvm_kernels.hip.h).  2^25 pair rows
    // = 309 GB of values: beyond the HBM of one GPU.
    inline uint32_t pack_offset(int64_t slot, int64_t j) const {
        int64_t s = slot / SLICE, l = slot % SLICE;
        return (uint32_t)((((uint64_t)(sptr[s] / 2 + (j >> 1))) << 7) | ((uint64_t)l << 1) | (uint64_t)(j & 1));
    }
};

// elnodes0: 0-based int32 [ne*10].  Throws std::runtime_error on bad input.
void build_pattern(int64_t ne, int64_t nn, const int32_t *elnodes0, Pattern &p, int nthreads,
                   int sort_window = SORT_WINDOW, int npe = 10, const int32_t *row_order = nullptr);

// nodes sorted along a Morton (Z-order) curve through their coordinates: a locality-preserving order in which
// the SpMV walks the rows when the caller's node numbering has none (unstructured meshes)
std::vector<int32_t> morton_order(int64_t nn, const double *xyz);
// mean |max - min| of the node numbers of an element, relative to nn: ~0 for band-like numberings
double numbering_spread(int64_t ne, int64_t nn, const int32_t *el, int npe);

}  // namespace fcvm
