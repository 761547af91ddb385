// libfcvm_hip.so : C-ABI (include/fcvm_hip.h) over the gfx950 kernels in fcvm_kernels.hip.h.
#include "../../include/fcvm_hip.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <string>
#include <thread>
#include <type_traits>
#include <memory>
#include <vector>

#include "fcvm_kernels.hip.h"
#include "fcvm_spmv_win.hip.h"
#include "fcvm_mg_kernels.hip.h"
#include "fcvm_assemble_lds.hip.h"
#include "fcvm_dense.hip.h"
#include "fcvm_l2s_kernels.hip.h"
#include "fcvm_patch_kernels.hip.h"
#include "fcvm_pattern.h"

using namespace fcvm;

static thread_local std::string g_err;
static int fail(const std::string &m) {
    g_err = m;
    return 1;
}
#define HCHK(x)                                                                                     \
    do {                                                                                            \
        hipError_t e__ = (x);                                                                       \
        if (e__ != hipSuccess)                                                                      \
            return fail(std::string(#x) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" + \
                        std::to_string(__LINE__) + ")");                                            \
    } while (0)
#define NEED(cond, msg) \
    do {                \
        if (!(cond)) return fail(msg); \
    } while (0)

// device side of a WinPattern (window-format copy of one matrix)
struct WinDev {
    bool on = false;
    WinPattern pat;
    int64_t *gptr = nullptr, *woff = nullptr;
    int32_t *wlist = nullptr, *rownode = nullptr, *srcslot = nullptr, *gextra = nullptr;
    uint32_t *seg = nullptr;
    uint16_t *rowlen = nullptr, *own = nullptr, *cols = nullptr, *jstart = nullptr;
    void *vals = nullptr;   // 9 * nblocks doubles
    size_t lds = 0;
    // 16-bit block-scaled copy of the matrix for the smoother of the multiplicative cycle (mg_additive == 2): 24 bytes per block
    short8 *lo8 = nullptr;
    WinLoTail *lo2 = nullptr;
    double *lo_scale = nullptr;   // [2] on the device: global exponent e0, value of one unit at block exponent 0 (2^(e0 - 15))
    double *part_lo = nullptr;    // [2 * ngroups] partials of r.z and r.r left by the post-smoothing pass
    bool lo_valid = false;
};

// symmetric form of the window format (WinSym): its own row tables, entries and FP64 values; groups and windows are WinDev's
struct WinSymDev {
    bool on = false, valid = false;   // built for this mesh / holds the values of the last assembly
    WinSym pat;
    int64_t *woff = nullptr;
    int32_t *rownode = nullptr, *srcslot = nullptr, *gextra = nullptr;
    uint32_t *seg = nullptr, *cols = nullptr;
    uint16_t *rowlen = nullptr, *own = nullptr;
    uint8_t *jell = nullptr;
    double *vals = nullptr;
    size_t lds = 0;
    // 16-bit block-scaled copy in the same layout (the smoother passes of the multiplicative cycle)
    short8 *lo8 = nullptr;
    WinLoTailS *lo2 = nullptr;
    size_t lds_lo = 0;
    bool lo_valid = false;
    // assembly straight into this form (k_assemble_lds<.., SYM>, round 4): element map of kept block numbers, segment tables,
    // work items (wave * 4 + quarter, only quarters with a first segment) grouped by the kept length of their longest row
    bool asm_ok = false;
    uint8_t *emap = nullptr, *diagk = nullptr;
    uint16_t *jfirst = nullptr, *rowoff = nullptr;
    int32_t *nextseg = nullptr, *asm_items = nullptr;
    int maxl1 = 0;
    int asm_rows[7] = {ASM_ROWS, ASM_ROWS, ASM_ROWS, ASM_ROWS, ASM_ROWS, ASM_ROWS, ASM_ROWS};   // rows per work item of a class = 64 / its lanes per row
    static constexpr int ASM_BUCKETS = 7;
    int64_t asm_count[ASM_BUCKETS] = {0, 0, 0, 0, 0, 0, 0}, asm_first[ASM_BUCKETS] = {0, 0, 0, 0, 0, 0, 0}, asm_off[ASM_BUCKETS] = {0, 0, 0, 0, 0, 0, 0};
    double *part_vol = nullptr;     // [items] volume partials
    int64_t nitems = 0;
    // multi-GPU layer (fcvm_set_interface_nodes): the groups that hold a row of a node shared with another rank, and the others -- the three
    // fine-level passes launch the first list, hand the shared entries to the exchange, and launch the second while the messages travel
    int32_t *glist_b = nullptr, *glist_i = nullptr;
    int64_t n_gb = 0, n_gi = 0;
    bool split = false;
};

struct L2sDev {   // device state of the sparse aggregate level (owned by fcvm_ctx)
    bool on = false;
    int64_t nagg = 0, nslots = 0, nsuper = 0, n3 = 0, lda3 = 0;
    int k2 = 2, ncoarse = 1;   // smoothing steps per sweep; super-aggregate corrections per cycle (2: smooth, coarse, smooth, coarse, smooth)
    double lmax2 = 0.0, alpha2 = 8.0;
    int64_t *sptr_g = nullptr, *soff = nullptr, *sptr_s = nullptr;
    int32_t *slot_agg = nullptr, *sagg = nullptr, *smem = nullptr;
    float *blocks = nullptr, *A3invf = nullptr;
    double *dinv = nullptr, *C = nullptr;
    double *x = nullptr, *r = nullptr, *d = nullptr, *t = nullptr, *r3 = nullptr, *e3 = nullptr;
    double *symv_part3 = nullptr;   // [nt3][nt3][128] partial results of k_symv_tiles on the super-aggregate inverse (null: k_gemv_f32x4 streams the whole of it)
};

// Work vectors of the multigrid PCG for a second right-hand side (fcvm_solve2): the cycle's code works on the context's own
// pointers, so the second state is brought in by swapping pointers between launches (kernel arguments are taken at launch).
// vertex-patch smoother of the fine level (fcvm_patch_kernels.hip.h, fcvm_patch.inc)
struct PatchDev {
    bool on = false;       // tables built for this mesh (FCVM_MG_PATCH, patch sizes within the LDS)
    bool valid = false;    // the inverses belong to the matrix of the last assembly
    int64_t npatch = 0, nslots = 0, tri_total = 0, full_total = 0;
    int maxn = 0, maxtri4 = 0, wpb = 4;
    size_t lds_apply = 0, lds_invert = 0;
    uint16_t *posa = nullptr, *posb = nullptr;
    int64_t *poff = nullptr, *toff = nullptr;
    float *inv = nullptr;      // packed lower triangles of the patch inverses, FP32
    double *y = nullptr;       // patch-major result of k_patch_apply: 3 doubles per node slot
    int *fail = nullptr;
    double lmax = 0.0;         // largest eigenvalue of S K (Lanczos, once per mesh)
    int64_t refreshes = 0;
    double last_refresh_ms = 0.0;
};

struct MgAlt {
    bool ready = false;
    int64_t nn = 0, nc = 0, n2 = 0, nagg2 = 0, n3 = 0, ngroups = 0;
    double *r = nullptr, *z = nullptr, *p = nullptr, *q = nullptr, *S = nullptr, *part_vec = nullptr, *part_spmv = nullptr, *part_lo = nullptr;
    double *md = nullptr, *mr = nullptr, *mt = nullptr, *x1 = nullptr, *rr1 = nullptr, *d1 = nullptr, *t1 = nullptr, *r2 = nullptr, *e2 = nullptr;
    double *sx = nullptr, *sr = nullptr, *sd = nullptr, *st = nullptr, *sr3 = nullptr, *se3 = nullptr;   // sparse aggregate level
    float *d1_32 = nullptr, *t1_32 = nullptr;   // FP32 twins of d1 / t1 (gathers of k_spmv_l1_i16)
    double *spart3 = nullptr;      // ... and its own partial buffer of the super-aggregate level's k_symv_tiles
    double *py = nullptr;          // ... and of the patch smoother's patch-major result vector
    double *symv_part = nullptr;   // its own partial buffer of k_symv_tiles: the two coarse parts of a pair run side by side
    bool symv = false;
};

// host half of the level-1 (tet4) topology, see mg_topology_host in fcvm_mg.inc
struct TopoHost {
    bool ok = false;
    std::string err;
    int64_t nc = 0;
    std::vector<int32_t> c2f, el4, pa, pb, mids;
    std::vector<int64_t> mptr;
    std::vector<uint16_t> posa, posb;   // position of a mid-side node in the vertex patches of pa / pb (1 + its index in mids of that corner)
    Pattern pat1;
};

struct fcvm_ctx {
    int device = 0;
    hipStream_t own = nullptr, stream = nullptr;
    Pattern pat;
    int64_t ne = 0, nn = 0;
    bool have_mesh = false, have_mat = false, assembled = false;
    double E = 0, nu = 0, rho = 0;
    // device mesh
    int32_t *d_el = nullptr, *d_cols = nullptr, *d_rownode = nullptr, *d_noce = nullptr;
    int64_t *d_sptr = nullptr;
    uint32_t *d_emap = nullptr;
    uint64_t *d_diagoff = nullptr, *d_diagoff1 = nullptr;
    int64_t *d_nptr = nullptr, *d_nptr1 = nullptr;   // node -> element incidence for the row-owner assembly (fine / tet4 level)
    int32_t *d_nel = nullptr, *d_nel1 = nullptr;
    int assemble_lds = 1;        // rows accumulated in LDS, one coalesced write per value (FCVM_ASSEMBLE_LDS=0: global read-modify-writes)
    // work items of k_assemble_lds (slice * 4 + quarter), grouped by the LDS a quarter slice needs
    static constexpr int ASM_BUCKETS = 7;
    int32_t *d_asm_items = nullptr;
    int64_t asm_count[ASM_BUCKETS] = {0, 0, 0, 0, 0, 0, 0}, asm_first[ASM_BUCKETS] = {0, 0, 0, 0, 0, 0, 0};
    bool asm_lds_ok = false;
    // the length classes run side by side on streams of their own (a CU holds one long-row item next to two short-row ones)
    hipStream_t asm_stream[ASM_BUCKETS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t asm_fork = nullptr, asm_join[ASM_BUCKETS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int asm_overlap = 1;         // FCVM_ASSEMBLE_OVERLAP=0: one stream, class after class
    int qin_atomic = 0;          // FCVM_QIN_ATOMIC=1: internal forces scattered with f64 atomics
    double *d_fe = nullptr;      // [ne][10][3] element nodal forces of the last stress update
    double *d_part_vol = nullptr;  // [nslices] volume partials of the row-owner assembly
    uint8_t *d_fixmask = nullptr;
    double *d_coord = nullptr, *d_fixval = nullptr, *d_vals = nullptr, *d_dblk = nullptr, *d_dinv = nullptr;
    double *d_vals_g = nullptr;  // geometric stiffness (linear buckling), same pattern as d_vals
    // pcg work
    double *d_r = nullptr, *d_z = nullptr, *d_p = nullptr, *d_q = nullptr, *d_S = nullptr;
    double *d_part_vec = nullptr, *d_part_spmv = nullptr, *d_part_misc = nullptr, *d_out8 = nullptr, *d_part_mdot = nullptr;
    long long *d_pidx = nullptr;
    int64_t nb_vec = 0, nb_spmv = 0, nb_gp = 0;
    int64_t nb_part_spmv = 0;   // entries of d_part_spmv: max(workgroups of the sliced-ELL kernel, groups of the window format)
    int spmv_spw = 1;  // waves per SpMV workgroup: 1 keeps the tail of the launch short (long and short slices differ 3x): -5 % vs 4
    double last_spmv_ms = 0, last_total_ms = 0;
    double row_spread = 0.0;
    bool row_order_morton = false;
    // window-format copies of the matrices for the SpMV (fcvm_spmv_win.hip.h); 0: the sliced-ELL kernel does it
    bool win2_ok = false;           // twice the window fits the LDS: k_spmv_win2 can run
    int level1_overlap = 1;         // FCVM_LEVEL1_OVERLAP=0: the level-1 operator of a tangent refresh after the fine assembly, on the main stream
    hipStream_t level1_stream = nullptr;
    hipEvent_t level1_fork = nullptr, level1_done = nullptr;
    // round 5: where the driver puts an allocation decides 10-15 % of the time of a kernel that streams it or scatters its results into it
    // (the same binary, box and matrix: 0.335 or 0.385 ms for the FP64 SpMV -- profiles/r05_spmv_slow_box.txt); after the first stiffness
    // assembly of a mesh the PCG's result vector q and the value array of the symmetric form are each timed against fresh allocations
    // and the fastest is kept (tune_sym_placement; FCVM_TUNE_PLACEMENT = candidates per array, 1 = off)
    int tune_tries = 3;
    bool placement_tuned = false;
    double tune_info[6] = {0, 0, 0, 0, 0, 0};   // tuned, q first / best [ms], value array first / best [ms], time spent [ms]
    int pcg_events = 1;             // FCVM_PCG_EVENTS=0: no HIP events around the PCG's SpMV launches (no in-solver SpMV timing; experiment of round 5)
    int pair_overlap = 1;           // FCVM_PAIR_OVERLAP=0: the coarse parts of a pair of solves one after the other on the main stream
    hipStream_t pair_stream = nullptr;
    hipEvent_t pair_fork = nullptr, pair_done = nullptr;
    int repack_overlap = 1;         // FCVM_REPACK_OVERLAP=0: the copy into the window storage on the main stream
    hipStream_t repack_stream = nullptr;
    hipEvent_t repack_fork = nullptr, repack_done = nullptr;
    bool repack_pending = false;
    bool wsym_use = false;      // the matrix being copied is an SPD stiffness matrix: also into the symmetric form
    bool wsym_lo = true;        // ... and its 16-bit copy (FCVM_WIN_SYM_LO=0: the smoother passes stay on the full form)
    // round 5: the fixed-point sums of the symmetric form lose relative precision as max / min of the nodal diagonal entries (a graded
    // mesh): measured at every stiffness assembly (k_diag_grading), beyond grading_max the context runs the full window form
    bool wsym_graded = false;   // the symmetric form was declined for this mesh because of its grading
    double grading_max = 1.0e4; // FCVM_SYM_GRADING_MAX
    double grading[2] = {0.0, 0.0};   // largest / smallest nodal maximum of the free diagonal entries at the last assembly
    double *d_grading = nullptr;
    int32_t *d_slotof = nullptr;   // [nn] sliced-ELL slot of a node, kept from fcvm_set_mesh's uploads until the symmetric form is set up
    int asm_lpr = 4;            // lanes per row of the assembly into the symmetric window form (FCVM_ASM_LPR = 4, 8 or 16) ...
    int asm_lpr_long = 8;       // ... and in its long-row classes (FCVM_ASM_LPR_LONG), kept length above asm_lpr_from (FCVM_ASM_LPR_FROM)
    int asm_lpr_from = 28;
    int asm_nt = 1;             // FCVM_ASM_NT=0: plain instead of nontemporal stores when the assembly writes the symmetric window form
    int asm_sym = 1;            // FCVM_ASSEMBLE_SYM=0: stiffness matrices go through the sliced-ELL array and the copy kernel (rounds 2-3)
    bool sell_valid = false;    // d_vals holds the matrix of the last assembly (false after an assembly straight into the symmetric form)
    bool wsym_pending = false;  // the symmetric form's host build may still be running; its device half is done by wsym_finish
    bool sym_pair = true;       // two right-hand sides share one pass over the symmetric form (FCVM_SYM_PAIR=0: the one-vector kernels twice)
    bool mg_patch = false;         // FCVM_MG_PATCH=1: vertex-patch smoother on the fine level in the place of the block-Jacobi step (needs cycle form 2 on the symmetric window
                                   // form).  Off: on the 1 M plate it brings 46 -> 44 PCG iterations at 1.13 -> 1.68 ms per iteration -- the aggregate-based coarse solve, not the
                                   // fine smoother, is what bounds the cycle there (profiles/r05_patch_smoother.txt)
    PatchDev patch;
    std::vector<uint16_t> h_posa, h_posb;
    std::vector<int64_t> h_mptr;
    int32_t *d_l1_sb = nullptr, *d_l1_si = nullptr;   // multi-GPU layer: level-1 slices with a row of a shared node / the others (fcvm_mg_l1_set_interface)
    int64_t n_l1_sb = 0, n_l1_si = 0;
    int32_t *d_l1_nb = nullptr, *d_l1_ni = nullptr;   // ... and the shared level-1 nodes themselves / the others (fcvm_mg_restrict_part)
    int64_t n_l1_nb = 0, n_l1_ni = 0;
    bool mg_l1_x32 = true;         // FCVM_L1_X32=0: the level-1 passes gather from the FP64 direction vector as in rounds 3-4 (bitwise the same numbers)
    float *d_d1_32 = nullptr, *d_t1_32 = nullptr;   // FP32 twins of d_d1 / d_t1, one 16-byte record per node
    bool mg_fuse_prolong = false;  // FCVM_MG_FUSE_PROLONG=1: x + P e1 formed while the post-smoothing pass stages x instead of by k_prolong_add
                                   // (measured: 212 us against 177 + 29 -- every window node is staged by 2.9 groups -- so off)
    int spmv_win = 1, win_v2 = 1;   // win_v2: value components in 16-byte pairs (FCVM_WIN_V2=0: nine 8-byte runs)
    WinDev win;                     // the fine matrix.  (The level-1 matrix of the multigrid cycle was tried on the same format with
    WinSymDev wsym;     // the PCG's FP64 SpMV on the symmetric form (FCVM_WIN_SYM=0: off)
    std::thread sym_thread;   // host build of the symmetric form, beside the upload of the full form (joined by wsym_upload)
                                    // FP32 values and the smoother epilogue: 31.0 against 28.3 us per launch -- 686 workgroups are too
                                    // few and the staging barrier adds latency; it stays on the sliced-ELL kernel, profiles/r02v_*.)
    int spmv_variant = 5;  // 5 (shipped): nontemporal matrix loads, four pairs in flight, 8 + 16 byte x gathers; 4: three 8-byte gathers; 0: plain loads, one pair per iteration
    double last_spmv_timed = 0;
    // host copies needed to build the coarse level
    std::vector<int32_t> h_el;
    std::unique_ptr<TopoHost> topo_host;   // host half of the level-1 topology, built on topo_thread since fcvm_set_mesh (fcvm_mg.inc)
    std::thread topo_thread;
    std::vector<double> h_coord;
    std::vector<uint8_t> h_fixmask;
    std::vector<int32_t> h_c2f;   // level-1 node -> fine node
    std::vector<int32_t> h_morton;   // Morton order of the nodes (fcvm_set_mesh -> setup_window_format), released afterwards
    // p-multigrid preconditioner (mode 1): level 1 = linear tets on the corner nodes
    int precond_mode = 0, mg_k0 = 1, mg_k1 = 60;
    double mg_alpha0 = 4.0, mg_alpha1 = 1000.0, mg_lmax0 = 0, mg_lmax1 = 0;
    bool mg_topo = false, mg_lmax_stale = true, fixmask1_valid = false;
    int64_t nc = 0, nb_spmv1 = 0;
    Pattern pat1;
    int32_t *d_el4 = nullptr, *d_cols1 = nullptr, *d_rownode1 = nullptr, *d_c2f = nullptr, *d_mids = nullptr, *d_pa = nullptr,
            *d_pb = nullptr;
    int64_t *d_sptr1 = nullptr, *d_mptr = nullptr;
    uint32_t *d_emap4 = nullptr;
    uint8_t *d_fixmask1 = nullptr;
    double *d_coord1 = nullptr, *d_vals1 = nullptr, *d_dblk1 = nullptr, *d_dinv1 = nullptr;
    double *d_mr = nullptr, *d_md = nullptr, *d_mt = nullptr, *d_r1 = nullptr, *d_x1 = nullptr, *d_d1 = nullptr, *d_t1 = nullptr,
           *d_rr1 = nullptr;
    int64_t mg_spmv0 = 0, mg_spmv1 = 0;
    // level-0 correction: additive (x = w Dinv b + P M1^-1 P^T b: no fine SpMV inside the preconditioner; measured
    // fastest on the benchmark cube) or multiplicative (Chebyshev pre/post smoothing, FCVM_MG_ADDITIVE=0)
    int mg_additive = 2;   // 1: additive; 0: multiplicative, FP64 smoother SpMVs; 2: multiplicative V(1,1), smoother SpMVs on a 16-bit fixed-point copy
    int mg_fp32 = 1;           // coarse levels: level-1 matrix and aggregate inverse stored in FP32 (FP64 vectors / accumulation)
    float *d_vals1f = nullptr, *d_A2invf = nullptr;
    double *d_symv_part = nullptr;   // [nt][nt][128] partial results of k_symv_tiles (null: k_gemv_f32x4 applies the inverse)
    int mg_l1_16 = 1;              // level-1 SpMVs of the cycle on a 16-bit block-scaled copy of the level-1 matrix (FCVM_MG_L1_16=0: FP32)
    int4v *d_l1A = nullptr, *d_l1B = nullptr, *d_l1C = nullptr;
    int64_t *d_rp16 = nullptr;
    double *d_l1_scale = nullptr;  // [1024 partial maxima | exponent, unit value]
    float *d_dinv1_6f = nullptr;   // the same for level 1 (SpMV epilogue of the Chebyshev smoother)
    float *d_dinv6f = nullptr;     // FP32 copy of the six distinct entries of every fine inverse diagonal block (PCG update kernel)
    bool l2_has_inverse = false;
    double mg_add_w = 1.0;
    const double *mg_done = nullptr;  // device flag that turns the V-cycle kernels into no-ops once PCG has converged
    // level 2: aggregates + dense inverse (set from the host binding)
    bool l2 = false;
    L2sDev l2s;                    // sparse aggregate level + dense super-aggregate level (fcvm_l2s.inc), large meshes
    int64_t mg_max_dense = 3000;   // most aggregates a dense inverse is built for (FCVM_MG_MAX_AGGREGATES)
    int64_t mg_spmv2 = 0;
    int64_t nagg = 0, n2 = 0;
    int mg_k1s = 2;
    double mg_alpha1s = 8.0;
    int32_t *d_agg = nullptr, *d_anodes = nullptr;
    int64_t *d_aoff = nullptr, *d_aptr = nullptr;
    double *d_Z = nullptr, *d_A2inv = nullptr, *d_r2 = nullptr, *d_e2 = nullptr;
    // multi-GPU aggregate level: this rank's blocks of Z^T A1 Z between fcvm_mg_l2_galerkin and fcvm_mg_l2_galerkin_get
    std::vector<int64_t> gl_sptr;
    std::vector<int32_t> gl_slot;
    std::vector<double> gl_B;
    std::vector<hipEvent_t> evpool;
    double *h_pcgS = nullptr;      // pinned copies of the PCG scalars, one per queued chunk (fcvm_mg.inc)
    double *h_pcgS2 = nullptr;     // the same for the two states of fcvm_solve2
    MgAlt alt;                     // second PCG state (fcvm_solve2), allocated on first use
    int64_t mg_last_iters = 0, mg_prev_iters = 0;   // iterations of the two previous multigrid-PCG solves: they size the first chunk of the next one
};

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// the x gathers / LDS staging read a node's three values as 8 + 16 bytes: vector bases must sit on 16-byte boundaries
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
// LDS a workgroup can get on this device with the per-kernel opt-in (160 KiB on gfx950); the runtime reports it under
// different attributes depending on its version, so the largest answer counts, and never less than the 64 KiB every CDNA part has
static int device_lds_cap(int device) {
    int cap = 64 * 1024, v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device) == hipSuccess) cap = std::max(cap, v);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess) cap = std::max(cap, v);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess) cap = std::max(cap, v);
    (void)hipGetLastError();
    return cap;
}
template <class T>
static int dalloc(T **p, int64_t n) {
    // FCVM_POISON=1 (debugging): every fresh allocation filled with 0xFF bytes (NaN as a double, -1 as an index) -- a read of memory nothing
    // wrote, which a fresh process hides behind zero pages, shows at once; floating-point arrays get 512 poisoned bytes BEHIND them as well
    // (the read behind a vector that k_gemv_f32x4 multiplied with zero padding, round 5, was invisible to an exactly sized poisoned array)
    static const bool poison = getenv("FCVM_POISON") && atoi(getenv("FCVM_POISON")) != 0;
    const size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T), tail = (poison && std::is_floating_point<T>::value) ? 512 : 0;
    HCHK(hipMalloc((void **)p, bytes + tail));
    if (poison) {      // (hipMemset runs on the null stream and does not wait for the host: without the synchronisation it can land on top of what a
                       // context's own non-blocking stream writes into the array next)
        HCHK(hipMemset((void *)*p, 0xFF, bytes + tail));
        HCHK(hipDeviceSynchronize());
    }
    return 0;
}
template <class T>
static void dfree(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}
// FCVM_SETUP_TIMING=1: wall time of the stages of the one-time set-up (hierarchy, aggregate level), each closed by a stream
// synchronisation -- off in ordinary runs
struct SetupLap {
    bool on;
    hipStream_t s;
    std::chrono::steady_clock::time_point t;
    SetupLap(hipStream_t st) : on(getenv("FCVM_SETUP_TIMING") != nullptr), s(st), t(std::chrono::steady_clock::now()) {}
    void operator()(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm set-up: %-44s %.4f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
static MatConst matconst(const fcvm_ctx *c, double Et_E) {
    MatConst m;
    // fcVM.py:570-582 hooke and fcVM.py:2229-2234
    const double dm = c->E * (1.0 - c->nu) / (1.0 + c->nu) / (1.0 - 2.0 * c->nu);
    m.D11 = dm;
    m.D12 = c->nu / (1.0 - c->nu) * dm;
    m.D44 = 0.5 * (1.0 - 2.0 * c->nu) / (1.0 - c->nu) * dm;
    m.G = c->E / 2.0 / (1 + c->nu);
    if (Et_E > 0.95) Et_E = 0.95;
    m.Et = Et_E * c->E;
    m.H = m.Et / (1.0 - Et_E);
    m.rho = c->rho;
    return m;
}

struct MatView {
    int64_t nslices, nb;
    const int64_t *sptr;
    const int32_t *cols;
    const double *vals;
    const int32_t *rownode;
    const float *vals32 = nullptr;  // optional FP32 copy of the values (coarse multigrid level)
    // optional 16-bit block-scaled copy (level 1, k_spmv_l1_i16): records per block pair, see fcvm_spmv_win.hip.h
    const int4v *A16 = nullptr, *B16 = nullptr, *C16 = nullptr;
    const int64_t *rp16 = nullptr;
    const double *scale16 = nullptr;
};
static MatView fine_view(const fcvm_ctx *c) { return {c->pat.nslices, c->nb_spmv, c->d_sptr, c->d_cols, c->d_vals, c->d_rownode}; }
static MatView coarse_view(const fcvm_ctx *c) {
    MatView m{c->pat1.nslices, c->nb_spmv1, c->d_sptr1, c->d_cols1, c->d_vals1, c->d_rownode1, c->mg_fp32 ? c->d_vals1f : nullptr};
    if (c->mg_l1_16 && c->d_l1A) { m.A16 = c->d_l1A; m.B16 = c->d_l1B; m.C16 = c->d_l1C; m.rp16 = c->d_rp16; m.scale16 = c->d_l1_scale + 1024; }
    return m;
}

template <bool DOT, int WAVES, bool V2>
static void launch_spmv_win_t(fcvm_ctx *c, const WinDev &w, const double *x, double *y, const double *done) {
    hipLaunchKernelGGL((k_spmv_win<DOT, WAVES, V2>), dim3((unsigned)w.pat.ngroups), dim3(64 * WAVES), w.lds, c->stream, w.pat.ngroups,
                       w.pat.wmax, w.gptr, w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.own, w.cols, (const double *)w.vals, x,
                       y, DOT ? c->d_part_spmv : (double *)nullptr, done);
}

#define FCVM_SPMV_LAUNCH(DOT_, NT_, UN_, V2_, EPI_, vals_, epi_)                                                               \
    hipLaunchKernelGGL((k_spmv<DOT_, NT_, UN_, V2_, EPI_, 1>), grid, block, 0, c->stream, m.nslices, m.nb, m.sptr, m.cols, \
                       (const V2_ *)(vals_), m.rownode, x, y, part, done, epi_)

template <bool DOT>
static void launch_spmv_t(fcvm_ctx *c, const MatView &m, const double *x, double *y, double *part, const double *done) {
    dim3 grid((unsigned)m.nb), block(64 * c->spmv_spw);
    if (m.vals32) {  // FP32-stored matrix, FP64 vectors and accumulation
        FCVM_SPMV_LAUNCH(DOT, false, 4, float2, 0, m.vals32, ChebEpi{});
        return;
    }
    if (c->spmv_variant == 0) FCVM_SPMV_LAUNCH(DOT, false, 1, double2, 0, m.vals, ChebEpi{});
    else if (c->spmv_variant == 4)   // for comparison: three 8-byte gathers per node instead of 8 + 16
        hipLaunchKernelGGL((k_spmv<DOT, true, 4, double2, 0, 0>), grid, block, 0, c->stream, m.nslices, m.nb, m.sptr, m.cols,
                           (const double2 *)m.vals, m.rownode, x, y, part, done, ChebEpi{});
    else FCVM_SPMV_LAUNCH(DOT, true, 4, double2, 0, m.vals, ChebEpi{});
}
// SpMV t = A d fused with a Chebyshev smoother step on the same rows (see ChebEpi); nothing is written but epi's vectors
// FP32 twin of a level-1 direction vector of this context (null: none -- a caller's own vector)
static inline float *l1_twin(const fcvm_ctx *c, const double *v) {
    if (!c->mg_l1_x32 || !v) return nullptr;
    return v == c->d_d1 ? c->d_d1_32 : (v == c->d_t1 ? c->d_t1_32 : nullptr);
}
static void launch_spmv_cheb(fcvm_ctx *c, const MatView &m, const double *x, const double *done, const ChebEpi &epi_in) {
    ChebEpi epi = epi_in;
    if (m.A16 && m.A16 == c->d_l1A) {
        epi.dout32 = epi.update_d ? l1_twin(c, epi.dout) : nullptr;
        if (const float *x32 = l1_twin(c, x)) {      // round 5: the gathers read the FP32 twin the producer of x wrote beside it
            hipLaunchKernelGGL((k_spmv_l1_i16<1, false, true>), dim3((unsigned)m.nslices), dim3(64), 0, c->stream, m.nslices, m.rp16, m.A16, m.B16, m.C16,
                               m.scale16, m.rownode, x, (double *)nullptr, done, epi, (const f4v *)x32);
            return;
        }
    }
    if (m.A16) {   // level 1 on its 16-bit copy: 24 instead of 40 bytes per block
        // plain loads by default: the level-1 matrix (tens of MB) is read six times per V-cycle and stays in the memory-side cache
        // between those passes (1 M plate: 24.1 -> 21.2 us per pass, gpurun_out/r03ag); FCVM_L1_NT=1: nontemporal loads
        const bool nt = getenv("FCVM_L1_NT") && atoi(getenv("FCVM_L1_NT")) != 0;
        if (nt)
            hipLaunchKernelGGL((k_spmv_l1_i16<1, true>), dim3((unsigned)m.nslices), dim3(64), 0, c->stream, m.nslices, m.rp16, m.A16, m.B16, m.C16,
                               m.scale16, m.rownode, x, (double *)nullptr, done, epi);
        else
            hipLaunchKernelGGL((k_spmv_l1_i16<1, false>), dim3((unsigned)m.nslices), dim3(64), 0, c->stream, m.nslices, m.rp16, m.A16, m.B16, m.C16,
                               m.scale16, m.rownode, x, (double *)nullptr, done, epi);
        return;
    }
    dim3 grid((unsigned)m.nb), block(64 * c->spmv_spw);
    double *y = nullptr, *part = nullptr;
    if (m.vals32) FCVM_SPMV_LAUNCH(false, false, 4, float2, 1, m.vals32, epi);
    else FCVM_SPMV_LAUNCH(false, true, 4, double2, 1, m.vals, epi);
}
// leading dimension of the FP32 copy of the aggregate-level inverse: rows start on 128-byte lines (32 floats), so the 512-byte row
// segments of k_symv_tiles touch four lines, not five (PMC: 350 MB per launch for a 286 MB triangle with rows padded to 16 bytes)
static inline int64_t a2_lda(int64_t n2) { return (n2 + 31) / 32 * 32; }
// e2 = A2^-1 r2 with the FP32 copy of the dense inverse: from its lower triangle alone where that pays (k_symv_tiles; the partial
// buffer is allocated with the inverse, fcvm_level2.inc / fcvm_mg_set_level2), otherwise row by row
static void launch_gemv_f32(fcvm_ctx *c, const double *r2, double *e2, const double *done) {
    const int64_t lda = a2_lda(c->n2);
    if (c->d_symv_part) {
        const int nt = (int)cdiv(c->n2, SYMV_T);
        hipLaunchKernelGGL(k_symv_tiles, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(256), 0, c->stream, nt, c->n2, lda, c->d_A2invf, r2,
                           c->d_symv_part, done);
        hipLaunchKernelGGL(k_symv_reduce, dim3((unsigned)nt), dim3(2 * SYMV_T), 0, c->stream, nt, c->n2, c->d_symv_part, e2, done);
    } else {
        hipLaunchKernelGGL(k_gemv_f32x4, dim3((unsigned)cdiv(c->n2, 16)), dim3(256), 0, c->stream, c->n2, c->n2, lda, c->d_A2invf, r2, e2, done);
    }
    static const double over = getenv("FCVM_MG_L2_SCALE") ? atof(getenv("FCVM_MG_L2_SCALE")) : 1.0;   // (experiment: over-correction of the unsmoothed aggregation)
    if (over != 1.0) hipLaunchKernelGGL(k_axpby, dim3((unsigned)cdiv(c->n2, 256)), dim3(256), 0, c->stream, c->n2, over, e2, 0.0, e2, e2);
}
// (re)allocates the partial buffer of k_symv_tiles for the current n2, or drops it: small levels (the row kernel is launch-bound
// there anyway) and FCVM_L2_SYMV=0
static int symv_prepare(fcvm_ctx *c) {
    dfree(c->d_symv_part);
    const bool off = getenv("FCVM_L2_SYMV") && atoi(getenv("FCVM_L2_SYMV")) == 0;
    if (off || c->n2 < 2048) return 0;
    const int64_t nt = cdiv(c->n2, SYMV_T);
    return dalloc(&c->d_symv_part, nt * nt * SYMV_T);
}
// part (multi-GPU layer): 0 = all groups; 1 = the groups with rows of interface nodes only; 2 = the others only.  Storage forms without the
// lists (no symmetric form, or no interface nodes set) do everything in part 1 and nothing in part 2: the same result, no overlap.
static void launch_spmv_win(fcvm_ctx *c, bool dot, const double *x, double *y, const double *done, int part = 0) {
    const WinDev &w = c->win;
    if (c->wsym.on && c->wsym.valid) {   // symmetric form: 0.70 x the blocks, transposed products through LDS in fixed point
        const WinSymDev &ws = c->wsym;
        SymVecs<1> V{{x}, {y}, {dot ? c->d_part_spmv : (double *)nullptr}, {done}};
        int64_t grid = w.pat.ngroups;
        if (part && ws.split) {
            V.glist = part == 1 ? ws.glist_b : ws.glist_i;
            grid = part == 1 ? ws.n_gb : ws.n_gi;
            if (grid == 0) return;
        } else if (part == 2) {
            return;
        }
        if (dot)
            hipLaunchKernelGGL((k_spmv_win_sym<true, WIN_WAVES, 1>), dim3((unsigned)grid), dim3(64 * WIN_WAVES), ws.lds, c->stream,
                               w.pat.ngroups, w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own,
                               ws.cols, ws.vals, w.lo_scale, V);
        else
            hipLaunchKernelGGL((k_spmv_win_sym<false, WIN_WAVES, 1>), dim3((unsigned)grid), dim3(64 * WIN_WAVES), ws.lds, c->stream,
                               w.pat.ngroups, w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own,
                               ws.cols, ws.vals, w.lo_scale, V);
        return;
    }
    if (part == 2) return;
    if (w.pat.waves == 4) {   // meshes whose 512-row windows do not fit the LDS: groups of 256 rows
        if (dot) launch_spmv_win_t<true, 4, true>(c, w, x, y, done);
        else launch_spmv_win_t<false, 4, true>(c, w, x, y, done);
    } else if (c->win_v2) {
        if (dot) launch_spmv_win_t<true, WIN_WAVES, true>(c, w, x, y, done);
        else launch_spmv_win_t<false, WIN_WAVES, true>(c, w, x, y, done);
    } else {
        if (dot) launch_spmv_win_t<true, WIN_WAVES, false>(c, w, x, y, done);
        else launch_spmv_win_t<false, WIN_WAVES, false>(c, w, x, y, done);
    }
}
// p.Ap partials the fine SpMV leaves: one per group of the window format, one per workgroup of the sliced-ELL kernel
static inline int64_t spmv_partials(const fcvm_ctx *c) { return c->win.on ? c->win.pat.ngroups : c->nb_spmv; }
static void launch_spmv(fcvm_ctx *c, bool dot, const double *x, double *y, const double *done, int part = 0) {
    if (c->win.on) {   // window format: x through LDS (the p.Ap partials beyond the group count stay zero)
        launch_spmv_win(c, dot, x, y, done, part);
        return;
    }
    if (part == 2) return;
    if (dot) launch_spmv_t<true>(c, fine_view(c), x, y, c->d_part_spmv, done);
    else launch_spmv_t<false>(c, fine_view(c), x, y, nullptr, done);
}

// smoother passes of the multiplicative cycle on the 16-bit copy (EPI 1: out = r - K~ x; EPI 2: out = x + c0 Dinv (r - K~ x) + partials)
template <int EPI>
static void launch_spmv_win_lo(fcvm_ctx *c, const double *x, const double *r, double *out, double c0, const double *done, double *part = nullptr,
                               const double *fuse_e1 = nullptr, int gpart = 0, const double *rw = nullptr) {
    const WinDev &w = c->win;
    if (!part) part = w.part_lo;
    if (c->wsym.on && c->wsym.lo_valid) {      // symmetric form of the 16-bit copy
        const WinSymDev &ws = c->wsym;
        const ProlongFuse pf{c->d_pa, c->d_pb, c->d_fixmask, fuse_e1};      // (fuse_e1: x + P e1 is formed while x is staged, EPI 2 only)
        LoSymVecs<1> V{{x}, {r}, {out}, {part}, {done}};
        V.rw = rw;
        int64_t grid = w.pat.ngroups;
        if (gpart && ws.split) {       // (gpart: the boundary-groups-first split of the multi-GPU layer, as launch_spmv_win's part)
            V.glist = gpart == 1 ? ws.glist_b : ws.glist_i;
            grid = gpart == 1 ? ws.n_gb : ws.n_gi;
            if (grid == 0) return;
        } else if (gpart == 2) {
            return;
        }
        hipLaunchKernelGGL((k_spmv_win_lo_sym<EPI, WIN_WAVES, 1>), dim3((unsigned)grid), dim3(64 * WIN_WAVES), ws.lds_lo, c->stream, w.pat.ngroups,
                           w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own, ws.lo8,
                           (const WinLoTail *)ws.lo2, w.lo_scale, V, c0, (const float *)c->d_dinv6f, pf);
        return;
    }
    if (gpart == 2) return;
    if (w.pat.waves == 4)
        hipLaunchKernelGGL((k_spmv_win_lo<EPI, 4>), dim3((unsigned)w.pat.ngroups), dim3(256), w.lds / 2, c->stream, w.pat.ngroups, w.pat.wmax, w.gptr,
                           w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.lo8, w.lo2, w.lo_scale, x, r, out, c0,
                           (const float *)c->d_dinv6f, part, done);
    else
        hipLaunchKernelGGL((k_spmv_win_lo<EPI, WIN_WAVES>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), w.lds / 2, c->stream, w.pat.ngroups,
                           w.pat.wmax, w.gptr, w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.lo8, w.lo2, w.lo_scale, x,
                           r, out, c0, (const float *)c->d_dinv6f, part, done);
}
// two right-hand sides: one pass over the matrix for both (k_spmv_win2 / k_spmv_win_lo2); LDS holds both windows
static void launch_spmv_win2(fcvm_ctx *c, const double *xA, const double *xB, double *yA, double *yB, double *partA, double *partB,
                             const double *doneA, const double *doneB) {
    const WinDev &w = c->win;
    if (c->wsym.on && c->wsym.valid) {
        // symmetric form, both vectors in one pass (round 4): per vector bitwise the one-vector kernel's result
        const WinSymDev &ws = c->wsym;
        if (c->sym_pair) {
            const SymVecs<2> V{{xA, xB}, {yA, yB}, {partA, partB}, {doneA, doneB}};
            hipLaunchKernelGGL((k_spmv_win_sym<true, WIN_WAVES, 2>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), 2 * ws.lds, c->stream,
                               w.pat.ngroups, w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own,
                               ws.cols, ws.vals, w.lo_scale, V);
            return;
        }
        const double *xs2[2] = {xA, xB}, *dn[2] = {doneA, doneB};       // FCVM_SYM_PAIR=0: the one-vector kernel twice
        double *ys2[2] = {yA, yB}, *pt[2] = {partA, partB};
        for (int k = 0; k < 2; k++) {
            const SymVecs<1> V{{xs2[k]}, {ys2[k]}, {pt[k]}, {dn[k]}};
            hipLaunchKernelGGL((k_spmv_win_sym<true, WIN_WAVES, 1>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), ws.lds, c->stream,
                               w.pat.ngroups, w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own,
                               ws.cols, ws.vals, w.lo_scale, V);
        }
        return;
    }
    if (w.pat.waves == 4)
        hipLaunchKernelGGL((k_spmv_win2<true, 4>), dim3((unsigned)w.pat.ngroups), dim3(256), 2 * w.lds, c->stream, w.pat.ngroups, w.pat.wmax, w.gptr,
                           w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.own, w.cols, (const double *)w.vals, xA, xB, yA, yB, partA,
                           partB, doneA, doneB);
    else
        hipLaunchKernelGGL((k_spmv_win2<true, WIN_WAVES>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), 2 * w.lds, c->stream, w.pat.ngroups,
                           w.pat.wmax, w.gptr, w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.own, w.cols, (const double *)w.vals, xA, xB,
                           yA, yB, partA, partB, doneA, doneB);
}
template <int EPI>
static void launch_spmv_win_lo2(fcvm_ctx *c, const double *xA, const double *xB, const double *rA, const double *rB, double *outA, double *outB,
                                double c0, double *partA, double *partB, const double *doneA, const double *doneB) {
    const WinDev &w = c->win;
    if (c->wsym.on && c->wsym.lo_valid) {
        if (c->sym_pair) {      // both vectors in one pass over the 16-bit copy (round 4), per vector bitwise the one-vector kernel's result
            const WinSymDev &ws = c->wsym;
            const ProlongFuse pf{c->d_pa, c->d_pb, c->d_fixmask, nullptr};
            const LoSymVecs<2> V{{xA, xB}, {rA, rB}, {outA, outB}, {partA, partB}, {doneA, doneB}};
            hipLaunchKernelGGL((k_spmv_win_lo_sym<EPI, WIN_WAVES, 2>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), 2 * ws.lds_lo, c->stream,
                               w.pat.ngroups, w.pat.wmax, ws.pat.xmax, w.gptr, ws.gextra, w.wlist, ws.woff, ws.rowlen, ws.rownode, ws.seg, ws.own,
                               ws.lo8, (const WinLoTail *)ws.lo2, w.lo_scale, V, c0, (const float *)c->d_dinv6f, pf);
            return;
        }
        launch_spmv_win_lo<EPI>(c, xA, rA, outA, c0, doneA, partA);
        launch_spmv_win_lo<EPI>(c, xB, rB, outB, c0, doneB, partB);
        return;
    }
    if (w.pat.waves == 4)
        hipLaunchKernelGGL((k_spmv_win_lo2<EPI, 4>), dim3((unsigned)w.pat.ngroups), dim3(256), w.lds, c->stream, w.pat.ngroups, w.pat.wmax, w.gptr,
                           w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.lo8, w.lo2, w.lo_scale, xA, xB, rA, rB, outA, outB, c0,
                           (const float *)c->d_dinv6f, partA, partB, doneA, doneB);
    else
        hipLaunchKernelGGL((k_spmv_win_lo2<EPI, WIN_WAVES>), dim3((unsigned)w.pat.ngroups), dim3(64 * WIN_WAVES), w.lds, c->stream, w.pat.ngroups,
                           w.pat.wmax, w.gptr, w.gextra, w.wlist, w.woff, w.rowlen, w.rownode, w.seg, w.lo8, w.lo2, w.lo_scale, xA, xB, rA, rB,
                           outA, outB, c0, (const float *)c->d_dinv6f, partA, partB, doneA, doneB);
}

// the post-smoothing pass forms x + P e1 itself (k_spmv_win_lo_sym): the separate prolongation kernel is not needed
static inline bool mg_fuse_prolong(const fcvm_ctx *c) { return c->wsym.on && c->wsym.lo_valid && c->mg_fuse_prolong; }
// the multiplicative cycle with the 16-bit smoother matrix needs the window format and its 16-bit copy
static inline bool mg_use_lo(const fcvm_ctx *c) { return c->mg_additive == 2 && c->win.on && (c->win.lo_valid || (c->wsym.on && c->wsym.lo_valid)); }
// the form that runs: 2 without the window format (tiny meshes, windows beyond the LDS) becomes the additive form -- the FP64
// multiplicative form would stream the sliced-ELL matrix three times per iteration there
static inline bool mg_use_patch(const fcvm_ctx *c) { return mg_use_lo(c) && c->mg_patch && c->patch.on && c->patch.valid && c->wsym.on && c->wsym.lo_valid; }
static inline int mg_form(const fcvm_ctx *c) { return c->mg_additive == 2 ? (mg_use_lo(c) ? 2 : 1) : c->mg_additive; }

static int dot_common(fcvm_ctx *c, int64_t n, const double *w, const double *x, const double *y, double *result_dev) {
    NEED(c && c->d_part_misc, "vector reductions need a mesh (work buffers)");
    int64_t nb = std::min<int64_t>(std::max<int64_t>(cdiv(n, 256), 1), 2048);
    hipLaunchKernelGGL(k_dot_partial, dim3((unsigned)nb), dim3(256), 0, c->stream, n, w, x, y, c->d_part_misc);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, c->d_part_misc, nb, result_dev);
    HCHK(hipGetLastError());
    return 0;
}
static void alt_free(fcvm_ctx *c);
static void patch_free(fcvm_ctx *c);
static int win_join(fcvm_ctx *c);
static int patch_build_tables(fcvm_ctx *c);
static int patch_refresh(fcvm_ctx *c);
static void patch_apply(fcvm_ctx *c, const double *x, double *y);
static void patch_smooth_from_zero(fcvm_ctx *c, const double *x, double w, double *out);
static void patch_smooth_post(fcvm_ctx *c, const double *t, double w, const double *x, const double *r, double *out, double *part, int64_t nb);
static double mg_patch_weight(const fcvm_ctx *c);
#include "fcvm_mg.inc"
#include "fcvm_patch.inc"

static void wsym_free(WinSymDev &w) {
    dfree(w.woff); dfree(w.rownode); dfree(w.srcslot); dfree(w.gextra); dfree(w.seg); dfree(w.cols); dfree(w.rowlen); dfree(w.own);
    dfree(w.jell); dfree(w.vals); dfree(w.lo8); dfree(w.lo2);
    dfree(w.emap); dfree(w.diagk); dfree(w.jfirst); dfree(w.rowoff); dfree(w.nextseg); dfree(w.asm_items); dfree(w.part_vol);
    dfree(w.glist_b); dfree(w.glist_i);
    w.n_gb = w.n_gi = 0;
    w.split = false;
    w.asm_ok = false;
    w.nitems = 0;
    w.on = w.valid = w.lo_valid = false;
    w.pat = WinSym();
    w.lds = 0;
}
static void win_free(WinDev &w) {
    dfree(w.gptr); dfree(w.woff); dfree(w.wlist); dfree(w.rownode); dfree(w.srcslot); dfree(w.gextra); dfree(w.seg); dfree(w.rowlen);
    dfree(w.own); dfree(w.cols); dfree(w.jstart);
    if (w.vals) { (void)hipFree(w.vals); w.vals = nullptr; }
    dfree(w.lo8); dfree(w.lo2); dfree(w.lo_scale); dfree(w.part_lo);
    w.lo_valid = false;
    w.on = false;
    w.pat = WinPattern();
    w.lds = 0;
}

// device copy of a built WinPattern
static int win_upload(fcvm_ctx *c, WinDev &d) {
    const int value_bytes = 8;
    const WinPattern &w = d.pat;
    const int64_t nrows = w.nwaves * SLICE;
    if (dalloc(&d.gptr, w.ngroups + 1) || dalloc(&d.woff, w.nwaves + 1) || dalloc(&d.wlist, (int64_t)w.wlist.size()) ||
        dalloc(&d.rownode, nrows) || dalloc(&d.srcslot, nrows) || dalloc(&d.rowlen, nrows) || dalloc(&d.own, nrows) ||
        dalloc(&d.cols, w.nblocks) || dalloc(&d.gextra, w.ngroups) || dalloc(&d.seg, nrows) || dalloc(&d.jstart, nrows))
        return 1;
    // (the FP64 values of the full form are allocated when first needed, win_values_ensure: with the symmetric form in charge
    // only the buckling operators use them)
    hipStream_t s = c->stream;
    HCHK(hipMemcpyAsync(d.gptr, w.gptr.data(), sizeof(int64_t) * (w.ngroups + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.woff, w.woff.data(), sizeof(int64_t) * (w.nwaves + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.wlist, w.wlist.data(), sizeof(int32_t) * w.wlist.size(), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.rownode, w.rownode.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.srcslot, w.srcslot.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.rowlen, w.rowlen.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.own, w.own.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.jstart, w.jstart.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.seg, w.seg.data(), sizeof(uint32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.gextra, w.gextra.data(), sizeof(int32_t) * w.ngroups, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.cols, w.cols.data(), sizeof(uint16_t) * w.nblocks, hipMemcpyHostToDevice, s));
    HCHK(hipStreamSynchronize(s));
    (void)value_bytes;
    d.lds = sizeof(double) * 3 * (size_t)(w.wmax + w.xmax);
    d.on = true;
    return 0;
}

// FP64 values of the full window form, allocated at first use
static int win_values_ensure(fcvm_ctx *c) {
    WinDev &d = c->win;
    if (d.vals) return 0;
    if (hipMalloc(&d.vals, sizeof(double) * 9 * (size_t)d.pat.nblocks) != hipSuccess) return fail("out of device memory (window format values)");
    HCHK(hipMemsetAsync(d.vals, 0, sizeof(double) * 9 * (size_t)d.pat.nblocks, c->stream));
    return 0;
}
template <class K>
static int win_allow_lds(K kernel, size_t bytes) {
    HCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
// kept-length classes of the assembly into the symmetric form (blocks of the longest row of a work item's 16 lanes):
// 22 / 30 / 39 / 53 / 68 / 80 / 160 KiB of LDS per item = 7 / 5 / 4 / 3 / 2 / 2 / 1 items per CU
static const int64_t ASM_SYM_LIMIT[WinSymDev::ASM_BUCKETS] = {14, 20, 28, 40, 53, 64, 133};
// tables and work items of k_assemble_lds<.., SYM> (after wsym_upload; needs c->d_slotof, c->d_emap, c->d_sptr of fcvm_set_mesh).
// Not available (asm_ok stays false, the sliced-ELL path + copy kernel remain): rows beyond the largest class, no LDS for it.
struct StageLap {      // FCVM_PATTERN_TIMING=1: wall time of the stages of a set-up function on stderr
    const char *who;
    bool on;
    std::chrono::steady_clock::time_point t;
    explicit StageLap(const char *w) : who(w), on(getenv("FCVM_PATTERN_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void operator()(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm %s: %-28s %.3f s\n", who, what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
static int wsym_asm_setup(fcvm_ctx *c) {
    WinSymDev &d = c->wsym;
    WinSym &w = d.pat;
    d.asm_ok = false;
    StageLap lap("assembly into the symmetric form");
    const int64_t nrows = w.nwaves * SLICE;
    auto release_host = [&]() {
        w.kmap.clear(); w.kmap.shrink_to_fit();
        w.rowoff.clear(); w.rowoff.shrink_to_fit();
    };
    if (!c->asm_sym || !c->assemble_lds || !c->d_slotof || w.kmap.empty()) { release_host(); return 0; }
    // A work item = 16 rows of ONE group, taken from the group's rows sorted by their kept length (longest first): rows of one
    // length class together (the LDS of an item is sized by its longest row), every quad busy.  Their lanes in the window layout
    // are near each other but not consecutive -- the lanes of a group are sorted by SEGMENT length, later segments of cut rows
    // in between -- so the write-out's 16-byte stores of a block row land in one or two 1 KiB stretches, which the L2 merges
    // with the neighbouring items' (first version: item = 16 consecutive lanes; 38 % of the items then held a long row, most
    // of them with half their quads idle on later segments: 12.2 ms per assembly against 10.0 of the copy route).
    std::vector<int32_t> items[WinSymDev::ASM_BUCKETS];
    bool fits = true;
    {
        const int64_t R = (int64_t)w.waves * SLICE;
        for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++)      // rows per work item: the long-row classes get more lanes per row
            d.asm_rows[b] = 64 / (ASM_SYM_LIMIT[b] > c->asm_lpr_from ? c->asm_lpr_long : c->asm_lpr);
        int nt = (int)std::thread::hardware_concurrency();
        nt = std::min(std::max(nt, 1), 32);
        std::vector<std::vector<int32_t>> part(nt * WinSymDev::ASM_BUCKETS);      // per thread (= contiguous range of groups) and class
        std::vector<char> bad(nt, 0);
        // Round 5 experiment (FCVM_ASM_ITEMS=octet; NOT the default: slower, see below): a work item = whole OCTETS of consecutive lanes of a wave -- the eight lanes whose 16-byte stores
        // fill one 128-byte line of every value plane of a block row, so a finished item leaves whole lines (rounds 3-4: rows picked by kept length from
        // all over the group, every store a lone 16 bytes of a line that seven other items complete some time later: 4.02 GB written for 2.83).  The
        // lanes of a group are sorted by segment length with the kinds clustered (build_window_sym), so an octet's rows are of one kept-length class
        // but for the seams; octets of later segments alone hold nothing to assemble.  Measured on the 1 M plate (profiles/r05_assembly_items.txt): whole
        // lines are written (WRITE_SIZE falls), but calc_gsm takes 6.45 / 10.34 ms (elastic / tangent) with the lane order by wave and 7.73 / 12.19 with the
        // order inside a length, against 5.76 / 9.28 for the items of rounds 3-4 -- seam octets put short rows into the long-row classes (one octet
        // = one item of eight lanes per row there), and the kernel is bound by its dependent chain per element visit, not by its writes.
        const bool octets = getenv("FCVM_ASM_ITEMS") && std::string(getenv("FCVM_ASM_ITEMS")) == "octet";
        if (octets)
        parallel_for(w.ngroups, nt, [&](int64_t gb, int64_t ge, int t) {
            std::vector<int32_t> oct[WinSymDev::ASM_BUCKETS];      // per class: the group's octets, eight lane positions each (-1: nothing to assemble)
            for (int64_t g = gb; g < ge; g++) {
                for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) oct[b].clear();
                for (int64_t at0 = g * R; at0 < (g + 1) * R; at0 += 8) {
                    int m = 0;
                    for (int l = 0; l < 8; l++) {
                        const int64_t at = at0 + l;
                        if (w.rownode[at] >= 0 && (w.seg[at] & 255u) > 0) m = std::max<int>(m, w.rowtot[at]);
                    }
                    if (m == 0) continue;
                    if (m > ASM_SYM_LIMIT[WinSymDev::ASM_BUCKETS - 1]) { bad[t] = 1; return; }
                    int b = 0;
                    while (m > ASM_SYM_LIMIT[b]) b++;
                    for (int l = 0; l < 8; l++) {
                        const int64_t at = at0 + l;
                        oct[b].push_back(w.rownode[at] >= 0 && (w.seg[at] & 255u) > 0 ? (int32_t)at : -1);
                    }
                }
                for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) {
                    const int RW = d.asm_rows[b];
                    std::vector<int32_t> &dst = part[t * WinSymDev::ASM_BUCKETS + b];
                    const size_t n = oct[b].size();
                    for (size_t k = 0; k < n; k += RW) {    // (RW = 16: two octets per item, 8: one, 4: half an octet)
                        const size_t first = dst.size();
                        for (int l = 0; l < RW; l++) dst.push_back(k + l < n ? oct[b][k + l] : -1);
                        // the kernel reads the item's group from its FIRST row: a row there (the order of an item's rows is free); an item
                        // without any row (half an octet of later segments) goes
                        size_t v = first;
                        while (v < dst.size() && dst[v] < 0) v++;
                        if (v == dst.size()) dst.resize(first);
                        else std::swap(dst[first], dst[v]);
                    }
                }
            }
        }, 16);
        else
        parallel_for(w.ngroups, nt, [&](int64_t gb, int64_t ge, int t) {
            std::vector<int32_t> rows;
            for (int64_t g = gb; g < ge; g++) {
                rows.clear();
                for (int64_t at = g * R; at < (g + 1) * R; at++)
                    if (w.rownode[at] >= 0 && (w.seg[at] & 255u) > 0) rows.push_back((int32_t)at);
                std::stable_sort(rows.begin(), rows.end(), [&](int32_t x, int32_t y) { return w.rowtot[x] > w.rowtot[y]; });
                for (size_t k = 0; k < rows.size();) {
                    const int m = w.rowtot[rows[k]];
                    if (m > ASM_SYM_LIMIT[WinSymDev::ASM_BUCKETS - 1]) { bad[t] = 1; return; }
                    int b = 0;
                    while (m > ASM_SYM_LIMIT[b]) b++;
                    const int RW = d.asm_rows[b];
                    std::vector<int32_t> &dst = part[t * WinSymDev::ASM_BUCKETS + b];
                    for (int l = 0; l < RW; l++) dst.push_back(k + l < rows.size() ? rows[k + l] : -1);
                    k += RW;
                }
            }
        }, 16);
        for (int t = 0; t < nt; t++) {      // (threads in order: the same item lists whatever the thread count)
            if (bad[t]) fits = false;
            for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) {
                const std::vector<int32_t> &src = part[t * WinSymDev::ASM_BUCKETS + b];
                items[b].insert(items[b].end(), src.begin(), src.end());
            }
        }
    }
    lap("work items (host)");
    const int lds_cap = device_lds_cap(c->device);
    for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++)
        if (!items[b].empty() && (int64_t)asm_lds_bytes(ASM_SYM_LIMIT[b], true, d.asm_rows[b]) > lds_cap) fits = false;
    if (!fits || w.nwaves >= (int64_t(1) << 25)) { release_host(); return 0; }
    std::vector<int32_t> all;      // per class: [item][rows of the class] lane (wave * 64 + lane) of each of the item's rows, -1: none
    d.nitems = 0;
    for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) {
        d.asm_first[b] = d.nitems;                     // first volume partial of the class
        d.asm_off[b] = (int64_t)all.size();            // first entry of the class in asm_items
        d.asm_count[b] = (int64_t)items[b].size() / d.asm_rows[b];
        d.nitems += d.asm_count[b];
        all.insert(all.end(), items[b].begin(), items[b].end());
    }
    d.maxl1 = w.maxl + 1;
    uint8_t *d_kmap = nullptr;
    int64_t *d_rowptr = nullptr;
    if (dalloc(&d.emap, c->ne * 100) || dalloc(&d.diagk, c->nn) || dalloc(&d.jfirst, nrows) || dalloc(&d.rowoff, (int64_t)w.rowoff.size()) ||
        dalloc(&d.nextseg, nrows) || dalloc(&d.asm_items, (int64_t)all.size()) || dalloc(&d.part_vol, d.nitems) || dalloc(&d_kmap, (int64_t)w.kmap.size()) ||
        dalloc(&d_rowptr, c->nn + 1)) {
        (void)hipGetLastError();
        dfree(d_kmap); dfree(d_rowptr);
        dfree(d.emap); dfree(d.diagk); dfree(d.jfirst); dfree(d.rowoff); dfree(d.nextseg); dfree(d.asm_items); dfree(d.part_vol);
        release_host();
        return 0;       // (no memory for the tables: the copy-kernel path stays)
    }
    lap("allocations");
    hipStream_t s = c->stream;
    HCHK(hipMemcpyAsync(d.diagk, w.diagk.data(), sizeof(uint8_t) * c->nn, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.jfirst, w.jfirst.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.rowoff, w.rowoff.data(), sizeof(uint16_t) * w.rowoff.size(), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.nextseg, w.nextseg.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.asm_items, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d_kmap, w.kmap.data(), sizeof(uint8_t) * w.kmap.size(), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d_rowptr, c->pat.rowptr.data(), sizeof(int64_t) * (c->nn + 1), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_build_emap_sym, dim3((unsigned)cdiv(c->ne * 10, 256)), dim3(256), 0, s, c->ne, c->d_el, c->d_slotof, c->d_sptr, d_rowptr,
                       c->d_emap, d_kmap, d.emap);
    HCHK(hipGetLastError());
    HCHK(hipStreamSynchronize(s));
    lap("uploads, element map kernel");
    dfree(d_kmap); dfree(d_rowptr);
    release_host();
    lap("release");
    static int asm_sym_lds_allowed_dev[64] = {0};
    int &allowed = asm_sym_lds_allowed_dev[c->device & 63];
    if (allowed == 0) allowed = 64 * 1024;
    for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) {
        const int bytes = (int)asm_lds_bytes(ASM_SYM_LIMIT[b], true, d.asm_rows[b]);
        if (bytes <= allowed || !d.asm_count[b]) continue;
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<true, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<true, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<false, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<true, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        allowed = bytes;
    }
    d.asm_ok = true;
    if (getenv("FCVM_VERBOSE")) {
        fprintf(stderr, "fcvm: assembly into the symmetric form: %lld work items, by kept length (x rows per item)", (long long)d.nitems);
        for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) fprintf(stderr, " <=%lld: %lld x %d", (long long)ASM_SYM_LIMIT[b], (long long)d.asm_count[b], d.asm_rows[b]);
        fprintf(stderr, "\n");
    }
    return 0;
}

// device copy of the symmetric form built by win_build_host (nothing to do when it was not built); used only where two
// workgroups of it fit the LDS of a CU
static int wsym_upload(fcvm_ctx *c) {
    StageLap lap("symmetric form (device)");
    if (c->sym_thread.joinable()) c->sym_thread.join();
    lap("wait for the host build");
    WinSymDev &d = c->wsym;
    const WinSym &w = d.pat;
    if (w.nblocks <= 0 || !c->win.on) { d.pat = WinSym(); return 0; }
    d.lds = sizeof(double) * 3 * (size_t)(c->win.pat.wmax + w.xmax) + sizeof(unsigned long long) * 3 * 64 * WIN_WAVES;
    static size_t lds_allowed_dev[64] = {0};
    if (2 * d.lds > (size_t)device_lds_cap(c->device)) {
        if (getenv("FCVM_VERBOSE")) fprintf(stderr, "fcvm: symmetric window form not used (%zu B of LDS per workgroup)\n", d.lds);
        d.pat = WinSym();
        return 0;
    }
    const int64_t nrows = w.nwaves * SLICE;
    if (dalloc(&d.woff, w.nwaves + 1) || dalloc(&d.rownode, nrows) || dalloc(&d.srcslot, nrows) || dalloc(&d.gextra, w.ngroups) ||
        dalloc(&d.seg, nrows) || dalloc(&d.cols, w.nblocks) || dalloc(&d.rowlen, nrows) || dalloc(&d.own, nrows) ||
        dalloc(&d.jell, w.nblocks) || dalloc(&d.vals, 9 * w.nblocks)) {
        // no memory for the second copy: the full window form stays in charge instead of failing the mesh
        (void)hipGetLastError();
        if (getenv("FCVM_VERBOSE")) fprintf(stderr, "fcvm: symmetric window form not used (out of device memory)\n");
        wsym_free(d);
        return 0;
    }
    hipStream_t s = c->stream;
    HCHK(hipMemcpyAsync(d.woff, w.woff.data(), sizeof(int64_t) * (w.nwaves + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.rownode, w.rownode.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.srcslot, w.srcslot.data(), sizeof(int32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.gextra, w.gextra.data(), sizeof(int32_t) * w.ngroups, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.seg, w.seg.data(), sizeof(uint32_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.cols, w.cols.data(), sizeof(uint32_t) * w.nblocks, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.rowlen, w.rowlen.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.own, w.own.data(), sizeof(uint16_t) * nrows, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(d.jell, w.jell.data(), sizeof(uint8_t) * w.nblocks, hipMemcpyHostToDevice, s));
    HCHK(hipMemsetAsync(d.vals, 0, sizeof(double) * 9 * w.nblocks, s));
    HCHK(hipStreamSynchronize(s));
    lap("allocations, uploads");
    size_t &lds_allowed = lds_allowed_dev[c->device & 63];
    lds_allowed = std::max(lds_allowed, d.lds);
    if (win_allow_lds(k_spmv_win_sym<true, WIN_WAVES, 1>, lds_allowed) || win_allow_lds(k_spmv_win_sym<false, WIN_WAVES, 1>, lds_allowed) ||
        win_allow_lds(k_spmv_win_sym<true, WIN_WAVES, 2>, 2 * lds_allowed))      // (two workgroups of the form fit a CU: twice its LDS does)
        return 1;
    d.lds = lds_allowed;
    // the 16-bit passes: FP32 window and segment results (even count: the 64-bit accumulators follow), accumulators
    d.lds_lo = sizeof(float) * (size_t)((3 * (c->win.pat.wmax + w.xmax) + 1) & ~(int64_t)1) + sizeof(unsigned long long) * 3 * 64 * WIN_WAVES;
    if (win_allow_lds(k_spmv_win_lo_sym<0, WIN_WAVES, 1>, d.lds) || win_allow_lds(k_spmv_win_lo_sym<1, WIN_WAVES, 1>, d.lds) ||
        win_allow_lds(k_spmv_win_lo_sym<2, WIN_WAVES, 1>, d.lds) || win_allow_lds(k_spmv_win_lo_sym<1, WIN_WAVES, 2>, 2 * d.lds) ||
        win_allow_lds(k_spmv_win_lo_sym<2, WIN_WAVES, 2>, 2 * d.lds))
        return 1;
    d.pat.cols.clear(); d.pat.cols.shrink_to_fit();
    d.pat.jell.clear(); d.pat.jell.shrink_to_fit();
    d.on = true;
    d.valid = false;
    if (wsym_asm_setup(c)) return 1;
    if (getenv("FCVM_VERBOSE"))
        fprintf(stderr, "fcvm: symmetric window form: %lld entries (%.3f of the full form), %zu B LDS\n", (long long)w.nblocks,
                (double)w.nblocks / (double)c->win.pat.nblocks, d.lds);
    return 0;
}

// Second storage of the fine matrix for the SpMV (WinPattern): built when every group's window fits the LDS twice per CU
// host part of the window format (pattern only, no GPU call): runs on a thread of its own beside the uploads of fcvm_set_mesh
struct WinBuild {
    bool built = false;
    std::string err;
};
static void win_build_host(fcvm_ctx *c, const double *nocoord, WinBuild *res) {
    if (!c->spmv_win || c->nn < 2048) return;
    int nt = (int)std::thread::hardware_concurrency();
    nt = std::min(std::max(nt, 1), 32);
    int64_t max_window = 3200;   // 2 x 24 B x (3200 + partial-result slots) < 160 KB: two workgroups per CU
    if (const char *v = getenv("FCVM_WIN_MAX")) max_window = std::max(64, atoi(v));   // (tests: force the smaller groups)
    WinPattern &w = c->win.pat;
    try {
        int split = WIN_SPLIT;
        if (const char *v = getenv("FCVM_WIN_SPLIT")) split = std::min(512, std::max(4, atoi(v)));
        const int32_t *mo = c->h_morton.size() == (size_t)c->nn ? c->h_morton.data() : nullptr;   // the order fcvm_set_mesh computed
        bool built = build_window_format(c->pat, nocoord, nt, max_window, w, WIN_WAVES, split, mo);
        if (!built) {   // smaller groups have smaller windows (at the price of more x traffic from the L2)
            const long long w8 = (long long)w.wmax;
            w = WinPattern();
            built = build_window_format(c->pat, nocoord, nt, max_window, w, 4, split, mo);
            if (getenv("FCVM_VERBOSE"))
                fprintf(stderr, "fcvm: window format: largest window of 512-row groups %lld nodes, of 256-row groups %lld\n", w8, (long long)w.wmax);
            if (built) c->win_v2 = 1;
        }
        if (!built) {
            if (getenv("FCVM_VERBOSE")) fprintf(stderr, "fcvm: window format not used (largest window %lld nodes)\n", (long long)w.wmax);
            w = WinPattern();
            return;
        }
        res->built = true;
        // symmetric form for the PCG's FP64 SpMV (groups of WIN_WAVES waves, paired value layout)
        c->wsym.pat = WinSym();
        const bool sym = !(getenv("FCVM_WIN_SYM") && atoi(getenv("FCVM_WIN_SYM")) == 0);
        if (sym && w.waves == WIN_WAVES && c->win_v2)
            c->sym_thread = std::thread([c, nt]() {
                try {
                    if (!build_window_sym(c->pat, c->win.pat, nt, c->wsym.pat)) c->wsym.pat = WinSym();
                } catch (const std::exception &) {      // (out of memory: the full form stays in charge)
                    c->wsym.pat = WinSym();
                }
            });
    } catch (const std::exception &ex) {
        res->err = std::string("fcvm_set_mesh (window format): ") + ex.what();
    }
}

static int setup_window_format_body(fcvm_ctx *c, const WinBuild &res);
// every return leaves no builder thread behind: the thread reads c->pat / c->win.pat, which the caller may rebuild or free next
// (a failed fcvm_set_mesh followed by another one on the same context)
static int setup_window_format(fcvm_ctx *c, const WinBuild &res) {
    const int rc = setup_window_format_body(c, res);
    if (rc || !c->wsym_pending) {
        if (c->sym_thread.joinable()) c->sym_thread.join();
        c->wsym_pending = false;
    }
    if (rc) { wsym_free(c->wsym); }
    return rc;
}
// The symmetric form's host build (0.05-0.09 s at 1 M elements) can only start when the full form's is done, so fcvm_set_mesh
// would sit waiting for it; it returns instead and the device half (upload, the assembly's element map and work items) happens
// here, at the first call that needs the form -- the caller's own set-up work (supports, load vectors) runs beside the thread.
static int wsym_finish(fcvm_ctx *c) {
    if (!c->wsym_pending) return 0;
    c->wsym_pending = false;
    HCHK(hipSetDevice(c->device));
    const int rc = wsym_upload(c);      // (joins the builder)
    if (c->sym_thread.joinable()) c->sym_thread.join();
    WinPattern &w = c->win.pat;
    // the host copies of the big tables are not needed again (the symmetric form's builder has read them)
    w.wlist.clear(); w.wlist.shrink_to_fit();
    w.cols.clear(); w.cols.shrink_to_fit();
    dfree(c->d_slotof);
    if (rc) wsym_free(c->wsym);
    return rc;
}
static int setup_window_format_body(fcvm_ctx *c, const WinBuild &res) {
    if (!res.err.empty()) return fail(res.err);
    if (!res.built) return 0;
    WinPattern &w = c->win.pat;
    if (win_upload(c, c->win)) return 1;
    // the window kernel leaves one p.Ap partial per GROUP: more than the sliced-ELL kernel's one per workgroup when several
    // slices share a workgroup (FCVM_SPMV_SPW > 1) or when cut rows add virtual rows
    c->nb_part_spmv = std::max<int64_t>(c->nb_spmv, w.ngroups);
    dfree(c->d_part_spmv);
    if (dalloc(&c->d_part_spmv, c->nb_part_spmv)) return 1;
    HCHK(hipMemsetAsync(c->d_part_spmv, 0, sizeof(double) * c->nb_part_spmv, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    // the limit is a property of the kernel (per device), not of this context: never lower what another context of the
    // process asked for on the same device, and refuse windows the device's LDS cannot hold (-> sliced-ELL kernel)
    static size_t lds_allowed_dev[64] = {0};
    const int lds_cap = device_lds_cap(c->device);
    if (c->win.lds > (size_t)lds_cap) {
        if (getenv("FCVM_VERBOSE")) fprintf(stderr, "fcvm: window format not used (%zu B of LDS needed, device offers %d)\n", c->win.lds, lds_cap);
        if (c->sym_thread.joinable()) c->sym_thread.join();      // (it reads the pattern that goes away here)
        c->wsym.pat = WinSym();
        win_free(c->win);
        return 0;
    }
    size_t &lds_allowed = lds_allowed_dev[c->device & 63];
    lds_allowed = std::max(lds_allowed, c->win.lds);
    const size_t lds = lds_allowed;
    if (win_allow_lds(k_spmv_win<true, WIN_WAVES, true>, lds) || win_allow_lds(k_spmv_win<false, WIN_WAVES, true>, lds) ||
        win_allow_lds(k_spmv_win<true, WIN_WAVES, false>, lds) || win_allow_lds(k_spmv_win<false, WIN_WAVES, false>, lds) ||
        win_allow_lds(k_spmv_win<true, 4, true>, lds) || win_allow_lds(k_spmv_win<false, 4, true>, lds) ||
        win_allow_lds(k_spmv_win_lo<1, WIN_WAVES>, lds) || win_allow_lds(k_spmv_win_lo<2, WIN_WAVES>, lds) ||
        win_allow_lds(k_spmv_win_lo<1, 4>, lds) || win_allow_lds(k_spmv_win_lo<2, 4>, lds) ||
        win_allow_lds(k_spmv_win_lo<0, WIN_WAVES>, lds) || win_allow_lds(k_spmv_win_lo<0, 4>, lds) ||
        win_allow_lds(k_spmv_win_lo2<1, WIN_WAVES>, lds) || win_allow_lds(k_spmv_win_lo2<2, WIN_WAVES>, lds) ||
        win_allow_lds(k_spmv_win_lo2<1, 4>, lds) || win_allow_lds(k_spmv_win_lo2<2, 4>, lds))
        return 1;
    // the two-vector FP64 SpMV keeps both windows in LDS: only where twice the window fits (fcvm_solve2 solves one after the
    // other otherwise)
    c->win2_ok = 2 * lds <= (size_t)lds_cap;
    if (c->win2_ok && (win_allow_lds(k_spmv_win2<true, WIN_WAVES>, 2 * lds) || win_allow_lds(k_spmv_win2<true, 4>, 2 * lds))) return 1;
    c->wsym_pending = true;      // (wsym_finish: at the first assembly)
    if (getenv("FCVM_WSYM_EAGER") && atoi(getenv("FCVM_WSYM_EAGER")) != 0 && wsym_finish(c)) return 1;
    if (getenv("FCVM_VERBOSE"))
        fprintf(stderr, "fcvm: window format: %lld groups of %d waves, windows mean %.0f / max %lld nodes (%zu B LDS of %d), %lld entries\n",
                (long long)w.ngroups, w.waves, w.wmean, (long long)w.wmax, lds, lds_cap, (long long)w.nblocks);
    return 0;
}

// The same SpMV on the same GPU runs at 0.51 or at 0.59 ms depending on which physical pages hipMalloc handed out for the
// value array (measured: several model instances in one process, tools/alloc_probe.py; the x / y vectors do not matter).
// So for matrices of some size a few candidate allocations are timed with the real kernel (zero values, 8 launches each,
// ~5 ms per candidate at 1 M elements) and the fastest is kept.  FCVM_VALS_TRIES=1 switches this off.
static int tune_matrix_placement(fcvm_ctx *c) {
    int tries = 3;
    if (const char *v = getenv("FCVM_VALS_TRIES")) tries = std::max(1, atoi(v));
    const int64_t nv = c->pat.value_count(), n = 3 * c->nn;
    if (tries < 2 || nv < (int64_t)32 * 1024 * 1024) return 0;   // < 256 MB: the L2 / MALL hide it
    hipStream_t s = c->stream;
    hipEvent_t e0, e1;
    HCHK(hipEventCreate(&e0));
    HCHK(hipEventCreate(&e1));
    const bool verbose = getenv("FCVM_VERBOSE") != nullptr;
    auto time_spmv = [&](double *vals, double *x, double *y, float *out) -> int {
        MatView m = fine_view(c);
        m.vals = vals;
        for (int k = 0; k < 3; k++) launch_spmv_t<false>(c, m, x, y, nullptr, nullptr);
        HCHK(hipEventRecord(e0, s));
        for (int k = 0; k < 8; k++) launch_spmv_t<false>(c, m, x, y, nullptr, nullptr);
        HCHK(hipEventRecord(e1, s));
        HCHK(hipEventSynchronize(e1));
        HCHK(hipEventElapsedTime(out, e0, e1));
        *out /= 8;
        return 0;
    };
    HCHK(hipMemsetAsync(c->d_p, 0, sizeof(double) * n, s));
    // stage 1: candidates for the value array, measured with the solver's own x / y vectors (d_p / d_q)
    {
        std::vector<double *> cand;
        std::vector<float> ms;
        for (int t = 0; t < tries; t++) {
            double *v = c->d_vals;
            if (t > 0) {
                v = nullptr;
                if (hipMalloc((void **)&v, sizeof(double) * nv) != hipSuccess) {   // no memory for another candidate
                    (void)hipGetLastError();
                    break;
                }
                HCHK(hipMemsetAsync(v, 0, sizeof(double) * nv, s));
            }
            float t_ms = 0.f;
            if (time_spmv(v, c->d_p, c->d_q, &t_ms)) return 1;
            cand.push_back(v);
            ms.push_back(t_ms);
            if (verbose) fprintf(stderr, "fcvm: value array candidate %d: SpMV %.4f ms\n", t, t_ms);
        }
        size_t best = 0;
        for (size_t i = 1; i < cand.size(); i++)
            if (ms[i] < ms[best]) best = i;
        for (size_t i = 0; i < cand.size(); i++)
            if (i != best) HCHK(hipFree(cand[i]));
        c->d_vals = cand[best];
    }
    // stage 2: candidates for the (x, y) pair of the PCG's SpMV
    {
        std::vector<std::pair<double *, double *>> cand;
        std::vector<float> ms;
        for (int t = 0; t < tries; t++) {
            double *x = c->d_p, *y = c->d_q;
            if (t > 0) {
                x = y = nullptr;
                if (hipMalloc((void **)&x, sizeof(double) * n) != hipSuccess || hipMalloc((void **)&y, sizeof(double) * n) != hipSuccess) {
                    (void)hipGetLastError();
                    if (x) HCHK(hipFree(x));
                    break;
                }
                HCHK(hipMemsetAsync(x, 0, sizeof(double) * n, s));
            }
            float t_ms = 0.f;
            if (time_spmv(c->d_vals, x, y, &t_ms)) return 1;
            cand.push_back({x, y});
            ms.push_back(t_ms);
            if (verbose) fprintf(stderr, "fcvm: vector pair candidate %d: SpMV %.4f ms\n", t, t_ms);
        }
        size_t best = 0;
        for (size_t i = 1; i < cand.size(); i++)
            if (ms[i] < ms[best]) best = i;
        for (size_t i = 0; i < cand.size(); i++)
            if (i != best) {
                HCHK(hipFree(cand[i].first));
                HCHK(hipFree(cand[i].second));
            }
        c->d_p = cand[best].first;
        c->d_q = cand[best].second;
    }
    HCHK(hipEventDestroy(e0));
    HCHK(hipEventDestroy(e1));
    return 0;
}

// Round 5: placement of the two arrays the PCG's FP64 SpMV is sensitive to, decided by measurement on the ASSEMBLED matrix (round 4 timed
// candidate value arrays filled with zeros and found nothing that carried over: zeros skip the transposed products).  What the experiments
// of this round showed (profiles/r05_spmv_slow_box.txt): on one box, with one binary, the kernel takes 0.335 or 0.385 ms depending on the
// model instance -- not the clocks, the power, the temperature, the instruction mix, the occupancy, the address translation or what runs
// before it, but (a) which allocation its scattered 8-byte result stores go to (the solver's own q: +0.04 ms in the first context of a
// process, none in twelve other vectors) and (b) which allocation holds the values.  So: the result vector first (34 MB candidates), then
// the value array (candidates filled by a device copy), five timed launches each on the solver's stream, the fastest kept.
static int tune_sym_placement(fcvm_ctx *c) {
    c->placement_tuned = true;
    if (c->tune_tries < 2 || !(c->win.on && c->wsym.on && c->wsym.valid) || c->wsym.pat.nblocks * 72 < ((int64_t)256 << 20)) return 0;   // < 256 MB: caches hide it
    hipStream_t s = c->stream;
    WinSymDev &ws = c->wsym;
    const int64_t n = 3 * c->nn, nb = ws.pat.nblocks;
    const bool verbose = getenv("FCVM_VERBOSE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    hipEvent_t e0, e1;
    HCHK(hipEventCreate(&e0));
    HCHK(hipEventCreate(&e1));
    int which = 0;      // 0: the FP64 SpMV; 1 / 2: the 16-bit passes with epilogue 1 / 2
    auto time_kernel = [&](float *out) -> int {
        auto launch = [&]() {
            if (which == 0) launch_spmv_win(c, true, c->d_p, c->d_q, nullptr);
            else if (which == 1) launch_spmv_win_lo<1>(c, c->d_md, c->d_r, c->d_mr, 0.3, nullptr, nullptr, nullptr);
            else launch_spmv_win_lo<2>(c, c->d_md, c->d_r, c->d_z, 0.3, nullptr, nullptr, nullptr);
        };
        for (int k = 0; k < 2; k++) launch();
        HCHK(hipEventRecord(e0, s));
        for (int k = 0; k < 5; k++) launch();
        HCHK(hipEventRecord(e1, s));
        HCHK(hipEventSynchronize(e1));
        HCHK(hipEventElapsedTime(out, e0, e1));
        *out /= 5;
        HCHK(hipGetLastError());
        return 0;
    };
    // one array: the current allocation against tune_tries - 1 fresh ones (filled by a device copy when the kernel reads it); the fastest is
    // kept -- a fresh one only for a gain beyond the noise of the measurement.  first / best: times with the first and the kept allocation.
    auto tune = [&](const char *name, void **slot, size_t bytes, bool copy, double *first, double *best_ms) -> int {
        std::vector<void *> cand{*slot};
        std::vector<float> ms(1, 0.f);
        void *orig = *slot;
        if (time_kernel(&ms[0])) return 1;
        for (int t = 1; t < c->tune_tries; t++) {
            void *v = nullptr;
            if (hipMalloc(&v, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            if (copy && hipMemcpyAsync(v, orig, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { (void)hipFree(v); *slot = orig; return fail("tune_sym_placement: device copy"); }
            *slot = v;
            float tm = 0.f;
            if (time_kernel(&tm)) return 1;
            cand.push_back(v);
            ms.push_back(tm);
        }
        size_t best = 0;
        for (size_t i = 1; i < cand.size(); i++)
            if (ms[i] < 0.985f * ms[best]) best = i;
        HCHK(hipStreamSynchronize(s));
        for (size_t i = 0; i < cand.size(); i++)
            if (i != best) (void)hipFree(cand[i]);
        *slot = cand[best];
        if (first) *first = ms[0];
        if (best_ms) *best_ms = ms[best];
        if (verbose) {
            fprintf(stderr, "fcvm: placement of %s:", name);
            for (size_t i = 0; i < ms.size(); i++) fprintf(stderr, " %.4f%s", ms[i], i == best ? "*" : "");
            fprintf(stderr, " ms\n");
        }
        return 0;
    };
    const dim3 gv((unsigned)cdiv(n, 256)), blk(256);
    hipLaunchKernelGGL(k_lcg_vector, gv, blk, 0, s, n, c->d_p);     // a dense x: every transposed product is formed
    int rc = tune("the FP64 SpMV's result vector (q)", (void **)&c->d_q, sizeof(double) * n, false, &c->tune_info[1], &c->tune_info[2]);
    // the arrays the kernels STREAM: in every run of round 5 a fresh allocation was as good as the first one or worse (0.375 against 0.339 ms
    // for the value array, twice) -- two candidates of 2.1 GB each cost 25 of the tuning's 36 ms for nothing: only on request
    const bool arrays = getenv("FCVM_TUNE_PLACEMENT_ARRAYS") && atoi(getenv("FCVM_TUNE_PLACEMENT_ARRAYS")) != 0;
    c->tune_info[3] = c->tune_info[4] = c->tune_info[2];
    if (!rc && arrays) rc = tune("the symmetric form's value array", (void **)&ws.vals, sizeof(double) * 9 * nb, true, &c->tune_info[3], &c->tune_info[4]);
    if (!rc && arrays) rc = tune("the symmetric form's column array", (void **)&ws.cols, sizeof(uint32_t) * nb, true, nullptr, &c->tune_info[4]);
    if (!rc && ws.lo_valid && ws.lo8 && c->win.part_lo && c->d_md && c->d_mr && c->d_dinv6f) {   // the two 16-bit smoother passes of the cycle
        hipLaunchKernelGGL(k_lcg_vector, gv, blk, 0, s, n, c->d_md);
        hipLaunchKernelGGL(k_lcg_vector, gv, blk, 0, s, n, c->d_r);
        which = 1;
        rc = tune("the first 16-bit pass's result vector", (void **)&c->d_mr, sizeof(double) * n, false, nullptr, nullptr);
        if (!rc && arrays) rc = tune("the 16-bit copy's value array", (void **)&ws.lo8, sizeof(short8) * nb, true, nullptr, nullptr);
        which = 2;
        if (!rc) rc = tune("the second 16-bit pass's result vector (z)", (void **)&c->d_z, sizeof(double) * n, false, nullptr, nullptr);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->tune_info[0] = 1.0;
    c->tune_info[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return rc;
}

// slice-length classes of k_assemble_lds (blocks of the longest row): a quarter slice needs asm_lds_bytes() of LDS
static const int64_t ASM_LEN_LIMIT[fcvm_ctx::ASM_BUCKETS] = {19, 27, 41, 65, 82, 98, 133};   // 28 / 37 / 54 / 81 / 100 / 119 / 160 KiB: 5 / 4 / 3 / 2 / 1 / 1 / 1 per CU

static int build_asm_items(fcvm_ctx *c) {
    const Pattern &p = c->pat;
    c->asm_lds_ok = false;
    dfree(c->d_asm_items);
    if (p.maxlen > ASM_LEN_LIMIT[fcvm_ctx::ASM_BUCKETS - 1]) return 0;   // a row that does not fit the LDS: global read-modify-write kernel
    std::vector<int32_t> items[fcvm_ctx::ASM_BUCKETS];
    NEED(p.nslices < (int64_t(1) << 29), "too many slices for 32-bit assembly work items");
    for (int64_t s = 0; s < p.nslices; s++) {
        const int64_t m = p.slice_len(s);
        int b = 0;
        while (m > ASM_LEN_LIMIT[b]) b++;
        for (int q = 0; q < 4; q++) items[b].push_back((int32_t)(s * 4 + q));
    }
    std::vector<int32_t> all;
    for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) {
        c->asm_first[b] = (int64_t)all.size();
        c->asm_count[b] = (int64_t)items[b].size();
        all.insert(all.end(), items[b].begin(), items[b].end());
    }
    if (dalloc(&c->d_asm_items, (int64_t)all.size())) return 1;
    HCHK(hipMemcpyAsync(c->d_asm_items, all.data(), sizeof(int32_t) * all.size(), hipMemcpyHostToDevice, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    // dynamic LDS beyond 64 KiB has to be requested per kernel; the limit belongs to the kernel on a device, not to this
    // context: only raise it.  A length class the device's LDS cannot hold sends the mesh to k_assemble_rows.
    static int asm_lds_allowed_dev[64] = {0};
    int &asm_lds_allowed = asm_lds_allowed_dev[c->device & 63];
    if (asm_lds_allowed == 0) asm_lds_allowed = 64 * 1024;
    const int lds_cap = device_lds_cap(c->device);
    for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++)
        if (c->asm_count[b] && (int64_t)asm_lds_bytes(ASM_LEN_LIMIT[b]) > lds_cap) return 0;   // asm_lds_ok stays false
    for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) {
        const int bytes = (int)asm_lds_bytes(ASM_LEN_LIMIT[b]);
        if (bytes <= asm_lds_allowed || !c->asm_count[b]) continue;
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HCHK(hipFuncSetAttribute((const void *)k_assemble_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        asm_lds_allowed = bytes;
    }
    c->asm_lds_ok = true;
    return 0;
}

template <bool TANGENT>
static int launch_assemble_lds(fcvm_ctx *c, const double *disp, const double *sig, const uint8_t *pgp, const MatConst &mc,
                               const double grav[3], double *glv, double *modf, double *xgp, double *vol_part, int penalty_bc) {
    int nb = 0;
    for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) nb += c->asm_count[b] ? 1 : 0;
    const bool overlap = c->asm_overlap && nb > 1;
    if (overlap) {
        if (!c->asm_fork) {
            HCHK(hipEventCreateWithFlags(&c->asm_fork, hipEventDisableTiming));
            for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) {
                HCHK(hipStreamCreateWithFlags(&c->asm_stream[b], hipStreamNonBlocking));
                HCHK(hipEventCreateWithFlags(&c->asm_join[b], hipEventDisableTiming));
            }
        }
        HCHK(hipEventRecord(c->asm_fork, c->stream));
    }
    for (int b = fcvm_ctx::ASM_BUCKETS - 1; b >= 0; b--) {   // longest rows first: their items run longest
        if (!c->asm_count[b]) continue;
        hipStream_t st = overlap ? c->asm_stream[b] : c->stream;
        if (overlap) HCHK(hipStreamWaitEvent(st, c->asm_fork, 0));
        hipLaunchKernelGGL(k_assemble_lds<TANGENT>, dim3((unsigned)c->asm_count[b]), dim3(64), (size_t)asm_lds_bytes(ASM_LEN_LIMIT[b], TANGENT), st,
                           c->d_asm_items + c->asm_first[b], c->d_sptr, c->d_rownode, c->d_nptr, c->d_nel, c->d_el, c->d_coord, disp, c->d_fixmask,
                           c->d_fixval, c->d_emap, sig, pgp, mc, grav[0], grav[1], grav[2], c->d_vals, glv, modf, xgp,
                           vol_part ? vol_part + c->asm_first[b] : nullptr, penalty_bc);
        if (overlap) {
            HCHK(hipEventRecord(c->asm_join[b], st));
            HCHK(hipStreamWaitEvent(c->stream, c->asm_join[b], 0));
        }
    }
    HCHK(hipGetLastError());
    return 0;
}

// the same kernel on the work items of the symmetric window form: rows leave LDS straight into the SpMV's storage
template <bool TANGENT>
static int launch_assemble_sym(fcvm_ctx *c, const double *disp, const double *sig, const uint8_t *pgp, const MatConst &mc,
                               const double grav[3], double *glv, double *modf, double *xgp, bool want_lo) {
    static_assert(WinSymDev::ASM_BUCKETS <= fcvm_ctx::ASM_BUCKETS, "one stream per length class");
    WinSymDev &d = c->wsym;
    int nb = 0;
    for (int b = 0; b < WinSymDev::ASM_BUCKETS; b++) nb += d.asm_count[b] ? 1 : 0;
    const bool overlap = c->asm_overlap && nb > 1;
    if (overlap) {
        if (!c->asm_fork) {
            HCHK(hipEventCreateWithFlags(&c->asm_fork, hipEventDisableTiming));
            for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) {
                HCHK(hipStreamCreateWithFlags(&c->asm_stream[b], hipStreamNonBlocking));
                HCHK(hipEventCreateWithFlags(&c->asm_join[b], hipEventDisableTiming));
            }
        }
        HCHK(hipEventRecord(c->asm_fork, c->stream));
    }
    AsmSym sy;
    sy.rownode = d.rownode; sy.seg = d.seg; sy.emap = d.emap; sy.rowlen = d.rowlen; sy.jfirst = d.jfirst; sy.rowoff = d.rowoff;
    sy.nextseg = d.nextseg; sy.woff = d.woff; sy.cols = d.cols; sy.diagk = d.diagk; sy.vals = d.vals;
    sy.lo8 = want_lo ? d.lo8 : nullptr; sy.lo2 = want_lo ? d.lo2 : nullptr; sy.dblk = c->d_dblk; sy.maxl1 = d.maxl1; sy.waves = d.pat.waves; sy.len = 0;
    sy.nt = c->asm_nt;
    for (int b = WinSymDev::ASM_BUCKETS - 1; b >= 0; b--) {   // longest rows first: their items run longest
        if (!d.asm_count[b]) continue;
        hipStream_t st = overlap ? c->asm_stream[b] : c->stream;
        if (overlap) HCHK(hipStreamWaitEvent(st, c->asm_fork, 0));
        sy.len = (int)ASM_SYM_LIMIT[b];
        size_t lds_b = (size_t)asm_lds_bytes(ASM_SYM_LIMIT[b], TANGENT, d.asm_rows[b]);
        if (const char *v = getenv("FCVM_ASM_LDS_MIN")) lds_b = std::min<size_t>(std::max<size_t>(lds_b, (size_t)atol(v)), 64 * 1024);   // (experiment: fewer items per CU)
#define FCVM_LAUNCH_ASM_SYM(LPR_)                                                                                                              \
        hipLaunchKernelGGL((k_assemble_lds<TANGENT, true, LPR_>), dim3((unsigned)d.asm_count[b]), dim3(64), lds_b, st,                         \
                           d.asm_items + d.asm_off[b], c->d_sptr, c->d_rownode, c->d_nptr, c->d_nel, c->d_el, c->d_coord, disp, \
                           c->d_fixmask, c->d_fixval, c->d_emap, sig, pgp, mc, grav[0], grav[1], grav[2], (double *)nullptr, glv, modf, xgp,  \
                           d.part_vol + d.asm_first[b], 0, sy)
        if (d.asm_rows[b] == 16) FCVM_LAUNCH_ASM_SYM(4);
        else if (d.asm_rows[b] == 8) FCVM_LAUNCH_ASM_SYM(8);
        else FCVM_LAUNCH_ASM_SYM(16);
#undef FCVM_LAUNCH_ASM_SYM
        if (overlap) {
            HCHK(hipEventRecord(c->asm_join[b], st));
            HCHK(hipStreamWaitEvent(c->stream, c->asm_join[b], 0));
        }
    }
    HCHK(hipGetLastError());
    return 0;
}

extern "C" {

const char *fcvm_last_error(void) { return g_err.c_str(); }
int fcvm_version(void) { return 100; }

int fcvm_device_count(int *count) {
    *count = 0;
    HCHK(hipGetDeviceCount(count));
    return 0;
}

int fcvm_create(int device, fcvm_ctx **out) {
    *out = nullptr;
    int n = 0;
    HCHK(hipGetDeviceCount(&n));
    NEED(n > 0 && device >= 0 && device < n, "fcvm_create: no such HIP device");
    HCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HCHK(hipGetDeviceProperties(&prop, device));
    NEED(prop.warpSize == 64, "fcvm_create: this library is built for 64-wide wavefronts (gfx950)");
    fcvm_ctx *c = new fcvm_ctx();
    c->device = device;
    if (const char *v = getenv("FCVM_SPMV_VARIANT")) c->spmv_variant = atoi(v);
    if (const char *v = getenv("FCVM_SPMV_WIN")) c->spmv_win = atoi(v);
    if (const char *v = getenv("FCVM_WIN_V2")) c->win_v2 = atoi(v) != 0;
    if (const char *v = getenv("FCVM_ASSEMBLE_LDS")) c->assemble_lds = atoi(v);
    if (const char *v = getenv("FCVM_ASSEMBLE_OVERLAP")) c->asm_overlap = atoi(v);
    if (const char *v = getenv("FCVM_REPACK_OVERLAP")) c->repack_overlap = atoi(v);
    if (const char *v = getenv("FCVM_LEVEL1_OVERLAP")) c->level1_overlap = atoi(v);
    if (const char *v = getenv("FCVM_QIN_ATOMIC")) c->qin_atomic = atoi(v);
    if (const char *v = getenv("FCVM_SPMV_SPW")) c->spmv_spw = std::min(4, std::max(1, atoi(v)));   // slices (waves) per workgroup
    if (const char *v = getenv("FCVM_MG_ADDITIVE")) c->mg_additive = std::min(2, std::max(0, atoi(v)));
    if (const char *v = getenv("FCVM_MG_MAX_AGGREGATES")) c->mg_max_dense = std::max(8, atoi(v));
    if (const char *v = getenv("FCVM_MG_FP32")) c->mg_fp32 = atoi(v);
    if (const char *v = getenv("FCVM_MG_L1_16")) c->mg_l1_16 = atoi(v);
    if (const char *v = getenv("FCVM_MG_ADD_W")) c->mg_add_w = atof(v);
    if (const char *v = getenv("FCVM_WIN_SYM_LO")) c->wsym_lo = atoi(v) != 0;
    if (const char *v = getenv("FCVM_SYM_GRADING_MAX")) c->grading_max = atof(v) > 1.0 ? atof(v) : c->grading_max;
    if (const char *v = getenv("FCVM_MG_FUSE_PROLONG")) c->mg_fuse_prolong = atoi(v) != 0;
    if (const char *v = getenv("FCVM_MG_PATCH")) c->mg_patch = atoi(v) != 0;
    if (const char *v = getenv("FCVM_L1_X32")) c->mg_l1_x32 = atoi(v) != 0;
    if (const char *v = getenv("FCVM_SYM_PAIR")) c->sym_pair = atoi(v) != 0;
    if (const char *v = getenv("FCVM_PAIR_OVERLAP")) c->pair_overlap = atoi(v) != 0;
    if (const char *v = getenv("FCVM_PCG_EVENTS")) c->pcg_events = atoi(v) != 0;
    if (const char *v = getenv("FCVM_TUNE_PLACEMENT")) c->tune_tries = std::min(8, std::max(1, atoi(v)));
    if (const char *v = getenv("FCVM_ASSEMBLE_SYM")) c->asm_sym = atoi(v);
    if (const char *v = getenv("FCVM_ASM_NT")) c->asm_nt = atoi(v);
    if (const char *v = getenv("FCVM_ASM_LPR")) { const int l = atoi(v); if (l == 4 || l == 8 || l == 16) c->asm_lpr = l; }
    if (const char *v = getenv("FCVM_ASM_LPR_LONG")) { const int l = atoi(v); if (l == 4 || l == 8 || l == 16) c->asm_lpr_long = l; }
    if (const char *v = getenv("FCVM_ASM_LPR_FROM")) c->asm_lpr_from = atoi(v);
    HCHK(hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking));
    c->stream = c->own;
    if (dalloc(&c->d_S, S_COUNT) || dalloc(&c->d_out8, 16)) return 1;
    {   // group -> workgroup map of the symmetric-form kernels (win_group_of_block): chunks of 16 groups per XCD unless FCVM_WIN_GMAP says otherwise
        int mode = 16;
        if (const char *v = getenv("FCVM_WIN_GMAP")) mode = std::min(4096, std::max(0, atoi(v)));
        HCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_win_gmap), &mode, sizeof(int)));
    }
    *out = c;
    return 0;
}

static void alt_free(fcvm_ctx *c) {
    MgAlt &a = c->alt;
    dfree(a.r); dfree(a.z); dfree(a.p); dfree(a.q); dfree(a.S); dfree(a.part_vec); dfree(a.part_spmv); dfree(a.part_lo);
    dfree(a.md); dfree(a.mr); dfree(a.mt); dfree(a.x1); dfree(a.rr1); dfree(a.d1); dfree(a.t1); dfree(a.r2); dfree(a.e2);
    dfree(a.sx); dfree(a.sr); dfree(a.sd); dfree(a.st); dfree(a.sr3); dfree(a.se3); dfree(a.symv_part); dfree(a.spart3); dfree(a.py); dfree(a.d1_32); dfree(a.t1_32);
    a = MgAlt();
}

static void mg_topology_join(fcvm_ctx *c, bool discard);
static void mg_topology_start(fcvm_ctx *c);
static void free_mesh(fcvm_ctx *c) {
    mg_topology_join(c, true);
    if (c->sym_thread.joinable()) c->sym_thread.join();
    c->wsym_pending = false;
    dfree(c->d_el); dfree(c->d_cols); dfree(c->d_rownode); dfree(c->d_noce); dfree(c->d_sptr);
    dfree(c->d_emap); dfree(c->d_diagoff); dfree(c->d_fixmask); dfree(c->d_coord); dfree(c->d_fixval);
    dfree(c->d_nptr); dfree(c->d_nel); dfree(c->d_fe); dfree(c->d_part_vol); dfree(c->d_asm_items); dfree(c->d_slotof);
    c->sell_valid = false;
    win_free(c->win);
    wsym_free(c->wsym);
    dfree(c->d_vals); dfree(c->d_vals_g); dfree(c->d_dblk); dfree(c->d_dinv); dfree(c->d_dinv6f); dfree(c->d_r); dfree(c->d_z); dfree(c->d_p);
    dfree(c->d_q); dfree(c->d_part_vec); dfree(c->d_part_spmv); dfree(c->d_part_misc); dfree(c->d_pidx);
    mg_free(c);
    alt_free(c);
    c->have_mesh = c->assembled = false;
}

int fcvm_destroy(fcvm_ctx *c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_mesh(c);
    dfree(c->d_S); dfree(c->d_out8); dfree(c->d_part_mdot); dfree(c->d_grading);
    for (auto e : c->evpool) (void)hipEventDestroy(e);
    if (c->h_pcgS) (void)hipHostFree(c->h_pcgS);
    if (c->h_pcgS2) (void)hipHostFree(c->h_pcgS2);
    if (c->level1_fork) (void)hipEventDestroy(c->level1_fork);
    if (c->level1_done) (void)hipEventDestroy(c->level1_done);
    if (c->level1_stream) (void)hipStreamDestroy(c->level1_stream);
    if (c->pair_fork) (void)hipEventDestroy(c->pair_fork);
    if (c->pair_done) (void)hipEventDestroy(c->pair_done);
    if (c->pair_stream) (void)hipStreamDestroy(c->pair_stream);
    if (c->repack_fork) (void)hipEventDestroy(c->repack_fork);
    if (c->repack_done) (void)hipEventDestroy(c->repack_done);
    if (c->repack_stream) (void)hipStreamDestroy(c->repack_stream);
    if (c->asm_fork) (void)hipEventDestroy(c->asm_fork);
    for (int b = 0; b < fcvm_ctx::ASM_BUCKETS; b++) {
        if (c->asm_join[b]) (void)hipEventDestroy(c->asm_join[b]);
        if (c->asm_stream[b]) (void)hipStreamDestroy(c->asm_stream[b]);
    }
    if (c->own) (void)hipStreamDestroy(c->own);
    delete c;
    return 0;
}

/* free and total device memory in bytes (hipMemGetInfo) */
int fcvm_mem_info(fcvm_ctx *c, int64_t out[2]) {
    NEED(c && out, "fcvm_mem_info: bad arguments");
    size_t fr = 0, tot = 0;
    HCHK(hipSetDevice(c->device));
    HCHK(hipMemGetInfo(&fr, &tot));
    out[0] = (int64_t)fr; out[1] = (int64_t)tot;
    return 0;
}

int fcvm_synchronize(fcvm_ctx *c) {
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int fcvm_set_stream(fcvm_ctx *c, void *s, int external) {
    HCHK(hipStreamSynchronize(c->stream));
    c->stream = external ? (hipStream_t)s : c->own;  // external NULL = HIP's default (null) stream
    return 0;
}
int fcvm_malloc(fcvm_ctx *c, uint64_t bytes, void **dev) {
    HCHK(hipSetDevice(c->device));
    const bool poison = getenv("FCVM_POISON") && atoi(getenv("FCVM_POISON")) != 0;      // (see dalloc)
    HCHK(hipMalloc(dev, (bytes ? bytes : 8) + (poison ? 512 : 0)));
    if (poison) {
        HCHK(hipMemset(*dev, 0xFF, (bytes ? bytes : 8) + 512));
        HCHK(hipDeviceSynchronize());
    }
    return 0;
}
int fcvm_free(fcvm_ctx *c, void *dev) {
    HCHK(hipStreamSynchronize(c->stream));
    HCHK(hipFree(dev));
    return 0;
}
int fcvm_h2d(fcvm_ctx *c, void *dev, const void *host, uint64_t bytes) {
    HCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int fcvm_d2h(fcvm_ctx *c, void *host, const void *dev, uint64_t bytes) {
    HCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int fcvm_d2d(fcvm_ctx *c, void *dst, const void *src, uint64_t bytes) {
    HCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}
int fcvm_memset0(fcvm_ctx *c, void *dev, uint64_t bytes) {
    HCHK(hipMemsetAsync(dev, 0, bytes, c->stream));
    return 0;
}

int fcvm_set_mesh(fcvm_ctx *c, int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord) {
    NEED(c && ne > 0 && nn > 0 && elNodes && nocoord, "fcvm_set_mesh: empty mesh");
    NEED(nn < (1ll << 31) / 3, "fcvm_set_mesh: too many nodes for 32-bit node indices");
    HCHK(hipSetDevice(c->device));
    HCHK(hipStreamSynchronize(c->stream));
    free_mesh(c);
    const bool timing = getenv("FCVM_PATTERN_TIMING") != nullptr;
    auto tprev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "fcvm set_mesh: %-30s %.3f s\n", what, std::chrono::duration<double>(t - tprev).count());
        tprev = t;
    };
    std::vector<int32_t> el(ne * 10);
    for (int64_t i = 0; i < ne * 10; i++) {
        int64_t v = elNodes[i] - 1;
        NEED(v >= 0 && v < nn, "fcvm_set_mesh: element node number out of range (expected 1..nn)");
        el[i] = (int32_t)v;
    }
    bool emap_host = false;
    try {
        int nt = (int)std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 32) nt = 32;
        int win = SORT_WINDOW;
        if (const char *v = getenv("FCVM_SORT_WINDOW")) win = std::max(1, atoi(v));
        // row order of the matrix: the caller's numbering when it is band-like (structured meshes), otherwise a
        // Morton curve through the node coordinates (vectors keep the caller's order; only the order in which the
        // SpMV walks the rows -- and with it the locality of its x-gathers -- changes)
        int mode = -1;  // auto
        if (const char *v = getenv("FCVM_ROW_ORDER")) mode = atoi(v);
        c->row_spread = numbering_spread(ne, nn, el.data(), 10);
        // with the window format the sliced-ELL rows follow the same Morton curve: k_repack_win then copies between two
        // layouts whose neighbourhoods coincide (its scattered 8-byte reads are absorbed by the L2 instead of going to HBM)
        const bool use_morton = mode == 1 || (mode < 0 && nn > 4096 && (c->row_spread > 0.05 || c->spmv_win));
        lap("conversion, numbering spread");
        std::vector<int32_t> order;
        if (use_morton) order = morton_order(nn, nocoord);
        c->h_morton = order;      // (the window format walks the same curve)
        lap("Morton order");
        c->row_order_morton = use_morton;
        // the element map is filled on the device below unless FCVM_EMAP_HOST=1 asks for the host's
        emap_host = getenv("FCVM_EMAP_HOST") && atoi(getenv("FCVM_EMAP_HOST")) != 0;
        build_pattern(ne, nn, el.data(), c->pat, nt, win, 10, use_morton ? order.data() : nullptr, emap_host);
        lap("sliced-ELL pattern (total)");
    } catch (const std::exception &ex) {
        return fail(std::string("fcvm_set_mesh: ") + ex.what());
    }
    c->ne = ne;
    c->nn = nn;
    // the window format's host work (0.07 s at 1 M elements) beside the allocations and uploads below
    WinBuild winres;
    std::thread winthread(win_build_host, c, nocoord, &winres);
    // joined on every return path, the symmetric form's builder (started by winthread) after it
    struct Joiner {
        std::thread &t; std::thread &t2; bool keep_second = false;
        ~Joiner() { if (t.joinable()) t.join(); if (!keep_second && t2.joinable()) t2.join(); }
    } joiner{winthread, c->sym_thread};
    c->h_el = el;
    c->h_coord.assign(nocoord, nocoord + 3 * nn);
    c->h_fixmask.assign(nn, 0);
    mg_topology_start(c);      // level 1 of the multigrid hierarchy needs the mesh only: its host work starts here (joined at the first assembly)
    const Pattern &p = c->pat;
    std::vector<int32_t> noce(nn, 0);  // fcVM.py:183-185
    for (int64_t i = 0; i < ne * 10; i++) noce[el[i]]++;
    if (dalloc(&c->d_el, ne * 10) || dalloc(&c->d_cols, p.nslots) || dalloc(&c->d_rownode, p.nslices * SLICE) ||
        dalloc(&c->d_noce, nn) || dalloc(&c->d_sptr, p.nslices + 1) || dalloc(&c->d_emap, ne * 100) ||
        dalloc(&c->d_diagoff, nn) || dalloc(&c->d_fixmask, nn) || dalloc(&c->d_coord, nn * 3) ||
        dalloc(&c->d_fixval, nn * 3) || dalloc(&c->d_vals, p.value_count()) || dalloc(&c->d_dblk, nn * 9) ||
        dalloc(&c->d_dinv, nn * 9) || dalloc(&c->d_dinv6f, nn * 6) || dalloc(&c->d_r, nn * 3) || dalloc(&c->d_z, nn * 3) || dalloc(&c->d_p, nn * 3) ||
        dalloc(&c->d_q, nn * 3) || dalloc(&c->d_nptr, nn + 1) || dalloc(&c->d_nel, ne * 10) || dalloc(&c->d_part_vol, 4 * p.nslices))
        return 1;
    c->nb_vec = cdiv(nn, 256);
    c->nb_spmv = cdiv(p.nslices, c->spmv_spw);
    c->nb_gp = cdiv(4 * ne, 256);
    c->nb_part_spmv = c->nb_spmv;
    if (dalloc(&c->d_part_vec, 2 * c->nb_vec) || dalloc(&c->d_part_spmv, c->nb_spmv) ||
        dalloc(&c->d_part_misc, 2 * std::max<int64_t>(c->nb_gp, 4096)) || dalloc(&c->d_pidx, c->nb_gp))
        return 1;
    hipStream_t s = c->stream;
    HCHK(hipMemcpyAsync(c->d_el, el.data(), sizeof(int32_t) * ne * 10, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_cols, p.cols.data(), sizeof(int32_t) * p.nslots, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_rownode, p.rownode.data(), sizeof(int32_t) * p.nslices * SLICE, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_noce, noce.data(), sizeof(int32_t) * nn, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_sptr, p.sptr.data(), sizeof(int64_t) * (p.nslices + 1), hipMemcpyHostToDevice, s));
    dfree(c->d_slotof);
    if (dalloc(&c->d_slotof, nn)) return 1;      // (kept until the symmetric window form has built its own element map)
    HCHK(hipMemcpyAsync(c->d_slotof, p.slotof.data(), sizeof(int32_t) * nn, hipMemcpyHostToDevice, s));
    if (emap_host) {
        HCHK(hipMemcpyAsync(c->d_emap, p.emap.data(), sizeof(uint32_t) * ne * 100, hipMemcpyHostToDevice, s));
    } else {   // (after d_el, d_cols and d_sptr above, in stream order)
        hipLaunchKernelGGL(k_build_emap, dim3((unsigned)cdiv(ne * 10, 256)), dim3(256), 0, s, ne, c->d_el, c->d_slotof, c->d_sptr, c->d_cols, c->d_emap);
        HCHK(hipGetLastError());
    }
    HCHK(hipMemcpyAsync(c->d_diagoff, p.diagoff.data(), sizeof(uint64_t) * nn, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_nptr, p.nptr.data(), sizeof(int64_t) * (nn + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_nel, p.nel.data(), sizeof(int32_t) * ne * 10, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_coord, nocoord, sizeof(double) * nn * 3, hipMemcpyHostToDevice, s));
    HCHK(hipMemsetAsync(c->d_fixmask, 0, nn, s));
    HCHK(hipMemsetAsync(c->d_fixval, 0, sizeof(double) * nn * 3, s));
    HCHK(hipMemsetAsync(c->d_vals, 0, sizeof(double) * p.value_count(), s));
    HCHK(hipStreamSynchronize(s));
    lap("allocation, upload");
    c->ne = ne;
    c->nn = nn;
    if (build_asm_items(c)) return 1;
    lap("assembly work items");
    winthread.join();
    if (setup_window_format(c, winres)) return 1;
    lap("window format (rest of the build + upload)");
    if (!c->wsym_pending) dfree(c->d_slotof);
    joiner.keep_second = c->wsym_pending;      // (the symmetric form's builder may run on: wsym_finish joins it)
    c->h_morton.clear();
    c->h_morton.shrink_to_fit();
    if (!c->win.on && tune_matrix_placement(c)) return 1;
    // the element->slot map is only needed on the device from here on
    c->pat.emap.clear();
    c->pat.emap.shrink_to_fit();
    c->have_mesh = true;
    c->wsym_graded = false;     // (the grading is a property of the mesh: measured again at its first assembly)
    c->grading[0] = c->grading[1] = 0.0;
    c->placement_tuned = false;
    for (double &v : c->tune_info) v = 0.0;
    return 0;
}

/* completes what fcvm_set_mesh left running in the background (the symmetric window form's host build and its device half);
 * called implicitly by the first fcvm_assemble -- a caller with host work of its own between the two calls it after that work */
int fcvm_finish_setup(fcvm_ctx *c) {
    NEED(c && c->have_mesh, "fcvm_finish_setup: no mesh");
    return wsym_finish(c);
}

int fcvm_set_coords(fcvm_ctx *c, const double *nocoord) {
    NEED(c && c->have_mesh, "fcvm_set_coords: no mesh");
    HCHK(hipMemcpyAsync(c->d_coord, nocoord, sizeof(double) * c->nn * 3, hipMemcpyHostToDevice, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    mg_topology_join(c, false);    // (the thread reads h_coord for the row order of level 1; the numbering it built stays valid)
    c->h_coord.assign(nocoord, nocoord + 3 * c->nn);
    c->mg_topo = false;
    c->wsym_graded = false;     // (new geometry: the grading is measured again at the next assembly)
    return 0;
}

int fcvm_set_material(fcvm_ctx *c, double E, double nu, double density) {
    NEED(c && E > 0 && nu > -1.0 && nu < 0.5, "fcvm_set_material: need E > 0 and -1 < nu < 0.5");
    c->E = E; c->nu = nu; c->rho = density;
    c->have_mat = true;
    c->mg_lmax_stale = true;
    return 0;
}

int fcvm_set_fix(fcvm_ctx *c, int64_t nfix, const int64_t *dofs, const double *vals) {
    NEED(c && c->have_mesh, "fcvm_set_fix: set the mesh first");
    std::vector<uint8_t> mask(c->nn, 0);
    std::vector<double> val(c->nn * 3, 0.0);
    for (int64_t i = 0; i < nfix; i++) {
        NEED(dofs[i] >= 0 && dofs[i] < 3 * c->nn, "fcvm_set_fix: dof out of range");
        mask[dofs[i] / 3] |= (uint8_t)(1u << (dofs[i] % 3));
        val[dofs[i]] = vals[i];
    }
    c->h_fixmask = mask;
    c->mg_lmax_stale = true;
    c->fixmask1_valid = false;
    HCHK(hipMemcpyAsync(c->d_fixmask, mask.data(), c->nn, hipMemcpyHostToDevice, c->stream));
    HCHK(hipMemcpyAsync(c->d_fixval, val.data(), sizeof(double) * c->nn * 3, hipMemcpyHostToDevice, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_pattern_info(fcvm_ctx *c, int64_t out[8]) {
    NEED(c && c->have_mesh, "fcvm_pattern_info: no mesh");
    const Pattern &p = c->pat;
    out[0] = p.nnzb; out[1] = p.nslots; out[2] = p.nslices;
    out[3] = p.value_count() * 8 + p.nslots * 4;
    out[4] = p.maxlen; out[5] = 3 * p.nn;
    out[6] = 1; out[7] = c->spmv_variant;
    return 0;
}

int fcvm_spmv_info(fcvm_ctx *c, int64_t out[8]) {
    NEED(c && c->have_mesh, "fcvm_spmv_info: no mesh");
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = c->win.on ? c->win.pat.waves : 0;
    if (!c->win.on) return 0;
    const WinPattern &w = c->win.pat;
    out[1] = w.ngroups; out[2] = w.wmax; out[3] = (int64_t)(w.wmean + 0.5); out[4] = w.nblocks;
    // bytes one launch streams from this format: values, 16-bit columns, window lists, per-vrow tables
    out[5] = w.nblocks * 74 + w.gptr[w.ngroups] * 4 + w.nwaves * SLICE * 16 + (w.nwaves + w.ngroups) * 8;
    out[6] = (int64_t)c->win.lds; out[7] = w.split;
    return 0;
}

/* the symmetric form of the window format (the PCG's FP64 SpMV): out[0] = 1 if built for this mesh, out[1] = stored entries,
 * out[2] = blocks of the matrix not stored (their transposes are), out[3] = bytes one launch streams from the format, out[4] =
 * LDS bytes per workgroup, out[5] = 1 if the values of the last assembly are in it (fcvm_spmv / fcvm_solve then use it), out[6] = 1
 * if the assembly kernels can write the form themselves, out[7] = 1 if the last assembly did (no sliced-ELL array, no copy) */
/* group -> workgroup map of the symmetric-form kernels on this context's device (see win_group_of_block): 0 = an eighth of the groups per
 * XCD, 1 = identity, C >= 2 = chunks of C groups per XCD.  A device-wide switch (a __device__ variable): all contexts of the process on that
 * device follow it. */
int fcvm_set_group_map(fcvm_ctx *c, int mode) {
    NEED(c && mode >= 0 && mode <= 4096, "fcvm_set_group_map: mode 0, 1 or a chunk length up to 4096");
    HCHK(hipSetDevice(c->device));
    HCHK(hipStreamSynchronize(c->stream));
    HCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_win_gmap), &mode, sizeof(int)));
    return 0;
}

/* what tune_sym_placement measured for this mesh: out[0] = 1 if it ran, out[1] / out[2] = SpMV time [ms] with the result vector first
 * allocated / kept, out[3] / out[4] = with the value array first allocated / kept, out[5] = milliseconds the tuning took */
int fcvm_placement_info(fcvm_ctx *c, double out[6]) {
    NEED(c && out, "fcvm_placement_info: bad arguments");
    for (int k = 0; k < 6; k++) out[k] = c->tune_info[k];
    return 0;
}

/* experiment hook (round 5): device addresses of the PCG's own vectors r, z, p, q and of the partial buffer of its SpMV */
int fcvm_debug_pcg_vectors(fcvm_ctx *c, uint64_t out[5]) {
    NEED(c && c->have_mesh && out, "fcvm_debug_pcg_vectors: no mesh");
    out[0] = (uint64_t)(uintptr_t)c->d_r; out[1] = (uint64_t)(uintptr_t)c->d_z; out[2] = (uint64_t)(uintptr_t)c->d_p;
    out[3] = (uint64_t)(uintptr_t)c->d_q; out[4] = (uint64_t)(uintptr_t)c->d_part_spmv;
    return 0;
}

int fcvm_spmv_sym_info(fcvm_ctx *c, int64_t out[10]) {
    NEED(c && c->have_mesh && out, "fcvm_spmv_sym_info: no mesh");
    if (wsym_finish(c)) return 1;
    for (int k = 0; k < 10; k++) out[k] = 0;
    out[8] = c->wsym_graded ? 1 : 0;                            // declined for this mesh: graded beyond FCVM_SYM_GRADING_MAX
    out[9] = c->grading[1] > 0.0 ? (int64_t)std::min(c->grading[0] / c->grading[1], 9.0e18) : 0;   // max / min nodal diagonal entry, last assembly
    if (!c->wsym.on) return 0;
    const WinSym &w = c->wsym.pat;
    const WinPattern &f = c->win.pat;
    out[0] = 1; out[1] = w.nblocks; out[2] = w.dropped;
    out[3] = w.nblocks * 76 + f.gptr[f.ngroups] * 4 + w.nwaves * SLICE * 16 + (w.nwaves + w.ngroups) * 8;
    out[4] = (int64_t)c->wsym.lds; out[5] = c->wsym.valid ? 1 : 0;
    out[6] = c->wsym.asm_ok ? 1 : 0;                            // the assembly can write this form itself (k_assemble_lds<.., SYM>)
    out[7] = (c->wsym.valid && !c->sell_valid) ? 1 : 0;         // ... and the last one did
    return 0;
}

// the matrix of the last assembly read back from the SYMMETRIC window form (an assembly straight into it leaves the sliced-ELL
// array untouched): every stored entry gives block (i, j) and, where it carries a transposed product, block (j, i) = its transpose
static int matrix_to_coo_sym(fcvm_ctx *c, int64_t *row, int64_t *col, double *val) {
    const Pattern &p = c->pat;
    const WinSymDev &d = c->wsym;
    const WinDev &f = c->win;
    const int64_t nrows = d.pat.nwaves * SLICE, nb = d.pat.nblocks, ng = f.pat.ngroups;
    std::vector<double> v(9 * nb);
    std::vector<uint32_t> cols(nb);
    std::vector<int64_t> woff(d.pat.nwaves + 1), gptr(ng + 1);
    std::vector<uint16_t> rowlen(nrows);
    std::vector<int32_t> rownode(nrows);
    HCHK(hipStreamSynchronize(c->stream));
    HCHK(hipMemcpy(v.data(), d.vals, sizeof(double) * 9 * nb, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(cols.data(), d.cols, sizeof(uint32_t) * nb, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(woff.data(), d.woff, sizeof(int64_t) * (d.pat.nwaves + 1), hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(gptr.data(), f.gptr, sizeof(int64_t) * (ng + 1), hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(rowlen.data(), d.rowlen, sizeof(uint16_t) * nrows, hipMemcpyDeviceToHost));
    HCHK(hipMemcpy(rownode.data(), d.rownode, sizeof(int32_t) * nrows, hipMemcpyDeviceToHost));
    std::vector<int32_t> wlist(gptr[ng]);
    HCHK(hipMemcpy(wlist.data(), f.wlist, sizeof(int32_t) * wlist.size(), hipMemcpyDeviceToHost));
    int64_t k = 0;
    for (int64_t i = 0; i < p.nn; i++)
        for (int64_t q = p.rowptr[i]; q < p.rowptr[i + 1]; q++)
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) {
                    row[k] = 3 * i + r;
                    col[k] = 3 * (int64_t)p.colidx[q] + cc;
                    val[k] = 0.0;
                    k++;
                }
    auto put = [&](int32_t i, int32_t j, const double *b9, bool transposed) {
        const int32_t *rb = p.colidx.data() + p.rowptr[i], *re = p.colidx.data() + p.rowptr[i + 1];
        const int64_t q = std::lower_bound(rb, re, j) - p.colidx.data();
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) val[9 * q + 3 * r + cc] = transposed ? b9[3 * cc + r] : b9[3 * r + cc];
    };
    const int64_t R = (int64_t)d.pat.waves * SLICE;
    for (int64_t wave = 0; wave < d.pat.nwaves; wave++) {
        const int64_t g = wave / d.pat.waves;
        const uint16_t *len = &rowlen[wave * SLICE];
        int64_t off = woff[wave];
        for (int64_t j = 0; j < len[0]; j++) {
            int64_t n = 0;
            while (n < SLICE && len[n] > j) n++;
            for (int64_t l = 0; l < n; l++) {
                const uint32_t en = cols[off + l];
                const int32_t i = rownode[wave * SLICE + l], jn = wlist[gptr[g] + (en & 0xFFFFu)];
                double b9[9];
                for (int cmp = 0; cmp < 9; cmp++)       // V2 component layout, see win_value_at
                    b9[cmp] = v[9 * off + (cmp < 8 ? (cmp >> 1) * 2 * n + 2 * l + (cmp & 1) : 8 * n + l)];
                put(i, jn, b9, false);
                if ((en >> 16) != 0xFFFFu) put(rownode[g * R + (en >> 16)], i, b9, true);
            }
            off += n;
        }
    }
    return 0;
}

int fcvm_matrix_to_coo(fcvm_ctx *c, int64_t *row, int64_t *col, double *val) {
    NEED(c && c->have_mesh, "fcvm_matrix_to_coo: no mesh");
    if (!c->sell_valid && c->wsym.on && c->wsym.valid) return matrix_to_coo_sym(c, row, col, val);
    const Pattern &p = c->pat;
    std::vector<double> v(p.value_count());
    HCHK(hipMemcpyAsync(v.data(), c->d_vals, sizeof(double) * v.size(), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    int64_t k = 0;
    for (int64_t i = 0; i < p.nn; i++)
        for (int64_t q = p.rowptr[i]; q < p.rowptr[i + 1]; q++) {
            const uint64_t off = p.block_offset(p.slotof[i], q - p.rowptr[i]);
            const int64_t cs = p.comp_stride(p.slotof[i], q - p.rowptr[i]);
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) {
                    row[k] = 3 * i + r;
                    col[k] = 3 * (int64_t)p.colidx[q] + cc;
                    val[k] = v[off + (r * 3 + cc) * cs];
                    k++;
                }
        }
    return 0;
}

// main stream waits for a copy into the window storage that is still running on its side stream (refresh_precond)
static int win_join(fcvm_ctx *c) {
    if (c->repack_pending) {
        HCHK(hipStreamWaitEvent(c->stream, c->repack_done, 0));
        c->repack_pending = false;
    }
    return 0;
}

// after an assembly straight into the symmetric window form (k_assemble_lds<.., SYM> wrote the FP64 values, the 16-bit copy and
// the diagonal blocks): the matrix exponent for the fixed-point sums / the 16-bit copy, and the diagonal inverses
static int refresh_precond_direct(fcvm_ctx *c, bool lo) {
    WinDev &w = c->win;
    WinSymDev &ws = c->wsym;
    if (!w.lo_scale && dalloc(&w.lo_scale, 2)) return 1;
    hipLaunchKernelGGL(k_absmax_partial, dim3(1024), dim3(256), 0, c->stream, 9 * c->nn, c->d_dblk, c->d_part_misc);
    hipLaunchKernelGGL(k_lo_scale, dim3(1), dim3(256), 0, c->stream, c->d_part_misc, (int64_t)1024, w.lo_scale);
    ws.valid = true;
    ws.lo_valid = lo;
    w.lo_valid = false;
    hipLaunchKernelGGL(k_diag_invert, dim3(cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_dblk, c->d_dinv);
    hipLaunchKernelGGL(k_dinv_pack6f, dim3(cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_dinv, c->d_dinv6f);
    HCHK(hipGetLastError());
    return 0;
}

static int refresh_precond(fcvm_ctx *c) {   // after every assembly into d_vals
    hipLaunchKernelGGL(k_diag_get, dim3(cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_diagoff, c->d_vals, c->d_dblk);
    if (c->win.on) {
        WinDev &w = c->win;
        const int64_t nrows = w.pat.nwaves * SLICE;
        const bool lo = c->precond_mode >= 1 && c->mg_additive == 2;   // the multiplicative cycle smooths with a 16-bit fixed-point copy (mode 2: the multi-GPU layer's cycle)
        w.lo_valid = false;
        c->wsym.valid = false;
        const bool sym_lo = lo && c->wsym.on && c->wsym_use && c->wsym_lo;   // the 16-bit copy in the symmetric form too
        c->wsym.lo_valid = false;
        if (lo && !w.part_lo && dalloc(&w.part_lo, 2 * w.pat.ngroups)) return 1;
        if (lo && !sym_lo && !w.lo8 && (dalloc(&w.lo8, w.pat.nblocks) || dalloc(&w.lo2, w.pat.nblocks))) return 1;
        if (sym_lo && !c->wsym.lo8 && (dalloc(&c->wsym.lo8, c->wsym.pat.nblocks) || dalloc(&c->wsym.lo2, c->wsym.pat.nblocks))) return 1;
        if (lo || c->wsym.on) {   // exponent of the largest diagonal entry: scale of the 16-bit copy and of the fixed-point sums
            if (!w.lo_scale && dalloc(&w.lo_scale, 2)) return 1;
            hipLaunchKernelGGL(k_absmax_partial, dim3(1024), dim3(256), 0, c->stream, 9 * c->nn, c->d_dblk, c->d_part_misc);
            hipLaunchKernelGGL(k_lo_scale, dim3(1), dim3(256), 0, c->stream, c->d_part_misc, (int64_t)1024, w.lo_scale);
        }
        // the copy into the SpMV's storage (1.5 ms at 1 M elements, bandwidth-bound) runs on a stream of its own next to what
        // follows on the main stream -- diagonal inverses and, with the multigrid preconditioner, the tet4 level's assembly and the
        // eigenvalue estimates, none of which reads the window storage; win_join() makes the main stream wait for it
        const bool sym_now = c->wsym.on && c->wsym_use;
        if (!sym_now && win_values_ensure(c)) return 1;      // (zero-filled on the main stream: before the copy stream forks off)
        hipStream_t rs = c->stream;
        if (c->repack_overlap) {
            if (!c->repack_stream) {
                HCHK(hipStreamCreateWithFlags(&c->repack_stream, hipStreamNonBlocking));
                HCHK(hipEventCreateWithFlags(&c->repack_fork, hipEventDisableTiming));
                HCHK(hipEventCreateWithFlags(&c->repack_done, hipEventDisableTiming));
            }
            HCHK(hipEventRecord(c->repack_fork, c->stream));
            HCHK(hipStreamWaitEvent(c->repack_stream, c->repack_fork, 0));
            rs = c->repack_stream;
        }
        // with the symmetric form in charge of the FP64 products nobody reads the full FP64 window copy: only its 16-bit copy is written
        if (!sym_now || (lo && !sym_lo))
        hipLaunchKernelGGL(k_repack_win, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, rs, nrows, w.srcslot, w.rowlen,
                           w.jstart, w.woff, c->d_sptr, c->d_vals, sym_now ? (double *)nullptr : (double *)w.vals, c->win_v2, lo ? w.lo_scale : (const double *)nullptr,
                           lo ? w.lo8 : (short8 *)nullptr, lo ? w.lo2 : (WinLoTail *)nullptr, (const uint16_t *)w.cols);
        if (c->wsym.on && c->wsym_use) {   // (an SPD stiffness matrix: fcvm_assemble sets wsym_use, the buckling operators do not)
            WinSymDev &ws = c->wsym;
            hipLaunchKernelGGL(k_repack_sym, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, rs, nrows, ws.srcslot, ws.rowlen, ws.woff, ws.jell,
                               c->d_sptr, c->d_vals, ws.vals, sym_lo ? w.lo_scale : (const double *)nullptr, sym_lo ? ws.lo8 : (short8 *)nullptr,
                               sym_lo ? ws.lo2 : (WinLoTailS *)nullptr, (const uint32_t *)ws.cols);
            ws.valid = true;
            ws.lo_valid = sym_lo;
        }
        if (c->repack_overlap) {
            HCHK(hipEventRecord(c->repack_done, rs));
            c->repack_pending = true;
        }
        w.lo_valid = lo && !sym_lo;      // (the full-form 16-bit copy was written)
    }
    hipLaunchKernelGGL(k_diag_invert, dim3(cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_dblk, c->d_dinv);
    hipLaunchKernelGGL(k_dinv_pack6f, dim3(cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_dinv, c->d_dinv6f);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_assemble(fcvm_ctx *c, const double *disp, const double *sig, const uint8_t *pgp, double Et_E,
                  const double grav[3], double *glv_grav, double *modf, double *xgp, double *volume) {
    NEED(c && c->have_mesh && c->have_mat, "fcvm_assemble: mesh and material must be set");
    NEED(glv_grav && modf, "fcvm_assemble: glv_grav and modf are required");
    NEED((sig == nullptr) == (pgp == nullptr), "fcvm_assemble: sig and pgp go together");
    if (wsym_finish(c)) return 1;
    hipStream_t s = c->stream;
    const bool use_lds = c->assemble_lds && c->asm_lds_ok;
    if (!use_lds) HCHK(hipMemsetAsync(c->d_vals, 0, sizeof(double) * c->pat.value_count(), s));   // the LDS kernel writes every value once
    HCHK(hipMemsetAsync(glv_grav, 0, sizeof(double) * 3 * c->nn, s));
    HCHK(hipMemsetAsync(modf, 0, sizeof(double) * 3 * c->nn, s));
    HCHK(hipMemsetAsync(c->d_out8, 0, sizeof(double), s));
    MatConst mc = matconst(c, Et_E);
    dim3 block(256);
    const int64_t nslots = c->pat.nslices * SLICE;
    dim3 rgrid((unsigned)cdiv(nslots, 256));
    const bool level1_early = mg_level1_early(c, sig ? disp : nullptr, sig, pgp, Et_E);   // a refresh: beside the fine assembly
    // round 4: straight into the SpMV's storage (symmetric window form + its 16-bit copy + diagonal blocks) where that form is
    // in charge -- no sliced-ELL array, no copy kernel
    const bool lo = c->precond_mode >= 1 && c->mg_additive == 2;
    // (cycle form 0 smooths with Chebyshev steps fused into the sliced-ELL SpMV: it needs that array)
    const bool direct = use_lds && c->asm_sym && c->win.on && c->wsym.on && c->wsym.asm_ok && (!lo || c->wsym_lo) && !c->wsym_graded &&
                        !(c->precond_mode >= 1 && c->mg_additive == 0);
    if (direct) {
        WinSymDev &ws = c->wsym;
        if (lo && !c->win.part_lo && dalloc(&c->win.part_lo, 2 * c->win.pat.ngroups)) return 1;
        if (lo && !ws.lo8) {
            if (dalloc(&ws.lo8, ws.pat.nblocks) || dalloc(&ws.lo2, ws.pat.nblocks)) return 1;
            hipLaunchKernelGGL(k_lo2_static, dim3((unsigned)cdiv(ws.pat.nblocks, 256)), dim3(256), 0, s, ws.pat.nblocks, ws.cols, ws.lo2);
        }
        if (sig ? launch_assemble_sym<true>(c, disp, sig, pgp, mc, grav, glv_grav, modf, xgp, lo)
                : launch_assemble_sym<false>(c, disp, nullptr, nullptr, mc, grav, glv_grav, modf, xgp, lo))
            return 1;
        hipLaunchKernelGGL(k_sum_final_wide, dim3(1), dim3(1024), 0, s, ws.part_vol, ws.nitems, c->d_out8);
        HCHK(hipGetLastError());
        c->sell_valid = false;
    } else if (use_lds) {  // quarter slices accumulated in LDS, written once (fcvm_assemble_lds.hip.h)
        if (sig ? launch_assemble_lds<true>(c, disp, sig, pgp, mc, grav, glv_grav, modf, xgp, c->d_part_vol, 0)
                : launch_assemble_lds<false>(c, disp, nullptr, nullptr, mc, grav, glv_grav, modf, xgp, c->d_part_vol, 0))
            return 1;
        hipLaunchKernelGGL(k_sum_final_wide, dim3(1), dim3(1024), 0, s, c->d_part_vol, 4 * c->pat.nslices, c->d_out8);
    } else {  // one lane per matrix row, global read-modify-writes, no atomics, bitwise reproducible
        if (sig)
            hipLaunchKernelGGL(k_assemble_rows<true>, rgrid, block, 0, s, nslots, c->d_rownode, c->d_nptr, c->d_nel, c->d_el, c->d_coord,
                               disp, c->d_fixmask, c->d_fixval, c->d_emap, sig, pgp, mc, grav[0], grav[1], grav[2], c->d_vals, glv_grav,
                               modf, xgp, c->d_part_vol, 0);
        else
            hipLaunchKernelGGL(k_assemble_rows<false>, rgrid, block, 0, s, nslots, c->d_rownode, c->d_nptr, c->d_nel, c->d_el, c->d_coord,
                               disp, c->d_fixmask, c->d_fixval, c->d_emap, (const double *)nullptr, (const uint8_t *)nullptr, mc,
                               grav[0], grav[1], grav[2], c->d_vals, glv_grav, modf, xgp, c->d_part_vol, 0);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, s, c->d_part_vol, c->pat.nslices, c->d_out8);
    }
    HCHK(hipGetLastError());
    SetupLap lap(s);
    lap("assemble: kernels");
    c->wsym_use = !c->wsym_graded;
    if (direct) {
        if (refresh_precond_direct(c, lo)) return 1;
    } else {
        c->sell_valid = true;
        if (refresh_precond(c)) return 1;
    }
    const bool check_grading = c->wsym.on && !c->wsym_graded;
    if (check_grading) {      // read back behind the synchronisation this call ends with anyway
        if (!c->d_grading && dalloc(&c->d_grading, 2)) return 1;
        const int gb = (int)std::min<int64_t>(std::max<int64_t>(cdiv(c->nn, 1024), 1), 1024);     // (d_part_misc holds 2048 partials)
        hipLaunchKernelGGL(k_diag_grading_partial, dim3((unsigned)gb), dim3(256), 0, s, c->nn, c->d_dblk, c->d_fixmask, c->d_part_misc);
        hipLaunchKernelGGL(k_diag_grading_final, dim3(1), dim3(256), 0, s, c->d_part_misc, gb, c->d_grading);
        HCHK(hipMemcpyAsync(c->grading, c->d_grading, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    lap("assemble: diagonal blocks, window-format copy");
    if (c->precond_mode >= 1) {
        HCHK(hipStreamSynchronize(s));
        if (mg_setup(c, sig ? disp : nullptr, sig, pgp, Et_E, level1_early)) return 1;
    }
    lap("assemble: multigrid set-up (total)");
    if (volume) {
        HCHK(hipMemcpyAsync(volume, c->d_out8, sizeof(double), hipMemcpyDeviceToHost, s));
    }
    if (win_join(c)) return 1;
    HCHK(hipStreamSynchronize(s));
    c->assembled = true;
    if (!c->placement_tuned && c->wsym.on && c->wsym.valid && !(check_grading && c->grading[1] > 0.0 && c->grading[0] > c->grading_max * c->grading[1]) &&
        tune_sym_placement(c))
        return 1;
    if (check_grading && c->grading[1] > 0.0 && c->grading[0] > c->grading_max * c->grading[1]) {
        // a mesh graded beyond what the fixed-point sums resolve: this context keeps to the full window form from here on; the matrix
        // just assembled went (or may have gone) into the symmetric form, so it is assembled once more along the other route
        c->wsym_graded = true;
        if (getenv("FCVM_VERBOSE"))
            fprintf(stderr, "fcvm: nodal diagonal entries span %.3g .. %.3g (ratio %.3g > %.3g): symmetric window form declined, full form in charge\n",
                    c->grading[1], c->grading[0], c->grading[0] / c->grading[1], c->grading_max);
        return fcvm_assemble(c, disp, sig, pgp, Et_E, grav, glv_grav, modf, xgp, volume);
    }
    return 0;
}

int fcvm_update_stress_load(fcvm_ctx *c, const double *sig_yield, const double *disp_new, const double *du,
                            const double *sig_old, double *sig_new, double *sig_test, double *qin, double Et_E,
                            int LD, uint8_t *pgp) {
    NEED(c && c->have_mesh && c->have_mat, "fcvm_update_stress_load: mesh and material must be set");
    NEED(sig_yield && du && sig_old && sig_new && sig_test && qin && pgp, "fcvm_update_stress_load: null array");
    NEED(!LD || disp_new, "fcvm_update_stress_load: disp_new required when LD");
    hipStream_t s = c->stream;
    MatConst mc = matconst(c, Et_E);
    dim3 grid((unsigned)cdiv(c->ne * 4, 256)), block(256);
    if (!c->qin_atomic) {   // per-element nodal forces + fixed-order gather per node: no atomics, reproducible residuals
        if (!c->d_fe && dalloc(&c->d_fe, c->ne * 30)) return 1;
        if (LD)
            hipLaunchKernelGGL((k_update_stress<true, true>), grid, block, 0, s, c->ne, c->d_el, c->d_coord, disp_new, du, sig_old,
                               sig_new, sig_test, sig_yield, pgp, qin, mc, c->d_fe);
        else
            hipLaunchKernelGGL((k_update_stress<false, true>), grid, block, 0, s, c->ne, c->d_el, c->d_coord, disp_new, du, sig_old,
                               sig_new, sig_test, sig_yield, pgp, qin, mc, c->d_fe);
        hipLaunchKernelGGL(k_gather_qin, dim3((unsigned)cdiv(c->nn, 256)), block, 0, s, c->nn, c->d_nptr, c->d_nel, c->d_fe, qin);
        HCHK(hipGetLastError());
        return 0;
    }
    HCHK(hipMemsetAsync(qin, 0, sizeof(double) * 3 * c->nn, s));
    if (LD)
        hipLaunchKernelGGL((k_update_stress<true, false>), grid, block, 0, s, c->ne, c->d_el, c->d_coord, disp_new, du, sig_old,
                           sig_new, sig_test, sig_yield, pgp, qin, mc, (double *)nullptr);
    else
        hipLaunchKernelGGL((k_update_stress<false, false>), grid, block, 0, s, c->ne, c->d_el, c->d_coord, disp_new, du,
                           sig_old, sig_new, sig_test, sig_yield, pgp, qin, mc, (double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_update_peeq_csr(fcvm_ctx *c, const double *sig_test, const double *sig_new, double *sig_yield,
                         double ultimate_strain, double *peeq, double *csr, double *triax, double *pressure,
                         double *sigmises, double *ecr, double Et_E, double out[8]) {
    NEED(c && c->have_mesh && c->have_mat, "fcvm_update_peeq_csr: mesh and material must be set");
    MatConst mc = matconst(c, Et_E);
    if (ultimate_strain == 0.0) ultimate_strain = 1.0e12;
    const double alpha = std::sqrt(std::exp(1.0)) * ultimate_strain;
    const int64_t ngp = 4 * c->ne, nb = c->nb_gp;
    double *pmaxc = c->d_part_misc, *pmaxp = c->d_part_misc + nb;
    hipLaunchKernelGGL(k_peeq_csr, dim3((unsigned)nb), dim3(256), 0, c->stream, ngp, sig_test, sig_new, sig_yield, peeq,
                       csr, triax, pressure, sigmises, ecr, mc.G, mc.H, mc.Et, alpha, pmaxc, c->d_pidx, pmaxp);
    hipLaunchKernelGGL(k_peeq_csr_final, dim3(1), dim3(256), 0, c->stream, nb, ngp, pmaxc, c->d_pidx, pmaxp, pressure,
                       sigmises, triax, ecr, peeq, c->d_out8);
    HCHK(hipGetLastError());
    HCHK(hipMemcpyAsync(out, c->d_out8, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_map_stresses(fcvm_ctx *c, int averaged, const double *sig, const double *peeq, const double *sigvm,
                      const double *csr, double sig_yield, double *t10s, double *t10peeq, double *t10csr,
                      double *t10svm, double *t10triax) {
    NEED(c && c->have_mesh, "fcvm_map_stresses: no mesh");
    hipStream_t s = c->stream;
    HCHK(hipMemsetAsync(t10s, 0, sizeof(double) * 6 * c->nn, s));
    HCHK(hipMemsetAsync(t10peeq, 0, sizeof(double) * c->nn, s));
    HCHK(hipMemsetAsync(t10csr, 0, sizeof(double) * c->nn, s));
    HCHK(hipMemsetAsync(t10svm, 0, sizeof(double) * c->nn, s));
    HCHK(hipMemsetAsync(t10triax, 0, sizeof(double) * c->nn, s));
    hipLaunchKernelGGL(k_map_corners, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, s, c->nn, c->d_nptr, c->d_nel, c->d_noce,
                       averaged, sig, peeq, sigvm, csr, sig_yield, t10s, t10peeq, t10csr, t10svm, t10triax);
    hipLaunchKernelGGL(k_map_midside, dim3((unsigned)cdiv(6 * c->ne, 256)), dim3(256), 0, s, c->ne, c->d_el, t10s,
                       t10peeq, t10csr, t10svm, t10triax);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_principal_stress(fcvm_ctx *c, int64_t n, const double *t10s, double *ps, double *pv) {
    NEED(c && n >= 0, "fcvm_principal_stress: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_principal, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, t10s, ps, pv);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_spmv(fcvm_ctx *c, const double *x, double *y) {
    NEED(c && c->have_mesh, "fcvm_spmv: no mesh");
    NEED(x != y, "fcvm_spmv: x and y must not alias");
    NEED(aligned16(x) && aligned16(y), "fcvm_spmv: device vectors must be 16-byte aligned");
    launch_spmv(c, false, x, y, nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_precond(fcvm_ctx *c, const double *r, double *z) {
    NEED(c && c->have_mesh, "fcvm_precond: no mesh");
    hipLaunchKernelGGL(k_precond, dim3((unsigned)c->nb_vec), dim3(256), 0, c->stream, c->nn, c->d_dinv, r, z);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_get_diag_blocks(fcvm_ctx *c, double *dblk) {
    NEED(c && c->have_mesh, "fcvm_get_diag_blocks: no mesh");
    if (!c->sell_valid && c->wsym.on && c->wsym.valid) {      // assembled straight into the symmetric form: the kernel left the blocks in d_dblk
        HCHK(hipMemcpyAsync(dblk, c->d_dblk, sizeof(double) * 9 * c->nn, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    }
    hipLaunchKernelGGL(k_diag_get, dim3((unsigned)c->nb_vec), dim3(256), 0, c->stream, c->nn, c->d_diagoff, c->d_vals, dblk);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_set_diag_blocks(fcvm_ctx *c, const double *dblk) {
    NEED(c && c->have_mesh, "fcvm_set_diag_blocks: no mesh");
    hipLaunchKernelGGL(k_diag_invert, dim3((unsigned)c->nb_vec), dim3(256), 0, c->stream, c->nn, dblk, c->d_dinv);
    hipLaunchKernelGGL(k_dinv_pack6f, dim3((unsigned)c->nb_vec), dim3(256), 0, c->stream, c->nn, c->d_dinv, c->d_dinv6f);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_solve(fcvm_ctx *c, const double *f, double *u, double rtol, int64_t maxit, double info[3]) {
    NEED(c && c->have_mesh && c->assembled, "fcvm_solve: assemble the matrix first");
    NEED(f && u && f != u, "fcvm_solve: f and u must be distinct device arrays");
    NEED(aligned16(f) && aligned16(u), "fcvm_solve: device vectors must be 16-byte aligned");
    if (c->precond_mode == 1) return mg_pcg(c, f, u, rtol, maxit, info);
    NEED(c->precond_mode == 0, "fcvm_solve: in hierarchy-only mode the solve is driven by the host (multi-GPU)");
    hipStream_t s = c->stream;
    const int64_t nn = c->nn, nbv = c->nb_vec;
    const int CHUNK = 25;
    if (c->evpool.size() < (size_t)(2 * CHUNK + 2)) {
        while (c->evpool.size() < (size_t)(2 * CHUNK + 2)) {
            hipEvent_t e;
            HCHK(hipEventCreate(&e));
            c->evpool.push_back(e);
        }
    }
    hipEvent_t ev0 = c->evpool[2 * CHUNK], ev1 = c->evpool[2 * CHUNK + 1];
    HCHK(hipEventRecord(ev0, s));
    hipLaunchKernelGGL(k_pcg_init, dim3((unsigned)nbv), dim3(256), 0, s, nn, c->d_dinv, f, u, c->d_r, c->d_z, c->d_p,
                       c->d_part_vec, nbv);
    hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, 0, c->d_part_vec, nbv, c->nb_spmv, c->d_S, rtol);
    double S[S_COUNT];
    int64_t it = 0, nspmv = 0;
    double spmv_ms = 0.0;
    bool done = false;
    while (!done && it < maxit) {
        int k = (int)std::min<int64_t>(CHUNK, maxit - it);
        for (int j = 0; j < k; j++) {
            HCHK(hipEventRecord(c->evpool[2 * j], s));
            launch_spmv(c, true, c->d_p, c->d_q, c->d_S + S_DONE);
            HCHK(hipEventRecord(c->evpool[2 * j + 1], s));
            hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, 1, c->d_part_spmv, nbv, spmv_partials(c), c->d_S, rtol);
            hipLaunchKernelGGL(k_pcg_update_xr, dim3((unsigned)nbv), dim3(256), 0, s, nn, c->d_dinv, c->d_p, c->d_q, u,
                               c->d_r, c->d_z, c->d_part_vec, nbv, c->d_S);
            hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, 2, c->d_part_vec, nbv, c->nb_spmv, c->d_S, rtol);
            hipLaunchKernelGGL(k_pcg_update_p, dim3((unsigned)cdiv(3 * nn, 256)), dim3(256), 0, s, 3 * nn, c->d_z, c->d_p, c->d_S);
        }
        HCHK(hipGetLastError());
        HCHK(hipMemcpyAsync(S, c->d_S, sizeof(S), hipMemcpyDeviceToHost, s));
        HCHK(hipStreamSynchronize(s));
        int64_t it_new = (int64_t)S[S_ITERS];
        int64_t ran = it_new - it;  // SpMV launches that did real work in this chunk
        for (int j = 0; j < ran && j < k; j++) {
            float ms = 0.f;
            HCHK(hipEventElapsedTime(&ms, c->evpool[2 * j], c->evpool[2 * j + 1]));
            spmv_ms += ms;
        }
        nspmv += ran;
        it = it_new;
        done = S[S_DONE] != 0.0;
        if (ran < k && !done) break;  // defensive: no progress
        if (!std::isfinite(S[S_RR])) return fail("fcvm_solve: PCG broke down (non-finite residual; matrix not SPD?)");
    }
    HCHK(hipEventRecord(ev1, s));
    HCHK(hipEventSynchronize(ev1));
    float tot = 0.f;
    HCHK(hipEventElapsedTime(&tot, ev0, ev1));
    c->last_spmv_ms = spmv_ms;
    c->last_spmv_timed = (double)nspmv;
    c->last_total_ms = tot;
    if (info) {
        info[0] = (double)it;
        info[1] = (S[S_BB] > 0.0) ? std::sqrt(S[S_RR] / S[S_BB]) : 0.0;
        info[2] = (double)nspmv;
    }
    if (!done) return fail("fcvm_solve: PCG did not converge in " + std::to_string(maxit) + " iterations (rel. residual " +
                           std::to_string(info ? info[1] : -1.0) + ")");
    return 0;
}

/* K uA = fA and K uB = fB in lockstep: two multigrid PCG iterations that share every pass over the matrix (cycle form 2 on the
 * window format; otherwise, or when twice the window does not fit the LDS, the two systems are solved one after the other).
 * Each result is bitwise what fcvm_solve returns for that right-hand side.  info[6] = {iterations, relative residual,
 * SpMV launches} of A, then of B. */
int fcvm_solve2(fcvm_ctx *c, const double *fA, const double *fB, double *uA, double *uB, double rtolA, double rtolB, int64_t maxit,
                double info[6]) {
    NEED(c && c->have_mesh && c->assembled, "fcvm_solve2: assemble the matrix first");
    NEED(fA && fB && uA && uB && fA != uA && fB != uB && uA != uB, "fcvm_solve2: four distinct device arrays");
    NEED(aligned16(fA) && aligned16(fB) && aligned16(uA) && aligned16(uB), "fcvm_solve2: device vectors must be 16-byte aligned");
    if (c->precond_mode == 1 && mg_use_lo(c) && c->win2_ok) return mg_pcg2(c, fA, fB, uA, uB, rtolA, rtolB, maxit, info);
    if (getenv("FCVM_VERBOSE"))
        fprintf(stderr, "fcvm_solve2: one after the other (preconditioner mode %d, 16-bit cycle %d, two windows fit LDS %d)\n", c->precond_mode,
                (int)mg_use_lo(c), (int)c->win2_ok);
    if (fcvm_solve(c, fA, uA, rtolA, maxit, info)) return 1;
    return fcvm_solve(c, fB, uB, rtolB, maxit, info ? info + 3 : nullptr);
}

/* y = K~ x with the 16-bit block-scaled copy of the matrix the multiplicative cycle smooths with (the FP64 product where
 * no copy exists) */
int fcvm_spmv_lo(fcvm_ctx *c, const double *x, double *y) {
    NEED(c && c->have_mesh, "fcvm_spmv_lo: no mesh");
    NEED(x && y && x != y && aligned16(x) && aligned16(y), "fcvm_spmv_lo: x and y must be distinct 16-byte aligned device vectors");
    if (mg_use_lo(c)) launch_spmv_win_lo<0>(c, x, nullptr, y, 0.0, nullptr);
    else launch_spmv(c, false, x, y, nullptr);
    HCHK(hipGetLastError());
    return 0;
}

/* yA = K xA and yB = K xB with one pass over the matrix (k_spmv_win2); each bitwise what fcvm_spmv gives.  Window format with
 * room for two windows in LDS, otherwise two ordinary launches. */
int fcvm_spmv2(fcvm_ctx *c, const double *xA, const double *xB, double *yA, double *yB) {
    NEED(c && c->have_mesh, "fcvm_spmv2: no mesh");
    NEED(xA && xB && yA && yB && xA != yA && xB != yB && yA != yB && xA != yB && xB != yA, "fcvm_spmv2: x and y must not alias");
    NEED(aligned16(xA) && aligned16(xB) && aligned16(yA) && aligned16(yB), "fcvm_spmv2: device vectors must be 16-byte aligned");
    if (c->win.on && c->win2_ok && c->win.part_lo) {
        launch_spmv_win2(c, xA, xB, yA, yB, c->d_part_spmv, c->win.part_lo, nullptr, nullptr);   // (p.Ap partials: scratch here)
    } else {
        launch_spmv(c, false, xA, yA, nullptr);
        launch_spmv(c, false, xB, yB, nullptr);
    }
    HCHK(hipGetLastError());
    return 0;
}

/* zA = M^-1 rA and zB = M^-1 rB: the preconditioner of fcvm_solve2 applied to two vectors, its two 16-bit matrix passes shared
 * (bitwise fcvm_mg_apply of each).  Falls back to two applications where fcvm_solve2 would. */
int fcvm_mg_apply2(fcvm_ctx *c, const double *rA, const double *rB, double *zA, double *zB) {
    NEED(c && c->have_mesh && c->assembled && c->precond_mode == 1, "fcvm_mg_apply2: assemble with the multigrid preconditioner first");
    NEED(rA && rB && zA && zB && rA != zA && rB != zB && zA != zB, "fcvm_mg_apply2: distinct device vectors");
    NEED(aligned16(rA) && aligned16(rB) && aligned16(zA) && aligned16(zB), "fcvm_mg_apply2: device vectors must be 16-byte aligned");
    c->mg_done = nullptr;
    if (!(mg_use_lo(c) && c->win2_ok)) {
        if (mg_vcycle(c, rA, zA)) return 1;
        return mg_vcycle(c, rB, zB);
    }
    return mg_vcycle2(c, rA, rB, zA, zB);
}

/* z = M^-1 r : one application of the multigrid preconditioner of fcvm_solve (r, z: device vectors, r != z) */
int fcvm_mg_apply(fcvm_ctx *c, const double *r, double *z) {
    NEED(c && c->have_mesh && c->assembled && c->precond_mode == 1, "fcvm_mg_apply: assemble with the multigrid preconditioner first");
    NEED(r && z && r != z && aligned16(r) && aligned16(z), "fcvm_mg_apply: r and z must be distinct 16-byte aligned device vectors");
    c->mg_done = nullptr;
    return mg_vcycle(c, r, z);
}

int fcvm_solve_timing(fcvm_ctx *c, double out_ms[4]) {
    out_ms[0] = c->last_spmv_ms;
    out_ms[1] = c->last_total_ms;
    out_ms[2] = c->last_spmv_timed;
    out_ms[3] = (double)c->mg_spmv1;
    return 0;
}

int fcvm_set_precond(fcvm_ctx *c, int mode, int k0, int k1, double alpha0, double alpha1) {
    NEED(c && mode >= 0 && mode <= 2, "fcvm_set_precond: mode 0 (block-Jacobi), 1 (p-multigrid) or 2 (hierarchy only, host-driven V-cycle)");
    NEED(mode == 0 || (k0 >= 1 && k1 >= 1 && alpha0 > 1.0 && alpha1 > 1.0), "fcvm_set_precond: bad smoother parameters");
    c->precond_mode = mode;
    if (mode >= 1) { c->mg_k0 = k0; c->mg_k1 = k1; c->mg_alpha0 = alpha0; c->mg_alpha1 = alpha1; }
    c->assembled = false;  // the next fcvm_assemble builds what the mode needs
    return 0;
}

int fcvm_precond_info(fcvm_ctx *c, double out[8]) {
    out[0] = c->precond_mode; out[1] = (double)c->nc; out[2] = c->mg_lmax0; out[3] = c->mg_lmax1;
    out[4] = c->mg_topo ? (double)c->pat1.nnzb : 0.0; out[5] = (double)c->mg_spmv0;
    out[6] = c->l2 ? (double)c->nagg : 0.0; out[7] = c->l2 ? (double)c->n2 : 0.0;
    return 0;
}

/* vertex-patch smoother of the fine level (round 5; no reference counterpart -- the reference factorises): out[0] = tables built for this
 * mesh, out[1] = inverses valid for the last assembly (the cycle uses them), out[2] = patches, out[3] = largest patch [unknowns],
 * out[4] = mean patch [unknowns], out[5] = bytes of the FP32 inverses, out[6] = largest eigenvalue of S K, out[7] = ms of the last refresh */
int fcvm_patch_info(fcvm_ctx *c, double out[8]) {
    NEED(c && out, "fcvm_patch_info: bad arguments");
    const PatchDev &P = c->patch;
    out[0] = P.on ? 1.0 : 0.0; out[1] = mg_use_patch(c) ? 1.0 : 0.0; out[2] = (double)P.npatch; out[3] = (double)P.maxn;
    out[4] = P.npatch > 0 ? 3.0 * (double)P.nslots / (double)P.npatch : 0.0; out[5] = 4.0 * (double)P.tri_total; out[6] = P.lmax; out[7] = P.last_refresh_ms;
    return 0;
}
/* test hook: out = S x, the undamped additive Schwarz smoother over the vertex patches of the last assembly (device vectors of 3 nn) */
int fcvm_debug_patch_apply(fcvm_ctx *c, const double *x, double *out) {
    NEED(c && c->have_mesh && x && out, "fcvm_debug_patch_apply: bad arguments");
    NEED(c->patch.on && c->patch.valid, "fcvm_debug_patch_apply: no vertex-patch smoother on this context");
    c->mg_done = nullptr;
    patch_smooth_from_zero(c, x, 1.0, out);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_vec_axpby(fcvm_ctx *c, int64_t n, double a, const double *x, double b, const double *y, double *out) {
    NEED(c && n >= 0 && x && out && (b == 0.0 || y), "fcvm_vec_axpby: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, a, x, b, y ? y : x, out);
    HCHK(hipGetLastError());
    return 0;
}
int fcvm_vec_mul(fcvm_ctx *c, int64_t n, const double *x, const double *y, double *out) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_mul, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, x, y, out);
    HCHK(hipGetLastError());
    return 0;
}
int fcvm_vec_dot(fcvm_ctx *c, int64_t n, const double *x, const double *y, double *result_host) {
    if (dot_common(c, n, nullptr, x, y, c->d_out8 + 8)) return 1;
    HCHK(hipMemcpyAsync(result_host, c->d_out8 + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int fcvm_vec_mdot(fcvm_ctx *c, int64_t n, int k, const double *const *xs, const double *y, double *result_host) {
    NEED(c && n > 0 && k >= 1 && k <= 16 && xs && y && result_host, "fcvm_vec_mdot: 1..16 vectors");
    const int64_t nb = std::min<int64_t>(std::max<int64_t>(cdiv(n, 256), 1), 2048);   // as dot_common
    if (!c->d_part_mdot && dalloc(&c->d_part_mdot, 16 * 2048 + 16)) return 1;
    MdotPtrs p;
    for (int j = 0; j < 16; j++) p.x[j] = j < k ? xs[j] : nullptr;
    double *out = c->d_part_mdot + 16 * 2048;
    hipLaunchKernelGGL(k_mdot_partial, dim3((unsigned)nb), dim3(256), 0, c->stream, n, k, p, y, c->d_part_mdot, nb);
    hipLaunchKernelGGL(k_msum_final, dim3((unsigned)k), dim3(256), 0, c->stream, c->d_part_mdot, nb, out);
    HCHK(hipGetLastError());
    HCHK(hipMemcpyAsync(result_host, out, sizeof(double) * k, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int fcvm_vec_wdot_dev(fcvm_ctx *c, int64_t n, const double *w, const double *x, const double *y, double *result_dev) {
    return dot_common(c, n, w, x, y, result_dev);
}
int fcvm_dist_pcg_scalars(fcvm_ctx *c, int phase, double *S, const double *in) {
    NEED(c && S && in && phase >= 0 && phase <= 3, "fcvm_dist_pcg_scalars: bad arguments");
    hipLaunchKernelGGL(k_dist_pcg_scalars, dim3(1), dim3(1), 0, c->stream, phase, S, in);
    HCHK(hipGetLastError());
    return 0;
}
int fcvm_any_u8(fcvm_ctx *c, int64_t n, const uint8_t *a, int *result_host) {
    NEED(c && n >= 0 && result_host, "fcvm_any_u8: bad arguments");
    *result_host = 0;
    if (n == 0) return 0;
    int *flag = (int *)(c->d_out8 + 12);
    HCHK(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_any_u8, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0, c->stream, n, a, flag);
    HCHK(hipGetLastError());
    HCHK(hipMemcpyAsync(result_host, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_vec_max_node_norm(fcvm_ctx *c, int64_t nnodes, const double *u, double *result_host) {
    NEED(c && c->d_part_misc, "vector reductions need a mesh (work buffers)");
    int64_t nb = std::min<int64_t>(std::max<int64_t>(cdiv(nnodes, 256), 1), 2048);
    hipLaunchKernelGGL(k_max_node_norm_partial, dim3((unsigned)nb), dim3(256), 0, c->stream, nnodes, u, c->d_part_misc);
    hipLaunchKernelGGL(k_max_final, dim3(1), dim3(256), 0, c->stream, c->d_part_misc, nb, c->d_out8 + 8);
    HCHK(hipGetLastError());
    HCHK(hipMemcpyAsync(result_host, c->d_out8 + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_host_window_check(int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord, int64_t max_window, int64_t out[8]) {
    // host-only (no GPU needed): sliced-ELL pattern + window format of the same matrix, and the invariants the kernels rely on
    NEED(ne > 0 && nn > 0 && elNodes && nocoord, "fcvm_host_window_check: empty mesh");
    std::vector<int32_t> el(ne * 10);
    for (int64_t i = 0; i < ne * 10; i++) {
        int64_t v = elNodes[i] - 1;
        NEED(v >= 0 && v < nn, "fcvm_host_window_check: element node number out of range");
        el[i] = (int32_t)v;
    }
    Pattern p;
    WinPattern w;
    bool built = false;
    try {
        build_pattern(ne, nn, el.data(), p, 4, SORT_WINDOW);
        built = build_window_format(p, nocoord, 4, max_window > 0 ? max_window : 65535, w);
    } catch (const std::exception &ex) {
        return fail(std::string("fcvm_host_window_check: ") + ex.what());
    }
    for (int k = 0; k < 8; k++) out[k] = 0;
    out[0] = built ? 1 : 0; out[2] = w.wmax;
    if (!built) return 0;
    int64_t bad = 0, rows = 0, blocks = 0;
    std::vector<int64_t> covered(nn, 0);     // blocks of a node's row covered by its segments
    std::vector<uint8_t> firsts(nn, 0);
    for (int64_t wave = 0; wave < w.nwaves; wave++) {
        const int64_t g = wave / w.waves;
        const int32_t *wl = &w.wlist[w.gptr[g]];
        const int64_t W = w.gptr[g + 1] - w.gptr[g];
        const uint16_t *len = &w.rowlen[wave * SLICE];
        for (int l = 0; l < SLICE; l++) {
            const int64_t at = wave * SLICE + l;
            const int32_t nd = w.rownode[at];
            if (l > 0 && len[l] > len[l - 1]) bad++;                         // lanes sorted by length: prefixes
            if (nd < 0) { if (len[l]) bad++; continue; }
            rows++;
            const int nseg = (int)(w.seg[at] & 255u);
            const int64_t rl = p.rowptr[nd + 1] - p.rowptr[nd];
            if (len[l] > w.split || w.jstart[at] + len[l] > rl) bad++;
            if (nseg > 0) {                                                   // first segment: exactly one per node
                if (firsts[nd]++) bad++;
                if (w.jstart[at] != 0 || nseg != std::max<int64_t>(1, (rl + w.split - 1) / w.split)) bad++;
                if ((int64_t)(w.seg[at] >> 8) + nseg - 1 > w.gextra[g]) bad++;
            } else if ((int64_t)(w.seg[at] >> 8) >= w.gextra[g]) bad++;
            covered[nd] += len[l];
            if (w.srcslot[at] != p.slotof[nd]) bad++;
            if (w.own[at] >= W || wl[w.own[at]] != nd) bad++;
        }
        int64_t off = w.woff[wave];
        for (int64_t j = 0; j < len[0]; j++) {
            int64_t n = 0;
            while (n < SLICE && len[n] > j) n++;
            for (int64_t l = 0; l < n; l++) {
                const int64_t at = wave * SLICE + l;
                const int32_t nd = w.rownode[at];
                const uint16_t cidx = w.cols[off + l];
                if (cidx >= W || wl[cidx] != p.colidx[p.rowptr[nd] + w.jstart[at] + j]) bad++;   // its column through the window
            }
            off += n;
            blocks += n;
        }
        if (off > w.woff[wave + 1] || w.woff[wave + 1] - off >= 16 || w.woff[wave] % 16) bad++;   // waves start on 128-byte lines
    }
    for (int64_t i = 0; i < nn; i++)
        if (firsts[i] != 1 || covered[i] != p.rowptr[i + 1] - p.rowptr[i]) bad++;   // every block of every row exactly once
    for (int64_t g = 0; g < w.ngroups; g++)
        for (int64_t i = w.gptr[g] + 1; i < w.gptr[g + 1]; i++)
            if (w.wlist[i] <= w.wlist[i - 1]) bad++;                          // ascending, unique
    if (rows < nn || blocks != p.nnzb || w.nblocks < p.nnzb || w.nblocks >= p.nnzb + 16 * w.nwaves) bad++;
    out[1] = w.ngroups; out[3] = (int64_t)(w.wmean + 0.5); out[4] = bad; out[5] = w.nblocks; out[6] = p.nnzb; out[7] = p.nslots;
    return 0;
}

int fcvm_host_window_sym_check(int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord, int64_t out[8]) {
    // host-only: the symmetric form of the window format (WinSym) of a mesh and its invariants
    NEED(ne > 0 && nn > 0 && elNodes && nocoord && out, "fcvm_host_window_sym_check: empty mesh");
    std::vector<int32_t> el(ne * 10);
    for (int64_t i = 0; i < ne * 10; i++) {
        int64_t v = elNodes[i] - 1;
        NEED(v >= 0 && v < nn, "fcvm_host_window_sym_check: element node number out of range");
        el[i] = (int32_t)v;
    }
    for (int k = 0; k < 8; k++) out[k] = 0;
    try {
        Pattern p;
        WinPattern w;
        WinSym s;
        build_pattern(ne, nn, el.data(), p, 4, SORT_WINDOW);
        if (!build_window_format(p, nocoord, 4, 65535, w)) return 0;
        if (!build_window_sym(p, w, 4, s)) return 0;
        out[0] = 1; out[1] = s.nblocks; out[2] = s.dropped; out[3] = check_window_sym(p, w, s); out[4] = w.nblocks; out[5] = s.xmax;
        out[6] = p.nnzb; out[7] = w.wmax;
    } catch (const std::exception &ex) {
        return fail(std::string("fcvm_host_window_sym_check: ") + ex.what());
    }
    return 0;
}

int fcvm_check_device_emap(fcvm_ctx *c, int64_t out[2]) {
    // the element map on the device (k_build_emap, or the uploaded host map under FCVM_EMAP_HOST=1) against the host
    // builder's rule: block (a, b) of element e is block lower_bound(columns of row el[e][a], el[e][b]) of that row
    NEED(c && c->have_mesh && out, "fcvm_check_device_emap: no mesh");
    HCHK(hipSetDevice(c->device));
    const Pattern &p = c->pat;
    const int64_t ne = c->ne;
    std::vector<uint32_t> dev(ne * 100);
    HCHK(hipStreamSynchronize(c->stream));
    HCHK(hipMemcpy(dev.data(), c->d_emap, sizeof(uint32_t) * ne * 100, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int64_t e = 0; e < ne; e++)
        for (int a = 0; a < 10; a++) {
            const int32_t i = c->h_el[e * 10 + a];
            const int32_t *rb = p.colidx.data() + p.rowptr[i], *re = p.colidx.data() + p.rowptr[i + 1];
            for (int b = 0; b < 10; b++) {
                const int32_t *at = std::lower_bound(rb, re, c->h_el[e * 10 + b]);
                if (at == re || *at != c->h_el[e * 10 + b] || dev[(e * 10 + a) * 10 + b] != p.pack_offset(p.slotof[i], at - rb)) bad++;
            }
        }
    out[0] = ne * 100;
    out[1] = bad;
    return 0;
}

int fcvm_host_pattern_check(int64_t ne, int64_t nn, const int64_t *elNodes, int64_t out[8]) {
    // host-only (no GPU needed): builds the sliced-ELL pattern and verifies its invariants
    NEED(ne > 0 && nn > 0 && elNodes, "fcvm_host_pattern_check: empty mesh");
    std::vector<int32_t> el(ne * 10);
    for (int64_t i = 0; i < ne * 10; i++) {
        int64_t v = elNodes[i] - 1;
        NEED(v >= 0 && v < nn, "fcvm_host_pattern_check: element node number out of range");
        el[i] = (int32_t)v;
    }
    Pattern p;
    try {
        build_pattern(ne, nn, el.data(), p, 4, SORT_WINDOW);
    } catch (const std::exception &ex) {
        return fail(std::string("fcvm_host_pattern_check: ") + ex.what());
    }
    int64_t bad = 0;
    // every element block must land in its row's slice, on the block whose column entry is the block's column node;
    // pairs and the single last block of odd-length slices must be flagged consistently
    for (int64_t e = 0; e < ne; e++)
        for (int a = 0; a < 10; a++)
            for (int b = 0; b < 10; b++) {
                const uint32_t em = p.emap[e * 100 + a * 10 + b];   // packed: half chunk << 2 | (j & 1) << 1 | single
                const bool single = em & 1u;
                const int64_t unit = em >> 2, sub = (em >> 1) & 1u;
                const int64_t slot = p.slotof[el[e * 10 + a]], s = slot / SLICE, l = slot % SLICE;
                const int64_t j = unit - p.sptr[s] + sub;
                if (j < 0 || j >= p.slice_len(s) || single != p.is_single(s, j) || (single && sub) || (!single && ((unit - p.sptr[s]) & 1))) { bad++; continue; }
                if (p.cols[p.col_index(s, l, j)] != el[e * 10 + b]) bad++;
                if (p.pack_offset(slot, j) != em) bad++;
            }
    for (int64_t i = 0; i < nn; i++) {
        int64_t slot = p.slotof[i];
        if (slot < 0 || p.rownode[slot] != i) bad++;
    }
    out[0] = p.nnzb; out[1] = p.nslots; out[2] = p.nslices; out[3] = p.maxlen; out[4] = bad;
    out[5] = p.value_count(); out[6] = p.sptr[p.nslices]; out[7] = 0;
    return 0;
}

int fcvm_membench(fcvm_ctx *c, int64_t nbytes, double out[2]) {
    NEED(c && nbytes >= (1 << 20), "fcvm_membench: bad arguments");
    hipStream_t s = c->stream;
    const int64_t n2 = nbytes / 16;
    double2 *buf = nullptr;
    double *sink = nullptr;
    if (dalloc(&buf, n2) || dalloc(&sink, 1)) return 1;
    HCHK(hipMemsetAsync(buf, 0, (size_t)n2 * 16, s));
    hipEvent_t e0, e1;
    HCHK(hipEventCreate(&e0));
    HCHK(hipEventCreate(&e1));
    const unsigned grid = 256 * 16;   // 16 workgroups of 4 waves per CU
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, s, n2, buf, sink);
    HCHK(hipEventRecord(e0, s));
    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, s, n2, buf, sink);
    HCHK(hipEventRecord(e1, s));
    HCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HCHK(hipEventElapsedTime(&ms, e0, e1));
    out[0] = (double)n2 * 16.0 / (ms / 10.0 * 1e-3) / 1e9;   // GB/s read
    out[1] = ms / 10.0;
    // the same bytes with 8 B per lane (not timed into out[]: a known byte count for counter calibration, tools/pmc_summary.py)
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(k_stream_read8, dim3(grid), dim3(256), 0, s, 2 * n2, (const double *)buf, sink);
    HCHK(hipStreamSynchronize(s));
    HCHK(hipEventDestroy(e0));
    HCHK(hipEventDestroy(e1));
    dfree(buf);
    dfree(sink);
    return 0;
}

int fcvm_mg_coarse_nodes(fcvm_ctx *c, double *coord1, uint8_t *fixmask1) {
    NEED(c && c->mg_topo, "fcvm_mg_coarse_nodes: assemble with the multigrid preconditioner first");
    HCHK(hipMemcpyAsync(coord1, c->d_coord1, sizeof(double) * 3 * c->nc, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipMemcpyAsync(fixmask1, c->d_fixmask1, c->nc, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_mg_coarse_to_coo(fcvm_ctx *c, int64_t *row, int64_t *col, double *val) {
    NEED(c && c->mg_topo, "fcvm_mg_coarse_to_coo: assemble with the multigrid preconditioner first");
    const Pattern &p = c->pat1;
    std::vector<double> v(p.value_count());
    HCHK(hipMemcpyAsync(v.data(), c->d_vals1, sizeof(double) * v.size(), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    int64_t k = 0;
    for (int64_t i = 0; i < p.nn; i++)
        for (int64_t q = p.rowptr[i]; q < p.rowptr[i + 1]; q++) {
            const uint64_t off = p.block_offset(p.slotof[i], q - p.rowptr[i]);
            const int64_t cs = p.comp_stride(p.slotof[i], q - p.rowptr[i]);
            for (int r = 0; r < 3; r++)
                for (int cc = 0; cc < 3; cc++) {
                    row[k] = 3 * i + r;
                    col[k] = 3 * (int64_t)p.colidx[q] + cc;
                    val[k] = v[off + (r * 3 + cc) * cs];
                    k++;
                }
        }
    return 0;
}

int fcvm_mg_set_level2(fcvm_ctx *c, int64_t nagg, int64_t n2, const int32_t *agg, const int64_t *aoff, const double *Z,
                       const double *A2inv, int k1s, double alpha1s) {
    NEED(c && c->mg_topo, "fcvm_mg_set_level2: assemble with the multigrid preconditioner first");
    NEED(nagg > 0 && n2 > 0 && n2 <= 6 * nagg && k1s >= 1 && alpha1s > 1.0, "fcvm_mg_set_level2: bad sizes");
    const int64_t nc = c->nc;
    std::vector<int64_t> aptr(nagg + 1, 0);
    for (int64_t i = 0; i < nc; i++) {
        NEED(agg[i] >= 0 && agg[i] < nagg, "fcvm_mg_set_level2: aggregate id out of range");
        aptr[agg[i] + 1]++;
    }
    for (int64_t g = 0; g < nagg; g++) {
        aptr[g + 1] += aptr[g];
        NEED(aoff[g + 1] >= aoff[g] && aoff[g + 1] - aoff[g] <= 6, "fcvm_mg_set_level2: at most 6 modes per aggregate");
    }
    NEED(aoff[0] == 0 && aoff[nagg] == n2, "fcvm_mg_set_level2: aoff does not add up to n2");
    std::vector<int32_t> anodes(nc);
    {
        std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
        for (int64_t i = 0; i < nc; i++) anodes[fill[agg[i]]++] = (int32_t)i;
    }
    HCHK(hipStreamSynchronize(c->stream));
    dfree(c->d_agg); dfree(c->d_anodes); dfree(c->d_aoff); dfree(c->d_aptr); dfree(c->d_Z); dfree(c->d_A2inv);
    dfree(c->d_r2); dfree(c->d_e2);
    if (dalloc(&c->d_agg, nc) || dalloc(&c->d_anodes, nc) || dalloc(&c->d_aoff, nagg + 1) || dalloc(&c->d_aptr, nagg + 1) ||
        dalloc(&c->d_Z, nc * 18) || (A2inv && dalloc(&c->d_A2inv, n2 * n2)) || dalloc(&c->d_r2, n2) || dalloc(&c->d_e2, n2))
        return 1;
    hipStream_t s = c->stream;
    HCHK(hipMemcpyAsync(c->d_agg, agg, sizeof(int32_t) * nc, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_anodes, anodes.data(), sizeof(int32_t) * nc, hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_aoff, aoff, sizeof(int64_t) * (nagg + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_aptr, aptr.data(), sizeof(int64_t) * (nagg + 1), hipMemcpyHostToDevice, s));
    HCHK(hipMemcpyAsync(c->d_Z, Z, sizeof(double) * nc * 18, hipMemcpyHostToDevice, s));
    dfree(c->d_A2invf);
    if (A2inv) {
        HCHK(hipMemcpyAsync(c->d_A2inv, A2inv, sizeof(double) * n2 * n2, hipMemcpyHostToDevice, s));
        const int64_t lda = a2_lda(n2);
        if (dalloc(&c->d_A2invf, n2 * lda)) return 1;
        hipLaunchKernelGGL(k_to_f32_padded, dim3((unsigned)cdiv(n2 * lda, 256)), dim3(256), 0, s, n2, lda, c->d_A2inv, c->d_A2invf);
    }
    HCHK(hipGetLastError());
    HCHK(hipStreamSynchronize(s));
    c->nagg = nagg; c->n2 = n2; c->mg_k1s = k1s; c->mg_alpha1s = alpha1s;
    dfree(c->d_symv_part);
    if (A2inv && symv_prepare(c)) return 1;   // (a caller's inverse is applied from its lower triangle)
    c->l2 = true;
    c->l2_has_inverse = A2inv != nullptr;  // without it only the transfer operators exist (distributed aggregate level)
    return 0;
}

}  // extern "C"
#include "fcvm_l2s.inc"
#include "fcvm_level2.inc"
extern "C" {

/* Third multigrid level built inside the library: aggregates (uniform bins of about `target` level-1 nodes), rigid-body
 * modes, A2 = Z^T A1 Z and its dense inverse on the device.  out[0] = aggregates, out[1] = unknowns. */
int fcvm_mg_build_level2(fcvm_ctx *c, int64_t target, int k1s, double alpha1s, int64_t out[2]) {
    NEED(c && c->mg_topo, "fcvm_mg_build_level2: assemble with the multigrid preconditioner first");
    NEED(target >= 1 && k1s >= 1 && alpha1s > 1.0, "fcvm_mg_build_level2: bad parameters");
    HCHK(hipSetDevice(c->device));
    if (mg_build_level2(c, target, k1s, alpha1s, 1e-6)) return 1;
    if (out) { out[0] = c->nagg; out[1] = c->n2; }
    return 0;
}

/* sparse aggregate level (meshes with more aggregates than a dense inverse carries): out[0] = 1 if in use, out[1] = 6 x 6 blocks of
 * A2, out[2] = super-aggregates, out[3] = unknowns of the dense top level */
int fcvm_mg_sparse_info(fcvm_ctx *c, int64_t out[4]) {
    NEED(c && out, "fcvm_mg_sparse_info: bad arguments");
    out[0] = c->l2 && c->l2s.on ? 1 : 0; out[1] = c->l2s.nslots; out[2] = c->l2s.nsuper; out[3] = c->l2s.n3;
    return 0;
}

/* ---- aggregate level of a PARTITIONED mesh (fcvm_hip/dist.py DistMG): every rank holds the whole level, bit for bit ----
 * The host layer numbers the aggregates globally (bins over the global bounding box), forms Z from Gram matrices summed over the ranks
 * and hands both over with fcvm_mg_set_level2(..., aoff[g] = 6 g, A2inv = NULL).  Then:
 *   fcvm_mg_l2_galerkin      this rank's share of A2 = Z^T A1 Z as 6 x 6 blocks (k_galerkin_blocks on the rank's level-1 matrix);
 *                            *nslots = number of blocks
 *   fcvm_mg_l2_galerkin_get  sptr_g int64[nagg + 1] (blocks of aggregate g: sptr_g[g] .. sptr_g[g + 1]), slot_agg int32[nslots] (column
 *                            aggregate, ascending within a row), B f64[nslots][36]
 *   (host layer: union of the ranks' block patterns, ONE all-reduce of the block values -- a few 10 MB, where the dense matrix of
 *    rounds 1-4 was n2^2 doubles)
 *   fcvm_mg_l2_finish        from the summed blocks in the global pattern: nsuper == 0 -> dense inverse of the replicated matrix (n2 = 6
 *                            nagg, applied from its lower triangle by every rank: no gather of the correction); nsuper > 0 -> the sparse
 *                            level of fcvm_l2s.inc with the caller's super-aggregates (sagg, soff, C as in l2s_finish).
 * Identical input on every rank gives identical device state, so fcvm_mg_l2_solve returns the same correction everywhere. */
int fcvm_mg_l2_galerkin(fcvm_ctx *c, int64_t *nslots) {
    NEED(c && c->mg_topo && c->l2 && nslots, "fcvm_mg_l2_galerkin: fcvm_mg_set_level2 first");
    HCHK(hipSetDevice(c->device));
    const int64_t nc = c->nc, nagg = c->nagg;
    std::vector<int32_t> agg(nc), anodes(nc);
    std::vector<int64_t> aptr(nagg + 1);
    HCHK(hipMemcpyAsync(agg.data(), c->d_agg, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipMemcpyAsync(anodes.data(), c->d_anodes, sizeof(int32_t) * nc, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipMemcpyAsync(aptr.data(), c->d_aptr, sizeof(int64_t) * (nagg + 1), hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    int nthr = (int)std::thread::hardware_concurrency();
    nthr = std::min(std::max(nthr, 1), 16);
    std::vector<int64_t> eptr;
    std::vector<int32_t> ent_i, ent_j;
    std::vector<uint64_t> ent_loc;
    galerkin_lists(c->pat1, nagg, agg, aptr, anodes, nthr, c->gl_sptr, c->gl_slot, eptr, ent_i, ent_j, ent_loc);
    if (galerkin_blocks_host(c, nagg, c->gl_sptr, eptr, ent_i, ent_j, ent_loc, c->gl_B)) return 1;
    *nslots = (int64_t)c->gl_slot.size();
    return 0;
}

int fcvm_mg_l2_galerkin_get(fcvm_ctx *c, int64_t *sptr_g, int32_t *slot_agg, double *B) {
    NEED(c && c->l2 && (int64_t)c->gl_sptr.size() == c->nagg + 1 && sptr_g && slot_agg && B, "fcvm_mg_l2_galerkin_get: fcvm_mg_l2_galerkin first");
    std::copy(c->gl_sptr.begin(), c->gl_sptr.end(), sptr_g);
    std::copy(c->gl_slot.begin(), c->gl_slot.end(), slot_agg);
    std::copy(c->gl_B.begin(), c->gl_B.end(), B);
    c->gl_sptr = std::vector<int64_t>(); c->gl_slot = std::vector<int32_t>(); c->gl_B = std::vector<double>();
    return 0;
}

int fcvm_mg_l2_finish(fcvm_ctx *c, int64_t nagg, const int32_t *arank, int64_t nslots, const int64_t *sptr_g, const int32_t *slot_agg,
                      const double *B, int64_t nsuper, const int32_t *sagg, const int64_t *soff, const double *C, int k2, int ncoarse) {
    NEED(c && c->mg_topo && c->l2 && nagg == c->nagg && c->n2 == 6 * nagg, "fcvm_mg_l2_finish: fcvm_mg_set_level2 with six modes per aggregate first");
    NEED(arank && nslots > 0 && sptr_g && slot_agg && B && sptr_g[0] == 0 && sptr_g[nagg] == nslots, "fcvm_mg_l2_finish: bad block lists");
    NEED(nsuper == 0 || (sagg && soff && C && soff[0] == 0), "fcvm_mg_l2_finish: super-aggregates without their tables");
    HCHK(hipSetDevice(c->device));
    std::vector<int64_t> vs(sptr_g, sptr_g + nagg + 1);
    std::vector<int32_t> va(slot_agg, slot_agg + nslots);
    std::vector<double> vB(B, B + nslots * 36);
    std::vector<int> vr(nagg);
    for (int64_t g = 0; g < nagg; g++) {
        NEED(arank[g] >= 0 && arank[g] <= 6 && vs[g + 1] >= vs[g], "fcvm_mg_l2_finish: bad rank / row pointer");
        vr[g] = arank[g];
    }
    for (int64_t sl = 0; sl < nslots; sl++) NEED(va[sl] >= 0 && va[sl] < nagg, "fcvm_mg_l2_finish: column aggregate out of range");
    l2s_free(c->l2s);
    dfree(c->d_A2inv); dfree(c->d_A2invf); dfree(c->d_symv_part);
    c->l2_has_inverse = false;
    if (nsuper > 0) {
        std::vector<int32_t> vsa(sagg, sagg + nagg);
        std::vector<int64_t> vso(soff, soff + nsuper + 1);
        std::vector<double> vC(C, C + nagg * 36);
        for (int64_t S = 0; S < nsuper; S++) NEED(vso[S + 1] >= vso[S] && vso[S + 1] - vso[S] <= 6, "fcvm_mg_l2_finish: at most 6 modes per super-aggregate");
        if (l2s_finish(c, nagg, vs, va, vB, vr, nsuper, vsa, vso, vC)) return 1;
        // smoothing steps per sweep / super-aggregate corrections per cycle of the caller (0: the library's defaults; the environment still overrides)
        if (k2 > 0 && !getenv("FCVM_L2S_K")) c->l2s.k2 = k2;
        if (ncoarse > 0 && !getenv("FCVM_L2S_COARSE")) c->l2s.ncoarse = std::min(4, ncoarse);
        return 0;
    }
    // dense: the blocks scattered into the n2 x n2 matrix on the device, inverted by every rank (identical input, identical result)
    if (l2_symmetrise_blocks(nagg, vs, va, vB, vr, nullptr)) return 1;
    const int64_t n2 = c->n2;
    hipStream_t s = c->stream;
    std::vector<int32_t> srow(nslots);
    for (int64_t g = 0; g < nagg; g++)
        for (int64_t sl = vs[g]; sl < vs[g + 1]; sl++) srow[sl] = (int32_t)g;
    double *d_A2raw = nullptr, *d_B = nullptr;
    int32_t *d_row = nullptr, *d_col = nullptr;
    auto cleanup = [&]() { dfree(d_A2raw); dfree(d_B); dfree(d_row); dfree(d_col); };
    if (dalloc(&d_A2raw, n2 * n2) || dalloc(&d_B, nslots * 36) || dalloc(&d_row, nslots) || dalloc(&d_col, nslots)) { cleanup(); return 1; }
    if (hipMemsetAsync(d_A2raw, 0, sizeof(double) * n2 * n2, s) != hipSuccess ||
        hipMemcpyAsync(d_B, vB.data(), sizeof(double) * vB.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_row, srow.data(), sizeof(int32_t) * nslots, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_col, va.data(), sizeof(int32_t) * nslots, hipMemcpyHostToDevice, s) != hipSuccess) {
        cleanup();
        return fail("fcvm_mg_l2_finish: upload failed");
    }
    hipLaunchKernelGGL(k_scatter_blocks6, dim3((unsigned)cdiv(nslots * 36, 256)), dim3(256), 0, s, nslots, d_row, d_col, d_B, d_A2raw, n2);
    int rc = dalloc(&c->d_A2inv, n2 * n2);
    if (!rc) rc = spd_inverse_device(c, n2, d_A2raw, c->d_A2inv, 1e-13);
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail("fcvm_mg_l2_finish: HIP error");
    cleanup();
    if (rc) return 1;
    const int64_t lda = a2_lda(n2);
    if (dalloc(&c->d_A2invf, n2 * lda)) return 1;
    hipLaunchKernelGGL(k_to_f32_padded, dim3((unsigned)cdiv(n2 * lda, 256)), dim3(256), 0, s, n2, lda, c->d_A2inv, c->d_A2invf);
    HCHK(hipGetLastError());
    HCHK(hipStreamSynchronize(s));
    if (symv_prepare(c)) return 1;
    c->l2_has_inverse = true;
    return 0;
}

/* dst = inverse of the symmetric positive definite n x n matrix src (both row-major on the device, may not alias):
 * blocked Cholesky, triangular inverse and X^T X in HIP kernels (fcvm_dense.hip.h) */
int fcvm_spd_inverse(fcvm_ctx *c, int64_t n, const double *src, double *dst) {
    NEED(c && n > 0 && src && dst && src != dst, "fcvm_spd_inverse: bad arguments");
    HCHK(hipSetDevice(c->device));
    return spd_inverse_device(c, n, src, dst, 0.0);
}

/* ---- fine-grained multigrid building blocks for the host-driven multi-GPU V-cycle ---- */
static int mg_level(fcvm_ctx *c, int level, MatView *A, const double **dinv, int64_t *nnodes) {
    NEED(c && c->have_mesh && (level == 0 || (level == 1 && c->mg_topo)), "multigrid level not available");
    if (level == 0) { *A = fine_view(c); *dinv = c->d_dinv; *nnodes = c->nn; }
    else { *A = coarse_view(c); *dinv = c->d_dinv1; *nnodes = c->nc; }
    return 0;
}

int fcvm_mg_coarse_map(fcvm_ctx *c, int32_t *c2f) {
    NEED(c && c->mg_topo, "fcvm_mg_coarse_map: no multigrid hierarchy");
    HCHK(hipMemcpyAsync(c2f, c->d_c2f, sizeof(int32_t) * c->nc, hipMemcpyDeviceToHost, c->stream));
    HCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fcvm_spmv_level(fcvm_ctx *c, int level, const double *x, double *y) {
    MatView A; const double *dinv; int64_t n;
    if (mg_level(c, level, &A, &dinv, &n)) return 1;
    if (level == 0) launch_spmv(c, false, x, y, nullptr);      // (window format where it is in charge: the sliced-ELL array may be stale)
    else launch_spmv_t<false>(c, A, x, y, nullptr, nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_get_diag_blocks(fcvm_ctx *c, int level, double *dblk) {
    if (level == 0) return fcvm_get_diag_blocks(c, dblk);
    NEED(c && c->mg_topo, "fcvm_mg_get_diag_blocks: no multigrid hierarchy");
    hipLaunchKernelGGL(k_diag_get, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_diagoff1, c->d_vals1, dblk);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_set_diag_blocks(fcvm_ctx *c, int level, const double *dblk) {
    if (level == 0) return fcvm_set_diag_blocks(c, dblk);
    NEED(c && c->mg_topo, "fcvm_mg_set_diag_blocks: no multigrid hierarchy");
    hipLaunchKernelGGL(k_diag_invert, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, dblk, c->d_dinv1);
    hipLaunchKernelGGL(k_dinv_pack6f, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_dinv1, c->d_dinv1_6f);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_cheb_init(fcvm_ctx *c, int level, const double *b, double *x, double *r, double *d, double c0) {
    MatView A; const double *dinv; int64_t n;
    if (mg_level(c, level, &A, &dinv, &n)) return 1;
    hipLaunchKernelGGL(k_cheb_init, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, dinv, b, x, r, d, c0, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_cheb_step(fcvm_ctx *c, int level, const double *t, double *x, double *r, double *d, double cdd, double cdr, int update_d) {
    MatView A; const double *dinv; int64_t n;
    if (mg_level(c, level, &A, &dinv, &n)) return 1;
    hipLaunchKernelGGL(k_cheb_step, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, dinv, t, x, r, d, cdd, cdr, update_d,
                       (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_restrict(fcvm_ctx *c, const double *r0, double *r1) {
    NEED(c && c->mg_topo, "fcvm_mg_restrict: no multigrid hierarchy");
    hipLaunchKernelGGL(k_restrict, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_c2f, c->d_mptr, c->d_mids,
                       c->d_fixmask, r0, r1, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_restrict_weighted(fcvm_ctx *c, const double *r0, const double *wgt, double *r1) {
    NEED(c && c->mg_topo && wgt, "fcvm_mg_restrict_weighted: no multigrid hierarchy / weights");
    hipLaunchKernelGGL(k_restrict, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_c2f, c->d_mptr, c->d_mids,
                       c->d_fixmask, r0, r1, (const double *)nullptr, wgt);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_restrict_diff(fcvm_ctx *c, const double *r0, const double *wgt, const double *sub, double *r1) {
    NEED(c && c->mg_topo && wgt && sub, "fcvm_mg_restrict_diff: no multigrid hierarchy / weights / second vector");
    hipLaunchKernelGGL(k_restrict_diff4, dim3((unsigned)cdiv(4 * c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_c2f, c->d_mptr, c->d_mids,
                       c->d_fixmask, r0, wgt, sub, r1);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_jacobi6f(fcvm_ctx *c, const double *r, double weight, double *z) {
    NEED(c && c->have_mesh && c->d_dinv6f && r && z, "fcvm_jacobi6f: no mesh / diagonal blocks");
    hipLaunchKernelGGL(k_jacobi6f, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, (const float *)c->d_dinv6f, r, weight, z);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_dist_post_smooth(fcvm_ctx *c, const double *x2, const double *b, const double *t, double c0, const double *ow, double *z,
                          double *bz_dev, double *bb_dev) {
    NEED(c && c->have_mesh && c->d_dinv6f && c->d_part_misc, "fcvm_dist_post_smooth: no mesh / diagonal blocks");
    NEED(x2 && b && t && ow && z && bz_dev && bb_dev, "fcvm_dist_post_smooth: null argument");
    const int64_t nb = std::min<int64_t>(std::max<int64_t>(cdiv(c->nn, 256), 1), 1024);     // d_part_misc holds 2048 partials
    hipLaunchKernelGGL(k_dist_post_smooth, dim3((unsigned)nb), dim3(256), 0, c->stream, c->nn, (const float *)c->d_dinv6f, x2, b, t, c0, ow, z,
                       c->d_part_misc);
    hipLaunchKernelGGL(k_sum_final2, dim3(1), dim3(256), 0, c->stream, c->d_part_misc, nb, bz_dev, bb_dev);
    HCHK(hipGetLastError());
    return 0;
}

/* Multi-GPU layer: the local nodes shared with other ranks (0-based, any order).  The groups of the window format that hold a row of such a node
 * form the first of two launch lists, the rest the second: fcvm_spmv_dot_part / fcvm_spmv_lo_part(part = 1) compute every shared entry, the
 * caller hands them to the exchange, and part = 2 computes the interior while the messages travel.  Both parts together are bitwise the unsplit
 * launch (a group's numbers are formed by its own workgroup).  n = 0 removes the lists.  Needs the symmetric window form; without it part 1
 * does everything and part 2 nothing.  out[0] / out[1] (optional): groups in the first / second list. */
int fcvm_set_interface_nodes(fcvm_ctx *c, int64_t n, const int64_t *nodes, int64_t out[2]) {
    NEED(c && c->have_mesh && n >= 0 && (n == 0 || nodes), "fcvm_set_interface_nodes: no mesh / bad arguments");
    if (wsym_finish(c)) return 1;
    WinSymDev &ws = c->wsym;
    HCHK(hipStreamSynchronize(c->stream));
    dfree(ws.glist_b); dfree(ws.glist_i);
    ws.n_gb = ws.n_gi = 0;
    ws.split = false;
    if (out) out[0] = out[1] = 0;
    if (n == 0 || !ws.on) return 0;
    std::vector<uint8_t> shared(c->nn, 0);
    for (int64_t i = 0; i < n; i++) {
        NEED(nodes[i] >= 0 && nodes[i] < c->nn, "fcvm_set_interface_nodes: node out of range");
        shared[nodes[i]] = 1;
    }
    const int64_t ng = c->win.pat.ngroups, rows_per_group = 64 * (int64_t)ws.pat.waves, nrows = ws.pat.nwaves * SLICE;
    std::vector<int32_t> rownode(nrows);
    HCHK(hipMemcpy(rownode.data(), ws.rownode, sizeof(int32_t) * nrows, hipMemcpyDeviceToHost));
    std::vector<int32_t> gb, gi;
    for (int64_t g = 0; g < ng; g++) {
        bool b = false;
        for (int64_t r = g * rows_per_group; r < std::min((g + 1) * rows_per_group, nrows) && !b; r++)
            b = rownode[r] >= 0 && shared[rownode[r]];
        (b ? gb : gi).push_back((int32_t)g);
    }
    if (dalloc(&ws.glist_b, (int64_t)gb.size()) || dalloc(&ws.glist_i, (int64_t)gi.size())) return 1;
    if (!gb.empty()) HCHK(hipMemcpy(ws.glist_b, gb.data(), sizeof(int32_t) * gb.size(), hipMemcpyHostToDevice));
    if (!gi.empty()) HCHK(hipMemcpy(ws.glist_i, gi.data(), sizeof(int32_t) * gi.size(), hipMemcpyHostToDevice));
    ws.n_gb = (int64_t)gb.size(); ws.n_gi = (int64_t)gi.size();
    ws.split = true;
    if (out) { out[0] = ws.n_gb; out[1] = ws.n_gi; }
    return 0;
}

/* y = K_local x over the groups of `part` (0 all, 1 boundary groups, 2 interior groups) and, after part 0 or 2, xy_dev[0] = x . y over the local rows */
int fcvm_spmv_dot_part(fcvm_ctx *c, const double *x, double *y, double *xy_dev, int part) {
    NEED(c && c->have_mesh && xy_dev && part >= 0 && part <= 2, "fcvm_spmv_dot_part: no mesh / bad part");
    NEED(x != y && aligned16(x) && aligned16(y), "fcvm_spmv_dot_part: x and y must be distinct 16-byte aligned device vectors");
    launch_spmv(c, true, x, y, nullptr, part);
    if (part != 1) hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, c->d_part_spmv, spmv_partials(c), xy_dev);
    HCHK(hipGetLastError());
    return 0;
}

/* out = w r - K~_local x on the 16-bit copy (w per dof, e.g. 1 / ranks holding the node): the residual after the pre-smoothing step of the multi-GPU
 * cycle as the epilogue of its smoother pass -- the rank's product stays unsummed (the restriction is linear; the level-1 exchange completes the sum), r
 * enters once over the ranks.  Needs the symmetric form of the 16-bit copy. */
int fcvm_spmv_lo_resid(fcvm_ctx *c, const double *x, const double *r, const double *w, double *out) {
    NEED(c && c->have_mesh && x && r && w && out && x != out, "fcvm_spmv_lo_resid: no mesh / null or aliased vectors");
    NEED(aligned16(x) && aligned16(out), "fcvm_spmv_lo_resid: device vectors must be 16-byte aligned");
    NEED(mg_use_lo(c) && c->wsym.on && c->wsym.lo_valid, "fcvm_spmv_lo_resid: needs the symmetric form of the 16-bit copy (cycle form 2)");
    launch_spmv_win_lo<1>(c, x, r, out, 0.0, nullptr, nullptr, nullptr, 0, w);
    HCHK(hipGetLastError());
    return 0;
}

/* y = K~ x (16-bit copy) over the groups of `part`, as fcvm_spmv_dot_part */
int fcvm_spmv_lo_part(fcvm_ctx *c, const double *x, double *y, int part) {
    NEED(c && c->have_mesh && part >= 0 && part <= 2, "fcvm_spmv_lo_part: no mesh / bad part");
    NEED(x && y && x != y && aligned16(x) && aligned16(y), "fcvm_spmv_lo_part: x and y must be distinct 16-byte aligned device vectors");
    if (mg_use_lo(c)) launch_spmv_win_lo<0>(c, x, nullptr, y, 0.0, nullptr, nullptr, nullptr, part);
    else launch_spmv(c, false, x, y, nullptr, part);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_spmv_dot(fcvm_ctx *c, const double *x, double *y, double *xy_dev) {
    NEED(c && c->have_mesh && xy_dev, "fcvm_spmv_dot: no mesh");
    NEED(x != y, "fcvm_spmv_dot: x and y must not alias");
    NEED(aligned16(x) && aligned16(y), "fcvm_spmv_dot: device vectors must be 16-byte aligned");
    launch_spmv(c, true, x, y, nullptr);
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, c->stream, c->d_part_spmv, spmv_partials(c), xy_dev);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_prolong_add(fcvm_ctx *c, const double *e1, const double *z0, double *z) {
    NEED(c && c->mg_topo, "fcvm_mg_prolong_add: no multigrid hierarchy");
    hipLaunchKernelGGL(k_prolong_add, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_pa, c->d_pb, c->d_fixmask, e1,
                       z0, z);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_pcg_update_xr_z0(fcvm_ctx *c, const double *alpha_dev, const double *p, const double *q, double *x, double *r, double *z0,
                          double weight) {
    NEED(c && c->have_mesh && alpha_dev, "fcvm_pcg_update_xr_z0: no mesh");
    hipLaunchKernelGGL(k_update_xr_z0_ext, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, alpha_dev, p, q, x, r, c->d_dinv,
                       z0, weight);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_pcg_update_xr_z0_f32(fcvm_ctx *c, const double *alpha_dev, const double *p, const double *q, double *x, double *r, double *z0,
                              double weight) {
    NEED(c && c->have_mesh && alpha_dev && c->d_dinv6f, "fcvm_pcg_update_xr_z0_f32: no mesh / diagonal blocks");
    hipLaunchKernelGGL(k_update_xr_z0_ext, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, alpha_dev, p, q, x, r, c->d_dinv,
                       z0, weight, (const float *)c->d_dinv6f);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_prolong(fcvm_ctx *c, const double *e1, double *d0) {
    NEED(c && c->mg_topo, "fcvm_mg_prolong: no multigrid hierarchy");
    hipLaunchKernelGGL(k_prolong, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, c->d_pa, c->d_pb, c->d_fixmask, e1,
                       d0, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_l2_restrict(fcvm_ctx *c, const double *r1, double *r2) {
    NEED(c && c->l2, "fcvm_mg_l2_restrict: no aggregate level (fcvm_mg_set_level2)");
    hipLaunchKernelGGL(k_l2_restrict, dim3((unsigned)cdiv(c->nagg, 4)), dim3(256), 0, c->stream, c->nagg, c->d_aptr, c->d_anodes,
                       c->d_aoff, c->d_Z, r1, r2, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_l2_restrict_weighted(fcvm_ctx *c, const double *r1, const double *wgt, double *r2) {
    NEED(c && c->l2 && wgt, "fcvm_mg_l2_restrict_weighted: no aggregate level (fcvm_mg_set_level2) / weights");
    hipLaunchKernelGGL(k_l2_restrict, dim3((unsigned)cdiv(c->nagg, 4)), dim3(256), 0, c->stream, c->nagg, c->d_aptr, c->d_anodes,
                       c->d_aoff, c->d_Z, r1, r2, (const double *)nullptr, wgt);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_l2_restrict_diff(fcvm_ctx *c, const double *r1, const double *wgt, const double *sub, double *r2) {
    NEED(c && c->l2 && wgt && sub, "fcvm_mg_l2_restrict_diff: no aggregate level (fcvm_mg_set_level2) / weights / second vector");
    hipLaunchKernelGGL(k_l2_restrict, dim3((unsigned)cdiv(c->nagg, 4)), dim3(256), 0, c->stream, c->nagg, c->d_aptr, c->d_anodes,
                       c->d_aoff, c->d_Z, r1, r2, (const double *)nullptr, wgt, sub);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_dist_cg_update(fcvm_ctx *c, const double *S_dev, const double *z, const double *w, double *p, double *s, double *x, double *r,
                        double *z0, double weight) {
    NEED(c && c->have_mesh && c->d_dinv6f && S_dev && z && w && p && s && x && r && z0, "fcvm_dist_cg_update: no mesh / null argument");
    hipLaunchKernelGGL(k_dist_cg_update, dim3((unsigned)cdiv(c->nn, 256)), dim3(256), 0, c->stream, c->nn, S_dev, z, w, p, s, x, r,
                       (const float *)c->d_dinv6f, z0, weight);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_l2_solve(fcvm_ctx *c, const double *r2, double *e2) {
    NEED(c && c->l2 && (c->l2_has_inverse || c->l2s.on) && r2 && e2, "fcvm_mg_l2_solve: no aggregate level with an inverse or a sparse cycle "
                                                                      "(fcvm_mg_set_level2 with an inverse / fcvm_mg_l2_finish)");
    if (c->l2s.on) {
        // sparse aggregate level: block-Jacobi Chebyshev sweeps around the dense super-aggregate corrections (l2s_cycle works on
        // c->d_r2 and leaves the result in its own x)
        double *own = c->d_r2;
        c->d_r2 = const_cast<double *>(r2);
        c->mg_done = nullptr;
        l2s_cycle(c);
        c->d_r2 = own;
        HCHK(hipMemcpyAsync(e2, c->l2s.x, sizeof(double) * 6 * c->l2s.nagg, hipMemcpyDeviceToDevice, c->stream));
        HCHK(hipGetLastError());
        return 0;
    }
    if (c->mg_fp32)
        launch_gemv_f32(c, r2, e2, nullptr);
    else
        hipLaunchKernelGGL(k_gemv<double>, dim3((unsigned)cdiv(c->n2, 4)), dim3(256), 0, c->stream, c->n2, c->d_A2inv, r2, e2, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_gemv_f32(fcvm_ctx *c, int64_t m, int64_t n, int64_t lda, const float *A, const double *x, double *y) {
    NEED(c && m > 0 && n > 0 && lda >= n && lda % 4 == 0 && A && x && y, "fcvm_gemv_f32: bad arguments (lda must be a multiple of 4)");
    hipLaunchKernelGGL(k_gemv_f32x4, dim3((unsigned)cdiv(m, 16)), dim3(256), 0, c->stream, m, n, lda, A, x, y, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_mg_l2_prolong(fcvm_ctx *c, const double *e2, double *d1) {
    NEED(c && c->l2, "fcvm_mg_l2_prolong: no aggregate level (fcvm_mg_set_level2)");
    hipLaunchKernelGGL(k_l2_prolong, dim3((unsigned)cdiv(c->nc, 256)), dim3(256), 0, c->stream, c->nc, c->d_agg, c->d_aoff, c->d_Z, e2,
                       d1, (const double *)nullptr);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_pack_nodes(fcvm_ctx *c, int64_t n, const int64_t *idx, int ncomp, const double *v, double *out) {
    NEED(c && n >= 0 && ncomp > 0, "fcvm_pack_nodes: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)cdiv(n * ncomp, 256)), dim3(256), 0, c->stream, n, idx, ncomp, v, out);
    HCHK(hipGetLastError());
    return 0;
}

/* v[idx[k]][c] += a[k][c], idx distinct (device arrays): see k_scatter_add_nodes */
int fcvm_scatter_add_nodes(fcvm_ctx *c, int64_t n, const int64_t *idx, int ncomp, const double *a, double *v) {
    NEED(c && n >= 0 && ncomp > 0 && (n == 0 || (idx && a && v)), "fcvm_scatter_add_nodes: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scatter_add_nodes, dim3((unsigned)cdiv(n * ncomp, 256)), dim3(256), 0, c->stream, n, idx, ncomp, a, v);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_unpack_sum(fcvm_ctx *c, int64_t nshared, const int64_t *node, const int64_t *eptr, const int64_t *src, int ncomp, double *v,
                    const double *recv) {
    NEED(c && nshared >= 0 && ncomp > 0, "fcvm_unpack_sum: bad arguments");
    if (nshared == 0) return 0;
    hipLaunchKernelGGL(k_unpack_sum, dim3((unsigned)cdiv(nshared * ncomp, 256)), dim3(256), 0, c->stream, nshared, node, eptr, src, ncomp, v, recv);
    HCHK(hipGetLastError());
    return 0;
}

/* Multi-GPU layer, level 1 of the cycle on the library's current kernels (round 5; rounds 1-4 ran the FP32 sliced-ELL product and a separate
 * smoother-step kernel per pass): one pass of the Chebyshev smoother on this rank's 16-bit copy of the tet4 matrix (k_spmv_l1_i16) with the step
 * as epilogue -- t = A1_local din (+ t2: a pending product of this rank);  r -= t;  x += din;  dout = cdd din + cdr Dinv r (update_d);
 * x += dout (add_dout) -- for every row EXCEPT those of nodes flagged in defer (u8 per level-1 node: shared with another rank), whose raw
 * product goes to t_out and whose step fcvm_unpack_sum_cheb applies after the exchange has summed the ranks' parts.  dout != din. */
int fcvm_mg_l1_pass(fcvm_ctx *c, const double *din, double *dout, double *x, double *r, const double *t2, double *t_out, const uint8_t *defer,
                    double cdd, double cdr, int update_d, int add_dout) {
    NEED(c && c->mg_topo && din && dout && din != dout && x && r, "fcvm_mg_l1_pass: no multigrid hierarchy / null or aliased vectors");
    NEED((defer == nullptr) == (t_out == nullptr), "fcvm_mg_l1_pass: defer and t_out go together");
    MatView A = coarse_view(c);
    NEED(A.A16 && c->d_dinv1_6f, "fcvm_mg_l1_pass: needs the 16-bit copy of the level-1 matrix (FCVM_MG_L1_16=1)");
    ChebEpi epi{c->d_dinv1, x, r, dout, cdd, cdr, update_d ? 1 : 0, add_dout ? 1 : 0, c->d_dinv1_6f};
    epi.defer = defer; epi.t_out = t_out; epi.t2 = t2;
    launch_spmv_cheb(c, A, din, nullptr, epi);
    HCHK(hipGetLastError());
    return 0;
}

/* Boundary slices first for the level-1 passes (round 5, as fcvm_set_interface_nodes for the fine passes): nodes = the level-1 (corner) nodes, 0-based
 * in the level-1 numbering, that another rank holds too.  fcvm_mg_l1_pass_part(part = 1) then covers the slices of the 16-bit copy with a row of one of
 * them -- their raw products are complete and can be exchanged -- and part = 2 the others, beside the exchange; parts 1 + 2 are bitwise the unsplit pass
 * (every row is formed by the same lane either way).  out[0] / out[1] = slices in the two lists.  n = 0 drops the lists. */
int fcvm_mg_l1_set_interface(fcvm_ctx *c, int64_t n, const int64_t *nodes, int64_t out[2]) {
    NEED(c && c->mg_topo && n >= 0 && (n == 0 || nodes), "fcvm_mg_l1_set_interface: no multigrid hierarchy / bad arguments");
    HCHK(hipStreamSynchronize(c->stream));
    dfree(c->d_l1_sb); dfree(c->d_l1_si); dfree(c->d_l1_nb); dfree(c->d_l1_ni);
    c->n_l1_sb = c->n_l1_si = c->n_l1_nb = c->n_l1_ni = 0;
    if (out) out[0] = out[1] = 0;
    if (n == 0) return 0;
    std::vector<uint8_t> shared(c->nc, 0);
    for (int64_t i = 0; i < n; i++) {
        NEED(nodes[i] >= 0 && nodes[i] < c->nc, "fcvm_mg_l1_set_interface: node out of range");
        shared[nodes[i]] = 1;
    }
    const int64_t ns = c->pat1.nslices;
    std::vector<int32_t> rownode(ns * SLICE);
    HCHK(hipMemcpy(rownode.data(), c->d_rownode1, sizeof(int32_t) * ns * SLICE, hipMemcpyDeviceToHost));
    std::vector<int32_t> sb, si;
    for (int64_t s = 0; s < ns; s++) {
        bool b = false;
        for (int64_t r = s * SLICE; r < (s + 1) * SLICE && !b; r++) b = rownode[r] >= 0 && shared[rownode[r]];
        (b ? sb : si).push_back((int32_t)s);
    }
    if (dalloc(&c->d_l1_sb, (int64_t)sb.size()) || dalloc(&c->d_l1_si, (int64_t)si.size())) return 1;
    if (!sb.empty()) HCHK(hipMemcpy(c->d_l1_sb, sb.data(), sizeof(int32_t) * sb.size(), hipMemcpyHostToDevice));
    if (!si.empty()) HCHK(hipMemcpy(c->d_l1_si, si.data(), sizeof(int32_t) * si.size(), hipMemcpyHostToDevice));
    c->n_l1_sb = (int64_t)sb.size(); c->n_l1_si = (int64_t)si.size();
    std::vector<int32_t> nb, ni;
    for (int64_t i = 0; i < c->nc; i++) (shared[i] ? nb : ni).push_back((int32_t)i);
    if (dalloc(&c->d_l1_nb, (int64_t)nb.size()) || dalloc(&c->d_l1_ni, (int64_t)ni.size())) return 1;
    if (!nb.empty()) HCHK(hipMemcpy(c->d_l1_nb, nb.data(), sizeof(int32_t) * nb.size(), hipMemcpyHostToDevice));
    if (!ni.empty()) HCHK(hipMemcpy(c->d_l1_ni, ni.data(), sizeof(int32_t) * ni.size(), hipMemcpyHostToDevice));
    c->n_l1_nb = (int64_t)nb.size(); c->n_l1_ni = (int64_t)ni.size();
    if (out) { out[0] = c->n_l1_sb; out[1] = c->n_l1_si; }
    return 0;
}
/* r1 = P^T r0 for the level-1 nodes of `part` (0: all, as fcvm_mg_restrict; 1: the nodes given to fcvm_mg_l1_set_interface; 2: the others): the restricted
 * residual of the shared corner nodes can be exchanged while the rest is still being restricted */
int fcvm_mg_restrict_part(fcvm_ctx *c, const double *r0, double *r1, int part) {
    if (part == 0) return fcvm_mg_restrict(c, r0, r1);
    NEED(c && c->mg_topo && (part == 1 || part == 2) && (c->d_l1_nb || c->d_l1_ni), "fcvm_mg_restrict_part: no multigrid hierarchy / no node lists (fcvm_mg_l1_set_interface)");
    const int64_t n = part == 1 ? c->n_l1_nb : c->n_l1_ni;
    if (n > 0)
        hipLaunchKernelGGL(k_restrict, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, c->stream, n, c->d_c2f, c->d_mptr, c->d_mids, c->d_fixmask, r0, r1,
                           (const double *)nullptr, (const double *)nullptr, part == 1 ? c->d_l1_nb : c->d_l1_ni);
    HCHK(hipGetLastError());
    return 0;
}
/* fcvm_mg_l1_pass over the slices of `part` (0: all, as fcvm_mg_l1_pass; 1: the slices with rows of shared nodes; 2: the others) */
int fcvm_mg_l1_pass_part(fcvm_ctx *c, const double *din, double *dout, double *x, double *r, const double *t2, double *t_out, const uint8_t *defer,
                         double cdd, double cdr, int update_d, int add_dout, int part) {
    if (part == 0) return fcvm_mg_l1_pass(c, din, dout, x, r, t2, t_out, defer, cdd, cdr, update_d, add_dout);
    NEED(c && c->mg_topo && din && dout && din != dout && x && r && (part == 1 || part == 2), "fcvm_mg_l1_pass_part: no multigrid hierarchy / bad arguments");
    NEED((defer == nullptr) == (t_out == nullptr), "fcvm_mg_l1_pass_part: defer and t_out go together");
    NEED(c->d_l1_sb || c->d_l1_si, "fcvm_mg_l1_pass_part: no slice lists (fcvm_mg_l1_set_interface)");
    MatView A = coarse_view(c);
    NEED(A.A16 && c->d_dinv1_6f, "fcvm_mg_l1_pass_part: needs the 16-bit copy of the level-1 matrix (FCVM_MG_L1_16=1)");
    ChebEpi epi{c->d_dinv1, x, r, dout, cdd, cdr, update_d ? 1 : 0, add_dout ? 1 : 0, c->d_dinv1_6f};
    epi.defer = defer; epi.t_out = t_out; epi.t2 = t2;
    const int64_t n = part == 1 ? c->n_l1_sb : c->n_l1_si;
    const int32_t *list = part == 1 ? c->d_l1_sb : c->d_l1_si;
    if (n > 0)
        hipLaunchKernelGGL((k_spmv_l1_i16<1, false>), dim3((unsigned)n), dim3(64), 0, c->stream, A.nslices, A.rp16, A.A16, A.B16, A.C16, A.scale16, A.rownode, din,
                           (double *)nullptr, (const double *)nullptr, epi, (const f4v *)nullptr, list);
    HCHK(hipGetLastError());
    return 0;
}

/* raw product of the level-1 16-bit copy: y = A1_local x (the last pre-smoothing product, which stays unsummed) */
int fcvm_mg_l1_spmv(fcvm_ctx *c, const double *x, double *y) {
    NEED(c && c->mg_topo && x && y && x != y, "fcvm_mg_l1_spmv: no multigrid hierarchy / null or aliased vectors");
    MatView A = coarse_view(c);
    NEED(A.A16, "fcvm_mg_l1_spmv: needs the 16-bit copy of the level-1 matrix (FCVM_MG_L1_16=1)");
    hipLaunchKernelGGL((k_spmv_l1_i16<0, false>), dim3((unsigned)A.nslices), dim3(64), 0, c->stream, A.nslices, A.rp16, A.A16, A.B16, A.C16,
                       A.scale16, A.rownode, x, y, (const double *)nullptr, ChebEpi{});
    HCHK(hipGetLastError());
    return 0;
}

/* the shared-node sum of fcvm_unpack_sum (three components, level-1 nodes) fused with the smoother step fcvm_mg_l1_pass deferred */
int fcvm_unpack_sum_cheb(fcvm_ctx *c, int64_t nshared, const int64_t *node, const int64_t *eptr, const int64_t *src, const double *traw,
                         const double *recv, const double *din, double *dout, double *x, double *r, double cdd, double cdr, int update_d,
                         int add_dout) {
    NEED(c && c->mg_topo && c->d_dinv1_6f && nshared >= 0, "fcvm_unpack_sum_cheb: no multigrid hierarchy");
    if (nshared == 0) return 0;
    hipLaunchKernelGGL(k_unpack_sum_cheb, dim3((unsigned)cdiv(nshared, 256)), dim3(256), 0, c->stream, nshared, node, eptr, src, traw, recv,
                       (const float *)c->d_dinv1_6f, din, dout, x, r, cdd, cdr, update_d ? 1 : 0, add_dout ? 1 : 0);
    HCHK(hipGetLastError());
    return 0;
}

int fcvm_set_mg_cycle(fcvm_ctx *c, int additive, double weight) {
    NEED(c && weight > 0.0 && additive >= 0 && additive <= 2, "fcvm_set_mg_cycle: form 0, 1 or 2; weight must be positive");
    if ((additive == 2) != (c->mg_additive == 2)) c->assembled = false;   // the 16-bit copy is written by the next fcvm_assemble
    c->mg_additive = additive;
    c->mg_add_w = weight;
    return 0;
}

int fcvm_get_mg_cycle(fcvm_ctx *c, int *form, double *weight, int *fp16_copy_in_use) {
    NEED(c && form && weight && fp16_copy_in_use, "fcvm_get_mg_cycle: bad arguments");
    *form = mg_form(c);
    *weight = c->mg_add_w;
    *fp16_copy_in_use = mg_use_lo(c) ? 1 : 0;
    return 0;
}

/*
 * Linear-buckling matrices of calcTSM(nstep = 1) (fcVM.py:1002-1006, 1051-1062): K_b = elastic stiffness
 * without boundary-condition elimination, diagonal of fixed dofs x 100; K_g = geometric stiffness of the
 * Gauss-point stresses sig.  The solver matrix becomes A = K_b + shift * K_g (= K_b - shift * G with the
 * reference's G = -K_g), preconditioner and multigrid hierarchy are refreshed for it.
 */
int fcvm_assemble_buckling(fcvm_ctx *c, const double *sig, double shift) {
    NEED(c && c->have_mesh && c->have_mat && sig, "fcvm_assemble_buckling: mesh, material and stresses are required");
    if (wsym_finish(c)) return 1;
    hipStream_t s = c->stream;
    const int64_t nv = c->pat.value_count();
    if (!c->d_vals_g && dalloc(&c->d_vals_g, nv)) return 1;
    const bool use_lds = c->assemble_lds && c->asm_lds_ok;
    if (!use_lds) HCHK(hipMemsetAsync(c->d_vals, 0, sizeof(double) * nv, s));
    HCHK(hipMemsetAsync(c->d_vals_g, 0, sizeof(double) * nv, s));
    MatConst mc = matconst(c, 0.0);
    dim3 block(256);
    if (use_lds) {
        const double g0[3] = {0.0, 0.0, 0.0};
        if (launch_assemble_lds<false>(c, nullptr, nullptr, nullptr, mc, g0, c->d_r, c->d_z, nullptr, nullptr, 1)) return 1;
    } else {
        const int64_t nslots = c->pat.nslices * SLICE;
        hipLaunchKernelGGL(k_assemble_rows<false>, dim3((unsigned)cdiv(nslots, 256)), block, 0, s, nslots, c->d_rownode, c->d_nptr, c->d_nel,
                           c->d_el, c->d_coord, (const double *)nullptr, c->d_fixmask, c->d_fixval, c->d_emap, (const double *)nullptr,
                           (const uint8_t *)nullptr, mc, 0.0, 0.0, 0.0, c->d_vals, c->d_r, c->d_z, (double *)nullptr, (double *)nullptr, 1);
    }
    {
        // round 4: one double per block summed in LDS, written once (k_assemble_geometric_lds); FCVM_GEOM_LDS=0: the round-1 kernel
        const size_t gl = (size_t)asm_geom_lds_bytes(c->pat.maxlen);
        const bool geom_lds = !(getenv("FCVM_GEOM_LDS") && atoi(getenv("FCVM_GEOM_LDS")) == 0) && gl <= (size_t)64 * 1024;
        const int64_t nslots = c->pat.nslices * SLICE;
        if (geom_lds) {
            const int64_t nitems = 4 * c->pat.nslices;
            hipLaunchKernelGGL(k_assemble_geometric_lds, dim3((unsigned)nitems), dim3(64), gl, s, nitems, c->d_sptr, c->d_rownode, c->d_nptr, c->d_nel,
                               c->d_el, c->d_coord, c->d_emap, sig, c->d_vals_g);
        } else {
            hipLaunchKernelGGL(k_assemble_geometric, dim3((unsigned)cdiv(nslots, 256)), block, 0, s, nslots, c->d_rownode, c->d_nptr, c->d_nel,
                               c->d_el, c->d_coord, c->d_emap, sig, c->d_vals_g);
        }
    }
    if (shift != 0.0)
        hipLaunchKernelGGL(k_axpby, dim3((unsigned)cdiv(nv, 256)), dim3(256), 0, s, nv, 1.0, c->d_vals, shift, c->d_vals_g, c->d_vals);
    HCHK(hipGetLastError());
    c->wsym_use = false;      // K_b + sigma K_g is not positive definite in general: the bound behind the fixed-point sums does not hold
    c->sell_valid = true;
    if (refresh_precond(c)) return 1;
    if (c->precond_mode >= 1) {
        HCHK(hipStreamSynchronize(s));
        c->mg_lmax_stale = true;  // different operator (penalty rows, shift)
        if (mg_setup(c)) return 1;
        c->mg_lmax_stale = true;  // and the next ordinary assembly is different again
    }
    if (win_join(c)) return 1;
    HCHK(hipStreamSynchronize(s));
    c->assembled = true;
    return 0;
}

/* y = K_g x with the geometric stiffness of the last fcvm_assemble_buckling */
int fcvm_spmv_geometric(fcvm_ctx *c, const double *x, double *y) {
    NEED(c && c->have_mesh && c->d_vals_g, "fcvm_spmv_geometric: call fcvm_assemble_buckling first");
    NEED(x != y, "fcvm_spmv_geometric: x and y must not alias");
    MatView m = fine_view(c);
    m.vals = c->d_vals_g;
    launch_spmv_t<false>(c, m, x, y, nullptr, nullptr);
    HCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
