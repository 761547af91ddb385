/*
 * fcvm_hip.h -- C-ABI of libfcvm_hip.so: the MI355X (gfx950) hot path of the fcVM
 * elastoplastic tet10 solver.
 *
 * The reference (HarryvL/fcVM-workbench, "source code/fcVM.py") is Python + numba + CHOLMOD; the
 * entry points below are what a binding for its per-element / linear-solve path has to call.
 * Each one cites the reference routine it replaces.  All pointers are plain host or device
 * pointers (documented per argument as [host] / [dev]); no torch / numpy types cross this line.
 *
 * Conventions (same as the reference): elNodes 1-based [ne][10] in fcVM's internal node order;
 * dof = 3*(node-1)+k; stress vectors (xx,yy,zz,xy,zx,yz); 4 Gauss points / element, Gauss-point
 * arrays are [4*ne] or [24*ne]; everything is FP64.
 *
 * Every function returns 0 on success, non-zero on failure; fcvm_last_error() describes the
 * failure of the calling thread's last call.
 */
#ifndef FCVM_HIP_H
#define FCVM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fcvm_ctx fcvm_ctx;

const char *fcvm_last_error(void);
int fcvm_version(void);
int fcvm_device_count(int *count);

/* one context = one GPU + one HIP stream + one mesh partition */
int fcvm_create(int device, fcvm_ctx **out);
int fcvm_destroy(fcvm_ctx *c);
/* out[0] = free, out[1] = total device memory in bytes */
int fcvm_mem_info(fcvm_ctx *c, int64_t out[2]);
int fcvm_synchronize(fcvm_ctx *c);
/* external != 0: run the context's kernels on the caller's hipStream_t (e.g. torch's current
 * stream, so that torch ops and RCCL collectives issued by the host order correctly with them;
 * a NULL handle then means HIP's default stream).  external == 0: back to the context's own stream */
int fcvm_set_stream(fcvm_ctx *c, void *hip_stream, int external);

/* raw device memory helpers for the host binding */
int fcvm_malloc(fcvm_ctx *c, uint64_t bytes, void **dev);
int fcvm_free(fcvm_ctx *c, void *dev);
int fcvm_h2d(fcvm_ctx *c, void *dev, const void *host, uint64_t bytes);
int fcvm_d2h(fcvm_ctx *c, void *host, const void *dev, uint64_t bytes);
int fcvm_d2d(fcvm_ctx *c, void *dst, const void *src, uint64_t bytes);
int fcvm_memset0(fcvm_ctx *c, void *dev, uint64_t bytes);

/*
 * Mesh, material and displacement boundary conditions.
 * Replaces the (elNodes, nocoord, materialbyElement, fix) arguments of calcGSM / calcTSM /
 * update_stress_load (fcVM.py:621, 820, 2197).  Builds the node-node sparsity pattern and the
 * sliced-ELL 3x3-block matrix layout (host threads), uploads everything.
 * elNodes [host] int64 [ne*10] 1-based, nocoord [host] f64 [nn*3].
 * fix: nfix pairs (dof, value) -- the reference's numba Dict `fix` (fcVM.py:222-254).
 */
int fcvm_set_mesh(fcvm_ctx *c, int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord);
int fcvm_set_material(fcvm_ctx *c, double E, double nu, double density);
int fcvm_set_fix(fcvm_ctx *c, int64_t nfix, const int64_t *dofs, const double *vals);
/* fcvm_set_mesh returns while a background thread still builds the symmetric form of the window format (round 4); the first
 * fcvm_assemble waits for it and uploads it.  A caller with host work of its own after fcvm_set_mesh (supports, load vectors)
 * calls this when that work is done, so that the wait does not fall into the analysis.  No reference counterpart. */
int fcvm_finish_setup(fcvm_ctx *c);

/* nodal coordinate update (imperfections, fcVM.py:1240) */
int fcvm_set_coords(fcvm_ctx *c, const double *nocoord);

/* pattern statistics: out[0]=nnz blocks, out[1]=padded block slots, out[2]=slices,
 * out[3]=matrix bytes (values+cols), out[4]=max row length, out[5]=ndof, out[6]=1 (reserved),
 * out[7]=SpMV kernel variant */
int fcvm_pattern_info(fcvm_ctx *c, int64_t out[8]);
/* which kernel fcvm_spmv / the PCG iteration use for the fine matrix: out[0]=8 or 4: the window format with groups of that
 * many 64-row waves (second storage of the
 * matrix for the SpMV, csrc/fcvm_spmv_win.hip.h; FCVM_SPMV_WIN=0 or a mesh whose windows do not fit the LDS: 0 = the
 * sliced-ELL kernel), out[1]=groups (workgroups), out[2]=largest window in nodes, out[3]=mean window, out[4]=stored blocks,
 * out[5]=bytes streamed per launch from that storage, out[6]=LDS bytes per workgroup, out[7]=row split length */
int fcvm_spmv_info(fcvm_ctx *c, int64_t out[8]);
/* host copy of the assembled matrix as scalar COO (row, col, val) of the FULL symmetric matrix,
 * nnz = 9 * nnz blocks; for tests.  Arrays must hold 9*out[0] entries. */
int fcvm_matrix_to_coo(fcvm_ctx *c, int64_t *row, int64_t *col, double *val);

/*
 * calcGSM (fcVM.py:729-800) / material part of calcTSM (fcVM.py:940-1050).
 * Integrates K^e = sum_gp B^T D B |J| w for every element on the GPU and scatters it into the
 * global matrix with the displacement boundary conditions folded in exactly as the reference
 * does (unit diagonal per element occurrence on fixed dofs, modf corrections).
 *   disp   [dev|NULL] f64[3nn]  coordinates are nocoord+disp when given (calcTSM, fcVM.py:962-967)
 *   sig    [dev|NULL] f64[24ne], pgp [dev|NULL] u8[4ne]: plastic tangent D - pmat at plastic
 *          Gauss points (fcVM.py:983-998); Et_E hardening ratio
 *   grav   gravity vector; glv_grav [dev] f64[3nn] receives (=, not +=) the gravity load vector
 *   modf   [dev] f64[3nn] receives the boundary-condition load modification
 *   xgp    [dev|NULL] f64[4ne*3] Gauss point coordinates; volume [host|NULL]
 * Afterwards the block-Jacobi preconditioner is refreshed.
 */
int fcvm_assemble(fcvm_ctx *c, const double *disp, const double *sig, const uint8_t *pgp,
                  double Et_E, const double grav[3], double *glv_grav, double *modf, double *xgp,
                  double *volume);

/*
 * update_stress_load (fcVM.py:2197-2464): strain increment, trial stress, von Mises radial
 * return (vmises_original_optimised, fcVM.py:2469-2492), internal force vector.
 * All arrays [dev].  qin is overwritten (the reference passes a zeroed vector).  LD != 0 selects
 * the updated-Lagrange branch (fcVM.py:2256-2260, 2383-2429).
 */
int fcvm_update_stress_load(fcvm_ctx *c, const double *sig_yield, const double *disp_new,
                            const double *du, const double *sig_old, double *sig_new,
                            double *sig_test, double *qin, double Et_E, int LD, uint8_t *pgp);

/*
 * update_PEEQ_CSR (fcVM.py:2085-2137) plus the reductions calcDisp takes right after it
 * (fcVM.py:1546-1554): out[0]=argmax(csr) (first occurrence), out[1]=max(csr), out[2]=max(peeq),
 * out[3..7]= pressure, sigmises, triax, ecr, peeq at that Gauss point.  Arrays [dev], out [host].
 */
int fcvm_update_peeq_csr(fcvm_ctx *c, const double *sig_test, const double *sig_new,
                         double *sig_yield, double ultimate_strain, double *peeq, double *csr,
                         double *triax, double *pressure, double *sigmises, double *ecr,
                         double Et_E, double out[8]);

/* mapStresses (fcVM.py:2497-2554). sig etc. [dev]; outputs [dev]: t10s f64[nn*6], others f64[nn] */
int fcvm_map_stresses(fcvm_ctx *c, int averaged, const double *sig, const double *peeq,
                      const double *sigvm, const double *csr, double sig_yield, double *t10s,
                      double *t10peeq, double *t10csr, double *t10svm, double *t10triax);

/* calculate_principal_stress (fcVM.py:2954-2994): t10s [dev] f64[nn*6] ->
 * ps [dev] f64[nn*3] (descending), pv [dev] f64[nn*9] (three eigenvectors scaled by their
 * eigenvalue, row k = k-th principal direction) */
int fcvm_principal_stress(fcvm_ctx *c, int64_t n, const double *t10s, double *ps, double *pv);

/*
 * Linear solve K u = f, replacing sksparse.cholmod `factor(f)` (fcVM.py:1121-1135, 1401):
 * block-Jacobi preconditioned conjugate gradients on the assembled matrix, FP64, device resident.
 * f, u [dev] f64[3nn].  Stops at ||r|| <= rtol*||f|| or maxit.  info[0]=iterations,
 * info[1]=final relative residual, info[2]=SpMV launches.  [host]
 */
int fcvm_solve(fcvm_ctx *c, const double *f, double *u, double rtol, int64_t maxit, double info[3]);

/*
 * Linear buckling (gnl == "GNLY", fcVM.py:1199-1214).  fcvm_assemble_buckling builds the matrices of
 * calcTSM(nstep = 1): K_b (elastic stiffness, no boundary-condition elimination, fixed diagonal x 100,
 * fcVM.py:1051-1062) and K_g (geometric stiffness of the Gauss-point stresses sig [dev] f64[24ne],
 * fcVM.py:1002-1006), and makes A = K_b + shift*K_g the matrix of fcvm_solve / fcvm_spmv (the
 * reference's shifted operator K - sigma*G with G = -K_g).  fcvm_spmv_geometric: y = K_g x.
 * The next fcvm_assemble restores the ordinary stiffness matrix.
 */
int fcvm_assemble_buckling(fcvm_ctx *c, const double *sig, double shift);
int fcvm_spmv_geometric(fcvm_ctx *c, const double *x, double *y);

/* y = K x on the assembled matrix ([dev] f64[3nn]); building block of the multi-GPU solver */
int fcvm_spmv(fcvm_ctx *c, const double *x, double *y);
/* z = M^-1 r with the 3x3 block-Jacobi preconditioner */
int fcvm_precond(fcvm_ctx *c, const double *r, double *z);
/* diagonal 3x3 blocks [dev] f64[nn*9]: get (for the interface sum across ranks) / set+invert */
int fcvm_get_diag_blocks(fcvm_ctx *c, double *dblk);
int fcvm_set_diag_blocks(fcvm_ctx *c, const double *dblk);

/* small FP64 vector kernels used by the Newton / Riks driver (all [dev], n entries) */
int fcvm_vec_axpby(fcvm_ctx *c, int64_t n, double a, const double *x, double b, const double *y, double *out);
int fcvm_vec_mul(fcvm_ctx *c, int64_t n, const double *x, const double *y, double *out);
int fcvm_vec_dot(fcvm_ctx *c, int64_t n, const double *x, const double *y, double *result_host);
/* k (1..16) inner products xs[j] . y in one pass over y and one host synchronisation; every result is bitwise what
 * fcvm_vec_dot returns for that pair.  xs: host array of k device pointers; result_host[k]. */
int fcvm_vec_mdot(fcvm_ctx *c, int64_t n, int k, const double *const *xs, const double *y, double *result_host);
/* w-weighted dot: sum w_i x_i y_i (owner weights of the multi-GPU partition); result [dev] */
int fcvm_vec_wdot_dev(fcvm_ctx *c, int64_t n, const double *w, const double *x, const double *y, double *result_dev);
/* sqrt(max over the first nnodes nodes of |u_node|^2) (fcVM.py:1539-1542) */
/* scalars of the multi-GPU PCG on the device (fcvm_hip/dist.py): S [dev, 7] = {rz, rr, thresh, done, iterations, alpha, beta};
 * phase 0: in [dev, 1] = p.q -> alpha; phase 1: in [dev, 2] = {r.z, r.r} -> beta, rz, rr, iterations, done (alpha = beta = 0 once
 * done): one launch instead of a dozen one-element tensor operations per iteration */
int fcvm_dist_pcg_scalars(fcvm_ctx *c, int phase, double *S, const double *in);
/* (phases 2 / 3: the single-reduction CG of round 3, in [dev, 3] = {r.z, w.z, r.r} with w = K z -- 3 starts the loop (S[2] =
 * threshold set by the caller), 2 follows every update: iteration count, convergence flag, beta and the next alpha) */
/* *result_host = 1 if any of the n bytes at a (device) is non-zero (the "any plastic Gauss point?" test of fcVM.py:1351) */
int fcvm_any_u8(fcvm_ctx *c, int64_t n, const uint8_t *a, int *result_host);
int fcvm_vec_max_node_norm(fcvm_ctx *c, int64_t nnodes, const double *u, double *result_host);

/* Two systems with the same matrix, K uA = fA and K uB = fB, solved in lockstep: the pair of back substitutions the reference
 * makes with one tangent factor per GNLY Newton iteration (ue = factor(f), due = factor(relax * r), fcVM.py:1351-1401) and
 * the block solves of the buckling pre-analysis (fcVM.py:1199-1214).  Every pass over the matrix -- the FP64 SpMV and the two
 * 16-bit smoother passes of the cycle -- serves both right-hand sides (k_spmv_win2 / k_spmv_win_lo2); each result is bitwise
 * what fcvm_solve returns for that right-hand side.  Needs cycle form 2 on the window format with room for two windows in
 * LDS, otherwise the systems are solved one after the other.  info[6]: {iterations, relative residual, timed SpMV launches}
 * of A, then of B.  All vectors [dev, 16-byte aligned], pairwise distinct. */
int fcvm_solve2(fcvm_ctx *c, const double *fA, const double *fB, double *uA, double *uB, double rtolA, double rtolB, int64_t maxit,
                double info[6]);

/* y = K~ x: the product with the 16-bit copy of the matrix that cycle form 2 smooths with (the multi-GPU layer drives its
 * cycle from the host and exchanges the product over the shared nodes before using it); the FP64 product where no copy exists */
int fcvm_spmv_lo(fcvm_ctx *c, const double *x, double *y);
/* the two shared-pass building blocks of fcvm_solve2 on their own: yA = K xA, yB = K xB (bitwise fcvm_spmv of each) and
 * zA = M^-1 rA, zB = M^-1 rB (bitwise fcvm_mg_apply of each); for block methods and for the tests */
int fcvm_spmv2(fcvm_ctx *c, const double *xA, const double *xB, double *yA, double *yB);
int fcvm_mg_apply2(fcvm_ctx *c, const double *rA, const double *rB, double *zA, double *zB);

/* z = M^-1 r: one application of the multigrid preconditioner fcvm_solve uses (mode 1), r != z [dev, 16-byte aligned].
 * No reference counterpart (CHOLMOD's factor(f), fcVM.py:1130, is replaced by PCG); used by the solver-space tools and tests. */
int fcvm_mg_apply(fcvm_ctx *c, const double *r, double *z);

/* timing of the last fcvm_solve (HIP events on the solver's stream): out[0] = milliseconds spent in
 * the timed fine-level SpMV launches, out[1] = total milliseconds, out[2] = number of timed SpMV
 * launches, out[3] = coarse-level SpMV launches (multigrid) */
int fcvm_solve_timing(fcvm_ctx *c, double out[4]);

/*
 * Preconditioner of fcvm_solve.  mode 0: 3x3 block-Jacobi.  mode 2: build the multigrid hierarchy
 * only (host-driven V-cycle, multi-GPU).  mode 1: p-multigrid V-cycle
 * (level 0 = assembled tet10 matrix, level 1 = linear tets on the corner nodes, Chebyshev
 * smoothing in Dinv*A: k0 steps before and after the coarse correction on [lmax/alpha0, lmax],
 * k1 steps as approximate coarse solve on [lmax1/alpha1, lmax1]).  Takes effect at the next
 * fcvm_assemble.
 */
int fcvm_set_precond(fcvm_ctx *c, int mode, int k0, int k1, double alpha0, double alpha1);
/* Shared-node exchange of the element partition (one process per GPU): values of the nodes a rank shares with its
 * neighbours go into one send buffer (fcvm_pack_nodes: out[k][c] = v[idx[k]][c]); after the point-to-point transfers
 * fcvm_unpack_sum replaces v at every interface node by the sum over its entry list, src < 0 standing for the rank's
 * own value and src >= 0 for row src of the receive buffer.  The host sorts the lists by source rank, so every sharer
 * adds the same numbers in the same order.  Index arrays are device pointers (int64). */
int fcvm_pack_nodes(fcvm_ctx *c, int64_t n, const int64_t *idx, int ncomp, const double *v, double *out);
int fcvm_unpack_sum(fcvm_ctx *c, int64_t nshared, const int64_t *node, const int64_t *eptr, const int64_t *src, int ncomp,
                    double *v, const double *recv);
/* level-0 part of the V-cycle: additive != 0 (default): x = weight*c0*Dinv b + P M1^-1 P^T b, i.e. no
 * fine-level SpMV inside the preconditioner; additive == 0: multiplicative Chebyshev pre/post smoothing */
int fcvm_set_mg_cycle(fcvm_ctx *c, int additive, double weight);
/* additive == 2: multiplicative V(1,1) with one damped block-Jacobi step before and after the coarse correction whose two
 * fine-level SpMVs stream a 16-bit fixed-point copy of the matrix (20 instead of 74 bytes per 3x3 block; written by the next
 * fcvm_assemble; needs the window format, otherwise the FP64 form 0 runs).  The PCG itself -- p.Ap, the residual
 * recurrence and the convergence test -- stays on the FP64 matrix. */
int fcvm_get_mg_cycle(fcvm_ctx *c, int *form, double *weight, int *fp16_copy_in_use);
/* out[0]=mode, out[1]=coarse nodes, out[2]=lmax level 0, out[3]=lmax level 1, out[4]=coarse nnz
 * blocks, out[5]=fine SpMV launches inside V-cycles of the last solve, out[6]=level-2 aggregates,
 * out[7]=level-2 unknowns (0 = no level 2) */
int fcvm_precond_info(fcvm_ctx *c, double out[8]);

/*
 * Optional third multigrid level.  The host binding groups the tet4 (level-1) nodes into
 * aggregates, gives every aggregate up to 6 orthonormalised rigid-body modes Z (rows of fixed dofs
 * zeroed), forms A2 = Z^T A1 Z from the level-1 matrix and hands over its dense inverse:
 *   fcvm_mg_coarse_nodes   [host] coord1 f64[nc*3], fixmask1 u8[nc]   (inputs for the aggregation)
 *   fcvm_mg_coarse_to_coo  [host] scalar COO of the level-1 matrix, 9 * (coarse nnz blocks) entries
 *   fcvm_mg_set_level2     agg int32[nc] aggregate of each level-1 node, aoff int64[nagg+1] first
 *                          level-2 unknown of each aggregate, Z f64[nc][3][6], A2inv f64[n2*n2]
 *                          row-major (NULL: transfer operators only -- the multi-GPU host code keeps a
 *                          row block of the inverse per rank and applies it with fcvm_gemv_f32),
 *                          k1s/alpha1s: Chebyshev smoothing steps / range on level 1.
 * Level 2 survives re-assemblies on the same mesh (tangent updates) and is dropped when the mesh or
 * its coordinates change.
 */
int fcvm_mg_coarse_nodes(fcvm_ctx *c, double *coord1, uint8_t *fixmask1);
int fcvm_mg_coarse_to_coo(fcvm_ctx *c, int64_t *row, int64_t *col, double *val);
int fcvm_mg_set_level2(fcvm_ctx *c, int64_t nagg, int64_t n2, const int32_t *agg, const int64_t *aoff,
                       const double *Z, const double *A2inv, int k1s, double alpha1s);
/* Third multigrid level built inside the library (host bookkeeping in C++, triple product and dense SPD inverse in HIP
 * kernels): aggregates = uniform bins of about `target` level-1 nodes, six orthonormalised rigid-body modes each,
 * A2 = Z^T A1 Z, explicit inverse kept in FP32.  out[0] = aggregates, out[1] = unknowns.  Call after fcvm_assemble. */
int fcvm_mg_build_level2(fcvm_ctx *c, int64_t target, int k1s, double alpha1s, int64_t out[2]);
/* When the aggregates outnumber what a dense inverse is built for (FCVM_MG_MAX_AGGREGATES, default 3000: meshes beyond
 * ~3 M tet10), fcvm_mg_build_level2 keeps A2 sparse (6 x 6 blocks, block-Jacobi Chebyshev smoothing) and closes the cycle with a
 * dense level of super-aggregates (bins of 2 x 2 x 2 aggregates x rigid-body modes).  out[0] = 1 if that path is in use,
 * out[1] = blocks of A2, out[2] = super-aggregates, out[3] = unknowns of the dense top level. */
int fcvm_mg_sparse_info(fcvm_ctx *c, int64_t out[4]);
/* Aggregate level of a PARTITIONED mesh (north_star: elements range-partitioned over the GPUs of a node; no reference counterpart --
 * fcVM.py:1121-1135 factorises once on one process).  Every rank keeps the WHOLE level, bit for bit, so its correction needs no gather:
 * the host layer (fcvm_hip/dist.py, DistMG._level2) numbers the aggregates globally, forms Z from Gram matrices summed over the ranks and
 * calls fcvm_mg_set_level2(..., aoff[g] = 6 g, A2inv = NULL); then
 *   fcvm_mg_l2_galerkin      this rank's share of A2 = Z^T A1 Z as 6 x 6 blocks, formed on the device; *nslots = their number
 *   fcvm_mg_l2_galerkin_get  sptr_g int64[nagg+1], slot_agg int32[nslots] (column aggregates, ascending within a row), B f64[nslots][36]
 *   (host layer: union of the ranks' block patterns and ONE all-reduce of the block values)
 *   fcvm_mg_l2_finish        the summed blocks in the global pattern -> nsuper == 0: dense inverse of the replicated matrix (n2 = 6 nagg,
 *                            applied from its lower triangle); nsuper > 0: the sparse level (block-Jacobi Chebyshev sweeps) closed by a dense
 *                            level of the caller's super-aggregates: sagg int32[nagg], soff int64[nsuper+1] (first mode of each), C
 *                            f64[nagg][6][6] (the super-aggregate's modes in the basis of the aggregate, [k][m]); arank int32[nagg]: modes
 *                            kept per aggregate (the others are zero columns of Z and get a unit diagonal); k2 / ncoarse: smoothing steps per
 *                            sweep and super-aggregate corrections per cycle of the sparse form (0: the library's 8 / 2; every rank computes the
 *                            whole level, so the multi-GPU layer asks for 2 / 2: the same PCG iteration count at 7 instead of 25 launches).
 * fcvm_mg_l2_solve then returns e2 = M2^-1 r2 for the replicated r2 on every rank. */
int fcvm_mg_l2_galerkin(fcvm_ctx *c, int64_t *nslots);
int fcvm_mg_l2_galerkin_get(fcvm_ctx *c, int64_t *sptr_g, int32_t *slot_agg, double *B);
int fcvm_mg_l2_finish(fcvm_ctx *c, int64_t nagg, const int32_t *arank, int64_t nslots, const int64_t *sptr_g, const int32_t *slot_agg,
                      const double *B, int64_t nsuper, const int32_t *sagg, const int64_t *soff, const double *C, int k2, int ncoarse);
/* dst = inverse of the SPD n x n matrix src (row-major, device pointers, no aliasing) */
int fcvm_spd_inverse(fcvm_ctx *c, int64_t n, const double *src, double *dst);

/* diagnostic: streaming read rate of a freshly allocated buffer of nbytes (16 B per lane, nontemporal): out[0] = GB/s,
 * out[1] = ms per pass.  The HBM rate this GPU delivers right now, to put a measured SpMV time into perspective. */
int fcvm_membench(fcvm_ctx *c, int64_t nbytes, double out[2]);

/* The symmetric form of the window format that fcvm_spmv / the PCG use for an assembled stiffness matrix (round 3: one block
 * per pair K_ij, K_ji inside a group, the transposed product through LDS in 64-bit fixed point): out[0] = 1 if built for this
 * mesh, out[1] = stored entries, out[2] = blocks not stored, out[3] = bytes one launch streams, out[4] = LDS bytes per
 * workgroup, out[5] = 1 if it holds the values of the last assembly, out[6] = 1 if fcvm_assemble can write the form itself
 * (round 4: rows leave the assembly kernel's LDS straight into this storage, its 16-bit copy and the diagonal blocks), out[7] = 1
 * if the last assembly did.  Round 5: out[8] = 1 if the form was DECLINED for this mesh because of its grading -- the fixed-point sums take
 * their unit from the largest matrix entry, so products in the region of smallest entries lose relative precision as out[9] = max / min
 * over the nodes of the largest free diagonal entry (measured at every stiffness assembly); beyond FCVM_SYM_GRADING_MAX (1e4) the
 * context assembles into and multiplies with the full window form instead.  No reference counterpart (CHOLMOD, fcVM.py:1121). */
int fcvm_spmv_sym_info(fcvm_ctx *c, int64_t out[10]);
/* Which group of the window format a workgroup of the symmetric-form kernels takes (workgroups go to the eight XCDs round-robin): mode 0 = every
 * XCD walks one contiguous eighth of the groups (rounds 2-4), 1 = group = workgroup, C >= 2 = chunks of C consecutive groups per XCD inside
 * super-chunks of 8 C groups (one moving front through the value array instead of eight streams an eighth of it apart; L2 sharing of window
 * nodes kept inside the chunks; default C = 16).  Device-wide.  FCVM_WIN_GMAP sets it when a context is created. */
int fcvm_set_group_map(fcvm_ctx *c, int mode);
/* Multi-GPU layer, overlap of the shared-node exchange with the interior rows of the three fine-level passes (north_star: "halo displacement
 * exchange at shared nodes"; no reference counterpart -- fcVM.py is one process).  fcvm_set_interface_nodes: the local nodes (0-based) another rank
 * holds too; the groups of the window format with a row of such a node form the first launch list, the rest the second (out[0] / out[1]: their
 * sizes; n = 0: no lists).  fcvm_spmv_dot_part / fcvm_spmv_lo_part(part): 0 = all groups, 1 = the first list (every entry the exchange needs is
 * complete after it), 2 = the second list (launched after the sends are posted); after part 0 or 2 fcvm_spmv_dot_part also reduces x . y.  Parts
 * 1 + 2 are bitwise part 0.  Without the symmetric window form part 1 does everything and part 2 nothing. */
int fcvm_set_interface_nodes(fcvm_ctx *c, int64_t n, const int64_t *nodes, int64_t out[2]);
int fcvm_spmv_dot_part(fcvm_ctx *c, const double *x, double *y, double *xy_dev, int part);
int fcvm_spmv_lo_part(fcvm_ctx *c, const double *x, double *y, int part);
/* out = w r - K~_local x on the 16-bit copy (w per dof): the residual after the pre-smoothing step of the multi-GPU cycle formed by the smoother pass itself,
 * the rank's product left unsummed for the restriction (fcvm_mg_restrict) whose level-1 exchange completes the sum */
int fcvm_spmv_lo_resid(fcvm_ctx *c, const double *x, const double *r, const double *w, double *out);
/* Multi-GPU layer, level 1 of the cycle on the library's current kernels (round 5): fcvm_mg_l1_pass = one Chebyshev pass on this rank's 16-bit
 * copy of the tet4 matrix with the smoother step as epilogue (t = A1_local din (+ t2); r -= t; x += din; dout = cdd din + cdr Dinv r; x += dout
 * if add_dout) for every row except those of nodes flagged in defer (u8 per level-1 node: held by another rank too) -- their raw product goes to
 * t_out, the exchange sums the ranks' parts, and fcvm_unpack_sum_cheb (the sum of fcvm_unpack_sum fused with that step) finishes them.
 * fcvm_mg_l1_spmv: the raw product y = A1_local x of the same copy. */
int fcvm_mg_l1_pass(fcvm_ctx *c, const double *din, double *dout, double *x, double *r, const double *t2, double *t_out, const uint8_t *defer,
                    double cdd, double cdr, int update_d, int add_dout);
/* Boundary slices first for the level-1 passes (round 5; no reference counterpart): fcvm_mg_l1_set_interface gives the library the level-1 (corner)
 * nodes, 0-based in the level-1 numbering, that another rank holds too; fcvm_mg_l1_pass_part(part = 1) covers the slices with a row of one of them,
 * part = 2 the others (beside the exchange of the raw products), part = 0 everything; parts 1 + 2 are bitwise the unsplit pass.  out[0] / out[1] =
 * slices in the two lists; n = 0 drops them. */
int fcvm_mg_l1_set_interface(fcvm_ctx *c, int64_t n, const int64_t *nodes, int64_t out[2]);
/* r1 = P^T r0 (fcvm_mg_restrict) for the level-1 nodes of `part`: 1 = the nodes given to fcvm_mg_l1_set_interface, 2 = the others, 0 = all */
int fcvm_mg_restrict_part(fcvm_ctx *c, const double *r0, double *r1, int part);
int fcvm_mg_l1_pass_part(fcvm_ctx *c, const double *din, double *dout, double *x, double *r, const double *t2, double *t_out, const uint8_t *defer,
                         double cdd, double cdr, int update_d, int add_dout, int part);
int fcvm_mg_l1_spmv(fcvm_ctx *c, const double *x, double *y);
int fcvm_unpack_sum_cheb(fcvm_ctx *c, int64_t nshared, const int64_t *node, const int64_t *eptr, const int64_t *src, const double *traw,
                         const double *recv, const double *din, double *dout, double *x, double *r, double cdd, double cdr, int update_d,
                         int add_dout);
/* v[idx[k]][c] += a[k][c] for distinct idx (int64, device): the follower-pressure part of the external loads (calcTSM, fcVM.py:858-883) is computed on the
 * host from the displacements of the loaded faces' nodes alone (gathered with fcvm_pack_nodes) and added here -- no full displacement vector crosses PCIe
 * per GNLY tangent */
int fcvm_scatter_add_nodes(fcvm_ctx *c, int64_t n, const int64_t *idx, int ncomp, const double *a, double *v);
/* Placement tuning (round 5; no reference counterpart): where the driver puts an allocation decides 10-15 % of the time of the kernel that
 * streams it or scatters its results into it.  After the first stiffness assembly of a mesh the PCG's result vector and the value array of the
 * 16-bit passes' result vectors (and, with FCVM_TUNE_PLACEMENT_ARRAYS=1, the value / column arrays they stream) are each timed against
 * FCVM_TUNE_PLACEMENT - 1 (default 2) fresh allocations on the assembled matrix and the fastest is kept.  out[0] = 1 if it ran, out[1] / out[2] = SpMV time [ms] with the result vector first allocated / kept, out[3] / out[4] = with the value
 * array first allocated / kept, out[5] = milliseconds the tuning took. */
int fcvm_placement_info(fcvm_ctx *c, double out[6]);
/* Vertex-patch smoother of the fine multigrid level (round 5; no reference counterpart -- the reference factorises with CHOLMOD,
 * fcVM.py:1121-1135): additive Schwarz over the patches "corner node + mid-side nodes of its edges" in the place of the damped
 * block-Jacobi step of cycle form 2; patch matrices cut out of the assembled symmetric window form after every assembly, inverted on
 * the device, kept as FP32 triangles.  Off by default (FCVM_MG_PATCH=1 switches it on): with the library's aggregate-based coarse solve it
 * does not pay on large meshes (profiles/r05_patch_smoother.txt).  out[0] = tables built for this mesh, out[1] = the cycle uses
 * it (inverses valid for the last assembly), out[2] = patches, out[3] = largest patch [unknowns], out[4] = mean patch [unknowns],
 * out[5] = bytes of the inverses, out[6] = largest eigenvalue of S K (Lanczos), out[7] = milliseconds of the last refresh */
int fcvm_patch_info(fcvm_ctx *c, double out[8]);
/* test hook: out = S x (undamped) on device vectors of 3 nn doubles */
int fcvm_debug_patch_apply(fcvm_ctx *c, const double *x, double *out);
/* experiment hook: device addresses of the PCG's own vectors r, z, p, q and of its SpMV's partial buffer (placement studies) */
int fcvm_debug_pcg_vectors(fcvm_ctx *c, uint64_t out[5]);

/* host-only self check of the symmetric form of the window format (the PCG's FP64 SpMV stores one block of every pair
 * K_ij, K_ji inside a group and forms both products): out[0] = built, out[1] = stored entries, out[2] = blocks not stored,
 * out[3] = violated invariants (must be 0: every block of the matrix covered exactly once), out[4] = entries of the full
 * window format, out[5] = most extra row segments in a group, out[6] = blocks of the matrix, out[7] = largest window */
int fcvm_host_window_sym_check(int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord, int64_t out[8]);

/* The element -> block-location map of the assembly kernels is built on the device by fcvm_set_mesh (k_build_emap; the host
 * builder's map under FCVM_EMAP_HOST=1).  This compares it, entry by entry, with the host rule of fcvm_pattern.cpp:
 * out[0] = entries, out[1] = mismatches (must be 0).  Replaces nothing in the reference (its scipy COO -> CSC conversion,
 * fcVM.py:805-816, has no such table). */
int fcvm_check_device_emap(fcvm_ctx *c, int64_t out[2]);

/* host-only self check of the sparsity pattern builder (no GPU needed): out[0]=nnz blocks,
 * out[1]=padded slots, out[2]=slices, out[3]=max row length, out[4]=number of violated invariants
 * (must be 0), out[5]=value count, out[6]=sum of slice lengths */
int fcvm_host_pattern_check(int64_t ne, int64_t nn, const int64_t *elNodes, int64_t out[8]);

/* host-only self check of the SpMV's window format (second storage of the fine matrix: groups of 512 rows along a Morton
 * curve, x of a group's column nodes staged in LDS, 16-bit window columns, padding-free prefix storage; csrc/fcvm_pattern.h).
 * nocoord [nn*3]; max_window: largest admissible window in nodes (0: 65535).  out[0]=1 if the format was built (0: some
 * window is too large, the sliced-ELL kernel would be used), out[1]=groups, out[2]=largest window, out[3]=mean window,
 * out[4]=number of violated invariants (must be 0), out[5]=stored blocks, out[6]=nnz blocks, out[7]=sliced-ELL slots */
int fcvm_host_window_check(int64_t ne, int64_t nn, const int64_t *elNodes, const double *nocoord, int64_t max_window, int64_t out[8]);

/*
 * Fine-grained multigrid building blocks (levels 0 = tet10 matrix, 1 = tet4 matrix) for a V-cycle
 * driven by the host, as the multi-GPU solver does (it has to exchange shared-node sums between the
 * ranks after every local SpMV).  fcvm_set_precond mode 2 builds the hierarchy at fcvm_assemble but
 * leaves the solve to the host.  All vectors [dev]; level-1 vectors have 3*nc entries.
 *   fcvm_mg_coarse_map      c2f int32[nc] [host]: fine node of every level-1 node
 *   fcvm_spmv_level         y = A_level x
 *   fcvm_mg_get/set_diag_blocks   3x3 diagonal blocks of a level (set: inverts them for the smoother)
 *   fcvm_mg_cheb_init       x = 0, r = b, d = c0 * Dinv b
 *   fcvm_mg_cheb_step       x += d; r -= t (skipped when t is NULL); if update_d: d = cdd*d + cdr*Dinv r
 *   fcvm_mg_restrict        r1 = M1 P^T M0 r0        fcvm_mg_prolong   d0 = M0 P M1 e1
 */
int fcvm_mg_coarse_map(fcvm_ctx *c, int32_t *c2f);
int fcvm_spmv_level(fcvm_ctx *c, int level, const double *x, double *y);
int fcvm_mg_get_diag_blocks(fcvm_ctx *c, int level, double *dblk);
int fcvm_mg_set_diag_blocks(fcvm_ctx *c, int level, const double *dblk);
int fcvm_mg_cheb_init(fcvm_ctx *c, int level, const double *b, double *x, double *r, double *d, double c0);
int fcvm_mg_cheb_step(fcvm_ctx *c, int level, const double *t, double *x, double *r, double *d, double cdd,
                      double cdr, int update_d);
int fcvm_mg_restrict(fcvm_ctx *c, const double *r0, double *r1);
int fcvm_mg_prolong(fcvm_ctx *c, const double *e1, double *d0);
/* fused forms for the host-driven (multi-GPU) additive cycle: restriction with per-dof partition-of-unity weights;
 * z = z0 + M0 P M1 e1; x += alpha p, r -= alpha q, z0 = weight * Dinv r with alpha read from device memory */
int fcvm_mg_restrict_weighted(fcvm_ctx *c, const double *r0, const double *wgt, double *r1);
int fcvm_mg_prolong_add(fcvm_ctx *c, const double *e1, const double *z0, double *z);
int fcvm_pcg_update_xr_z0(fcvm_ctx *c, const double *alpha_dev, const double *p, const double *q, double *x, double *r,
                          double *z0, double weight);
/* round 4 -- fused forms for the host-driven cycle of form 2 (16-bit smoother passes), all with the six-entry FP32 form of
 * the inverse diagonal blocks, which pre- and post-smoothing must share:
 *   fcvm_pcg_update_xr_z0_f32: as fcvm_pcg_update_xr_z0;   fcvm_jacobi6f: z = weight * Dinv r;
 *   fcvm_mg_restrict_diff: r1 = P^T (wgt * r0 - sub) (the residual after pre-smoothing, formed on the fly);
 *   fcvm_dist_post_smooth: z = x2 + c0 * Dinv (b - t), and *bz_dev = sum ow b z, *bb_dev = sum ow b b over this rank's dofs
 *     (ow: owner weights; the two sums the single-reduction CG needs next, fixed summation order);
 *   fcvm_spmv_dot: y = K x and *xy_dev = x . y over this rank's rows (with consistent x and the rank's UNSUMMED product this
 *     is the rank's share of x^T K x: no owner weights, no exchange needed before it). */
int fcvm_pcg_update_xr_z0_f32(fcvm_ctx *c, const double *alpha_dev, const double *p, const double *q, double *x, double *r,
                              double *z0, double weight);
int fcvm_jacobi6f(fcvm_ctx *c, const double *r, double weight, double *z);
int fcvm_mg_restrict_diff(fcvm_ctx *c, const double *r0, const double *wgt, const double *sub, double *r1);
int fcvm_dist_post_smooth(fcvm_ctx *c, const double *x2, const double *b, const double *t, double c0, const double *ow, double *z,
                          double *bz_dev, double *bb_dev);
int fcvm_spmv_dot(fcvm_ctx *c, const double *x, double *y, double *xy_dev);
/*   fcvm_dist_cg_update: the vector half of a Chronopoulos-Gear iteration in one pass -- p = z + beta p, s = w + beta s (the
 *     previous iteration's direction updates), x += alpha p, r -= alpha s, z0 = weight * Dinv r; alpha = S_dev[5], beta = S_dev[6]
 *     as fcvm_dist_pcg_scalars leaves them;
 *   fcvm_mg_l2_restrict_weighted: r2 = Z^T (wgt * r1) (owner weights of the level-1 nodes) */
int fcvm_dist_cg_update(fcvm_ctx *c, const double *S_dev, const double *z, const double *w, double *p, double *s, double *x, double *r,
                        double *z0, double weight);
int fcvm_mg_l2_restrict_weighted(fcvm_ctx *c, const double *r1, const double *wgt, double *r2);
/*   fcvm_mg_l2_restrict_diff: r2 = Z^T (wgt * r1 - sub): the level-1 residual whose last product (sub: this rank's UNSUMMED A1 d)
 *     has not been exchanged -- the all-reduce of r2 completes the sum over the ranks */
int fcvm_mg_l2_restrict_diff(fcvm_ctx *c, const double *r1, const double *wgt, const double *sub, double *r2);
/* aggregate level after fcvm_mg_set_level2 / fcvm_mg_l2_finish: r2 = Z^T r1 (f64[n2]), e2 = A2inv r2 (dense inverse) or one cycle of the
 * sparse level, d1 = Z e2 */
int fcvm_mg_l2_restrict(fcvm_ctx *c, const double *r1, double *r2);
int fcvm_mg_l2_solve(fcvm_ctx *c, const double *r2, double *e2);
int fcvm_mg_l2_prolong(fcvm_ctx *c, const double *e2, double *d1);
/* y[m] = A x, A f32[m][lda] (device, rows zero-padded to lda = multiple of 4), x f64[n], FP64 accumulation:
 * one rank's row block of the distributed aggregate-level inverse */
int fcvm_gemv_f32(fcvm_ctx *c, int64_t m, int64_t n, int64_t lda, const float *A, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
