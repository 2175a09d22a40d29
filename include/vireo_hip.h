/*
 * vireo_hip.h -- C ABI of libvireo_hip.so: the MI355X (gfx950) implementation of the
 * vireoSNP variational-EM hot path.
 *
 * The reference (vireoSNP 0.5.9, pure Python) has no FFI layer; its boundary for this
 * path is the Python API (class Vireo, class BinomMixtureVB, vireo_wrap).  This header is
 * what a ctypes binding of that API calls.  Each entry point names the reference
 * interface (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; vrx_last_error() gives the
 *     thread-local message of the last failing call.
 *   - all pointers are HOST pointers to caller-owned memory; the library copies in/out
 *     and retains nothing across calls (device state lives behind the opaque handles).
 *   - dense arrays are C-contiguous float64 in the reference's own shapes:
 *       ID_prob (n_cell, n_donor)   GT_prob (n_var, n_donor, n_gt)
 *       beta_mu/beta_sum (theta_rows, n_gt) for Vireo [theta_rows = n_var in ASE mode
 *       else 1], (n_var, n_donor) for BinomMixtureVB.
 *   - one host thread per handle; distinct handles may be driven concurrently.
 */
#ifndef VIREO_HIP_H
#define VIREO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrx_problem vrx_problem; /* AD/DP resident in HBM, both orientations   */
typedef struct vrx_model vrx_model;     /* variational state + priors + scratch in HBM */
typedef struct vrx_comm vrx_comm;       /* RCCL communicator (restart shard)           */

#define VRX_OK 0
#define VRX_ERR_ARG (-1)
#define VRX_ERR_HIP (-2)
#define VRX_ERR_NOMEM (-3)
#define VRX_ERR_UNSUPPORTED (-4)
#define VRX_ERR_COMM (-5)

const char* vrx_last_error(void);
int vrx_device_count(int* n);
/* name / CU count / HBM bytes of a device (for logs and bench.py) */
int vrx_device_info(int device, char* name, int name_len, int* n_cu, int64_t* hbm_bytes);

/* PCI bus id ("0000:c1:00.0") of a device: the physical GPU behind a rank of the restart shard
 * (the reference's pool workers are anonymous, vireo_wrap.py:74-83; a multi-GPU record must say
 * which GPUs it ran on). */
int vrx_device_pci_bus_id(int device, char* out, int out_len /* >= 16 */);

/* ---- the sparse count matrices -------------------------------------------------------
 * AD and DP (n_var x n_cell), given once, as ONE CSC pattern (the union of both patterns,
 * row indices strictly increasing inside a column) carrying an (ad, dp) pair per entry.
 * Replaces: the scipy.sparse operands of Vireo.fit (vireoSNP/utils/vireo_model.py:278)
 * and BinomMixtureVB.fit (vireoSNP/utils/bmm_model.py:204); "BD = DP - AD"
 * (vireo_model.py:168,190,228; bmm_model.py:122,136) is never materialised.
 * Builds the variant-major (CSR) copy and the segment tables on the way in. */
int vrx_problem_create(int device, int64_t n_var, int64_t n_cell, int64_t nnz,
                       const int64_t* csc_colptr /* n_cell+1 */,
                       const int32_t* csc_rowidx /* nnz */,
                       const int32_t* ad /* nnz */, const int32_t* dp /* nnz */,
                       vrx_problem** out);
/* The same with build options.  VRX_PROBLEM_BALANCED: *balanced slabs* -- for problems that run the
 * LDS-resident passes with AD/BD words and no split rows, the builder chooses PER ROW TILE which
 * contracted rows share a slab so that every row of the tile carries about the same number of words in
 * every slab (the passes execute a round of 16 rows at the pace of its longest row: 1.62 -> ~1.2 executed
 * slots per word at c3, passes 10-17 % shorter).  Costs a one-off ~0.5 s at 1e8 entries, so it pays from a
 * few thousand iterations on the same problem (vireo_wrap's restarts, vireo_wrap.py:64-94, all run on one);
 * results differ from the unbalanced build's in summation order only (<= 1e-13 relative per update).
 * The reference has no counterpart (SciPy's CSC / CSR are what they are). */
#define VRX_PROBLEM_BALANCED 1
int vrx_problem_create2(int device, int64_t n_var, int64_t n_cell, int64_t nnz,
                        const int64_t* csc_colptr, const int32_t* csc_rowidx, const int32_t* ad,
                        const int32_t* dp, int32_t flags, vrx_problem** out);
/* info4 = { variant stream balanced (0 | 1), cell stream balanced, seconds the balancing added to the build,
 * built on the device (0 | 1) } */
int vrx_problem_build_info(vrx_problem* p, double* info4);
void vrx_problem_destroy(vrx_problem* p);

/* sum over entries with dp>0 of float32(min(log C(dp,ad), 700)), accumulated in float64.
 * Replaces: np.sum(get_binom_coeff(AD, DP))  (vireoSNP/utils/vireo_base.py:7-22; called at
 * vireo_model.py:313 and bmm_model.py:239).  Computed once per problem and cached. */
int vrx_problem_binom_const(vrx_problem* p, double* sum_out);

/* Checksums (64-bit FNV-1a) of the device-resident problem, per orientation (variant-major
 * first): packed entries, tiled stream words, stream boundaries, wave starts, row map, number of
 * gather segments.  Two builds of one input are identical iff the first five agree; the tests
 * hold the device builder (vrx_build.h) to the host builder with it. */
int vrx_problem_digest(vrx_problem* p, uint64_t* out12);
/* per-cell count of variants with dp>0 (vireoSNP/vireo.py:191, "n_vars") */
int vrx_problem_n_vars(vrx_problem* p, int32_t* out /* n_cell */);

/* ---- model ------------------------------------------------------------------------- */
#define VRX_KIND_VIREO 0 /* class Vireo            vireoSNP/utils/vireo_model.py:11 */
#define VRX_KIND_BMM 1   /* class BinomMixtureVB   vireoSNP/utils/bmm_model.py:9    */

typedef struct {
    int32_t kind;         /* VRX_KIND_*                                                  */
    int32_t n_donor;      /* K                                                           */
    int32_t n_gt;         /* T (Vireo only; vireo_model.py:61)                           */
    int32_t learn_gt;     /* Vireo.learn_GT      (vireo_model.py:65)                     */
    int32_t learn_theta;  /* Vireo.learn_theta   (vireo_model.py:67)                     */
    int32_t ase_mode;     /* Vireo.ASE_mode      (vireo_model.py:66): theta per variant  */
    int32_t fix_beta_sum; /* fix_beta_sum        (vireo_model.py:68, bmm_model.py:52)    */
    int32_t n_batch;      /* restarts held by ONE model (0 or 1: a single model; <= 16).  The
                             random restarts of vireo_wrap (vireo_wrap.py:64-87) fitted side by
                             side: state arrays gain a restart axis -- ID_prob [n_cell][R][K],
                             GT_prob [n_var][R][K][T], beta [R][rows][cols], logLik_ID
                             [n_cell][R][K] -- and fit / run_iters / step(ELBO) / get_elbo_parts
                             return one trace / value per restart                           */
} vrx_model_cfg;

int vrx_model_create(vrx_problem* p, const vrx_model_cfg* cfg, vrx_model** out);
void vrx_model_destroy(vrx_model* m);

/* Upload the variational state (attributes ID_prob, GT_prob, beta_mu, beta_sum of the
 * reference objects: vireo_model.py:84-104, bmm_model.py:69-83).  A NULL pointer leaves
 * that array as it is on the device.  GT_prob is ignored for VRX_KIND_BMM. */
int vrx_model_set_state(vrx_model* m, const double* ID_prob, const double* GT_prob,
                        const double* beta_mu, const double* beta_sum);
/* Initial state straight from the random draws (Vireo.set_initial, vireo_model.py:98-104:
 * ID_prob = normalize(rand(M, K)), GT_prob = normalize(rand(N, K, T))): uploads the RAW draws
 * and normalises them on the device over the last axis, bit-identically to NumPy's
 * X / X.sum(-1) (pairwise summation order restated in vrx_normalize_rows).  NULL arguments are
 * left untouched.  <= 128 columns. */
int vrx_model_set_state_raw(vrx_model* m, const double* ID_raw, const double* GT_raw,
                            const double* beta_mu, const double* beta_sum);
/* Staged uploads for back-to-back restarts (vireo_wrap.py:64-87 constructs and fits n_init models
 * one after the other): the raw draws of restart i + 1 travel to the device while restart i fits.
 * stage_reserve: once, from the thread that owns the model (single Vireo models): two staging
 * buffers, a copy stream.  stage_raw: buffer buf (0 | 1) <- the raw draws; MAY BE CALLED FROM A SECOND
 * HOST THREAD while vrx_model_fit runs on the same model (it touches only the staging buffer and the
 * copy stream; it returns when the host arrays may be reused).  set_state_staged: the model's
 * state <- normalised buffer buf (+ beta_mu / beta_sum, NULL = untouched): the same bits as
 * vrx_model_set_state_raw from the same draws. */
int vrx_model_stage_reserve(vrx_model* m);
int vrx_model_stage_raw(vrx_model* m, int32_t buf, const double* ID_raw, const double* GT_raw);
int vrx_model_set_state_staged(vrx_model* m, int32_t buf, const double* beta_mu, const double* beta_sum);
/* restore == 0: save (ID_prob, GT_prob, beta_mu, beta_sum) in a device-side slot;
 * restore != 0: bring them back.  The best restart so far (vireo_wrap.py:90-91) stays in HBM. */
int vrx_model_snapshot(vrx_model* m, int32_t restore);
int vrx_model_get_state(vrx_model* m, double* ID_prob, double* GT_prob, double* beta_mu,
                        double* beta_sum);
/* Restart batches (vrx_model_cfg.n_batch = R > 1).  vireo_wrap.py:64-87 builds n_init models
 * and fits them one after the other; here R of them share every sparse pass.
 * set_restart: slot r <- one restart's state, in the SINGLE-model layouts of
 * vrx_model_set_state (raw == 0) or vrx_model_set_state_raw (raw != 0: normalised on the
 * device).  copy_restart: the single model dst <- slot r of src, device to device (the
 * winner, vireo_wrap.py:90-91).  Prior tables (vrx_model_set_prior) are shared by the batch. */
int vrx_model_set_restart(vrx_model* m, int32_t r, const double* ID, const double* GT,
                          const double* beta_mu, const double* beta_sum, int32_t raw);
int vrx_model_copy_restart(vrx_model* dst, vrx_model* src, int32_t r);

/* Priors (Vireo.set_prior vireo_model.py:107-137; BinomMixtureVB.set_prior
 * bmm_model.py:87-105).  ID_prior: id_rows = 0 -> uniform 1/K (pointer ignored), 1 -> one
 * row broadcast to every cell, n_cell -> full.  GT_prior: gt_rows = 0 -> uniform 1/T,
 * 1 -> one (K,T) slab broadcast over variants, n_var -> full.  Priors are given as
 * probabilities (not logs); rows are re-normalised for the KL terms exactly like
 * scipy.stats.entropy does (vireo_model.py:237-238).  theta priors have the shape of
 * beta_mu (prior_rows = 1 or the state's row count). */
int vrx_model_set_prior(vrx_model* m, const double* ID_prior, int64_t id_rows,
                        const double* GT_prior, int64_t gt_rows,
                        const double* theta_s1_prior, const double* theta_s2_prior,
                        int64_t theta_prior_rows);

/* The coordinate-ascent loop.
 * Replaces Vireo._fit_VB (vireo_model.py:251-276) / BinomMixtureVB._fit_BV
 * (bmm_model.py:178-201): same update order, same convergence test, same "it" on exit.
 *   elbo_trace[0 .. *it_out] receives EVERY computed ELBO (without the binomial
 *   constant), i.e. one more than the reference keeps: the reference returns
 *   ELBO[:it] -- the host mirrors that truncation.
 *   warn_flags: bit0 = "lower bound decreases" seen, bit1 = "did not converge".
 * Continues from the state currently on the device (warm restart, vireo_wrap.py:94).
 * A batch model (n_batch = R) runs until its last restart has stopped; a restart that has
 * stopped is frozen (its kernels return at once).  Outputs are then per restart:
 * elbo_trace [R][max_iter], it_out [R], warn_flags [R]. */
int vrx_model_fit(vrx_model* m, int32_t max_iter, int32_t min_iter, double epsilon_conv,
                  int32_t delay_fit_theta, double* elbo_trace /* [R][max_iter] */,
                  int32_t* it_out /* [R] */, int32_t* warn_flags /* [R] */);

/* Single coordinate updates, for the public step methods:
 *   VRX_STEP_THETA  Vireo.update_theta_size (vireo_model.py:165) / bmm_model.py:133
 *   VRX_STEP_GT     Vireo.update_GT_prob    (vireo_model.py:204)
 *   VRX_STEP_ID     Vireo.update_ID_prob    (vireo_model.py:187) -> logLik_ID kept on device
 *                   BMM: get_E_logLik + update_ID_prob (bmm_model.py:118,147)
 *   VRX_STEP_LOGLIK recompute logLik_ID only (get_ELBO(None, AD, DP), vireo_model.py:227)
 *   VRX_STEP_ELBO   get_ELBO with the logLik_ID on the device (vireo_model.py:236-248) */
#define VRX_STEP_THETA 1
#define VRX_STEP_GT 2
#define VRX_STEP_ID 3
#define VRX_STEP_LOGLIK 4
#define VRX_STEP_ELBO 5
#define VRX_STEP_SOFTMAX 6 /* ID_prob <- softmax(logLik_ID on device + log ID_prior): the tail of
                              update_ID_prob alone (bmm_model.py:153-154 with logLik_ID given) */
int vrx_model_step(vrx_model* m, int32_t which, double* elbo_out /* VRX_STEP_ELBO only */);
int vrx_model_get_loglik(vrx_model* m, double* logLik_ID /* n_cell x n_donor */);
int vrx_model_set_loglik(vrx_model* m, const double* logLik_ID);
/* the four ELBO terms of the last VRX_STEP_ELBO / fit iteration:
 * LB_p, KL_ID, KL_GT, KL_theta (vireo_model.py:236-245) */
int vrx_model_get_elbo_parts(vrx_model* m, double* parts4);

/* Doublet scoring with a fitted model, the device form of predict_doublet
 * (vireoSNP/utils/vireo_doublet.py:11-82): the genotype table of all donor pairs
 * (add_doublet_GT, :105-136) is formed on the fly inside the kernel that builds the cell
 * pass's W tables; logLik / prob_out are (n_cell x C), C = K + K(K-1)/2 (singlets first,
 * pairs in itertools.combinations order).  psi*: digammas of the T + T(T-1)/2 class thetas
 * of add_doublet_theta (:85-102), (psi_rows x G).  n_gt <= 3. */
int vrx_problem_doublet(vrx_problem* p, int64_t n_donor, int64_t n_gt,
                        const double* GT_prob /* n_var x n_donor x n_gt */,
                        const double* psi1, const double* psi2, const double* psis,
                        int64_t psi_rows /* 1 or n_var */,
                        const double* ID_prior, int64_t id_rows,
                        double* logLik, double* prob_out /* may be NULL */);

/* Expected reads per variant and donor: AD @ ID_prob and DP @ ID_prob, (n_var x n_col) each
 * -- one variant pass.  Replaces the two products the command line makes for the donor VCF
 * (vireoSNP/vireo.py:240-241, "cell_dat['AD'] * res_vireo['ID_prob']"). */
int vrx_problem_donor_reads(vrx_problem* p, int64_t n_col, const double* ID_prob,
                            double* AD_reads, double* DP_reads);

/* One-shot cell log-likelihood against caller-supplied genotype/theta tables:
 *   logLik[m,c] = sum_n sum_g GT[n,c,g] * (AD[n,m] psi1[g] + BD[n,m] psi2[g] - DP[n,m] psis[g])
 * with psi*(n_rows_psi x G): the 3*G transposed products of predict_doublet
 * (vireoSNP/utils/vireo_doublet.py:53-62), C = K + K(K-1)/2 columns, G = 6 classes.
 * If prob_out is non-NULL it receives softmax_c(logLik + log ID_prior)
 * (vireo_doublet.py:67-68; ID_prior rows as in vrx_model_set_prior). */
int vrx_problem_cell_loglik(vrx_problem* p, int64_t n_col, int64_t n_class,
                            const double* GT /* n_var x n_col x n_class */,
                            const double* psi1, const double* psi2, const double* psis,
                            int64_t psi_rows /* 1 or n_var */,
                            const double* ID_prior, int64_t id_rows,
                            double* logLik /* n_cell x n_col */,
                            double* prob_out /* n_cell x n_col, may be NULL */);

/* ---- timing (bench.py roofline leg) ---------------------------------------------------
 * When enabled, every launch of a pass kernel is bracketed by hipEvents on the model's
 * stream; totals are read back after a sync.  Kernel ids: */
#define VRX_KERN_VARIANT_PASS 0 /* variant-major sparse pass (AD,DP)*ID_prob          */
#define VRX_KERN_CELL_PASS 1    /* cell-major sparse pass (AD,DP)^T*W                 */
#define VRX_KERN_DENSE 2        /* all dense/epilogue kernels together                */
#define VRX_KERN_COUNT 3
/* which kernels this model's passes run: info[0]/[1] = 1 if the variant/cell pass is
 * LDS-resident (vrx_spmm_lds) else 0 (vrx_spmm, global gathers); info[2]/[3] = entry format
 * of the variant/cell orientation (0: 4 B, 1: 8 B, 2: 12 B per non-zero); info[4]/[5] =
 * L2 tiles of the variant/cell orientation; info[6]/[7] = partial arrays of the LDS-resident
 * variant/cell pass (the most pieces its work list cuts a tile into); info[8]/[9] = its stream
 * words (padding included) per 1000 non-zeros; info[10]/[11] = extra row pieces (long rows are
 * cut into interleaved pieces); info[12] = form of the LDS-resident cell stream (0: (ad, dp)
 * pairs, 1: single-valued AD / BD entries); info[13] = form of the variant stream (0: pairs,
 * 2: AD entries then BD entries per round, 3: AD / BD entries as 2N single-accumulator virtual
 * rows -- the cell pass's kernel); info[14] = restarts in the model (n_batch); info[15]
 * = longest / mean wave stream of the tiled streams x 1000 (variant: low 16 bits, cell: high 16).  The forms follow
 * the depth of the data: AD/BD words unless a count needs so many of them (> 1.56 words per
 * entry, estimated at vrx_problem_create) that one pair word per entry is cheaper. */
int vrx_model_info(vrx_model* m, int32_t* info16);
int vrx_model_profile(vrx_model* m, int32_t enable);
int vrx_model_profile_read(vrx_model* m, double* ms_total /* VRX_KERN_COUNT */,
                           int64_t* launches /* VRX_KERN_COUNT */);
/* run `n_iter` full iterations with no convergence test and no host sync in between
 * (the timed region of bench.py); returns wall ms measured with hipEvents on the stream. */
int vrx_model_run_iters(vrx_model* m, int32_t n_iter, int32_t theta_from_iter,
                        double* elbo_trace /* n_iter */, double* ms_out);

/* ---- restart shard over RCCL ---------------------------------------------------------
 * vireo_wrap.py:74-91 farms n_init restarts to a multiprocessing.Pool and takes
 * argmax(ELBO_[-1]).  Here restarts are sharded one process per GPU and the per-restart
 * ELBOs are all-gathered over RCCL/xGMI. */
#define VRX_UNIQUE_ID_BYTES 128
int vrx_comm_unique_id(uint8_t* id /* VRX_UNIQUE_ID_BYTES */);
int vrx_comm_create(int device, int rank, int world, const uint8_t* id, vrx_comm** out);
void vrx_comm_destroy(vrx_comm* c);
/* all ranks contribute n_local doubles; out has world*n_local doubles, rank-major */
int vrx_comm_allgather_f64(vrx_comm* c, const double* local, int64_t n_local, double* out);
int vrx_comm_barrier(vrx_comm* c);
/* broadcast a host buffer of doubles from `root` (winner's state to rank 0 / everyone) */
int vrx_comm_bcast_f64(vrx_comm* c, double* buf, int64_t n, int root);

/* info4 = rank, world, device, librccl version code (ncclGetVersion; 0 = unknown) */
int vrx_comm_info(vrx_comm* c, int32_t* info4);
/* The winner's state to every rank, DEVICE TO DEVICE: vireo_wrap.py:90-94 continues with
 * `_models_all[_idx]`; here the owner of the best restart broadcasts ID_prob, GT_prob, beta_mu,
 * beta_sum straight from its model's HBM buffers into the same buffers of every other rank's model
 * (one RCCL group call, no host staging).  All ranks pass models of one shape and configuration. */
int vrx_comm_bcast_model(vrx_comm* c, vrx_model* m, int root);

/* ---- restart initialisation ------------------------------------------------------------
 * Every restart's initial ID_prob / GT_prob comes from NumPy's legacy global stream
 * (np.random.rand in vireo_model.py:98,103, one Vireo per restart in vireo_wrap.py:66-71).
 * Continue that stream in C: write the next n doubles of RandomState.random_sample() to
 * `out`, or (out == NULL) skip them, advancing the MT19937 state of np.random.get_state()
 * (key[624], pos) in place.  Host only; no GPU needed. */
int vrx_mt19937_random_sample(uint32_t* key624, int32_t* pos, double* out, int64_t n);
/* Skip n doubles of the same stream.  A rank of the restart shard (vireo_wrap.py:66-71: restart i
 * on rank i % world) must pass every other rank's draws; far skips are a JUMP of the generator --
 * the polynomial x^(624 R) mod phi of its GF(2) transition matrix applied to the state, cached per
 * skip length, ~1 ms whatever the distance -- instead of R regenerations.  jump: 1 = jump
 * whenever the skip spans >= 3 regenerations, 0 = always step (the specification the jump is
 * tested against, bit for bit), -1 = the library's threshold (what out == NULL above uses). */
int vrx_mt19937_skip(uint32_t* key624, int32_t* pos, int64_t n, int32_t jump);
/* np.sum() of a contiguous float32 array, bit for bit (NumPy's pairwise summation per 8192-element
 * iterator chunk).  vrx_problem_binom_const adds the float32 terms of get_binom_coeff
 * (vireo_base.py:7-22) with it, like vireo_model.py:313 does with np.sum.  Host only. */
int vrx_np_sum_f32(const float* a, int64_t n, float* out);

/* ---- MatrixMarket input (f4) -----------------------------------------------------------
 * cellSNP.tag.{AD,DP}.mtx and the vartrix matrices, which the reference reads with
 * scipy.io.mmread (io_utils.py:57,72-73): a multi-threaded parser of `coordinate`
 * `integer` / `real` (integral values) / `pattern`, `general` files.  vrx_mtx_header returns
 * the shape and entry count; vrx_mtx_read fills caller-allocated COO arrays (0-based, file
 * order, duplicates kept).  n_threads <= 0: all cores (at most 64).  Host only. */
int vrx_mtx_header(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
/* AD and DP (canonical CSC, any of the dtypes the reference's loaders produce) -> the union
 * pattern with an (ad, dp) pair per entry, the input of vrx_problem_create: what
 * `BD = DP - AD` and the separate AD / DP products of the reference (vireo_model.py:167-170)
 * are replaced by.  Two passes: with rowidx == NULL the per-column entry counts go to
 * colptr[1..n_cell]; the caller accumulates them and calls again to fill.  *_ptr64 / *_idx64:
 * the index arrays are int64 (else int32); *_kind: counts are 0 int32, 1 int64, 2 float64.
 * Host only, multi-threaded. */
int vrx_merge_counts(int64_t n_var, int64_t n_cell, const void* ad_ptr, const void* ad_idx,
                     const void* ad_dat, int ad_ptr64, int ad_idx64, int ad_kind,
                     const void* dp_ptr, const void* dp_idx, const void* dp_dat, int dp_ptr64,
                     int dp_idx64, int dp_kind, int64_t* colptr, int32_t* rowidx, int32_t* ad,
                     int32_t* dp, int n_threads);
int vrx_mtx_read(const char* path, int64_t nnz, int32_t* row, int32_t* col, int32_t* val,
                 int n_threads);
/* The COO arrays of vrx_mtx_read -> CSC: `mmread(...).tocsc()` of read_cellSNP / read_vartrix
 * (io_utils.py:57,72-73; SciPy's single-threaded coo_tocsr) as a stable counting sort by column
 * on all cores.  colptr [n_cols + 1], rowidx / data [nnz] (data int64 like mmread's).
 * *canonical = 1 when every column's rows come out strictly increasing (no duplicates, file was
 * row-major: what cellSNP writes); 0: the caller must sort / sum duplicates (SciPy does).
 * Host only. */
int vrx_coo_to_csc(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* row,
                   const int32_t* col, const int32_t* val, int64_t* colptr, int32_t* rowidx,
                   int64_t* data, int32_t* canonical, int n_threads);
/* The inverse of vrx_mtx_read: a `coordinate integer general` file from 0-based COO arrays, in
 * the given order (the format of cellSNP's cellSNP.tag.{AD,DP}.mtx that read_cellSNP loads,
 * io_utils.py:57; the reference itself never writes one).  Multi-threaded formatting; used by
 * bench.py's end-to-end leg and the parser's round-trip test.  Host only. */
int vrx_mtx_write(const char* path, int64_t n_rows, int64_t n_cols, int64_t nnz,
                  const int32_t* row, const int32_t* col, const int32_t* val);

/* ---- text writers of the command (host only) ----------------------------------------------
 * vrx_write_table: prob_singlet.tsv / prob_doublet.tsv of write_donor_id (io_utils.py:147-170):
 * `header`, then per row its label and `cols` numbers in the printf format `fmt` ("%.2e"), tab
 * separated.  vrx_write_vcf_records: the donor genotype VCF of write_VCF (vcf_utils.py:234-296)
 * with the tags GT:AD:DP:PL -- `head`, then per variant its prefix (fixed columns + FORMAT) and
 * per sample <0/0|1/0|1/1>:AD:DP:PL,PL,PL from integer arrays.  gz != 0 writes gzip (one member
 * per chunk of rows).  Chunks are formatted and deflated on several threads. */
int vrx_write_table(const char* path, const char* header, const char* names,
                    const int64_t* name_off /* rows + 1 */, const double* table, int64_t rows,
                    int64_t cols, const char* fmt, int32_t gz);
int vrx_write_vcf_records(const char* path, const char* head, const char* prefix,
                          const int64_t* prefix_off /* n_var + 1 */, const int8_t* call,
                          const int64_t* ad, const int64_t* dp, const int64_t* pl /* [..][3] */,
                          int64_t n_var, int64_t n_sample, int32_t gz);

#ifdef __cplusplus
}
#endif
#endif /* VIREO_HIP_H */
