/*
 * metmhn_amd C ABI - MI355X (gfx950) likelihood / gradient engine for metMHN.
 *
 * The reference (cbg-ethz/metMHN @ 2024_08_07) has no FFI layer: its boundary is the
 * Python call surface of metmhn/regularized_optimization.py and metmhn/jx/{kronvec,likelihood,vanilla}.py.  Every
 * entry point below replaces one of those Python functions; the ctypes binding lives in
 * metmhn_amd/_lib.py and the reference-side stub a maintainer would add is shown in
 * INTEGRATION.md.  All pointers are caller-owned HOST buffers, row-major, fp64 at the
 * interface whatever the engine's internal dtype; every function returns 0 on success
 * and a non-zero status otherwise (message: mmhn_last_error(), thread-local).  A handle
 * is bound to one GPU and is not thread-safe (the reference is single-threaded:
 * scipy.optimize.minimize calls score_and_grad_reg synchronously,
 * regularized_optimization.py:328-330).
 *
 * Layout conventions (regularized_optimization.py:63-66, metmhn/jx/kronvec.py:223-250):
 *   dat    int8 [n_pat][2n+3]  = PT_0,MT_0,...,PT_{n-1},MT_{n-1},seed, order, type
 *   state  int8 [2n+1]  joint observation;  [n+1] for the single-tumour functions
 *   vectors have 2^k entries, k = #ones in state; index bit b <-> b-th active slot.
 *   log_theta fp64 [n+1][n+1], log_d_p / log_d_m fp64 [n+1].
 */
#ifndef METMHN_AMD_H
#define METMHN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmhn_engine* mmhn_handle;

enum { MMHN_F64 = 0, MMHN_F32 = 1 };

/* ---- lifetime ------------------------------------------------------------------- */
int mmhn_create(int device_id, int n_mut, int dtype, mmhn_handle* out);
void mmhn_destroy(mmhn_handle h);
const char* mmhn_last_error(void);
/* upper bound on device workspace used per batch of patients (default: 70 % of free HBM) */
int mmhn_set_workspace_limit(mmhn_handle h, size_t bytes);

/* ---- cohort objective -------------------------------------------------------------
 * mmhn_set_cohort copies `dat`, derives every patient's bit layout and batches once.
 * mmhn_score            <-> regularized_optimization.score           (:55-130)
 * mmhn_score_and_grad   <-> regularized_optimization.score_and_grad  (:163-267)
 * mmhn_cohort_sums returns the UNWEIGHTED partial sums of this handle's patients so that
 * patient shards on several GPUs can be combined with one all-reduce:
 *   sums[0]            = sum of log-probs of the rows with type != 0 ("EM")
 *   sums[1]            = sum of log-probs of the rows with type == 0 ("NM")
 *   sums[2]            = number of rows with seeding == 1,  sums[3] = number of rows
 *   then d_theta_EM [N*N], d_theta_NM [N*N], d_dp_EM [N], d_dp_NM [N], d_dm_EM [N]
 * (N = n_mut + 1; 4 + 2*N*N + 3*N doubles; gradient blocks are zero if with_grad == 0).
 */
int mmhn_set_cohort(mmhn_handle h, const int8_t* dat, int64_t n_pat, int n_cols);
int mmhn_score(mmhn_handle h, const double* log_theta, const double* log_d_p, const double* log_d_m,
               double perc_met, double* score);
int mmhn_score_and_grad(mmhn_handle h, const double* log_theta, const double* log_d_p,
                        const double* log_d_m, double perc_met, double* score, double* d_theta,
                        double* d_dp, double* d_dm);
int mmhn_cohort_sums(mmhn_handle h, const double* log_theta, const double* log_d_p,
                     const double* log_d_m, int with_grad, double* sums);
/* The same in two halves: _begin issues the evaluation (and the all-reduce) and returns without waiting - the
 * parameter arrays may be reused at once -, _end waits and delivers.  In between the caller is free to do host work: the
 * reference computes its penalty terms on the host after the score (regularized_optimization.py:292-298), here they run
 * next to the GPU.  One evaluation in flight per handle. */
int mmhn_cohort_sums_begin(mmhn_handle h, const double* log_theta, const double* log_d_p,
                           const double* log_d_m, int with_grad);
int mmhn_cohort_sums_end(mmhn_handle h, double* sums);
/* The same evaluation with the EM / NM weighting of regularized_optimization.py:256-266 applied on the device:
 * wsums[1 + N*N + 2 N] = [w s_EM + s_NM, w G_EM + G_NM, w p_EM + p_NM, w m_EM]; the caller divides by
 * n_full = w n_em + n_nm.  w needs the GLOBAL counts only (known to every rank once the cohort is set), so with a
 * communicator attached the all-reduce carries 1 + N^2 + 2 N doubles (484 at n = 20). */
int mmhn_cohort_wsums_begin(mmhn_handle h, const double* log_theta, const double* log_d_p,
                            const double* log_d_m, int with_grad, double w);
int mmhn_cohort_wsums_end(mmhn_handle h, double* wsums);
/* One more double riding in the SAME all-reduce as the wsums buffer (ABI 5): the value set here travels with the NEXT
 * mmhn_cohort_wsums_begin, is summed over the ranks of the communicator, and is read back after its _end.  The Python host
 * uses it for the "this rank's cohort array was edited in place" bit of its layout cache, which used to be a second
 * collective per evaluation (metmhn_amd/regularized_optimization.py: _result). */
int mmhn_set_reduce_flag(mmhn_handle h, double value);
int mmhn_get_reduce_flag(mmhn_handle h, double* summed);
/* per-patient results of the current cohort (tests): lp[n_pat], and if non-NULL
 * d_theta[n_pat][N*N], d_dp[n_pat][N], d_dm[n_pat][N]  (ssr._g_coupled_*, _grad_*_obs) */
int mmhn_patient_grads(mmhn_handle h, const double* log_theta, const double* log_d_p,
                       const double* log_d_m, double* lp, double* d_theta, double* d_dp, double* d_dm);

/* ---- joint PT/MT primitives (metmhn/jx/kronvec.py, likelihood.py) ------------------
 * mmhn_kronvec        <-> kronvec.kronvec        (:499-539)   y = Q p, flags diag / transpose
 * mmhn_kron_diag      <-> kronvec.kron_diag      (:964-999)
 * mmhn_diag_scal      <-> kronvec.diag_scal_p/m  (:574-602, :646-671)  which: 0 = p, 1 = m
 * mmhn_obs_states     <-> kronvec.obs_states + jnp.where(size=) (:1056-1095, likelihood.py:280)
 *                         ascending compatible indices; *count receives how many
 * mmhn_resolvent      <-> likelihood.R_i_inv_vec (:231-262)   (D_p + D_m - Q)^-1 x
 * mmhn_x_partial_Q_y  <-> likelihood.x_partial_Q_y (:163-201) G[N][N]
 * mmhn_x_partial_D_y  <-> likelihood.x_partial_D_y (:204-228) (takes log_d_p, log_d_m in THAT order)
 * mmhn_partial_diag_scal <-> kronvec.partial_diag_scal_p / _m (:605-644, :674-710): (dD/dlog d[i]) * p,
 *                         which: 0 = p, 1 = m; i in [0, n_mut] (n_mut = the seeding entry)
 * mmhn_kronvec_batched: the same product for `batch` vectors p[b][2^k] of ONE restricted space (what a vmapped
 *   kronvec.kronvec does; SURVEY 8b "+ _batched"): one launch over every tile of every vector into y[b][2^k].  The
 *   device output starts as a NaN pattern and is written completely by that launch (tiles where Q_off has no entries
 *   are zeroed inside it) - it is the launch sequence mmhn_bench_kronvec times.
 */
int mmhn_kronvec_batched(mmhn_handle h, const double* log_theta, const int8_t* state, int64_t batch, const double* p,
                         double* y, int diag, int transpose);
/* one sweep of likelihood.R_i_inv_vec's Jacobi iteration (likelihood.py:253-255), batched as above:
 * y[b] = lidg * (Q_off p[b] + rhs[b]) (transpose: Q_off^T), lidg = 1 / (D_p + D_m - diag Q); the launch
 * mmhn_bench_kronvec times with jacobi != 0 */
int mmhn_jacobi_step_batched(mmhn_handle h, const double* log_theta, const double* log_d_p, const double* log_d_m,
                             const int8_t* state, int64_t batch, const double* p, const double* rhs, double* y,
                             int transpose);
int mmhn_kronvec(mmhn_handle h, const double* log_theta, const int8_t* state, const double* p,
                 double* y, int diag, int transpose);
int mmhn_kron_diag(mmhn_handle h, const double* log_theta, const int8_t* state, double* out);
int mmhn_diag_scal(mmhn_handle h, const double* log_d, const int8_t* state, const double* p,
                   double* y, int which);
int mmhn_obs_states(mmhn_handle h, const int8_t* state, int pt_first, int64_t* idx, int64_t* count);
int mmhn_resolvent(mmhn_handle h, const double* log_theta, const double* log_d_p,
                   const double* log_d_m, const int8_t* state, const double* x, double* y,
                   int transpose);
int mmhn_x_partial_Q_y(mmhn_handle h, const double* log_theta, const int8_t* state, const double* x,
                       const double* y, double* G);
int mmhn_x_partial_D_y(mmhn_handle h, const double* log_d_p, const double* log_d_m,
                       const int8_t* state, const double* x, const double* y, double* d_dp,
                       double* d_dm);
int mmhn_partial_diag_scal(mmhn_handle h, const double* log_d, const int8_t* state, const double* p,
                           int i, int which, double* y);

/* ---- single-tumour primitives (metmhn/jx/vanilla.py); state has n+1 entries ---------
 * mmhn_v_kronvec       <-> vanilla.kronvec        (:78-106)
 * mmhn_v_resolvent     <-> vanilla.R_inv_vec      (:269-305)  d_rates == NULL means 1
 * mmhn_v_x_partial_Q_y <-> vanilla.x_partial_Q_y  (:328-393)  G[N][N], d_diag[N]
 * mmhn_v_kron_diag     <-> vanilla.kron_diag      (:247-260)  diag(Q) * diag  (diag == NULL means ones)
 * mmhn_v_scal_d_pt     <-> vanilla.scal_d_pt      (:125-142)  observation rates of an MT-only datapoint:
 *                          out_p = [seeding clear] prod d_p * vec, out_m = [seeding set] d_m[n] prod d_m * vec
 * mmhn_v_d_scal_d_pt   <-> vanilla.d_scal_d_pt    (:144-187)  their derivatives w.r.t. log d[i]
 * mmhn_v_x_partial_D_y <-> vanilla.x_partial_D_y  (:190-203)  (d_dp[N], d_dm[N]); argument order (log_d_p, log_d_m)
 * The three scal_d_pt functions need state[n_mut] == 1 (the reference applies a 2-state factor to the seeding bit).
 */
int mmhn_v_kronvec(mmhn_handle h, const double* log_theta, const int8_t* state, const double* p,
                   double* y, int diag, int transpose);
int mmhn_v_resolvent(mmhn_handle h, const double* log_theta, const int8_t* state,
                     const double* d_rates, const double* x, double* y, int transpose);
int mmhn_v_x_partial_Q_y(mmhn_handle h, const double* log_theta, const int8_t* state,
                         const double* x, const double* y, double* G, double* d_diag);
int mmhn_v_kron_diag(mmhn_handle h, const double* log_theta, const int8_t* state, const double* diag,
                     double* out);
int mmhn_v_scal_d_pt(mmhn_handle h, const double* log_d_p, const double* log_d_m, const int8_t* state,
                     const double* vec, double* out_p, double* out_m);
int mmhn_v_d_scal_d_pt(mmhn_handle h, const double* log_d_p, const double* log_d_m, const int8_t* state,
                       const double* vec, int i, double* out_p, double* out_m);
int mmhn_v_x_partial_D_y(mmhn_handle h, const double* log_d_p, const double* log_d_m,
                         const int8_t* state, const double* x, const double* y, double* d_dp,
                         double* d_dm);

/* ---- patient shards on several GPUs (one process and one engine per GPU) ----------------
 * The reference is single-process; its cohort sum (regularized_optimization.py:256-266) is what shards.
 * Rank 0 draws an id (mmhn_comm_unique_id, 128 bytes) and hands it to every rank by any host channel; after
 * mmhn_comm_init every mmhn_cohort_sums / mmhn_score / mmhn_score_and_grad of the handle returns the sums over
 * ALL ranks' cohorts: one RCCL all-reduce of the 4 + 2 N^2 + 3 N doubles on the engine's stream per evaluation,
 * no host staging.  Every rank must make the same calls in the same order (collective semantics).
 */
int mmhn_comm_unique_id(void* id128);
int mmhn_comm_init(mmhn_handle h, const void* id128, int rank, int n_ranks);
int mmhn_comm_destroy(mmhn_handle h);

/* ---- simulation ---------------------------------------------------------
 * mmhn_simulate: Gillespie sampler of the joint PT/MT process, one trajectory per thread; replaces
 * metmhn/simulations.py:117-147 (`simulate_dat`) and :87-114 (`simulate_orders`).
 *   log_theta [N][N], pt_d_ef / mt_d_ef [N]: log-parameters as the reference passes them (N = n_mut + 1)
 *   seed: Philox key; the samples depend on (seed, n_sim index) only
 *   dat_out    int8 [n_sim][2 n_mut + 2] = [PT_0, MT_0, ..., seeding, order (0 unpaired / 1 PT first / 2 MT first)]
 *   orders_out int8 [n_sim][2 N + 2] event sequences padded with -99 (events numbered as simulations.py:100-107), or NULL
 */
int mmhn_simulate(mmhn_handle h, const double* log_theta, const double* pt_d_ef, const double* mt_d_ef,
                  int64_t n_sim, uint64_t seed, int8_t* dat_out, int8_t* orders_out);

/* ---- measurement -------------------------------------------------------------------
 * mmhn_bench_kronvec: `batch` resident copies of a 2^k vector, `iters` back-to-back
 * launches of mmhn_kronvec_batched's launch (diag = 0: y = Q_off p into a NaN-filled y, every tile of every vector,
 * structurally zero tiles zeroed inside the launch) or of the fused Jacobi step if jacobi != 0, timed
 * with HIP events on the engine's stream; returns the average launch duration in ms.
 * tiles (optional): [0] = tiles per launch where Q_off has entries, [1] = tiles per launch.
 * mmhn_get_counters: cumulative figures since mmhn_reset_counters, per class of dominant kernel (events recorded
 * on the engine's stream around every launch).
 */
enum { MMHN_K_OTHER_SOLVE = 0,  /* tile solves / Jacobi sweeps of the single-tumour problems (k_csolve, k_tsolve, k_sweep), API calls */
       MMHN_K_PSOLVE_FWD = 1,   /* k_wsolve / k_psolve2 forward: a chain of patients / one patient per workgroup, (D - Q) pi = e_0 */
       MMHN_K_PSOLVE_ADJ = 2,   /* ... adjoint: (D - Q)^T q = rhs */
       MMHN_K_PCLASS = 3,       /* k_wclass / k_pclass: class marginals of pi (x) q */
       MMHN_K_CSOLVE_FWD = 4,   /* k_csolve forward: the tiles of the joint problems on the tile route, one cooperative launch */
       MMHN_K_CSOLVE_ADJ = 5,   /* ... adjoint */
       MMHN_K_COUNT = 6 };
typedef struct {
  double ms;         /* total duration of the launches (HIP events on the engine's stream) */
  int64_t launches;
  double alg_bytes;  /* algorithmic bytes of those launches: solves = the solution written once (live tiles),
                        marginals = pi and q read once, Jacobi sweep = 4 * 2^k * sizeof(dtype) per vector */
} mmhn_kernel_counter;
typedef struct {
  mmhn_kernel_counter kernel[MMHN_K_COUNT];
  double eval_ms;    /* host wall time spent inside evaluations */
  int64_t evals;
  int32_t comm_ranks; /* ranks of the RCCL communicator attached by mmhn_comm_init as RCCL itself reports them (ncclCommCount), 0: none */
  int32_t comm_rank;  /* this engine's rank in it (ncclCommUserRank), -1: none */
} mmhn_counters;
/* ABI version of this header: bumped whenever an exported signature or structure changes (4: mmhn_bench_kronvec has its
 * `tiles` argument, mmhn_debug_lane_moves exists; 5: mmhn_counters has six kernel classes and the communicator's size / rank).  A client built against another header must refuse to run:
 * mmhn_abi_version() != MMHN_ABI_VERSION (metmhn_amd/_lib.py checks it on load). */
#define MMHN_ABI_VERSION 5
int mmhn_abi_version(void);
int mmhn_bench_kronvec(mmhn_handle h, const double* log_theta, const int8_t* state, int64_t batch,
                       int iters, int transpose, int jacobi, double* ms_per_launch, int64_t* tiles);
/* device-memory bandwidth of this GPU for a plain 16-byte-per-lane stream over arrays of `bytes` each
 * (kind 0: copy, 1: triad a = b + s c), GB/s of the 2 x / 3 x bytes moved: the measured denominator next to the
 * nominal HBM peak */
int mmhn_bench_stream(mmhn_handle h, size_t bytes, int iters, int kind, double* gbps);
int mmhn_get_counters(mmhn_handle h, mmhn_counters* out);
int mmhn_reset_counters(mmhn_handle h);
/* diagnostic of the window-layout solve (csrc/wsolve.h): out[6][64], out[i][lane] = the lane whose value `lane` receives
 * through the exchange along lane bit i (DPP / swizzle / permute forms) - lane ^ (1 << i) on every lane that has the
 * move (forward: bit i set, transposed: bit i clear).  No reference counterpart (the reference has no lanes). */
int mmhn_debug_lane_moves(mmhn_handle h, int transposed, int* out);

#ifdef __cplusplus
}
#endif
#endif /* METMHN_AMD_H */
