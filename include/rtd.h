/*
 * rtd.h -- C ABI of the MI355X discrete-ordinate radiative-transfer engine (librtd.so).
 *
 * This is the drop-in boundary for the hot path of LDEO-CREW/Pythonic-DISORT:
 *
 *   _assemble_intensity_and_fluxes   src/PythonicDISORT/_assemble_intensity_and_fluxes.py:8-32
 *     -> _solve_for_gen_and_part_sols   src/PythonicDISORT/_solve_for_gen_and_part_sols.py:5-16
 *     -> _solve_for_coeffs              src/PythonicDISORT/_solve_for_coeffs.py:8-29
 *     -> closures u / u0 / flux_up / flux_down   _assemble_intensity_and_fluxes.py:170,334,446,527
 *
 * The reference has no FFI of its own (it is pure Python over NumPy/SciPy); the entry points
 * below are what a ctypes binding of that path binds.  They take the *prepared* arguments of
 * `_assemble_intensity_and_fluxes` (delta-M scaled, source-rescaled; pydisort.py:184-372), with a
 * leading column axis C added to every per-atmosphere array so that many independent
 * atmospheric columns are solved in one call.  C = 1 reproduces one `pydisort()` call.
 *
 * Conventions: plain C, all arrays contiguous row-major float64, caller-allocated HOST memory
 * unless a name says "dev"; sizes are int32; every function returns 0 on success or a nonzero
 * status (rtd_last_error() gives the text).  No exceptions cross the boundary.  A plan owns its
 * device buffers and one HIP stream; plans are independent and may be used from different host
 * threads (one thread per plan).
 */
#ifndef RTD_H
#define RTD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtd_plan rtd_plan; /* opaque */

typedef struct {
  int32_t ncols;    /* C : independent atmospheric columns                                  */
  int32_t nlayers;  /* NLayers                                                              */
  int32_t nquad;    /* NQuad (streams, even, 2 ... 128; N = NQuad/2 per hemisphere)         */
  int32_t nleg;     /* NLeg  (phase-function moments used, <= NQuad)                        */
  int32_t nfourier; /* NFourier (1 when only fluxes are wanted)                             */
  int32_t nscoeffs; /* Nscoeffs: polynomial order+1 of the isotropic source (0 = none)      */
  int32_t nbdrf;    /* NBDRF: number of tabulated BDRF Fourier modes (0 = black surface)    */
  int32_t beam;     /* there_is_beam_source (any I0 > 0)                                    */
} rtd_dims;

/* --- library / device ------------------------------------------------------------------- */
int rtd_version(void);
const char* rtd_last_error(void);
int rtd_device_count(int32_t* count);
/* free and total memory of a device right now (hipMemGetInfo; either pointer may be NULL): what a caller sizes a retention
 * budget or a pool limit against */
int rtd_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);

/* --- plan life cycle ---------------------------------------------------------------------- */
/* Allocates every device buffer the path needs for `dims` on HIP device `device`.
 * Inputs and results are held for all `ncols` columns; the intermediates of the solve (eigenvector blocks, BC workspace:
 * ~8 MB per 20-layer 32-stream column) are held for one *window* of columns at a time and the kernels run window after
 * window on the plan's stream, so ncols is bounded by the inputs and results only (10^5 columns of BASELINE.json's
 * configs[3] need 2.3 GB).  rtd_plan_create sizes the window itself (environment RTD_WORK_BYTES, default 24 GiB of
 * intermediates); rtd_plan_create_windowed takes it from the caller (work_columns <= 0: automatic).
 * A plan of more than one window pipelines them: it keeps the eigen stage's hand-off buffers twice and runs the eigen
 * stage of window w + 1 on a second HIP stream beside the boundary-condition and evaluation stage of window w (events
 * order the two; RTD_NO_PIPELINE=1 in the environment runs the windows one after the other).  Nothing of this is visible
 * at the ABI: every call below is ordered on the plan's stream as before, rtd_plan_synchronize covers both. */
int rtd_plan_create(const rtd_dims* dims, int32_t device, rtd_plan** plan);
int rtd_plan_create_windowed(const rtd_dims* dims, int32_t device, int32_t work_columns, rtd_plan** plan);
/* A plan whose evaluators can be called again after a solve WITHOUT solving again, whatever its window count -- the contract
 * of the reference's closures, which keep GC_collect, K_collect, B_collect of the solve and evaluate any (tau, phi) from them
 * (_assemble_intensity_and_fluxes.py:170-262).  Everything rtd_plan_evaluate / rtd_plan_get_tensors read of a solve (the
 * eigenvector blocks Y, A, the eigenvalues, particular solutions and boundary-condition coefficients: 3.1 of the 8.4 MB of
 * intermediates of a 20-layer 32-stream column) is then held for ALL columns; only the boundary-condition workspace stays
 * windowed.  retain_bytes: budget for that state in bytes; < 0: three tenths of the device memory that is free at creation;
 * 0: never (= rtd_plan_create_windowed).
 * A batch whose full state does not fit the budget is created in the LEAN retained form when that fits (10 streams and up): only
 * what cannot be recomputed cheaply is held for all columns -- the boundary-condition coefficients, k, E = exp(-k dtau), the beam
 * and thermal particular solutions: 0.5 MB per 20-layer 32-stream column, 50 GB for BASELINE's 10^5-column batch -- and
 * rtd_plan_evaluate re-runs the EIGEN stage (never the boundary-condition solve) for the layers its points touch, window by window,
 * in the wavefront composition the solve had them in: results bit-identical to the full form; an evaluation at one depth per column
 * costs about a tenth of a solve.  Else the plan is a plain windowed one, whose rtd_plan_evaluate solves the windows again.
 * rtd_plan_retained tells which it is: 1 full (one-window plans: always), 2 lean, 0 neither.
 * rtd_plan_create_retained_form: form 0 = as above (full, else lean, else neither), 1 = full or neither, 2 = lean or neither. */
int rtd_plan_create_retained(const rtd_dims* dims, int32_t device, int32_t work_columns, int64_t retain_bytes, rtd_plan** plan);
int rtd_plan_create_retained_form(const rtd_dims* dims, int32_t device, int32_t work_columns, int64_t retain_bytes, int32_t form,
                                  rtd_plan** plan);
int rtd_plan_retained(rtd_plan* plan, int32_t* retained);
/* columns per window and number of windows of a plan */
int rtd_plan_windows(rtd_plan* plan, int32_t* work_columns, int32_t* nwindows);
int rtd_plan_destroy(rtd_plan* plan);
int rtd_plan_synchronize(rtd_plan* plan);
/* bytes of device memory held by the plan */
int rtd_plan_device_bytes(rtd_plan* plan, int64_t* bytes);
/* Device memory of destroyed plans.  Small arenas (blocks up to 64 MB, 512 MB in all) are always kept for the next plan of about
 * the same size: one-column calls create and destroy a plan per call.  LARGE blocks are kept only when the caller asks for it --
 * rtd_pool_set_limit, or RTD_POOL_BYTES in the environment; the default limit is 0: a library imported under someone else's
 * process must not sit on gigabytes of a GPU it shares with that process's allocator.  With a limit (bytes PER DEVICE; < 0: an
 * eighth of that device's memory; oldest blocks of the same device out first), creating and destroying a plan of gigabytes per call
 * costs neither hipMalloc nor the seconds-long stalls the runtime's lazy reclaim of freed gigabytes puts on later allocations
 * (profiles/r05_alloc_outliers.txt: 4 of 40 creations of a 14 GB plan stalled for 1.8 ... 4.7 s without, none with) -- a serving
 * loop opts in once.  No counterpart in the reference (NumPy's allocator).
 * rtd_pool_set_limit: the new limit (previous may be NULL; lowering it frees what no longer fits); rtd_pool_bytes: what is held
 * right now; rtd_pool_trim: give it back to the runtime (device -1: every device; released may be NULL).  An allocation of the
 * library that fails with out-of-memory trims the pool and is tried again by itself. */
int rtd_pool_set_limit(int64_t bytes, int32_t device, int64_t* previous);
int rtd_pool_bytes(int64_t* cached);
int rtd_pool_trim(int32_t device, int64_t* released);
/* Which columns of the batch failed numerically in the last solve: status[ncols], 0 = fine.  Bits 1..4 of the low byte:
 * 2 eigen-iteration not converged, 4 non-positive Cholesky pivot / non-finite eigenvalue, 8 singular boundary-condition
 * system, 16 non-finite beam particular solution -- raised by Fourier mode 0; the same bits shifted left by 8: raised by
 * a mode m > 0 (the fluxes and u0 of such a column are still valid: they come from mode 0; the reference returns NaN
 * intensities and valid fluxes there, _solve_for_gen_and_part_sols.py:186).  A column's failure does not touch the other
 * columns' results; rtd_plan_fetch / _evaluate report RTD_ERR_NUMERIC when any column they return has failed. */
int rtd_plan_get_column_status(rtd_plan* plan, int32_t* status);

/* --- inputs (host -> device) -------------------------------------------------------------- */
/* Quadrature of one hemisphere: mu_arr_pos[N], W[N]  (pydisort.py:304-306). */
int rtd_plan_set_quadrature(rtd_plan* plan, const double* mu_pos, const double* weights);

/* Per-column prepared arguments of _assemble_intensity_and_fluxes (_assemble.py:8-32):
 *   scaled_omega      [C][L]        scaled_omega_arr
 *   tau               [C][L]        tau_arr (unscaled lower boundaries, for layer lookup)
 *   scaled_tau_with_0 [C][L+1]      scaled_tau_arr_with_0
 *   scale_tau         [C][L]        scale_tau
 *   wleg              [C][L][NLeg]  weighted_scaled_Leg_coeffs
 *   mu0, I0, phi0, rescale [C]      beam parameters (I0 already divided by rescale_factor)
 *   b_pos, b_neg      [C][M][N]     Dirichlet BCs per Fourier mode (NULL = 0); _coeffs.py:142-158
 *   s_poly            [C][L][Ns]    scaled_s_poly_coeffs (NULL when Ns = 0)
 *   bdrf_q            [C][NBDRF][N][N]  q^m(mu_i, mu_j)     } BDRF_Fourier_modes evaluated on the
 *   bdrf_q0           [C][NBDRF][N]     q^m(mu_i, mu0)      } quadrature grid; _coeffs.py:121-134
 */
int rtd_plan_set_columns(rtd_plan* plan, const double* scaled_omega, const double* tau,
                         const double* scaled_tau_with_0, const double* scale_tau, const double* wleg,
                         const double* mu0, const double* I0, const double* phi0, const double* rescale,
                         const double* b_pos, const double* b_neg, const double* s_poly,
                         const double* bdrf_q, const double* bdrf_q0);

/* The same batch from RAW inputs: the preparation of pydisort.py:316-372 (delta-M scaling of tau, omega and the moments,
 * the thermal source polynomial in the scaled optical depth, the rescaling of every source by the largest one) runs on
 * the device.  For throughput batches: the host hands over what the user gave, nothing is computed per column on the CPU.
 *   tau_arr, omega_arr, f_arr [C][L]   as pydisort()'s arguments (f_arr = 0: no delta-M scaling)
 *   leg_all   [C][L][nleg_all]         Leg_coeffs_all (nleg_all >= nleg; the first nleg moments are used)
 *   mu0, I0, phi0 [C]                  I0 NOT rescaled
 *   b_pos, b_neg  [C][M][N]            Dirichlet BCs per Fourier mode, NOT rescaled (NULL = 0)
 *   s_poly    [C][L][Ns]               s_poly_coeffs as given by the user (NULL when Ns = 0)
 *   bdrf_q, bdrf_q0                    as in rtd_plan_set_columns */
int rtd_plan_set_columns_raw(rtd_plan* plan, const double* tau_arr, const double* omega_arr, const double* leg_all,
                             int32_t nleg_all, const double* f_arr, const double* mu0, const double* I0,
                             const double* phi0, const double* b_pos, const double* b_neg, const double* s_poly,
                             const double* bdrf_q, const double* bdrf_q0);

/* BDRF Fourier modes formed on the device.  The reference takes the surface as callables
 * BDRF_Fourier_modes[m](mu, -mu') evaluated on the quadrature grid (_solve_for_coeffs.py:121-134); for a reflectance
 * rho(mu, mu', dphi) its tests integrate every mode on the host (pydisotest/6_test.py:194-201).  Here the caller passes
 * rho sampled at dphi_p = 2 pi p / nphi, p < nphi, and the device forms the plan's NBDRF modes
 *   q^m = (2 - delta_m0)/nphi  sum_p rho_p cos(m dphi_p)      (trapezoid rule of 1/((1+delta_m0) pi) Int rho cos(m dphi))
 *   rho_qq [C][N][N][nphi]   rho(mu_i, mu_j, dphi_p)
 *   rho_q0 [C][N][nphi]      rho(mu_i, mu0, dphi_p)   (NULL when there is no beam)
 * Call after rtd_plan_set_columns (whose bdrf_q / bdrf_q0 may then be NULL) and before rtd_plan_solve. */
int rtd_plan_set_bdrf_samples(rtd_plan* plan, int32_t nphi, const double* rho_qq, const double* rho_q0);

/* Fourier-mode shard (the partition for fewer columns than GPUs): the plan's nfourier local modes
 * stand for the modes first, first + stride, ... of `total` (the reference's NFourier; modes are independent through
 * the eigen stage and the boundary-condition solve, _gen_part.py:91, _coeffs.py:111, and meet only in the Fourier sum,
 * _assemble.py:256-260).  b_pos / b_neg of rtd_plan_set_columns are then those of the local modes.  The evaluators
 * return the shard's partial sums (u0, fluxes and the thermal terms from the shard that owns mode 0 only); summing the
 * shards gives the full result: rtd_comm_allreduce_results.  Default: first 0, stride 1, total = nfourier. */
int rtd_plan_set_mode_shard(rtd_plan* plan, int32_t first, int32_t stride, int32_t total);

/* Declares the uploaded inputs NEW without uploading them again: the next solve recomputes everything that depends on
 * them -- the per-column associated-Legendre tables at -mu0 and the beam attenuations, which a plan otherwise keeps from run
 * to run while its inputs are unchanged (the reference recomputes them in every call, _solve_for_gen_and_part_sols.py:96-109)
 * -- and starts behind everything queued on the plan's stream, as after rtd_plan_set_columns.  For measurements of the
 * fresh-input rate with inputs resident in HBM (bench.py: `value`; repeated inputs: `value_cached_tables`). */
int rtd_plan_invalidate_tables(rtd_plan* plan);

/* --- solve: _solve_for_gen_and_part_sols + _solve_for_coeffs on the device ---------------- */
/* Asynchronous on the plan's stream. */
int rtd_plan_solve(rtd_plan* plan);

/* --- evaluators: the closures of _assemble_intensity_and_fluxes --------------------------- */
/* tau: [C][ntau] optical depths (0 <= tau <= tau_arr[-1]); phi: [nphi].
 * u      [C][NQuad][ntau][nphi]   (axes mu, tau, phi as the reference's u(tau, phi))
 * u0     [C][NQuad][ntau]
 * fluxes [C][ntau]: flux_up, flux_down diffuse, flux_down direct
 * ulast  [C][NQuad][ntau]: last Fourier mode u^{M-1} (for return_Fourier_error), may be NULL
 * antiderivative bit 0 switches every output to the tau-antiderivative (is_antiderivative_wrt_tau);
 * bit 1 evaluates u without the Nakajima-Tanaka corrections even when rtd_plan_set_nt is active.
 * Any output pointer may be NULL.  Synchronous (returns when the host arrays are filled).
 * Returns RTD_ERR_TAU_RANGE if some tau lies outside its column (the reference raises ValueError). */
int rtd_plan_evaluate(rtd_plan* plan, int32_t ntau, const double* tau, int32_t nphi, const double* phi,
                      int32_t antiderivative, double* u, double* u0, double* flux_up,
                      double* flux_down_diffuse, double* flux_down_direct, double* ulast);

/* Nakajima-Tanaka intensity corrections (TMS + IMS; reference pydisort.py:375-698), applied on the device to the
 * `u` output of rtd_plan_evaluate once set (the reference returns u_corrected in place of u).
 *   weighted_leg_all [C][L][nleg_all]  (2l+1) g_l of the FULL phase function (weighted_Leg_coeffs_all)
 *   f_arr            [C][L]            delta-M truncation fractions
 *   ims_coef         [C][nleg_all]     (2l+1)(2 g~_l - g~_l^2), g~ the tau-omega weighted residual moments (:601-611)
 *   ims_par          [C][2]            scaled_mu0 = mu0/(1 - omega_avg f_avg), amplitude I0/(4pi) (omega_avg f_avg)^2/(1 - omega_avg f_avg)
 * nleg_all <= 0 switches the corrections off. */
int rtd_plan_set_nt(rtd_plan* plan, int32_t nleg_all, const double* weighted_leg_all, const double* f_arr,
                    const double* ims_coef, const double* ims_par);

/* Throughput form: evaluation points are uploaded once, results stay in HBM. */
int rtd_plan_set_eval_points(rtd_plan* plan, int32_t ntau, const double* tau, int32_t nphi, const double* phi);
/* solve + evaluate at the stored points, asynchronous on the plan's stream */
int rtd_plan_run(rtd_plan* plan);
/* copy the results of the last rtd_plan_run to the host (any pointer may be NULL) */
int rtd_plan_fetch(rtd_plan* plan, double* u, double* u0, double* flux_up, double* flux_down_diffuse,
                   double* flux_down_direct);
/* rtd_plan_run + rtd_plan_fetch as one host-to-host pipeline: window w's results travel to the host (device -> pinned
 * staging on a copy stream -> the caller's arrays) while window w+1 is being solved.  Synchronous; any pointer may be NULL. */
int rtd_plan_run_fetch(rtd_plan* plan, double* u, double* u0, double* flux_up, double* flux_down_diffuse,
                       double* flux_down_direct);
/* device pointers of the result buffers of rtd_plan_run (for a device-side collective) */
int rtd_plan_result_dev_ptrs(rtd_plan* plan, void** u_dev, int64_t* u_bytes, void** flux_dev, int64_t* flux_bytes);

/* --- the reference's tensors for one column (layout of the reference) ---------------------- */
/* GC [M][L][Q][Q], K [M][L][Q], B [M][L][Q], G_inv_mu_inv [L][Q], G [M][L][Q][Q]; any may be NULL.
 * These are what _solve_for_coeffs returns (_coeffs.py:390) and what the closures capture. */
int rtd_plan_get_tensors(rtd_plan* plan, int32_t column, double* GC, double* K, double* B,
                         double* G_inv_mu_inv, double* G);

/* --- one-call forms (rtd_solve_batch / rtd_solve_tensors) -------------------------------- */
/* A struct of the prepared arguments of _assemble_intensity_and_fluxes (_assemble.py:8-32) with the column axis; the
 * field meanings are those of rtd_plan_set_quadrature / rtd_plan_set_columns.  NULL optional fields as there. */
typedef struct {
  const double *mu_pos, *weights;                                         /* [N] */
  const double *scaled_omega, *tau, *scaled_tau_with_0, *scale_tau, *wleg; /* per column and layer */
  const double *mu0, *I0, *phi0, *rescale;                                /* [C] */
  const double *b_pos, *b_neg, *s_poly, *bdrf_q, *bdrf_q0;                /* optional */
} rtd_inputs;

/* create plan -> upload -> solve -> evaluate at (tau [C][ntau], phi [nphi]) -> destroy, in one call: what one batched
 * call of the reference's _assemble_intensity_and_fluxes + closures would return.  Output pointers may be NULL. */
int rtd_solve_batch(const rtd_dims* dims, int32_t device, const rtd_inputs* in, int32_t ntau, const double* tau,
                    int32_t nphi, const double* phi, double* u, double* u0, double* flux_up,
                    double* flux_down_diffuse, double* flux_down_direct);

/* the same up to the solve, returning the tensors the reference's closures capture for column `column`
 * (GC, K, B, G_inv_mu_inv, G in the reference layout; _solve_for_coeffs.py:390). */
int rtd_solve_tensors(const rtd_dims* dims, int32_t device, const rtd_inputs* in, int32_t column, double* GC, double* K,
                      double* B, double* G_inv_mu_inv, double* G);

/* --- measurement --------------------------------------------------------------------------- */
/* When enabled, rtd_plan_run/solve bracket each kernel with HIP events on the plan's stream. */
int rtd_plan_enable_timing(rtd_plan* plan, int32_t enable);
/* accumulated milliseconds per kernel slot since the last reset (slot names as returned by the Python wrapper):
 * [0] "tables" Legendre tables; [2] "jacobi" the fused rtd_eigen_kernel; [1] "asm" and [3] "post" are the empty slots of
 * the earlier three-kernel eigen stage; [4] "iface" rtd_iface_kernel -- or, at 64 streams, the tiled fused
 * rtd_bc_tile_kernel; [5] "sweep" rtd_sweep_kernel -- or the fused rtd_bc_mfma_kernel when 16 < NQuad <= 32 (slot 4 is
 * then empty) -- or, at 64 streams, the pivoted kernels' pass over the chains the tiled kernel flagged; [6] "eval"
 * rtd_eval_kernel or, with the fused interface evaluation, rtd_fourier_kernel, plus the NT corrections.  Launches counted in nlaunch[7] (one per window of columns).  Synchronises; with timing
 * enabled every window starts with a stream synchronisation: a measurement mode, not the throughput path. */
int rtd_plan_get_timing(rtd_plan* plan, double ms[7], int64_t nlaunch[7], int32_t reset);
/* number of (column, mode) chains of the last window solved whose speculative (diagonal-pivot) elimination failed in
 * the tiled fused boundary-condition kernel (64 streams) and that the pivoted row-per-lane kernels solved instead
 * (diagnostic; 0 for other stream counts) */
int rtd_plan_pivoted_chains(rtd_plan* plan, int32_t* chains);
/* maximum Jacobi sweeps used by any eigenproblem of the last solve (diagnostic) */
int rtd_plan_max_sweeps(rtd_plan* plan, int32_t* sweeps);

/* --- multi-GPU: RCCL over xGMI, one communicator rank per plan (one process per GPU) -------- */
/* The path shards by atmospheric column with no exchange during the solve (the reference's
 * independent loops are _solve_for_gen_and_part_sols.py:88-91 and _solve_for_coeffs.py:110-111, the meeting point is
 * the Fourier sum _assemble_intensity_and_fluxes.py:256-260); the one collective stitches the evaluated results of
 * every rank -- u [C][NQuad][ntau][nphi] and the fluxes [3][C][ntau] -- with ncclAllGather.  rank 0 creates the id
 * (ncclGetUniqueId) and hands the 128 bytes to the other ranks by any host channel. */
/* Load RCCL now (call before anything else in the process loads another RCCL/HIP runtime, e.g. PyTorch). */
int rtd_comm_preload(void);
int rtd_comm_unique_id(char id[128]);
int rtd_comm_init(rtd_plan* plan, const char id[128], int32_t rank, int32_t nranks);
/* RCCL's own statement about the plan's communicator: ncclCommCount, ncclCommUserRank, ncclCommCuDevice (any pointer may be
 * null).  RTD_ERR_STATE when they disagree with the arguments of rtd_comm_init. */
int rtd_comm_size(rtd_plan* plan, int32_t* nranks, int32_t* rank, int32_t* device);
/* Which transport carries the collectives, as text: "rccl", or "stub:ipc" / "stub:shm" when RTD_RCCL_STUB named the tests'
 * stand-in (plan may be null: "stub" without the communicator's mode).  Never a rate through the stub. */
int rtd_comm_transport(rtd_plan* plan, char* buf, int32_t nbuf);
int rtd_comm_allgather_fluxes(rtd_plan* plan);               /* asynchronous on the plan's stream */
int rtd_comm_fetch_gathered(rtd_plan* plan, double* out);    /* host [nranks][3][C][ntau] */
/* u AND fluxes of the last rtd_plan_run: the rank's results are snapshot into its own slot of the gathered arrays (device
 * copy on the plan's stream), then two IN-PLACE ncclAllGather run from there on the plan's communication stream, ordered
 * after the copy by an event: the next rtd_plan_run overlaps them completely (nothing of it waits for the collective; the
 * next gather's snapshot does, a whole step later).  rtd_plan_synchronize waits for both streams. */
int rtd_comm_allgather_results(rtd_plan* plan);
/* The same results gathered on ONE rank only ("if only rank 0 needs results"): the other ranks
 * ncclSend their u and fluxes, `root` ncclRecv's them into the layout of rtd_comm_allgather_results (one group call on
 * the communication stream, overlapped with the next run like the all-gather).  Every rank of the communicator calls
 * it; rtd_comm_fetch_gathered_results is then valid on the root. */
int rtd_comm_gather_results(rtd_plan* plan, int32_t root);
/* host copies of the gathered results: u [nranks][C][NQuad][ntau][nphi] (= all nranks * C columns in rank order),
 * fluxes [nranks][3][C][ntau]; either may be NULL */
int rtd_comm_fetch_gathered_results(rtd_plan* plan, double* u, double* fluxes);
/* A slice of the gathered results: the columns [first, first + count) of rank `rank`'s shard -- u [count][NQuad][ntau][nphi],
 * fluxes [3][count][ntau]; either may be NULL.  What a consumer (or a verification: bench.py checks every rank's slot
 * against a local solve of the same columns) reads without copying all nranks * C columns to the host. */
int rtd_comm_fetch_gathered_columns(rtd_plan* plan, int32_t rank, int32_t first, int32_t count, double* u, double* fluxes);
/* Layer shards (the north star's variant "(layer x mode x column) instances sharded, a single
 * all-gather to stitch the boundary-condition system"): the eigen stage is independent per layer
 * (_solve_for_gen_and_part_sols.py:114), the boundary-condition solve couples the layers (_solve_for_coeffs.py:296-323).
 * Rank r decomposes the layers [r * count, (r + 1) * count) of every (column, mode) with rtd_plan_solve_layers; ONE
 * ncclAllGather (rtd_comm_allgather_layers: pack, gather, unpack on the plan's stream) gives every rank the eigen-stage
 * results of all layers -- Y, A, k, E, B per (column, mode, layer) and the thermal-source vectors per (column, layer):
 * 8 Lloc [C M (2 NP^2 + 2 NP + 2 NP) + C (2 NP Ns + NP)] bytes per rank (cfg4, one column, 4 ranks: 737 KB per rank) --
 * and rtd_plan_solve_bc finishes the solve on each rank (or on the one that wants the result).  Worth it only for few
 * columns with very large nlayers x nfourier; plans of one window only. */
int rtd_plan_solve_layers(rtd_plan* plan, int32_t first_layer, int32_t count);
int rtd_comm_allgather_layers(rtd_plan* plan, int32_t count_per_rank);
int rtd_plan_solve_bc(rtd_plan* plan);

/* mode shards: ncclAllReduce(sum) of the u, u0 and flux results of rtd_plan_run over the ranks, in place, asynchronous
 * on the plan's stream (rtd_plan_fetch then returns the complete fields on every rank) */
int rtd_comm_allreduce_results(rtd_plan* plan);
int rtd_comm_destroy(rtd_plan* plan);

enum {
  RTD_OK = 0,
  RTD_ERR_ARG = 1,        /* bad argument / unsupported size */
  RTD_ERR_HIP = 2,        /* HIP runtime failure             */
  RTD_ERR_TAU_RANGE = 3,  /* tau outside [0, tau_arr[-1]]    */
  RTD_ERR_STATE = 4,      /* call order (e.g. evaluate before solve) */
  RTD_ERR_NUMERIC = 5     /* numerical failure on the device: non-positive Cholesky pivot (phase function not positive
                             definite after delta-M scaling), Jacobi iteration not converged, singular boundary-condition
                             system or 1/mu0 on an eigenvalue (non-finite result).  The reference raises LinAlgError from
                             np.linalg.solve / returns NaN in these cases (_solve_for_gen_and_part_sols.py:186, :226-231) */
};

/* --- environment read by the library (the complete list; tests/test_host_logic.py greps the sources against it) ---------
 * None of these changes a result beyond rounding: they select between implementations that the test suite holds to the same
 * parity (tests/test_gpu_parity.py runs the suite under each), size buffers, or print diagnostics.  Timing experiments whose
 * results are NOT valid (e.g. the aliased hand-off reads of DESIGN.md section 7a) exist only behind compile-time -D flags.
 *   RTD_WORK_BYTES        bytes of solve intermediates per plan (default 24 GiB): sizes the automatic column window
 *   RTD_POOL_BYTES        bytes of device memory (per device) of destroyed plans kept for the next plan in blocks above 64 MB
 *                         (default 0: large blocks go straight back to the runtime); the same as rtd_pool_set_limit
 *   RTD_NO_PIPELINE       windows one after the other on one stream (no second hand-off slot, no eigen stream)
 *   RTD_BC_FORCE_PIVOT    fused boundary-condition kernels: =1 every elimination is redone by the column-pivoted LDS path (the
 *                         path of a failed speculation); =2 every chain of the 32-stream kernel takes the register-resident
 *                         column-pivoted elimination throughout (the path of near-conservative mode-0 chains)
 *   RTD_BC_FORCE_HANDOVER tiled (64-stream) kernel: every third chain goes to the pivoted row-per-lane kernels
 *   RTD_BC_TILED          32 streams through the tiled kernel's T = 1 instance instead of rtd_bc_mfma_kernel
 *   RTD_BC_TILE_V1        64 streams through rtd_bc_tile_kernel<2> (one wavefront per SIMD, rounds 2-3) instead of the lean
 *                         two-wavefronts-per-SIMD kernel of rtd_bc_tile2.hip (round 4)
 *   RTD_BC_WIDE_V1        66 ... 128 streams through the row-per-lane kernels (one wavefront per chain, rounds 1-3) instead of the
 *                         four-wavefronts-per-chain kernels of rtd_bc_wide.hip (round 4)
 *   RTD_EIG_MFMA          eigen stage with its assembly GEMMs on the matrix cores (measured slower; a tested variant)
 *   RTD_EIG_SMALL_V1      2 ... 8 streams: eigen stage through the four-lanes-per-problem instance of the general eigen kernel
 *                         instead of the one-lane-per-problem kernel of rtd_eig_small.hip
 *   RTD_SMALL_SPLIT       2 ... 16 streams through the separate interface / sweep / evaluation kernels instead of the fused
 *                         rtd_bc_small_kernel
 *   RTD_RCCL_STUB         TESTS ONLY: path of a stand-in for the RCCL entry points (tests/stub/librccl_stub.so) for rank processes
 *                         that share one GPU, where RCCL itself refuses a second rank; rtd_comm_transport() then says "stub"
 *   RTD_DEBUG             diagnostics on stderr
 */
#ifdef __cplusplus
}
#endif
#endif /* RTD_H */
