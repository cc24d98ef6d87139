/*
 * mfgp.h -- C-ABI of libmfgp_hip.so: the MI355X (gfx950) exact-GP engine that sits where the
 * reference delegates to GPy.
 *
 * The reference has no FFI of its own: its "engine boundary" is the slice of the GPy object API
 * that src/abstractMFGP.py and src/MFDataFusion.py touch.  Every entry point below names the
 * reference call site (file:line under /root/reference) whose work it replaces.  The Python shim
 * (multifidelity_datafusion_gps_amd/engine.py) binds these with ctypes.CDLL; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int32 status: 0 = OK; >0 = LAPACK-style info (1-based index of the first
 *     non-positive pivot met by the Cholesky); <0 = error (text via mfgp_last_error): -1 argument / state, -2 HIP,
 *     -4 RCCL, MFGP_ERR_OOM (-6) a batch slab the device cannot hold (the handle stays usable: see mfgp_eval_batch).
 *   - all matrices are row-major (C order) fp64, caller-owned HOST buffers unless a name says "dev";
 *     the library copies what it needs and owns every device allocation behind the opaque handle.
 *   - hyper-parameters cross the boundary in natural units (variance, lengthscale, noise variance);
 *     the positivity transforms stay in Python (reference: paramz Logexp, [GPy-recall]).
 *   - a handle is bound to one device (and its own HIP streams) and is not thread-safe; distinct handles may
 *     be driven from distinct threads / processes (one process per GPU is the multi-GPU model).
 */
#ifndef MFGP_H
#define MFGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfgp_handle mfgp_handle;

#define MFGP_ERR_OOM (-6)

/* kernel-part types: GPy.kern.RBF / Matern32 / Matern52 (src/abstractMFGP.py:60, :62 kern_class1..3) */
enum { MFGP_KERN_RBF = 0, MFGP_KERN_MATERN32 = 1, MFGP_KERN_MATERN52 = 2 };
/* OR-ed into mfgp_kern_part.type: one lengthscale PER active column (GPy's ARD=True; the "ARD weights" the reference's model
 * docstrings speak of, src/models/NARGP.py:13, src/models/GPDF.py:12 -- the reference itself never passes the flag):
 * k = variance * shape( sqrt( sum_d (x_d - x'_d)^2 / lengthscale_d^2 ) ) */
#define MFGP_KERN_ARD 0x100

/*
 * One stationary factor k_f(x,x') = variance_f * shape_f(|x[c0:c1]-x'[c0:c1]| / lengthscale_f).
 * The covariance is   K = sum over distinct `term` ids of ( product of the factors with that id ),
 * which covers GPy.kern.RBF(D) (one factor, one term; src/abstractMFGP.py:59-60) and the NARGP
 * composite kern1*kern2 + kern3 with active_dims column slices (src/abstractMFGP.py:73-80).
 * Parameter vector layout used by every call, factor after factor: variance_f, then the factor's lengthscale(s) -- ONE for an
 * isotropic factor (so theta[2*f] = variance_f, theta[2*f+1] = lengthscale_f when no factor is ARD), col_end - col_begin of
 * them, in column order, for an MFGP_KERN_ARD factor.  P = the total (mfgp_num_params), at most MFGP_MAX_THETA.  Gradients use
 * the same layout followed by the noise-variance entry: P + 1 doubles.
 */
typedef struct mfgp_kern_part {
    int32_t type;      /* MFGP_KERN_* */
    int32_t col_begin; /* active_dims = [col_begin, col_end) */
    int32_t col_end;
    int32_t term;      /* factors sharing a term id are multiplied; terms are summed; ids ascending */
} mfgp_kern_part;

#define MFGP_MAX_PARTS 6
#define MFGP_MAX_THETA 40

/* P for a kernel description (no handle needed); < 0 for an invalid description */
int32_t mfgp_num_params(const mfgp_kern_part* parts, int32_t n_parts);

/* hipEvent stage timers of the most recent mfgp_eval / mfgp_predict (milliseconds) and the
 * algorithmic work of the two single-launch kernels the bench reports rooflines for. */
typedef struct mfgp_timings {
    double kbuild_ms;    /* K(X,X)+noise lower-triangle build: ONE launch of mfgp_kbuild_tri_f64      */
    double cholinv_ms;   /* blocked Cholesky + triangular inverse (leaf kernels + MFMA tile GEMMs)    */
    double solve_ms;     /* z = L^-1 y, alpha = L^-T z, log-det, quadratic form                        */
    double kinv_ms;      /* K^-1 = L^-T L^-1 lower triangle: ONE launch of the MFMA tile-GEMM kernel   */
    double grad_ms;      /* fused dNLML/dtheta reduction over the lower triangle                       */
    double predict_panel_ms; /* K(X*,X) panel + mean GEMV                                              */
    double predict_var_ms;   /* V = K(X*,X) L^-T (MFMA) + row sum of squares                           */
    double total_ms;     /* first event -> last event of the call                                       */
    double kbuild_bytes; /* algorithmic bytes of the K-build launch (4*Np*(Np+64) B)       */
    double kinv_flops;   /* algorithmic flops of the K^-1 launch (Np^3/3)                               */
    double cholinv_flops;/* 2*Np^3/3                                                                    */
    int64_t n_launches;  /* kernel launches issued by the call                                          */
    int64_t timed;       /* which of the millisecond fields were measured by the call: 0 = none (no event was
                          * recorded: below Np = 4096 a handle records none unless MFGP_TIMING / MFGP_STAGE_TIMING
                          * ask for them -- the fields are then 0, not a measurement), 1 = total_ms and the two
                          * predict fields, 3 = the per-stage fields as well.  Bytes / flops are always filled.   */
} mfgp_timings;

/* running sums over every mfgp_eval / mfgp_predict since the last reset: lets a benchmark attribute
 * the timed region to kernels with HIP events recorded on the engine's own stream. */
typedef struct mfgp_counters {
    double evals, grad_evals, predicts, predict_rows;
    double kbuild_ms, cholinv_ms, solve_ms, kinv_ms, grad_ms, total_ms, predict_ms;
    double kbuild_bytes, kinv_flops, cholinv_flops;
    double predict_panel_ms;   /* K(X*,X) panel + mean                                                          */
    double predict_var_ms;     /* V = K(X*,X) L^-T + row sums of squares                                         */
    double predict_var_flops;  /* algorithmic flops of the variance products: Np^2 * rows per predict,
                                * counted whether or not the call was timed                                         */
    double timed_evals;        /* evaluations whose total_ms entered the sums (see mfgp_timings.timed)              */
    double timed_predict_var_flops; /* variance-product flops of the predicts whose predict_var_ms entered the sums  */
} mfgp_counters;

/* ---- lifecycle --------------------------------------------------------------------------------- */

/* create an engine on HIP device `device_id`. Fails loudly (<0) when no HIP device. */
int32_t mfgp_create(int32_t device_id, mfgp_handle** out);
int32_t mfgp_destroy(mfgp_handle* h);
/* text of the last error on this handle (or a global one when h == NULL); owned by the library */
const char* mfgp_last_error(mfgp_handle* h);
/* library / device identification string ("mfgp_hip gfx950 <device name> CUs=..") */
const char* mfgp_device_info(mfgp_handle* h);
/* hash of the sources this library was built from (build.py: sha256 over csrc/ + this header, 16 hex digits; "unknown" for a
 * build that did not pass it): measurement files under profiles/ carry the id of the library they were taken with */
const char* mfgp_build_id(void);

/* replaces GPy.models.GPRegression(X=, Y=, kernel=) data capture
 * (src/MFDataFusion.py:93-98, src/abstractMFGP.py:100-102): one upload per fit.
 * X is (N, D) row-major, Y is (N,) [the reference's (N,1) column]. */
int32_t mfgp_set_data(mfgp_handle* h, const double* X, int64_t N, int32_t D, const double* Y);

/* replaces the kernel objects built at src/abstractMFGP.py:51-80 (RBF / Prod / Add with active_dims) */
int32_t mfgp_set_kernel(mfgp_handle* h, const mfgp_kern_part* parts, int32_t n_parts);

/* ---- hot calls --------------------------------------------------------------------------------- */

/* One objective(+gradient) evaluation = what every paramz/GPy parameter change triggers
 * (ExactGaussianInference.inference behind src/abstractMFGP.py:103,132-137; src/MFDataFusion.py:93-100):
 *   Ky = K(theta) + (noise + jitter) I ; L = chol(Ky) ; alpha = Ky^-1 y ; logdet ;
 *   nlml = 0.5*(N log 2pi + logdet + y^T alpha) ;
 *   want_grad: grad[0 .. P) = dNLML/d theta in theta's own layout (mfgp_kern_part above: per factor its variance, then its
 *   lengthscale(s)); grad[P] = dNLML/d noise -- P + 1 doubles, P = mfgp_num_params().
 * theta has P entries.  GPy adds 1e-8 to the diagonal itself; pass it (plus any jitchol
 * retry jitter) as `jitter`.  Returns >0 (pivot index) when Ky is not positive definite. */
int32_t mfgp_eval(mfgp_handle* h, const double* theta, double noise, double jitter,
                  int32_t want_grad, double* nlml, double* grad);

/* B evaluations of mfgp_eval at once: the same data and kernel structure at B hyper-parameter points -- what the independent
 * restarts of optimize_restarts(6, ...) (src/abstractMFGP.py:137) ask for between them, one evaluation per run and round.
 * thetas is (B, P) row-major, noises / jitters are (B); nlml (B), grads (B, P + 1) row-major (may be NULL when want_grad == 0),
 * status (B): 0, or the pivot index of an evaluation whose Ky is not positive definite (its nlml / gradient are then
 * undefined; the others are unaffected).  The return value covers the call as a whole (0, or < 0 for an argument / HIP error).
 * One pass of the factorisation plan carries the B matrix sets side by side (1 <= B <= 16; 4 Np^2 doubles of device memory
 * per set, kept by the handle), each evaluation's arithmetic is mfgp_eval's tile for tile: results are bitwise those of B
 * mfgp_eval calls.  The handle's own factorisation (what mfgp_predict / mfgp_nlml read) is left untouched.
 * Memory: a request whose sets do not fit -- hipMalloc says so, or they exceed the environment's MFGP_BATCH_MEM_CAP (bytes per
 * handle) -- returns MFGP_ERR_OOM and changes nothing: the sets the handle already held stay, every other call works; the caller
 * retries with fewer sets, or with mfgp_eval, which needs none (the host layer does: same results bit for bit, only slower).
 * mfgp_mem_info: free / total bytes of the handle's device (hipMemGetInfo).  mfgp_batch_mem: the bytes `sets` sets would take on
 * this handle at its current capacity, MFGP_BATCH_MEM_CAP (0: none) and the number of sets it holds -- what a caller sizes its
 * batches from BEFORE asking. */
int32_t mfgp_eval_batch(mfgp_handle* h, int32_t B, const double* thetas, const double* noises, const double* jitters,
                        int32_t want_grad, double* nlml, double* grads, int32_t* status);
int32_t mfgp_mem_info(mfgp_handle* h, int64_t* free_bytes, int64_t* total_bytes);
int32_t mfgp_batch_mem(mfgp_handle* h, int32_t sets, int64_t* bytes, int64_t* cap_bytes, int32_t* sets_held);

/* Row-block form of the K build for the multi-GPU layout of SURVEY 8(e3) / north_star: each rank builds its row blocks of
 * Ky = K + (noise+jitter) I -- all columns -- in place in the device matrix, the ranks all-gather their blocks
 * (mfgp_allgather_rows: RCCL inside the library; or any other transport through mfgp_rows_download / mfgp_rows_upload or the
 * pointer mfgp_dev_matrix returns), then mfgp_eval_prebuilt factorises what is there instead of building K itself.  Same
 * arithmetic as mfgp_eval.  mfgp_kbuild_rows: rows [row_begin, row_end) (multiples of 64, within the padded size);
 * mfgp_kbuild_owned_rows: the rows of every 128-row block b with mfgp_row_block_owner(b, size) == rank -- the serpentine
 * block-cyclic deal 0 1 .. G-1 G-1 .. 1 0 .. under which every rank's blocks hold the same share of the lower triangle. */
int32_t mfgp_kbuild_rows(mfgp_handle* h, const double* theta, double noise, double jitter, int64_t row_begin,
                         int64_t row_end);
int32_t mfgp_kbuild_owned_rows(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t rank, int32_t size);
int32_t mfgp_row_block_owner(int32_t block, int32_t size);
int32_t mfgp_dev_matrix(mfgp_handle* h, void** dev_ptr, int64_t* padded_n); /* Np x Np fp64, row-major, ld = Np */
int32_t mfgp_eval_prebuilt(mfgp_handle* h, int32_t want_grad, double* nlml, double* grad);

/* ---- multi-GPU exchange steps (one process per GPU, RCCL over xGMI) -----------------------------
 * The reference has no multi-device path; these are the collectives of the decompositions SURVEY 8(e) lists.
 * mfgp_comm_unique_id: 128 opaque bytes (ncclUniqueId), produced on rank 0 and carried to the other ranks by the
 *   host side's own rendezvous; mfgp_comm_init: collective over all ranks, binds a communicator to the handle
 *   (its device, its stream); librccl is loaded lazily by these two calls only.
 * mfgp_allgather_rows: after every rank has run mfgp_kbuild_owned_rows(.., rank, size) with the communicator's rank and size,
 *   the LOWER part of every block (128 x 128 (b + 1) doubles: all the factorisation reads) is packed by owner and ONE in-place
 *   ncclAllGather of equal chunks completes the lower triangle of Ky on every rank -- 4 Np (Np + 128) / size bytes per rank,
 *   half of what full rows would move; the part of a foreign block right of its diagonal block is left as it was.
 * mfgp_allgather_host: recv[rank*count .. ) = send of that rank, for the small results that shard by rows
 *   (predictive mean / variance of hf_model.predict(X*), src/MFDataFusion.py:156: 16 B per test row).
 * mfgp_rows_download / mfgp_rows_upload: the same row blocks of the device matrix through host memory, for
 *   transports other than RCCL. */
int32_t mfgp_comm_unique_id(uint8_t* out128);
int32_t mfgp_comm_init(mfgp_handle* h, const uint8_t* id128, int32_t rank, int32_t size);
int32_t mfgp_comm_destroy(mfgp_handle* h);
/* mfgp_comm_state: 0 = no communicator, n > 0 = a live one of n ranks, -1 = ABORTED: a rank whose sharded pass failed after the group
 *   had been told to start it (or whose peers went silent for MFGP_SHARD_TIMEOUT_S, default 600 s) tears its communicator down
 *   without them (ncclCommAbort) instead of issuing collectives nobody will match; every further collective call on the handle
 *   returns -4 and the process is expected to END with an error, so that its launcher stops the peers.  The same deadline guards
 *   mfgp_allgather_rows and mfgp_allgather_host: a gather the peers never join returns -4 after it instead of blocking for ever. */
int32_t mfgp_comm_state(mfgp_handle* h);
/* Collective (every rank of the handle's communicator, same arguments): what one small collective of this group costs on the
 * handle's stream, MEASURED -- the median of `reps` ncclBroadcast of a diagonal message (2 x 128^2 + 2 doubles) and of `reps`
 * in-place ncclAllGather of `panel_blocks` 128 x 128 blocks per rank (0: what a middle block column of the current matrix carries),
 * host clock from the enqueue to the stream running dry, two untimed repetitions first; rank 0's medians are broadcast so that
 * every rank holds the same figures.  out6 = {broadcast us, all-gather us, all-gather bytes per rank, all-gather GB/s received,
 * reps, this rank's own worst median us}.  The figure stays on the handle and decides, when a shared evaluation is planned,
 * whether its Cholesky is distributed over the group as well (2 nblk - 1 such collectives on the serial chain) or replicated
 * (mfgp_shard_decision); without a calibration it is replicated.  There is no counterpart in the reference (one process, no GPU). */
int32_t mfgp_comm_calibrate(mfgp_handle* h, int32_t reps, int32_t panel_blocks, double* out6);
/* the choice a shared evaluation of the handle's current matrix will be planned under: out6 = {1 distributed / 0 replicated
 * Cholesky, projected saving ms, cost ms = collectives x measured us, collectives on the chain, measured us per collective
 * (0: never calibrated), 1 if MFGP_DIST_CHOL forced the choice} */
int32_t mfgp_shard_decision(mfgp_handle* h, double* out6);
/* the rule itself (pure host arithmetic, no handle, no GPU): 1 if a group of `size` ranks should distribute the Cholesky of an
 * nblk x 128 matrix when one collective costs coll_us microseconds (saving: (1 - 1/size) N^3/3 flops at 60 TFLOP/s; cost:
 * (2 nblk - 1) collectives; taken at saving > 1.25 x cost, never at coll_us <= 0) */
int32_t mfgp_dist_cholesky_pays(int32_t nblk, int32_t size, double coll_us, double* saving_ms, double* cost_ms);
int32_t mfgp_allgather_rows(mfgp_handle* h);
int32_t mfgp_allgather_host(mfgp_handle* h, const double* send, int64_t count, double* recv);
/* mfgp_eval_sharded: mfgp_eval as ONE evaluation across the ranks of the handle's communicator (collective: every rank calls it
 *   with the same arguments on the same data; a handle without a communicator is the group of one).  The Cholesky's serial chain
 *   does not shard and runs on every rank; the other two thirds of the evaluation's N^3 flops -- the rows of L^-T and, after one
 *   exchange of those rows (their upper-triangular part packed by owner, ONE in-place ncclAllGather of 4 Np^2 bytes in all), the
 *   rows of K^-1 and the gradient's tile
 *   sums (one ncclAllReduce of P + 1 sums per tile) -- are split by 128-row block.  Every rank returns the same nlml / grad,
 *   BITWISE those of mfgp_eval, and is left with the complete factorisation (mfgp_predict works; mfgp_get_Kinv does not: a rank
 *   holds only its own rows).  What a fit's SEQUENTIAL evaluations -- the low-fidelity run, first run -> restart 0 of
 *   src/abstractMFGP.py:131-137 -- gain from more GPUs.
 *   From 128 block columns (N >= 16384; MFGP_DIST_CHOL=1 / 0 forces it on / off when the handle plans) the Cholesky itself is
 *   distributed over the group too: 1-D block-cyclic rows with the same ownership, a rank factorises the diagonal blocks it owns
 *   and runs its rows of every panel and trailing update; per block column one ncclBroadcast (the diagonal blocks) and one
 *   ncclAllGather (the panel column) sit on the chain.  The same tasks, so the same bits.
 *   Failure: a rank whose pass fails after the group has started it, or whose peers make no progress for MFGP_SHARD_TIMEOUT_S
 *   (600), aborts its communicator (mfgp_comm_state = -1; further collective calls return -4) and must end -- nothing hangs. */
int32_t mfgp_eval_sharded(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, double* nlml,
                          double* grad);
/* The same with ONE optimiser: rank 0 of the communicator LEADS (mfgp_sharded_lead = mfgp_eval_sharded whose arguments reach the
 *   other ranks in a 64-double control block, one ncclBroadcast per evaluation), the others SERVE (mfgp_sharded_serve blocks,
 *   runs its share of every evaluation the leader asks for, returns -- with their number -- when the leader calls
 *   mfgp_sharded_release).  This is how the host layer runs an L-BFGS-B run's evaluations on a group: the optimiser exists once. */
int32_t mfgp_sharded_lead(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, double* nlml,
                          double* grad);
int32_t mfgp_sharded_serve(mfgp_handle* h, int64_t* served);
int32_t mfgp_sharded_release(mfgp_handle* h);
int32_t mfgp_rows_download(mfgp_handle* h, int64_t row_begin, int64_t row_end, double* out);
int32_t mfgp_rows_upload(mfgp_handle* h, int64_t row_begin, int64_t row_end, const double* in);

/* the pieces of mfgp_eval, for callers that want them separately (same state machine) */
int32_t mfgp_factorize(mfgp_handle* h, const double* theta, double noise, double jitter);
int32_t mfgp_nlml(mfgp_handle* h, double* value);
int32_t mfgp_nlml_grad(mfgp_handle* h, double* grad);

/* Rank-1 append at the CURRENT hyper-parameters: extends L, L^-1, alpha, log-det and the quadratic form by one
 * training row in O(N^2) instead of refactorising (the adaptation loop adds one row per step:
 * src/abstractMFGP.py:320,354 -> src/MFDataFusion.py:93-98).  0 = appended; 1 = no padding slot left (N is a
 * multiple of 128): call mfgp_set_data + mfgp_factorize; >1 = not positive definite with the new row. */
int32_t mfgp_append_row(mfgp_handle* h, const double* x_new, double y_new);

/* replaces hf_model.predict(X*) / lf_model.predict(t) (src/MFDataFusion.py:156, src/abstractMFGP.py:104,114):
 * mean = K(X*,X) alpha ; var = kdiag(X*) - rowsum((K(X*,X) L^-T)^2), floored at 1e-15,
 * + noise when include_noise (GPy's predict() default).  Needs a successful mfgp_factorize/mfgp_eval.
 * var may be NULL when want_var == 0. */
int32_t mfgp_predict(mfgp_handle* h, const double* Xstar, int64_t Nstar, double* mean, double* var,
                     int32_t want_var, int32_t include_noise);

/* ---- device-resident level chaining ------------------------------------------------------
 * The reference builds the high-fidelity level's inputs on the host: for every row x it calls f_low at the
 * stencil points x + i*tau and concatenates (src/MFDataFusion.py:177-208); with a data-driven low-fidelity level
 * each f_low call is lf_model.predict(t)[0] (src/abstractMFGP.py:104).  These two calls keep that hand-over on
 * the device.  `lf` is the factorised low-fidelity level (d = its column count); `offsets` is the (c, d)
 * row-major array of stencil offsets ALREADY multiplied by tau, in iteration order.
 *
 * mfgp_augment: out (N, d + c) = [ X | mean_lf(X + offsets[0]) ... mean_lf(X + offsets[c-1]) ]  -- the matrix
 *   __augment_Data returns, in one call (used for the training inputs of the next level).
 * mfgp_predict_chained: mfgp_predict of level `h` (D = d + c columns) at the augmented rows of Xstar (Nstar, d),
 *   the low-fidelity means never leaving the device; aug_out (Nstar, d + c) optionally receives the augmented
 *   rows.  Same numbers as mfgp_predict(lf) + host concatenation + mfgp_predict(h).
 * Both handles must live on the same device and must not be used by another thread during the call. */
int32_t mfgp_augment(mfgp_handle* lf, const double* X, int64_t N, const double* offsets, int32_t c, double* out);
int32_t mfgp_predict_chained(mfgp_handle* h, mfgp_handle* lf, const double* Xstar, int64_t Nstar,
                             const double* offsets, int32_t c, double* mean, double* var, int32_t want_var,
                             int32_t include_noise, double* aug_out);

/* ---- parity / debug read-back (host buffers sized N*N, N) ---------------------------------------- */
int32_t mfgp_get_K(mfgp_handle* h, double* out);      /* K(X,X) WITHOUT noise, full symmetric (rebuilt)  */
int32_t mfgp_get_L(mfgp_handle* h, double* out);      /* lower Cholesky factor of Ky, zeros above diag   */
int32_t mfgp_get_Linv(mfgp_handle* h, double* out);   /* L^-1 lower                                       */
int32_t mfgp_get_Kinv(mfgp_handle* h, double* out);   /* Ky^-1 full symmetric (valid after a gradient)    */
int32_t mfgp_get_alpha(mfgp_handle* h, double* out);  /* alpha (N)                                        */
/* HIP-event times of the last call.  Event records cost stream time, so below N = 4096 none are taken (all times 0) unless
 * the environment asks at mfgp_set_data: MFGP_TIMING=1 (start / end of a call), MFGP_STAGE_TIMING=1 (stages inside an evaluation). */
int32_t mfgp_get_timings(mfgp_handle* h, mfgp_timings* out);
int32_t mfgp_get_counters(mfgp_handle* h, mfgp_counters* out, int32_t reset);
/* hipDeviceSynchronize on the handle's device: what a benchmark brackets its timed region with (every stream of
 * every handle on that device has drained when it returns) */
int32_t mfgp_device_synchronize(mfgp_handle* h);

/* ---- kernel-level test hooks (tests/ only) -------------------------------------------------------- */
/* C = alpha * A B^T + beta * C on Mp x Np x Kp host matrices (multiples of 128) through the MFMA
 * tile-GEMM kernel; tile = 128 or 64, -64 for the serial-chain variant of the 64-tile kernel, 32 for the 32x32 chain kernel.
 * Any other tile is reported as an error (status < 0): the library never terminates the host process. */
int32_t mfgp_dbg_gemm_nt(mfgp_handle* h, const double* A, const double* B, double* C, int32_t M,
                         int32_t N, int32_t K, double alpha, double beta, int32_t tile);
/* the device work of rank `rank` of `size` of one mfgp_eval_sharded WITHOUT its exchange steps; *ms = its duration (what one GPU
 * of a `size`-GPU group would spend; the results are not an evaluation's and the handle is left without a factorisation) */
int32_t mfgp_dbg_eval_as_rank(mfgp_handle* h, const double* theta, double noise, double jitter, int32_t want_grad, int32_t rank,
                              int32_t size, double* ms);
/* Cholesky + inverse of one SPD 128x128 block through the leaf kernel: Lout, Xout are 128x128. */
/* test hook: the n-th sharded pass from now (leader or follower form) fails on THIS rank after the control exchange (0: off) */
int32_t mfgp_dbg_fail_sharded_after(mfgp_handle* h, int32_t n);
/* test hook: the n-th ncclAllGather this handle issues from now (mfgp_allgather_rows, mfgp_allgather_host, the exchange of a shared
 * pass) comes back with an RCCL error WITHOUT having been issued (0: off) -- what the failure path of a gather does is then
 * observable: the communicator is aborted, the handle's collective calls are poisoned (mfgp_comm_state = -1), the peers give up at
 * their deadline */
int32_t mfgp_dbg_fail_collective_after(mfgp_handle* h, int32_t n);
int32_t mfgp_dbg_leaf(mfgp_handle* h, const double* A, double* Lout, double* Xout, double* logdet_half);
#ifdef __cplusplus
}
#endif
#endif /* MFGP_H */
