/*
 * psoap_gp.h -- C ABI of the MI355X (gfx950) Gaussian-process likelihood library.
 *
 * This is the drop-in boundary for PSOAP's dense GP hot path.  The reference has
 * no FFI: the path sits behind Python functions taking NumPy arrays
 * (psoap/covariance.py, psoap/matrix_functions.pyx).  Each entry point below
 * names the reference interface it replaces (paths relative to the reference
 * tree).  Host code (the psoap_amd Python package, or a maintainer's ctypes stub -- see
 * INTEGRATION.md) binds these symbols; there are no torch types, only plain
 * pointers and sizes.  All arrays are C-contiguous fp64 on the HOST unless a
 * parameter is documented as device-resident.
 *
 * Return value: 0 on success, non-zero on a runtime/HIP error (message via
 * psoap_last_error()).  Numerical failure is NOT an error: a non positive-
 * definite matrix or a negative amp/l yields -inf in the output, exactly like
 * psoap/covariance.py:317-318,326-327.
 *
 * Threading: a handle may be used by one host thread at a time.  HIP is
 * initialised lazily by the first call in the calling process (safe after
 * fork(), as psoap/sample_parallel.py:258-278 requires).
 *
 * Several processes on one GPU (that reference's worker-per-chunk model with
 * more chunks than GPUs, psoap/sample_parallel.py:258-278):
 *   - the library serves them in turn: an advisory lock on
 *     <dir>/gpu_<PCI bus id>.lock is held from an upload to the fetch / sync of
 *     the evaluation that reads it, by every other call that touches the device
 *     for its duration, and by a stream while it has tickets outstanding
 *     (released only once its resident launch has left the device when other
 *     processes are there).  <dir> = $PSOAP_LOCK_DIR, else
 *     $XDG_RUNTIME_DIR/psoap, else /tmp/psoap-<uid>: per user, 0700.  Within a
 *     process the lock is counted.  A wait longer than
 *     PSOAP_DEVICE_LOCK_TIMEOUT_S (300) is an error that names the holder.
 *   - the processes on the device are counted: slot files beside the lock, and
 *     where the driver lists them (/sys/class/kfd/kfd/proc/<pid>/queues/<q>/gpuid)
 *     every process with a hardware queue on the device, library or not;
 *   - every persistent launch reports workgroups that the device's scheduler
 *     moved between compute units while a task ran (the one disturbance its
 *     hand-off protocol does not survive): such an evaluation is
 *     run again (PSOAP_SHARE_RETRIES, 3; bit-identical when clean), then by the
 *     staged path (kernel boundaries only).  Measured clean up to 8 processes per device
 *     (0 wrong in 246,000 evaluations, rounds 4-5); the check sees a workgroup that is
 *     somewhere else at the end of a task than at its start, not one that moved and came back;
 *   - with PSOAP_DEVICE_LOCK=0 and several processes on the device, or more
 *     than PSOAP_SHARE_DAG_MAX (8) of them, evaluations take the staged path
 *     from the start and without the lock (kernel boundaries only: immune, and
 *     the device interleaves the processes' kernels); psoap_stream_open is
 *     refused there.  PSOAP_SHARE_POLICY=dag|staged pins the path.
 *   psoap_share_stats reports what all this cost.
 */
#ifndef PSOAP_GP_H
#define PSOAP_GP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psoap_chunk psoap_chunk; /* opaque per-chunk device state */

/* ---- library ---------------------------------------------------------------- */
int psoap_version(void);
const char *psoap_last_error(void);
int psoap_device_count(int *count);

/* What sharing a device with other processes has cost THIS process so far
 * (process-wide counters; out[k], k < min(n, PSOAP_SHARE_N)). */
enum {
    PSOAP_SHARE_PROCS = 0,            /* processes with a slot on the device now (this one included) */
    PSOAP_SHARE_DAG_LAUNCHES = 1,     /* persistent launches issued */
    PSOAP_SHARE_TAINTED = 2,          /* ... that reported a moved workgroup (results withheld) */
    PSOAP_SHARE_RETRIES = 3,          /* evaluations issued again because of that */
    PSOAP_SHARE_STAGED_FALLBACKS = 4, /* evaluations that went to the staged path after the retries */
    PSOAP_SHARE_STAGED_POLICY = 5,    /* evaluations sent down the staged path from the start */
    PSOAP_SHARE_MOVED_TASKS = 6,      /* tasks that ended on another compute unit than they started on */
    PSOAP_SHARE_MOVED_XCD = 7,        /* ... on another XCD */
    PSOAP_SHARE_LOCK_ACQUISITIONS = 8,
    PSOAP_SHARE_LOCK_WAIT_US = 9,
    PSOAP_SHARE_STREAM_RESUBMITS = 10,
    PSOAP_SHARE_LOCK_ENABLED = 11,
    PSOAP_SHARE_N = 12
};
int psoap_share_stats(int device, long long *out, int n);

/* ---- per-chunk handle ---------------------------------------------------------
 * Replaces the per-chunk state of Worker.initialize (psoap/sample_parallel.py:
 * 126-166): fl, sigma and the N x N scratch matrix V11 (:163) live on the device
 * for the life of the handle; max_batch matrices are pre-allocated so that
 * max_batch proposals can be evaluated concurrently. */
int psoap_chunk_create(psoap_chunk **out, int device, int N, const double *fl,
                       const double *sigma, int max_batch);
int psoap_chunk_destroy(psoap_chunk *h);
int psoap_chunk_set_data(psoap_chunk *h, const double *fl, const double *sigma);
/* Observed-frame grid for the device-side Doppler shift
 * (replicate_wls, psoap/data.py:40-63): lwl[N], epoch[N] in [0, n_epochs). */
int psoap_chunk_set_grid(psoap_chunk *h, const double *lwl, const int32_t *epoch,
                         int n_epochs);

/* ---- lnlike -------------------------------------------------------------------
 * lnlike_f / lnlike_f_g / lnlike_f_g_h (psoap/covariance.py:299-376), c = 1,2,3.
 *   lwl : (c, N) rest-frame ln-wavelengths (the *lwls splat of
 *         sample_parallel.py:193); gp : (amp_0, l_0, amp_1, l_1, ...).
 *   out : -0.5 * (r^T K^-1 r + logdet K), r = fl - mu_GP; -inf conventions above. */
int psoap_lnlike(psoap_chunk *h, int c, const double *lwl, const double *gp,
                 double mu_GP, double *out);
/* B proposals at once: lwl (B, c, N), gp (B, 2c), out (B).  B <= max_batch. */
int psoap_lnlike_batch(psoap_chunk *h, int B, int c, const double *lwl,
                       const double *gp, double mu_GP, double *out);

/* Split-phase form of psoap_lnlike_batch: upload (H2D, async, on a copy stream of
 * its own), eval (kernels only, async), fetch (sync + D2H of B doubles).
 * A handle holds TWO proposal batches: an upload always goes to the one that is
 * not being evaluated, and eval consumes the most recent upload (or re-evaluates
 * the current batch when nothing new was uploaded).  So
 *     upload(0); loop { eval(); upload(k+1); fetch() -> results of k }
 * moves the proposals of step k+1 over PCIe while step k is being factored
 * (bench.py's timed step: one upload, one eval, one fetch), and the plain
 * upload / eval / fetch sequence keeps working unchanged. */
int psoap_batch_upload(psoap_chunk *h, int B, int c, const double *lwl,
                       const double *gp, double mu_GP);
/* Same, but the Doppler shift runs on the device: vel (B, c, n_epochs) km/s,
 * lwl_c = lwl - vel[c, epoch]/c_kms (psoap/data.py:37,61).  Needs set_grid. */
int psoap_batch_upload_velocities(psoap_chunk *h, int B, int c, const double *vel,
                                  const double *gp, double mu_GP);
/* Orbit proposals: the batched Kepler solve, the Doppler shift and the
 * |v| >= c_kms -> -inf rule of Worker.lnprob (psoap/sample_parallel.py:183-193) run on the device.
 * model: 0 SB1, 1 SB2, 2 ST1, 3 ST2, 4 ST3; p_orb (B, n_orb) in the order of utils.registered_params
 * up to and including gamma (psoap/utils.py:4-14): n_orb = 6, 7, 11, 12, 13.  Needs set_grid + set_dates. */
int psoap_chunk_set_dates(psoap_chunk *h, const double *dates, int n_epochs);
int psoap_batch_upload_orbits(psoap_chunk *h, int B, int model, const double *p_orb, const double *gp,
                              double mu_GP);
/* orbit.models[model](*p_orb, dates).get_velocities() for B parameter vectors at once
 * (psoap/orbit.py:95-115,148-170,301-320,394-417,463-487): vel_out (B, c, n_dates) km/s. */
int psoap_orbit_velocities(int device, int model, int B, const double *p_orb, int n_dates, const double *dates,
                           double *vel_out);
int psoap_batch_eval(psoap_chunk *h);
int psoap_batch_fetch(psoap_chunk *h, double *out);
int psoap_chunk_sync(psoap_chunk *h);

/* ---- kernel-matrix fills --------------------------------------------------------
 * fill_V11_f / fill_V11_f_g / fill_V11_f_g_h (psoap/matrix_functions.pyx:19-59,
 * 99-146,149-201): full symmetric (N,N) matrix written IN PLACE into the
 * caller's host array; sigma == NULL -> no noise term (the pyx contract);
 * sigma != NULL fuses V11[diag] += sigma**2 (covariance.py:322). */
int psoap_fill_sym(int device, int c, int N, const double *lwl, const double *gp,
                   const double *sigma, double *out);
/* fill_V12_f (psoap/matrix_functions.pyx:61-96): out (M,N),
 * out[i,j] = amp^2 exp(p (lwl_col[j]-lwl_row[i])^2), M = len(lwl_row). */
int psoap_fill_cross(int device, int M, int N, const double *lwl_row,
                     const double *lwl_col, double amp, double l, double *out);

/* ---- predict --------------------------------------------------------------------
 * mode 0: joint conditional of the c components -- predict_f_g
 *         (covariance.py:81-148), predict_f_g_h (:190-251):
 *         mu (c*M), Sigma (c*M, c*M); mean offset fl - 1.0 (:140,:248).
 * mode 1: conditional of the sum -- predict_f_g_sum (:151-187; 1e-8 nugget,
 *         offset fl - 1.0), predict_f_g_h_sum (:253-297; offset fl - mu_c[0],
 *         M == N): mu (M), Sigma (M, M).  mu_c[0] is mu_fg / mu_fgh.
 * mode 2: single component predict_f (:25-54) with the evidently intended
 *         N = len(lwl_predict) (the reference raises NameError at :38):
 *         offset fl - mu_c[0].
 * lwl (c,N), lwl_pred (c,M); Sigma_out may be NULL (get_Sigma=False, :145-148).
 * status_out: 0 ok, 1 data covariance not positive definite (the reference
 * raises LinAlgError there; outputs are then NaN). */
int psoap_predict(int device, int mode, int c, int N, int M, const double *lwl,
                  const double *fl, const double *sigma, const double *lwl_pred,
                  const double *mu_c, const double *gp, double *mu_out,
                  double *Sigma_out, int *status_out);
/* The same on a chunk handle (fl, sigma and N are the handle's): every device
 * buffer lives in a grow-only workspace owned by the handle, so the retrieve loop
 * (scripts/psoap_retrieve_ST3.py:148, one predict per chunk) allocates once.
 * psoap_chunk_predict_release frees that workspace early (destroy frees it too). */
int psoap_chunk_predict(psoap_chunk *h, int mode, int c, int M, const double *lwl,
                        const double *lwl_pred, const double *mu_c, const double *gp,
                        double *mu_out, double *Sigma_out, int *status_out);
/* Mean and diag(Sigma) only (R doubles instead of R^2, no N R^2 product, no 75 MB download):
 * what the retrieve scripts use of Sigma -- sigma_f = sqrt(diag(Sigma))[0:M]
 * (scripts/psoap_retrieve_ST3.py:111-119, psoap_retrieve_SB2.py:107-113). */
int psoap_chunk_predict_var(psoap_chunk *h, int mode, int c, int M, const double *lwl,
                            const double *lwl_pred, const double *mu_c, const double *gp,
                            double *mu_out, double *var_out, int *status_out);
int psoap_chunk_predict_release(psoap_chunk *h);
/* Timings of the handle's last predict call, in ms: device_ms = first upload ->
 * mu and Sigma complete on the device (HIP events), factor_ms = the persistent
 * launch over [B | Cx^T], sigma_ms = mean + prior fill + Sigma = A - W^T W,
 * download_ms = what the Sigma download adds to the call beyond the device work (it
 * travels in groups of tile rows while later groups are computed), total_ms = the whole call
 * (host clock); flops = N^3/3 + N^2 R + N R^2 + 2 N R (SURVEY.md 8(d) F_pred,
 * padded sizes, R = prediction columns). */
typedef struct {
    double device_ms, factor_ms, sigma_ms, download_ms, total_ms, flops;
} psoap_predict_timings;
int psoap_chunk_predict_timings(psoap_chunk *h, psoap_predict_timings *t);
/* A reusable predict workspace that is not tied to a chunk: fl and sigma travel with
 * every call (2 N doubles), all N^2-sized buffers are kept and only grow.  One
 * predictor serves the whole retrieve loop (chunk after chunk); psoap_predict is
 * this with a workspace created and destroyed inside the call. */
typedef struct psoap_predictor psoap_predictor;
int psoap_predictor_create(psoap_predictor **out, int device);
int psoap_predictor_run(psoap_predictor *p, int mode, int c, int N, int M,
                        const double *lwl, const double *fl, const double *sigma,
                        const double *lwl_pred, const double *mu_c, const double *gp,
                        double *mu_out, double *Sigma_out, int *status_out);
int psoap_predictor_run_var(psoap_predictor *p, int mode, int c, int N, int M,
                            const double *lwl, const double *fl, const double *sigma,
                            const double *lwl_pred, const double *mu_c, const double *gp,
                            double *mu_out, double *var_out, int *status_out);
int psoap_predictor_timings(psoap_predictor *p, psoap_predict_timings *t);
int psoap_predictor_destroy(psoap_predictor *p);

/* ---- calibration ---------------------------------------------
 * Chebyshev re-normalisation of one epoch's flux against reference epochs:
 *   fl' = mu + C B^-1 (fl_fixed - mu),  C' = A - C B^-1 C^T,  D = fl_cal * T_k(lwl_cal),
 *   X = (D^T C'^-1 D)^-1 D^T C'^-1 fl',  fl_cor = D X.
 * psoap_calibrate_explicit replaces optimize_calibration (covariance.py:560-624):
 *   caller-filled A (M,M) with sigma_cal^2 on the diagonal, B (N,N) with
 *   sigma_fixed^2, C (M,N); row-major host arrays.
 * psoap_calibrate evaluates the three matrices on the device from c-component
 *   rest-frame grids: optimize_calibration_static (covariance.py:628-707, c = 1,
 *   lwls_cal == lwl_cal) and the per-epoch body of
 *   scripts/psoap_process_calibration_ST3.py:147-183 (c = 3).  lwls_cal (c,M),
 *   lwls_fixed (c,N), gp (2c); lwl_cal (M) is the Chebyshev abscissa, mapped from
 *   [lwl0, lwl1] onto [-1, 1] as numpy's Chebyshev(domain=...) does.
 * Outputs: fl_cor (M), X (order+1).  order <= 15.
 * status_out: 0 ok; 1 B, 2 C', 3 the normal equations not positive definite (the
 * reference raises LinAlgError there; outputs are then unspecified). */
int psoap_calibrate(int device, int c, int M, int N, int order, double lwl0,
                    double lwl1, const double *lwl_cal, const double *lwls_cal,
                    const double *fl_cal, const double *sigma_cal,
                    const double *lwls_fixed, const double *fl_fixed,
                    const double *sigma_fixed, const double *gp, double mu_GP,
                    double *fl_cor, double *X, int *status_out);
int psoap_calibrate_explicit(int device, int M, int N, int order, double lwl0,
                             double lwl1, const double *lwl_cal,
                             const double *fl_cal, const double *fl_fixed,
                             const double *A, const double *B, const double *C,
                             double mu_GP, double *fl_cor, double *X,
                             int *status_out);

/* ---- several chunks, one launch ---------------------------------------------------
 * A group evaluates the uploaded batches of several chunk handles (one device, one
 * component count, sizes may differ) in ONE launch of the persistent kernel over the
 * heterogeneous batch, so that the matrices of all chunks hide each other's dependency
 * chains.  The reference evaluates its chunks in separate worker processes
 * (sample_parallel.py:258-278, :378-387); this is the one-GPU form of that fan-out.
 * Usage: psoap_batch_upload* on every member, psoap_group_eval, psoap_batch_fetch on
 * every member (each handle's stream waits for the group launch). */
typedef struct psoap_group psoap_group;
int psoap_group_create(psoap_group **out, psoap_chunk *const *handles, int n);
int psoap_group_eval(psoap_group *g);
int psoap_group_destroy(psoap_group *g);
/* How often the group rebuilt its task list (batch sizes changed) and refreshed its matrix
 * records (a member moved to its other proposal slot: device-to-device, no host sync). */
int psoap_group_stats(psoap_group *g, long long *plan_builds, long long *record_refreshes);

/* ---- measurement ----------------------------------------------------------------
 * With profiling on, every kernel launch of psoap_batch_eval is bracketed by
 * hipEvents on its own stream (single stream group, so launches serialise);
 * get_timings returns per-kernel-class totals of the LAST eval. */
enum {
    PSOAP_K_FILL = 0,
    PSOAP_K_PANEL_UPDATE = 1, /* MFMA f64 left-looking panel update (dominant) */
    PSOAP_K_POTRF = 2,
    PSOAP_K_TRSM = 3,
    PSOAP_K_MISC = 4,
    PSOAP_K_DAG = 5,          /* the persistent dependency-graph kernel (whole factorisation) */
    PSOAP_K_CLASSES = 6
};
typedef struct {
    double ms[PSOAP_K_CLASSES];      /* summed device time per class */
    int64_t launches[PSOAP_K_CLASSES];
    double flops[PSOAP_K_CLASSES];   /* executed flops per class (MFMA classes) */
    double bytes[PSOAP_K_CLASSES];   /* algorithmic HBM bytes per class (fill) */
    double total_ms;                 /* first launch -> last launch, event time */
} psoap_timings;
int psoap_chunk_set_profiling(psoap_chunk *h, int enabled);
int psoap_chunk_get_timings(psoap_chunk *h, psoap_timings *t);
/* Number of concurrent stream groups a batch is split into (staged mode; default 2). */
int psoap_chunk_set_stream_groups(psoap_chunk *h, int groups);
/* Execution mode of psoap_batch_eval: 1 (default) = one persistent dependency-graph kernel
 * for the whole batched factorisation; 0 = staged, three kernels per 128-row panel.
 * Inside mode 1 the tile updates are cut by one of two schemes chosen from the batch size (latency
 * for small batches, throughput for large ones); the environment variable
 * PSOAP_DAG_SCHEME=0|1, read when a task list is built, pins one for experiments. */
int psoap_chunk_set_mode(psoap_chunk *h, int mode);

/* Debug aid for the persistent kernel: the first call allocates a per-task timestamp log, later
 * calls copy it out (4 x 100 MHz stamps per task, indexed by ticket). */
int psoap_chunk_dag_tasklog(psoap_chunk *h, unsigned long long *out, long long max_tasks);
/* ---- Streamed evaluation (round 4): consecutive ensemble steps through ONE resident launch ----------------
 * Replaces: the back-to-back iterations of the sampler's loop, /root/reference/psoap/sample_parallel.py:434-438 (every
 * iteration calls lnprob -> Worker.lnprob -> covariance.lnlike[...] at :193, and the master gathers at :378-387).
 * A plain psoap_batch_eval is one launch of the persistent kernel per step, which ramps up and drains every time;
 * a stream keeps the launch resident and lets the MATRICES come and go: `lanes` matrix workspaces of the handle
 * (lanes <= max_batch, <= 64) each run the same single-matrix task list, a dispatcher workgroup inside the launch pulls
 * submitted proposals from pinned host memory and opens lanes, and the workgroup that finishes a matrix writes its
 * lnprob straight into pinned host memory.  A proposal's result does not depend on what else is in flight: it is
 * bit-identical for every batch size, submission order and number of GPUs.
 * Typical use -- two half-ensembles in flight (neither half's proposals depend on the other's accept / reject):
 *     open; tA = submit(A0); tB = submit(B0); loop { fetch(tA); tA = submit(A_next); fetch(tB); tB = submit(B_next); }
 * While a stream is open the handle's batch entry points (psoap_batch_*, psoap_lnlike*) are refused -- the matrix
 * workspaces belong to the resident launch -- and other launches on the device wait until it leaves (it does so by
 * itself once nothing has been in flight for PSOAP_STREAM_IDLE_MS, default 20 ms, and comes back on the next submit). */
/* c: number of components of every submission; scheme: -1 automatic (by lanes and N, as for a batch of `lanes` matrices),
 * 0 throughput, 1 latency, 2 following (refused in late round 5 for a rare wrong value; round 6 removed its cause --
 * LABNOTES 14 -- and all three run here). */
int psoap_stream_open(psoap_chunk *h, int c, int lanes, int scheme);
/* n proposals -- lwl (n, c, N), gp (n, 2c) as psoap_batch_upload -- into n free lanes; tickets[n] identify them.
 * Fails when fewer than n lanes are free. */
int psoap_stream_submit(psoap_chunk *h, int n, const double *lwl, const double *gp, double mu_GP, long long *tickets);
/* The same with radial velocities (n, c, n_epochs) -- the resident launch shifts the chunk's grid itself (as
 * psoap_batch_upload_velocities: replicate_wls + lredshift, psoap/data.py:37,61) -- or with orbital parameters
 * (n, n_orb(model)): Kepler solve per epoch, the |v| >= c_kms -> -inf rule (sample_parallel.py:183-187) and the shift
 * inside the launch (as psoap_batch_upload_orbits).  psoap_chunk_set_grid / _set_dates BEFORE psoap_stream_open. */
int psoap_stream_submit_velocities(psoap_chunk *h, int n, const double *vel, const double *gp, double mu_GP, long long *tickets);
int psoap_stream_submit_orbits(psoap_chunk *h, int n, int model, const double *p_orb, const double *gp, double mu_GP,
                               long long *tickets);
/* blocks until the n results are there (any order of tickets); frees their lanes.  -inf for a negative hyper-parameter
 * (covariance.py:317) or a matrix that is not positive definite. */
int psoap_stream_fetch(psoap_chunk *h, int n, const long long *tickets, double *out);
/* blocks until one of the n tickets has its result: *which = its index (then psoap_stream_fetch of that one returns at
 * once) -- independent chains resubmit each walker the moment it completes instead of waiting for a whole group */
int psoap_stream_wait_any(psoap_chunk *h, int n, const long long *tickets, int *which);
/* *ready = 1 when the result of `ticket` has arrived (never blocks) */
int psoap_stream_ready(psoap_chunk *h, long long ticket, int *ready);
/* the resident launch leaves as soon as what is in flight is done (instead of after the idle time-out) and the call
 * returns when it has; the stream stays open and the next submit relaunches.  Call before a device-wide synchronise. */
int psoap_stream_pause(psoap_chunk *h);
/* duration (HIP events around it) of the resident launch that psoap_stream_pause ended last, and the matrices it completed */
int psoap_stream_last_launch(psoap_chunk *h, double *ms, long long *matrices);
/* waits for what is in flight, ends the resident launch, frees the stream */
int psoap_stream_close(psoap_chunk *h);
/* counters: launches of the resident kernel so far (> 1 after an idle time-out), submissions, completed results, the
 * scheme of the lanes' task list and its length; any pointer may be NULL */
int psoap_stream_stats(psoap_chunk *h, long long *launches, long long *submitted, long long *completed, int *scheme,
                       long long *tasks_per_matrix);
/* Pure host function (touches no device): the task list every lane of a stream of `lanes` lanes runs for matrices of
 * P block rows on `workers` workgroups (record format of psoap_chunk_dag_tasks; the field `b` carries the burst marks:
 * 0x8000 = the last ticket of a burst, i.e. of a block row).  scheme -1: automatic; *scheme_out the one taken. */
int psoap_stream_plan(int P, int lanes, int workers, int scheme, void *out, long long max_tasks, long long *n_tasks,
                      long long *n_slots, long long *n_ctrs, int *scheme_out);
/* Debug aids: per-task time stamps of the last `cap` submissions (allocate with out == NULL before the first submit;
 * read with nothing in flight), and the task list every lane runs (record format of psoap_chunk_dag_tasks). */
int psoap_stream_tasklog(psoap_chunk *h, int cap, unsigned long long *out, long long max_words);
int psoap_stream_tasks(psoap_chunk *h, void *out, long long max_tasks, long long *n_tasks);

/* Debug aid: the persistent kernel's task list (16-byte records: type, q, j, S, b(u16), pa, pb,
 * slot(u32), ctr(u32)) in ticket order; *n_tasks receives the list length. */
int psoap_chunk_dag_tasks(psoap_chunk *h, void *out, long long max_tasks, long long *n_tasks);

/* Pure host function (touches no device): the task list of the persistent kernel for B matrices of
 * P block rows on `workers` workgroups, same record format as psoap_chunk_dag_tasks. */
int psoap_dag_plan(int B, int P, int workers, void *out, long long max_tasks, long long *n_tasks,
                   long long *n_slots, long long *n_ctrs, unsigned int *queue_first /* 9 entries or NULL */);
/* The same for a heterogeneous batch (matrices of several chunks in one launch): matrix b has Ps[b]
 * block rows. */
int psoap_dag_plan_multi(int B, const int *Ps, int workers, void *out, long long max_tasks,
                         long long *n_tasks, long long *n_slots, long long *n_ctrs,
                         unsigned int *queue_first);
/* One matrix of P block rows with Mt appended column tiles (the predict launch) and, when
 * Ms > 0, the Ms x Ms upper tiles of their Schur complement (Sigma = A - W^T W) as tasks of
 * the same launch; scheme -1 automatic, 0 throughput, 1 latency. */
int psoap_dag_plan_aug(int P, int Mt, int Ms, int workers, int scheme, void *tasks_out,
                       long long max_tasks, long long *n_tasks, long long *n_slots,
                       long long *n_ctrs, unsigned int *queue_first);

/* Pure host function: a batch's task list with the two hand-out orders of the ready-only scheme (round 5: finals in list
 * order, PART tasks taken only when their panels and predecessor are there) -- order[], dep[], n_main[8]; *has_pool = 0 and
 * no orders for the throughput scheme.  Mt / Ms > 0 (appended column tiles, Schur block): one matrix (predict).
 * tests/test_dag_plan.py plays the hand-out through on the CPU. */
int psoap_dag_plan_pool(int B, const int *Ps, int workers, int Mt, int Ms, int scheme, void *tasks_out,
                        long long max_tasks, long long *n_tasks, unsigned int *order_out, unsigned int *dep_out,
                        unsigned int *n_main_out, unsigned int *queue_first, int *has_pool, long long *n_ctrs);
/* Pure host function: the number of persistent workgroups a batch gets -- all the device admits (two per
 * compute unit), or one per compute unit when the batch is bound by the row-to-row chains of its matrices
 * (algorithmic flops <= 3.3e9 x block rows of the largest matrix).  Ps[b]: block rows of
 * matrix b, Mt: appended column tiles (predict). */
int psoap_dag_pick_workers(int B, const int *Ps, int Mt, int compute_units, int max_workers, int *workers);

#ifdef __cplusplus
}
#endif
#endif /* PSOAP_GP_H */
