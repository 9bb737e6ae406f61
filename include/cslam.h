/*
 * cslam.h -- C ABI of the MI355X-native EKF-SLAM / FastSLAM-2 engine (libcslam_hip.so).
 *
 * This is the drop-in boundary for the hot path of mfkiwl/conan-slam: the entry points below are what
 * a binding behind the reference's `class Slam` virtuals would call.  Each one cites the reference
 * interface it replaces (file:line relative to the reference tree).  Plain C types only: opaque handles,
 * raw pointers, sizes and status codes; no exceptions cross the boundary (the reference swallows its
 * exceptions and continues, e.g. EKF.cpp:125-128 -- a binding maps a non-zero status to that behaviour).
 *
 * Conventions (identical to the reference, which uses Eigen's defaults):
 *   - matrices are COLUMN-MAJOR; element (r,c) of a matrix with leading dimension ld is a[c*ld + r];
 *   - state X = [x, y, phi, lm1.x, lm1.y, ...], n = 3 + 2*N;  P is n x n;
 *   - feature indices `idf` are 1-BASED positions in the state (EKF.cpp:357, PF.cpp:86);
 *   - Z is 2 x m column-major: Z[2*i] = range, Z[2*i+1] = bearing (rad);
 *   - "scalar" pointers (const void*) point at elements of the dtype chosen at create time
 *     (float for CSLAM_F32 -- the reference's own precision -- or double for CSLAM_F64).
 *
 * Ownership: the handle owns the authoritative X and P in HBM (P preallocated for max_landmarks,
 * padded leading dimension).  Host copies are refreshed by get_x / get_state.  Calls on one handle are
 * stream-ordered and asynchronous unless stated otherwise; a handle is NOT thread-safe; distinct
 * handles are independent (own stream) and may be driven from distinct host threads.
 *
 * There is NO CPU fallback: every entry point fails with CSLAM_ERR_NO_DEVICE / CSLAM_ERR_HIP when no
 * gfx950 device or kernel image is available.
 */
#ifndef CSLAM_H
#define CSLAM_H

#ifdef __cplusplus
extern "C" {
#endif

#define CSLAM_VERSION 100

/* ---- status codes ---- */
#define CSLAM_OK 0
#define CSLAM_ERR_BAD_ARG 1    /* null pointer, negative size, idf out of range, wrong dtype ...      */
#define CSLAM_ERR_CAPACITY 2   /* state would exceed max_landmarks / max_features                     */
#define CSLAM_ERR_HIP 3        /* a HIP runtime call failed; see cslam_last_error()                   */
#define CSLAM_ERR_NO_DEVICE 4  /* no usable GPU                                                       */
#define CSLAM_ERR_ALLOC 5      /* host or device allocation failed                                    */
/* sticky per-handle factorisation flags (bit-or), read with cslam_ekf_factor_status() */
#define CSLAM_FACTOR_OK 0
#define CSLAM_FACTOR_FALLBACK 1 /* LLT of S failed; eigen "square root" taken (slam.h:425-429)        */
#define CSLAM_FACTOR_ZEROED 2   /* factor or its inverse non-finite -> update was a no-op (slam.h:252-255, 431-434) */
#define CSLAM_FACTOR_SKIPPED 4  /* async mode: LLT failed and the update was skipped (see set_sync_mode) */
#define CSLAM_FACTOR_INTERNAL 16 /* an internal wait of the look-ahead factor chain timed out (must never happen; results of
                                   that window are undefined) */
#define CSLAM_FACTOR_BAD_IDF 8  /* cslam_ekf_update_device: a device-resident feature index was outside 1..N; the kernels
                                   clamped it (no out-of-bounds access), the update used the clamped index           */

/* ---- precision ---- */
#define CSLAM_F32 0 /* the reference's precision (Eigen::MatrixXf everywhere)                          */
#define CSLAM_F64 1

/* ---- quirk flags: which of the reference's behaviours to reproduce ---- */
#define CSLAM_Q_LOWER_CHOL_GAIN 1 /* slam.h:250-260 with 423: gain built from inv(L), L lower          */
#define CSLAM_Q_PREDICT_NM4 2     /* EKF.cpp:442-443: cross-covariance stripe n-4 wide                 */
#define CSLAM_Q_REF_EXACT 3       /* both: bug-compatible with the reference (default for parity)      */
#define CSLAM_Q_TEXTBOOK 0        /* the algebra the reference meant                                   */

/* ---- profiling stages reported by cslam_ekf_get_stage_times ---- */
#define CSLAM_STAGE_GATHER 0   /* PHT = P*H^T (sparse gather form of slam.h:243)                       */
#define CSLAM_STAGE_FACTOR 1   /* S, symmetrise, chol, inverse (slam.h:244-255)                        */
#define CSLAM_STAGE_GAIN 2     /* W1 = PHT*G, X += W1*(G^T V) (slam.h:257-259)                         */
#define CSLAM_STAGE_DOWNDATE 3 /* P -= W1*W1^T (slam.h:260) -- the P-GEMM                              */
#define CSLAM_N_STAGES 4

typedef struct cslam_ekf* cslam_ekf_t;

/* ---- library-level ---- */
const char* cslam_last_error(void);       /* message of the last failing call on this thread          */
int cslam_version(void);
int cslam_device_count(int* count);

/* ================================ EKF-SLAM ================================================== */

/* Replaces: `new EKF(LM, WP)` + the caller-owned `Eigen::VectorXf X; Eigen::MatrixXf P`
 * (test/main.cpp:89,107-108).  State starts as X = 0_3, P = 0_3x3 as in the reference driver.
 * device < 0 selects the current HIP device. */
int cslam_ekf_create(int max_landmarks, int dtype, int device, int quirks, cslam_ekf_t* out);
int cslam_ekf_destroy(cslam_ekf_t h);

/* sync_mode = 1 (default): update() waits for its own completion and performs the reference's
 * eigen-decomposition fallback (slam.h:425-429) on the host when the LLT of S fails, exactly as the
 * reference would.  sync_mode = 0: fully asynchronous pipeline; a failed LLT makes that update a no-op
 * and raises CSLAM_FACTOR_SKIPPED (check cslam_ekf_factor_status). */
int cslam_ekf_set_sync_mode(cslam_ekf_t h, int sync_mode);

/* Upload / download the state.  X: n scalars; P: n x n, leading dimension ldp (>= n), both host memory.
 * These are what a binding uses where the reference passes X and P by reference into every call. */
int cslam_ekf_set_state(cslam_ekf_t h, const void* X, int n, const void* P, int ldp);
int cslam_ekf_get_state(cslam_ekf_t h, void* X, void* P, int ldp); /* synchronises */
int cslam_ekf_get_x(cslam_ekf_t h, void* X, int capacity);         /* synchronises; writes n scalars */
int cslam_ekf_get_n(cslam_ekf_t h, int* n);
int cslam_ekf_trace(cslam_ekf_t h, double* trace);                 /* synchronises */
int cslam_ekf_synchronize(cslam_ekf_t h);
int cslam_ekf_factor_status(cslam_ekf_t h, int* flags, int clear); /* synchronises */

/* Replaces Slam::predict(X, P, v, swa, Q, wb, dt)  -- slam.h:841-847, EKF.cpp:406-455.
 * Q: 2x2 scalars. */
int cslam_ekf_predict(cslam_ekf_t h, double v, double swa, const void* Q, double wb, double dt);

/* Replaces Slam::update(X, P, Z, R, idf, batch)  -- slam.h:938-943, EKF.cpp:481-496
 * (batchUpdate EKF.cpp:93-129, singleUpdate EKF.cpp:457-479, observeModel EKF.cpp:354-404,
 *  choleskyUpdate slam.h:235-266).  Z: 2 x m scalars, R: 2x2 scalars, idf: m ints (1-based), all host
 * memory, consumed before return.  m = 0 is a no-op, as in the reference. */
int cslam_ekf_update(cslam_ekf_t h, const void* Z, int m, const void* R, const int* idf, int batch);

/* Same, with Z and idf already resident in device memory (HBM); R stays a host pointer (4 scalars).  The host cannot
 * check device-resident indices: every kernel clamps them into 1..N before it forms an address and
 * CSLAM_FACTOR_BAD_IDF is raised (cslam_ekf_factor_status) when one was out of range. */
int cslam_ekf_update_device(cslam_ekf_t h, const void* dZ, int m, const void* R, const int* d_idf, int batch);

/* Replaces Slam::augment(X, P, Z, R)  -- slam.h:190-191, EKF.cpp:9-26 / addOneNewFeature EKF.cpp:28-91.
 * Z: 2 x q scalars (host). Fails with CSLAM_ERR_CAPACITY beyond max_landmarks. */
int cslam_ekf_augment(cslam_ekf_t h, const void* Z, int q, const void* R);

/* Replaces Slam::observeHeading(X, P, phi, useHeading)  -- slam.h:788, EKF.cpp:328-352 with
 * josephUpdate slam.h:700-725.  With H = e_2^T the Joseph form equals P - p p^T / S (p = P[:,2], S = P22 + R) for a
 * symmetric P: the pose stripe is updated at once, the map block receives the rank-1 downdate as one more pending
 * column of the next P-GEMM.  O(n); a predict() issued just before it runs in the same launch. */
int cslam_ekf_observe_heading(cslam_ekf_t h, double phi, int use_heading);

/* Gated nearest-neighbour data association: Slam::dataAssociate (slam.h, implemented at EKF.cpp:235-326 with
 * EKF::computeAssociation EKF.cpp:131-144).  Z: 2 x m observations (host pointer, column-major), R: 2 x 2.
 * For observation i: kind_out[i] = 1 and idf_out[i] = the 1-based feature with the smallest normalised distance
 * among those whose normalised innovation squared is below gate1; otherwise idf_out[i] = 0 and kind_out[i] = 2 when
 * the best NIS of the remaining features exceeds gate2 (far enough to be a new feature) or 0 (ambiguous: dropped).
 * The reference's own return value carries an EMPTY new-feature list (EKF.cpp:307 re-declares ZN, :311 never
 * advances the column): a REF_EXACT caller ignores kind == 2, see conan_slam_amd/ekf.py::data_associate.
 * Synchronous (results are written to host memory). Pending deferred downdates are applied first. */
int cslam_ekf_associate(cslam_ekf_t h, const void* Z, int m, const void* R, double gate1, double gate2, int* idf_out,
                        int* kind_out);

/* Deferred downdates.  slam.h:260 (P = P - W1*W1^T) is linear in the W1 panels, so the engine may keep
 * P = Ps - Wp*Wp^T with up to max_pending_columns columns of not-yet-applied panels and apply them in ONE
 * P-GEMM (k = pending columns): every reader of P adds the rank-k correction for the columns it touches, so
 * results are the reference's up to rounding.  0 (default) applies each update's downdate at once.  The
 * sequential form update(batch = 0) always defers its m rank-2 downdates to one pass at the end of the call
 * (SURVEY.md 8f rank 2).  get_state / trace / associate / flush apply whatever is pending.
 * The shipped engine is single-stream and applies every batch update's P-GEMM at once unless a window is set here. */
int cslam_ekf_set_deferred(cslam_ekf_t h, int max_pending_columns);
int cslam_ekf_flush(cslam_ekf_t h);

/* The two HIP streams (hipStream_t) of a handle: the chain stream carries everything except the covariance downdate,
 * the P-GEMM stream carries P -= W1*W1^T of the previous update (the same stream when the engine is not pipelined).
 * For callers that want to order their own work (event records, input copies) against the engine's. */
int cslam_ekf_get_streams(cslam_ekf_t h, void** chain_stream, void** pgemm_stream);

/* Cap on the workgroups of the persistent covariance-downdate kernel (0 = default: two per compute unit, i.e. the
 * whole chip).  For several filter instances that run side by side on one GPU (Monte-Carlo runs, one stream each): with
 * the default every instance's P-GEMM occupies all compute units for its duration and the other instances' small
 * kernels queue behind it; with e.g. 2 * CUs / instances each instance keeps to its share and their kernels interleave. */
int cslam_ekf_set_pgemm_workgroups(cslam_ekf_t h, int workgroups);

/* Monte-Carlo driver (BASELINE configs[4]; the reference's unit is one filter loop, test/main.cpp:132-200): runs
 * `steps` x { predict(v[t], swa[t], Q, wb, dt); update(Z_t, R, idf_t, batch) } on each of `count` INDEPENDENT filter
 * handles at once, one host thread and one stream pair per handle.  dZ[i] / d_idf[i]: device-resident inputs of
 * instance i, steps x (2*m scalars) and steps x (m ints), step-major.  Returns when every instance has been enqueued
 * (asynchronous mode) or has finished (sync mode); cslam_ekf_synchronize waits for an instance. */
int cslam_ekf_run_many(cslam_ekf_t* handles, int count, int steps, const double* v, const double* swa, const void* Q,
                       double wb, double dt, const void* const* dZ, const int* const* d_idf, int m, const void* R,
                       int batch);

/* Batched Monte-Carlo engine (BASELINE configs[4], test/main.cpp:132-200 x I): `instances` INDEPENDENT f32 filters of the
 * same size (n = 3 + 2 * n_landmarks, fixed) advance in lockstep, every stage of the step ONE launch for all of them
 * (conan_slam_amd/csrc/cslam_ekf_batch.hip).  The arithmetic per instance is cslam_ekf_update's (batch form with the
 * predict held back and applied inside the update, look-ahead windows of two updates): an instance's results are bitwise
 * those of a single handle that runs the same pairs of updates as look-ahead windows.
 *   run: `steps` x { predict(v[t], swa[t], Q, wb, dt); update(Z_t, R, idf_t, batch) } on every instance; the controls are
 *        common to the instances (as in cslam_ekf_run_many), dZ[i] / d_idf[i] are instance i's device-resident inputs,
 *        steps x (2*m floats) and steps x (m ints), step-major; 9 <= m <= 32.  Asynchronous: returns when the work has
 *        been enqueued.  Feature indices are checked on the device (CSLAM_FACTOR_BAD_IDF).  Windows are formed within a
 *        call -- steps (0,1), (2,3), ...; an odd call ends with a window of one update -- so the rounding of a run depends
 *        on how its steps are cut into calls (as a single handle's does on when its updates arrive).
 *   flush applies the pending covariance panels; get_state / trace flush and synchronise; factor_status synchronises and
 *   writes one flag word per instance.  instances * (round_up(n, 128))^2 * 4 must stay below 4 GiB. */
typedef struct cslam_ekf_batch* cslam_ekf_batch_t;
int cslam_ekf_batch_create(int instances, int n_landmarks, int device, int quirks, cslam_ekf_batch_t* out);
int cslam_ekf_batch_destroy(cslam_ekf_batch_t h);
int cslam_ekf_batch_set_state(cslam_ekf_batch_t h, int instance, const float* X, int n, const float* P, int ldp);
int cslam_ekf_batch_get_state(cslam_ekf_batch_t h, int instance, float* X, float* P, int ldp);
int cslam_ekf_batch_run(cslam_ekf_batch_t h, int steps, const double* v, const double* swa, const float* Q, double wb,
                        double dt, const float* const* dZ, const int* const* d_idf, int m, const float* R);
int cslam_ekf_batch_flush(cslam_ekf_batch_t h);
int cslam_ekf_batch_synchronize(cslam_ekf_batch_t h);
int cslam_ekf_batch_trace(cslam_ekf_batch_t h, double* traces /* [instances] */);
int cslam_ekf_batch_factor_status(cslam_ekf_batch_t h, int* flags /* [instances] */);
int cslam_ekf_batch_info(cslam_ekf_batch_t h, int* instances, int* n, long long* windows);
/* HIP events around one covariance-downdate launch in `every` (0 stops; at most 256 launches are kept; synchronises);
 * get: synchronises, sum of milliseconds and number of the launches timed since profiling was switched on. */
int cslam_ekf_batch_set_profiling(cslam_ekf_batch_t h, int every);
int cslam_ekf_batch_get_pgemm_time(cslam_ekf_batch_t h, double* ms_sum, int* launches);

/* Per-stage device times of update() measured with HIP events on the handle's streams.
 * on = 1 starts recording (events around every stage of every update), on = 2 brackets the covariance downdate
 * (P-GEMM) launches only, on = 3 one downdate launch in sixteen, on = 4 one in four (an event pair costs ~11 us of
 * stream time), on = 0 stops.
 * get: synchronises, writes the SUM of milliseconds per stage since profiling was switched on and the
 * number of launches per stage. */
int cslam_ekf_set_profiling(cslam_ekf_t h, int on);
int cslam_ekf_get_stage_times(cslam_ekf_t h, double* ms_sum, int* launches);

/* Introspection for tests: copy the update intermediates of the LAST batch update to the host.
 * PHT and W1 are n x k (leading dimension n on output), S and G are k x k, V and t have k entries.
 * Any pointer may be NULL. Synchronises. */
int cslam_ekf_debug_last_update(cslam_ekf_t h, void* PHT, void* S, void* G, void* W1, void* V, int* k);

/* ================================ FastSLAM-2 particle set ==================================== */
typedef struct cslam_pf* cslam_pf_t;

/* Replaces PF::initializeParticles(numParticles)  -- slam.h:688, PF.cpp:319-341, for the
 * n_particles this process owns (one shard of the global set; see INTEGRATION.md for the sharding).
 * Layout is structure-of-arrays in HBM. */
int cslam_pf_create(int n_particles, int max_features, int dtype, int device, int quirks, cslam_pf_t* out);
int cslam_pf_destroy(cslam_pf_t h);
int cslam_pf_synchronize(cslam_pf_t h);
int cslam_pf_get_counts(cslam_pf_t h, int* n_particles, int* n_features);

/* set every particle's weight to w0 (PF.cpp:327 uses 1/N of the GLOBAL particle count) */
int cslam_pf_set_uniform_weight(cslam_pf_t h, double w0);

/* Replaces PF::predict(particle, v, swa, Q, wb, dt) for every owned particle -- slam.h:858-863,
 * PF.cpp:419-471. */
int cslam_pf_predict(cslam_pf_t h, double v, double swa, const void* Q, double wb, double dt);

/* Replaces PF::observeHeading(particle, phi, use) for every owned particle -- PF.cpp:382-417. */
int cslam_pf_observe_heading(cslam_pf_t h, double phi, int use_heading);

/* Replaces PF::sampleProposal(particle, Z, idf, R) for every owned particle -- slam.h:881-884,
 * PF.cpp:502-544 (computeJacobians PF.cpp:70-135, likelihood 343-359, gaussEvaluate 279-317).
 * normals: 3 * n_particles scalars (host), the N(0,1) draws slam.h:753-764 would make, COMPONENT-major:
 * normals[e * n_particles + p] is draw e (0..2) of particle p (i.e. an n_particles x 3 column-major matrix).
 * Like every per-particle call of this section, it consumes its host arrays before it returns but does
 * not wait for the device: the work is ordered on the handle's stream, and cslam_pf_synchronize() (or
 * any call that returns data to the host) waits for it. */
int cslam_pf_sample_proposal(cslam_pf_t h, const void* Z, int m, const int* idf, const void* R,
                             const void* normals);

/* Replaces PF::featureUpdate(particle, Z, idf, R) for every owned particle -- slam.h:549-552,
 * PF.cpp:222-277. */
int cslam_pf_feature_update(cslam_pf_t h, const void* Z, int m, const int* idf, const void* R);

/* Replaces PF::addOneNewFeature(particle, Z, R) for every owned particle -- slam.h:134, PF.cpp:9-60. */
int cslam_pf_add_features(cslam_pf_t h, const void* Z, int q, const void* R);

/* ---- the resample step (PF.cpp:473-500, 546-577), split so that a multi-GPU driver can put its
 *      collectives between the pieces; see INTEGRATION.md ---- */
/* local partial sums: sums[0] = sum w, sums[1] = sum w^2 (doubles, host). Synchronises. */
int cslam_pf_weight_sums(cslam_pf_t h, double* sums);
/* w *= scale (the 1/ws of PF.cpp:482-487 with ws the GLOBAL sum) */
int cslam_pf_scale_weights(cslam_pf_t h, double scale);
/* device pointer to the owned weights (n_particles scalars) for an all-gather */
int cslam_pf_weights_device_ptr(cslam_pf_t h, void** dptr);
int cslam_pf_get_weights(cslam_pf_t h, void* w_host); /* synchronises */
int cslam_pf_set_weights(cslam_pf_t h, const void* w_host);
/* size in bytes of one packed particle record (w, Xv, Pv, XF, PF for max_features) */
int cslam_pf_record_bytes(cslam_pf_t h, long long* bytes);
/* pack the particles src_idx[0..count) (local 0-based indices, host array) into the device buffer
 * d_records (count * record_bytes), e.g. a send buffer of the all-to-all-v */
int cslam_pf_pack(cslam_pf_t h, const int* src_idx, int count, void* d_records);
/* overwrite local slots dst_idx[0..count) from packed records in device memory */
int cslam_pf_unpack(cslam_pf_t h, const int* dst_idx, int count, const void* d_records);
/* purely local resample: slot i <- copy of local particle keep[i] (0-based), all weights = w_new */
/* PF::resampleParticles (PF.cpp:473-500) with stratifiedResample (PF.cpp:546-574) when ONE handle holds the whole
 * particle set: weight sums, normalisation, Neff, the decision (Neff < n_effective && resample_status), keep[] and
 * the particle moves all run on the device; `select` are the N strata positions (PF.cpp:557, host pointer).
 * neff / resampled may be NULL (then nothing returns to the host). Same results as weight_sums + scale_weights +
 * host keep[] + gather_local.  Any particle count (the running sum is sequential, as in the reference: 8192 weights
 * are staged in LDS at a time). */
int cslam_pf_resample_local(cslam_pf_t h, const void* select, double n_effective, int resample_status, double* neff,
                            int* resampled);
int cslam_pf_gather_local(cslam_pf_t h, const int* keep, double w_new);
/* One whole observation step of the reference's FastSLAM-2 loop (test/main.cpp:279-311) for a handle that holds every
 * particle: PF::predict (PF.cpp:419-471), PF::sampleProposal (PF.cpp:502-544), PF::featureUpdate (PF.cpp:222-277) and
 * PF::resampleParticles (PF.cpp:473-500), same arguments as the individual calls, with ONE staged host-to-device copy
 * for all the small inputs and nothing returned to the host (cslam_pf_resample_stats reports what happened). */
int cslam_pf_observation_step(cslam_pf_t h, double v, double swa, const void* Q, double wb, double dt, const void* Z, int m,
                              const int* idf, const void* R, const void* normals, const void* select, double n_effective,
                              int resample_status);
/* resample calls / resamples performed since the handle was created and the last Neff (device-side counters of
 * cslam_pf_resample_local and cslam_pf_observation_step; any pointer may be NULL).  Synchronises. */
int cslam_pf_resample_stats(cslam_pf_t h, double* calls, double* resamples, double* last_neff);

/* ---- the resample step over a particle set SHARDED across GPUs, one process (rank) per GPU:
 * PF::resampleParticles (slam.h:871-872, PF.cpp:473-500) with stratifiedResample (PF.cpp:546-574) where every rank
 * holds n_particles of the world * n_particles particles (block partition: global slot g lives on rank g / n_particles).
 * The collectives run over RCCL (xGMI inside a node) on the handle's stream:
 *     1. all-reduce(sum) of [sum w, sum w^2]                       -> normalisation and Neff, identical on every rank
 *     2. only when it resamples: all-gather of the normalised weights; every rank derives the identical keep[] on its
 *        own device from the shared strata positions
 *     3. one grouped send/recv of the packed particle records whose source rank differs from the destination rank
 * `select`: the world * n_particles strata positions of PF.cpp:557 (host pointer), identical on every rank.
 * neff / resampled may be NULL.  The librccl of the process is bound at run time (dlopen): there is no link-time
 * dependency, and a process that never shards never loads it.
 * A communicator is made the RCCL way: rank 0 calls cslam_comm_unique_id, distributes the CSLAM_COMM_ID_BYTES bytes by
 * whatever means the application has (MPI_Bcast, a file, torch.distributed), every rank calls cslam_comm_create. */
typedef struct cslam_comm* cslam_comm_t;
#define CSLAM_COMM_ID_BYTES 128
int cslam_comm_unique_id(void* id_bytes);
int cslam_comm_create(const void* id_bytes, int rank, int world, int device, cslam_comm_t* out);
int cslam_comm_destroy(cslam_comm_t c);
int cslam_comm_info(cslam_comm_t c, int* rank, int* world); /* either pointer may be NULL */
/* A LOOPBACK communicator: `world` (<= 16) ranks inside ONE process on ONE device, out[0..world) their handles.  RCCL
 * refuses the same device twice in one communicator, so this is how the multi-rank paths of cslam_pf_resample_sharded
 * (ranks > 0, the exchange plan, the receive ordering by source rank) run on a one-GPU box: all-reduce, all-gather and
 * the grouped send/recv become device-to-device copies ordered by a host barrier.  Every rank's
 * cslam_pf_resample_sharded must be called from its OWN host thread (the ranks meet inside the call, as processes do
 * over RCCL); a rank that fails or does not arrive within 60 s breaks the communicator for all.  For tests and for
 * sharding one GPU's particle set by hand -- not a fast path. */
int cslam_comm_create_loopback(int world, int device, cslam_comm_t* out);
int cslam_pf_resample_sharded(cslam_pf_t h, cslam_comm_t comm, const void* select, double n_effective,
                              int resample_status, double* neff, int* resampled);
/* Introspection for tests: the exchange plan of the LAST cslam_pf_resample_sharded on this handle.  counts: 2 * world
 * ints -- records sent to each destination rank, then records received from each source rank (all 0 when it did not
 * resample); send_idx: the local source indices in send order (capacity >= *n_send).  Any pointer may be NULL. */
int cslam_pf_debug_last_exchange(cslam_pf_t h, int* counts, int* send_idx, int capacity, int* n_send);
/* download one particle (host buffers; any may be NULL): w (1), Xv (3), Pv (9), XF (2*nf), PF (4*nf) */
int cslam_pf_get_particle(cslam_pf_t h, int index, void* w, void* Xv, void* Pv, void* XF, void* PF);
/* upload one particle with nf features (nf must equal the current feature count, or set it when the
 * store is empty of features) */
int cslam_pf_set_particle(cslam_pf_t h, int index, const void* w, const void* Xv, const void* Pv,
                          const void* XF, const void* PF, int nf);

/* ------------------------------------------------------------------------------------------------
 * Device-side observation generator and known-association table: the per-step host work of
 * the reference's driver (test/main.cpp:139-165) -- a visibility filter over all landmarks and a table lookup per
 * observation -- with the map, the table, the scan and its split resident in HBM.
 *   cslam_sim_get_observations      Slam::getObservations, slam.h:575-683 with computeRangeBearing slam.h:339-368:
 *                                   landmarks with |dx|,|dy| < rmax, in front of the vehicle and inside the range
 *                                   circle, ascending tag order; Z (2 x m) and tags (1-based) are copied to the host
 *                                   pointers when these are not NULL.
 *   cslam_sim_add_observation_noise slam.h:168-178: Z[r][i] += normals[2i+r] * sqrt(R[r][r]) on the device-resident
 *                                   scan (the N(0,1) draws are an input, as for the particle filter).
 *   cslam_sim_associate_table       EKF::dataAssociateTable, EKF.cpp:146-233, on the device-resident scan: known tags
 *                                   -> (ZF, idf = their state position), unknown tags -> ZN, and the table assigns
 *                                   them the positions n_features+1, n_features+2, ... in scan order.
 *   cslam_sim_device_ptrs           the device-resident ZF / idf / ZN / Z / tags, e.g. for cslam_ekf_update_device.
 * All calls are synchronous with respect to their host outputs. */
typedef struct cslam_sim* cslam_sim_t;
int cslam_sim_create(const void* LM, int n_landmarks, int dtype, int device, cslam_sim_t* out);
int cslam_sim_destroy(cslam_sim_t h);
int cslam_sim_get_observations(cslam_sim_t h, const void* xv_true, double rmax, void* Z, int* tags, int* m);
int cslam_sim_add_observation_noise(cslam_sim_t h, const void* R, const void* normals);
int cslam_sim_associate_table(cslam_sim_t h, int n_features, void* ZF, int* idf, int* mf, void* ZN, int* mn);
int cslam_sim_device_ptrs(cslam_sim_t h, const void** dZF, const int** dIdf, const void** dZN, const void** dZ,
                          const int** dTags);
int cslam_sim_get_table(cslam_sim_t h, int* table);
int cslam_sim_set_table(cslam_sim_t h, const int* table);

#ifdef __cplusplus
}
#endif
#endif /* CSLAM_H */
