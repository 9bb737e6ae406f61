/*
 * slam_oracle.h -- CPU restatement of the mfkiwl/conan-slam EKF-SLAM / FastSLAM hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link or call it; the HIP engine never does.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (its arithmetic lives in the
 * un-vendored Eigen 3.4.0 / Boost 1.84.0, conanfile.py:54-55) and ships no tests, golden vectors or
 * known-answer values.  This restatement is pinned instead by
 *   (i)  an independent numpy/scipy restatement (oracle/np_restatement.py) -> tests/golden/,
 *   (ii) first-principles known-answer tests (tests/test_oracle_*.py).
 *
 * Conventions (same as the reference, which uses Eigen's defaults):
 *   - all matrices column-major; element (r,c) of a matrix with leading dimension ld is a[c*ld+r];
 *   - state X = [x, y, phi, lm1x, lm1y, lm2x, ...], length n = 3 + 2*N;
 *   - feature indices idf are 1-BASED (EKF.cpp:357);
 *   - Z is 2 x m column-major: Z[2*i] = range, Z[2*i+1] = bearing.
 * Every function exists in an _f32 (reference-faithful, the reference is MatrixXf throughout) and
 * an _f64 (high-precision) flavour generated from slam_oracle_impl.inc.
 *
 * Quirk flags: REF_EXACT reproduces the reference's behaviour bug for bug,
 * TEXTBOOK is the algebra the reference meant.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* quirk bit-flags */
#define ORC_Q_LOWER_CHOL_GAIN 1 /* slam.h:250-260 + 423: W1 = PHT * inv(L) with L lower (quirk #1)        */
#define ORC_Q_PREDICT_NM4 2     /* EKF.cpp:442-443: cross-covariance stripe is n-4 wide (quirk #2)       */
#define ORC_Q_REF_EXACT (ORC_Q_LOWER_CHOL_GAIN | ORC_Q_PREDICT_NM4)
#define ORC_Q_TEXTBOOK 0

/* return codes of the factorisation helpers */
#define ORC_CHOL_OK 0       /* LLT succeeded                                   (slam.h:421-424) */
#define ORC_CHOL_EIGEN 1    /* LLT failed, eigen "square root" used            (slam.h:425-429) */
#define ORC_CHOL_ZEROED 2   /* factor (or its inverse) non-finite -> zeros     (slam.h:431-434, 252-255) */

#define ORC_DECL(T, S)                                                                                      \
    /* slam.h:816-829 */                                                                                   \
    T orc_pi2pi_##S(T angle);                                                                               \
    /* slam.h:776-779 (in place, k x k, ld = k) */                                                          \
    void orc_make_symmetric_##S(T* A, int k);                                                               \
    /* slam.h:413-436: lower factor (or eigen fallback) of the k x k matrix M into Lout */                  \
    int orc_cholesky_decomposition_##S(const T* M, int k, T* Lout);                                         \
    /* general inverse by partially pivoted LU (what Eigen's MatrixXf::inverse() does), k x k */            \
    void orc_inverse_##S(const T* A, int k, T* Ainv);                                                       \
    /* slam.h:250-255: the k x k factor the reference calls SCHOLINV, from the symmetrised S.            \
       REF_EXACT: inv(L); TEXTBOOK: inv(L)^T; zeros when non-finite. Returns an ORC_CHOL_* code. */         \
    int orc_gain_factor_##S(const T* Smat, int k, int quirks, T* G);                                        \
    /* slam.h:235-266, dense operation order. X (n), P (n x n, ldp), V (k), R (k x k), H (k x n, ldh=k).    \
       work-free interface: allocates internally. Returns an ORC_CHOL_* code. */                            \
    int orc_cholesky_update_##S(T* X, T* P, int n, int ldp, const T* V, const T* R, const T* H, int k,      \
                                int quirks);                                                                \
    /* slam.h:700-725, dense operation order (n^3). */                                                      \
    void orc_joseph_update_##S(T* X, T* P, int n, int ldp, const T* V, const T* R, const T* H, int k);      \
    /* EKF.cpp:354-404: Zp (2), H (2 x n, ld 2, zero-filled) */                                             \
    void orc_ekf_observe_model_##S(const T* X, int n, int idf, T* Zp, T* H);                                \
    /* EKF.cpp:131-144: normalised innovation squared and normalised distance of observation z against    \
       feature idf (1-based), dense operation order; out[0] = nis, out[1] = nd */                          \
    void orc_ekf_compute_association_##S(const T* X, const T* P, int n, int ldp, const T* z, const T* R,    \
                                         int idf, T* out);                                                  \
    /* EKF.cpp:235-326: gated nearest neighbour, linear search.  kind[i]: 0 dropped, 1 associated with     \
       feature idf_out[i] (1-based), 2 far enough from everything (outer > gate2) to be a new feature.     \
       (The reference's return statement hands back an EMPTY new-feature list because line 307 re-declares \
       ZN inside the try block and line 311 never advances its column index; kind = 2 is what the loop     \
       decided, the caller applies that quirk.) */                                                         \
    void orc_ekf_data_associate_##S(const T* X, const T* P, int n, int ldp, const T* Z, int m, const T* R,  \
                                    T gate1, T gate2, int* idf_out, int* kind);                             \
    /* EKF.cpp:406-455 */                                                                                   \
    void orc_ekf_predict_##S(T* X, T* P, int n, int ldp, T v, T swa, const T* Q, T wb, T dt, int quirks);   \
    /* EKF.cpp:93-129 */                                                                                    \
    int orc_ekf_batch_update_##S(T* X, T* P, int n, int ldp, const T* Z, int m, const T* R, const int* idf, \
                                 int quirks);                                                               \
    /* EKF.cpp:457-479 */                                                                                   \
    int orc_ekf_single_update_##S(T* X, T* P, int n, int ldp, const T* Z, int m, const T* R,                \
                                  const int* idf, int quirks);                                              \
    /* EKF.cpp:481-496 */                                                                                   \
    int orc_ekf_update_##S(T* X, T* P, int n, int ldp, const T* Z, int m, const T* R, const int* idf,       \
                           int batch, int quirks);                                                          \
    /* EKF.cpp:9-91: appends q features; X has room for n+2q, P for (n+2q)^2 with leading dim ldp.          \
       Returns the new n. */                                                                                \
    int orc_ekf_augment_##S(T* X, T* P, int n, int ldp, const T* Z, int q, const T* R);                     \
    /* EKF.cpp:328-352 (dense Joseph form) */                                                               \
    void orc_ekf_observe_heading_##S(T* X, T* P, int n, int ldp, T phi, int use_heading);                   \
    /* the same update in its rank-structured O(n^2) form; used to check the algebra the        \
       device kernel implements against the dense form above */                                            \
    void orc_ekf_observe_heading_structured_##S(T* X, T* P, int n, int ldp, T phi, int use_heading);        \
    /* ---- the reference's update in DENSE operation order, written for speed: this is the timed           \
       cpu_baseline ("port").  Same arithmetic as orc_ekf_batch_update, blocked loops. ---- */              \
    int orc_ekf_batch_update_fast_##S(T* X, T* P, int n, int ldp, const T* Z, int m, const T* R,            \
                                      const int* idf, int quirks);                                          \
    /* ---------------- simulator helpers (harness side) ---------------- */                     \
    /* slam.h:952-966 */                                                                                    \
    void orc_vehicle_model_##S(T* Xv, T v, T swa, T wb, T dt);                                              \
    /* slam.h:279-332; WP is 2 x nwp; iwp 1-based in/out (0 = finished); swa in/out; int_signum = 1     \
       reproduces signum<int>()'s truncation of its float argument (slam.h:317,324) */                      \
    void orc_compute_swa_##S(const T* Xv, const T* WP, int nwp, int* iwp, T minD, T* swa, T rateSWA,        \
                             T maxSWA, T dt, int int_signum);                                               \
    /* slam.h:575-683 + 339-368: visible landmarks -> Z (2 x out), tags (1-based); returns count */         \
    int orc_get_observations_##S(const T* Xv, const T* LM, int nlm, T rmax, T* Z, int* tags);               \
    /* ---------------- FastSLAM-2 per-particle path (slam_oracle_pf.inc) ---------------- */               \
    /* PF.cpp:419-471 */                                                                                    \
    void orc_pf_predict_##S(T* Xv, T* Pv, T v, T swa, const T* Q, T wb, T dt);                              \
    /* PF.cpp:382-417 */                                                                                    \
    void orc_pf_observe_heading_##S(T* Xv, T* Pv, T phi, int use_heading);                                  \
    /* PF.cpp:70-135 */                                                                                     \
    void orc_pf_compute_jacobians_##S(const T* Xv, const T* XF, const T* PF, const int* idf, int len,       \
                                      const T* R, T* ZP, T* HV, T* HF, T* SF);                              \
    /* PF.cpp:279-317 */                                                                                    \
    T orc_pf_gauss_evaluate_##S(const T* V, const T* Smat, int D, int log_flag);                            \
    /* PF.cpp:343-359 */                                                                                    \
    T orc_pf_likelihood_##S(const T* Xv, const T* XF, const T* PF, const T* Z, const int* idf, int m,       \
                            const T* R);                                                                    \
    /* PF.cpp:502-544; normals = the 3 N(0,1) draws (input) */                               \
    void orc_pf_sample_proposal_##S(T* w, T* Xv, T* Pv, const T* XF, const T* PF, const T* Z,               \
                                    const int* idf, int m, const T* R, const T* normals);                   \
    /* PF.cpp:222-277 */                                                                                    \
    void orc_pf_feature_update_##S(const T* Xv, T* XF, T* PF, const T* Z, const int* idf, int m,            \
                                   const T* R, int quirks);                                                 \
    /* PF.cpp:9-60 */                                                                                       \
    int orc_pf_add_features_##S(const T* Xv, T* XF, T* PF, int nf, const T* Z, int q, const T* R);          \
    /* PF.cpp:579-596 */                                                                                    \
    void orc_pf_stratified_random_##S(int n, const T* noise, int ref_exact, T* out);                        \
    /* PF.cpp:546-577 */                                                                                    \
    T orc_pf_stratified_resample_##S(T* w, int n, const T* select, int* keep, int ref_exact);               \
    /* PF.cpp:473-500 with the index bug removed */                                         \
    T orc_pf_normalize_resample_##S(T* w, int n, int n_effective, int flag, const T* select, int* keep,     \
                                    int* resampled);

ORC_DECL(float, f32)
ORC_DECL(double, f64)

/* EKF.cpp:146-233: known-association table. table has one int per landmark tag (0 = unseen).
 * Splits the m observations into associated (ZF, idf: state index 1-based) and new (ZN).
 * nf = number of features already in the state.  Returns mf (count in ZF); *mn gets count in ZN.
 * Works on raw 2 x m blocks of either precision via elem_size (4 or 8). */
int orc_data_associate_table(const void* Z, const int* tags, int m, int* table, int nf, void* ZF, int* idf,
                             void* ZN, int* mn, int elem_size);

#ifdef __cplusplus
}
#endif
#endif
