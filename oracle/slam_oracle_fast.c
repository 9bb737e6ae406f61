/*
 * slam_oracle_fast.c -- the reference's batch update in its DENSE operation order, written with
 * blocked loops so that it is a fair single-core CPU baseline ("port") for bench.py.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see slam_oracle.h); never part of the HIP engine.
 *
 * What is restated (slam.h:235-266 as reached from EKF.cpp:93-129):
 *     PHT = P * H^T            dense n x n by n x k GEMM, zeros of H included, as Eigen does it
 *     S   = H * PHT + R ; symmetrise ; L = chol(S) ; G = inv(L)
 *     W1  = PHT * G ; W = W1 * G^T ; X += W * V
 *     P   = P - W1 * W1^T      via an n x n temporary (Eigen evaluates `P - W1*W1^T` into a
 *                              temporary because products are assumed to alias, then copies back)
 * The GEMMs use a small register-blocked NT micro-kernel on GCC vector extensions; it is not Eigen's
 * kernel, hence "port" and not "reference" in bench.py's cpu_baseline.kind.
 */
#include "slam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(r, c, ld) ((size_t)(c) * (size_t)(ld) + (size_t)(r))

#define FAST_IMPL(T, S, VL)                                                                                  \
    typedef T vec_##S __attribute__((vector_size(VL * sizeof(T)), aligned(sizeof(T))));                     \
    /* C (M x N, ldc) = beta*C + alpha * A (M x K, lda) * Bm^T, Bm is N x K with leading dim ldb */          \
    static void gemm_nt_##S(int M, int N, int K, T alpha, const T* A, int lda, const T* Bm, int ldb,         \
                            T beta, T* C, int ldc)                                                           \
    {                                                                                                        \
        const int MR = 4 * VL, NR = 4;                                                                       \
        int       Mb = (M / MR) * MR, Nb = (N / NR) * NR;                                                    \
        for (int i0 = 0; i0 < Mb; i0 += MR)                                                                  \
        {                                                                                                    \
            for (int j0 = 0; j0 < Nb; j0 += NR)                                                              \
            {                                                                                                \
                vec_##S acc[4][4];                                                                           \
                for (int a = 0; a < 4; a++)                                                                  \
                    for (int b = 0; b < 4; b++)                                                              \
                        acc[a][b] = (vec_##S){0};                                                            \
                for (int l = 0; l < K; l++)                                                                  \
                {                                                                                            \
                    const T* ap = &A[IDX(i0, l, lda)];                                                       \
                    vec_##S  a0 = *(const vec_##S*)(ap), a1 = *(const vec_##S*)(ap + VL),                    \
                            a2 = *(const vec_##S*)(ap + 2 * VL), a3 = *(const vec_##S*)(ap + 3 * VL);        \
                    const T* bp = &Bm[IDX(j0, l, ldb)];                                                      \
                    for (int b = 0; b < 4; b++)                                                              \
                    {                                                                                        \
                        T bv = bp[b];                                                                        \
                        acc[b][0] += a0 * bv;                                                                \
                        acc[b][1] += a1 * bv;                                                                \
                        acc[b][2] += a2 * bv;                                                                \
                        acc[b][3] += a3 * bv;                                                                \
                    }                                                                                        \
                }                                                                                            \
                for (int b = 0; b < 4; b++)                                                                  \
                {                                                                                            \
                    T* cp = &C[IDX(i0, j0 + b, ldc)];                                                        \
                    for (int a = 0; a < 4; a++)                                                              \
                    {                                                                                        \
                        vec_##S* cv = (vec_##S*)(cp + a * VL);                                               \
                        if (beta == (T)0)                                                                    \
                            *cv = alpha * acc[b][a];                                                         \
                        else                                                                                 \
                            *cv = beta * (*cv) + alpha * acc[b][a];                                          \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
        /* edges: plain loops */                                                                             \
        for (int j = 0; j < N; j++)                                                                          \
        {                                                                                                    \
            int istart = (j < Nb) ? Mb : 0;                                                                  \
            for (int i = istart; i < M; i++)                                                                 \
            {                                                                                                \
                T s = (T)0;                                                                                  \
                for (int l = 0; l < K; l++)                                                                  \
                    s += A[IDX(i, l, lda)] * Bm[IDX(j, l, ldb)];                                             \
                T* c = &C[IDX(i, j, ldc)];                                                                   \
                *c   = (beta == (T)0) ? alpha * s : beta * (*c) + alpha * s;                                 \
            }                                                                                                \
        }                                                                                                    \
    }                                                                                                        \
                                                                                                             \
    int orc_ekf_batch_update_fast_##S(T* X, T* P, int n, int ldp, const T* Z, int m, const T* R,             \
                                      const int* idf, int quirks)                                            \
    {                                                                                                        \
        int k = 2 * m;                                                                                       \
        if (m == 0)                                                                                          \
            return ORC_CHOL_OK;                                                                              \
        size_t nk  = (size_t)n * (size_t)k;                                                                  \
        T*     H   = (T*)calloc(nk, sizeof(T)); /* k x n, ld k */                                            \
        T*     V   = (T*)calloc((size_t)k, sizeof(T));                                                       \
        T*     RR  = (T*)calloc((size_t)k * k, sizeof(T));                                                   \
        T*     HT  = (T*)calloc(2 * (size_t)n, sizeof(T));                                                   \
        T*     PHT = (T*)malloc(nk * sizeof(T)); /* n x k, ld n */                                           \
        T*     Sm  = (T*)malloc((size_t)k * k * sizeof(T));                                                  \
        T*     G   = (T*)malloc((size_t)k * k * sizeof(T));                                                  \
        T*     Gt  = (T*)malloc((size_t)k * k * sizeof(T));                                                  \
        T*     W1  = (T*)malloc(nk * sizeof(T));                                                             \
        T*     W   = (T*)malloc(nk * sizeof(T));                                                             \
        T*     TMP = (T*)malloc((size_t)n * n * sizeof(T));                                                  \
        for (int i = 0; i < m; i++) /* EKF.cpp:108-121 */                                                    \
        {                                                                                                    \
            T Zp[2];                                                                                         \
            orc_ekf_observe_model_##S(X, n, idf[i], Zp, HT);                                                 \
            for (int j = 0; j < n; j++)                                                                      \
            {                                                                                                \
                H[IDX(2 * i, j, k)]     = HT[IDX(0, j, 2)];                                                  \
                H[IDX(2 * i + 1, j, k)] = HT[IDX(1, j, 2)];                                                  \
            }                                                                                                \
            V[2 * i]     = Z[2 * i] - Zp[0];                                                                 \
            V[2 * i + 1] = orc_pi2pi_##S(Z[2 * i + 1] - Zp[1]);                                              \
            for (int c = 0; c < 2; c++)                                                                      \
                for (int r = 0; r < 2; r++)                                                                  \
                    RR[IDX(2 * i + r, 2 * i + c, k)] = R[IDX(r, c, 2)];                                      \
        }                                                                                                    \
        /* slam.h:243 */                                                                                     \
        gemm_nt_##S(n, k, n, (T)1, P, ldp, H, k, (T)0, PHT, n);                                              \
        /* slam.h:244: S = H*PHT + R  (k x n by n x k) */                                                    \
        for (int c = 0; c < k; c++)                                                                          \
            for (int r = 0; r < k; r++)                                                                      \
            {                                                                                                \
                T s = (T)0;                                                                                  \
                for (int j = 0; j < n; j++)                                                                  \
                    s += H[IDX(r, j, k)] * PHT[IDX(j, c, n)];                                                \
                Sm[IDX(r, c, k)] = s + RR[IDX(r, c, k)];                                                     \
            }                                                                                                \
        orc_make_symmetric_##S(Sm, k);                     /* slam.h:247 */                                  \
        int code = orc_gain_factor_##S(Sm, k, quirks, G);  /* slam.h:250-255 */                              \
        for (int c = 0; c < k; c++)                                                                          \
            for (int r = 0; r < k; r++)                                                                      \
                Gt[IDX(r, c, k)] = G[IDX(c, r, k)];                                                          \
        gemm_nt_##S(n, k, k, (T)1, PHT, n, Gt, k, (T)0, W1, n); /* slam.h:257 W1 = PHT*G     */              \
        gemm_nt_##S(n, k, k, (T)1, W1, n, G, k, (T)0, W, n);    /* slam.h:258 W  = W1*G^T    */              \
        for (int i = 0; i < n; i++)                              /* slam.h:259 */                            \
        {                                                                                                    \
            T s = (T)0;                                                                                      \
            for (int q = 0; q < k; q++)                                                                      \
                s += W[IDX(i, q, n)] * V[q];                                                                 \
            X[i] = X[i] + s;                                                                                 \
        }                                                                                                    \
        /* slam.h:260: tmp = P; tmp -= W1*W1^T; P = tmp */                                                   \
        for (int j = 0; j < n; j++)                                                                          \
            memcpy(&TMP[IDX(0, j, n)], &P[IDX(0, j, ldp)], sizeof(T) * (size_t)n);                           \
        gemm_nt_##S(n, n, k, (T)-1, W1, n, W1, n, (T)1, TMP, n);                                             \
        for (int j = 0; j < n; j++)                                                                          \
            memcpy(&P[IDX(0, j, ldp)], &TMP[IDX(0, j, n)], sizeof(T) * (size_t)n);                           \
        free(H);                                                                                             \
        free(V);                                                                                             \
        free(RR);                                                                                            \
        free(HT);                                                                                            \
        free(PHT);                                                                                           \
        free(Sm);                                                                                            \
        free(G);                                                                                             \
        free(Gt);                                                                                            \
        free(W1);                                                                                            \
        free(W);                                                                                             \
        free(TMP);                                                                                           \
        return code;                                                                                         \
    }

FAST_IMPL(float, f32, 8)
FAST_IMPL(double, f64, 4)
