// pf_kernels.hpp -- gfx950 kernels of the FastSLAM-2 per-particle path and of the resample data movement.
//
// Particle store, structure-of-arrays in HBM with the particle index fastest (np = particles owned by
// this handle, so a wave reads 64 consecutive particles' values with one coalesced access):
//   w [np]                    weights                                     (slam.h:122)
//   xv[3][np]                 pose means                                  (slam.h:123)
//   pv[9][np]                 pose covariances, column-major 3x3          (slam.h:124)
//   xf[nfcap][2][np]          feature means                               (slam.h:125)
//   pf[nfcap][4][np]          feature covariances, column-major 2x2       (slam.h:126)
// One lane owns one particle (or one (particle, observation) pair where observations are independent);
// all the 2x2 / 3x3 algebra is in registers, in the reference's operation order (citations per function).
#pragma once

#include <hip/hip_runtime.h>

#include "device_math.hpp" // pi2pi and the libm shims

// Contraction is switched off here: these kernels are latency-bound, and a*b+c with two roundings is what
// a non-FMA build of the reference (and the CPU oracle) computes.
#pragma clang fp contract(off)

namespace cslam
{

__device__ inline float  dexp(float x) { return expf(x); }
__device__ inline double dexp(double x) { return exp(x); }
__device__ inline float  dfabs(float x) { return fabsf(x); }
__device__ inline double dfabs(double x) { return fabs(x); }

template <typename T>
struct PfStore
{
    T*  w;
    T*  xv;
    T*  pv;
    T*  xf;
    T*  pf;
    int np;
    int nf;
};

// a predict step riding inside pf_sample_proposal_kernel (on = 0: none)
template <typename T>
struct PfPredict
{
    int on;
    T   v, swa, q00, q10, q01, q11, wb, dt;
};

// ---------------------------------------------------------------- tiny dense helpers (column-major)
template <typename T, int RA, int CA, int CB>
__device__ inline void mm(const T* A, const T* B, T* C)
{
#pragma unroll
    for (int c = 0; c < CB; c++)
    {
#pragma unroll
        for (int r = 0; r < RA; r++)
        {
            T s = (T)0;
#pragma unroll
            for (int q = 0; q < CA; q++)
            {
                s += A[r + RA * q] * B[q + CA * c];
            }
            C[r + RA * c] = s;
        }
    }
}

template <typename T, int RA, int CA>
__device__ inline void tr(const T* A, T* At)
{
#pragma unroll
    for (int c = 0; c < CA; c++)
    {
#pragma unroll
        for (int r = 0; r < RA; r++)
        {
            At[c + CA * r] = A[r + RA * c];
        }
    }
}

template <typename T, int D>
__device__ inline bool all_finite(const T* A)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D * D; i++)
    {
        ok = ok && dfinite(A[i]);
    }
    return ok;
}

// Eigen::LLT, lower, reads the lower triangle; returns true on failure (pivot <= 0), slam.h:417-421
template <typename T, int D>
__device__ inline bool llt_lower(const T* M, T* L)
{
#pragma unroll
    for (int i = 0; i < D * D; i++)
    {
        L[i] = (T)0;
    }
#pragma unroll
    for (int c = 0; c < D; c++)
    {
#pragma unroll
        for (int r = c; r < D; r++)
        {
            L[r + D * c] = M[r + D * c];
        }
    }
    bool failed = false;
#pragma unroll
    for (int j = 0; j < D; j++)
    {
        if (!failed)
        {
            T x = L[j + D * j];
#pragma unroll
            for (int q = 0; q < j; q++)
            {
                x -= L[j + D * q] * L[j + D * q];
            }
            if (x <= (T)0)
            {
                failed = true;
            }
            else
            {
                x            = dsqrt(x);
                L[j + D * j] = x;
#pragma unroll
                for (int r = j + 1; r < D; r++)
                {
                    T s = L[r + D * j];
#pragma unroll
                    for (int q = 0; q < j; q++)
                    {
                        s -= L[r + D * q] * L[j + D * q];
                    }
                    L[r + D * j] = s / x;
                }
            }
        }
    }
    return failed;
}

// cyclic Jacobi, eigenvalues ascending (stands in for SelfAdjointEigenSolver, slam.h:427)
template <typename T, int D>
__device__ inline void jacobi_eigh(const T* M, T* ev, T* V)
{
    T A[D * D];
#pragma unroll
    for (int c = 0; c < D; c++)
    {
#pragma unroll
        for (int r = 0; r < D; r++)
        {
            A[r + D * c] = (r >= c) ? M[r + D * c] : M[c + D * r];
            V[r + D * c] = (r == c) ? (T)1 : (T)0;
        }
    }
    for (int sweep = 0; sweep < 64; sweep++)
    {
        double off = 0.0;
#pragma unroll
        for (int c = 0; c < D; c++)
        {
#pragma unroll
            for (int r = c + 1; r < D; r++)
            {
                off += (double)A[r + D * c] * (double)A[r + D * c];
            }
        }
        if (!(off > 0.0))
        {
            break;
        }
#pragma unroll
        for (int p = 0; p < D - 1; p++)
        {
#pragma unroll
            for (int q = p + 1; q < D; q++)
            {
                T apq = A[p + D * q];
                if (apq != (T)0)
                {
                    T theta = (A[q + D * q] - A[p + D * p]) / ((T)2 * apq);
                    T t     = (theta >= (T)0 ? (T)1 : (T)-1) / (dfabs(theta) + dsqrt(theta * theta + (T)1));
                    T cs = (T)1 / dsqrt(t * t + (T)1), sn = t * cs;
#pragma unroll
                    for (int r = 0; r < D; r++)
                    {
                        T arp = A[r + D * p], arq = A[r + D * q];
                        A[r + D * p] = cs * arp - sn * arq;
                        A[r + D * q] = sn * arp + cs * arq;
                    }
#pragma unroll
                    for (int c = 0; c < D; c++)
                    {
                        T apc = A[p + D * c], aqc = A[q + D * c];
                        A[p + D * c] = cs * apc - sn * aqc;
                        A[q + D * c] = sn * apc + cs * aqc;
                    }
#pragma unroll
                    for (int r = 0; r < D; r++)
                    {
                        T vrp = V[r + D * p], vrq = V[r + D * q];
                        V[r + D * p] = cs * vrp - sn * vrq;
                        V[r + D * q] = sn * vrp + cs * vrq;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < D; i++)
    {
        ev[i] = A[i + D * i];
    }
    // ascending selection sort with a compile-time network (D <= 3)
#pragma unroll
    for (int i = 0; i < D - 1; i++)
    {
#pragma unroll
        for (int j = i + 1; j < D; j++)
        {
            // pick the smallest of the remainder: equivalent to selection sort for these sizes
            if (ev[j] < ev[i])
            {
                T tmp = ev[i];
                ev[i] = ev[j];
                ev[j] = tmp;
#pragma unroll
                for (int r = 0; r < D; r++)
                {
                    T tv         = V[r + D * i];
                    V[r + D * i] = V[r + D * j];
                    V[r + D * j] = tv;
                }
            }
        }
    }
}

// slam.h:413-436
template <typename T, int D>
__device__ inline void chol_decomp(const T* M, T* L)
{
    if (llt_lower<T, D>(M, L))
    {
        T ev[D], V[D * D];
        jacobi_eigh<T, D>(M, ev, V);
#pragma unroll
        for (int c = 0; c < D; c++)
        {
            T s = dsqrt(ev[c]);
#pragma unroll
            for (int r = 0; r < D; r++)
            {
                L[r + D * c] = V[r + D * c] * s;
            }
        }
    }
    if (!all_finite<T, D>(L))
    {
#pragma unroll
        for (int i = 0; i < D * D; i++)
        {
            L[i] = (T)0;
        }
    }
}

// MatrixXf::inverse() for dynamic sizes = LU with partial pivoting (slam.h:251, PF.cpp:292,518,523-524).
// Written branch-light with compile-time loops; the row permutation is applied to the right-hand sides.
template <typename T, int D>
__device__ inline void inverse_lu(const T* Ain, T* Ainv)
{
    T LU[D * D];
    T Bm[D * D]; // permuted identity
#pragma unroll
    for (int i = 0; i < D * D; i++)
    {
        LU[i] = Ain[i];
    }
#pragma unroll
    for (int c = 0; c < D; c++)
    {
#pragma unroll
        for (int r = 0; r < D; r++)
        {
            Bm[r + D * c] = (r == c) ? (T)1 : (T)0;
        }
    }
#pragma unroll
    for (int c = 0; c < D; c++)
    {
        int piv  = c;
        T   best = dfabs(LU[c + D * c]);
#pragma unroll
        for (int r = c + 1; r < D; r++)
        {
            T a = dfabs(LU[r + D * c]);
            if (a > best)
            {
                best = a;
                piv  = r;
            }
        }
        if (best != (T)0)
        {
#pragma unroll
            for (int r = c + 1; r < D; r++)
            {
                if (r == piv) // swap rows c and r of LU and of the right-hand sides
                {
#pragma unroll
                    for (int cc = 0; cc < D; cc++)
                    {
                        T t1          = LU[c + D * cc];
                        LU[c + D * cc] = LU[r + D * cc];
                        LU[r + D * cc] = t1;
                        T t2          = Bm[c + D * cc];
                        Bm[c + D * cc] = Bm[r + D * cc];
                        Bm[r + D * cc] = t2;
                    }
                }
            }
            T d = LU[c + D * c];
#pragma unroll
            for (int r = c + 1; r < D; r++)
            {
                LU[r + D * c] /= d;
            }
        }
#pragma unroll
        for (int cc = c + 1; cc < D; cc++)
        {
            T u = LU[c + D * cc];
#pragma unroll
            for (int r = c + 1; r < D; r++)
            {
                LU[r + D * cc] -= LU[r + D * c] * u;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < D; c++)
    {
        T x[D];
#pragma unroll
        for (int r = 0; r < D; r++)
        {
            x[r] = Bm[r + D * c];
        }
#pragma unroll
        for (int r = 0; r < D; r++)
        {
            T s = x[r];
#pragma unroll
            for (int q = 0; q < r; q++)
            {
                s -= LU[r + D * q] * x[q];
            }
            x[r] = s;
        }
#pragma unroll
        for (int r = D - 1; r >= 0; r--)
        {
            T s = x[r];
#pragma unroll
            for (int q = r + 1; q < D; q++)
            {
                s -= LU[r + D * q] * x[q];
            }
            x[r] = s / LU[r + D * r];
        }
#pragma unroll
        for (int r = 0; r < D; r++)
        {
            Ainv[r + D * c] = x[r];
        }
    }
}

// PF.cpp:279-317, likelihood form (logFlag = false)
template <typename T, int D>
__device__ inline T gauss_evaluate(const T* V, const T* S)
{
    T L[D * D], SC[D * D], SCI[D * D], nin[D];
    chol_decomp<T, D>(S, L);
    tr<T, D, D>(L, SC);
    inverse_lu<T, D>(SC, SCI);
    mm<T, D, D, 1>(SCI, V, nin);
    T sum = (T)0;
#pragma unroll
    for (int i = 0; i < D; i++)
    {
        nin[i] = nin[i] * nin[i];
        sum += nin[i];
    }
    T E    = (T)-0.5f * sum;
    T prod = (T)1;
#pragma unroll
    for (int i = 0; i < D; i++)
    {
        prod *= SC[i + D * i];
    }
    // std::pow(2.0F * pi, D / 2.0F): (2 pi)^(D/2) in double
    const double twopi = 2.0 * kPi;
    double       Cn    = (D == 2) ? twopi : twopi * sqrt(twopi);
    Cn                 = Cn * (double)prod;
    return (T)((double)dexp(E) / Cn);
}

// PF.cpp:70-135 for ONE feature: ZP(2), HV(2x3), HF(2x2), SF(2x2)
template <typename T>
__device__ inline void compute_jacobians(const T* X, const T* xf, const T* pf, const T* R, T* ZP, T* HV, T* HF, T* SF)
{
    T dx = xf[0] - X[0];
    T dy = xf[1] - X[1];
    T d2 = dx * dx + dy * dy;
    T d  = dsqrt(d2);
    ZP[0] = d;
    ZP[1] = pi2pi<T>(datan2(dy, dx) - X[2]);
    HV[0] = -dx / d;
    HV[2] = -dy / d;
    HV[4] = (T)0;
    HV[1] = dy / d2;
    HV[3] = -dx / d2;
    HV[5] = (T)-1;
    HF[0] = dx / d;
    HF[2] = dy / d;
    HF[1] = -dy / d2;
    HF[3] = dx / d2;
    T t1[4], hft[4], t2[4];
    mm<T, 2, 2, 2>(HF, pf, t1);
    tr<T, 2, 2>(HF, hft);
    mm<T, 2, 2, 2>(t1, hft, t2);
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        SF[e] = t2[e] + R[e];
    }
}

template <typename T>
__device__ inline void load_feature(const PfStore<T>& s, int p, int f, T* xf, T* pf)
{
    xf[0] = s.xf[((size_t)f * 2 + 0) * s.np + p];
    xf[1] = s.xf[((size_t)f * 2 + 1) * s.np + p];
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        pf[e] = s.pf[((size_t)f * 4 + e) * s.np + p];
    }
}

template <typename T>
__device__ inline void motion_jacobians(T phi, T v, T swa, T wb, T dt, T* Gv, T* Gu)
{
    T s = dsin(swa + phi), c = dcos(swa + phi);
    Gv[0] = (T)1;
    Gv[1] = (T)0;
    Gv[2] = (T)0;
    Gv[3] = (T)0;
    Gv[4] = (T)1;
    Gv[5] = (T)0;
    Gv[6] = -v * dt * s;
    Gv[7] = v * dt * c;
    Gv[8] = (T)1;
    Gu[0] = dt * c;
    Gu[1] = dt * s;
    Gu[2] = dt * dsin(swa) / wb;
    Gu[3] = -v * dt * s;
    Gu[4] = v * dt * c;
    Gu[5] = v * dt * dcos(swa) / wb;
}

// ---------------------------------------------------------------- PF.cpp:419-471
// PF::predict (PF.cpp:419-471) for one particle, in registers: X and P (column-major 3 x 3) in, predicted values out
template <typename T>
__device__ inline void pf_predict_state(T* X, T* P, T v, T swa, const T* Q, T wb, T dt)
{
    T Gv[9], Gu[6];
    T phi = X[2];
    motion_jacobians<T>(phi, v, swa, wb, dt, Gv, Gu);
    T GvT[9], t1[9], t2[9], GuQ[6], GuT[6], t3[9];
    tr<T, 3, 3>(Gv, GvT);
    mm<T, 3, 3, 3>(Gv, P, t1);
    mm<T, 3, 3, 3>(t1, GvT, t2);
    mm<T, 3, 2, 2>(Gu, Q, GuQ);
    tr<T, 3, 2>(Gu, GuT);
    mm<T, 3, 2, 3>(GuQ, GuT, t3);
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        P[i] = t2[i] + t3[i];
    }
    const T x0 = X[0] + v * dt * dcos(swa + phi);
    const T x1 = X[1] + v * dt * dsin(swa + phi);
    const T x2 = pi2pi<T>(X[2] + v * dt * dsin(swa) / wb);
    X[0]       = x0;
    X[1]       = x1;
    X[2]       = x2;
}

template <typename T>
__global__ void __launch_bounds__(64) pf_predict_kernel(PfStore<T> s, T v, T swa, T q00, T q10, T q01, T q11, T wb, T dt)
{
    int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= s.np)
    {
        return;
    }
    T X[3], P[9], Q[4] = {q00, q10, q01, q11};
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        X[i] = s.xv[(size_t)i * s.np + p];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        P[i] = s.pv[(size_t)i * s.np + p];
    }
    pf_predict_state<T>(X, P, v, swa, Q, wb, dt);
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        s.pv[(size_t)i * s.np + p] = P[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        s.xv[(size_t)i * s.np + p] = X[i];
    }
}

// ---------------------------------------------------------------- PF.cpp:382-417 -> slam.h:700-725 with n = 3, k = 1
template <typename T>
__global__ void __launch_bounds__(64) pf_heading_kernel(PfStore<T> s, T phi, T R)
{
    int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= s.np)
    {
        return;
    }
    T X[3], P[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        X[i] = s.xv[(size_t)i * s.np + p];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        P[i] = s.pv[(size_t)i * s.np + p];
    }
    const T H[3] = {(T)0, (T)0, (T)1};
    T       V    = pi2pi<T>(phi - X[2]);
    T       PHT[3], W[3];
#pragma unroll
    for (int i = 0; i < 3; i++) // P*H^T, dense order
    {
        T a = (T)0;
#pragma unroll
        for (int j = 0; j < 3; j++)
        {
            a += P[i + 3 * j] * H[j];
        }
        PHT[i] = a;
    }
    T S = (T)0;
#pragma unroll
    for (int j = 0; j < 3; j++)
    {
        S += H[j] * PHT[j];
    }
    S    = S + R;
    T SI = (T)1 / S;
    SI   = (SI + SI) * (T)0.5;
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        W[i] = PHT[i] * SI;
        X[i] = X[i] + W[i] * V;
    }
    T Cm[9], CP[9], out[9];
#pragma unroll
    for (int j = 0; j < 3; j++)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
        {
            Cm[i + 3 * j] = ((i == j) ? (T)1 : (T)0) - W[i] * H[j];
        }
    }
    mm<T, 3, 3, 3>(Cm, P, CP);
#pragma unroll
    for (int j = 0; j < 3; j++)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
        {
            T a = (T)0;
#pragma unroll
            for (int l = 0; l < 3; l++)
            {
                a += CP[i + 3 * l] * Cm[j + 3 * l];
            }
            T b  = (W[i] * R) * W[j];
            T o  = a + b;
            o    = o + ((i == j) ? (T)1 : (T)0) * (T)1.17549435e-38f;
            out[i + 3 * j] = o;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        s.xv[(size_t)i * s.np + p] = X[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        s.pv[(size_t)i * s.np + p] = out[i];
    }
}

// ---------------------------------------------------------------- PF.cpp:502-544 (+ 343-359, 279-317, 62-68)
// normals: [3][np] standard-normal draws (input)
// choleskyUpdate(XF, PF, V, R, HF) of one feature (slam.h:243-260 through PF.cpp:222-277): xf += W V, pf -= W1 W1^T
template <typename T>
__device__ inline void pf_feature_kf(const T* xf, const T* pf, const T* HF, const T* V, const T* R, int textbook, T* xf_new,
                                     T* pf_new)
{
    // choleskyUpdate(XF, PF, V, R, HF): slam.h:243-260
    T HFt[4], PHT[4], S[4], Lc[4], G[4], W1[4], Gt[4], W[4];
    tr<T, 2, 2>(HF, HFt);
    mm<T, 2, 2, 2>(pf, HFt, PHT);
    mm<T, 2, 2, 2>(HF, PHT, S);
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        S[e] = S[e] + R[e];
    }
    {
        T o  = (S[1] + S[2]) * (T)0.5;
        S[1] = o;
        S[2] = o;
        S[0] = (S[0] + S[0]) * (T)0.5;
        S[3] = (S[3] + S[3]) * (T)0.5;
    }
    chol_decomp<T, 2>(S, Lc);
    inverse_lu<T, 2>(Lc, G);
    if (textbook)
    {
        T t0 = G[1];
        G[1] = G[2];
        G[2] = t0;
    }
    if (!all_finite<T, 2>(G))
    {
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            G[e] = (T)0;
        }
    }
    mm<T, 2, 2, 2>(PHT, G, W1);
    tr<T, 2, 2>(G, Gt);
    mm<T, 2, 2, 2>(W1, Gt, W);
    T dx[2];
    mm<T, 2, 2, 1>(W, V, dx);
    T W1t[4], WW[4];
    tr<T, 2, 2>(W1, W1t);
    mm<T, 2, 2, 2>(W1, W1t, WW);
    xf_new[0] = xf[0] + dx[0];
    xf_new[1] = xf[1] + dx[1];
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        pf_new[e] = pf[e] - WW[e];
    }
}

constexpr int kPfSubLanes = 8; // lanes per particle in pf_sample_proposal_kernel (= its observation chunk)

template <typename T>
__global__ void __launch_bounds__(64) pf_sample_proposal_kernel(PfStore<T> s, const T* __restrict__ Z,
                                                                 const int* __restrict__ idf, int m, T r00, T r10, T r01,
                                                                 T r11, const T* __restrict__ normals, PfPredict<T> pred,
                                                                 int fu_mode)
{
    // fu_mode != 0: PF::featureUpdate (PF.cpp:222-277) of the m observed features rides along too (1: the reference's
    // gain, 2: the textbook one): it needs exactly the Jacobians and innovation at the SAMPLED pose that the likelihood
    // factor of the same observation is built from, so the sub-lane that has them finishes the feature's Kalman update
    // and stores it (pf_feature_update_kernel would reload the pose and the feature and recompute them)
    // pred.on: the particle's predict step (PF.cpp:419-471, pf_predict_state) is applied to the loaded pose and
    // covariance first -- cslam_pf_observation_step's predict + observe in one launch; this kernel overwrites xv and Pv
    // anyway, so the predicted values never go to memory
    // kPfSubLanes lanes per particle.  The sequential proposal updates (PF.cpp:502-530) are computed by all of them
    // alike (same instructions on the same values: lanes are free, a wave of 64 particles used 8 waves of the whole
    // chip); the m likelihood factors at the sampled pose (PF.cpp:343-359) are independent of one another, so sub-lane j
    // evaluates observation j and the product is then formed in the reference's order from the shuffled factors.
    const int gl  = blockIdx.x * 64 + threadIdx.x;
    const int p   = gl / kPfSubLanes;
    const int sub = threadIdx.x & (kPfSubLanes - 1);
    if (p >= s.np) // (whole groups of sub-lanes leave together)
    {
        return;
    }
    const T R[4] = {r00, r10, r01, r11};
    T       X[3], P[9], X0[3], P0[9], PX[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        X[i] = s.xv[(size_t)i * s.np + p];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        P[i] = s.pv[(size_t)i * s.np + p];
    }
    if (pred.on) // (kernel-uniform)
    {
        const T Q[4] = {pred.q00, pred.q10, pred.q01, pred.q11};
        pf_predict_state<T>(X, P, pred.v, pred.swa, Q, pred.wb, pred.dt);
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        X0[i] = X[i];
        PX[i] = X[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        P0[i] = P[i];
    }
    // The features of up to kObsChunk observations are requested together before the (strictly sequential) pose
    // updates consume them: one memory round trip per chunk instead of one per observation (the arithmetic and its
    // order are untouched).  The same registers serve the likelihood loop when m <= kObsChunk.
    constexpr int kObsChunk = 8;
    T             xfc[kObsChunk][2], pfc[kObsChunk][4];
    for (int base = 0; base < m; base += kObsChunk)
    {
#pragma unroll
        for (int j = 0; j < kObsChunk; j++)
        {
            // (clamped index, unconditional load: a branch per load would serialise the round trips again)
            load_feature<T>(s, p, idf[min(base + j, m - 1)] - 1, xfc[j], pfc[j]);
        }
#pragma unroll
        for (int j = 0; j < kObsChunk; j++)
        {
            const int i = base + j;
            if (i >= m)
            {
                continue;
            }
        T xf[2] = {xfc[j][0], xfc[j][1]}, pf[4] = {pfc[j][0], pfc[j][1], pfc[j][2], pfc[j][3]};
        T ZP[2], HV[6], HF[4], SF[4], SFI[4], VI[2];
        compute_jacobians<T>(PX, xf, pf, R, ZP, HV, HF, SF);
        inverse_lu<T, 2>(SF, SFI);
        VI[0] = Z[2 * i] - ZP[0];
        VI[1] = pi2pi<T>(Z[2 * i + 1] - ZP[1]);
        T HVt[6], t32[6], t33[9], Pinv[9], PT[9];
        tr<T, 2, 3>(HV, HVt);
        mm<T, 3, 2, 2>(HVt, SFI, t32);
        mm<T, 3, 2, 3>(t32, HV, t33);
        inverse_lu<T, 3>(P, Pinv);
#pragma unroll
        for (int e = 0; e < 9; e++)
        {
            PT[e] = t33[e] + Pinv[e];
        }
        inverse_lu<T, 3>(PT, P);
        T a32[6], b32[6], dx[3];
        mm<T, 3, 3, 2>(P, HVt, a32);
        mm<T, 3, 2, 2>(a32, SFI, b32);
        mm<T, 3, 2, 1>(b32, VI, dx);
#pragma unroll
        for (int e = 0; e < 3; e++)
        {
            X[e]  = X[e] + dx[e];
            PX[e] = X[e];
        }
        }
    }
    T L[9], XS[3], z[3];
#pragma unroll
    for (int e = 0; e < 3; e++)
    {
        z[e] = normals[(size_t)e * s.np + p];
    }
    chol_decomp<T, 3>(P, L);
    mm<T, 3, 3, 1>(L, z, XS);
#pragma unroll
    for (int e = 0; e < 3; e++)
    {
        XS[e] = XS[e] + X[e];
    }
    // likelihood at the sampled pose (PF.cpp:343-359)
    T like = (T)1;
    static_assert(kObsChunk == kPfSubLanes, "one observation of a chunk per sub-lane");
    for (int base = 0; base < m; base += kObsChunk)
    {
        if (m > kObsChunk) // (otherwise the chunk loaded above is still the right one)
        {
#pragma unroll
            for (int j = 0; j < kObsChunk; j++)
            {
                load_feature<T>(s, p, idf[min(base + j, m - 1)] - 1, xfc[j], pfc[j]);
            }
        }
        // this sub-lane's observation of the chunk (selects, not a dynamically indexed register array)
        T xf[2] = {xfc[0][0], xfc[0][1]}, pf[4] = {pfc[0][0], pfc[0][1], pfc[0][2], pfc[0][3]};
#pragma unroll
        for (int j = 1; j < kObsChunk; j++)
        {
            const bool mine = (j == sub);
            xf[0]           = mine ? xfc[j][0] : xf[0];
            xf[1]           = mine ? xfc[j][1] : xf[1];
            pf[0]           = mine ? pfc[j][0] : pf[0];
            pf[1]           = mine ? pfc[j][1] : pf[1];
            pf[2]           = mine ? pfc[j][2] : pf[2];
            pf[3]           = mine ? pfc[j][3] : pf[3];
        }
        const int i  = min(base + sub, m - 1); // (a sub-lane beyond m repeats the last one; its factor is not used)
        T         lf;
        {
            T ZP[2], HV[6], HF[4], SF[4], V[2];
            compute_jacobians<T>(XS, xf, pf, R, ZP, HV, HF, SF);
            V[0] = Z[2 * i] - ZP[0];
            V[1] = pi2pi<T>(Z[2 * i + 1] - ZP[1]);
            lf   = gauss_evaluate<T, 2>(V, SF);
            if (fu_mode != 0 && base + sub < m)
            {
                T xn[2], pn[4];
                pf_feature_kf<T>(xf, pf, HF, V, R, fu_mode == 2 ? 1 : 0, xn, pn);
                const int f = idf[i] - 1;
                s.xf[((size_t)f * 2 + 0) * s.np + p] = xn[0];
                s.xf[((size_t)f * 2 + 1) * s.np + p] = xn[1];
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    s.pf[((size_t)f * 4 + e) * s.np + p] = pn[e];
                }
            }
        }
        const int lane0 = (int)(threadIdx.x & ~(kPfSubLanes - 1));
#pragma unroll
        for (int j = 0; j < kObsChunk; j++)
        {
            const T lj = __shfl(lf, lane0 + j);
            if (base + j < m)
            {
                like = like * lj; // the reference's order: ((1 * l0) * l1) * ...
            }
        }
    }
    T d1[3] = {X0[0] - XS[0], X0[1] - XS[1], pi2pi<T>(X0[2] - XS[2])};
    T d2[3] = {X[0] - XS[0], X[1] - XS[1], pi2pi<T>(X[2] - XS[2])};
    T prior = gauss_evaluate<T, 3>(d1, P0);
    T prop  = gauss_evaluate<T, 3>(d2, P);
    if (sub != 0)
    {
        return;
    }
    T w     = s.w[p];
    s.w[p]  = w * like * prior / prop;
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        s.xv[(size_t)i * s.np + p] = XS[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
        s.pv[(size_t)i * s.np + p] = (T)0; // PF.cpp:537
    }
}

// ---------------------------------------------------------------- PF.cpp:222-277 with slam.h:235-266 at n = k = 2
// one lane per (particle, observation); observations of distinct features are independent given the pose
template <typename T>
__global__ void __launch_bounds__(64) pf_feature_update_kernel(PfStore<T> s, const T* __restrict__ Z,
                                                                const int* __restrict__ idf, int m, T r00, T r10, T r01,
                                                                T r11, int textbook)
{
    int p = blockIdx.x * 64 + threadIdx.x;
    int i = blockIdx.y;
    if (p >= s.np || i >= m)
    {
        return;
    }
    const T R[4] = {r00, r10, r01, r11};
    T       X[3], xf[2], pf[4], ZP[2], HV[6], HF[4], SF[4], V[2];
#pragma unroll
    for (int e = 0; e < 3; e++)
    {
        X[e] = s.xv[(size_t)e * s.np + p];
    }
    const int f = idf[i] - 1;
    load_feature<T>(s, p, f, xf, pf);
    compute_jacobians<T>(X, xf, pf, R, ZP, HV, HF, SF);
    V[0] = Z[2 * i] - ZP[0];
    V[1] = pi2pi<T>(Z[2 * i + 1] - ZP[1]);
    T xn[2], pn[4];
    pf_feature_kf<T>(xf, pf, HF, V, R, textbook, xn, pn);
    s.xf[((size_t)f * 2 + 0) * s.np + p] = xn[0];
    s.xf[((size_t)f * 2 + 1) * s.np + p] = xn[1];
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        s.pf[((size_t)f * 4 + e) * s.np + p] = pn[e];
    }
}

// ---------------------------------------------------------------- PF.cpp:9-60; one lane per (particle, new obs)
template <typename T>
__global__ void __launch_bounds__(64) pf_add_features_kernel(PfStore<T> s, const T* __restrict__ Z, int q, T r00, T r10,
                                                              T r01, T r11)
{
    int p = blockIdx.x * 64 + threadIdx.x;
    int i = blockIdx.y;
    if (p >= s.np || i >= q)
    {
        return;
    }
    const T R[4] = {r00, r10, r01, r11};
    T       x = s.xv[(size_t)0 * s.np + p], y = s.xv[(size_t)1 * s.np + p], phi = s.xv[(size_t)2 * s.np + p];
    T       r = Z[2 * i], b = Z[2 * i + 1];
    T       sn = dsin(phi + b), cs = dcos(phi + b);
    const int f = s.nf + i;
    s.xf[((size_t)f * 2 + 0) * s.np + p] = x + (r * cs);
    s.xf[((size_t)f * 2 + 1) * s.np + p] = y + (r * sn);
    T Gz[4] = {cs, sn, -r * sn, r * cs};
    T GzR[4], Gzt[4], out[4];
    mm<T, 2, 2, 2>(Gz, R, GzR);
    tr<T, 2, 2>(Gz, Gzt);
    mm<T, 2, 2, 2>(GzR, Gzt, out);
#pragma unroll
    for (int e = 0; e < 4; e++)
    {
        s.pf[((size_t)f * 4 + e) * s.np + p] = out[e];
    }
}

// ---------------------------------------------------------------- resample pieces (PF.cpp:473-500)
// sums[0] = sum w, sums[1] = sum w^2, accumulated in double; one workgroup, deterministic order
template <typename T>
__global__ void __launch_bounds__(256) pf_weight_sums_kernel(const T* __restrict__ w, int np, double* __restrict__ sums)
{
    __shared__ double s1[256], s2[256];
    double            a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < np; i += 256)
    {
        double x = (double)w[i];
        a += x;
        b += x * x;
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1)
    {
        if ((int)threadIdx.x < st)
        {
            s1[threadIdx.x] += s1[threadIdx.x + st];
            s2[threadIdx.x] += s2[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        sums[0] = s1[0];
        sums[1] = s2[0];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) pf_scale_weights_kernel(T* __restrict__ w, int np, T scale, int set_instead)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < np)
    {
        w[i] = set_instead ? scale : w[i] * scale;
    }
}

// ------------------------------------------------------------------------------------------------
// PF::resampleParticles (PF.cpp:473-500) with stratifiedResample (PF.cpp:546-574) for ONE shard holding the whole
// particle set, entirely on the device: weight sums (same reduction as pf_weight_sums_kernel), normalisation,
// Neff, the decision, the running sum of the normalised weights in the particle dtype IN INDEX ORDER (one lane: the
// same sequence of roundings as the reference's loop), and keep[c] = first i with cum[i] > select[c] (equal to the
// reference's two-pointer walk because select is increasing).  info[0] = Neff, info[1] = 1 if resampling happens.
// The kernels that move the particles afterwards take `enable` and return at once when it is 0.
// One workgroup (the running sum is sequential); any np: beyond kPfPlanMax the weights go through LDS in stages.
// ------------------------------------------------------------------------------------------------
constexpr int kPfPlanMax = 8192; // particles staged in LDS at once (more: the running sum goes on chunk by chunk)

// PF.cpp:559-563: cum[i] = w[0] + ... + w[i], summed sequentially in the particle dtype by ONE lane (the reference's
// sequence of roundings), staged through LDS kPfPlanMax weights at a time (one lane walking global memory took 65 us for
// 512 particles: a dependent L2 round trip per element).  The sums go to cum[] (global); for np <= kPfPlanMax they are
// also still in s_cum afterwards.  Called by every thread of the one workgroup.
template <typename T>
__device__ __forceinline__ void pf_running_sum(const T* __restrict__ w, int np, T* __restrict__ cum,
                                               __attribute__((address_space(3))) T* s_cum)
{
    // (s_cum is typed as an LDS pointer so that the chain below uses ds_read / ds_write whatever the inliner does)
    T run = (T)0; // (lives in thread 0)
    for (int base = 0; base < np; base += kPfPlanMax)
    {
        const int len = min(kPfPlanMax, np - base);
        if (base > 0)
        {
            __syncthreads(); // the previous chunk has been copied out
        }
        for (int i = threadIdx.x; i < len; i += 256)
        {
            s_cum[i] = w[base + i];
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            int i = 0;
            if (base == 0)
            {
                run = s_cum[0];
                i   = 1;
            }
            // eight weights are read ahead of the eight dependent additions: left to the compiler this form of the loop
            // ran one LDS round trip per element (the plan kernel took 25 us for 512 particles instead of 11.6)
            for (; i + 8 <= len; i += 8)
            {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    v[u] = s_cum[i + u];
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                {
                    run          = run + v[u];
                    s_cum[i + u] = run;
                }
            }
            for (; i < len; i++)
            {
                run      = run + s_cum[i];
                s_cum[i] = run;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < len; i += 256)
        {
            cum[base + i] = s_cum[i];
        }
    }
    __syncthreads();
}

// first i in [0, np) with cum[i] > sc, else 0 (PF.cpp:565-574: the reference leaves keep[c] = 0 then)
template <typename T>
__device__ __forceinline__ int pf_first_above(const __attribute__((address_space(3))) T* cumv, int np, T sc)
{
    int lo = 0, hi = np;
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (sc < cumv[mid])
        {
            hi = mid;
        }
        else
        {
            lo = mid + 1;
        }
    }
    return (lo < np) ? lo : 0;
}

template <typename T>
__global__ void __launch_bounds__(256) pf_resample_plan_kernel(T* __restrict__ w, int np, const T* __restrict__ select,
                                                                double n_effective, int resample_status,
                                                                T* __restrict__ cum, int* __restrict__ keep,
                                                                double* __restrict__ info, int* __restrict__ enable)
{
    __shared__ double s1[256], s2[256];
    __shared__ int    s_do;
    double            a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < np; i += 256)
    {
        double x = (double)w[i];
        a += x;
        b += x * x;
    }
    s1[threadIdx.x] = a;
    s2[threadIdx.x] = b;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1)
    {
        if ((int)threadIdx.x < st)
        {
            s1[threadIdx.x] += s1[threadIdx.x + st];
            s2[threadIdx.x] += s2[threadIdx.x + st];
        }
        __syncthreads();
    }
    const double ws = s1[0], ws2r = s2[0];
    const T      scale = (T)(1.0 / ws); // PF.cpp:482-487
    for (int i = threadIdx.x; i < np; i += 256)
    {
        w[i] = w[i] * scale;
    }
    // Neff from the raw sums, as the host path does: 1 / sum (w/ws)^2 = ws^2 / sum w^2  (PF.cpp:549-554)
    const double neff = (ws2r > 0.0) ? (ws * ws) / ws2r : 0.0;
    if (threadIdx.x == 0)
    {
        const int go = (neff < n_effective && resample_status) ? 1 : 0; // PF.cpp:490
        s_do         = go;
        info[0]      = neff;
        info[1]      = (double)go;
        info[2] += 1.0;        // calls and resamples since the handle was created (cslam_pf_resample_stats): lets a
        info[3] += (double)go; // driver leave every step on the device and still report what happened
        *enable      = go;
    }
    __syncthreads();
    if (!s_do)
    {
        return;
    }
    __shared__ T s_cum[kPfPlanMax];
    pf_running_sum<T>(w, np, cum, (__attribute__((address_space(3))) T*)s_cum);
    if (np <= kPfPlanMax) // (workgroup-uniform) the search reads LDS
    {
        for (int c = threadIdx.x; c < np; c += 256)
        {
            keep[c] = pf_first_above<T>((const __attribute__((address_space(3))) T*)s_cum, np, select[c]);
        }
    }
    else // beyond one LDS stage: the sums this workgroup has just written to global memory
    {
        __threadfence();
        __syncthreads();
        const volatile T* cv = cum;
        for (int c = threadIdx.x; c < np; c += 256)
        {
            const T sc = select[c];
            int     lo = 0, hi = np;
            while (lo < hi)
            {
                const int mid = (lo + hi) >> 1;
                if (sc < cv[mid])
                {
                    hi = mid;
                }
                else
                {
                    lo = mid + 1;
                }
            }
            keep[c] = (lo < np) ? lo : 0;
        }
    }
}

// The stratified selection alone (PF.cpp:559-574) on ALREADY NORMALISED weights: the sharded resample runs it on every
// rank over the all-gathered weights of the whole set (identical input -> identical keep[] everywhere).  Same running
// sum (one lane, particle dtype, index order) and the same search as pf_resample_plan_kernel.  One workgroup.
template <typename T>
__global__ void __launch_bounds__(256) pf_keep_kernel(const T* __restrict__ w, int np, const T* __restrict__ select,
                                                       int* __restrict__ keep, T* __restrict__ cum)
{
    __shared__ T s_cum[kPfPlanMax];
    pf_running_sum<T>(w, np, cum, (__attribute__((address_space(3))) T*)s_cum);
    if (np <= kPfPlanMax)
    {
        for (int c = threadIdx.x; c < np; c += 256)
        {
            keep[c] = pf_first_above<T>((const __attribute__((address_space(3))) T*)s_cum, np, select[c]);
        }
    }
    else
    {
        __threadfence();
        __syncthreads();
        const volatile T* cv = cum;
        for (int c = threadIdx.x; c < np; c += 256)
        {
            const T sc = select[c];
            int     lo = 0, hi = np;
            while (lo < hi)
            {
                const int mid = (lo + hi) >> 1;
                if (sc < cv[mid])
                {
                    hi = mid;
                }
                else
                {
                    lo = mid + 1;
                }
            }
            keep[c] = (lo < np) ? lo : 0;
        }
    }
}

// Who sends what where, on the device.  Global slot g lives on rank g / L at local index g % L; it is
// refilled from global particle keep[g], which lives on rank keep[g] / L.  keep[] is non-decreasing (select is
// increasing), so for the slots of one destination the source ranks come in ascending order and, within one source,
// in ascending g: no sorting is needed anywhere.
//   send_idx[j]   local index of the j-th particle this rank sends = keep[g] - rank*L over all g whose source is this
//                 rank, ascending g (hence grouped by destination rank)
//   counts[d]             records this rank sends to rank d      (d = 0..world-1, including itself)
//   counts[world + s]     records this rank receives from rank s (they fill its slots in order)
// One workgroup.
template <int DUMMY = 0>
__global__ void __launch_bounds__(256) pf_exchange_plan_kernel(const int* __restrict__ keep, int N, int L, int rank, int world,
                                                                int* __restrict__ send_idx, int* __restrict__ counts)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * world; i += 256)
    {
        counts[i] = 0;
    }
    // flags, then an exclusive scan (blocked: each thread owns a contiguous run of g)
    const int per = (N + 255) / 256;
    const int g0 = tid * per, g1 = min(N, g0 + per);
    int       cnt = 0;
    for (int g = g0; g < g1; g++)
    {
        cnt += (keep[g] / L == rank) ? 1 : 0;
    }
    __shared__ int s_part[257];
    s_part[tid + 1] = cnt;
    if (tid == 0)
    {
        s_part[0] = 0;
    }
    __syncthreads();
    if (tid == 0)
    {
        for (int i = 1; i <= 256; i++)
        {
            s_part[i] += s_part[i - 1];
        }
    }
    __syncthreads();
    int pos = s_part[tid];
    for (int g = g0; g < g1; g++)
    {
        if (keep[g] / L == rank)
        {
            send_idx[pos++] = keep[g] - rank * L;
            atomicAdd(&counts[g / L], 1);
        }
        if (g / L == rank)
        {
            atomicAdd(&counts[world + keep[g] / L], 1);
        }
    }
}

// The particle moves of a resample in ONE pass along the particle index (coalesced both ways), from the store into its
// twin (cslam_pf.hip keeps two sets of xv / Pv / xf / Pf buffers and swaps them after this kernel; the first form went
// through a scratch copy and back in two gated launches): dst slot i <- src particle keep[i] when the plan kernel decided to resample
// (*enable), else <- src particle i (an identity copy costs what the two gated launches it replaces cost; the host
// then knows which set is current without asking the device).  The weight row is not doubled: w = w_new in place when
// resampling.  grid = (13 + 6 nf, ceil(np/256)).
template <typename T>
__global__ void __launch_bounds__(256) pf_gather_move_kernel(PfStore<T> s, PfStore<T> d, const int* __restrict__ keep,
                                                             const int* __restrict__ enable, T w_new)
{
    const int en = *enable;
    const int e  = blockIdx.x;
    const int i  = blockIdx.y * 256 + threadIdx.x;
    if (i >= s.np)
    {
        return;
    }
    if (e == 0)
    {
        if (en)
        {
            s.w[i] = w_new; // PF.cpp:495-499
        }
        return;
    }
    const T* src;
    T*       dst;
    if (e < 4)
    {
        src = s.xv + (size_t)(e - 1) * s.np;
        dst = d.xv + (size_t)(e - 1) * s.np;
    }
    else if (e < 13)
    {
        src = s.pv + (size_t)(e - 4) * s.np;
        dst = d.pv + (size_t)(e - 4) * s.np;
    }
    else if (e < 13 + 2 * s.nf)
    {
        src = s.xf + (size_t)(e - 13) * s.np;
        dst = d.xf + (size_t)(e - 13) * s.np;
    }
    else
    {
        src = s.pf + (size_t)(e - 13 - 2 * s.nf) * s.np;
        dst = d.pf + (size_t)(e - 13 - 2 * s.nf) * s.np;
    }
    dst[i] = src[en ? keep[i] : i];
}


template <typename T>
__global__ void __launch_bounds__(256) pf_set_weights_if_kernel(T* __restrict__ w, int np, T value, const int* __restrict__ enable)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (*enable && i < np)
    {
        w[i] = value;
    }
}

// Packed record of one particle: [w, xv(3), pv(9), xf(2*nf), pf(4*nf)] -- rec_len = 13 + 6*nf scalars.
// pack: records[j] <- particle idx[j];  grid.x = count, threads stride over the record.
template <typename T>
__global__ void __launch_bounds__(256) pf_pack_kernel(PfStore<T> s, const int* __restrict__ idx, int count,
                                                       T* __restrict__ rec, const int* __restrict__ enable = nullptr)
{
    const int j = blockIdx.x;
    if (j >= count || (enable != nullptr && *enable == 0))
    {
        return;
    }
    const int    p   = idx[j];
    const int    len = 13 + 6 * s.nf;
    T*           out = rec + (size_t)j * len;
    for (int e = threadIdx.x; e < len; e += 256)
    {
        T v;
        if (e == 0)
        {
            v = s.w[p];
        }
        else if (e < 4)
        {
            v = s.xv[(size_t)(e - 1) * s.np + p];
        }
        else if (e < 13)
        {
            v = s.pv[(size_t)(e - 4) * s.np + p];
        }
        else if (e < 13 + 2 * s.nf)
        {
            v = s.xf[(size_t)(e - 13) * s.np + p];
        }
        else
        {
            v = s.pf[(size_t)(e - 13 - 2 * s.nf) * s.np + p];
        }
        out[e] = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) pf_unpack_kernel(PfStore<T> s, const int* __restrict__ idx, int count,
                                                         const T* __restrict__ rec, const int* __restrict__ enable = nullptr)
{
    const int j = blockIdx.x;
    if (j >= count || (enable != nullptr && *enable == 0))
    {
        return;
    }
    const int p   = (idx != nullptr) ? idx[j] : j; // nullptr: identity (record j -> slot j)
    const int len = 13 + 6 * s.nf;
    const T*  in  = rec + (size_t)j * len;
    for (int e = threadIdx.x; e < len; e += 256)
    {
        T v = in[e];
        if (e == 0)
        {
            s.w[p] = v;
        }
        else if (e < 4)
        {
            s.xv[(size_t)(e - 1) * s.np + p] = v;
        }
        else if (e < 13)
        {
            s.pv[(size_t)(e - 4) * s.np + p] = v;
        }
        else if (e < 13 + 2 * s.nf)
        {
            s.xf[(size_t)(e - 13) * s.np + p] = v;
        }
        else
        {
            s.pf[(size_t)(e - 13 - 2 * s.nf) * s.np + p] = v;
        }
    }
}

} // namespace cslam
