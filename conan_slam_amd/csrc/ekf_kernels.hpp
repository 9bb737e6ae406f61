// ekf_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the EKF-SLAM hot path.
//
// Data layout in HBM (all column-major, as the reference's Eigen objects):
//   P    n x n covariance inside an ldp x ldp buffer, ldp = round_up(3+2*Nmax, 128); padding is inert.
//   X    state, ldp scalars.
//   PHT  n x k panel, leading dimension ldw = ldp       (slam.h:243)
//   W1   n x k panel, leading dimension ldw; rows [n, round_up(n,128)) are written as ZERO so that the
//        downdate can process whole 128-row tiles without bounds checks (P_pad -= 0).
//   S, G k x k (leading dimension k), V, t length k.
//
// Kernels (reference lines they implement):
//   ekf_gather_kernel    PHT = P*H^T using the 5 non-zero columns of each row of H  (slam.h:243, EKF.cpp:394-395)
//   ekf_factor_kernel    S = H*PHT+R, symmetrise, LLT, G = inv(L) or inv(L)^T, t = G^T V (slam.h:244-255, EKF.cpp:108-121)
//   ekf_gain_kernel      W1 = PHT*G, X += W1*t                                       (slam.h:257-259)
//   ekf_downdate_f32/f64 P -= W1*W1^T on MFMA, LDS-tiled                             (slam.h:260)
//   (predict, heading, augment and the pose-stripe downdate: ekf_pose_kernels.hpp; the tuned factor / gain / P-GEMM
//    kernels: ekf_kernels_fast.hpp)
#pragma once

#include <hip/hip_runtime.h>

#include "device_math.hpp"

namespace cslam
{

constexpr int kFlagLltFailed = 1; // device-side factor flags
constexpr int kFlagZeroed    = 2;
constexpr int kFlagBadIdf    = 8; // a feature index outside 1..(n-3)/2 reached the device (cslam_ekf_update_device)

// Feature indices that arrive through device memory cannot be checked by the host: every kernel clamps them into
// 1..(n-3)/2 before it forms an address (no out-of-bounds access whatever the caller sends) and the gather kernel
// raises kFlagBadIdf.
__device__ inline int clamp_idf(int idf, int n)
{
    const int nf = (n - 3) >> 1;
    return idf < 1 ? 1 : (idf > nf ? (nf > 0 ? nf : 1) : idf);
}

typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Vec4;
template <>
struct Vec4<float>
{
    typedef float4 type;
};

// Block-lower storage (cslam_ekf.hip "lower" mode): P is symmetric and only 128x128 tiles on or below the
// tile diagonal are maintained; element (i,j) with tile(i) < tile(j) is read from its mirror (j,i).  Inside a
// DIAGONAL tile both halves are stored and updated, and the element-wise lower triangle is the authoritative one:
// every reader takes (max(i,j), min(i,j)) there.  (The f32-MFMA and f64 P-GEMMs produce the two halves bit-identical
// anyway; the bf16-limb P-GEMM adds the limb products of (i,j) and (j,i) in different orders, so its halves differ in
// the last bit -- reading one of them keeps P exactly symmetric for every consumer.)
template <typename T>
__device__ inline T p_sym(const T* __restrict__ P, int ldp, int i, int j, int lower)
{
    const int  ti = i >> 7, tj = j >> 7;
    const bool direct = !lower || (ti > tj) || (ti == tj && i >= j);
    return direct ? P[(size_t)j * ldp + i] : P[(size_t)i * ldp + j];
}

// Pose stripe.  Columns 0..2 of P (= rows 0..2 by symmetry, and the 3 x 3 pose block) live in their OWN buffer
//     Pv[c*ldp + i] = P[i, c],  c < 3, i < n,
// which is always current: predict (EKF.cpp:439-443) and the heading observation only touch it, every update applies its
// downdate to it at once (ekf_pose_downdate_kernel), and the pending W1 panels carry ZERO pose rows.  The P buffer's own
// rows/columns 0..2 are dead storage (the P-GEMM may scribble there; get_state patches them from Pv).  This is what lets a
// predict or a heading update run while the P-GEMM of the previous update is still sweeping P on another stream.
template <typename T>
__device__ inline T p_get(const T* __restrict__ P, const T* __restrict__ Pv, int ldp, int i, int j, int lower)
{
    if (j < 3)
    {
        return Pv[(size_t)j * ldp + i];
    }
    if (i < 3)
    {
        return Pv[(size_t)i * ldp + j];
    }
    return p_sym<T>(P, ldp, i, j, lower);
}

// EKF.cpp:354-404 for one observation. coef[0..4] = row 0 of H at columns {0,1,2,fx,fx+1},
// coef[5..9] = row 1; v = innovation (EKF.cpp:117-118, bearing wrapped); fx = 0-based index of the
// feature's x in the state (= fpos-1 of the reference).
template <typename T>
__device__ inline void observe_model_pose(const T* __restrict__ X, int n, int idf, T zr, T zb, T px, T py, T pphi, T* coef,
                                          T* v, int* fx)
{
    int f = 3 + 2 * clamp_idf(idf, n) - 2;
    *fx   = f;
    if (n > 3)
    {
        T dx  = X[f] - px;
        T dy  = X[f + 1] - py;
        T d2  = dx * dx + dy * dy;
        T d   = dsqrt(d2);
        T xd  = dx / d;
        T yd  = dy / d;
        T xd2 = dx / d2;
        T yd2 = dy / d2;
        coef[0] = -xd;
        coef[1] = -yd;
        coef[2] = (T)0;
        coef[3] = xd;
        coef[4] = yd;
        coef[5] = yd2;
        coef[6] = -xd2;
        coef[7] = (T)-1;
        coef[8] = -yd2;
        coef[9] = xd2;
        v[0]    = zr - d;
        v[1]    = pi2pi<T>(zb - (datan2(dy, dx) - pphi));
    }
    else
    {
        for (int i = 0; i < 10; i++)
        {
            coef[i] = (T)0;
        }
        v[0] = zr;
        v[1] = pi2pi<T>(zb);
    }
}

template <typename T>
__device__ inline void observe_model(const T* __restrict__ X, int n, int idf, T zr, T zb, T* coef, T* v, int* fx)
{
    observe_model_pose<T>(X, n, idf, zr, zb, X[0], X[1], X[2], coef, v, fx);
}

// ------------------------------------------------------------------------------------------------
// A predict() that has been accepted but not launched yet (EKF.cpp:406-455): when the next call is a batch update on
// the fast path, its three kernels apply it on the fly -- the gather and the factor kernel see the predicted pose and
// the predicted pose rows of P, the gain kernel writes them back -- and the predict launch disappears.
// valid = 0: nothing pending (all helpers reduce to the stored state).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct PredictArgs
{
    int valid;
    T   v, swa, q00, q10, q01, q11, wb, dt;
    int w; // stripe width: n-4 (REF_EXACT, quirk #2) or n-3
};

// predicted pose (EKF.cpp:445-452)
template <typename T>
__device__ inline void predicted_pose(const PredictArgs<T>& pp, const T* __restrict__ X, T* px, T* py, T* pphi)
{
    const T x0 = X[0], x1 = X[1], x2 = X[2];
    if (!pp.valid)
    {
        *px = x0;
        *py = x1;
        *pphi = x2;
        return;
    }
    const T s = dsin(pp.swa + x2), c = dcos(pp.swa + x2);
    *px   = x0 + pp.v * pp.dt * c;
    *py   = x1 + pp.v * pp.dt * s;
    *pphi = pi2pi<T>(x2 + pp.v * pp.dt * dsin(pp.swa) / pp.wb);
}

// Gv = [[1,0,g02],[0,1,g12],[0,0,1]] (EKF.cpp:419-428) from the OLD heading
template <typename T>
__device__ inline void predict_gv(const PredictArgs<T>& pp, T phi_old, T* g02, T* g12)
{
    *g02 = -pp.v * pp.dt * dsin(pp.swa + phi_old);
    *g12 = pp.v * pp.dt * dcos(pp.swa + phi_old);
}

// one column of the cross-covariance stripe: Gv * (a0, a1, a2), dense summation order (zeros of Gv included)
template <typename T>
__device__ inline void predict_stripe_col(T g02, T g12, T a0, T a1, T a2, T* o0, T* o1, T* o2)
{
    T t0 = (T)1 * a0;
    t0 += (T)0 * a1;
    t0 += g02 * a2;
    T t1 = (T)0 * a0;
    t1 += (T)1 * a1;
    t1 += g12 * a2;
    T t2 = (T)0 * a0;
    t2 += (T)0 * a1;
    t2 += (T)1 * a2;
    *o0 = t0;
    *o1 = t1;
    *o2 = t2;
}

// Pvv = Gv Pvv Gv^T + Gu Q Gu^T (EKF.cpp:430-440), column-major 3 x 3 in and out, from the OLD heading
template <typename T>
__device__ inline void predict_pvv(const PredictArgs<T>& pp, T phi_old, const T* Pv, T* out)
{
    const T v = pp.v, dt = pp.dt, swa = pp.swa, wb = pp.wb;
    T s = dsin(swa + phi_old), c = dcos(swa + phi_old);
    T Gv[9] = {(T)1, (T)0, (T)0, (T)0, (T)1, (T)0, -v * dt * s, v * dt * c, (T)1}; // column-major
    T Gu[6] = {dt * c, dt * s, dt * dsin(swa) / wb, -v * dt * s, v * dt * c, v * dt * dcos(swa) / wb};
    T Q[4]  = {pp.q00, pp.q10, pp.q01, pp.q11};
    T t1[9], t2[9];
    for (int cc = 0; cc < 3; cc++) // t1 = Gv*Pvv
    {
        for (int r = 0; r < 3; r++)
        {
            T acc = (T)0;
            for (int l = 0; l < 3; l++)
            {
                acc += Gv[r + 3 * l] * Pv[l + 3 * cc];
            }
            t1[r + 3 * cc] = acc;
        }
    }
    for (int cc = 0; cc < 3; cc++) // t2 = t1*Gv^T
    {
        for (int r = 0; r < 3; r++)
        {
            T acc = (T)0;
            for (int l = 0; l < 3; l++)
            {
                acc += t1[r + 3 * l] * Gv[cc + 3 * l];
            }
            t2[r + 3 * cc] = acc;
        }
    }
    T GuQ[6];
    for (int cc = 0; cc < 2; cc++)
    {
        for (int r = 0; r < 3; r++)
        {
            T acc = (T)0;
            for (int l = 0; l < 2; l++)
            {
                acc += Gu[r + 3 * l] * Q[l + 2 * cc];
            }
            GuQ[r + 3 * cc] = acc;
        }
    }
    for (int cc = 0; cc < 3; cc++)
    {
        for (int r = 0; r < 3; r++)
        {
            T acc = (T)0;
            for (int l = 0; l < 2; l++)
            {
                acc += GuQ[r + 3 * l] * Gu[cc + 3 * l];
            }
            out[r + 3 * cc] = t2[r + 3 * cc] + acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: PHT = P * H^T   (slam.h:243).  Row i of H^T's product only touches columns {0,1,2,fx,fx+1} of P
// (EKF.cpp:394-395), which in column-major P are contiguous columns -> fully coalesced reads.
// grid = (ceil(n/256), ceil(m/kGatherObs)), block = 256.
// ------------------------------------------------------------------------------------------------
constexpr int kGatherCorr = 16; // pending columns the gather kernel corrects for by itself (its template default)
constexpr int kGatherCorrMax = 64; // ... and its <T, 64, kGatherObsWide> form: one deferred batch panel (m <= 32)
constexpr int kGatherObsWide = 2;  // observations per workgroup of that form (each workgroup reads its rows of the panel)
constexpr int kGatherObs = 1; // measured at N = 5000, m = 32: 11.1 us (8 per block), 10.2 (4), 9.5 (2), 8.7 (1): the kernel is a latency chain, more blocks win

template <typename T, int KC = kGatherCorr, int OBS = kGatherObs>
__global__ void __launch_bounds__(256) ekf_gather_kernel(const T* __restrict__ X, const T* __restrict__ P,
                                                          const T* __restrict__ Pv, int ldp,
                                                          int n, const T* __restrict__ Z, const int* __restrict__ idf,
                                                          int m, T* __restrict__ PHT, int ldw, int lower,
                                                          T* __restrict__ sub = nullptr,
                                                          PredictArgs<T> pp = PredictArgs<T>{0, (T)0, (T)0, (T)0, (T)0,
                                                                                             (T)0, (T)0, (T)0, (T)0, 0},
                                                          T* __restrict__ pred_out = nullptr,
                                                          const T* __restrict__ Wc = nullptr, int ldwc = 0, int kc = 0,
                                                          const int* __restrict__ sgn = nullptr,
                                                          int* __restrict__ flags = nullptr,
                                                          T* __restrict__ Yout = nullptr)
{
    // Yout (kc > KC, the template bound): the pending panels are too many to correct for here: this kernel only publishes
    // Y = H*Wc (k x kc, Y[q*k + row]) -- with the coefficients of the (possibly predicted) pose it has anyway -- and the
    // MFMA panel kernel applies PHT -= Wc*Y^T behind it.
    if (flags != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < m)
    {
        const int id = idf[threadIdx.x];
        if (id < 1 || id > ((n - 3) >> 1))
        {
            atomicOr(&flags[0], kFlagBadIdf);
        }
    }
    // Wc / kc (kc <= KC): a FEW pending columns (heading observations: rank-1 columns, ekf_pose_step_kernel)
    // are corrected for right here -- PHT = Ps*H^T - Wc*(H*Wc)^T with H*Wc built per workgroup from the two landmark
    // rows of Wc (its pose rows are zero) -- so that the update keeps its fast path (compact block, fused predict)
    // instead of going through the separate correction kernels.
    // pp.valid: a predict() is pending (see PredictArgs): this kernel works on the PREDICTED pose and pose rows of P,
    // formed on the fly from the stored ones (which the gain kernel replaces afterwards); nothing is written to X or P.
    // sub (optional, m <= 32): the (3 + 2m) x 2m block of PHT that S = H*PHT reads -- rows 0,1,2 and the two rows
    // of every observed landmark -- stored compactly as sub[slot*2m + col] (slot 3+2o+a <-> row fx_o + a), so that
    // the one-workgroup factor kernel loads 17 KB of contiguous data instead of 2000 scattered cache lines.
    __shared__ T   s_coef[OBS * 10];
    __shared__ int s_fx[OBS];
    __shared__ int s_idf[32];
    int            o0 = blockIdx.y * OBS;
    int            no = min(OBS, m - o0);
    __shared__ unsigned s_cand; // observations whose landmark rows fall into this block's 256 rows
    if (sub != nullptr && threadIdx.x == 0)
    {
        s_cand = 0u;
    }
    // Everything this thread reads from P depends on the landmark ids only, not on the observation model: it is
    // requested here, so that the P round trip and the idf -> X -> atan2 chain of the model below run side by side
    // instead of one after the other (the kernel is a chain of dependent round trips, nothing else).
    const int i  = blockIdx.x * 256 + threadIdx.x;
    const int il = min(i, n - 1); // (rows past n: clamped loads, no stores)
    int       fxe[OBS];
    T         ea[OBS], eb[OBS];
    T         sa[OBS][3], sb[OBS][3]; // pp.valid, i < 3: rows 0..2 of the landmark's two columns
    T         Pvv[9];
    const T   phi_old = X[2];
#pragma unroll
    for (int oo = 0; oo < OBS; oo++)
    {
        fxe[oo] = 3 + 2 * clamp_idf(idf[o0 + min(oo, no - 1)], n) - 2; // as observe_model_pose
        ea[oo]  = p_get<T>(P, Pv, ldp, il, fxe[oo], lower);
        eb[oo]  = p_get<T>(P, Pv, ldp, il, fxe[oo] + 1, lower);
    }
    T p0 = Pv[(size_t)0 * ldp + il], p1 = Pv[(size_t)1 * ldp + il], p2 = Pv[(size_t)2 * ldp + il];
    T wc[KC];
#pragma unroll
    for (int q = 0; q < KC; q++)
    {
        wc[q] = (q < kc && Yout == nullptr) ? Wc[(size_t)q * ldwc + il] : (T)0;
    }
    __shared__ T s_y[OBS][2][KC];
    // (kc > 0, in-kernel correction) the two landmark rows of pending column q that thread (oo, q) turns into Y below:
    // requested here with everything else (the landmark id fixes the rows)
    T         ywa = (T)0, ywb = (T)0;
    const int yoo = threadIdx.x / KC, yq = threadIdx.x % KC;
    if (kc > 0 && Yout == nullptr && yoo < no && yq < kc)
    {
        const int fy = 3 + 2 * clamp_idf(idf[o0 + yoo], n) - 2;
        ywa          = Wc[(size_t)yq * ldwc + fy];
        ywb          = Wc[(size_t)yq * ldwc + fy + 1];
    }
    if (pp.valid && i < 3)
    {
        for (int cc = 0; cc < 3; cc++)
        {
            for (int r = 0; r < 3; r++)
            {
                Pvv[r + 3 * cc] = Pv[(size_t)cc * ldp + r];
            }
        }
#pragma unroll
        for (int oo = 0; oo < OBS; oo++)
        {
            for (int r = 0; r < 3; r++)
            {
                sa[oo][r] = Pv[(size_t)r * ldp + fxe[oo]];
                sb[oo][r] = Pv[(size_t)r * ldp + fxe[oo] + 1];
            }
        }
    }
    if ((int)threadIdx.x < no)
    {
        T v[2];
        int o = o0 + threadIdx.x;
        T px, py, pphi;
        predicted_pose<T>(pp, X, &px, &py, &pphi);
        observe_model_pose<T>(X, n, idf[o], Z[2 * o], Z[2 * o + 1], px, py, pphi, &s_coef[threadIdx.x * 10], v,
                              &s_fx[threadIdx.x]);
    }
    if (sub != nullptr)
    {
        __syncthreads(); // s_cand = 0 above
        if ((int)threadIdx.x < m && threadIdx.x < 32)
        {
            const int id        = clamp_idf(idf[threadIdx.x], n);
            s_idf[threadIdx.x]  = id;
            const int fxo       = 3 + 2 * id - 2; // first state row of that landmark
            const int i0        = blockIdx.x * 256;
            if (fxo + 1 >= i0 && fxo < i0 + 256)
            {
                atomicOr(&s_cand, 1u << threadIdx.x);
            }
        }
    }
    __syncthreads();
    if (Yout != nullptr && blockIdx.x == 0) // (workgroup-uniform) one workgroup per observation publishes its two rows of Y
    {
        for (int e = threadIdx.x; e < no * kc; e += 256)
        {
            const int oo = e / kc, q = e - oo * kc;
            const T*  c  = &s_coef[oo * 10];
            const T   wa = Wc[(size_t)q * ldwc + s_fx[oo]], wb = Wc[(size_t)q * ldwc + s_fx[oo] + 1];
            T         y0 = c[3] * wa;
            y0 += c[4] * wb;
            T y1 = c[8] * wa;
            y1 += c[9] * wb;
            if (sgn != nullptr && sgn[q] != 0)
            {
                y0 = -y0;
                y1 = -y1;
            }
            Yout[(size_t)q * (2 * m) + 2 * (o0 + oo)]     = y0;
            Yout[(size_t)q * (2 * m) + 2 * (o0 + oo) + 1] = y1;
        }
    }
    if (kc > 0 && Yout == nullptr) // (workgroup-uniform)
    {
        // Y = H*Wc for this workgroup's observations: only the landmark columns of H meet non-zero rows of Wc
        if ((int)threadIdx.x < no * KC)
        {
            const int oo = yoo, q = yq;
            T         y0 = (T)0, y1 = (T)0;
            if (q < kc)
            {
                const T* c  = &s_coef[oo * 10];
                const T  wa = ywa, wb = ywb;
                y0          = c[3] * wa;
                y0 += c[4] * wb;
                y1 = c[8] * wa;
                y1 += c[9] * wb;
                if (sgn != nullptr && sgn[q] != 0)
                {
                    y0 = -y0;
                    y1 = -y1;
                }
            }
            s_y[oo][0][q] = y0;
            s_y[oo][1][q] = y1;
        }
        __syncthreads();
    }
    if (i >= n)
    {
        return;
    }
    T g02 = (T)0, g12 = (T)0;
    if (pp.valid)
    {
        predict_gv<T>(pp, phi_old, &g02, &g12);
        if (i >= 3)
        {
            if (i - 3 < pp.w) // column i of the stripe (= row i of the pose columns, by symmetry)
            {
                T o0, o1, o2;
                predict_stripe_col<T>(g02, g12, p0, p1, p2, &o0, &o1, &o2);
                p0 = o0;
                p1 = o1;
                p2 = o2;
            }
        }
        else // row i of Pvv
        {
            T out[9];
            predict_pvv<T>(pp, phi_old, Pvv, out);
            p0 = out[i];
            p1 = out[i + 3];
            p2 = out[i + 6];
            if (pred_out != nullptr && i == 0 && blockIdx.y == 0)
            {
                // for the factor kernel (predicted pose) and the gain kernel (which commits the predict without racing
                // on X[2]): {g02, g12, predicted pose (3), predicted Pvv (9)}
                T px, py, pphi;
                predicted_pose<T>(pp, X, &px, &py, &pphi);
                pred_out[0] = g02;
                pred_out[1] = g12;
                pred_out[2] = px;
                pred_out[3] = py;
                pred_out[4] = pphi;
                for (int e = 0; e < 9; e++)
                {
                    pred_out[5 + e] = out[e];
                }
            }
        }
    }
    unsigned hit = 0; // observations whose landmark owns row i
    if (sub != nullptr && i >= 3)
    {
        const int lm1 = ((i - 3) >> 1) + 1; // 1-based landmark id of row i
        unsigned  cc  = s_cand;
        while (cc)
        {
            const int o = __builtin_ctz(cc);
            hit |= (s_idf[o] == lm1) ? (1u << o) : 0u;
            cc &= cc - 1;
        }
    }
#pragma unroll
    for (int oo = 0; oo < OBS; oo++)
    {
        if (oo >= no)
        {
            break;
        }
        const T* c  = &s_coef[oo * 10];
        int      fx = fxe[oo];
        T        a  = ea[oo];
        T        b  = eb[oo];
        if (pp.valid && i < 3) // pose rows of the landmark's two columns: elements of the predicted stripe
        {
            T o[3];
            if (fx - 3 < pp.w)
            {
                predict_stripe_col<T>(g02, g12, sa[oo][0], sa[oo][1], sa[oo][2], &o[0], &o[1], &o[2]);
                a = o[i];
            }
            if (fx + 1 - 3 < pp.w)
            {
                predict_stripe_col<T>(g02, g12, sb[oo][0], sb[oo][1], sb[oo][2], &o[0], &o[1], &o[2]);
                b = o[i];
            }
        }
        // same summation order as the dense product: columns 0,1,2,fx,fx+1 ascending
        T s0 = p0 * c[0];
        s0 += p1 * c[1];
        s0 += p2 * c[2];
        s0 += a * c[3];
        s0 += b * c[4];
        T s1 = p0 * c[5];
        s1 += p1 * c[6];
        s1 += p2 * c[7];
        s1 += a * c[8];
        s1 += b * c[9];
        if (kc > 0 && Yout == nullptr)
        {
            T c0 = (T)0, c1 = (T)0;
#pragma unroll
            for (int q = 0; q < KC; q++)
            {
                c0 += wc[q] * s_y[oo][0][q];
                c1 += wc[q] * s_y[oo][1][q];
            }
            s0 -= c0;
            s1 -= c1;
        }
        int col = 2 * (o0 + oo);
        PHT[(size_t)col * ldw + i]       = s0;
        PHT[(size_t)(col + 1) * ldw + i] = s1;
        if (sub != nullptr)
        {
            const int k = 2 * m;
            if (i < 3)
            {
                sub[i * k + col]     = s0;
                sub[i * k + col + 1] = s1;
            }
            unsigned hh = hit;
            while (hh)
            {
                const int o    = __builtin_ctz(hh);
                const int slot = 3 + 2 * o + ((i - 3) & 1);
                sub[slot * k + col]     = s0;
                sub[slot * k + col + 1] = s1;
                hh &= hh - 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gated nearest-neighbour data association (EKF.cpp:131-144 computeAssociation, EKF.cpp:235-326 dataAssociate).
// The reference evaluates S = H P H^T + R, its inverse and determinant for every (observation, feature) pair,
// but S, inv(S), det(S) and the predicted observation depend on the FEATURE only:
//   ekf_assoc_feature_kernel : one lane per feature j: the 5 x 5 block of P that H_j touches -> S_j (dense summation
//                              order: (H*P) first, then (H*P)*H^T, structural zeros add nothing), 2 x 2 partially
//                              pivoted LU inverse and determinant (what Eigen's dynamic inverse()/determinant() do),
//                              out[j] = {Sinv00, Sinv10, Sinv01, Sinv11, log det, zp_r, zp_b, 0}
//   ekf_assoc_scan_kernel    : one wave per observation walks the features IN ORDER, 64 at a time, and reproduces the
//                              sequential gate logic exactly: a feature sets a new best iff it is inside gate1 and its
//                              nd is below every earlier gated nd (exclusive prefix minimum); every other feature
//                              feeds `outer` with its nis (EKF.cpp:280-283: the else-branch also sees gated
//                              features that did not set a record).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ekf_assoc_feature_kernel(const T* __restrict__ X, const T* __restrict__ P,
                                                                 const T* __restrict__ Pv, int ldp,
                                                                 int n, T r00, T r10, T r01, T r11, int lower,
                                                                 T* __restrict__ out)
{
    const int nf = (n - 3) / 2;
    const int j  = blockIdx.x * 256 + threadIdx.x; // 0-based feature
    if (j >= nf)
    {
        return;
    }
    T   coef[10], v[2];
    int fx;
    // observe_model with z = 0: v = -zp (range), pi2pi(-bearing): recompute zp directly instead
    observe_model<T>(X, n, j + 1, (T)0, (T)0, coef, v, &fx);
    const T dx = X[fx] - X[0], dy = X[fx + 1] - X[1];
    const T zr = dsqrt(dx * dx + dy * dy);
    const T zb = datan2(dy, dx) - X[2];
    const int idx[5] = {0, 1, 2, fx, fx + 1};
    // HP[r][c] for the five columns c that H touches
    T HP[2][5];
#pragma unroll
    for (int c = 0; c < 5; c++)
    {
        T p[5];
#pragma unroll
        for (int l = 0; l < 5; l++)
        {
            p[l] = p_get<T>(P, Pv, ldp, idx[l], idx[c], lower);
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
            T sm = coef[5 * r + 0] * p[0];
            sm += coef[5 * r + 1] * p[1];
            sm += coef[5 * r + 2] * p[2];
            sm += coef[5 * r + 3] * p[3];
            sm += coef[5 * r + 4] * p[4];
            HP[r][c] = sm;
        }
    }
    T S[2][2]; // S[r][c]
    const T R[2][2] = {{r00, r01}, {r10, r11}};
#pragma unroll
    for (int c = 0; c < 2; c++)
    {
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
            T sm = HP[r][0] * coef[5 * c + 0];
            sm += HP[r][1] * coef[5 * c + 1];
            sm += HP[r][2] * coef[5 * c + 2];
            sm += HP[r][3] * coef[5 * c + 3];
            sm += HP[r][4] * coef[5 * c + 4];
            S[r][c] = sm + R[r][c];
        }
    }
    // partially pivoted LU of [[a b],[c d]]
    const T a = S[0][0], b = S[0][1], c2 = S[1][0], d = S[1][1];
    const bool sw = dabs(c2) > dabs(a);
    const T u00 = sw ? c2 : a, u01 = sw ? d : b;
    const T l10 = (sw ? a : c2) / u00;
    const T u11 = (sw ? b : d) - l10 * u01;
    const T det = sw ? -(u00 * u11) : (u00 * u11);
    // inverse, column by column: x = e_perm, forward (unit lower), backward (upper)
    T inv[2][2]; // inv[r][c]
#pragma unroll
    for (int c = 0; c < 2; c++)
    {
        T x0 = ((sw ? 1 : 0) == c) ? (T)1 : (T)0; // perm[0] == c
        T x1 = ((sw ? 0 : 1) == c) ? (T)1 : (T)0; // perm[1] == c
        x1 -= l10 * x0;
        x1        = x1 / u11;
        x0        = (x0 - u01 * x1) / u00;
        inv[0][c] = x0;
        inv[1][c] = x1;
    }
    T* o = out + (size_t)j * 8;
    o[0] = inv[0][0];
    o[1] = inv[1][0];
    o[2] = inv[0][1];
    o[3] = inv[1][1];
    o[4] = dlog(det);
    o[5] = zr;
    o[6] = zb;
    o[7] = (T)0;
}

template <typename T>
__global__ void __launch_bounds__(64) ekf_assoc_scan_kernel(const T* __restrict__ feat, int nf, const T* __restrict__ Z,
                                                             int m, T gate1, T gate2, int* __restrict__ idf_out,
                                                             int* __restrict__ kind)
{
    const int i    = blockIdx.x;
    const int lane = threadIdx.x;
    const T   z0 = Z[2 * i], z1 = Z[2 * i + 1];
    const T   inf = (T)INFINITY;
    T         nbest = inf, outer = inf;
    int       jbest = 0;
    for (int base = 0; base < nf; base += 64)
    {
        const int  j     = base + lane;
        const bool valid = j < nf;
        const T*   f     = feat + (size_t)(valid ? j : 0) * 8;
        const T    v0    = z0 - f[5];
        const T    v1    = pi2pi<T>(z1 - f[6]);
        const T    t0    = v0 * f[0] + v1 * f[1]; // (V^T Sinv)[0] = v0*Sinv00 + v1*Sinv10
        const T    t1    = v0 * f[2] + v1 * f[3];
        const T    nis   = t0 * v0 + t1 * v1;
        const T    nd    = nis + f[4];
        const bool gated = valid && (nis < gate1);
        const T    cand  = (gated && nd == nd) ? nd : inf;
        // inclusive prefix minimum over the wave, then shift by one lane
        T incl = cand;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1)
        {
            const T up = __shfl_up(incl, dlt);
            if (lane >= dlt)
            {
                incl = (up < incl) ? up : incl;
            }
        }
        T excl = __shfl_up(incl, 1);
        excl   = (lane == 0) ? inf : excl;
        const T    before = (nbest < excl) ? nbest : excl;
        const bool record = gated && (nd < before);
        if (valid && !record && nis < outer)
        {
            outer = nis;
        }
        const T cmin = __shfl(incl, 63);
        if (cmin < nbest)
        {
            const unsigned long long hits = __ballot(cand == cmin);
            jbest                         = base + __builtin_ctzll(hits) + 1;
            nbest                         = cmin;
        }
    }
#pragma unroll
    for (int dlt = 32; dlt >= 1; dlt >>= 1)
    {
        const T o = __shfl_xor(outer, dlt);
        outer     = (o < outer) ? o : outer;
    }
    if (lane == 0)
    {
        idf_out[i] = jbest;
        kind[i]    = (jbest != 0) ? 1 : ((outer > gate2) ? 2 : 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Deferred downdates.  The engine may hold the covariance as  P = Ps - Wp*Wp^T  with Ps the matrix stored
// in HBM ("stale") and Wp (n x kp) the W1 panels of updates whose P-GEMM has not been applied yet; one
// P-GEMM with k = kp then applies them all (slam.h:260 is linear in the panels).  Every reader of P adds the
// rank-kp correction for the few columns it touches:
//   PHT = P*H^T = Ps*H^T - Wp*(H*Wp)^T,   Y = H*Wp (k x kp) from the 5 non-zero columns of each H row.
// ekf_pending_y_kernel: Y[r, q] for the two rows of one observation; grid = (m, ceil(kp/256)).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ekf_pending_y_kernel(const T* __restrict__ X, int n, const T* __restrict__ Z,
                                                             const int* __restrict__ idf, int m,
                                                             const T* __restrict__ Wp, int ldw, int kp,
                                                             T* __restrict__ Y, const int* __restrict__ sgn = nullptr)
{
    // sgn (optional): columns with sgn[q] != 0 enter P with the opposite sign (P = Ps - Wp diag(+-1) Wp^T, see
    // ekf_pose_step_kernel): their row of Y is negated, which is all the correction PHT -= Wp Y^T needs.
    __shared__ T   s_coef[10];
    __shared__ int s_fx;
    const int      o = blockIdx.x;
    if (threadIdx.x == 0)
    {
        T v[2];
        observe_model<T>(X, n, idf[o], Z[2 * o], Z[2 * o + 1], s_coef, v, &s_fx);
    }
    __syncthreads();
    const int q = blockIdx.y * 256 + threadIdx.x;
    if (q >= kp)
    {
        return;
    }
    const T* w  = Wp + (size_t)q * ldw;
    const int fx = s_fx;
    const T  w0 = w[0], w1 = w[1], w2 = w[2], wa = w[fx], wb = w[fx + 1];
    T        y0 = s_coef[0] * w0;
    y0 += s_coef[1] * w1;
    y0 += s_coef[2] * w2;
    y0 += s_coef[3] * wa;
    y0 += s_coef[4] * wb;
    T y1 = s_coef[5] * w0;
    y1 += s_coef[6] * w1;
    y1 += s_coef[7] * w2;
    y1 += s_coef[8] * wa;
    y1 += s_coef[9] * wb;
    const int k = 2 * m;
    if (sgn != nullptr && sgn[q] != 0)
    {
        y0 = -y0;
        y1 = -y1;
    }
    Y[(size_t)q * k + 2 * o]     = y0;
    Y[(size_t)q * k + 2 * o + 1] = y1;
}

// PHT -= Wp * Y^T for the 2*kGatherObs columns of one gather block; Y slice staged in LDS in chunks.
// Same grid as ekf_gather_kernel; runs right after it on the same stream.
constexpr int kCorrChunk = 64;

template <typename T>
__global__ void __launch_bounds__(256) ekf_pending_corr_kernel(int n, int m, const T* __restrict__ Wp, int ldw, int kp,
                                                                const T* __restrict__ Y, T* __restrict__ PHT,
                                                                int ldpht)
{
    __shared__ T   s_y[2 * kGatherObs][kCorrChunk + 1];
    const int      o0  = blockIdx.y * kGatherObs;
    const int      no  = min(kGatherObs, m - o0);
    const int      nc  = 2 * no; // PHT columns of this block
    const int      k   = 2 * m;
    const int      i   = blockIdx.x * 256 + threadIdx.x;
    const bool     in  = i < n;
    T              acc[2 * kGatherObs];
#pragma unroll
    for (int c = 0; c < 2 * kGatherObs; c++)
    {
        acc[c] = (T)0;
    }
    for (int q0 = 0; q0 < kp; q0 += kCorrChunk)
    {
        const int qn = min(kCorrChunk, kp - q0);
        __syncthreads();
        for (int e = threadIdx.x; e < nc * qn; e += 256)
        {
            const int c = e % nc, q = e / nc;
            s_y[c][q]   = Y[(size_t)(q0 + q) * k + 2 * o0 + c];
        }
        __syncthreads();
        if (in)
        {
            for (int q = 0; q < qn; q++)
            {
                const T w = Wp[(size_t)(q0 + q) * ldw + i];
#pragma unroll
                for (int c = 0; c < 2 * kGatherObs; c++)
                {
                    acc[c] += w * s_y[c][q]; // rows of s_y beyond nc are never stored
                }
            }
        }
    }
    if (in)
    {
#pragma unroll
        for (int c = 0; c < 2 * kGatherObs; c++)
        {
            if (c < nc)
            {
                T* p = PHT + (size_t)(2 * o0 + c) * ldpht + i;
                *p   = *p - acc[c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2+K3: one workgroup.  S = H*PHT + RR (slam.h:244), symmetrise (247), lower Cholesky (250 / 417-423),
// G = inv(L) [REF_EXACT] or inv(L)^T [TEXTBOOK] (251 + quirk #1), non-finite -> zeros (252-255),
// t = G^T V.  S and G live in LDS when they fit (lds_ld = k+1 to spread banks), else in global scratch.
// Outputs: dS (symmetrised S), dG (k x k), dGt (= G^T, so that the gain kernel reads rows of G
// contiguously), dV, dt, flags[0] |= code (sticky), flags[1] = code.
// ------------------------------------------------------------------------------------------------
constexpr int kFactorThreads = 256;

template <typename T>
struct FactorArgs
{
    const T*   X;
    int        n;
    const T*   Z;
    const int* idf;
    int        m;
    T          R[4];
    const T*   PHT;
    int        ldw;
    T*         dS;
    T*         dG;   // G (k x k); ekf_factor_mfma_f32 leaves it alone and publishes only dGt (debug transposes it back)
    T*         dGt;
    T*         dV;
    T*         dt;
    int*       flags;
    T*         scratchS; // global k x (k+1) scratch used when LDS is too small
    T*         scratchG;
    long long* stamps; // diagnostic: s_memtime at phase boundaries (nullptr in production)
    const T*   sub;    // compact (3+2m) x 2m block of PHT written by ekf_gather_kernel, or nullptr
    T*         dM;     // optional (ekf_factor_mfma_f32): M = G*(G^T*PHT[0:3,:]^T), 3 x k (row c at dM + c*k): the gain kernel
                       // then applies the pose-stripe downdate P[:,0:3] -= W1*W1[0:3,:]^T = PHT*M itself (see ekf_panel_mfma_f32)
    PredictArgs<T> pp;   // pending predict (valid = 0: none); honoured by ekf_factor_mfma_f32 / ekf_factor_mfma_f64 only
    const T*   P3;       // P (for Pvv) and its leading dimension, used with pp.valid
    int        ldp3;
    const T*   pred_out; // pp.valid: {g02, g12, predicted pose (3), predicted Pvv (9)} written by the gather kernel
    int        lds_S; // 1: S in LDS
    int        lds_G; // 1: G in LDS
    int        textbook;
};

template <typename T>
__global__ void __launch_bounds__(kFactorThreads) ekf_factor_kernel(FactorArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int     k   = 2 * a.m;
    const int     ld  = k + 1;
    const int     tid = threadIdx.x;
    const int     nth = kFactorThreads;
    T*            sm  = reinterpret_cast<T*>(smem_raw);
    size_t        off = 0;
    T*            S   = a.lds_S ? (sm + off) : a.scratchS;
    off += a.lds_S ? (size_t)k * ld : 0;
    T* G = a.lds_G ? (sm + off) : a.scratchG;
    off += a.lds_G ? (size_t)k * ld : 0;
    T* coef = sm + off;
    off += (size_t)a.m * 10;
    T* V = sm + off;
    off += (size_t)k;
    int* fxs  = reinterpret_cast<int*>(sm + off);
    int* sflg = fxs + a.m; // [0] llt failed, [1] non-finite

    if (tid == 0)
    {
        sflg[0] = 0;
        sflg[1] = 0;
    }
    // observation models and innovations (EKF.cpp:108-121)
    for (int o = tid; o < a.m; o += nth)
    {
        observe_model<T>(a.X, a.n, a.idf[o], a.Z[2 * o], a.Z[2 * o + 1], &coef[o * 10], &V[2 * o], &fxs[o]);
        a.dV[2 * o]     = V[2 * o];
        a.dV[2 * o + 1] = V[2 * o + 1];
    }
    __syncthreads();
    // S = H*PHT + RR : 5-term sums in ascending column order, then + R on the 2x2 diagonal blocks
    for (int e = tid; e < k * k; e += nth)
    {
        int      r  = e % k;
        int      c  = e / k;
        int      ob = r >> 1, ra = r & 1;
        const T* cf = &coef[ob * 10 + ra * 5];
        int      fx = fxs[ob];
        const T* ph = a.PHT + (size_t)c * a.ldw;
        T        s  = cf[0] * ph[0];
        s += cf[1] * ph[1];
        s += cf[2] * ph[2];
        s += cf[3] * ph[fx];
        s += cf[4] * ph[fx + 1];
        T rr = ((c >> 1) == ob) ? a.R[ra + 2 * (c & 1)] : (T)0;
        S[r + (size_t)c * ld] = s + rr;
    }
    __syncthreads();
    // makeSymmetric (slam.h:776-779): each unordered pair owned by one thread
    for (int e = tid; e < k * k; e += nth)
    {
        int r = e % k;
        int c = e / k;
        if (r > c)
        {
            T v                    = (S[r + (size_t)c * ld] + S[c + (size_t)r * ld]) * (T)0.5;
            S[r + (size_t)c * ld] = v;
            S[c + (size_t)r * ld] = v;
        }
        else if (r == c)
        {
            T d                    = S[r + (size_t)c * ld];
            S[r + (size_t)c * ld] = (d + d) * (T)0.5;
        }
    }
    __syncthreads();
    for (int e = tid; e < k * k; e += nth)
    {
        a.dS[e] = S[(e % k) + (size_t)(e / k) * ld];
    }
    __syncthreads();
    // right-looking lower Cholesky in place (Eigen::LLT reads the lower triangle only); a pivot <= 0
    // is the LLT failure of slam.h:421
    for (int j = 0; j < k; j++)
    {
        if (tid == 0)
        {
            T d = S[j + (size_t)j * ld];
            if (d <= (T)0)
            {
                sflg[0] = 1;
            }
            else
            {
                S[j + (size_t)j * ld] = dsqrt(d);
            }
        }
        __syncthreads();
        if (sflg[0])
        {
            break;
        }
        T dj = S[j + (size_t)j * ld];
        for (int r = j + 1 + tid; r < k; r += nth)
        {
            S[r + (size_t)j * ld] = S[r + (size_t)j * ld] / dj;
        }
        __syncthreads();
        int cnt = k - j - 1;
        for (int e = tid; e < cnt * cnt; e += nth)
        {
            int c = j + 1 + e / cnt;
            int r = j + 1 + e % cnt;
            if (r >= c)
            {
                S[r + (size_t)c * ld] -= S[r + (size_t)j * ld] * S[c + (size_t)j * ld];
            }
        }
        __syncthreads();
    }
    const bool failed = (sflg[0] != 0);
    // G = inv(L): one column per thread, uniform loops (G is zero above its diagonal, so the dot
    // products may start at q = 0 and every lane reads the same L element -> LDS broadcast)
    if (!failed)
    {
        for (int c = tid; c < k; c += nth)
        {
            T* g = G + (size_t)c * ld;
            for (int r = 0; r < k; r++)
            {
                T s = (T)0;
                for (int q = 0; q < r; q++)
                {
                    s += S[r + (size_t)q * ld] * g[q];
                }
                T e  = (r == c) ? (T)1 : (T)0;
                g[r] = (r < c) ? (T)0 : (e - s) / S[r + (size_t)r * ld];
            }
        }
    }
    __syncthreads();
    // finite check (slam.h:252-255)
    if (!failed)
    {
        int bad = 0;
        for (int e = tid; e < k * k; e += nth)
        {
            bad |= !dfinite(G[(e % k) + (size_t)(e / k) * ld]);
        }
        if (bad)
        {
            atomicOr(&sflg[1], 1);
        }
    }
    __syncthreads();
    const bool zero = failed || (sflg[1] != 0);
    // write G in its final orientation (and its transpose), t = G^T V
    for (int e = tid; e < k * k; e += nth)
    {
        int r = e % k, c = e / k;
        T   g = (T)0;
        if (!zero)
        {
            g = a.textbook ? G[c + (size_t)r * ld] : G[r + (size_t)c * ld];
        }
        a.dG[r + (size_t)c * k]  = g;
        a.dGt[c + (size_t)r * k] = g;
    }
    __syncthreads();
    for (int c = tid; c < k; c += nth)
    {
        T s = (T)0;
        if (!zero)
        {
            for (int r = 0; r < k; r++)
            {
                T g = a.textbook ? G[c + (size_t)r * ld] : G[r + (size_t)c * ld];
                s += g * V[r];
            }
        }
        a.dt[c] = s;
    }
    if (tid == 0)
    {
        int code = (failed ? kFlagLltFailed : 0) | ((!failed && sflg[1]) ? kFlagZeroed : 0);
        a.flags[1] = code;
        if (code)
        {
            atomicOr(&a.flags[0], code);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4: W1 = PHT * G (slam.h:257) and X += W1 * t (slam.h:258-259 with W never formed: W*V = W1*(G^T V)).
// block = 256 threads = 64 rows x 4 column groups; Gt (= G^T, so row q of G is contiguous) is read
// through wave-uniform addresses (one wave = one column group).  Rows [n, n_pad) of W1 are zeroed.
// ------------------------------------------------------------------------------------------------
constexpr int kGainCols = 16; // columns per thread per pass

template <typename T>
__global__ void __launch_bounds__(256) ekf_gain_kernel(const T* __restrict__ PHT, int ldw, int n, int n_pad, int k,
                                                        const T* __restrict__ Gt, const T* __restrict__ t,
                                                        T* __restrict__ W1, T* __restrict__ X)
{
    __shared__ T s_part[4][64];
    const int    ri = threadIdx.x & 63;
    const int    cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int    i  = blockIdx.x * 64 + ri;
    const bool   in = (i < n);
    T            xs = (T)0;
    for (int cbase = 0; cbase < k; cbase += 4 * kGainCols)
    {
        const int c0 = cbase + cg * kGainCols;
        T         acc[kGainCols];
#pragma unroll
        for (int cc = 0; cc < kGainCols; cc++)
        {
            acc[cc] = (T)0;
        }
        if (c0 < k)
        {
            // q in chunks of 8: the 8 row values are requested together (one memory round trip per chunk, not
            // per q); out-of-range rows read row 0 and are zeroed by the select (no conditional loads)
            const int ii = in ? i : 0;
            for (int q0 = 0; q0 < k; q0 += 8)
            {
                T pv[8];
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const int q = (q0 + t < k) ? (q0 + t) : (k - 1);
                    pv[t]       = PHT[(size_t)q * ldw + ii];
                }
#pragma unroll
                for (int t = 0; t < 8; t++)
                {
                    const T  p  = (in && (q0 + t < k)) ? pv[t] : (T)0;
                    const int q = (q0 + t < k) ? (q0 + t) : (k - 1);
                    const T* gr = Gt + (size_t)q * k + c0; // G[q, c0..]
#pragma unroll
                    for (int cc = 0; cc < kGainCols; cc++)
                    {
                        const bool ok = (c0 + cc < k);
                        const T    g  = gr[ok ? cc : 0]; // unconditional (wave-uniform) load, value selected
                        acc[cc] += p * (ok ? g : (T)0);
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < kGainCols; cc++)
            {
                if (c0 + cc < k)
                {
                    if (i < n_pad)
                    {
                        W1[(size_t)(c0 + cc) * ldw + i] = in ? acc[cc] : (T)0;
                    }
                    xs += acc[cc] * t[c0 + cc];
                }
            }
        }
    }
    s_part[cg][ri] = xs;
    __syncthreads();
    if (cg == 0 && in)
    {
        T s = s_part[0][ri];
        s += s_part[1][ri];
        s += s_part[2][ri];
        s += s_part[3][ri];
        X[i] = X[i] + s;
    }
}

// ------------------------------------------------------------------------------------------------
// K5 (f64): same structure on v_mfma_f64_16x16x4_f64.
// A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15], D[i = (lane>>4) + 4*g][j = lane&15], g = 0..3.
// i <-> P column, j <-> P rows 2j+b (b = 0,1: one 16-byte double2 per lane).
// Workgroup tile = 128 rows x 64 cols; wave w owns rows [32w, 32w+32) and all 64 columns (4 column blocks).
// ------------------------------------------------------------------------------------------------
constexpr int kDownKC64 = 16;

// lower != 0: block-lower storage (see p_sym): 128 x 64 tiles that lie in 128 x 128 blocks above the block diagonal are
// not maintained -- their workgroups leave at once, the P-GEMM then reads and writes half of P.
// kcm: columns of W1 staged per pass (dynamic LDS = kcm * 192 doubles).  With kcm = k <= 64 the whole panel is staged
// at once: one load phase and one barrier per tile instead of four of each (the kernel is a latency chain at N = 1000).
// CB: 16-column blocks per workgroup tile (tile = 128 rows x 16*CB columns).  At N = 1000 the kernel is a latency chain
// (P sits in the L2 / Infinity Cache): narrower tiles mean more workgroups in flight per CU.
template <int CB>
__global__ void __launch_bounds__(256, 2) ekf_downdate_f64(double* __restrict__ P, int ldp,
                                                            const double* __restrict__ W1, int ldw, int k, int tiles_r,
                                                            int lower, int kcm)
{
    constexpr int TC = 16 * CB; // tile columns
    if (lower && (((int)(blockIdx.x / tiles_r) * TC) >> 7) > (int)(blockIdx.x % tiles_r))
    {
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double s_pan[];
    double* sB = s_pan;             // [kc][128] rows
    double* sA = s_pan + kcm * 128; // [kc][TC]  columns

    const int tid  = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int lj   = lane & 15;
    const int lq   = lane >> 4;
    const int tj   = blockIdx.x / tiles_r; // column tile of TC
    const int ti   = blockIdx.x % tiles_r; // row tile of 128
    const int row0 = ti * 128;
    const int col0 = tj * TC;

    f64x4 acc[2][CB];
#pragma unroll
    for (int b = 0; b < 2; b++)
    {
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
        {
            acc[b][cb] = (f64x4){0.0, 0.0, 0.0, 0.0};
        }
    }
    // the P tile is requested before the panels: its latency hides behind the panel loads and the MFMA loop
    // (it used to be loaded in the epilogue: 22-26 us per launch at N = 1000 although P sits in the L2 / Infinity Cache)
    double2* ptr[4 * CB];
    double2  v[4 * CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++)
    {
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            const int col   = col0 + cb * 16 + lq + 4 * g;
            ptr[cb * 4 + g] = reinterpret_cast<double2*>(P + (size_t)col * ldp + row0 + wave * 32 + 2 * lj);
            v[cb * 4 + g]   = *ptr[cb * 4 + g];
        }
    }
    if (kcm == 16)
    {
        // Register-staged, double-buffered panels: the loads of pass p+1 are in flight while pass p is multiplied, and
        // one barrier per pass is enough (pass p+2 overwrites buffer p & 1 only behind the barrier of pass p+1, which
        // every wave reaches after it has finished reading pass p).  The first form of this loop -- load, store to LDS,
        // barrier, multiply, barrier, per pass and per staging iteration -- cost 13 us per 64 columns (five dependent
        // round trips per pass) where the MFMAs need 1.
        constexpr int KCM = 16;
        constexpr int NRB = KCM * 64 / 256;                   // row-panel double2 per thread (4)
        constexpr int NRA = (KCM * (TC / 2) + 255) / 256;     // column-panel double2 per thread (1 or 2)
        double2       rb[NRB], ra[NRA];
        auto fetch = [&](int k0) {
            const int kc = min(KCM, k - k0);
#pragma unroll
            for (int it = 0; it < NRB; it++)
            {
                const int id = tid + it * 256;
                const int kk = id >> 6, r2 = (id & 63) * 2;
                rb[it] = (kk < kc) ? *reinterpret_cast<const double2*>(W1 + (size_t)(k0 + kk) * ldw + row0 + r2)
                                   : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int it = 0; it < NRA; it++)
            {
                const int id = tid + it * 256;
                const int kk = id / (TC / 2), r2 = (id % (TC / 2)) * 2;
                ra[it] = (kk < kc && kk < KCM) ? *reinterpret_cast<const double2*>(W1 + (size_t)(k0 + kk) * ldw + col0 + r2)
                                               : make_double2(0.0, 0.0);
            }
        };
        constexpr int kBuf = KCM * (128 + TC); // doubles per buffer (the launch provides two)
        fetch(0);
        int par = 0;
        for (int k0 = 0; k0 < k; k0 += KCM, par ^= 1)
        {
            double* bB = s_pan + par * kBuf;
            double* bA = bB + KCM * 128;
#pragma unroll
            for (int it = 0; it < NRB; it++)
            {
                const int id = tid + it * 256;
                *reinterpret_cast<double2*>(&bB[(id >> 6) * 128 + (id & 63) * 2]) = rb[it];
            }
#pragma unroll
            for (int it = 0; it < NRA; it++)
            {
                const int id = tid + it * 256;
                if (id < KCM * (TC / 2))
                {
                    *reinterpret_cast<double2*>(&bA[(id / (TC / 2)) * TC + (id % (TC / 2)) * 2]) = ra[it];
                }
            }
            if (k0 + KCM < k)
            {
                fetch(k0 + KCM);
            }
            __syncthreads();
            // (rows of the panel beyond k were stored as zeros: every pass is KCM deep)
#pragma unroll
            for (int kk = 0; kk < KCM; kk += 4)
            {
                const int     kq = kk + lq;
                const double2 b  = *reinterpret_cast<const double2*>(&bB[kq * 128 + wave * 32 + 2 * lj]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
                {
                    const double a = bA[kq * TC + cb * 16 + lj];
                    acc[0][cb]     = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b.x, acc[0][cb], 0, 0, 0);
                    acc[1][cb]     = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b.y, acc[1][cb], 0, 0, 0);
                }
            }
        }
    }
    else
    for (int k0 = 0; k0 < k; k0 += kcm)
    {
        const int kc = min(kcm, k - k0);
        if (k0 > 0)
        {
            __syncthreads();
        }
        // row panel: kc x 128 doubles = kc*64 double2 ; column panel: kc x 64 doubles = kc*32 double2
        for (int id = tid; id < kc * 64; id += 256)
        {
            const int     kk = id >> 6;
            const int     r2 = (id & 63) * 2;
            const double* w  = W1 + (size_t)(k0 + kk) * ldw;
            *reinterpret_cast<double2*>(&sB[kk * 128 + r2]) = *reinterpret_cast<const double2*>(w + row0 + r2);
        }
        for (int id = tid; id < kc * (TC / 2); id += 256)
        {
            const int     kk = id / (TC / 2);
            const int     r2 = (id % (TC / 2)) * 2;
            const double* w  = W1 + (size_t)(k0 + kk) * ldw;
            *reinterpret_cast<double2*>(&sA[kk * TC + r2]) = *reinterpret_cast<const double2*>(w + col0 + r2);
        }
        __syncthreads();
        // k advances by 4 per MFMA; when kc is not a multiple of 4 (k = 2 mod 4) the tail lanes feed zeros
        for (int kk = 0; kk < kc; kk += 4)
        {
            const int     kq  = kk + lq;
            const bool    ok  = (kq < kc);
            const double2 b   = ok ? *reinterpret_cast<const double2*>(&sB[kq * 128 + wave * 32 + 2 * lj])
                                   : make_double2(0.0, 0.0);
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
            {
                const double a = ok ? sA[kq * TC + cb * 16 + lj] : 0.0;
                acc[0][cb]     = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b.x, acc[0][cb], 0, 0, 0);
                acc[1][cb]     = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b.y, acc[1][cb], 0, 0, 0);
            }
        }
    }
    {
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
        {
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                v[cb * 4 + g].x -= acc[0][cb][g];
                v[cb * 4 + g].y -= acc[1][cb][g];
                *ptr[cb * 4 + g] = v[cb * 4 + g];
            }
        }
    }
}

} // namespace cslam
