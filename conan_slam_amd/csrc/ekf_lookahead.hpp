// ekf_lookahead.hpp -- the small-matrix side of the LOOK-AHEAD update engine (cslam_ekf.hip, "look-ahead windows").
//
// The batch update of slam.h:235-266 is a strict chain per update -- gather PHT = P H^T (n x k), factor S (k x k, one
// workgroup, ~22 us), gain W1 = PHT G (n x k) -- and the covariance downdate P -= W1 W1^T (the P-GEMM) sweeps all of P.
// But the factorisation only ever looks at the (3 + 2m) x (3 + 2m) principal block of P over the pose and the m observed
// landmarks, and at 3 + 2m entries of X:
//     S = H P H^T + R,   H non-zero in the pose columns and the two columns of every observed landmark (EKF.cpp:394-395).
// So the factor chain of the NEXT two updates can run ahead on a stream of its own, from a few small blocks of the
// current P, while the wide kernels (gather, gain, P-GEMM) of the previous updates still occupy the chip:
//
//   ekf_la_rows_kernel     rows of X, of the pose stripe Pv and of the pending W1 panels at the landmark rows of the
//                          window's two updates a and b                                       (main stream, many workgroups)
//   ekf_la_blocks_kernel   one workgroup per row of the blocks D_aa, D_ba, D_bb of the TRUE covariance P = Ps - Wp Wp^T
//                          between those rows (Ps block-lower in HBM, Wp the pending panels).  Every workgroup also
//                          applies update a's held predict (EKF.cpp:406-455) to its row and evaluates a's observation
//                          model (EKF.cpp:354-404), so the rows leave as what the factor chain needs: row of
//                          sub_a = PHT_a[rows(a), :] (ekf_gather_kernel's compact block), row of PHT_a[rows(b), :],
//                          row of D_bb                                                         (main stream, many workgroups)
//   ekf_la_carry_kernel    update b on stream F, between the two factor kernels: carries the rows of b through update a
//                          WITHOUT the wide kernels -- W1_a[rows(b)] = PHT_a[rows(b)] G_a, P[rows(b), rows(b)] -= W1 W1^T,
//                          X[rows(b)] += PHT_a[rows(b)] u_a, the pose stripe rows -= PHT_a M_a (slam.h:257-260 restricted
//                          to 67 rows) -- then b's predict and observation model, and forms sub_b.  One workgroup; every
//                          global load is issued up front (under the P-GEMM a round trip costs microseconds), the two
//                          64 x 64 x 64 products run on 4 x 4 register tiles fed by 16-byte LDS reads.
// The factor kernels themselves (ekf_factor_mfma_f32 / _f64) run unchanged on these compact inputs (a local state vector
// of 3 + 2m entries with local feature ids 1..m).  The wide kernels then apply both updates to all n rows afterwards
// with the factors already known.  Same algebra as the reference's sequence of two choleskyUpdate calls, re-associated;
// results agree to rounding (tests/test_timed_path_gpu.py runs this path against the oracle).
#pragma once

#include <hip/hip_runtime.h>

#include "ekf_kernels.hpp"
#include "ekf_kernels_fast.hpp"

namespace cslam
{

constexpr int kLaMaxObs = 32; // observations per update on this path (k <= 64)

// 0-based state row of slot s (two per observation) of an update's landmark rows
__device__ inline int la_row(const int* __restrict__ idf, int s, int n)
{
    return 3 + 2 * clamp_idf(idf[s >> 1], n) - 2 + (s & 1);
}

// ------------------------------------------------------------------------------------------------
// rows: slot s < ra = 2 m_a belongs to update a, ra <= s < ra + rb to update b.
//   XL[s] = X[row], PvL[s*3 + c] = Pv[c*ldp + row], WR[q*128 + s] = Wp[q*ldw + row] (q < kp; 128 slots per column)
// grid = ra + rb workgroups of 128 threads.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct LaRowsArgs
{
    const T* X;
    const T* Pv;
    int      ldp, n;
    const int* idf_a;
    int        ra;
    const int* idf_b;
    int        rb;
    const T*   Wp;
    int        ldw, kp, kpad;
    T *        XL, *PvL, *WR;
    int*       flags;
};

template <typename T>
__device__ __forceinline__ void ekf_la_rows_body(const LaRowsArgs<T>& a)
{
    const T* __restrict__ X = a.X;
    const T* __restrict__ Pv = a.Pv;
    const int ldp = a.ldp, n = a.n, ra = a.ra, rb = a.rb, ldw = a.ldw, kp = a.kp;
    const int* __restrict__ idf_a = a.idf_a;
    const int* __restrict__ idf_b = a.idf_b;
    const T* __restrict__ Wp = a.Wp;
    T* __restrict__ XL = a.XL;
    T* __restrict__ PvL = a.PvL;
    T* __restrict__ WR = a.WR;
    int* __restrict__ flags = a.flags;

    const int s   = blockIdx.x;
    if (threadIdx.x == 4) // (device-resident feature ids cannot be checked by the host: clamped everywhere, flagged here)
    {
        const int id = (s < ra) ? idf_a[s >> 1] : idf_b[(s - ra) >> 1];
        if (id < 1 || id > ((n - 3) >> 1))
        {
            atomicOr(&flags[0], kFlagBadIdf);
        }
    }
    const int row = (s < ra) ? la_row(idf_a, s, n) : la_row(idf_b, s - ra, n);
    for (int q = threadIdx.x; q < kp; q += 128)
    {
        WR[(size_t)q * 128 + s] = Wp[(size_t)q * ldw + row]; // slot-contiguous: the blocks kernel reads it lane = slot
    }
    if (threadIdx.x < 3)
    {
        PvL[s * 3 + threadIdx.x] = Pv[(size_t)threadIdx.x * ldp + row];
    }
    if (threadIdx.x == 3)
    {
        XL[s] = X[row];
    }
}

template <typename T>
__global__ void __launch_bounds__(128) ekf_la_rows_kernel(LaRowsArgs<T> a)
{
    ekf_la_rows_body<T>(a);
}


// ------------------------------------------------------------------------------------------------
// What an update's predict + observation model leave behind.  The head {g02, g12, pose, pvv} has the layout the
// gain kernels read as `pred` (ekf_panel_mfma_f32).
// ------------------------------------------------------------------------------------------------
template <typename T>
struct LaModel
{
    T g02, g12; // Gv entries of the update's predict (EKF.cpp:419-428; 0 when none was held)
    T pose[3];  // pose after the predict (EKF.cpp:445-452)
    T pvv[9];   // pose block after the predict, element (r, c) at r + 3c (EKF.cpp:430-440)
    T coef[kLaMaxObs * 10]; // H coefficients per observation (observe_model_pose)
    T pad[2];               // (the struct is copied with 16-byte LDS-DMA pieces: 336 scalars)
};

// applies a held predict to one stripe row (row index `row` >= 3): Gv * (a0, a1, a2), EKF.cpp:442-443 with the n-4 quirk
template <typename T>
__device__ inline void la_predict_row(const PredictArgs<T>& pp, T g02, T g12, int row, T* a)
{
    if (pp.valid && row - 3 < pp.w)
    {
        T o0, o1, o2;
        predict_stripe_col<T>(g02, g12, a[0], a[1], a[2], &o0, &o1, &o2);
        a[0] = o0;
        a[1] = o1;
        a[2] = o2;
    }
}

// pose, Gv entries and pose block after a held predict (one thread)
template <typename T>
__device__ inline void la_predict_pose(const PredictArgs<T>& pp, const T* pose_in, const T* pvv_in, T* pose, T* g, T* pvv)
{
    pose[0] = pose_in[0], pose[1] = pose_in[1], pose[2] = pose_in[2];
    g[0] = g[1] = (T)0;
    for (int e = 0; e < 9; e++)
    {
        pvv[e] = pvv_in[e];
    }
    if (pp.valid)
    {
        predicted_pose<T>(pp, pose_in, &pose[0], &pose[1], &pose[2]);
        predict_gv<T>(pp, pose_in[2], &g[0], &g[1]);
        predict_pvv<T>(pp, pose_in[2], pvv_in, pvv);
    }
}

// the five-term sums of ekf_gather_kernel for one (row, observation): columns 0, 1, 2, fx, fx+1 ascending
template <typename T>
__device__ inline void la_pht_pair(const T* c, T p0, T p1, T p2, T pa, T pb, T* v0, T* v1)
{
    T s0 = p0 * c[0];
    s0 += p1 * c[1];
    s0 += p2 * c[2];
    s0 += pa * c[3];
    s0 += pb * c[4];
    T s1 = p0 * c[5];
    s1 += p1 * c[6];
    s1 += p2 * c[7];
    s1 += pa * c[8];
    s1 += pb * c[9];
    *v0 = s0;
    *v1 = s1;
}

template <typename T>
struct LaPrepArgs
{
    const T* P;
    int      ldp, n, lower;
    const T* X;  // base state (pose X[0..2])
    const T* Pv; // base pose stripe (its 3 x 3 head is the pose block)
    const int* idf_a;
    const int* idf_b;
    int ra, rb; // 2 m_a, 2 m_b (rb = 0: a window of one update)
    PredictArgs<T> pp_a;
    const T *XL, *PvL, *WR; // ekf_la_rows_kernel
    int kp, kpad;
    // outputs
    T* sub_a;  // (3 + ra) x ra: sub[slot*ra + col]
    T* PH;     // rb x ra: PHT_a[rows_lm(b), :]
    T* Dbb;    // rb x rb
    T* PvLb;   // rb x 3: stripe rows of b after a's predict
    LaModel<T>* model_a;
    T*   xloc_a; // 3 + ra
    int* idloc;  // 1 .. kLaMaxObs
    unsigned* done; // 16 counters (stride 16 words) of finished workgroups: the chain kernel on stream F waits for their sum
};

// ------------------------------------------------------------------------------------------------
// grid = 3 + ra + 2 rb workgroups of 64 threads:
//   b < 3              pose row b of sub_a (workgroup 0 also publishes model_a, xloc_a, idloc)
//   b < 3 + ra         landmark row of update a: row of D_aa -> row 3 + s of sub_a
//   b < 3 + ra + rb    landmark row of update b against the columns of a: row of D_ba -> row of PHT_a[rows_lm(b)]
//   else               landmark row of update b against its own columns: row of D_bb
// D[s][s'] = Ps(row s, row s') - sum_q WR[s][q] WR[s'][q] (the pending panels' pose rows are zero: pose parts come from
// the stripe).  (s, s') and (s', s) read the same element of Ps and add the same products in the same order.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ekf_la_blocks_body(const LaPrepArgs<T>& a)
{
    __shared__ T wrow[256];
    __shared__ T s_xl[3 + 2 * kLaMaxObs];
    __shared__ T s_coef[kLaMaxObs * 10];
    __shared__ T s_pvv[9], s_g[2], s_e3[3];
    __shared__ T s_d[2 * kLaMaxObs];
    __shared__ T s_e[2 * kLaMaxObs * 3];
    const int tid = threadIdx.x;
    const int ra = a.ra, rb = a.rb, ma = ra >> 1;
    int       b    = blockIdx.x;
    int       type = 0, s = b; // 0 pose row, 1 row of a, 2 row of b vs a, 3 row of b vs b
    if (b >= 3)
    {
        b -= 3;
        type = (b < ra) ? 1 : ((b < ra + rb) ? 2 : 3);
        s    = (type == 1) ? b : ((type == 2) ? b - ra : b - ra - rb);
    }
    // everything the row needs from global memory is requested before the model is evaluated
    const int* idf_r = (type <= 1) ? a.idf_a : a.idf_b;
    const int* idf_c = (type == 3) ? a.idf_b : a.idf_a;
    const int  nc    = (type == 3) ? rb : ra;
    const int  wr_r  = (type <= 1) ? s : ra + s; // row of WR / XL / PvL of this workgroup's row
    const int  wr_c0 = (type == 3) ? ra : 0;
    int        row   = 0;
    T          pcell = (T)0;
    T          e3[3] = {(T)0, (T)0, (T)0};
    if (type != 0)
    {
        row = la_row(idf_r, s, a.n);
        for (int q = tid; q < a.kp; q += 64) // (kp <= 256: the host keeps longer windows on the classic path)
        {
            wrow[q] = a.WR[(size_t)q * 128 + wr_r];
        }
        if (tid < nc)
        {
            pcell = p_sym<T>(a.P, a.ldp, row, la_row(idf_c, tid, a.n), a.lower);
        }
        if (tid == 0)
        {
            e3[0] = a.PvL[wr_r * 3 + 0], e3[1] = a.PvL[wr_r * 3 + 1], e3[2] = a.PvL[wr_r * 3 + 2];
        }
    }
    T ee[3] = {(T)0, (T)0, (T)0};
    int erow = 0;
    if (type == 0 && tid < ra)
    {
        ee[0] = a.PvL[tid * 3 + 0], ee[1] = a.PvL[tid * 3 + 1], ee[2] = a.PvL[tid * 3 + 2];
        erow  = la_row(a.idf_a, tid, a.n);
    }
    if (tid < ra)
    {
        s_xl[3 + tid] = a.XL[tid];
    }
    if (tid == 0)
    {
        T pose_in[3] = {a.X[0], a.X[1], a.X[2]}, pvv_in[9], pose[3], g[2], pvv[9];
        for (int e = 0; e < 9; e++)
        {
            pvv_in[e] = a.Pv[(size_t)(e / 3) * a.ldp + (e % 3)];
        }
        la_predict_pose<T>(a.pp_a, pose_in, pvv_in, pose, g, pvv);
        s_xl[0] = pose[0], s_xl[1] = pose[1], s_xl[2] = pose[2];
        s_g[0] = g[0], s_g[1] = g[1];
        for (int e = 0; e < 9; e++)
        {
            s_pvv[e] = pvv[e];
        }
        if (type != 0)
        {
            la_predict_row<T>(a.pp_a, g[0], g[1], row, e3);
            s_e3[0] = e3[0], s_e3[1] = e3[1], s_e3[2] = e3[2];
        }
    }
    __syncthreads();
    if (tid < ma)
    {
        T   v[2];
        int fx;
        observe_model_pose<T>(s_xl, 3 + ra, tid + 1, (T)0, (T)0, s_xl[0], s_xl[1], s_xl[2], &s_coef[tid * 10], v, &fx);
    }
    if (type == 0)
    {
        if (tid < ra)
        {
            la_predict_row<T>(a.pp_a, s_g[0], s_g[1], erow, ee);
            s_e[tid * 3 + 0] = ee[0], s_e[tid * 3 + 1] = ee[1], s_e[tid * 3 + 2] = ee[2];
        }
    }
    else if (tid < nc)
    {
        const T* wc  = a.WR + wr_c0 + tid; // column q of the pending panels at this thread's slot: wc[q * 128]
        T        dot = (T)0;
        int      q   = 0;
        for (; q + 8 <= a.kp; q += 8) // (eight loads in flight; the products are still added in ascending q)
        {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                v[u] = wc[(size_t)(q + u) * 128];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                dot += wrow[q + u] * v[u];
            }
        }
        for (; q < a.kp; q++)
        {
            dot += wrow[q] * wc[(size_t)q * 128];
        }
        s_d[tid] = pcell - dot;
    }
    __syncthreads();
    if (type == 0)
    {
        const int r = s; // pose row
        if (tid < ma)
        {
            T v0, v1;
            la_pht_pair<T>(&s_coef[tid * 10], s_pvv[r], s_pvv[r + 3], s_pvv[r + 6], s_e[(2 * tid) * 3 + r],
                           s_e[(2 * tid + 1) * 3 + r], &v0, &v1);
            a.sub_a[r * ra + 2 * tid]     = v0;
            a.sub_a[r * ra + 2 * tid + 1] = v1;
        }
        if (r == 0)
        {
            LaModel<T>& mo = *a.model_a;
            if (tid == 0)
            {
                mo.g02 = s_g[0];
                mo.g12 = s_g[1];
            }
            if (tid < 3)
            {
                mo.pose[tid] = s_xl[tid];
            }
            if (tid < 9)
            {
                mo.pvv[tid] = s_pvv[tid];
            }
            for (int e = tid; e < ma * 10; e += 64)
            {
                mo.coef[e] = s_coef[e];
            }
            for (int e = tid; e < 3 + ra; e += 64)
            {
                a.xloc_a[e] = s_xl[e];
            }
            if (tid < kLaMaxObs)
            {
                a.idloc[tid] = tid + 1;
            }
        }
    }
    else if (type == 3)
    {
        if (tid < nc)
        {
            a.Dbb[(size_t)s * rb + tid] = s_d[tid];
        }
    }
    else
    {
        if (tid < ma)
        {
            T v0, v1;
            la_pht_pair<T>(&s_coef[tid * 10], s_e3[0], s_e3[1], s_e3[2], s_d[2 * tid], s_d[2 * tid + 1], &v0, &v1);
            T* out           = (type == 1) ? (a.sub_a + (size_t)(3 + s) * ra) : (a.PH + (size_t)s * ra);
            out[2 * tid]     = v0;
            out[2 * tid + 1] = v1;
        }
        if (type == 2 && tid < 3)
        {
            a.PvLb[s * 3 + tid] = s_e3[tid];
        }
    }
    // this workgroup's rows are out: release them to the chain kernel (device scope: it may run on another XCD).
    // (done == nullptr: the batched engine signals its chains from the NEXT kernel on the stream instead -- a release fence
    // writes back the unit's L2, and 8 instances x 195 workgroups of them took 24 us against 8 us for one instance)
    if (a.done == nullptr)
    {
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0)
    {
        atomicAdd(a.done + 16 * (blockIdx.x & 15u), 1u); // (16 counters, 64 bytes apart: ~200 atomics on ONE word take ~5 us)
    }
}

template <typename T>
__global__ void __launch_bounds__(64) ekf_la_blocks_kernel(LaPrepArgs<T> a)
{
    ekf_la_blocks_body<T>(a);
}

template <typename T>
struct LaCarryArgs
{
    int n, m_a, m_b;
    const int* idf_b;
    PredictArgs<T> pp_b;
    const T *PH, *Dbb, *PvLb, *XLb; // blocks / rows kernels (XLb: landmark coordinates of b in the base state)
    const LaModel<T>* model_a;
    const T *Gt_a, *u_a, *M_a, *sub_a; // factor outputs of update a; sub_a rows 0..2 are PHT_a's pose rows
    // outputs
    T*          sub_b;  // (3 + kb) x kb
    T*          xloc_b; // 3 + kb
    LaModel<T>* model_b;
    T*          Y_b;    // H_b * W1_a (kb x ka): Y_b[q*kb + row], the layout of ekf_gather_kernel's Yout
};

template <typename T>
inline size_t la_carry_lds()
{
    constexpr int KM = 2 * kLaMaxObs, LD = KM + 4, LB = 3 + KM + 1;
    return ((size_t)KM * LD + (size_t)(3 + KM) * LB + kLaMaxObs * 10 + (3 + KM) + 4 * KM + 3 * KM) * sizeof(T) + 64;
}

// One workgroup of 256 threads between factor(a) and factor(b) (inside ekf_la_chain_kernel).  smem: la_carry_lds<T>() bytes.
template <typename T>
__device__ __forceinline__ void ekf_la_carry_body(const LaCarryArgs<T>& a, unsigned char* smem)
{
    constexpr int KM = 2 * kLaMaxObs; // 64
    constexpr int LD = KM + 4;        // 16-byte aligned rows, conflict-free 16-byte reads
    constexpr int LB = 3 + KM + 1;
    // Two areas, each used twice (the carry step must fit beside a workgroup of the wide kernel on one compute unit,
    // see ekf_la_chain_kernel):
    T* pht  = reinterpret_cast<T*>(smem);    // pht[q*LD + s] = PHT_a[row s of b][q]; once the first product has read it:
    T* blk  = pht;                           // the (3 + kb) square block of P in rows(b) order, row-major LB
    T* gm   = pht + (3 + KM) * LB;           // gm[q*LD + c]  = G_a(q, c); after the first product the SAME space holds
    T* w1t  = gm;                            // w1t[c*LD + s] = W1_a[row s of b][c]
    T* coef = gm + KM * LD;
    T* xl   = coef + kLaMaxObs * 10;
    T* ua   = xl + (3 + KM);                 // u_a (KM), M_a (3 x KM at ua + KM)
    T* php  = ua + 4 * KM;                   // PHT_a pose rows (3 x ka)
    __shared__ T s_pose[3], s_pvv[9], s_g[2];

    const int tid = threadIdx.x;
    const int ka = 2 * a.m_a, kb = 2 * a.m_b, mb = a.m_b;
    const int ts = tid & 15, tc = tid >> 4;
    const int s0 = 4 * ts, c0 = 4 * tc;

    // ---- every global load up front
    T rph[16], rg[16], rd[16];
#pragma unroll
    for (int it = 0; it < 16; it++)
    {
        const int e = tid + it * 256; // 64 x 64 padded index space: (row, col) = (e / 64, e % 64)
        const int r = e >> 6, c = e & 63;
        rph[it]     = (r < kb && c < ka) ? a.PH[(size_t)r * ka + c] : (T)0;
        rg[it]      = (r < ka && c < ka) ? a.Gt_a[(size_t)r * ka + c] : (T)0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int r = s0 + i, c = c0 + j;
            rd[i * 4 + j] = (r < kb && c < kb) ? a.Dbb[(size_t)r * kb + c] : (T)0;
        }
    }
    T   ru = (T)0, rm[3] = {(T)0, (T)0, (T)0}, rp[3] = {(T)0, (T)0, (T)0}, re[3] = {(T)0, (T)0, (T)0}, rx = (T)0;
    int rrow = 0;
    if (tid < ka)
    {
        ru    = a.u_a[tid];
        rm[0] = a.M_a[tid], rm[1] = a.M_a[ka + tid], rm[2] = a.M_a[2 * ka + tid];
        rp[0] = a.sub_a[tid], rp[1] = a.sub_a[ka + tid], rp[2] = a.sub_a[2 * ka + tid];
    }
    if (tid < kb)
    {
        re[0] = a.PvLb[tid * 3 + 0], re[1] = a.PvLb[tid * 3 + 1], re[2] = a.PvLb[tid * 3 + 2];
        rx    = a.XLb[tid];
        rrow  = la_row(a.idf_b, tid, a.n);
    }
    const LaModel<T>& ma_ = *a.model_a;
    T                 mpose = (T)0, mpvv = (T)0;
    if (tid < 3)
    {
        mpose = ma_.pose[tid];
    }
    if (tid < 9)
    {
        mpvv = ma_.pvv[tid];
    }
    // ---- park them in LDS
#pragma unroll
    for (int it = 0; it < 16; it++)
    {
        const int e = tid + it * 256;
        const int r = e >> 6, c = e & 63;
        pht[c * LD + r] = rph[it]; // transposed: q = c
        gm[r * LD + c]  = rg[it];
    }
    if (tid < KM)
    {
        ua[tid]          = ru;
        ua[KM + tid]     = rm[0];
        ua[2 * KM + tid] = rm[1];
        ua[3 * KM + tid] = rm[2];
        php[tid]          = rp[0];
        php[KM + tid]     = rp[1];
        php[2 * KM + tid] = rp[2];
    }
    __syncthreads();
    // ---- the sums that read PHT_a's rows (its LDS area is recycled below): stripe rows of b -= PHT_a[row] M_a^T,
    //      landmark coordinates += PHT_a[row] u_a; the same for the three pose rows
    T sd0 = (T)0, sd1 = (T)0, sd2 = (T)0, sxs = (T)0;
    if (tid < kb)
    {
        for (int q = 0; q < ka; q++)
        {
            const T p = pht[q * LD + tid];
            sd0 += p * ua[KM + q];
            sd1 += p * ua[2 * KM + q];
            sd2 += p * ua[3 * KM + q];
            sxs += p * ua[q];
        }
    }
    // ---- W1_a[rows_lm(b)] = PHT_a[rows_lm(b)] * G_a on 4 x 4 register tiles
    T acc[16];
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        acc[e] = (T)0;
    }
    for (int q = 0; q < ka; q++)
    {
        T pa[4], pg[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            pa[i] = pht[q * LD + s0 + i];
            pg[i] = gm[q * LD + c0 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                acc[i * 4 + j] += pa[i] * pg[j];
            }
        }
    }
    __syncthreads(); // (everybody is done reading G_a: its space takes the product)
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            w1t[(c0 + j) * LD + s0 + i] = acc[i * 4 + j];
        }
    }
    __syncthreads();
    // ---- P[rows_lm(b), rows_lm(b)] = D_bb - W1 W1^T ((s, t) and (t, s): the same products in the same order)
#pragma unroll
    for (int e = 0; e < 16; e++)
    {
        acc[e] = (T)0;
    }
    for (int q = 0; q < ka; q++)
    {
        T pa[4], pb[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            pa[i] = w1t[q * LD + s0 + i];
            pb[i] = w1t[q * LD + c0 + i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                acc[i * 4 + j] += pa[i] * pb[j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            blk[(3 + s0 + i) * LB + 3 + c0 + j] = rd[i * 4 + j] - acc[i * 4 + j];
        }
    }
    // ---- stripe rows of b and landmark coordinates after update a (sums formed above)
    if (tid < kb)
    {
        blk[(3 + tid) * LB + 0] = re[0] - sd0;
        blk[(3 + tid) * LB + 1] = re[1] - sd1;
        blk[(3 + tid) * LB + 2] = re[2] - sd2;
        xl[3 + tid]             = rx + sxs;
    }
    // the pose and the 3 x 3 pose block (the gain kernel's rule: the thread of the larger index applies its increment
    // to both (r, c) and (c, r))
    if (tid < 9)
    {
        s_pvv[tid] = mpvv;
    }
    __syncthreads();
    if (tid < 3)
    {
        const int r = tid;
        T         d[3] = {(T)0, (T)0, (T)0}, xs = (T)0;
        for (int q = 0; q < ka; q++)
        {
            const T p = php[r * KM + q];
            d[0] += p * ua[KM + q];
            d[1] += p * ua[2 * KM + q];
            d[2] += p * ua[3 * KM + q];
            xs += p * ua[q];
        }
        s_pose[r] = mpose + xs;
        T nv[3], nt[3]; // new (r, c) and new (c, r) for c < r
        for (int c = 0; c < 3; c++)
        {
            nv[c] = s_pvv[r + 3 * c] - d[c];
            nt[c] = s_pvv[c + 3 * r] - d[c];
        }
        // (each (r, c) pair is written by exactly one thread: r for c <= r)
        for (int c = 0; c <= r; c++)
        {
            blk[r * LB + c] = nv[c];
            if (c < r)
            {
                blk[c * LB + r] = nt[c];
            }
        }
    }
    __syncthreads();
    // ---- update b's own predict (EKF.cpp:406-455), held back by the engine
    if (tid == 0)
    {
        T pvv_in[9], pose[3], g[2], pvv[9];
        for (int e = 0; e < 9; e++)
        {
            pvv_in[e] = blk[(e % 3) * LB + e / 3];
        }
        la_predict_pose<T>(a.pp_b, s_pose, pvv_in, pose, g, pvv);
        xl[0] = pose[0], xl[1] = pose[1], xl[2] = pose[2];
        s_g[0] = g[0], s_g[1] = g[1];
        for (int e = 0; e < 9; e++)
        {
            s_pvv[e]                  = pvv[e];
            blk[(e % 3) * LB + e / 3] = pvv[e];
        }
    }
    __syncthreads();
    if (tid < kb)
    {
        T e3[3] = {blk[(3 + tid) * LB + 0], blk[(3 + tid) * LB + 1], blk[(3 + tid) * LB + 2]};
        la_predict_row<T>(a.pp_b, s_g[0], s_g[1], rrow, e3);
        for (int c = 0; c < 3; c++)
        {
            blk[(3 + tid) * LB + c] = e3[c]; // P[row, c]
            blk[c * LB + 3 + tid]   = e3[c]; // P[c, row] (the stripe is stored once)
        }
    }
    if (tid < mb)
    {
        T   v[2];
        int fx;
        observe_model_pose<T>(xl, 3 + kb, tid + 1, (T)0, (T)0, xl[0], xl[1], xl[2], &coef[tid * 10], v, &fx);
    }
    __syncthreads();
    // ---- sub_b = PHT_b[rows(b), :], the five terms of ekf_gather_kernel in its order; Y_b = H_b * W1_a
    for (int e = tid; e < (3 + kb) * mb; e += 256)
    {
        const int slot = e / mb, o = e % mb;
        const T*  br   = &blk[slot * LB];
        T         v0, v1;
        la_pht_pair<T>(&coef[o * 10], br[0], br[1], br[2], br[3 + 2 * o], br[3 + 2 * o + 1], &v0, &v1);
        a.sub_b[slot * kb + 2 * o]     = v0;
        a.sub_b[slot * kb + 2 * o + 1] = v1;
    }
    for (int e = tid; e < ka * mb; e += 256)
    {
        const int q = e / mb, o = e % mb;
        const T*  c = &coef[o * 10];
        const T   wa = w1t[q * LD + 2 * o], wb = w1t[q * LD + 2 * o + 1];
        T         y0 = c[3] * wa;
        y0 += c[4] * wb;
        T y1 = c[8] * wa;
        y1 += c[9] * wb;
        a.Y_b[(size_t)q * kb + 2 * o]     = y0;
        a.Y_b[(size_t)q * kb + 2 * o + 1] = y1;
    }
    for (int e = tid; e < 3 + kb; e += 256)
    {
        a.xloc_b[e] = xl[e];
    }
    LaModel<T>& mo = *a.model_b;
    if (tid == 0)
    {
        mo.g02 = s_g[0];
        mo.g12 = s_g[1];
    }
    if (tid < 3)
    {
        mo.pose[tid] = xl[tid];
    }
    if (tid < 9)
    {
        mo.pvv[tid] = s_pvv[tid];
    }
    for (int e = tid; e < mb * 10; e += 256)
    {
        mo.coef[e] = coef[e];
    }
}

// ------------------------------------------------------------------------------------------------
// The factor chain of one look-ahead window in ONE launch on stream F: factor(a), carry, factor(b).
//   * It is launched BEFORE the window's rows / blocks kernels are even submitted to the main stream and waits for them
//     on a counter in memory (the blocks kernel's workgroups count themselves out).  Being early is the point: the
//     workgroup asks for so much LDS that it owns its compute unit, so the persistent P-GEMM that follows on the main
//     stream (grid reduced by two workgroups) cannot share the unit -- next to its matrix-bound waves the chain runs four
//     times slower (factor kernel: 83 us instead of 22), wave priority changes nothing, queue CU masks cost the P-GEMM
//     14 % (the dispatcher deals workgroups to the shader engines evenly: an engine that lost a unit is short of two
//     slots), and a unit vacated by P-GEMM workgroups is refilled by the dispatcher.
//   * One launch, no events on the way in: the kernel boundaries and hand-overs of three launches cost ~20 us of stream F.
// Same arithmetic as the three kernels launched one after another.  A wait that outlasts `timeout` (a host error path
// that never submitted the blocks kernel) raises kFlagLaTimeout and goes on, so the grid always drains.
// ------------------------------------------------------------------------------------------------
constexpr int kFlagLaTimeout = 16;

template <typename T>
struct LaChainArgs
{
    FactorArgs<T>   fa, fb;
    T *             du_a, *du_b;
    LaCarryArgs<T>  ca;
    int             nu;     // updates in the window (1 or 2)
    const unsigned* done;   // counter of the blocks kernels' workgroups
    unsigned        target; // its value once this window's blocks kernel has finished
    unsigned long long timeout; // in s_memrealtime ticks (100 MHz)
    unsigned*       chain_done; // set to `seq` when the chain has finished (the wide kernel waits for it: no stream event)
    unsigned        seq;
};

template <typename T, int K>
__device__ __forceinline__ void ekf_la_chain_body(const LaChainArgs<T>& a)
{
    extern __shared__ __align__(16) unsigned char la_chain_smem[];
    if (threadIdx.x < 64) // (wave 0: lanes 0..15 read one counter each)
    {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (true)
        {
            unsigned c = threadIdx.x < 16
                             ? __hip_atomic_load(a.done + 16 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0u;
            c += __shfl_xor(c, 1);
            c += __shfl_xor(c, 2);
            c += __shfl_xor(c, 4);
            c += __shfl_xor(c, 8);
            c = __shfl(c, 0);
            if ((int)(c - a.target) >= 0)
            {
                break;
            }
            __builtin_amdgcn_s_sleep(32);
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout)
            {
                if (threadIdx.x == 0)
                {
                    atomicOr(&a.fa.flags[0], kFlagLaTimeout);
                }
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // acquire: the blocks kernel's rows
    if constexpr (std::is_same<T, float>::value)
    {
        ekf_factor_mfma_f32_body<K>(a.fa, a.du_a);
    }
    else
    {
        ekf_factor_mfma_f64_body<K>(a.fa, a.du_a);
    }
    if (a.nu == 2)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // factor(a)'s outputs -> the carry step (same workgroup)
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        ekf_la_carry_body<T>(a.ca, la_chain_smem);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if constexpr (std::is_same<T, float>::value)
        {
            ekf_factor_mfma_f32_body<K>(a.fb, a.du_b);
        }
        else
        {
            ekf_factor_mfma_f64_body<K>(a.fb, a.du_b);
        }
    }
    // release the window's factors to the wide kernel on the main stream (device scope: other XCDs read them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x < 32) // (32 copies, 64 bytes apart: the wide kernel's workgroups poll different words)
    {
        __hip_atomic_store(a.chain_done + 16 * threadIdx.x, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, int K>
__global__ void __launch_bounds__(256) ekf_la_chain_kernel(LaChainArgs<T> a)
{
    ekf_la_chain_body<T, K>(a);
}

// ------------------------------------------------------------------------------------------------
// The WIDE half of a look-ahead window in ONE launch (f32): with the factors of both updates known, every block of 32
// rows applies update a and update b to itself -- what ekf_gather_kernel + ekf_panel_mfma_f32 (+ the in-kernel
// correction of the second gather) did in four launches:
//   a:  PHT_a[rows] = P[rows, 0:3] HU_a^T + Ps[rows, lm(a)] LU_a^T      (slam.h:243, five non-zero columns of H per row;
//                                                                         the predict of EKF.cpp:439-443 applied to the stripe)
//       W1_a[rows]  = PHT_a[rows] G_a;  X[rows] += PHT_a[rows] u_a;  Pv[rows] -= PHT_a[rows] M_a^T      (slam.h:257-260)
//   b:  PHT_b[rows] = P'[rows, 0:3] HU_b^T + Ps[rows, lm(b)] LU_b^T - W1_a[rows] Y_b^T     (Y_b = H_b W1_a from the carry step)
//       W1_b[rows]  = PHT_b[rows] G_b;  X[rows] += ...;  Pv[rows] -= ...
// Everything is row-local: the only cross-row quantities (H's coefficients at the updated state, Y_b, the pose rows of
// PHT) come from the factor chain.  Two waves per workgroup, 32 rows; the three n x 64 x 64 products run on
// v_mfma_f32_32x32x2_f32 with the SAME operand layout throughout: a lane (row j = lane & 31, half h = lane >> 5) holds
// entry q = 32 t' + (r & 3) + 8 (r >> 2) + 4 h of its row in register r of tile t' -- the accumulator layout -- and the
// k-steps of the next product walk q in that order, so a product's result feeds the next one without a shuffle.
// The pose rows (0..2) of PHT are the factor chain's (sub rows 0..2); their W1 entries are stored as zero and the pose
// block follows the gain kernel's rule (the thread of the larger index applies its increment to both (r, c) and (c, r)).
// grid = n_pad / 32 workgroups of 128 threads.
// ------------------------------------------------------------------------------------------------
struct LaWideArgs
{
    const float* P;
    int          ldp, n, lower;
    float*       X;
    float*       Pv;
    int          nu; // updates in the window
    const int *  idf_a, *idf_b;
    int          ma, mb;
    int          valid_a, valid_b, w_a, w_b; // held predicts: valid, stripe width (n - 3, or n - 4 under REF_EXACT)
    const LaModel<float>*model_a, *model_b;
    const float *Gt_a, *u_a, *M_a, *sub_a;
    const float *Gt_b, *u_b, *M_b, *sub_b, *Y_b;
    float *      W1a, *W1b; // the two slots of the pending store
    int          ldw;
    float*       wv_out; // pose rows of the LAST update's W1 (3 x k), kept for the debug entry point
    const unsigned* chain_done; // the chain kernel's completion word and the value this window waits for
    unsigned        seq;
    unsigned long long timeout;
    int*            flags;
    long long*      stamps; // diagnostics (CSLAM_LA_STAMPS): s_memrealtime at phase boundaries of one workgroup, or nullptr
    long long*      wg_times; // diagnostics (CSLAM_BATCH_STAMPS): {start, end} of every workgroup (s_memrealtime), or nullptr
};

__device__ __forceinline__ int la_q_of(int t, int r, int lh)
{
    return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
}

// PAIRS: pairs of waves per workgroup (1: the single filter's kernel; 2: the batched engine's -- two blocks of 32 rows share
// ONE copy of the staged factor outputs, so that twice the waves fit a compute unit: the batch has ~1000 blocks of rows
// where a single filter has ~300).
// KFIX: 0, or the number of columns of BOTH updates' panels known at compile time (64: m = 32 observations per update, the
// benchmark shape).  The operand reads of the three matrix products then have immediate LDS offsets and no range selects:
// the gain and correction phases are ~600 vector instructions of address arithmetic and selects each otherwise.
template <int PAIRS = 1, int KFIX = 0>
__device__ __forceinline__ void ekf_la_wide_body(const LaWideArgs& a)
{
    // TWO waves per block of 32 rows.  Wave w owns column tile w (32 columns) of every product -- half the matrix-core
    // chain, half the operand reads -- and builds tile w of PHT; the halves meet through LDS (one 8 KB exchange area,
    // carved out of the G areas while they are not in use).  The wave's work is one dependent chain, so every global
    // read is issued as early as its address is known:
    //   round 0: the feature ids, this row's stripe entries, then the landmark columns of Ps for BOTH updates (their
    //            addresses need the feature ids) -- the longest round trip of the kernel (scattered, cold);
    //   round 1: once the chain has been seen finished: G_a^T, Y_b and the small shared inputs straight into LDS by LDS-DMA
    //            (no registers in between: 16 KB each, one 1 KB piece per instruction); G_b^T follows once the exchange
    //            area has moved out of its space.
    // After that the kernel computes from registers and LDS only.  (Measured on the batched engine: issuing the poll with
    // round 0, or the DMA before the columns, does not help -- the column round trip, 12-18 us under load, is the floor.)
    __shared__ __attribute__((aligned(16))) float s_G[2][4 * 1024]; // G^T of a / b, as in memory: [q * k + c]
    __shared__ __attribute__((aligned(16))) float s_Y[4 * 1024];    // Y_b: [c * kb + q]
    __shared__ __attribute__((aligned(16))) float s_model[2][512];  // LaModel image: {g02, g12, pose, pvv} then coef at 14
    __shared__ __attribute__((aligned(16))) float s_u[2][256];
    __shared__ __attribute__((aligned(16))) float s_M[2][256];
    __shared__ int   s_fx[2][kLaMaxObs];
    __shared__ float s_sum_all[PAIRS][4][32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lj = lane & 31, lh = lane >> 5;
    const int wv = wave & 1, pr = wave >> 1; // column tile of this wave; its pair (block of rows) within the workgroup
    float (*s_sum)[32] = s_sum_all[pr];
    const int row0 = (blockIdx.x * PAIRS + pr) * 32;
    const int row  = row0 + lj;
    const int rowc = row < a.n ? row : a.n - 1;
    const int ka = KFIX ? KFIX : 2 * a.ma, kb = KFIX ? KFIX : 2 * a.mb; // (KFIX: the host checked ma, and mb when nu == 2)
    int       stamp_i = 0;
    auto      stamp   = [&]() {
        if (a.stamps != nullptr && blockIdx.x == (gridDim.x >> 1) && tid == 0)
        {
            a.stamps[stamp_i++] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    };
    stamp();
    if constexpr (PAIRS == 2) // (diagnostics of the batched engine only)
    {
        if (a.wg_times != nullptr && tid == 0)
        {
            a.wg_times[2 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    }
    // ---- round 0: what does not come from the factor chain is requested before the chain is waited for: the feature ids,
    //      this row's stripe entries and state entry ...
    const int id_a = lane < a.ma ? a.idf_a[lane] : 1;
    const int id_b = (a.nu == 2 && lane < a.mb) ? a.idf_b[lane] : 1;
    float pv0 = a.Pv[(size_t)0 * a.ldp + rowc], pv1 = a.Pv[(size_t)1 * a.ldp + rowc], pv2 = a.Pv[(size_t)2 * a.ldp + rowc];
    float x   = a.X[rowc];
    if (tid < kLaMaxObs)
    {
        s_fx[0][tid] = 3 + 2 * clamp_idf(id_a, a.n) - 2;
        s_fx[1][tid] = 3 + 2 * clamp_idf(id_b, a.n) - 2;
    }
    __syncthreads();
    // ---- ... and the landmark columns of Ps of this wave's tile (8 observations per lane), both updates
    float pcol[2][16];
    {
        int fxv[2][8];
#pragma unroll
        for (int ub = 0; ub < 2; ub++) // (all the LDS reads first, then every global load back to back)
        {
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const int o = (32 * wv + (2 * i & 3) + 8 * (2 * i >> 2) + 4 * lh) >> 1;
                fxv[ub][i]  = s_fx[ub][o < kLaMaxObs ? o : 0];
            }
        }
#pragma unroll
        for (int ub = 0; ub < 2; ub++)
        {
            const int k = ub == 0 ? ka : kb;
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                const int  q  = 32 * wv + (2 * i & 3) + 8 * (2 * i >> 2) + 4 * lh;
                const bool in = q < k && (ub == 0 || a.nu == 2);
                const int  fx = in ? fxv[ub][i] : 3;
                // (unconditional loads, the VALUE is selected: a branch per load would serialise them)
                const float va = p_sym<float>(a.P, a.ldp, rowc, fx, a.lower);
                const float vb = p_sym<float>(a.P, a.ldp, rowc, fx + 1, a.lower);
                pcol[ub][2 * i]     = in ? va : 0.f;
                pcol[ub][2 * i + 1] = in ? vb : 0.f;
            }
        }
    }
    stamp();
    // ---- the factor chain of this window (stream F) finished long ago in the steady state: one poll, no stream event
    if (tid == 0)
    {
        const unsigned long long t0   = __builtin_amdgcn_s_memrealtime();
        const unsigned*          word = a.chain_done + 16 * (blockIdx.x & 31u);
        while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.seq) < 0)
        {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout)
            {
                atomicOr(&a.flags[0], kFlagLaTimeout);
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // (acquire only: no write-back of this unit's L2)
    // ---- round 1: the chain's outputs straight into LDS by LDS-DMA
    typedef __attribute__((address_space(3))) void* lptr_t;
    // count floats, whole 1 KB pieces (zeros beyond); the two waves take alternate pieces, every workgroup starts at a
    // different piece (they all stage the same 16 KB: this way they do not walk the L2 channels in step)
    auto dma = [&](const float* src, int count, float* dst) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (unsigned)(count * 4), 0x00020000);
        const int pieces = (count + 255) / 256;
        int       it     = (int)(blockIdx.x & 15u);
        it               = it < pieces ? it : 0;
        for (int j = 0; j < pieces; j++, it = (it + 1 < pieces) ? it + 1 : 0)
        {
            if ((j % (2 * PAIRS)) == wave)
            {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + it * 256), 16, (unsigned)(lane * 16),
                                                         (unsigned)(it * 1024), 0, 0);
            }
        }
    };
    dma(a.Gt_a, ka * ka, s_G[0]);
    dma(reinterpret_cast<const float*>(a.model_a), 336, s_model[0]);
    dma(a.u_a, ka, s_u[0]);
    dma(a.M_a, 3 * ka, s_M[0]);
    if (a.nu == 2)
    {
        dma(a.Y_b, ka * kb, s_Y);
        dma(reinterpret_cast<const float*>(a.model_b), 336, s_model[1]);
        dma(a.u_b, kb, s_u[1]);
        dma(a.M_b, 3 * kb, s_M[1]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the DMA pieces (and the column loads above) have landed
    __syncthreads();
    stamp();
    // exchange area (16 registers x 64 lanes per tile): in G_b's space until G_b is staged, then in G_a's
    float* xch = s_G[1] + pr * 2048;
    float  own[16];  // this wave's tile of the row vector in hand (PHT or W1), accumulator layout
    float  full[32]; // both tiles
    auto share = [&]() { // own -> LDS, then both tiles back
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            xch[(wv * 16 + r) * 64 + lane] = own[r];
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; t++)
        {
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                full[t * 16 + r] = xch[(t * 16 + r) * 64 + lane];
            }
        }
    };
    // this wave's tile of PHT[row, :] of update ub (0 = a, 1 = b); rows >= n give zeros, pose rows come from sub
    auto build_own = [&](auto UB, int k, const float* sub) {
        constexpr int ub = decltype(UB)::value;
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const int    r = 2 * i;
            const int    q = 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int    o = (q >> 1) < kLaMaxObs ? (q >> 1) : 0;
            const float* c = &s_model[ub][14 + o * 10];
            float        v0, v1;
            la_pht_pair<float>(c, pv0, pv1, pv2, pcol[ub][2 * i], pcol[ub][2 * i + 1], &v0, &v1);
            const bool in = (q < k) && (row < a.n);
            if (row < 3 && in)
            {
                v0 = sub[row * k + q];
                v1 = sub[row * k + q + 1];
            }
            own[r]     = in ? v0 : 0.f;
            own[r + 1] = in ? v1 : 0.f;
        }
    };
    // this wave's column tile of W1 = PHT * G on the matrix cores, and its share of the row sums:
    // wave 0: X += PHT u and Pv[:, 0] -= PHT M_0;  wave 1: Pv[:, 1], Pv[:, 2]
    f32x16 w1;
    auto   gain = [&](auto UB, int k) {
        constexpr int ub = decltype(UB)::value;
        const float*  Gs = s_G[ub];
        const float*  v0 = wv == 0 ? s_u[ub] : s_M[ub] + k;
        const float*  v1 = wv == 0 ? s_M[ub] : s_M[ub] + 2 * k;
        f32x16        ac0 = f32x16{0}, ac1 = f32x16{0}; // two chains (even / odd steps), added at the end
        float         sa = 0.f, sb = 0.f;
        const int     cg   = 32 * wv + lj;
        const bool    cin  = cg < k;
        const int     cgc  = cin ? cg : 0;
        // operands first (clamped addresses, unconditional LDS reads, the VALUE is selected: a guarded read compiles to a
        // branch per read and serialises the whole loop), then the matrix-core steps
        float g[32], e0[32], e1[32];
#pragma unroll
        for (int i = 0; i < 32; i++)
        {
            const int q  = la_q_of(i >> 4, i & 15, lh);
            const int qc = q < k ? q : 0;
            const float gv = Gs[qc * k + cgc];
            g[i]  = (q < k && cin) ? gv : 0.f;
            e0[i] = v0[qc];
            e1[i] = v1[qc];
        }
#pragma unroll
        for (int i = 0; i < 32; i++)
        {
            const float b = full[i]; // (zero beyond k: the clamped operands above meet a zero)
            if (i & 1)
            {
                ac1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g[i], b, ac1, 0, 0, 0);
            }
            else
            {
                ac0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g[i], b, ac0, 0, 0, 0);
            }
            sa += b * e0[i];
            sb += b * e1[i];
        }
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            w1[r] = ac0[r] + ac1[r];
        }
        sa += __shfl_xor(sa, 32);
        sb += __shfl_xor(sb, 32);
        if (lh == 0)
        {
            s_sum[2 * wv][lj]     = sa;
            s_sum[2 * wv + 1][lj] = sb;
        }
    };
    auto store_w1 = [&](float* W, int k, bool last) {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int c = la_q_of(wv, r, lh);
            if (c < k)
            {
                if (row < 3) // pose rows of the panel: kept aside, stored as zero (the stripe carries them)
                {
                    if (last)
                    {
                        a.wv_out[(size_t)row * k + c] = w1[r];
                    }
                    W[(size_t)c * a.ldw + row] = 0.f;
                }
                else
                {
                    W[(size_t)c * a.ldw + row] = w1[r];
                }
            }
        }
    };
    using U0 = std::integral_constant<int, 0>;
    using U1 = std::integral_constant<int, 1>;

    // ================= update a =================
    if (a.valid_a && row >= 3 && row - 3 < a.w_a)
    {
        float o0, o1, o2;
        predict_stripe_col<float>(s_model[0][0], s_model[0][1], pv0, pv1, pv2, &o0, &o1, &o2);
        pv0 = o0, pv1 = o1, pv2 = o2;
    }
    build_own(U0{}, ka, a.sub_a);
    share();
    stamp();
    gain(U0{}, ka);
    stamp();
    store_w1(a.W1a, ka, a.nu == 1);
    const float* pred_last = s_model[0];
    if (a.nu == 2)
    {
        // W1_a[row, :] for the correction: both tiles through the exchange area (after everybody has read PHT_a from it)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            own[r] = (row >= 3) ? w1[r] : 0.f; // (the panel's pose rows are zero)
        }
        share(); // (its barrier also publishes the row sums of update a)
        const float xs = s_sum[0][lj], xm0 = s_sum[1][lj], xm1 = s_sum[2][lj], xm2 = s_sum[3][lj];
        if (row >= 3) // state of the rows after update a (pose rows: the chain's values are taken at the end)
        {
            x += xs;
            pv0 -= xm0, pv1 -= xm1, pv2 -= xm2;
        }
        stamp();
        // ================= update b =================
        if (a.valid_b && row >= 3 && row - 3 < a.w_b)
        {
            float o0, o1, o2;
            predict_stripe_col<float>(s_model[1][0], s_model[1][1], pv0, pv1, pv2, &o0, &o1, &o2);
            pv0 = o0, pv1 = o1, pv2 = o2;
        }
        build_own(U1{}, kb, a.sub_b);
        // this wave's tile of PHT_b -= W1_a * Y_b^T : D[i = q][j = row] += Y_b[c][q] * W1_a[row][c] over c (both tiles)
        {
            f32x16     c0 = f32x16{0}, c1 = f32x16{0};
            const int  qg  = 32 * wv + lj;
            const bool qin = qg < kb;
            const int  qgc = qin ? qg : 0;
            float      y[32];
#pragma unroll
            for (int i = 0; i < 32; i++)
            {
                const int   c  = la_q_of(i >> 4, i & 15, lh);
                const int   cc = c < ka ? c : 0;
                const float yv = s_Y[cc * kb + qgc];
                y[i]           = (c < ka && qin) ? yv : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 32; i++)
            {
                if (i & 1)
                {
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y[i], full[i], c1, 0, 0, 0);
                }
                else
                {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y[i], full[i], c0, 0, 0, 0);
                }
            }
            if (row >= 3)
            {
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    own[r] -= c0[r] + c1[r];
                }
            }
        }
        stamp();
        // the exchange area moves into G_a's space (dead since gain a); G_b is staged into its own
        __syncthreads(); // (everybody is done with W1_a in the old exchange area)
        dma(a.Gt_b, kb * kb, s_G[1]);
        xch = s_G[0] + pr * 2048;
        share();
        __builtin_amdgcn_s_waitcnt(0x0F70); // G_b has landed ...
        __syncthreads();                    // ... for both waves; the row sums of update a have been read by everybody
        stamp();
        gain(U1{}, kb);
        stamp();
        store_w1(a.W1b, kb, true);
        pred_last = s_model[1];
    }
    __syncthreads(); // the row sums of the last update
    stamp();
    if constexpr (PAIRS == 2)
    {
        if (a.wg_times != nullptr && tid == 0)
        {
            a.wg_times[2 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    }
    // ================= commit X and the stripe =================
    if (wv == 0 && lh == 0 && row < a.n)
    {
        const float xs = s_sum[0][lj], xm0 = s_sum[1][lj], xm1 = s_sum[2][lj], xm2 = s_sum[3][lj];
        if (row >= 3)
        {
            a.X[row]                        = x + xs;
            a.Pv[(size_t)0 * a.ldp + row] = pv0 - xm0;
            a.Pv[(size_t)1 * a.ldp + row] = pv1 - xm1;
            a.Pv[(size_t)2 * a.ldp + row] = pv2 - xm2;
        }
        else
        {
            // pose rows: start from the chain's predicted pose / pose block of the last update
            const int   r     = row;
            const float av[3] = {pred_last[5 + r], pred_last[5 + r + 3], pred_last[5 + r + 6]};
            const float dv[3] = {xm0, xm1, xm2};
            a.X[r]            = pred_last[2 + r] + xs;
#pragma unroll
            for (int c = 0; c < 3; c++)
            {
                if (c == r)
                {
                    a.Pv[(size_t)c * a.ldp + r] = av[c] - dv[c];
                }
                else if (c < r)
                {
                    const float bcr             = pred_last[5 + c + 3 * r]; // (c, r)
                    a.Pv[(size_t)c * a.ldp + r] = av[c] - dv[c];
                    a.Pv[(size_t)r * a.ldp + c] = bcr - dv[c];
                }
            }
        }
    }
}

// (the __global__ wrapper ekf_la_wide_f32 lives in cslam_ekf.hip, the batched one in cslam_ekf_batch.hip: a non-template kernel
// in a header would be defined once per translation unit)

// ------------------------------------------------------------------------------------------------
// BATCHED look-ahead windows (cslam_ekf_batch.hip): I independent f32 filters of the same size advance in lockstep -- the
// Monte-Carlo runs of BASELINE configs[4], test/main.cpp:132-200 x I.  Every stage of a window is ONE launch for all
// instances (blockIdx.y = instance): rows, blocks, the factor chain (I workgroups on stream F) and the wide half; the
// P-GEMMs are I launches of the single-filter kernel.  The instances live in slabs with a fixed stride per buffer class,
// so a kernel derives its instance's arguments from this one struct (passed by value: no per-window copies).
// ------------------------------------------------------------------------------------------------
namespace labatch
{
// per-instance factor block (floats): two slots (update a / b of the window)
constexpr int kS = 0, kG = 4160, kGt = 8320, kV = 12480, kT = 12544, kU = 12608, kM = 12672, kSub = 12864, kXloc = 17152;
constexpr int kSlot = 17280, kFoBlock = 2 * kSlot;
// per-instance look-ahead block (floats)
constexpr int kXL = 0, kPvL = 128, kPH = 512, kPvLb = 4608, kDbb = 4864, kY = 8960, kModel = 13056, kWR = 13760;
constexpr int kModelStride = 336, kKpad = 128, kLaBlock = kWR + kKpad * 128;
constexpr int kDoneBlock = 768; // unsigned words per instance: 16 counters (stride 16) + 32 chain words (stride 16) at 256
} // namespace labatch

struct LaBatchWin
{
    int I, n, ldp, lower, textbook;
    float*  X;    // [I][ldp]
    float*  Pv;   // [I][3 ldp]
    float*  P;    // [I][ldp ldp]
    float*  Wp;   // pending region of instance 0 (its kp columns are what the coming P-GEMM applies)
    float*  Wn;   // the region the window's W1 panels go to (instance 0, column 0)
    long    sW;   // per-instance stride of the pending store (floats)
    float*  fo;   // [I][kFoBlock]
    float*  la;   // [I][kLaBlock]
    float*  wv;   // [I][3 * 64]
    unsigned* done; // [I][kDoneBlock]
    int*      flags; // [I][2]
    int*      idloc; // [I][kLaMaxObs]: 1 .. m
    const float* const* Ztab; // user pointers per instance
    const int* const*   idftab;
    long zoff_a, zoff_b, ioff_a, ioff_b; // element offsets of update a / b in each instance's arrays
    int  ma, mb, nu;
    long long* stamps; // diagnostics (CSLAM_BATCH_STAMPS): phase stamps of one workgroup of instance 0's wide kernel
    int  wide_direct; // timing experiment only (wrong results): the wide kernel reads P(row, col) where it lies, no mirroring
    int  wg_signal; // 1: every workgroup of the blocks kernel releases its rows itself (A/B switch of the batched engine)
    PredictArgs<float> pp_a, pp_b;
    float R[4];
    int   kp;
    unsigned target, seq;
    unsigned long long timeout;
};

__device__ __forceinline__ LaRowsArgs<float> la_batch_rows(const LaBatchWin& w, int i)
{
    using namespace labatch;
    float*            la = w.la + (size_t)i * kLaBlock;
    LaRowsArgs<float> a;
    a.X     = w.X + (size_t)i * w.ldp;
    a.Pv    = w.Pv + (size_t)i * 3 * w.ldp;
    a.ldp   = w.ldp;
    a.n     = w.n;
    a.idf_a = w.idftab[i] + w.ioff_a;
    a.ra    = 2 * w.ma;
    a.idf_b = w.idftab[i] + (w.nu == 2 ? w.ioff_b : w.ioff_a);
    a.rb    = 2 * w.mb;
    a.Wp    = w.Wp + (size_t)i * w.sW;
    a.ldw   = w.ldp;
    a.kp    = w.kp;
    a.kpad  = kKpad;
    a.XL    = la + kXL;
    a.PvL   = la + kPvL;
    a.WR    = la + kWR;
    a.flags = w.flags + 2 * i;
    return a;
}

__device__ __forceinline__ LaPrepArgs<float> la_batch_prep(const LaBatchWin& w, int i)
{
    using namespace labatch;
    float*            la = w.la + (size_t)i * kLaBlock;
    float*            fo = w.fo + (size_t)i * kFoBlock;
    LaPrepArgs<float> a;
    a.P       = w.P + (size_t)i * w.ldp * w.ldp;
    a.ldp     = w.ldp;
    a.n       = w.n;
    a.lower   = w.lower;
    a.X       = w.X + (size_t)i * w.ldp;
    a.Pv      = w.Pv + (size_t)i * 3 * w.ldp;
    a.idf_a   = w.idftab[i] + w.ioff_a;
    a.idf_b   = w.idftab[i] + (w.nu == 2 ? w.ioff_b : w.ioff_a);
    a.ra      = 2 * w.ma;
    a.rb      = 2 * w.mb;
    a.pp_a    = w.pp_a;
    a.XL      = la + kXL;
    a.PvL     = la + kPvL;
    a.WR      = la + kWR;
    a.kp      = w.kp;
    a.kpad    = kKpad;
    a.sub_a   = fo + kSub;
    a.PH      = la + kPH;
    a.Dbb     = la + kDbb;
    a.PvLb    = la + kPvLb;
    a.model_a = reinterpret_cast<LaModel<float>*>(la + kModel);
    a.xloc_a  = fo + kXloc;
    a.idloc   = w.idloc + kLaMaxObs * i;
    a.done    = w.wg_signal ? w.done + (size_t)i * kDoneBlock : nullptr; // (0: signalled by the kernel that follows)
    return a;
}

__device__ __forceinline__ FactorArgs<float> la_batch_factor(const LaBatchWin& w, int i, int slot)
{
    using namespace labatch;
    float*            fo = w.fo + (size_t)i * kFoBlock + (size_t)slot * kSlot;
    FactorArgs<float> a;
    a.X   = fo + kXloc;
    a.n   = 3 + 2 * (slot == 0 ? w.ma : w.mb);
    a.Z   = w.Ztab[i] + (slot == 0 ? w.zoff_a : w.zoff_b);
    a.idf = w.idloc + kLaMaxObs * i;
    a.m   = slot == 0 ? w.ma : w.mb;
    for (int e = 0; e < 4; e++)
    {
        a.R[e] = w.R[e];
    }
    a.PHT      = nullptr;
    a.ldw      = w.ldp;
    a.dS       = fo + kS;
    a.dG       = fo + kG;
    a.dGt      = fo + kGt;
    a.dV       = fo + kV;
    a.dt       = fo + kT;
    a.flags    = w.flags + 2 * i;
    a.scratchS = nullptr;
    a.scratchG = nullptr;
    a.stamps   = nullptr;
    a.sub      = fo + kSub;
    a.dM       = fo + kM;
    a.pp       = PredictArgs<float>{0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0};
    a.P3       = w.Pv + (size_t)i * 3 * w.ldp;
    a.ldp3     = w.ldp;
    a.pred_out = nullptr;
    a.lds_S    = 1;
    a.lds_G    = 1;
    a.textbook = w.textbook;
    return a;
}

__device__ __forceinline__ LaChainArgs<float> la_batch_chain(const LaBatchWin& w, int i)
{
    using namespace labatch;
    float*             la = w.la + (size_t)i * kLaBlock;
    float*             f0 = w.fo + (size_t)i * kFoBlock;
    float*             f1 = f0 + kSlot;
    LaChainArgs<float> c;
    c.fa   = la_batch_factor(w, i, 0);
    c.fb   = la_batch_factor(w, i, w.nu == 2 ? 1 : 0);
    c.du_a = f0 + kU;
    c.du_b = f1 + kU;
    c.nu   = w.nu;
    c.done = w.done + (size_t)i * kDoneBlock;
    c.target     = w.target;
    c.timeout    = w.timeout;
    c.chain_done = w.done + (size_t)i * kDoneBlock + 256;
    c.seq        = w.seq;
    LaCarryArgs<float>& ca = c.ca;
    ca.n       = w.n;
    ca.m_a     = w.ma;
    ca.m_b     = w.nu == 2 ? w.mb : 0;
    ca.idf_b   = w.idftab[i] + (w.nu == 2 ? w.ioff_b : w.ioff_a);
    ca.pp_b    = w.nu == 2 ? w.pp_b : w.pp_a;
    ca.PH      = la + kPH;
    ca.Dbb     = la + kDbb;
    ca.PvLb    = la + kPvLb;
    ca.XLb     = la + kXL + 2 * w.ma;
    ca.model_a = reinterpret_cast<const LaModel<float>*>(la + kModel);
    ca.Gt_a    = f0 + kGt;
    ca.u_a     = f0 + kU;
    ca.M_a     = f0 + kM;
    ca.sub_a   = f0 + kSub;
    ca.sub_b   = f1 + kSub;
    ca.xloc_b  = f1 + kXloc;
    ca.model_b = reinterpret_cast<LaModel<float>*>(la + kModel + kModelStride);
    ca.Y_b     = la + kY;
    return c;
}

__device__ __forceinline__ LaWideArgs la_batch_wide(const LaBatchWin& w, int i)
{
    using namespace labatch;
    float*     la = w.la + (size_t)i * kLaBlock;
    float*     f0 = w.fo + (size_t)i * kFoBlock;
    float*     f1 = f0 + kSlot;
    LaWideArgs a;
    a.P       = w.P + (size_t)i * w.ldp * w.ldp;
    a.ldp     = w.ldp;
    a.n       = w.n;
    a.lower   = w.wide_direct ? 0 : w.lower;
    a.X       = w.X + (size_t)i * w.ldp;
    a.Pv      = w.Pv + (size_t)i * 3 * w.ldp;
    a.nu      = w.nu;
    a.idf_a   = w.idftab[i] + w.ioff_a;
    a.idf_b   = w.idftab[i] + (w.nu == 2 ? w.ioff_b : w.ioff_a);
    a.ma      = w.ma;
    a.mb      = w.nu == 2 ? w.mb : 0;
    a.valid_a = w.pp_a.valid;
    a.valid_b = w.nu == 2 ? w.pp_b.valid : 0;
    a.w_a     = w.pp_a.w;
    a.w_b     = w.nu == 2 ? w.pp_b.w : 0;
    a.model_a = reinterpret_cast<const LaModel<float>*>(la + kModel);
    a.model_b = reinterpret_cast<const LaModel<float>*>(la + kModel + kModelStride);
    a.Gt_a    = f0 + kGt;
    a.u_a     = f0 + kU;
    a.M_a     = f0 + kM;
    a.sub_a   = f0 + kSub;
    a.Gt_b    = f1 + kGt;
    a.u_b     = f1 + kU;
    a.M_b     = f1 + kM;
    a.sub_b   = f1 + kSub;
    a.Y_b     = la + kY;
    a.W1a     = w.Wn + (size_t)i * w.sW;
    a.W1b     = a.W1a + (size_t)(2 * w.ma) * w.ldp;
    a.ldw     = w.ldp;
    a.wv_out  = w.wv + (size_t)i * 192;
    a.chain_done = w.done + (size_t)i * kDoneBlock + 256;
    a.seq        = w.seq;
    a.timeout    = w.timeout;
    a.flags      = w.flags + 2 * i;
    a.stamps     = i == 0 ? w.stamps : nullptr;
    a.wg_times   = w.stamps ? w.stamps + 32 + (size_t)i * 2 * 128 : nullptr;
    return a;
}

} // namespace cslam
