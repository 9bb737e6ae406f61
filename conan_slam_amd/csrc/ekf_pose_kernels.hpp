// ekf_pose_kernels.hpp -- the O(n) kernels that work on the pose stripe Pv (see p_get in ekf_kernels.hpp):
//
//   ekf_pose_step_kernel      EKF::predict (EKF.cpp:406-455) and / or EKF::observeHeading (EKF.cpp:328-352 with
//                             josephUpdate slam.h:700-725) in ONE launch -- the reference's driver calls the two back to
//                             back on every control step (test/main.cpp:165-168).
//   ekf_pose_downdate_kernel  the pose-stripe share of slam.h:260, P[:,0:3] -= W1 * W1[0:3,:]^T, applied right behind
//                             the gain kernel; it then ZEROES the pose rows of the W1 panel, so that the pending-panel
//                             algebra (P = Ps - Wp Wp^T) never touches the stripe.
//   ekf_augment_kernel        EKF::addOneNewFeature (EKF.cpp:28-91), O(n) with the preallocated buffers.
//   ekf_patch_pose_kernel     Pv -> rows/columns 0..2 of the P buffer (get_state).
//
// Why the heading update is O(n) here.  With H = e_2^T, p = P[:,2], S = P22 + R, W = p/S, the Joseph form of
// slam.h:713-718 is, for a symmetric P,
//     (I - W H) P (I - W H)^T + W R W^T  =  P - p p^T / S                                   (exactly),
// i.e. ONE more rank-1 downdate of the same kind the update's W1 panels are: the column w = p / sqrt(S) is appended to
// the pending panels (applied by the next P-GEMM together with them) and only the pose stripe, which must stay current,
// is updated directly: P'[i,c] = P[i,c] - p_i p_c / S for c = 0,1 and P'[i,2] = p_i (R/S) -- the latter WITHOUT the
// cancellation 1 - P22/S that any evaluation of C = I - W H in floating point suffers (R/S ~ 1e-6 with the
// reference's sigma = 0.01 deg).  The reference's dense form costs 4 n^3 flops per control step; the earlier O(n^2)
// form of this engine swept all of P; this one reads and writes 3 n scalars.
#pragma once

#include "ekf_kernels.hpp"

namespace cslam
{

template <typename T>
struct HeadingArgs
{
    int valid;
    T   phi; // observed heading
    T   R;   // sigmaPhi^2 (EKF.cpp:337-343)
};

// A run of consecutive control steps: step s = [predict pp[s]] then [heading hd[s]] (either may be absent).  The 3 x 3
// pose block and the pose evolve through the run WITHOUT reference to the map rows (Pvv' = Gv Pvv Gv^T + Gu Q Gu^T;
// the heading update of the block needs the block only), so every thread replays that small recursion for itself and
// applies the per-step coefficients to its own row of the stripe: a whole run is ONE launch with no grid-wide
// synchronisation.  The reference's driver issues predict + observeHeading six times between two updates
// (test/main.cpp:165-174); the engine queues them and launches the run when something else needs X or Pv.
constexpr int kPoseSeqMax = 8;

template <typename T>
struct PoseSeq
{
    int            count;
    PredictArgs<T> pp[kPoseSeqMax];
    HeadingArgs<T> hd[kPoseSeqMax];
    int            col[kPoseSeqMax]; // heading steps: index of the pending column they write (-1: scratch / none)
};

// grid = ceil(n_pad/256) x 256 threads.  wbase / ldw: the pending store's current region (column c at wbase + c*ldw;
// n_pad entries written per heading column, zeros in rows 0..2 and [n, n_pad)); scratch: a column for heading steps
// without a map (col < 0).  sgn / neg_count: see below.  done: ticket counter (reset by the last workgroup, which is
// also the one that rewrites the 3 x 3 pose block and the pose -- everybody else has consumed the old values by then).
template <typename T>
__global__ void __launch_bounds__(256) ekf_pose_step_kernel(T* __restrict__ X, T* __restrict__ Pv, int ldp, int n,
                                                             int n_pad, PoseSeq<T> seq, T* __restrict__ wbase, int ldw,
                                                             T* __restrict__ scratch, int* __restrict__ sgn,
                                                             int* __restrict__ neg_count, int* __restrict__ done)
{
    // sgn / neg_count: the reference's Joseph form is finite for S = P22 + R < 0 too (an indefinite P, which
    // REF_EXACT's gain produces: SURVEY 2.1 #1/#3) and equals P - p p^T / S there as well, i.e. P + w w^T with
    // w = p / sqrt(|S|): the column is stored with sgn[col] = 1 and counted in *neg_count; the pending-panel correction
    // flips its sign (ekf_pending_y_kernel, ekf_gather_kernel) and ekf_negcol_fix adds 2 w w^T ahead of the P-GEMM.
    __shared__ int s_last;
    T              pvv[9];
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
    {
#pragma unroll
        for (int r = 0; r < 3; r++)
        {
            pvv[r + 3 * cc] = Pv[(size_t)cc * ldp + r];
        }
    }
    T         xo[3] = {X[0], X[1], X[2]};
    const int i     = blockIdx.x * 256 + threadIdx.x;
    const bool row  = (i >= 3 && i < n);
    T         a0 = (T)0, a1 = (T)0, a2 = (T)0, xi = (T)0;
    if (row)
    {
        a0 = Pv[(size_t)0 * ldp + i];
        a1 = Pv[(size_t)1 * ldp + i];
        a2 = Pv[(size_t)2 * ldp + i];
        xi = X[i];
    }
    int negs = 0;
    for (int s = 0; s < seq.count; s++)
    {
        const PredictArgs<T> pp = seq.pp[s];
        const HeadingArgs<T> hd = seq.hd[s];
        if (pp.valid)
        {
            T g02, g12;
            predict_gv<T>(pp, xo[2], &g02, &g12);
            T nv[9];
            predict_pvv<T>(pp, xo[2], pvv, nv);
            const T sn = dsin(pp.swa + xo[2]), cs = dcos(pp.swa + xo[2]); // as predicted_pose (EKF.cpp:445-452)
            const T nx = xo[0] + pp.v * pp.dt * cs;
            const T ny = xo[1] + pp.v * pp.dt * sn;
            const T nphi = pi2pi<T>(xo[2] + pp.v * pp.dt * dsin(pp.swa) / pp.wb);
            xo[0] = nx;
            xo[1] = ny;
            xo[2] = nphi;
#pragma unroll
            for (int e = 0; e < 9; e++)
            {
                pvv[e] = nv[e];
            }
            if (row && (i - 3) < pp.w) // column i of the stripe P[0:3, 3:3+w] = Gv * stripe, mirrored (EKF.cpp:442-443)
            {
                T o0, o1, o2;
                predict_stripe_col<T>(g02, g12, a0, a1, a2, &o0, &o1, &o2);
                a0 = o0;
                a1 = o1;
                a2 = o2;
            }
        }
        if (hd.valid)
        {
            // column 2 of the pose block, S, 1/S, the innovation (EKF.cpp:337-347, slam.h:708-713)
            const T pcol[3] = {pvv[0 + 6], pvv[1 + 6], pvv[2 + 6]}; // P[r,2]
            const T prow[3] = {pvv[2 + 0], pvv[2 + 3], pvv[2 + 6]}; // P[2,c]
            const T S       = pcol[2] + hd.R;
            const T si      = (T)1 / S;
            const T rs      = (T)1 / dsqrt(dabs(S));
            const T ros     = hd.R * si; // R/S = 1 - P22/S without the cancellation
            const T vin     = pi2pi<T>(hd.phi - xo[2]);
            T*      wcol    = (seq.col[s] >= 0) ? wbase + (size_t)seq.col[s] * ldw : scratch;
            if (row)
            {
                const T pi_ = a2;       // p_i = P[i,2]
                const T wi  = pi_ * si; // W = PHT * SI (slam.h:712)
                a0          = a0 - wi * pcol[0];
                a1          = a1 - wi * pcol[1];
                a2          = pi_ * ros;
                xi          = xi + wi * vin; // slam.h:713
                wcol[i]     = pi_ * rs;
            }
            else if (i < n_pad) // rows 0..2 (the stripe is never pending) and the padding rows
            {
                wcol[i] = (T)0;
            }
            T nv[9];
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
            {
#pragma unroll
                for (int r = 0; r < 3; r++)
                {
                    T val;
                    if (cc == 2)
                    {
                        val = pcol[r] * ros;
                    }
                    else if (r == 2)
                    {
                        val = prow[cc] * ros;
                    }
                    else
                    {
                        val = pvv[r + 3 * cc] - (pcol[r] * si) * prow[cc];
                    }
                    // + I * FLT_MIN (slam.h:719): the reference's driver starts from P = 0, which this keeps at FLT_MIN * I.
                    // Only the pose block carries it here: a map diagonal entry absorbs 1.2e-38 in rounding unless it is
                    // below 2^-102 (deliberate deviation)
                    nv[r + 3 * cc] = val + ((r == cc) ? (T)1.17549435e-38f : (T)0);
                }
            }
#pragma unroll
            for (int r = 0; r < 3; r++)
            {
                xo[r] = xo[r] + (pcol[r] * si) * vin;
            }
#pragma unroll
            for (int e = 0; e < 9; e++)
            {
                pvv[e] = nv[e];
            }
            if (seq.col[s] >= 0 && i == 0)
            {
                const int neg  = (S > (T)0) ? 0 : 1;
                sgn[seq.col[s]] = neg; // (workgroup 0 only: nobody reads it before the kernel ends)
                negs += neg;
            }
        }
    }
    if (row)
    {
        Pv[(size_t)0 * ldp + i] = a0;
        Pv[(size_t)1 * ldp + i] = a1;
        Pv[(size_t)2 * ldp + i] = a2;
        X[i]                    = xi;
    }
    if (i == 0 && negs > 0)
    {
        atomicAdd(neg_count, negs);
    }
    __syncthreads(); // every thread of this workgroup has consumed the old pose / pose block
    if (threadIdx.x == 0)
    {
        const int t = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last      = (t == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last == 0 || threadIdx.x != 0)
    {
        return;
    }
    *done = 0;
    for (int e = 0; e < 9; e++)
    {
        Pv[(size_t)(e / 3) * ldp + (e % 3)] = pvv[e];
    }
    X[0] = xo[0];
    X[1] = xo[1];
    X[2] = xo[2];
}

// Pose-stripe downdate behind the gain kernel: Pv[:, c] -= sum_q W1[:, q] * W1[c, q], c = 0..2; then the LAST workgroup
// zeroes W1[0:3, 0:k8) (everybody has read those rows by then) after saving them to wv_out (3 x k, row c at
// wv_out + c*k: what cslam_ekf_debug_last_update reports).  Four lanes share a row (columns q = part, part + 4, ...;
// partial sums joined by two shuffles) so that a thread has at most 16 independent loads in flight per 64 columns:
// the kernel is a latency chain, not a bandwidth problem (n*k*s bytes).  grid = ceil(n/64) x 256.
template <typename T>
__global__ void __launch_bounds__(256) ekf_pose_downdate_kernel(T* __restrict__ W1, int ldw, int k, int k8, int n,
                                                                 T* __restrict__ Pv, int ldp, T* __restrict__ wv_out,
                                                                 int* __restrict__ done)
{
    constexpr int QC = 64;
    __shared__ T   s_wv[3][QC];
    __shared__ int s_last;
    const int      part = threadIdx.x & 3;
    const int      i    = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int      il   = min(i, n - 1);
    T              acc0 = (T)0, acc1 = (T)0, acc2 = (T)0;
    for (int q0 = 0; q0 < k; q0 += QC)
    {
        const int qn = min(QC, k - q0);
        __syncthreads();
        if ((int)threadIdx.x < 3 * QC)
        {
            const int c = threadIdx.x / QC, q = threadIdx.x - c * QC;
            const T   v = (q < qn) ? W1[(size_t)(q0 + q) * ldw + c] : (T)0;
            s_wv[c][q]  = v;
            if (blockIdx.x == 0 && q < qn)
            {
                wv_out[(size_t)c * k + q0 + q] = v;
            }
        }
        T w[QC / 4];
#pragma unroll
        for (int j = 0; j < QC / 4; j++)
        {
            const int q = part + 4 * j;
            w[j]        = W1[(size_t)(q0 + min(q, qn - 1)) * ldw + il];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < QC / 4; j++)
        {
            const int q = part + 4 * j; // (columns past qn: s_wv holds zeros)
            acc0 += w[j] * s_wv[0][q];
            acc1 += w[j] * s_wv[1][q];
            acc2 += w[j] * s_wv[2][q];
        }
    }
    acc0 += __shfl_xor(acc0, 1);
    acc1 += __shfl_xor(acc1, 1);
    acc2 += __shfl_xor(acc2, 1);
    acc0 += __shfl_xor(acc0, 2);
    acc1 += __shfl_xor(acc1, 2);
    acc2 += __shfl_xor(acc2, 2);
    if (i < n && part < 3)
    {
        const T a = (part == 0) ? acc0 : ((part == 1) ? acc1 : acc2);
        Pv[(size_t)part * ldp + i] -= a;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int t = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last      = (t == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last == 0)
    {
        return;
    }
    if (threadIdx.x == 0)
    {
        *done = 0;
    }
    for (int e = threadIdx.x; e < 3 * k8; e += 256)
    {
        W1[(size_t)(e / 3) * ldw + (e % 3)] = (T)0;
    }
}

// Exceptional path of the heading update (see ekf_pose_step_kernel): for every pending column q of this region with
// sgn[q] != 0 the map block needs P += w_q w_q^T; the P-GEMM that follows subtracts w_q w_q^T, so 2 w_q w_q^T is added
// here.  Launched in front of a P-GEMM whose region holds heading columns; exits at once when *neg_count == 0 (the
// normal case; pending-panel corrections that read these signs, in the two-stream mode, are ordered before the NEXT
// use of the region, see Ekf::flush).  Persistent: workgroup b takes tiles b, b + grid, ... (tile_list: the block-lower tiles, or nullptr
// for every tile of the full matrix).
template <typename T>
__global__ void __launch_bounds__(256) ekf_negcol_fix_kernel(T* __restrict__ P, int ldp, int n, const T* __restrict__ W,
                                                              int ldw, int kp, int* __restrict__ sgn,
                                                              int* __restrict__ neg_count,
                                                              const int2* __restrict__ tile_list, int ntiles, int tiles_1d,
                                                              int* __restrict__ done, int reset)
{
    if (*neg_count == 0)
    {
        return;
    }
    __shared__ int s_last;
    for (int tt = blockIdx.x; tt < ntiles; tt += gridDim.x)
    {
        const int2 t = tile_list ? tile_list[tt] : make_int2(tt % tiles_1d, tt / tiles_1d);
        for (int e = threadIdx.x; e < 128 * 128; e += 256)
        {
            const int i = t.x * 128 + (e & 127), j = t.y * 128 + (e >> 7);
            if (i >= n || j >= n)
            {
                continue;
            }
            T acc = (T)0;
            for (int q = 0; q < kp; q++)
            {
                if (sgn[q])
                {
                    acc += W[(size_t)q * ldw + i] * W[(size_t)q * ldw + j];
                }
            }
            P[(size_t)j * ldp + i] += (T)2 * acc;
        }
    }
    // the workgroup that finishes last clears the signs and the count: the normal path never has to
    if (!reset)
    {
        return;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const int t = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last      = (t == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last)
    {
        for (int q = threadIdx.x; q < kp; q += 256)
        {
            sgn[q] = 0;
        }
        if (threadIdx.x == 0)
        {
            *neg_count = 0;
            *done      = 0;
        }
    }
}

// EKF.cpp:28-91 for one new feature observed at (r, b): X grows by 2, the new rows/columns of P are Gv*P[0:3, :]
// (EKF.cpp:77-84, from the pose stripe, which is current), the new 2 x 2 block is Gv Pvv Gv^T + Gz R Gz^T (EKF.cpp:74).
// The pending W1 panels keep ZERO rows for the new feature (their rows beyond n are zero), which is exactly right: the
// values written here are those of the true P.  grid = ceil(len/256) x 256 (nobody writes what another thread reads:
// X[2] and rows < len of the stripe are only read, rows len, len+1 only written).
template <typename T>
__global__ void __launch_bounds__(256) ekf_augment_kernel(T* __restrict__ X, T* __restrict__ P, T* __restrict__ Pv,
                                                           int ldp, int len, T r, T b, T r00, T r10, T r01, T r11,
                                                           int lower)
{
    const T s = dsin(X[2] + b), c = dcos(X[2] + b);
    const T Gv[6] = {(T)1, (T)0, (T)0, (T)1, -r * s, r * c}; // 2x3 column-major
    const int j   = blockIdx.x * 256 + threadIdx.x;
    if (j == 0)
    {
        X[len]     = X[0] + (r * c);
        X[len + 1] = X[1] + (r * s);
        T Gz[4] = {c, s, -r * s, r * c};
        T R[4]  = {r00, r10, r01, r11};
        T GvP[6];
        for (int cc = 0; cc < 3; cc++)
        {
            for (int rr = 0; rr < 2; rr++)
            {
                T acc = (T)0;
                for (int l = 0; l < 3; l++)
                {
                    acc += Gv[rr + 2 * l] * Pv[(size_t)cc * ldp + l];
                }
                GvP[rr + 2 * cc] = acc;
            }
        }
        T GzR[4];
        for (int cc = 0; cc < 2; cc++)
        {
            for (int rr = 0; rr < 2; rr++)
            {
                T acc = (T)0;
                for (int l = 0; l < 2; l++)
                {
                    acc += Gz[rr + 2 * l] * R[l + 2 * cc];
                }
                GzR[rr + 2 * cc] = acc;
            }
        }
        for (int cc = 0; cc < 2; cc++)
        {
            for (int rr = 0; rr < 2; rr++)
            {
                T a1 = (T)0;
                for (int l = 0; l < 3; l++)
                {
                    a1 += GvP[rr + 2 * l] * Gv[cc + 2 * l];
                }
                T a2 = (T)0;
                for (int l = 0; l < 2; l++)
                {
                    a2 += GzR[rr + 2 * l] * Gz[cc + 2 * l];
                }
                P[(size_t)(len + cc) * ldp + len + rr] = a1 + a2; // EKF.cpp:74
            }
        }
    }
    if (j >= len)
    {
        return;
    }
    // EKF.cpp:77-78, 83-84: new rows = Gv * P[0:3, 0:len], mirrored into the new columns
    const T a0 = Pv[(size_t)0 * ldp + j], a1 = Pv[(size_t)1 * ldp + j], a2 = Pv[(size_t)2 * ldp + j];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
    {
        T acc = (T)0;
        acc += Gv[rr + 0] * a0;
        acc += Gv[rr + 2] * a1;
        acc += Gv[rr + 4] * a2;
        if (j < 3)
        {
            Pv[(size_t)j * ldp + len + rr] = acc; // pose columns of the new rows: the stripe
        }
        else
        {
            P[(size_t)j * ldp + len + rr] = acc;
            if (!lower || ((j >> 7) == ((len + rr) >> 7)))
            {
                P[(size_t)(len + rr) * ldp + j] = acc; // the mirror exists under full storage / inside a diagonal tile
            }
        }
    }
}

// rows / columns 0..2 of the P buffer <- Pv (before P is handed to the host); grid = ceil(n/256) x 256
template <typename T>
__global__ void __launch_bounds__(256) ekf_patch_pose_kernel(T* __restrict__ P, const T* __restrict__ Pv, int ldp, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
    {
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
        const T v              = Pv[(size_t)c * ldp + i];
        P[(size_t)c * ldp + i] = v;
        if (i >= 3)
        {
            P[(size_t)i * ldp + c] = v;
        }
    }
}

} // namespace cslam
