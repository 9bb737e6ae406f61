// ekf_kernels_fast.hpp -- the tuned gfx950 kernels of the update chain (same arithmetic as the general
// kernels in ekf_kernels.hpp, which remain the path for shapes these do not cover).
//
// Shipped path (f32, 16 < k <= 64 per update):
//   ekf_factor_mfma_f32<K>        S build from the gather kernel's compact block, then the Cholesky factorisation and
//                                 the triangular inverse as rank-1 updates on the matrix cores (accumulator-layout
//                                 tiles: the pivot row is an MFMA operand vector, no broadcasts).
//   ekf_panel_mfma_f32            W1 = PHT*G on v_mfma_f32_32x32x2_f32, X += PHT*u fused (u = G G^T V); commits a
//                                 pending predict; also the deferred-mode correction PHT -= Wp*Y^T.
//   ekf_downdate_psym4_f32<.,NCH> the P-GEMM P -= W1*W1^T: persistent, symmetric (block-lower storage), dynamic tile
//                                 tickets, every memory operation issued from inside the MFMA loop; NCH = 2 (k <= 64)
//                                 or 4 (k <= 128) chunks of 32 columns.
//   (predict, heading, augment and the pose-stripe downdate live in ekf_pose_kernels.hpp)
// Other shapes:
//   ekf_factor_small_kernel<T,K>  k <= 16: one wave holds the matrix a row per lane; every multiplier broadcast by v_readlane.
//   ekf_factor_mfma_f64<K>        the f64 counterpart of the matrix-core factorisation (v_mfma_f64_16x16x4_f64).
//   ekf_factor_mfma_big_f32<128>  64 < k <= 128 in f32: four 32-wide blocks.
//   ekf_panel_mfma_f64            the f64 gain / correction panel product.
//   ekf_downdate_psym_f32         the unpipelined persistent symmetric P-GEMM: any k (windows beyond 128 columns), and the
//                                 full-storage form (CSLAM_STORAGE=full) with mirror stores.
// Earlier generations (a tile-per-workgroup P-GEMM, a P-GEMM pipelined across tiles only, workgroup-parallel and 2x2-blocked
// factorisations, a triangular-solve gain kernel) were measured against these and removed; DESIGN.md 8 keeps the numbers.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "ekf_kernels.hpp"

namespace cslam
{

// ------------------------------------------------------------------------------------------------
// wave-level broadcast of lane `src`'s value (src is a compile-time constant after unrolling)
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline float bcast(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ inline double bcast(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// reciprocal square root of the pivot: one v_rsq_f32 (<= 1 ulp) for f32 instead of the ~35-instruction
// IEEE sqrt + divide chain that otherwise sits on the serial path of every column; exact for f64.
__device__ inline float  pivot_rsqrt(float d) { return __builtin_amdgcn_rsqf(d); }
__device__ inline double pivot_rsqrt(double d) { return 1.0 / sqrt(d); }

// ------------------------------------------------------------------------------------------------
// K2+K3 for k <= K (K a power of two <= 64).  Same outputs as ekf_factor_kernel plus du = G*(G^T V).
// Padding rows/columns [k, K) of S are the identity, so L and inv(L) are blkdiag(., I).
// ------------------------------------------------------------------------------------------------
template <typename T, int K>
__global__ void __launch_bounds__(256) ekf_factor_small_kernel(FactorArgs<T> a, T* __restrict__ du)
{
    constexpr int LD = K + 1;
    __shared__ T   S[K * LD];
    __shared__ T   coef[(K / 2) * 10];
    __shared__ T   V[K];
    __shared__ T   tvec[K];
    __shared__ int fxs[K / 2];
    __shared__ int sflg[2];
    const int      k   = 2 * a.m;
    const int      tid = threadIdx.x;

    auto stamp = [&](int i) {
        if (a.stamps && tid == 0)
        {
            a.stamps[i] = (long long)__builtin_readcyclecounter();
        }
    };
    stamp(0);
    if (tid == 0)
    {
        sflg[0] = 0;
        sflg[1] = 0;
    }
    if (tid < K)
    {
        V[tid] = (T)0;
    }
    __syncthreads();
    for (int o = tid; o < a.m; o += 256)
    {
        observe_model<T>(a.X, a.n, a.idf[o], a.Z[2 * o], a.Z[2 * o + 1], &coef[o * 10], &V[2 * o], &fxs[o]);
        a.dV[2 * o]     = V[2 * o];
        a.dV[2 * o + 1] = V[2 * o + 1];
    }
    __syncthreads();
    // S = H*PHT + RR (slam.h:244): 5-term sums in ascending column order; identity padding.
    // Two passes with compile-time trip counts: first every global load of the thread's elements is issued
    // (they are independent L2 hits), then the sums are formed -- one round trip instead of one per element.
    {
        constexpr int NE = (K * K + 255) / 256;
        const T       r00 = a.R[0], r10 = a.R[1], r01 = a.R[2], r11 = a.R[3]; // no indexed kernarg loads below
        T             ph[NE][5];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e  = tid + it * 256;
            const int r  = e & (K - 1);
            const int c  = e / K;
            const bool in = (e < K * K) && (r < k) && (c < k);
            const int rc = in ? r : 0, cc = in ? c : 0; // clamped: loads stay unconditional
            const int fx = fxs[rc >> 1];
            const T*  p  = a.PHT + (size_t)cc * a.ldw;
            ph[it][0]    = p[0];
            ph[it][1]    = p[1];
            ph[it][2]    = p[2];
            ph[it][3]    = p[fx];
            ph[it][4]    = p[fx + 1];
        }
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e = tid + it * 256;
            if (e < K * K)
            {
                const int r = e & (K - 1);
                const int c = e / K;
                T         v;
                if (r < k && c < k)
                {
                    const int ob = r >> 1, ra = r & 1;
                    const T*  cf = &coef[ob * 10 + ra * 5];
                    T         sm = cf[0] * ph[it][0];
                    sm += cf[1] * ph[it][1];
                    sm += cf[2] * ph[it][2];
                    sm += cf[3] * ph[it][3];
                    sm += cf[4] * ph[it][4];
                    const int ri = ra + 2 * (c & 1);
                    const T   rv = (ri == 0) ? r00 : ((ri == 1) ? r10 : ((ri == 2) ? r01 : r11));
                    v = sm + (((c >> 1) == ob) ? rv : (T)0);
                }
                else
                {
                    v = (r == c) ? (T)1 : (T)0;
                }
                S[r + c * LD] = v;
            }
        }
    }
    __syncthreads();
    // makeSymmetric (slam.h:776-779)
    for (int e = tid; e < K * K; e += 256)
    {
        const int r = e & (K - 1);
        const int c = e / K;
        if (r > c)
        {
            T v           = (S[r + c * LD] + S[c + r * LD]) * (T)0.5;
            S[r + c * LD] = v;
            S[c + r * LD] = v;
        }
        else if (r == c)
        {
            T d           = S[r + c * LD];
            S[r + c * LD] = (d + d) * (T)0.5;
        }
    }
    __syncthreads();
    for (int e = tid; e < K * K; e += 256)
    {
        const int r = e & (K - 1), c = e / K;
        if (r < k && c < k)
        {
            a.dS[r + c * k] = S[r + c * LD];
        }
    }
    __syncthreads();

    stamp(1);
    if (tid < 64) // ---------------- one wave: lane = row of S / column of inv(L)
    {
        const int lane = tid;
        T         row[K];
#pragma unroll
        for (int c = 0; c < K; c++)
        {
            row[c] = (lane < K) ? S[lane + c * LD] : ((c == lane) ? (T)1 : (T)0);
        }
        bool failed = false;
        T    rdiag[K]; // wave-uniform reciprocals of the diagonal of L
        // right-looking lower Cholesky; a pivot <= 0 is the LLT failure of slam.h:421
#pragma unroll
        for (int j = 0; j < K; j++)
        {
            if (!failed)
            {
                const T dj = bcast(row[j], j);
                if (dj <= (T)0)
                {
                    failed = true;
                }
                else
                {
                    const T rs = pivot_rsqrt(dj); // 1/sqrt(pivot)
                    row[j]     = (lane == j) ? dj * rs : row[j] * rs;
                    rdiag[j]   = rs; // 1/L[j][j], reused by the inverse
#pragma unroll
                    for (int c = j + 1; c < K; c++)
                    {
                        const T l = bcast(row[j], c);
                        row[c] -= row[j] * l;
                    }
                }
            }
        }
        stamp(2);
        // inv(L) by forward substitution, lane = column; L[r][q] is lane r's register q
        T    x[K];
        bool bad = false;
        if (!failed)
        {
#pragma unroll
            for (int r = 0; r < K; r++)
            {
                T s = (T)0;
#pragma unroll
                for (int q = 0; q < r; q++)
                {
                    s += bcast(row[q], r) * x[q];
                }
                x[r] = (((lane == r) ? (T)1 : (T)0) - s) * rdiag[r];
                bad       = bad || !dfinite(x[r]);
            }
            bad = (__ballot(bad && lane < k) != 0ull);
        }
        stamp(3);
        const bool zero = failed || bad;
        // G back into LDS (over S): REF_EXACT G = inv(L) -> G[r][c] = x[r] of lane c; TEXTBOOK G = inv(L)^T
        if (lane < K)
        {
#pragma unroll
            for (int r = 0; r < K; r++)
            {
                const T g = zero ? (T)0 : x[r];
                if (a.textbook)
                {
                    S[lane + r * LD] = g; // G[c][r] = inv(L)[r][c]
                }
                else
                {
                    S[r + lane * LD] = g;
                }
            }
        }
        if (lane == 0)
        {
            sflg[0] = failed ? 1 : 0;
            sflg[1] = (!failed && bad) ? 1 : 0;
        }
    }
    __syncthreads();
    // outputs: G, G^T (coalesced), t = G^T V, u = G t
    for (int e = tid; e < K * K; e += 256)
    {
        const int r = e & (K - 1), c = e / K;
        if (r < k && c < k)
        {
            a.dG[r + c * k] = S[r + c * LD];
        }
    }
    for (int e = tid; e < K * K; e += 256)
    {
        const int c = e & (K - 1), r = e / K;
        if (r < k && c < k)
        {
            a.dGt[c + r * k] = S[r + c * LD];
        }
    }
    // t = G^T V and u = G t: 4 lanes per output element, partial sums combined in a fixed order
    {
        const int o = tid >> 2, part = tid & 3; // 256 threads = 64 outputs x 4 parts
        T         s = (T)0;
        if (o < K)
        {
#pragma unroll 4
            for (int r = part; r < K; r += 4)
            {
                s += S[r + o * LD] * V[r]; // padding rows of V are zero
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0 && o < K)
        {
            if (o < k)
            {
                a.dt[o] = s;
            }
            tvec[o] = (o < k) ? s : (T)0;
        }
        __syncthreads();
        T s2 = (T)0;
        if (o < K)
        {
#pragma unroll 4
            for (int c = part; c < K; c += 4)
            {
                s2 += S[o + c * LD] * tvec[c];
            }
        }
        s2 += __shfl_xor(s2, 1);
        s2 += __shfl_xor(s2, 2);
        if (part == 0 && o < k)
        {
            du[o] = s2;
        }
        // M[:, c] = G (G^T PHT[c, :]^T), c = 0..2: lets the gain kernel apply the pose-stripe downdate itself
        // (see ekf_factor_mfma_f32; the sequential update runs this kernel once per observation)
        if (a.dM != nullptr)
        {
            for (int c = 0; c < 3; c++)
            {
                __syncthreads();
                T s3 = (T)0;
                if (o < K)
                {
#pragma unroll 4
                    for (int r = part; r < K; r += 4)
                    {
                        s3 += S[r + o * LD] * ((r < k) ? a.PHT[(size_t)r * a.ldw + c] : (T)0);
                    }
                }
                s3 += __shfl_xor(s3, 1);
                s3 += __shfl_xor(s3, 2);
                if (part == 0 && o < K)
                {
                    tvec[o] = (o < k) ? s3 : (T)0;
                }
                __syncthreads();
                T s4 = (T)0;
                if (o < K)
                {
#pragma unroll 4
                    for (int q = part; q < K; q += 4)
                    {
                        s4 += S[o + q * LD] * tvec[q];
                    }
                }
                s4 += __shfl_xor(s4, 1);
                s4 += __shfl_xor(s4, 2);
                if (part == 0 && o < k)
                {
                    a.dM[c * k + o] = s4;
                }
            }
        }
    }
    stamp(4);
    if (tid == 0)
    {
        const int code = (sflg[0] ? kFlagLltFailed : 0) | (sflg[1] ? kFlagZeroed : 0);
        a.flags[1]     = code;
        if (code)
        {
            atomicOr(&a.flags[0], code);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2+K3 (f32) for 16 < k <= 64 with the rank-1 updates of the Cholesky factorisation and of the triangular
// inverse on the matrix cores.  Same inputs/outputs/flags as ekf_factor_small_kernel<float, K>.
//
// Why: in the register-resident kernel above every multiplier of a rank-1 update is broadcast with v_readlane
// (8 issue cycles each, measured with tools/probes/issue_probe.hip: 12 cycles per element pair), 2016 pairs for
// the factorisation and again for the inverse: 36k + 23k cycles at k = 64.  v_mfma_f32_32x32x2_f32 performs the
// whole rank-1 update of a 32 x 32 tile in ONE instruction and needs no broadcast: its operands are vectors
// spread over the lanes, and the pivot row of a symmetric matrix held in the accumulator layout is exactly that.
//
// Layout (wave 0 only; lane l: h = l>>5, c = l&31): tile T[I][J] is an f32x16 accumulator,
//     T[I][J][r] of lane l  =  M[32 I + (r&3) + 8 (r>>2) + 4 h][32 J + c].
// Row j of the matrix (j = 32 J' + jj) therefore sits in register rj = (jj&3) + 4 (jj>>3) of the lanes with
// h = (jj>>2)&1, one column per lane: a ready-made MFMA operand (A[i = l&31][kslot = l>>5], B[kslot][c = l&31];
// the other k-slot is fed zeros).  The matrix is kept fully symmetric (T01 is updated as well) so that row j is
// column j.
//   Cholesky step j:  d = M[j][j] (one v_readlane), rs = 1/sqrt(d), a = row_j * rs (= column j of L, masked to
//                     rows >= j), L[:, j] -> LDS,  T -= a a^T  (3 MFMAs for j < 32, 1 for j >= 32)
//   inverse step q:   X[q][:] = R[q][:] * rs_q -> G in LDS,  R -= L[:, q] X[q][:]  (R starts as I; 2 MFMAs)
// 64 + 64 short dependent steps (~120 cycles each) instead of 2 x 2016 broadcast pairs.
// ------------------------------------------------------------------------------------------------
// (a __device__ body: the kernel below runs it as one workgroup; the look-ahead chain kernel, ekf_lookahead.hpp, runs it
// twice in one launch with the carry step in between)
template <int K>
__device__ __forceinline__ void ekf_factor_mfma_f32_body(const FactorArgs<float>& a, float* __restrict__ du)
{
    static_assert(K == 32 || K == 64, "one or two 32-wide tiles per dimension");
    typedef float  T;
    struct alignas(4) float2_u // landmark rows start at odd indices: an 8-byte load with 4-byte alignment
    {
        float x, y;
    };
    constexpr int  LD = K + 1;
    __shared__ T   S[K * LD];         // S, read once into accumulators
    __shared__ T   Gm[K * LD + 128];  // X = inv(L): X[q][c] at q + c*LD; + scratch slots (lane + q)
    __shared__ T   sub[(3 + K) * LD]; // the rows of PHT that H touches: 0,1,2, then fx_o, fx_o+1 per observation
    __shared__ T   coef[(K / 2) * 10];
    __shared__ T   V[K];
    __shared__ int fxs[K / 2];
    __shared__ int sflg[2];
    const int      k   = 2 * a.m;
    const int      tid = threadIdx.x;
    auto stamp = [&](int i) {
        if (a.stamps && tid == 0)
        {
            a.stamps[i] = (long long)__builtin_readcyclecounter();
        }
    };
    stamp(0);
    const T r00 = a.R[0], r10 = a.R[1], r01 = a.R[2], r11 = a.R[3]; // requested here, used by the S sums two barriers on
    if (tid == 0)
    {
        sflg[0] = 0;
        sflg[1] = 0;
    }
    if (tid >= k && tid < K)
    {
        V[tid] = (T)0; // padding entries only: observe_model below writes [0, k) (no barrier needed in between)
    }
#pragma unroll
    for (int it = 0; it < (K * LD + 255) / 256; it++)
    {
        const int e = tid + it * 256;
        if (e < K * LD)
        {
            Gm[e] = (T)0;
        }
    }
    // The rows of PHT that H touches.  Normal path: the compact block written by the gather kernel, contiguous and
    // independent of the observation model: its loads are issued first, the observation model (its own dependent
    // loads, sqrt/atan2) runs while they are in flight, then they are parked in LDS -- one barrier for both.
    // Without the block (deferred downdates correct PHT after the gather): rows 0..2 here, the landmark rows after
    // observe_model.
    constexpr int NS = ((3 + K) * K + 255) / 256;
    T             sv[NS];
    if (a.sub != nullptr)
    {
#pragma unroll
        for (int it = 0; it < NS; it++)
        {
            const int  e    = tid + it * 256;
            const int  slot = e / K, c = e & (K - 1);
            const bool in   = (slot < 3 + k) && (c < k);
            sv[it]          = a.sub[in ? slot * k + c : 0];
        }
    }
    else if (tid < K && tid < k)
    {
        const T* p       = a.PHT + (size_t)tid * a.ldw;
        sub[0 * LD + tid] = p[0];
        sub[1 * LD + tid] = p[1];
        sub[2 * LD + tid] = p[2];
    }
    for (int o = tid; o < a.m; o += 256)
    {
        // the predicted pose when a predict() is pending (PredictArgs; the gather kernel left it in pred_out),
        // else the stored one
        const T* ps = a.pp.valid ? (a.pred_out + 2) : a.X;
        const T  px = ps[0], py = ps[1], pphi = ps[2];
        observe_model_pose<T>(a.X, a.n, a.idf[o], a.Z[2 * o], a.Z[2 * o + 1], px, py, pphi, &coef[o * 10], &V[2 * o],
                              &fxs[o]);
        a.dV[2 * o]     = V[2 * o];
        a.dV[2 * o + 1] = V[2 * o + 1];
    }
    if (a.sub != nullptr)
    {
#pragma unroll
        for (int it = 0; it < NS; it++)
        {
            const int e    = tid + it * 256;
            const int slot = e / K, c = e & (K - 1);
            if (slot < 3 + K)
            {
                sub[slot * LD + c] = sv[it];
            }
        }
    }
    __syncthreads();
    stamp(6);
    // the landmark rows: one 8-byte load per (observation, column) instead of two 4-byte loads per element of S
    if (a.sub == nullptr)
    {
        constexpr int NP = (K / 2) * K / 256; // (o, c) pairs per thread
        float2        pv[NP];
#pragma unroll
        for (int it = 0; it < NP; it++)
        {
            const int  e  = tid + it * 256;
            const int  c  = e & (K - 1);
            const int  o  = e / K;
            const bool in = (o < a.m) && (c < k);
            const int  fx = fxs[in ? o : 0];
            const float2_u q = *reinterpret_cast<const float2_u*>(a.PHT + (size_t)(in ? c : 0) * a.ldw + fx);
            pv[it]           = make_float2(q.x, q.y);
        }
#pragma unroll
        for (int it = 0; it < NP; it++)
        {
            const int e = tid + it * 256;
            const int c = e & (K - 1);
            const int o = e / K;
            sub[(3 + 2 * o) * LD + c]     = pv[it].x;
            sub[(3 + 2 * o + 1) * LD + c] = pv[it].y;
        }
    }
    __syncthreads();
    // S = H*PHT + RR (slam.h:244): 5-term sums in ascending column order; identity padding.
    // A thread's 16 elements share the row r = tid & (K-1): its H coefficients are read once.
    {
        constexpr int NE = (K * K + 255) / 256;
        const int     r   = tid & (K - 1);
        const int     ob = r >> 1, ra = r & 1;
        const bool    rin = r < k;
        const T*      cf  = &coef[(rin ? ob : 0) * 10 + ra * 5];
        const T       c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4];
        const int     s3 = (3 + 2 * (rin ? ob : 0)) * LD;
        // The thread's columns are cb, cb + CS, ...: every LDS address is a per-thread base plus a compile-time offset
        // and the R entry is the same for all of them (CS is even).  All reads are issued first, then the sums.
        // Columns >= k of `sub` hold finite or stale values that the final select discards.
        constexpr int CS = 256 / K;
        const int     cb = tid / K;
        const int     ri = ra + 2 * (cb & 1);
        const T       rv = (ri == 0) ? r00 : ((ri == 1) ? r10 : ((ri == 2) ? r01 : r11));
        const T*      sb = &sub[cb];
        const T*      sl = &sub[s3 + cb];
        T*            so = &S[r + cb * LD];
        T             p0[NE], p1[NE], p2[NE], p3[NE], p4[NE];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            p0[it] = sb[0 * LD + CS * it];
            p1[it] = sb[1 * LD + CS * it];
            p2[it] = sb[2 * LD + CS * it];
            p3[it] = sl[CS * it];
            p4[it] = sl[LD + CS * it];
        }
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int c  = cb + CS * it;
            T         sm = c0 * p0[it];
            sm += c1 * p1[it];
            sm += c2 * p2[it];
            sm += c3 * p3[it];
            sm += c4 * p4[it];
            const T vv       = sm + (((c >> 1) == ob) ? rv : (T)0);
            const T pad      = (r == c) ? (T)1 : (T)0;
            so[CS * it * LD] = (rin && c < k) ? vv : pad;
        }
    }
    __syncthreads();
    stamp(7);
    // makeSymmetric (slam.h:776-779): every thread owns 16 elements (r, c); it reads (r, c) and (c, r), then all
    // write -- (x + y) * 0.5 is the same value from both sides.  Same constant-offset addressing as above.
    {
        constexpr int NE = (K * K + 255) / 256;
        constexpr int CS = 256 / K;
        const int     r  = tid & (K - 1);
        const int     cb = tid / K;
        T*            so = &S[r + cb * LD];
        const T*      st = &S[cb + r * LD];
        T             sv[NE];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int c = cb + CS * it;
            const T   x = so[CS * it * LD], y = st[CS * it];
            sv[it]      = (r > c) ? (x + y) * (T)0.5 : ((r < c) ? (y + x) * (T)0.5 : (x + x) * (T)0.5);
        }
        __syncthreads();
        T* go = a.dS + r + (size_t)cb * k;
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int c      = cb + CS * it;
            so[CS * it * LD] = sv[it];
            if (r < k && c < k)
            {
                go[(size_t)CS * it * k] = sv[it];
            }
        }
    }
    __syncthreads();
    stamp(1);

    if (tid < 64) // ---------------- wave 0
    {
        const int lane  = tid;
        const int h     = lane >> 5;
        const int lc    = lane & 31;
        const int trash = K * LD + lane;
        f32x16    T00, T01, T11;
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            T00[r]        = S[row + lc * LD];
            if (K == 64)
            {
                T01[r] = S[row + (32 + lc) * LD];
                T11[r] = S[32 + row + (32 + lc) * LD];
            }
        }
        // Both loops are branch-free straight-line code (a failed pivot just lets NaNs run through: the flag
        // zeroes the outputs afterwards), so the accumulators stay where the MFMAs leave them and each step reads
        // one register per tile.  Measured (tools/probes/mfma_overlap_probe.hip): within ONE wave vector
        // instructions do not overlap an MFMA in flight -- a step costs (MFMAs x 64 cycles) + (pivot chain, ~80
        // cycles), whatever the order.  So the MFMA count is what is minimised: updates that the next pivot rows do
        // not need (T11 during the first 32 steps; R10 likewise in the inverse) are deferred and applied two columns
        // per MFMA (one per k-slot: the operands of a step live in one half of the wave, so the masked operands of a
        // "half 0" step and a "half 1" step simply add).
        bool failed = false;
        T    rdiag[K]; // wave-uniform 1/L[j][j]
        T    la0[32];  // column j of L, rows 0-31 (j < 32), as the masked MFMA operand; reused by the inverse
        T    la1[K];   // column j of L, rows 32-63
        auto halfof = [](int j) { return ((j & 31) >> 2) & 1; };
        auto rowreg = [](int j) { return ((j & 31) & 3) + 4 * ((j & 31) >> 3); };
        auto rowlo  = [&](int j) { return 32 * halfof(j) + (j & 31); };
        auto on_ge  = [&](int j) { return (lane >= rowlo(j)) && (lane <= 32 * halfof(j) + 31); };
        auto on_half = [&](int j) { return (lane >= 32 * halfof(j)) && (lane <= 32 * halfof(j) + 31); };
        // the 16 (half-0 step, half-1 step) pairs of 0..31: p -> (8*(p>>2) + (p&3), that + 4)
        auto pair_lo = [](int p) { return 8 * (p >> 2) + (p & 3); };
        // ---- Cholesky (right-looking, symmetric storage) ----
#pragma unroll
        for (int j = 0; j < 32; j++)
        {
            const T dj = bcast(T00[rowreg(j)], rowlo(j));
            failed     = failed || !(dj > (T)0);
            const T rs = pivot_rsqrt(dj);
            rdiag[j]   = rs;
            const T a0 = on_ge(j) ? T00[rowreg(j)] * rs : (T)0; // L[lc][j]
            la0[j]     = a0;
            T00        = __builtin_amdgcn_mfma_f32_32x32x2f32(-a0, a0, T00, 0, 0, 0);
            if (K == 64)
            {
                const T a1 = on_half(j) ? T01[rowreg(j)] * rs : (T)0; // L[32 + lc][j]
                la1[j]     = a1;
                T01        = __builtin_amdgcn_mfma_f32_32x32x2f32(-a0, a1, T01, 0, 0, 0);
            }
        }
        if (K == 64)
        {
            stamp(5);
            // T11 -= sum_{j<32} a1_j a1_j^T, two columns per MFMA
#pragma unroll
            for (int p = 0; p < 16; p++)
            {
                const T v = la1[pair_lo(p)] + la1[pair_lo(p) + 4];
                T11       = __builtin_amdgcn_mfma_f32_32x32x2f32(-v, v, T11, 0, 0, 0);
            }
#pragma unroll
            for (int j = 32; j < K; j++)
            {
                const T dj = bcast(T11[rowreg(j)], rowlo(j));
                failed     = failed || !(dj > (T)0);
                const T rs = pivot_rsqrt(dj);
                rdiag[j]   = rs;
                const T a1 = on_ge(j) ? T11[rowreg(j)] * rs : (T)0; // L[32 + lc][j]
                la1[j]     = a1;
                if (j + 1 < K)
                {
                    T11 = __builtin_amdgcn_mfma_f32_32x32x2f32(-a1, a1, T11, 0, 0, 0);
                }
            }
        }
        stamp(2);
        T chk = (T)0;
        {
        // ---- inv(L): R = I, then for every q: X[q][:] = R[q][:] / L[q][q], R -= L[:, q] X[q][:] ----
        // X[q][c] goes to Gm[q + c*LD] (the transposition for TEXTBOOK happens when G is read back); per-lane base
        // addresses, one per half: lanes outside the half point at their scratch slot.  chk turns NaN as soon as
        // one entry of X is not finite (0*Inf = NaN, 0*NaN = NaN): no compare in the loop.
        {
            f32x16 R00, R10, R11;
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                R00[r]        = (row == lc) ? (T)1 : (T)0;
                R10[r]        = (T)0;
                R11[r]        = (row == lc) ? (T)1 : (T)0;
            }
            const int gA0 = (h == 0) ? lc * LD : trash, gA1 = (h == 0) ? (32 + lc) * LD : trash;
            const int gB0 = (h == 1) ? lc * LD : trash, gB1 = (h == 1) ? (32 + lc) * LD : trash;
            T         xs[32]; // rows 0..31 of X (cols 0-31), kept for the deferred R10 update
#pragma unroll
            for (int q = 0; q < 32; q++)
            {
                const T x0 = on_half(q) ? R00[rowreg(q)] * rdiag[q] : (T)0; // X[q][lc]  (zero for lc > q)
                xs[q]      = x0;
                chk        = __builtin_fmaf(x0, (T)0, chk);
                Gm[(halfof(q) == 0 ? gA0 : gB0) + q] = x0;
                if (q + 1 < 32)
                {
                    R00 = __builtin_amdgcn_mfma_f32_32x32x2f32(-la0[q], x0, R00, 0, 0, 0);
                }
            }
            if (K == 64)
            {
                // R10 -= sum_{q<32} L[32.., q] X[q][:], two q per MFMA
#pragma unroll
                for (int p = 0; p < 16; p++)
                {
                    const T l = la1[pair_lo(p)] + la1[pair_lo(p) + 4];
                    const T x = xs[pair_lo(p)] + xs[pair_lo(p) + 4];
                    R10       = __builtin_amdgcn_mfma_f32_32x32x2f32(-l, x, R10, 0, 0, 0);
                }
#pragma unroll
                for (int q = 32; q < K; q++)
                {
                    const T x0 = on_half(q) ? R10[rowreg(q)] * rdiag[q] : (T)0; // X[q][lc]
                    const T x1 = on_half(q) ? R11[rowreg(q)] * rdiag[q] : (T)0; // X[q][32 + lc]
                    chk        = __builtin_fmaf(x0, (T)0, chk);
                    chk        = __builtin_fmaf(x1, (T)0, chk);
                    Gm[(halfof(q) == 0 ? gA0 : gB0) + q] = x0;
                    Gm[(halfof(q) == 0 ? gA1 : gB1) + q] = x1;
                    if (q + 1 < K)
                    {
                        R10 = __builtin_amdgcn_mfma_f32_32x32x2f32(-la1[q], x0, R10, 0, 0, 0);
                        R11 = __builtin_amdgcn_mfma_f32_32x32x2f32(-la1[q], x1, R11, 0, 0, 0);
                    }
                }
            }
        }
        }
        stamp(3);
        const bool bad = (__ballot(!(chk == chk)) != 0ull);
        if (lane == 0)
        {
            sflg[0] = failed ? 1 : 0;
            sflg[1] = (!failed && bad) ? 1 : 0;
        }
    }
    __syncthreads();
    stamp(8);
    const bool zero = (sflg[0] | sflg[1]) != 0;
    if (zero) // LLT failure (slam.h:421-429 handled by the host in sync mode) or a non-finite inverse: G = 0
    {
#pragma unroll
        for (int it = 0; it < (K * LD + 255) / 256; it++)
        {
            const int e = tid + it * 256;
            if (e < K * LD)
            {
                Gm[e] = (T)0;
            }
        }
        __syncthreads();
    }
    // outputs: G^T (what the gain kernel reads; the debug entry point transposes it back), t = G^T V, u = G t.
    // G[r][c]: REF_EXACT G = inv(L) = X, TEXTBOOK G = X^T
    const int gr = a.textbook ? LD : 1, gc = a.textbook ? 1 : LD;
    {
        constexpr int NE = (K * K + 255) / 256;
        constexpr int CS = 256 / K;
        const int     x  = tid & (K - 1), yb = tid / K;
        const T*      gp = &Gm[yb * gr + x * gc];
        T*            go = a.dGt + x + (size_t)yb * k;
        T             g2[NE];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            g2[it] = gp[CS * it * gr]; // G[y][x], y = yb + CS it
        }
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            if (x < k && yb + CS * it < k)
            {
                go[(size_t)CS * it * k] = g2[it]; // Gt[x][y] = G[y][x]
            }
        }
    }
    stamp(9);
    {
        // four right-hand sides at once, 64 threads each: vec 0: t = G^T V, u = G t (slam.h:258-259 regrouped);
        // vec 1..3: M[:, c] = G (G^T PHT[c, :]^T) for the pose rows c = 0..2 (the rows of `sub`), with which the gain
        // kernel forms the pose-stripe downdate PHT*M = W1*W1[0:3,:]^T without waiting for W1's pose rows
        __shared__ T t4[4][K];
        const int    o = tid & 63, vec = tid >> 6;
        T            s1 = (T)0;
        if (o < K)
        {
            const T* vin = (vec == 0) ? V : &sub[(vec - 1) * LD];
#pragma unroll 8
            for (int r = 0; r < K; r++)
            {
                const T x = (vec == 0 || r < k) ? vin[r] : (T)0; // padding rows of V are zero; those of sub are stale
                s1 += Gm[r * gr + o * gc] * x;
            }
            t4[vec][o] = (o < k) ? s1 : (T)0;
            if (vec == 0 && o < k)
            {
                a.dt[o] = s1;
            }
        }
        __syncthreads();
        T s2 = (T)0;
        if (o < K)
        {
#pragma unroll 8
            for (int c = 0; c < K; c++)
            {
                s2 += Gm[o * gr + c * gc] * t4[vec][c];
            }
            if (o < k)
            {
                if (vec == 0)
                {
                    du[o] = s2;
                }
                else if (a.dM != nullptr)
                {
                    a.dM[(vec - 1) * k + o] = s2;
                }
            }
        }
    }
    stamp(4);
    if (tid == 0)
    {
        const int code = (sflg[0] ? kFlagLltFailed : 0) | (sflg[1] ? kFlagZeroed : 0);
        a.flags[1]     = code;
        if (code)
        {
            atomicOr(&a.flags[0], code);
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256) ekf_factor_mfma_f32(FactorArgs<float> a, float* __restrict__ du)
{
    ekf_factor_mfma_f32_body<K>(a, du);
}

// ------------------------------------------------------------------------------------------------
// K2+K3 (f64) for 16 < k <= 64 on v_mfma_f64_16x16x4_f64: the f64 counterpart of ekf_factor_mfma_f32.
//
// The register-resident kernel (ekf_factor_small_kernel<double, 64>) spends 80 k cycles in the factorisation and 44 k
// in the inverse at k = 64 (two v_readlane per multiplier, 128 KB of straight-line code): 71 of a 119 us step at
// N = 1000.  Here the matrix lives in the accumulator layout of the f64 MFMA,
//     tile T[I][J] (16 x 16) = f64x4:  element g of lane l  =  M[16 I + (l>>4) + 4 g][16 J + (l&15)],
// so row j = 16 J' + jj of a tile sits in register jj>>2 of the 16 lanes with l>>4 == (jj&3), one column per lane:
// exactly an MFMA operand with the other three k-slots zero (A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]).
//   Cholesky step j : d = M[j][j] (v_readlane pair), rs = 1/sqrt(d) (v_rsq_f64 + two Newton steps), a = row_j * rs
//                     (= column j of L, masked to rows >= j) -> LDS; the tiles of block row J' are updated at once
//                     (one MFMA each, the next pivots need them); the tiles below wait until the 16 columns of the
//                     block are done and then take them four per MFMA (the operands of steps with different jj&3 occupy
//                     different k-slots: they simply add).
//   inverse         : R = I; step q: X[q][:] = R[q][:] * rs_q -> LDS (final orientation of G), R -= L[:, q] X[q][:]
//                     with the same immediate / deferred split.  200 + 200 MFMAs instead of 2 x 2016 broadcast pairs.
// Only the upper tiles T[I][J], I <= J, of the symmetric matrix are kept (row j of the upper part = column j of the
// lower one); R is lower triangular: tiles I >= J.  Same inputs / outputs / flags as ekf_factor_small_kernel; dM is
// produced as in ekf_factor_mfma_f32.  LDS (dynamic): S | sub (later: the L columns) | small arrays.
// ------------------------------------------------------------------------------------------------
// compile-time loop: the index is an integral_constant, so register arrays indexed with it never fall back to scratch
// memory when the optimiser declines to unroll a large `#pragma unroll` loop
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E)
    {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ inline double rsqrt_f64(double d)
{
    double       y = __builtin_amdgcn_rsq(d); // ~26 bits
    const double h = 0.5 * d;
    y              = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    y              = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    return y;
}

// (a __device__ body: the kernel below runs it as one workgroup; the look-ahead chain kernel, ekf_lookahead.hpp, runs it
// twice in one launch with the carry step in between)
template <int K>
__device__ __forceinline__ void ekf_factor_mfma_f64_body(const FactorArgs<double>& a, double* __restrict__ du)
{
    static_assert(K == 32 || K == 64, "two or four 16-wide tiles per dimension");
    typedef double T;
    constexpr int  NB = K / 16;
    constexpr int  LD = K + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f64[];
    T*   S    = reinterpret_cast<T*>(smem_f64);  // K x LD: S, then G in its final orientation
    T*   sub  = S + K * LD;                      // (3 + K) x LD: rows of PHT that H touches; then Lm[r + j*LD] = L[r][j]
    T*   Lm   = sub;
    T*   coef = sub + (3 + K) * LD;              // (K/2) x 10
    T*   V    = coef + (K / 2) * 10;             // K
    T*   rd   = V + K;                           // K: 1 / L[j][j]
    T*   t4   = rd + K;                          // 4 x K
    int* fxs  = reinterpret_cast<int*>(t4 + 4 * K); // K/2
    int* sflg = fxs + K / 2;                     // 2 flags + [2]: 16-column blocks of L that wave 0 has published
    const int k   = 2 * a.m;
    const int tid = threadIdx.x;
    auto stamp = [&](int i) {
        if (a.stamps && tid == 0)
        {
            a.stamps[i] = (long long)__builtin_readcyclecounter();
        }
    };
    stamp(0);
    const T   r00 = a.R[0], r10 = a.R[1], r01 = a.R[2], r11 = a.R[3];
    if (tid == 0)
    {
        sflg[0] = 0;
        sflg[1] = 0;
        sflg[2] = 0;
    }
    if (tid < K)
    {
        V[tid] = (T)0;
    }
    __syncthreads();
    // The rows of PHT that H touches -> sub[slot*LD + c]: slots 0..2 = rows 0..2, 3+2o+a = row fx_o + a.  With the
    // compact block of the gather kernel their loads do not depend on the observation model: they are issued first
    // and the model (its own dependent loads, sqrt / atan2 in f64) runs while they are in flight.
    constexpr int NS = ((3 + K) * K + 255) / 256;
    T             sv[NS];
    const int     tot = (3 + k) * k;
    if (a.sub != nullptr)
    {
#pragma unroll
        for (int it = 0; it < NS; it++)
        {
            const int e = tid + it * 256;
            sv[it]      = a.sub[(e < tot) ? e : 0];
        }
    }
    for (int o = tid; o < a.m; o += 256)
    {
        // the predicted pose when a predict() is pending (the gather kernel left it in pred_out), else the stored one
        const T* ps = a.pp.valid ? (a.pred_out + 2) : a.X;
        observe_model_pose<T>(a.X, a.n, a.idf[o], a.Z[2 * o], a.Z[2 * o + 1], ps[0], ps[1], ps[2], &coef[o * 10], &V[2 * o],
                              &fxs[o]);
        a.dV[2 * o]     = V[2 * o];
        a.dV[2 * o + 1] = V[2 * o + 1];
    }
    if (a.sub == nullptr)
    {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NS; it++)
        {
            const int e  = tid + it * 256;
            const int ec = (e < tot) ? e : 0;
            const int c = ec / (3 + k), slot = ec - c * (3 + k); // consecutive threads walk down one column of PHT
            const int row = (slot < 3) ? slot : fxs[(slot - 3) >> 1] + ((slot - 3) & 1);
            sv[it]        = a.PHT[(size_t)c * a.ldw + row];
        }
    }
#pragma unroll
    for (int it = 0; it < NS; it++)
    {
        const int e = tid + it * 256;
        if (e < tot)
        {
            int slot, c;
            if (a.sub != nullptr)
            {
                slot = e / k;
                c    = e - slot * k;
            }
            else
            {
                c    = e / (3 + k);
                slot = e - c * (3 + k);
            }
            sub[slot * LD + c] = sv[it];
        }
    }
    __syncthreads();
    stamp(6);
    // S = H*PHT + RR (slam.h:244): 5-term sums in ascending column order; identity padding
#pragma unroll
    for (int e = tid; e < K * K; e += 256)
    {
        const int r = e & (K - 1), c = e / K;
        T         v;
        if (r < k && c < k)
        {
            const int ob = r >> 1, ra = r & 1;
            const T*  cf = &coef[ob * 10 + ra * 5];
            T         sm = cf[0] * sub[0 * LD + c];
            sm += cf[1] * sub[1 * LD + c];
            sm += cf[2] * sub[2 * LD + c];
            sm += cf[3] * sub[(3 + 2 * ob) * LD + c];
            sm += cf[4] * sub[(4 + 2 * ob) * LD + c];
            const int ri = ra + 2 * (c & 1);
            const T   rv = (ri == 0) ? r00 : ((ri == 1) ? r10 : ((ri == 2) ? r01 : r11));
            v            = sm + (((c >> 1) == ob) ? rv : (T)0);
        }
        else
        {
            v = (r == c) ? (T)1 : (T)0;
        }
        S[r + c * LD] = v;
    }
    __syncthreads();
    stamp(7);
    // makeSymmetric (slam.h:776-779): every thread reads (r, c) and (c, r) of its elements, then all write --
    // (x + y) * 0.5 is the same value from both sides
    {
        constexpr int NE = K * K / 256;
        T             sv[NE];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e = tid + it * 256;
            const int r = e & (K - 1), c = e / K;
            const T   x = S[r + c * LD], y = S[c + r * LD];
            sv[it]      = (r > c) ? (x + y) * (T)0.5 : ((r < c) ? (y + x) * (T)0.5 : (x + x) * (T)0.5);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e = tid + it * 256;
            const int r = e & (K - 1), c = e / K;
            S[r + c * LD] = sv[it];
            if (r < k && c < k)
            {
                a.dS[r + c * k] = sv[it];
            }
        }
    }
    // pose rows of PHT (sub rows 0..2) are needed again at the end: park them in t4[1..3] before sub becomes Lm
    if (tid < 3 * K)
    {
        const int c = tid / K, q = tid - c * K;
        t4[(1 + c) * K + q] = (q < k) ? sub[c * LD + q] : (T)0;
    }
    __syncthreads();
    const int lane = tid & 63;
    const int lj = lane & 15, lq = lane >> 4;
    f64x4     Tt[NB][NB]; // upper tiles (I <= J) of S, then (as Rt) lower tiles (I >= J) of the inverse recurrence
    if (tid < 64)
    {
#pragma unroll
        for (int I = 0; I < NB; I++)
        {
#pragma unroll
            for (int J = I; J < NB; J++)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    Tt[I][J][g] = S[(16 * I + lq + 4 * g) + (16 * J + lj) * LD];
                }
            }
        }
    }
    else // the other three waves clear the L-column store (sub is dead)
    {
        for (int e = tid - 64; e < K * LD; e += 192)
        {
            Lm[e] = (T)0;
        }
    }
    __syncthreads(); // #A: T is in registers, Lm is clear
    stamp(1);
    bool failed = false;
    if (tid < 64)
    {
        // ---- Cholesky (right-looking; the matrix stays symmetric, only upper tiles are touched) ----
#pragma unroll
        for (int J = 0; J < NB; J++)
        {
            T A4[NB][4]; // per block row I > J: the operands of this block's 16 columns, four columns per entry
#pragma unroll
            for (int I = 0; I < NB; I++)
            {
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    A4[I][t] = (T)0;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 16; jj++)
            {
                const int j    = 16 * J + jj;
                const int g    = jj >> 2, slot = jj & 3;
                const T   dj   = bcast(Tt[J][J][g], 16 * slot + jj);
                failed         = failed || !(dj > (T)0);
                const T    rs  = rsqrt_f64(dj);
                const bool on  = (lq == slot);
                T          av[NB];
#pragma unroll
                for (int Jc = J; Jc < NB; Jc++)
                {
                    const bool keep = on && (Jc > J || lj >= jj);
                    av[Jc]          = keep ? Tt[J][Jc][g] * rs : (T)0; // L[16 Jc + lj][j]
                    if (on)
                    {
                        Lm[(16 * Jc + lj) + j * LD] = av[Jc];
                    }
                    if (Jc > J)
                    {
                        A4[Jc][g] += av[Jc];
                    }
                }
                if (lane == 0)
                {
                    rd[j] = rs;
                }
                if (jj < 15) // (after its last column the rows of block J are never read again)
                {
#pragma unroll
                    for (int Jc = J; Jc < NB; Jc++) // block row J: needed by the next pivots
                    {
                        Tt[J][Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[J], av[Jc], Tt[J][Jc], 0, 0, 0);
                    }
                }
            }
            // the 16 columns of L of this block (and their 1/diag) are complete: the inverse (wave 1) may take them
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the LDS stores above have been performed
            if (lane == 0)
            {
                __hip_atomic_store(&sflg[2], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // the tiles below block row J: the block's 16 columns, four per MFMA
#pragma unroll
            for (int I = J + 1; I < NB; I++)
            {
#pragma unroll
                for (int Jc = I; Jc < NB; Jc++)
                {
#pragma unroll
                    for (int t = 0; t < 4; t++)
                    {
                        Tt[I][Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(-A4[I][t], A4[Jc][t], Tt[I][Jc], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (tid == 0)
    {
        sflg[0] = failed ? 1 : 0;
        stamp(2);
    }
    // ---- wave 1: the inverse, one 16-column block behind the factorisation (it needs L[:, 16Q..16Q+15] and the
    //      1/diag of block Q, nothing later); waves 2 and 3 wait at the barrier below
    T chk = (T)0;
    if (tid >= 64 && tid < 128)
    {
        for (int e = lane; e < K * LD; e += 64) // S is in wave 0's registers: clear it for G
        {
            S[e] = (T)0;
        }
        // ---- inv(L): R = I; q ascending: X[q][:] = R[q][:] / L[q][q]; R -= L[:, q] X[q][:] ----
        // G[r][c]: REF_EXACT G = inv(L) = X, TEXTBOOK G = X^T; S[r + c*LD] = G[r][c]
        const int sq = a.textbook ? LD : 1, sc = a.textbook ? 1 : LD; // X[q][c] -> S[q*sq + c*sc]
#pragma unroll
        for (int I = 0; I < NB; I++)
        {
#pragma unroll
            for (int J = 0; J <= I; J++)
            {
#pragma unroll
                for (int g = 0; g < 4; g++)
                {
                    Tt[I][J][g] = (I == J && (lq + 4 * g) == lj) ? (T)1 : (T)0;
                }
            }
        }
#pragma unroll
        for (int Q = 0; Q < NB; Q++)
        {
            while (__hip_atomic_load(&sflg[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= Q)
            {
                __builtin_amdgcn_s_sleep(2);
            }
            T X4[NB][4];
#pragma unroll
            for (int Jc = 0; Jc < NB; Jc++)
            {
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    X4[Jc][t] = (T)0;
                }
            }
            // the block's 1/diag values and L operands are read up front: the LDS stores of the X rows below would
            // otherwise fence every read behind them (everything here is carved out of one LDS allocation)
            T rsv[16], lqv[16];
#pragma unroll
            for (int qq = 0; qq < 16; qq++)
            {
                rsv[qq] = rd[16 * Q + qq];
                lqv[qq] = Lm[(16 * Q + lj) + (16 * Q + qq) * LD];
            }
#pragma unroll
            for (int qq = 0; qq < 16; qq++)
            {
                const int  q  = 16 * Q + qq;
                const int  g  = qq >> 2, slot = qq & 3;
                const bool on = (lq == slot);
                const T    rs = rsv[qq];
                const T    lq_ = on ? lqv[qq] : (T)0; // L[16 Q + lj][q] (zero above the diagonal)
                T          xv[NB];
#pragma unroll
                for (int Jc = 0; Jc <= Q; Jc++)
                {
                    xv[Jc] = on ? Tt[Q][Jc][g] * rs : (T)0; // X[q][16 Jc + lj]
                    chk    = __builtin_fma(xv[Jc], (T)0, chk);
                    if (on)
                    {
                        S[q * sq + (16 * Jc + lj) * sc] = xv[Jc];
                    }
                    X4[Jc][g] += xv[Jc];
                }
                if (qq < 15)
                {
#pragma unroll
                    for (int Jc = 0; Jc <= Q; Jc++)
                    {
                        Tt[Q][Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(-lq_, xv[Jc], Tt[Q][Jc], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int I = Q + 1; I < NB; I++)
            {
#pragma unroll
                for (int t = 0; t < 4; t++)
                {
                    const T li = Lm[(16 * I + lj) + (16 * Q + 4 * t + lq) * LD]; // L[16 I + lj][16 Q + 4 t + lq]
#pragma unroll
                    for (int Jc = 0; Jc <= Q; Jc++)
                    {
                        Tt[I][Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(-li, X4[Jc][t], Tt[I][Jc], 0, 0, 0);
                    }
                }
            }
        }
        const bool bad = (__ballot(!(chk == chk)) != 0ull);
        if (lane == 0)
        {
            sflg[1] = bad ? 1 : 0;
        }
    }
    __syncthreads(); // #C
    stamp(3);
    stamp(8);
    if (tid == 0 && sflg[0])
    {
        sflg[1] = 0; // (a failed factorisation is reported as such, not as a non-finite inverse)
    }
    __syncthreads();
    const bool zero = (sflg[0] | sflg[1]) != 0;
    if (zero) // LLT failure (slam.h:421-429 handled by the host in sync mode) or a non-finite inverse: G = 0
    {
        for (int e = tid; e < K * LD; e += 256)
        {
            S[e] = (T)0;
        }
        __syncthreads();
    }
    // outputs: G, G^T, then t = G^T V, u = G t and M (see ekf_factor_mfma_f32)
    // (only G^T is published: it is what the gain kernel reads; the debug entry point transposes it back)
#pragma unroll
    for (int e = tid; e < K * K; e += 256)
    {
        const int c = e & (K - 1), r = e / K;
        if (r < k && c < k)
        {
            a.dGt[c + r * k] = S[r + c * LD];
        }
    }
    stamp(9);
    {
        const int o = tid & 63, vec = tid >> 6;
        T         s1 = (T)0;
        if (o < K)
        {
            const T* vin = (vec == 0) ? V : &t4[vec * K];
#pragma unroll 16
            for (int r = 0; r < K; r++)
            {
                s1 += S[r + o * LD] * vin[r]; // (G^T x)[o]; padding entries of the inputs are zero
            }
        }
        __syncthreads(); // (t4[1..3] are inputs above and outputs below)
        if (o < K)
        {
            t4[vec * K + o] = (o < k) ? s1 : (T)0;
            if (vec == 0 && o < k)
            {
                a.dt[o] = s1;
            }
        }
        __syncthreads();
        if (o < K)
        {
            T s2 = (T)0;
#pragma unroll 16
            for (int c = 0; c < K; c++)
            {
                s2 += S[o + c * LD] * t4[vec * K + c]; // (G t)[o]
            }
            if (o < k)
            {
                if (vec == 0)
                {
                    du[o] = s2;
                }
                else if (a.dM != nullptr)
                {
                    a.dM[(vec - 1) * k + o] = s2;
                }
            }
        }
    }
    stamp(4);
    if (tid == 0)
    {
        const int code = (sflg[0] ? kFlagLltFailed : 0) | (sflg[1] ? kFlagZeroed : 0);
        a.flags[1]     = code;
        if (code)
        {
            atomicOr(&a.flags[0], code);
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256) ekf_factor_mfma_f64(FactorArgs<double> a, double* __restrict__ du)
{
    ekf_factor_mfma_f64_body<K>(a, du);
}

// ------------------------------------------------------------------------------------------------
// K2+K3 (f32) for 64 < k <= 128 on v_mfma_f32_32x32x2_f32: the structure of ekf_factor_mfma_f64 (upper tiles only,
// block-row updates at once, the tiles below a block four... here SIXTEEN pairs of columns per block and two columns
// per MFMA, the inverse on a second wave one block behind) with 32 x 32 tiles: NB = K/32 = 4 blocks, 10 tiles of 16
// VGPRs.  Tile element r of lane l = M[32 I + (r&3) + 8 (r>>2) + 4 (l>>5)][32 J + (l&31)]; row jj of a tile sits in
// register (jj&3) + 4 (jj>>3) of the lanes with l>>5 == (jj>>2)&1.  Batches of 33..64 observations used to fall back to
// the workgroup-parallel kernel (ekf_factor_par_kernel<float, 128>).  LDS (dynamic, 139 KB): S | sub / L | small arrays.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) ekf_factor_mfma_big_f32(FactorArgs<float> a, float* __restrict__ du)
{
    static_assert(K == 64 || K == 128, "two or four 32-wide tiles per dimension");
    typedef float T;
    constexpr int NB = K / 32;
    constexpr int LD = K + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
    T*   S    = reinterpret_cast<T*>(smem_big);
    T*   sub  = S + K * LD;
    T*   Lm   = sub;
    T*   coef = sub + (3 + K) * LD;
    T*   V    = coef + (K / 2) * 10;
    T*   rd   = V + K;
    T*   t4   = rd + K;
    int* fxs  = reinterpret_cast<int*>(t4 + 4 * K);
    int* sflg = fxs + K / 2;
    const int k   = 2 * a.m;
    const int tid = threadIdx.x;
    auto stamp = [&](int i) {
        if (a.stamps && tid == 0)
        {
            a.stamps[i] = (long long)__builtin_readcyclecounter();
        }
    };
    stamp(0);
    const T   r00 = a.R[0], r10 = a.R[1], r01 = a.R[2], r11 = a.R[3];
    if (tid == 0)
    {
        sflg[0] = 0;
        sflg[1] = 0;
        sflg[2] = 0;
    }
    if (tid < K)
    {
        V[tid] = (T)0;
    }
    __syncthreads();
    for (int o = tid; o < a.m; o += 256)
    {
        observe_model<T>(a.X, a.n, a.idf[o], a.Z[2 * o], a.Z[2 * o + 1], &coef[o * 10], &V[2 * o], &fxs[o]);
        a.dV[2 * o]     = V[2 * o];
        a.dV[2 * o + 1] = V[2 * o + 1];
    }
    __syncthreads();
    // the rows of PHT that H touches (rows 0..2, then the two rows of every observed landmark), straight from PHT:
    // consecutive threads walk down one column
    for (int e0 = 0; e0 < (3 + k) * k; e0 += 256 * 24)
    {
        T v8[24];
#pragma unroll
        for (int u = 0; u < 24; u++)
        {
            const int e  = e0 + u * 256 + tid;
            const int ec = (e < (3 + k) * k) ? e : 0;
            const int c = ec / (3 + k), slot = ec - c * (3 + k);
            const int row = (slot < 3) ? slot : fxs[(slot - 3) >> 1] + ((slot - 3) & 1);
            v8[u]         = a.PHT[(size_t)c * a.ldw + row];
        }
#pragma unroll
        for (int u = 0; u < 24; u++)
        {
            const int e = e0 + u * 256 + tid;
            if (e < (3 + k) * k)
            {
                const int c = e / (3 + k), slot = e - c * (3 + k);
                sub[slot * LD + c] = v8[u];
            }
        }
    }
    __syncthreads();
    stamp(6);
    // S = H*PHT + RR (slam.h:244): 5-term sums in ascending column order; identity padding
#pragma unroll 8
    for (int e = tid; e < K * K; e += 256)
    {
        const int r = e & (K - 1), c = e / K;
        T         v;
        if (r < k && c < k)
        {
            const int ob = r >> 1, ra = r & 1;
            const T*  cf = &coef[ob * 10 + ra * 5];
            T         sm = cf[0] * sub[0 * LD + c];
            sm += cf[1] * sub[1 * LD + c];
            sm += cf[2] * sub[2 * LD + c];
            sm += cf[3] * sub[(3 + 2 * ob) * LD + c];
            sm += cf[4] * sub[(4 + 2 * ob) * LD + c];
            const int ri = ra + 2 * (c & 1);
            const T   rv = (ri == 0) ? r00 : ((ri == 1) ? r10 : ((ri == 2) ? r01 : r11));
            v            = sm + (((c >> 1) == ob) ? rv : (T)0);
        }
        else
        {
            v = (r == c) ? (T)1 : (T)0;
        }
        S[r + c * LD] = v;
    }
    __syncthreads();
    stamp(7);
    // makeSymmetric (slam.h:776-779): every thread reads (r, c) and (c, r) of ALL its elements, then all write
    {
        constexpr int NE = K * K / 256;
        T             sv[NE];
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e = tid + it * 256;
            const int r = e & (K - 1), c = e / K;
            const T   x = S[r + c * LD], y = S[c + r * LD];
            sv[it]      = (r > c) ? (x + y) * (T)0.5 : ((r < c) ? (y + x) * (T)0.5 : (x + x) * (T)0.5);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NE; it++)
        {
            const int e = tid + it * 256;
            const int r = e & (K - 1), c = e / K;
            S[r + c * LD] = sv[it];
            if (r < k && c < k)
            {
                a.dS[r + c * k] = sv[it];
            }
        }
    }
    for (int e = tid; e < 3 * K; e += 256) // pose rows of PHT for M, before sub becomes the L store
    {
        const int c = e / K, q = e - c * K;
        t4[(1 + c) * K + q] = (q < k) ? sub[c * LD + q] : (T)0;
    }
    __syncthreads();
    const int lane = tid & 63;
    const int lj = lane & 31, lh = lane >> 5;
    f32x16    Tt[NB][NB];
    if (tid < 64)
    {
#pragma unroll
        for (int I = 0; I < NB; I++)
        {
#pragma unroll
            for (int J = I; J < NB; J++)
            {
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    Tt[I][J][r] = S[(32 * I + (r & 3) + 8 * (r >> 2) + 4 * lh) + (32 * J + lj) * LD];
                }
            }
        }
    }
    else
    {
        for (int e = tid - 64; e < K * LD; e += 192)
        {
            Lm[e] = (T)0;
        }
    }
    __syncthreads(); // #A
    stamp(1);
    bool failed = false;
    if (tid < 64)
    {
        static_for<0, NB>([&](auto Jt) {
            constexpr int J = decltype(Jt)::value;
            T A4[NB][16]; // per block row I > J: the block's 32 columns, two per entry (the pair shares a register index)
#pragma unroll
            for (int I = 0; I < NB; I++)
            {
#pragma unroll
                for (int t = 0; t < 16; t++)
                {
                    A4[I][t] = (T)0;
                }
            }
            static_for<0, 32>([&](auto jt) {
                constexpr int jj = decltype(jt)::value;
                constexpr int j  = 32 * J + jj;
                constexpr int rg = (jj & 3) + 4 * (jj >> 3), hf = (jj >> 2) & 1;
                const T    dj = bcast(Tt[J][J][rg], 32 * hf + jj);
                failed        = failed || !(dj > (T)0);
                const T    rs = pivot_rsqrt(dj);
                const bool on = (lh == hf);
                T          av[NB];
#pragma unroll
                for (int Jc = J; Jc < NB; Jc++)
                {
                    const bool keep = on && (Jc > J || lj >= jj);
                    av[Jc]          = keep ? Tt[J][Jc][rg] * rs : (T)0; // L[32 Jc + lj][j]
                    if (on)
                    {
                        Lm[(32 * Jc + lj) + j * LD] = av[Jc];
                    }
                    if (Jc > J)
                    {
                        A4[Jc][rg] += av[Jc];
                    }
                }
                if (lane == 0)
                {
                    rd[j] = rs;
                }
                if (jj < 31)
                {
#pragma unroll
                    for (int Jc = J; Jc < NB; Jc++)
                    {
                        Tt[J][Jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av[J], av[Jc], Tt[J][Jc], 0, 0, 0);
                    }
                }
            });
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0)
            {
                __hip_atomic_store(&sflg[2], J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int I = J + 1; I < NB; I++)
            {
#pragma unroll
                for (int Jc = I; Jc < NB; Jc++)
                {
#pragma unroll
                    for (int t = 0; t < 16; t++)
                    {
                        Tt[I][Jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(-A4[I][t], A4[Jc][t], Tt[I][Jc], 0, 0, 0);
                    }
                }
            }
        });
    }
    if (tid == 0)
    {
        sflg[0] = failed ? 1 : 0;
        stamp(2);
    }
    T chk = (T)0;
    if (tid >= 64 && tid < 128) // ---- wave 1: the inverse, one 32-column block behind
    {
        for (int e = lane; e < K * LD; e += 64)
        {
            S[e] = (T)0;
        }
        const int sq = a.textbook ? LD : 1, sc = a.textbook ? 1 : LD; // X[q][c] -> S[q*sq + c*sc]
#pragma unroll
        for (int I = 0; I < NB; I++)
        {
#pragma unroll
            for (int J = 0; J <= I; J++)
            {
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    Tt[I][J][r] = (I == J && ((r & 3) + 8 * (r >> 2) + 4 * lh) == lj) ? (T)1 : (T)0;
                }
            }
        }
        static_for<0, NB>([&](auto Qt) {
            constexpr int Q = decltype(Qt)::value;
            while (__hip_atomic_load(&sflg[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= Q)
            {
                __builtin_amdgcn_s_sleep(2);
            }
            T X4[NB][16];
#pragma unroll
            for (int Jc = 0; Jc < NB; Jc++)
            {
#pragma unroll
                for (int t = 0; t < 16; t++)
                {
                    X4[Jc][t] = (T)0;
                }
            }
            T rsv[32], lqv[32];
#pragma unroll
            for (int qq = 0; qq < 32; qq++)
            {
                rsv[qq] = rd[32 * Q + qq];
                lqv[qq] = Lm[(32 * Q + lj) + (32 * Q + qq) * LD];
            }
            static_for<0, 32>([&](auto qt) {
                constexpr int qq = decltype(qt)::value;
                constexpr int q  = 32 * Q + qq;
                constexpr int rg = (qq & 3) + 4 * (qq >> 3), hf = (qq >> 2) & 1;
                const bool on = (lh == hf);
                const T    rs = rsv[qq];
                const T    lq_ = on ? lqv[qq] : (T)0;
                T          xv[NB];
#pragma unroll
                for (int Jc = 0; Jc <= Q; Jc++)
                {
                    xv[Jc] = on ? Tt[Q][Jc][rg] * rs : (T)0; // X[q][32 Jc + lj]
                    chk    = __builtin_fmaf(xv[Jc], (T)0, chk);
                    if (on)
                    {
                        S[q * sq + (32 * Jc + lj) * sc] = xv[Jc];
                    }
                    X4[Jc][rg] += xv[Jc];
                }
                if (qq < 31)
                {
#pragma unroll
                    for (int Jc = 0; Jc <= Q; Jc++)
                    {
                        Tt[Q][Jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(-lq_, xv[Jc], Tt[Q][Jc], 0, 0, 0);
                    }
                }
            });
#pragma unroll
            for (int I = Q + 1; I < NB; I++)
            {
#pragma unroll
                for (int t = 0; t < 16; t++)
                {
                    // group t = the two columns whose row register is t: qq = (t&3) + 8 (t>>2) + 4 lh
                    const T li = Lm[(32 * I + lj) + (32 * Q + (t & 3) + 8 * (t >> 2) + 4 * lh) * LD];
#pragma unroll
                    for (int Jc = 0; Jc <= Q; Jc++)
                    {
                        Tt[I][Jc] = __builtin_amdgcn_mfma_f32_32x32x2f32(-li, X4[Jc][t], Tt[I][Jc], 0, 0, 0);
                    }
                }
            }
        });
        const bool bad = (__ballot(!(chk == chk)) != 0ull);
        if (lane == 0)
        {
            sflg[1] = bad ? 1 : 0;
        }
    }
    __syncthreads(); // #C
    stamp(3);
    stamp(8);
    if (tid == 0 && sflg[0])
    {
        sflg[1] = 0;
    }
    __syncthreads();
    const bool zero = (sflg[0] | sflg[1]) != 0;
    if (zero)
    {
        for (int e = tid; e < K * LD; e += 256)
        {
            S[e] = (T)0;
        }
        __syncthreads();
    }
    // outputs: G^T (what the gain kernel reads), t = G^T V, u = G t, M
#pragma unroll 8
    for (int e = tid; e < K * K; e += 256)
    {
        const int c = e & (K - 1), r = e / K;
        if (r < k && c < k)
        {
            a.dGt[c + r * k] = S[r + c * LD];
        }
    }
    stamp(9);
    for (int vec = 0; vec < 4; vec++) // (K = 128: two outputs per thread pair: 256 threads = 128 outputs x 2 parts)
    {
        const int o = tid >> 1, part = tid & 1;
        const T*  vin = (vec == 0) ? V : &t4[vec * K];
        T         s1 = (T)0;
        if (o < K)
        {
#pragma unroll 16
            for (int r = part; r < K; r += 2)
            {
                s1 += S[r + o * LD] * vin[r];
            }
        }
        s1 += __shfl_xor(s1, 1);
        __syncthreads();
        if (part == 0 && o < K)
        {
            t4[vec * K + o] = (o < k) ? s1 : (T)0;
            if (vec == 0 && o < k)
            {
                a.dt[o] = s1;
            }
        }
        __syncthreads();
        T s2 = (T)0;
        if (o < K)
        {
#pragma unroll 16
            for (int c = part; c < K; c += 2)
            {
                s2 += S[o + c * LD] * t4[vec * K + c];
            }
        }
        s2 += __shfl_xor(s2, 1);
        if (part == 0 && o < k)
        {
            if (vec == 0)
            {
                du[o] = s2;
            }
            else if (a.dM != nullptr)
            {
                a.dM[(vec - 1) * k + o] = s2;
            }
        }
    }
    stamp(4);
    if (tid == 0)
    {
        const int code = (sflg[0] ? kFlagLltFailed : 0) | (sflg[1] ? kFlagZeroed : 0);
        a.flags[1]     = code;
        if (code)
        {
            atomicOr(&a.flags[0], code);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4 (f32) on MFMA: W1 = PHT * G (slam.h:257), X += PHT * u with u = G*(G^T V) (slam.h:258-259 regrouped).
// Tile = 32 rows x 32 columns per wave.  D[i][j]: i <-> output column, j <-> row (contiguous across lanes).
//   A[i = lane&31][kq = lane>>5] = G[q0+kq][c0+i]   (read from Gt, contiguous across lanes)
//   B[kq][j]                     = PHT[row0 + j][q0+kq]
// PHT rows >= n are zero (never written), so W1's padding rows come out zero as the downdate needs.
// grid = (n_pad/32, ceil(k/32)) single-wave workgroups; column tile 0 also updates X.
// ------------------------------------------------------------------------------------------------
// Generic form: OUT[i, c] (= or -=) sum_q A[i, q] * Bt[q*ldb + c],  i < n_pad rows, c < nc columns, q < kq.
//   gain:        A = PHT, kq = k,  Bt = G^T (ldb = k), OUT = W1 slot,       XUPD: X += A * u
//   correction:  A = Wp,  kq = kp, Bt = Y   (ldb = k), OUT = PHT, SUB (PHT -= Wp * Y^T, deferred downdates)
template <bool SUB, bool XUPD>
__global__ void __launch_bounds__(64) ekf_panel_mfma_f32(const float* __restrict__ A, int lda, int n, int kq, int nc,
                                                          const float* __restrict__ Bt, int ldb,
                                                          const float* __restrict__ u, float* __restrict__ OUT, int ldo,
                                                          float* __restrict__ X, const float* __restrict__ pred = nullptr,
                                                          int pred_w = 0, float* __restrict__ P = nullptr, int ldp = 0,
                                                          const float* __restrict__ Mv = nullptr,
                                                          float* __restrict__ wv_out = nullptr)
{
    // (P here is the pose stripe Pv, see p_get: the predicted stripe and pose block are committed there)
    // Mv != nullptr (gain, XUPD): the factor kernel supplied M = G G^T PHT[0:3,:]^T (3 x kq): the column-tile-0
    // workgroups also apply the pose-stripe share of slam.h:260, Pv[:, c] -= PHT * M[c, :]^T (= W1 * W1[c, :]^T), and the
    // pose rows of the W1 panel are written as ZERO (saved to wv_out, 3 x nc, for the debug entry point): the pending-
    // panel algebra never touches the stripe.
    // pred != nullptr (gain, XUPD): a predict() was applied on the fly by the gather and factor kernels (PredictArgs);
    // this kernel commits it: the column-tile-0 workgroups write the predicted stripe and Pvv into P and add the state
    // correction to the PREDICTED pose.  pred = {g02, g12, pose (3), Pvv (9)} from the factor kernel.
    // one wave per workgroup, one 32x32 MFMA tile per wave: grid = (n_pad/32 row tiles, ceil(nc/32) column tiles).
    // Small tiles on purpose: the kernel is a latency chain (load -> MFMA -> store), so it wants many waves.
    const int  lane = threadIdx.x;
    const int  lj   = lane & 31;
    const int  lh   = lane >> 5;
    const int  row0 = blockIdx.x * 32;
    const int  c0   = blockIdx.y * 32;
    const bool cok  = (c0 + lj) < nc;
    const int  cc   = cok ? (c0 + lj) : (nc - 1); // clamped: loads are unconditional, the VALUE is selected
    const bool dox  = XUPD && (blockIdx.y == 0);
    f32x16     acc  = {0};
    float      xs   = 0.f, xm0 = 0.f, xm1 = 0.f, xm2 = 0.f;
    const float* pa = A + row0 + lj;
    // SUB: the old values of the output tile are requested first, so that their latency (the tile was written by
    // another kernel on other XCDs a moment ago) hides behind the operand loads and the MFMAs
    float oldv[16];
    if (SUB)
    {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const int col = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            oldv[r]       = OUT[(size_t)(col < nc ? col : nc - 1) * ldo + row0 + lj];
        }
    }
    constexpr int NP = 32; // k-pairs per trip: every load of the trip is issued before its first MFMA
    for (int qb = 0; qb < kq; qb += 2 * NP)
    {
        float b[NP], g[NP], uq[NP];
#pragma unroll
        for (int t = 0; t < NP; t++)
        {
            const int q  = qb + 2 * t + lh;
            const int qc = (q < kq) ? q : (kq - 1);
            b[t]         = pa[(size_t)qc * lda];
            g[t]         = Bt[(size_t)qc * ldb + cc];
            uq[t]        = XUPD ? u[qc] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NP; t++)
        {
            const int   q  = qb + 2 * t + lh;
            const bool  ok = q < kq;
            const float gg = (ok && cok) ? g[t] : 0.f;
            const float bb = ok ? b[t] : 0.f;
            acc            = __builtin_amdgcn_mfma_f32_32x32x2f32(gg, bb, acc, 0, 0, 0);
            xs += bb * uq[t];
        }
        if (XUPD && dox && Mv != nullptr) // (workgroup-uniform)
        {
#pragma unroll
            for (int t = 0; t < NP; t++)
            {
                const int   q  = qb + 2 * t + lh;
                const int   qc = (q < kq) ? q : (kq - 1);
                const float bb = (q < kq) ? b[t] : 0.f;
                xm0 += bb * Mv[qc];
                xm1 += bb * Mv[kq + qc];
                xm2 += bb * Mv[2 * kq + qc];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        const int col = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (col < nc)
        {
            float* o = OUT + (size_t)col * ldo + row0 + lj;
            if (!SUB && Mv != nullptr && row0 + lj < 3) // pose rows of the W1 panel: kept aside, stored as zero
            {
                wv_out[(size_t)(row0 + lj) * nc + col] = acc[r];
                *o                                     = 0.f;
            }
            else
            {
                *o = SUB ? (oldv[r] - acc[r]) : acc[r];
            }
        }
    }
    if (dox)
    {
        xs += __shfl_xor(xs, 32);
        xm0 += __shfl_xor(xm0, 32);
        xm1 += __shfl_xor(xm1, 32);
        xm2 += __shfl_xor(xm2, 32);
        const int r = row0 + lj;
        if (lh == 0 && r < n)
        {
            X[r] = ((pred != nullptr && r < 3) ? pred[2 + r] : X[r]) + xs;
            if (pred != nullptr || Mv != nullptr)
            {
                // row r of the pose stripe: [predicted (EKF.cpp:439-443)] then [downdated]
                float a0 = P[(size_t)0 * ldp + r], a1 = P[(size_t)1 * ldp + r], a2 = P[(size_t)2 * ldp + r];
                if (pred != nullptr)
                {
                    if (r >= 3)
                    {
                        if (r - 3 < pred_w)
                        {
                            float o0, o1, o2;
                            predict_stripe_col<float>(pred[0], pred[1], a0, a1, a2, &o0, &o1, &o2);
                            a0 = o0;
                            a1 = o1;
                            a2 = o2;
                        }
                    }
                    else // the pose block: predicted Pvv, row r
                    {
                        a0 = pred[5 + r];
                        a1 = pred[5 + r + 3];
                        a2 = pred[5 + r + 6];
                    }
                }
                if (r >= 3)
                {
                    P[(size_t)0 * ldp + r] = a0 - xm0;
                    P[(size_t)1 * ldp + r] = a1 - xm1;
                    P[(size_t)2 * ldp + r] = a2 - xm2;
                }
                else
                {
                    // the 3 x 3 pose block: the increment W1 W1^T is symmetric bit for bit in the reference's product;
                    // here (r, c) and (c, r) come from different dot products, so the thread of the larger index
                    // applies ITS increment to both
                    const float av[3] = {a0, a1, a2};
                    const float dv[3] = {xm0, xm1, xm2};
#pragma unroll
                    for (int c = 0; c < 3; c++)
                    {
                        if (c == r)
                        {
                            P[(size_t)c * ldp + r] = av[c] - dv[c];
                        }
                        else if (c < r)
                        {
                            const float bcr = (pred != nullptr) ? pred[5 + c + 3 * r] : P[(size_t)r * ldp + c]; // (c, r)
                            P[(size_t)c * ldp + r] = av[c] - dv[c];
                            P[(size_t)r * ldp + c] = bcr - dv[c];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4 (f64) on v_mfma_f64_16x16x4_f64: the f64 counterpart of ekf_panel_mfma_f32 (same generic form, same fusions except
// the pending predict, which the f64 engine launches on its own).  Tile = 16 rows x 16 columns per single-wave
// workgroup: D[i][j], i <-> output column (A[i = lane&15][kq = lane>>4] = Bt[(q0+kq)*ldb + c0 + i]), j <-> row
// (B[kq][j = lane&15] = A[row0 + j][q0 + kq]); element g of a lane is D[(lane>>4) + 4 g][lane&15].
// grid = (n_pad/16, ceil(nc/16)).  The vector-unit kernel it replaces (ekf_gain_lds_kernel<double>) ran 32 workgroups
// for 19-25 us at N = 1000, k = 64.
// ------------------------------------------------------------------------------------------------
template <bool SUB, bool XUPD>
__global__ void __launch_bounds__(64) ekf_panel_mfma_f64(const double* __restrict__ A, int lda, int n, int kq, int nc,
                                                          const double* __restrict__ Bt, int ldb,
                                                          const double* __restrict__ u, double* __restrict__ OUT, int ldo,
                                                          double* __restrict__ X, double* __restrict__ Pv = nullptr,
                                                          int ldp = 0, const double* __restrict__ Mv = nullptr,
                                                          double* __restrict__ wv_out = nullptr,
                                                          const double* __restrict__ pred = nullptr, int pred_w = 0)
{
    // pred != nullptr (gain, XUPD): commits a predict() that the gather and factor kernels applied on the fly, exactly as
    // ekf_panel_mfma_f32 does: pred = {g02, g12, pose (3), Pvv (9)} from the gather kernel.
    const int  lane = threadIdx.x;
    const int  lj   = lane & 15;
    const int  lq   = lane >> 4;
    const int  row0 = blockIdx.x * 16;
    const int  c0   = blockIdx.y * 16;
    const bool cok  = (c0 + lj) < nc;
    const int  cc   = cok ? (c0 + lj) : (nc - 1);
    const bool dox  = XUPD && (blockIdx.y == 0);
    f64x4      acc  = {0.0, 0.0, 0.0, 0.0};
    double     xs = 0.0, xm0 = 0.0, xm1 = 0.0, xm2 = 0.0;
    const double* pa = A + row0 + lj;
    double oldv[4];
    if (SUB)
    {
#pragma unroll
        for (int g = 0; g < 4; g++)
        {
            const int col = c0 + lq + 4 * g;
            oldv[g]       = OUT[(size_t)(col < nc ? col : nc - 1) * ldo + row0 + lj];
        }
    }
    constexpr int NQ = 16; // k-quads per trip: every load of the trip is issued before its first MFMA
    for (int qb = 0; qb < kq; qb += 4 * NQ)
    {
        double b[NQ], g[NQ], uq[NQ], m0[NQ], m1[NQ], m2[NQ];
#pragma unroll
        for (int t = 0; t < NQ; t++)
        {
            const int q  = qb + 4 * t + lq;
            const int qc = (q < kq) ? q : (kq - 1);
            b[t]         = pa[(size_t)qc * lda];
            g[t]         = Bt[(size_t)qc * ldb + cc];
            uq[t]        = (XUPD && dox) ? u[qc] : 0.0;
            if (XUPD && dox && Mv != nullptr)
            {
                m0[t] = Mv[qc];
                m1[t] = Mv[kq + qc];
                m2[t] = Mv[2 * kq + qc];
            }
            else
            {
                m0[t] = m1[t] = m2[t] = 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < NQ; t++)
        {
            const int    q  = qb + 4 * t + lq;
            const bool   ok = q < kq;
            const double gg = (ok && cok) ? g[t] : 0.0;
            const double bb = ok ? b[t] : 0.0;
            acc             = __builtin_amdgcn_mfma_f64_16x16x4f64(gg, bb, acc, 0, 0, 0);
            xs += bb * uq[t];
            xm0 += bb * m0[t];
            xm1 += bb * m1[t];
            xm2 += bb * m2[t];
        }
    }
#pragma unroll
    for (int g = 0; g < 4; g++)
    {
        const int col = c0 + lq + 4 * g;
        if (col < nc)
        {
            double* o = OUT + (size_t)col * ldo + row0 + lj;
            if (!SUB && Mv != nullptr && row0 + lj < 3) // pose rows of the W1 panel: kept aside, stored as zero
            {
                wv_out[(size_t)(row0 + lj) * nc + col] = acc[g];
                *o                                     = 0.0;
            }
            else
            {
                *o = SUB ? (oldv[g] - acc[g]) : acc[g];
            }
        }
    }
    if (dox)
    {
        xs += __shfl_xor(xs, 16);
        xs += __shfl_xor(xs, 32);
        xm0 += __shfl_xor(xm0, 16);
        xm0 += __shfl_xor(xm0, 32);
        xm1 += __shfl_xor(xm1, 16);
        xm1 += __shfl_xor(xm1, 32);
        xm2 += __shfl_xor(xm2, 16);
        xm2 += __shfl_xor(xm2, 32);
        const int r = row0 + lj;
        if (lq == 0 && r < n)
        {
            X[r] = ((pred != nullptr && r < 3) ? pred[2 + r] : X[r]) + xs;
            if (pred != nullptr || Mv != nullptr)
            {
                // row r of the pose stripe: [predicted (EKF.cpp:439-443)] then [downdated] (xm* are zero without Mv)
                double a0 = Pv[(size_t)0 * ldp + r], a1 = Pv[(size_t)1 * ldp + r], a2 = Pv[(size_t)2 * ldp + r];
                if (pred != nullptr)
                {
                    if (r >= 3)
                    {
                        if (r - 3 < pred_w)
                        {
                            double o0, o1, o2;
                            predict_stripe_col<double>(pred[0], pred[1], a0, a1, a2, &o0, &o1, &o2);
                            a0 = o0;
                            a1 = o1;
                            a2 = o2;
                        }
                    }
                    else // the pose block: predicted Pvv, row r
                    {
                        a0 = pred[5 + r];
                        a1 = pred[5 + r + 3];
                        a2 = pred[5 + r + 6];
                    }
                }
                if (r >= 3)
                {
                    Pv[(size_t)0 * ldp + r] = a0 - xm0;
                    Pv[(size_t)1 * ldp + r] = a1 - xm1;
                    Pv[(size_t)2 * ldp + r] = a2 - xm2;
                }
                else // the 3 x 3 pose block: one increment for (r, c) and (c, r) (see ekf_panel_mfma_f32)
                {
                    const double av[3] = {a0, a1, a2};
                    const double dv[3] = {xm0, xm1, xm2};
#pragma unroll
                    for (int c = 0; c < 3; c++)
                    {
                        if (c == r)
                        {
                            Pv[(size_t)c * ldp + r] = av[c] - dv[c];
                        }
                        else if (c < r)
                        {
                            const double bcr = (pred != nullptr) ? pred[5 + c + 3 * r] : Pv[(size_t)r * ldp + c]; // (c, r)
                            Pv[(size_t)c * ldp + r] = av[c] - dv[c];
                            Pv[(size_t)r * ldp + c] = bcr - dv[c];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K5 (f32), persistent symmetric form -- the shipped P-GEMM.
//   * symmetric: only tiles with ti >= tj are computed (half the MFMA work, half the P reads); the result of
//     an off-diagonal tile is stored in place and (full storage only) transposed into the mirror tile: for MFMA
//     block b and register group g the registers 4g..4g+3 of a lane are four consecutive ROWS of the mirror
//     tile in column row0+4*lj+b, i.e. one 16-byte store; L2 merges the pieces of a 128-byte line;
//   * persistent: gridDim.x workgroups (two per CU) walk a host-built tile list with stride gridDim.x, and the
//     P tile of the NEXT list entry is requested (16 x 16-byte loads per lane into a second register set)
//     right before the MFMA loop of the current one, so every workgroup keeps 64 KiB of HBM reads in flight
//     while it computes;
//   * the W1 panels go global -> LDS by LDS-DMA (no staging registers): one 1 KiB piece = two k-rows of 128
//     floats per wave-instruction, lane-linear LDS image; k8 = round_up(k, 8), columns [k, k8) of W1 are zero.
// Register plan per lane: 64 accumulators + 2 x 64 P-tile values; pinned to two waves per SIMD.
// ------------------------------------------------------------------------------------------------
template <int KC, bool NT, bool MIRROR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
ekf_downdate_psym_f32(float* __restrict__ P, int ldp, const float* __restrict__ W1, int ldw, int k8,
                      const int2* __restrict__ tile_list, int ntiles, long long* __restrict__ stamps)
{
    int        stamp_slot = 0; // diagnostic: s_memtime of workgroup 0 at the phase boundaries of its first tiles
    const bool stamping   = (stamps != nullptr) && blockIdx.x == 0 && threadIdx.x == 0;
    auto       stamp      = [&]() {
        if (stamping && stamp_slot < 60)
        {
            stamps[stamp_slot++] = (long long)__builtin_readcyclecounter();
        }
    };

    __shared__ __attribute__((aligned(16))) float s_pan[2 * KC * 128];
    float* sB = s_pan;            // rows of the tile   [kk][128]
    float* sA = s_pan + KC * 128; // columns of the tile [kk][128]

    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lj   = lane & 31;
    const int lh   = lane >> 5;
    const int G    = gridDim.x;

    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void*       lptr_t;

    auto tile_base = [&](int2 t) -> float* {
        return P + (size_t)(t.y * 128 + wave * 32 + 4 * lh) * ldp + t.x * 128 + 4 * lj;
    };
    auto load_tile = [&](float* pbase, f32x4 (&pv)[16]) {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const float* src = pbase + (size_t)((r & 3) + 8 * (r >> 2)) * ldp;
            if (NT)
            {
                pv[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            }
            else
            {
                pv[r] = *reinterpret_cast<const f32x4*>(src);
            }
        }
    };
    // one tile: panels -> LDS, (prefetch next P tile), MFMA, P -= acc, store in place + mirror
    auto process = [&](int2 cur, f32x4 (&pv)[16], bool have_next, int2 nxt, f32x4 (&pn)[16]) {
        const int row0 = cur.x * 128;
        const int col0 = cur.y * 128;
        f32x16    acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        for (int k0 = 0; k0 < k8; k0 += KC)
        {
            const int kc = min(KC, k8 - k0);
            stamp();
            __syncthreads(); // previous readers of the panels are done
            stamp();
#pragma unroll
            for (int it = 0; it < KC / 8; it++)
            {
                const int kkb = it * 8 + wave * 2;
                if (kkb < kc)
                {
                    const float* w = W1 + (size_t)(k0 + kkb + lh) * ldw + 4 * lj;
                    __builtin_amdgcn_global_load_lds((gptr_t)(w + row0), (lptr_t)(sB + kkb * 128), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(w + col0), (lptr_t)(sA + kkb * 128), 16, 0, 0);
                }
            }
            __syncthreads(); // drains the DMA (vmcnt) and publishes the panels
            stamp();
            __builtin_amdgcn_sched_barrier(0);
            if (k0 == 0 && have_next)
            {
                load_tile(tile_base(nxt), pn);
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int kk = 0; kk < kc; kk += 8)
            {
#pragma unroll
                for (int t = 0; t < 8; t += 2)
                {
                    const float4 b = *reinterpret_cast<const float4*>(&sB[(kk + t + lh) * 128 + 4 * lj]);
                    const float  a = sA[(kk + t + lh) * 128 + wave * 32 + lj];
                    acc0           = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc0, 0, 0, 0);
                    acc1           = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc1, 0, 0, 0);
                    acc2           = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc2, 0, 0, 0);
                    acc3           = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, acc3, 0, 0, 0);
                }
            }
        }
        stamp();
        float* pbase = tile_base(cur);
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            pv[r][0] -= acc0[r];
            pv[r][1] -= acc1[r];
            pv[r][2] -= acc2[r];
            pv[r][3] -= acc3[r];
            float* dst = pbase + (size_t)((r & 3) + 8 * (r >> 2)) * ldp;
            if (NT)
            {
                __builtin_nontemporal_store(pv[r], reinterpret_cast<f32x4*>(dst));
            }
            else
            {
                *reinterpret_cast<f32x4*>(dst) = pv[r];
            }
        }
        if (MIRROR && cur.x != cur.y) // mirror tile (tj, ti); skipped under block-lower storage
        {
            float* mbase = P + (size_t)(row0 + 4 * lj) * ldp + col0 + wave * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
#pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    const f32x4 m = {pv[4 * g + 0][b], pv[4 * g + 1][b], pv[4 * g + 2][b], pv[4 * g + 3][b]};
                    *reinterpret_cast<f32x4*>(mbase + (size_t)b * ldp + 8 * g) = m;
                }
            }
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles)
    {
        return;
    }
    f32x4 pvA[16], pvB[16];
    int2   cur = tile_list[t];
    load_tile(tile_base(cur), pvA);
    while (true)
    {
        int        tn  = t + G;
        bool       hn  = tn < ntiles;
        int2       nxt = hn ? tile_list[tn] : cur;
        process(cur, pvA, hn, nxt, pvB);
        if (!hn)
        {
            break;
        }
        t               = tn;
        cur             = nxt;
        tn              = t + G;
        hn              = tn < ntiles;
        nxt             = hn ? tile_list[tn] : cur;
        process(cur, pvB, hn, nxt, pvA);
        if (!hn)
        {
            break;
        }
        t   = tn;
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// K5 (f32), block-lower storage, k8 <= 64: the P-GEMM with every memory operation issued from INSIDE the MFMA
// loop.  A pure read-modify-write of the same 3160 tiles (tools/probes/tile_rmw_probe.hip) takes 63 us; the
// kernels above take 87-94 us because a wave's memory traffic comes in bursts between its MFMA loops (16 loads,
// then 8200+ cycles of MFMA, then 16 stores): the memory system idles while the waves compute.  Here a tile's
// k <= 64 is cut into two chunks of 32 (two pairs of LDS panel buffers, 16 KB each) and a wave's stream is
//     chunk 0 of tile i : [barrier] DMA panels 1(i)   ; 16 k-pairs of MFMA; behind k-pairs 0-7 the 16 STORES of
//                                                       tile i-1's results, behind 8-15 the first 8 LOADS of P_i
//     chunk 1 of tile i : [wait panels 1, barrier] DMA panels 0(i+1) ; 16 k-pairs; behind 0-7 the other 8 LOADS
//     P_i -= acc        (the results stay in registers until the next chunk 0)
// One register set serves "results of tile i-1 going out" and "P_i coming in" (a load may overwrite a register
// whose store has been issued: VMEM instructions leave in order), so the kernel needs acc (64) + P (64) VGPRs,
// every tile runs the same code with the same register roles, and no tile waits for a store to complete.
// All traffic uses buffer instructions (resource + uniform SGPR offset + one fixed per-lane VGPR offset): no
// VGPRs hold 64-bit pointers.  Rows of W1 beyond k8 are outside the W1 resource: the DMA delivers zeros for them,
// so every chunk is 32 deep and the MFMA loops have fixed trip counts.
// vmcnt retires in order: "all but the 24 youngest" at the top of chunk 1 = the DMA issued at the top of chunk 0
// has landed (16 stores + 8 loads were issued after it; 8 for a workgroup's first tile).
// Tile hand-out by atomic ticket; a ticket requested at the top of tile i is collected at the top of
// tile i+1, its tile looked up, published through LDS at the top of chunk 1 of tile i+1 and consumed there.
// ------------------------------------------------------------------------------------------------
// NTMODE: 0 ordinary accesses, 1 non-temporal loads and stores, 2 loads only, 3 stores only,
//         4 agent-scope (sc1) loads, 5 agent-scope loads and stores, 6 sc1 loads + nt stores (cache-policy experiments)
// NCH: chunks of 32 columns per tile (2: k <= 64; 4: k <= 128, the deferred flushes).  Chunk c uses LDS buffer pair
// c & 1; the stores go behind chunk 0, the loads behind chunk 0/1 (NCH = 2) or chunks 1 and 2 (NCH = 4).
// KC: columns per chunk (32; 24 with NCH = 4 covers 64 < k <= 96 -- an update's 64 columns plus a few rank-1 heading
// columns -- at 1.5x instead of 2x the matrix-core work of k = 64, which keeps that launch bandwidth-bound).
// PF: the LDS operands of k-pair g+1 are requested BEFORE the four MFMAs of k-pair g are issued (software pipelining:
// the scheduling barriers between k-pairs otherwise put every group's LDS latency in front of its MFMAs).  It matters
// where the launch is matrix-bound (k = 128: 4 chunks).
// BATCH (cslam_ekf_batch.hip): P and W1 are slabs of independent filters of the same size, instance i at byte offsets
// i * sP / i * sW; a tile's x carries the instance in its upper 16 bits, p_span / w_span are the slabs' sizes in bytes
// (< 4 GiB), and every instance's panel must be zero beyond its pending columns up to NCH * KC (the host clears them:
// the resource bound that delivers those zeros for one filter spans the whole slab here).
template <int NTMODE, int NCH, int KC = 32, bool PF = false, bool BATCH = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
ekf_downdate_psym4_f32(float* __restrict__ P, int ldp, const float* __restrict__ W1, int ldw, int k8,
                       const int2* __restrict__ tile_list_in, int ntiles_in, int* __restrict__ ticket_in,
                       int* __restrict__ ticket_reset, unsigned long long* __restrict__ hwids,
                       const int* __restrict__ seg_off = nullptr, unsigned sP = 0, unsigned sW = 0, unsigned p_span = 0,
                       unsigned w_span = 0, unsigned* __restrict__ signal = nullptr, unsigned sig_add = 0, int sig_count = 0,
                       int sig_stride = 0)
{
    // Look-ahead windows: the kernels launched before this one on the stream have finished and their results are visible
    // device-wide (kernel boundary): tell the `sig_count` factor chains that wait for them (ekf_la_chain_kernel /
    // ekf_la_chain_batch; one counter each).  One atomic per filter here replaces a release fence + atomic in every workgroup
    // of the blocks kernel.
    if (signal != nullptr && blockIdx.x == 0 && (int)threadIdx.x < sig_count)
    {
        atomicAdd(signal + (size_t)threadIdx.x * sig_stride, sig_add);
    }
    // seg_off != nullptr: one tile queue per XCD.  tile_list_in is then Morton-ordered and cut into eight segments
    // (seg_off[0..8]), ticket_in / ticket_reset are eight counters each, and workgroup b works on queue b & 7 -- its
    // first two tiles by its rank b >> 3 in that queue, the rest by tickets -- exactly as on the single queue.  Blocks
    // are dealt round-robin over the XCDs (b and b + 8 share one), so each L2 sees one compact patch of the triangle and
    // the W1 panels of a few block rows and columns instead of all of them.  Placement is a speed matter only: queue
    // membership is by block index, every queue has its workgroups whatever the hardware does with them.
    const bool xq = seg_off != nullptr; // (kernel-uniform)
    const int  qid = xq ? (int)(blockIdx.x & 7u) : 0;
    const int2* tile_list = xq ? tile_list_in + seg_off[qid] : tile_list_in;
    const int   ntiles    = xq ? seg_off[qid + 1] - seg_off[qid] : ntiles_in;
    int*        ticket    = xq ? ticket_in + qid : ticket_in;
    static_assert(KC % 8 == 0 && KC / 2 >= ((NCH == 2) ? 16 : 8), "chunk depth: DMA granularity 8, room for the memory-op schedule");
    // four separate LDS objects (not one array): the compiler's wait-count pass can then tell that the panel
    // DMA in flight (other chunk's buffers) does not alias the buffers the MFMA loop is reading
    __shared__ __attribute__((aligned(16))) float s_b0[KC * 128]; // chunk 0: rows of the tile    [kk][128]
    __shared__ __attribute__((aligned(16))) float s_a0[KC * 128]; // chunk 0: columns of the tile [kk][128]
    __shared__ __attribute__((aligned(16))) float s_b1[KC * 128];
    __shared__ __attribute__((aligned(16))) float s_a1[KC * 128];
    __shared__ int2 s_next;

    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lj   = lane & 31;
    const int lh   = lane >> 5;
    const int G    = xq ? (int)((gridDim.x - (blockIdx.x & 7u) + 7u) >> 3) : (int)gridDim.x; // workgroups on this queue

    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kAuxLd = (NTMODE == 1 || NTMODE == 2) ? 2 : ((NTMODE >= 4) ? 16 : 0); // bit 1: nt, bit 4: sc1
    constexpr int kAuxSt = (NTMODE == 1 || NTMODE == 3 || NTMODE == 6) ? 2 : ((NTMODE == 5) ? 16 : 0);
    // (P must be < 4 GiB: ldp < 32768; the host checks.)
    const __amdgpu_buffer_rsrc_t rsP =
        __builtin_amdgcn_make_buffer_rsrc(P, 0, BATCH ? p_span : (unsigned)((size_t)ldp * ldp * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(W1), 0, BATCH ? w_span : (unsigned)((size_t)k8 * ldw * 4), 0x00020000);
    const unsigned lane_off = (unsigned)(((wave * 32 + 4 * lh) * ldp + 4 * lj) * 4);
    auto tile_base = [&](int2 t) -> unsigned {
        if constexpr (BATCH)
        {
            return (unsigned)(t.x >> 16) * sP + (unsigned)(((size_t)(t.y * 128) * ldp + (t.x & 0xFFFF) * 128) * 4);
        }
        return (unsigned)(((size_t)(t.y * 128) * ldp + t.x * 128) * 4);
    };
    auto row_off   = [&](int r) -> unsigned { return (unsigned)(((r & 3) + 8 * (r >> 2)) * ldp * 4); };
    auto load1     = [&](unsigned tbase, int r) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, lane_off, tbase + row_off(r), kAuxLd));
    };
    auto store1 = [&](unsigned tbase, int r, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsP, lane_off, tbase + row_off(r), kAuxSt);
    };
    // panel chunk C (rows C*32 .. C*32+31 of W1) of tile t into the LDS buffers of chunk C: always 8 DMA
    // instructions per wave (each: 2 rows x 128 floats)
    const unsigned dma_lane_off = (unsigned)((lh * ldw + 4 * lj) * 4);
    auto dma_chunk = [&](int2 t, auto C) { // C: chunk index (compile time); LDS buffer pair C & 1
        constexpr int  c    = decltype(C)::value;
        const unsigned ibase = BATCH ? (unsigned)(t.x >> 16) * sW : 0u;
        const unsigned row0  = ibase + (unsigned)((BATCH ? (t.x & 0xFFFF) : t.x) * 128 * 4);
        const unsigned col0  = ibase + (unsigned)(t.y * 128 * 4);
#pragma unroll
        for (int it = 0; it < KC / 8; it++)
        {
            const int      kkb  = it * 8 + wave * 2;
            const unsigned roff = (unsigned)((c * KC + kkb) * ldw * 4);
            lptr_t         db   = (lptr_t)(((c & 1) == 0 ? s_b0 : s_b1) + kkb * 128);
            lptr_t         da   = (lptr_t)(((c & 1) == 0 ? s_a0 : s_a1) + kkb * 128);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, db, 16, dma_lane_off, roff + row0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, da, 16, dma_lane_off, roff + col0, 0, 0);
        }
    };

    f32x16 acc0, acc1, acc2, acc3;
    // one k-pair (4 MFMAs) of the chunk in LDS buffer pair B at rows g*2, g*2+1 (g = 0..15)
    auto mfma_group = [&](auto B, int g) {
        constexpr int bsel = decltype(B)::value;
        const float*  sB   = bsel == 0 ? s_b0 : s_b1;
        const float*  sA   = bsel == 0 ? s_a0 : s_a1;
        const float4  b    = *reinterpret_cast<const float4*>(&sB[(2 * g + lh) * 128 + 4 * lj]);
        const float   a    = sA[(2 * g + lh) * 128 + wave * 32 + lj];
        acc0               = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.x, acc0, 0, 0, 0);
        acc1               = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.y, acc1, 0, 0, 0);
        acc2               = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.z, acc2, 0, 0, 0);
        acc3               = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b.w, acc3, 0, 0, 0);
    };
    using C0 = std::integral_constant<int, 0>;
    // schedule of the 16 P loads of a tile over its chunks: NCH = 2: chunk 0 k-pairs 8-15 and chunk 1 k-pairs 0-7;
    // NCH = 4: chunk 1 and chunk 2, k-pairs 0-7 (chunk 0 carries the 16 stores, the last chunk nothing)
    constexpr int kLoadChunkA = (NCH == 2) ? 0 : 1; // loads 0-7
    constexpr int kLoadChunkB = (NCH == 2) ? 1 : 2; // loads 8-15
    constexpr int kLoadA_g0   = (NCH == 2) ? 8 : 0; // first k-pair behind which loads 0-7 go

    f32x4 pv[16]; // results of the previous tile -> P of the current tile -> results of the current tile

    // One tile.  prev_base: where the results held in pv go (not FIRST).  t_next_in: FIRST: index of the next
    // tile; otherwise thread 0's raw ticket from one tile ago.  Returns false after the workgroup's last tile.
    auto process = [&](auto FIRST, int2 cur, unsigned cbase, unsigned prev_base, int t_next_in, int& t_next_out,
                       int2& nxt_out) -> bool {
        constexpr bool first = decltype(FIRST)::value;
        int2           look   = make_int2(-1, -1);
        int            tk_new = 0;
        int2           nxt    = make_int2(-1, -1);
        bool           have_next = false;
        auto chunk = [&](auto CC) {
            constexpr int c = decltype(CC)::value;
            // VM operations issued behind the panel DMA at the top of chunk x ("all but that many" = the DMA landed)
            constexpr int nb_prev = (c == 0) ? ((NCH - 1 == kLoadChunkA ? 8 : 0) + (NCH - 1 == kLoadChunkB ? 8 : 0))
                                             : ((c - 1 == 0 && !first ? 16 : 0) + (c - 1 == kLoadChunkA ? 8 : 0) +
                                                (c - 1 == kLoadChunkB ? 8 : 0));
            // ---- top of chunk c: its panels have landed; everybody is done with the other buffer pair ----
            if constexpr (c == 0)
            {
                // panels 0 were requested at the top of the previous tile's last chunk; whatever followed them (the
                // loads of the subtraction) has been waited for, unless nothing followed (NCH = 4) or this is the
                // first tile
                __builtin_amdgcn_s_waitcnt((first || nb_prev == 0) ? 0x0070 : 0xC07F);
            }
            else
            {
                __builtin_amdgcn_s_waitcnt(nb_prev == 0 ? 0x0F70 : (nb_prev == 8 ? 0x0F78 : (nb_prev == 16 ? 0x4F70 : 0x4F78)));
                if (c == 1 && !first && tid == 0)
                {
                    s_next = look;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (c == 0)
            {
                // thread 0: the ticket requested one tile ago has returned (it is older than the DMA group waited for
                // above): look its tile up (8-byte load, again covered by the next wait), then request the next one
                if (tid == 0)
                {
                    if (!first)
                    {
                        int tk_raw = t_next_in;
                        asm volatile("" : "+v"(tk_raw));
                        const int tt = 2 * G + tk_raw;
                        if (tt >= 0 && tt < ntiles)
                        {
                            look = tile_list[tt];
                        }
                    }
                    const int zero = 0, one = 1;
                    asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(tk_new) : "v"(zero), "v"(one), "s"(ticket) : "memory");
                }
                acc0 = acc1 = acc2 = acc3 = f32x16{0};
            }
            if constexpr (c == 1)
            {
                if (first)
                {
                    nxt = t_next_in < ntiles ? tile_list[t_next_in] : make_int2(-1, -1);
                }
                else
                {
                    const int2 sn = s_next;
                    nxt           = make_int2(__builtin_amdgcn_readfirstlane(sn.x), __builtin_amdgcn_readfirstlane(sn.y));
                }
                have_next = nxt.x >= 0;
                asm volatile("" : "+v"(tk_new)); // (its request has returned: covered by the wait above)
            }
            // the next panels into the other buffer pair
            if constexpr (c + 1 < NCH)
            {
                dma_chunk(cur, std::integral_constant<int, c + 1>{});
            }
            else
            {
                dma_chunk(have_next ? nxt : cur, C0{}); // (re-reads this tile's panel after the last tile: harmless)
            }
            __builtin_amdgcn_sched_barrier(0);
            float4 b_cur = make_float4(0.f, 0.f, 0.f, 0.f);
            float  a_cur = 0.f;
            if constexpr (PF)
            {
                const float* sB = (c & 1) == 0 ? s_b0 : s_b1;
                const float* sA = (c & 1) == 0 ? s_a0 : s_a1;
                b_cur           = *reinterpret_cast<const float4*>(&sB[lh * 128 + 4 * lj]);
                a_cur           = sA[lh * 128 + wave * 32 + lj];
            }
#pragma unroll
            for (int g = 0; g < KC / 2; g++)
            {
                if constexpr (PF)
                {
                    const float* sB = (c & 1) == 0 ? s_b0 : s_b1;
                    const float* sA = (c & 1) == 0 ? s_a0 : s_a1;
                    float4       b_nxt = b_cur;
                    float        a_nxt = a_cur;
                    if (g + 1 < KC / 2)
                    {
                        b_nxt = *reinterpret_cast<const float4*>(&sB[(2 * (g + 1) + lh) * 128 + 4 * lj]);
                        a_nxt = sA[(2 * (g + 1) + lh) * 128 + wave * 32 + lj];
                    }
                    acc0  = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.x, acc0, 0, 0, 0);
                    acc1  = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.y, acc1, 0, 0, 0);
                    acc2  = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.z, acc2, 0, 0, 0);
                    acc3  = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur.w, acc3, 0, 0, 0);
                    b_cur = b_nxt;
                    a_cur = a_nxt;
                }
                else
                {
                    mfma_group(std::integral_constant<int, (c & 1)>{}, g);
                }
                if (c == 0 && g < 8 && !first)
                {
                    store1(prev_base, 2 * g, pv[2 * g]);
                    store1(prev_base, 2 * g + 1, pv[2 * g + 1]);
                }
                if (c == kLoadChunkA && g >= kLoadA_g0 && g < kLoadA_g0 + 8)
                {
                    pv[g - kLoadA_g0] = load1(cbase, g - kLoadA_g0);
                }
                if (c == kLoadChunkB && g < 8)
                {
                    pv[g + 8] = load1(cbase, g + 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        if constexpr (NCH == 4)
        {
            chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{});
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            pv[r][0] -= acc0[r];
            pv[r][1] -= acc1[r];
            pv[r][2] -= acc2[r];
            pv[r][3] -= acc3[r];
        }
        t_next_out = tk_new; // raw ticket (valid in thread 0: its request has returned)
        nxt_out    = nxt;
        return have_next;
    };

    const int t = xq ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; // rank on the queue
    if (blockIdx.x == 0 && tid < (xq ? 8 : 1))
    {
        ticket_reset[tid] = 0; // the counter(s) the NEXT launch on this stream will use
    }
    if (t >= ntiles)
    {
        return;
    }
    int2     cur   = tile_list[t];
    unsigned cbase = tile_base(cur);
    dma_chunk(cur, C0{});
    if (hwids != nullptr && tid == 0)
    {
        hwids[4 * blockIdx.x]     = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        hwids[4 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        hwids[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
    int  tk = 0;
    int2 nxt;
    bool hn = process(std::true_type{}, cur, cbase, 0u, t + G, tk, nxt);
    while (hn)
    {
        const unsigned rbase = cbase;
        cur                  = nxt;
        cbase                = tile_base(cur);
        hn                   = process(std::false_type{}, cur, cbase, rbase, tk, tk, nxt);
    }
    // the last tile's results
#pragma unroll
    for (int r = 0; r < 16; r++)
    {
        store1(cbase, r, pv[r]);
    }
    if (hwids != nullptr && tid == 0)
    {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        hwids[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------
// Block-lower storage -> full symmetric matrix (used by get_state): every tile above the tile diagonal is
// filled with the transpose of its mirror, and so is the element-wise upper triangle of every diagonal tile.  32x32 sub-tiles through LDS so both sides are coalesced.
// grid = (ceil(n/32), ceil(n/32)); blocks not strictly above the 128-tile diagonal exit.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) ekf_mirror_upper_kernel(T* __restrict__ P, int ldp, int n)
{
    __shared__ T tile[32][33];
    const int    bi = blockIdx.x * 32; // destination rows (upper part: small row index)
    const int    bj = blockIdx.y * 32; // destination columns
    // tiles above the tile diagonal, and inside a diagonal tile the element-wise upper triangle (see p_sym)
    const bool diag_tile = (bi >> 7) == (bj >> 7);
    if ((bi >> 7) > (bj >> 7) || (diag_tile && bi > bj))
    {
        return;
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    for (int r = ty; r < 32; r += 8)
    {
        // source element (bj + tx, bi + r): column bi + r, rows bj..bj+31 (coalesced along tx)
        const int si = bj + tx, sj = bi + r;
        tile[r][tx] = (si < n && sj < n) ? P[(size_t)sj * ldp + si] : (T)0;
    }
    __syncthreads();
    for (int c = ty; c < 32; c += 8)
    {
        // destination element (bi + tx, bj + c) = source (bj + c, bi + tx) = tile[tx][c]
        const int di = bi + tx, dj = bj + c;
        if (di < n && dj < n && (bi != bj || di < dj))
        {
            P[(size_t)dj * ldp + di] = tile[tx][c];
        }
    }
}

} // namespace cslam
