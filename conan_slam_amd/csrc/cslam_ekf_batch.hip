// cslam_ekf_batch.hip -- the batched Monte-Carlo engine: I independent f32 EKF-SLAM filters of the same size advance in
// lockstep on one GPU (BASELINE configs[4]; the unit that is replicated is the reference's filter loop,
// test/main.cpp:132-200: predict + batch update per control step, slam.h:235-266 via EKF.cpp:93-129, 406-455).
//
// One handle per filter (cslam_ekf_run_many) leaves the runs launch-bound: 8 co-resident instances of N = 2000 reach 2.0x
// one instance.  Here every stage of a look-ahead window (ekf_lookahead.hpp: two updates per window, their factor chain
// underneath the previous window's P-GEMM) is ONE launch for all instances:
//     stream F : ekf_la_chain_batch   I workgroups, each owning a compute unit (factor a -> carry -> factor b)
//     main     : ekf_la_rows_batch -> ekf_la_blocks_batch -> ekf_downdate_psym4_f32<BATCH> -> ekf_la_wide_batch
// The P-GEMM draws tickets over the union of the instances' lower-triangular tiles (instance-major: the workgroups of a
// launch work on one or two instances' panels at a time, which fit the L2s).  The device code of a window is the single
// filter's (the same *_body functions), so an instance's results are BITWISE those of a solo engine running look-ahead
// windows of the same pairs of updates (tests/test_batch_gpu.py).
//
// State lives in slabs with a fixed stride per instance: X [I][ldp], Pv [I][3 ldp], P [I][ldp ldp] (block-lower), the
// pending store W [2 regions][I][128][ldp] (P = Ps - Wp Wp^T, Wp = the previous window's panels), factor slots, the
// look-ahead scratch and the wait counters (labatch:: layout in ekf_lookahead.hpp).
#include <algorithm>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#include "cslam_common.hpp"
#include "ekf_kernels.hpp"
#include "ekf_kernels_fast.hpp"
#include "ekf_lookahead.hpp"
#include "ekf_pose_kernels.hpp"

using namespace cslam;

namespace cslam
{
// one launch per stage for all instances: blockIdx.y (the chain: blockIdx.x) is the instance
__global__ void __launch_bounds__(128) ekf_la_rows_batch(LaBatchWin w)
{
    const LaRowsArgs<float> a = la_batch_rows(w, blockIdx.y);
    ekf_la_rows_body<float>(a);
}
__global__ void __launch_bounds__(64) ekf_la_blocks_batch(LaBatchWin w)
{
    const LaPrepArgs<float> a = la_batch_prep(w, blockIdx.y);
    ekf_la_blocks_body<float>(a);
}
template <int K>
__global__ void __launch_bounds__(256) ekf_la_chain_batch(LaBatchWin w)
{
    const LaChainArgs<float> a = la_batch_chain(w, blockIdx.x); // (one workgroup per instance)
    ekf_la_chain_body<float, K>(a);
}
__global__ void __launch_bounds__(128) ekf_la_wide_batch1(LaBatchWin w) // (A/B: CSLAM_BATCH_WIDE_PAIRS=1)
{
    const LaWideArgs a = la_batch_wide(w, blockIdx.y);
    ekf_la_wide_body<1, 0>(a);
}
// (two pairs of waves per workgroup, 256 registers: 8 waves per compute unit instead of 4, see ekf_la_wide_body)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) ekf_la_wide_batch(LaBatchWin w)
{
    const LaWideArgs a = la_batch_wide(w, blockIdx.y);
    ekf_la_wide_body<2, 0>(a);
}
// ... with both updates of m = 32 observations (k = 64 known at compile time)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) ekf_la_wide_batch_k64(LaBatchWin w)
{
    const LaWideArgs a = la_batch_wide(w, blockIdx.y);
    ekf_la_wide_body<2, 64>(a);
}

// windows without a P-GEMM (nothing pending): the chains' go-ahead as a kernel of its own
__global__ void __launch_bounds__(256) ekf_la_signal_batch(unsigned* signal, unsigned add, int count, int stride)
{
    if ((int)threadIdx.x < count)
    {
        atomicAdd(signal + (size_t)threadIdx.x * stride, add);
    }
}
} // namespace cslam

struct cslam_ekf_batch
{
    int device = 0, I = 0, n = 0, ldp = 0, quirks = 0, num_cus = 0;
    hipStream_t stream = nullptr, stream_f = nullptr;
    float *   dX = nullptr, *dPv = nullptr, *dP = nullptr, *dW = nullptr, *dFo = nullptr, *dLa = nullptr, *dWv = nullptr;
    unsigned* dDone  = nullptr;
    int *     dFlags = nullptr, *dIdloc = nullptr;
    const float** dZtab   = nullptr; // [2][I] (two generations: a run() may be enqueued while the previous one executes)
    const int**   dIdftab = nullptr;
    int           tab_gen = 0;
    hipEvent_t    ev_gen[2]   = {nullptr, nullptr}; // the last kernel that reads generation g has finished
    bool          gen_used[2] = {false, false};
    int2*         dTiles  = nullptr;
    int*          dTicket = nullptr;
    int           n_tiles = 0, parity = 0;
    int           wcur = 0, kp = 0; // pending region and its columns
    unsigned      target = 0, seq = 0;
    long long     windows = 0;
    // A/B switches (env CSLAM_BATCH_WG_SIGNAL=1, CSLAM_BATCH_WIDE_PAIRS=1): the first forms of two stages, kept measurable
    int wg_signal = 0, wide_pairs = 2;
    int la_k64 = 1; // CSLAM_LA_K64=0: the general wide kernel for m = 32 too (A/B)
    long long* dStamps = nullptr; // CSLAM_BATCH_STAMPS=1: see LaBatchWin::stamps (printed after 300 windows)
    // bench support: HIP events around one P-GEMM launch in `prof_every` (an event pair costs ~11 us of stream time)
    int                                          prof_every = 0;
    long long                                    prof_seen  = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    size_t                                       prof_used = 0;

    static constexpr int kWcols = 128; // columns per instance and region: one window's panels

    size_t sW() const { return (size_t)kWcols * ldp; }
    float* wregion(int r) const { return dW + (size_t)r * I * sW(); }

    // dynamic LDS of ekf_la_chain_batch<K>: as Ekf<float>::la_chain_lds (cslam_ekf.hip) -- the workgroup's total is ~99 KB,
    // which keeps the P-GEMM's 64 KB workgroups off its compute unit and still fits beside one wide-kernel workgroup
    static size_t chain_lds(int K)
    {
        const size_t fixed = (K == 64) ? 53984 : 15008;
        return std::max(la_carry_lds<float>(), (size_t)101 * 1024 - fixed);
    }

    int use_device()
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        return CSLAM_OK;
    }

    void release()
    {
        (void)hipSetDevice(device);
        if (stream)
        {
            (void)hipStreamSynchronize(stream);
        }
        if (stream_f)
        {
            (void)hipStreamSynchronize(stream_f);
        }
        (void)hipFree(dX);
        (void)hipFree(dPv);
        (void)hipFree(dP);
        (void)hipFree(dW);
        (void)hipFree(dFo);
        (void)hipFree(dLa);
        (void)hipFree(dWv);
        (void)hipFree(dDone);
        (void)hipFree(dFlags);
        (void)hipFree(dIdloc);
        (void)hipFree(dZtab);
        (void)hipFree(dIdftab);
        (void)hipFree(dTiles);
        (void)hipFree(dTicket);
        (void)hipFree(dStamps);
        for (auto& e : prof_ev)
        {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        prof_ev.clear();
        for (hipEvent_t& e : ev_gen)
        {
            if (e)
            {
                (void)hipEventDestroy(e);
                e = nullptr;
            }
        }
        if (stream)
        {
            (void)hipStreamDestroy(stream);
        }
        if (stream_f)
        {
            (void)hipStreamDestroy(stream_f);
        }
    }

    int init()
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if (const char* e = getenv("CSLAM_BATCH_WG_SIGNAL"))
        {
            wg_signal = atoi(e) ? 1 : 0;
        }
        if (const char* e = getenv("CSLAM_LA_K64"))
        {
            la_k64 = atoi(e) ? 1 : 0;
        }
        if (const char* e = getenv("CSLAM_BATCH_WIDE_PAIRS"))
        {
            wide_pairs = atoi(e) == 1 ? 1 : 2;
        }
        if (getenv("CSLAM_BATCH_STAMPS"))
        {
            // 32 phase stamps, then {start, end} of up to 128 wide-kernel workgroups per instance
            CSLAM_HIP_TRY(hipMalloc(&dStamps, (32 + (size_t)I * 256) * sizeof(long long)));
            CSLAM_HIP_TRY(hipMemset(dStamps, 0, (32 + (size_t)I * 256) * sizeof(long long)));
        }
        hipDeviceProp_t prop;
        CSLAM_HIP_TRY(hipGetDeviceProperties(&prop, device));
        num_cus = prop.multiProcessorCount;
        if (I >= num_cus / 2)
        {
            return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_create: %d instances need %d compute units for their factor chains", I, I);
        }
        int lo = 0, hi = 0;
        CSLAM_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CSLAM_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        CSLAM_HIP_TRY(hipStreamCreateWithPriority(&stream_f, hipStreamNonBlocking, hi));
        CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_gen[0], hipEventDisableTiming));
        CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_gen[1], hipEventDisableTiming));
        const size_t L = (size_t)ldp;
        CSLAM_HIP_TRY(hipMalloc(&dX, I * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dPv, I * 3 * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dP, I * L * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dW, 2 * I * sW() * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dFo, (size_t)I * labatch::kFoBlock * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dLa, (size_t)I * labatch::kLaBlock * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dWv, (size_t)I * 192 * sizeof(float)));
        CSLAM_HIP_TRY(hipMalloc(&dDone, (size_t)I * labatch::kDoneBlock * sizeof(unsigned)));
        CSLAM_HIP_TRY(hipMalloc(&dFlags, (size_t)I * 2 * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dIdloc, (size_t)I * kLaMaxObs * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dZtab, (size_t)2 * I * sizeof(float*)));
        CSLAM_HIP_TRY(hipMalloc(&dIdftab, (size_t)2 * I * sizeof(int*)));
        CSLAM_HIP_TRY(hipMalloc(&dTicket, 2 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemset(dX, 0, I * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dPv, 0, I * 3 * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dP, 0, I * L * L * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dW, 0, 2 * I * sW() * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dFo, 0, (size_t)I * labatch::kFoBlock * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dLa, 0, (size_t)I * labatch::kLaBlock * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dWv, 0, (size_t)I * 192 * sizeof(float)));
        CSLAM_HIP_TRY(hipMemset(dDone, 0, (size_t)I * labatch::kDoneBlock * sizeof(unsigned)));
        CSLAM_HIP_TRY(hipMemset(dFlags, 0, (size_t)I * 2 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemset(dTicket, 0, 2 * sizeof(int)));
        // the union of the instances' lower-triangular tiles, instance-major; x = row tile | instance << 16
        const int         tiles = ldp / kTile;
        std::vector<int2> h;
        h.reserve((size_t)I * tiles * (tiles + 1) / 2);
        for (int i = 0; i < I; i++)
        {
            for (int tj = 0; tj < tiles; tj++)
            {
                for (int ti = tj; ti < tiles; ti++)
                {
                    if (ti * kTile < n) // (tiles of pure padding rows never change)
                    {
                        h.push_back(make_int2(ti | (i << 16), tj));
                    }
                }
            }
        }
        n_tiles = (int)h.size();
        CSLAM_HIP_TRY(hipMalloc(&dTiles, h.size() * sizeof(int2)));
        CSLAM_HIP_TRY(hipMemcpy(dTiles, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
        CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_la_chain_batch<32>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds(32)));
        CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_la_chain_batch<64>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds(64)));
        return CSLAM_OK;
    }

    int sync()
    {
        CSLAM_HIP_TRY(hipStreamSynchronize(stream_f));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    // P -= Wp Wp^T of every instance: one persistent launch over the union tile list.  sig_add != 0: the launch also tells
    // the waiting factor chains that the blocks kernel in front of it has finished (see ekf_la_blocks_body).
    int flush(unsigned sig_add = 0)
    {
        if (kp == 0)
        {
            if (sig_add)
            {
                hipLaunchKernelGGL(ekf_la_signal_batch, dim3(1), dim3(256), 0, stream, dDone, sig_add, I, labatch::kDoneBlock);
                CSLAM_HIP_TRY(hipGetLastError());
            }
            return CSLAM_OK;
        }
        float*    W     = wregion(wcur);
        const int k8    = round_up(kp, 8);
        const int cover = k8 <= 64 ? 64 : (k8 <= 96 ? 96 : 128);
        if (kp < cover) // (the kernel reads `cover` columns of every instance's panel)
        {
            CSLAM_HIP_TRY(hipMemset2DAsync(W + (size_t)kp * ldp, sW() * sizeof(float), 0, (size_t)(cover - kp) * ldp * sizeof(float),
                                           (size_t)I, stream));
        }
        const int      G      = std::min(n_tiles, 2 * (num_cus - I));
        const unsigned sPb    = (unsigned)((size_t)ldp * ldp * 4);
        const unsigned sWb    = (unsigned)(sW() * 4);
        const unsigned p_span = (unsigned)((size_t)I * ldp * ldp * 4);
        const unsigned w_span = (unsigned)((size_t)I * sW() * 4);
        parity ^= 1;
        const bool timed = prof_every > 0 && (prof_seen++ % prof_every) == 0 && prof_used < prof_ev.size();
        if (timed)
        {
            CSLAM_HIP_TRY(hipEventRecord(prof_ev[prof_used].first, stream));
        }
#define CSLAM_LAUNCH_PSYM4B(NCH, KC)                                                                                  \
    hipLaunchKernelGGL((ekf_downdate_psym4_f32<0, NCH, KC, false, true>), dim3(G), dim3(256), 0, stream, dP, ldp, W, ldp, \
                       kp, (const int2*)dTiles, n_tiles, dTicket + parity, dTicket + (parity ^ 1),                  \
                       (unsigned long long*)nullptr, (const int*)nullptr, sPb, sWb, p_span, w_span,                   \
                       sig_add ? dDone : (unsigned*)nullptr, sig_add, I, (int)labatch::kDoneBlock)
        if (k8 <= 64)
        {
            CSLAM_LAUNCH_PSYM4B(2, 32);
        }
        else if (k8 <= 96)
        {
            CSLAM_LAUNCH_PSYM4B(4, 24);
        }
        else
        {
            CSLAM_LAUNCH_PSYM4B(4, 32);
        }
#undef CSLAM_LAUNCH_PSYM4B
        CSLAM_HIP_TRY(hipGetLastError());
        if (timed)
        {
            CSLAM_HIP_TRY(hipEventRecord(prof_ev[prof_used++].second, stream));
        }
        wcur ^= 1;
        kp = 0;
        return CSLAM_OK;
    }

    // one window: updates a (and b when nu == 2) of every instance, with their held predicts
    int window(const LaBatchWin& w0)
    {
        LaBatchWin w = w0;
        const int  ka = 2 * w.ma, kb = w.nu == 2 ? 2 * w.mb : 0;
        const unsigned n_blocks = (unsigned)(3 + ka + 2 * kb);
        w.I        = I;
        w.n        = n;
        w.ldp      = ldp;
        w.lower    = 1;
        w.textbook = (quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 0 : 1;
        w.X        = dX;
        w.Pv       = dPv;
        w.P        = dP;
        w.Wp       = wregion(wcur);
        w.sW       = (long)sW();
        w.fo       = dFo;
        w.la       = dLa;
        w.wv       = dWv;
        w.done     = dDone;
        w.flags    = dFlags;
        w.idloc    = dIdloc;
        w.kp       = kp;
        w.target   = target + n_blocks;
        w.seq      = ++seq;
        w.wg_signal = wg_signal;
        w.wide_direct = getenv("CSLAM_BATCH_TIMING_DIRECT") ? 1 : 0;
        w.stamps      = dStamps;
        w.timeout  = 20000000ull; // 0.2 s of s_memrealtime ticks: a stuck wait raises CSLAM_FACTOR_INTERNAL instead of hanging
        // 1. the factor chains first: each takes a compute unit and waits there for its instance's blocks
        if (std::max(ka, kb) <= 32)
        {
            hipLaunchKernelGGL(ekf_la_chain_batch<32>, dim3(I), dim3(256), chain_lds(32), stream_f, w);
        }
        else
        {
            hipLaunchKernelGGL(ekf_la_chain_batch<64>, dim3(I), dim3(256), chain_lds(64), stream_f, w);
        }
        // 2. rows of the pending panels, then the small blocks of the current covariance (nothing may fail in between:
        //    the chains are waiting)
        hipLaunchKernelGGL(ekf_la_rows_batch, dim3(ka + kb, I), dim3(128), 0, stream, w);
        hipLaunchKernelGGL(ekf_la_blocks_batch, dim3(n_blocks, I), dim3(64), 0, stream, w);
        CSLAM_HIP_TRY(hipGetLastError());
        target += n_blocks;
        // 3. the P-GEMM of the previous window's panels: its first workgroup gives the chains their go-ahead (the blocks
        //    kernel has finished by then), and they run underneath it
        int rc = flush(wg_signal ? 0u : n_blocks);
        if (rc)
        {
            return rc;
        }
        // 4. the wide half of both updates (waits in the kernel for its instance's chain); its W1 panels become the pending
        //    columns of the region the P-GEMM has just left
        w.Wn = wregion(wcur);
        if (wide_pairs == 1)
        {
            hipLaunchKernelGGL(ekf_la_wide_batch1, dim3(round_up(n, kTile) / 32, I), dim3(128), 0, stream, w);
        }
        else
        {
            if (la_k64 && w.ma == 32 && (w.nu == 1 || w.mb == 32))
            {
                hipLaunchKernelGGL(ekf_la_wide_batch_k64, dim3(round_up(n, kTile) / 64, I), dim3(256), 0, stream, w);
            }
            else
            {
                hipLaunchKernelGGL(ekf_la_wide_batch, dim3(round_up(n, kTile) / 64, I), dim3(256), 0, stream, w);
            }
        }
        CSLAM_HIP_TRY(hipGetLastError());
        kp = ka + kb;
        windows++;
        if (dStamps && windows == 300)
        {
            long long h[32];
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            CSLAM_HIP_TRY(hipMemcpy(h, dStamps, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[cslam batch wide stamps, 10 ns ticks] ids+columns issue:%lld poll+DMA wait:%lld pht_a:%lld gain_a:%lld "
                            "store+share W1_a:%lld pht_b+corr:%lld share+G_b:%lld gain_b:%lld store_b:%lld\n",
                    h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[7], h[9] - h[8]);
            std::vector<long long> wt((size_t)I * 256);
            CSLAM_HIP_TRY(hipMemcpy(wt.data(), dStamps + 32, wt.size() * sizeof(long long), hipMemcpyDeviceToHost));
            const int nwg = std::min(128, round_up(n, kTile) / (32 * wide_pairs));
            long long t0  = wt[0];
            for (int i = 0; i < I; i++)
            {
                for (int b = 0; b < nwg; b++)
                {
                    t0 = std::min(t0, wt[(size_t)i * 256 + 2 * b]);
                }
            }
            fprintf(stderr, "[cslam batch wide workgroups] instance 0, duration by block of rows (10 ns ticks):");
            for (int b = 0; b < nwg; b++)
            {
                fprintf(stderr, " %lld", wt[2 * b + 1] - wt[2 * b]);
            }
            fprintf(stderr, "\n");
            for (int i = 0; i < I; i++)
            {
                long long s0 = 1ll << 62, s1 = 0, e0 = 1ll << 62, e1 = 0, dsum = 0;
                for (int b = 0; b < nwg; b++)
                {
                    const long long st = wt[(size_t)i * 256 + 2 * b] - t0, en = wt[(size_t)i * 256 + 2 * b + 1] - t0;
                    s0 = std::min(s0, st), s1 = std::max(s1, st), e0 = std::min(e0, en), e1 = std::max(e1, en);
                    dsum += en - st;
                }
                fprintf(stderr, "[cslam batch wide workgroups, 10 ns ticks] instance %d: start %lld..%lld end %lld..%lld mean duration %lld\n",
                        i, s0, s1, e0, e1, dsum / nwg);
            }
        }
        return CSLAM_OK;
    }
};

extern "C" {

int cslam_ekf_batch_create(int instances, int n_landmarks, int device, int quirks, cslam_ekf_batch_t* out)
{
    if (!out || instances < 1 || instances > 255 || n_landmarks < 1 || (quirks & ~CSLAM_Q_REF_EXACT))
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_create: bad arguments");
    }
    *out  = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0)
    {
        return fail(CSLAM_ERR_NO_DEVICE, "ekf_batch_create: no HIP device (this engine has no CPU fallback)");
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
    {
        device = 0;
    }
    if (device >= c)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_create: device %d of %d", device, c);
    }
    const int    n   = 3 + 2 * n_landmarks;
    const int    ldp = round_up(n, kTile);
    const size_t pb  = (size_t)instances * ldp * ldp * 4;
    if (pb >= ((size_t)1 << 32))
    {
        // (the P-GEMM addresses the slab through one buffer resource with 32-bit offsets)
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_create: %d instances x %d landmarks exceed the 4 GiB covariance slab", instances,
                    n_landmarks);
    }
    cslam_ekf_batch* b = new (std::nothrow) cslam_ekf_batch();
    if (!b)
    {
        return fail(CSLAM_ERR_ALLOC, "ekf_batch_create: out of host memory");
    }
    b->device = device;
    b->I      = instances;
    b->n      = n;
    b->ldp    = ldp;
    b->quirks = quirks;
    int rc    = b->init();
    if (rc)
    {
        b->release();
        delete b;
        return rc;
    }
    live_engines().fetch_add(1); // (a single-filter handle created beside this one keeps its kernels free of waits)
    *out = b;
    return CSLAM_OK;
}

int cslam_ekf_batch_destroy(cslam_ekf_batch_t h)
{
    if (!h)
    {
        return CSLAM_OK;
    }
    live_engines().fetch_sub(1);
    h->release();
    delete h;
    return CSLAM_OK;
}

int cslam_ekf_batch_set_state(cslam_ekf_batch_t h, int instance, const float* X, int n, const float* P, int ldp)
{
    if (!h || !X || !P || instance < 0 || instance >= h->I || n != h->n || ldp < n)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_set_state: bad arguments (instance %d, n %d)", instance, n);
    }
    int rc = h->use_device();
    if (rc || (rc = h->sync()))
    {
        return rc;
    }
    const size_t L  = (size_t)h->ldp;
    float*       dX = h->dX + instance * L;
    float*       dP = h->dP + instance * L * L;
    float*       dV = h->dPv + instance * 3 * L;
    CSLAM_HIP_TRY(hipMemcpyAsync(dX, X, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CSLAM_HIP_TRY(hipMemcpy2DAsync(dP, L * sizeof(float), P, (size_t)ldp * sizeof(float), (size_t)n * sizeof(float), (size_t)n,
                                   hipMemcpyHostToDevice, h->stream));
    CSLAM_HIP_TRY(hipMemcpyAsync(dV, dP, 3 * L * sizeof(float), hipMemcpyDeviceToDevice, h->stream)); // the pose stripe
    // a new state discards this instance's pending panels (the other instances' stay)
    for (int r = 0; r < 2; r++)
    {
        CSLAM_HIP_TRY(hipMemsetAsync(h->wregion(r) + instance * h->sW(), 0, h->sW() * sizeof(float), h->stream));
    }
    CSLAM_HIP_TRY(hipMemsetAsync(h->dFlags + 2 * instance, 0, 2 * sizeof(int), h->stream));
    CSLAM_HIP_TRY(hipStreamSynchronize(h->stream));
    return CSLAM_OK;
}

int cslam_ekf_batch_flush(cslam_ekf_batch_t h)
{
    if (!h)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_flush: null handle");
    }
    int rc = h->use_device();
    return rc ? rc : h->flush();
}

int cslam_ekf_batch_synchronize(cslam_ekf_batch_t h)
{
    if (!h)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_synchronize: null handle");
    }
    int rc = h->use_device();
    return rc ? rc : h->sync();
}

int cslam_ekf_batch_get_state(cslam_ekf_batch_t h, int instance, float* X, float* P, int ldp)
{
    if (!h || instance < 0 || instance >= h->I || (P && ldp < h->n))
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_get_state: bad arguments (instance %d)", instance);
    }
    int rc = h->use_device();
    if (rc || (P && (rc = h->flush())))
    {
        return rc;
    }
    const size_t L = (size_t)h->ldp;
    const int    n = h->n;
    if (X)
    {
        CSLAM_HIP_TRY(hipMemcpyAsync(X, h->dX + instance * L, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    if (P)
    {
        float*    dP = h->dP + instance * L * L;
        const int g  = (n + 31) / 32;
        hipLaunchKernelGGL(ekf_mirror_upper_kernel<float>, dim3(g, g), dim3(256), 0, h->stream, dP, h->ldp, n);
        hipLaunchKernelGGL(ekf_patch_pose_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, h->stream, dP,
                           h->dPv + instance * 3 * L, h->ldp, n);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipMemcpy2DAsync(P, (size_t)ldp * sizeof(float), dP, L * sizeof(float), (size_t)n * sizeof(float), (size_t)n,
                                       hipMemcpyDeviceToHost, h->stream));
    }
    CSLAM_HIP_TRY(hipStreamSynchronize(h->stream));
    return CSLAM_OK;
}

int cslam_ekf_batch_trace(cslam_ekf_batch_t h, double* traces)
{
    if (!h || !traces)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_trace: bad arguments");
    }
    int rc = h->use_device();
    if (rc || (rc = h->flush()))
    {
        return rc;
    }
    const size_t       L = (size_t)h->ldp;
    const int          n = h->n;
    std::vector<float> diag((size_t)n);
    for (int i = 0; i < h->I; i++)
    {
        CSLAM_HIP_TRY(hipMemcpy2DAsync(diag.data(), sizeof(float), h->dP + i * L * L, (L + 1) * sizeof(float), sizeof(float),
                                       (size_t)n, hipMemcpyDeviceToHost, h->stream));
        CSLAM_HIP_TRY(hipMemcpy2DAsync(diag.data(), sizeof(float), h->dPv + i * 3 * L, (L + 1) * sizeof(float), sizeof(float),
                                       (size_t)3, hipMemcpyDeviceToHost, h->stream)); // the pose block lives in the stripe
        CSLAM_HIP_TRY(hipStreamSynchronize(h->stream));
        double s = 0.0;
        for (float d : diag)
        {
            s += (double)d;
        }
        traces[i] = s;
    }
    return CSLAM_OK;
}

int cslam_ekf_batch_factor_status(cslam_ekf_batch_t h, int* flags)
{
    if (!h || !flags)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_factor_status: bad arguments");
    }
    int rc = h->use_device();
    if (rc || (rc = h->sync()))
    {
        return rc;
    }
    std::vector<int> f((size_t)2 * h->I);
    CSLAM_HIP_TRY(hipMemcpy(f.data(), h->dFlags, f.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->I; i++)
    {
        flags[i] = f[2 * i];
    }
    return CSLAM_OK;
}

int cslam_ekf_batch_run(cslam_ekf_batch_t h, int steps, const double* v, const double* swa, const float* Q, double wb, double dt,
                        const float* const* dZ, const int* const* d_idf, int m, const float* R)
{
    if (!h || steps < 0 || !v || !swa || !Q || !dZ || !d_idf || !R)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_run: bad arguments");
    }
    if (2 * m <= 16 || m > kLaMaxObs)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_run: m=%d outside the batched engine's 9..%d observations per update", m, kLaMaxObs);
    }
    for (int i = 0; i < h->I; i++)
    {
        if (!dZ[i] || !d_idf[i])
        {
            return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_run: instance %d has no inputs", i);
        }
    }
    if (steps == 0)
    {
        return CSLAM_OK;
    }
    int rc = h->use_device();
    if (rc)
    {
        return rc;
    }
    // The per-instance input pointers of this call go into the table generation the previous call does not use, with a
    // BLOCKING copy: the chain kernels read them on stream F, which is not ordered behind copies on the main stream.  The
    // generation was last used two calls ago: wait for that call's last kernel (the call in between stays in flight).
    h->tab_gen ^= 1;
    const int     g  = h->tab_gen;
    const float** zt = h->dZtab + (size_t)g * h->I;
    const int**   it = h->dIdftab + (size_t)g * h->I;
    if (h->gen_used[g])
    {
        CSLAM_HIP_TRY(hipEventSynchronize(h->ev_gen[g]));
    }
    CSLAM_HIP_TRY(hipMemcpy(zt, dZ, (size_t)h->I * sizeof(float*), hipMemcpyHostToDevice));
    CSLAM_HIP_TRY(hipMemcpy(it, d_idf, (size_t)h->I * sizeof(int*), hipMemcpyHostToDevice));
    const int pw = (h->quirks & CSLAM_Q_PREDICT_NM4) ? (h->n - 4) : (h->n - 3);
    auto      pp = [&](int t) {
        return PredictArgs<float>{1, (float)v[t], (float)swa[t], Q[0], Q[1], Q[2], Q[3], (float)wb, (float)dt, std::max(pw, 0)};
    };
    for (int t = 0; t < steps; t += 2)
    {
        LaBatchWin w;
        memset(&w, 0, sizeof(w));
        w.nu     = (t + 1 < steps) ? 2 : 1;
        w.ma     = m;
        w.mb     = m;
        w.Ztab   = zt;
        w.idftab = it;
        w.zoff_a = (long)t * 2 * m;
        w.ioff_a = (long)t * m;
        w.zoff_b = (long)(t + 1) * 2 * m;
        w.ioff_b = (long)(t + 1) * m;
        w.pp_a   = pp(t);
        w.pp_b   = w.nu == 2 ? pp(t + 1) : pp(t);
        for (int e = 0; e < 4; e++)
        {
            w.R[e] = R[e];
        }
        if ((rc = h->window(w)))
        {
            return rc;
        }
    }
    CSLAM_HIP_TRY(hipEventRecord(h->ev_gen[g], h->stream)); // (the chains of a window finish before its wide kernel does)
    h->gen_used[g] = true;
    return CSLAM_OK;
}

int cslam_ekf_batch_set_profiling(cslam_ekf_batch_t h, int every)
{
    if (!h || every < 0)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_set_profiling: bad arguments");
    }
    int rc = h->use_device();
    if (rc || (rc = h->sync()))
    {
        return rc;
    }
    h->prof_every = every;
    h->prof_seen  = 0;
    h->prof_used  = 0;
    while (every > 0 && h->prof_ev.size() < 256)
    {
        hipEvent_t a = nullptr, b = nullptr;
        CSLAM_HIP_TRY(hipEventCreate(&a));
        CSLAM_HIP_TRY(hipEventCreate(&b));
        h->prof_ev.push_back({a, b});
    }
    return CSLAM_OK;
}

int cslam_ekf_batch_get_pgemm_time(cslam_ekf_batch_t h, double* ms_sum, int* launches)
{
    if (!h || !ms_sum || !launches)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_get_pgemm_time: bad arguments");
    }
    int rc = h->use_device();
    if (rc || (rc = h->sync()))
    {
        return rc;
    }
    double s = 0.0;
    for (size_t i = 0; i < h->prof_used; i++)
    {
        float ms = 0.f;
        CSLAM_HIP_TRY(hipEventElapsedTime(&ms, h->prof_ev[i].first, h->prof_ev[i].second));
        s += ms;
    }
    *ms_sum   = s;
    *launches = (int)h->prof_used;
    return CSLAM_OK;
}

int cslam_ekf_batch_info(cslam_ekf_batch_t h, int* instances, int* n, long long* windows)
{
    if (!h)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_batch_info: null handle");
    }
    if (instances)
    {
        *instances = h->I;
    }
    if (n)
    {
        *n = h->n;
    }
    if (windows)
    {
        *windows = h->windows;
    }
    return CSLAM_OK;
}

} // extern "C"
