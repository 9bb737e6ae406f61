// cslam_ekf.hip -- host side of the EKF-SLAM engine behind the C ABI of include/cslam.h.
//
// One handle = one filter instance bound to one device; X and P live in HBM for the lifetime of the handle.
// update() is a chain of launches on the handle's stream: gather (+ the correction for pending panels), factor, gain
// (which also takes the pose stripe's share of the downdate), and the covariance downdate P -= W1 W1^T (the P-GEMM).
//
//   immediate mode (default): every update launches its own P-GEMM behind its gain kernel.
//   deferred mode (cslam_ekf_set_deferred, and always inside a sequential update): the covariance is held as
//     P = Ps - Wp Wp^T with up to kmax columns of W1 panels (and the rank-1 columns of heading observations) pending;
//     readers of P correct for them, ONE P-GEMM applies them all.
//   two-stream mode (CSLAM_PIPELINE=1, an experiment that is NOT the default: measured slower): the
//     pending columns' P-GEMM runs on a second stream underneath the next update's factor / gain chain.  What makes it
//     legal is the pose stripe Pv (ekf_kernels.hpp p_get): predicts and heading observations never touch Ps.
// Nothing returns to the host unless the caller asks (get_x / get_state / sync mode).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cslam_common.hpp"
#include "ekf_kernels.hpp"
#include "ekf_kernels_fast.hpp"
#include "ekf_lookahead.hpp"
#include "ekf_pgemm_limbs.hpp"
#include "ekf_pose_kernels.hpp"
#include "host_linalg.hpp"

namespace cslam
{
// the chain's go-ahead as a kernel of its own (only when the P-GEMM launch that should carry it did not happen)
__global__ void ekf_la_signal_kernel(unsigned* signal, unsigned add)
{
    if (threadIdx.x == 0)
    {
        atomicAdd(signal, add);
    }
}
// the wide half of a look-ahead window (ekf_lookahead.hpp: ekf_la_wide_body), one filter
__global__ void __launch_bounds__(128) ekf_la_wide_f32(LaWideArgs a)
{
    ekf_la_wide_body<1, 0>(a);
}
// ... with both updates of m = 32 observations (k = 64 known at compile time)
__global__ void __launch_bounds__(128) ekf_la_wide_f32_k64(LaWideArgs a)
{
    ekf_la_wide_body<1, 64>(a);
}
} // namespace cslam

using namespace cslam;

namespace
{

// Live engines of this process.  The look-ahead windows' fast hand-overs are waits INSIDE kernels (the chain kernel is
// launched early and waits for the blocks kernel; the wide kernel polls the chain's completion word).  They are only safe
// while the two streams of ONE engine have the device to themselves: with several engines, streams share the few hardware
// queues (a chain kernel spinning at the head of a queue blocks the kernel it waits for when that one sits behind another
// engine's waiting kernel in a second queue) and spinning wide kernels can hold every compute unit -- measured as 0.2 s
// time-outs with 8 co-running instances.  So with more than one engine alive: windows only if forced
// (CSLAM_LOOKAHEAD=1), and then with stream events only (ev_raw / ev_fb): a plain dependency graph, no waiting kernels.
std::atomic<int>& g_engines = ::cslam::live_engines();

constexpr int    kStagingSlots = 64;
constexpr size_t kLdsBudget    = 150 * 1024; // of the 160 KiB per CU, leave room for the small arrays

struct EkfBase
{
    virtual ~EkfBase() {}
    int         dtype    = CSLAM_F32;
    int         fuse_f64 = 1; // the f64 MFMA kernels take a held predict too (env CSLAM_FUSE_F64=0: its own launch)
    int         pgemm_limbs_req = -1, limbs_kmin_req = -1; // env CSLAM_PGEMM_LIMBS / CSLAM_LIMBS_KMIN (-1: default)
    int         xcd_queues_req = -1;                       // env CSLAM_XCD_QUEUES
    int         device   = 0;
    int         quirks   = CSLAM_Q_REF_EXACT;
    int         nmax     = 0; // max landmarks
    int         ncap     = 0; // 3 + 2*nmax
    int         ldp      = 0; // padded leading dimension / column count
    int         n        = 3;
    int         sync_mode = 1;
    int         seq_defer     = 1; // sequential update(): one P-GEMM per call (env CSLAM_SEQ_DEFER=0 restores m passes)
    int         lower         = 0; // block-lower storage of P (f32 default; env CSLAM_STORAGE=full|lower)
    int         pipeline      = 0; // P-GEMM of update t on stream B under the chain of update t+1 (env CSLAM_PIPELINE)
    int         pgemm_spare   = 16; // pipelined: workgroups the persistent P-GEMM grid leaves out (env CSLAM_PGEMM_SPARE)
    int         gather_corr_wide = 1; // a pending batch panel (<= 64 columns) corrected for inside the gather kernel (env CSLAM_GATHER_WIDE)
    int         pgemm_wgs     = 0;  // > 0: cap on the persistent P-GEMM grid (cslam_ekf_set_pgemm_workgroups: co-running instances)
    int         lookahead     = -1; // look-ahead windows (ekf_lookahead.hpp): -1 where they pay, env CSLAM_LOOKAHEAD=1 / 0 forces
    hipStream_t stream   = nullptr; // A: everything except the P-GEMM
    hipStream_t stream_b = nullptr; // B: the P-GEMM (== stream when not pipelined)

    virtual int init()                                                                        = 0;
    virtual int set_state(const void* X, int n, const void* P, int ldp)                        = 0;
    virtual int get_state(void* X, void* P, int ldp)                                           = 0;
    virtual int get_x(void* X, int cap)                                                        = 0;
    virtual int trace(double* tr)                                                              = 0;
    virtual int predict(double v, double swa, const void* Q, double wb, double dt)             = 0;
    virtual int update(const void* Z, int m, const void* R, const int* idf, int batch, bool on_device) = 0;
    virtual int augment(const void* Z, int q, const void* R)                                   = 0;
    virtual int observe_heading(double phi, int use)                                           = 0;
    virtual int associate(const void* Z, int m, const void* R, double g1, double g2, int* idf_out, int* kind) = 0;
    virtual int factor_status(int* flags, int clear)                                           = 0;
    virtual int set_profiling(int on)                                                          = 0;
    virtual int get_stage_times(double* ms, int* launches)                                     = 0;
    virtual int debug_last_update(void* PHT, void* S, void* G, void* W1, void* V, int* k)      = 0;
    virtual int set_deferred(int max_cols)                                                     = 0;
    virtual int do_flush()                                                                     = 0;
    virtual int resolve_predict()                                                              = 0;
    virtual void set_fuse_predict(int on)                                                      = 0;
    virtual int sync_all()                                                                     = 0;
    virtual int la_drain()                                                                     = 0;
};

template <typename T>
struct Ekf : EkfBase
{
    T*   dX = nullptr;
    T*   dP = nullptr;
    T*   dPv = nullptr; // pose stripe: columns 0..2 of P (always current; see p_get in ekf_kernels.hpp)
    T*   dWv = nullptr; // pose rows of the last update's W1 (3 x kcap), saved by the pose downdate before it zeroes them
    int* dPoseDone = nullptr; // ticket counters: [0] ekf_pose_step_kernel, [1] ekf_pose_downdate_kernel
    int* dSign     = nullptr; // per region: wcap column signs (heading columns with S < 0), then [2*wcap + r] their count
    int  hd_cols[2] = {0, 0}; // heading columns appended to each region since it became the pending store
    // the pending W1 store is two regions of wcap columns: `wcur` collects pending columns, the other one may still be
    // read by a P-GEMM in flight on stream B
    int        wcur = 0;
    unsigned   inflight_mask = 0; // regions an unfinished P-GEMM reads (cleared when stream A has waited for it)
    hipEvent_t ev_a2b = nullptr, ev_pgemm = nullptr;
    T*         last_slot = nullptr; // W1 of the last update
    // update workspace
    int  kcap  = 0;
    T*   dPHT  = nullptr;
    T*   dW1   = nullptr; // pending W1 panels, two regions of ldp x wcap
    T*   dY    = nullptr; // Y = H*Wp (kcap x wcap), correction of PHT under pending panels
    int  wcap  = 0;       // columns per region of dW1
    int  kp    = 0;       // pending columns (downdates not applied to P yet)
    int  defer_max = 0;   // > 0: keep up to this many pending columns across calls (cslam_ekf_set_deferred)
    T*   dS    = nullptr;
    T*   dG    = nullptr;
    T*   dSub  = nullptr; // (3 + 64) x 64 compact block of PHT (see ekf_gather_kernel)
    T*   dM    = nullptr; // 3 x 64: M = G G^T PHT[0:3,:]^T from ekf_factor_mfma_f32 (pose-stripe downdate in the gain kernel)
    bool m_valid = false; // the last factor launch produced dM
    bool g_from_gt  = false; // the last factor launch wrote only G^T (ekf_factor_mfma_f32): debug transposes it
    bool sub_valid = false;
    T*   dGt   = nullptr;
    T*   dV    = nullptr;
    T*   dt_   = nullptr;
    T*   dU    = nullptr; // u = G*(G^T V), the gain kernel's X update vector
    T*   dScrS = nullptr;
    T*   dScrG = nullptr;
    int* dFlags = nullptr; // [0] sticky, [1] last
    int* hFlags = nullptr; // pinned mirror
    // heading scratch: w, cp2, rrow (ldp each) + 2 scalars
    T* dHead = nullptr;
    // observation staging: pinned host ring + one device buffer
    int         mcap   = 0;
    void*       hStage = nullptr;
    void*       dStage = nullptr;
    hipEvent_t  stage_ev[kStagingSlots];
    bool        stage_ev_used[kStagingSlots];
    int         stage_next = 0;
    // tile list of the persistent symmetric downdate
    int*       dTicket     = nullptr; // two tile-ticket counters used alternately by successive P-GEMM launches
    // f32 P-GEMM on the bf16 matrix cores (ekf_pgemm_limbs.hpp): limb pairs per product (9 exact, 6, 0 = the f32 MFMA
    // kernel; env CSLAM_PGEMM_LIMBS), from how many columns on (env CSLAM_LIMBS_KMIN), the limb store and its size.
    // Off by default: correct and as accurate as the f32 MFMA kernel (tests), but measured no faster -- 120 - 132 us
    // against 115 at k = 128, N = 5000 -- see DESIGN.md 8.
    int        pgemm_limbs = 0;
    // the f32 P-GEMM draws its tiles from one queue per XCD over a Morton-ordered list (env CSLAM_XCD_QUEUES=1).  Off by
    // default: it cuts the HBM fetch traffic of a launch by a sixth (k = 64: 487 -> 435 MB, 1.04x the algorithmic bytes;
    // k = 128: 584 -> 483 MB) but not its time (81.7 vs 80.9 us, 115.3 vs 114.6 us), and the loops built on it came out
    // 0 - 4 % slower (profiles/r02_pmc_xcd_queues.txt)
    int        xcd_queues  = 0;
    int        limbs_kmin  = 65;
    uint4*     dWb         = nullptr;
    size_t     wb_bytes    = 0;
    int2*      dTilesM     = nullptr; // the tile list in Morton order, cut into eight segments (one per XCD) ...
    int*       dSegOff     = nullptr; // ... at these offsets (9), and two sets of eight ticket counters
    int*       dTicketX    = nullptr;
    int        tilesM_built = 0;
    int        limb_parity = 0;
    unsigned   launch_parity = 0;
    int        psym_nt = -1; // CSLAM_PSYM_NT=0|1: non-temporal P accesses in the P-GEMM (-1: by footprint)
    int2* dTiles      = nullptr;
    int   tiles_built = 0;
    int   n_sym_tiles = 0;
    int   num_cus     = 256;
    // status
    int sticky_host = 0; // flags raised by host-side decisions (FALLBACK/SKIPPED)
    int last_k      = 0;
    int kp_call_limit = 0; // pending columns a sequential update() may accumulate within the call
    // profiling
    int                     profiling = 0;
    unsigned                prof_count = 0;
    bool                    prof_sampled = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int>        ev_stage; // stage id of interval [2i, 2i+1]
    size_t                  ev_used = 0;

    ~Ekf() override
    {
        if (stream)
        {
            (void)hipStreamSynchronize(stream);
        }
        if (stream_b && stream_b != stream)
        {
            (void)hipStreamSynchronize(stream_b);
        }
        if (stream_f)
        {
            (void)hipStreamSynchronize(stream_f);
            (void)hipEventDestroy(ev_fb);
            (void)hipEventDestroy(ev_raw);
            (void)hipStreamDestroy(stream_f);
        }
        la_free();
        if (ev_a2b)
        {
            (void)hipEventDestroy(ev_a2b);
        }
        if (ev_pgemm)
        {
            (void)hipEventDestroy(ev_pgemm);
        }
        for (hipEvent_t e : ev_pool)
        {
            (void)hipEventDestroy(e);
        }
        for (int i = 0; i < kStagingSlots; i++)
        {
            if (stage_ev_used[i])
            {
                (void)hipEventDestroy(stage_ev[i]);
            }
        }
        (void)hipFree(dX);
        (void)hipFree(dP);
        (void)hipFree(dPv);
        (void)hipFree(dWv);
        (void)hipFree(dPoseDone);
        (void)hipFree(dSign);
        free_workspace();
        (void)hipFree(dFlags);
        (void)hipFree(dHead);
        (void)hipFree(dTicket);
        (void)hipFree(dWb);
        (void)hipFree(dTilesM);
        (void)hipFree(dSegOff);
        (void)hipFree(dTicketX);
        (void)hipFree(dPred);
        (void)hipFree(dAssoc);
        (void)hipFree(dAssocOut);
        (void)hipFree(dW1);
        (void)hipFree(dY);
        (void)hipFree(dTiles);
        (void)hipFree(dStage);
        if (hStage)
        {
            (void)hipHostFree(hStage);
        }
        if (hFlags)
        {
            (void)hipHostFree(hFlags);
        }
        if (stream_b && stream_b != stream)
        {
            (void)hipStreamDestroy(stream_b);
        }
        if (stream)
        {
            (void)hipStreamDestroy(stream);
        }
    }

    T* wbase(int region) const { return dW1 + (size_t)region * wcap * ldp; }

    void free_workspace()
    {
        (void)hipFree(dPHT);
        (void)hipFree(dS);
        (void)hipFree(dG);
        (void)hipFree(dSub);
        (void)hipFree(dM);
        dM = nullptr;
        (void)hipFree(dGt);
        (void)hipFree(dV);
        (void)hipFree(dt_);
        (void)hipFree(dU);
        (void)hipFree(dScrS);
        (void)hipFree(dScrG);
        dPHT = dS = dG = dSub = dGt = dV = dt_ = dU = dScrS = dScrG = nullptr;
    }

    int use_device() { CSLAM_HIP_TRY(hipSetDevice(device)); return CSLAM_OK; }

    int init() override
    {
        for (int i = 0; i < kStagingSlots; i++)
        {
            stage_ev_used[i] = false;
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if (xcd_queues_req >= 0)
        {
            xcd_queues = xcd_queues_req;
        }
        if (pgemm_limbs_req >= 0)
        {
            pgemm_limbs = (pgemm_limbs_req == 6 || pgemm_limbs_req == 9) ? pgemm_limbs_req : 0;
        }
        if (limbs_kmin_req >= 0)
        {
            limbs_kmin = std::max(57, limbs_kmin_req); // (at least four chunks of 16: k8 >= 57 rounds to 64)
        }
        if (pipeline)
        {
            // the chain (A) outranks the P-GEMM (B): its small kernels must get in while the P-GEMM fills the chip
            int lo = 0, hi = 0;
            CSLAM_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CSLAM_HIP_TRY(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));
            CSLAM_HIP_TRY(hipStreamCreateWithPriority(&stream_b, hipStreamNonBlocking, lo));
        }
        else
        {
            CSLAM_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            stream_b = stream;
        }
        CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_a2b, hipEventDisableTiming));
        CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_pgemm, hipEventDisableTiming));
        {
            hipDeviceProp_t prop;
            CSLAM_HIP_TRY(hipGetDeviceProperties(&prop, device));
            num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        size_t pbytes = (size_t)ldp * ldp * sizeof(T);
        CSLAM_HIP_TRY(hipMalloc(&dX, (size_t)ldp * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dP, pbytes));
        CSLAM_HIP_TRY(hipMemsetAsync(dX, 0, (size_t)ldp * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dP, 0, pbytes, stream));
        CSLAM_HIP_TRY(hipMalloc(&dPv, (size_t)3 * ldp * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dPv, 0, (size_t)3 * ldp * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMalloc(&dPoseDone, 4 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemsetAsync(dPoseDone, 0, 4 * sizeof(int), stream));
        CSLAM_HIP_TRY(hipMalloc(&dFlags, 2 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemsetAsync(dFlags, 0, 2 * sizeof(int), stream));
        CSLAM_HIP_TRY(hipHostMalloc(&hFlags, 2 * sizeof(int), hipHostMallocDefault));
        CSLAM_HIP_TRY(hipMalloc(&dHead, ((size_t)3 * ldp + 2) * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dHead, 0, ((size_t)3 * ldp + 2) * sizeof(T), stream));
        // allow the factor kernel its large dynamic LDS
        CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_factor_kernel<T>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (sizeof(T) == 4)
        {
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_factor_mfma_big_f32<128>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        else
        {
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_factor_mfma_f64<64>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_factor_mfma_f64<32>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_downdate_f64<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_downdate_f64<2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        CSLAM_HIP_TRY(hipMalloc(&dTicket, 2 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemsetAsync(dTicket, 0, 2 * sizeof(int), stream));
        if (const char* sv = getenv("CSLAM_LA_K64"))
        {
            la_k64 = atoi(sv) ? 1 : 0;
        }
        if (const char* sv = getenv("CSLAM_LA_WG_SIGNAL"))
        {
            la_wg_signal = atoi(sv) ? 1 : 0;
        }
        if (const char* sv = getenv("CSLAM_PSYM_NT"))
        {
            psym_nt = atoi(sv) ? 1 : 0;
        }
        rc = ensure_k(64);
        if (rc)
        {
            return rc;
        }
        rc = ensure_w(128);
        if (rc)
        {
            return rc;
        }
        rc = ensure_m(64);
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    // workspace for batches of up to k rows of H
    int ensure_k(int k)
    {
        if (k <= kcap)
        {
            return CSLAM_OK;
        }
        int newk = round_up(std::max(k, 2 * kcap), 8);
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        free_workspace();
        size_t pan = (size_t)ldp * newk * sizeof(T);
        size_t kk  = (size_t)newk * (newk + 1) * sizeof(T);
        CSLAM_HIP_TRY(hipMalloc(&dPHT, pan));
        CSLAM_HIP_TRY(hipMalloc(&dS, kk));
        CSLAM_HIP_TRY(hipMalloc(&dG, kk));
        if (dSub == nullptr)
        {
            CSLAM_HIP_TRY(hipMalloc(&dSub, (size_t)(3 + 64) * 64 * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dM, (size_t)3 * 128 * sizeof(T)));
        }
        CSLAM_HIP_TRY(hipMalloc(&dGt, kk));
        CSLAM_HIP_TRY(hipMalloc(&dScrS, kk));
        CSLAM_HIP_TRY(hipMalloc(&dScrG, kk));
        CSLAM_HIP_TRY(hipMalloc(&dV, (size_t)newk * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dt_, (size_t)newk * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dU, (size_t)newk * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dPHT, 0, pan, stream));
        (void)hipFree(dWv);
        dWv = nullptr;
        CSLAM_HIP_TRY(hipMalloc(&dWv, (size_t)3 * newk * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dWv, 0, (size_t)3 * newk * sizeof(T), stream));
        kcap = newk;
        if (dY)
        {
            (void)hipFree(dY);
            dY = nullptr;
        }
        if (wcap > 0)
        {
            CSLAM_HIP_TRY(hipMalloc(&dY, (size_t)kcap * wcap * sizeof(T)));
        }
        return CSLAM_OK;
    }

    // room for `cols` pending columns per region; applies what is pending first when the buffer has to move
    int ensure_w(int cols)
    {
        cols = round_up(cols, 8);
        if (cols <= wcap)
        {
            return CSLAM_OK;
        }
        int rc = flush();
        if (rc || (rc = sync_all()))
        {
            return rc;
        }
        int neww = round_up(std::max(cols, 2 * wcap), 8);
        (void)hipFree(dW1);
        (void)hipFree(dY);
        dW1 = nullptr;
        dY  = nullptr;
        CSLAM_HIP_TRY(hipMalloc(&dW1, (size_t)2 * ldp * neww * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dY, (size_t)std::max(kcap, 8) * neww * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dW1, 0, (size_t)2 * ldp * neww * sizeof(T), stream));
        (void)hipFree(dSign);
        dSign = nullptr;
        CSLAM_HIP_TRY(hipMalloc(&dSign, ((size_t)2 * neww + 2) * sizeof(int)));
        CSLAM_HIP_TRY(hipMemsetAsync(dSign, 0, ((size_t)2 * neww + 2) * sizeof(int), stream));
        hd_cols[0] = hd_cols[1] = 0;
        wcap      = neww;
        wcur      = 0;
        last_slot = nullptr;
        return CSLAM_OK;
    }

    int sync_all() override
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        if (stream_b != stream)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream_b));
        }
        if (stream_f)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream_f));
        }
        inflight_mask = 0;
        return CSLAM_OK;
    }

    // stream A waits until every P-GEMM launched so far (stream B) has completed: needed before anything reads or
    // writes Ps, or writes a W1 region such a P-GEMM reads
    int wait_pgemm()
    {
        if (inflight_mask != 0 && stream_b != stream)
        {
            CSLAM_HIP_TRY(hipStreamWaitEvent(stream, ev_pgemm, 0));
        }
        inflight_mask = 0;
        return CSLAM_OK;
    }

    // before stream A writes into W1 region r
    int own_region(int r)
    {
        return (inflight_mask & (1u << r)) ? wait_pgemm() : CSLAM_OK;
    }

    // launch the P-GEMM of every pending column (slam.h:260 is linear in the panels: ONE pass with k = kp) on stream B,
    // ordered behind everything enqueued on stream A so far; the pending store moves on to the other region
    int flush()
    {
        int rc = launch_pose_queue(); // queued heading steps write pending columns of this region
        if (rc)
        {
            return rc;
        }
        if (kp == 0)
        {
            return CSLAM_OK;
        }
        if ((rc = use_device()))
        {
            return rc;
        }
        T*        W   = wbase(wcur);
        const int kp8 = round_up(kp, 8);
        // (the shipped f32 P-GEMM bounds its W1 buffer resource at kp columns: the hardware returns zeros beyond, no
        // padding needed; the other kernels read whole blocks of 8 columns)
        if (kp8 > kp && !psym4_takes(kp8))
        {
            CSLAM_HIP_TRY(hipMemset2DAsync(W + (size_t)kp * ldp, (size_t)ldp * sizeof(T), 0,
                                           (size_t)round_up(n, kTile) * sizeof(T), (size_t)(kp8 - kp), stream));
        }
        if (stream_b != stream)
        {
            CSLAM_HIP_TRY(hipEventRecord(ev_a2b, stream));
            CSLAM_HIP_TRY(hipStreamWaitEvent(stream_b, ev_a2b, 0));
        }
        if (hd_cols[wcur] > 0) // exceptional heading columns (S < 0) enter with the opposite sign: exits at once otherwise
        {
            if ((rc = launch_negcol_fix(W, kp, stream_b)))
            {
                return rc;
            }
        }
        if ((rc = prof_begin(CSLAM_STAGE_DOWNDATE, stream_b)) || (rc = launch_downdate(W, kp, stream_b)) ||
            (rc = prof_end(CSLAM_STAGE_DOWNDATE, stream_b)))
        {
            return rc;
        }
        if (stream_b != stream)
        {
            CSLAM_HIP_TRY(hipEventRecord(ev_pgemm, stream_b));
            inflight_mask |= 1u << wcur;
        }
        wcur ^= 1;
        kp = 0;
        // the other region becomes the pending store: a still older P-GEMM may be reading it
        if ((rc = own_region(wcur)))
        {
            return rc;
        }
        if (hd_cols[wcur] > 0) // its column signs belong to columns that have been applied
        {
            if (stream_b != stream) // (single stream: ekf_negcol_fix_kernel clears them itself when it had work)
            {
                CSLAM_HIP_TRY(hipMemsetAsync(dSign + (size_t)wcur * wcap, 0, (size_t)wcap * sizeof(int), stream));
                CSLAM_HIP_TRY(hipMemsetAsync(dSign + (size_t)2 * wcap + wcur, 0, sizeof(int), stream));
            }
            hd_cols[wcur] = 0;
        }
        return CSLAM_OK;
    }

    // the tile list in Morton order, cut into eight equal segments (one per XCD), and the 2 x 8 ticket counters
    int ensure_tiles_morton(int tiles, hipStream_t stream)
    {
        if (tilesM_built == tiles)
        {
            return CSLAM_OK;
        }

        // Morton order over the lower triangle, eight equal segments
        auto spread = [](unsigned v) {
            v &= 0xFFFF;
            v = (v | (v << 8)) & 0x00FF00FF;
            v = (v | (v << 4)) & 0x0F0F0F0F;
            v = (v | (v << 2)) & 0x33333333;
            v = (v | (v << 1)) & 0x55555555;
            return v;
        };
        std::vector<std::pair<unsigned, int2>> keyed;
        keyed.reserve((size_t)tiles * (tiles + 1) / 2);
        for (int tj = 0; tj < tiles; tj++)
        {
            for (int ti = tj; ti < tiles; ti++)
            {
                keyed.push_back({spread((unsigned)ti) | (spread((unsigned)tj) << 1), make_int2(ti, tj)});
            }
        }
        std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        std::vector<int2> hl(keyed.size());
        for (size_t i = 0; i < keyed.size(); i++)
        {
            hl[i] = keyed[i].second;
        }
        int off[9];
        for (int sgi = 0; sgi <= 8; sgi++)
        {
            off[sgi] = (int)((size_t)hl.size() * sgi / 8);
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(dTilesM);
        dTilesM = nullptr;
        CSLAM_HIP_TRY(hipMalloc(&dTilesM, hl.size() * sizeof(int2)));
        CSLAM_HIP_TRY(hipMemcpy(dTilesM, hl.data(), hl.size() * sizeof(int2), hipMemcpyHostToDevice));
        if (dSegOff == nullptr)
        {
            CSLAM_HIP_TRY(hipMalloc(&dSegOff, 9 * sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dTicketX, 16 * sizeof(int)));
            CSLAM_HIP_TRY(hipMemset(dTicketX, 0, 16 * sizeof(int)));
        }
        CSLAM_HIP_TRY(hipMemcpy(dSegOff, off, sizeof(off), hipMemcpyHostToDevice));
        tilesM_built = tiles;
        return CSLAM_OK;
    }

    // k8 columns go through ekf_downdate_psym4_f32 (see launch_downdate)
    bool psym4_takes(int k8) const
    {
        return sizeof(T) == 4 && (k8 <= 128 || limbs_take(k8)) && lower && ldp < 32768;
    }
    // ... or through ekf_downdate_psym5_bf16 (which pads its own limb store)
    bool limbs_take(int k8) const
    {
        return sizeof(T) == 4 && pgemm_limbs > 0 && k8 >= limbs_kmin && k8 <= 256 && lower && ldp < 32768;
    }

    int launch_negcol_fix(T* W, int kcols, hipStream_t st)
    {
        const int tiles = round_up(n, kTile) / kTile;
        int       rc    = CSLAM_OK;
        if (lower && (rc = ensure_tile_list(tiles)))
        {
            return rc;
        }
        const int nt = lower ? n_sym_tiles : tiles * tiles;
        hipLaunchKernelGGL(ekf_negcol_fix_kernel<T>, dim3(std::min(nt, 2 * num_cus)), dim3(256), 0, st, dP, ldp, n, W, ldp,
                           kcols, dSign + (size_t)wcur * wcap, dSign + (size_t)2 * wcap + wcur,
                           lower ? dTiles : (const int2*)nullptr, nt, tiles, dPoseDone + 2, (stream_b == stream) ? 1 : 0);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    // everything applied and Ps quiescent as far as stream A is concerned
    int flush_wait()
    {
        int rc = flush();
        return rc ? rc : wait_pgemm();
    }

    size_t slot_bytes(int mc) const { return (size_t)mc * (2 * sizeof(T) + sizeof(int)); }

    int ensure_m(int m)
    {
        if (m <= mcap)
        {
            return CSLAM_OK;
        }
        int newm = std::max(m, 2 * mcap);
        if (int rc = la_drain()) // (queued updates read the ring that is about to move)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        if (stream_f)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream_f));
        }
        if (hStage)
        {
            (void)hipHostFree(hStage);
            hStage = nullptr;
        }
        (void)hipFree(dStage);
        dStage = nullptr;
        CSLAM_HIP_TRY(hipHostMalloc(&hStage, slot_bytes(newm) * kStagingSlots, hipHostMallocDefault));
        // (a device slot per host slot: queued look-ahead updates read their inputs up to two calls later; a slot comes round
        // again kStagingSlots calls later, stream-ordered behind every kernel that read it)
        CSLAM_HIP_TRY(hipMalloc(&dStage, slot_bytes(newm) * kStagingSlots));
        mcap = newm;
        return CSLAM_OK;
    }

    // copies (Z, idf) of one call into a pinned slot and enqueues the H2D copy; returns device pointers
    int stage_obs(const void* Z, const int* idf, int m, const T** dZ, const int** dIdf)
    {
        int rc = ensure_m(m);
        if (rc)
        {
            return rc;
        }
        int slot = stage_next;
        stage_next = (stage_next + 1) % kStagingSlots;
        if (stage_ev_used[slot])
        {
            CSLAM_HIP_TRY(hipEventSynchronize(stage_ev[slot]));
        }
        else
        {
            CSLAM_HIP_TRY(hipEventCreateWithFlags(&stage_ev[slot], hipEventDisableTiming));
            stage_ev_used[slot] = true;
        }
        unsigned char* hs = static_cast<unsigned char*>(hStage) + slot_bytes(mcap) * slot;
        size_t         zb = (size_t)m * 2 * sizeof(T);
        memcpy(hs, Z, zb);
        memcpy(hs + zb, idf, (size_t)m * sizeof(int));
        unsigned char* ds = static_cast<unsigned char*>(dStage) + slot_bytes(mcap) * slot;
        CSLAM_HIP_TRY(hipMemcpyAsync(ds, hs, zb + (size_t)m * sizeof(int), hipMemcpyHostToDevice, stream));
        CSLAM_HIP_TRY(hipEventRecord(stage_ev[slot], stream));
        *dZ   = reinterpret_cast<const T*>(ds);
        *dIdf = reinterpret_cast<const int*>(ds + zb);
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- state transfer
    int set_state(const void* X, int nn, const void* P, int ldph) override
    {
        if (!X || !P || nn < 3 || nn > ncap || ((nn - 3) & 1) || ldph < nn)
        {
            return fail(CSLAM_ERR_BAD_ARG, "set_state: bad n=%d (cap %d) or ldp=%d", nn, ncap, ldph);
        }
        int rc = sync_all();
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(dX, X, (size_t)nn * sizeof(T), hipMemcpyHostToDevice, stream));
        CSLAM_HIP_TRY(hipMemcpy2DAsync(dP, (size_t)ldp * sizeof(T), P, (size_t)ldph * sizeof(T), (size_t)nn * sizeof(T),
                                       (size_t)nn, hipMemcpyHostToDevice, stream));
        // the pose stripe = columns 0..2 of P (contiguous in the column-major buffer)
        CSLAM_HIP_TRY(hipMemcpyAsync(dPv, dP, (size_t)3 * ldp * sizeof(T), hipMemcpyDeviceToDevice, stream));
        // panels: rows beyond the new n must read as zero (the tuned gain kernel relies on it)
        kp        = 0; // a new state discards updates that were never applied
        wcur      = 0;
        last_slot = nullptr;
        hd_cols[0] = hd_cols[1] = 0;
        CSLAM_HIP_TRY(hipMemsetAsync(dSign, 0, ((size_t)2 * wcap + 2) * sizeof(int), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dPHT, 0, (size_t)ldp * kcap * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dW1, 0, (size_t)2 * ldp * wcap * sizeof(T), stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        n = nn;
        return CSLAM_OK;
    }

    int get_state(void* X, void* P, int ldph) override
    {
        if (ldph < n && P)
        {
            return fail(CSLAM_ERR_BAD_ARG, "get_state: ldp=%d < n=%d", ldph, n);
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if (P && (rc = flush_wait()))
        {
            return rc;
        }
        if (X)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(X, dX, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (P)
        {
            if (lower)
            {
                const int g = (n + 31) / 32;
                hipLaunchKernelGGL(ekf_mirror_upper_kernel<T>, dim3(g, g), dim3(256), 0, stream, dP, ldp, n);
                CSLAM_HIP_TRY(hipGetLastError());
            }
            // rows / columns 0..2 of the buffer come from the pose stripe
            hipLaunchKernelGGL(ekf_patch_pose_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, stream, dP, dPv, ldp, n);
            CSLAM_HIP_TRY(hipGetLastError());
            CSLAM_HIP_TRY(hipMemcpy2DAsync(P, (size_t)ldph * sizeof(T), dP, (size_t)ldp * sizeof(T),
                                           (size_t)n * sizeof(T), (size_t)n, hipMemcpyDeviceToHost, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int get_x(void* X, int cap) override
    {
        if (!X || cap < n)
        {
            return fail(CSLAM_ERR_BAD_ARG, "get_x: capacity %d < n=%d", cap, n);
        }
        return get_state(X, nullptr, 0);
    }

    int trace(double* tr) override
    {
        if (!tr)
        {
            return fail(CSLAM_ERR_BAD_ARG, "trace: null");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if ((rc = flush_wait()))
        {
            return rc;
        }
        std::vector<T> diag((size_t)n);
        CSLAM_HIP_TRY(hipMemcpy2DAsync(diag.data(), sizeof(T), dP, ((size_t)ldp + 1) * sizeof(T), sizeof(T), (size_t)n,
                                       hipMemcpyDeviceToHost, stream));
        // the pose block lives in the stripe
        CSLAM_HIP_TRY(hipMemcpy2DAsync(diag.data(), sizeof(T), dPv, ((size_t)ldp + 1) * sizeof(T), sizeof(T), (size_t)3,
                                       hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        double s = 0.0;
        for (T d : diag)
        {
            s += (double)d;
        }
        *tr = s;
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- profiling
    // mode 1: every stage; 2: every P-GEMM launch; 3: one P-GEMM launch in sixteen (an event pair costs about 11 us of
    // stream time around the kernel it brackets -- rocprofv3 trace: 5.9 us before, 5.6 us after -- so the timed region
    // of the bench samples instead of bracketing every launch)
    bool prof_skip(int stage, bool begin)
    {
        if (!profiling)
        {
            return true;
        }
        if (profiling >= 2 && stage != CSLAM_STAGE_DOWNDATE)
        {
            return true;
        }
        if (profiling == 3 || profiling == 4)
        {
            if (begin)
            {
                prof_sampled = (prof_count++ % (profiling == 3 ? 16u : 4u)) == 0;
            }
            return !prof_sampled;
        }
        return false;
    }
    int prof_begin(int stage, hipStream_t st = nullptr)
    {
        st = st ? st : stream;
        if (prof_skip(stage, true))
        {
            return CSLAM_OK;
        }
        if (ev_used + 2 > ev_pool.size())
        {
            for (int i = 0; i < 2; i++)
            {
                hipEvent_t e;
                CSLAM_HIP_TRY(hipEventCreate(&e));
                ev_pool.push_back(e);
            }
        }
        ev_stage.resize(ev_pool.size() / 2);
        ev_stage[ev_used / 2] = stage;
        CSLAM_HIP_TRY(hipEventRecord(ev_pool[ev_used], st));
        return CSLAM_OK;
    }
    int prof_end(int stage, hipStream_t st = nullptr)
    {
        st = st ? st : stream;
        if (prof_skip(stage, false))
        {
            return CSLAM_OK;
        }
        CSLAM_HIP_TRY(hipEventRecord(ev_pool[ev_used + 1], st));
        ev_used += 2;
        return CSLAM_OK;
    }
    int set_profiling(int on) override
    {
        if (int rc = la_drain())
        {
            return rc;
        }
        if (int rc = sync_all())
        {
            return rc;
        }
        profiling  = on;
        ev_used    = 0;
        prof_count = 0;
        return CSLAM_OK;
    }
    int get_stage_times(double* ms, int* launches) override
    {
        if (!ms || !launches)
        {
            return fail(CSLAM_ERR_BAD_ARG, "get_stage_times: null");
        }
        if (int rc = la_drain())
        {
            return rc;
        }
        if (int rc = sync_all())
        {
            return rc;
        }
        for (int s = 0; s < CSLAM_N_STAGES; s++)
        {
            ms[s]       = 0.0;
            launches[s] = 0;
        }
        for (size_t i = 0; i + 1 < ev_used; i += 2)
        {
            float t = 0.f;
            CSLAM_HIP_TRY(hipEventElapsedTime(&t, ev_pool[i], ev_pool[i + 1]));
            int s = ev_stage[i / 2];
            ms[s] += (double)t;
            launches[s] += 1;
        }
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- data association (EKF.cpp:131-144, 235-326)
    T*   dAssoc    = nullptr; // 8 scalars per feature
    int  assoc_cap = 0;
    int* dAssocOut = nullptr; // idf[m], kind[m]
    int  assoc_mcap = 0;
    int associate(const void* Zv, int m, const void* Rv, double g1, double g2, int* idf_out, int* kind_out) override
    {
        if (m < 0 || !Rv || (m > 0 && (!Zv || !idf_out || !kind_out)))
        {
            return fail(CSLAM_ERR_BAD_ARG, "associate: bad arguments (m=%d)", m);
        }
        if (m == 0)
        {
            return CSLAM_OK;
        }
        int rc = use_device();
        if (rc || (rc = flush_wait()))
        {
            return rc;
        }
        const int nf = (n - 3) / 2;
        if (nf > assoc_cap)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(dAssoc);
            dAssoc = nullptr;
            CSLAM_HIP_TRY(hipMalloc(&dAssoc, (size_t)std::max(nf, 1) * 8 * sizeof(T)));
            assoc_cap = nf;
        }
        if (m > assoc_mcap)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(dAssocOut);
            dAssocOut = nullptr;
            CSLAM_HIP_TRY(hipMalloc(&dAssocOut, (size_t)2 * m * sizeof(int)));
            assoc_mcap = m;
        }
        const T* R  = static_cast<const T*>(Rv);
        const T* dZ = nullptr;
        const int* dummy = nullptr;
        std::vector<int> zero_idf((size_t)m, 1);
        if ((rc = stage_obs(Zv, zero_idf.data(), m, &dZ, &dummy)))
        {
            return rc;
        }
        if (nf > 0)
        {
            hipLaunchKernelGGL(ekf_assoc_feature_kernel<T>, dim3((nf + 255) / 256), dim3(256), 0, stream, dX, dP, dPv, ldp, n,
                               R[0], R[1], R[2], R[3], lower, dAssoc);
            CSLAM_HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(ekf_assoc_scan_kernel<T>, dim3(m), dim3(64), 0, stream, dAssoc, nf, dZ, m, (T)g1, (T)g2,
                           dAssocOut, dAssocOut + m);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipMemcpyAsync(idf_out, dAssocOut, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipMemcpyAsync(kind_out, dAssocOut + m, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- predict (EKF.cpp:406-455) / heading (EKF.cpp:328-352)
    // Control steps are accepted and held back until something needs their result:
    //   predict()          is held in `pp`;
    //   observe_heading()  joins the held predict into ONE step of the pose queue `seq` (the reference's driver calls
    //                      the two back to back on every control step, test/main.cpp:165-168); up to kPoseSeqMax steps
    //                      queue up and run in ONE ekf_pose_step_kernel launch (they only touch the pose stripe Pv, X and
    //                      the pending store: see ekf_pose_kernels.hpp);
    //   a batch update on the fast path (16 < k <= 64) applies a held predict on the fly in its gather / factor /
    //                      gain kernels and commits it (PredictArgs in ekf_kernels.hpp);
    //   anything else that reads X or P launches what is queued first (resolve_predict).
    // CSLAM_FUSE_PREDICT=0 launches every predict / heading at once.
    PredictArgs<T> pp{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, 0};
    PoseSeq<T>     seq{};
    T*             dPred = nullptr; // 16 scalars: factor kernel -> gain kernel (see FactorArgs::pred_out)
    bool           fuse_now = false; // the batch in flight consumes pp
    int            fuse_predict = 1;

    int predict(double v, double swa, const void* Qv, double wb, double dt) override
    {
        if (!Qv)
        {
            return fail(CSLAM_ERR_BAD_ARG, "predict: Q is null");
        }
        int rc = CSLAM_OK;
        if (pp.valid && (rc = queue_step(pp, HeadingArgs<T>{0, (T)0, (T)0}))) // two predicts in a row
        {
            return rc;
        }
        const T* Q = static_cast<const T*>(Qv);
        int      w = 0;
        if (n > 3)
        {
            w = (quirks & CSLAM_Q_PREDICT_NM4) ? (n - 4) : (n - 3);
        }
        pp = PredictArgs<T>{1, (T)v, (T)swa, Q[0], Q[1], Q[2], Q[3], (T)wb, (T)dt, std::max(w, 0)};
        if (!fuse_predict)
        {
            return resolve_predict();
        }
        return CSLAM_OK;
    }

    void set_fuse_predict(int on) override { fuse_predict = on; }

    // append one control step (predict and / or heading) to the pose queue
    int queue_step(const PredictArgs<T>& p, const HeadingArgs<T>& hd)
    {
        int rc = la_drain(); // (queued look-ahead updates come before this control step)
        if (rc)
        {
            return rc;
        }
        if (seq.count == kPoseSeqMax && (rc = launch_pose_queue()))
        {
            return rc;
        }
        int col = -1;
        if (hd.valid && n > 3)
        {
            // the rank-1 downdate -p p^T / S of the map block is one more pending column
            if (kp + 1 > wcap)
            {
                if ((rc = launch_pose_queue()) || (rc = flush()))
                {
                    return rc;
                }
            }
            if ((rc = own_region(wcur)))
            {
                return rc;
            }
            col = kp;
            kp += 1;
            hd_cols[wcur] += 1;
        }
        const int s = seq.count++;
        seq.pp[s]   = p;
        seq.hd[s]   = hd;
        seq.col[s]  = col;
        if (&p == &pp)
        {
            pp.valid = 0;
        }
        return CSLAM_OK;
    }

    int launch_pose_queue()
    {
        if (seq.count == 0)
        {
            return CSLAM_OK;
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const int n_pad = round_up(n, kTile);
        hipLaunchKernelGGL(ekf_pose_step_kernel<T>, dim3((n_pad + 255) / 256), dim3(256), 0, stream, dX, dPv, ldp, n, n_pad,
                           seq, wbase(wcur), ldp, dHead, dSign + (size_t)wcur * wcap, dSign + (size_t)2 * wcap + wcur,
                           dPoseDone);
        CSLAM_HIP_TRY(hipGetLastError());
        seq.count = 0;
        return CSLAM_OK;
    }

    // everything queued (and the held predict) runs now
    int resolve_predict() override
    {
        int rc = la_drain();
        if (rc)
        {
            return rc;
        }
        if (pp.valid && (rc = queue_step(pp, HeadingArgs<T>{0, (T)0, (T)0})))
        {
            return rc;
        }
        return launch_pose_queue();
    }

    // ---------------------------------------------------------------- update
    int launch_factor(const T* dZ, const int* dIdf, int m, const T* R)
    {
        FactorArgs<T> a;
        a.X   = dX;
        a.n   = n;
        a.Z   = dZ;
        a.idf = dIdf;
        a.m   = m;
        for (int i = 0; i < 4; i++)
        {
            a.R[i] = R[i];
        }
        a.PHT      = dPHT;
        a.ldw      = ldp;
        a.dS       = dS;
        a.dG       = dG;
        a.dGt      = dGt;
        a.dV       = dV;
        a.dt       = dt_;
        a.flags    = dFlags;
        a.scratchS = dScrS;
        a.scratchG = dScrG;
        a.textbook = (quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 0 : 1;
        a.stamps   = nullptr;
        a.sub      = sub_valid ? dSub : nullptr;
        a.dM       = nullptr;
        m_valid    = false;
        g_from_gt  = false;
        a.pp       = fuse_now ? pp : PredictArgs<T>{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, 0};
        a.P3       = dPv; // the pose block lives in the stripe
        a.ldp3     = ldp;
        a.pred_out = dPred;
        a.lds_S    = 1;
        a.lds_G    = 1;
        return launch_factor_args(a, dU, stream);
    }

    // dispatch on k (a.m); du: the X-update vector u = G (G^T V)
    int launch_factor_args(FactorArgs<T>& a, T* du, hipStream_t st)
    {
        const int k = 2 * a.m, m = a.m;
        const bool own = (a.dM == nullptr); // the handle's own workspace: record what the gain kernel will find
        // Which kernel factorises S (all of them produce G^T, t, u; the tuned ones also M for the gain kernel's
        // pose-stripe downdate):
        //   k <= 16            ekf_factor_small_kernel: one wave, a row per lane, v_readlane broadcasts
        //   16 < k <= 64       ekf_factor_mfma_f32 / _f64: rank-1 updates on the matrix cores
        //   64 < k <= 128 f32  ekf_factor_mfma_big_f32: four 32-wide blocks
        //   beyond             ekf_factor_kernel: the general LDS / global-scratch form
        if (k <= 4)
        {
            a.dM    = own ? dM : a.dM;
            m_valid = own ? true : m_valid;
            hipLaunchKernelGGL((ekf_factor_small_kernel<T, 4>), dim3(1), dim3(256), 0, st, a, du);
        }
        else if (k <= 16)
        {
            a.dM    = own ? dM : a.dM;
            m_valid = own ? true : m_valid;
            hipLaunchKernelGGL((ekf_factor_small_kernel<T, 16>), dim3(1), dim3(256), 0, st, a, du);
        }
        else if (k <= 64)
        {
            a.dM      = own ? dM : a.dM;
            m_valid   = own ? true : m_valid;
            g_from_gt = own ? true : g_from_gt;
            if constexpr (std::is_same<T, float>::value)
            {
                if (k <= 32)
                {
                    hipLaunchKernelGGL((ekf_factor_mfma_f32<32>), dim3(1), dim3(256), 0, st, a, du);
                }
                else
                {
                    hipLaunchKernelGGL((ekf_factor_mfma_f32<64>), dim3(1), dim3(256), 0, st, a, du);
                }
            }
            else
            {
                auto lds64 = [](int K) {
                    return (size_t)(K * (K + 1) + (3 + K) * (K + 1) + (K / 2) * 10 + 6 * K) * sizeof(double) +
                           (size_t)(K / 2 + 4) * sizeof(int) + 16;
                };
                if (k <= 32)
                {
                    hipLaunchKernelGGL((ekf_factor_mfma_f64<32>), dim3(1), dim3(256), lds64(32), st, a, du);
                }
                else
                {
                    hipLaunchKernelGGL((ekf_factor_mfma_f64<64>), dim3(1), dim3(256), lds64(64), st, a, du);
                }
            }
        }
        else if (std::is_same<T, float>::value && k <= 128)
        {
            if constexpr (std::is_same<T, float>::value)
            {
                constexpr int K   = 128;
                const size_t  lds = (size_t)(K * (K + 1) + (3 + K) * (K + 1) + (K / 2) * 10 + 6 * K) * sizeof(float) +
                                   (size_t)(K / 2 + 4) * sizeof(int) + 16;
                a.dM      = own ? dM : a.dM;
                m_valid   = own ? true : m_valid;
                g_from_gt = own ? true : g_from_gt;
                hipLaunchKernelGGL((ekf_factor_mfma_big_f32<128>), dim3(1), dim3(256), lds, st, a, du);
            }
        }
        else
        {
            size_t mat   = (size_t)k * (k + 1) * sizeof(T);
            size_t small = ((size_t)m * 10 + k) * sizeof(T) + ((size_t)m + 4) * sizeof(int) + 64;
            a.lds_S      = (mat + small <= kLdsBudget) ? 1 : 0;
            a.lds_G      = (a.lds_S && 2 * mat + small <= kLdsBudget) ? 1 : 0;
            size_t lds   = small + (a.lds_S ? mat : 0) + (a.lds_G ? mat : 0);
            hipLaunchKernelGGL(ekf_factor_kernel<T>, dim3(1), dim3(kFactorThreads), lds, st, a);
        }
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    // W1 of this update goes to `slot` (n_pad x k8 columns of the pending store); then the pose stripe takes its share
    // of the downdate at once and the panel's pose rows are zeroed (ekf_pose_downdate_kernel)
    int launch_gain(int k, T* slot, const T* Gt = nullptr, const T* U = nullptr, const T* M = nullptr)
    {
        // (Gt / U / M: the factor outputs to apply -- the handle's own workspace unless a look-ahead window passes a slot's)
        const int n_pad    = round_up(n, kTile);
        pose_fused_in_gain = false;
        if (Gt == nullptr)
        {
            Gt = dGt;
            U  = dU;
            M  = m_valid ? dM : nullptr;
        }
        if (!launch_gain_fast(k, n_pad, slot, Gt, U, M))
        {
            // the general vector-unit form (f32 beyond k = 128, f64 beyond k = 64): no X vector u, no fused pose downdate
            hipLaunchKernelGGL(ekf_gain_kernel<T>, dim3(n_pad / 64), dim3(256), 0, stream, dPHT, ldp, n, n_pad, k, dGt, dt_,
                               slot, dX);
        }
        CSLAM_HIP_TRY(hipGetLastError());
        if (pose_fused_in_gain)
        {
            return CSLAM_OK; // ekf_panel_mfma_f32 applied the pose-stripe downdate and zeroed the panel's pose rows itself
        }
        hipLaunchKernelGGL(ekf_pose_downdate_kernel<T>, dim3((n + 63) / 64), dim3(256), 0, stream, slot, ldp, k,
                           round_up(k, 8), n, dPv, ldp, dWv, dPoseDone + 1);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }
    bool pose_fused_in_gain = false;

    int  launch_downdate(const T* W, int k, hipStream_t st);
    bool launch_corr_fast(int k, const T* Wp, int kc);          // PHT -= Wp*Y^T on MFMA (f32)
    int  ensure_tile_list(int tiles);
    bool launch_gain_fast(int k, int n_pad, T* slot, const T* Gt, const T* U, const T* M); // MFMA gain

    // one batch of m observations with device-resident Z / idf (slam.h:235-266 via EKF.cpp:93-129).
    // keep_pending: never start this update's (or any pending) P-GEMM inside the call (sequential mode).
    int batch_on_device(const T* dZ, const int* dIdf, int m, const T* R, bool keep_pending)
    {
        const int k  = 2 * m;
        int       rc = ensure_k(k);
        if (rc)
        {
            return rc;
        }
        // a few pending columns (heading observations) are corrected for inside the gather kernel: the fast path stays
        // ... and so is one deferred batch panel (up to kGatherCorrMax columns), by the kernel's wider form
        const bool small_corr = kp > 0 && kp <= (gather_corr_wide ? kGatherCorrMax : kGatherCorr) && !pipeline;
        const bool wide_corr  = small_corr && kp > kGatherCorr;
        // a pending predict() rides along when this batch takes the (non-pipelined) fast path
        fuse_now = pp.valid && (sizeof(T) == 4 || fuse_f64) && !pipeline && !keep_pending && k > 16 && k <= 64;
        if ((rc = fuse_now ? launch_pose_queue() : resolve_predict())) // (queued control steps come first either way)
        {
            return rc;
        }
        if (fuse_now && dPred == nullptr)
        {
            CSLAM_HIP_TRY(hipMalloc(&dPred, 16 * sizeof(T)));
        }
        // the general gain kernels and ekf_pose_downdate_kernel write whole blocks of 8 columns of the slot: the slot starts
        // at column kp (heading columns make kp any number), so kp + round_up(k, 8) must stay inside the region
        const int k8w = round_up(k, 8);
        if ((rc = ensure_w(std::max(k8w, kp_call_limit)))) // (may flush and move the store)
        {
            return rc;
        }
        // Pipelined: the pending columns' P-GEMM starts right behind this update's gather and runs under its chain.
        // It is held back while an explicit deferral window (cslam_ekf_set_deferred) still has room for this panel.
        const bool overlap = pipeline && !keep_pending && kp > 0 && (defer_max == 0 || kp + k > defer_max);
        // pending columns stay pending through this update while they fit the window: the explicit deferral window, the
        // sequential call's own columns, else the store (e.g. heading columns in immediate mode: applied together with
        // this update's panel by the flush below)
        const int window = std::min(wcap, defer_max > 0 ? defer_max : (kp_call_limit > 0 ? kp_call_limit : wcap));
        if (!overlap && kp > 0 && (kp + k > window || kp + k8w > wcap))
        {
            if ((rc = flush())) // no room to keep them pending: apply them first
            {
                return rc;
            }
        }
        if ((rc = wait_pgemm())) // Ps must be quiescent for the gather
        {
            return rc;
        }
        last_k = k;
        dbgS = dbgGt = dbgV = nullptr; // (debug_last_update reads the handle's own workspace again)
        if ((rc = prof_begin(CSLAM_STAGE_GATHER)))
        {
            return rc;
        }
        const dim3 ggrid((n + 255) / 256, (m + kGatherObs - 1) / kGatherObs);
        // the compact H-rows block for the MFMA factor kernel (f32, 16 < k <= 64, no pending panels to correct)
        sub_valid = (k > 16 && k <= 64 && (kp == 0 || small_corr) && dSub != nullptr);
        PredictArgs<T> pnone{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, 0};
        {
            T*             sub  = sub_valid ? dSub : nullptr;
            PredictArgs<T> pa   = fuse_now ? pp : pnone;
            T*             pred = fuse_now ? dPred : (T*)nullptr;
            const T*       Wg   = (kp > 0 && !pipeline) ? (const T*)wbase(wcur) : (const T*)nullptr;
            const int      kg   = pipeline ? 0 : kp;
            const int*     sg =
                (kp > 0 && !pipeline && hd_cols[wcur] > 0) ? dSign + (size_t)wcur * wcap : (const int*)nullptr;
            T* yout = (kp > 0 && !small_corr && !pipeline) ? dY : (T*)nullptr;
            if (wide_corr)
            {
                const dim3 wgrid((n + 255) / 256, (m + kGatherObsWide - 1) / kGatherObsWide);
                hipLaunchKernelGGL((ekf_gather_kernel<T, kGatherCorrMax, kGatherObsWide>), wgrid, dim3(256), 0, stream, dX,
                                   dP, dPv, ldp, n, dZ, dIdf, m, dPHT, ldp, lower, sub, pa, pred, Wg, ldp, kg, sg, dFlags,
                                   yout);
            }
            else
            {
                hipLaunchKernelGGL(ekf_gather_kernel<T>, ggrid, dim3(256), 0, stream, dX, dP, dPv, ldp, n, dZ, dIdf, m, dPHT,
                                   ldp, lower, sub, pa, pred, Wg, ldp, kg, sg, dFlags, yout);
            }
        }
        CSLAM_HIP_TRY(hipGetLastError());
        // the panels this update's P*H^T must be corrected with, and where its own W1 goes
        const T*  Wc        = wbase(wcur);
        const int kc        = kp;
        const int rc_region = wcur;
        if (overlap)
        {
            if ((rc = flush())) // P-GEMM of the pending columns on stream B, behind the gather; store -> other region
            {
                return rc;
            }
        }
        T* slot = wbase(wcur) + (size_t)kp * ldp;
        if ((rc = own_region(wcur)))
        {
            return rc;
        }
        if (kc > 0 && !small_corr) // PHT -= Wp * (H*Wp)^T : the pending panels' share of P*H^T (their pose rows are zero)
        {
            if (pipeline) // (single stream: the gather kernel has published Y = H*Wp already)
            {
                hipLaunchKernelGGL(ekf_pending_y_kernel<T>, dim3(m, (kc + 255) / 256), dim3(256), 0, stream, dX, n, dZ, dIdf,
                                   m, Wc, ldp, kc, dY,
                                   hd_cols[rc_region] > 0 ? dSign + (size_t)rc_region * wcap : (const int*)nullptr);
                CSLAM_HIP_TRY(hipGetLastError());
            }
            if (!launch_corr_fast(k, Wc, kc))
            {
                hipLaunchKernelGGL(ekf_pending_corr_kernel<T>, ggrid, dim3(256), 0, stream, n, m, Wc, ldp, kc, dY, dPHT,
                                   ldp);
            }
            CSLAM_HIP_TRY(hipGetLastError());
        }
        if ((rc = prof_end(CSLAM_STAGE_GATHER)) || (rc = prof_begin(CSLAM_STAGE_FACTOR)) ||
            (rc = launch_factor(dZ, dIdf, m, R)) || (rc = prof_end(CSLAM_STAGE_FACTOR)) ||
            (rc = prof_begin(CSLAM_STAGE_GAIN)) || (rc = launch_gain(k, slot)) || (rc = prof_end(CSLAM_STAGE_GAIN)))
        {
            return rc;
        }
        if (fuse_now) // the gain kernel committed the predicted pose, stripe and Pvv
        {
            pp.valid = 0;
            fuse_now = false;
        }
        last_slot = slot;
        kp += k;
        const bool deferring = keep_pending || pipeline || defer_max > 0;
        if (!deferring && (rc = flush()))
        {
            return rc;
        }
        if (sync_mode)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(hFlags, dFlags, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            if (hFlags[1] & kFlagLltFailed)
            {
                return eigen_fallback(k, deferring);
            }
        }
        // explicit deferral without pipelining: once the window is full, apply it now rather than at the start of
        // the next update
        if (!pipeline && !keep_pending && defer_max > 0 && kp >= defer_max && (rc = flush()))
        {
            return rc;
        }
        return CSLAM_OK;
    }

    // slam.h:425-429 on the host: the device left X and P untouched (G = 0, t = 0)
    int eigen_fallback(int k, bool still_pending)
    {
        sticky_host |= CSLAM_FACTOR_FALLBACK;
        std::vector<T> S((size_t)k * k), V((size_t)k), G;
        CSLAM_HIP_TRY(hipMemcpyAsync(S.data(), dS, S.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipMemcpyAsync(V.data(), dV, V.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        bool textbook = !(quirks & CSLAM_Q_LOWER_CHOL_GAIN);
        if (!host_eigen_fallback_gain(S.data(), k, textbook, G))
        {
            sticky_host |= CSLAM_FACTOR_ZEROED;
            return CSLAM_OK; // zeros: the update is a no-op, which is what the device already did
        }
        std::vector<T> Gt((size_t)k * k), t((size_t)k, (T)0);
        for (int c = 0; c < k; c++)
        {
            for (int r = 0; r < k; r++)
            {
                Gt[(size_t)r * k + c] = G[(size_t)c * k + r];
                t[c] += G[(size_t)c * k + r] * V[r];
            }
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(dG, G.data(), G.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        CSLAM_HIP_TRY(hipMemcpyAsync(dGt, Gt.data(), Gt.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        g_from_gt  = false; // (the fallback gain is a general matrix, uploaded as G and G^T)
        m_valid    = false; // (M belonged to the zeroed G: the separate pose downdate kernel runs)
        CSLAM_HIP_TRY(hipMemcpyAsync(dt_, t.data(), t.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        std::vector<T> u((size_t)k, (T)0);
        for (int q = 0; q < k; q++)
        {
            for (int c = 0; c < k; c++)
            {
                u[q] += G[(size_t)c * k + q] * t[c];
            }
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(dU, u.data(), u.size() * sizeof(T), hipMemcpyHostToDevice, stream));
        int rc;
        if ((rc = own_region(wcur)))
        {
            return rc;
        }
        // the zero G made this update's W1 slot zero (a no-op wherever it was or will be applied): rewrite it
        // (the gain kernel's pose downdate runs again on the new panel: the first one subtracted zeros)
        if ((rc = launch_gain(k, last_slot)))
        {
            return rc;
        }
        if (!still_pending) // its P-GEMM already ran (with zeros): apply this panel alone
        {
            const int k8 = round_up(k, 8);
            if (k8 > k)
            {
                CSLAM_HIP_TRY(hipMemset2DAsync(last_slot + (size_t)k * ldp, (size_t)ldp * sizeof(T), 0,
                                               (size_t)round_up(n, kTile) * sizeof(T), (size_t)(k8 - k), stream));
            }
            if ((rc = launch_downdate(last_slot, k, stream)))
            {
                return rc;
            }
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }


    // ---------------------------------------------------------------- look-ahead windows (ekf_lookahead.hpp)
    // Asynchronous batch updates (16 < k <= 64) are collected two at a time.  For a window (a, b):
    //   stream   : rows / blocks kernels (small blocks of the current P for the factor chain)
    //   stream F : prefactor(a), factor(a), prefactor(b), factor(b) -- 1-workgroup kernels, ~60 us in all
    //   stream   : P-GEMM of the PREVIOUS window (k = 128), under which stream F runs;
    //              then gather(a), gain(a), gather(b) (corrected for W1_a in the kernel), gain(b) with the factors known.
    // The state the rest of the engine sees afterwards is the deferred engine's after two updates (k_a + k_b pending
    // columns in the store), so every other call simply drains the queue first (la_drain) and carries on.
    struct LaUpd
    {
        const T*       dZ;
        const int*     dIdf;
        int            m;
        T              R[4];
        PredictArgs<T> pp;
    };
    struct FactorOut
    {
        T *  S = nullptr, *G = nullptr, *Gt = nullptr, *V = nullptr, *t = nullptr, *U = nullptr, *M = nullptr;
        T *  sub = nullptr, *xloc = nullptr;
        int* idloc = nullptr;
    };
    LaUpd       la_q[2];
    int         la_n = 0;
    FactorOut   fo[2];
    hipStream_t stream_f = nullptr;
    hipEvent_t  ev_fb = nullptr;  // the chain kernel of the last window has finished
    hipEvent_t  ev_raw = nullptr; // the blocks kernel of the last window has finished (several engines alive only)
    T *         la_XL = nullptr, *la_PvL = nullptr, *la_WR = nullptr, *la_PH = nullptr, *la_PvLb = nullptr, *la_Dbb = nullptr;
    T*          la_Y = nullptr;         // H_b * W1_a of the last window (for a fused wide kernel)
    LaModel<T>* la_model = nullptr;     // [2]: predict + observation model of update a / b
    int         la_kpad  = 0;
    long long   la_windows = 0; // windows launched (diagnostics)
    int         la_cus = 0;        // compute units the persistent P-GEMM leaves to the chain kernel (0 until stream F exists)
    unsigned*   la_done   = nullptr; // device counter: workgroups of the blocks kernels that have finished
    unsigned    la_target = 0;       // its value once every blocks kernel launched so far has finished
    unsigned    la_seq    = 0;       // windows whose chain kernel has been launched
    // != 0: the next P-GEMM launch (ekf_downdate_psym4_f32) adds this to la_done[0] -- the chain's go-ahead, in place of a
    // release fence + atomic in every workgroup of the blocks kernel (8.8 -> 6.6 us per window); see la_launch_window
    unsigned    la_sig_add = 0;
    int         la_k64 = 1;       // CSLAM_LA_K64=0: the general wide kernel for m = 32 too (A/B)
    int         la_wg_signal = 0; // CSLAM_LA_WG_SIGNAL=1: always the blocks kernel's own release (A/B: the first form)

    // Dynamic LDS of the chain kernel <T, K>: the carry step's arrays (in f64 the factor body's arrays live in the same
    // space), padded so that the workgroup's total LDS is ~99 KB: more than 96 KB keeps the persistent P-GEMM's 64 KB
    // workgroups off its compute unit, less than 106 KB lets it start beside ONE workgroup of the wide kernel (54 KB) --
    // a chain kernel that found no unit before the P-GEMM filled the chip must still be able to start while the wide
    // kernel's workgroups wait for it.
    static size_t la_chain_lds(int K)
    {
        size_t need = la_carry_lds<T>();
        if (sizeof(T) == 8)
        {
            need = std::max(need, (size_t)(K * (K + 1) + (3 + K) * (K + 1) + (K / 2) * 10 + 6 * K) * sizeof(double) +
                                      (size_t)(K / 2 + 4) * sizeof(int) + 16);
            return need;
        }
        const size_t fixed = (K == 64) ? 53984 : 15008; // static LDS of ekf_la_chain_kernel<float, K> (tools/kernel_resources.py)
        return std::max(need, (size_t)101 * 1024 - fixed);
    }
    int         la_fused   = 1; // env CSLAM_LA_FUSED=0: gather + gain per update instead of the one wide launch (A/B)
    long long*  la_stamps  = nullptr; // CSLAM_LA_STAMPS=1: phase stamps of factor(a) underneath the P-GEMM (diagnostics)
    // what debug_last_update reads (the handle's workspace, or the factor slot of a window's last update)
    const T *dbgS = nullptr, *dbgGt = nullptr, *dbgV = nullptr;

    void la_free()
    {
        for (FactorOut& f : fo)
        {
            (void)hipFree(f.S);
            (void)hipFree(f.G);
            (void)hipFree(f.Gt);
            (void)hipFree(f.V);
            (void)hipFree(f.t);
            (void)hipFree(f.U);
            (void)hipFree(f.M);
            (void)hipFree(f.sub);
            (void)hipFree(f.xloc);
            (void)hipFree(f.idloc);
            f = FactorOut();
        }
        (void)hipFree(la_XL);
        (void)hipFree(la_PvL);
        (void)hipFree(la_WR);
        (void)hipFree(la_PH);
        (void)hipFree(la_PvLb);
        (void)hipFree(la_Dbb);
        (void)hipFree(la_Y);
        (void)hipFree(la_model);
        (void)hipFree(la_stamps);
        (void)hipFree(la_done);
        la_stamps = nullptr;
        la_done   = nullptr;
        la_XL = la_PvL = la_WR = la_PH = la_PvLb = la_Dbb = la_Y = nullptr;
        la_model = nullptr;
        la_kpad  = 0;
    }

    int la_ensure(int kp_cols)
    {
        constexpr int KM = 2 * kLaMaxObs;
        if (stream_f == nullptr)
        {
            // The chain kernel owns its compute unit by the LDS it asks for (see ekf_la_chain_kernel); the persistent
            // P-GEMM's grid is reduced by that unit's two workgroups (la_cus).
            int lo = 0, hi = 0;
            CSLAM_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            CSLAM_HIP_TRY(hipStreamCreateWithPriority(&stream_f, hipStreamNonBlocking, hi));
            la_cus = 1;
            CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_fb, hipEventDisableTiming));
            CSLAM_HIP_TRY(hipEventCreateWithFlags(&ev_raw, hipEventDisableTiming));
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_la_chain_kernel<T, 32>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)la_chain_lds(32)));
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_la_chain_kernel<T, 64>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)la_chain_lds(64)));
            // words 0..255: 16 counters (stride 16) of finished blocks-kernel workgroups; words 256..767: 32 copies
            // (stride 16) of the last window whose chain kernel has finished
            CSLAM_HIP_TRY(hipMalloc(&la_done, 768 * sizeof(unsigned)));
            CSLAM_HIP_TRY(hipMemset(la_done, 0, 768 * sizeof(unsigned)));
            la_target = 0;
            la_seq    = 0;
            for (FactorOut& f : fo)
            {
                CSLAM_HIP_TRY(hipMalloc(&f.S, (size_t)KM * (KM + 1) * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.G, (size_t)KM * (KM + 1) * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.Gt, (size_t)KM * (KM + 1) * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.V, (size_t)KM * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.t, (size_t)KM * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.U, (size_t)KM * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.M, (size_t)3 * KM * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.sub, (size_t)(3 + KM) * KM * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.xloc, (size_t)(3 + KM) * sizeof(T)));
                CSLAM_HIP_TRY(hipMalloc(&f.idloc, (size_t)kLaMaxObs * sizeof(int)));
            }
            CSLAM_HIP_TRY(hipMalloc(&la_XL, (size_t)2 * KM * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_PvL, (size_t)2 * KM * 3 * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_PH, (size_t)KM * KM * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_PvLb, (size_t)KM * 3 * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_Dbb, (size_t)KM * KM * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_Y, (size_t)KM * KM * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&la_model, 2 * sizeof(LaModel<T>)));
            if (const char* fv = getenv("CSLAM_LA_FUSED"))
            {
                la_fused = atoi(fv) ? 1 : 0;
            }
            if (getenv("CSLAM_LA_STAMPS"))
            {
                CSLAM_HIP_TRY(hipMalloc(&la_stamps, 32 * sizeof(long long)));
                CSLAM_HIP_TRY(hipMemset(la_stamps, 0, 32 * sizeof(long long)));
            }
        }
        if (kp_cols > la_kpad)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream_f));
            (void)hipFree(la_WR);
            la_WR   = nullptr;
            la_kpad = 0;
            const int kpad = round_up(std::max(kp_cols, 128), 64);
            CSLAM_HIP_TRY(hipMalloc(&la_WR, (size_t)2 * KM * kpad * sizeof(T))); // [kpad columns][128 slots]
            la_kpad = kpad;
        }
        return CSLAM_OK;
    }

    // may this batch join a look-ahead window?  (everything the window's fixed schedule does not cover stays classic)
    bool la_eligible(int m) const
    {
        const int k = 2 * m;
        // lookahead: 1 on, 0 off, -1 (default) where it pays: f32 and a P-GEMM long enough to hide the factor chain
        // (~70 us) underneath it -- about N >= 3500 landmarks; a short P-GEMM leaves the chain on the critical path
        // (measured: f64 N = 1000 12.9 k steps/s with windows against 15.8 k without)
        const bool on = lookahead > 0 || (lookahead < 0 && sizeof(T) == 4 && n >= 7000 && g_engines.load() == 1);
        return on && !sync_mode && !pipeline && profiling != 1 && k > 16 && k <= 2 * kLaMaxObs && gather_corr_wide &&
               fuse_predict && (sizeof(T) == 4 || fuse_f64) && defer_max >= k + (la_n ? 2 * la_q[0].m : k) &&
               wcap >= k + (la_n ? 2 * la_q[0].m : k) && seq.count == 0 && hd_cols[0] == 0 && hd_cols[1] == 0 && n > 3 &&
               kp_call_limit == 0;
    }

    int la_enqueue(const T* dZ, const int* dIdf, int m, const T* R)
    {
        LaUpd u;
        u.dZ   = dZ;
        u.dIdf = dIdf;
        u.m    = m;
        for (int i = 0; i < 4; i++)
        {
            u.R[i] = R[i];
        }
        u.pp     = pp; // the held predict belongs to this update
        pp.valid = 0;
        la_q[la_n++] = u;
        return la_n == 2 ? la_launch_window() : CSLAM_OK;
    }

    // the factor kernel's arguments for one update of a window: compact inputs (a local state vector of 3 + 2m entries with
    // local feature ids 1..m and the block sub), outputs into the factor slot f
    FactorArgs<T> la_factor_args(const LaUpd& u, FactorOut& f)
    {
        FactorArgs<T> a;
        a.X   = f.xloc;
        a.n   = 3 + 2 * u.m;
        a.Z   = u.dZ;
        a.idf = fo[0].idloc; // 1, 2, ... (written once per window by the blocks kernel)
        a.m   = u.m;
        for (int i = 0; i < 4; i++)
        {
            a.R[i] = u.R[i];
        }
        a.PHT      = dPHT; // (not read: the compact block is supplied)
        a.ldw      = ldp;
        a.dS       = f.S;
        a.dG       = f.G;
        a.dGt      = f.Gt;
        a.dV       = f.V;
        a.dt       = f.t;
        a.flags    = dFlags;
        a.scratchS = dScrS;
        a.scratchG = dScrG;
        a.textbook = (quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 0 : 1;
        a.stamps   = (la_stamps && &f == &fo[0]) ? la_stamps : nullptr;
        a.sub      = f.sub;
        a.dM       = f.M;
        a.pp       = PredictArgs<T>{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, 0}; // (applied by blocks / carry)
        a.P3       = dPv;
        a.ldp3     = ldp;
        a.pred_out = nullptr;
        a.lds_S    = 1;
        a.lds_G    = 1;
        return a;
    }

    // gather + gain of one update of the window on the main stream, with the factor outputs of slot f
    int la_wide(const LaUpd& u, const FactorOut& f, hipEvent_t ev_factor)
    {
        const int k = 2 * u.m;
        pp          = u.pp;
        fuse_now    = pp.valid != 0;
        if (fuse_now && dPred == nullptr)
        {
            CSLAM_HIP_TRY(hipMalloc(&dPred, 16 * sizeof(T)));
        }
        last_k = k;
        PredictArgs<T> pnone{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, 0};
        PredictArgs<T> pa   = fuse_now ? pp : pnone;
        T*             pred = fuse_now ? dPred : (T*)nullptr;
        const T*       Wg   = kp > 0 ? (const T*)wbase(wcur) : (const T*)nullptr;
        if (kp > kGatherCorr)
        {
            const dim3 wgrid((n + 255) / 256, (u.m + kGatherObsWide - 1) / kGatherObsWide);
            hipLaunchKernelGGL((ekf_gather_kernel<T, kGatherCorrMax, kGatherObsWide>), wgrid, dim3(256), 0, stream, dX, dP, dPv,
                               ldp, n, u.dZ, u.dIdf, u.m, dPHT, ldp, lower, (T*)nullptr, pa, pred, Wg, ldp, kp,
                               (const int*)nullptr, dFlags, (T*)nullptr);
        }
        else
        {
            const dim3 ggrid((n + 255) / 256, (u.m + kGatherObs - 1) / kGatherObs);
            hipLaunchKernelGGL(ekf_gather_kernel<T>, ggrid, dim3(256), 0, stream, dX, dP, dPv, ldp, n, u.dZ, u.dIdf, u.m, dPHT,
                               ldp, lower, (T*)nullptr, pa, pred, Wg, ldp, kp, (const int*)nullptr, dFlags, (T*)nullptr);
        }
        CSLAM_HIP_TRY(hipGetLastError());
        if (ev_factor != nullptr)
        {
            CSLAM_HIP_TRY(hipStreamWaitEvent(stream, ev_factor, 0));
        }
        T*  slot = wbase(wcur) + (size_t)kp * ldp;
        int rc   = launch_gain(k, slot, f.Gt, f.U, f.M);
        if (rc)
        {
            return rc;
        }
        pp.valid  = 0;
        fuse_now  = false;
        last_slot = slot;
        kp += k;
        dbgS      = f.S;
        dbgGt     = f.Gt;
        dbgV      = f.V;
        g_from_gt = true;
        sub_valid = false;
        return CSLAM_OK;
    }

    int la_launch_window()
    {
        const int nu = la_n;
        if (nu == 0)
        {
            return CSLAM_OK;
        }
        la_n = 0;
        const LaUpd ua = la_q[0], ub = la_q[1];
        const PredictArgs<T> held = pp; // a predict accepted AFTER the queued updates stays held
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const int ka = 2 * ua.m, kb = nu == 2 ? 2 * ub.m : 0;
        if (kp > 256 && (rc = flush())) // (the blocks kernel stages one row of at most 256 pending columns)
        {
            return rc;
        }
        if ((rc = ensure_k(2 * kLaMaxObs)) || (rc = ensure_w(ka + kb)) || (rc = la_ensure(kp)))
        {
            return rc;
        }
        // (the P-GEMM's tile list is (re)built here, not inside the launch below: building it waits for both streams, and
        // from the chain launch on stream F waits for a go-ahead that only that P-GEMM launch delivers)
        if (lower && (rc = ensure_tile_list(round_up(n, kTile) / kTile)))
        {
            return rc;
        }
        // 1. the factor chain of the window on stream F, ONE launch, submitted first: it takes a compute unit for itself
        //    and waits there (on a counter) for the blocks kernel below.  (safe: several engines alive -- the chain kernel is
        //    launched behind the blocks kernel's event instead, see g_engines)
        const bool     safe     = g_engines.load() > 1;
        const unsigned n_blocks = (unsigned)(3 + ka + 2 * kb);
        auto launch_chain = [&]() -> int {
            LaChainArgs<T> ch;
            ch.fa   = la_factor_args(ua, fo[0]);
            ch.fb   = la_factor_args(nu == 2 ? ub : ua, fo[1]);
            ch.du_a = fo[0].U;
            ch.du_b = fo[1].U;
            ch.nu   = nu;
            ch.done = la_done;
            ch.target  = la_target + n_blocks;
            ch.timeout = 20000000ull; // 0.2 s of s_memrealtime ticks
            ch.chain_done = la_done + 256;
            ch.seq        = ++la_seq;
            LaCarryArgs<T>& ca = ch.ca;
            ca.n       = n;
            ca.m_a     = ua.m;
            ca.m_b     = nu == 2 ? ub.m : 0;
            ca.idf_b   = nu == 2 ? ub.dIdf : ua.dIdf;
            ca.pp_b    = nu == 2 ? ub.pp : ua.pp;
            ca.PH      = la_PH;
            ca.Dbb     = la_Dbb;
            ca.PvLb    = la_PvLb;
            ca.XLb     = la_XL + ka;
            ca.model_a = la_model;
            ca.Gt_a    = fo[0].Gt;
            ca.u_a     = fo[0].U;
            ca.M_a     = fo[0].M;
            ca.sub_a   = fo[0].sub;
            ca.sub_b   = fo[1].sub;
            ca.xloc_b  = fo[1].xloc;
            ca.model_b = la_model + 1;
            ca.Y_b     = la_Y;
            if (std::max(ka, kb) <= 32)
            {
                hipLaunchKernelGGL((ekf_la_chain_kernel<T, 32>), dim3(1), dim3(256), la_chain_lds(32), stream_f, ch);
            }
            else
            {
                hipLaunchKernelGGL((ekf_la_chain_kernel<T, 64>), dim3(1), dim3(256), la_chain_lds(64), stream_f, ch);
            }
            CSLAM_HIP_TRY(hipGetLastError());
            CSLAM_HIP_TRY(hipEventRecord(ev_fb, stream_f));
            return CSLAM_OK;
        };
        if (!safe && (rc = launch_chain()))
        {
            return rc;
        }
        // 2. what the chain needs of the current covariance P = Ps - Wp Wp^T (before Ps changes): rows of the pending
        //    panels, then one workgroup per row of the small blocks; update a's compact block sub_a comes out of it ready
        //    for the factor step.  (From here to the blocks launch nothing may fail: the chain kernel is waiting.)
        const T* Wp = wbase(wcur);
        LaRowsArgs<T> ra;
        ra.X     = dX;
        ra.Pv    = dPv;
        ra.ldp   = ldp;
        ra.n     = n;
        ra.idf_a = ua.dIdf;
        ra.ra    = ka;
        ra.idf_b = nu == 2 ? ub.dIdf : ua.dIdf;
        ra.rb    = kb;
        ra.Wp    = Wp;
        ra.ldw   = ldp;
        ra.kp    = kp;
        ra.kpad  = la_kpad;
        ra.XL    = la_XL;
        ra.PvL   = la_PvL;
        ra.WR    = la_WR;
        ra.flags = dFlags;
        hipLaunchKernelGGL(ekf_la_rows_kernel<T>, dim3(ka + kb), dim3(128), 0, stream, ra);
        LaPrepArgs<T> pa;
        pa.P       = dP;
        pa.ldp     = ldp;
        pa.n       = n;
        pa.lower   = lower;
        pa.X       = dX;
        pa.Pv      = dPv;
        pa.idf_a   = ua.dIdf;
        pa.idf_b   = nu == 2 ? ub.dIdf : ua.dIdf;
        pa.ra      = ka;
        pa.rb      = kb;
        pa.pp_a    = ua.pp;
        pa.XL      = la_XL;
        pa.PvL     = la_PvL;
        pa.WR      = la_WR;
        pa.kp      = kp;
        pa.kpad    = la_kpad;
        pa.sub_a   = fo[0].sub;
        pa.PH      = la_PH;
        pa.Dbb     = la_Dbb;
        pa.PvLb    = la_PvLb;
        pa.model_a = la_model;
        pa.xloc_a  = fo[0].xloc;
        pa.idloc   = fo[0].idloc;
        // the P-GEMM that follows signals the chain when it is the plain single-stream psym4 launch (always in the steady state);
        // otherwise the blocks kernel's workgroups release their rows themselves
        const int  k8f      = round_up(kp, 8);
        const bool pg_signal = !la_wg_signal && !safe && kp > 0 && seq.count == 0 && sizeof(T) == 4 && k8f <= 128 && lower && ldp < 32768 &&
                               !limbs_take(k8f) && stream_b == stream && hd_cols[wcur] == 0;
        pa.done    = pg_signal ? (unsigned*)nullptr : la_done;
        hipLaunchKernelGGL(ekf_la_blocks_kernel<T>, dim3(n_blocks), dim3(64), 0, stream, pa);
        CSLAM_HIP_TRY(hipGetLastError());
        if (safe)
        {
            CSLAM_HIP_TRY(hipEventRecord(ev_raw, stream));
            CSLAM_HIP_TRY(hipStreamWaitEvent(stream_f, ev_raw, 0));
            if ((rc = launch_chain())) // (its wait for the blocks kernel's counters passes at once)
            {
                return rc;
            }
        }
        la_target += n_blocks;
        // 3. the P-GEMM of everything pending (the previous window's panels): stream F works underneath it
        la_sig_add = pg_signal ? n_blocks : 0u;
        rc         = flush();
        if (la_sig_add != 0) // (the launch did not happen: release the chain from here -- it times out otherwise)
        {
            const unsigned add = la_sig_add;
            la_sig_add         = 0;
            hipLaunchKernelGGL(ekf_la_signal_kernel, dim3(1), dim3(64), 0, stream, la_done, add);
        }
        if (rc)
        {
            return rc;
        }
        // 4. the wide half of both updates, factors known: ONE launch in f32 (ekf_la_wide_f32), gather + gain per update
        //    otherwise (CSLAM_LA_FUSED=0: A/B)
        bool fused = false;
        if constexpr (std::is_same<T, float>::value)
        {
            if (la_fused)
            {
                fused = true;
                if (safe) // (see g_engines: no waiting inside the wide kernel then)
                {
                    CSLAM_HIP_TRY(hipStreamWaitEvent(stream, ev_fb, 0));
                }
                LaWideArgs wa;
                wa.chain_done = la_done + 256; // (waits for the chain kernel in the kernel: a stream event costs ~6 us here)
                wa.seq        = la_seq;
                wa.timeout    = 20000000ull;
                wa.flags      = dFlags;
                wa.stamps     = la_stamps ? la_stamps + 16 : nullptr;
                wa.wg_times   = nullptr;
                wa.P       = dP;
                wa.ldp     = ldp;
                wa.n       = n;
                wa.lower   = getenv("CSLAM_LA_TIMING_DIRECT") ? 0 : lower; // (timing experiment only: wrong results)
                wa.X       = dX;
                wa.Pv      = dPv;
                wa.nu      = nu;
                wa.idf_a   = ua.dIdf;
                wa.idf_b   = nu == 2 ? ub.dIdf : ua.dIdf;
                wa.ma      = ua.m;
                wa.mb      = nu == 2 ? ub.m : 0;
                wa.valid_a = ua.pp.valid;
                wa.valid_b = nu == 2 ? ub.pp.valid : 0;
                wa.w_a     = ua.pp.w;
                wa.w_b     = nu == 2 ? ub.pp.w : 0;
                wa.model_a = la_model;
                wa.model_b = la_model + 1;
                wa.Gt_a    = fo[0].Gt;
                wa.u_a     = fo[0].U;
                wa.M_a     = fo[0].M;
                wa.sub_a   = fo[0].sub;
                wa.Gt_b    = fo[1].Gt;
                wa.u_b     = fo[1].U;
                wa.M_b     = fo[1].M;
                wa.sub_b   = fo[1].sub;
                wa.Y_b     = la_Y;
                wa.W1a     = wbase(wcur) + (size_t)kp * ldp;
                wa.W1b     = wa.W1a + (size_t)ka * ldp;
                wa.ldw     = ldp;
                wa.wv_out  = dWv;
                if (la_k64 && ua.m == 32 && (nu == 1 || ub.m == 32))
                {
                    hipLaunchKernelGGL(ekf_la_wide_f32_k64, dim3(round_up(n, kTile) / 32), dim3(128), 0, stream, wa);
                }
                else
                {
                    hipLaunchKernelGGL(ekf_la_wide_f32, dim3(round_up(n, kTile) / 32), dim3(128), 0, stream, wa);
                }
                CSLAM_HIP_TRY(hipGetLastError());
                last_slot = nullptr; // (PHT is not materialised on this path: nothing for debug_last_update)
                last_k    = 0;
                kp += ka + kb;
                sub_valid = false;
            }
        }
        if (!fused && ((rc = la_wide(ua, fo[0], ev_fb)) || (nu == 2 && (rc = la_wide(ub, fo[1], nullptr)))))
        {
            return rc;
        }
        pp = held;
        la_windows++;
        if (la_stamps && la_windows == 300)
        {
            long long h[32];
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            CSLAM_HIP_TRY(hipMemcpy(h, la_stamps, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[cslam la wide stamps, 10 ns ticks] ids+columns issue:%lld poll+DMA wait:%lld pht_a:%lld gain_a:%lld store+share W1_a:%lld pht_b+corr:%lld share+G_b:%lld gain_b:%lld store_b:%lld\n",
                    h[17] - h[16], h[18] - h[17], h[19] - h[18], h[20] - h[19], h[21] - h[20], h[22] - h[21], h[23] - h[22],
                    h[24] - h[23], h[25] - h[24]);
            fprintf(stderr, "[cslam la stamps, cycles] load+observe:%lld sums:%lld symmetrise:%lld cholesky:%lld (first half %lld) inverse:%lld outputs:%lld total:%lld\n",
                    h[6] - h[0], h[7] - h[6], h[1] - h[7], h[2] - h[1], h[5] ? h[5] - h[1] : 0, h[3] - h[2], h[4] - h[3], h[4] - h[0]);
        }
        return CSLAM_OK;
    }

    // every call that needs the state as of the last update() goes through here first
    int la_drain() override { return la_n ? la_launch_window() : CSLAM_OK; }

    int update(const void* Zv, int m, const void* Rv, const int* idf, int batch, bool on_device) override
    {
        if (m < 0 || !Rv || (m > 0 && (!Zv || !idf)))
        {
            return fail(CSLAM_ERR_BAD_ARG, "update: bad arguments (m=%d)", m);
        }
        if (m == 0)
        {
            return CSLAM_OK; // EKF.cpp:101-123 with an empty Z: nothing changes
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const T*   R    = static_cast<const T*>(Rv);
        const T*   dZ   = static_cast<const T*>(Zv);
        const int* dIdf = idf;
        if (!on_device)
        {
            const int nf = (n - 3) / 2;
            for (int i = 0; i < m; i++)
            {
                if (idf[i] < 1 || idf[i] > nf)
                {
                    return fail(CSLAM_ERR_BAD_ARG, "update: idf[%d]=%d outside 1..%d", i, idf[i], nf);
                }
            }
            if ((rc = stage_obs(Zv, idf, m, &dZ, &dIdf)))
            {
                return rc;
            }
        }
        if (batch && la_eligible(m))
        {
            return la_enqueue(dZ, dIdf, m, R);
        }
        if ((rc = la_drain())) // (the held predict of THIS call survives the drain: la_launch_window keeps it)
        {
            return rc;
        }
        if (batch)
        {
            return batch_on_device(dZ, dIdf, m, R, false);
        }
        if ((rc = resolve_predict()))
        {
            return rc;
        }
        // EKF.cpp:457-479: m successive rank-2 updates, relinearised on the updated state each time.  Their m
        // rank-2 downdates stay pending and are applied by ONE P-GEMM with k = 2m: each observation reads the columns
        // it needs as Ps[:,c] - Wp*Wp[c,:]^T.
        if (seq_defer)
        {
            // (every rank-2 slot is written as a block of 8 columns: the last one reaches column kp + 2m + 6)
            if ((rc = ensure_w(2 * m + 8)))
            {
                return rc;
            }
            if (kp + 2 * m + 6 > wcap && (rc = flush()))
            {
                return rc;
            }
        }
        kp_call_limit = seq_defer ? kp + 2 * m : 0;
        for (int i = 0; i < m; i++)
        {
            if ((rc = batch_on_device(dZ + 2 * i, dIdf + i, 1, R, seq_defer != 0)))
            {
                kp_call_limit = 0;
                return rc;
            }
        }
        kp_call_limit = 0;
        if (!pipeline && defer_max == 0 && (rc = flush()))
        {
            return rc;
        }
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- augment (EKF.cpp:9-91)
    int augment(const void* Zv, int q, const void* Rv) override
    {
        if (q < 0 || !Rv || (q > 0 && !Zv))
        {
            return fail(CSLAM_ERR_BAD_ARG, "augment: bad arguments (q=%d)", q);
        }
        if (n + 2 * q > ncap)
        {
            return fail(CSLAM_ERR_CAPACITY, "augment: %d features would exceed max_landmarks=%d", (n - 3) / 2 + q, nmax);
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if (q > 0 && (rc = wait_pgemm())) // the new rows / columns of Ps are written here: no P-GEMM may be sweeping it
        {
            return rc;
        }
        const T* Z = static_cast<const T*>(Zv);
        const T* R = static_cast<const T*>(Rv);
        for (int i = 0; i < q; i++)
        {
            // (pending panels: their rows for the new feature are zero, which is right -- the kernel writes values of
            // the true P, built from the pose stripe)
            hipLaunchKernelGGL(ekf_augment_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, stream, dX, dP, dPv, ldp, n, Z[2 * i],
                               Z[2 * i + 1], R[0], R[1], R[2], R[3], lower);
            CSLAM_HIP_TRY(hipGetLastError());
            n += 2;
        }
        return CSLAM_OK;
    }

    // ---------------------------------------------------------------- heading (EKF.cpp:328-352)
    int observe_heading(double phi, int use) override
    {
        if (!use)
        {
            return CSLAM_OK; // EKF.cpp:332-335 (a pending predict stays pending)
        }
        // float sigmaPhi = 0.01F * pi / 180.0F; R = pow(sigmaPhi, 2)
        T   sigma = (T)(((double)0.01f * kPi) / 180.0);
        int rc    = queue_step(pp, HeadingArgs<T>{1, (T)phi, sigma * sigma}); // with the held predict, if any
        if (rc || fuse_predict)
        {
            return rc;
        }
        return launch_pose_queue();
    }

    int factor_status(int* flags, int clear) override
    {
        if (!flags)
        {
            return fail(CSLAM_ERR_BAD_ARG, "factor_status: null");
        }
        int rc = use_device();
        if (rc || (rc = la_drain()))
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(hFlags, dFlags, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        int f = sticky_host;
        if (hFlags[0] & kFlagZeroed)
        {
            f |= CSLAM_FACTOR_ZEROED;
        }
        if (hFlags[0] & kFlagBadIdf)
        {
            f |= CSLAM_FACTOR_BAD_IDF;
        }
        if (hFlags[0] & kFlagLaTimeout)
        {
            f |= CSLAM_FACTOR_INTERNAL;
        }
        if ((hFlags[0] & kFlagLltFailed) && !(sticky_host & CSLAM_FACTOR_FALLBACK))
        {
            f |= CSLAM_FACTOR_SKIPPED; // async mode: the failed factorisation was not followed up
        }
        *flags = f;
        if (clear)
        {
            sticky_host = 0;
            CSLAM_HIP_TRY(hipMemsetAsync(dFlags, 0, 2 * sizeof(int), stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        }
        return CSLAM_OK;
    }

    int set_deferred(int max_cols) override
    {
        if (max_cols < 0)
        {
            return fail(CSLAM_ERR_BAD_ARG, "set_deferred: negative");
        }
        int rc = flush();
        if (rc)
        {
            return rc;
        }
        defer_max = max_cols;
        return max_cols > 0 ? ensure_w(max_cols + 64) : CSLAM_OK;
    }

    int do_flush() override { return flush(); }

    int debug_last_update(void* PHT, void* S, void* G, void* W1, void* V, int* kout) override
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const int k = last_slot ? last_k : 0;
        if (kout)
        {
            *kout = k;
        }
        if (k == 0)
        {
            return CSLAM_OK;
        }
        if (PHT)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(PHT, (size_t)n * sizeof(T), dPHT, (size_t)ldp * sizeof(T),
                                           (size_t)n * sizeof(T), (size_t)k, hipMemcpyDeviceToHost, stream));
        }
        if (W1)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(W1, (size_t)n * sizeof(T), last_slot, (size_t)ldp * sizeof(T),
                                           (size_t)n * sizeof(T), (size_t)k, hipMemcpyDeviceToHost, stream));
            // its pose rows were zeroed in the store after the pose stripe took its share (ekf_pose_downdate_kernel
            // saved them): rows 0..2 <- dWv (3 x k, row c at c*k)
            CSLAM_HIP_TRY(hipMemcpy2DAsync(W1, (size_t)n * sizeof(T), dWv, sizeof(T), sizeof(T), (size_t)k,
                                           hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipMemcpy2DAsync(static_cast<T*>(W1) + 1, (size_t)n * sizeof(T), dWv + k, sizeof(T), sizeof(T),
                                           (size_t)k, hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipMemcpy2DAsync(static_cast<T*>(W1) + 2, (size_t)n * sizeof(T), dWv + 2 * k, sizeof(T), sizeof(T),
                                           (size_t)k, hipMemcpyDeviceToHost, stream));
        }
        if (S)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(S, dbgS ? dbgS : dS, (size_t)k * k * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (G && !g_from_gt)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(G, dG, (size_t)k * k * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (G && g_from_gt) // the tuned factor kernels publish only G^T (what the gain kernel reads)
        {
            std::vector<T> Gt((size_t)k * k);
            CSLAM_HIP_TRY(hipMemcpyAsync(Gt.data(), dbgGt ? dbgGt : dGt, Gt.size() * sizeof(T), hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            T* out = static_cast<T*>(G);
            for (int c = 0; c < k; c++)
            {
                for (int r = 0; r < k; r++)
                {
                    out[(size_t)c * k + r] = Gt[(size_t)r * k + c];
                }
            }
        }
        if (V)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(V, dbgV ? dbgV : dV, (size_t)k * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }
};

// lower-triangular tile list (ti >= tj), column-of-tiles major so that consecutive entries share the column
// panel; rebuilt when the number of 128-row tiles changes (n grew past a tile boundary)
template <typename T>
int Ekf<T>::ensure_tile_list(int tiles)
{
    if (tiles == tiles_built)
    {
        return CSLAM_OK;
    }
    std::vector<int2> h;
    h.reserve((size_t)tiles * (tiles + 1) / 2);
    for (int tj = 0; tj < tiles; tj++)
    {
        for (int ti = tj; ti < tiles; ti++)
        {
            h.push_back(make_int2(ti, tj));
        }
    }
    if (int rc = sync_all()) // (a P-GEMM in flight still reads the old list)
    {
        return rc;
    }
    (void)hipFree(dTiles);
    dTiles = nullptr;
    CSLAM_HIP_TRY(hipMalloc(&dTiles, h.size() * sizeof(int2)));
    CSLAM_HIP_TRY(hipMemcpy(dTiles, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
    tiles_built = tiles;
    n_sym_tiles = (int)h.size();
    return CSLAM_OK;
}

template <>
int Ekf<float>::launch_downdate(const float* W, int k, hipStream_t stream)
{
    const int tiles = round_up(n, kTile) / kTile;
    const int k8    = round_up(k, 8); // W columns [k, k8) are zero where the kernel reads them (flush() clears the tail)
    // persistent symmetric kernels over the list of lower-triangular tiles; mirror stores only under full storage
    int rc = ensure_tile_list(tiles);
    if (rc)
    {
        return rc;
    }
    const dim3 block(256);
    // Persistent grid: two workgroups per CU, minus `pgemm_spare` in two-stream mode -- a few CUs keep one workgroup
    // (64 of 160 KB LDS) so that the one-workgroup factor kernel of the NEXT update (53 KB LDS, stream A) finds room
    // while this P-GEMM fills the chip.
    // (two-stream mode: a few CUs keep one workgroup; look-ahead windows: the main stream's queue mask excludes the
    // compute units of the factor chain's stream, see la_ensure)
    int G = std::min(n_sym_tiles, std::max(1, 2 * (num_cus - la_cus) - (pipeline ? pgemm_spare : 0)));
    if (pgemm_wgs > 0)
    {
        G = std::min(G, pgemm_wgs);
    }
    const bool nt = psym_nt >= 0 ? psym_nt != 0 : (size_t)n_sym_tiles * 65536 > ((size_t)230 << 20);
    if (limbs_take(k8))
    {
        // f32 products as exact bf16 limb products on the bf16 matrix cores (ekf_pgemm_limbs.hpp)
        const int    nch  = k8 <= 64 ? 4 : (k8 <= 96 ? 6 : (k8 <= 128 ? 8 : (k8 <= 192 ? 12 : 16)));
        const int    kgs  = 2 * nch;
        const int    rows = round_up(n, kTile);
        const size_t need = (size_t)2 * 3 * kgs * rows * 16;
        if (need > wb_bytes)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream)); // (grows with n and the window: rare)
            (void)hipFree(dWb);
            dWb      = nullptr;
            wb_bytes = 0;
            const size_t cap = (size_t)2 * 3 * 32 * round_up(ncap, kTile) * 16; // every window up to 256 columns
            CSLAM_HIP_TRY(hipMalloc(&dWb, cap));
            wb_bytes = cap;
        }
        if ((rc = ensure_tiles_morton(tiles, stream)))
        {
            return rc;
        }
        hipLaunchKernelGGL(ekf_limb_split_kernel, dim3((rows + 255) / 256, kgs), dim3(256), 0, stream, W, ldp, k, rows, kgs,
                           dWb);
        // (ring of 3 panel buffers, 72 KB: two workgroups per compute unit)
        limb_parity ^= 1;
#define CSLAM_LAUNCH_PSYM5(MODE, NCH, NP, RR)                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        static bool attr_set = false;                                                                                  \
        if (!attr_set)                                                                                                 \
        {                                                                                                              \
            CSLAM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ekf_downdate_psym5_bf16<MODE, NCH, NP, RR>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));               \
            attr_set = true;                                                                                           \
        }                                                                                                              \
        hipLaunchKernelGGL((ekf_downdate_psym5_bf16<MODE, NCH, NP, RR>), dim3(G), block, (size_t)RR * 24576, stream, dP, ldp, \
                           (const uint4*)dWb, rows, (const int2*)dTilesM, (const int*)dSegOff, dTicketX + 8 * limb_parity,  \
                           dTicketX + 8 * (limb_parity ^ 1));                                                          \
    } while (0)
#define CSLAM_LAUNCH_PSYM5R(NCH, RR)                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        if (pgemm_limbs == 6)                                                                                          \
        {                                                                                                              \
            if (nt) { CSLAM_LAUNCH_PSYM5(1, NCH, 6, RR); } else { CSLAM_LAUNCH_PSYM5(0, NCH, 6, RR); }                 \
        }                                                                                                              \
        else                                                                                                           \
        {                                                                                                              \
            if (nt) { CSLAM_LAUNCH_PSYM5(1, NCH, 9, RR); } else { CSLAM_LAUNCH_PSYM5(0, NCH, 9, RR); }                 \
        }                                                                                                              \
    } while (0)
        switch (nch)
        {
        case 4: CSLAM_LAUNCH_PSYM5R(4, 3); break;
        case 6: CSLAM_LAUNCH_PSYM5R(6, 3); break;
        case 8: CSLAM_LAUNCH_PSYM5R(8, 3); break;
        case 12: CSLAM_LAUNCH_PSYM5R(12, 3); break;
        default: CSLAM_LAUNCH_PSYM5R(16, 3); break;
        }
#undef CSLAM_LAUNCH_PSYM5R
#undef CSLAM_LAUNCH_PSYM5
    }
    else if (k8 <= 128 && lower && ldp < 32768)
    {
        // the shipped P-GEMM: every memory operation interleaved with the MFMA loop; two chunks of 32 columns (k <= 64),
        // four of 24 (k <= 96) or four of 32 (k <= 128)
        launch_parity++;
        // per-XCD tile queues (see the kernel): env CSLAM_XCD_QUEUES (every queue needs workgroups: small grids stay on
        // the single queue)
        const bool xq = xcd_queues != 0 && G >= 64;
        if (xq)
        {
            if ((rc = ensure_tiles_morton(tiles, stream)))
            {
                return rc;
            }
            limb_parity ^= 1;
        }
#define CSLAM_LAUNCH_PSYM4(MODE, NCH, KC)                                                                             \
    hipLaunchKernelGGL((ekf_downdate_psym4_f32<MODE, NCH, KC>), dim3(G), block, 0, stream, dP, ldp, W, ldp, k,         \
                       xq ? (const int2*)dTilesM : (const int2*)dTiles, n_sym_tiles,                                  \
                       xq ? dTicketX + 8 * limb_parity : dTicket + (launch_parity & 1),                              \
                       xq ? dTicketX + 8 * (limb_parity ^ 1) : dTicket + ((launch_parity + 1) & 1),                  \
                       (unsigned long long*)nullptr, xq ? (const int*)dSegOff : (const int*)nullptr, 0u, 0u, 0u, 0u,           \
                       la_sig_add ? la_done : (unsigned*)nullptr, la_sig_add, 1, 0)
        if (k8 <= 64)
        {
            if (nt) { CSLAM_LAUNCH_PSYM4(1, 2, 32); } else { CSLAM_LAUNCH_PSYM4(0, 2, 32); }
        }
        else if (k8 <= 96)
        {
            if (nt) { CSLAM_LAUNCH_PSYM4(1, 4, 24); } else { CSLAM_LAUNCH_PSYM4(0, 4, 24); }
        }
        else
        {
            if (nt) { CSLAM_LAUNCH_PSYM4(1, 4, 32); } else { CSLAM_LAUNCH_PSYM4(0, 4, 32); }
        }
#undef CSLAM_LAUNCH_PSYM4
        la_sig_add = 0; // (delivered)
    }
    else if (lower)
    {
        // any k, block-lower storage: the unpipelined persistent symmetric kernel (windows beyond 128 columns)
        hipLaunchKernelGGL((ekf_downdate_psym_f32<64, true, false>), dim3(G), block, 0, stream, dP, ldp, W, ldp, k8, dTiles,
                           n_sym_tiles, (long long*)nullptr);
    }
    else
    {
        // full storage (CSLAM_STORAGE=full): the same kernel with mirror stores
        hipLaunchKernelGGL((ekf_downdate_psym_f32<64, true, true>), dim3(G), block, 0, stream, dP, ldp, W, ldp, k8, dTiles,
                           n_sym_tiles, (long long*)nullptr);
    }
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

template <>
bool Ekf<float>::launch_gain_fast(int k, int n_pad, float* slot, const float* Gt, const float* U, const float* M)
{
    if (k > 128)
    {
        return false; // u is produced by the tuned factor kernels only
    }
    hipLaunchKernelGGL((ekf_panel_mfma_f32<false, true>), dim3(n_pad / 32, (k + 31) / 32), dim3(64), 0, stream, dPHT, ldp, n,
                       k, k, Gt, k, U, slot, ldp, dX, fuse_now ? (const float*)dPred : (const float*)nullptr, pp.w, dPv, ldp,
                       M, dWv);
    pose_fused_in_gain = (M != nullptr);
    return true;
}

template <>
bool Ekf<float>::launch_corr_fast(int k, const float* Wp, int kc)
{
    const int n_pad = round_up(n, kTile);
    hipLaunchKernelGGL((ekf_panel_mfma_f32<true, false>), dim3(n_pad / 32, (k + 31) / 32), dim3(64), 0, stream, Wp, ldp, n,
                       kc, k, dY, k, nullptr, dPHT, ldp, nullptr);
    return true;
}

template <>
bool Ekf<double>::launch_corr_fast(int k, const double* Wp, int kc)
{
    const int n_pad = round_up(n, kTile);
    hipLaunchKernelGGL((ekf_panel_mfma_f64<true, false>), dim3(n_pad / 16, (k + 15) / 16), dim3(64), 0, stream, Wp, ldp, n, kc,
                       k, dY, k, nullptr, dPHT, ldp, nullptr);
    return true;
}

template <>
bool Ekf<double>::launch_gain_fast(int k, int n_pad, double* slot, const double* Gt, const double* U, const double* M)
{
    if (k > 64) // u (and M) come from the tuned factor kernels
    {
        return false;
    }
    hipLaunchKernelGGL((ekf_panel_mfma_f64<false, true>), dim3(n_pad / 16, (k + 15) / 16), dim3(64), 0, stream, dPHT, ldp, n, k,
                       k, Gt, k, U, slot, ldp, dX, dPv, ldp, M, dWv, fuse_now ? (const double*)dPred : (const double*)nullptr,
                       pp.w);
    pose_fused_in_gain = (M != nullptr);
    return true;
}

template <>
int Ekf<double>::launch_downdate(const double* W, int k, hipStream_t stream)
{
    const int tiles_r = round_up(n, kTile) / kTile;
    // columns of W1 staged per pass: 16, register-staged and double-buffered (see the kernel).  The synchronous staging
    // loop (other values of CSLAM_F64_KCM) measured at N = 1000, k = 64: 8 / 16 / 32 -> 22.8 / 21.9 / 22.9 us, 64 (the whole
    // panel at once, 96 KB of LDS, one workgroup per CU) -> 30.5 us.
    int kcm = 16;
    if (const char* e = getenv("CSLAM_F64_KCM"))
    {
        kcm = std::max(4, std::min(64, round_up(atoi(e), 4)));
    }
    // tile width: 64 columns, or 32 for small states where the launch would not fill the chip (CSLAM_F64_CB overrides)
    int cb = (tiles_r * (round_up(n, kTile) / 64) < 4 * num_cus) ? 2 : 4;
    if (const char* e = getenv("CSLAM_F64_CB"))
    {
        cb = (atoi(e) == 2) ? 2 : 4;
    }
    const int tiles_c = round_up(n, kTile) / (16 * cb);
    const size_t lds  = (size_t)(kcm == 16 ? 2 : 1) * kcm * (128 + 16 * cb) * sizeof(double); // (16: two buffers)
    if (cb == 2)
    {
        hipLaunchKernelGGL(ekf_downdate_f64<2>, dim3(tiles_r * tiles_c), dim3(256), lds, stream, dP, ldp, W, ldp, k, tiles_r, lower,
                           kcm);
    }
    else
    {
        hipLaunchKernelGGL(ekf_downdate_f64<4>, dim3(tiles_r * tiles_c), dim3(256), lds, stream, dP, ldp, W, ldp, k, tiles_r, lower,
                           kcm);
    }
    CSLAM_HIP_TRY(hipGetLastError());
    return CSLAM_OK;
}

inline EkfBase* B(cslam_ekf_t h)
{
    return reinterpret_cast<EkfBase*>(h);
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* cslam_last_error(void)
{
    return last_error_buf();
}

int cslam_version(void)
{
    return CSLAM_VERSION;
}

int cslam_device_count(int* count)
{
    if (!count)
    {
        return fail(CSLAM_ERR_BAD_ARG, "device_count: null");
    }
    int        c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess)
    {
        *count = 0;
        return fail(CSLAM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return CSLAM_OK;
}

int cslam_ekf_create(int max_landmarks, int dtype, int device, int quirks, cslam_ekf_t* out)
{
    if (!out || max_landmarks < 0 || (dtype != CSLAM_F32 && dtype != CSLAM_F64) || (quirks & ~CSLAM_Q_REF_EXACT))
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_create: bad arguments");
    }
    *out  = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0)
    {
        return fail(CSLAM_ERR_NO_DEVICE, "ekf_create: no HIP device (this engine has no CPU fallback)");
    }
    if (device < 0)
    {
        if (hipGetDevice(&device) != hipSuccess)
        {
            device = 0;
        }
    }
    if (device >= c)
    {
        return fail(CSLAM_ERR_BAD_ARG, "ekf_create: device %d of %d", device, c);
    }
    EkfBase* b = nullptr;
    if (dtype == CSLAM_F32)
    {
        b = new (std::nothrow) Ekf<float>();
    }
    else
    {
        b = new (std::nothrow) Ekf<double>();
    }
    if (!b)
    {
        return fail(CSLAM_ERR_ALLOC, "ekf_create: out of host memory");
    }
    b->dtype  = dtype;
    b->device = device;
    b->quirks = quirks;
    b->nmax   = max_landmarks;
    b->ncap   = 3 + 2 * max_landmarks;
    b->ldp    = round_up(b->ncap, kTile);
    b->n      = 3;
    if (const char* sd = getenv("CSLAM_SEQ_DEFER"))
    {
        b->seq_defer = atoi(sd);
    }
    if (const char* ff = getenv("CSLAM_FUSE_F64"))
    {
        b->fuse_f64 = atoi(ff) ? 1 : 0;
    }
    if (const char* fp = getenv("CSLAM_FUSE_PREDICT"))
    {
        b->set_fuse_predict(atoi(fp));
    }
    // Block-lower storage (default): only the 128x128 tiles on / below the tile diagonal of the symmetric P are maintained
    // (the P-GEMM writes each tile once).  CSLAM_STORAGE=full keeps both triangles (mirror stores in the P-GEMM).
    b->lower = 1;
    if (const char* sv = getenv("CSLAM_STORAGE"))
    {
        b->lower = strcmp(sv, "full") ? 1 : 0;
    }
    // two-stream pipelining (see the top of this file) is an option, not the default: measured at N = 5000, k = 64
    // (profiles/r02_*): the kernels of update t+1 that touch memory crawl underneath the persistent P-GEMM (its waves
    // are older and keep ~24 KB of requests in flight each: the pending-panel correction takes 65 us instead of 5.5)
    // and every cross-stream hand-over costs ~6 us, so the period is 138 us against 112 us on one stream.
    b->pipeline = 0;
    if (const char* pv = getenv("CSLAM_PIPELINE"))
    {
        b->pipeline = atoi(pv) ? 1 : 0;
    }
    if (const char* xv = getenv("CSLAM_XCD_QUEUES"))
    {
        b->xcd_queues_req = atoi(xv) ? 1 : 0;
    }
    if (const char* lv = getenv("CSLAM_PGEMM_LIMBS"))
    {
        b->pgemm_limbs_req = atoi(lv);
    }
    if (const char* lk = getenv("CSLAM_LIMBS_KMIN"))
    {
        b->limbs_kmin_req = atoi(lk);
    }
    if (const char* lv = getenv("CSLAM_LOOKAHEAD"))
    {
        b->lookahead = atoi(lv) > 0 ? 1 : (atoi(lv) < 0 ? -1 : 0);
    }
    if (const char* gw = getenv("CSLAM_GATHER_WIDE"))
    {
        b->gather_corr_wide = atoi(gw) ? 1 : 0;
    }
    if (const char* sp = getenv("CSLAM_PGEMM_SPARE"))
    {
        b->pgemm_spare = std::max(0, atoi(sp));
    }
    int rc    = b->init();
    if (rc)
    {
        delete b;
        return rc;
    }
    g_engines.fetch_add(1);
    *out = reinterpret_cast<cslam_ekf_t>(b);
    return CSLAM_OK;
}

int cslam_ekf_destroy(cslam_ekf_t h)
{
    if (!h)
    {
        return CSLAM_OK;
    }
    (void)hipSetDevice(B(h)->device);
    delete B(h);
    g_engines.fetch_sub(1);
    return CSLAM_OK;
}

#define CSLAM_NEED(h)                                                 \
    if (!(h))                                                         \
    {                                                                 \
        return fail(CSLAM_ERR_BAD_ARG, "%s: null handle", __func__);  \
    }

int cslam_ekf_set_sync_mode(cslam_ekf_t h, int sync_mode)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->la_drain()) // (queued look-ahead updates belong to the asynchronous mode they were accepted in)
    {
        return rc;
    }
    B(h)->sync_mode = sync_mode ? 1 : 0;
    return CSLAM_OK;
}

int cslam_ekf_set_state(cslam_ekf_t h, const void* X, int n, const void* P, int ldp)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->set_state(X, n, P, ldp);
}

int cslam_ekf_get_state(cslam_ekf_t h, void* X, void* P, int ldp)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->get_state(X, P, ldp);
}

int cslam_ekf_get_x(cslam_ekf_t h, void* X, int capacity)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->get_x(X, capacity);
}

int cslam_ekf_get_n(cslam_ekf_t h, int* n)
{
    CSLAM_NEED(h);
    if (!n)
    {
        return fail(CSLAM_ERR_BAD_ARG, "get_n: null");
    }
    *n = B(h)->n;
    return CSLAM_OK;
}

int cslam_ekf_trace(cslam_ekf_t h, double* trace)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->trace(trace);
}

int cslam_ekf_synchronize(cslam_ekf_t h)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->sync_all();
}

int cslam_ekf_factor_status(cslam_ekf_t h, int* flags, int clear)
{
    CSLAM_NEED(h);
    return B(h)->factor_status(flags, clear);
}

int cslam_ekf_predict(cslam_ekf_t h, double v, double swa, const void* Q, double wb, double dt)
{
    CSLAM_NEED(h);
    return B(h)->predict(v, swa, Q, wb, dt);
}

int cslam_ekf_update(cslam_ekf_t h, const void* Z, int m, const void* R, const int* idf, int batch)
{
    CSLAM_NEED(h);
    return B(h)->update(Z, m, R, idf, batch, false);
}

int cslam_ekf_update_device(cslam_ekf_t h, const void* dZ, int m, const void* R, const int* d_idf, int batch)
{
    CSLAM_NEED(h);
    return B(h)->update(dZ, m, R, d_idf, batch, true);
}

int cslam_ekf_augment(cslam_ekf_t h, const void* Z, int q, const void* R)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->augment(Z, q, R);
}

int cslam_ekf_associate(cslam_ekf_t h, const void* Z, int m, const void* R, double gate1, double gate2, int* idf_out,
                        int* kind_out)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->associate(Z, m, R, gate1, gate2, idf_out, kind_out);
}

int cslam_ekf_observe_heading(cslam_ekf_t h, double phi, int use_heading)
{
    CSLAM_NEED(h);
    return B(h)->observe_heading(phi, use_heading); // (a pending predict rides in the same launch)
}

int cslam_ekf_set_pgemm_workgroups(cslam_ekf_t h, int workgroups)
{
    CSLAM_NEED(h);
    if (workgroups < 0)
    {
        return fail(CSLAM_ERR_BAD_ARG, "set_pgemm_workgroups: negative");
    }
    B(h)->pgemm_wgs = workgroups;
    return CSLAM_OK;
}

int cslam_ekf_get_streams(cslam_ekf_t h, void** chain_stream, void** pgemm_stream)
{
    CSLAM_NEED(h);
    if (chain_stream)
    {
        *chain_stream = reinterpret_cast<void*>(B(h)->stream);
    }
    if (pgemm_stream)
    {
        *pgemm_stream = reinterpret_cast<void*>(B(h)->stream_b);
    }
    return CSLAM_OK;
}

int cslam_ekf_run_many(cslam_ekf_t* handles, int count, int steps, const double* v, const double* swa, const void* Q,
                       double wb, double dt, const void* const* dZ, const int* const* d_idf, int m, const void* R,
                       int batch)
{
    if (!handles || count < 0 || steps < 0 || m < 0 || !R || !Q || (steps > 0 && (!v || !swa)) || (m > 0 && (!dZ || !d_idf)))
    {
        return fail(CSLAM_ERR_BAD_ARG, "run_many: bad arguments");
    }
    for (int i = 0; i < count; i++)
    {
        if (!handles[i] || (m > 0 && (!dZ[i] || !d_idf[i])))
        {
            return fail(CSLAM_ERR_BAD_ARG, "run_many: null handle or input for instance %d", i);
        }
    }
    // one host thread per instance: every instance is an independent filter on its own stream pair, so the launches of
    // different instances are issued concurrently and their kernels interleave on the device (no ordering between them)
    std::vector<int>         rcs((size_t)count, CSLAM_OK);
    std::vector<std::string> msgs((size_t)count);
    auto                     body = [&](int i) {
        EkfBase*     e     = B(handles[i]);
        const size_t esz   = (e->dtype == CSLAM_F32) ? sizeof(float) : sizeof(double);
        const char*  zbase = m > 0 ? static_cast<const char*>(dZ[i]) : nullptr;
        for (int t = 0; t < steps; t++)
        {
            int rc = e->predict(v[t], swa[t], Q, wb, dt);
            if (!rc && m > 0)
            {
                rc = e->update(zbase + (size_t)t * 2 * m * esz, m, R, d_idf[i] + (size_t)t * m, batch, true);
            }
            if (rc)
            {
                rcs[(size_t)i]  = rc;
                msgs[(size_t)i] = last_error_buf(); // (thread-local: carried back to the caller's thread below)
                return;
            }
        }
    };
    if (count == 1)
    {
        body(0);
    }
    else
    {
        std::vector<std::thread> th;
        th.reserve((size_t)count);
        for (int i = 0; i < count; i++)
        {
            th.emplace_back(body, i);
        }
        for (auto& t : th)
        {
            t.join();
        }
    }
    for (int i = 0; i < count; i++)
    {
        if (rcs[(size_t)i])
        {
            return fail(rcs[(size_t)i], "run_many: instance %d: %s", i, msgs[(size_t)i].c_str());
        }
    }
    return CSLAM_OK;
}

int cslam_ekf_set_profiling(cslam_ekf_t h, int on)
{
    CSLAM_NEED(h);
    return B(h)->set_profiling(on);
}

int cslam_ekf_get_stage_times(cslam_ekf_t h, double* ms_sum, int* launches)
{
    CSLAM_NEED(h);
    return B(h)->get_stage_times(ms_sum, launches);
}

int cslam_ekf_set_deferred(cslam_ekf_t h, int max_pending_columns)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->set_deferred(max_pending_columns);
}

int cslam_ekf_flush(cslam_ekf_t h)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->do_flush();
}

int cslam_ekf_debug_last_update(cslam_ekf_t h, void* PHT, void* S, void* G, void* W1, void* V, int* k)
{
    CSLAM_NEED(h);
    if (int rc = B(h)->resolve_predict())
    {
        return rc;
    }
    return B(h)->debug_last_update(PHT, S, G, W1, V, k);
}

} // extern "C"
