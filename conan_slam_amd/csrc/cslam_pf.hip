// cslam_pf.hip -- host side of the FastSLAM-2 particle store behind the C ABI of include/cslam.h.
//
// One handle owns np particles (one shard of the global particle set) in structure-of-arrays form in HBM
// and launches one lane per particle (or per particle x observation).  The only step of the reference that
// couples particles -- resampleParticles, PF.cpp:473-500 -- is exposed in pieces (weight sums, scaling,
// pack / unpack of particle records, local gather) so that a multi-GPU driver can put its collectives
// between them (conan_slam_amd/pf.py does that with torch.distributed over RCCL).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h> // types only: the library itself is bound with dlopen (see Rccl below)

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "cslam_common.hpp"
#include "pf_kernels.hpp"

using namespace cslam;

namespace
{

// ------------------------------------------------------------------------------------------------
// RCCL, bound at run time.  One process must hold ONE copy of librccl (PyTorch wheels bundle their own, as they do
// libamdhip64): dlopen by SONAME returns the copy the process already has, else the system one.
// ------------------------------------------------------------------------------------------------
struct Rccl
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                            = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                     = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                               = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t)        = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)               = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                     = nullptr;
    ncclResult_t (*GroupStart)()                                                                          = nullptr;
    ncclResult_t (*GroupEnd)()                                                                            = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                           = nullptr;
};

inline Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried)
    {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
            {
                break;
            }
        }
        if (r.lib)
        {
#define CSLAM_RCCL_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
            CSLAM_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
            CSLAM_RCCL_SYM(CommInitRank, "ncclCommInitRank");
            CSLAM_RCCL_SYM(CommDestroy, "ncclCommDestroy");
            CSLAM_RCCL_SYM(AllReduce, "ncclAllReduce");
            CSLAM_RCCL_SYM(AllGather, "ncclAllGather");
            CSLAM_RCCL_SYM(Send, "ncclSend");
            CSLAM_RCCL_SYM(Recv, "ncclRecv");
            CSLAM_RCCL_SYM(GroupStart, "ncclGroupStart");
            CSLAM_RCCL_SYM(GroupEnd, "ncclGroupEnd");
            CSLAM_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef CSLAM_RCCL_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.AllGather || !r.Send || !r.Recv ||
                !r.GroupStart || !r.GroupEnd)
            {
                r.lib = nullptr;
            }
        }
    }
    return r.lib ? &r : nullptr;
}

#define CSLAM_RCCL_TRY(expr)                                                                                         \
    do                                                                                                               \
    {                                                                                                                \
        ncclResult_t r__ = (expr);                                                                                   \
        if (r__ != ncclSuccess)                                                                                      \
        {                                                                                                            \
            return ::cslam::fail(CSLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                                      \
                                 rccl()->GetErrorString ? rccl()->GetErrorString(r__) : "rccl error", __FILE__, __LINE__); \
        }                                                                                                            \
    } while (0)

// ------------------------------------------------------------------------------------------------
// The communicator of the sharded resample.  Two back-ends behind one interface:
//   RCCL      one process (rank) per GPU, collectives over xGMI -- production;
//   loopback  `world` ranks that live in ONE process on ONE device, one host thread per rank (RCCL refuses the same device
//             twice in a communicator, SURVEY 7 "hard parts"): all-reduce / all-gather / send-recv are device-to-device
//             copies ordered by a host barrier.  It exists so that the multi-rank code paths of
//             cslam_pf_resample_sharded (ranks > 0, the exchange plan, the receive ordering) can run under test on a
//             one-GPU box; it is slow on purpose (every collective synchronises the calling rank's stream twice).
// ------------------------------------------------------------------------------------------------
constexpr int kLoopMaxWorld = 16;

struct LoopPtrs
{
    const double* p[kLoopMaxWorld];
};

__global__ void comm_loop_sum_kernel(LoopPtrs ptrs, int world, double* __restrict__ out, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
    {
        double s = 0.0;
        for (int r = 0; r < world; r++) // rank order: the same sum on every rank
        {
            s += ptrs.p[r][i];
        }
        out[i] = s;
    }
}

struct LoopShared
{
    int                     world = 1;
    int                     refs  = 0;
    std::mutex              mu;
    std::condition_variable cv;
    int                     arrived = 0;
    unsigned                gen     = 0;
    bool                    broken  = false; // a rank gave up (error / timeout): every later barrier fails at once
    std::vector<const void*> src;            // [rank] buffer registered for the collective in flight
    struct P2P
    {
        const void* ptr   = nullptr;
        size_t      bytes = 0;
    };
    std::vector<P2P> sends; // [from * world + to] of the group in flight

    // all `world` ranks arrive, or false after `seconds` (a peer failed and never came)
    bool barrier(double seconds = 60.0)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken)
        {
            return false;
        }
        const unsigned g = gen;
        if (++arrived == world)
        {
            arrived = 0;
            gen++;
            cv.notify_all();
            return true;
        }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(seconds), [&] { return gen != g || broken; });
        if (!ok || broken)
        {
            broken = true;
            cv.notify_all();
            return false;
        }
        return true;
    }
    void poison()
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

struct Comm
{
    ncclComm_t  comm   = nullptr; // RCCL back-end
    LoopShared* loop   = nullptr; // loopback back-end
    int         rank   = 0;
    int         world  = 1;
    int         device = 0;
    bool        in_group = false;
    struct Rv
    {
        void*  ptr;
        size_t bytes;
        int    peer;
    };
    std::vector<Rv> recvs; // loopback: receives of the open group

    int loop_fail(const char* what)
    {
        loop->poison();
        return ::cslam::fail(CSLAM_ERR_HIP, "loopback communicator: %s (rank %d of %d)", what, rank, world);
    }

    // recv[i] = sum over ranks of send[i], i < count doubles; identical on every rank
    int all_reduce_sum_f64(const double* send, double* recv, int count, hipStream_t st)
    {
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->AllReduce(send, recv, (size_t)count, ncclDouble, ncclSum, comm, st));
            return CSLAM_OK;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(st)); // `send` is complete
        loop->src[(size_t)rank] = send;
        if (!loop->barrier())
        {
            return loop_fail("all-reduce: a peer never arrived");
        }
        LoopPtrs ptrs{};
        for (int r = 0; r < world; r++)
        {
            ptrs.p[r] = static_cast<const double*>(loop->src[(size_t)r]);
        }
        hipLaunchKernelGGL(comm_loop_sum_kernel, dim3((count + 63) / 64), dim3(64), 0, st, ptrs, world, recv, count);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipStreamSynchronize(st)); // every peer's `send` has been read before anybody moves on
        if (!loop->barrier())
        {
            return loop_fail("all-reduce: a peer never finished");
        }
        return CSLAM_OK;
    }

    // recv[r * bytes .. (r+1) * bytes) = rank r's send
    int all_gather(const void* send, void* recv, size_t count, ncclDataType_t dt, size_t elt, hipStream_t st)
    {
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->AllGather(send, recv, count, dt, comm, st));
            return CSLAM_OK;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(st));
        loop->src[(size_t)rank] = send;
        if (!loop->barrier())
        {
            return loop_fail("all-gather: a peer never arrived");
        }
        const size_t bytes = count * elt;
        for (int r = 0; r < world; r++)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)r * bytes, loop->src[(size_t)r], bytes,
                                         hipMemcpyDeviceToDevice, st));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(st));
        if (!loop->barrier())
        {
            return loop_fail("all-gather: a peer never finished");
        }
        return CSLAM_OK;
    }

    int group_start()
    {
        in_group = true;
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->GroupStart());
            return CSLAM_OK;
        }
        recvs.clear();
        for (int r = 0; r < world; r++)
        {
            loop->sends[(size_t)rank * world + r] = LoopShared::P2P{};
        }
        return CSLAM_OK;
    }
    int send(const void* buf, size_t count, ncclDataType_t dt, size_t elt, int peer, hipStream_t st)
    {
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->Send(buf, count, dt, peer, comm, st));
            return CSLAM_OK;
        }
        loop->sends[(size_t)rank * world + peer] = LoopShared::P2P{buf, count * elt};
        return CSLAM_OK;
    }
    int recv(void* buf, size_t count, ncclDataType_t dt, size_t elt, int peer, hipStream_t st)
    {
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->Recv(buf, count, dt, peer, comm, st));
            return CSLAM_OK;
        }
        recvs.push_back(Rv{buf, count * elt, peer});
        return CSLAM_OK;
    }
    // closes the group on every path (a group left open would swallow the next collective)
    int group_end(hipStream_t st)
    {
        if (!in_group)
        {
            return CSLAM_OK;
        }
        in_group = false;
        if (!loop)
        {
            CSLAM_RCCL_TRY(rccl()->GroupEnd());
            return CSLAM_OK;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(st)); // the send buffers are packed
        if (!loop->barrier())
        {
            return loop_fail("send/recv: a peer never arrived");
        }
        for (const Rv& rv : recvs)
        {
            const LoopShared::P2P& sp = loop->sends[(size_t)rv.peer * world + rank];
            if (sp.ptr == nullptr || sp.bytes != rv.bytes)
            {
                return loop_fail("send/recv: a receive has no matching send of the same size");
            }
            CSLAM_HIP_TRY(hipMemcpyAsync(rv.ptr, sp.ptr, rv.bytes, hipMemcpyDeviceToDevice, st));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(st));
        if (!loop->barrier()) // nobody reuses a send buffer before its receiver has copied it
        {
            return loop_fail("send/recv: a peer never finished");
        }
        return CSLAM_OK;
    }
};

struct PfBase
{
    virtual ~PfBase() {}
    int         dtype  = CSLAM_F32;
    int         device = 0;
    int         quirks = CSLAM_Q_REF_EXACT;
    int         np     = 0;
    int         nfcap  = 0;
    int         nf     = 0;
    hipStream_t stream = nullptr;

    virtual int init()                                                                       = 0;
    virtual int set_uniform_weight(double w0)                                                 = 0;
    virtual int predict(double v, double swa, const void* Q, double wb, double dt)            = 0;
    virtual int observe_heading(double phi, int use)                                          = 0;
    virtual int sample_proposal(const void* Z, int m, const int* idf, const void* R, const void* normals) = 0;
    virtual int feature_update(const void* Z, int m, const int* idf, const void* R)           = 0;
    virtual int add_features(const void* Z, int q, const void* R)                             = 0;
    virtual int weight_sums(double* sums)                                                     = 0;
    virtual int scale_weights(double scale)                                                   = 0;
    virtual int weights_ptr(void** p)                                                         = 0;
    virtual int get_weights(void* w)                                                          = 0;
    virtual int set_weights(const void* w)                                                    = 0;
    virtual int record_bytes(long long* b)                                                    = 0;
    virtual int pack(const int* idx, int count, void* drec)                                   = 0;
    virtual int unpack(const int* idx, int count, const void* drec)                           = 0;
    virtual int gather_local(const int* keep, double w_new)                                   = 0;
    virtual int resample_local(const void* select, double n_eff, int status, double* neff, int* did) = 0;
    virtual int resample_sharded(Comm* c, const void* select, double n_eff, int status, double* neff, int* did) = 0;
    virtual int debug_last_exchange(int* counts, int* send_idx, int cap, int* n_send)                = 0;
    virtual int observation_step(double v, double swa, const void* Q, double wb, double dt, const void* Z, int m,
                                 const int* idf, const void* R, const void* normals, const void* select, double n_eff,
                                 int status) = 0;
    virtual int resample_stats(double* calls, double* resamples, double* last_neff) = 0;
    virtual int get_particle(int i, void* w, void* Xv, void* Pv, void* XF, void* PF)          = 0;
    virtual int set_particle(int i, const void* w, const void* Xv, const void* Pv, const void* XF, const void* PF,
                             int nf)                                                          = 0;
};

template <typename T>
struct Pf : PfBase
{
    T*      dW   = nullptr;
    T*      dXv  = nullptr;
    T*      dPv  = nullptr;
    T*      dXF  = nullptr;
    T*      dPF  = nullptr;
    // the twin set the single-pass resample gathers into; the two sets are swapped after every resample call
    T*      dXv2 = nullptr;
    T*      dPv2 = nullptr;
    T*      dXF2 = nullptr;
    T*      dPF2 = nullptr;
    T*      dObs = nullptr; // staging: Z (2*mcap T) | idf (mcap int) | normals (3*np T), filled by one copy per call
    int*    dIdx = nullptr; // index lists of pack/unpack (max(mcap, np))
    double* dSums = nullptr;
    T*      dRec = nullptr; // scratch for gather_local
    int     mcap = 0;

    // Pinned staging ring.  The small host inputs of a call (Z, idf, normals, select) are copied into the next slot and
    // sent with ONE asynchronous copy, so the call returns without waiting for the stream (the caller's arrays are
    // consumed before return all the same).  The stream is drained once per lap of the ring, never per call.
    static constexpr int kStageSlots = 16;
    char*             hStage         = nullptr;
    size_t            stage_slot     = 0;
    int               stage_pos      = 0;
    int               stage_inflight = 0;
    hipEvent_t        stage_ev[kStageSlots] = {};
    bool              stage_ev_set[kStageSlots] = {};
    int               stage_last = 0; // slot handed out by the last stage_slot_for()
    std::vector<char> staged; // Z || idf bytes currently in dObs (empty = unknown)
    double*           hInfo = nullptr;

    ~Pf() override
    {
        if (stream)
        {
            (void)hipStreamSynchronize(stream);
        }
        (void)hipFree(dW);
        (void)hipFree(dSumsG);
        (void)hipFree(dWall);
        (void)hipFree(dSelG);
        (void)hipFree(dCumG);
        (void)hipFree(dKeepG);
        (void)hipFree(dSendIdx);
        (void)hipFree(dCounts);
        (void)hipFree(dSendBuf);
        (void)hipFree(dRecvBuf);
        if (hCounts)
        {
            (void)hipHostFree(hCounts);
        }
        (void)hipFree(dSel);
        (void)hipFree(dCum);
        (void)hipFree(dKeep);
        (void)hipFree(dEnable);
        (void)hipFree(dInfo);
        (void)hipFree(dXv);
        (void)hipFree(dPv);
        (void)hipFree(dXF);
        (void)hipFree(dPF);
        (void)hipFree(dXv2);
        (void)hipFree(dPv2);
        (void)hipFree(dXF2);
        (void)hipFree(dPF2);
        (void)hipFree(dObs);
        (void)hipFree(dIdx);
        (void)hipFree(dSums);
        (void)hipFree(dRec);
        for (int i = 0; i < kStageSlots; i++)
        {
            if (stage_ev[i])
            {
                (void)hipEventDestroy(stage_ev[i]);
            }
        }
        (void)hipHostFree(hStage);
        (void)hipHostFree(hInfo);
        if (stream)
        {
            (void)hipStreamDestroy(stream);
        }
    }

    size_t off_idf() const
    {
        return (size_t)2 * mcap * sizeof(T);
    }
    size_t off_normals() const
    {
        return off_idf() + (size_t)mcap * sizeof(int);
    }
    int* dIdf() const
    {
        return reinterpret_cast<int*>(reinterpret_cast<char*>(dObs) + off_idf());
    }
    T* dNormals() const
    {
        return reinterpret_cast<T*>(reinterpret_cast<char*>(dObs) + off_normals());
    }

    int stage_slot_for(size_t bytes, char** out)
    {
        if (bytes > stage_slot)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            (void)hipHostFree(hStage);
            hStage         = nullptr;
            size_t newsz   = std::max((bytes + 4095) / 4096 * 4096, 2 * stage_slot);
            stage_slot     = 0;
            CSLAM_HIP_TRY(hipHostMalloc(&hStage, newsz * kStageSlots, hipHostMallocDefault));
            stage_slot     = newsz;
            stage_pos      = 0;
            stage_inflight = 0; // (the slots' events are all complete after the synchronisation above)
        }
        // a slot is reused one lap later: wait for the copy that read it last (long done in the steady state) instead of
        // draining the stream once per lap (which cost a ~60 us bubble every 16 calls)
        if (stage_ev_set[stage_pos])
        {
            CSLAM_HIP_TRY(hipEventSynchronize(stage_ev[stage_pos]));
        }
        stage_last = stage_pos;
        *out      = hStage + (size_t)stage_pos * stage_slot;
        stage_pos = (stage_pos + 1) % kStageSlots;
        stage_inflight++;
        return CSLAM_OK;
    }

    // the copy out of the slot handed out last has been enqueued: mark it
    int stage_commit()
    {
        if (!stage_ev_set[stage_last])
        {
            CSLAM_HIP_TRY(hipEventCreateWithFlags(&stage_ev[stage_last], hipEventDisableTiming));
            stage_ev_set[stage_last] = true;
        }
        CSLAM_HIP_TRY(hipEventRecord(stage_ev[stage_last], stream));
        return CSLAM_OK;
    }


    int use_device()
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        return CSLAM_OK;
    }

    PfStore<T> store() const
    {
        PfStore<T> s;
        s.w  = dW;
        s.xv = dXv;
        s.pv = dPv;
        s.xf = dXF;
        s.pf = dPF;
        s.np = np;
        s.nf = nf;
        return s;
    }

    int ensure_m(int m)
    {
        if (m <= mcap)
        {
            return CSLAM_OK;
        }
        int newm = (std::max(m, std::max(64, 2 * mcap)) + 3) / 4 * 4; // keeps the normals 16-byte aligned
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(dObs);
        (void)hipFree(dIdx);
        dObs = nullptr;
        dIdx = nullptr;
        mcap = 0;
        staged.clear();
        CSLAM_HIP_TRY(hipMalloc(&dObs, ((size_t)2 * newm + (size_t)4 * np) * sizeof(T) + (size_t)newm * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dIdx, (size_t)std::max(newm, np) * sizeof(int)));
        mcap = newm;
        return CSLAM_OK;
    }

    int init() override
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        size_t n1 = (size_t)np;
        size_t cf = (size_t)std::max(nfcap, 1);
        CSLAM_HIP_TRY(hipMalloc(&dW, n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dXv, 3 * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dPv, 9 * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dXF, 2 * cf * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dPF, 4 * cf * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dSums, 2 * sizeof(double)));
        CSLAM_HIP_TRY(hipMalloc(&dRec, n1 * (13 + 6 * cf) * sizeof(T)));
        // PF.cpp:319-341: X = 0, P = 0, empty map; w = 1/np until the driver sets the global value
        CSLAM_HIP_TRY(hipMemsetAsync(dXv, 0, 3 * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dPv, 0, 9 * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dXF, 0, 2 * cf * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dPF, 0, 4 * cf * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMalloc(&dXv2, 3 * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dPv2, 9 * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dXF2, 2 * cf * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dPF2, 4 * cf * n1 * sizeof(T)));
        CSLAM_HIP_TRY(hipMemsetAsync(dXv2, 0, 3 * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dPv2, 0, 9 * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dXF2, 0, 2 * cf * n1 * sizeof(T), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dPF2, 0, 4 * cf * n1 * sizeof(T), stream));
        rc = ensure_m(64);
        if (rc)
        {
            return rc;
        }
        rc = set_uniform_weight(1.0 / (double)np);
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int set_uniform_weight(double w0) override
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_scale_weights_kernel<T>, dim3((np + 255) / 256), dim3(256), 0, stream, dW, np, (T)w0, 1);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int predict(double v, double swa, const void* Qv, double wb, double dt) override
    {
        if (!Qv)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_predict: Q is null");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const T* Q = static_cast<const T*>(Qv);
        hipLaunchKernelGGL(pf_predict_kernel<T>, dim3((np + 63) / 64), dim3(64), 0, stream, store(), (T)v, (T)swa, Q[0],
                           Q[1], Q[2], Q[3], (T)wb, (T)dt);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int observe_heading(double phi, int use) override
    {
        if (!use)
        {
            return CSLAM_OK;
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        T sigma = (T)(((double)0.01f * kPi) / 180.0); // PF.cpp:391
        hipLaunchKernelGGL(pf_heading_kernel<T>, dim3((np + 63) / 64), dim3(64), 0, stream, store(), (T)phi,
                           sigma * sigma);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int check_idf(const int* idf, int m, const char* who)
    {
        for (int i = 0; i < m; i++)
        {
            if (idf[i] < 1 || idf[i] > nf)
            {
                return fail(CSLAM_ERR_BAD_ARG, "%s: idf[%d]=%d outside 1..%d", who, i, idf[i], nf);
            }
        }
        return CSLAM_OK;
    }

    // stage Z (2*m) and idf (m) of one call, plus `extra` (the normals) when given; inputs are consumed before return.
    // A call whose Z/idf are byte-identical to what dObs already holds (featureUpdate right after sampleProposal,
    // PF.cpp:150-156) sends nothing.
    int stage(const void* Z, int m, const int* idf, const void* extra = nullptr, size_t extra_bytes = 0)
    {
        int rc = ensure_m(m);
        if (rc)
        {
            return rc;
        }
        const size_t zb = (size_t)2 * m * sizeof(T), ib = idf ? (size_t)m * sizeof(int) : 0;
        const bool   same = !extra && !staged.empty() && staged.size() == zb + ib && std::memcmp(staged.data(), Z, zb) == 0 &&
                          (ib == 0 || std::memcmp(staged.data() + zb, idf, ib) == 0);
        if (same)
        {
            return CSLAM_OK;
        }
        const size_t bytes = extra ? off_normals() + extra_bytes : (idf ? off_idf() + ib : zb);
        char*        slot  = nullptr;
        if ((rc = stage_slot_for(bytes, &slot)))
        {
            return rc;
        }
        if (zb)
        {
            std::memcpy(slot, Z, zb);
        }
        if (ib)
        {
            std::memcpy(slot + off_idf(), idf, ib);
        }
        if (extra)
        {
            std::memcpy(slot + off_normals(), extra, extra_bytes);
        }
        staged.clear();
        CSLAM_HIP_TRY(hipMemcpyAsync(dObs, slot, bytes, hipMemcpyHostToDevice, stream));
        if ((rc = stage_commit()))
        {
            return rc;
        }
        staged.resize(zb + ib);
        if (zb)
        {
            std::memcpy(staged.data(), Z, zb);
        }
        if (ib)
        {
            std::memcpy(staged.data() + zb, idf, ib);
        }
        return CSLAM_OK;
    }

    int sample_proposal(const void* Z, int m, const int* idf, const void* Rv, const void* normals) override
    {
        if (m < 0 || !Rv || !normals || (m > 0 && (!Z || !idf)))
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_sample_proposal: bad arguments");
        }
        int rc = use_device();
        if (rc || (rc = check_idf(idf, m, "pf_sample_proposal")) ||
            (rc = stage(Z, m, idf, normals, (size_t)3 * np * sizeof(T))))
        {
            return rc;
        }
        const T* R = static_cast<const T*>(Rv);
        hipLaunchKernelGGL(pf_sample_proposal_kernel<T>, dim3((np * kPfSubLanes + 63) / 64), dim3(64), 0, stream, store(), dObs, dIdf(), m,
                           R[0], R[1], R[2], R[3], dNormals(), PfPredict<T>{0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0, (T)0}, 0);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int feature_update(const void* Z, int m, const int* idf, const void* Rv) override
    {
        if (m < 0 || !Rv || (m > 0 && (!Z || !idf)))
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_feature_update: bad arguments");
        }
        if (m == 0)
        {
            return CSLAM_OK;
        }
        int rc = use_device();
        if (rc || (rc = check_idf(idf, m, "pf_feature_update")) || (rc = stage(Z, m, idf)))
        {
            return rc;
        }
        const T* R = static_cast<const T*>(Rv);
        hipLaunchKernelGGL(pf_feature_update_kernel<T>, dim3((np + 63) / 64, m), dim3(64), 0, stream, store(), dObs, dIdf(),
                           m, R[0], R[1], R[2], R[3], (quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 0 : 1);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int add_features(const void* Z, int q, const void* Rv) override
    {
        if (q < 0 || !Rv || (q > 0 && !Z))
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_add_features: bad arguments");
        }
        if (q == 0)
        {
            return CSLAM_OK;
        }
        if (nf + q > nfcap)
        {
            return fail(CSLAM_ERR_CAPACITY, "pf_add_features: %d features would exceed max_features=%d", nf + q, nfcap);
        }
        int rc = use_device();
        if (rc || (rc = stage(Z, q, nullptr)))
        {
            return rc;
        }
        const T* R = static_cast<const T*>(Rv);
        hipLaunchKernelGGL(pf_add_features_kernel<T>, dim3((np + 63) / 64, q), dim3(64), 0, stream, store(), dObs, q, R[0],
                           R[1], R[2], R[3]);
        CSLAM_HIP_TRY(hipGetLastError());
        nf += q;
        return CSLAM_OK;
    }

    int weight_sums(double* sums) override
    {
        if (!sums)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_weight_sums: null");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_weight_sums_kernel<T>, dim3(1), dim3(256), 0, stream, dW, np, dSums);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipMemcpyAsync(sums, dSums, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int scale_weights(double scale) override
    {
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_scale_weights_kernel<T>, dim3((np + 255) / 256), dim3(256), 0, stream, dW, np, (T)scale, 0);
        CSLAM_HIP_TRY(hipGetLastError());
        return CSLAM_OK;
    }

    int weights_ptr(void** p) override
    {
        if (!p)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_weights_device_ptr: null");
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream)); // the caller will read it from another stream
        *p = dW;
        return CSLAM_OK;
    }

    int get_weights(void* w) override
    {
        if (!w)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_get_weights: null");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(w, dW, (size_t)np * sizeof(T), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int set_weights(const void* w) override
    {
        if (!w)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_set_weights: null");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(dW, w, (size_t)np * sizeof(T), hipMemcpyHostToDevice, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int record_bytes(long long* b) override
    {
        if (!b)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_record_bytes: null");
        }
        *b = (long long)(13 + 6 * nf) * (long long)sizeof(T);
        return CSLAM_OK;
    }

    int stage_idx(const int* idx, int count, const char* who)
    {
        if (count < 0 || (count > 0 && !idx))
        {
            return fail(CSLAM_ERR_BAD_ARG, "%s: bad index list", who);
        }
        for (int i = 0; i < count; i++)
        {
            if (idx[i] < 0 || idx[i] >= np)
            {
                return fail(CSLAM_ERR_BAD_ARG, "%s: index %d outside 0..%d", who, idx[i], np - 1);
            }
        }
        int rc = ensure_m(count);
        if (rc)
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(dIdx, idx, (size_t)count * sizeof(int), hipMemcpyHostToDevice, stream));
        return CSLAM_OK;
    }

    int pack(const int* idx, int count, void* drec) override
    {
        if (count == 0)
        {
            return CSLAM_OK;
        }
        if (!drec)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_pack: null buffer");
        }
        int rc = use_device();
        if (rc || (rc = stage_idx(idx, count, "pf_pack")))
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_pack_kernel<T>, dim3(count), dim3(256), 0, stream, store(), dIdx, count,
                           static_cast<T*>(drec));
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipStreamSynchronize(stream)); // the buffer is handed to a collective on another stream
        return CSLAM_OK;
    }

    int unpack(const int* idx, int count, const void* drec) override
    {
        if (count == 0)
        {
            return CSLAM_OK;
        }
        if (!drec)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_unpack: null buffer");
        }
        int rc = use_device();
        if (rc || (rc = stage_idx(idx, count, "pf_unpack")))
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_unpack_kernel<T>, dim3(count), dim3(256), 0, stream, store(), dIdx, count,
                           static_cast<const T*>(drec));
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    // PF.cpp:490-499 for a single shard: slot i <- particle keep[i], weights = w_new
    int gather_local(const int* keep, double w_new) override
    {
        int rc = use_device();
        if (rc || (rc = stage_idx(keep, np, "pf_gather_local")))
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_pack_kernel<T>, dim3(np), dim3(256), 0, stream, store(), dIdx, np, dRec);
        CSLAM_HIP_TRY(hipGetLastError());
        std::vector<int> ident((size_t)np);
        for (int i = 0; i < np; i++)
        {
            ident[i] = i;
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        CSLAM_HIP_TRY(hipMemcpyAsync(dIdx, ident.data(), (size_t)np * sizeof(int), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(pf_unpack_kernel<T>, dim3(np), dim3(256), 0, stream, store(), dIdx, np, dRec);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return set_uniform_weight(w_new);
    }

    // PF.cpp:473-500 for a single shard that holds the whole particle set, without leaving the device: plan
    // (sums, normalise, Neff, decision, keep[]) -> pack(keep) -> unpack(identity) -> w = 1/N, the last three gated by a
    // device flag.  One D2H of {Neff, flag} at the end, and only if the caller asks for them.
    T*      dSel  = nullptr;
    T*      dCum  = nullptr;
    int*    dKeep = nullptr;
    int*    dEnable = nullptr;
    double* dInfo = nullptr;
    int resample_local(const void* select, double n_eff, int status, double* neff, int* did) override
    {
        if (!select)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_resample_local: null select");
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        if (!dSel)
        {
            CSLAM_HIP_TRY(hipMalloc(&dSel, (size_t)np * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dCum, (size_t)np * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dKeep, (size_t)np * sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dEnable, sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dInfo, 4 * sizeof(double)));
            CSLAM_HIP_TRY(hipMemsetAsync(dInfo, 0, 4 * sizeof(double), stream));
            CSLAM_HIP_TRY(hipHostMalloc(&hInfo, 4 * sizeof(double), hipHostMallocDefault));
        }
        char* slot = nullptr;
        if ((rc = stage_slot_for((size_t)np * sizeof(T), &slot)))
        {
            return rc;
        }
        std::memcpy(slot, select, (size_t)np * sizeof(T));
        CSLAM_HIP_TRY(hipMemcpyAsync(dSel, slot, (size_t)np * sizeof(T), hipMemcpyHostToDevice, stream));
        if ((rc = stage_commit()))
        {
            return rc;
        }
        if ((rc = launch_resample(dSel, n_eff, status)))
        {
            return rc;
        }
        if (neff || did)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(hInfo, dInfo, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            stage_inflight = 0;
            if (neff)
            {
                *neff = hInfo[0];
            }
            if (did)
            {
                *did = hInfo[1] != 0.0 ? 1 : 0;
            }
        }
        return CSLAM_OK;
    }

    // PF.cpp:473-500 over a particle set sharded across ranks (one rank per GPU): see cslam_pf_resample_sharded in
    // include/cslam.h.  Everything is ordered on the handle's stream; the host reads back the two global sums (the
    // decision must be the same on every rank and drives which collectives run) and, when it resamples, the
    // 2 x world record counts of the exchange.
    double* dSumsG   = nullptr;
    T*      dWall    = nullptr;
    T*      dSelG    = nullptr;
    T*      dCumG    = nullptr; // running sum of the gathered weights (pf_keep_kernel)
    int*    dKeepG   = nullptr;
    int*    dSendIdx = nullptr;
    int*    dCounts  = nullptr;
    int*    hCounts  = nullptr;
    T*      dSendBuf = nullptr;
    T*      dRecvBuf = nullptr;
    int     sh_world = 0;
    int     sh_nf    = -1;
    std::vector<int> last_counts; // 2 * world record counts of the last exchange (send per destination, receive per source)
    int              last_n_send = 0;

    template <typename P>
    static void refree(P*& p)
    {
        (void)hipFree(p);
        p = nullptr;
    }

    // buffers of the sharded resample: everything that can fail is allocated BEFORE the first collective, so that a rank
    // never leaves its peers waiting inside one because a local allocation failed
    int ensure_sharded_buffers(int world)
    {
        const int N = np * world;
        if (sh_world != world)
        {
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            refree(dSumsG);
            refree(dWall);
            refree(dSelG);
            refree(dCumG);
            refree(dKeepG);
            refree(dSendIdx);
            refree(dCounts);
            if (hCounts)
            {
                (void)hipHostFree(hCounts);
                hCounts = nullptr;
            }
            sh_world = 0;
            sh_nf    = -1;
            CSLAM_HIP_TRY(hipMalloc(&dSumsG, 2 * sizeof(double)));
            CSLAM_HIP_TRY(hipMalloc(&dWall, (size_t)N * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dSelG, (size_t)N * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dCumG, (size_t)N * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dKeepG, (size_t)N * sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dSendIdx, (size_t)N * sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dCounts, (size_t)2 * world * sizeof(int)));
            CSLAM_HIP_TRY(hipHostMalloc(&hCounts, ((size_t)2 * world + 4) * sizeof(double), hipHostMallocDefault));
            sh_world = world;
        }
        if (sh_nf != nf)
        {
            const size_t rec = (size_t)(13 + 6 * nf);
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            refree(dSendBuf);
            refree(dRecvBuf);
            sh_nf = -1;
            CSLAM_HIP_TRY(hipMalloc(&dSendBuf, (size_t)N * rec * sizeof(T))); // worst case: every slot keeps a particle of this rank
            CSLAM_HIP_TRY(hipMalloc(&dRecvBuf, (size_t)np * rec * sizeof(T)));
            sh_nf = nf;
        }
        return CSLAM_OK;
    }

    int resample_sharded(Comm* c, const void* select, double n_eff, int status, double* neff, int* did) override
    {
        if (!c || !select)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_resample_sharded: null communicator or select");
        }
        if (!c->loop && !rccl())
        {
            return fail(CSLAM_ERR_HIP, "pf_resample_sharded: librccl could not be loaded");
        }
        const int world = c->world, rank = c->rank, L = np, N = np * world;
        int rc = use_device();
        if (rc || (rc = ensure_sharded_buffers(world)))
        {
            return rc;
        }
        const ncclDataType_t dt = (sizeof(T) == 4) ? ncclFloat : ncclDouble;
        // the strata positions go to the device up front as well (staging can fail; the copy is cheap when unused)
        char* slot = nullptr;
        if ((rc = stage_slot_for((size_t)N * sizeof(T), &slot)))
        {
            return rc;
        }
        std::memcpy(slot, select, (size_t)N * sizeof(T));
        CSLAM_HIP_TRY(hipMemcpyAsync(dSelG, slot, (size_t)N * sizeof(T), hipMemcpyHostToDevice, stream));
        if ((rc = stage_commit()))
        {
            return rc;
        }
        // 1. global weight sums
        hipLaunchKernelGGL(pf_weight_sums_kernel<T>, dim3(1), dim3(256), 0, stream, dW, np, dSums);
        CSLAM_HIP_TRY(hipGetLastError());
        if ((rc = c->all_reduce_sum_f64(dSums, dSumsG, 2, stream)))
        {
            return rc;
        }
        double* hs = reinterpret_cast<double*>(hCounts); // (pinned; the counts use it later)
        CSLAM_HIP_TRY(hipMemcpyAsync(hs, dSumsG, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        stage_inflight = 0;
        const double ws = hs[0], ws2 = hs[1];
        // 2. w /= ws (PF.cpp:482-487), Neff = 1 / sum (w/ws)^2 (PF.cpp:549-554), the decision (PF.cpp:490)
        hipLaunchKernelGGL(pf_scale_weights_kernel<T>, dim3((np + 255) / 256), dim3(256), 0, stream, dW, np, (T)(1.0 / ws), 0);
        CSLAM_HIP_TRY(hipGetLastError());
        const double ne = (ws2 > 0.0) ? (ws * ws) / ws2 : 0.0;
        const bool   go = (ne < n_eff) && status;
        if (neff)
        {
            *neff = ne;
        }
        if (did)
        {
            *did = go ? 1 : 0;
        }
        last_counts.assign((size_t)2 * world, 0);
        last_n_send = 0;
        if (!go)
        {
            return CSLAM_OK;
        }
        // 3. every rank plans the same keep[] from the gathered weights and the shared strata
        if ((rc = c->all_gather(dW, dWall, (size_t)L, dt, sizeof(T), stream)))
        {
            return rc;
        }
        hipLaunchKernelGGL(pf_keep_kernel<T>, dim3(1), dim3(256), 0, stream, dWall, N, dSelG, dKeepG, dCumG);
        CSLAM_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(pf_exchange_plan_kernel<0>, dim3(1), dim3(256), 0, stream, dKeepG, N, L, rank, world, dSendIdx,
                           dCounts);
        CSLAM_HIP_TRY(hipGetLastError());
        int* hc = reinterpret_cast<int*>(hCounts) + 8;
        CSLAM_HIP_TRY(hipMemcpyAsync(hc, dCounts, (size_t)2 * world * sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        stage_inflight = 0;
        int n_send = 0, n_recv = 0;
        for (int r = 0; r < world; r++)
        {
            n_send += hc[r];
            n_recv += hc[world + r];
        }
        last_counts.assign(hc, hc + 2 * world);
        last_n_send = n_send;
        // (every rank derives the plan from the same gathered weights, so these checks fail on all ranks or on none)
        if (n_recv != L || hc[rank] != hc[world + rank])
        {
            return fail(CSLAM_ERR_HIP, "pf_resample_sharded: inconsistent exchange plan (%d of %d slots filled, self %d / %d)",
                        n_recv, L, hc[rank], hc[world + rank]);
        }
        const size_t rec = (size_t)(13 + 6 * nf);
        // 4. records out of the store (before any slot is overwritten), exchange, records into the slots in order
        if (n_send > 0)
        {
            hipLaunchKernelGGL(pf_pack_kernel<T>, dim3(n_send), dim3(256), 0, stream, store(), dSendIdx, n_send, dSendBuf);
            CSLAM_HIP_TRY(hipGetLastError());
        }
        // the records that stay on this rank: a device copy, outside the group
        {
            size_t soff = 0, roff = 0;
            for (int r = 0; r < rank; r++)
            {
                soff += (size_t)hc[r];
                roff += (size_t)hc[world + r];
            }
            if (hc[rank] > 0)
            {
                CSLAM_HIP_TRY(hipMemcpyAsync(dRecvBuf + roff * rec, dSendBuf + soff * rec, (size_t)hc[rank] * rec * sizeof(T),
                                             hipMemcpyDeviceToDevice, stream));
            }
        }
        if ((rc = c->group_start()))
        {
            return rc;
        }
        size_t soff = 0, roff = 0;
        for (int r = 0; r < world && rc == CSLAM_OK; r++)
        {
            const size_t sc = (size_t)hc[r], rcv = (size_t)hc[world + r];
            if (r != rank)
            {
                if (sc > 0)
                {
                    rc = c->send(dSendBuf + soff * rec, sc * rec, dt, sizeof(T), r, stream);
                }
                if (rcv > 0 && rc == CSLAM_OK)
                {
                    rc = c->recv(dRecvBuf + roff * rec, rcv * rec, dt, sizeof(T), r, stream);
                }
            }
            soff += sc;
            roff += rcv;
        }
        const int rc_end = c->group_end(stream); // (closed on the failure path too)
        if (rc || rc_end)
        {
            return rc ? rc : rc_end;
        }
        hipLaunchKernelGGL(pf_unpack_kernel<T>, dim3(L), dim3(256), 0, stream, store(), (const int*)nullptr, L, dRecvBuf);
        CSLAM_HIP_TRY(hipGetLastError());
        return set_uniform_weight(1.0 / (double)N); // PF.cpp:495-499
    }

    // test introspection: the record counts (2 * world) and the send list of the last sharded resample
    int debug_last_exchange(int* counts, int* send_idx, int cap, int* n_send) override
    {
        if (n_send)
        {
            *n_send = last_n_send;
        }
        if (counts)
        {
            for (size_t i = 0; i < last_counts.size(); i++)
            {
                counts[i] = last_counts[i];
            }
        }
        if (send_idx && last_n_send > 0)
        {
            if (cap < last_n_send)
            {
                return fail(CSLAM_ERR_BAD_ARG, "debug_last_exchange: capacity %d < %d", cap, last_n_send);
            }
            int rc = use_device();
            if (rc)
            {
                return rc;
            }
            CSLAM_HIP_TRY(hipMemcpyAsync(send_idx, dSendIdx, (size_t)last_n_send * sizeof(int), hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
            stage_inflight = 0;
        }
        return CSLAM_OK;
    }

    // plan (sums, normalise, Neff, decision, keep[]) -> gather -> copy back + w = 1/N, the last two gated by a device flag
    int launch_resample(const T* d_select, double n_eff, int status)
    {
        hipLaunchKernelGGL(pf_resample_plan_kernel<T>, dim3(1), dim3(256), 0, stream, dW, np, d_select, n_eff, status, dCum,
                           dKeep, dInfo, dEnable);
        CSLAM_HIP_TRY(hipGetLastError());
        const dim3 ggrid(13 + 6 * store().nf, (np + 255) / 256);
        const T    w_new = (T)(1.0 / (double)np);
        PfStore<T> twin  = store();
        twin.xv          = dXv2;
        twin.pv          = dPv2;
        twin.xf          = dXF2;
        twin.pf          = dPF2;
        hipLaunchKernelGGL(pf_gather_move_kernel<T>, ggrid, dim3(256), 0, stream, store(), twin, dKeep, dEnable, w_new);
        CSLAM_HIP_TRY(hipGetLastError());
        std::swap(dXv, dXv2); // the twin set is the store now (whether particles moved or were copied in place)
        std::swap(dPv, dPv2);
        std::swap(dXF, dXF2);
        std::swap(dPF, dPF2);
        return CSLAM_OK;
    }

    int ensure_resample_buffers()
    {
        if (!dSel)
        {
            CSLAM_HIP_TRY(hipMalloc(&dSel, (size_t)np * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dCum, (size_t)np * sizeof(T)));
            CSLAM_HIP_TRY(hipMalloc(&dKeep, (size_t)np * sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dEnable, sizeof(int)));
            CSLAM_HIP_TRY(hipMalloc(&dInfo, 4 * sizeof(double)));
            CSLAM_HIP_TRY(hipMemsetAsync(dInfo, 0, 4 * sizeof(double), stream));
            CSLAM_HIP_TRY(hipHostMalloc(&hInfo, 4 * sizeof(double), hipHostMallocDefault));
        }
        return CSLAM_OK;
    }

    // One whole FastSLAM-2 observation step for a shard that holds every particle -- predict, sampleProposal,
    // featureUpdate, resampleParticles (PF.cpp:419-471, 502-544, 222-277, 473-500) -- with ONE staged host-to-device
    // copy for all its small inputs (Z | idf | normals | select) and nothing returned to the host.
    int observation_step(double v, double swa, const void* Qv, double wb, double dt, const void* Z, int m, const int* idf,
                         const void* Rv, const void* normals, const void* select, double n_eff, int status) override
    {
        if (!Qv || !Rv || m < 0 || (m > 0 && (!Z || !idf || !normals)) || !select)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_observation_step: bad arguments");
        }
        int rc = use_device();
        if (rc || (rc = check_idf(idf, m, "pf_observation_step")) || (rc = ensure_m(std::max(m, 1))) ||
            (rc = ensure_resample_buffers()))
        {
            return rc;
        }
        const size_t zb = (size_t)2 * m * sizeof(T), ib = (size_t)m * sizeof(int), nb = (size_t)3 * np * sizeof(T);
        const size_t off_sel = off_normals() + nb, bytes = off_sel + (size_t)np * sizeof(T);
        char*        slot    = nullptr;
        if ((rc = stage_slot_for(bytes, &slot)))
        {
            return rc;
        }
        if (m > 0)
        {
            std::memcpy(slot, Z, zb);
            std::memcpy(slot + off_idf(), idf, ib);
            std::memcpy(slot + off_normals(), normals, nb);
        }
        std::memcpy(slot + off_sel, select, (size_t)np * sizeof(T));
        // (zero-copy -- the kernels reading the pinned slot over the host link -- was tried instead of this staged copy:
        // 13.7 k instead of 15.0 k steps/s)
        staged.clear();
        // (a small kernel reading the pinned slot in place of this copy command: 19.6 k instead of 19.9 k steps/s)
        // (a second stream for this copy, double-buffered inputs and event hand-overs so that it runs under the previous
        // step's kernels was tried: 14.8 k instead of 16.7 k steps/s -- four more runtime calls per step cost more host
        // time than the 10 us of stream time they free)
        CSLAM_HIP_TRY(hipMemcpyAsync(dObs, slot, bytes, hipMemcpyHostToDevice, stream));
        if ((rc = stage_commit()))
        {
            return rc;
        }
        char*      base = reinterpret_cast<char*>(dObs);
        const T*   sZ   = reinterpret_cast<const T*>(base);
        const int* sIdf = reinterpret_cast<const int*>(base + off_idf());
        const T*   sNrm = reinterpret_cast<const T*>(base + off_normals());
        const T* Q = static_cast<const T*>(Qv);
        const T* R = static_cast<const T*>(Rv);
        if (m > 0) // predict rides inside the proposal kernel (which overwrites xv / Pv anyway)
        {
            const PfPredict<T> pr{1, (T)v, (T)swa, Q[0], Q[1], Q[2], Q[3], (T)wb, (T)dt};
            // ... and so does the feature update, unless an observation list names a feature twice (the separate kernel
            // then updates it twice from the same old value, last writer wins: kept as it was)
            bool dup = false;
            for (int a = 0; a < m && !dup; a++)
            {
                for (int c = a + 1; c < m; c++)
                {
                    if (idf[a] == idf[c])
                    {
                        dup = true;
                        break;
                    }
                }
            }
            const int fu = dup ? 0 : ((quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 1 : 2);
            hipLaunchKernelGGL(pf_sample_proposal_kernel<T>, dim3((np * kPfSubLanes + 63) / 64), dim3(64), 0, stream, store(), sZ, sIdf,
                               m, R[0], R[1], R[2], R[3], sNrm, pr, fu);
            CSLAM_HIP_TRY(hipGetLastError());
            if (fu == 0)
            {
                hipLaunchKernelGGL(pf_feature_update_kernel<T>, dim3((np + 63) / 64, m), dim3(64), 0, stream, store(), sZ,
                                   sIdf, m, R[0], R[1], R[2], R[3], (quirks & CSLAM_Q_LOWER_CHOL_GAIN) ? 0 : 1);
                CSLAM_HIP_TRY(hipGetLastError());
            }
        }
        else
        {
            hipLaunchKernelGGL(pf_predict_kernel<T>, dim3((np + 63) / 64), dim3(64), 0, stream, store(), (T)v, (T)swa, Q[0],
                               Q[1], Q[2], Q[3], (T)wb, (T)dt);
            CSLAM_HIP_TRY(hipGetLastError());
        }
        return launch_resample(reinterpret_cast<const T*>(base + off_sel), n_eff, status);
    }

    int resample_stats(double* calls, double* resamples, double* last_neff) override
    {
        int rc = use_device();
        if (rc || (rc = ensure_resample_buffers()))
        {
            return rc;
        }
        CSLAM_HIP_TRY(hipMemcpyAsync(hInfo, dInfo, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        stage_inflight = 0;
        if (last_neff)
        {
            *last_neff = hInfo[0];
        }
        if (calls)
        {
            *calls = hInfo[2];
        }
        if (resamples)
        {
            *resamples = hInfo[3];
        }
        return CSLAM_OK;
    }

    int get_particle(int i, void* w, void* Xv, void* Pv, void* XF, void* PF) override
    {
        if (i < 0 || i >= np)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_get_particle: index %d", i);
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const size_t s = sizeof(T), pitch = (size_t)np * s;
        if (w)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(w, dW + i, s, hipMemcpyDeviceToHost, stream));
        }
        if (Xv)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(Xv, s, dXv + i, pitch, s, 3, hipMemcpyDeviceToHost, stream));
        }
        if (Pv)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(Pv, s, dPv + i, pitch, s, 9, hipMemcpyDeviceToHost, stream));
        }
        if (XF && nf > 0)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(XF, s, dXF + i, pitch, s, (size_t)2 * nf, hipMemcpyDeviceToHost, stream));
        }
        if (PF && nf > 0)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(PF, s, dPF + i, pitch, s, (size_t)4 * nf, hipMemcpyDeviceToHost, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int set_particle(int i, const void* w, const void* Xv, const void* Pv, const void* XF, const void* PF,
                     int nfeat) override
    {
        if (i < 0 || i >= np || nfeat < 0 || nfeat > nfcap)
        {
            return fail(CSLAM_ERR_BAD_ARG, "pf_set_particle: index %d / nf %d", i, nfeat);
        }
        int rc = use_device();
        if (rc)
        {
            return rc;
        }
        const size_t s = sizeof(T), pitch = (size_t)np * s;
        if (w)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(dW + i, w, s, hipMemcpyHostToDevice, stream));
        }
        if (Xv)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(dXv + i, pitch, Xv, s, s, 3, hipMemcpyHostToDevice, stream));
        }
        if (Pv)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(dPv + i, pitch, Pv, s, s, 9, hipMemcpyHostToDevice, stream));
        }
        if (XF && nfeat > 0)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(dXF + i, pitch, XF, s, s, (size_t)2 * nfeat, hipMemcpyHostToDevice, stream));
        }
        if (PF && nfeat > 0)
        {
            CSLAM_HIP_TRY(hipMemcpy2DAsync(dPF + i, pitch, PF, s, s, (size_t)4 * nfeat, hipMemcpyHostToDevice, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        nf = nfeat; // every particle of the store carries the same number of features
        return CSLAM_OK;
    }
};

inline PfBase* B(cslam_pf_t h)
{
    return reinterpret_cast<PfBase*>(h);
}

} // namespace

extern "C" {

#define CSLAM_NEED(h)                                                \
    if (!(h))                                                        \
    {                                                                \
        return fail(CSLAM_ERR_BAD_ARG, "%s: null handle", __func__); \
    }

int cslam_pf_create(int n_particles, int max_features, int dtype, int device, int quirks, cslam_pf_t* out)
{
    if (!out || n_particles < 1 || max_features < 0 || (dtype != CSLAM_F32 && dtype != CSLAM_F64) ||
        (quirks & ~CSLAM_Q_REF_EXACT))
    {
        return fail(CSLAM_ERR_BAD_ARG, "pf_create: bad arguments");
    }
    *out  = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0)
    {
        return fail(CSLAM_ERR_NO_DEVICE, "pf_create: no HIP device (this engine has no CPU fallback)");
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
    {
        device = 0;
    }
    if (device >= c)
    {
        return fail(CSLAM_ERR_BAD_ARG, "pf_create: device %d of %d", device, c);
    }
    PfBase* b = (dtype == CSLAM_F32) ? static_cast<PfBase*>(new (std::nothrow) Pf<float>())
                                     : static_cast<PfBase*>(new (std::nothrow) Pf<double>());
    if (!b)
    {
        return fail(CSLAM_ERR_ALLOC, "pf_create: out of host memory");
    }
    b->dtype  = dtype;
    b->device = device;
    b->quirks = quirks;
    b->np     = n_particles;
    b->nfcap  = max_features;
    int rc    = b->init();
    if (rc)
    {
        delete b;
        return rc;
    }
    *out = reinterpret_cast<cslam_pf_t>(b);
    return CSLAM_OK;
}

int cslam_pf_destroy(cslam_pf_t h)
{
    if (!h)
    {
        return CSLAM_OK;
    }
    (void)hipSetDevice(B(h)->device);
    delete B(h);
    return CSLAM_OK;
}

int cslam_pf_synchronize(cslam_pf_t h)
{
    CSLAM_NEED(h);
    CSLAM_HIP_TRY(hipSetDevice(B(h)->device));
    CSLAM_HIP_TRY(hipStreamSynchronize(B(h)->stream));
    return CSLAM_OK;
}

int cslam_pf_get_counts(cslam_pf_t h, int* n_particles, int* n_features)
{
    CSLAM_NEED(h);
    if (n_particles)
    {
        *n_particles = B(h)->np;
    }
    if (n_features)
    {
        *n_features = B(h)->nf;
    }
    return CSLAM_OK;
}

int cslam_pf_set_uniform_weight(cslam_pf_t h, double w0)
{
    CSLAM_NEED(h);
    return B(h)->set_uniform_weight(w0);
}

int cslam_pf_predict(cslam_pf_t h, double v, double swa, const void* Q, double wb, double dt)
{
    CSLAM_NEED(h);
    return B(h)->predict(v, swa, Q, wb, dt);
}

int cslam_pf_observe_heading(cslam_pf_t h, double phi, int use_heading)
{
    CSLAM_NEED(h);
    return B(h)->observe_heading(phi, use_heading);
}

int cslam_pf_sample_proposal(cslam_pf_t h, const void* Z, int m, const int* idf, const void* R, const void* normals)
{
    CSLAM_NEED(h);
    return B(h)->sample_proposal(Z, m, idf, R, normals);
}

int cslam_pf_feature_update(cslam_pf_t h, const void* Z, int m, const int* idf, const void* R)
{
    CSLAM_NEED(h);
    return B(h)->feature_update(Z, m, idf, R);
}

int cslam_pf_add_features(cslam_pf_t h, const void* Z, int q, const void* R)
{
    CSLAM_NEED(h);
    return B(h)->add_features(Z, q, R);
}

int cslam_pf_weight_sums(cslam_pf_t h, double* sums)
{
    CSLAM_NEED(h);
    return B(h)->weight_sums(sums);
}

int cslam_pf_scale_weights(cslam_pf_t h, double scale)
{
    CSLAM_NEED(h);
    return B(h)->scale_weights(scale);
}

int cslam_pf_weights_device_ptr(cslam_pf_t h, void** dptr)
{
    CSLAM_NEED(h);
    return B(h)->weights_ptr(dptr);
}

int cslam_pf_get_weights(cslam_pf_t h, void* w_host)
{
    CSLAM_NEED(h);
    return B(h)->get_weights(w_host);
}

int cslam_pf_set_weights(cslam_pf_t h, const void* w_host)
{
    CSLAM_NEED(h);
    return B(h)->set_weights(w_host);
}

int cslam_pf_record_bytes(cslam_pf_t h, long long* bytes)
{
    CSLAM_NEED(h);
    return B(h)->record_bytes(bytes);
}

int cslam_pf_pack(cslam_pf_t h, const int* src_idx, int count, void* d_records)
{
    CSLAM_NEED(h);
    return B(h)->pack(src_idx, count, d_records);
}

int cslam_pf_unpack(cslam_pf_t h, const int* dst_idx, int count, const void* d_records)
{
    CSLAM_NEED(h);
    return B(h)->unpack(dst_idx, count, d_records);
}

int cslam_pf_resample_local(cslam_pf_t h, const void* select, double n_effective, int resample_status, double* neff,
                            int* resampled)
{
    CSLAM_NEED(h);
    return B(h)->resample_local(select, n_effective, resample_status, neff, resampled);
}

int cslam_pf_gather_local(cslam_pf_t h, const int* keep, double w_new)
{
    CSLAM_NEED(h);
    if (!keep)
    {
        return fail(CSLAM_ERR_BAD_ARG, "pf_gather_local: null");
    }
    return B(h)->gather_local(keep, w_new);
}

int cslam_pf_get_particle(cslam_pf_t h, int index, void* w, void* Xv, void* Pv, void* XF, void* PF)
{
    CSLAM_NEED(h);
    return B(h)->get_particle(index, w, Xv, Pv, XF, PF);
}

int cslam_pf_set_particle(cslam_pf_t h, int index, const void* w, const void* Xv, const void* Pv, const void* XF,
                          const void* PF, int nf)
{
    CSLAM_NEED(h);
    return B(h)->set_particle(index, w, Xv, Pv, XF, PF, nf);
}

int cslam_pf_observation_step(cslam_pf_t h, double v, double swa, const void* Q, double wb, double dt, const void* Z, int m,
                              const int* idf, const void* R, const void* normals, const void* select, double n_effective,
                              int resample_status)
{
    CSLAM_NEED(h);
    return B(h)->observation_step(v, swa, Q, wb, dt, Z, m, idf, R, normals, select, n_effective, resample_status);
}

int cslam_pf_resample_stats(cslam_pf_t h, double* calls, double* resamples, double* last_neff)
{
    CSLAM_NEED(h);
    return B(h)->resample_stats(calls, resamples, last_neff);
}

int cslam_comm_unique_id(void* id_bytes)
{
    if (!id_bytes)
    {
        return fail(CSLAM_ERR_BAD_ARG, "comm_unique_id: null");
    }
    Rccl* R = rccl();
    if (!R)
    {
        return fail(CSLAM_ERR_HIP, "comm_unique_id: librccl could not be loaded");
    }
    static_assert(sizeof(ncclUniqueId) == CSLAM_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    CSLAM_RCCL_TRY(R->GetUniqueId(&id));
    std::memcpy(id_bytes, &id, sizeof(id));
    return CSLAM_OK;
}

int cslam_comm_create(const void* id_bytes, int rank, int world, int device, cslam_comm_t* out)
{
    if (!id_bytes || !out || world < 1 || rank < 0 || rank >= world)
    {
        return fail(CSLAM_ERR_BAD_ARG, "comm_create: bad arguments");
    }
    *out    = nullptr;
    Rccl* R = rccl();
    if (!R)
    {
        return fail(CSLAM_ERR_HIP, "comm_create: librccl could not be loaded");
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
    {
        device = 0;
    }
    CSLAM_HIP_TRY(hipSetDevice(device));
    Comm* c = new (std::nothrow) Comm();
    if (!c)
    {
        return fail(CSLAM_ERR_ALLOC, "comm_create: out of host memory");
    }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = R->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess)
    {
        delete c;
        return fail(CSLAM_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "rccl error");
    }
    c->rank   = rank;
    c->world  = world;
    c->device = device;
    *out      = reinterpret_cast<cslam_comm_t>(c);
    return CSLAM_OK;
}

int cslam_comm_create_loopback(int world, int device, cslam_comm_t* out)
{
    if (!out || world < 1 || world > kLoopMaxWorld)
    {
        return fail(CSLAM_ERR_BAD_ARG, "comm_create_loopback: world must be 1..%d", kLoopMaxWorld);
    }
    for (int r = 0; r < world; r++)
    {
        out[r] = nullptr;
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
    {
        device = 0;
    }
    LoopShared* sh = new (std::nothrow) LoopShared();
    if (!sh)
    {
        return fail(CSLAM_ERR_ALLOC, "comm_create_loopback: out of host memory");
    }
    sh->world = world;
    sh->refs  = world;
    sh->src.assign((size_t)world, nullptr);
    sh->sends.assign((size_t)world * world, LoopShared::P2P{});
    for (int r = 0; r < world; r++)
    {
        Comm* c = new (std::nothrow) Comm();
        if (!c)
        {
            for (int q = 0; q < r; q++)
            {
                delete reinterpret_cast<Comm*>(out[q]);
                out[q] = nullptr;
            }
            delete sh;
            return fail(CSLAM_ERR_ALLOC, "comm_create_loopback: out of host memory");
        }
        c->loop   = sh;
        c->rank   = r;
        c->world  = world;
        c->device = device;
        out[r]    = reinterpret_cast<cslam_comm_t>(c);
    }
    return CSLAM_OK;
}

int cslam_comm_destroy(cslam_comm_t c)
{
    if (!c)
    {
        return CSLAM_OK;
    }
    Comm* cc = reinterpret_cast<Comm*>(c);
    if (cc->loop)
    {
        bool last = false;
        {
            std::lock_guard<std::mutex> lk(cc->loop->mu);
            last = (--cc->loop->refs == 0);
        }
        if (last)
        {
            delete cc->loop;
        }
    }
    else if (Rccl* R = rccl())
    {
        (void)R->CommDestroy(cc->comm);
    }
    delete cc;
    return CSLAM_OK;
}

int cslam_comm_info(cslam_comm_t c, int* rank, int* world)
{
    if (!c)
    {
        return fail(CSLAM_ERR_BAD_ARG, "comm_info: null communicator");
    }
    const Comm* cc = reinterpret_cast<const Comm*>(c);
    if (rank)
    {
        *rank = cc->rank;
    }
    if (world)
    {
        *world = cc->world;
    }
    return CSLAM_OK;
}

int cslam_pf_debug_last_exchange(cslam_pf_t h, int* counts, int* send_idx, int capacity, int* n_send)
{
    CSLAM_NEED(h);
    return B(h)->debug_last_exchange(counts, send_idx, capacity, n_send);
}

int cslam_pf_resample_sharded(cslam_pf_t h, cslam_comm_t comm, const void* select, double n_effective,
                              int resample_status, double* neff, int* resampled)
{
    CSLAM_NEED(h);
    return B(h)->resample_sharded(reinterpret_cast<Comm*>(comm), select, n_effective, resample_status, neff, resampled);
}

} // extern "C"
