// cslam_sim.hip -- device-side observation generator and known-association table.
//
// The reference's driver does three things on the host between two filter calls, each a loop over the whole map
// or the whole scan:
//   Slam::getObservations    slam.h:575-683 (visibility filter over all landmarks) + computeRangeBearing slam.h:339-368
//   Slam::addObservationNoise slam.h:168-178 (the driver's noise draw; here the N(0,1) draws are an input, as for the PF)
//   EKF::dataAssociateTable  EKF.cpp:146-233 (split the scan into known features (ZF, idf) and new ones (ZN); new tags
//                            get the next state positions)
// Here the map, the table, the scan and its split stay in HBM; the filter is fed with cslam_ekf_update_device and the
// only thing that returns to the host per step is three counters.
//
// Kernels: one workgroup of 1024 threads walks the landmarks / the scan in order and compacts with a block-wide
// exclusive scan, so that the outputs come out in ascending tag order exactly as the reference's sequential loops
// produce them (integer outputs are bit-exact against the oracle).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "cslam_common.hpp"
#include "device_math.hpp"
#include "../../include/cslam.h"

namespace cslam
{
namespace
{

constexpr int kSimThreads = 1024;

// exclusive prefix sum of one flag per thread over the workgroup; returns the thread's offset, *total = block sum
__device__ inline int block_exclusive_scan(int flag, int* s_wave, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag != 0);
    const int                in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0)
    {
        s_wave[wave] = __popcll(bal);
    }
    __syncthreads();
    int base = 0, sum = 0;
    for (int w = 0; w < kSimThreads / 64; w++)
    {
        const int c = s_wave[w];
        base += (w < wave) ? c : 0;
        sum += c;
    }
    __syncthreads();
    *total = sum;
    return base + in_wave;
}

// slam.h:575-683 getVisibleLandmarks + slam.h:339-368 computeRangeBearing.  The visibility test is evaluated in
// double on the float differences, as the reference does (slam.h:627-628 stores float subtractions in doubles).
template <typename T>
__global__ void __launch_bounds__(kSimThreads) sim_get_observations_kernel(const T* __restrict__ LM, int nlm, T x, T y,
                                                                             T phi, T rmax, T* __restrict__ Z,
                                                                             int* __restrict__ tags, int* __restrict__ count)
{
    __shared__ int s_wave[kSimThreads / 64];
    const double   cphi = cos((double)phi), sphi = sin((double)phi), rm = (double)rmax;
    int            done = 0;
    for (int base = 0; base < nlm; base += kSimThreads)
    {
        const int i   = base + threadIdx.x;
        bool      vis = false;
        T         fx = (T)0, fy = (T)0;
        if (i < nlm)
        {
            fx = LM[2 * i] - x;
            fy = LM[2 * i + 1] - y;
            const double dx = (double)fx, dy = (double)fy;
            vis = (fabs(dx) < rm && fabs(dy) < rm) && ((dx * cphi + dy * sphi) > 0.0) && ((dx * dx + dy * dy) < rm * rm);
        }
        int       total;
        const int off = block_exclusive_scan(vis ? 1 : 0, s_wave, &total);
        if (vis)
        {
            const int o  = done + off;
            Z[2 * o]     = dsqrt(fx * fx + fy * fy);
            Z[2 * o + 1] = datan2(fy, fx) - phi;
            tags[o]      = i + 1;
        }
        done += total;
    }
    if (threadIdx.x == 0)
    {
        *count = done;
    }
}

// slam.h:168-178: z += N(0,1) * sqrt(R_ii), the draws are an input (2 per observation, range then bearing)
template <typename T>
__global__ void sim_add_noise_kernel(T* __restrict__ Z, const T* __restrict__ normals, int m, T s0, T s1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m)
    {
        Z[2 * i]     = Z[2 * i] + normals[2 * i] * s0;
        Z[2 * i + 1] = Z[2 * i + 1] + normals[2 * i + 1] * s1;
    }
}

// EKF.cpp:146-233.  The scan is split by the table as it stands before this scan (EKF.cpp:169-182); then the new
// tags receive the state positions nf+1, nf+2, ... in scan order (EKF.cpp:213-226).
template <typename T>
__global__ void __launch_bounds__(kSimThreads) sim_associate_table_kernel(const T* __restrict__ Z, const int* __restrict__ tags,
                                                                            const int* __restrict__ count, int* __restrict__ table,
                                                                            int nf, T* __restrict__ ZF, int* __restrict__ idf,
                                                                            T* __restrict__ ZN, int* __restrict__ out_counts)
{
    __shared__ int s_wave[kSimThreads / 64];
    const int      m = *count;
    int            nknown = 0, nnew = 0;
    for (int base = 0; base < m; base += kSimThreads)
    {
        const int i  = base + threadIdx.x;
        const int id = (i < m) ? tags[i] : 0;
        const int pos = (i < m) ? table[id - 1] : 0;
        const bool known = (i < m) && pos != 0, fresh = (i < m) && pos == 0;
        int        tk, tn;
        const int  ok = block_exclusive_scan(known ? 1 : 0, s_wave, &tk);
        const int  on = block_exclusive_scan(fresh ? 1 : 0, s_wave, &tn);
        if (known)
        {
            const int o = nknown + ok;
            ZF[2 * o]     = Z[2 * i];
            ZF[2 * o + 1] = Z[2 * i + 1];
            idf[o]        = pos;
        }
        if (fresh)
        {
            const int o = nnew + on;
            ZN[2 * o]     = Z[2 * i];
            ZN[2 * o + 1] = Z[2 * i + 1];
            table[id - 1] = nf + o + 1; // (read above by the same thread only: tags within a scan are distinct)
        }
        nknown += tk;
        nnew += tn;
    }
    if (threadIdx.x == 0)
    {
        out_counts[0] = nknown;
        out_counts[1] = nnew;
    }
}

struct SimBase
{
    int         dtype = CSLAM_F32, device = 0, nlm = 0;
    hipStream_t stream = nullptr;
    virtual ~SimBase() {}
    virtual int init(const void* LM)                                                                 = 0;
    virtual int get_observations(const void* xv, double rmax, void* Zh, int* tagsh, int* m)         = 0;
    virtual int add_noise(const void* R, const void* normals)                                       = 0;
    virtual int associate(int nf, void* ZFh, int* idfh, int* mf, void* ZNh, int* mn)                = 0;
    virtual int ptrs(const void** dZF, const int** dIdf, const void** dZN, const void** dZ, const int** dTags) = 0;
    virtual int get_table(int* t)                                                                    = 0;
    virtual int set_table(const int* t)                                                              = 0;
};

template <typename T>
struct Sim : SimBase
{
    T*   dLM = nullptr;
    T*   dZ = nullptr;
    T*   dZF = nullptr;
    T*   dZN = nullptr;
    T*   dNorm = nullptr;
    int* dTags = nullptr;
    int* dIdf = nullptr;
    int* dTable = nullptr;
    int* dCount = nullptr; // [0] scan size, [1] known, [2] new
    int  m_last = 0;

    ~Sim() override
    {
        (void)hipSetDevice(device);
        (void)hipFree(dLM);
        (void)hipFree(dZ);
        (void)hipFree(dZF);
        (void)hipFree(dZN);
        (void)hipFree(dNorm);
        (void)hipFree(dTags);
        (void)hipFree(dIdf);
        (void)hipFree(dTable);
        (void)hipFree(dCount);
        if (stream)
        {
            (void)hipStreamDestroy(stream);
        }
    }

    int init(const void* LM) override
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        CSLAM_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        const size_t cap = (size_t)std::max(nlm, 1);
        CSLAM_HIP_TRY(hipMalloc(&dLM, 2 * cap * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dZ, 2 * cap * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dZF, 2 * cap * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dZN, 2 * cap * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dNorm, 2 * cap * sizeof(T)));
        CSLAM_HIP_TRY(hipMalloc(&dTags, cap * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dIdf, cap * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dTable, cap * sizeof(int)));
        CSLAM_HIP_TRY(hipMalloc(&dCount, 4 * sizeof(int)));
        CSLAM_HIP_TRY(hipMemsetAsync(dTable, 0, cap * sizeof(int), stream));
        CSLAM_HIP_TRY(hipMemsetAsync(dCount, 0, 4 * sizeof(int), stream));
        if (nlm > 0)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(dLM, LM, 2 * (size_t)nlm * sizeof(T), hipMemcpyHostToDevice, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int get_observations(const void* xvv, double rmax, void* Zh, int* tagsh, int* m) override
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        const T* xv = static_cast<const T*>(xvv);
        hipLaunchKernelGGL(sim_get_observations_kernel<T>, dim3(1), dim3(kSimThreads), 0, stream, dLM, nlm, xv[0], xv[1], xv[2],
                           (T)rmax, dZ, dTags, dCount);
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipMemcpyAsync(&m_last, dCount, sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        if (m)
        {
            *m = m_last;
        }
        if (m_last > 0 && Zh)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(Zh, dZ, 2 * (size_t)m_last * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (m_last > 0 && tagsh)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(tagsh, dTags, (size_t)m_last * sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int add_noise(const void* Rv, const void* normals) override
    {
        if (m_last <= 0)
        {
            return CSLAM_OK;
        }
        CSLAM_HIP_TRY(hipSetDevice(device));
        const T* R = static_cast<const T*>(Rv);
        CSLAM_HIP_TRY(hipMemcpyAsync(dNorm, normals, 2 * (size_t)m_last * sizeof(T), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(sim_add_noise_kernel<T>, dim3((m_last + 255) / 256), dim3(256), 0, stream, dZ, dNorm, m_last,
                           (T)std::sqrt(R[0]), (T)std::sqrt(R[3]));
        CSLAM_HIP_TRY(hipGetLastError());
        CSLAM_HIP_TRY(hipStreamSynchronize(stream)); // (the pageable host buffer may be reused by the caller)
        return CSLAM_OK;
    }

    int associate(int nf, void* ZFh, int* idfh, int* mf, void* ZNh, int* mn) override
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        hipLaunchKernelGGL(sim_associate_table_kernel<T>, dim3(1), dim3(kSimThreads), 0, stream, dZ, dTags, dCount, dTable, nf,
                           dZF, dIdf, dZN, dCount + 1);
        CSLAM_HIP_TRY(hipGetLastError());
        int c[3] = {0, 0, 0};
        CSLAM_HIP_TRY(hipMemcpyAsync(c, dCount, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        if (mf)
        {
            *mf = c[1];
        }
        if (mn)
        {
            *mn = c[2];
        }
        if (c[1] > 0 && ZFh)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(ZFh, dZF, 2 * (size_t)c[1] * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        if (c[1] > 0 && idfh)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(idfh, dIdf, (size_t)c[1] * sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        if (c[2] > 0 && ZNh)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(ZNh, dZN, 2 * (size_t)c[2] * sizeof(T), hipMemcpyDeviceToHost, stream));
        }
        CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        return CSLAM_OK;
    }

    int ptrs(const void** pZF, const int** pIdf, const void** pZN, const void** pZ, const int** pTags) override
    {
        if (pZF) *pZF = dZF;
        if (pIdf) *pIdf = dIdf;
        if (pZN) *pZN = dZN;
        if (pZ) *pZ = dZ;
        if (pTags) *pTags = dTags;
        return CSLAM_OK;
    }

    int get_table(int* t) override
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        if (nlm > 0)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(t, dTable, (size_t)nlm * sizeof(int), hipMemcpyDeviceToHost, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        }
        return CSLAM_OK;
    }
    int set_table(const int* t) override
    {
        CSLAM_HIP_TRY(hipSetDevice(device));
        if (nlm > 0)
        {
            CSLAM_HIP_TRY(hipMemcpyAsync(dTable, t, (size_t)nlm * sizeof(int), hipMemcpyHostToDevice, stream));
            CSLAM_HIP_TRY(hipStreamSynchronize(stream));
        }
        return CSLAM_OK;
    }
};

inline SimBase* S(cslam_sim_t h)
{
    return reinterpret_cast<SimBase*>(h);
}

} // namespace
} // namespace cslam

using namespace cslam;

#define CSLAM_NEED_SIM(h)                                            \
    if (!(h))                                                        \
    {                                                                \
        return fail(CSLAM_ERR_BAD_ARG, "%s: null handle", __func__); \
    }

extern "C" {

int cslam_sim_create(const void* LM, int n_landmarks, int dtype, int device, cslam_sim_t* out)
{
    if (!out || n_landmarks < 0 || (n_landmarks > 0 && !LM) || (dtype != CSLAM_F32 && dtype != CSLAM_F64))
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_create: bad arguments");
    }
    *out  = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c == 0)
    {
        return fail(CSLAM_ERR_NO_DEVICE, "sim_create: no HIP device (this engine has no CPU fallback)");
    }
    if (device < 0 && hipGetDevice(&device) != hipSuccess)
    {
        device = 0;
    }
    if (device >= c)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_create: device %d of %d", device, c);
    }
    SimBase* b = (dtype == CSLAM_F32) ? static_cast<SimBase*>(new (std::nothrow) Sim<float>())
                                      : static_cast<SimBase*>(new (std::nothrow) Sim<double>());
    if (!b)
    {
        return fail(CSLAM_ERR_ALLOC, "sim_create: out of host memory");
    }
    b->dtype  = dtype;
    b->device = device;
    b->nlm    = n_landmarks;
    int rc    = b->init(LM);
    if (rc)
    {
        delete b;
        return rc;
    }
    *out = reinterpret_cast<cslam_sim_t>(b);
    return CSLAM_OK;
}

int cslam_sim_destroy(cslam_sim_t h)
{
    if (h)
    {
        delete S(h);
    }
    return CSLAM_OK;
}

int cslam_sim_get_observations(cslam_sim_t h, const void* xv_true, double rmax, void* Z, int* tags, int* m)
{
    CSLAM_NEED_SIM(h);
    if (!xv_true)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_get_observations: null pose");
    }
    return S(h)->get_observations(xv_true, rmax, Z, tags, m);
}

int cslam_sim_add_observation_noise(cslam_sim_t h, const void* R, const void* normals)
{
    CSLAM_NEED_SIM(h);
    if (!R || !normals)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_add_observation_noise: null argument");
    }
    return S(h)->add_noise(R, normals);
}

int cslam_sim_associate_table(cslam_sim_t h, int n_features, void* ZF, int* idf, int* mf, void* ZN, int* mn)
{
    CSLAM_NEED_SIM(h);
    if (n_features < 0)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_associate_table: negative feature count");
    }
    return S(h)->associate(n_features, ZF, idf, mf, ZN, mn);
}

int cslam_sim_device_ptrs(cslam_sim_t h, const void** dZF, const int** dIdf, const void** dZN, const void** dZ,
                          const int** dTags)
{
    CSLAM_NEED_SIM(h);
    return S(h)->ptrs(dZF, dIdf, dZN, dZ, dTags);
}

int cslam_sim_get_table(cslam_sim_t h, int* table)
{
    CSLAM_NEED_SIM(h);
    if (!table)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_get_table: null");
    }
    return S(h)->get_table(table);
}

int cslam_sim_set_table(cslam_sim_t h, const int* table)
{
    CSLAM_NEED_SIM(h);
    if (!table)
    {
        return fail(CSLAM_ERR_BAD_ARG, "sim_set_table: null");
    }
    return S(h)->set_table(table);
}

} // extern "C"
