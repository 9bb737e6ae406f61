// Probe: does hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM is still running?
// Kernel A: one workgroup spinning ~40 us.  Kernel B: 256 workgroups, short.  Device timestamps (wall_clock64) written
// by both show whether B started before A ended.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void spin_kernel(long long* stamps, long long ticks)
{
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[0] = t0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void short_kernel(long long* stamps, float* buf)
{
    if (threadIdx.x == 0) stamps[16 + blockIdx.x] = wall_clock64();
    buf[blockIdx.x * 256 + threadIdx.x] += 1.0f;
}
__global__ void __launch_bounds__(256) spin_big_kernel(long long* stamps, long long ticks)
{
    __shared__ float st_lds[13500];                  // 54 KB static
    extern __shared__ float dyn_lds[];               // + 52 KB dynamic
    st_lds[threadIdx.x] = 1.0f; dyn_lds[threadIdx.x] = 2.0f;
    __syncthreads();
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[0] = t0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) stamps[1] = wall_clock64() + (long long)(st_lds[5] + dyn_lds[7] - 3.0f);
}
__global__ void __launch_bounds__(256) ticket_kernel(long long* stamps, float* buf, int* tickets, int tiles)
{
    __shared__ int s_t;
    if (threadIdx.x == 0) stamps[16 + (blockIdx.x & 255)] = wall_clock64();
    for (;;)
    {
        if (threadIdx.x == 0) s_t = atomicAdd(&tickets[0], 1);
        __syncthreads();
        const int t = s_t;
        if (t >= tiles) break;
        buf[(t & 255) * 256 + threadIdx.x] += 1.0f;
        __syncthreads();
    }
    if (threadIdx.x == 0 && atomicAdd(&tickets[1], 1) == (int)gridDim.x - 1) { atomicExch(&tickets[0], 0); atomicExch(&tickets[1], 0); stamps[8] = wall_clock64(); }
}
int main()
{
    long long* st; float* buf;
    CK(hipMalloc(&st, 1024 * sizeof(long long))); CK(hipMalloc(&buf, 256 * 256 * sizeof(float)));
    CK(hipMemset(buf, 0, 256 * 256 * sizeof(float)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; mode++)
    {
        for (int rep = 0; rep < 3; rep++)
        {
            CK(hipMemsetAsync(st, 0, 1024 * sizeof(long long), s));
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, st, 4000LL); // 100 MHz clock: 40 us
            if (mode == 0) hipLaunchKernelGGL(short_kernel, dim3(256), dim3(256), 0, s, st, buf);
            else hipExtLaunchKernelGGL(short_kernel, dim3(256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, st, buf);
            CK(hipGetLastError());
            hipLaunchKernelGGL(short_kernel, dim3(1), dim3(256), 0, s, st + 512, buf); // ordered successor
            CK(hipStreamSynchronize(s));
            static long long h[1024]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
            long long mn = 1LL << 62, mx = -(1LL << 62); int early = 0;
            for (int b = 0; b < 256; b++) { long long t = h[16 + b] - h[0]; mn = t < mn ? t : mn; mx = t > mx ? t : mx; early += (h[16 + b] < h[1]); }
            printf("mode %d (%s): A 0..%lld  B blocks start min %+lld max %+lld, %d of 256 before A ended;  C start %+lld (10 ns ticks after A start)\n", mode,
                   mode ? "any-order" : "ordered", h[1] - h[0], mn, mx, early, h[512 + 16] - h[0]);
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    int* tk; CK(hipMalloc(&tk, 8)); CK(hipMemset(tk, 0, 8));
    for (int mode = 0; mode < 3; mode++)
    {
        for (int rep = 0; rep < 3; rep++)
        {
            CK(hipMemsetAsync(st, 0, 1024 * sizeof(long long), s));
            hipLaunchKernelGGL(spin_big_kernel, dim3(1), dim3(256), 52 * 1024, s, st, 2500LL); // 25 us
            if (mode == 0) hipExtLaunchKernelGGL(short_kernel, dim3(256), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, st, buf);
            else hipExtLaunchKernelGGL(ticket_kernel, dim3(mode == 1 ? 1463 : 183), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, st, buf, tk, mode == 1 ? 1280 : 160);
            CK(hipGetLastError());
            hipLaunchKernelGGL(short_kernel, dim3(1), dim3(256), 0, s, st + 512, buf);
            CK(hipStreamSynchronize(s));
            static long long h[1024]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
            long long mn = 1LL << 62, mx = -(1LL << 62); int early = 0;
            for (int b = 0; b < 256; b++) { if (!h[16 + b]) continue; long long t = h[16 + b] - h[0]; mn = t < mn ? t : mn; mx = t > mx ? t : mx; early += (h[16 + b] < h[1]); }
            printf("big-A mode %d: A 0..%lld  B blocks start min %+lld max %+lld, %d early; B all done %+lld; C start %+lld\n", mode, h[1] - h[0], mn, mx, early, h[8] ? h[8] - h[0] : -1, h[512 + 16] - h[0]);
        }
    }
    return 0;
}
