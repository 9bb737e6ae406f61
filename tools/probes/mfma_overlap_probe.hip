// Does vector work issued behind an MFMA overlap its execution within one wave, and which vector instructions do?
// Each iteration: 3 x { v_mfma_f32_32x32x2_f32 ; 8 dependent "filler" instructions not using MFMA results }.
// If the fillers overlap, an iteration costs 3 x 64 = 192 cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap_probe mfma_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__device__ inline float filler(float x)
{
#pragma unroll
    for (int i = 0; i < 8; i++)
    {
        if (MODE == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        if (MODE == 1) x = __builtin_fmaf(x, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 5)), 0.5f);
        if (MODE == 2) x = __builtin_amdgcn_rsqf(x + 2.0f);
        if (MODE == 3) { unsigned u = __float_as_uint(x); auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); x = __uint_as_float(r[0]) + 1.0f; }
    }
    return x;
}
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters)
{
    f32x16 A = {0}, B = {0}, C = {0};
    for (int r = 0; r < 16; r++) { A[r] = 1.0f + threadIdx.x * 0.01f + r; B[r] = A[r]; C[r] = A[r]; }
    const float a0 = 0.001f * threadIdx.x, a1 = 0.002f;
    float x = 1.0f + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
        A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 0) x = filler<MODE < 0 ? 0 : MODE>(x);
        __builtin_amdgcn_sched_barrier(0);
        B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 0) x = filler<MODE < 0 ? 0 : MODE>(x);
        __builtin_amdgcn_sched_barrier(0);
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 0) x = filler<MODE < 0 ? 0 : MODE>(x);
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    float s = x; for (int r = 0; r < 16; r++) s += A[r] + B[r] + C[r];
    out[threadIdx.x] = s;
}
// reading one accumulator register of the tile the PREVIOUS iteration's MFMA wrote, behind this iteration's MFMAs
__global__ void kread(float* out, long long* cyc, int iters)
{
    f32x16 A = {0}, B = {0}, C = {0};
    for (int r = 0; r < 16; r++) { A[r] = 1.0f + threadIdx.x * 0.01f + r; B[r] = A[r]; C[r] = A[r]; }
    const float a0 = 0.001f * threadIdx.x, a1 = 0.002f;
    float x = 1.0f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
        A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        x += A[3]; // A's MFMA was issued 128+ cycles ago
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    float s = x; for (int r = 0; r < 16; r++) s += A[r] + B[r] + C[r];
    out[threadIdx.x] = s;
}
int main()
{
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
    const int it = 1000;
    k<-1><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 MFMAs, no filler               %.1f cycles/iter\n", (double)h / it);
    k<0><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 x (MFMA + 8 v_fma)             %.1f\n", (double)h / it);
    k<1><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 x (MFMA + 8 readlane+fma)      %.1f\n", (double)h / it);
    k<2><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 x (MFMA + 8 add+rsq)           %.1f\n", (double)h / it);
    k<3><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 x (MFMA + 8 permlane32_swap+add) %.1f\n", (double)h / it);
    kread<<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("3 MFMAs + read of A[3]           %.1f\n", (double)h / it);
    return 0;
}
