// MFMA timing probe (tools/, not part of the library): v_mfma_f32_32x32x2_f32 issue interval, and the cost of the
// dependent "read one accumulator register -> v_readlane -> v_rsq -> scale -> next MFMA" chain of the factor kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters)
{
    f32x16 A = {0}, B = {0}, C = {0};
    for (int r = 0; r < 16; r++) { A[r] = 1.0f + threadIdx.x * 0.01f + r; B[r] = A[r]; C[r] = A[r]; }
    float a0 = 0.001f * threadIdx.x, a1 = 0.002f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
        if (MODE == 0) // three independent MFMAs
        {
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
            C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
        }
        else if (MODE == 1) // one MFMA + dependent chain
        {
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A[3]), 5));
            float rs = __builtin_amdgcn_rsqf(d);
            a0 = (threadIdx.x > 7) ? A[3] * rs : 0.0f;
        }
        else if (MODE == 2) // three MFMAs, chain on the first
        {
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
            C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(A[3]), 5));
            float rs = __builtin_amdgcn_rsqf(d);
            a0 = (threadIdx.x > 7) ? A[3] * rs : 0.0f;
            a1 = (threadIdx.x > 3) ? B[3] * rs : 0.0f;
        }
        else if (MODE == 3) // one MFMA, no chain (dependent accumulate)
        {
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
        }
        else if (MODE == 5) // three MFMAs + a VALU chain that does not depend on them
        {
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
            C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), 5));
            float rs = __builtin_amdgcn_rsqf(d + 1.0f);
            a0 = (threadIdx.x > 7) ? a0 * rs : 0.0f;
            a1 = (threadIdx.x > 3) ? a1 * rs : 0.0f;
        }
        else if (MODE == 6) // look-ahead: the chain reads the accumulators BEFORE this iteration's MFMAs (one step lag)
        {
            const float t0 = A[3], t1 = B[3];
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            B = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, B, 0, 0, 0);
            C = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, C, 0, 0, 0);
            float l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), 9));
            float r0 = __builtin_fmaf(-a0, l, t0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), 5));
            float rs = __builtin_amdgcn_rsqf(d);
            a0 = (threadIdx.x > 7) ? r0 * rs : 0.0f;
            a1 = (threadIdx.x > 3) ? __builtin_fmaf(-a1, l, t1) * rs : 0.0f;
        }
        else if (MODE == 7) // as 6 with one MFMA per iteration
        {
            const float t0 = A[3];
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, A, 0, 0, 0);
            float l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a0), 9));
            float r0 = __builtin_fmaf(-a0, l, t0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), 5));
            float rs = __builtin_amdgcn_rsqf(d);
            a0 = (threadIdx.x > 7) ? r0 * rs : 0.0f;
        }
        else if (MODE == 4) // 16x16x4 variant: one MFMA + chain
        {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            f32x4 D = {A[0], A[1], A[2], A[3]};
            D = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, D, 0, 0, 0);
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D[3]), 5));
            float rs = __builtin_amdgcn_rsqf(d);
            a0 = (threadIdx.x > 7) ? D[3] * rs : 0.0f;
            A[0] = D[0]; A[1] = D[1]; A[2] = D[2]; A[3] = D[3];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0; for (int r = 0; r < 16; r++) s += A[r] + B[r] + C[r];
    out[threadIdx.x] = s + a0 + a1;
}
int main()
{
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
    const int it = 1000;
    const char* names[8] = {"3 independent 32x32x2 MFMAs", "1 MFMA + pivot chain", "3 MFMAs + chain on #1/#2", "1 MFMA (dependent accumulate)", "1 16x16x4 MFMA + chain", "3 MFMAs + independent VALU chain", "3 MFMAs + look-ahead chain", "1 MFMA + look-ahead chain"};
    for (int rep = 0; rep < 2; rep++)
    {
        k<0><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[0], (double)h / it);
        k<1><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[1], (double)h / it);
        k<2><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[2], (double)h / it);
        k<3><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[3], (double)h / it);
        k<4><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[4], (double)h / it);
        k<5><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[5], (double)h / it);
        k<6><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[6], (double)h / it);
        k<7><<<1, 64>>>(out, cyc, it); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %.1f cycles/iter\n", names[7], (double)h / it);
    }
    return 0;
}
