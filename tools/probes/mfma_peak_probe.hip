// MFMA peak probe (tools/, not part of the library): back-to-back independent matrix-core instructions on every SIMD of
// the chip, for the two instructions the P-GEMMs use -- v_mfma_f32_32x32x2_f32 (ekf_downdate_psym4_f32) and
// v_mfma_f64_16x16x4_f64 (ekf_downdate_f64).  SURVEY.md 8(d) asks for the datasheet peaks (f32 157.3 TF, f64 78.6 TF) to
// be confirmed on the box; bench.py prices `roofline.frac` against the datasheet figure and quotes this measurement
// beside it (profiles/r03_mfma_peak.txt).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip ; run: ./mfma_peak_probe
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float  f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_f32(float* out, int iters)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = 1.0f + 1e-3f * threadIdx.x, y = 1.0f - 1e-3f * threadIdx.x;
    for (int i = 0; i < iters; i++)
    {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; r++)
    {
        s += a0[r] + a1[r] + a2[r] + a3[r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_f64(double* out, int iters)
{
    f64x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const double x = 1.0 + 1e-3 * threadIdx.x, y = 1.0 - 1e-3 * threadIdx.x;
    for (int i = 0; i < iters; i++)
    {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    double s = 0.0;
    for (int r = 0; r < 4; r++)
    {
        s += a0[r] + a1[r] + a2[r] + a3[r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, wgs = 2 * cus, iters = 20000; // 8 waves per compute unit: 2 per SIMD
    void* buf = nullptr;
    hipMalloc(&buf, (size_t)wgs * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which < 2; which++)
    {
        double best = 0.0;
        for (int rep = 0; rep < 5; rep++)
        {
            hipEventRecord(e0, 0);
            if (which == 0)
            {
                hipLaunchKernelGGL(k_f32, dim3(wgs), dim3(256), 0, 0, (float*)buf, iters);
            }
            else
            {
                hipLaunchKernelGGL(k_f64, dim3(wgs), dim3(256), 0, 0, (double*)buf, iters);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            // flops per instruction and wave: 32x32x2 -> 2*32*32*2 = 4096; 16x16x4 -> 2*16*16*4 = 2048
            const double flops = (double)wgs * 4 * iters * 4 * (which == 0 ? 4096.0 : 2048.0);
            best               = flops / (ms * 1e-3) / 1e12 > best ? flops / (ms * 1e-3) / 1e12 : best;
        }
        printf("%s: %.1f TFLOP/s on %d compute units (%d MHz max clock), datasheet %s\n",
               which == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f64_16x16x4_f64", best, cus, p.clockRate / 1000,
               which == 0 ? "157.3" : "78.6");
    }
    return 0;
}
