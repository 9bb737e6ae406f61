#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float xhalf(float v)
{
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__global__ void k(float* o) { o[threadIdx.x] = xhalf((float)threadIdx.x); }
int main()
{
    float* d; float h[64]; hipMalloc(&d, 256); k<<<1, 64>>>(d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int ok = 1; for (int i = 0; i < 64; i++) ok &= (h[i] == (float)(i ^ 32));
    printf("xhalf %s: lane0=%g lane1=%g lane32=%g lane63=%g\n", ok ? "OK" : "WRONG", h[0], h[1], h[32], h[63]);
    return 0;
}
