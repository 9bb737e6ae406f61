// Probe: HW_REG_XCC_ID per workgroup (which XCD a block lands on) for a 512-block launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20);
}
int main()
{
    int* d; hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d);
    int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[16] = {0};
    for (int i = 0; i < 512; i++) hist[h[i] & 15]++;
    printf("raw[0..15]:"); for (int i = 0; i < 16; i++) printf(" %x", h[i]); printf("\nhist:");
    for (int i = 0; i < 16; i++) printf(" %d", hist[i]); printf("\n");
    return 0;
}
