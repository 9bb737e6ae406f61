// Probe: which compute units does a queue CU mask select?  Streams are created with hipExtStreamCreateWithCUMask for a few
// bit patterns; a kernel of 512 workgroups records (HW_REG_XCC_ID, HW_REG_HW_ID bits 8..15 = CU / SH / SE) per workgroup;
// the host prints how many distinct units every XCD contributed.  Build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned* out)
{
    if (threadIdx.x == 0)
    {
        out[2 * blockIdx.x]     = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
        out[2 * blockIdx.x + 1] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFF;
    }
    __builtin_amdgcn_s_sleep(100);
}
static void run(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess)
    {
        printf("%s: create failed\n", name);
        return;
    }
    unsigned* d;
    hipMalloc(&d, 2 * 2048 * sizeof(unsigned));
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(2 * 2048);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> per[16];
    for (int i = 0; i < 2048; i++)
    {
        per[h[2 * i] & 15].insert(h[2 * i + 1]);
    }
    printf("%s:", name);
    for (int x = 0; x < 8; x++)
    {
        printf(" xcc%d=%zu", x, per[x].size());
    }
    printf("\n");
    hipFree(d);
    hipStreamDestroy(st);
}
int main()
{
    std::vector<uint32_t> all(8, 0xFFFFFFFFu), hi8(8, 0u), lo8(8, 0u), first32(8, 0u), stride(8, 0u);
    hi8[7]     = 0xFF000000u;  // bits 248..255
    lo8[0]     = 0x000000FFu;  // bits 0..7
    first32[0] = 0xFFFFFFFFu;  // bits 0..31
    for (int i = 0; i < 256; i += 32) stride[i / 32] |= 1u; // bits 0, 32, 64, ...
    run("all", all);
    run("bits 248..255", hi8);
    run("bits 0..7", lo8);
    run("bits 0..31", first32);
    run("bits 0,32,64,...", stride);
    return 0;
}
