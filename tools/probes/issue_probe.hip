// Issue-rate probe (tools/, not part of the library): cycles per {2 x v_readlane, s_nop 1, v_pk_fma_f32} group -- the
// inner step of the register-resident Cholesky -- when the code sits in the instruction cache (loop) and when it
// is straight-line code larger than the cache (the factor kernel's situation).
// build: hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GROUP(acc)                                                                                     \
    asm volatile("v_readlane_b32 s20, %1, 10\n v_readlane_b32 s21, %1, 11\n s_nop 1\n"                \
                 "v_pk_fma_f32 %0, %2, s[20:21], %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" \
                 : "+v"(acc) : "v"(col), "v"(colp) : "s20", "s21");
#define GROUPS(acc, SA, SB, SP)                                                                          \
    asm volatile("v_readlane_b32 " SA ", %1, 10\n v_readlane_b32 " SB ", %1, 11\n"                      \
                 : : "v"(acc), "v"(col) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
#define FMAS(acc, SP)                                                                                   \
    asm volatile("v_pk_fma_f32 %0, %1, " SP ", %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"       \
                 : "+v"(acc) : "v"(colp) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35");
// batched: 8 pairs of readlanes into 8 different SGPR pairs, then the 8 FMAs
#define B8 GROUPS(a0, "s20", "s21", "") GROUPS(a1, "s22", "s23", "") GROUPS(a2, "s24", "s25", "") GROUPS(a3, "s26", "s27", "") \
           GROUPS(a4, "s28", "s29", "") GROUPS(a5, "s30", "s31", "") GROUPS(a6, "s32", "s33", "") GROUPS(a7, "s34", "s35", "") \
           FMAS(a0, "s[20:21]") FMAS(a1, "s[22:23]") FMAS(a2, "s[24:25]") FMAS(a3, "s[26:27]")                                 \
           FMAS(a4, "s[28:29]") FMAS(a5, "s[30:31]") FMAS(a6, "s[32:33]") FMAS(a7, "s[34:35]")
#define B64 B8 B8 B8 B8 B8 B8 B8 B8
#define G8 GROUP(a0) GROUP(a1) GROUP(a2) GROUP(a3) GROUP(a4) GROUP(a5) GROUP(a6) GROUP(a7)
#define G64 G8 G8 G8 G8 G8 G8 G8 G8
#define G512 G64 G64 G64 G64 G64 G64 G64 G64

__global__ void looped(float* out, long long* cyc, int iters)
{
    float col = threadIdx.x * 0.001f;
    f32x2 colp = {col, col};
    f32x2 a0 = {1, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { G64 }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x;
}
__global__ void batched(float* out, long long* cyc, int iters)
{
    float col = threadIdx.x * 0.001f;
    f32x2 colp = {col, col};
    f32x2 a0 = {1, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) { B64 }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x;
}
// LDS broadcast: one ds_read_b128 from a uniform address feeds 2 pk_fma (4 multipliers)
__global__ void ldsb(float* out, long long* cyc, int iters)
{
    __shared__ float4 lds[64];
    lds[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    float col = threadIdx.x * 0.001f;
    f32x2 colp = {col, col};
    f32x2 a0 = {1, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            const float4 m = lds[(u + i) & 63];
            a0 -= colp * f32x2{m.x, m.y};
            a1 -= colp * f32x2{m.z, m.w};
            asm volatile("" : "+v"(a0), "+v"(a1));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = a0.x + a1.x;
}
__global__ void straight(float* out, long long* cyc)
{
    float col = threadIdx.x * 0.001f;
    f32x2 colp = {col, col};
    f32x2 a0 = {1, 1}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    long long t0 = __builtin_readcyclecounter();
    G512 G512 G512 G512 G512 G512 G512 G512   // 4096 groups = 16384 instructions (~100 KB)
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x;
}
int main()
{
    float* out; long long* cyc; long long h;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
    for (int rep = 0; rep < 3; rep++)
    {
        looped<<<1, 64>>>(out, cyc, 64); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("looped   (64 groups x 64 iterations): %.2f cycles/group\n", (double)h / 4096);
        batched<<<1, 64>>>(out, cyc, 64); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("batched  (8 SGPR pairs in flight)   : %.2f cycles/group\n", (double)h / 4096);
        ldsb<<<1, 64>>>(out, cyc, 256); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("lds b128 broadcast (4 multipliers)  : %.2f cycles per 2 pk_fma (= 2 groups)\n", (double)h / 4096);
        straight<<<1, 64>>>(out, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("straight (4096 groups, ~100 KB code): %.2f cycles/group\n", (double)h / 4096);
    }
    return 0;
}
