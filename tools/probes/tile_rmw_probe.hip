// Memory-pattern probe for the P-GEMM (tools/, not part of the library): how fast can the chip read-modify-write the
// block-lower covariance (n = 10003 -> 79 x 79 tiles of 128 x 128 f32, lower triangle = 3160 tiles = 207 MB) when
// no arithmetic is in the way, for different layouts / launch shapes?
//   linear      : float4 grid-stride over 207 MB
//   colmajor    : the P-GEMM's access pattern (16 x 16 B per lane, 512 B runs strided by ldp*4 B), persistent
//   tilemajor   : each tile contiguous (64 KB), same per-lane work
// build: hipcc --offload-arch=gfx950 -O3 -o tile_rmw_probe tile_rmw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) linear_rmw(f32x4* p, size_t n4)
{
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    {
        f32x4 v = p[i];
        v -= 1.0f;
        p[i] = v;
    }
}

template <bool NT, bool TILEMAJOR, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) tiled_rmw(float* P, int ldp, const int2* tiles, int ntiles)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lj = lane & 31, lh = lane >> 5;
    constexpr int ROWS_PER_WAVE = 128 / WAVES; // columns of the tile handled by one wave
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x)
    {
        const int2 c = tiles[t];
        f32x4 pv[ROWS_PER_WAVE / 2];
        float* base;
        size_t stride;
        if (TILEMAJOR)
        {
            base   = P + (size_t)t * 128 * 128 + (size_t)(wave * ROWS_PER_WAVE) * 128 + lane * 4;
            stride = 256; // 64 lanes x 4 floats = 2 columns of the tile per instruction
        }
        else
        {
            base   = P + (size_t)(c.y * 128 + wave * ROWS_PER_WAVE + lh) * ldp + c.x * 128 + 4 * lj;
            stride = (size_t)2 * ldp;
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE / 2; r++)
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(base + r * stride);
            pv[r] = NT ? __builtin_nontemporal_load(src) : *src;
        }
#pragma unroll
        for (int r = 0; r < ROWS_PER_WAVE / 2; r++)
        {
            pv[r] -= 1.0f;
            f32x4* dst = reinterpret_cast<f32x4*>(base + r * stride);
            if (NT) __builtin_nontemporal_store(pv[r], dst); else *dst = pv[r];
        }
    }
}

int main()
{
    const int T = 79, ldp = T * 128;
    std::vector<int2> h;
    for (int tj = 0; tj < T; tj++) for (int ti = tj; ti < T; ti++) h.push_back(make_int2(ti, tj));
    const int ntiles = (int)h.size();
    float* P; int2* dT;
    CK(hipMalloc(&P, (size_t)ldp * ldp * 4));
    CK(hipMemset(P, 0, (size_t)ldp * ldp * 4));
    CK(hipMalloc(&dT, h.size() * sizeof(int2)));
    CK(hipMemcpy(dT, h.data(), h.size() * sizeof(int2), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 2.0 * ntiles * 65536.0;
    auto report = [&](const char* name, float ms, int iters) {
        printf("%-34s %8.1f us  %7.0f GB/s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9);
    };
    const int iters = 50;
    float ms;
    auto run = [&](const char* name, auto&& launch) -> int {
        for (int i = 0; i < 5; i++) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        report(name, ms, iters);
        return 0;
    };
    const size_t n4 = (size_t)ntiles * 4096;
    run("linear grid 2048", [&] { linear_rmw<<<2048, 256>>>((f32x4*)P, n4); });
    run("linear grid 8192", [&] { linear_rmw<<<8192, 256>>>((f32x4*)P, n4); });
    run("colmajor nt  256thr G=512", [&] { tiled_rmw<true, false, 4><<<512, 256>>>(P, ldp, dT, ntiles); });
    run("colmajor     256thr G=512", [&] { tiled_rmw<false, false, 4><<<512, 256>>>(P, ldp, dT, ntiles); });
    run("colmajor     256thr G=1024", [&] { tiled_rmw<false, false, 4><<<1024, 256>>>(P, ldp, dT, ntiles); });
    run("colmajor     256thr G=3160", [&] { tiled_rmw<false, false, 4><<<3160, 256>>>(P, ldp, dT, ntiles); });
    run("colmajor     512thr G=512", [&] { tiled_rmw<false, false, 8><<<512, 512>>>(P, ldp, dT, ntiles); });
    run("colmajor     512thr G=1024", [&] { tiled_rmw<false, false, 8><<<1024, 512>>>(P, ldp, dT, ntiles); });
    run("tilemajor nt 256thr G=512", [&] { tiled_rmw<true, true, 4><<<512, 256>>>(P, ldp, dT, ntiles); });
    run("tilemajor    256thr G=512", [&] { tiled_rmw<false, true, 4><<<512, 256>>>(P, ldp, dT, ntiles); });
    run("tilemajor    256thr G=1024", [&] { tiled_rmw<false, true, 4><<<1024, 256>>>(P, ldp, dT, ntiles); });
    run("tilemajor    256thr G=3160", [&] { tiled_rmw<false, true, 4><<<3160, 256>>>(P, ldp, dT, ntiles); });
    run("tilemajor    512thr G=1024", [&] { tiled_rmw<false, true, 8><<<1024, 512>>>(P, ldp, dT, ntiles); });
    return 0;
}
