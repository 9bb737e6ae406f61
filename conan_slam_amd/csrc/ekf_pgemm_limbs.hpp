// The f32 covariance downdate P -= W*W^T (slam.h:260) on the bf16 matrix cores, with f32 results.
//
// gfx950 has no fast f32 MFMA: v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TFLOP/s), 1/16 of
// v_mfma_f32_32x32x16_bf16, and at k = 128 (the deferred flush) the f32 P-GEMM is bound by it (121 us where its HBM
// traffic needs 52).  An f32 number is EXACTLY the sum of three bf16 numbers (24 significand bits = 3 x 8):
//     w = h + m + l,   h = bf16(w), m = bf16(w - h), l = bf16(w - h - m)        (round to nearest even, exact residuals)
// and the product of two bf16 numbers is exact in f32 (16 significand bits).  So
//     a * b = sum over the nine limb pairs (a_i * b_j),
// every one of them formed exactly by the bf16 MFMA and accumulated in f32: the same f32 arithmetic as the f32 MFMA's
// fma chain up to the order of the f32 additions -- at 9/16 of its matrix-core time (NP = 9), or 6/16 without the three
// pairs below 2^-24 of the product (m*l, l*m, l*l; NP = 6).  The kernel below is ekf_downdate_psym4_f32's skeleton
// (persistent workgroups, block-lower symmetric storage, tile tickets, every P load / store issued from inside the
// MFMA loop, LDS-DMA panels) around this inner product; the limbs are produced once per flush by ekf_limb_split_kernel.
//
// Limb store (bf16), two images of the same data so that both operands are lane-linear in LDS:
//     Wb[image][limb][kg][row'][8]      kg = column / 8, 8 consecutive columns per 16-byte granule
//     image 0 ("A", the tile's COLUMN block): row' = row
//     image 1 ("B", the tile's ROW block):    row' = (row & ~127) + (row & 3) * 32 + ((row & 127) >> 2)
// (the accumulator tile's D-column j of accumulator c is memory row 4 j + c -- that is what makes a lane's four values of
// P one 16-byte access -- so the B operand of accumulator c wants rows c, 4 + c, 8 + c, ... on consecutive lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "ekf_kernels.hpp"

namespace cslam
{

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned bf16_rne_bits(float x)
{
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; // (finite inputs; the filter's panels hold no NaN that matters here)
}

// one thread per (row, kg): 8 columns of the f32 panel -> 3 limbs x 2 images.  W is column-major (ldw), k columns are
// valid, the kgs * 8 - k padding columns are written as zeros.  grid = (ceil(rows / 256), kgs), rows = n_pad (multiple of 128)
__global__ void __launch_bounds__(256) ekf_limb_split_kernel(const float* __restrict__ W, int ldw, int k, int rows, int kgs,
                                                              uint4* __restrict__ Wb)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int kg  = blockIdx.y;
    if (row >= rows)
    {
        return;
    }
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        const int q = kg * 8 + j;
        w[j]        = (q < k) ? W[(size_t)q * ldw + row] : 0.f;
    }
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
        hb[j]          = bf16_rne_bits(w[j]);
        const float r1 = w[j] - __uint_as_float(hb[j] << 16);
        mb[j]          = bf16_rne_bits(r1);
        const float r2 = r1 - __uint_as_float(mb[j] << 16);
        lb[j]          = bf16_rne_bits(r2);
    }
    auto pack = [](const unsigned* b) -> uint4 {
        return make_uint4(b[0] | (b[1] << 16), b[2] | (b[3] << 16), b[4] | (b[5] << 16), b[6] | (b[7] << 16));
    };
    const size_t image = (size_t)3 * kgs * rows;
    const int    rowp  = (row & ~127) + (row & 3) * 32 + ((row & 127) >> 2);
    const uint4  hv = pack(hb), mv = pack(mb), lv = pack(lb);
    Wb[((size_t)0 * kgs + kg) * rows + row]          = hv;
    Wb[((size_t)1 * kgs + kg) * rows + row]          = mv;
    Wb[((size_t)2 * kgs + kg) * rows + row]          = lv;
    Wb[image + ((size_t)0 * kgs + kg) * rows + rowp] = hv;
    Wb[image + ((size_t)1 * kgs + kg) * rows + rowp] = mv;
    Wb[image + ((size_t)2 * kgs + kg) * rows + rowp] = lv;
}

// VM operations a wave issues inside chunk j behind that chunk's panel DMA: the 16 stores of the previous tile's results
// (chunk 0, not for a workgroup's first tile), 8 loads of P each in chunks 1 and 2.
constexpr int limb_mops(int j, bool first) { return j == 0 ? (first ? 0 : 16) : ((j == 1 || j == 2) ? 8 : 0); }
// VM operations issued AFTER the panel DMA of chunk c and before the top of chunk c (vmcnt retires in order: "all but
// that many" = that DMA has landed).  The DMA of chunk c is issued D chunks ahead: at the top of chunk c - D of the same
// tile, of chunk NCH + c - D of the previous tile, or -- a workgroup's first tile -- in the prologue together with those
// of chunks 0 .. D-1; each chunk top issues 6 DMA instructions per wave.
constexpr int limb_vm_after(int c, bool first, int D, int NCH)
{
    int n = 0;
    if (first && c < D)
    {
        n = (D - 1 - c) * 6;
        for (int j = 0; j < c; j++)
        {
            n += 6 + limb_mops(j, true);
        }
    }
    else if (c >= D)
    {
        n = limb_mops(c - D, first);
        for (int j = c - D + 1; j < c; j++)
        {
            n += 6 + limb_mops(j, first);
        }
    }
    else
    {
        const int p = NCH + c - D; // (>= 1: the previous tile's chunk 0 is never in the window)
        n           = limb_mops(p, true);
        for (int j = p + 1; j < NCH; j++)
        {
            n += 6 + limb_mops(j, true);
        }
        for (int j = 0; j < c; j++)
        {
            n += 6 + limb_mops(j, false);
        }
    }
    return n;
}

// NTMODE: 0 ordinary accesses to P, 1 non-temporal loads and stores.  NCH: chunks of 16 columns (k8 <= 16 * NCH; >= 4).
// NP: limb pairs per product (9: all; 6: without m*l, l*m, l*l).  R: ring of R panel buffers of 24 KB in LDS, the DMA of
// a chunk is issued R - 1 chunks ahead of its use (a chunk's MFMAs last ~0.5 us, a panel fetch 1.5 - 2.5 us: with two
// buffers -- one chunk of lead -- every chunk waited for its panels and the kernel ran at a third of its matrix rate).
template <int NTMODE, int NCH, int NP, int R>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
ekf_downdate_psym5_bf16(float* __restrict__ P, int ldp, const uint4* __restrict__ Wb, int rows,
                        const int2* __restrict__ tile_list_all, const int* __restrict__ seg_off,
                        int* __restrict__ ticket_all, int* __restrict__ ticket_reset)
{
    // Tile queues per XCD.  The tile list is Morton-ordered and cut into eight equal segments (seg_off[0..8]); a workgroup
    // reads the XCD it runs on (HW_REG_XCC_ID) and draws tickets from THAT segment's counter, so the eight L2s each see
    // one compact patch of the triangle -- a few row and column panels of the limb store -- instead of all of them
    // (the limb store is 2 x 3 x n x k x 2 bytes, 15 MB at k = 128: four L2s' worth).  Placement is a speed matter only:
    // a workgroup whose own segment is exhausted goes on with whatever the other seven have left (one tile at a time),
    // so every tile is done exactly once wherever the workgroups land.
    constexpr int D = R - 1;
    static_assert(NCH >= 4 && D >= 1 && D <= NCH - 1, "chunks 0, 1, 2 carry the memory operations; the next tile is known from chunk 1 on");
    static_assert(NP == 9 || NP == 6, "limb pairs");
    static_assert(limb_vm_after(D < NCH ? D - 1 : 0, false, D, NCH) <= 63, "vmcnt is 6 bits");
    constexpr int KGS = 2 * NCH; // 8-column groups
    // ring slot: [operand 2][limb 3][kg 2][row 128] granules of 16 bytes = 24 KB
    extern __shared__ __attribute__((aligned(16))) uint4 s_ring[];
    __shared__ unsigned long long s_next; // the looked-up tile (x | y << 32), all ones = none
    __shared__ int      s_tk;
    __shared__ unsigned s_mask;

    const int tid  = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int lj   = lane & 31;
    const int lh   = lane >> 5;
    const int xcd  = __builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u));
    // the segment this workgroup is drawing from (changes in the stealing phase)
    const int2* tile_list = tile_list_all;
    int         ntiles    = 0;
    int*        ticket    = ticket_all;

    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kAuxLd = (NTMODE == 1) ? 2 : 0;
    constexpr int kAuxSt = (NTMODE == 1) ? 2 : 0;
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(P, 0, (unsigned)((size_t)ldp * ldp * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(Wb), 0, (unsigned)((size_t)2 * 3 * KGS * rows * 16), 0x00020000);
    const unsigned lane_off = (unsigned)(((wave * 32 + 4 * lh) * ldp + 4 * lj) * 4);
    auto tile_base = [&](int2 t) -> unsigned { return (unsigned)(((size_t)(t.y * 128) * ldp + t.x * 128) * 4); };
    auto row_off   = [&](int r) -> unsigned { return (unsigned)(((r & 3) + 8 * (r >> 2)) * ldp * 4); };
    auto load1     = [&](unsigned tbase, int r) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, lane_off, tbase + row_off(r), kAuxLd));
    };
    auto store1 = [&](unsigned tbase, int r, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsP, lane_off, tbase + row_off(r), kAuxSt);
    };
    // Panels of chunk cc (k-groups 2 cc, 2 cc + 1) of tile t into ring slot `sl`: 24 half-blocks of 64 granules (1 KB), six
    // per wave.  Half-block hb = 0..23: operand = hb / 12 (0: A image, rows of t.y; 1: B image, rows of t.x),
    // limb = (hb % 12) / 4, kg = ((hb % 4) >> 1), half = hb & 1.
    const unsigned dma_lane_off = (unsigned)(lane * 16);
    const unsigned image_bytes  = (unsigned)((size_t)3 * KGS * rows * 16);
    auto dma_chunk = [&](int2 t, int cc, int sl) {
#pragma unroll
        for (int it = 0; it < 6; it++)
        {
            const int      hb   = it * 4 + wave; // (wave-uniform)
            const int      op   = hb / 12, limb = (hb % 12) / 4, kgl = (hb % 4) >> 1, half = hb & 1;
            const unsigned row0 = (unsigned)((op == 0 ? t.y : t.x) * 128 + half * 64);
            const unsigned goff = (unsigned)op * image_bytes +
                                  (unsigned)((((size_t)limb * KGS + (2 * cc + kgl)) * rows + row0) * 16);
            uint4* dst = s_ring + (sl * 2 + op) * 768 + (limb * 2 + kgl) * 128 + half * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)dst, 16, dma_lane_off, goff, 0, 0);
        }
    };

    f32x16 acc0, acc1, acc2, acc3;
    using C0 = std::integral_constant<int, 0>;
    f32x4 pv[16]; // results of the previous tile -> P of the current tile -> results of the current tile
    int   slot = 0; // ring slot of the chunk being computed

    auto process = [&](auto FIRST, auto STEAL, int2 cur, unsigned cbase, unsigned prev_base, int t_next_in, int& t_next_out,
                       int2& nxt_out) -> bool {
        constexpr bool first = decltype(FIRST)::value;
        constexpr bool steal = decltype(STEAL)::value; // a lone tile: no look-ahead, no ticket request
        unsigned long long look = ~0ull; // (kept as ONE 64-bit value: an int2 made the compiler shuffle components right
                                         // behind the load, i.e. wait for it: a global round trip per tile in wave 0)
        int            tk_new = 0;
        int2           nxt    = make_int2(-1, -1);
        bool           have_next = false;
        auto chunk = [&](auto CC) {
            constexpr int c  = decltype(CC)::value;
            constexpr int nb = limb_vm_after(c, first, D, NCH);
            static_assert(nb <= 63, "vmcnt is 6 bits");
            // ---- top of chunk c: its panels have landed; everybody is done with the slot of chunk c - 1 ----
            __builtin_amdgcn_s_waitcnt((nb & 15) | 0x0F70 | ((nb >> 4) << 14));
            if (c == 1 && !first && tid == 0)
            {
                s_next = look;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (c == 0)
            {
                if (tid == 0 && !steal)
                {
                    if (!first)
                    {
                        int tk_raw = t_next_in;
                        asm volatile("" : "+v"(tk_raw));
                        const int tt = tk_raw;
                        if (tt >= 0 && tt < ntiles)
                        {
                            look = reinterpret_cast<const unsigned long long*>(tile_list)[tt];
                        }
                    }
                    const int zero = 0, one = 1;
                    // (s_nop: the pointer may have just been restored from a spill lane by v_readlane, and the hazard
                    // recogniser does not look inside an asm block: VALU-written SGPR -> VMEM address needs 5 wait states;
                    // without them the atomic went to whatever the SGPR pair held before: a memory fault at N >= 3000)
                    asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0"
                                 : "=v"(tk_new)
                                 : "v"(zero), "v"(one), "s"(ticket)
                                 : "memory");
                }
                acc0 = acc1 = acc2 = acc3 = f32x16{0};
            }
            if constexpr (c == 1)
            {
                if (first)
                {
                    nxt = (!steal && t_next_in < ntiles) ? tile_list[t_next_in] : make_int2(-1, -1);
                }
                else
                {
                    const unsigned long long sn = s_next;
                    nxt = make_int2(__builtin_amdgcn_readfirstlane((int)(unsigned)sn),
                                    __builtin_amdgcn_readfirstlane((int)(unsigned)(sn >> 32)));
                }
                have_next = nxt.x >= 0;
                asm volatile("" : "+v"(tk_new));
            }
            // the panels of the chunk D ahead, into the slot chunk c - 1 has just left (the workgroup's last tile
            // re-requests its own panels: the wait counts above stay the same for every tile)
            {
                const int tslot = (slot == 0) ? (R - 1) : (slot - 1);
                if constexpr (c + D < NCH)
                {
                    dma_chunk(cur, c + D, tslot);
                }
                else
                {
                    dma_chunk(have_next ? nxt : cur, c + D - NCH, tslot);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const uint4* sA = s_ring + slot * 1536;
            const uint4* sB = sA + 768;
            // A fragments: limb L, k-group lh, row wave*32 + lj of the tile's column block
            bf16x8 af[3];
#pragma unroll
            for (int L = 0; L < 3; L++)
            {
                af[L] = __builtin_bit_cast(bf16x8, sA[(L * 2 + lh) * 128 + wave * 32 + lj]);
            }
            auto group = [&](auto CI, f32x16& acc) {
                constexpr int ci = decltype(CI)::value;
                bf16x8        bfr[3];
#pragma unroll
                for (int L = 0; L < 3; L++)
                {
                    bfr[L] = __builtin_bit_cast(bf16x8, sB[(L * 2 + lh) * 128 + ci * 32 + lj]);
                }
                // smallest pairs first: (l,l) (m,l) (l,m) | (h,l) (l,h) (m,m) | (h,m) (m,h) | (h,h)
                if constexpr (NP == 9)
                {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[1], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], acc, 0, 0, 0);
            };
            // memory operations of this chunk in four portions, one behind each accumulator's group of MFMAs:
            // chunk 0: the 16 stores of the previous tile's results; chunk 1: loads 0-7; chunk 2: loads 8-15
            auto memops = [&](auto PART) {
                constexpr int part = decltype(PART)::value;
                if (c == 0 && !first)
                {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                    {
                        store1(prev_base, 4 * part + r, pv[4 * part + r]);
                    }
                }
                if (c == 1)
                {
                    pv[2 * part]     = load1(cbase, 2 * part);
                    pv[2 * part + 1] = load1(cbase, 2 * part + 1);
                }
                if (c == 2)
                {
                    pv[8 + 2 * part]     = load1(cbase, 8 + 2 * part);
                    pv[8 + 2 * part + 1] = load1(cbase, 8 + 2 * part + 1);
                }
            };
            group(std::integral_constant<int, 0>{}, acc0);
            memops(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 1>{}, acc1);
            memops(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 2>{}, acc2);
            memops(std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 3>{}, acc3);
            memops(std::integral_constant<int, 3>{});
            __builtin_amdgcn_sched_barrier(0);
            slot = (slot + 1 == R) ? 0 : (slot + 1);
        };
        auto chunks_from = [&](auto self, auto CC) {
            constexpr int c = decltype(CC)::value;
            if constexpr (c < NCH)
            {
                chunk(CC);
                self(self, std::integral_constant<int, c + 1>{});
            }
        };
        chunks_from(chunks_from, C0{});
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            pv[r][0] -= acc0[r];
            pv[r][1] -= acc1[r];
            pv[r][2] -= acc2[r];
            pv[r][3] -= acc3[r];
        }
        t_next_out = tk_new;
        nxt_out    = nxt;
        return have_next;
    };
    // a workgroup's first tile (and every lone tile): the panels of its first D chunks, slots 0 .. D-1
    auto prologue = [&](int2 t) {
        slot = 0;
#pragma unroll
        for (int c = 0; c < D; c++)
        {
            dma_chunk(t, c, c);
        }
    };

    if (blockIdx.x == 0 && tid < 8)
    {
        ticket_reset[tid] = 0; // the counters the NEXT launch on this stream will use
    }
    using TT = std::true_type;
    using FF = std::false_type;
    int  tk = 0;
    int2 nxt;
    // ---- own segment: two tickets at once for the first two tiles, then one per tile, requested a tile ahead
    {
        tile_list = tile_list_all + seg_off[xcd];
        ntiles    = seg_off[xcd + 1] - seg_off[xcd];
        ticket    = ticket_all + xcd;
        if (tid == 0)
        {
            s_tk = __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int tk0 = __builtin_amdgcn_readfirstlane(s_tk);
        if (tk0 < ntiles)
        {
            int2     cur   = tile_list[tk0];
            unsigned cbase = tile_base(cur);
            prologue(cur);
            bool hn = process(TT{}, FF{}, cur, cbase, 0u, tk0 + 1, tk, nxt);
            while (hn)
            {
                const unsigned rbase = cbase;
                cur                  = nxt;
                cbase                = tile_base(cur);
                hn                   = process(FF{}, FF{}, cur, cbase, rbase, tk, tk, nxt);
            }
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                store1(cbase, r, pv[r]);
            }
        }
    }
    // ---- whatever the other segments have left (normally nothing: ONE look at the eight counters)
    __builtin_amdgcn_s_waitcnt(0x0070); // this wave's panel DMA (a last tile re-requests its own) and stores are done
    __syncthreads();
    if (tid < 64)
    {
        bool has = false;
        if (tid < 8)
        {
            const int seen = __hip_atomic_load(ticket_all + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            has            = seen < seg_off[tid + 1] - seg_off[tid];
        }
        const unsigned long long bal = __ballot(has);
        if (tid == 0)
        {
            s_mask = (unsigned)bal & 0xFFu;
        }
    }
    __syncthreads();
    const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask);
    for (int sft = 1; sft < 8 && mask != 0u; sft++)
    {
        const int x2 = (xcd + sft) & 7;
        if (!((mask >> x2) & 1u))
        {
            continue;
        }
        tile_list = tile_list_all + seg_off[x2];
        ntiles    = seg_off[x2 + 1] - seg_off[x2];
        ticket    = ticket_all + x2;
        for (;;)
        {
            __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads(); // s_tk and the ring are free again
            if (tid == 0)
            {
                s_tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            const int tks = __builtin_amdgcn_readfirstlane(s_tk);
            if (tks >= ntiles)
            {
                break;
            }
            const int2     cur   = tile_list[tks];
            const unsigned cbase = tile_base(cur);
            prologue(cur);
            (void)process(TT{}, TT{}, cur, cbase, 0u, 0, tk, nxt);
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                store1(cbase, r, pv[r]);
            }
        }
    }
}

} // namespace cslam
