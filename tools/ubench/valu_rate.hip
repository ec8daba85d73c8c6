#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef unsigned u32;
template <int OP> __global__ __launch_bounds__(256) void k(u32 *out, u32 seed, int iters)
{
    u32 a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    const u32 K = (128u) | (4u << 16);
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
            {
                if (OP == 0) a[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[i]), __builtin_bit_cast(us2, K), seed, false);
                if (OP == 1) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 2) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 3) asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 4) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 5) asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 6) asm volatile("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 7) asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 8) asm volatile("v_and_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 9) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(a[i]) : "v"(seed), "v"(a[i]));
                if (OP == 10) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(a[i]) : "v"(a[i]));
                if (OP == 11) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 12) asm volatile("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 13) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 14) asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 15) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(K));
                if (OP == 16) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 17) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(K));
                if (OP == 18) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 19) asm volatile("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
                if (OP == 20) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(a[i]) : "v"(a[i]), "v"(K), "s"(seed));
                if (OP == 21) asm volatile("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(a[i]) : "v"(K), "v"(a[i]));
                if (OP == 22) asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(K), "v"(seed));
            }
    }
    u32 s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name)
{
    u32 *d; hipMalloc(&d, 256 * 1024 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 4; // 4 blocks of 4 waves per CU = 4 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 3, 10);
    hipEventRecord(e0); k<OP><<<blocks, 256>>>(d, 3, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 64; // wave-instructions
    printf("%-16s %8.3f ms  %.1f G wave-inst/s (1024 SIMDs: %.2f cycles/inst at 2.1 GHz)\n", name, ms, insts / ms / 1e6, 1024 * 2.1e9 / (insts / ms * 1e3));
}
int main()
{
    run<0>("udot2_u32_u16"); run<1>("mad_u32_u24"); run<2>("mul_lo_u32"); run<3>("mad_u32_u16"); run<4>("bitop3"); run<5>("dot4_u32_u8"); run<6>("xor_sdwa"); run<7>("alignbit"); run<8>("and"); run<9>("lshrrev var"); run<10>("lshrrev const"); run<11>("and_or"); run<12>("lshl_or"); run<13>("add_u32"); run<14>("bfe"); run<15>("mov_dpp wave_shr"); run<16>("perm"); run<17>("xor"); run<18>("bfi"); run<19>("lshl_add"); run<20>("bitop3 sgpr"); run<21>("pk_lshrrev_b16"); run<22>("pk_mad_u16");
    return 0;
}
