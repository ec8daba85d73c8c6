// write_ceiling.hip — what does WRITING 16-byte records cost on this part, alone and next to a streaming read?
// (development probe; not product code).  Mirrors the store pattern of kg_single.hip: a wave owns "tickets"; per ticket
// it writes CHUNK bytes of records (64 lanes x 16 B per store instruction) at offset ticket * CHUNK + SKEW.
//   fill        : stores only
//   read+fill   : every ticket first streams RD bytes (dwordx4 nt loads, xor-reduced), then writes its chunk
//   read        : loads only
// build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/wc tools/ubench/write_ceiling.hip && /tmp/wc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the record store with explicit cache-control bits (gfx942/950: sc0 / sc1 = write-through scope, nt = streaming hint)
template <int ST>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v)
{
    if (ST == 0) *p = v;
    else if (ST == 1) __builtin_nontemporal_store(v, p);
    else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (ST == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else if (ST == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
}

// MODE bit0: read, bit1: write.  NTS: 0 plain stores, 1 non-temporal (builtin: nt), 2..5 explicit bit combinations.
template <int MODE, int NTS>
__global__ __launch_bounds__(256, 4) void rw(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n_tickets,
                                             size_t rd_bytes, size_t chunk, size_t skew, unsigned long long *ticket, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (;;)
    {
        unsigned long long t = 0;
        if (lane == 0)
            t = atomicAdd(ticket, 1ull);
        t = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)t);
        if (t >= n_tickets)
            break;
        if (MODE & 1)
        {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(src + t * rd_bytes) + lane;
            for (size_t r = 0; r < rd_bytes / 8192; ++r)
            {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = __builtin_nontemporal_load(p + r * 512 + j * 64);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
            }
        }
        if (MODE & 2)
        {
            uint8_t *q = dst + t * chunk + skew;
            for (size_t i = lane * 16; i + 16 <= chunk; i += 1024)
            {
                const u32x4 v = {(uint32_t)i, acc, (uint32_t)t, 7u};
                store16<NTS>(reinterpret_cast<u32x4 *>(q + i), v);
            }
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <int MODE, int NTS>
static float run(const uint8_t *src, uint8_t *dst, size_t n_tickets, size_t rd, size_t chunk, size_t skew, unsigned long long *tk,
                 uint32_t *out, int grid)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep)
    {
        CHK(hipMemsetAsync(tk, 0, 8, nullptr));
        CHK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((rw<MODE, NTS>), dim3(grid), dim3(256), 0, nullptr, src, dst, n_tickets, rd, chunk, skew, tk, out);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        float t;
        CHK(hipEventElapsedTime(&t, e0, e1));
        if (rep)
            ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const size_t rd = 128 << 10;                       // bytes read per ticket
    const size_t n_tickets = (size_t)(gib * (1 << 30)) / rd;
    uint8_t *src, *dst;
    unsigned long long *tk;
    uint32_t *out;
    CHK(hipMalloc(&src, n_tickets * rd));
    CHK(hipMemset(src, 1, n_tickets * rd));
    CHK(hipMalloc(&tk, 8));
    CHK(hipMalloc(&out, 4));
    const size_t max_chunk = 24 << 10;
    CHK(hipMalloc(&dst, n_tickets * max_chunk + 4096));
    const int grid = 256 * 4;
    const double rgb = n_tickets * (double)rd / 1e9;
    printf("tickets %zu x %zu KiB read (%.1f GB)\n", n_tickets, rd >> 10, rgb);
    printf("read only                          : %7.3f ms  %7.1f GB/s\n", run<1, 0>(src, dst, n_tickets, rd, 0, 0, tk, out, grid),
           rgb / run<1, 0>(src, dst, n_tickets, rd, 0, 0, tk, out, grid) * 1e3);
    for (size_t chunk : {(size_t)20976, (size_t)21504, (size_t)16384})
        for (size_t skew : {(size_t)0, (size_t)48})
        {
            const double wgb = n_tickets * (double)(chunk / 16 * 16) / 1e9;
            const float f = run<2, 0>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid);
            const float fn = run<2, 1>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid);
            const float b = run<3, 0>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid);
            const float bn = run<3, 1>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid);
            printf("chunk %6zu skew %2zu (%.2f GB): fill %6.3f ms %6.1f GB/s | nt %6.3f ms | read+fill %6.3f ms (%6.1f GB/s total) | nt %6.3f ms\n",
                   chunk, skew, wgb, f, wgb / f * 1e3, fn, b, (rgb + wgb) / b * 1e3, bn);
            if (chunk == 20976 && skew == 48)
                printf("   read+fill with explicit bits: sc0 sc1 nt %6.3f ms | sc1 nt %6.3f ms | sc0 sc1 %6.3f ms | sc0 nt %6.3f ms\n",
                       run<3, 2>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid), run<3, 3>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid),
                       run<3, 4>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid), run<3, 5>(src, dst, n_tickets, rd, chunk, skew, tk, out, grid));
        }
    return 0;
}
