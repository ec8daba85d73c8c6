// window_probe.hip — does the SIZE OF THE ACTIVE WINDOW (resident waves x bytes per ticket) bound a streaming scan?
// (development probe; not product code).  The product kernels hand every wave a ticket of 128-256 KiB, so the ~4096 resident
// waves of the chip read a window of 0.5-1 GiB at any moment; tools/ubench/read_ceiling.hip strides 32-KiB block tiles
// statically (a 32 MiB window) and reads 7.07 TB/s where the ticketed readers get 6.3-6.7.  Here: per-wave tickets of
// T KiB drawn from NC interleaved counters (counter c owns the tickets congruent to c mod NC; one counter sustains ~88
// fetch-adds per microsecond, so small tickets need several), read-only and read + 16 % non-temporal record writes.
// build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/wp tools/ubench/window_probe.hip && /tmp/wp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// counters are 128 bytes apart (own L2 lines)
template <bool WRITE>
__global__ __launch_bounds__(256, 4) void scan(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n_tickets, size_t tk_bytes,
                                               size_t rec_bytes, unsigned long long *counters, uint32_t nc, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t c = (blockIdx.x * 4 + (threadIdx.x >> 6)) % nc;
    unsigned long long *ctr = counters + (size_t)c * 16;
    uint32_t acc = 0;
    for (;;)
    {
        unsigned long long t = 0;
        if (lane == 0)
            t = atomicAdd(ctr, 1ull);
        t = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)t);
        t = t * nc + c;
        if (t >= n_tickets)
            break;
        const u32x4 *p = reinterpret_cast<const u32x4 *>(src + t * tk_bytes) + lane;
        for (size_t r = 0; r < tk_bytes / 8192; ++r)
        {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = __builtin_nontemporal_load(p + r * 512 + j * 64);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
        if (WRITE)
        {
            uint8_t *q = dst + t * rec_bytes;
            for (size_t i = lane * 16; i + 16 <= rec_bytes; i += 1024)
            {
                const u32x4 v = {(uint32_t)i, acc, (uint32_t)t, 7u};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(q + i));
            }
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <bool WRITE>
static float run(const uint8_t *src, uint8_t *dst, size_t n_tickets, size_t tk, size_t rec, unsigned long long *ctr, uint32_t nc, uint32_t *out)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 6; ++rep)
    {
        CHK(hipMemsetAsync(ctr, 0, 128 * 64, nullptr));
        CHK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((scan<WRITE>), dim3(1024), dim3(256), 0, nullptr, src, dst, n_tickets, tk, rec, ctr, nc, out);
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

int main()
{
    const size_t n = (size_t)32 << 30;
    uint8_t *src, *dst;
    unsigned long long *ctr;
    uint32_t *out;
    CHK(hipMalloc(&src, n));
    CHK(hipMemset(src, 1, n));
    CHK(hipMalloc(&dst, n / 6 + (1 << 20)));
    CHK(hipMalloc(&ctr, 128 * 64));
    CHK(hipMalloc(&out, 4));
    printf("32 GiB, 4096 resident waves; ticket KiB x counters -> active window; read-only ms (TB/s) | read + 16 %% nt record writes ms\n");
    const struct { size_t kib; uint32_t nc; } cfg[] = {{256, 1}, {128, 1}, {64, 4}, {32, 8}, {16, 16}, {8, 32}, {32, 1}, {128, 8}};
    for (auto &c : cfg)
    {
        const size_t tk = c.kib << 10, nt = n / tk, rec = (tk / 128 * 21 / 1024 * 1024 + 15) / 16 * 16; // ~16 % of the ticket, as records
        const float r = run<false>(src, dst, nt, tk, rec, ctr, c.nc, out);
        const float w = run<true>(src, dst, nt, tk, (tk * 20976 / 131072) / 16 * 16, ctr, c.nc, out);
        printf("ticket %4zu KiB x %2u counters (window %5zu MiB): read %6.3f ms (%5.2f TB/s) | read+write %6.3f ms\n", c.kib, c.nc,
               (4096 * tk) >> 20, r, n / r / 1e9, w);
    }
    return 0;
}
