// read_ceiling.hip — what read bandwidth can ANY kernel get out of this part?  (development probe; not product code)
//   variants: register loads (dwordx4 per lane, LOADS in flight, nontemporal or plain), LDS-DMA (global_load_lds_dwordx4,
//   nt or plain) + ds_read_b128 read-back; persistent grid, static tile striding, 256-thread blocks.
// build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/rc tools/ubench/read_ceiling.hip && /tmp/rc [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, bool NT>
__global__ __launch_bounds__(256) void rd_regs(const uint8_t *__restrict__ p, size_t n, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_bytes = (size_t)LOADS * 1024, tile = wave_bytes * 4;
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t * tile + tile <= n; t += gridDim.x)
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p + t * tile + wave * wave_bytes) + lane;
        u32x4 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            v[j] = NT ? __builtin_nontemporal_load(src + j * 64) : src[j * 64];
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

// LDS-DMA: every wave owns LOADS KiB of LDS; 16 B per lane per instruction land at base + lane * 16
template <int LOADS, int AUX>
__global__ __launch_bounds__(256) void rd_ldsdma(const uint8_t *__restrict__ p, size_t n, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wave_bytes = (size_t)LOADS * 1024, tile = wave_bytes * 4;
    uint8_t *mine = smem + wave * wave_bytes;
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t * tile + tile <= n; t += gridDim.x)
    {
        const uint8_t *src = p + t * tile + wave * wave_bytes + lane * 16;
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                             (__attribute__((address_space(3))) void *)(mine + j * 1024), 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
        {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(mine + j * 1024 + lane * 16);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

// the skeleton of kg::lit_scan without its compute: one atomic ticket per 128 KiB tile (prefetched), one __syncthreads per
// tile, per wave R rounds of 8 x 1 KiB loads + the 8-byte look-ahead load behind each round.
// WORK: 0 = xor only; 1 = the compare work of the 8-byte literal (12 v_alignbyte + 16 v_cmp + ballots per cell)
template <int R, int WORK, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void rd_skeleton(const uint8_t *__restrict__ p, size_t n, uint32_t *out,
                                                                   unsigned long long *ticket, uint32_t p0, uint32_t p1)
{
    __shared__ unsigned long long s_ticket[2];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t unit = (size_t)R * 8192, tile = unit * 4, ntiles = n / tile;
    uint32_t acc = 0;
    unsigned long long hits = 0;
    unsigned long long next = 0;
    if (threadIdx.x == 0)
        next = atomicAdd(ticket, 1ull);
    for (uint32_t it = 0;; ++it)
    {
        if (threadIdx.x == 0)
            s_ticket[it & 1] = next;
        __syncthreads();
        const unsigned long long t = s_ticket[it & 1];
        if (t >= ntiles)
            break;
        if (threadIdx.x == 0)
            next = atomicAdd(ticket, 1ull);
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const uint8_t *seg = p + t * tile + wave * unit + (size_t)r * 8192;
            const u32x4 *src = reinterpret_cast<const u32x4 *>(seg) + lane;
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = __builtin_nontemporal_load(src + j * 64);
            const uint2 after = (t * tile + wave * unit + (size_t)(r + 1) * 8192 + 8 <= n)
                                    ? *reinterpret_cast<const uint2 *>(seg + 8192) : make_uint2(0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if (WORK == 0)
                    acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w ^ after.x;
                else
                {
                    uint32_t D[6] = {v[j].x, v[j].y, v[j].z, v[j].w, 0, 0};
                    const uint32_t n0 = __shfl_down(D[0], 1), n1 = __shfl_down(D[1], 1);
                    const uint32_t e0 = j + 1 < 8 ? __builtin_amdgcn_readfirstlane(v[j + 1 < 8 ? j + 1 : j].x) : after.x;
                    const uint32_t e1 = j + 1 < 8 ? __builtin_amdgcn_readfirstlane(v[j + 1 < 8 ? j + 1 : j].y) : after.y;
                    D[4] = lane == 63 ? e0 : n0;
                    D[5] = lane == 63 ? e1 : n1;
                    unsigned long long any = 0;
                    uint32_t A0[16];
                    bool c[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                    {
                        A0[k] = (k & 3) == 0 ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1], D[k >> 2], (uint32_t)(k & 3));
                        c[k] = A0[k] == p0;
                        any |= __ballot(c[k]);
                    }
                    if (any)
                    {
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                        {
                            const uint32_t a4 = k < 12 ? A0[k + 4] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 2], D[(k >> 2) + 1], (uint32_t)(k & 3));
                            if (c[k] && a4 == p1)
                                hits += 1;
                        }
                    }
                }
            }
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
    if (hits)
        atomicAdd((unsigned long long *)(out + 2), hits);
}

// static striding (what kg::lit_scan does since round 2) with the literal8 compare work; PF: rolling prefetch — as soon as cell j
// has been copied out of its registers they receive cell j of the NEXT round (next unit at the end of a unit)
template <int R, bool PF>
__global__ __launch_bounds__(256, 4) void rd_static(const uint8_t *__restrict__ p, size_t n, uint32_t *out, unsigned long long *,
                                                    uint32_t p0, uint32_t p1)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t unit_bytes = (size_t)R * 8192, n_units = n / unit_bytes;
    unsigned long long hits = 0;
    u32x4 v[8];
    bool have = false;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t unit = (size_t)blockIdx.x * 4 + wave; unit < n_units; unit += stride)
    {
        const uint8_t *ubase = p + unit * unit_bytes;
#pragma unroll
        for (int r = 0; r < R; ++r)
        {
            const uint8_t *seg = ubase + (size_t)r * 8192;
            const u32x4 *src = reinterpret_cast<const u32x4 *>(seg) + lane;
            if (!PF || !have)
            {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = __builtin_nontemporal_load(src + j * 64);
            }
            const bool last = r + 1 == R;
            const bool more = !last || unit + stride < n_units;
            const u32x4 *nsrc = !last ? src + 8192 / 16
                                      : (more ? reinterpret_cast<const u32x4 *>(p + (unit + stride) * unit_bytes) + lane : src);
            uint2 after = make_uint2(0, 0);
            if ((!PF || last) && (size_t)(seg - p) + 8192 + 8 <= n)
                after = *reinterpret_cast<const uint2 *>(seg + 8192);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                uint32_t D[6] = {v[j].x, v[j].y, v[j].z, v[j].w, 0, 0};
                if (PF)
                    v[j] = __builtin_nontemporal_load(nsrc + j * 64);
                const uint32_t n0 = __shfl_down(D[0], 1), n1 = __shfl_down(D[1], 1);
                uint32_t e0, e1;
                if (j + 1 < 8)
                {
                    e0 = __builtin_amdgcn_readfirstlane(v[j + 1 < 8 ? j + 1 : j].x);
                    e1 = __builtin_amdgcn_readfirstlane(v[j + 1 < 8 ? j + 1 : j].y);
                }
                else if (PF && !last)
                {
                    e0 = __builtin_amdgcn_readfirstlane(v[0].x); // the next round's first bytes are already on their way in
                    e1 = __builtin_amdgcn_readfirstlane(v[0].y);
                }
                else
                {
                    e0 = after.x;
                    e1 = after.y;
                }
                D[4] = lane == 63 ? e0 : n0;
                D[5] = lane == 63 ? e1 : n1;
                unsigned long long any = 0;
                uint32_t A0[16];
                bool c[16];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                {
                    A0[k] = (k & 3) == 0 ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1], D[k >> 2], (uint32_t)(k & 3));
                    c[k] = A0[k] == p0;
                    any |= __ballot(c[k]);
                }
                if (any)
                {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                    {
                        const uint32_t a4 = k < 12 ? A0[k + 4] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 2], D[(k >> 2) + 1], (uint32_t)(k & 3));
                        if (c[k] && a4 == p1)
                            hits += 1;
                    }
                }
            }
            have = PF && more;
        }
    }
    if (hits)
        atomicAdd((unsigned long long *)(out + 2), hits);
}


// LDS-DMA streaming WITH the literal8 compare (round 6): every wave owns NBUF buffers of KB KiB; the DMA of round
// g + NBUF - 1 is issued before round g is consumed (counted vmcnt), the consumer reads its 16 bytes per lane with ds_read_b128 and
// the 8 bytes behind them with ds_read_b64 (the neighbour's bytes: no shuffle).  Static interleaved deal of R-round units.
template <int KB, int NBUF, int AUX, int R>
__global__ __launch_bounds__(256) void rd_ldsdma_cmp(const uint8_t *__restrict__ p, size_t n, uint32_t *out, unsigned long long *,
                                                     uint32_t p0, uint32_t p1)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t *mine = smem + wave * (NBUF * KB * 1024 + 16);
    const size_t round_bytes = (size_t)KB * 1024, unit_bytes = round_bytes * R, n_units = n / unit_bytes;
    const size_t stride = (size_t)gridDim.x * 4, first = (size_t)blockIdx.x * 4 + wave;
    const size_t my_units = first < n_units ? (n_units - first + stride - 1) / stride : 0, G = my_units * R;
    auto src_of = [&](size_t g) -> const uint8_t * { return p + (first + (g / R) * stride) * unit_bytes + (g % R) * round_bytes + lane * 16; };
    auto issue = [&](size_t g) {
        const uint8_t *src = src_of(g);
        uint8_t *dst = mine + (g % NBUF) * round_bytes;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, AUX);
    };
    unsigned long long hits = 0;
    for (size_t g = 0; g < (size_t)(NBUF - 1) && g < G; ++g)
        issue(g);
    for (size_t g = 0; g < G; ++g)
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the buffer about to be refilled has been read)
        if (g + NBUF - 1 < G)
        {
            issue(g + NBUF - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * KB) : "memory");
        }
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint8_t *b = mine + (g % NBUF) * round_bytes + lane * 16;
        // all LDS reads of the round first (one LDS latency per round, not per cell)
        u32x4 vv[KB];
        uint2 nn[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j)
        {
            vv[j] = *reinterpret_cast<const u32x4 *>(b + j * 1024);
            nn[j] = *reinterpret_cast<const uint2 *>(b + j * 1024 + 16); // (the round's last 8 positions see the next buffer: ubench)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KB; ++j)
        {
            const u32x4 v = vv[j];
            const uint2 nx = nn[j];
            uint32_t D[6] = {v.x, v.y, v.z, v.w, nx.x, nx.y};
            unsigned long long any = 0;
            uint32_t A0[16];
            bool c[16];
#pragma unroll
            for (int k = 0; k < 16; ++k)
            {
                A0[k] = (k & 3) == 0 ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1], D[k >> 2], (uint32_t)(k & 3));
                c[k] = A0[k] == p0;
                any |= __ballot(c[k]);
            }
            if (any)
            {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                {
                    const uint32_t a4 = k < 12 ? A0[k + 4] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 2], D[(k >> 2) + 1], (uint32_t)(k & 3));
                    if (c[k] && a4 == p1)
                        hits += 1;
                }
            }
        }
    }
    if (hits)
        atomicAdd((unsigned long long *)(out + 2), hits);
}

__global__ void fill(uint32_t *p, size_t nwords)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x)
    {
        uint64_t x = i * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        p[i] = (uint32_t)(x * 0xBF58476D1CE4E5B9ull >> 17);
    }
}

template <typename F>
static double time_it(F launch, int reps = 7)
{
    hipEvent_t a, b;
    CHK(hipEventCreate(&a));
    CHK(hipEventCreate(&b));
    launch();
    launch();
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r)
    {
        CHK(hipEventRecord(a));
        launch();
        CHK(hipEventRecord(b));
        CHK(hipEventSynchronize(b));
        float t;
        CHK(hipEventElapsedTime(&t, a, b));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 32.0;
    const size_t n = (size_t)(gib * (1ull << 30));
    uint8_t *buf;
    uint32_t *out;
    CHK(hipMalloc(&buf, n));
    CHK(hipMalloc(&out, 64));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t *)buf, n / 4);
    CHK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
#define RUN(name, kern, bpc, lds)                                                                                     \
    {                                                                                                                 \
        if (lds > 65536) CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        double ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(cu * bpc), dim3(256), lds, 0, buf, n, out); });       \
        CHK(hipGetLastError());                                                                                      \
        printf("%-44s blocks/CU=%d  %7.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, bpc, ms, n / ms / 1e6, n / ms / 1e6 / 8000.0); \
        fflush(stdout);                                                                                               \
    }
    unsigned long long *ticket;
    CHK(hipMalloc(&ticket, 8));
#define RUNS(name, kern, bpc)                                                                                        \
    {                                                                                                                 \
        double ms = time_it([&] {                                                                                     \
            (void)hipMemsetAsync(ticket, 0, 8, 0);                                                                    \
            hipLaunchKernelGGL(kern, dim3(cu * bpc), dim3(256), 0, 0, buf, n, out, ticket, 0x72656853u, 0x6b636f6cu); \
        });                                                                                                           \
        CHK(hipGetLastError());                                                                                      \
        printf("%-44s blocks/CU=%d  %7.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, bpc, ms, n / ms / 1e6, n / ms / 1e6 / 8000.0); \
        fflush(stdout);                                                                                               \
    }

#define RUNL(name, kern, bpc, lds)                                                                                   \
    {                                                                                                                 \
        if (lds > 65536) CHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
        double ms = time_it([&] {                                                                                     \
            hipLaunchKernelGGL(kern, dim3(cu * bpc), dim3(256), lds, 0, buf, n, out, ticket, 0x72656853u, 0x6b636f6cu); \
        });                                                                                                           \
        CHK(hipGetLastError());                                                                                      \
        printf("%-52s blocks/CU=%d  %7.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, bpc, ms, n / ms / 1e6, n / ms / 1e6 / 8000.0); \
        fflush(stdout);                                                                                               \
    }
    if (argc > 2 && !strcmp(argv[2], "ldsdma"))
    {
        // round 6: LDS-DMA streaming with the literal8 compare against the register-load variants with the same compare
        RUNS("static R=4, literal8 compare, ROLLING PREFETCH", (rd_static<4, true>), 4);
        RUNS("static R=4, literal8 compare, no prefetch", (rd_static<4, false>), 4);
        RUN("regs  8 x dwordx4, nontemporal (bare reader)", (rd_regs<8, true>), 2, 0);
        RUN("lds-dma  8 KiB/wave, nt (bare reader)", (rd_ldsdma<8, 2>), 1, 4 * 8 * 1024);
        for (int bpc : {1, 2})
        {
            RUNL("lds-dma + literal8 compare, 2 x 8 KiB per wave, nt", (rd_ldsdma_cmp<8, 2, 2, 4>), bpc, 4 * (2 * 8 * 1024 + 16));
            RUNL("lds-dma + literal8 compare, 2 x 8 KiB per wave, default", (rd_ldsdma_cmp<8, 2, 0, 4>), bpc, 4 * (2 * 8 * 1024 + 16));
        }
        RUNL("lds-dma + literal8 compare, 3 x 8 KiB per wave, nt", (rd_ldsdma_cmp<8, 3, 2, 4>), 1, 4 * (3 * 8 * 1024 + 16));
        RUNL("lds-dma + literal8 compare, 4 x 8 KiB per wave, nt", (rd_ldsdma_cmp<8, 4, 2, 4>), 1, 4 * (4 * 8 * 1024 + 16));
        for (int bpc : {1, 2, 4})
        {
            RUNL("lds-dma + literal8 compare, 2 x 4 KiB per wave, nt", (rd_ldsdma_cmp<4, 2, 2, 8>), bpc, 4 * (2 * 4 * 1024 + 16));
            RUNL("lds-dma + literal8 compare, 4 x 4 KiB per wave, nt", (rd_ldsdma_cmp<4, 4, 2, 8>), bpc > 2 ? 2 : bpc, 4 * (4 * 4 * 1024 + 16));
        }
        RUNL("lds-dma + literal8 compare, 4 x 2 KiB per wave, nt", (rd_ldsdma_cmp<2, 4, 2, 16>), 4, 4 * (4 * 2 * 1024 + 16));
        RUNL("lds-dma + literal8 compare, 8 x 2 KiB per wave, nt", (rd_ldsdma_cmp<2, 8, 2, 16>), 2, 4 * (8 * 2 * 1024 + 16));
        return 0;
    }
    for (int bpc : {4})
    {
        RUNS("static R=4, literal8 compare, no prefetch", (rd_static<4, false>), bpc);
        RUNS("static R=4, literal8 compare, ROLLING PREFETCH", (rd_static<4, true>), bpc);
        RUNS("static R=8, literal8 compare, no prefetch", (rd_static<8, false>), bpc);
        RUNS("static R=8, literal8 compare, ROLLING PREFETCH", (rd_static<8, true>), bpc);
        RUNS("static R=1, literal8 compare, ROLLING PREFETCH", (rd_static<1, true>), bpc);
        RUNS("static R=4, literal8 compare, no prefetch", (rd_static<4, false>), bpc);
        RUNS("static R=4, literal8 compare, ROLLING PREFETCH", (rd_static<4, true>), bpc);
    }
    for (int bpc : {4})
    {
        RUNS("skeleton R=4, xor only, 4 waves/SIMD", (rd_skeleton<4, 0, 4>), bpc);
        RUNS("skeleton R=4, literal8 compare, 4 w/SIMD", (rd_skeleton<4, 1, 4>), bpc);
        RUNS("skeleton R=4, literal8 compare, 2 w/SIMD", (rd_skeleton<4, 1, 2>), bpc);
        RUNS("skeleton R=1, literal8 compare, 4 w/SIMD", (rd_skeleton<1, 1, 4>), bpc);
        RUNS("skeleton R=2, literal8 compare, 4 w/SIMD", (rd_skeleton<2, 1, 4>), bpc);
    }
    for (int bpc : {2, 4})
    {
        RUN("regs  8 x dwordx4, nontemporal", (rd_regs<8, true>), bpc, 0);
        RUN("regs  8 x dwordx4, plain", (rd_regs<8, false>), bpc, 0);
        RUN("regs 16 x dwordx4, nontemporal", (rd_regs<16, true>), bpc, 0);
        RUN("regs  4 x dwordx4, nontemporal", (rd_regs<4, true>), bpc, 0);
    }
    for (int bpc : {1, 2, 4})
    {
        RUN("lds-dma  8 KiB/wave, default policy", (rd_ldsdma<8, 0>), bpc, 4 * 8 * 1024);
        RUN("lds-dma  8 KiB/wave, nt (aux=2)", (rd_ldsdma<8, 2>), bpc, 4 * 8 * 1024);
        RUN("lds-dma 16 KiB/wave, nt (aux=2)", (rd_ldsdma<16, 2>), bpc > 2 ? 2 : bpc, 4 * 16 * 1024);
    }
    return 0;
}
