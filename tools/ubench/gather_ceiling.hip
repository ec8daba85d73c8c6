// gather_ceiling.hip — what does the FORMULATION of the multi-pattern scan cost on this part, with no filter and no verifier
// in it?  (development probe; not product code — VERDICT r04 item 1a.)
//
// kg::ac_scan_kernel (krep_amd/csrc/kg_ac.hip) is: one 1024-thread workgroup per CU (the 128-KiB LDS table), 16 autonomous
// waves, each streaming 16-KiB units with TEMPORAL 16 B/lane loads (8 KiB in flight per wave, rolling register prefetch),
// and once per unit verifying its ~42 filter candidates, one per lane: ONE unaligned 20-byte text window read back from the
// unit it has just streamed, then ONE dependent 64-byte bucket of an L2-resident table.  This probe keeps exactly that memory
// behaviour and drops everything else: the "filter" is an XOR of the loaded dwords, a "candidate" is a pseudo-random offset
// inside the unit, the "table" is 512 KiB of buckets indexed by the window's last dword.
//   K      windows per 16-KiB unit (0, 21, 42, 84; > 64: two batches), issued 64 at a time
//   PROBE  the dependent bucket read(s) behind the window: 1 = one 64-byte bucket, 2 = two (both ends of a stride-2 candidate)
//   GRAN   verify granularity: 2 = once per unit (as shipped), 1 = once per 8-KiB round with K/2 windows each
//   DEFER  the windows of granule g are issued, granule g + 1 is streamed and "filtered", THEN they are consumed
//          (the deferred-by-one verification VERDICT r04 asks to measure)
//   NT     non-temporal stream loads (the literal scan's choice; round 3 measured it slower for this kernel)
// build & run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gc tools/ubench/gather_ceiling.hip && /tmp/gc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr u32 kCell = 1024, kCells = 8, kSeg = kCell * kCells, kRounds = 2, kUnit = kSeg * kRounds, kUpt = 8;
constexpr u32 kBuckets = 8192; // x 64 B = 512 KiB (the product's chain-compressed 4-gram table is of this order)

__global__ void fill(u32 *p, size_t words, u32 seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
    {
        u64 x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        p[i] = (u32)(x >> 32);
    }
}

struct __attribute__((packed)) U32p { u32 v; };

template <int GRAN, bool DEFER, int PROBE, bool NT>
__global__ __launch_bounds__(1024) void scan(const uint8_t *__restrict__ text, size_t n_units, const uint4 *__restrict__ table, u32 K,
                                             u64 *ticket, u32 *out)
{
    extern __shared__ u32 s_mem[]; // 128 KiB claimed like the product's table (one workgroup per CU); touched once
    s_mem[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const u32 lane = threadIdx.x & 63u;
    u32 acc = s_mem[(threadIdx.x * 7u) & 1023u];
    const u32 kg = K / (GRAN == 1 ? 2u : 1u); // windows per granule
    // deferred windows: registers of the granule before
    u32 pw[2][5];
    bool pend = false;
    u32 pend_n = 0;
    for (;;)
    {
        u64 tk = 0;
        if (lane == 0)
            tk = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = ((u64)__builtin_amdgcn_readfirstlane((u32)(tk >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)tk);
        const u64 u_begin = tk * kUpt;
        if (u_begin >= n_units)
            break;
        const u64 u_end = u_begin + kUpt < n_units ? u_begin + kUpt : n_units;
        u32x4 d[kCells];
        bool have = false;
        for (u64 unit = u_begin; unit < u_end; ++unit)
        {
            const u64 useg = unit * (u64)kUnit;
#pragma unroll
            for (int r = 0; r < (int)kRounds; ++r)
            {
                const u64 seg = useg + (u64)r * kSeg;
                const u32x4 *src = reinterpret_cast<const u32x4 *>(text + seg) + lane;
                if (!have)
                {
#pragma unroll
                    for (int j = 0; j < (int)kCells; ++j)
                        d[j] = NT ? __builtin_nontemporal_load(src + j * 64) : src[j * 64];
                }
                const bool pf_next = r + 1 < (int)kRounds || unit + 1 < u_end;
                const u32x4 *nsrc = pf_next ? src + kSeg / 16 : src;
#pragma unroll
                for (int j = 0; j < (int)kCells; ++j)
                {
                    acc ^= d[j].x ^ (d[j].y >> 1) ^ (d[j].z << 1) ^ d[j].w;
                    d[j] = NT ? __builtin_nontemporal_load(nsrc + j * 64) : nsrc[j * 64];
                }
                have = pf_next;
                if (GRAN == 2 && r + 1 < (int)kRounds)
                    continue;
                // ---- "verify" this granule: kg windows, 64 at a time --------------------------------------------------------
                const u64 gbase = GRAN == 2 ? useg : seg;
                const u32 gbytes = GRAN == 2 ? kUnit : kSeg;
                auto consume = [&](const u32 (&w)[5], bool live) {
                    if (!live)
                        return;
                    u32 x = w[0] ^ w[1] ^ w[2] ^ w[3] ^ w[4];
                    if (PROBE)
                    {
                        const uint4 *b = table + 4 * (size_t)(((w[3] * 0x9E3779B1u) >> 9) & (kBuckets - 1));
                        const uint4 q0 = b[0], q1 = b[1], q2 = b[2], q3 = b[3];
                        if (PROBE == 2) // both ends of a stride-2 candidate: a second bucket, keyed one byte further, in flight with the first
                        {
                            const u32 wb = __builtin_amdgcn_alignbyte(w[4], w[3], 1);
                            const uint4 *c = table + 4 * (size_t)(((wb * 0x9E3779B1u) >> 9) & (kBuckets - 1));
                            const uint4 r0 = c[0], r1 = c[1], r2 = c[2], r3 = c[3];
                            x ^= r0.y ^ r1.z ^ r2.w ^ r3.x;
                        }
                        x ^= q0.x ^ q1.y ^ q2.z ^ q3.w;
                    }
                    acc += x;
                };
                auto issue = [&](u32 (&w)[5], u32 idx, bool live) {
                    if (!live)
                        return;
                    // pseudo-random position inside the granule (>= 19 bytes in, so the window never leaves it)
                    u32 h = (u32)(gbase >> 10) * 0x85EBCA6Bu + idx * 0xC2B2AE35u;
                    h ^= h >> 15;
                    h *= 0x2C1B3C6Du;
                    h ^= h >> 13;
                    const u32 off = 19u + h % (gbytes - 21u);
                    const U32p *q = reinterpret_cast<const U32p *>(text + gbase + off - 15u);
                    w[0] = q[0].v; w[1] = q[1].v; w[2] = q[2].v; w[3] = q[3].v;
                    w[4] = text[gbase + off + 1u];
                };
                if (DEFER)
                {
                    if (pend)
                    {
                        consume(pw[0], lane < pend_n);
                        consume(pw[1], lane + 64u < pend_n);
                    }
                    issue(pw[0], lane, lane < kg);
                    if (kg > 64u)
                        issue(pw[1], lane + 64u, lane + 64u < kg);
                    pend = true;
                    pend_n = kg;
                }
                else
                {
                    for (u32 b0 = 0; b0 < kg; b0 += 64u)
                    {
                        u32 w[5] = {0, 0, 0, 0, 0};
                        issue(w, b0 + lane, b0 + lane < kg);
                        consume(w, b0 + lane < kg);
                    }
                }
            }
        }
    }
    if (DEFER && pend)
    {
        if (lane < pend_n) acc += pw[0][0] ^ pw[0][1] ^ pw[0][2] ^ pw[0][3] ^ pw[0][4];
        if (lane + 64u < pend_n) acc += pw[1][0] ^ pw[1][4];
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <int GRAN, bool DEFER, int PROBE, bool NT>
static float run(const uint8_t *text, size_t n_units, const uint4 *table, u32 K, u64 *ticket, u32 *out, int cus)
{
    auto kern = scan<GRAN, DEFER, PROBE, NT>;
    const u32 lds = 128u << 10;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 7; ++rep)
    {
        CHK(hipMemsetAsync(ticket, 0, 8, nullptr));
        CHK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(kern, dim3(cus), dim3(1024), lds, nullptr, text, n_units, table, K, ticket, out);
        CHK(hipEventRecord(e1, nullptr));
        CHK(hipEventSynchronize(e1));
        CHK(hipGetLastError());
        float t;
        CHK(hipEventElapsedTime(&t, e0, e1));
        if (rep)
            ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    return ms[ms.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 32;
    const size_t n = gib << 30, n_units = n / kUnit;
    uint8_t *text;
    uint4 *table;
    u64 *ticket;
    u32 *out;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    CHK(hipMalloc(&text, n + 64));
    CHK(hipMalloc(&table, (size_t)kBuckets * 64));
    CHK(hipMalloc(&ticket, 8));
    CHK(hipMalloc(&out, 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, nullptr, (u32 *)text, (n + 64) / 4, 1u);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, nullptr, (u32 *)table, (size_t)kBuckets * 16, 77u);
    CHK(hipDeviceSynchronize());
    printf("# gather_ceiling: %zu GiB, %d CUs x 16 waves, 16-KiB units, 8 KiB in flight per wave (rolling prefetch), temporal stream unless NT\n", gib, cus);
    printf("# ms (median of 6) and TB/s of text; K = windows per 16-KiB unit (the shipped kernel verifies 42 per unit on BASELINE config 4)\n");
    printf("%-44s %8s %8s %8s %8s\n", "variant", "K=0", "K=21", "K=42", "K=84");
    const u32 Ks[4] = {0, 21, 42, 84};
#define ROW(name, G, D, P, N)                                                                      \
    {                                                                                              \
        printf("%-44s", name);                                                                     \
        for (u32 K : Ks)                                                                           \
        {                                                                                          \
            const float t = run<G, D, P, N>(text, n_units, table, K, ticket, out, cus);            \
            printf(" %8.3f", t);                                                                   \
        }                                                                                          \
        printf("\n");                                                                              \
        fflush(stdout);                                                                            \
    }
    ROW("per unit, window only", 2, false, 0, false)
    ROW("per unit, window + one dependent bucket", 2, false, 1, false)
    ROW("per unit, window + two buckets (as shipped)", 2, false, 2, false)
    ROW("per unit, deferred by one unit, + 2 buckets", 2, true, 2, false)
    ROW("per 8-KiB round, window + 2 buckets", 1, false, 2, false)
    ROW("per 8-KiB round, deferred by one, + 2 buckets", 1, true, 2, false)
    ROW("NT stream, per unit, window + 2 buckets", 2, false, 2, true)
    ROW("NT stream, per 8-KiB round, deferred, + 2", 1, true, 2, true)
    return 0;
}
