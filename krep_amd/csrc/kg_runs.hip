// kg_runs.hip — the greedy families on a pattern of ONE repeated byte, counted (round 6).
//
// simd_sse42_search (krep.c:4839-4848: a hit advances by the pattern length) and kmp_search / boyer_moore_search under -o report
// leftmost NON-overlapping occurrences.  For a pattern with a border the library resolves that on the ordered list of ALL
// occurrences (kg_greedy.hip) — which for `  `, `--`, `==`, `aa` on real text means materialising hundreds of millions of them
// before a single cluster is walked: `-c -o '  '` ran at 0.65 TB/s.  For a pattern that is m copies of one byte b the answer needs
// no list: inside a maximal run of R bytes b the kept matches start at every m-th byte, floor(R / m) of them.  A kept match ENDS
// where the number of b's of its run, counted from the run's start, is a multiple of m — a property of the position and of the run
// length in front of it, which crosses lanes as a carry: per lane {is every byte b, trailing run length}, a wave scan, the round's
// and the unit's carry in a scalar; a unit finds the run length in front of it by looking back (a run longer than 4 KiB in front
// of a unit switches to the two-level form: per-unit summaries, the carries from a prefix pass on the host).  Counting only: with
// records wanted the list road stays.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <vector>
#include "kg_common.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;
std::atomic<uint64_t> g_runs_launches{0}; // (test hook: krep_gpu_debug_runs_launches)

namespace {
constexpr u64 kRunUnit = (u64)kRoundsBig * kSegBytes; // 32 KiB per wave and step
constexpr u64 kLookBack = 4u * 1024u; // (a run this long in front of a unit: the two-level form)

__device__ __forceinline__ u32 r_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 r_eq16(const uint4 &v, u32 splat, u32 fold)
{ // bit k: byte k of the lane's 16 equals b ((x | fold) == splat: fold = 0x20 per byte for a letter under -i)
    auto eq = [&](u32 x) -> u32 {
        const u32 y = (x | fold) ^ splat;
        const u32 t = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
        return (((t >> 7) * 0x00204081u) >> 21) & 0xfu;
    };
    return eq(v.x) | (eq(v.y) << 4) | (eq(v.z) << 8) | (eq(v.w) << 12);
}
} // namespace

// text[lo, hi): the window whose STARTS are owned, counted from lo (the reference's scan stands at lo: nothing in front of it belongs
// to a run).  out[0] += kept matches, out[1] = max(out[1], end of the last kept match + 1), out[2] |= 1 when a unit gave up.
__global__ __launch_bounds__(256) void run_count_kernel(const uint8_t *__restrict__ text, u64 text_len, u64 lo, u64 hi, u32 m, u32 splat, u32 fold,
                                                        u64 n_units, unsigned long long *out, const u32 *__restrict__ carries)
{
    const u32 lane = r_lane();
    const u64 anchor = lo & ~(u64)15;
    const u64 end_lo = lo + m - 1, end_hi = (hi + m - 1 < text_len) ? hi + m - 1 : text_len; // ENDs of the owned matches
    u64 total = 0, last_end = 0; // (per LANE: summed / maximised over the wave once, when the kernel ends)
    bool gave_up = false;
    const u32 b = splat & 0xffu;
    const u32 inv_m = (u32)(0x100000000ull / m) + 1u; // floor(x / m) == umulhi(x, inv_m) for x < 2^16
    for (u64 unit = (u64)blockIdx.x * kWavesPerBlk + (threadIdx.x >> 6); unit < n_units; unit += (u64)gridDim.x * kWavesPerBlk)
    {
        const u64 ubase = anchor + unit * kRunUnit;
        if (!carries && __hip_atomic_load(&out[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            break; // (some unit gave up: this launch's count is void, the two-level form follows)
        // ---- the run of b's in front of the unit (not in front of lo): 64 bytes per step, backwards
        u32 carry = 0; // length of the run entering the next lane / cell, modulo m (uniform)
        if (carries)
            carry = carries[unit]; // (the two-level form: a run in front of some unit was too long to look back over)
        else
        {
            u64 q = ubase, run = 0;
            while (q > lo)
            {
                const u64 p = q - 1 - lane;
                const bool in = q >= lo + 1 + lane; // p >= lo
                const u32 x = in ? text[p] : 0x100u;
                const u64 ne = __ballot(!in || ((x | (fold & 0xffu)) != b)); // lanes whose byte ends the run (lane 0 = the byte right in front)
                if (ne)
                {
                    run += (u64)__builtin_ctzll(ne);
                    break;
                }
                run += 64;
                q -= 64;
                if (run >= kLookBack)
                {
                    gave_up = true;
                    if (lane == 0)
                        atomicOr(&out[2], 1ull); // (at once: the other waves stop at their next unit)
                    break;
                }
            }
            carry = (u32)(run % m);
        }
        if (gave_up)
            break;
#pragma unroll 1
        for (int r = 0; r < kRoundsBig; ++r)
        {
            const u64 seg = ubase + (u64)r * kSegBytes;
            if (seg >= end_hi)
                break;
            const bool fast = seg + kSegBytes <= text_len;
            const bool interior = seg >= end_lo && seg + kSegBytes <= end_hi && seg >= lo;
            uint4 d[kCells];
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 off = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                if (fast)
                {
                    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(text + off));
                    d[j] = make_uint4(v.x, v.y, v.z, v.w);
                }
                else
                {
                    u32 w[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 16; ++k) // (bytes behind the text stay 0: they are masked out of the run test below)
                        if (off + k < text_len)
                            w[k >> 2] |= (u32)text[off + k] << (8 * (k & 3));
                    d[j] = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                u32 e = r_eq16(d[j], splat, fold);
                // positions in front of lo are not part of any run; positions at or behind the text's end neither
                if (lbase < lo)
                    e &= lo - lbase < 16 ? ~((1u << (u32)(lo - lbase)) - 1u) : 0u;
                if (lbase + 16 > text_len)
                    e &= lbase < text_len ? (1u << (u32)(text_len - lbase)) - 1u : 0u;
                e &= 0xffffu;
                const u64 any = __ballot(e != 0u);
                if (!any)
                {
                    carry = 0;
                    continue;
                }
                // the run entering each lane: {all b, trailing run length} scanned over the wave, the cell's carry in front of lane 0.
                // A lane of sixteen b's is rare in text: without one, what enters a lane is its left neighbour's own trailing run.
                const bool all = e == 0xffffu;
                u32 len = all ? 16u : (u32)__builtin_clz(~(e << 16)); // trailing run = leading ones of the 16-bit mask
                bool al = all;
                if (__ballot(all))
                {
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1)
                    {
                        const u32 pl = __shfl_up(len, o);
                        const bool pa = __shfl_up((int)al, o) != 0;
                        if (lane >= (u32)o && al)
                        {
                            len += pl;
                            al = pa;
                        }
                    }
                }
                // inclusive -> exclusive: what enters THIS lane is the left neighbour's inclusive result (lane 0: the carry) — one DPP
                // wave shift of {length | all << 31}
                const u32 packed = len | (al ? 0x80000000u : 0u);
                const u32 inp = (u32)__builtin_amdgcn_update_dpp((int)0x80000000u, (int)packed, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const u32 x0 = (inp & 0x7fffffffu) + ((inp >> 31) ? carry : 0u);
                const u32 rp0 = x0 - __umulhi(x0, inv_m) * m; // b's of the current run in front of the lane, modulo m (x0 < 2^12)
                u32 cnt = 0, lastk = 0xffu, rest = e;
                if (m == 2u && interior) // (uniform) `  `, `--`, `==`, `aa`: the kept ENDs of every run of the lane at once, no loop
                {
                    // a kept match ends at every byte whose b-count from its run's start is even: odd OFFSETS inside the run — even ones in
                    // the run that enters the lane with one b already counted.  Runs that start at an even bit index: adding their start
                    // bits to e ripples through exactly those runs.
                    const u32 starts = e & ~(e << 1);
                    const u32 re = e & ~(e + (starts & 0x5555u));        // the runs that start at an even index
                    const u32 evenoff = (re & 0x5555u) | (e & ~re & 0xaaaau); // bytes at an even offset inside their run
                    const u32 r0 = (e & 1u) ? (e & ~(e + 1u)) : 0u;       // the run that contains the lane's first byte
                    const u32 flip = rp0 ? r0 : 0u;
                    const u32 kept = ((e & ~evenoff) ^ flip) & e & 0xffffu; // (inside r0 with rp0: even offsets are the kept ones)
                    cnt = (u32)__popc(kept);
                    if (kept)
                        lastk = 31u - (u32)__builtin_clz(kept);
                    rest = 0;
                }
                // run by run (a lane holds one to three): a run of L bytes entered with c b's already counted keeps floor((c + L) / m)
                while (rest)
                {
                    const u32 s0 = (u32)__builtin_ctz(rest), L = (u32)__builtin_ctz(~(rest >> s0));
                    const u32 c = s0 == 0u ? rp0 : 0u, tot = c + L, kq = __umulhi(tot, inv_m); // tot < 32: exact
                    rest = L + s0 >= 32u ? 0u : rest & ~((1u << (L + s0)) - 1u);
                    if (kq)
                    {
                        if (interior)
                        {
                            cnt += kq;
                            lastk = s0 + kq * m - c - 1u; // the last kept END of the run
                        }
                        else
                            for (u32 i = 1; i <= kq; ++i) // (a boundary cell: every END against the owned window)
                            {
                                const u32 k = s0 + i * m - c - 1u;
                                const u64 p = lbase + (u64)k;
                                if (p >= end_lo && p < end_hi)
                                {
                                    ++cnt;
                                    lastk = k;
                                }
                            }
                    }
                }
                // the carry leaving the cell: lane 63's inclusive scan result (+ the old carry when the whole cell was b)
                const u32 p63 = (u32)__builtin_amdgcn_readlane((int)packed, 63);
                const u32 x63 = (p63 & 0x7fffffffu) + ((p63 >> 31) ? carry : 0u);
                carry = x63 - __umulhi(x63, inv_m) * m;
                if (cnt) // (per lane: a lane's later cells lie behind its earlier ones)
                {
                    total += cnt;
                    last_end = lbase + lastk + 1;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        total += ((u64)(u32)__shfl_xor((int)(u32)(total >> 32), o) << 32 | (u32)__shfl_xor((int)(u32)total, o));
        const u64 other = (u64)(u32)__shfl_xor((int)(u32)(last_end >> 32), o) << 32 | (u32)__shfl_xor((int)(u32)last_end, o);
        last_end = other > last_end ? other : last_end;
    }
    if (lane == 0)
    {
        if (total)
        {
            atomicAdd(&out[0], total);
            atomicMax(&out[1], last_end);
        }
        if (gave_up)
            atomicOr(&out[2], 1ull);
    }
}

// The two-level form's first pass: per unit {is every owned byte b, length of the run of b's that ends the unit}
__global__ __launch_bounds__(256) void run_summary_kernel(const uint8_t *__restrict__ text, u64 text_len, u64 lo, u32 splat, u32 fold, u64 n_units,
                                                          uint2 *__restrict__ summ)
{
    const u32 lane = r_lane();
    const u64 anchor = lo & ~(u64)15;
    for (u64 unit = (u64)blockIdx.x * kWavesPerBlk + (threadIdx.x >> 6); unit < n_units; unit += (u64)gridDim.x * kWavesPerBlk)
    {
        const u64 ubase = anchor + unit * kRunUnit;
        u32 trail = 0;
        bool unit_all = true;
        for (u64 cell = 0; cell < kRunUnit / kCellBytes; ++cell)
        {
            const u64 lbase = ubase + cell * kCellBytes + (u64)lane * 16u;
            u32 w[4] = {0, 0, 0, 0};
            if (lbase + 16 <= text_len)
            {
                const uint4 v = *reinterpret_cast<const uint4 *>(text + lbase);
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
            }
            else
                for (int k = 0; k < 16; ++k)
                    if (lbase + k < text_len)
                        w[k >> 2] |= (u32)text[lbase + k] << (8 * (k & 3));
            u32 e = r_eq16(make_uint4(w[0], w[1], w[2], w[3]), splat, fold);
            if (lbase < lo)
                e &= lo - lbase < 16 ? ~((1u << (u32)(lo - lbase)) - 1u) : 0u;
            if (lbase + 16 > text_len)
                e &= lbase < text_len ? (1u << (u32)(text_len - lbase)) - 1u : 0u;
            e &= 0xffffu;
            const bool all = e == 0xffffu;
            u32 len = all ? 16u : (u32)__builtin_clz(~(e << 16));
            bool al = all;
            if (__ballot(all))
            {
#pragma unroll
                for (int o = 1; o < 64; o <<= 1)
                {
                    const u32 pl = __shfl_up(len, o);
                    const bool pa = __shfl_up((int)al, o) != 0;
                    if (lane >= (u32)o && al)
                    {
                        len += pl;
                        al = pa;
                    }
                }
            }
            const u32 l63 = __shfl(len, 63);
            const bool a63 = __shfl((int)al, 63) != 0;
            trail = a63 ? trail + l63 : l63;
            unit_all = unit_all && a63;
        }
        if (lane == 0)
            summ[unit] = make_uint2(trail, unit_all ? 1u : 0u);
    }
}

// -> 0 and *total / *end_p1 (buffer offset behind the last kept match, 0 if none), 1 when a run was too long for the look-back (the
// caller takes the list road), 2 on a HIP error.  d_slots: three zeroable u64 in device memory, h_slots: their pinned mirror.
int runs_count_greedy(const uint8_t *d_text, uint64_t text_len, uint64_t lo, uint64_t hi, uint32_t m, uint8_t byte, bool ci, int num_cu,
                      unsigned long long *d_slots, unsigned long long *h_slots, hipStream_t st, uint64_t *total, uint64_t *end_p1)
{
    *total = 0;
    *end_p1 = 0;
    if (hi <= lo || text_len < m)
        return 0;
    const bool letter = ci && ((byte | 0x20) >= 'a' && (byte | 0x20) <= 'z');
    const u32 splat = 0x01010101u * (u32)(letter ? (byte | 0x20) : byte), fold = letter ? 0x20202020u : 0u;
    const u64 anchor = lo & ~(u64)15, end_hi = std::min<u64>(hi + m - 1, text_len);
    const u64 n_units = (end_hi - anchor + kRunUnit - 1) / kRunUnit;
    if (hipMemsetAsync(d_slots, 0, 3 * sizeof(unsigned long long), st) != hipSuccess)
        return 2;
    const u32 grid = (u32)std::min<u64>((n_units + kWavesPerBlk - 1) / kWavesPerBlk, (u64)num_cu * 4);
    hipLaunchKernelGGL(run_count_kernel, dim3(grid ? grid : 1), dim3(kBlock), 0, st, d_text, (u64)text_len, (u64)lo, (u64)hi, m, splat, fold, n_units, d_slots,
                       (const u32 *)nullptr);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h_slots, d_slots, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return 2;
    g_runs_launches.fetch_add(1, std::memory_order_relaxed);
    if (h_slots[2])
    {
        // A run of more than 64 KiB in front of some unit (a text of `aaaa...`): the two-level form — every unit's {all b, trailing run},
        // the carries from a prefix pass over the units on the host (one step per 32 KiB of text), then the count with the carries given.
        uint2 *d_sum = nullptr;
        u32 *d_car = nullptr;
        std::vector<uint2> sum(n_units);
        std::vector<u32> car(n_units, 0);
        int rc = 2;
        if (hipMalloc(&d_sum, n_units * sizeof(uint2)) == hipSuccess && hipMalloc(&d_car, n_units * sizeof(u32)) == hipSuccess)
        {
            hipLaunchKernelGGL(run_summary_kernel, dim3(grid ? grid : 1), dim3(kBlock), 0, st, d_text, (u64)text_len, (u64)lo, splat, fold, n_units, d_sum);
            if (hipGetLastError() == hipSuccess && hipMemcpyAsync(sum.data(), d_sum, n_units * sizeof(uint2), hipMemcpyDeviceToHost, st) == hipSuccess &&
                hipStreamSynchronize(st) == hipSuccess)
            {
                u64 run = 0; // b's in front of the unit, modulo m
                for (u64 u = 0; u < n_units; ++u)
                {
                    car[u] = (u32)run;
                    run = sum[u].y ? (run + sum[u].x) % m : sum[u].x % m;
                }
                if (hipMemcpyAsync(d_car, car.data(), n_units * sizeof(u32), hipMemcpyHostToDevice, st) == hipSuccess &&
                    hipMemsetAsync(d_slots, 0, 3 * sizeof(unsigned long long), st) == hipSuccess)
                {
                    hipLaunchKernelGGL(run_count_kernel, dim3(grid ? grid : 1), dim3(kBlock), 0, st, d_text, (u64)text_len, (u64)lo, (u64)hi, m, splat, fold,
                                       n_units, d_slots, (const u32 *)d_car);
                    if (hipGetLastError() == hipSuccess &&
                        hipMemcpyAsync(h_slots, d_slots, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st) == hipSuccess &&
                        hipStreamSynchronize(st) == hipSuccess)
                        rc = 0;
                }
            }
        }
        if (d_sum) (void)hipFree(d_sum);
        if (d_car) (void)hipFree(d_car);
        if (rc)
            return (void)hipGetLastError(), 1; // (the list road, which says what it cannot hold)
        g_runs_launches.fetch_add(2, std::memory_order_relaxed);
    }
    *total = h_slots[0];
    *end_p1 = h_slots[1];
    return 0;
}

} // namespace kg
