// kg_post.hip — the ordering post-pass of the scan kernels (gfx950).
//
// The scan kernels never wait on each other: every wave leaves, per 8/32-KiB unit, one info word
// {line bits, line count, hit count} and its hits as unit-ordered start offsets in a fixed staging
// slot.  Because units are contiguous pieces of the haystack in index order, the global emission
// order of the reference (ascending start; for Aho-Corasick ascending end, then start) is the
// concatenation of the unit lists.  This file turns that into the final match_result_t layout:
//   K1 post_reduce : per 1024-unit block: hit-count sum and composed line summary
//   K2 post_carry  : one wave walks the (<= a few thousand) block records: exclusive offsets + line carry
//   K3 post_offsets: per unit: exclusive global index of its first hit; distinct-line corrections
//   K4 post_gather : coalesced copy staging slot -> match_position_t[offset ..]
// All of it touches only O(units + matches) bytes — ~1 % of the scan for BASELINE config 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "kg_common.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

constexpr int kPostBlock = 256;          // threads
constexpr int kPostUnitsPerThread = 4;
constexpr int kPostUnitsPerBlock = kPostBlock * kPostUnitsPerThread; // 1024

struct Bits { bool nl, head, tail; };
__device__ __forceinline__ Bits bits_of(u64 w) { return Bits{(w & kLnNl) != 0, (w & kLnHead) != 0, (w & kLnTail) != 0}; }
__device__ __forceinline__ u64 word_of(Bits b) { return (b.nl ? kLnNl : 0) | (b.head ? kLnHead : 0) | (b.tail ? kLnTail : 0); }
// a then b (a is earlier in the text)
__device__ __forceinline__ Bits compose(Bits a, Bits b)
{
    return Bits{a.nl || b.nl, a.nl ? a.head : (a.head || b.head), b.nl ? b.tail : (a.tail || b.tail)};
}
__device__ __forceinline__ u32 plane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// compose the 64 lane summaries of a wave in lane order (lane 0 = earliest)
__device__ __forceinline__ Bits wave_compose(Bits mine)
{
    const u64 nlm = __ballot(mine.nl), hdm = __ballot(mine.head), tlm = __ballot(mine.tail);
    Bits r;
    r.nl = nlm != 0;
    if (r.nl)
    {
        const int first = __builtin_ctzll(nlm), last = 63 - __builtin_clzll(nlm);
        r.head = (hdm & ((2ull << first) - 1ull)) != 0; // lanes <= first newline holder
        r.tail = (tlm & ~((1ull << last) - 1ull)) != 0; // lanes >= last newline holder
    }
    else
        r.head = r.tail = (hdm | tlm) != 0;
    return r;
}
// "is a matched line open on entry to my lane", given the state on entry to the wave
__device__ __forceinline__ bool wave_open_in(Bits mine, bool wave_in, u32 lane)
{
    const u64 nlm = __ballot(mine.nl), tlm = __ballot(mine.tail);
    const u64 lt = (1ull << lane) - 1ull;
    const u64 nl_below = nlm & lt;
    if (nl_below)
    {
        const int q = 63 - __builtin_clzll(nl_below);
        return (tlm & lt & ~((1ull << q) - 1ull)) != 0; // lanes q .. me-1
    }
    return wave_in || (tlm & lt) != 0;
}

__global__ __launch_bounds__(kPostBlock) void post_reduce(const u64 *__restrict__ info, u64 n_units, u64 *__restrict__ blk_sum,
                                                          u64 *__restrict__ blk_bits)
{
    __shared__ u64 s_sum[4];
    __shared__ u64 s_bits[4];
    const u32 lane = plane_id(), wave = threadIdx.x >> 6;
    const u64 u0 = (u64)blockIdx.x * kPostUnitsPerBlock + (u64)threadIdx.x * kPostUnitsPerThread;
    u64 sum = 0;
    Bits b{false, false, false};
#pragma unroll
    for (int k = 0; k < kPostUnitsPerThread; ++k)
    {
        const u64 w = (u0 + k < n_units) ? info[u0 + k] : 0ull;
        sum += w & kUiCountMask;
        b = compose(b, bits_of(w));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        sum += __shfl_xor(sum, o);
    const Bits wb = wave_compose(b);
    if (lane == 0)
    {
        s_sum[wave] = sum;
        s_bits[wave] = word_of(wb);
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        u64 t = 0;
        Bits c{false, false, false};
        for (int w = 0; w < 4; ++w)
        {
            t += s_sum[w];
            c = compose(c, bits_of(s_bits[w]));
        }
        blk_sum[blockIdx.x] = t;
        blk_bits[blockIdx.x] = word_of(c);
    }
}

// one wave scans the block records 64 at a time (n_blocks = units/1024: 1024 for a 32 GiB shard)
__global__ __launch_bounds__(64) void post_carry(u64 n_blocks, u64 *__restrict__ blk_sum, u64 *__restrict__ blk_bits, Counters *ctr)
{
    const u32 lane = plane_id();
    u64 run = 0;                 // hits in all earlier blocks (uniform)
    Bits c{false, false, false}; // composition of all earlier blocks (uniform)
    for (u64 b0 = 0; b0 < n_blocks; b0 += 64)
    {
        const u64 b = b0 + lane;
        const bool live = b < n_blocks;
        const u64 s = live ? blk_sum[b] : 0ull;
        const Bits mine = live ? bits_of(blk_bits[b]) : Bits{false, false, false};
        u64 incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const u64 v = __shfl_up(incl, o);
            if (lane >= (u32)o)
                incl += v;
        }
        const bool open = wave_open_in(mine, c.tail, lane);
        if (live)
        {
            blk_sum[b] = run + (incl - s);      // exclusive offset of the block
            blk_bits[b] = open ? 1ull : 0ull;   // a matched line is open on entry to the block
        }
        run += __shfl(incl, 63);
        c = compose(c, wave_compose(mine));
    }
    if (lane == 0)
        ctr->summary = word_of(c);
}

__global__ __launch_bounds__(kPostBlock) void post_offsets(const u64 *__restrict__ info, u64 n_units,
                                                           const u64 *__restrict__ blk_off, const u64 *__restrict__ blk_open,
                                                           u64 *__restrict__ offsets, Counters *ctr, int want_lines)
{
    __shared__ u32 s_wsum[4];
    __shared__ u64 s_wbits[4];
    __shared__ u64 s_lines[4];
    const u32 lane = plane_id(), wave = threadIdx.x >> 6;
    const u64 u0 = (u64)blockIdx.x * kPostUnitsPerBlock + (u64)threadIdx.x * kPostUnitsPerThread;
    u64 w[kPostUnitsPerThread];
    u32 tsum = 0;
    Bits tb{false, false, false};
#pragma unroll
    for (int k = 0; k < kPostUnitsPerThread; ++k)
    {
        w[k] = (u0 + k < n_units) ? info[u0 + k] : 0ull;
        tsum += (u32)(w[k] & kUiCountMask);
        tb = compose(tb, bits_of(w[k]));
    }
    // exclusive prefix of tsum inside the wave
    u32 incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const u32 v = __shfl_up(incl, o);
        if (lane >= (u32)o)
            incl += v;
    }
    const Bits wb = wave_compose(tb);
    if (lane == 63)
        s_wsum[wave] = incl;
    if (lane == 0)
        s_wbits[wave] = word_of(wb);
    __syncthreads();
    u64 base = blk_off[blockIdx.x];
    bool open = blk_open[blockIdx.x] != 0;
    for (u32 q = 0; q < wave; ++q)
    {
        base += s_wsum[q];
        const Bits qb = bits_of(s_wbits[q]);
        open = qb.nl ? qb.tail : (open || qb.head);
    }
    u64 off = base + (incl - tsum);
    bool lane_open = wave_open_in(tb, open, lane);
    u64 lines = 0;
#pragma unroll
    for (int k = 0; k < kPostUnitsPerThread; ++k)
    {
        if (u0 + k < n_units)
        {
            offsets[u0 + k] = off;
            off += w[k] & kUiCountMask;
            if (want_lines)
            {
                const Bits ub = bits_of(w[k]);
                lines += (w[k] >> kUiLineShift) & kUiLineMask;
                if (lane_open && ub.head)
                    lines -= 1; // the first matched line of this unit continues one already counted
                lane_open = ub.nl ? ub.tail : (lane_open || ub.head);
            }
        }
    }
    if (want_lines)
    {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            lines += __shfl_xor(lines, o);
        if (lane == 0)
            s_lines[wave] = lines;
        __syncthreads();
        if (threadIdx.x == 0)
        {
            const u64 t = s_lines[0] + s_lines[1] + s_lines[2] + s_lines[3];
            if (t)
                atomicAdd(&ctr->lines, t);
        }
    }
}

// Staging -> final records.  fixed_len > 0 (single literal): the slot holds 16-bit offsets relative to the unit's
// first byte, records are {unit_origin + rel, + fixed_len}; fixed_len == 0 (Aho-Corasick): the staged 64-bit word
// (unit-relative start + 1024) << 11 | len, len <= 1024 (the start of a match that ends in the unit lies at most 1023 bytes
// in front of it).
__global__ __launch_bounds__(kPostBlock) void post_gather(const u64 *__restrict__ info, u64 n_units,
                                                          const u64 *__restrict__ offsets, const u64 *__restrict__ stage,
                                                          u32 stage_cap, u32 fixed_len, u64 origin, u64 unit_bytes,
                                                          u64 *__restrict__ positions, u64 pos_cap)
{
    const unsigned short *stage16 = reinterpret_cast<const unsigned short *>(stage);
    const u32 lane = plane_id();
    const u64 n_waves = (u64)gridDim.x * (kPostBlock / 64);
    const u64 wid = (u64)blockIdx.x * (kPostBlock / 64) + (threadIdx.x >> 6);
    const u32 *stage32 = reinterpret_cast<const u32 *>(stage);
    // record i of the unit whose first byte is `org` and whose staging slot starts at entry `sbase`
    auto put = [&](u64 idx, u64 org, u64 sbase, u32 i) {
        u64 s, e;
        if (fixed_len)
        {
            s = org + stage16[sbase + i];
            e = s + fixed_len;
        }
        else
        {
            const u32 w = stage32[sbase + i];
            s = org + (u64)(w >> 11) - 1024ull;
            e = s + (w & 2047u);
        }
        *reinterpret_cast<uint4 *>(positions + 2 * idx) = make_uint4((u32)s, (u32)(s >> 32), (u32)e, (u32)(e >> 32));
    };
    for (u64 g = wid * 64; g < n_units; g += n_waves * 64)
    {
        const u64 u = g + lane;
        u32 cnt = 0;
        u64 off = 0;
        if (u < n_units)
        {
            cnt = (u32)(info[u] & kUiCountMask);
            off = offsets[u];
            if (cnt > stage_cap || off >= pos_cap)
                cnt = 0; // overflowed units are written by the scan kernel's emit mode
        }
        if (!fixed_len)
        {
            // multi-pattern (5.3 matches per 16 KiB unit): the 64 units of the wave are one contiguous run of records — the
            // offsets of consecutive units follow each other — so lane j takes record j of the RUN: which unit it belongs to is
            // a 6-step search over the wave's inclusive counts, its staged word one scattered 4-byte read (L2), and the 16-byte
            // stores of a wave instruction are consecutive.  (Round 2 let every lane copy its own unit's records: 64 serial
            // little loops with scattered stores, 0.23 ms for 11.2 M records.)
            const u32 raw = u < n_units ? (u32)(info[u] & kUiCountMask) : 0u; // counts as the offsets saw them (overflowed units too)
            u32 incl = raw;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const u32 t = __shfl_up(incl, o);
                if (lane >= (u32)o)
                    incl += t;
            }
            const u32 total = __shfl(incl, 63);
            const u64 off0 = __shfl(u < n_units ? offsets[u] : 0ull, 0); // g < n_units: lane 0 is a real unit
            for (u32 j0 = 0; j0 < total; j0 += 64u) // wave-uniform trip count: every lane takes part in the shuffles below
            {
                const u32 j = j0 + lane;
                u32 own = 0; // first lane whose inclusive count exceeds j
#pragma unroll
                for (u32 step = 32; step; step >>= 1)
                {
                    const u32 t = __shfl(incl, (own + step - 1u) & 63u);
                    if (t <= j)
                        own += step;
                }
                own &= 63u;
                const u32 oincl = __shfl(incl, own), oraw = __shfl(raw, own);
                const u32 i = j - (oincl - oraw);
                const u64 idx = off0 + j;
                if (j < total && oraw <= stage_cap && idx < pos_cap) // (an overflowed unit's records come from the scan kernel's emit mode)
                    put(idx, origin + (g + (u64)own) * unit_bytes, (g + (u64)own) * (u64)stage_cap, i);
            }
            continue;
        }
        // few records: the owning lane copies them itself
        if (cnt && cnt <= 4u)
        {
            const u64 sbase = u * (u64)stage_cap, org = origin + u * unit_bytes;
            for (u32 i = 0; i < cnt; ++i)
                if (off + i < pos_cap)
                    put(off + i, org, sbase, i);
            cnt = 0;
        }
        // many records: the whole wave copies one unit at a time (coalesced)
        u64 big = __ballot(cnt != 0);
        if (fixed_len && stage_cap <= 512)
        {
            // dense single-literal units (the 1 % single byte: ~330 records per 32 KiB unit).  A wave is a serial chain of
            // load -> store steps, and only ~8192 waves are resident: with one 64-record step per memory round trip the gather
            // ran at 262 K records/us (1.6 ms for 343 M).  All (<= 8) staged loads of a unit are issued before its first
            // store — UNCONDITIONALLY, with clamped indices: predicated loads each got their own branch and wait.
            while (big)
            {
                const int l = __builtin_ctzll(big);
                big &= big - 1;
                const u32 c = __shfl(cnt, l);
                const u64 o = __shfl(off, l);
                const u64 sbase = (g + (u64)l) * (u64)stage_cap, org = origin + (g + (u64)l) * unit_bytes;
                u32 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                {
                    const u32 i = lane + 64u * (u32)k;
                    v[k] = stage16[sbase + (i < c ? i : c - 1u)];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                {
                    const u32 i = lane + 64u * (u32)k;
                    if (i < c && o + i < pos_cap)
                    {
                        const u64 s0 = org + v[k], e0 = s0 + fixed_len;
                        *reinterpret_cast<uint4 *>(positions + 2 * (o + i)) = make_uint4((u32)s0, (u32)(s0 >> 32), (u32)e0, (u32)(e0 >> 32));
                    }
                }
            }
        }
        while (big)
        {
            const int l = __builtin_ctzll(big);
            big &= big - 1;
            const u32 c = __shfl(cnt, l);
            const u64 o = __shfl(off, l);
            const u64 sbase = (g + (u64)l) * (u64)stage_cap, org = origin + (g + (u64)l) * unit_bytes;
            for (u32 i = lane; i < c; i += 64)
                if (o + i < pos_cap)
                    put(o + i, org, sbase, i);
        }
    }
}

// ---- host driver --------------------------------------------------------------------------------
void post_free(PostScratch &s)
{
    if (s.d_unitinfo) (void)hipFree(s.d_unitinfo);
    if (s.d_offsets) (void)hipFree(s.d_offsets);
    if (s.d_blk) (void)hipFree(s.d_blk);
    if (s.d_stage) (void)hipFree(s.d_stage);
    if (s.d_occ) (void)hipFree(s.d_occ);
    if (s.d_keep) (void)hipFree(s.d_keep);
    if (s.d_gblk) (void)hipFree(s.d_gblk);
    if (s.d_surv) (void)hipFree(s.d_surv);
    if (s.d_tk) (void)hipFree(s.d_tk);
    s = PostScratch{};
}

#define PCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int post_reserve(PostScratch &s, uint64_t n_units, uint64_t stage_words)
{
    if (n_units > s.units_cap)
    {
        if (s.d_unitinfo) (void)hipFree(s.d_unitinfo);
        if (s.d_offsets) (void)hipFree(s.d_offsets);
        if (s.d_blk) (void)hipFree(s.d_blk);
        s.d_unitinfo = nullptr; s.d_offsets = nullptr; s.d_blk = nullptr; s.units_cap = 0;
        const uint64_t nb = (n_units + kPostUnitsPerBlock - 1) / kPostUnitsPerBlock;
        PCHK(hipMalloc(&s.d_unitinfo, n_units * sizeof(u64)));
        PCHK(hipMalloc(&s.d_offsets, n_units * sizeof(u64)));
        PCHK(hipMalloc(&s.d_blk, 2 * nb * sizeof(u64)));
        s.units_cap = n_units;
    }
    if (stage_words > s.stage_cap_words)
    {
        if (s.d_stage) (void)hipFree(s.d_stage);
        s.d_stage = nullptr; s.stage_cap_words = 0;
        PCHK(hipMalloc(&s.d_stage, stage_words * sizeof(u64)));
        s.stage_cap_words = stage_words;
    }
    if (getenv("KREP_GPU_DEBUG_ALLOC")) // tools/placement_probe.py
        fprintf(stderr, "[krep_gpu] scratch: unitinfo %p offsets %p blk %p stage %p (%llu units, %llu stage words)\n", (void *)s.d_unitinfo,
                (void *)s.d_offsets, (void *)s.d_blk, (void *)s.d_stage, (unsigned long long)s.units_cap, (unsigned long long)s.stage_cap_words);
    return 0;
}

int post_offsets_pass(PostScratch &s, uint64_t n_units, bool want_lines, Counters *d_ctr, hipStream_t st)
{
    const uint64_t nb = (n_units + kPostUnitsPerBlock - 1) / kPostUnitsPerBlock;
    u64 *blk_sum = (u64 *)s.d_blk, *blk_bits = (u64 *)s.d_blk + nb;
    hipLaunchKernelGGL(post_reduce, dim3((u32)nb), dim3(kPostBlock), 0, st, (const u64 *)s.d_unitinfo, (u64)n_units, blk_sum,
                       blk_bits);
    hipLaunchKernelGGL(post_carry, dim3(1), dim3(64), 0, st, (u64)nb, blk_sum, blk_bits, d_ctr);
    hipLaunchKernelGGL(post_offsets, dim3((u32)nb), dim3(kPostBlock), 0, st, (const u64 *)s.d_unitinfo, (u64)n_units,
                       (const u64 *)blk_sum, (const u64 *)blk_bits, (u64 *)s.d_offsets, d_ctr, want_lines ? 1 : 0);
    PCHK(hipGetLastError());
    return 0;
}

int post_gather_pass(PostScratch &s, uint64_t n_units, uint32_t stage_cap, uint32_t fixed_len, uint64_t origin,
                     uint64_t unit_bytes, uint64_t *d_pos, uint64_t pos_cap, int num_cu, hipStream_t st)
{
    if (!d_pos || !pos_cap)
        return 0;
    const uint64_t groups = (n_units + 63) / 64;
    const u32 grid = (u32)std::min<uint64_t>((groups + 3) / 4, (uint64_t)num_cu * 16);
    hipLaunchKernelGGL(post_gather, dim3(grid ? grid : 1), dim3(kPostBlock), 0, st, (const u64 *)s.d_unitinfo, (u64)n_units,
                       (const u64 *)s.d_offsets, (const u64 *)s.d_stage, stage_cap, fixed_len, (u64)origin, (u64)unit_bytes,
                       (u64 *)d_pos, (u64)pos_cap);
    PCHK(hipGetLastError());
    return 0;
}

int post_order(PostScratch &s, uint64_t n_units, uint32_t stage_cap, uint32_t fixed_len, uint64_t origin, uint64_t unit_bytes,
               bool want_lines, uint64_t *d_pos, uint64_t pos_cap, Counters *d_ctr, int num_cu, hipStream_t st)
{
    if (post_offsets_pass(s, n_units, want_lines, d_ctr, st))
        return 2;
    if (stage_cap == 0 && pos_cap) // nothing was staged (kg_ac.hip: a dense tiny dictionary): every record comes from the emit-mode launch
        return 0;
    return post_gather_pass(s, n_units, stage_cap, fixed_len, origin, unit_bytes, d_pos, pos_cap, num_cu, st);
}

} // namespace kg
