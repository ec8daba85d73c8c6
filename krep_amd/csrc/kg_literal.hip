// kg_literal.hip — single-literal scan kernel for gfx950 (CDNA4, wave64).
//
// Replaces, behind krep's search_func_t boundary, the hot loops of boyer_moore_search
// (krep.c:1294-1382), memchr_search (:3918-4023), memchr_short_search (:4396-4500) and the
// SIMD literal scans (:4737-4866, :4914-5056, :5145-5257): it produces the set of ALL occurrences
// of one literal (optionally case-folded, optionally -w filtered) whose START lies in the owned
// window, their exact count, their ordered match_position_t list and the line bookkeeping of -c.
//
// Design (HBM-bound byte scan, no MFMA):
//  * every haystack byte is read exactly once with coalesced 16-byte-per-lane loads
//    (global_load_dwordx4, 1 KiB per wave instruction), kCells loads in flight per lane;
//  * all 16 candidate positions of a lane are tested in registers: v_alignbyte_b32 builds the 20
//    unaligned dwords of the lane's 24-byte window (16 own + 8 from the next lane), which are
//    compared with the first 4 (+4) pattern bytes; the per-position lane masks live in SGPR pairs
//    (v_cmp -> s_or / s_bcnt1), so a cell without a candidate costs no further vector work;
//  * longer patterns (> 8 B) use those 8 bytes as the filter and verify the rest on the (rare)
//    candidate lanes only; m == 1 uses an exact SWAR byte compare (memchr path, 1 % hit rate);
//  * ordered output WITHOUT re-reading the haystack and without any inter-workgroup waiting: a
//    wave turns its unit's per-lane 16-bit hit masks into unit-local ranks (ballot bit-planes +
//    v_mbcnt) and stages the start offsets, already in order, in the unit's fixed slot; it
//    publishes one info word per unit {line bits, line count, hit count}.  The post-pass
//    (kg_post.hip: offset scan over the info words + line carry + coalesced gather) produces the
//    globally ordered match_position_t list.  A chained decoupled look-back was measured first and
//    capped the ordered rate at ~3 TB/s (status round trips), see DESIGN.md;
//  * a unit with more hits than its staging slot (very dense inputs) is re-scanned by the same
//    kernel in emit mode, writing its records straight to their final offsets;
//  * units are dealt out statically and interleaved (no ticket, no barrier: every wave runs on its own).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "kg_common.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// popcount(mask & lanes_below_me)
__device__ __forceinline__ u32 mbcnt64(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u64 rfl64(u64 v)
{
    u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// 0x80 in every byte of x that equals the byte replicated in c4 (exact, no false positives)
__device__ __forceinline__ u32 eq_bytes(u32 x, u32 c4)
{
    u32 y = x ^ c4;
    u32 t = (y & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | y | 0x7f7f7f7fu);
}
// gather the four 0x80 flags of a dword into a 4-bit mask
__device__ __forceinline__ u32 movemask4(u32 t) { return (((t >> 7) * 0x00204081u) >> 21) & 0xfu; }

__device__ __forceinline__ bool is_wordc(u32 c)
{
    return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_';
}

// ---- line bookkeeping monoid: a window summarised as (cnt, has_nl, head, tail) -------------
struct LS { u32 cnt; bool nl, head, tail; };
__device__ __forceinline__ LS ls_combine(const LS &a, const LS &b)
{
    LS r;
    r.cnt = a.cnt + b.cnt - ((a.tail && b.head) ? 1u : 0u);
    r.nl = a.nl || b.nl;
    r.head = a.nl ? a.head : (a.head || b.head);
    r.tail = b.nl ? b.tail : (a.tail || b.tail);
    return r;
}
__device__ __forceinline__ u64 ls_bits(const LS &s) { return (s.nl ? kLnNl : 0) | (s.head ? kLnHead : 0) | (s.tail ? kLnTail : 0); }

// wave sum of a small per-lane value (< 32) through ballot bit-planes: SALU only
// (two planes when no lane exceeds 3 — even the 1 % single-byte workload almost never has 4 hits in 16 bytes)
__device__ __forceinline__ u32 wave_sum5(u32 v)
{
    u32 s = (u32)__popcll(__ballot(v & 1u)) + ((u32)__popcll(__ballot((v >> 1) & 1u)) << 1);
    if (__ballot(v > 3u))
    {
#pragma unroll
        for (int b = 2; b < 5; ++b)
            s += (u32)__popcll(__ballot((v >> b) & 1u)) << b;
    }
    return s;
}
// exclusive prefix over lanes of a small per-lane value (< 32)
__device__ __forceinline__ u32 wave_excl5(u32 v)
{
    u32 s = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b)
        s += mbcnt64(__ballot((v >> b) & 1u)) << b;
    return s;
}

// guarded 24-byte window for the (at most two) tiles that touch the end of the buffer
struct W6 { u32 v[6]; };
__device__ __noinline__ W6 load_window_guarded(const uint8_t *text, u64 text_len, u64 off)
{
    W6 r;
#pragma unroll
    for (int w = 0; w < 6; ++w)
    {
        u32 v = 0;
        for (int b = 0; b < 4; ++b)
        {
            u64 o = off + (u64)(w * 4 + b);
            if (o < text_len)
                v |= (u32)text[o] << (8 * b);
        }
        r.v[w] = v;
    }
    return r;
}

// KIND: 1 -> m == 1 (SWAR), 4 -> 2..4 bytes (one word), 8 -> 5..8 bytes (two words), 9 -> m > 8 (filter+verify),
//       10 -> m > 16 without -c (round 5): the same 8-byte filter, but the tail of every candidate of a UNIT is verified at the
//       unit's end, one candidate per lane — one memory round trip per 32-KiB unit where KIND 9 pays one in every cell that
//       holds a candidate (a verify load waits behind the round's stream loads in the in-order queue): m = 128 0.69 -> see
//       profiles/r05_literal_sweep_32gib.txt
// MASKED: the last compared word is partial (m = 2,3 or 5,6,7), so its compare needs the byte mask.
// R: load rounds per UNIT.  A wave scans one contiguous R x 8 KiB unit at a time (statically dealt, see below), keeps the
// R x 8 per-lane hit masks in registers and publishes the unit's aggregate + staged hits; no wave waits for another.
template <int KIND, bool MASKED, bool CI, bool LINES, int R>
// >= 4 waves per SIMD (<= 128 VGPRs) for the plain variants: the allocator otherwise drifts to 137 and loses a wave.  The
// single byte with -c on 32-KiB units (32 deferred mask registers + the line bookkeeping) does not fit 128 without 16 bytes of
// scratch per lane; 3 blocks per CU on 168 registers without the spill were measured and are SLOWER (32 GiB, 1 % hits: 11.39
// against 9.81 ms, profiles/r04_literal_sweep_32gib.txt) — the fourth wave per SIMD is worth more than the 16 bytes.
#ifndef KG_LIT1_LINES_WAVES
#define KG_LIT1_LINES_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, (KIND == 1 && LINES && R == 4) ? KG_LIT1_LINES_WAVES : 4) void lit_scan(const LitArgs a)
{
    const u32 lane = lane_id();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool want_pos = !LINES && (a.flags & F_POS) != 0; // (-c never produces records: kg_scan.hip scan_literal)
    const bool ww = (a.flags & F_WW) != 0;
    const bool chain = want_pos || LINES;
    const u64 hi_match = (a.own_hi < a.text_len - a.m + 1) ? a.own_hi : (a.text_len - a.m + 1); // exclusive start bound
    constexpr u64 kUnitBytes = (u64)R * kSegBytes;

    // 2..8-byte kinds with 16-entry slots: NOTHING is stored while the wave streams.  Info words and staged offsets of up to
    // kPark units (30 tickets of 8) are parked in LDS (~9.8 KiB per wave: the four blocks of a CU use 154 of the 160 KiB) and
    // written out in one burst of ~12 store instructions; at 32 GiB a wave scans ~256 units, i.e. it stores twice.
    constexpr u32 kPark = (KIND == 4 || KIND == 8) ? 240u : 8u; // 30 tickets of 8 units: 4 x (240 x 40 + 240) B x 4 blocks = 154 KiB
    __shared__ u64 s_info[kWavesPerBlk][kPark];
    __shared__ __attribute__((aligned(16))) unsigned short s_slots[kWavesPerBlk][kPark * 16u];
    // (only the plain offsets-producing scan: -c measured 1 % slower parked, emit mode writes final records; m > 8 measured 3 %
    //  slower parked.  Round 5: the pool is shared out by the plan's slot size — 240 units of 16 entries, 56 of 64, 24 of 128 — so a
    //  plan whose slot has grown (`-i th`: 35 hits per unit) still stores nothing while it streams: its 37 M scattered 2-byte
    //  stores were why it ran at 0.52 of the roofline.)
    const bool park = (KIND == 4 || KIND == 8) && !LINES && want_pos && !a.emit_mode && a.stage_cap >= 16u && a.stage_cap <= 128u &&
                      (a.stage_cap & 7u) == 0u && (a.upt == 0u || a.upt >= 4u) && a.num_tiles < (1ull << 29);
    // units the pool holds: whole tickets only (s_tk maps a parked unit to its ticket)
    const u32 park_max = park ? (a.upt ? (kPark * 16u / a.stage_cap) / a.upt * a.upt : kPark * 16u / a.stage_cap) : kPark;
    // KIND 10: the unit's candidate lane-cells, {lane-cell of the unit (11 bits) | its 16-bit candidate mask << 16}, in position order
    constexpr u32 kCandCap = KIND == 10 ? 256u : 1u;
    __shared__ u32 s_cand[kWavesPerBlk][kCandCap];
    u32 n_park = 0;     // parked units (uniform); they are the units of the wave's last tickets, a.upt consecutive ones each
    __shared__ u32 s_tk[kWavesPerBlk][kPark / 4u]; // first unit of each parked ticket (4 or 8 units each)
    u64 park_first = 0;                            // static deal: the first parked unit, the others follow at the wave's stride
    auto flush_parked = [&]() __attribute__((always_inline)) {
        for (u32 i = lane; i < n_park; i += 64u)
        {
            const u64 u = a.upt ? (u64)s_tk[wave][i / a.upt] + (u64)(i % a.upt) : park_first + (u64)i * ((u64)gridDim.x * kWavesPerBlk);
            a.unitinfo[u] = s_info[wave][i];
            if (want_pos && (s_info[wave][i] & kUiCountMask))
            {
                uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(a.stage) + u * (u64)a.stage_cap);
                const uint4 *src = reinterpret_cast<const uint4 *>(&s_slots[wave][i * a.stage_cap]);
                const u32 c = (u32)(s_info[wave][i] & kUiCountMask), nv = ((c < a.stage_cap ? c : a.stage_cap) + 7u) >> 3;
                for (u32 q = 0; q < nv; ++q) // (16-entry slots: at most two)
                    dst[q] = src[q];
            }
        }
        n_park = 0;
    };
    u64 acc_total = 0; // wave-uniform accumulator
    // Every WAVE draws its own ticket for a.upt consecutive units (8 = 256 KiB on large texts; ~26 fetch-adds/us on the one
    // ticket word at 6.5 TB/s, a third of what it sustains) and runs on its own: no barrier, no LDS broadcast.  Round 1 drew one
    // ticket per 128 KiB tile for the four waves of a block and broadcast it behind a __syncthreads(): measured in isolation
    // (tools/ubench/read_ceiling.hip, 32 GiB) that skeleton alone caps a pure reader at 6.5 TB/s against 7.07 TB/s without
    // the barrier — every wave waits for the slowest of its tile, 262 144 times per scan.  A static interleaved deal (wave w
    // of block b scans units (i * grid + b) * 4 + w) was the first replacement; it leaves the waves that the store path
    // slows down (4.1, store placement) as stragglers: per-wave tickets measured 5.39 -> 5.17 ms counting and 8.5 -> 7.9 ms
    // on the single-byte workload on a box where that happens, and the same elsewhere.
    const u64 n_units = a.num_tiles * kWavesPerBlk;
    // (a.upt == 0, texts below 512 MiB: the static interleaved deal — such a scan lasts < 100 us, the ticket word would be its limit)
    u64 tk_next = (u64)blockIdx.x * kWavesPerBlk + wave, tk_end = a.upt ? tk_next : ~0ull;
    // Emit mode (re-scan of the units that overflowed their staging slot) draws GROUPS of 64 units: one info word per lane, a
    // ballot of the overflowed ones.  Walking the units one dependent load at a time cost the launch 1.6 ms per million units
    // even when three of them had overflowed (`-i sh`, 32 GiB).
    u64 em_base = 0, em_mask = 0;
    for (;;)
    {
        if (a.emit_mode)
        {
            while (!em_mask)
            {
                u64 g = 0;
                if (lane == 0)
                    g = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                em_base = rfl64(g) * 64ull;
                if (em_base >= n_units)
                    break;
                const u64 u = em_base + lane;
                em_mask = __ballot(u < n_units && (u32)(a.unitinfo[u] & kUiCountMask) > a.stage_cap);
            }
            if (!em_mask)
                break;
            tk_next = em_base + (u64)__builtin_ctzll(em_mask);
            em_mask &= em_mask - 1ull;
            tk_end = ~0ull;
        }
        else if (a.upt && tk_next == tk_end)
        {
            u64 tk = 0;
            if (lane == 0)
                tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk_next = rfl64(tk) * (u64)a.upt;
            tk_end = tk_next + a.upt;
            if (park && tk_next < n_units)
            {
                if (lane == 0)
                    s_tk[wave][n_park / a.upt] = (u32)tk_next;
            }
        }
        const u64 unit = tk_next;
        tk_next += a.upt ? 1ull : (u64)gridDim.x * kWavesPerBlk;
        if (unit >= n_units)
            break; // tickets (and the static stride) ascend: nothing is left for this wave
        const u64 ubase = a.anchor + unit * kUnitBytes;

        // KIND 1 (single byte, ~330 hits per unit at 1 %): every cell holds hits, so the masks are kept and written out in one go
        // at the unit's end (inline stores between the loads cost it 7 %); the sparse kinds write their rare hits where they find them
        constexpr bool kInline = KIND != 1;
        u32 M[kInline ? 1 : R][kInline ? 1 : kCells];
        u32 wcnt = 0;     // unit total (uniform)
        u32 n_cand = 0;   // KIND 10: parked candidate lane-cells of the unit (uniform)
        // KIND 10: verify the parked candidates, 64 lane-cells at a time, ONE candidate per lane and iteration (a lane-cell with two
        // candidates — a pattern that overlaps itself within 16 bytes — takes a second iteration), every lane's loads in flight
        // together; the survivors are ranked behind the unit's hits so far and staged (or, in emit mode, written out) in order
        auto verify_parked = [&]() __attribute__((always_inline)) {
            struct __attribute__((packed)) U64p { unsigned long long v; };
            typedef const __attribute__((address_space(4))) unsigned long long cu64;
            cu64 *pc = (cu64 *)(size_t)a.pat_chunks;
            const u32 last = a.n_chunks - 1u;
            for (u32 b0 = 0; b0 < n_cand; b0 += 64u)
            {
                const bool live = b0 + lane < n_cand;
                const u32 it = live ? s_cand[wave][b0 + lane] : 0u;
                const u32 rel0 = (it & 0xffffu) * 16u; // unit-relative offset of the lane-cell's first byte
                u32 rest = it >> 16, okm = 0;
                while (__ballot(rest != 0u))
                {
                    const bool act = rest != 0u;
                    const u32 k = act ? (u32)__builtin_ctz(rest) : 0u;
                    rest &= rest - 1u;
                    const u64 p = ubase + rel0 + k;
                    bool ok = act;
                    if (act)
                    {
                        u32 cL = 0, cR = 0;
                        if (ww)
                        {
                            if (p > 0)
                                cL = a.text[p - 1];
                            if (p + a.m < a.text_len)
                                cR = a.text[p + a.m];
                        }
                        const unsigned char *tp = a.text + p;
                        unsigned long long diff = 0;
                        // bytes 8..m-1 in independent 8-byte chunks, sixteen loads in flight, the last chunk re-anchored at m - 8
                        // (inside the match, hence inside the text); pattern words and -i letter masks as scalar loads
                        auto group = [&](auto width_c, u32 g0) {
                            constexpr u32 W = decltype(width_c)::value;
                            unsigned long long t[W];
#pragma unroll
                            for (u32 i = 0; i < W; ++i)
                            {
                                const u32 kk = g0 + i < last ? g0 + i : last;
                                const u32 q = 8u + 8u * kk < a.m - 8u ? 8u + 8u * kk : a.m - 8u;
                                t[i] = reinterpret_cast<const U64p *>(tp + q)->v;
                            }
#pragma unroll
                            for (u32 i = 0; i < W; ++i)
                            {
                                const u32 kk = g0 + i < last ? g0 + i : last;
                                const unsigned long long x = CI ? (t[i] | pc[a.n_chunks + kk]) : t[i];
                                diff |= x ^ pc[kk];
                            }
                        };
                        if (a.n_chunks <= 4u) // m <= 40
                            group(std::integral_constant<u32, 4>{}, 0u);
                        else if (a.n_chunks <= 8u) // m <= 72
                            group(std::integral_constant<u32, 8>{}, 0u);
                        else
                            for (u32 g0 = 0; g0 < a.n_chunks; g0 += 16u)
                                group(std::integral_constant<u32, 16>{}, g0);
                        ok = diff == 0;
                        if (ok && ww)
                        {
                            if (p > 0 && p != a.ww_exempt_left && is_wordc(cL))
                                ok = false;
                            else if (p + a.m < a.text_len && is_wordc(cR))
                                ok = false;
                        }
                    }
                    okm |= ok ? (1u << k) : 0u;
                }
                // rank the survivors (a lane holds at most 16) and write them out in position order
                const u32 c = __popc(okm);
                if (__ballot(c != 0u))
                {
                    const u32 idx0 = wcnt + wave_excl5(c);
                    wcnt += wave_sum5(c);
                    if (want_pos)
                    {
                        u32 idx = idx0, bits = okm;
                        if (!a.emit_mode)
                        {
                            unsigned short *slot = reinterpret_cast<unsigned short *>(a.stage) + unit * (u64)a.stage_cap;
                            while (bits)
                            {
                                const u32 k = __builtin_ctz(bits);
                                bits &= bits - 1u;
                                if (idx < a.stage_cap)
                                    slot[idx] = (unsigned short)(rel0 + k);
                                ++idx;
                            }
                        }
                        else
                        {
                            u64 o = a.offsets[unit] + idx;
                            while (bits)
                            {
                                const u32 k = __builtin_ctz(bits);
                                bits &= bits - 1u;
                                if (o < a.pos_cap)
                                {
                                    const u64 st = ubase + rel0 + k + a.global_base, en = st + a.m;
                                    *reinterpret_cast<uint4 *>(a.positions + 2 * o) =
                                        make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                                }
                                ++o;
                            }
                        }
                    }
                }
            }
            n_cand = 0;
        };
        LS wls{0, false, false, false};
        // ---- -c: the line bookkeeping of a unit (round 5).  A line that holds a match is counted at its FIRST match.  Inside a
        // lane's 16 bytes that is carry arithmetic on the hit / newline masks (l_cnt: first match behind a newline OF THIS LANE,
        // summed over the unit in a vector register, reduced once per unit); the part of a lane in front of its first newline
        // continues whatever line enters the lane, and "does the entering line already hold a match" is a CARRY CHAIN over the 64
        // lanes: generate = the lane ends with a match behind its last newline (or holds one and no newline), propagate = the lane
        // holds neither — one 64-bit scalar add resolves all 64 lanes, its carry-out is the state entering the next cell.  Round 4
        // resolved every cell with four ballots, a per-lane search for the nearest newline lane below and a ballot-plane sum:
        // 0.44 of the HBM roofline for a single byte at 1 % hits where the count-only scan runs at 0.85.
        u32 l_cnt = 0, l_hits = 0;                                  // per lane, summed at the unit's end
        u32 s_new = 0;                                              // lines opened by the head part of some lane (uniform)
        bool s_open = false, s_seen = false, s_head = false;        // open line entering the next cell | a newline seen | match before it
        auto line_cell = [&](u64 B_nl, u64 B_any, u64 B_head, u64 B_tail) __attribute__((always_inline)) {
            // (for a lane without a newline head == tail == any, by construction of both front ends)
            const u64 G = B_tail, P = ~(B_nl | B_any);
            const unsigned __int128 sum = (unsigned __int128)(G | P) + G + (s_open ? 1u : 0u);
            const u64 O = (u64)sum ^ P; // bit l: the line entering lane l already holds a match
            s_new += (u32)__popcll(B_head & ~O);
            if (!s_seen && B_nl)
            {
                const int f = __builtin_ctzll(B_nl);
                s_head = (((O | B_head) >> f) & 1ull) != 0ull;
                s_seen = true;
            }
            s_open = (u64)(sum >> 64) != 0ull;
        };

        // the rounds of a unit are a real loop for the sparse kinds (nothing is indexed by r any more): a quarter of the code
#pragma unroll(KIND == 1 ? R : 1)
        for (int r = 0; r < R; ++r)
        {
            const u64 seg = ubase + (u64)r * kSegBytes;
            const bool fast = seg + kSegBytes + (KIND == 9 ? 16 : 8) <= a.text_len;
            static_assert(KIND != 10 || !LINES, "the deferred verify has no per-cell hit masks for the line bookkeeping");
            const bool interior = seg >= a.own_lo && seg + kSegBytes <= hi_match &&
                                  (seg + kSegBytes <= a.excl_lo || seg >= a.excl_hi);

            uint4 d[kCells];
            uint4 after = make_uint4(0u, 0u, 0u, 0u); // the bytes behind the round (16 of them for the m = 9..16 verify)
            if (fast)
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                {
                    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + j * kWave));
                    d[j] = make_uint4(v.x, v.y, v.z, v.w);
                }
                if (KIND == 9)
                    after = *reinterpret_cast<const uint4 *>(a.text + seg + kSegBytes);
                else
                {
                    const uint2 t = *reinterpret_cast<const uint2 *>(a.text + seg + kSegBytes);
                    after.x = t.x; after.y = t.y;
                }
            }

#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u; // this lane's first byte
                u32 D[6];
                if (fast)
                {
                    D[0] = d[j].x; D[1] = d[j].y; D[2] = d[j].z; D[3] = d[j].w;
                    if (KIND != 1)
                    {
                        const u32 n0 = __shfl_down(D[0], 1), n1 = __shfl_down(D[1], 1);
                        u32 e0, e1;
                        if (j + 1 < kCells)
                        {
                            e0 = __builtin_amdgcn_readfirstlane(d[(j + 1 < kCells) ? j + 1 : j].x);
                            e1 = __builtin_amdgcn_readfirstlane(d[(j + 1 < kCells) ? j + 1 : j].y);
                        }
                        else
                        {
                            e0 = after.x;
                            e1 = after.y;
                        }
                        D[4] = (lane == 63u) ? e0 : n0;
                        D[5] = (lane == 63u) ? e1 : n1;
                    }
                }
                else
                {
                    const W6 g = load_window_guarded(a.text, a.text_len, lbase);
#pragma unroll
                    for (int w = 0; w < 6; ++w)
                        D[w] = g.v[w];
                }

                if (KIND == 1 && LINES && fast && interior && !ww)
                {
                    // ---- single byte, -c, interior cell: everything on the 0x80-per-byte flag words (no movemask, no 16-bit
                    //      masks): the lane's 16 bytes are ONE 128-bit integer with a flag at bit 8k + 7 for byte k ----
                    u32 HF[4], NF[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                    {
                        HF[w] = eq_bytes(CI ? (D[w] | a.l0) : D[w], a.p0);
                        NF[w] = eq_bytes(D[w], 0x0a0a0a0au);
                        l_hits += (u32)__popc(HF[w]);
                    }
                    const u64 B_any = __ballot((HF[0] | HF[1] | HF[2] | HF[3]) != 0u);
                    const u64 B_nl = __ballot((NF[0] | NF[1] | NF[2] | NF[3]) != 0u);
                    if (!B_any)
                    {
                        if (B_nl)
                        {
                            if (!s_seen)
                            {
                                s_head = s_open;
                                s_seen = true;
                            }
                            s_open = false;
                        }
                        continue;
                    }
                    typedef unsigned __int128 u128;
                    auto pack = [](const u32 (&x)[4]) -> u128 {
                        return ((u128)(((u64)x[3] << 32) | x[2]) << 64) | (((u64)x[1] << 32) | x[0]);
                    };
                    const u128 H = pack(HF), N = pack(NF), Hs = H | N;
                    const u128 firsts = H & ~(Hs - (N << 8)); // the first match behind each newline of the lane
                    l_cnt += (u32)(__popcll((u64)firsts) + __popcll((u64)(firsts >> 64)));
                    const u128 low = H & (Hs ^ (Hs - 1));      // the lowest flag, if it is a match: a match in front of the first newline
                    line_cell(B_nl, B_any, __ballot(low != 0), __ballot(N < H)); // N < H: the highest flag is a match
                    continue;
                }
                // newline mask before folding (folding never touches '\n')
                // interior cells only need "does this lane hold a newline" (a zero-byte test on D ^ '\n'); the exact
                // 16-bit mask (a multiply per dword) is computed where it is consumed: in cells that hold a hit, and in
                // boundary cells, where it has to be clipped to the owned window
                u32 NL = 0;
                auto exact_nl = [&]() -> u32 {
                    u32 v = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        v |= movemask4(eq_bytes(D[w], 0x0a0a0a0au)) << (4 * w);
                    return v;
                };
                auto has_nl = [&]() -> bool { // "does this lane hold a newline": a zero-byte test on D ^ '\n'
                    u32 z = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                    {
                        const u32 t = D[w] ^ 0x0a0a0a0au;
                        z |= (t - 0x01010101u) & ~t;
                    }
                    return (z & 0x80808080u) != 0u;
                };
                if (LINES && !interior)
                    NL = exact_nl();
                // -i: the text is NOT folded; a letter of the (folded) pattern is compared as (x | 0x20) == p, which
                // holds exactly for its two cases (C locale) — one OR per compared dword instead of a SWAR fold of
                // the whole window

                u32 m16 = 0;
                if (KIND == 1)
                {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        m16 |= movemask4(eq_bytes(CI ? (D[w] | a.l0) : D[w], a.p0)) << (4 * w);
                }
                else
                {
                    // unaligned dwords of the 24-byte window: A(k) = bytes [k, k+4)
                    auto A = [&](int k) -> u32 {
                        return ((k & 3) == 0) ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1], D[k >> 2], (u32)(k & 3));
                    };
                    // -i, common path: a SUPERSET filter on the window with 0x20 set in every byte, compared with the pattern's
                    // first word treated the same way — 6 ORs per cell where the exact compare ((x | letter mask) == p, the
                    // mask aligned to the candidate) costs one per position.  It also passes '@' for '`', '[' for '{' ...; the
                    // exact compare runs below, in the cells that hold a candidate.
                    u32 Dq[6];
#pragma unroll
                    for (int w = 0; w < 6; ++w)
                        Dq[w] = CI ? (D[w] | 0x20202020u) : D[w];
                    auto Aq = [&](int k) -> u32 {
                        return ((k & 3) == 0) ? Dq[k >> 2] : __builtin_amdgcn_alignbyte(Dq[(k >> 2) + 1], Dq[k >> 2], (u32)(k & 3));
                    };
                    const u32 p0q = CI ? (a.p0 | 0x20202020u) : a.p0;
                    u32 A0[16];
                    bool c[16];
                    u64 any = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                    {
                        A0[k] = Aq(k);
                        c[k] = (KIND == 4 && MASKED) ? (((A0[k] ^ p0q) & a.k0) == 0u) : (A0[k] == p0q);
                        any |= __ballot(c[k]);
                    }
                    // The superset IS the exact test when every compared byte of the pattern's first word is a letter
                    // ((x | 0x20) == (p | 0x20) <=> x is one of p's two cases): `-i th`, `-i the`, `-i sherlock` ... — the common
                    // case, and for 2-3-byte patterns (a candidate in nearly every 1-KiB cell) the difference between re-doing
                    // all 16 compares and not: m = 2 -i ran at 0.42 of the roofline (r03), the case-sensitive m = 2 at 0.73.
                    const bool cix0 = !CI || (a.l0 & a.k0 & 0x20202020u) == (a.k0 & 0x20202020u); // (uniform)
                    if (any) // wave-uniform: almost never taken for a selective 4-byte prefix
                    {
#pragma unroll
                        for (int k = 0; k < 16; ++k)
                        {
                            bool h = c[k];
                            if (CI && !cix0)
                            {
                                const u32 e0 = A(k) | a.l0; // exact: (x | 0x20 in the pattern's letter lanes) == folded pattern
                                h = h && ((KIND == 4 && MASKED) ? (((e0 ^ a.p0) & a.k0) == 0u) : (e0 == a.p0));
                            }
                            if (KIND >= 8)
                            {
                                const u32 a4 = (CI ? A(k + 4) : ((k < 12) ? A0[k + 4] : A(k + 4))) | (CI ? a.l1 : 0u);
                                h = h && ((KIND == 8 && MASKED) ? (((a4 ^ a.p1) & a.k1) == 0u) : (a4 == a.p1));
                            }
                            m16 |= h ? (1u << k) : 0u;
                        }
                    }
                }

                // window limits (boundary segments only)
                u32 nlm = NL;
                if (!interior)
                {
                    auto clip = [&](u64 lo, u64 hi) -> u32 { // bit mask of k with lo <= lbase+k < hi
                        u32 klo = lo > lbase ? (u32)((lo - lbase) < 16 ? (lo - lbase) : 16) : 0u;
                        u32 khi = hi > lbase ? (u32)((hi - lbase) < 16 ? (hi - lbase) : 16) : 0u;
                        return khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                    };
                    m16 &= clip(a.own_lo, hi_match) & ~clip(a.excl_lo, a.excl_hi);
                    if (LINES)
                        nlm &= clip(a.own_lo, a.own_hi);
                }

                if (KIND == 10)
                {
                    // candidates of the 8-byte filter: parked as lane-cell items, verified once per unit (verify_parked)
                    const u64 bm = __ballot(m16 != 0u);
                    if (bm)
                    {
                        const u32 nb = (u32)__popcll(bm);
                        if (n_cand + nb > kCandCap)
                            verify_parked(); // (a unit with more than 256 candidate lane-cells: what is parked goes first, in order)
                        if (m16)
                            s_cand[wave][n_cand + mbcnt64(bm)] = ((u32)(r * kCells + j) * kWave + lane) | (m16 << 16);
                        n_cand += nb;
                        m16 = 0;
                    }
                }
                // -w for m <= 8 (round 6): both neighbours of every start position of the lane lie in registers — the 24-byte window and,
                // for position 0, the last byte of the lane below (one shuffle per cell that holds a candidate) — and are picked out of
                // them by a select chain: ~35 VALU per candidate, no memory access (lane 0's position 0 excepted: the byte in front of
                // the cell).  The loop used to pay two dependent byte loads per candidate: `-w the` on word text (a candidate
                // every 130 bytes) counted at 2.5 TB/s where `tion` counts at 6.5.
                if (KIND != 10 && KIND != 9 && ww && __ballot(m16 != 0u))
                {
                    u32 d4 = D[4], d5 = D[5];
                    if (KIND == 1 && fast) // (the single-byte kinds do not build the window's tail: byte 16 is the lane above's first)
                    {
                        const u32 n0 = __shfl_down(D[0], 1);
                        const u32 e0 = (j + 1 < kCells) ? __builtin_amdgcn_readfirstlane(d[(j + 1 < kCells) ? j + 1 : j].x) : after.x;
                        d4 = (lane == 63u) ? e0 : n0;
                        d5 = 0u;
                    }
                    const u32 below = __shfl_up(D[3], 1); // byte 3: the byte in front of this lane's position 0
                    // byte i of the 28-byte sequence [below | D0 D1 D2 D3 | d4 d5]: the left neighbour of position k is byte k + 3, the right
                    // one byte k + m + 4
                    auto pick = [&](u32 i) -> u32 {
                        const u32 q = i >> 2;
                        const u32 w = q == 0u ? below : q == 1u ? D[0] : q == 2u ? D[1] : q == 3u ? D[2] : q == 4u ? D[3] : q == 5u ? d4 : d5;
                        return (w >> (8u * (i & 3u))) & 0xffu;
                    };
                    u32 rest = m16;
                    while (rest)
                    {
                        const u32 k = __builtin_ctz(rest);
                        rest &= rest - 1u;
                        const u64 p = lbase + k;
                        bool left = is_wordc(pick(k + 3u));
                        if (lane == 0u && k == 0u)
                            left = p > 0 && is_wordc(a.text[p - 1]);
                        if (p == a.ww_exempt_left) // (the first byte of a scalar tail call has no left neighbour: krep.c:5059-5097)
                            left = false;
                        // (a right neighbour behind the text reads as 0 in the guarded window, and a full round has its 8 bytes)
                        if (left || is_wordc(pick(k + a.m + 4u)))
                            m16 &= ~(1u << k);
                    }
                }
                // rare refinement on candidate lanes: verify the pattern tail (m > 8) and its -w
                if (KIND == 9 && __ballot(m16 != 0u))
                {
                    // (wave-uniform branch) the next lane's bytes 8..15 for the in-register verify of m = 9..16
                    const bool inreg = KIND == 9 && fast && a.m <= 16u;
                    u32 D6 = 0, D7 = 0;
                    if (inreg)
                    {
                        const u32 n2 = __shfl_down(D[2], 1), n3 = __shfl_down(D[3], 1);
                        u32 e2, e3;
                        if (j + 1 < kCells)
                        {
                            e2 = __builtin_amdgcn_readfirstlane(d[(j + 1 < kCells) ? j + 1 : j].z);
                            e3 = __builtin_amdgcn_readfirstlane(d[(j + 1 < kCells) ? j + 1 : j].w);
                        }
                        else
                        {
                            e2 = after.z;
                            e3 = after.w;
                        }
                        D6 = (lane == 63u) ? e2 : n2;
                        D7 = (lane == 63u) ? e3 : n3;
                    }
                    u32 rest = m16;
                    while (rest)
                    {
                        const u32 k = __builtin_ctz(rest);
                        rest &= rest - 1u;
                        const u64 p = lbase + k;
                        bool ok = true;
                        // -w neighbours: both byte loads are issued here, ahead of the pattern-tail loads, so that one
                        // memory round trip serves the whole candidate
                        u32 cL = 0, cR = 0;
                        if (ww)
                        {
                            if (p > 0)
                                cL = a.text[p - 1];
                            if (p + a.m < a.text_len)
                                cR = a.text[p + a.m];
                        }
                        if (KIND == 9 && inreg)
                        {
                            // m = 9..16 on a full round: bytes 8..15 of the candidate lie in the lane's own 16 bytes
                            // and the 16 of the next lane — compared in registers, no memory access at all
                            const u32 q = (k + 8u) >> 2, sh = k & 3u; // dword index 2..5 of the 32-byte window
                            auto sel = [&](u32 i) -> u32 {
                                return i == 2u ? D[2] : i == 3u ? D[3] : i == 4u ? D[4] : i == 5u ? D[5] : i == 6u ? D6 : D7;
                            };
                            const u32 w0 = sel(q), w1 = sel(q + 1u), w2 = q + 2u <= 7u ? sel(q + 2u) : 0u;
                            const u32 a8 = __builtin_amdgcn_alignbyte(w1, w0, sh), a12 = __builtin_amdgcn_alignbyte(w2, w1, sh);
                            ok = ((((CI ? (a8 | a.l2) : a8) ^ a.p2) & a.k2) | (((CI ? (a12 | a.l3) : a12) ^ a.p3) & a.k3)) == 0u;
                        }
                        else if (KIND == 9)
                        {
                            // bytes 8..m-1 in independent 8-byte chunks (all loads in flight together, no early
                            // exit: a byte loop paid one dependent global access per byte — 5.6 -> 3.3 TB/s at
                            // m = 9, 1.3 at m = 64); the last partial chunk is re-anchored at m - 8, which is
                            // inside [p, p + m) and therefore inside the text
                            struct __attribute__((packed)) U64p { unsigned long long v; };
                            const unsigned char *tp = a.text + p;
                            // groups of eight chunks: the eight text loads are issued back to back and waited for ONCE.  Round 1's
                            // counted loop cost one dependent memory round trip per chunk (two register pairs under the 128-VGPR
                            // cap of the old kernel; ~0.3 us of wave time each): 4.8 TB/s at m = 32, 4.2 at m = 64.  The
                            // pattern side comes as aligned 8-byte words (LitArgs::pat_chunks): uniform scalar loads, no
                            // vector registers.  A chunk index past the end repeats the last chunk (always inside the match).
                            unsigned long long diff = 0;
                            const u32 last = a.n_chunks - 1u;
                            // constant address space: tells the compiler the words are never written while the kernel runs,
                            // which is what lets it use s_load (SMEM) for them
                            typedef const __attribute__((address_space(4))) unsigned long long cu64;
                            cu64 *pc = (cu64 *)(size_t)a.pat_chunks;
                            auto group = [&](auto width_c, u32 g0) {
                                constexpr u32 W = decltype(width_c)::value;
                                unsigned long long t[W];
#pragma unroll
                                for (u32 i = 0; i < W; ++i)
                                {
                                    const u32 k = g0 + i < last ? g0 + i : last;
                                    const u32 q = 8u + 8u * k < a.m - 8u ? 8u + 8u * k : a.m - 8u;
                                    t[i] = reinterpret_cast<const U64p *>(tp + q)->v;
                                }
#pragma unroll
                                for (u32 i = 0; i < W; ++i)
                                {
                                    const u32 k = g0 + i < last ? g0 + i : last;
                                    // -i: the text is not folded — (x | letter mask) == pattern word is the C-locale compare of the
                                    // eight bytes (0x20 where the folded pattern holds a letter; the masks follow the pattern
                                    // words in the same scalar-loaded array): one OR where a SWAR fold cost ~16 VALU per chunk
                                    const unsigned long long x = CI ? (t[i] | pc[a.n_chunks + k]) : t[i];
                                    diff |= x ^ pc[k];
                                }
                            };
                            if (a.n_chunks <= 2u) // m <= 24
                                group(std::integral_constant<u32, 2>{}, 0u);
                            else if (a.n_chunks <= 4u) // m <= 40
                                group(std::integral_constant<u32, 4>{}, 0u);
                            else if (a.n_chunks <= 8u) // m <= 72
                                group(std::integral_constant<u32, 8>{}, 0u);
                            else // sixteen loads in flight, waited for once: two groups of eight cost m = 128 a second round trip per candidate
                                for (u32 g0 = 0; g0 < a.n_chunks; g0 += 16u)
                                    group(std::integral_constant<u32, 16>{}, g0);
                            ok = diff == 0;
                        }
                        if (ok && ww)
                        {
                            if (p > 0 && p != a.ww_exempt_left && is_wordc(cL))
                                ok = false;
                            else if (p + a.m < a.text_len && is_wordc(cR))
                                ok = false;
                        }
                        if (!ok)
                            m16 &= ~(1u << k);
                    }
                }

                const u64 anyhit = __ballot(m16 != 0u);
                if (!kInline)
                {
                    M[kInline ? 0 : r][kInline ? 0 : j] = m16;
                    if (anyhit)
                        wcnt += wave_sum5(__popc(m16));
                }
                else if (KIND != 10 && anyhit)
                {
                    // the cell holds hits (10 % of the cells at 1e-4 hits per byte): rank them inside the unit — running unit
                    // count + exclusive lane prefix, both from the same ballot bit-planes — and write them out HERE, in order:
                    // 16-bit unit-relative offsets into the unit's staging slot, or (emit mode: a unit that overflowed its slot
                    // is being re-scanned) final records at the unit's global offset.  Round 1 kept the R x 8 masks of the unit
                    // in 32 registers and walked them again at the unit's end; without them the offsets-producing scan runs at
                    // the count-only rate (the kernel is bound by how much load latency its registers let it cover, not by VALU).
                    const u32 c = __popc(m16);
                    u32 idx = wcnt, tot = 0;
                    auto plane = [&](int b) { // exclusive lane prefix and wave total from the same ballot
                        const u64 bm = __ballot((c >> b) & 1u);
                        idx += mbcnt64(bm) << b;
                        tot += (u32)__popcll(bm) << b;
                    };
                    plane(0);
                    plane(1);
                    if (__ballot(c > 3u)) // rare: some lane holds 4+ hits in its 16 bytes
                    {
                        plane(2);
                        plane(3);
                        plane(4);
                    }
                    wcnt += tot;
                    if (want_pos)
                    {
                        u32 rest = m16;
                        if (!a.emit_mode)
                        {
                            unsigned short *slot = park ? &s_slots[wave][n_park * a.stage_cap]
                                                        : reinterpret_cast<unsigned short *>(a.stage) + unit * (u64)a.stage_cap;
                            const u32 rel0 = (u32)(r * kCells + j) * kCellBytes + lane * 16u;
                            while (rest)
                            {
                                const u32 k = __builtin_ctz(rest);
                                rest &= rest - 1u;
                                if (idx < a.stage_cap)
                                    slot[idx] = (unsigned short)(rel0 + k);
                                ++idx;
                            }
                        }
                        else
                        {
                            u64 o = a.offsets[unit] + idx;
                            const u64 lb = lbase + a.global_base;
                            while (rest)
                            {
                                const u32 k = __builtin_ctz(rest);
                                rest &= rest - 1u;
                                if (o < a.pos_cap)
                                {
                                    const u64 st = lb + k, en = st + a.m;
                                    *reinterpret_cast<uint4 *>(a.positions + 2 * o) =
                                        make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                                }
                                ++o;
                            }
                        }
                    }
                }

                // A hit-free interior cell only matters through "was there a newline", and only while that changes the state: a
                // newline closes the open line and settles the unit's head — idempotent.  Once a newline has been seen and no line
                // is open, hit-free cells are not even looked at until the next hit (a wave-uniform test of two scalars).
                if (LINES && interior && !anyhit)
                {
                    if (!s_seen || s_open)
                        if (__ballot(has_nl()))
                        {
                            if (!s_seen)
                            {
                                s_head = s_open;
                                s_seen = true;
                            }
                            s_open = false;
                        }
                }
                else if (LINES)
                {
                    // the same on the 16-bit masks (cells with a hit of the sparse kinds, -w, boundary cells)
                    const u32 N = interior ? exact_nl() : nlm, H = m16, Hs = H | N;
                    l_cnt += (u32)__popc(H & ~(Hs - ((N << 1) & 0xffffu)));
                    line_cell(__ballot(N != 0u), anyhit, __ballot((H & (Hs ^ (Hs - 1u))) != 0u), __ballot(H > N));
                }
            }

        }

        if (KIND == 10 && n_cand)
            verify_parked();
        if (LINES)
        {
            u32 t = l_cnt, h = l_hits;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1)
            {
                t += __shfl_xor(t, o);
                if (KIND == 1)
                    h += __shfl_xor(h, o);
            }
            if (KIND == 1)
                wcnt += h; // (the flag-word cells count their hits per lane)
            wls = LS{t + s_new, s_seen, s_seen ? s_head : s_open, s_open};
        }
        acc_total += wcnt;
        if (!chain)
            continue;

        if (!a.emit_mode)
        {
            // ---- publish the unit: info word + staged, unit-ordered start offsets ----------------------
            if (lane == 0)
            {
                u64 info = (u64)wcnt;
                if (LINES)
                    info |= ls_bits(wls) | ((u64)(wls.cnt & kUiLineMask) << kUiLineShift);
                else if (wcnt)
                    info |= kLnHead | kLnTail;
                if (park)
                    s_info[wave][n_park] = info;
                else
                    a.unitinfo[unit] = info;
                if (want_pos && wcnt > a.stage_cap)
                {
                    atomicAdd(&a.ctr->overflow_units, 1ull);
                    atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
                }
            }
            if (park)
            {
                if (n_park == 0)
                    park_first = unit;
                if (++n_park == park_max)
                    flush_parked();
            }
            if (!kInline && want_pos && wcnt)
            {
                // unit-relative 16-bit offsets (a unit spans <= 32 KiB): 2 B staged per hit instead of 8
                unsigned short *slot = reinterpret_cast<unsigned short *>(a.stage) + unit * (u64)a.stage_cap;
                u32 out = 0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
#pragma unroll
                    for (int j = 0; j < kCells; ++j)
                    {
                        u32 m16 = M[kInline ? 0 : r][kInline ? 0 : j];
                        const u64 anyhit = __ballot(m16 != 0u);
                        if (!anyhit)
                            continue;
                        const u32 c = __popc(m16);
                        u32 idx = out, tot = 0;
                        auto plane = [&](int b) { // exclusive lane prefix and wave total from the same ballot
                            const u64 bm = __ballot((c >> b) & 1u);
                            idx += mbcnt64(bm) << b;
                            tot += (u32)__popcll(bm) << b;
                        };
                        plane(0);
                        plane(1);
                        if (__ballot(c > 3u)) // rare: some lane holds 4+ hits in its 16 bytes
                        {
                            plane(2);
                            plane(3);
                            plane(4);
                        }
                        out += tot;
                        const u32 rel0 = (u32)(r * kCells + j) * kCellBytes + lane * 16u;
                        while (m16)
                        {
                            const u32 k = __builtin_ctz(m16);
                            m16 &= m16 - 1u;
                            if (idx < a.stage_cap)
                                slot[idx] = (unsigned short)(rel0 + k);
                            ++idx;
                        }
                    }
                }
            }
        }
        else if (!kInline && want_pos && wcnt > a.stage_cap)
        {
            // ---- emit mode: this unit overflowed its staging slot; write its records in place -----------
            u64 out = a.offsets[unit];
            if (out < a.pos_cap)
            {
#pragma unroll
                for (int r = 0; r < R; ++r)
                {
#pragma unroll
                    for (int j = 0; j < kCells; ++j)
                    {
                        u32 m16 = M[kInline ? 0 : r][kInline ? 0 : j];
                        const u64 anyhit = __ballot(m16 != 0u);
                        if (!anyhit)
                            continue;
                        const u32 c = __popc(m16);
                        u64 idx = out + wave_excl5(c);
                        out += wave_sum5(c);
                        const u64 lb = ubase + (u64)(r * kCells + j) * kCellBytes + (u64)lane * 16u + a.global_base;
                        while (m16)
                        {
                            const u32 k = __builtin_ctz(m16);
                            m16 &= m16 - 1u;
                            if (idx < a.pos_cap)
                            {
                                const u64 st = lb + k, en = st + a.m;
                                *reinterpret_cast<uint4 *>(a.positions + 2 * idx) =
                                    make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                            }
                            ++idx;
                        }
                    }
                }
            }
        }
    }

    if (park && n_park)
        flush_parked();
    if (lane == 0 && acc_total && !a.emit_mode)
        atomicAdd(&a.ctr->total, acc_total);
}

// ---- launcher ----------------------------------------------------------------------------------
// The units are dealt out statically, so the grid is exactly the resident set: blocks per CU as the occupancy calculator
// reports them for the chosen instantiation, at most 4 (16 waves per CU; more waves measured slower), one phase, no tail of
// late blocks.  KREP_GPU_LIT_BLOCKS_PER_CU overrides (measurement aid).
template <typename K>
static u32 resident_blocks_per_cu(K kernel)
{
    static const u32 forced = [] { const char *e = getenv("KREP_GPU_LIT_BLOCKS_PER_CU"); return e && atoi(e) > 0 ? (u32)atoi(e) : 0u; }();
    if (forced)
        return forced;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlock, 0) != hipSuccess || n < 1)
    {
        (void)hipGetLastError();
        n = 4;
    }
    return (u32)std::min(n, 4); // measured: a fifth wave per SIMD costs the 8-byte literal 4-8 % (5.40 -> 5.59 ms at 32 GiB)
}
template <int KIND, bool MASKED, bool CI, bool LINES, int R>
static hipError_t launch4(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    static const u32 bpc = resident_blocks_per_cu(lit_scan<KIND, MASKED, CI, LINES, R>);
    const u32 grid = (u32)std::min<u64>(a.num_tiles, (u64)num_cu * bpc);
    hipLaunchKernelGGL((lit_scan<KIND, MASKED, CI, LINES, R>), dim3(grid ? grid : 1), dim3(kBlock), 0, st, a);
    return hipGetLastError();
}
template <int KIND, bool MASKED, bool CI, int R>
static hipError_t launch3(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    return (a.flags & F_LINES) ? launch4<KIND, MASKED, CI, true, R>(a, num_cu, st) : launch4<KIND, MASKED, CI, false, R>(a, num_cu, st);
}
template <int KIND, bool MASKED, bool CI>
static hipError_t launch2(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    return a.rounds == kRoundsBig ? launch3<KIND, MASKED, CI, kRoundsBig>(a, num_cu, st) : launch3<KIND, MASKED, CI, 1>(a, num_cu, st);
}
template <int KIND, bool MASKED>
static hipError_t launch1(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    return (a.flags & F_CI) ? launch2<KIND, MASKED, true>(a, num_cu, st) : launch2<KIND, MASKED, false>(a, num_cu, st);
}

bool literal_dma_eligible(const LitArgs &a);                                  // kg_literal_dma.hip
hipError_t launch_literal_dma(const LitArgs &a, u32 num_cu, hipStream_t st);
// a.num_tiles counts workgroup tiles of 4 x a.rounds x 8 KiB; a.rounds is 1 or kRoundsBig
hipError_t launch_literal(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    if (literal_dma_eligible(a)) // 2..8-byte patterns on 32-KiB units without -c: the LDS-DMA streaming kernel (kg_literal_dma.hip, round 6)
        return launch_literal_dma(a, num_cu, st);
    if (a.m == 1)
        return launch1<1, false>(a, num_cu, st);
    if (a.m < 4)
        return launch1<4, true>(a, num_cu, st);
    if (a.m == 4)
        return launch1<4, false>(a, num_cu, st);
    if (a.m < 8)
        return launch1<8, true>(a, num_cu, st);
    if (a.m == 8)
        return launch1<8, false>(a, num_cu, st);
    if (a.m > 16 && !(a.flags & F_LINES) && !getenv("KREP_GPU_LIT_NO_DEFER"))
    { // the tail verified once per unit (KIND 10; -c keeps the per-cell verify: its line bookkeeping wants the hits cell by cell)
        if (a.flags & F_CI)
            return a.rounds == kRoundsBig ? launch4<10, false, true, false, kRoundsBig>(a, num_cu, st) : launch4<10, false, true, false, 1>(a, num_cu, st);
        return a.rounds == kRoundsBig ? launch4<10, false, false, false, kRoundsBig>(a, num_cu, st) : launch4<10, false, false, false, 1>(a, num_cu, st);
    }
    return launch1<9, false>(a, num_cu, st);
}

} // namespace kg
