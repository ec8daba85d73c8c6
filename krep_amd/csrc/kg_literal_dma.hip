// kg_literal_dma.hip — the single-literal scan of 2..8-byte patterns on LDS-DMA streaming (round 6).
//
// Same job as kg::lit_scan<4 | 8> without -c (kg_literal.hip: all occurrences of one literal whose START lies in the owned window,
// counted, staged in unit order for the post-pass; replaces the hot loops of boyer_moore_search krep.c:1294-1382, memchr_short_search
// :4396-4500 and simd_sse42_search :4737-4866), same tickets, same staging slots and info words, same parked stores — but the
// haystack does not pass through vector registers on its way in:
//   * every wave owns a ring of two 8-KiB buffers in LDS and fills it with global_load_lds_dwordx4 ... nt (16 B per lane straight
//     into LDS, 1 KiB per instruction); the DMA of round g + 1 is issued before round g is consumed and retired by a COUNTED
//     s_waitcnt vmcnt (loads return in order; a store in the queue can only make the wait longer);
//   * a lane reads its 16 bytes with ds_read_b128 and the 8 bytes behind them with ds_read_b64 — the neighbour lane's bytes come
//     from LDS, not from a shuffle — and compares all 16 start positions in registers exactly as lit_scan does;
//   * the up-to-seven start positions whose window crosses the end of a round wait for the next round's first 8 bytes (a ticket's
//     last round: a 256-byte DMA piece behind the ticket) and are tested by lanes 0..7 then.
// Why: a bare LDS-DMA reader runs at 7.3 TB/s on this part where the register reader reaches 7.06, and with the literal compare
// at 6.98 (2 workgroups per CU) where the register version with a rolling prefetch reaches 6.77 and lit_scan 6.5-6.65
// (tools/ubench/read_ceiling.hip `ldsdma`, profiles/r06_ldsdma_ubench.txt).  No MFMA anywhere: an HBM-bound byte scan.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "kg_common.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

namespace {

constexpr u32 kDmaPark = 80;                 // parked units per wave (info word + 16 staged offsets each)
constexpr u32 kDmaRing = 2u * kSegBytes;     // two rounds
constexpr u32 kDmaTail = 256u;               // the DMA piece behind a ticket (4 B per lane)
constexpr u32 kDmaWaveLds = kDmaRing + kDmaTail; // dynamic LDS per wave; + 44 B per parked unit in static arrays: 20160 B per wave, two workgroups per CU

__device__ __forceinline__ u32 d_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 d_mbcnt(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u64 d_rfl64(u64 v)
{
    return ((u64)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)v);
}
__device__ __forceinline__ bool d_wordc(u32 c) { return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_'; }

struct W6d { u32 v[6]; };
__device__ __noinline__ W6d d_window_guarded(const uint8_t *text, u64 text_len, u64 off)
{
    W6d r;
#pragma unroll
    for (int w = 0; w < 6; ++w)
    {
        u32 v = 0;
        for (int b = 0; b < 4; ++b)
        {
            const u64 o = off + (u64)(w * 4 + b);
            if (o < text_len)
                v |= (u32)text[o] << (8 * b);
        }
        r.v[w] = v;
    }
    return r;
}

} // namespace

template <int KIND, bool MASKED, bool CI>
__global__ __launch_bounds__(kBlock, 2) void lit_scan_dma(const LitArgs a)
{
    // The ring is the ONLY thing in the dynamic LDS block and is read through inline ds_read (below): the compiler orders every LDS
    // access that may alias an LDS-DMA destination behind s_waitcnt vmcnt(0) — which would retire the next round's DMA before this
    // round is looked at (measured: 6.3 ms where the register kernel takes 5.3).  The parked stores live in static arrays of their own.
    extern __shared__ __attribute__((aligned(16))) uint8_t d_smem[];
    __shared__ u64 s_info_all[kWavesPerBlk][kDmaPark];
    __shared__ __attribute__((aligned(16))) unsigned short s_slots_all[kWavesPerBlk][kDmaPark * 16u];
    __shared__ u32 s_unit_all[kWavesPerBlk][kDmaPark];
    const u32 lane = d_lane();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t *ring = d_smem + wave * kDmaWaveLds;
    uint8_t *tailb = ring + kDmaRing;
    u64 *s_info = s_info_all[wave];
    unsigned short *s_slots = s_slots_all[wave];
    u32 *s_unit = s_unit_all[wave];
    const u32 ring_lds = (u32)(size_t)(__attribute__((address_space(3))) uint8_t *)ring; // LDS byte address of the wave's ring

    const bool want_pos = (a.flags & F_POS) != 0;
    const bool ww = (a.flags & F_WW) != 0;
    const u64 hi_match = (a.own_hi < a.text_len - a.m + 1) ? a.own_hi : (a.text_len - a.m + 1); // exclusive start bound
    constexpr u64 kUnitBytes = (u64)kRoundsBig * kSegBytes;
    const u64 n_units = a.num_tiles * kWavesPerBlk;

    // parked stores (kg_literal.hip): info words and staged offsets wait in LDS, written out in bursts
    const bool park = want_pos && a.stage_cap >= 16u && a.stage_cap <= 128u && (a.stage_cap & 7u) == 0u && n_units < (1ull << 32);
    const u32 park_max = park ? kDmaPark * 16u / a.stage_cap : 0u;
    u32 n_park = 0;
    auto flush_parked = [&]() __attribute__((always_inline)) {
        for (u32 i = lane; i < n_park; i += 64u)
        {
            const u64 u = s_unit[i];
            a.unitinfo[u] = s_info[i];
            if (s_info[i] & kUiCountMask)
            {
                uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(a.stage) + u * (u64)a.stage_cap);
                const uint4 *src = reinterpret_cast<const uint4 *>(&s_slots[i * a.stage_cap]);
                const u32 c = (u32)(s_info[i] & kUiCountMask), nv = ((c < a.stage_cap ? c : a.stage_cap) + 7u) >> 3;
                for (u32 q = 0; q < nv; ++q)
                    dst[q] = src[q];
            }
        }
        n_park = 0;
    };

    // tickets of a.upt consecutive units per wave, or (a.upt == 0) the static interleaved deal, one unit at a time
    const u32 tk_units = a.upt ? a.upt : 1u;
    u64 static_next = (u64)blockIdx.x * kWavesPerBlk + wave;
    auto next_ticket = [&]() -> u64 {
        if (a.upt)
        {
            u64 tk = 0;
            if (lane == 0)
                tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return d_rfl64(tk) * (u64)a.upt;
        }
        const u64 u = static_next;
        static_next += (u64)gridDim.x * kWavesPerBlk;
        return u;
    };
    // the DMA of one round: only a round that lies inside the text (the ragged end is read through guarded loads)
    auto issue_round = [&](u64 seg, u32 buf) -> bool {
        if (seg + kSegBytes > a.text_len)
            return false;
        const uint8_t *src = a.text + seg + lane * 16u;
        uint8_t *dst = ring + buf * kSegBytes;
#pragma unroll
        for (int j = 0; j < kCells; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * kCellBytes),
                                             (__attribute__((address_space(3))) void *)(dst + j * kCellBytes), 16, 0, 2 /* nt */);
        return true;
    };
    auto issue_tail = [&](u64 at) -> bool {
        if (at + kDmaTail > a.text_len)
            return false;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.text + at + lane * 4u),
                                         (__attribute__((address_space(3))) void *)tailb, 4, 0, 0);
        return true;
    };

    u64 acc_total = 0;
    u32 pf_cells = 0; // 1-KiB cells the prefilter let through (uniform; Counters::candidates — what the host's choice of kernel follows)
    u32 wcnt = 0; // hits of the unit being scanned (uniform)
    // a hit at unit-relative offset `rel` of `unit`, ranked idx: staged (parked or in the unit's slot)
    auto stage_hit = [&](u64 unit, u32 idx, u32 rel) __attribute__((always_inline)) {
        if (idx < a.stage_cap)
        {
            unsigned short *slot = park ? &s_slots[n_park * a.stage_cap] : reinterpret_cast<unsigned short *>(a.stage) + unit * (u64)a.stage_cap;
            slot[idx] = (unsigned short)rel;
        }
    };
    auto publish = [&](u64 unit) __attribute__((always_inline)) {
        acc_total += wcnt;
        if (want_pos)
        {
            if (lane == 0)
            {
                const u64 info = (u64)wcnt | (wcnt ? (kLnHead | kLnTail) : 0ull);
                if (park)
                {
                    s_info[n_park] = info;
                    s_unit[n_park] = (u32)unit;
                }
                else
                    a.unitinfo[unit] = info;
                if (wcnt > a.stage_cap)
                {
                    atomicAdd(&a.ctr->overflow_units, 1ull);
                    atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
                }
            }
            if (park && ++n_park == park_max)
                flush_parked();
        }
        wcnt = 0;
    };
    // exact compare of the window whose first word is A0 and second word A4 (the -i superset filter's second step included)
    auto exact = [&](u32 A0, u32 A4) -> bool {
        const u32 e0 = CI ? (A0 | a.l0) : A0;
        bool h = (KIND == 4 && MASKED) ? (((e0 ^ a.p0) & a.k0) == 0u) : (e0 == a.p0);
        if (KIND >= 8)
        {
            const u32 e4 = CI ? (A4 | a.l1) : A4;
            h = h && ((KIND == 8 && MASKED) ? (((e4 ^ a.p1) & a.k1) == 0u) : (e4 == a.p1));
        }
        return h;
    };
    auto owned = [&](u64 p) -> bool { return p >= a.own_lo && p < hi_match && !(p >= a.excl_lo && p < a.excl_hi); };
    auto word_ok = [&](u64 p) -> bool { // -w: is_whole_word_match (krep.h:312-319)
        if (p > 0 && p != a.ww_exempt_left && d_wordc(a.text[p - 1]))
            return false;
        if (p + a.m < a.text_len && d_wordc(a.text[p + a.m]))
            return false;
        return true;
    };
    // the start positions of a round whose window crosses its end: lanes 0..7 hold position seg_prev + 8184 + lane; c0, c1 = the
    // round's last 8 bytes, n0, n1 = the 8 bytes behind it
    auto boundary = [&](u64 seg_prev, u64 unit, u32 r_prev, u32 c0, u32 c1, u32 n0, u32 n1) __attribute__((always_inline)) {
        {
            // on the scalar unit: does one of the round's last 8 bytes equal the pattern's first byte at all?
            const u32 f = 0x01010101u * ((CI ? (a.p0 | 0x20u) : a.p0) & 0xffu);
            const u32 y0 = (CI ? (c0 | 0x20202020u) : c0) ^ f, y1 = (CI ? (c1 | 0x20202020u) : c1) ^ f;
            if (((((y0 - 0x01010101u) & ~y0) | ((y1 - 0x01010101u) & ~y1)) & 0x80808080u) == 0u)
                return;
        }
        const u32 q = lane & 7u, sh = q & 3u;
        const bool up = (q >> 2) != 0u;
        const u32 lo = up ? c1 : c0, mid = up ? n0 : c1, hi = up ? n1 : n0;
        const u32 A0 = __builtin_amdgcn_alignbyte(mid, lo, sh), A4 = __builtin_amdgcn_alignbyte(hi, mid, sh);
        const u64 p = seg_prev + (kSegBytes - 8u) + q;
        bool h = lane < 8u && q + a.m > 8u && exact(A0, A4) && owned(p); // (q + m <= 8: the window ended inside the round, tested there)
        if (ww && __ballot(h))
            h = h && word_ok(p);
        const u64 bm = __ballot(h);
        if (bm)
        {
            if (h && want_pos)
                stage_hit(unit, wcnt + d_mbcnt(bm), r_prev * kSegBytes + (kSegBytes - 8u) + q);
            wcnt += (u32)__popcll(bm);
        }
    };

    u64 u0 = next_ticket();
    bool dma_cur = u0 < n_units && issue_round(a.anchor + u0 * kUnitBytes, 0u);
    while (u0 < n_units)
    {
        const u32 nun = (u32)((n_units - u0 < (u64)tk_units) ? (n_units - u0) : (u64)tk_units), nr = nun * (u32)kRoundsBig;
        const u64 tbase = a.anchor + u0 * kUnitBytes;
        u64 u_next = ~0ull;
        bool tail_ok = false;
        u32 c0 = 0, c1 = 0;      // the last 8 bytes of the round in front (uniform)
        bool deferred = false;   // ... whose crossing positions are still to be tested
#pragma unroll 1
        for (u32 g = 0; g < nr; ++g)
        {
            const u64 seg = tbase + (u64)g * kSegBytes, unit = u0 + (g >> 2);
            const u32 r = g & 3u;
            // ---- the next round's DMA (the next ticket's first round behind this ticket's last), then the counted wait for this one
            bool dma_nx = false, tail_now = false;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the buffer about to be refilled has been read)
            if (g + 1u < nr)
                dma_nx = issue_round(seg + kSegBytes, (g + 1u) & 1u);
            else
            {
                u_next = next_ticket();
                if (u_next < n_units)
                    dma_nx = issue_round(a.anchor + u_next * kUnitBytes, (g + 1u) & 1u);
            }
            if (g + 2u == nr)
                tail_ok = tail_now = issue_tail(tbase + (u64)nr * kSegBytes);
            if (dma_cur)
            {
                if (dma_nx && tail_now)
                    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else if (dma_nx)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (tail_now)
                    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const bool interior = seg >= a.own_lo && seg + kSegBytes <= hi_match && (seg + kSegBytes <= a.excl_lo || seg >= a.excl_hi);
            // ---- the round's bytes: all LDS reads first (one LDS latency per round, not per cell)
            uint4 vv[kCells];
            uint2 nn[kCells];
            if (dma_cur)
            {
                typedef u32 v4 __attribute__((ext_vector_type(4)));
                typedef u32 v2 __attribute__((ext_vector_type(2)));
                const u32 addr = ring_lds + (g & 1u) * kSegBytes + lane * 16u;
                v4 x[kCells];
                v2 y[kCells];
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[j]) : "v"(addr), "n"(j * (int)kCellBytes));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(y[j]) : "v"(addr), "n"(j * (int)kCellBytes + 16)); // (lane 63 of the last cell: not this round's bytes — masked below)
                }
                // (the wait names every destination: nothing that uses one can be scheduled in front of it)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(y[0]), "+v"(y[1]),
                               "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7])
                             :
                             : "memory");
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                {
                    vv[j] = make_uint4(x[j].x, x[j].y, x[j].z, x[j].w);
                    nn[j] = make_uint2(y[j].x, y[j].y);
                }
            }
            else
            {
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                {
                    const W6d w = d_window_guarded(a.text, a.text_len, seg + (u64)j * kCellBytes + (u64)lane * 16u);
                    vv[j] = make_uint4(w.v[0], w.v[1], w.v[2], w.v[3]);
                    nn[j] = make_uint2(w.v[4], w.v[5]);
                }
            }
            // ---- the crossing positions of the round in front, now that its next 8 bytes are here
            if (deferred)
            {
                boundary(seg - kSegBytes, u0 + ((g - 1u) >> 2), (g - 1u) & 3u, c0, c1, __builtin_amdgcn_readfirstlane(vv[0].x),
                         __builtin_amdgcn_readfirstlane(vv[0].y));
                deferred = false;
                if (r == 0u)
                    publish(unit - 1u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                const u32 D[6] = {vv[j].x, vv[j].y, vv[j].z, vv[j].w, nn[j].x, nn[j].y};
                auto A = [&](int k) -> u32 {
                    return ((k & 3) == 0) ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1], D[k >> 2], (u32)(k & 3));
                };
                // -i: a SUPERSET filter on the window with 0x20 set in every byte (kg_literal.hip); the exact compare in the cells that
                // hold a candidate
                u32 Dq[6];
#pragma unroll
                for (int w = 0; w < 6; ++w)
                    Dq[w] = CI ? (D[w] | 0x20202020u) : D[w];
                auto Aq = [&](int k) -> u32 {
                    return ((k & 3) == 0) ? Dq[k >> 2] : __builtin_amdgcn_alignbyte(Dq[(k >> 2) + 1], Dq[k >> 2], (u32)(k & 3));
                };
                const u32 p0q = CI ? (a.p0 | 0x20202020u) : a.p0;
                if (a.prefilter) // (uniform) a rare first byte: one zero-byte test per dword says whether the cell can start a match at all
                {
                    u32 z = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                    {
                        const u32 y = Dq[w] ^ a.prefilter;
                        z |= (y - 0x01010101u) & ~y;
                    }
                    if (!__ballot((z & 0x80808080u) != 0u))
                        continue;
                    ++pf_cells;
                }
                bool c[16];
                u64 any = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                {
                    const u32 x = Aq(k);
                    c[k] = (KIND == 4 && MASKED) ? (((x ^ p0q) & a.k0) == 0u) : (x == p0q);
                    any |= __ballot(c[k]);
                }
                if (!any) // wave-uniform: almost always for a selective first word
                    continue;
                u32 m16 = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (__ballot(c[k])) // (uniform: the exact compare only at the start positions some lane's first word matched at)
                        m16 |= (c[k] && exact(A(k), A(k + 4))) ? (1u << k) : 0u;
                // lane 63 of a DMA round's last cell does not have the bytes behind the round: its crossing positions are deferred
                if (dma_cur && j == kCells - 1 && lane == 63u)
                    m16 &= (1u << (17u - a.m)) - 1u;
                if (!interior)
                {
                    auto clip = [&](u64 lo, u64 hi) -> u32 { // bit mask of k with lo <= lbase + k < hi
                        const u32 klo = lo > lbase ? (u32)((lo - lbase) < 16 ? (lo - lbase) : 16) : 0u;
                        const u32 khi = hi > lbase ? (u32)((hi - lbase) < 16 ? (hi - lbase) : 16) : 0u;
                        return khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                    };
                    m16 &= clip(a.own_lo, hi_match) & ~clip(a.excl_lo, a.excl_hi);
                }
                if (ww && __ballot(m16 != 0u))
                {
                    // -w in registers (kg_literal.hip): the neighbours picked out of the lane's window and the lane below's last dword; the
                    // byte in front of lane 0 and the bytes behind a DMA round (lane 63 of its last cell) come from memory
                    const u32 below = __shfl_up(D[3], 1);
                    auto pick = [&](u32 i) -> u32 {
                        const u32 q = i >> 2;
                        const u32 w = q == 0u ? below : q == 1u ? D[0] : q == 2u ? D[1] : q == 3u ? D[2] : q == 4u ? D[3] : q == 5u ? D[4] : D[5];
                        return (w >> (8u * (i & 3u))) & 0xffu;
                    };
                    const bool edge = dma_cur && j == kCells - 1 && lane == 63u;
                    u32 rest = m16;
                    while (rest)
                    {
                        const u32 k = __builtin_ctz(rest);
                        rest &= rest - 1u;
                        const u64 p = lbase + k;
                        bool bad;
                        if (edge || (lane == 0u && k == 0u))
                            bad = !word_ok(p);
                        else
                            bad = (p != a.ww_exempt_left && d_wordc(pick(k + 3u))) || d_wordc(pick(k + a.m + 4u));
                        if (bad)
                            m16 &= ~(1u << k);
                    }
                }
                if (!__ballot(m16 != 0u))
                    continue;
                // rank the cell's hits behind the unit's hits so far (ballot bit-planes + mbcnt) and stage them in order
                const u32 cnt = __popc(m16);
                u32 idx = wcnt, tot = 0;
                auto plane = [&](int b) {
                    const u64 bm = __ballot((cnt >> b) & 1u);
                    idx += d_mbcnt(bm) << b;
                    tot += (u32)__popcll(bm) << b;
                };
                plane(0);
                plane(1);
                if (__ballot(cnt > 3u))
                {
                    plane(2);
                    plane(3);
                    plane(4);
                }
                wcnt += tot;
                if (want_pos)
                {
                    const u32 rel0 = (r * (u32)kCells + (u32)j) * kCellBytes + lane * 16u;
                    u32 rest = m16;
                    while (rest)
                    {
                        const u32 k = __builtin_ctz(rest);
                        rest &= rest - 1u;
                        stage_hit(unit, idx++, rel0 + k);
                    }
                }
            }
            // ---- what the next round (or the ticket's tail piece) has to finish
            if (dma_cur)
            {
                c0 = __builtin_amdgcn_readlane(vv[kCells - 1].z, 63);
                c1 = __builtin_amdgcn_readlane(vv[kCells - 1].w, 63);
                deferred = true;
            }
            if (g + 1u == nr)
            {
                if (deferred)
                {
                    u32 n0 = 0, n1 = 0;
                    if (tail_ok)
                    { // (the piece was issued in front of the next ticket's first round: the wait above covered it)
                        typedef u32 v2 __attribute__((ext_vector_type(2)));
                        v2 t;
                        const u32 taddr = ring_lds + kDmaRing;
                        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(taddr) : "memory");
                        n0 = __builtin_amdgcn_readfirstlane(t.x);
                        n1 = __builtin_amdgcn_readfirstlane(t.y);
                    }
                    else
                    {
                        const W6d w = d_window_guarded(a.text, a.text_len, seg + kSegBytes);
                        n0 = __builtin_amdgcn_readfirstlane(w.v[0]);
                        n1 = __builtin_amdgcn_readfirstlane(w.v[1]);
                    }
                    boundary(seg, unit, r, c0, c1, n0, n1);
                    deferred = false;
                }
                publish(unit);
            }
            else if (!deferred && r == 3u)
                publish(unit); // (a guarded round ended the unit: nothing is pending)
            dma_cur = dma_nx;
        }
        u0 = u_next;
    }
    if (park && n_park)
        flush_parked();
    if (lane == 0 && acc_total)
        atomicAdd(&a.ctr->total, acc_total);
    if (lane == 0 && pf_cells)
        atomicAdd(&a.ctr->candidates, (u64)pf_cells);
}

// The look before the first launch (kg_scan.hip lit_pass): in how many 1-KiB cells of a sample does the prefilter's byte occur at all?
// A cell that holds it costs lit_scan_dma the full 16-position compare, and at two workgroups per CU that work is not hidden; the register
// kernel (kg_literal.hip, more waves per SIMD) is the faster one from ~4 cells in 10 on (measured: kg_scan.hip lit_pass).  One wave per cell, 16 B per lane.
__global__ __launch_bounds__(256) void dma_byte_look(const uint8_t *text, u64 lo, u32 n_cells, u32 b4, u32 fold, unsigned long long *out)
{
    const u32 lane = d_lane();
    const u32 wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    u32 cnt = 0;
    for (u32 c = wave; c < n_cells; c += n_waves)
    {
        const uint4 v = *reinterpret_cast<const uint4 *>(text + lo + (u64)c * kCellBytes + lane * 16u);
        const u32 d[4] = {v.x, v.y, v.z, v.w};
        u32 z = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            const u32 y = (d[w] | fold) ^ b4;
            z |= (y - 0x01010101u) & ~y;
        }
        if (__ballot((z & 0x80808080u) != 0u))
            ++cnt;
    }
    if (lane == 0 && cnt)
        atomicAdd(out, (u64)cnt);
}
hipError_t launch_dma_byte_look(const uint8_t *text, uint64_t lo, uint32_t n_cells, uint32_t prefilter, bool ci, unsigned long long *out, hipStream_t st)
{
    hipLaunchKernelGGL(dma_byte_look, dim3(256), dim3(256), 0, st, text, (u64)lo, n_cells, prefilter, ci ? 0x20202020u : 0u, out);
    return hipGetLastError();
}

// ---- launcher ----------------------------------------------------------------------------------
std::atomic<uint64_t> g_lit_dma_launches{0};

template <int KIND, bool MASKED, bool CI>
static hipError_t dma_launch3(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    constexpr u32 lds = kWavesPerBlk * kDmaWaveLds; // dynamic part (the rings); the parked stores are static
    constexpr int kMaxDev = 64;
    static std::atomic<bool> granted[kMaxDev];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev || !granted[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&lit_scan_dma<KIND, MASKED, CI>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess)
            return e;
        if (dev >= 0 && dev < kMaxDev)
            granted[dev].store(true, std::memory_order_release);
    }
    static const u32 bpc = [] { const char *e = getenv("KREP_GPU_LIT_DMA_BLOCKS_PER_CU"); return e && atoi(e) > 0 ? (u32)atoi(e) : 2u; }();
    const u32 grid = (u32)std::min<u64>(a.num_tiles, (u64)num_cu * bpc);
    hipLaunchKernelGGL((lit_scan_dma<KIND, MASKED, CI>), dim3(grid ? grid : 1), dim3(kBlock), lds, st, a);
    g_lit_dma_launches.fetch_add(1, std::memory_order_relaxed);
    return hipGetLastError();
}
template <int KIND, bool MASKED>
static hipError_t dma_launch2(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    return (a.flags & F_CI) ? dma_launch3<KIND, MASKED, true>(a, num_cu, st) : dma_launch3<KIND, MASKED, false>(a, num_cu, st);
}

// does this launch take the LDS-DMA kernel?  2..8-byte patterns, 32-KiB units, no -c, not the emit-mode re-scan
bool literal_dma_eligible(const LitArgs &a)
{
    // (and only with the rare-first-byte prefilter: without it the compare's VALU work is not hidden at two waves per SIMD —
    //  `-i sherlock`: 6.26 ms against 5.21 ms for the register kernel, which keeps such patterns)
    return !getenv("KREP_GPU_LIT_NO_DMA") && a.prefilter != 0u && a.m >= 2 && a.m <= 8 && a.rounds == (u32)kRoundsBig && !(a.flags & F_LINES) && !a.emit_mode &&
           (a.upt >= 4 || getenv("KREP_GPU_LIT_DMA_ALL")) && a.text_len >= 4u * kSegBytes;
    // (tickets of >= 4 units — texts from ~24 GiB — : with one-unit tickets / the static deal of smaller texts every fourth round ends
    //  a ticket, and the register kernel is as fast or faster: 8 GiB 1.42 against 1.36 ms.  $KREP_GPU_LIT_DMA_ALL: the tests' switch.)
}
hipError_t launch_literal_dma(const LitArgs &a, u32 num_cu, hipStream_t st)
{
    if (a.m < 4)
        return dma_launch2<4, true>(a, num_cu, st);
    if (a.m == 4)
        return dma_launch2<4, false>(a, num_cu, st);
    if (a.m < 8)
        return dma_launch2<8, true>(a, num_cu, st);
    return dma_launch2<8, false>(a, num_cu, st);
}

} // namespace kg
