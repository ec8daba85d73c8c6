// kg_single.hip — memchr_search (krep.c:3891-4041) with its match_position_t records in ONE pass over the text (gfx950).
//
// A single byte at 1 % density (BASELINE config 3: 343.6 M records per 32 GiB) is the worst case of the two-pass scheme of
// kg_literal.hip + kg_post.hip: the scan stages 0.7 GB of unit-relative offsets, publishes 1 M info words, and a second
// kernel turns the staging into 5.5 GB of records — 5.8-6.9 ms of scan (depending on where the driver places the staging
// stores) plus 1.4 ms of gather for a workload whose bytes (34.4 GB read + 5.5 GB written) fit into ~6 ms.
// Here the records are written by the wave that found the hits, at their FINAL index, and nothing else is stored:
//   * a wave draws a ticket of kUpt units (128 KiB; one counter), scans it exactly as lit_scan<1> does (SWAR byte equality, 16 start
//     positions per lane in registers) and puts every hit, already ranked inside the ticket (ballot bit-planes + v_mbcnt),
//     as a 16-bit unit-relative offset into its LDS ring — nothing goes to memory while it streams;
//   * it publishes the ticket's hit count (one 8-byte store) and goes on to the NEXT ticket;
//   * one RESOLVER wave — whichever wave 0 of a block arrives first claims the role (see "Progress" below); it does not scan —
//     turns the published counts into their exclusive prefix, the global index of each ticket's first record, ticket by ticket
//     as far as the run of published counts extends;
//   * only after scanning that next ticket (~40 us later) does the wave pick up its previous ticket's prefix — by then the
//     resolver has long passed it, nobody waits — and writes that ticket's records from the ring: coalesced 1-KiB stores,
//     10 KiB per ticket, ascending with the ticket number.
// This is the chained scan that round 1 measured at 2.9 TB/s (every tile waited for its predecessor's status) with the wait
// taken out of it: counts flow forward through one wave, the consumers are a whole ticket period behind.
// Progress without assumptions about residency (round 4, ADVICE r03 — the round-3 build could wait circularly when fewer
// waves than expected were really running: a GPU shared with another process, a partitioned device, a grid that is not
// fully resident):
//   * tickets (128 KiB) come from ONE counter, so the set of drawn tickets is always a PREFIX of the ticket space, and
//     whoever drew a ticket is a wave that is running;
//   * the resolver is whichever wave-0-of-a-block arrives first (an atomic claim) — a running wave by construction — and it
//     publishes prefixes AS FAR AS THE READY RUN EXTENDS, ticket by ticket, not in steps of 512;
//   * a wave waits for the prefix of ticket p only while holding tickets it drew AFTER p.
//   Let m be the smallest ticket whose count is not published.  Everything below m is ready, so every prefix up to m is
//   published.  If m is drawn, its holder is scanning it or waits for the prefix of an EARLIER ticket, which is published: it
//   goes on.  If m is not drawn, every wave that waits does so for a ticket below m: published.  No wait is circular, whatever
//   subset of the grid runs.  (tests/test_gpu_literal.py::test_single_byte_one_pass_with_a_starved_grid forces 1-, 2- and 3-block
//   grids over 8 192 tickets.)
// A ticket with more hits than the ring holds raises ctr->overflow_units (the scan still COUNTS: the resolver's running sum is
// the total).  Six shapes (template parameters UPT = 32-KiB units per ticket, RING = 16-bit entries per wave, WPE = waves per
// SIMD; the table is in front of launch_shape below): from 128-KiB tickets / 8 KiB of ring / 16 waves per CU for up to ~1.2 % hits
// (BASELINE config 3) to 32-KiB tickets / 32 KiB / 4 waves for up to ~20 % (80 tickets per microsecond is what one counter gives:
// fine while the records, 16 bytes per hit, are most of the traffic).  The host picks the shape from the density the
// first scan of a plan counted (kg_scan.hip, lit_pass) and falls back to the two-pass kernels beyond that.
// SET: up to four needle bytes instead of one — a dictionary of single-byte patterns (`-e e -e t`) is this scan with a set
// (aho_corasick_search reports such matches in text order, one per position: the records are memchr_search's).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "kg_common.h"
#include "kg_internal.h"
#include "kg_tickets.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

// Ticket size (A/B builds: python -m krep_amd.build --variant x -DKG_S1_UPT=2).  ONE ticket counter: the drawn tickets are
// always a prefix of the ticket space, which is what the progress argument above needs.  Measured in one process (32 GiB,
// tools/ab_bench.py, profiles/r04_single_byte_tickets.txt): 128-KiB tickets from the one counter 6.58 ms against 6.55 for
// round 3's 64-KiB tickets from eight interleaved counters (which could wait circularly on a starved grid); 64-KiB tickets from
// one counter 8.06 (80 fetch-adds/us on one word: the dequeue limit of the part); TWO consecutive 64-KiB tickets per draw
// 11.6 ms and four 7 s — a ticket that is drawn long before it is scanned holds back the prefix of everything behind it, and
// the waves end up scanning in lock step.  A resolver window of 1024 or 2048 tickets instead of 512 gained nothing.
#ifndef KG_S1_UPT
#define KG_S1_UPT 4
#endif
constexpr u32 kUptStd = KG_S1_UPT;            // units (32 KiB each) per ticket: 128 KiB (at most 4: the flush tells units apart by three bounds)
static_assert(kUptStd >= 1 && kUptStd <= 4, "flush() derives a record's unit from three boundaries");
constexpr u32 kRingStd = 1024u * kUptStd;     // 16-bit entries per wave: the ticket being scanned + the one waiting
constexpr u32 kRingDense = 8192u;             // ... of the dense shapes (16 KiB per wave: 2 workgroups per CU; kRingMid = 12 KiB: 3)
constexpr u32 kRingDensest = 16384u;          // ... of the densest one (32 KiB per wave: one workgroup per CU)
constexpr u64 kReady = kTkReady;              // (the resolver itself: kg_tickets.h, shared with kg_ac_tiny.hip)
constexpr u64 kUnitBytes1 = (u64)kRoundsBig * kSegBytes;

__device__ __forceinline__ u32 s_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 s_mbcnt(u64 m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }
__device__ __forceinline__ u64 s_rfl64(u64 v)
{
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 s_eq_bytes(u32 x, u32 c4)
{
    const u32 y = x ^ c4;
    const u32 t = (y & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    return ~(t | y | 0x7f7f7f7fu);
}
__device__ __forceinline__ u32 s_movemask4(u32 t) { return (((t >> 7) * 0x00204081u) >> 21) & 0xfu; }

// the (at most one) round that touches the end of the buffer: byte-wise, bounds-checked, out of line
__device__ __noinline__ uint4 s_load16_guarded(const uint8_t *text, u64 text_len, u64 off)
{
    u32 w[4] = {0, 0, 0, 0};
    for (u32 b = 0; b < 16; ++b)
        if (off + b < text_len)
            w[b >> 2] |= (u32)text[off + b] << (8 * (b & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// MULTI (round 5): a literal of 2..8 bytes instead of one byte — the same ring, resolver and deferred stores, the match test is
// lit_scan's (kg_literal.hip: the 16 start positions of a lane compared in registers against the pattern's first word, and its
// second word where the pattern is longer than four bytes); the 8 bytes behind a lane come from its right neighbour, behind the
// round from one extra 8-byte load.  For DENSE short literals with records (`-i sh`: 35 hits per 32-KiB unit, ` a`: 180), whose
// staging slots overflow and whose text the two-pass kernels then read twice: chosen by lit_pass from the density a scan counted.
// BDRAW (round 5, the shapes with 32-KiB tickets): ONE draw per WORKGROUP — thread 0 fetches kWavesPerBlk consecutive tickets, a
// barrier hands wave w ticket base + w, the four waves start their tickets together.  The one counter gives ~65 draws per
// microsecond (same-address atomics serialise in one L2 channel): a million 32-KiB tickets are 15 ms whatever the text holds.
// The drawn tickets stay a prefix of the ticket space and a wave's tickets ascend, so the progress argument above holds as it
// stands; a wave at the barrier holds nothing unpublished.  The resolver's own workgroup cannot use the barrier (its wave 0 never
// gets there): its other three waves keep drawing one ticket each from the same counter — any mix of draws leaves a prefix — so a
// device that runs a single workgroup of the grid still makes progress.
__device__ __forceinline__ bool s_wordc(u32 c) { return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_'; }

template <bool CI, u32 kUpt, u32 kRing, int WPE, bool SET, bool MULTI = false, bool BDRAW = false, bool WW = false>
__global__ __launch_bounds__(kBlock, WPE) void single_fused(const LitArgs a, u64 *__restrict__ agg, u64 *__restrict__ pref,
                                                            const u64 n_tickets)
{
    static_assert(!MULTI || !SET, "a byte set is a single-byte scan");
    static_assert(kUpt >= 1 && kUpt <= 4, "ticket shape");
    // ring positions wrap by a mask where the ring is a power of two, by one compare-and-subtract where it is not (x < 2 kRing)
    auto rwrap = [](const u32 x) -> u32 {
        if constexpr ((kRing & (kRing - 1u)) == 0u)
            return x & (kRing - 1u);
        else
            return x >= kRing ? x - kRing : x;
    };
    const u32 lane = s_lane();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 n_units = a.num_tiles * kWavesPerBlk;

    // ---- the resolver: counts -> exclusive prefixes, in ticket order.  Claimed by the first wave 0 of any block that gets
    // here (a.ctr->pad[0], zeroed with the counters before the launch): a wave that RUNS, whatever part of the grid is resident.
    bool resolver = false;
    if (wave == 0)
    {
        u64 r = 1;
        if (lane == 0)
            r = __hip_atomic_fetch_add(&a.ctr->pad[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        resolver = s_rfl64(r) == 0ull;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned short s_ring[]; // [kWavesPerBlk][kRing] (dynamic: the densest shape asks for 128 KiB)
    u64 *s_draw = reinterpret_cast<u64 *>(s_ring + (size_t)kWavesPerBlk * kRing); // BDRAW: [0] the workgroup's ticket base, [1] "the resolver's workgroup"
    bool bmode = false; // BDRAW: this workgroup draws as one (uniform over the workgroup; false in the resolver's)
    if constexpr (BDRAW)
    {
        if (threadIdx.x == 0)
            s_draw[1] = resolver ? 1ull : 0ull;
        __syncthreads();
        bmode = s_draw[1] == 0ull;
    }
    if (resolver)
    {
        tk_resolve(agg, pref, n_tickets, a.ctr, lane);
        return;
    }
    unsigned short *ring = s_ring + (size_t)wave * kRing;
    const u64 hi_match = a.own_hi < a.text_len - a.m + 1 ? a.own_hi : a.text_len - a.m + 1; // exclusive start bound

    // the ticket whose records are still in the ring (uniform)
    bool pend = false;
    u64 pend_t = 0;
    u32 pend_at = 0, pend_cnt = 0, pend_c0 = 0, pend_c1 = 0, pend_c2 = 0;
    u32 wp = 0; // ring write position (uniform, modulo kRing at use)
    bool overflowed = false;

    auto flush = [&]() __attribute__((always_inline)) {
#ifdef KG_S1_NOWAIT // (ablation build: records at a made-up index, nobody waits for the resolver)
        const u64 first = pend_t * 1400ull;
#else
        const u64 first = tk_wait_prefix(pref, pend_t, a.ctr, lane);
#endif
        const u64 tbase = a.anchor + pend_t * (u64)kUpt * kUnitBytes1 + a.global_base;
        const u32 b1 = pend_c0, b2 = pend_c0 + pend_c1, b3 = pend_c0 + pend_c1 + pend_c2;
        // Lanes are mapped to GLOBAL record indices rounded down to 8 (= one 128-byte line), so every store instruction covers
        // whole lines except at the two ends of the ticket's run; the stores are NON-TEMPORAL: next to the streaming reads that
        // is worth 9 % of the kernel (tools/ubench/write_ceiling.hip: 34.4 GB read + 5.5 GB written in 7.30 ms with plain
        // stores, 6.64 ms non-temporal; line-aligned runs another 0.1-0.3 ms).
        const u32 pad = (u32)(first & 7ull);
        for (u32 g = lane; g < pend_cnt + pad; g += 64u)
        {
            if (g < pad)
                continue;
            const u32 i = g - pad;
            const u32 off = ring[rwrap(pend_at + i)];
            const u32 unit = (i >= b1 ? 1u : 0u) + (i >= b2 ? 1u : 0u) + (i >= b3 ? 1u : 0u);
            const u64 idx = first + i;
#ifdef KG_S1_NOSTORE // (ablation build: everything but the record stores)
            if (idx == ~0ull)
#else
            if (idx < a.pos_cap)
#endif
            {
                const u64 st = tbase + (u64)unit * kUnitBytes1 + off, en = st + (MULTI ? (u64)a.m : 1ull);
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 rec = {(u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32)};
#ifdef KG_S1_PLAIN_STORES
                *reinterpret_cast<u32x4 *>(a.positions + 2 * idx) = rec;
#else
                __builtin_nontemporal_store(rec, reinterpret_cast<u32x4 *>(a.positions + 2 * idx));
#endif
            }
        }
        pend = false;
    };

    // ---- the rounds of a ticket (16 x 8 KiB, contiguous).  The first round of the NEXT ticket is requested before the previous
    // ticket's records are written, so the flush runs under load latency.  (A rolling prefetch of every next round into a
    // second register set — 123 instead of 59 VGPRs — was built and measured: 7.23 vs 7.20 ms, nothing; the kernel is bound by
    // the mixed read/write traffic, not by exposed load latency: without its record stores it runs at 5.4 ms.)  Loads whose
    // target may not exist are issued unconditionally at a fallback address (uniform select: a conditional load makes the
    // compiler wait vmcnt(0) at the join); a round that touches the end of the buffer is loaded byte-wise.
    constexpr u32 kRoundsPerTicket = kUpt * kRoundsBig;
    constexpr u64 kTicketBytes = (u64)kUpt * kUnitBytes1;
    const u64 scan_end = a.anchor + n_units * kUnitBytes1 < a.text_len ? a.anchor + n_units * kUnitBytes1 : a.text_len;
    const u64 fb = (a.text_len - kSegBytes) & ~15ull; // a round that is always inside the buffer (the host guarantees text_len >= 8 KiB):
                                                      // where a prefetch with nothing to fetch points, so that it stays unconditional
    auto issue = [&](uint4 (&X)[kCells], u64 seg) __attribute__((always_inline)) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
#pragma unroll
        for (int j = 0; j < kCells; ++j)
        {
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + j * kWave));
            X[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    u32 cnt = 0;  // hits of the ticket so far (uniform)
    u32 at = 0, room = 0;
    auto process = [&](uint4 (&X)[kCells], u64 seg, bool have, u32 rr) __attribute__((always_inline)) {
        if (!have)
        {
            if (seg + kSegBytes <= a.text_len)
                issue(X, seg);
            else
            {
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                    X[j] = s_load16_guarded(a.text, a.text_len, seg + (u64)j * kCellBytes + (u64)lane * 16u);
            }
        }
        const bool interior = seg >= a.own_lo && seg + kSegBytes <= hi_match;
        // MULTI: the 8 bytes behind the round (the last lane of its last cell looks that far); zeros behind the end of the text
        u32 aft0 = 0, aft1 = 0;
        if (MULTI)
        {
            const u64 q = seg + kSegBytes;
            if (q + 8 <= a.text_len)
            {
                const uint2 t2 = *reinterpret_cast<const uint2 *>(a.text + q);
                aft0 = __builtin_amdgcn_readfirstlane(t2.x);
                aft1 = __builtin_amdgcn_readfirstlane(t2.y);
            }
            else
            {
                u32 w2[2] = {0, 0};
                for (u32 b = 0; b < 8; ++b)
                    if (q + b < a.text_len)
                        w2[b >> 2] |= (u32)a.text[q + b] << (8 * (b & 3u));
                aft0 = __builtin_amdgcn_readfirstlane(w2[0]);
                aft1 = __builtin_amdgcn_readfirstlane(w2[1]);
            }
        }
#pragma unroll
        for (int j = 0; j < kCells; ++j)
        {
            const u32 D[4] = {X[j].x, X[j].y, X[j].z, X[j].w};
            u32 m16 = 0;
            if (MULTI)
            {
                // the lane's 16 bytes + the 8 behind them: the right neighbour's first two dwords, for the last lane the next cell's
                // (the round's cells are all in registers) or the bytes behind the round
                const u32 n0 = __shfl_down(D[0], 1), n1 = __shfl_down(D[1], 1);
                const u32 e0 = j + 1 < kCells ? __builtin_amdgcn_readfirstlane(X[j + 1 < kCells ? j + 1 : j].x) : aft0;
                const u32 e1 = j + 1 < kCells ? __builtin_amdgcn_readfirstlane(X[j + 1 < kCells ? j + 1 : j].y) : aft1;
                const u32 W[6] = {D[0], D[1], D[2], D[3], lane == 63u ? e0 : n0, lane == 63u ? e1 : n1};
                auto A = [&](int k) -> u32 {
                    return ((k & 3) == 0) ? W[k >> 2] : __builtin_amdgcn_alignbyte(W[(k >> 2) + 1], W[k >> 2], (u32)(k & 3));
                };
                // (x | l) == p on the compared bytes: exact with and without -i (l = 0x20 under the pattern's letters, p folded)
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    m16 |= ((((CI ? (A(k) | a.l0) : A(k)) ^ a.p0) & a.k0) == 0u) ? (1u << k) : 0u;
                if (a.m > 4u && __ballot(m16 != 0u)) // (uniform) a second word to compare, where the first one matched
                {
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if ((((CI ? (A(k + 4) | a.l1) : A(k + 4)) ^ a.p1) & a.k1) != 0u)
                            m16 &= ~(1u << k);
                }
                // -w (round 6; is_whole_word_match krep.h:312-319): both neighbours of a start position lie in the 24-byte window or — position
                // 0 — in the lane below's last dword, and are picked out of registers (kg_literal.hip); only lane 0's position 0 asks memory
                // (WW: a template parameter — behind a run-time flag this block cost the plain one-pass writer 1.3-1.6 %: ` a` 6.40 -> 6.49 ms
                //  in one process, register allocation — so the -w instantiations are twins)
                if (WW && __ballot(m16 != 0u))
                {
                    const u32 below = __shfl_up(D[3], 1);
                    const u64 lb = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                    auto pick = [&](u32 i) -> u32 {
                        const u32 q = i >> 2;
                        const u32 w = q == 0u ? below : q == 1u ? W[0] : q == 2u ? W[1] : q == 3u ? W[2] : q == 4u ? W[3] : q == 5u ? W[4] : W[5];
                        return (w >> (8u * (i & 3u))) & 0xffu;
                    };
                    u32 rest = m16;
                    while (rest)
                    {
                        const u32 k = __builtin_ctz(rest);
                        rest &= rest - 1u;
                        const u64 p = lb + k;
                        bool left = s_wordc(pick(k + 3u));
                        if (lane == 0u && k == 0u)
                            left = p > 0 && s_wordc(a.text[p - 1]);
                        if (p == a.ww_exempt_left)
                            left = false;
                        if (left || s_wordc(pick(k + a.m + 4u)))
                            m16 &= ~(1u << k);
                    }
                }
            }
            else if (SET)
            { // a byte SET: the equality flags of its (<= 4) needles OR-ed before the one movemask per dword
#pragma unroll
                for (int w = 0; w < 4; ++w)
                {
                    u32 f = s_eq_bytes(CI ? (D[w] | a.set_l[0]) : D[w], a.set_p[0]);
                    if (a.set_n > 1u) f |= s_eq_bytes(CI ? (D[w] | a.set_l[1]) : D[w], a.set_p[1]);
                    if (a.set_n > 2u) f |= s_eq_bytes(CI ? (D[w] | a.set_l[2]) : D[w], a.set_p[2]);
                    if (a.set_n > 3u) f |= s_eq_bytes(CI ? (D[w] | a.set_l[3]) : D[w], a.set_p[3]);
                    m16 |= s_movemask4(f) << (4 * w);
                }
            }
            else
            {
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    m16 |= s_movemask4(s_eq_bytes(CI ? (D[w] | a.l0) : D[w], a.p0)) << (4 * w);
            }
            if (!interior)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                const u32 klo = a.own_lo > lbase ? (u32)((a.own_lo - lbase) < 16 ? (a.own_lo - lbase) : 16) : 0u;
                const u32 khi = hi_match > lbase ? (u32)((hi_match - lbase) < 16 ? (hi_match - lbase) : 16) : 0u;
                m16 &= khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
            }
            if (__ballot(m16 != 0u))
            {
                // rank inside the ticket: hits so far (uniform) + exclusive lane prefix, from the same ballot bit-planes
                const u32 c = __popc(m16);
                u32 idx = cnt, tot = 0;
                auto plane = [&](int b) {
                    const u64 bm = __ballot((c >> b) & 1u);
                    idx += s_mbcnt(bm) << b;
                    tot += (u32)__popcll(bm) << b;
                };
                plane(0);
                plane(1);
                if (__ballot(c > 3u))
                {
                    plane(2);
                    plane(3);
                    plane(4);
                }
                cnt += tot;
                const u32 rel0 = ((rr & (kRoundsBig - 1u)) * kCells + (u32)j) * kCellBytes + lane * 16u; // unit-relative
                u32 rest = m16;
                while (rest)
                {
                    const u32 k = __builtin_ctz(rest);
                    rest &= rest - 1u;
                    if (idx < room)
                        ring[rwrap(at + idx)] = (unsigned short)(rel0 + k);
                    ++idx;
                }
            }
        }
    };

    // ONE ticket counter (a.ctr->ticket): what has been drawn is always a prefix of the ticket space, and a wave's tickets
    // ascend — the two facts the progress argument in the header rests on.
    bool blk_live = true; // BDRAW: the workgroup's last draw still lay inside the ticket space (uniform over the workgroup)
    auto draw = [&]() __attribute__((always_inline)) -> u64 {
        u64 tk = 0;
        if (BDRAW && bmode)
        {
            __syncthreads(); // every wave has read the previous base
            if (threadIdx.x == 0)
                s_draw[0] = __hip_atomic_fetch_add(&a.ctr->ticket, (u64)kWavesPerBlk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const u64 base = s_draw[0];
            blk_live = base < n_tickets;
            tk = base + wave;
        }
        else
        {
            if (lane == 0)
                tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk = s_rfl64(tk);
        }
        return tk < n_tickets ? tk : ~0ull;
    };
    uint4 A[kCells];
    u64 t = draw();
    bool haveA = t < n_tickets && a.anchor + t * kTicketBytes + kSegBytes <= a.text_len;
    issue(A, haveA ? a.anchor + t * kTicketBytes : fb);
    while (bmode ? blk_live : t < n_tickets)
    {
        const bool mine = t < n_tickets; // (BDRAW: the last draw of a workgroup may leave some of its waves without a ticket)
        const u64 tbase = a.anchor + (mine ? t : 0ull) * kTicketBytes;
        at = wp;
        room = kRing - pend_cnt; // entries this ticket may use while the previous one is still parked
        cnt = 0;
        u32 c0 = 0, c1 = 0, c2 = 0, before = 0, units_done = 0; // hits of the ticket's first three units
#pragma unroll 1
        for (u32 rr = 0; rr < (mine ? kRoundsPerTicket : 0u); ++rr)
        {
            const u64 seg = tbase + (u64)rr * kSegBytes;
            if (seg >= scan_end)
                break;
            process(A, seg, haveA, rr);
            haveA = false;
            if ((rr & (kRoundsBig - 1u)) == kRoundsBig - 1u)
            { // a unit is complete (selects, not an indexed array: that would live in scratch)
                const u32 cu = cnt - before, uu = rr / kRoundsBig;
                before = cnt;
                units_done = uu + 1u;
                c0 = uu == 0u ? cu : c0;
                c1 = uu == 1u ? cu : c1;
                c2 = uu == 2u ? cu : c2;
            }
        }
        {
            // a ticket cut short by the end of the text: what was found behind the last complete unit belongs to the next one
            const u32 cu = cnt - before;
            c0 = units_done == 0u ? cu : c0;
            c1 = units_done == 1u ? cu : c1;
            c2 = units_done == 2u ? cu : c2;
        }
        if (cnt > room)
            overflowed = true; // too dense for the ring: counted, not recorded — the host re-runs the two-pass kernels
        // publish the count BEFORE waiting for anything; the ticket drawn next is behind every ticket this wave has parked
        if (lane == 0 && mine)
            __hip_atomic_store(&agg[t], (u64)cnt | kReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the next ticket, and its first round on the way, before the previous ticket's records are written
        const u64 tn = draw();
        haveA = tn < n_tickets && a.anchor + tn * kTicketBytes + kSegBytes <= a.text_len;
        issue(A, haveA ? a.anchor + tn * kTicketBytes : fb);
        if (pend)
            flush();
        if (cnt && cnt <= room)
        {
            pend = true;
            pend_t = t;
            pend_at = at;
            pend_cnt = cnt;
            pend_c0 = c0;
            pend_c1 = c1;
            pend_c2 = c2;
            wp = rwrap(at + cnt);
        }
        else
            pend_cnt = 0;
        t = tn;
    }
    if (pend)
        flush();
    if (overflowed && lane == 0)
        atomicAdd(&a.ctr->overflow_units, 1ull);
}

int g_s1_force_grid = 0; // test hook: at most this many blocks (0 = auto)
// grid = the resident blocks of the instantiation x CUs
template <bool CI, u32 UPT, u32 RING, int WPE, bool SET, bool MULTI, bool BDRAW = false, bool WW = false>
static hipError_t launch_fused(const LitArgs &a, u64 *agg, u64 *pref, u64 n_tickets, u32 num_cu, hipStream_t st)
{
    constexpr size_t kLds = (size_t)kWavesPerBlk * RING * sizeof(unsigned short) + (BDRAW ? 16u : 0u);
    // per DEVICE: the granted-LDS attribute and the resident blocks per CU of this instantiation are properties of the
    // device the launch goes to — asked once per device, not once per process (multi-device searches launch from one process) and
    // not on every launch
    static std::atomic<u32> s_bpc[64]; // 0 = not asked yet
    int dev = 0;
    (void)hipGetDevice(&dev);
    const u32 slot = (u32)dev < 64u ? (u32)dev : 63u;
    u32 bpc = s_bpc[slot].load(std::memory_order_acquire);
    if (!bpc)
    {
        if (kLds > 64 * 1024) // more than 64 KiB of dynamic LDS has to be asked for
        {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&single_fused<CI, UPT, RING, WPE, SET, MULTI, BDRAW, WW>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
            if (e != hipSuccess)
                return e;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, single_fused<CI, UPT, RING, WPE, SET, MULTI, BDRAW, WW>, kBlock, kLds) != hipSuccess || n < 1)
        {
            (void)hipGetLastError();
            n = 1;
        }
        bpc = (u32)std::min(n, 4);
        s_bpc[slot].store(bpc, std::memory_order_release);
    }
    // one block more than the tickets need is never useful; at least 1 scanning wave next to the resolver
    const u64 want = (n_tickets + kWavesPerBlk - 1) / kWavesPerBlk + 1;
    u32 grid = (u32)std::max<u64>(1, std::min<u64>(want, (u64)num_cu * bpc));
    if (g_s1_force_grid > 0) // test hook: a starved grid (krep_gpu_debug_force_single_grid)
        grid = std::min<u32>(grid, (u32)g_s1_force_grid);
    hipLaunchKernelGGL((single_fused<CI, UPT, RING, WPE, SET, MULTI, BDRAW, WW>), dim3(grid), dim3(kBlock), kLds, st, a, agg, pref, n_tickets);
    return hipGetLastError();
}

// The shapes, by the density they hold (hits per byte; half a ring per ticket with a margin for clustering):
//   0: 128-KiB tickets,  8-KiB rings, 16 waves per CU   <= ~1.2 %   (BASELINE config 3)
//   1:  64-KiB tickets, 12-KiB rings, 12 waves per CU   <= ~3.7 %   (round 5: 6144 entries, wrap by compare; a single byte at
//   2:  64-KiB tickets, 16-KiB rings,  8 waves per CU   <= ~5 %      3.1-3.5 % ran 8-11 % faster than in shape 2 — the dense
//   3:  32-KiB tickets, 12-KiB rings, 12 waves per CU   <= ~7.5 %    shapes are bound by their occupancy, not by the stores:
//   4:  32-KiB tickets, 16-KiB rings,  8 waves per CU   <= ~10 %     profiles/r05_dense_one_pass_ablation.txt); 3 and 4 draw
//                                                                    their tickets once per workgroup (BDRAW: e t 2.32 -> 2.67 TB/s)
//   5:  32-KiB tickets, 32-KiB rings,  4 waves per CU   <= ~20 %    (by then the records are 3 bytes per byte of text)
constexpr u32 kRingMid = 6144u;
static u32 shape_upt(int shape) { return shape == 0 ? kUptStd : (shape <= 2 ? 2u : 1u); }
static u32 shape_ring(int shape) { return shape == 0 ? kRingStd : (shape == 5 ? kRingDensest : ((shape & 1) ? kRingMid : kRingDense)); }
uint64_t single_fused_tickets(uint64_t n_units, int shape) { return (n_units + shape_upt(shape) - 1) / shape_upt(shape); }
uint64_t single_fused_scratch_words(uint64_t n_tickets) { return 2 * n_tickets; } // counts | prefixes
// the densest text (hits per byte) a shape's ring is sure to hold
double single_fused_max_density(int shape)
{
    return 0.4 * (double)shape_ring(shape) / ((double)shape_upt(shape) * (double)kUnitBytes1);
}

template <bool CI, bool SET, bool MULTI = false, bool WW = false>
static hipError_t launch_shape(const LitArgs &a, u64 *agg, u64 *pref, u64 n_tickets, u32 num_cu, int shape, hipStream_t st)
{
    static_assert(kFusedShapeMax == 5, "six shapes");
    if (shape == 0) return launch_fused<CI, kUptStd, kRingStd, 4, SET, MULTI, false, WW>(a, agg, pref, n_tickets, num_cu, st);
    if (shape == 1) return launch_fused<CI, 2u, kRingMid, 3, SET, MULTI, false, WW>(a, agg, pref, n_tickets, num_cu, st);
    if (shape == 2) return launch_fused<CI, 2u, kRingDense, 2, SET, MULTI, false, WW>(a, agg, pref, n_tickets, num_cu, st);
#ifndef KG_S1_NO_BDRAW // (A/B switch)
    constexpr bool kB = true; // 32-KiB tickets: one draw per workgroup
#else
    constexpr bool kB = false;
#endif
    if (shape == 3) return launch_fused<CI, 1u, kRingMid, 3, SET, MULTI, kB, WW>(a, agg, pref, n_tickets, num_cu, st);
    if (shape == 4) return launch_fused<CI, 1u, kRingDense, 2, SET, MULTI, kB, WW>(a, agg, pref, n_tickets, num_cu, st);
    // (not the 32-KiB-ring shape: with ONE workgroup per CU its four waves would scan and flush in lock step — measured 959 against 1000 GB/s)
    return launch_fused<CI, 1u, kRingDensest, 1, SET, MULTI, false, WW>(a, agg, pref, n_tickets, num_cu, st);
}

hipError_t launch_single_fused(const LitArgs &a, unsigned long long *d_agg, unsigned long long *d_pref, uint64_t n_tickets,
                               uint32_t num_cu, int shape, hipStream_t st)
{
    const bool ci = (a.flags & F_CI) != 0, set = a.set_n != 0u;
    if (!set && a.m > 1u)
    {
        if (a.m > 8u)
            return hipErrorInvalidValue;
        if (a.flags & F_WW)
            return ci ? launch_shape<true, false, true, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st)
                      : launch_shape<false, false, true, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st);
        return ci ? launch_shape<true, false, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st)
                  : launch_shape<false, false, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st);
    }
    if (set)
        return ci ? launch_shape<true, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st)
                  : launch_shape<false, true>(a, d_agg, d_pref, n_tickets, num_cu, shape, st);
    return ci ? launch_shape<true, false>(a, d_agg, d_pref, n_tickets, num_cu, shape, st)
              : launch_shape<false, false>(a, d_agg, d_pref, n_tickets, num_cu, shape, st);
}

} // namespace kg
