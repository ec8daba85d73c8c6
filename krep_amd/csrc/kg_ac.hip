// kg_ac.hip — multi-pattern literal scan for gfx950, replacing aho_corasick_search
// (aho_corasick.c:299-466) behind the same search_func_t boundary.
//
// What the reference computes: for every text index i (ascending), every pattern that ENDS at i,
// longest first (it walks the goto/fail automaton and then the whole fail chain,
// aho_corasick.c:353-431), duplicates of a pattern string once per copy.  That set is a pure
// function of the bytes text[i-Lmax+1 .. i], so the sequential automaton walk is not needed to
// reproduce it.  MI355X-first formulation:
//   * FILTER (every byte): one bit table in LDS, indexed by the 5-bit classes (b & 31) of the last four bytes
//     (2^20 bits = 128 KiB; a pattern shorter than 4 bytes sets every class of the bytes in front of it).  A lane
//     turns its 20 bytes into a 100-bit class stream once; a position then costs a funnel shift, the dword address,
//     one ds_read_b32 and two shifts.  With STRIDE == 2 only the even positions are looked up (see below).
//     Positions whose bit is clear cannot end a pattern.  The haystack is read once with the same coalesced
//     16 B/lane loads as the literal scan, the next round streaming into the registers the current one vacates.
//   * VERIFY (candidates only, ~0.2 % of positions for 1000 random patterns): one candidate per lane, 64 at a time
//     from the unit's candidate bitmap in LDS.  The exact last 4 bytes select an entry of a sparse hash table that carries the
//     next <= 12 bytes of the reversed-trie chain and the depths at which patterns end (kg_ac_common.h); 1-3-byte
//     patterns are exact bitmap lookups.  Two dependent accesses per candidate; the level-by-level walk of the
//     REVERSED-pattern trie (edges in an open-addressing table, L2-resident) remains for what that cannot express.
//     The reference order (longest first at one end index) is the depth mask read from the top.
//   * ORDER: same as the literal kernel — unit-local ranks, staging slot, info word, post-pass.
// -w, -c (line counting), max_count and start-offset ownership are applied exactly as in the literal
// kernel.  A dense DFA in LDS is impossible for the benchmark set (8605 states x 256 x 2 B = 4.4 MB
// against 160 KiB), which is why the LDS holds the filter, not the automaton.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_ac_common.h"
#include "kg_ac_tables.h"
#include "kg_internal.h"

namespace kg {

// SHORT: the dictionary holds 1-3-byte patterns (wildcard-expanded in the filter table, exact bitmaps in the verifier).
// STRIDE == 2: the filter tests the EVEN text positions only.  The table then holds, besides every
// pattern's final 4-gram (a match ends at the tested position t), the 4-gram one byte earlier (the match ends at t + 1;
// for a 4-byte pattern that gram has an unknown first byte: all 32 classes are set).  Half the LDS lookups — the
// bank-conflict wall of 4.2 — and half the lookup VALU; a candidate verifies both ends, with both probes in flight.
#ifndef KG_AC_LINES_ROLL_CELLS
#define KG_AC_LINES_ROLL_CELLS 4 // cells (1 KiB each) of the next round a -c scan prefetches (A/B: krep_amd/build.py --variant x -DKG_AC_LINES_ROLL_CELLS=8)
#endif
// ANCH (round 6, kg_ac_anchor.hip; only with STRIDE == 2, no short patterns, no -c): the table holds rarity-chosen ANCHOR grams.
// A candidate's exact anchor gram names the offsets k at which patterns END behind it; those ends are marked in a second bitmap
// of the unit — same layout as the candidate bitmap: bit b <-> the END pair (2b + 1, 2b + 2), in the KiB that otherwise parks a
// ticket's stores — and the marked pairs, not the candidates, are what the end-anchored verifier looks at, in position order:
// ranking, staging and emission as before.
// ANCH == 2: the filter's index holds FIVE classes — the class of the byte in front of the 4-gram is multiplied out over the class
// fields of the pair register (ac_mix5) before the slot address is formed, three VALU per tested position — for dictionaries (almost)
// without patterns of 4 or 5 bytes: a 5-gram window of a word is several times rarer than its rarest 4-gram (kg_ac_anchor.hip).
template <bool CI, bool LINES, bool SHORT, int STRIDE, int ANCH = 0, bool WW = false>
__global__ __launch_bounds__(kAcBlock) void ac_scan_kernel(const AcArgs a)
{
    // WW: -w with the neighbour filter of the depth masks compiled in (stride-2 instantiations without short patterns).  A flag tested at run
    // time cost the plain scans 1.5-3 % — the filter's code in the verifier moved BASELINE config 4 from 6.27 to 6.38-6.46 ms in one process —
    // so the eight instantiations it applies to exist twice; every other instantiation sends -w through the level walk as before.
    static_assert(!WW || (STRIDE == 2 && !SHORT), "the -w filter of the depth masks: pair filter, no short patterns");
    static_assert(!ANCH || (STRIDE == 2 && !LINES && !SHORT), "anchored scan: pair filter, records / counts only");
    extern __shared__ __attribute__((aligned(16))) u32 s_mem[]; // filter table | per wave: candidate bitmap (+ hit and newline bitmaps for -c)
    const u32 lane = ac_lane();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((u32)(size_t)((__attribute__((address_space(3))) u32 *)s_mem) != 0u)
        __builtin_trap(); // the table lookups address LDS absolutely (no static __shared__ in this kernel)
    for (u32 w = threadIdx.x; w < a.filter_words; w += kAcBlock)
        s_mem[w] = a.filter[w];
    const u32 fw = (a.filter_words + 3u) & ~3u;
    constexpr u32 kPerWave = kAcBitmapWords + (LINES ? 2u * kAcBitmapWords : 0u); // candidate | hit | newline bitmaps
    constexpr u32 XB = LINES ? kXBitsLines : kXBitsBig; // index bits of the exact-class table
    constexpr u32 kTabMask = XB == 20 ? 0x1fffcu : 0xfffcu; // byte address of a pair-layout slot
    constexpr bool PAIR = STRIDE == 2;                  // pair-layout table, odd positions tested (see cell_body)
    constexpr bool PIPE = PAIR && !LINES;               // ... with the table reads software-pipelined over the cells of a round
    // -c owns a match by its END and records it in the unit's hit bitmap, so the end j + 2 of the unit's last candidate bit
    // (= the first byte of the next unit) cannot be reported from here: every unit instead verifies one EXTRA candidate, the
    // tested position in front of its first byte, for its second end only
    constexpr bool XCAND = PAIR && LINES;
    // candidate bitmap of the unit: one bit per end position, written as the lane's 16-bit filter result per cell —
    // entry (r * 8 + j) * 64 + lane, so that index order is position order
    u32 *cbits = s_mem + fw + wave * kPerWave;
    unsigned short *cbits16 = reinterpret_cast<unsigned short *>(cbits);
    // PIPE: the pair filter only ever sets the even bits of a lane's 16-bit result, so it stores EIGHT bits per lane and cell
    // (bit q <-> tested position 2q + 1): the bitmap is 1 KiB, lane L owns 4 dwords, and the other KiB of the wave's bitmap
    // area parks the info words and staging slots of a ticket's units until the ticket's end (store
    // placement: the waves store nothing while they stream)
    constexpr u32 kWPL = PIPE ? 4u : 8u; // bitmap dwords per enumerating lane (256 positions)
    uint8_t *cbits8 = reinterpret_cast<uint8_t *>(cbits);
    u32 *park_slots = cbits + 256;       // [kAcUnitsPerTicketMax][16] staged words
    u64 *park_info = reinterpret_cast<u64 *>(cbits + 256 + kAcUnitsPerTicketMax * 16); // [kAcUnitsPerTicketMax]
    u32 *bitmap = cbits + kAcBitmapWords;
    unsigned short *nlmap = reinterpret_cast<unsigned short *>(bitmap + kAcBitmapWords); // 16 bits per lane and cell
    if (LINES)
        for (u32 w = lane; w < kAcBitmapWords; w += 64)
            bitmap[w] = 0u;
    u32 *ebits = cbits + 256; // ANCH: the unit's END-pair bitmap (1 KiB, the park area: an anchored scan parks nothing)
    if (ANCH)
        *reinterpret_cast<uint4 *>(ebits + lane * 4u) = make_uint4(0u, 0u, 0u, 0u);
    const bool want_pos = (a.flags & F_POS) != 0;
    const bool chain = want_pos || LINES;
    const bool emit_final = a.emit_mode != 0;

    u64 acc_total = 0;
    u32 acc_cand = 0; // (wave-uniform, an SGPR: a wave's share of a text holds far fewer than 2^32 tested positions)
    __syncthreads(); // filter tables are in LDS from here on; the waves never synchronise again

    // Waves are autonomous: each draws its own ticket for a.upt consecutive 16-KiB units (128 KiB on large texts:
    // ~25 fetch-adds/us on the single ticket word at 3 TB/s).  A per-tile workgroup barrier (one block per CU
    // because of the table) made every wave wait for the slowest verifier of its tile.
    for (;;)
    {
        u64 tk = 0;
        if (lane == 0)
            tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = ac_rfl64(tk);
        // (emit mode with a list: ticket k is the k-th unit that has to be scanned again — the launch no longer walks the info words of all
        //  units, one dependent load each: 0.7 ms per 8 GiB whenever a single unit had overflowed its slot)
        const bool listed = emit_final && a.redo_list != nullptr;
        if (listed && tk >= (u64)a.n_redo)
            break;
        const u64 u_begin = listed ? (u64)a.redo_list[tk] : tk * (u64)a.upt;
        if (u_begin >= a.num_tiles)
            break;
        const u64 u_end = listed ? u_begin + 1 : ((u_begin + a.upt < a.num_tiles) ? u_begin + a.upt : a.num_tiles);
      // Rolling prefetch: as soon as cell j of a round has been copied out of d[j], the same registers receive cell j
      // of the NEXT round (the rounds of a ticket are contiguous), so a wave always has 8 KiB in flight while it
      // filters and verifies — no second buffer (1024-thread blocks cap a wave at 128 VGPRs).
      uint4 d[kCells];
      bool have = false; // d[] holds (or is receiving) the round about to be processed (uniform)
      u32 carry = 0;     // the 4 bytes in front of that round (valid when have)
      // ANCH, stage 3 DEFERRED over the ticket (round 6): a unit of word text marks ~6 END pairs — a verify batch of 6 lanes that waits
      // for three dependent round trips (window, length masks, bucket) like a full one.  The marked pairs of the ticket's units are
      // collected instead — one per lane in `pend` (unit of the ticket << 14 | pair), verified when 64 are there or the ticket ends —
      // and the units' match counts wait in lanes 0..7 of `ucnt` until the ticket's info words are written.  Ranking and emission
      // stay per unit, in END order: the pairs arrive unit by unit, ascending.
      // (The end-gram kernel with the pipelined pair filter can do the same with its CANDIDATES — -DKG_AC_DEFER_GENERIC — and is slower with
      //  it: BASELINE config 4 holds ~40 candidates per unit, its batches are two thirds full already, and a batch that spans two units
      //  ranks and emits twice: 6.45-6.54 ms against 6.32-6.40 in alternating runs on one box.  Off.)
#ifdef KG_AC_DEFER_GENERIC
      constexpr bool kPend = ANCH != 0 || (PIPE && !SHORT);
#else
      constexpr bool kPend = ANCH != 0;
#endif
      const bool defer = kPend && !emit_final && !(a.flags & ((1u << 27) | (1u << 28) | (1u << 30) | (1u << 31))) && a.upt <= kAcUnitsPerTicketMax &&
                         (ANCH == 0 || a.xtab != nullptr);
      u32 pend = 0, npend = 0, ucnt = 0;
      for (u64 unit = u_begin; unit < u_end; ++unit)
      {
        const u64 useg = a.anchor + unit * (u64)kAcUnitBytes; // the unit = kAcRounds load rounds of 8 KiB
        if (emit_final && (u32)(a.unitinfo[unit] & kUiCountMask) <= a.stage_cap)
            continue;

        const bool parked = PIPE && !ANCH && !emit_final && chain && a.stage_cap == 16u && a.upt <= kAcUnitsPerTicketMax;
        u32 *slot = parked ? park_slots + (u32)(unit - u_begin) * 16u
                           : reinterpret_cast<u32 *>(a.stage) + unit * (u64)a.stage_cap; // 32-bit staged words, see write() below
        const bool do_final = emit_final && want_pos;
        const bool do_stage = !emit_final && want_pos;
        const u64 fbase = do_final ? a.offsets[unit] : 0ull;
        u32 wcnt = 0; // matches of the unit so far == rank of the next one (uniform)

#pragma unroll
        for (int r = 0; r < kAcRounds; ++r)
        {
        const u64 seg = useg + (u64)r * kSegBytes;
        const bool fast_now = seg + kSegBytes <= a.text_len;
        const bool interior = seg >= a.end_lo && seg + kSegBytes <= a.end_hi;
        const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
        // `before` (the 4 bytes in front of the round) is settled BEFORE the round's loads are issued and kept SCALAR: as a
        // vector register filled on a cold path it made the compiler wait vmcnt(0) where the paths join — at the start of
        // EVERY round, draining the rolling prefetch the filter never needed to wait for
        u32 before = 0;
        if (have)
            before = carry;
        else
        {
            u32 bb = 0;
            if (seg >= 4 && seg <= a.text_len)
                bb = *reinterpret_cast<const u32 *>(a.text + seg - 4);
            else
                for (u32 b = 0; b < 4; ++b)
                    if (seg + b >= 4 && seg + b - 4 < a.text_len)
                        bb |= (u32)a.text[seg + b - 4] << (8 * b);
            before = __builtin_amdgcn_readfirstlane(bb);
        }
        // -c prefetches only the first kRoll cells of the next round with the stride-1 filter and none with the pair filter (every
        // -c instantiation then fits the 128 VGPRs of a 1024-thread block without scratch; with the full prefetch they spilled 8-44
        // bytes per lane.  Measured at 32 GiB on the 1000-pattern dictionary, in-kernel -c road: 9.90 ms without the prefetch,
        // 9.61 ms with it and 20 bytes of scratch, profiles/r05_line_counting.txt; texts of 32 MiB and more count their lines on the
        // record list, kg_scan.hip, at 6.8 ms); the rest of a prefetched round is requested here, at its start
        // (the anchored instantiations roll six of the eight cells: the last two are requested at the round's start and consumed last —
        //  their eight registers are free while the two verify stages run, which otherwise spilled 12-20 bytes per lane)
        constexpr int kRoll = LINES ? (STRIDE == 2 ? 0 : KG_AC_LINES_ROLL_CELLS) : (ANCH != 0 ? kCells - 2 : kCells);
        if (fast_now)
        {
#pragma unroll
            for (int j = 0; j < kCells; ++j)
                if (!have || j >= kRoll)
                    d[j] = src[j * kWave]; // (temporal on purpose: non-temporal stream loads measured 7.45 vs 6.87 ms, the verifier's text windows hit in cache)
        }
        // the next round of this ticket, if it is a full one, streams in behind this one
        constexpr bool ROLL = kRoll > 0;
        const bool pf_next = ROLL && fast_now && !emit_final && seg + 2 * (u64)kSegBytes <= a.text_len &&
                             (r + 1 < kAcRounds || unit + 1 < u_end);
        // always issued in the fast path (a uniform address select, not a branch: the s_waitcnt counts stay static);
        // without a next round every lane re-reads the first bytes of this one (one cached line per load, dropped)
        const uint4 *nsrc = pf_next ? src + kSegBytes / 16 : reinterpret_cast<const uint4 *>(a.text + seg);
        // W[0] = the 4 bytes before the lane, W[1..4] = the lane's 16 bytes of cell j
        // have_c: the caller already holds the classes of W[0] and W[4] (fast path: it shuffles the 20-bit class word
        // of the neighbour lane instead of its raw bytes, which saves one compress per cell)
        auto cell_body = [&](const int j, u32 (&W)[5], const bool have_c, const u32 pc0, const u32 pc4) __attribute__((always_inline)) {
            u32 NL = 0;
            if (LINES)
            {
                // (exact 16-bit mask in every cell, ~40 VALU: storing only a has-newline flag and fetching the hit lanes' bytes
                //  in the line pass was measured — one more memory round trip per unit cost more than it saved: 10.15 -> 11.3 ms)
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    NL |= ac_movemask4(ac_eq_bytes(W[w + 1], 0x0a0a0a0au)) << (4 * w);
            }
            u32 cand = 0;
            if constexpr (PAIR)
            {
                // ---- filter, pair layout (stride 2 without -c): the ODD positions are tested, so the gram of position k is two
                //      16-bit-aligned byte pairs.  A dword becomes two 10-bit pair classes in its halves (3 VALU); the gram of
                //      k = 3 (mod 4) is that register as it is, the gram of k = 1 (mod 4) one v_alignbit over two of them.  The
                //      table slot is chosen for this register (ac_pair_slot): bit = the earliest class (the shifter reads the low
                //      5 bits itself), byte address = ((u >> 3) ^ (u >> 13)) & 0x1fffc — two shifts and one v_bitop3, and the
                //      LDS bank is the XOR of two classes (one class alone sends the 16 % blanks of a text to ONE bank: 10.6
                //      instead of 7.2 cycles per 64-lane read).  Per tested position: 0.5 + 3 + 2 VALU and 1.5 for the
                //      classes — 58 per 1-KiB cell where the 100-bit class stream took ~95 ----
                u32 t[5];
#pragma unroll
                for (int w = 1; w < 4; ++w)
                    t[w] = ac_pair(W[w]);
                t[0] = have_c ? pc0 : ac_pair(W[0]);
                t[4] = have_c ? pc4 : ac_pair(W[4]);
                u32 xs[8], dws[8];
                typedef __attribute__((address_space(3))) const u32 lds_u32;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const int w = q / 2 + 1;
                    xs[q] = (q & 1) ? t[w] : __builtin_amdgcn_alignbit(t[w], t[w - 1], 16u);
                    if constexpr (ANCH == 2)
                        xs[q] = ac_mix5(xs[q], __builtin_amdgcn_ubfe(t[(2 * q + 1) / 4], ((2 * q + 1) % 4 == 1) ? 5u : 21u, 5u));
                    // (-c: a 2^19-bit table, address bit 16 = bit 3 of the class c2 dropped)
                    dws[q] = *(lds_u32 *)(size_t)(((xs[q] >> 3) ^ (xs[q] >> 13)) & kTabMask);
                }
                __builtin_amdgcn_sched_barrier(0);
                u32 acc = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    acc = __builtin_amdgcn_alignbit(dws[q] >> (xs[q] & 31u), acc, 2u);
                cand = (acc >> 16) & 0x5555u; // bit 2q <-> tested position 2q + 1 (the verify stage adds the 1)
            }
            else
            {
                // ---- filter, exact-class table: the lane's 20 bytes as a 100-bit stream of 5-bit classes; the
                //      index of end position k is the 20-bit window at bit 5(k+1): one v_alignbit, no hash ----
                u32 c[5];
#pragma unroll
                for (int w = 1; w < 4; ++w)
                    c[w] = ac_cls4(W[w]);
                c[0] = have_c ? pc0 : ac_cls4(W[0]);
                c[4] = have_c ? pc4 : ac_cls4(W[4]);
                u32 R[5];
                R[0] = c[0] | (c[1] << 20);
                R[1] = (c[1] >> 12) | (c[2] << 8) | (c[3] << 28);
                R[2] = (c[3] >> 4) | (c[4] << 16);
                R[3] = c[4] >> 16;
                R[4] = 0;
                // per position 5 VALU: window, dword address, ds_read_b32, shift by the low 5 index bits (the shifter
                // masks them itself), and a funnel shift that pushes the hit bit into the accumulator from the top
                // all table reads of the cell are issued before the first one is consumed (the scheduler otherwise
                // serialises read -> wait -> shift through the accumulator chain: 8-16 LDS latencies per cell)
                constexpr int NK = 16 / STRIDE;
                // (stride 1 with the rolling prefetch: in two batches of 8 — 32 index / data registers at once spilled 12-20 bytes
                //  per lane under the 128-VGPR cap)
#ifndef KG_AC_S1_ONE_BATCH // (A/B switch)
                constexpr int NB = (NK == 16 && !LINES) ? 8 : NK;
#else
                constexpr int NB = NK;
#endif
                typedef __attribute__((address_space(3))) const u32 lds_u32;
                u32 acc = 0;
#pragma unroll
                for (int q0 = 0; q0 < NK; q0 += NB)
                {
                    u32 xs[NB], dws[NB];
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                    {
                        const int o = 5 * ((q0 + q) * STRIDE + 1);
                        xs[q] = (o & 31) ? __builtin_amdgcn_alignbit(R[(o >> 5) + 1], R[o >> 5], (u32)(o & 31)) : R[o >> 5];
                        // the table sits at LDS address 0 (checked at kernel entry): an absolute LDS pointer saves the
                        // v_add of the (link-time) base of s_mem on every lookup
                        dws[q] = *(lds_u32 *)(size_t)((xs[q] >> 3) & ((1u << (XB - 3)) - 4u));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        acc = __builtin_amdgcn_alignbit(dws[q] >> (xs[q] & 31u), acc, (u32)STRIDE);
                    if (q0 + NB < NK)
                        __builtin_amdgcn_sched_barrier(0);
                }
                cand = STRIDE == 2 ? (acc >> 16) & 0x5555u : acc >> 16;
            }
            u32 nlm = NL;
            if (!interior)
            {
                // window bounds relative to the round (uniform, clamped to [0, 8 KiB]) against the lane's 32-bit offset
                const u32 lrel = (u32)j * kCellBytes + lane * 16u;
                auto clip = [&](u64 lo, u64 hi) -> u32 {
                    const u32 rlo = lo > seg ? (u32)((lo - seg) < kSegBytes ? (lo - seg) : kSegBytes) : 0u;
                    const u32 rhi = hi > seg ? (u32)((hi - seg) < kSegBytes ? (hi - seg) : kSegBytes) : 0u;
                    const u32 klo = rlo > lrel ? ((rlo - lrel) < 16u ? (rlo - lrel) : 16u) : 0u;
                    const u32 khi = rhi > lrel ? ((rhi - lrel) < 16u ? (rhi - lrel) : 16u) : 0u;
                    return khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                };
                const u32 endm = clip(a.end_lo, a.end_hi);
                if (!PAIR) // (pair layout: a bit stands for the ends k + 1 and k + 2, the second possibly in the next lane; the
                           //  verify stage applies the exact window)
                    cand &= STRIDE == 2 ? (endm | (endm >> 1)) : endm; // stride 2: t or t + 1 is an end of this launch
                if (LINES)
                    nlm &= clip(a.own_lo, a.own_hi);
            }
            if (LINES) // kept in LDS, not in 16 registers, across the verify stage
                nlmap[(u32)r * (kSegBytes / 16) + (u32)j * kWave + lane] = (unsigned short)nlm;

            // ---- the lane's candidates go into the unit's bitmap; they are enumerated once per unit (below) instead
            //      of ranked per cell (ballots + a divergent store loop: ~15 VALU per cell), and nothing overflows ----
            if (PIPE)
            {
                u32 x = cand & 0x5555u; // even bits -> 8 contiguous bits
                x = (x | (x >> 1)) & 0x3333u;
                x = (x | (x >> 2)) & 0x0f0fu;
                x = (x | (x >> 4)) & 0x00ffu;
                cbits8[(u32)r * (kSegBytes / 16) + (u32)j * kWave + lane] = (uint8_t)x;
            }
            else
                cbits16[(u32)r * (kSegBytes / 16) + (u32)j * kWave + lane] = (unsigned short)cand;
        };
        if (PIPE && fast_now)
        {
            // Pair layout, software-pipelined over the cells of the round: the table reads of cell j + 1 are issued before the
            // results of cell j are consumed, so a wave waits for LDS once per round instead of once per cell (the LDS pipe
            // is ~50 % busy with 4 waves per SIMD: the exposed read latency, not its throughput, was the limit).
            u32 xs[2][8], dw[2][8];
            typedef __attribute__((address_space(3))) const u32 lds_u32;
            auto issue = [&](const int j, u32 (&x)[8], u32 (&v)[8]) __attribute__((always_inline)) {
                u32 t[5];
                t[1] = ac_pair(d[j].x); t[2] = ac_pair(d[j].y); t[3] = ac_pair(d[j].z); t[4] = ac_pair(d[j].w);
                const u32 last = d[j].w;
                if (j < kRoll)
                    d[j] = nsrc[j * kWave];
                t[0] = (u32)__builtin_amdgcn_update_dpp((int)ac_pair(before), (int)t[4], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                before = __builtin_amdgcn_readlane(last, 63);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                {
                    const int w = q / 2 + 1;
                    x[q] = (q & 1) ? t[w] : __builtin_amdgcn_alignbit(t[w], t[w - 1], 16u);
                    if constexpr (ANCH == 2) // the class of the byte in front of the gram (byte 2q - 3 of the lane), mixed into the class fields
                        x[q] = ac_mix5(x[q], __builtin_amdgcn_ubfe(t[(2 * q + 1) / 4], ((2 * q + 1) % 4 == 1) ? 5u : 21u, 5u));
                    v[q] = *(lds_u32 *)(size_t)(((x[q] >> 3) ^ (x[q] >> 13)) & kTabMask);
                }
            };
            auto finish = [&](const int j, const u32 (&x)[8], const u32 (&v)[8]) __attribute__((always_inline)) {
                u32 acc = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    acc = __builtin_amdgcn_alignbit(v[q] >> (x[q] & 31u), acc, 1u);
                cbits8[(u32)r * (kSegBytes / 16) + (u32)j * kWave + lane] = (uint8_t)(acc >> 24); // bit q <-> tested position 2q + 1
            };
            issue(0, xs[0], dw[0]);
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                if (j + 1 < kCells)
                    issue(j + 1, xs[(j + 1) & 1], dw[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                finish(j, xs[j & 1], dw[j & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        else if (fast_now)
        { // straight-line: no branch between a prefetch and the next cell's read of d[]
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                u32 W[5];
                W[1] = d[j].x; W[2] = d[j].y; W[3] = d[j].z; W[4] = d[j].w;
                if (j < kRoll)
                    d[j] = nsrc[j * kWave];
                {
                    // the class word of the 4 bytes in front of the lane = the left neighbour's last one: a DPP wave shift
                    // (lane 0 keeps `old` = the classes of `before`, uniform: scalar ALU) — no LDS permute, no branch
                    const u32 c4 = PAIR ? ac_pair(W[4]) : ac_cls4(W[4]);
                    const u32 cb = PAIR ? ac_pair(before) : ac_cls4(before);
                    const u32 c0 = (u32)__builtin_amdgcn_update_dpp((int)cb, (int)c4, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    W[0] = 0;
                    before = __builtin_amdgcn_readlane(W[4], 63); // the next cell's (and round's) left neighbour
                    cell_body(j, W, true, c0, c4);
                }
            }
        }
        else
        { // the ragged end of the text: bytewise, bounds-checked
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                u32 W[5];
#pragma unroll
                for (int w = 0; w < 5; ++w)
                {
                    u32 v = 0;
                    for (int b = 0; b < 4; ++b)
                    {
                        const u64 o = lbase + (u64)(w * 4 + b);
                        if (o >= 4 && o - 4 < a.text_len)
                            v |= (u32)a.text[o - 4] << (8 * b);
                    }
                    W[w] = v;
                }
                cell_body(j, W, false, 0u, 0u);
            }
        }

        have = pf_next;
        carry = before;
        } // rounds

        // ---- verify (and stage/emit) the candidates, 64 at a time, one per lane.  Lane L owns the 256 positions
        //      [256 L, 256 L + 256) of the bitmap (8 dwords); a wave scan of the popcounts gives every lane its rank
        //      range, and the lane verifying rank q finds its candidate by a binary search over those sums and a
        //      select of the t-th set bit in the owner's block ------------------------------------------------
        // the matches of one batch (one END pair per lane: cA at pos, cB at pos + 1, depth masks or the level walk's verdict):
        // unit-local ranks by a wave prefix, then staging slots / final records / the -c hit bitmap, longest first at one end
        // (what an emission refers to: this unit — or, for the deferred stage 3 of the anchored scan, the unit a pending pair belongs to)
        u64 e_useg = useg, e_fbase = fbase;
        u32 *e_slot = slot;
        auto rank_and_emit = [&](const u64 pos, const u32 cA, const u32 cB, const u64 dmA, const u64 dmB, const bool simA, const bool simB)
            __attribute__((always_inline)) {
            const u32 c = cA + cB;
            u32 incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const u32 t = __shfl_up(incl, o);
                if (lane >= (u32)o)
                    incl += t;
            }
            const u32 rank0 = wcnt + incl - c;
            wcnt += __shfl(incl, 63);
#pragma unroll 1
            for (int e = 0; e < (STRIDE == 2 ? 2 : 1); ++e)
            {
                const u32 ce = e ? cB : cA;
                if (!ce)
                    continue;
                const u64 pe = pos + (u64)e, dme = e ? dmB : dmA;
                const u32 re = rank0 + (e ? cA : 0u), rele = (u32)(pe - e_useg); // the END's bit in the unit's hit bitmap
                const bool sime = e ? simB : simA;
                if (LINES)
                    atomicOr(&bitmap[rele >> 5], 1u << (rele & 31u));
                if (do_stage || do_final)
                {
                    auto write = [&](u32 at, u64 s0, u32 len) {
                        if (do_stage)
                        {
                            if (at < a.stage_cap)
                                e_slot[at] = ((u32)(s0 + 1024u - e_useg) << 11) | len; // start relative to the unit (>= -1023), length <= 1024
                        }
                        else
                        {
                            const u64 g = e_fbase + at;
                            if (g < a.pos_cap)
                            {
                                const u64 st = s0 + a.global_base, en = st + len;
                                *reinterpret_cast<uint4 *>(a.positions + 2 * g) =
                                    make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                            }
                        }
                    };
                    if (sime)
                    {
                        u32 at = re;
                        for (u64 rest = dme; rest;) // longest first
                        {
                            const u32 d = 63u - (u32)__builtin_clzll(rest);
                            rest &= ~(1ull << d);
                            write(at++, pe + 1 - (u64)d, d);
                        }
                    }
                    else
                        ac_walk<CI, true, !SHORT>(a, pe, ce, [&](u32 r, u64 s2, u32 len) { write(re + r, s2, len); });
                }
            }
        };
        if constexpr (ANCH)
        {
            // ================= anchored, stage 2: candidates -> END-pair bitmap =================
            // the unit verifies the ends useg + 1 .. useg + 16384 (bit b <-> the pair 2b + 1, 2b + 2), as the end-gram kernel does
            auto mark = [&](u64 e) {
                if (e > useg && e <= useg + kAcUnitBytes)
                {
                    const u32 b = (u32)(e - useg - 1u) >> 1;
                    atomicOr(&ebits[b >> 5], 1u << (b & 31u));
                }
            };
            u32 mycnt = 0;
            {
                const uint4 lo = *reinterpret_cast<const uint4 *>(cbits + lane * kWPL);
                mycnt = (u32)(__popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w));
            }
            const u32 incl = ac_wave_scan_incl(mycnt); // (on the DPP network: the shuffle version's six address registers spilled in the -i -w instantiations)
            const bool off = (a.flags & (1u << 31)) != 0u; // (ablation hook KREP_GPU_AC_NOVERIFY: filter cost only)
            const u32 n = off ? 0u : __shfl(incl, 63);
            acc_cand += (u32)__builtin_amdgcn_readfirstlane((int)n);
            // an END lies up to 13 bytes behind the tested position that names it: the six tested positions in front of the unit
            // whose ends can fall into it are looked at again here (their own unit drops the ends beyond its last pair)
            const u32 n_x = (!off && useg >= 16u) ? 6u : 0u;
            const u32 n_tot = n + n_x;
            // the candidates of a batch: lane -> tested position (binary search over the owners' prefix sums, then the t-th set bit)
            auto locate = [&](const u32 b0, u64 &t, bool &live, bool &valid) {
                const u32 qi = b0 + lane;
                live = qi < n;
                const bool isx = qi >= n && qi < n_tot;
                u32 rel = 0, own = 0;
#pragma unroll
                for (u32 step = 32; step; step >>= 1)
                {
                    const u32 x = __shfl(incl, (own + step - 1u) & 63u);
                    if (x <= qi)
                        own += step;
                }
                own &= 63u;
                const u32 oincl = __shfl(incl, own), ocnt = __shfl(mycnt, own);
                if (live)
                {
                    u32 x = qi - (oincl - ocnt);
                    const u32 *blk = cbits + own * kWPL;
                    u32 w = 0, word = blk[0];
                    for (;;)
                    {
                        const u32 c = (u32)__popc(word);
                        if (x < c)
                            break;
                        x -= c;
                        word = blk[++w];
                    }
                    for (; x; --x)
                        word &= word - 1u;
                    rel = 2u * (own * 128u + w * 32u + (u32)__builtin_ctz(word)); // (bit b <-> tested position 2b + 1)
                }
                t = isx ? useg - 11u + 2u * (u64)(qi - n) : useg + rel + 1u; // the tested position (odd)
                valid = (live || isx) && t < a.text_len;
            };
            // its eight bytes t - 6 .. t + 1 (without the last one where the text ends: one byte lower, shifted back)
            auto fetch = [&](const u64 t, const bool valid) -> u64 {
                struct __attribute__((packed)) U64p { u64 v; };
                if (!valid || t < 16u)
                    return 0ull;
                const bool hasB = t + 1 < a.text_len;
                const u64 q8 = reinterpret_cast<const U64p *>(a.text + (t - (hasB ? 6u : 7u)))->v;
                return hasB ? q8 : (q8 >> 8);
            };
            // Two dependent round trips per batch (window, then bucket) and, on a text where most candidates reach their bucket, three
            // or four batches per unit: the NEXT batch's windows are requested before this batch's buckets, so a batch waits once
            u64 tN = 0, QN = 0;
            bool liveN = false, validN = false;
            if (n_tot)
            {
                locate(0u, tN, liveN, validN);
                QN = fetch(tN, validN);
            }
            for (u32 b0 = 0; b0 < n_tot; b0 += 64)
            {
                const u64 t = tN, Q = QN;
                const bool live = liveN, valid = validN;
                if (b0 + 64u < n_tot)
                {
                    locate(b0 + 64u, tN, liveN, validN);
                    QN = fetch(tN, validN);
                }
                if (a.flags & (1u << 30))
                { // (ablation hook KREP_GPU_AC_NOPROBE: the count that comes back is the number of filter candidates)
                    wcnt += (u32)__popcll(__ballot(live));
                    continue;
                }
                if (valid)
                {
                    if (t < 16u)
                    { // (the first bytes of the text: no window in front — every end the position could name)
                        for (u32 k = 0; k <= kAnchMaxK + 1u; ++k)
                            mark(t + k);
                    }
                    else
                    {
                        const bool hasB = t + 1 < a.text_len;
                        typedef __attribute__((address_space(3))) const u32 lds_u32;
                        // the table's answer for the gram whose LAST byte is byte `last` of the window (ANCH == 2: with the class of the
                        // byte in front of the gram, byte last - 4)
                        auto gtest = [&](const int last) -> bool {
                            u32 u = ac_pair((u32)(Q >> (8 * (last - 3))));
                            if constexpr (ANCH == 2)
                                u = ac_mix5(u, (u32)(Q >> (8 * (last - 4))) & 31u);
                            return ((*(lds_u32 *)(size_t)(((u >> 3) ^ (u >> 13)) & kTabMask) >> (u & 31u)) & 1u) != 0u;
                        };
                        // the position's own gram again for the six in front of the unit; then, as in the end-gram kernel: an anchor
                        // gram ENDS at t where the window's other gram sits at t - 1, at t + 1 where this one is that other gram
                        // (window bytes 0..7 = text bytes t - 6 .. t + 1)
                        const bool own_ok = live || gtest(6);
                        const bool liveA = own_ok && gtest(5);
                        const bool liveB = own_ok && hasB && gtest(7);
                        u32 kA = (u32)(Q >> 24), kB = (u32)(Q >> 32); // the exact anchor gram: bytes t - 3 .. t | t - 2 .. t + 1
                        if (CI)
                        {
                            kA = ac_fold4(kA);
                            kB = ac_fold4(kB);
                        }
                        // buckets of two 16-byte entries {exact anchor gram, 1 << 31 | offset mask, the three bytes in front of the
                        // 5-byte window, their byte mask}: seven exact bytes decide, all of them in the window already fetched
                        u32 cxA = (u32)Q & 0xffffffu, cxB = (u32)(Q >> 8) & 0xffffffu; // bytes a - 6 .. a - 4 of either parity
                        if (CI)
                        {
                            cxA = ac_fold4(cxA);
                            cxB = ac_fold4(cxB);
                        }
                        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, b0_ = a0, b1_ = a0;
                        if (liveA)
                        {
                            const uint4 *bk = a.anch + 2u * (size_t)(((kA * a.anch_mul) >> 9) & a.anch_mask);
                            a0 = bk[0];
                            a1 = bk[1];
                        }
                        if (liveB)
                        {
                            const uint4 *bk = a.anch + 2u * (size_t)(((kB * a.anch_mul) >> 9) & a.anch_mask);
                            b0_ = bk[0];
                            b1_ = bk[1];
                        }
                        auto pick = [](const uint4 &e0, const uint4 &e1, u32 key, u32 cx) -> u32 {
                            const uint4 &e = (e0.x == key && (e0.y >> 31)) ? e0 : e1;
                            return (e.x == key && (e.y >> 31) && ((cx ^ e.z) & e.w) == 0u) ? (e.y & 0x7fffffffu) : 0u;
                        };
                        u32 mA = liveA ? pick(a0, a1, kA, cxA) : 0u;
                        u32 mB = liveB ? pick(b0_, b1_, kB, cxB) : 0u;
                        while (mA)
                        {
                            mark(t + (u32)__builtin_ctz(mA));
                            mA &= mA - 1u;
                        }
                        while (mB)
                        {
                            mark(t + 1u + (u32)__builtin_ctz(mB));
                            mB &= mB - 1u;
                        }
                    }
                }
            }
        }
        // the verification of ONE END pair per lane (ends pos and pos + 1): the number of matches at either end, their depth masks (or the
        // level walk's verdict) — everything of a verify batch except its ranking and emission
        auto verify_ends = [&](const u64 pos, const bool live, const bool isx, const u64 vuseg, u32 &cA, u32 &cB, u64 &dmA, u64 &dmB, bool &simA,
                               bool &simB) __attribute__((always_inline)) {
            bool liveA = live, liveB = false;
            if (STRIDE == 2)
                liveA = live && pos >= a.end_lo && pos < a.end_hi;
            if (STRIDE == 2) // a candidate stands for the ends t and t + 1
                liveB = (live || isx) && pos + 1 >= a.end_lo && pos + 1 < a.end_hi &&
                        (!XCAND || pos + 1 < vuseg + kAcUnitBytes); // (-c: that end belongs to the next unit's extra candidate)
            cA = cB = 0;
            dmA = dmB = 0;
            simA = simB = false;
            if (STRIDE == 2)
            {
                bool slA = false, slB = false;
                u32 mA = 0, mB = 0;
#ifndef KG_AC_NO_BTEST // (A/B switches of krep_amd/build.py --variant)
                constexpr bool kGram = PAIR && XB == 20 && !ANCH; // the probe asks the class table about both ends (kg_ac_common.h; ANCH: the table holds anchor grams, not end grams)
#else
                constexpr bool kGram = false;
#endif
#ifndef KG_AC_NO_STAGE
                constexpr bool kStaged = kGram;
#else
                constexpr bool kStaged = false;
#endif
                bool exact_done = false;
                if constexpr (ANCH != 0)
#ifndef KG_AC_EXACT_SERIAL
                    if (a.xtab && pos >= 15u && (WW || !(a.flags & F_WW))) // (-w: tested inside ac_exact_end2<.., WW>, on the window's own bytes)
#else
                    if (a.xtab && !(a.flags & F_WW) && pos >= 15u)
#endif
                    {
                        // stage 3 through the length-keyed exact dictionary (kg_ac_common.h ac_exact_end): no trie, no chain
                        bool muA = false, muB = false;
#ifndef KG_AC_EXACT_SERIAL // (A/B switch of krep_amd/build.py --variant: the two ends one behind the other, as first built)
                        if (liveA || liveB)
                            ac_exact_end2<CI, WW>(a, pos, liveA, liveB, mA, mB, muA, muB);
#else
                        if (liveA)
                            mA = ac_exact_end<CI>(a, pos, muA);
                        if (liveB)
                            mB = ac_exact_end<CI>(a, pos + 1u, muB);
#endif
                        // start ownership (the same clip as ac_eval_entry's)
                        auto clip = [&](u32 m, u64 end) -> u32 {
                            const u64 e = end + 1;
                            if (e <= a.own_lo)
                                return 0u;
                            if (e - a.own_lo < 32)
                                m &= (2u << (u32)(e - a.own_lo)) - 1u;
                            if (e > a.own_hi)
                                m = (e - a.own_hi < 32) ? (m & ~((2u << (u32)(e - a.own_hi)) - 1u)) : 0u;
                            return m;
                        };
                        mA = clip(mA, pos);
                        mB = clip(mB, pos + 1u);
                        slA = muA && mA != 0u; // (a pattern the dictionary holds twice: counted and emitted by the level walk)
                        slB = muB && mB != 0u;
                        exact_done = true;
                    }
                if (exact_done)
                {
                }
                else if (!(a.flags & (1u << 30))) // (ablation hook KREP_GPU_AC_NOPROBE: candidate enumeration without the probes)
                    ac_walk_probe2<CI, SHORT, kStaged, WW>(a, pos, liveA, liveB, LINES, mA, slA, mB, slB, [&](u32 E) -> bool {
                        if constexpr (kGram)
                        {
                            // the filter's own lookup for a gram that lies in one dword (cell_body, k = 3 mod 4)
                            typedef __attribute__((address_space(3))) const u32 lds_u32;
                            const u32 u = ac_pair(E);
                            return ((*(lds_u32 *)(size_t)(((u >> 3) ^ (u >> 13)) & kTabMask) >> (u & 31u)) & 1u) != 0u;
                        }
                        else
                            return true;
                    });
                else
                    mA = liveA ? 1u : 0u; // ... and the count that comes back is the number of candidates
                dmA = mA; dmB = mB;
                cA = (u32)__popc(mA); cB = (u32)__popc(mB);
                simA = simB = true;
#ifndef KG_AC_NO_EXACT_SLOW // (A/B switch of krep_amd/build.py --variant)
                if constexpr (!SHORT && ANCH == 0 && !CI) // (-i: eight bytes of scratch per lane in the BASELINE-shaped kernel for a path the anchored instantiations cover)
                    if (a.xtab && !(a.flags & F_WW) && pos >= 15u && (slA || slB))
                    {
                        // an end the chain-compressed entry cannot answer (the trie branches behind its final gram): the exact dictionary
                        // instead of the level walk, where there is one (a word-like text, patterns of 4..16 bytes)
                        auto clipx = [&](u32 m, u64 end) -> u32 {
                            if (LINES)
                                return m; // (-c owns by END)
                            const u64 e = end + 1;
                            if (e <= a.own_lo)
                                return 0u;
                            if (e - a.own_lo < 32)
                                m &= (2u << (u32)(e - a.own_lo)) - 1u;
                            if (e > a.own_hi)
                                m = (e - a.own_hi < 32) ? (m & ~((2u << (u32)(e - a.own_hi)) - 1u)) : 0u;
                            return m;
                        };
                        bool mu = false;
                        if (slA)
                        {
                            const u32 m = clipx(ac_exact_end<CI>(a, pos, mu), pos);
                            if (!mu || !m) { mA = m; slA = false; }
                        }
                        if (slB)
                        {
                            const u32 m = clipx(ac_exact_end<CI>(a, pos + 1u, mu), pos + 1u);
                            if (!mu || !m) { mB = m; slB = false; }
                        }
                        dmA = mA; dmB = mB;
                        cA = (u32)__popc(mA); cB = (u32)__popc(mB);
                    }
#endif
#pragma unroll 1
                for (int e = 0; e < 2; ++e) // the one call site of the level walk
                    if (e ? slB : slA)
                    {
                        u64 dm;
                        bool sim;
                        const u32 c = ac_walk_slow<CI, SHORT>(a, pos + (u64)e, LINES, dm, sim);
                        if (e) { cB = c; dmB = dm; simB = sim; }
                        else { cA = c; dmA = dm; simA = sim; }
                    }
            }
            else if (liveA)
                cA = ac_walk_fast<CI, SHORT>(a, pos, LINES, dmA, simA);
        };
        // (ANCH: what follows is stage 3 — the same verify, over the END-pair bitmap instead of the candidate bitmap)
        const u32 *vbits = ANCH ? ebits : cbits;
        if constexpr (kPend)
        {
          if (ANCH != 0 && (a.flags & (1u << 28))) // (ablation hook KREP_GPU_AC_NOSTAGE3: the anchor stage alone; the count is the number of marked END pairs)
          {
            const uint4 m = *reinterpret_cast<const uint4 *>(ebits + lane * 4u);
            const u32 c = (u32)(__popc(m.x) + __popc(m.y) + __popc(m.z) + __popc(m.w));
            wcnt += (u32)__builtin_amdgcn_readlane((int)ac_wave_scan_incl(c), 63);
            *reinterpret_cast<uint4 *>(ebits + lane * 4u) = make_uint4(0u, 0u, 0u, 0u);
          }
          else if (ANCH == 0 || !(a.flags & (1u << 30)))
          {
            // ---- the unit's marked pairs (end-gram kernel: its candidates) join the pending ones; 64 pending pairs are a verify batch.  Deferred: what is left waits for the
            //      ticket's next unit; otherwise (emit mode, -w, no exact dictionary) it is verified here, at the unit's end ----
            // (registers: nothing of the collection below lives across a batch, and a pair's position is re-derived from its `pend` word —
            //  kept live they cost the anchored instantiations 20-28 B/lane of scratch, whose reloads drain the text prefetch)
            auto pos_of = [&]() -> u64 { // (bit b of a unit's END bitmap <-> the pair 2b + 1, 2b + 2)
                return a.anchor + (u_begin + (u64)(pend >> 14)) * (u64)kAcUnitBytes + ((pend & 0x3fffu) << 1) + 1u;
            };
            auto flush = [&](const u32 cnt) __attribute__((always_inline)) {
                const bool live = lane < cnt;
                u32 cA, cB;
                u64 dmA, dmB;
                bool simA, simB;
                verify_ends(pos_of(), live, false, 0ull, cA, cB, dmA, dmB, simA, simB);
                const u32 cc = cA | (cB << 8);
                const u32 v0 = (u32)__builtin_amdgcn_readfirstlane((int)(pend >> 14)), v1 = (u32)__builtin_amdgcn_readlane((int)(pend >> 14), (int)(cnt - 1u));
                for (u32 v = v0; v <= v1; ++v) // (uniform: the units the batch spans, ascending; not deferred: this unit)
                {
                    const bool in = live && (pend >> 14) == v;
                    if (!__ballot(in))
                        continue;
                    if (defer)
                    {
                        e_useg = a.anchor + (u_begin + (u64)v) * (u64)kAcUnitBytes;
                        e_slot = parked ? park_slots + v * 16u : reinterpret_cast<u32 *>(a.stage) + (u_begin + (u64)v) * (u64)a.stage_cap;
                        wcnt = (u32)__builtin_amdgcn_readlane((int)ucnt, (int)v);
                    }
                    rank_and_emit(pos_of(), in ? (cc & 0xffu) : 0u, in ? (cc >> 8) : 0u, dmA, dmB, simA, simB);
                    if (defer && lane == v)
                        ucnt = wcnt;
                }
                if (defer)
                    wcnt = 0;
            };
            const u32 uv = (u32)(unit - u_begin);
            u32 n = 0, mycnt = 0, incl = 0;
            auto recount = [&]() { // the marked pairs per lane (four dwords of the bitmap each) and their inclusive prefix
                const uint4 lo = *reinterpret_cast<const uint4 *>(vbits + lane * 4u);
                mycnt = (u32)(__popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w));
                incl = ac_wave_scan_incl(mycnt);
                n = (a.flags & (1u << 31)) ? 0u : (u32)__builtin_amdgcn_readlane((int)incl, 63); // (ablation hook KREP_GPU_AC_NOVERIFY: filter cost only)
            };
            recount();
            if (ANCH == 0)
                acc_cand += n;
            u32 taken = 0;
            do // (ONE call site of the verify batch: it is inlined, and two copies cost the anchored instantiations 12-20 B/lane of scratch)
            {
                const u32 room = 64u - npend, take = room < n - taken ? room : n - taken;
                const bool mine = lane >= npend && lane < npend + take;
                const u32 qi = mine ? taken + (lane - npend) : 0u;
                u32 own = 0;
#pragma unroll
                for (u32 step = 32; step; step >>= 1)
                {
                    const u32 t = __shfl(incl, (own + step - 1u) & 63u);
                    if (t <= qi)
                        own += step;
                }
                own &= 63u;
                const u32 oincl = __shfl(incl, own), ocnt = __shfl(mycnt, own);
                if (mine)
                {
                    u32 t = qi - (oincl - ocnt); // the t-th set bit of the owner's four dwords
                    const u32 *blk = vbits + own * 4u;
                    u32 w = 0, word = blk[0];
                    for (;;)
                    {
                        const u32 c = (u32)__popc(word);
                        if (t < c)
                            break;
                        t -= c;
                        word = blk[++w];
                    }
                    for (; t; --t)
                        word &= word - 1u;
                    pend = (uv << 14) | (own * 128u + w * 32u + (u32)__builtin_ctz(word));
                }
                npend += take;
                taken += take;
                if (npend == 64u || (taken >= n && npend && (!defer || unit + 1 == u_end)))
                {
                    flush(npend);
                    npend = 0;
                    if (taken < n)
                        recount(); // (see flush)
                }
            } while (taken < n);
            if (ANCH != 0 && n) // (wave-uniform) the END-pair bitmap is the next unit's again
            {
                // (the zeros are made HERE: as a loop-invariant uint4 they were kept in four registers from the kernel's start, spilled,
                //  and reloaded in front of this store — a scratch load whose wait drained the text prefetch once per unit)
                u32 z;
                asm volatile("v_mov_b32 %0, 0" : "=v"(z));
                *reinterpret_cast<uint4 *>(ebits + lane * 4u) = make_uint4(z, z, z, z);
            }
          }
        }
        else
        {
            u32 mycnt = 0;
            {
                const uint4 lo = *reinterpret_cast<const uint4 *>(vbits + lane * kWPL);
                mycnt = (u32)(__popc(lo.x) + __popc(lo.y) + __popc(lo.z) + __popc(lo.w));
                if (!PIPE)
                {
                    const uint4 hi = *reinterpret_cast<const uint4 *>(vbits + lane * kWPL + 4u);
                    mycnt += (u32)(__popc(hi.x) + __popc(hi.y) + __popc(hi.z) + __popc(hi.w));
                }
            }
            u32 incl = mycnt; // inclusive prefix of the candidate counts over lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const u32 t = __shfl_up(incl, o);
                if (lane >= (u32)o)
                    incl += t;
            }
            const u32 n = (a.flags & (1u << 31)) ? 0u : __shfl(incl, 63); // (ablation hook KREP_GPU_AC_NOVERIFY: filter cost only)
            if (!ANCH)
                acc_cand += (u32)__builtin_amdgcn_readfirstlane((int)n);
            constexpr bool pair = STRIDE == 2; // a candidate stands for the ends t and t + 1
            const u32 n_tot = n + ((XCAND && !(a.flags & (1u << 31)) && useg >= 1u) ? 1u : 0u); // rank n: the extra candidate
            for (u32 b0 = 0; b0 < n_tot; b0 += 64)
            {
                const u32 qi = b0 + lane;
                const bool live = qi < n;
                const bool isx = XCAND && qi == n && qi < n_tot;
                u32 rel = 0;
                {
                    // owner = first lane whose inclusive sum exceeds my rank
                    u32 own = 0;
#pragma unroll
                    for (u32 step = 32; step; step >>= 1)
                    {
                        const u32 t = __shfl(incl, (own + step - 1u) & 63u);
                        if (t <= qi)
                            own += step;
                    }
                    own &= 63u;
                    const u32 oincl = __shfl(incl, own), ocnt = __shfl(mycnt, own);
                    if (live)
                    {
                        u32 t = qi - (oincl - ocnt); // my candidate is the t-th set bit of the owner's block
                        const u32 *blk = vbits + own * kWPL;
                        u32 w = 0, word = blk[0];
                        for (;;)
                        {
                            const u32 c = (u32)__popc(word);
                            if (t < c)
                                break;
                            t -= c;
                            word = blk[++w];
                        }
                        for (; t; --t)
                            word &= word - 1u;
                        rel = PIPE ? 2u * (own * 128u + w * 32u + (u32)__builtin_ctz(word)) // (bit b <-> tested position 2b + 1)
                                   : own * 256u + w * 32u + (u32)__builtin_ctz(word);
                    }
                }
                // pair layout: bit j of the bitmap is tested position j + 1
                const u64 pos = isx ? useg - 1u : useg + rel + (PAIR ? 1u : 0u);
                u32 cA, cB;
                u64 dmA, dmB;
                bool simA, simB;
                verify_ends(pos, live, isx, useg, cA, cB, dmA, dmB, simA, simB);
                rank_and_emit(pos, cA, cB, dmA, dmB, simA, simB);
            }
            if (ANCH && n) // (wave-uniform) the END-pair bitmap is the next unit's again
                *reinterpret_cast<uint4 *>(ebits + lane * 4u) = make_uint4(0u, 0u, 0u, 0u);
        }
        LS2 wls{0, false, false, false};
        if (LINES)
        {
            wls = ac_line_pass(kAcRounds * kCells, [&](int rj, u32 &H, u32 &N) {
                const u32 bitoff = (u32)rj * kCellBytes + lane * 16u;
                H = (bitmap[bitoff >> 5] >> (bitoff & 31u)) & 0xffffu;
                N = nlmap[bitoff >> 4];
            });
            for (u32 w = lane; w < kAcBitmapWords; w += 64)
                bitmap[w] = 0u;
        }

        acc_total += wcnt; // (deferred stage 3: zero here — the ticket's counts are added below)
        if (chain && !emit_final && lane == 0 && !(kPend && defer))
        {
            u64 info = (u64)wcnt;
            if (LINES)
                info |= (wls.nl ? kLnNl : 0) | (wls.head ? kLnHead : 0) | (wls.tail ? kLnTail : 0) |
                        ((u64)(wls.cnt & kUiLineMask) << kUiLineShift);
            else if (wcnt)
                info |= kLnHead | kLnTail;
            if (parked)
                park_info[(u32)(unit - u_begin)] = info;
            else
                a.unitinfo[unit] = info;
            if (want_pos && wcnt > a.stage_cap)
            {
                atomicAdd(&a.ctr->overflow_units, 1ull);
                atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
            }
        }
      }
      if (kPend && defer)
      {
        // the ticket's units: their counts from `ucnt`, their info words in one store
        const u32 nun = (u32)(u_end - u_begin);
        const u32 c = lane < nun ? ucnt : 0u;
        acc_total += (u32)__builtin_amdgcn_readlane((int)ac_wave_scan_incl(c), 63); // (lanes 0..7 hold the counts; on the DPP network: no address registers)
        if (chain && lane < nun)
        {
            u32 l2 = lane;
            asm volatile("" : "+v"(l2)); // (the address is formed here, not kept — and spilled — from the kernel's start)
            const u64 info = (u64)c | (c ? (kLnHead | kLnTail) : 0ull);
            if (PIPE && !ANCH && a.stage_cap == 16u) // (parked: written with the ticket's slots below)
                park_info[l2] = info;
            else
                a.unitinfo[u_begin + l2] = info;
            if (want_pos && c > a.stage_cap)
            {
                atomicAdd(&a.ctr->overflow_units, 1ull);
                atomicMax(&a.ctr->max_unit_count, (u64)c);
            }
        }
      }
      if (PIPE && !ANCH && !emit_final && chain && a.stage_cap == 16u && a.upt <= kAcUnitsPerTicketMax)
      {
        // the ticket's parked info words and slots (consecutive units: one contiguous 64-byte-per-unit region), two stores
        const u32 nun = (u32)(u_end - u_begin);
        u32 l3 = lane;
        asm volatile("" : "+v"(l3)); // (the two addresses are formed here, not kept — and spilled — from the kernel's start)
        if (l3 < nun)
            a.unitinfo[u_begin + l3] = park_info[l3];
        if (want_pos && l3 < nun * 4u)
            reinterpret_cast<uint4 *>(reinterpret_cast<u32 *>(a.stage) + u_begin * 16u)[l3] = reinterpret_cast<const uint4 *>(park_slots)[l3];
      }
    }
    if (lane == 0 && acc_total && !a.emit_mode)
        atomicAdd(&a.ctr->total, acc_total);
    if (lane == 0 && acc_cand && !a.emit_mode)
        atomicAdd(&a.ctr->candidates, (unsigned long long)acc_cand);
}

// the units of a scan whose matches did not fit their staging slot, compacted (any order: an emitted unit writes at its own offset)
__global__ void ac_redo_list_kernel(const u64 *__restrict__ info, u64 n_units, u32 cap, u32 *__restrict__ list, u32 list_cap, unsigned long long *counter)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_units && (u32)(info[i] & kUiCountMask) > cap)
    {
        const u64 at = atomicAdd(counter, 1ull);
        if (at < list_cap)
            list[at] = (u32)i;
    }
}

// ---------------------------------------------------------------------------------------------- host: launches and the scan driver
// (the tables: kg_ac_tables.h, built by kg_ac_build.hip)

#define SCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int g_ac_force_stage_cap = 0; // test hook (krep_gpu_debug_force_stage_cap)

bool ac_counts_lines_in_registers(const AcTables *t) { return t && t->tiny.ok && !t->tiny.five; }

static u32 ac_lds_bytes(u32 filter_words, bool lines)
{
    const u32 per_wave = kAcBitmapWords + (lines ? 2u * kAcBitmapWords : 0u);
    return (((filter_words + 3u) & ~3u) + kAcWaves * per_wave) * (u32)sizeof(u32);
}

constexpr u32 kAcMaxLds = 160u * 1024u; // LDS of a gfx950 CU: the most a launch of the scan kernel can ask for
template <bool CI, bool LN, bool SHORT, int STRIDE, int ANCH = 0, bool WW = false>
static hipError_t ac_launch3(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    if constexpr (!WW && STRIDE == 2 && !SHORT)
        if (a.flags & F_WW)
            return ac_launch3<CI, LN, SHORT, STRIDE, ANCH, true>(a, grid, lds, st);
    // more than 64 KiB of dynamic LDS has to be requested explicitly — once per instantiation and device, not on every launch
    // (the call sits on the latency path of small host buffers), and always for the MOST this kernel can ask for (the whole
    // 160 KiB of a CU), so that no later, larger request can find a stale grant.  Atomic flags: concurrent scans from several
    // host threads may launch the same instantiation; devices beyond the table simply set it every time.
    constexpr int kMaxDev = 64;
    static std::atomic<bool> granted[kMaxDev];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev || !granted[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ac_scan_kernel<CI, LN, SHORT, STRIDE, ANCH, WW>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAcMaxLds);
        if (e != hipSuccess)
            return e;
        if (dev >= 0 && dev < kMaxDev)
            granted[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((ac_scan_kernel<CI, LN, SHORT, STRIDE, ANCH, WW>), dim3(grid), dim3(kAcBlock), lds, st, a);
    return hipGetLastError();
}
template <bool CI, bool LN>
static hipError_t ac_launch2(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    const bool shorts = a.has1 || a.has2 || a.has3;
    if constexpr (!LN)
        if (a.anch && a.stride == 2 && !shorts)
        {
            g_ac_anchored_launches.fetch_add(1, std::memory_order_relaxed);
            return a.anch_five ? ac_launch3<CI, false, false, 2, 2>(a, grid, lds, st) : ac_launch3<CI, false, false, 2, 1>(a, grid, lds, st);
        }
    if (a.stride == 2)
        return shorts ? ac_launch3<CI, LN, true, 2>(a, grid, lds, st) : ac_launch3<CI, LN, false, 2>(a, grid, lds, st);
    return shorts ? ac_launch3<CI, LN, true, 1>(a, grid, lds, st) : ac_launch3<CI, LN, false, 1>(a, grid, lds, st);
}
static hipError_t ac_launch(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    const bool ci = a.flags & F_CI, ln = a.flags & F_LINES;
    if (ci && ln) return ac_launch2<true, true>(a, grid, lds, st);
    if (ci) return ac_launch2<true, false>(a, grid, lds, st);
    if (ln) return ac_launch2<false, true>(a, grid, lds, st);
    return ac_launch2<false, false>(a, grid, lds, st);
}

int ac_scan(AcTables *t, Counters *d_ctr, Counters *h_ctr, PostScratch &post, int num_cu, const uint8_t *d_text, size_t text_len,
            size_t own_lo, size_t own_hi, size_t global_base, match_position_t *d_pos, uint64_t cap, bool ww, bool lines,
            bool track, size_t max_count, hipStream_t st, int time_it, hipEvent_t ev0, hipEvent_t ev1, krep_gpu_scan_out_t *out,
            int list_mode)
{
    // list_mode 1: the record list multi-pattern -c is counted on (END-owned, line gaps counted behind the post-pass);
    // 2: END-owned records only (the emission-order list of -c with a newline inside a pattern: kg_scan.hip)
    const bool lines_on_list = list_mode == 1, own_by_end = list_mode != 0;
    memset(out, 0, sizeof *out);
    if (max_count == 0) // aho_corasick.c:316
        return 0;
    if (own_hi > text_len)
        own_hi = text_len;
    if (text_len == 0)
    {
        // the empty pattern matches the empty text once (aho_corasick.c:441-463)
        if (t->has_empty)
        {
            out->count = out->total_matches = 1;
            if (d_pos && cap && track)
            {
                const match_position_t z{global_base, global_base};
                SCHK(hipMemcpyAsync(d_pos, &z, sizeof z, hipMemcpyHostToDevice, st));
                SCHK(hipStreamSynchronize(st));
                out->stored = 1;
            }
        }
        return 0;
    }
    if (own_lo >= own_hi || t->lmax == 0)
        return 0;
    if (lines && t->has_nl)
        return fail("-c with a pattern containing a newline is not supported by the multi-pattern scan");

    AcArgs a{};
    a.text = d_text;
    a.text_len = text_len;
    a.own_lo = own_lo;
    a.own_hi = own_hi;
    a.global_base = global_base;
    // END indices to examine.  Positions/counts own a match by its START (start-offset ownership), so ends
    // run up to own_hi + Lmax - 2.  -c owns by END index instead (a pattern without '\n' starts and ends
    // on the same line), which keeps the line bookkeeping inside the owned window.
    a.end_lo = own_lo;
    a.end_hi = (lines || own_by_end) ? own_hi : std::min<u64>(text_len, (u64)own_hi + t->lmax - 1);
    if (own_by_end)
    {
        // the record list -c is counted on (kg_scan.hip scan_ac_lines_on_list) owns by END like the in-kernel -c road does: two
        // neighbouring pieces / shards may take different roads, and a match across their cut must belong to exactly one of them
        // (ADVICE r04).  No start clip: every start the buffer holds is in.
        a.own_lo = 0;
        a.own_hi = text_len;
    }
    // (END-owned records: a unit verifies the ends behind its first byte, so the end AT own_lo belongs to the unit in front —
    //  which has to exist; round 6)
    a.anchor = ((own_by_end && !lines && own_lo && !getenv("KREP_GPU_AC_R05_ANCHOR")) ? own_lo - 1 : own_lo) & ~(u64)15; // (the hook: the round-5 grid, for the regression test's own check)
    const u64 unit_bytes = (u64)kAcUnitBytes;
    a.num_tiles = (a.end_hi - a.anchor + unit_bytes - 1) / unit_bytes;
    a.flags = (t->ci ? F_CI : 0) | (ww ? F_WW : 0) | (lines ? F_LINES : 0);
    if (getenv("KREP_GPU_AC_NOVERIFY")) // measurement hook: filter cost only (wrong results by design)
        a.flags |= 1u << 31;
    if (getenv("KREP_GPU_AC_NOSTAGE3")) // measurement hook (anchored scan): candidates -> END pairs, no verify; a count-only scan returns the number of marked pairs
        a.flags |= 1u << 28;
    if (getenv("KREP_GPU_AC_NODEFER")) // A/B hook (anchored scan): stage 3 at every unit's end, as first built
        a.flags |= 1u << 27;
    if (getenv("KREP_GPU_AC_NOPROBE")) // measurement hook: filter + candidate enumeration, no probes: a count-only scan returns the number of candidates
        a.flags |= 1u << 30;
    a.lmax = t->lmax;
    a.has1 = t->has1; a.has2 = t->has2; a.has3 = t->has3; a.has4 = t->has4;
    a.stride = 1;
    a.filter = lines ? t->d_filterx19 : t->d_filterx20;
    a.filter_words = (1u << (lines ? kXBitsLines : kXBitsBig)) / 32;
    if (t->d_filters20)
    {
        a.filter = lines ? t->d_filters19 : t->d_filters20;
        a.stride = 2;
    }
    // anchors (kg_ac_anchor.hip): decided once per dictionary on the first text of >= 1 MiB; every launch without in-kernel -c then
    // filters on the anchor grams and verifies the ends they name
    if (t->anch_state == 0 && text_len >= (1u << 20) && own_hi - own_lo >= (1u << 19))
    {
        SCHK(hipSetDevice(t->device));
        if (ac_anchor_prepare(t, d_text, text_len, own_lo, own_hi, st) == 2)
            (void)hipGetLastError(); // (the end grams stay; a failed sample is not a failed scan)
    }
    if (t->anch_state == 2 && !lines && a.stride == 2 && !getenv("KREP_GPU_AC_NO_ANCHOR"))
    {
        a.filter = t->d_filtera20;
        a.anch = t->d_anch;
        a.anch_mask = t->anch_mask;
        a.anch_mul = t->anch_mul;
        a.anch_five = t->anch_five;
    }
    if (t->d_xtab && !getenv("KREP_GPU_AC_NO_EXACT")) // the exact dictionary: stage 3 of the anchored scan, the slow path of the others
    {
        a.xlen = t->d_xlen;
        a.xtab = t->d_xtab;
        a.xmask = t->xmask;
        a.xmul = t->xmul;
    }
    a.s1 = t->d_s1; a.s2 = t->d_s2; a.s3 = t->d_s3;
    if (t->short_dup)
        a.flags |= F_AC_SHORT_DUP;
    a.edges = t->d_edges;
    a.emask = t->emask;
    a.copies = t->d_copies;
    a.gram4 = t->d_gram4;
    a.g4x = t->d_g4x;
    a.g4mask = t->g4mask;
    a.g4x_mode = t->g4x_mode; a.g4x_mask = t->g4x_mask; a.g4x_mul = t->g4x_mul;
    a.ctr = d_ctr;
    const u64 want = (d_pos && cap && !lines) ? std::min<u64>(cap, (u64)max_count) : 0;
    if (want)
        a.flags |= F_POS;
    a.positions = (u64 *)d_pos;
    a.pos_cap = want;
    // ---- a dictionary of single bytes, records wanted: memchr_search's one-pass kernel with a needle set (kg_single.hip).  The
    // matches of aho_corasick_search for such a dictionary are one (i, i + 1) per matching position in text order
    // (aho_corasick.c:383-437) — exactly that kernel's records.  Shapes by counted density as in lit_pass (kg_scan.hip).
    if (t->set_n && want && !ww && !own_by_end && t->set_ok && text_len >= (size_t)16 * kSegBytes && !g_ac_force_stage_cap &&
        !getenv("KREP_GPU_NO_FUSED1") && !getenv("KREP_GPU_AC_NO_TINY"))
    {
        LitArgs la{};
        la.text = d_text;
        la.text_len = text_len;
        la.own_lo = own_lo;
        la.own_hi = own_hi;
        la.anchor = own_lo & ~(u64)15;
        la.rounds = kRoundsBig;
        const u64 tile_bytes = (u64)kRoundsBig * kSegBytes * kWavesPerBlk;
        la.num_tiles = (own_hi - la.anchor + tile_bytes - 1) / tile_bytes;
        la.global_base = global_base;
        la.ww_exempt_left = ~0ull;
        la.m = 1;
        la.flags = (t->ci ? F_CI : 0) | F_POS;
        la.ctr = d_ctr;
        la.positions = (uint64_t *)d_pos;
        la.pos_cap = want;
        la.set_n = t->set_n;
        for (u32 k = 0; k < t->set_n; ++k)
        {
            la.set_p[k] = 0x01010101u * t->set_b[k];
            la.set_l[k] = (t->ci && t->set_b[k] >= 'a' && t->set_b[k] <= 'z') ? 0x20202020u : 0u;
        }
        la.p0 = la.set_p[0];
        la.l0 = la.set_l[0];
        la.k0 = 0xffu;
        const u64 units32 = la.num_tiles * kWavesPerBlk;
        SCHK(hipSetDevice(t->device));
        if (time_it) SCHK(hipEventRecord(ev0, st));
        for (;;)
        {
            const int shape = t->set_shape;
            const u64 n_tk = single_fused_tickets(units32, shape);
            if (n_tk > post.tk_cap)
            {
                if (post.d_tk) (void)hipFree(post.d_tk);
                post.d_tk = nullptr;
                post.tk_cap = 0;
                SCHK(hipMalloc(&post.d_tk, single_fused_scratch_words(n_tk) * sizeof(unsigned long long)));
                post.tk_cap = n_tk;
            }
            SCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), st));
            SCHK(hipMemsetAsync(post.d_tk, 0, single_fused_scratch_words(n_tk) * sizeof(unsigned long long), st));
            SCHK(launch_single_fused(la, post.d_tk, post.d_tk + n_tk, n_tk, (u32)num_cu, shape, st));
            if (time_it) SCHK(hipEventRecord(ev1, st));
            SCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
            SCHK(hipStreamSynchronize(st));
            if (!h_ctr->overflow_units)
            {
                if (time_it)
                {
                    float ms = 0;
                    SCHK(hipEventElapsedTime(&ms, ev0, ev1));
                    out->kernel_ms = ms;
                }
                const u64 total = h_ctr->total;
                if (shape > 0 && (double)total / (double)(own_hi - la.anchor) < 0.5 * single_fused_max_density(shape - 1))
                    t->set_shape = shape - 1; // (re-evaluated by every scan: a sparser text goes back to the larger tickets)
                out->total_matches = total;
                out->head_line_hit = out->tail_line_hit = total != 0;
                out->count = std::min<u64>(total, (u64)max_count);
                if (track)
                {
                    out->stored = std::min<u64>(out->count, want);
                    out->overflow = out->count > cap;
                }
                return 0;
            }
            const double density = (double)h_ctr->total / (double)(own_hi - la.anchor);
            int next = shape + 1;
            while (next <= kFusedShapeMax && density > single_fused_max_density(next))
                ++next;
            if (next > kFusedShapeMax || h_ctr->total == 0)
                break;
            t->set_shape = next;
        }
        t->set_ok = false; // denser than the largest rings hold: the register-compare kernel below, for good
    }
    // ---- a tiny dictionary, records wanted: ONE pass (kg_ac_tiny.hip FUSED, round 5) — matches ranked into an LDS ring per
    // 128-KiB ticket, the tickets' counts resolved into prefixes by one wave, records written at their final index: no masks kept,
    // no staging, no info words, no post-pass.  A ticket with more matches than its ring holds (~0.6 % of the bytes) is counted,
    // not recorded; the count gives the density, and the density the ticket size (1..4 units) of the DENSE flavour — 16-bit ring
    // entries, matches decoded where they are found — which takes the scan again and the dictionary's next ones (up to ~15 % of
    // the bytes; a long length included, which the item flavour does not take).  Beyond that, and for 4-byte patterns beside a long
    // length, the staging road below; every decision is re-evaluated by the following scans.
    const bool one_pass_ok = t->tiny.ok && !t->tiny.five && want && !ww && list_mode != 2 && track && !g_ac_force_stage_cap &&
                             text_len >= (size_t)64 * kAcUnitBytes && !getenv("KREP_GPU_AC_NO_TINY_FUSED") && !getenv("KREP_GPU_AC_NO_TINY");
    // tickets of the DENSE flavour for e matches per unit: two consecutive tickets share the ring, a quarter is left for clustering
    auto dense_upt = [&](const double e) -> u32 {
        if (e <= 0.0)
            return 4u;
        const double u = (double)ac_tiny_dense_ring() / (2.5 * e);
        return u >= 4.0 ? 4u : (u32)u;
    };
    const bool dense_first = getenv("KREP_GPU_AC_TINY_DENSE_FIRST") != nullptr; // measurement hook: the DENSE flavour at any density
    if (dense_first && one_pass_ok && !t->tiny_dense_upt && t->tiny_dense_ok)
        t->tiny_dense_upt = 4;
    for (int attempt = 0; one_pass_ok && attempt < 3; ++attempt)
    {
        const bool dense = t->tiny_dense_upt != 0 && !getenv("KREP_GPU_AC_NO_TINY_DENSE");
        if (!dense && (!t->tiny_fused_ok || t->tiny.llong))
            break;
        AcArgs f = a;
        f.upt = dense ? t->tiny_dense_upt : (u32)std::min<u64>(kAcUnitsPerTicketMax, std::max<u64>(1, f.num_tiles / ((u64)num_cu * kTinyWaves * 4)));
        const u64 n_tk = (f.num_tiles + f.upt - 1) / f.upt;
        if (n_tk > post.tk_cap)
        {
            if (post.d_tk) (void)hipFree(post.d_tk);
            post.d_tk = nullptr;
            post.tk_cap = 0;
            SCHK(hipMalloc(&post.d_tk, single_fused_scratch_words(n_tk) * sizeof(unsigned long long)));
            post.tk_cap = n_tk;
        }
        f.tk_agg = post.d_tk;
        f.tk_pref = post.d_tk + n_tk;
        f.n_tk = n_tk;
        f.stage_cap = 0;
        SCHK(hipSetDevice(t->device));
        if (time_it) SCHK(hipEventRecord(ev0, st));
        SCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), st));
        SCHK(hipMemsetAsync(post.d_tk, 0, single_fused_scratch_words(n_tk) * sizeof(unsigned long long), st));
        SCHK(ac_tiny_launch_fused(f, t->tiny, n_tk, (u32)num_cu, st, dense));
        // (the record list -c is counted on: its line gaps behind the scan on the same stream, skipped by the kernel itself when
        //  a ticket overflowed its ring)
        if (lines_on_list && tail_launch_line_gaps(d_text, text_len, global_base, (const uint64_t *)d_pos, &d_ctr->total, &d_ctr->overflow_units,
                                                   want, &d_ctr->lines, st))
            return 2;
        if (time_it) SCHK(hipEventRecord(ev1, st));
        SCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        const double per_unit = (double)h_ctr->total / (double)f.num_tiles;
        if (getenv("KREP_GPU_DEBUG"))
            fprintf(stderr, "krep-gpu: tiny one-pass (%s, %u units per ticket): %llu matches, %.1f per unit, %llu waves overflowed\n",
                    dense ? "dense" : "items", f.upt, (unsigned long long)h_ctr->total, per_unit, (unsigned long long)h_ctr->overflow_units);
        if (!h_ctr->overflow_units)
        {
            if (time_it)
            {
                float ms = 0;
                SCHK(hipEventElapsedTime(&ms, ev0, ev1));
                out->kernel_ms = ms;
            }
            const u64 total = h_ctr->total;
            out->total_matches = total;
            out->head_line_hit = out->tail_line_hit = total != 0;
            out->count = std::min<u64>(total, (u64)max_count);
            out->stored = std::min<u64>(out->count, want);
            out->overflow = out->count > cap;
            if (lines_on_list)
                out->line_count = total <= want ? h_ctr->lines : ~0ull; // ~0: the list did not fit when the gaps were counted
            if (dense)
            {
                // re-evaluated by every scan: the ticket size follows the density, and a text with fewer than 24 matches per unit
                // goes back to the item flavour (or, with a long length, to the staging road)
                if (per_unit < 24.0 && !dense_first)
                {
                    t->tiny_dense_upt = 0;
                    t->tiny_fused_ok = true;
                }
                else
                    t->tiny_dense_upt = std::max(1u, dense_upt(per_unit));
            }
            return 0;
        }
        // counted, not recorded (the resolver's running sum is the total — unless the spin-limit safety net fired: total 0)
        u32 nu = (t->tiny_dense_ok && h_ctr->total) ? dense_upt(per_unit) : 0u;
        if (dense && nu >= t->tiny_dense_upt)
            nu = t->tiny_dense_upt - 1u; // (clustered matches: the next smaller ticket)
        if (!dense)
            t->tiny_fused_ok = false; // (re-opened below by a text sparse enough for the rings)
        t->tiny_dense_upt = nu;
        if (!nu)
        {
            t->tiny_dense_ok = false; // (re-opened below, like tiny_fused_ok)
            break;
        }
    }
    const bool chain = want || lines;
    const u64 n_units = a.num_tiles;
    // 16 staged matches (32-bit words: unit-relative start + length) = one 64-byte slot per 16 KiB unit; BASELINE config 4 puts 5.3
    // matches in a unit.  Units that hold more are re-scanned in emit mode, and a scan in which more than 1 in 64 units did
    // raises the dictionary's slot to 64 for its next scans.  (Small slots keep the store stream dense: kg_scan.hip, lit_pass.)
    a.stage_cap = want ? (g_ac_force_stage_cap ? (u32)g_ac_force_stage_cap : t->stage_cap) : 0u;
    if (chain)
    {
        if (post_reserve(post, n_units, (n_units * a.stage_cap + 1) / 2))
            return 2;
        a.unitinfo = post.d_unitinfo;
        a.stage = (u64 *)post.d_stage;
        a.offsets = (const u64 *)post.d_offsets;
    }
    SCHK(hipSetDevice(t->device));
    // tiny dictionaries run in kg_ac_tiny.hip (same units, staging, info words and post-pass); -w takes the general kernel
    // (4-byte patterns beside a long length — AcTiny::five: only the case-sensitive COUNT runs in the register-compare kernel, every
    //  other mode of such a dictionary measured faster here)
    const bool tiny = t->tiny.ok && !ww && (!t->tiny.five || (!want && !lines && !t->ci));
    const u32 waves = tiny ? (u32)kTinyWaves : (u32)kAcWaves;
    const u32 lds = ac_lds_bytes(a.filter_words, lines);
    const u32 per_cu = lds <= 80 * 1024 ? 2u : 1u;
    // ticket size by text size: >= ~4 tickets per resident wave before tickets grow (small host buffers keep every
    // CU busy), 8 units (128 KiB) on large texts
    a.upt = (u32)std::min<u64>(kAcUnitsPerTicketMax, std::max<u64>(1, a.num_tiles / ((u64)num_cu * waves * 4)));
    if (const char *e = getenv("KREP_GPU_AC_UPT")) // test hook: the ticket size large texts get (the anchored scan defers its verify stage over a ticket's units)
    {
        const int v = atoi(e);
        if (v >= 1 && v <= (int)kAcUnitsPerTicketMax)
            a.upt = (u32)v;
    }
    const u64 n_tickets = (a.num_tiles + a.upt - 1) / a.upt;
    const u32 grid = (u32)std::min<u64>((n_tickets + waves - 1) / waves, (u64)num_cu * per_cu);
    auto launch = [&](const AcArgs &args) { return tiny ? ac_tiny_launch(args, t->tiny, n_tickets, (u32)num_cu, st) : ac_launch(args, grid, lds, st); };
    if (time_it) SCHK(hipEventRecord(ev0, st));
    // KREP_GPU_SYNC_DEBUG=1: synchronise behind every launch of this function and say which one faulted
    const bool dbg_sync = getenv("KREP_GPU_SYNC_DEBUG") != nullptr;
    auto dbg = [&](const char *what) -> int {
        if (!dbg_sync)
            return 0;
        const hipError_t e = hipStreamSynchronize(st);
        return e == hipSuccess ? 0 : fail("ac_scan: %s faulted: %s (tiny=%d stride=%u units=%llu stage_cap=%u lines=%d list=%d own=[%llu,%llu) end=[%llu,%llu) n=%llu)", what,
                                          hipGetErrorString(e), (int)tiny, a.stride, (unsigned long long)n_units, a.stage_cap, (int)lines, (int)lines_on_list,
                                          (unsigned long long)a.own_lo, (unsigned long long)a.own_hi, (unsigned long long)a.end_lo,
                                          (unsigned long long)a.end_hi, (unsigned long long)text_len);
    };
    SCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), st));
    SCHK(launch(a));
    if (dbg("scan kernel")) return 2;
    if (chain && post_order(post, n_units, a.stage_cap, 0, a.anchor + global_base, unit_bytes, lines, (uint64_t *)d_pos, want, d_ctr, num_cu, st))
        return 2;
    // lines_on_list (kg_scan.hip scan_ac_lines_on_list): the distinct lines of the record list just gathered, counted by the
    // newline gaps between neighbours, behind the post-pass on the same stream — valid when no unit overflowed its staging
    // slot and the list fitted (the caller checks both and repeats otherwise)
    if (dbg("post-pass")) return 2;
    if (lines_on_list && want && tail_launch_line_gaps(d_text, text_len, global_base, (const uint64_t *)d_pos, &d_ctr->total, &d_ctr->overflow_units, want, &d_ctr->lines, st))
        return 2;
    if (dbg("line-gap kernel")) return 2;
    if (time_it) SCHK(hipEventRecord(ev1, st));
    SCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    // The anchor decision follows the TEXT, and a dictionary meets more than one: the kernel counts the positions its filter passed, and a
    // scan whose measured rate is both one that matters (the decision's own 0.8 % bar) and more than twice what the decision estimated for
    // the filter it chose re-opens the decision, so that the next scan samples the text it is given (an i.i.d. text first, a word-like
    // one after it: 0.04 of the roofline without this, 0.49 with).  Matches are candidates too, so a text dense with matches can ask as
    // well; the repeats are bounded, and a repeat on the same kind of text decides as before.
    if (!tiny && !lines && a.stride == 2 && !a.emit_mode && own_hi > own_lo)
    {
        const double tested = 0.5 * (double)(own_hi - own_lo);
        t->anch_measured = (double)h_ctr->candidates / tested;
        const double est = t->anch_state == 2 ? t->anch_rate : t->anch_rate0;
        if (t->anch_state != 0 && t->anch_resamples < kAnchResamples && text_len >= (1u << 20) && own_hi - own_lo >= (1u << 19) &&
            t->anch_measured > 0.008 && t->anch_measured > 2.0 * est + 0.002 && !getenv("KREP_GPU_AC_ANCHOR") && !getenv("KREP_GPU_AC_NO_RESAMPLE"))
        {
            if (getenv("KREP_GPU_DEBUG"))
                fprintf(stderr, "krep-gpu: anchors: measured %.3f %% candidates per tested position against %.3f %% estimated (%s): the next scan decides again\n",
                        100.0 * t->anch_measured, 100.0 * est, t->anch_state == 2 ? "anchored" : "end grams");
            t->anch_state = 0;
            ++t->anch_resamples;
        }
    }
    if (want && !g_ac_force_stage_cap)
    {
        // a dense dictionary / text: the next scans stage 64 matches per unit — or, when most units of a tiny dictionary hold more
        // than that and the average is beyond 256 (`-e e -e t`: a thousand per unit), none at all: the first launch only counts (no masks kept, nothing staged)
        // and the emit-mode launch writes every record
        if (tiny && h_ctr->overflow_units * 2 > n_units && h_ctr->total > n_units * 256)
            t->stage_cap = 0;
        else if (t->stage_cap && h_ctr->overflow_units * 64 > n_units)
            t->stage_cap = 64;
        else if (!t->stage_cap && h_ctr->total < n_units * 32)
            t->stage_cap = 64; // the dense road (count pass + emit pass) is left again when a text holds < 32 matches per unit
        if (tiny && !t->tiny_fused_ok && h_ctr->total < n_units * 20) // (a 128-KiB ticket's ring holds ~1000: 125 per unit)
            t->tiny_fused_ok = true;
        if (tiny && !t->tiny_dense_ok && h_ctr->total < n_units * 20)
            t->tiny_dense_ok = true;
        // a dense text through the staging road (a dictionary with a long length starts here: the item flavour of the one-pass
        // writer does not take it): its next scans take the DENSE flavour, in tickets sized by the density just counted
        if (tiny && !t->tiny.five && t->tiny_dense_ok && !t->tiny_dense_upt && h_ctr->total >= n_units * 48)
            t->tiny_dense_upt = dense_upt((double)h_ctr->total / (double)n_units);
        // a byte-set dictionary that proved too dense for the one-pass rings gets them back on a text half as dense as they hold
        if (t->set_n && !t->set_ok && own_hi > a.anchor &&
            (double)h_ctr->total / (double)(own_hi - a.anchor) < 0.5 * single_fused_max_density(kFusedShapeMax))
            t->set_ok = true;
    }
    const bool list_lines_valid = lines_on_list && want && !h_ctr->overflow_units && h_ctr->total <= want;
    const u64 list_lines = h_ctr->lines;
    if (want && h_ctr->overflow_units)
    {
        AcArgs e = a;
        e.emit_mode = 1;
        SCHK(hipMemsetAsync(&d_ctr->ticket, 0, sizeof(unsigned long long), st));
        if (!tiny && !getenv("KREP_GPU_AC_NO_REDO_LIST"))
        {
            // the overflowed units as a list (their number is known: the scan counted them), one ticket each
            const u64 n_over = h_ctr->overflow_units;
            if (n_over > t->redo_cap)
            {
                if (t->d_redo) (void)hipFree(t->d_redo);
                t->d_redo = nullptr;
                t->redo_cap = 0;
                if (hipMalloc(&t->d_redo, (n_over + n_over / 4 + 64) * sizeof(u32)) == hipSuccess)
                    t->redo_cap = n_over + n_over / 4 + 64;
                else
                    (void)hipGetLastError();
            }
            if (t->d_redo && n_over <= 0xffffffffull)
            {
                SCHK(hipMemsetAsync(&d_ctr->pad[3], 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(ac_redo_list_kernel, dim3((u32)((n_units + 255) / 256)), dim3(256), 0, st, (const u64 *)a.unitinfo, n_units, a.stage_cap, t->d_redo,
                                   (u32)t->redo_cap, &d_ctr->pad[3]);
                e.redo_list = t->d_redo;
                e.n_redo = (u32)n_over;
            }
        }
        SCHK(launch(e));
        if (time_it) SCHK(hipEventRecord(ev1, st));
        SCHK(hipStreamSynchronize(st));
    }
    if (time_it)
    {
        float ms = 0;
        SCHK(hipEventElapsedTime(&ms, ev0, ev1));
        out->kernel_ms = ms;
    }
    const u64 total = h_ctr->total, nl = h_ctr->lines;
    out->total_matches = total;
    out->line_count = nl;
    const u64 summary = chain ? h_ctr->summary : (total ? (kLnHead | kLnTail) : 0);
    out->has_newline = (summary & kLnNl) != 0;
    out->head_line_hit = (summary & kLnHead) != 0;
    out->tail_line_hit = (summary & kLnTail) != 0;
    if (lines_on_list)
        out->line_count = list_lines_valid ? list_lines : ~0ull; // ~0: the list was not complete when the gaps were counted
    out->count = std::min<u64>(lines ? nl : total, (u64)max_count);
    if (want && track)
    {
        out->stored = std::min<u64>(out->count, want);
        out->overflow = out->count > cap;
    }
    return 0;
}

} // namespace kg
