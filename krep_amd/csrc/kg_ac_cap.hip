// kg_ac_cap.hip — the multi-pattern scan for dictionaries whose patterns all have >= 4 bytes (pair-layout stride-2 filter,
// no -c): aho_corasick_search (aho_corasick.c:299-466) with a verify stage that (almost) reads no text.
//
// What bounds the scan is bytes moved, not instructions or latency (profiles/r04_ac1000_where_the_time_goes.txt, 32 GiB, 1000
// patterns): the filter alone runs at the rate of the stream — 5.5 ms, with 16, 13 or 8 of a workgroup's waves filtering
// alike — and the round-3 kernel's verify stage cost exactly its gathers: 42 candidates per 16 KiB unit, each re-fetching its
// text window through L2 (the unit has long left the 4 MiB L2 by then), 6.9 GB on top of the 34.4 GB stream:
// (34.4 + 6.9) GB / 6.2 TB/s = 6.65 ms measured.  Dedicated verifier waves (kg_ac.hip, SPEC) moved that time, they could not
// remove it.  Here a lane that holds a candidate writes its own 16 bytes and the 4 bytes in front of them (one DPP move) into a
// 24-byte RECORD in the wave's LDS area while the cell is still in registers.  The verify stage takes the unit's records, one
// per lane, cuts the windows of the candidate's two ends out of the record — 6 to 20 real bytes, depending on where in the
// lane the candidate sits — and goes to memory for the table bucket only (L2-resident, 1000 hot lines).  Nearly every false
// candidate dies on the 5th and 6th byte it has; an end whose chain agrees with everything the record holds and goes on
// beyond it (a true match of a long pattern, ~2 per unit) fetches its full window from the text: 1/20 of the gathers.  The
// stream itself is loaded non-temporally, like the literal scan's.
//
// Geometry: 768-thread workgroups (12 waves: the filter is memory-bound from 8 waves on), 2720 bytes of LDS per wave beside
// the 128 KiB exact-class table: 64 records + the parked info words and staging slots of its ticket; up to 168 VGPRs, no
// scratch.  A unit (16 KiB) holds 43 records on BASELINE config 4; a cell whose records do not fit (or a lane with three
// candidates) sends the rest of its round down the slow road (GLOBAL records: windows fetched from the text).
// Ends and ownership: a record of tested position p stands for the ends p and p + 1.  The end that is the FIRST byte of a
// unit always belongs to that unit (its first lane's record, opened by the previous cell's last lane through `prev_c8`, or —
// when the wave did not scan the bytes in front of the unit itself — an unconditional extra candidate): emit-mode re-scans
// of single units count exactly what the first pass counted.
// Everything else — table layouts, the probe, the level walk, ranks, staging, info words, emit mode, -w, ownership — is
// shared with kg_ac.hip (kg_ac_common.h).
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/krep_gpu.h"
#include "kg_ac_common.h"
#include "kg_internal.h"

namespace kg {

constexpr int kCapBlock = 768, kCapWaves = kCapBlock / 64;
constexpr u32 kCapRecs = 64;                     // records per wave (one per lane of a verify batch)
constexpr u32 kCapWaveWords = 680;               // per-wave LDS area in dwords (2720 B, 16-byte multiple)
constexpr u32 kCapOwnAt = 2u * kCapRecs;         // [kCapRecs] uint4: the lane's own 16 bytes, behind [kCapRecs] uint2 {header, the 4 bytes in front}
constexpr u32 kCapParkAt = kCapOwnAt + 4u * kCapRecs;              // [kAcUnitsPerTicketMax][16] staged words
constexpr u32 kCapInfoAt = kCapParkAt + 16u * kAcUnitsPerTicketMax; // [kAcUnitsPerTicketMax] u64 info words
static_assert(kCapInfoAt + 2u * kAcUnitsPerTicketMax <= kCapWaveWords && (kCapOwnAt % 4u) == 0u && (kCapInfoAt % 2u) == 0u, "LDS layout");
static_assert(kCapRecs <= 64, "one record per lane");
// header word of a record: bits 0-15 = the lane's first byte relative to the unit; bits 16-19 = k: the tested position is the LEFT
// lane's last byte (k == 0) or byte 2k - 1 of the lane (k = 1..8) — ascending k is ascending position.  Window A = the 16 bytes
// ending at the tested position, window B = one byte further; the record holds the text from 4 bytes in front of the lane on:
// min(16, 2k + 4) real bytes of window A, min(16, 2k + 5) of window B (k == 0: 5 of B).  k == 0: the left lane verifies end A
// itself, this record only end B (= the lane's own first byte); k == 8: only end A (end B is the next lane's first byte and
// comes with THAT lane's k == 0 record).
constexpr u32 kRecGlobal = 1u << 21; // no usable bytes in the record: the windows are fetched from the text (bits 0-15 = tested position + 1 then), both ends ...
constexpr u32 kRecBOnly = 1u << 22;  // ... or only end B (a unit's extra candidate) ...
constexpr u32 kRecAOnly = 1u << 23;  // ... or only end A (the last byte of a unit: end B belongs to the next unit)

u32 ac_cap_lds_bytes(u32 filter_words) { return (((filter_words + 3u) & ~3u) + kCapWaves * kCapWaveWords) * (u32)sizeof(u32); }
u32 ac_cap_waves() { return kCapWaves; }

template <bool CI, bool SHORT>
__global__ __launch_bounds__(kCapBlock) void ac_cap_kernel(const AcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_mem[]; // filter table | per wave: records, parked slots and info words
    const u32 lane = ac_lane();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((u32)(size_t)((__attribute__((address_space(3))) u32 *)s_mem) != 0u)
        __builtin_trap(); // the table lookups address LDS absolutely
    for (u32 w = threadIdx.x; w < a.filter_words; w += kCapBlock)
        s_mem[w] = a.filter[w];
    const u32 fw = (a.filter_words + 3u) & ~3u;
    u32 *const area = s_mem + fw + wave * kCapWaveWords;
    uint2 *const rec_hdr = reinterpret_cast<uint2 *>(area);            // {header, the 4 bytes in front of the lane}
    uint4 *const rec_own = reinterpret_cast<uint4 *>(area + kCapOwnAt); // the lane's own 16 bytes
    u32 *const park_slots = area + kCapParkAt;
    u64 *const park_info = reinterpret_cast<u64 *>(area + kCapInfoAt);
    const bool want_pos = (a.flags & F_POS) != 0;
    const bool emit_final = a.emit_mode != 0;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));

    u32 fill = 0;      // records waiting (uniform)
    u32 unit_wcnt = 0; // matches of the current unit so far (uniform)
    u64 cur_useg = 0, cur_fbase = 0;
    u32 *cur_slot = nullptr;
    bool cur_do_stage = false, cur_do_final = false;

    // ---- ranks of a verified batch inside its unit and their staging / emission, in the reference's order: end ascending
    //      (= lane order), longest first at one end (aho_corasick.c:353-431) ----
    auto rank_emit = [&](const u64 pos, const u32 cA, const u32 cB, const u64 dmA, const u64 dmB, const bool simA, const bool simB)
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
        const u32 rank0 = unit_wcnt + incl - c;
        unit_wcnt += __shfl(incl, 63);
        if (!(cur_do_stage || cur_do_final))
            return;
#pragma unroll 1
        for (int e = 0; e < 2; ++e)
        {
            const u32 ce = e ? cB : cA;
            if (!ce)
                continue;
            const u64 pe = pos + (u64)e, dme = e ? dmB : dmA;
            const u32 re = rank0 + (e ? cA : 0u);
            const bool sime = e ? simB : simA;
            auto write = [&](u32 at, u64 s0, u32 len) {
                if (cur_do_stage)
                {
                    if (at < a.stage_cap)
                        cur_slot[at] = ((u32)(s0 + 1024u - cur_useg) << 11) | len; // start relative to the unit (>= -1023), length <= 1024
                }
                else
                {
                    const u64 g = cur_fbase + at;
                    if (g < a.pos_cap)
                    {
                        const u64 st = s0 + a.global_base, en = st + len;
                        *reinterpret_cast<uint4 *>(a.positions + 2 * g) = make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
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
    };

    // ---- verifies the waiting records (at most kCapRecs: one per lane) and empties the buffer ----
    auto verify_records = [&]() __attribute__((always_inline)) {
        const u32 n = (a.flags & (1u << 31)) ? 0u : fill; // (ablation hook KREP_GPU_AC_NOVERIFY: filter and capture only)
        fill = 0;
        if (n == 0u)
            return;
        const bool live = lane < n;
        const uint2 hl = live ? rec_hdr[lane] : make_uint2(0u, 0u);
        const u32 h = hl.x;
        const bool glob = (h & kRecGlobal) != 0u;
        const u32 k = (h >> 16) & 15u;
        // the tested position (useg - 1: the unit's extra candidate); GLOBAL records hold it (+ 1) in bits 0-15
        const u64 pos = cur_useg + (u64)(h & 0xffffu) + (glob ? 0ull : (u64)(2u * k)) - 1ull;
        const bool allowA = glob ? !(h & kRecBOnly) : k != 0u, allowB = glob ? !(h & kRecAOnly) : k != 8u;
        const bool liveA = live && allowA && pos >= a.end_lo && pos < a.end_hi;
        const bool liveB = live && allowB && pos + 1 >= a.end_lo && pos + 1 < a.end_hi;
        u32 mA = 0, mB = 0;
        bool slA = false, slB = false;
        if (!(a.flags & (1u << 30))) // (ablation hook KREP_GPU_AC_NOPROBE)
        {
            // the level walk for what the table entries cannot express, and where the 16-byte window does not exist
            const bool walk = (a.flags & (F_WW | (SHORT ? F_AC_SHORT_DUP : 0u))) != 0u || pos < 15ull;
            const bool pa = liveA && !walk, pb = liveB && !walk;
            // Phase 1, from the record: V = 16 zero bytes | the 4 bytes in front of the lane | its 16 bytes; window A = V[o, o + 16),
            // window B = V[o + 1, o + 17) with o = 2k + 4; the zero bytes stand for text the lane never held.
            u32 TA[4] = {0u, 0u, 0u, 0u}, TB[4] = {0u, 0u, 0u, 0u};
            u32 availA = 12u, availB = 12u; // real chain bytes (those in front of a window's last 4)
            if (live && !glob)
            {
                const uint4 O = rec_own[lane];
                u32 W[14] = {0u, 0u, 0u, 0u, hl.y, O.x, O.y, O.z, O.w, 0u, 0u, 0u, 0u, 0u};
                const u32 o = 2u * k + 4u, sh = o >> 2; // 1..5 whole dwords
                if (sh & 4u)
                {
#pragma unroll
                    for (int q = 0; q < 10; ++q)
                        W[q] = W[q + 4];
                }
                if (sh & 2u)
                {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        W[q] = W[q + 2];
                }
                if (sh & 1u)
                {
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        W[q] = W[q + 1];
                }
                const u32 b = o & 3u; // 0 or 2
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                    TA[q] = __builtin_amdgcn_alignbyte(W[q + 1], W[q], b);
                    TB[q] = __builtin_amdgcn_alignbyte(W[q + 1], W[q], b + 1u);
                }
                availA = 2u * k < 12u ? 2u * k : 12u;           // window A holds min(16, 2k + 4) real bytes
                availB = 2u * k + 1u < 12u ? 2u * k + 1u : 12u; // window B one more
            }
            bool fetch = glob && (pa || pb); // GLOBAL records (ragged rounds, extra candidates, overflowed rounds): the window from the text
            u32 F[5] = {0u, 0u, 0u, 0u, 0u};
            auto gather = [&]() __attribute__((always_inline)) {
                if (fetch)
                {
                    struct __attribute__((packed)) U32p { u32 v; };
                    const U32p *qa = reinterpret_cast<const U32p *>(a.text + (pos - 15));
                    F[0] = qa[0].v; F[1] = qa[1].v; F[2] = qa[2].v; F[3] = qa[3].v;
                    F[4] = liveB ? (u32)a.text[pos + 1] : 0u;
                }
            };
            auto take_gather = [&]() __attribute__((always_inline)) {
                if (fetch)
                {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        TA[q] = F[q];
                        TB[q] = __builtin_amdgcn_alignbyte(F[q + 1], F[q], 1u);
                    }
                    availA = availB = 12u;
                }
            };
            if (__ballot(fetch) != 0ull)
            {
                gather();
                take_gather();
            }
            ac_fold_windows<CI>(TA, TB);
            const u32 sbA = SHORT ? ac_short_bits(a, TA[3]) : 0u, sbB = SHORT ? ac_short_bits(a, TB[3]) : 0u;
            AcPairEntries e;
            ac_fetch_pair(a, TA[3], TB[3], pa, pb, e);
            bool undA = false, undB = false;
            if (pa)
                ac_eval_entry(a, e.foundA, TA, e.a0, e.a1, sbA, pos, false, mA, slA, availA, undA);
            if (pb)
                ac_eval_entry(a, e.foundB, TB, e.b0, e.b1, sbB, pos + 1, false, mB, slB, availB, undB);
            // Phase 2: an end whose chain agrees with all the record holds and goes on beyond it — the full window from the text,
            // the entries are still in registers (true matches of long patterns that start in the left lane: ~2 per unit)
            fetch = undA || undB;
            if (__ballot(fetch) != 0ull)
            {
                gather();
                take_gather();
                if (fetch)
                {
                    ac_fold_windows<CI>(TA, TB);
                    bool u2;
                    if (pa)
                        ac_eval_entry(a, e.foundA, TA, e.a0, e.a1, sbA, pos, false, mA, slA, 12u, u2);
                    if (pb)
                        ac_eval_entry(a, e.foundB, TB, e.b0, e.b1, sbB, pos + 1, false, mB, slB, 12u, u2);
                }
            }
            slA = slA || (liveA && walk);
            slB = slB || (liveB && walk);
        }
        u32 cA = (u32)__popc(mA), cB = (u32)__popc(mB);
        u64 dmA = mA, dmB = mB;
        bool simA = true, simB = true;
#pragma unroll 1
        for (int e = 0; e < 2; ++e) // the one call site of the level walk
            if (e ? slB : slA)
            {
                u64 dm;
                bool sim;
                const u32 c = ac_walk_slow<CI, SHORT>(a, pos + (u64)e, false, dm, sim);
                if (e) { cB = c; dmB = dm; simB = sim; }
                else { cA = c; dmA = dm; simA = sim; }
            }
        rank_emit(pos, cA, cB, dmA, dmB, simA, simB);
    };

    // the pair-layout filter of one cell from raw dwords (W[0] = the 4 bytes in front of the lane): the slow road's copy of
    // what the pipelined fast path does in place — bit q of the result <-> tested position 2q + 1 of the lane
    auto filter_cell = [&](const u32 (&W)[5]) __attribute__((always_inline)) -> u32 {
        u32 t[5];
#pragma unroll
        for (int w = 0; w < 5; ++w)
            t[w] = ac_pair(W[w]);
        typedef __attribute__((address_space(3))) const u32 lds_u32;
        u32 acc = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
            const int w = q / 2 + 1;
            const u32 x = (q & 1) ? t[w] : __builtin_amdgcn_alignbit(t[w], t[w - 1], 16u);
            const u32 v = *(lds_u32 *)(size_t)(((x >> 3) ^ (x >> 13)) & 0x1fffcu);
            acc = __builtin_amdgcn_alignbit(v >> (x & 31u), acc, 1u);
        }
        return acc >> 24;
    };

    u64 acc_total = 0;
    __syncthreads(); // the filter table is in LDS from here on; the waves never synchronise again

    for (;;)
    {
        u64 tk = 0;
        if (lane == 0)
            tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = ac_rfl64(tk);
        const u64 u_begin = tk * (u64)a.upt;
        if (u_begin >= a.num_tiles)
            break;
        const u64 u_end = (u_begin + a.upt < a.num_tiles) ? u_begin + a.upt : a.num_tiles;
        // Rolling prefetch: as soon as cell j of a round has been copied out of d[j], the same registers receive cell j of the
        // NEXT round (the rounds of a ticket are contiguous): 8 KiB per wave always in flight without a second buffer.
        uint4 d[kCells];
        bool have = false;  // d[] holds (or is receiving) the round about to be processed (uniform)
        bool ctx = false;   // prev_c8 describes the cell in front of that round (it went through the fast path of this wave)
        u32 carry = 0;      // the 4 bytes in front of that round = the last dword of lane 63 of the previous cell (valid when have)
        u32 prev_c8 = 0;    // that lane's candidate bits (uniform): bit 7 opens a record in lane 0 of the next cell
        const bool parked = !emit_final && want_pos && a.stage_cap == 16u && a.upt <= kAcUnitsPerTicketMax;
        for (u64 unit = u_begin; unit < u_end; ++unit)
        {
            const u64 useg = a.anchor + unit * (u64)kAcUnitBytes; // the unit = kAcRounds load rounds of 8 KiB
            if (emit_final && (u32)(a.unitinfo[unit] & kUiCountMask) <= a.stage_cap)
                continue;
            cur_useg = useg;
            cur_slot = parked ? park_slots + (u32)(unit - u_begin) * 16u : reinterpret_cast<u32 *>(a.stage) + unit * (u64)a.stage_cap;
            cur_do_final = emit_final && want_pos;
            cur_do_stage = !emit_final && want_pos;
            cur_fbase = cur_do_final ? a.offsets[unit] : 0ull;
            unit_wcnt = 0;
            if (!(have && ctx))
            {
                // the wave did not scan the bytes in front of this unit itself: the end that is the unit's first byte gets an
                // unconditional candidate (fill == 0 here: every unit ends with an empty buffer)
                prev_c8 = 0;
                if (useg >= 1u)
                {
                    if (lane == 0)
                        rec_hdr[0] = make_uint2(0u | kRecGlobal | kRecBOnly, 0u);
                    fill = 1;
                }
            }

#pragma unroll
            for (int r = 0; r < kAcRounds; ++r)
            {
                const u64 seg = useg + (u64)r * kSegBytes;
                const bool fast_now = seg + kSegBytes <= a.text_len;
                const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
                // `before` (the 4 bytes in front of the round) is settled BEFORE the round's loads are issued and kept SCALAR (a vector
                // register filled on a cold path made the compiler wait vmcnt(0) where the paths join)
                u32 before = 0;
                if (have)
                    before = carry;
                else
                {
                    u32 bb = 0;
                    if (seg >= 4 && seg <= a.text_len)
                        bb = *reinterpret_cast<const u32 *>(a.text + seg - 4);
                    else
                        for (u32 k = 0; k < 4; ++k)
                            if (seg + k >= 4 && seg + k - 4 < a.text_len)
                                bb |= (u32)a.text[seg + k - 4] << (8 * k);
                    before = __builtin_amdgcn_readfirstlane(bb);
                }
                if (r != 0 && !ctx)
                    prev_c8 = 0; // (a slow round in front of this one has verified its own last end B)
                if (fast_now && !have)
                {
#pragma unroll
                    for (int j = 0; j < kCells; ++j)
                    {
                        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + j * kWave));
                        d[j] = make_uint4(v.x, v.y, v.z, v.w);
                    }
                }
                // the next round of this ticket, if it is a full one, streams in behind this one; always issued in the fast
                // path (a uniform address select, not a branch: the s_waitcnt counts stay static) — without a next round
                // every lane re-reads the first bytes of this one (one cached line per load, dropped)
                const bool pf_next = fast_now && !emit_final && seg + 2 * (u64)kSegBytes <= a.text_len && (r + 1 < kAcRounds || unit + 1 < u_end);
                const uint4 *nsrc = pf_next ? src + kSegBytes / 16 : reinterpret_cast<const uint4 *>(a.text + seg);
                u32 ovf_from = kCells; // first cell of this round whose records did not fit (uniform): the slow road from there
                u32 ovf_prev = 0;      // ... and the candidate bits of the last lane in front of it
                if (fast_now)
                {
                    // Pair layout (see kg_ac.hip), software-pipelined over the cells of the round: the table reads of cell j + 1
                    // are issued before the results of cell j are consumed.
                    u32 xs[2][8], dw[2][8];
                    uint4 raw[2];
                    typedef __attribute__((address_space(3))) const u32 lds_u32;
                    auto issue = [&](const int j, u32 (&x)[8], u32 (&v)[8], uint4 &rw, u32 &bf) __attribute__((always_inline)) {
                        rw = d[j];
                        bf = before; // the 4 bytes in front of lane 0 of this cell
                        u32 t[5];
                        t[1] = ac_pair(rw.x); t[2] = ac_pair(rw.y); t[3] = ac_pair(rw.z); t[4] = ac_pair(rw.w);
                        const u32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(nsrc + j * kWave));
                        d[j] = make_uint4(nv.x, nv.y, nv.z, nv.w);
                        t[0] = (u32)__builtin_amdgcn_update_dpp((int)ac_pair(before), (int)t[4], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                        before = __builtin_amdgcn_readlane(rw.w, 63); // the next cell's (and round's) left neighbour
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                        {
                            const int w = q / 2 + 1;
                            x[q] = (q & 1) ? t[w] : __builtin_amdgcn_alignbit(t[w], t[w - 1], 16u);
                            v[q] = *(lds_u32 *)(size_t)(((x[q] >> 3) ^ (x[q] >> 13)) & 0x1fffcu);
                        }
                    };
                    auto finish = [&](const int j, const u32 (&x)[8], const u32 (&v)[8], const uint4 &rw, const u32 bf) __attribute__((always_inline)) {
                        u32 acc = 0;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            acc = __builtin_amdgcn_alignbit(v[q] >> (x[q] & 31u), acc, 1u);
                        const u32 c8 = acc >> 24; // bit q <-> tested position 2q + 1
                        // the left neighbour: its candidate bits (bit 7 = its last byte: end B is MY first byte) and its last 4 bytes
                        const u32 pc_old = prev_c8;
                        const u32 l8 = (u32)__builtin_amdgcn_update_dpp((int)pc_old, (int)c8, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                        const u32 l4 = (u32)__builtin_amdgcn_update_dpp((int)bf, (int)rw.w, 0x138, 0xf, 0xf, false);
                        prev_c8 = __builtin_amdgcn_readlane(c8, 63);
                        const u32 m9 = (c8 << 1) | (l8 >> 7); // bit k: k == 0 the left lane's last byte, else my byte 2k - 1
                        const u64 bal = __ballot(m9 != 0u);
                        if (bal != 0ull && ovf_from == (u32)kCells)
                        {
                            const u32 m9b = m9 & (m9 - 1u); // without its lowest bit
                            const u32 base = ((u32)r * kCells + (u32)j) * kCellBytes + lane * 16u;
                            if (__ballot(m9b != 0u) == 0ull)
                            { // no lane holds a second candidate (96 % of the cells): one compacting store
                                const u32 nb = (u32)__popcll(bal);
                                if (fill + nb <= kCapRecs)
                                {
                                    if (m9)
                                    {
                                        const u32 at = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, fill));
                                        rec_hdr[at] = make_uint2(base | ((u32)__builtin_ctz(m9) << 16), l4);
                                        rec_own[at] = rw;
                                    }
                                    fill += nb;
                                }
                                else
                                {
                                    ovf_from = (u32)j;
                                    ovf_prev = pc_old;
                                }
                            }
                            else
                            { // a lane with two candidates writes two records, one with three sends the round down the slow road
                                const u64 bal2 = __ballot(m9b != 0u);
                                const u32 nb = (u32)__popcll(bal) + (u32)__popcll(bal2);
                                if (__ballot((m9b & (m9b - 1u)) != 0u) == 0ull && fill + nb <= kCapRecs)
                                {
                                    if (m9)
                                    {
                                        u32 at = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, fill));
                                        at = __builtin_amdgcn_mbcnt_hi((u32)(bal2 >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal2, at));
                                        rec_hdr[at] = make_uint2(base | ((u32)__builtin_ctz(m9) << 16), l4);
                                        rec_own[at] = rw;
                                        if (m9b)
                                        {
                                            rec_hdr[at + 1u] = make_uint2(base | ((u32)__builtin_ctz(m9b) << 16), l4);
                                            rec_own[at + 1u] = rw;
                                        }
                                    }
                                    fill += nb;
                                }
                                else
                                {
                                    ovf_from = (u32)j;
                                    ovf_prev = pc_old;
                                }
                            }
                        }
                    };
                    u32 bfs[2];
                    issue(0, xs[0], dw[0], raw[0], bfs[0]);
#pragma unroll
                    for (int j = 0; j < kCells; ++j)
                    {
                        if (j + 1 < kCells)
                            issue(j + 1, xs[(j + 1) & 1], dw[(j + 1) & 1], raw[(j + 1) & 1], bfs[(j + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        finish(j, xs[j & 1], dw[j & 1], raw[j & 1], bfs[j & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (!fast_now || ovf_from < (u32)kCells)
                {
                    // ---- the slow road: the ragged end of the text, and the rest of a round whose records overflowed the buffer.
                    //      Cell by cell from memory (bounds-checked), GLOBAL records, the buffer verified whenever it is full ----
                    const u32 j0 = fast_now ? ovf_from : 0u;
                    auto put_global = [&](const u32 m8, const u32 lanebase, const u32 extra_flags) __attribute__((always_inline)) {
                        // one GLOBAL record per set bit of m8, in position order, whatever the buffer holds
                        const u32 cntl = (u32)__popc(m8);
                        u32 incl = cntl;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1)
                        {
                            const u32 t = __shfl_up(incl, o);
                            if (lane >= (u32)o)
                                incl += t;
                        }
                        const u32 excl = incl - cntl, total = __shfl(incl, 63);
                        for (u32 done = 0; done < total;)
                        {
                            if (fill == kCapRecs)
                                verify_records();
                            const u32 room = kCapRecs - fill, take = room < total - done ? room : total - done;
                            u32 k = 0;
                            for (u32 mm = m8; mm; mm &= mm - 1u, ++k)
                            {
                                const u32 g = excl + k;
                                if (g >= done && g < done + take)
                                {
                                    const u32 posp1 = lanebase + 2u * (u32)__builtin_ctz(mm) + 2u; // tested position 2q + 1, plus one
                                    // the unit's last byte: its end B is the next unit's first byte and belongs to that unit
                                    rec_hdr[fill + (g - done)] = make_uint2(posp1 | kRecGlobal | extra_flags | (posp1 == kAcUnitBytes ? kRecAOnly : 0u), 0u);
                                }
                            }
                            fill += take;
                            done += take;
                        }
                    };
                    // the last fast cell's last byte may have left its end B — the first byte of cell j0 — open: iteration -1
                    const bool open_b = ((fast_now ? ovf_prev : prev_c8) & 0x80u) != 0u;
                    prev_c8 = 0;
#pragma unroll 1
                    for (int it = open_b ? -1 : 0; it < (int)kCells - (int)j0; ++it)
                    {
                        const u32 j = j0 + (u32)(it < 0 ? 0 : it);
                        u32 m8 = lane == 0u ? 1u : 0u, lb = ((u32)r * kCells + j) * kCellBytes - 2u, fl = kRecBOnly;
                        if (it >= 0)
                        {
                            const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                            u32 W[5];
#pragma unroll
                            for (int w = 0; w < 5; ++w)
                            {
                                u32 v = 0;
                                for (int bq = 0; bq < 4; ++bq)
                                {
                                    const u64 ob = lbase + (u64)(w * 4 + bq);
                                    if (ob >= 4 && ob - 4 < a.text_len)
                                        v |= (u32)a.text[ob - 4] << (8 * bq);
                                }
                                W[w] = v;
                            }
                            m8 = filter_cell(W);
                            lb = ((u32)r * kCells + j) * kCellBytes + lane * 16u;
                            fl = 0u;
                        }
                        put_global(m8, lb, fl);
                    }
                }
                // what the next round finds in front of it
                ctx = fast_now && ovf_from == (u32)kCells;
                have = pf_next;
                carry = before;
            } // rounds

            verify_records();
            acc_total += unit_wcnt;
            if (want_pos && !emit_final && lane == 0)
            {
                const u64 info = (u64)unit_wcnt | (unit_wcnt ? (kLnHead | kLnTail) : 0ull);
                if (parked)
                    park_info[(u32)(unit - u_begin)] = info;
                else
                    a.unitinfo[unit] = info;
                if (unit_wcnt > a.stage_cap)
                {
                    atomicAdd(&a.ctr->overflow_units, 1ull);
                    atomicMax(&a.ctr->max_unit_count, (u64)unit_wcnt);
                }
            }
        }
        if (parked)
        {
            // the ticket's parked info words and slots (consecutive units: one contiguous 64-byte-per-unit region), two stores
            const u32 nun = (u32)(u_end - u_begin);
            if (lane < nun)
                a.unitinfo[u_begin + lane] = park_info[lane];
            if (lane < nun * 4u)
                reinterpret_cast<uint4 *>(reinterpret_cast<u32 *>(a.stage) + u_begin * 16u)[lane] = reinterpret_cast<const uint4 *>(park_slots)[lane];
        }
    }
    if (lane == 0 && acc_total && !a.emit_mode)
        atomicAdd(&a.ctr->total, acc_total);
}

template <bool CI, bool SHORT>
static hipError_t cap_launch2(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    // more than 64 KiB of dynamic LDS has to be requested explicitly — once per instantiation and device (see ac_launch3)
    constexpr int kMaxDev = 64;
    static std::atomic<bool> granted[kMaxDev];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDev || !granted[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ac_cap_kernel<CI, SHORT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess)
            return e;
        if (dev >= 0 && dev < kMaxDev)
            granted[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((ac_cap_kernel<CI, SHORT>), dim3(grid), dim3(kCapBlock), lds, st, a);
    return hipGetLastError();
}

hipError_t ac_cap_launch(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    const bool ci = a.flags & F_CI, shorts = a.has1 || a.has2 || a.has3;
    if (ci)
        return shorts ? cap_launch2<true, true>(a, grid, lds, st) : cap_launch2<true, false>(a, grid, lds, st);
    return shorts ? cap_launch2<false, true>(a, grid, lds, st) : cap_launch2<false, false>(a, grid, lds, st);
}

} // namespace kg
