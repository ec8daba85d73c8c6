// kg_ac_common.h — device code of the multi-pattern scan shared by the kernel variants of kg_ac.hip (verifiers, table
// layouts, constants).  See kg_ac.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include "kg_common.h"

namespace kg {


using u32 = uint32_t;
using u64 = unsigned long long;

#ifndef KG_AC_BLOCK
#define KG_AC_BLOCK 1024
#endif
constexpr int kAcBlock = KG_AC_BLOCK;      // 16 waves share one copy of the filter tables in LDS (A/B builds: -DKG_AC_BLOCK=512)
constexpr int kAcWaves = kAcBlock / 64;
constexpr u32 kS1Words = 256 / 32, kS2Words = 65536 / 32, kS3Words = (1u << 24) / 32; // exact bitmaps of 1-/2-/3-byte patterns
constexpr u32 kHashMul = 0x9E3779B1u;

struct AcArgs
{
    const uint8_t *text;
    u64 text_len, own_lo, own_hi, anchor, num_tiles, global_base;
    u64 end_lo, end_hi;          // range of END indices this launch examines
    u32 flags;                   // F_CI | F_WW | F_POS | F_LINES
    u32 lmax;
    u32 has1, has2, has3, has4;  // which length classes exist
    const u32 *filter;           // exact-class bit table (2^20 or 2^19 bits), copied to LDS address 0
    u32 filter_words;
    const u32 *s1, *s2, *s3;     // exact bitmaps of the 1-/2-/3-byte patterns (keys: last 1/2/3 text bytes, folded under -i)
    const uint2 *edges;          // open addressing: {key = node << 8 | byte, val = child | has_out << 31}
    u32 emask;
    const u32 *copies;           // per node: number of patterns equal to the node's string
    u32 stride;                  // 1 or 2: text positions per filter lookup (2 = even positions only, see ac_scan_kernel)
    u32 upt;                     // units per wave ticket of the fused kernel (1..kAcUnitsPerTicketMax, by text size)
    const uint2 *gram4;          // exact last-4-bytes -> {key, depth-4 node | has_out << 31} (val 0 = empty)
    u32 g4mask;
    const uint4 *g4x;            // entries of 2 x uint4: {key, child, info, endmask} {chain bytes x3, -}; layout by g4x_mode:
    u32 g4x_mode, g4x_mask, g4x_mul; // 0: the slots of gram4 (linear probing); 1: buckets of TWO entries (64 B), no bucket
                                 //    overfull, bucket = ((key * g4x_mul) >> 9) & g4x_mask: exactly one round trip per probe
    unsigned long long *unitinfo;
    Counters *ctr;
    u64 *stage;
    u32 stage_cap;
    u32 emit_mode;
    const u64 *offsets;
    u64 *positions;
    u64 pos_cap;
    // one-pass records of the register-compare kernel (kg_ac_tiny.hip, FUSED): per ticket its published match count and the
    // exclusive prefix the resolver wave derives from them (kg_single.hip's scheme); n_tk tickets of `upt` units
    u64 *tk_agg, *tk_pref;
    u64 n_tk;
    // ANCHORED scan (round 6, kg_ac_anchor.hip): the filter table holds, per pattern, the class grams of ONE 5-byte window chosen
    // by rarity in the text (its anchor, k bytes before the pattern's end) instead of always its last five bytes; a candidate's
    // exact anchor gram selects {key, 1 << 31 | mask of the offsets k} in buckets of two (one 16-byte load), every end t + k the
    // mask names is marked in the unit's END bitmap, and the marked ends are verified exactly by ac_walk_fast (the end-anchored
    // verifier, unchanged): the anchor stage is a superset filter of the ENDS, nothing else.
    const uint4 *anch;
    u32 anch_mask, anch_mul;
    u32 anch_five; // the table is indexed with FIVE classes (ac_scan_kernel<.., ANCH = 2>)
    // stage 3 of the anchored scan for dictionaries of 4..16-byte patterns: which LENGTHS end in these four bytes (a hashed 16-bit mask per
    // final gram), and per length an exact entry {the pattern right-aligned in 16 bytes, length, copies} in buckets of two — every pattern
    // that ends at a marked position, longest first, without walking a trie (ac_exact_end)
    const u32 *redo_list;       // emit mode: the units to scan again (those whose matches did not fit their staging slot), one per ticket; NULL: every unit is looked at
    u32 n_redo;
    const unsigned short *xlen; // [2][65536]: bit l - 4 set: a pattern of length l ends in these bytes — [0] lengths 4..7 by the hash of the
                                //   last FOUR bytes, [1] lengths 8..16 by the hash of the last EIGHT (a word's last eight bytes are all but its
                                //   own: one length, one probe; by the last four alone `tion` named nine lengths, nine round trips in turn)
    const uint4 *xtab;          // buckets of two 32-byte entries {w0..w3}{length, copies, 0, 0}
    u32 xmask, xmul;
};
constexpr u32 kAnchMaxK = 12; // an anchor gram ends at most 12 bytes before its pattern's end: an END lies within 13 bytes of the tested position

// Tiny dictionaries (kg_ac_tiny.hip): every pattern 1..4 bytes (or 1..3 and ONE length of 5..8), at most kTinyPer of each length,
// no duplicates.  The patterns
// travel as kernel arguments, one dword each: byte s = the pattern byte s places before its END (folded under -i), and a
// second dword of flags: bit s = that byte is a letter (-i compares it as (x | 0x20) == c).
constexpr u32 kTinyPer = 4;
constexpr int kTinyWaves = 4;  // small workgroups: 8 KiB of LDS per wave when records are wanted (the length words of a
                               // 16-KiB unit), 4 KiB under -c, none for a count
struct AcTiny
{
    u32 ok;           // the dictionary qualifies
    u32 lmax, ncls;   // longest pattern; number of distinct lengths
    u32 n[4];         // patterns of length 1, 2, 3, 4
    u32 pk[4][kTinyPer];
    u32 lf[4][kTinyPer];
    // ONE long length (5..8 bytes, llong; 0 = none) may take the place of length 4: pk[3] / lf[3] then hold the LAST four bytes of
    // those patterns and pk2 / lf2 their first llong - 4 bytes (byte s = s places before the end of that part)
    u32 llong;
    u32 pk2[kTinyPer], lf2[kTinyPer];
    // five: 4-byte patterns AND a long length (`-e if -e else -e while`, round 5): the long patterns are a FIFTH class — pk5 / lf5
    // their last four bytes, pk2 / lf2 their first llong - 4 — which only the counting and the one-pass (FUSED) instantiations
    // know; every other mode of such a dictionary runs in the general kernel (ac_scan)
    u32 five, n5;
    u32 pk5[kTinyPer], lf5[kTinyPer];
};
hipError_t ac_tiny_launch(const AcArgs &a, const AcTiny &td, u64 n_tickets, u32 num_cu, hipStream_t st); // sizes its own grid
hipError_t ac_tiny_launch_fused(const AcArgs &a, const AcTiny &td, u64 n_tickets, u32 num_cu, hipStream_t st, bool dense = false); // one-pass records
u32 ac_tiny_dense_ring(); // 16-bit ring entries per wave of its DENSE flavour
constexpr u32 kTinyRing = 1024; // items (12 bytes: a lane-cell's two length words + its index) per wave of the one-pass kernel's LDS ring:
                                // the ticket being scanned + the one waiting

__device__ __forceinline__ u32 ac_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 ac_rfl64(u64 v)
{
    return ((u64)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)v);
}
// inclusive prefix sum over the 64 lanes on the DPP network: four row shifts inside the rows of 16, then lane 15 of rows 0 / 2 into rows
// 1 / 3 and lane 31 into rows 2 and 3 — six v_add with a DPP modifier, no LDS traffic and no per-lane address registers (the shuffle
// version keeps six of them alive for as long as it is loop-invariant)
__device__ __forceinline__ u32 ac_wave_scan_incl(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ u32 ac_fold4(u32 x)
{
    u32 t = x & 0x7f7f7f7fu;
    return x | (((t + 0x3f3f3f3fu) & ~(t + 0x25252525u) & ~x & 0x80808080u) >> 2);
}
__device__ __forceinline__ u32 ac_eq_bytes(u32 x, u32 c4)
{
    u32 y = x ^ c4;
    return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu);
}
__device__ __forceinline__ u32 ac_movemask4(u32 t) { return (((t >> 7) * 0x00204081u) >> 21) & 0xfu; }
__device__ __forceinline__ bool ac_wordc(u32 c) { return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_'; }

struct LS2 { u32 cnt; bool nl, head, tail; };
// The distinct lines of a unit that hold a match END, from the 16-bit hit / newline masks H, N of its lane-cells (get(rj, H, N),
// rj = cell of the unit, this lane's 16 bytes): a line is counted at its first match.  Inside a lane that is carry arithmetic on
// the two masks; across the 64 lanes of a cell "does the line that enters this lane already hold a match" is a carry chain —
// generate = a match behind the lane's last newline (any match in a lane without one), propagate = neither a match nor a
// newline — which ONE 64-bit scalar add resolves, its carry-out being the state that enters the next cell (round 5; the same
// scheme as kg_literal.hip line_cell.  Round 4 resolved every cell with four ballots, a per-lane search for the nearest newline
// lane below and a shuffle reduction, and the -c instantiations spilled 8-44 bytes per lane).
template <typename Get>
__device__ __forceinline__ LS2 ac_line_pass(int ncells, Get get)
{
    u32 l_cnt = 0, s_new = 0;
    bool s_open = false, s_seen = false, s_head = false;
#pragma unroll 1
    for (int rj = 0; rj < ncells; ++rj)
    {
        u32 H, N;
        get(rj, H, N);
        const u64 B_any = __ballot(H != 0u), B_nl = __ballot(N != 0u);
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
        const u32 Hs = H | N;
        l_cnt += (u32)__popc(H & ~(Hs - ((N << 1) & 0xffffu)));          // first match behind each newline of the lane
        const u64 B_head = __ballot((H & (Hs ^ (Hs - 1u))) != 0u);       // the lane's lowest flag is a match
        const u64 G = __ballot(H > N), P = ~(B_nl | B_any);              // the highest flag is a match | nothing in the lane
        const unsigned __int128 sum = (unsigned __int128)(G | P) + G + (s_open ? 1u : 0u);
        const u64 O = (u64)sum ^ P; // bit l: the line entering lane l already holds a match
        s_new += (u32)__popcll(B_head & ~O);
        if (!s_seen && B_nl)
        {
            s_head = (((O | B_head) >> __builtin_ctzll(B_nl)) & 1ull) != 0ull;
            s_seen = true;
        }
        s_open = (u64)(sum >> 64) != 0ull;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        l_cnt += __shfl_xor(l_cnt, o);
    return LS2{l_cnt + s_new, s_seen, s_seen ? s_head : s_open, s_open};
}
__device__ __forceinline__ LS2 ls2_combine(const LS2 &a, const LS2 &b)
{
    return LS2{a.cnt + b.cnt - ((a.tail && b.head) ? 1u : 0u), a.nl || b.nl, a.nl ? a.head : (a.head || b.head),
               b.nl ? b.tail : (a.tail || b.tail)};
}

// Walk the reversed trie from end index i.  EMIT == false: returns the number of matches ending at i
// (after ownership and -w).  EMIT == true: additionally hands them to `put`, longest first.
// Deliberately NOT inlined per call site: the scan loop keeps one copy of each instantiation.
template <bool CI, bool EMIT, bool JUMP, typename Put>
__device__ __forceinline__ u32 ac_walk(const AcArgs &a, u64 i, u32 total, Put put)
{
    u32 node = 0, seen = 0;
    const bool ww = (a.flags & F_WW) != 0, lines = (a.flags & F_LINES) != 0;
    const u64 maxd = (i + 1 < (u64)a.lmax) ? i + 1 : (u64)a.lmax;
    u64 d = 1;
    u32 child = 0xffffffffu;
    if (JUMP && i >= 3)
    {
        // every pattern has >= 4 bytes: resolve trie levels 1..4 with ONE probe keyed by the exact last 4 bytes
        struct __attribute__((packed)) U32p { u32 v; };
        u32 E = reinterpret_cast<const U32p *>(a.text + (i - 3))->v; // one unaligned dword load
        if (CI)
            E = ac_fold4(E);
        for (u32 h = (E * kHashMul) >> 9;; ++h)
        {
            const uint2 e = a.gram4[h & a.g4mask];
            if (e.y == 0u)
                return 0u; // not a suffix of any pattern
            if (e.x == E)
            {
                child = e.y;
                break;
            }
        }
        d = 4;
    }
    else if (JUMP)
        return 0u; // fewer than 4 bytes before i: no pattern of length >= 4 can end here
    for (; d <= maxd; ++d)
    {
        if (!(JUMP && d == 4))
        {
            u32 c = a.text[i + 1 - d];
            if (CI && (c - 'A' < 26u))
                c += 32u;
            const u32 key = (node << 8) | c;
            u32 h = (key * kHashMul) >> 7;
            child = 0xffffffffu;
            for (;; ++h)
            {
                const uint2 e = a.edges[h & a.emask];
                if (e.x == key)
                {
                    child = e.y;
                    break;
                }
                if (e.x == 0xffffffffu)
                    break;
            }
            if (child == 0xffffffffu)
                break;
        }
        node = child & 0x7fffffffu;
        if (child & 0x80000000u)
        {
            const u64 s = i + 1 - d;
            bool ok = lines ? true : (s >= a.own_lo && s < a.own_hi); // -c owns by END index (see ac_scan)
            if (ok && ww)
            {
                if (s > 0 && ac_wordc(a.text[s - 1]))
                    ok = false;
                else if (i + 1 < a.text_len && ac_wordc(a.text[i + 1]))
                    ok = false;
            }
            if (ok)
            {
                const u32 k = a.copies[node];
                if (EMIT)
                    for (u32 q = 0; q < k; ++q)
                        put(total - seen - k + q, s, (u32)d);
                seen += k;
            }
        }
    }
    return seen;
}

// Fast verifier for pattern sets whose patterns all have >= 4 bytes (CLS == 8): ONE pass.
//  * the 16 bytes ending at the candidate are loaded once (one unaligned 16-byte load) — the walk then needs one
//    dependent access per level (the edge probe) instead of two (text byte + edge probe);
//  * levels 1..4 are resolved by the exact 4-gram table;
//  * the depths at which a pattern ends are remembered in a bit mask, so the longest-first emission needs no second
//    walk (falls back to it when a pattern has duplicate copies or the set has patterns longer than 64 bytes).
template <bool CI>
__device__ __forceinline__ u32 ac_walk_levels(const AcArgs &a, u64 i, bool own_by_end, u64 &depthmask, bool &simple)
{
    depthmask = 0;
    simple = true;
    const bool ww = (a.flags & F_WW) != 0;
    if (i < 15)
    { // too close to the start of the text for the 16-byte window: generic walk
        simple = false;
        return ac_walk<CI, false, true>(a, i, 0u, [](u32, u64, u32) {});
    }
    struct __attribute__((packed)) U32p { u32 v; };
    const U32p *q = reinterpret_cast<const U32p *>(a.text + (i - 15));
    u32 T[4] = {q[0].v, q[1].v, q[2].v, q[3].v};
    if (CI)
    {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            T[w] = ac_fold4(T[w]);
    }
    u32 child = 0xffffffffu;
    for (u32 h = (T[3] * kHashMul) >> 9;; ++h)
    {
        const uint2 e = a.gram4[h & a.g4mask];
        if (e.y == 0u)
            return 0u; // not a suffix of any pattern
        if (e.x == T[3])
        {
            child = e.y;
            break;
        }
    }
    u32 node = 0, seen = 0;
    const u64 maxd = (i + 1 < (u64)a.lmax) ? i + 1 : (u64)a.lmax;
    for (u64 d = 4; d <= maxd; ++d)
    {
        if (d > 4)
        {
            u32 c;
            if (d <= 16)
            {
                const u32 bi = 16u - (u32)d; // byte i-d+1 sits at index 16-d of the window
                const u32 w = bi >> 2;
                const u32 word = w == 0 ? T[0] : w == 1 ? T[1] : w == 2 ? T[2] : T[3];
                c = (word >> (8 * (bi & 3u))) & 0xffu;
            }
            else
            {
                c = a.text[i + 1 - d];
                if (CI && (c - 'A' < 26u))
                    c += 32u;
            }
            const u32 key = (node << 8) | c;
            child = 0xffffffffu;
            for (u32 h = (key * kHashMul) >> 7;; ++h)
            {
                const uint2 e = a.edges[h & a.emask];
                if (e.x == key)
                {
                    child = e.y;
                    break;
                }
                if (e.x == 0xffffffffu)
                    break;
            }
            if (child == 0xffffffffu)
                break;
        }
        node = child & 0x7fffffffu;
        if (child & 0x80000000u)
        {
            const u64 s = i + 1 - d;
            bool ok = own_by_end ? true : (s >= a.own_lo && s < a.own_hi);
            if (ok && ww)
            {
                if (s > 0 && ac_wordc(a.text[s - 1]))
                    ok = false;
                else if (i + 1 < a.text_len && ac_wordc(a.text[i + 1]))
                    ok = false;
            }
            if (ok)
            {
                const u32 k = a.copies[node];
                seen += k;
                if (k != 1u || d > 63)
                    simple = false;
                else
                    depthmask |= 1ull << d;
            }
        }
    }
    return seen;
}

// Chain-compressed verifier (the shipped path).  Below depth 4 almost every node of the reversed trie of a large
// dictionary lies on a unary chain, so the 4-gram entry carries the next <= 12 bytes of that chain (in text order, the
// window text[i-15 .. i-4] compares against it dword by dword) and the depths at which patterns end: a candidate costs
// TWO dependent accesses (text window, table entry) however long the match is, instead of one probe per trie level
// (12 serial L2 round trips for a 16-byte match — the latency that bounded the verify stage).  1-3-byte patterns
// (SHORT dictionaries) are exact bitmap lookups keyed by the last bytes of the same window, in flight with the probe.
// Anything this cannot express (branching below depth 4, duplicate patterns, chains continuing past depth 16, -w, a
// candidate within 15 bytes of the text start) takes the level-by-level walk (ac_walk_slow, ONE call site per kernel).
constexpr u32 kG4Simple = 1u << 4, kG4Cont = 1u << 5; // info bits above the chain length [3:0]
constexpr u32 F_AC_SHORT_DUP = 1u << 29;              // AcArgs.flags: a 1-3-byte pattern occurs twice: level walk only

// bits 1..3 of the depth mask: which 1-/2-/3-byte patterns end with the 4 text bytes E (byte i on top)
__device__ __forceinline__ u32 ac_short_bits(const AcArgs &a, u32 E)
{
    u32 sb = 0;
    if (a.has1)
    {
        const u32 k = E >> 24;
        sb |= ((a.s1[k >> 5] >> (k & 31u)) & 1u) << 1;
    }
    if (a.has2)
    {
        const u32 k = E >> 16;
        sb |= ((a.s2[k >> 5] >> (k & 31u)) & 1u) << 2;
    }
    if (a.has3)
    {
        const u32 k = E >> 8;
        sb |= ((a.s3[k >> 5] >> (k & 31u)) & 1u) << 3;
    }
    return sb;
}

// depth mask of the matches ending at `end` from a probed entry (found) and the short-pattern bits sb
__device__ __forceinline__ void ac_eval_entry(const AcArgs &a, bool found, const u32 (&T)[4], const uint4 &e0, const uint4 &e1,
                                              u32 sb, u64 end, bool own_by_end, u32 &dm, bool &slow)
{
    u32 m = sb;
    if (found)
    {
        const u32 info = e0.z, clen = info & 15u;
        auto same = [](u32 x) -> u32 { return x ? (u32)__builtin_clz(x) >> 3 : 4u; }; // equal bytes from the top
        u32 L = same(T[2] ^ e1.z);
        if (L == 4u)
        {
            L += same(T[1] ^ e1.y);
            if (L == 8u)
                L += same(T[0] ^ e1.x);
        }
        L = L < clen ? L : clen;
        slow = !(info & kG4Simple) || (L == clen && (info & kG4Cont));
        m |= ((e0.y >> 31) << 4) | ((e0.w & ((1u << L) - 1u)) << 5); // bit d: a pattern of length d ends here
    }
    if (!own_by_end)
    { // the match start s = end + 1 - d has to lie in [own_lo, own_hi)
        const u64 e = end + 1;
        if (e <= a.own_lo)
            m = 0;
        else
        {
            if (e - a.own_lo < 32)
                m &= (2u << (u32)(e - a.own_lo)) - 1u; // d <= e - own_lo
            if (e > a.own_hi)
                m = (e - a.own_hi < 32) ? (m & ~((2u << (u32)(e - a.own_hi)) - 1u)) : 0u; // d > e - own_hi
        }
    }
    dm = m;
}

// the level-by-level walk behind the fast verifiers: count, and (dictionaries without short patterns) the depth mask
template <bool CI, bool SHORT>
__device__ __forceinline__ u32 ac_walk_slow(const AcArgs &a, u64 i, bool own_by_end, u64 &depthmask, bool &simple)
{
    if (SHORT)
    {
        depthmask = 0;
        simple = false; // the emit pass walks again
        return ac_walk<CI, false, false>(a, i, 0u, [](u32, u64, u32) {});
    }
    return ac_walk_levels<CI>(a, i, own_by_end, depthmask, simple);
}

template <bool CI, bool SHORT>
__device__ __forceinline__ u32 ac_walk_fast(const AcArgs &a, u64 i, bool own_by_end, u64 &depthmask, bool &simple)
{
    bool slow = i < 15 || (a.flags & (F_WW | (SHORT ? F_AC_SHORT_DUP : 0u))); // one (inlined) call site for the level walk
    u32 dm = 0;
    if (!slow)
    {
        struct __attribute__((packed)) U32p { u32 v; };
        const U32p *q = reinterpret_cast<const U32p *>(a.text + (i - 15));
        u32 T[4] = {q[0].v, q[1].v, q[2].v, q[3].v};
        if (CI)
        {
#pragma unroll
            for (int w = 0; w < 4; ++w)
                T[w] = ac_fold4(T[w]);
        }
        const u32 sb = SHORT ? ac_short_bits(a, T[3]) : 0u;
        uint4 e0 = make_uint4(0, 0, 0, 0), e1 = e0;
        bool found = a.has4 != 0;
        if (found && a.g4x_mode)
        {
            const uint4 *b = a.g4x + 4 * (size_t)(((T[3] * a.g4x_mul) >> 9) & a.g4x_mask);
            const uint4 q0 = b[0], q1 = b[1], q2 = b[2], q3 = b[3];
            const bool hit0 = q0.y != 0u && q0.x == T[3], hit1 = q2.y != 0u && q2.x == T[3];
            e0 = hit0 ? q0 : q2;
            e1 = hit0 ? q1 : q3;
            found = hit0 || hit1;
        }
        else if (found)
            for (u32 h = (T[3] * kHashMul) >> 9;; ++h)
            {
                const uint4 *e = a.g4x + 2 * (size_t)(h & a.g4mask);
                e0 = e[0];
                e1 = e[1]; // issued with e[0]: one latency
                if (e0.y == 0u)
                {
                    found = false; // not a suffix of any pattern
                    break;
                }
                if (e0.x == T[3])
                    break;
            }
        ac_eval_entry(a, found, T, e0, e1, sb, i, own_by_end, dm, slow);
    }
    if (slow)
        return ac_walk_slow<CI, SHORT>(a, i, own_by_end, depthmask, simple);
    depthmask = dm;
    simple = true;
    return (u32)__popc(dm);
}

// The same for the two end positions i and i + 1 of a stride-2 candidate: both text windows and both table probes
// are in flight together (one latency for the pair).  No level walk in here: an end that needs it comes back with
// slow = true and the caller runs ac_walk_slow from its single call site.
// gtest(E): is the class gram of the four text bytes E in the filter table the caller holds in LDS?  (round 5)  A candidate is a
// tested position t whose gram is in the table — as some pattern's final gram (a match may end at t) or as some pattern's gram
// one byte earlier (a match may end at t + 1).  The table cannot say which, but it can be asked twice more, for free, once the
// text window is in registers: a match that ends at t also put ITS gram-one-byte-earlier into the table, i.e. the gram of
// t - 1; a match that ends at t + 1 put its final gram there, i.e. the gram of t + 1.  An end is probed only where its second
// gram is present as well — about one end in ten, where both 64-byte buckets of every candidate used to be read: they were
// 0.8 ms of the 6.5 on BASELINE config 4 (700 M L2 requests against the stream's 270 M, most onto a few thousand hot lines;
// profiles/r05_ac1000_where_the_time_goes.txt).
// STAGED (the caller has a real gtest): the eight bytes i - 6 .. i + 1 come first — ONE request per candidate, and all that both
// gram tests need; the 16-byte window in front of them is fetched only by the lanes an end survives in, together with their
// buckets (the same two dependent round trips as before, a third of the requests).
template <bool CI, bool SHORT, bool STAGED, bool WW, typename GTest>
__device__ __forceinline__ void ac_walk_probe2(const AcArgs &a, u64 i, bool liveA, bool liveB, bool own_by_end,
                                               u32 &dmA, bool &slowA, u32 &dmB, bool &slowB, GTest gtest)
{
    dmA = dmB = 0;
    slowA = slowB = false;
    // (-w no longer sends every candidate to the level walk, round 6: the depth masks the entries answer with are filtered by the matches'
    //  neighbours at the end of this function — BASELINE config 4 with -w: 15.0 ms for 32 GiB where the plain scan takes 6.5)
    // (WW: a template parameter of the kernel, not the run-time flag — see ac_scan_kernel)
    if (i < 15 || (a.flags & ((WW ? 0u : F_WW) | (SHORT ? F_AC_SHORT_DUP : 0u))))
    {
        slowA = liveA;
        slowB = liveB;
        return;
    }
    // ONE 17-byte window serves both ends: A = its first 16 bytes (the text up to i), B = the same one byte further (up to
    // i + 1) by funnel shifts.  Only the byte text[i + 1] is read behind A, and only when it exists (liveB): nothing is
    // touched past the end of the buffer.
    struct __attribute__((packed)) U32p { u32 v; };
    struct __attribute__((packed)) U64p { u64 v; };
    u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
    if constexpr (STAGED)
    {
        // bytes i - 6 .. i + 1 (without the last one where it does not exist: one byte lower, shifted back)
        const u64 q8 = reinterpret_cast<const U64p *>(a.text + (i - (liveB ? 6u : 7u)))->v;
        const u64 Q = liveB ? q8 : (q8 >> 8);
        w3 = (u32)(Q >> 24);                                // bytes i - 3 .. i
        w4 = (u32)(Q >> 56);                                // byte i + 1
        if (liveA)
            liveA = gtest((u32)(Q >> 16));                  // bytes i - 4 .. i - 1 (classes ignore the case: asked before the fold)
        if (liveB)
            liveB = gtest((u32)(Q >> 32));                  // bytes i - 2 .. i + 1
        if (liveA || liveB)
        {
            const U32p *qa = reinterpret_cast<const U32p *>(a.text + (i - 15));
            w0 = qa[0].v; w1 = qa[1].v; w2 = qa[2].v;
        }
    }
    else
    {
        const U32p *qa = reinterpret_cast<const U32p *>(a.text + (i - 15));
        w0 = qa[0].v; w1 = qa[1].v; w2 = qa[2].v; w3 = qa[3].v;
        w4 = liveB ? (u32)a.text[i + 1] : 0u;
        if (liveA)
            liveA = gtest(__builtin_amdgcn_alignbyte(w3, w2, 3));
        if (liveB)
            liveB = gtest(__builtin_amdgcn_alignbyte(w4, w3, 1));
    }
    u32 TA[4] = {w0, w1, w2, w3};
    u32 TB[4] = {w0, w1, w2, w3};
    if (liveB)
    {
        TB[0] = __builtin_amdgcn_alignbyte(w1, w0, 1);
        TB[1] = __builtin_amdgcn_alignbyte(w2, w1, 1);
        TB[2] = __builtin_amdgcn_alignbyte(w3, w2, 1);
        TB[3] = __builtin_amdgcn_alignbyte(w4, w3, 1);
    }
    if (CI)
    {
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            TA[w] = ac_fold4(TA[w]);
            TB[w] = ac_fold4(TB[w]);
        }
    }
    const u32 sbA = SHORT ? ac_short_bits(a, TA[3]) : 0u, sbB = SHORT ? ac_short_bits(a, TB[3]) : 0u;
    u32 hA = (TA[3] * kHashMul) >> 9, hB = (TB[3] * kHashMul) >> 9;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;
    bool doneA = !liveA || !a.has4, doneB = !liveB || !a.has4, foundA = false, foundB = false;
#if defined(KG_AC_ABLATE) && KG_AC_ABLATE == 1 // (measurement build: the text windows only — no table probe, no evaluation)
    dmA = (liveA && TA[3] == 0x12345678u) ? 16u : 0u;
    dmB = (liveB && TB[3] == 0x12345678u) ? 16u : 0u;
    return;
#endif
    if (a.g4x_mode)
    {
        const uint4 *ba = a.g4x + 4 * (size_t)(((TA[3] * a.g4x_mul) >> 9) & a.g4x_mask);
        const uint4 *bb = a.g4x + 4 * (size_t)(((TB[3] * a.g4x_mul) >> 9) & a.g4x_mask);
        uint4 x0 = make_uint4(0, 0, 0, 0), x1 = x0, x2 = x0, x3 = x0, y0 = x0, y1 = x0, y2 = x0, y3 = x0;
        if (!doneA) // (a lane whose end is dead issues no requests for it; the buckets of both ends are still in flight together)
        {
            x0 = ba[0]; x1 = ba[1]; x2 = ba[2]; x3 = ba[3];
        }
        if (!doneB)
        {
            y0 = bb[0]; y1 = bb[1]; y2 = bb[2]; y3 = bb[3];
        }
        const bool ha0 = x0.y != 0u && x0.x == TA[3], ha1 = x2.y != 0u && x2.x == TA[3];
        const bool hb0 = y0.y != 0u && y0.x == TB[3], hb1 = y2.y != 0u && y2.x == TB[3];
        a0 = ha0 ? x0 : x2; a1 = ha0 ? x1 : x3;
        b0 = hb0 ? y0 : y2; b1 = hb0 ? y1 : y3;
        foundA = !doneA && (ha0 || ha1);
        foundB = !doneB && (hb0 || hb1);
        doneA = doneB = true;
    }
    while (!(doneA && doneB))
    {
        const uint4 *ea = a.g4x + 2 * (size_t)(hA & a.g4mask), *eb = a.g4x + 2 * (size_t)(hB & a.g4mask);
        const uint4 x0 = ea[0], x1 = ea[1], y0 = eb[0], y1 = eb[1];
        if (!doneA)
        {
            a0 = x0; a1 = x1;
            if (x0.y == 0u) doneA = true;
            else if (x0.x == TA[3]) doneA = foundA = true;
            else ++hA;
        }
        if (!doneB)
        {
            b0 = y0; b1 = y1;
            if (y0.y == 0u) doneB = true;
            else if (y0.x == TB[3]) doneB = foundB = true;
            else ++hB;
        }
    }
#if defined(KG_AC_ABLATE) && KG_AC_ABLATE == 2 // (measurement build: windows + both buckets, no evaluation)
    dmA = (liveA && (a0.x ^ a1.y ^ b0.z ^ b1.w) == 0x12345678u) ? 16u : 0u;
    return;
#endif
    if (liveA)
        ac_eval_entry(a, foundA, TA, a0, a1, sbA, i, own_by_end, dmA, slowA);
#if defined(KG_AC_ABLATE) && KG_AC_ABLATE == 3 // (measurement build: the second end is not evaluated)
    return;
#endif
    if (liveB)
        ac_eval_entry(a, foundB, TB, b0, b1, sbB, i + 1, own_by_end, dmB, slowB);
    if constexpr (WW)
    {
        // -w (is_whole_word_match krep.h:312-319) on what matched: bit d of a mask is a pattern of d bytes ending here — the byte behind the
        // end decides for all of them, the byte in front of each is byte 15 - d of the window (d >= 16: one load).  An end that takes the
        // level walk (slow) is filtered there.
        auto wwf = [&](u32 m, const u32 (&T)[4], u64 e) -> u32 {
            if (!m)
                return 0u;
            if (e + 1 < a.text_len && ac_wordc(a.text[e + 1]))
                return 0u;
            u32 r = m;
            for (u32 rest = m; rest;)
            {
                const u32 d = 31u - (u32)__builtin_clz(rest);
                rest &= ~(1u << d);
                u32 c = 0;
                if (d <= 15u)
                {
                    const u32 at = 15u - d, w = at >> 2;
                    c = ((w == 0u ? T[0] : w == 1u ? T[1] : w == 2u ? T[2] : T[3]) >> (8u * (at & 3u))) & 0xffu;
                }
                else if (e >= d)
                    c = a.text[e - d];
                if (ac_wordc(c))
                    r &= ~(1u << d);
            }
            return r;
        };
        if (!slowA)
            dmA = wwf(dmA, TA, i);
        if (!slowB)
            dmB = wwf(dmB, TB, i + 1);
    }
}

// ---- the length-keyed exact dictionary (stage 3 of the anchored scan, kg_ac_anchor.hip) ----
// A word dictionary's reversed trie branches right behind its final grams (`tion`, `ness` end dozens of words each), which is exactly what
// the chain-compressed entries cannot express: nearly every marked end took the level-by-level walk, ~12 dependent L2 round trips, and
// stage 3 cost 7.7 ms for 12 M ends.  Here an end is answered per LENGTH: the (hashed) final gram names the lengths that can end on it, and
// for each of them the text's last l bytes are looked up whole.  Longest first is the order of the loop.
__host__ __device__ __forceinline__ u32 ac_xhash(u32 w0, u32 w1, u32 w2, u32 w3, u32 len, u32 mul)
{ // (w0..w3: the 16 bytes ending at the end position, the bytes in front of the pattern zeroed)
    u32 h = w3 * mul;
    h = (h ^ (h >> 15)) + w2 * 0x85EBCA6Bu;
    h = (h ^ (h >> 13)) + w1 * 0xC2B2AE35u;
    h = (h ^ (h >> 16)) + w0 * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) + len * 0x165667B1u;
    return h ^ (h >> 14);
}
__host__ __device__ __forceinline__ u32 ac_xlen_slot(u32 last4) { return (last4 * kHashMul) >> 16; }
__host__ __device__ __forceinline__ u32 ac_xlen_slot8(u32 w2, u32 w3)
{
    const u32 h = w3 * kHashMul + w2 * 0x85EBCA6Bu;
    return (h ^ (h >> 15)) >> 16;
}
// depth mask of the patterns ending at i (i >= 15) from the exact dictionary; `multi` when one of them occurs more than once in the
// dictionary (the caller then counts and emits through the level walk, which knows the copies)
template <bool CI>
__device__ __forceinline__ u32 ac_exact_end(const AcArgs &a, u64 i, bool &multi)
{
    struct __attribute__((packed)) U32p { u32 v; };
    const U32p *q = reinterpret_cast<const U32p *>(a.text + (i - 15));
    u32 T[4] = {q[0].v, q[1].v, q[2].v, q[3].v};
    if (CI)
    {
#pragma unroll
        for (int w = 0; w < 4; ++w)
            T[w] = ac_fold4(T[w]);
    }
    u32 lm = ((u32)a.xlen[ac_xlen_slot(T[3])] & 0xfu) | ((u32)a.xlen[65536u + ac_xlen_slot8(T[2], T[3])] & 0x1ff0u);
    u32 dm = 0;
    multi = false;
    while (lm)
    {
        const u32 b = 31u - (u32)__builtin_clz(lm), len = b + 4u;
        lm &= ~(1u << b);
        // keep the last `len` bytes of the 16: word w holds bytes 4w .. 4w + 3, the pattern starts at byte 16 - len
        const u32 drop = 16u - len; // leading bytes that are not the pattern's
        u32 M[4];
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            const u32 lo = 4u * (u32)w; // first byte of the word
            M[w] = drop >= lo + 4u ? 0u : drop <= lo ? T[w] : (T[w] & (0xffffffffu << (8u * (drop - lo))));
        }
        const uint4 *bk = a.xtab + 4u * (size_t)(ac_xhash(M[0], M[1], M[2], M[3], len, a.xmul) & a.xmask);
        const uint4 e0 = bk[0], m0 = bk[1], e1 = bk[2], m1 = bk[3];
        const bool h0 = m0.x == len && e0.x == M[0] && e0.y == M[1] && e0.z == M[2] && e0.w == M[3];
        const bool h1 = m1.x == len && e1.x == M[0] && e1.y == M[1] && e1.z == M[2] && e1.w == M[3];
        if (h0 || h1)
        {
            dm |= 1u << len;
            multi = multi || (h0 ? m0.y : m1.y) != 1u;
        }
    }
    return dm;
}

// The same for BOTH ends of a marked pair, i and i + 1 (stage 3 of the anchored scan), interleaved: one 17-byte window, the four length
// masks requested together, then per step the longest remaining length of EACH end — both buckets in flight at once.  Two calls of
// ac_exact_end run their round trips one behind the other (2 + nA, then 2 + nB); this runs 2 + max(nA, nB).
template <bool CI, bool WW>
__device__ __forceinline__ void ac_exact_end2(const AcArgs &a, u64 i, bool liveA, bool liveB, u32 &dmA, u32 &dmB, bool &multiA, bool &multiB)
{
    struct __attribute__((packed)) U32p { u32 v; };
    constexpr bool ww = WW; // -w (is_whole_word_match krep.h:312-319): a match needs a non-word byte (or the text's edge) on either side
    const U32p *q = reinterpret_cast<const U32p *>(a.text + (i - 15));
    u32 TA[4] = {q[0].v, q[1].v, q[2].v, q[3].v};
    const u32 nxt = ((liveB || ww) && i + 1 < a.text_len) ? (u32)a.text[i + 1] : 0u; // (a live end i + 1 lies inside the text)
    // -w: the byte behind end i + 1, and the byte in front of a 16-byte pattern that ends at i (the other neighbours are in the window)
    const u32 nxt2 = (ww && liveB && i + 2 < a.text_len) ? (u32)a.text[i + 2] : 0u;
    const u32 prv = (ww && liveA && i >= 16u) ? (u32)a.text[i - 16] : 0u;
    u32 TB[4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
        TB[w] = __builtin_amdgcn_alignbyte(w < 3 ? TA[w < 3 ? w + 1 : 3] : nxt, TA[w], 1u);
    const u32 firstA = TA[0] & 0xffu; // (text[i - 15]: the byte in front of a 16-byte pattern that ends at i + 1; before any fold)
    // (the neighbour tests read the unfolded window's classes: folding maps letters to letters, so either copy will do; the folded one is at hand)
    if (CI)
    {
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            TA[w] = ac_fold4(TA[w]);
            TB[w] = ac_fold4(TB[w]);
        }
    }
    const u32 a4 = a.xlen[ac_xlen_slot(TA[3])], a8 = a.xlen[65536u + ac_xlen_slot8(TA[2], TA[3])];
    const u32 b4 = a.xlen[ac_xlen_slot(TB[3])], b8 = a.xlen[65536u + ac_xlen_slot8(TB[2], TB[3])];
    u32 lmA = liveA ? ((a4 & 0xfu) | (a8 & 0x1ff0u)) : 0u, lmB = liveB ? ((b4 & 0xfu) | (b8 & 0x1ff0u)) : 0u;
    if (ww)
    { // a word character behind the end: no pattern ends here as a whole word
        if (ac_wordc(nxt))
            lmA = 0u;
        if (ac_wordc(nxt2))
            lmB = 0u;
    }
    dmA = dmB = 0;
    multiA = multiB = false;
    auto keep = [](const u32 (&T)[4], u32 len, u32 (&M)[4]) { // the last `len` bytes of the 16, the bytes in front zeroed
        const u32 drop = 16u - len;
#pragma unroll
        for (int w = 0; w < 4; ++w)
        {
            const u32 lo = 4u * (u32)w;
            M[w] = drop >= lo + 4u ? 0u : drop <= lo ? T[w] : (T[w] & (0xffffffffu << (8u * (drop - lo))));
        }
    };
    // -w: is the byte in front of a pattern of `len` bytes that ends the window a word character?  (byte 15 - len of the window; a pattern of
    // 16 bytes: `far`, the byte in front of the window — 0 where the text starts there)
    auto left_is_word = [](const u32 (&T)[4], u32 len, u32 far) -> bool {
        if (len >= 16u)
            return ac_wordc(far);
        const u32 at = 15u - len, w = at >> 2;
        const u32 word = w == 0u ? T[0] : w == 1u ? T[1] : w == 2u ? T[2] : T[3];
        return ac_wordc((word >> (8u * (at & 3u))) & 0xffu);
    };
    while (lmA | lmB)
    {
        const bool doA = lmA != 0u, doB = lmB != 0u;
        const u32 bA = doA ? 31u - (u32)__builtin_clz(lmA) : 0u, bB = doB ? 31u - (u32)__builtin_clz(lmB) : 0u;
        const u32 lenA = bA + 4u, lenB = bB + 4u;
        lmA &= ~(1u << bA);
        lmB &= ~(1u << bB);
        u32 MA[4], MB[4];
        keep(TA, lenA, MA);
        keep(TB, lenB, MB);
        const uint4 *bkA = a.xtab + 4u * (size_t)(ac_xhash(MA[0], MA[1], MA[2], MA[3], lenA, a.xmul) & a.xmask);
        const uint4 *bkB = a.xtab + 4u * (size_t)(ac_xhash(MB[0], MB[1], MB[2], MB[3], lenB, a.xmul) & a.xmask);
        uint4 eA0 = make_uint4(0, 0, 0, 0), mA0 = eA0, eA1 = eA0, mA1 = eA0, eB0 = eA0, mB0 = eA0, eB1 = eA0, mB1 = eA0;
        if (doA)
        {
            eA0 = bkA[0]; mA0 = bkA[1]; eA1 = bkA[2]; mA1 = bkA[3];
        }
        if (doB)
        {
            eB0 = bkB[0]; mB0 = bkB[1]; eB1 = bkB[2]; mB1 = bkB[3];
        }
        if (doA)
        {
            const bool h0 = mA0.x == lenA && eA0.x == MA[0] && eA0.y == MA[1] && eA0.z == MA[2] && eA0.w == MA[3];
            const bool h1 = mA1.x == lenA && eA1.x == MA[0] && eA1.y == MA[1] && eA1.z == MA[2] && eA1.w == MA[3];
            const bool mu = (h0 ? mA0.y : mA1.y) != 1u; // (copies != 1: the level walk answers this end, -w included — the entry may be a longer pattern's tail)
            if ((h0 || h1) && (mu || !(ww && left_is_word(TA, lenA, prv))))
            {
                dmA |= 1u << lenA;
                multiA = multiA || mu;
            }
        }
        if (doB)
        {
            const bool h0 = mB0.x == lenB && eB0.x == MB[0] && eB0.y == MB[1] && eB0.z == MB[2] && eB0.w == MB[3];
            const bool h1 = mB1.x == lenB && eB1.x == MB[0] && eB1.y == MB[1] && eB1.z == MB[2] && eB1.w == MB[3];
            const bool mu = (h0 ? mB0.y : mB1.y) != 1u;
            if ((h0 || h1) && (mu || !(ww && left_is_word(TB, lenB, firstA))))
            {
                dmB |= 1u << lenB;
                multiB = multiB || mu;
            }
        }
    }
}

constexpr u32 kAcUnitsPerTicketMax = 8; // fused kernel: up to 128 KiB per wave ticket (one cold round in 16), fewer on small texts
constexpr int kAcRounds = 2;            // load rounds per unit: one candidate drain per 16 KiB (53 of 64 lanes busy)
constexpr u32 kAcUnitBytes = kAcRounds * kSegBytes;
constexpr u32 kAcBitmapWords = kAcUnitBytes / 32; // one bit per end position of a unit (LINES)

// Exact-class filter index of the fused kernel when every pattern has >= 4 bytes: class(b) = b & 31 (letters keep
// their identity, upper/lower case share a class, so the table needs no case fold), index = the four classes of the
// suffix 4-gram, 5 bits each, first byte lowest.  2^20 bits (128 KiB) in LDS; -c keeps 2^19 (top bit dropped) to
// leave room for the per-wave line bitmaps.  A superset test like the hashed tables, but for text over one
// 5-bit-class alphabet a hit IS a suffix 4-gram match: no hash false positives, and no multiply per position.
constexpr u32 kXBitsBig = 20, kXBitsLines = 19;
__host__ __device__ __forceinline__ u32 ac_cls4(u32 w)
{ // bytes b0..b3 of w -> c(b0) | c(b1) << 5 | c(b2) << 10 | c(b3) << 15
    const u32 t = (w & 0x001f001fu) | ((w >> 3) & ~0x001f001fu); // two v_bfi: junk above each 10-bit pair ...
    return ((t & 0x3ffu) | ((t >> 6) & ~0x3ffu)) & 0xfffffu;       // ... shifted out or masked here
}


// Pair layout of the stride-2 filter (kernel without -c): the ODD text positions are tested, so the suffix 4-gram of a
// tested position is two 16-bit-aligned byte pairs of the lane's dwords.
// ac_pair: dword -> {c(b0) | c(b1) << 5} in the low half, {c(b2) | c(b3) << 5} in the high half (bits 10-15 of each half 0).
__host__ __device__ __forceinline__ u32 ac_pair(u32 w) { return (w & 0x001f001fu) | ((w >> 3) & 0x03e003e0u); }
// Table slot of the gram with classes x = c0 | c1 << 5 | c2 << 10 | c3 << 15 (c3 = the tested position): chosen so that the
// kernel gets it from the pair register u = {c0, c1 | c2, c3} with three VALU — bit = c0 = u & 31 (taken by the shifter),
// byte address = ((u >> 3) ^ (u >> 13)) & 0x1fffc, i.e. dword = (c1 ^ (c2 & 15) << 1) | (c2 >> 4) << 5 | c3 << 6 | (c2 & 15) << 11:
// a bijection of (c1, c2, c3) whose low five bits (the LDS bank) mix two classes.
__host__ __device__ __forceinline__ void ac_pair_slot(u32 x, u32 &dword, u32 &bit)
{
    const u32 c1 = (x >> 5) & 31u, c2 = (x >> 10) & 31u, c3 = (x >> 15) & 31u;
    bit = x & 31u;
    dword = (c1 ^ ((c2 & 15u) << 1)) | ((c2 >> 4) << 5) | (c3 << 6) | ((c2 & 15u) << 11);
}
// The same with the class e of the byte in FRONT of the gram (five-class index of the anchored scan, ANCH == 2): e is multiplied out over
// the class fields c1, c2, c3 of the pair register and XOR-ed in before the slot address is formed (the bit index stays c0).  Measured on word
// text (CPU model of the filter, 1000 rare words): e in the register's five unused bits — next to c3 in the address — left the candidates at
// 2.0 % of the positions, as if there were no fifth class: the grams that then share a slot differ in e and c3 only and share the 3-gram
// c0 c1 c2, which is where a text's mass is.  Spread over all three fields the colliding grams are unrelated ones: 0.55 %, what a uniform hash gives.
constexpr u32 kAnch5Mul = 0x3779B1u, kAnch5Mask = (0x1fu << 5) | (0x1fu << 16) | (0x1fu << 21);
__host__ __device__ __forceinline__ u32 ac_mix5(u32 u, u32 e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return u ^ (__umul24(e, kAnch5Mul) & kAnch5Mask);
#else
    return u ^ ((e * kAnch5Mul) & kAnch5Mask);
#endif
}
__host__ __device__ __forceinline__ void ac_pair_slot5(u32 x, u32 e, u32 &dword, u32 &bit)
{
    const u32 u = ac_mix5((x & 0x3ffu) | (((x >> 10) & 0x3ffu) << 16), e & 31u);
    bit = u & 31u;
    dword = (((u >> 3) ^ (u >> 13)) & 0x1fffcu) >> 2;
}

} // namespace kg
