// kg_ac.hip — multi-pattern literal scan for gfx950, replacing aho_corasick_search
// (aho_corasick.c:299-466) behind the same search_func_t boundary.
//
// What the reference computes: for every text index i (ascending), every pattern that ENDS at i,
// longest first (it walks the goto/fail automaton and then the whole fail chain,
// aho_corasick.c:353-431), duplicates of a pattern string once per copy.  That set is a pure
// function of the bytes text[i-Lmax+1 .. i], so the sequential automaton walk is not needed to
// reproduce it.  MI355X-first formulation:
//   * FILTER (every byte, HBM-rate): the last min(len,4) bytes of every pattern are hashed into bit
//     tables resident in LDS (<= 88 KiB: 256 b for 1-byte patterns, 64 Kib for 2-byte, 128 Kib hashed
//     for 3-byte, 512 Kib hashed for >= 4-byte patterns).  A lane tests its 16 end positions with one
//     v_alignbyte + multiply-shift + ds_read_b32 each; positions that miss every table cannot end a
//     pattern.  The haystack is read once with the same coalesced 16 B/lane loads as the literal scan.
//   * VERIFY (candidates only, ~0.4 % of positions for 1000 random patterns): walk the REVERSED-pattern
//     trie backwards from i (edges in an open-addressing table in global memory, L2-resident); each
//     node on the path that is a pattern end yields its copies.  Walking visits lengths ascending; the
//     reference order (longest first) falls out of writing slot = base + (total_i - seen - copies).
//   * ORDER: same as the literal kernel — unit-local ranks, staging slot, info word, post-pass.
// -w, -c (line counting), max_count and start-offset ownership are applied exactly as in the literal
// kernel.  A dense DFA in LDS is impossible for the benchmark set (8605 states x 256 x 2 B = 4.4 MB
// against 160 KiB), which is why the LDS holds the filter, not the automaton.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

constexpr int kAcBlock = 1024;             // 16 waves share one copy of the filter tables in LDS
constexpr int kAcWaves = kAcBlock / 64;
constexpr u32 kT1Words = 256 / 32;         // 1-byte patterns: direct
constexpr u32 kT2Words = 65536 / 32;       // 2-byte patterns: direct (8 KiB)
constexpr u32 kT3Bits = 17, kT3Words = (1u << kT3Bits) / 32; // 3-byte patterns: hashed (16 KiB)
constexpr u32 kT4Bits = 19, kT4Words = (1u << kT4Bits) / 32; // >= 4-byte patterns: hashed (64 KiB)
constexpr u32 kHashMul = 0x9E3779B1u;

struct AcArgs
{
    const uint8_t *text;
    u64 text_len, own_lo, own_hi, anchor, num_tiles, global_base;
    u64 end_lo, end_hi;          // range of END indices this launch examines
    u32 flags;                   // F_CI | F_WW | F_POS | F_LINES
    u32 lmax;
    u32 has1, has2, has3, has4;  // which length classes exist
    const u32 *filter;           // T1 | T2 | T3 | T4 (only the present ones, in this order)
    u32 off2, off3, off4, filter_words;
    const uint2 *edges;          // open addressing: {key = node << 8 | byte, val = child | has_out << 31}
    u32 emask;
    const u32 *copies;           // per node: number of patterns equal to the node's string
    unsigned long long *unitinfo;
    Counters *ctr;
    u64 *stage;
    u32 stage_cap;
    u32 emit_mode;
    const u64 *offsets;
    u64 *positions;
    u64 pos_cap;
};

__device__ __forceinline__ u32 ac_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u64 ac_rfl64(u64 v)
{
    return ((u64)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)v);
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
__device__ __forceinline__ LS2 ls2_combine(const LS2 &a, const LS2 &b)
{
    return LS2{a.cnt + b.cnt - ((a.tail && b.head) ? 1u : 0u), a.nl || b.nl, a.nl ? a.head : (a.head || b.head),
               b.nl ? b.tail : (a.tail || b.tail)};
}

// Walk the reversed trie from end index i.  EMIT == false: returns the number of matches ending at i
// (after ownership and -w).  EMIT == true: additionally writes them, longest first, at slot[base ...].
template <bool CI, bool EMIT, typename Put>
__device__ __forceinline__ u32 ac_walk(const AcArgs &a, u64 i, u32 total, Put put)
{
    u32 node = 0, seen = 0;
    const bool ww = (a.flags & F_WW) != 0, lines = (a.flags & F_LINES) != 0;
    const u64 maxd = (i + 1 < (u64)a.lmax) ? i + 1 : (u64)a.lmax;
    for (u64 d = 1; d <= maxd; ++d)
    {
        u32 c = a.text[i + 1 - d];
        if (CI && (c - 'A' < 26u))
            c += 32u;
        const u32 key = (node << 8) | c;
        u32 h = (key * kHashMul) >> 7;
        u32 child = 0xffffffffu;
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

template <bool CI, bool LINES>
__global__ __launch_bounds__(kAcBlock) void ac_scan_kernel(const AcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_filter[]; // filter tables, then 2 ticket words
    const u32 lane = ac_lane();
    const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (u32 w = threadIdx.x; w < a.filter_words; w += kAcBlock)
        s_filter[w] = a.filter[w];
    u64 *s_ticket = reinterpret_cast<u64 *>(s_filter + ((a.filter_words + 3u) & ~3u));
    const bool want_pos = (a.flags & F_POS) != 0;
    const bool chain = want_pos || LINES;

    u64 acc_total = 0;
    u64 next_ticket = 0;
    if (threadIdx.x == 0)
        next_ticket = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (u32 it = 0;; ++it)
    {
        if (threadIdx.x == 0)
            s_ticket[it & 1u] = next_ticket;
        __syncthreads(); // also orders the filter-table fill before its first use
        const u64 tile = ac_rfl64(s_ticket[it & 1u]);
        if (tile >= a.num_tiles)
            break;
        if (threadIdx.x == 0)
            next_ticket = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        const u64 unit = tile * kAcWaves + wave;
        const u64 seg = a.anchor + unit * (u64)kSegBytes;
        if (a.emit_mode && (u32)(a.unitinfo[unit] & kUiCountMask) <= a.stage_cap)
            continue;
        const bool fast = seg + kSegBytes <= a.text_len;
        const bool interior = seg >= a.end_lo && seg + kSegBytes <= a.end_hi;

        uint4 d[kCells];
        u32 before = 0; // the 4 bytes in front of the segment
        if (fast)
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
#pragma unroll
            for (int j = 0; j < kCells; ++j)
                d[j] = src[j * kWave];
        }
        if (seg >= 4 && seg - 4 + 4 <= a.text_len)
            before = *reinterpret_cast<const u32 *>(a.text + seg - 4);
        else
            for (u32 b = 0; b < 4; ++b)
                if (seg + b >= 4 && seg + b - 4 < a.text_len)
                    before |= (u32)a.text[seg + b - 4] << (8 * b);

        u32 CM[kCells]; // per lane: low 16 bits = end positions with >= 1 match, high 16 = matches of the lane
        u32 wcnt = 0;
        LS2 wls{0, false, false, false};

#pragma unroll
        for (int j = 0; j < kCells; ++j)
        {
            const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
            u32 W[5]; // W[0] = the 4 bytes before the lane, W[1..4] = the lane's 16 bytes
            if (fast)
            {
                W[1] = d[j].x; W[2] = d[j].y; W[3] = d[j].z; W[4] = d[j].w;
                const u32 up = __shfl_up(W[4], 1);
                const u32 edge = (j == 0) ? before : __builtin_amdgcn_readlane(d[j > 0 ? j - 1 : 0].w, 63);
                W[0] = (lane == 0u) ? edge : up;
            }
            else
            {
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
            }
            u32 NL = 0;
            if (LINES)
            {
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    NL |= ac_movemask4(ac_eq_bytes(W[w + 1], 0x0a0a0a0au)) << (4 * w);
            }
            if (CI)
            {
#pragma unroll
                for (int w = 0; w < 5; ++w)
                    W[w] = ac_fold4(W[w]);
            }

            // ---- filter: which of my 16 end positions can end a pattern? -----------------------------
            u32 cand = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k)
            {
                // E = bytes [k-3, k] of the lane (little endian: the byte at k is the top byte)
                const int o = k + 1;
                const u32 E = ((o & 3) == 0) ? W[o >> 2] : __builtin_amdgcn_alignbyte(W[(o >> 2) + 1], W[o >> 2], (u32)(o & 3));
                bool hit = false;
                if (a.has4)
                {
                    const u32 h = (E * kHashMul) >> (32 - kT4Bits);
                    hit = (s_filter[a.off4 + (h >> 5)] >> (h & 31u)) & 1u;
                }
                if (a.has3)
                {
                    const u32 h = ((E >> 8) * kHashMul) >> (32 - kT3Bits);
                    hit = hit || ((s_filter[a.off3 + (h >> 5)] >> (h & 31u)) & 1u);
                }
                if (a.has2)
                {
                    const u32 h = E >> 16;
                    hit = hit || ((s_filter[a.off2 + (h >> 5)] >> (h & 31u)) & 1u);
                }
                if (a.has1)
                {
                    const u32 h = E >> 24;
                    hit = hit || ((s_filter[h >> 5] >> (h & 31u)) & 1u);
                }
                cand |= hit ? (1u << k) : 0u;
            }
            u32 nlm = NL;
            if (!interior)
            {
                auto clip = [&](u64 lo, u64 hi) -> u32 {
                    u32 klo = lo > lbase ? (u32)((lo - lbase) < 16 ? (lo - lbase) : 16) : 0u;
                    u32 khi = hi > lbase ? (u32)((hi - lbase) < 16 ? (hi - lbase) : 16) : 0u;
                    return khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                };
                cand &= clip(a.end_lo, a.end_hi);
                if (LINES)
                    nlm &= clip(a.own_lo, a.own_hi);
            }

            // ---- verify the candidates --------------------------------------------------------------
            u32 hits = 0, lcnt = 0;
            if (__ballot(cand != 0u))
            {
                u32 rest = cand;
                while (rest)
                {
                    const u32 k = __builtin_ctz(rest);
                    rest &= rest - 1u;
                    const u32 c = ac_walk<CI, false>(a, lbase + k, 0u, [](u32, u64, u32) {});
                    if (c)
                    {
                        hits |= 1u << k;
                        lcnt += c;
                    }
                }
            }
            CM[j] = hits | (lcnt << 16);
            const u64 anyhit = __ballot(hits != 0u);
            if (anyhit)
            {
                u32 v = lcnt;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1)
                    v += __shfl_xor(v, o);
                wcnt += v;
            }

            if (LINES)
            {
                const u32 H = hits, N = nlm;
                const bool l_nl = N != 0u;
                const u64 B_nl = __ballot(l_nl);
                LS2 cell{0, B_nl != 0, false, false};
                if (anyhit)
                {
                    const u32 S = ((N << 1) | 1u) & 0xffffu, Hs = H | N;
                    const u32 firsts = H & Hs & ~(Hs - S);
                    bool l_head, l_tail;
                    if (l_nl)
                    {
                        const u32 lo_nl = N & (0u - N);
                        l_head = (H & (lo_nl | (lo_nl - 1u))) != 0u;
                        l_tail = (H >> (32 - __builtin_clz(N))) != 0u;
                    }
                    else
                        l_head = l_tail = H != 0u;
                    const u64 B_any = anyhit, B_tail = __ballot(l_tail), B_head = __ballot(l_head);
                    const u64 lt = (1ull << lane) - 1ull;
                    const u64 nl_below = B_nl & lt;
                    bool open;
                    if (nl_below)
                    {
                        const int q = 63 - __builtin_clzll(nl_below);
                        open = ((B_tail >> q) & 1ull) || (B_any & lt & ~((2ull << q) - 1ull)) != 0;
                    }
                    else
                        open = (B_any & lt) != 0;
                    u32 lc = __popc(firsts) - ((open && l_head) ? 1u : 0u);
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1)
                        lc += __shfl_xor(lc, o);
                    cell.cnt = lc;
                    if (cell.nl)
                    {
                        const int f = __builtin_ctzll(B_nl), l = 63 - __builtin_clzll(B_nl);
                        cell.head = (B_any & ((1ull << f) - 1ull)) != 0 || ((B_head >> f) & 1ull);
                        cell.tail = (l < 63 && (B_any >> (l + 1)) != 0) || ((B_tail >> l) & 1ull);
                    }
                    else
                        cell.head = cell.tail = true;
                }
                wls = ls2_combine(wls, cell);
            }
        }

        acc_total += wcnt;
        if (!chain)
            continue;

        // ---- publish / emit ----------------------------------------------------------------------------
        const bool emit_final = a.emit_mode != 0;
        if (!emit_final && lane == 0)
        {
            u64 info = (u64)wcnt;
            if (LINES)
                info |= (wls.nl ? kLnNl : 0) | (wls.head ? kLnHead : 0) | (wls.tail ? kLnTail : 0) |
                        ((u64)(wls.cnt & kUiLineMask) << kUiLineShift);
            else if (wcnt)
                info |= kLnHead | kLnTail;
            a.unitinfo[unit] = info;
            if (want_pos && wcnt > a.stage_cap)
            {
                atomicAdd(&a.ctr->overflow_units, 1ull);
                atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
            }
        }
        const bool do_stage = !emit_final && want_pos && wcnt != 0;
        const bool do_final = emit_final && want_pos && wcnt > a.stage_cap;
        if (do_stage || do_final)
        {
            u64 *slot = reinterpret_cast<u64 *>(a.stage) + unit * (u64)a.stage_cap;
            const u64 fbase = do_final ? a.offsets[unit] : 0ull;
            u32 out = 0;
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                u32 hits = CM[j] & 0xffffu;
                const u32 lcnt = CM[j] >> 16;
                if (!__ballot(hits != 0u))
                    continue;
                // exclusive prefix of lcnt over lanes
                u32 incl = lcnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1)
                {
                    const u32 t = __shfl_up(incl, o);
                    if (lane >= (u32)o)
                        incl += t;
                }
                u32 idx = out + incl - lcnt;
                out += __shfl(incl, 63);
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                while (hits)
                {
                    const u32 k = __builtin_ctz(hits);
                    hits &= hits - 1u;
                    const u32 c = ac_walk<CI, false>(a, lbase + k, 0u, [](u32, u64, u32) {});
                    const u32 base_i = idx;
                    ac_walk<CI, true>(a, lbase + k, c, [&](u32 rank, u64 s, u32 len) {
                        const u32 at = base_i + rank;
                        if (do_stage)
                        {
                            if (at < a.stage_cap)
                                slot[at] = ((s + a.global_base) << 11) | len;
                        }
                        else
                        {
                            const u64 g = fbase + at;
                            if (g < a.pos_cap)
                            {
                                const u64 st = s + a.global_base, en = st + len;
                                *reinterpret_cast<uint4 *>(a.positions + 2 * g) =
                                    make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                            }
                        }
                    });
                    idx += c;
                }
            }
        }
    }
    if (lane == 0 && acc_total && !a.emit_mode)
        atomicAdd(&a.ctr->total, acc_total);
}

// ---------------------------------------------------------------------------------------------- host
struct AcTables
{
    int device = 0;
    u32 npat = 0, lmin = 0, lmax = 0;
    bool ci = false, has_nl = false, has_empty = false;
    u32 has1 = 0, has2 = 0, has3 = 0, has4 = 0, off2 = 0, off3 = 0, off4 = 0, filter_words = 0;
    u32 *d_filter = nullptr;
    uint2 *d_edges = nullptr;
    u32 emask = 0;
    u32 *d_copies = nullptr;
    u32 nnodes = 0;
};

#define ACHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);       \
            goto bad;                                                                          \
        }                                                                                      \
    } while (0)

static inline uint8_t ac_lo8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

AcTables *ac_build(const search_params_t &sp, int device)
{
    auto *t = new AcTables();
    t->device = device;
    t->ci = !sp.case_sensitive;
    t->npat = (u32)sp.num_patterns;
    if (sp.num_patterns > 4095)
    {
        fail("too many patterns (%zu > 4095; the reference CLI accepts 1024)", (size_t)sp.num_patterns);
        delete t;
        return nullptr;
    }
    std::vector<std::vector<uint8_t>> pats;
    for (size_t i = 0; i < sp.num_patterns; ++i)
    {
        std::vector<uint8_t> p((const uint8_t *)sp.patterns[i], (const uint8_t *)sp.patterns[i] + sp.pattern_lens[i]);
        if (t->ci)
            for (auto &c : p)
                c = ac_lo8(c); // the trie is built on folded bytes (aho_corasick.c:161)
        if (p.empty())
        {
            t->has_empty = true; // only ever matches the empty text (aho_corasick.c:441-463)
            continue;
        }
        if (p.size() > 1024)
        {
            fail("pattern %zu longer than 1024 bytes", i);
            delete t;
            return nullptr;
        }
        if (memchr(p.data(), '\n', p.size()))
            t->has_nl = true;
        t->lmax = std::max<u32>(t->lmax, (u32)p.size());
        t->lmin = t->lmin ? std::min<u32>(t->lmin, (u32)p.size()) : (u32)p.size();
        pats.push_back(std::move(p));
    }
    // ---- filter tables ----
    std::vector<u32> T1(kT1Words, 0), T2, T3, T4;
    for (auto &p : pats)
    {
        const size_t n = p.size();
        if (n == 1)
        {
            t->has1 = 1;
            T1[p[0] >> 5] |= 1u << (p[0] & 31);
        }
        else if (n == 2)
        {
            if (!t->has2) T2.assign(kT2Words, 0);
            t->has2 = 1;
            const u32 h = (u32)p[0] | ((u32)p[1] << 8);
            T2[h >> 5] |= 1u << (h & 31);
        }
        else if (n == 3)
        {
            if (!t->has3) T3.assign(kT3Words, 0);
            t->has3 = 1;
            const u32 x = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16);
            const u32 h = (x * kHashMul) >> (32 - kT3Bits);
            T3[h >> 5] |= 1u << (h & 31);
        }
        else
        {
            if (!t->has4) T4.assign(kT4Words, 0);
            t->has4 = 1;
            const u32 x = (u32)p[n - 4] | ((u32)p[n - 3] << 8) | ((u32)p[n - 2] << 16) | ((u32)p[n - 1] << 24);
            const u32 h = (x * kHashMul) >> (32 - kT4Bits);
            T4[h >> 5] |= 1u << (h & 31);
        }
    }
    std::vector<u32> filter(T1);
    t->off2 = (u32)filter.size(); filter.insert(filter.end(), T2.begin(), T2.end());
    t->off3 = (u32)filter.size(); filter.insert(filter.end(), T3.begin(), T3.end());
    t->off4 = (u32)filter.size(); filter.insert(filter.end(), T4.begin(), T4.end());
    t->filter_words = (u32)filter.size();
    // ---- reversed trie ----
    std::unordered_map<u32, u32> edge; // key = node << 8 | byte
    std::vector<u32> copies(1, 0);
    for (auto &p : pats)
    {
        u32 node = 0;
        for (size_t k = p.size(); k-- > 0;)
        {
            const u32 key = (node << 8) | p[k];
            auto it = edge.find(key);
            if (it == edge.end())
            {
                const u32 nn = (u32)copies.size();
                copies.push_back(0);
                edge.emplace(key, nn);
                node = nn;
            }
            else
                node = it->second;
        }
        copies[node]++;
    }
    t->nnodes = (u32)copies.size();
    if (t->nnodes >= (1u << 23))
    {
        fail("pattern set too large (%u trie nodes)", t->nnodes);
        delete t;
        return nullptr;
    }
    u32 cap = 1024;
    while (cap < edge.size() * 2 + 16)
        cap <<= 1;
    t->emask = cap - 1;
    std::vector<uint2> tab(cap, make_uint2(0xffffffffu, 0u));
    for (auto &kv : edge)
    {
        u32 h = (kv.first * kHashMul) >> 7;
        while (tab[h & t->emask].x != 0xffffffffu)
            ++h;
        tab[h & t->emask] = make_uint2(kv.first, kv.second | (copies[kv.second] ? 0x80000000u : 0u));
    }
    if (hipSetDevice(device) != hipSuccess)
        goto bad;
    ACHK(hipMalloc(&t->d_filter, filter.size() * sizeof(u32)));
    ACHK(hipMemcpy(t->d_filter, filter.data(), filter.size() * sizeof(u32), hipMemcpyHostToDevice));
    ACHK(hipMalloc(&t->d_edges, tab.size() * sizeof(uint2)));
    ACHK(hipMemcpy(t->d_edges, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
    ACHK(hipMalloc(&t->d_copies, copies.size() * sizeof(u32)));
    ACHK(hipMemcpy(t->d_copies, copies.data(), copies.size() * sizeof(u32), hipMemcpyHostToDevice));
    return t;
bad:
    ac_free(t);
    return nullptr;
}

void ac_free(AcTables *t)
{
    if (!t)
        return;
    (void)hipSetDevice(t->device);
    if (t->d_filter) (void)hipFree(t->d_filter);
    if (t->d_edges) (void)hipFree(t->d_edges);
    if (t->d_copies) (void)hipFree(t->d_copies);
    delete t;
}

#define SCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int g_ac_force_stage_cap = 0; // test hook (krep_gpu_debug_force_stage_cap)

template <bool CI, bool LN>
static void ac_allow_lds(u32 lds)
{
    // more than 64 KiB of dynamic LDS has to be requested explicitly
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ac_scan_kernel<CI, LN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
}
static hipError_t ac_launch(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    const bool ci = a.flags & F_CI, ln = a.flags & F_LINES;
    if (ci && ln) ac_allow_lds<true, true>(lds);
    else if (ci) ac_allow_lds<true, false>(lds);
    else if (ln) ac_allow_lds<false, true>(lds);
    else ac_allow_lds<false, false>(lds);
    if (ci && ln) hipLaunchKernelGGL((ac_scan_kernel<true, true>), dim3(grid), dim3(kAcBlock), lds, st, a);
    else if (ci) hipLaunchKernelGGL((ac_scan_kernel<true, false>), dim3(grid), dim3(kAcBlock), lds, st, a);
    else if (ln) hipLaunchKernelGGL((ac_scan_kernel<false, true>), dim3(grid), dim3(kAcBlock), lds, st, a);
    else hipLaunchKernelGGL((ac_scan_kernel<false, false>), dim3(grid), dim3(kAcBlock), lds, st, a);
    return hipGetLastError();
}

int ac_scan(AcTables *t, Counters *d_ctr, Counters *h_ctr, PostScratch &post, int num_cu, const uint8_t *d_text, size_t text_len,
            size_t own_lo, size_t own_hi, size_t global_base, match_position_t *d_pos, uint64_t cap, bool ww, bool lines,
            bool track, size_t max_count, hipStream_t st, int time_it, hipEvent_t ev0, hipEvent_t ev1, krep_gpu_scan_out_t *out)
{
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
    a.end_hi = lines ? own_hi : std::min<u64>(text_len, (u64)own_hi + t->lmax - 1);
    a.anchor = own_lo & ~(u64)15;
    const u64 tile_bytes = (u64)kSegBytes * kAcWaves;
    a.num_tiles = (a.end_hi - a.anchor + tile_bytes - 1) / tile_bytes;
    a.flags = (t->ci ? F_CI : 0) | (ww ? F_WW : 0) | (lines ? F_LINES : 0);
    a.lmax = t->lmax;
    a.has1 = t->has1; a.has2 = t->has2; a.has3 = t->has3; a.has4 = t->has4;
    a.filter = t->d_filter;
    a.off2 = t->off2; a.off3 = t->off3; a.off4 = t->off4;
    a.filter_words = t->filter_words;
    a.edges = t->d_edges;
    a.emask = t->emask;
    a.copies = t->d_copies;
    a.ctr = d_ctr;
    const u64 want = (d_pos && cap && !lines) ? std::min<u64>(cap, (u64)max_count) : 0;
    if (want)
        a.flags |= F_POS;
    a.positions = (u64 *)d_pos;
    a.pos_cap = want;
    const bool chain = want || lines;
    const u64 n_units = a.num_tiles * kAcWaves;
    a.stage_cap = want ? (g_ac_force_stage_cap ? (u32)g_ac_force_stage_cap : 64u) : 0u;
    if (chain)
    {
        if (post_reserve(post, n_units, n_units * a.stage_cap))
            return 2;
        a.unitinfo = post.d_unitinfo;
        a.stage = (u64 *)post.d_stage;
        a.offsets = (const u64 *)post.d_offsets;
    }
    const u32 lds = (((t->filter_words + 3u) & ~3u) + 4u) * sizeof(u32);
    const u32 per_cu = lds <= 80 * 1024 ? 2u : 1u;
    const u32 grid = (u32)std::min<u64>(a.num_tiles, (u64)num_cu * per_cu);
    SCHK(hipSetDevice(t->device));
    if (time_it) SCHK(hipEventRecord(ev0, st));
    SCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), st));
    SCHK(ac_launch(a, grid, lds, st));
    if (chain && post_order(post, n_units, a.stage_cap, 0, lines, (uint64_t *)d_pos, want, d_ctr, num_cu, st))
        return 2;
    if (time_it) SCHK(hipEventRecord(ev1, st));
    SCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    if (want && h_ctr->overflow_units)
    {
        AcArgs e = a;
        e.emit_mode = 1;
        SCHK(hipMemsetAsync(&d_ctr->ticket, 0, sizeof(unsigned long long), st));
        SCHK(ac_launch(e, grid, lds, st));
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
    out->count = std::min<u64>(lines ? nl : total, (u64)max_count);
    if (want && track)
    {
        out->stored = std::min<u64>(out->count, want);
        out->overflow = out->count > cap;
    }
    return 0;
}

} // namespace kg
