// kg_ac_split.hip — measured ALTERNATIVES of the multi-pattern scan, kept for reproducibility behind
// KREP_GPU_AC_SPLIT=1 (DESIGN.md §4.2 lists their numbers; all pass the same parity tests as the shipped path):
//   filter kernel -> per-unit candidate lists in HBM -> verify kernel (trie walk, or per-length hash probes).
// They were expected to hide the verifier's latency behind occupancy; on MI355X they are slower than the fused
// kernel of kg_ac.hip (1.0 / 0.72 TB/s against 1.8 TB/s).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kg_ac_common.h"
#include "kg_internal.h"

namespace kg {

static const int g_ac_force_walk = getenv("KREP_GPU_AC_FORCE_WALK") ? 1 : 0; // test hook: trie-walk verifier

// ================================================================================================
// Split pipeline (positions / counts): FILTER kernel -> candidate lists -> VERIFY kernel.
// The trie walk is a chain of dependent L2 accesses (2-3 us each under a saturated HBM stream); inside
// the streaming kernel (128 VGPRs, 16 waves/CU) it cost 17 us per 8 KiB unit and capped the scan at
// ~1.2 TB/s.  Split, the filter keeps streaming and the verifier runs as a small-footprint kernel whose
// latency is hidden by occupancy.
// ================================================================================================

template <bool CI, int CLS>
__global__ __launch_bounds__(kAcBlock) void ac_filter_kernel(const AcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) u32 s_mem[]; // filter tables
    const u32 lane = ac_lane();
    for (u32 w = threadIdx.x; w < a.filter_words; w += kAcBlock)
        s_mem[w] = a.filter[w];
    __syncthreads();
    for (;;)
    {
        u64 tk = 0;
        if (lane == 0)
            tk = __hip_atomic_fetch_add(&a.ctr->ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = ac_rfl64(tk);
        const u64 u_begin = tk * (u64)kAcUnitsPerTicket;
        if (u_begin >= a.num_tiles)
            break;
        const u64 u_end = (u_begin + kAcUnitsPerTicket < a.num_tiles) ? u_begin + kAcUnitsPerTicket : a.num_tiles;
        for (u64 unit = u_begin; unit < u_end; ++unit)
        {
            const u64 seg = a.anchor + unit * (u64)kSegBytes;
            const bool fast = seg + kSegBytes <= a.text_len;
            const bool interior = seg >= a.end_lo && seg + kSegBytes <= a.end_hi;
            uint4 d[kCells];
            u32 before = 0;
            if (fast)
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(a.text + seg) + lane;
#pragma unroll
                for (int j = 0; j < kCells; ++j)
                    d[j] = src[j * kWave];
            }
            if (seg >= 4 && seg <= a.text_len)
                before = *reinterpret_cast<const u32 *>(a.text + seg - 4);
            else
                for (u32 b = 0; b < 4; ++b)
                    if (seg + b >= 4 && seg + b - 4 < a.text_len)
                        before |= (u32)a.text[seg + b - 4] << (8 * b);
            u32 *out = a.cand + (a.unit_base + unit) * (u64)a.cand_cap;
            u32 qn = 0;
            bool flooded = false;
#pragma unroll
            for (int j = 0; j < kCells; ++j)
            {
                const u64 lbase = seg + (u64)j * kCellBytes + (u64)lane * 16u;
                u32 W[5];
                if (fast)
                {
                    W[1] = d[j].x; W[2] = d[j].y; W[3] = d[j].z; W[4] = d[j].w;
                    const u32 up = __shfl_up(W[4], 1);
                    const u32 edge = (j == 0) ? before : __builtin_amdgcn_readlane(d[j > 0 ? j - 1 : 0].w, 63);
                    W[0] = (lane == 0u) ? edge : up;
                }
                else
                {
#pragma unroll 1
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
                if (CI)
                {
#pragma unroll
                    for (int w = 0; w < 5; ++w)
                        W[w] = ac_fold4(W[w]);
                }
                u32 cand = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                {
                    const int o = k + 1;
                    const u32 E = ((o & 3) == 0) ? W[o >> 2] : __builtin_amdgcn_alignbyte(W[(o >> 2) + 1], W[o >> 2], (u32)(o & 3));
                    u32 hit = 0;
                    if (CLS & 8)
                        {
                        // the >= 4-byte table sits at LDS byte 0: byte-addressed, no base add
                        const u32 t = E * kHashMul;
                        const u32 by = reinterpret_cast<const unsigned char *>(s_mem)[t >> (32 - kT4Bits + 3)];
                        hit |= (by >> ((t >> (32 - kT4Bits)) & 7u)) & 1u;
                    }
                    if (CLS & 4)
                        hit |= ac_tbit(s_mem, a.off3, ((E >> 8) * kHashMul) >> (32 - kT3Bits));
                    if (CLS & 2)
                        hit |= ac_tbit(s_mem, a.off2, E >> 16);
                    if (CLS & 1)
                        hit |= ac_tbit(s_mem, a.off1, E >> 24);
                    cand |= hit << k;
                }
                if (!interior)
                {
                    const u64 lo = a.end_lo, hi = a.end_hi;
                    const u32 klo = lo > lbase ? (u32)((lo - lbase) < 16 ? (lo - lbase) : 16) : 0u;
                    const u32 khi = hi > lbase ? (u32)((hi - lbase) < 16 ? (hi - lbase) : 16) : 0u;
                    cand &= khi > klo ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                }
                if (a.flags & (1u << 31)) // ablation hook (KREP_GPU_AC_NOVERIFY)
                    cand = 0;
                if (!flooded && __ballot(cand != 0u))
                {
                    const u32 c = __popc(cand);
                    u32 tot = 0, ex = 0;
#pragma unroll
                    for (int b = 0; b < 5; ++b)
                    {
                        const u64 m = __ballot((c >> b) & 1u);
                        tot += (u32)__popcll(m) << b;
                        ex += (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)) << b;
                    }
                    if (qn + tot > a.cand_cap)
                        flooded = true;
                    else
                    {
                        u32 at = qn + ex, rest = cand;
                        const u32 rel0 = (u32)j * kCellBytes + lane * 16u;
                        while (rest)
                        {
                            const u32 k = __builtin_ctz(rest);
                            rest &= rest - 1u;
                            out[at++] = rel0 + k;
                        }
                        qn += tot;
                    }
                }
            }
            if (lane == 0)
                a.candcnt[a.unit_base + unit] = flooded ? kAcFlooded : qn;
        }
    }
}

// one wave per unit: verify its candidates (or, for a flooded unit, every end position), rank, stage / emit
template <bool CI, bool JUMP>
__global__ __launch_bounds__(256) void ac_verify_kernel(const AcArgs a)
{
    const u32 lane = ac_lane();
    const u64 n_waves = (u64)gridDim.x * 4, wid = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool want_pos = (a.flags & F_POS) != 0;
    const bool emit_final = a.emit_mode != 0;
    u64 acc_total = 0;
    for (u64 lunit = wid; lunit < a.num_tiles; lunit += n_waves)
    {
        const u64 unit = a.unit_base + lunit; // global unit index (arrays); lunit addresses the chunk's text
        if (emit_final && (u32)(a.unitinfo[unit] & kUiCountMask) <= a.stage_cap)
            continue;
        const u32 cc = a.candcnt[unit];
        const bool flooded = cc == kAcFlooded;
        const u32 n = flooded ? kSegBytes : cc;
        const u64 seg = a.anchor + lunit * (u64)kSegBytes;
        const u32 *cl = a.cand + unit * (u64)a.cand_cap;
        u64 *slot = reinterpret_cast<u64 *>(a.stage) + unit * (u64)a.stage_cap;
        const bool do_final = emit_final && want_pos, do_stage = !emit_final && want_pos;
        const u64 fbase = do_final ? a.offsets[unit] : 0ull;
        u32 wcnt = 0;
        for (u32 b0 = 0; b0 < n; b0 += 64)
        {
            const u32 qi = b0 + lane;
            bool live = qi < n;
            const u32 rel = flooded ? qi : (live ? cl[qi] : 0u);
            const u64 pos = seg + rel;
            if (flooded)
                live = pos >= a.end_lo && pos < a.end_hi;
            u32 c = 0;
            if (live)
                c = ac_walk<CI, false, JUMP>(a, pos, 0u, [](u32, u64, u32) {});
            if (!__ballot(c != 0u))
                continue;
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
            if (c && (do_stage || do_final))
                ac_walk<CI, true, JUMP>(a, pos, c, [&](u32 r, u64 s, u32 len) {
                    const u32 at = rank0 + r;
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
        }
        acc_total += wcnt;
        if (want_pos && !emit_final && lane == 0)
        {
            a.unitinfo[unit] = (u64)wcnt | (wcnt ? (kLnHead | kLnTail) : 0ull);
            if (wcnt > a.stage_cap)
            {
                atomicAdd(&a.ctr->overflow_units, 1ull);
                atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
            }
        }
    }
    if (lane == 0 && acc_total && !emit_final)
        atomicAdd(&a.ctr->total, acc_total);
}

// ---- verify, fast form: every pattern is <= 16 bytes -------------------------------------------------
// A trie walk costs two dependent memory accesses per matched byte and the reference order needs the total
// before the first record (two walks): ~40 dependent accesses for a 10-byte match.  Here the 16 bytes ending at
// the candidate are loaded once and each PRESENT pattern length L is resolved by ONE probe of a hash table of
// whole patterns keyed by (L, suffix hash) — all probes independent, exact byte compare, results kept in
// registers per length, emitted longest first.  Latency per candidate ~ 2 memory round trips, any match length.
__device__ __forceinline__ u32 sfx_hash_step(u32 h, u32 byte) { return (h ^ byte) * 0x01000193u; } // FNV-1a over bytes i, i-1, ...
__device__ __forceinline__ u32 sfx_slot(u32 h, u32 L) { return ((h ^ (L * 0x9E3779B1u)) * 0x85EBCA6Bu) >> 8; }

template <bool CI>
__global__ __launch_bounds__(256) void ac_verify16_kernel(const AcArgs a)
{
    const u32 lane = ac_lane();
    const u64 n_waves = (u64)gridDim.x * 4, wid = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool want_pos = (a.flags & F_POS) != 0, ww = (a.flags & F_WW) != 0;
    const bool emit_final = a.emit_mode != 0;
    u64 acc_total = 0;
    for (u64 lunit = wid; lunit < a.num_tiles; lunit += n_waves)
    {
        const u64 unit = a.unit_base + lunit;
        if (emit_final && (u32)(a.unitinfo[unit] & kUiCountMask) <= a.stage_cap)
            continue;
        const u32 cc = a.candcnt[unit];
        const bool flooded = cc == kAcFlooded;
        const u32 n = flooded ? kSegBytes : cc;
        const u64 seg = a.anchor + lunit * (u64)kSegBytes;
        const u32 *cl = a.cand + unit * (u64)a.cand_cap;
        u64 *slot = reinterpret_cast<u64 *>(a.stage) + unit * (u64)a.stage_cap;
        const bool do_final = emit_final && want_pos, do_stage = !emit_final && want_pos;
        const u64 fbase = do_final ? a.offsets[unit] : 0ull;
        u32 wcnt = 0;
        for (u32 b0 = 0; b0 < n; b0 += 64)
        {
            const u32 qi = b0 + lane;
            bool live = qi < n;
            const u32 rel = flooded ? qi : (live ? cl[qi] : 0u);
            const u64 i = seg + rel; // END index
            if (flooded)
                live = i >= a.end_lo && i < a.end_hi;
            // the 16 bytes ending at i: T[w] = bytes [i-15+4w, i-12+4w]
            u32 T[4] = {0u, 0u, 0u, 0u};
            if (live)
            {
                if (i >= 15)
                {
                    struct __attribute__((packed)) U32p { u32 v; };
                    const U32p *q = reinterpret_cast<const U32p *>(a.text + (i - 15));
                    T[0] = q[0].v; T[1] = q[1].v; T[2] = q[2].v; T[3] = q[3].v;
                }
                else
                    for (u32 b = 0; b < 16; ++b)
                        if (i + b >= 15)
                            T[b >> 2] |= (u32)a.text[i + b - 15] << (8 * (b & 3));
                if (CI)
                {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        T[w] = ac_fold4(T[w]);
                }
            }
            // gate: when every pattern has >= 4 bytes, a candidate whose exact last 4 bytes are no pattern suffix
            // is a hash false positive of the LDS filter — one probe, then done
            if (live && a.gram4 && a.lenmask >= (1u << 4) && !(a.lenmask & 0xeu))
            {
                bool any4 = false;
                if (i >= 3)
                    for (u32 hh = (T[3] * kHashMul) >> 9;; ++hh)
                    {
                        const uint2 e = a.gram4[hh & a.g4mask];
                        if (e.y == 0u)
                            break;
                        if (e.x == T[3])
                        {
                            any4 = true;
                            break;
                        }
                    }
                live = any4;
            }
            // pass 1: suffix hashes of every length; ONE 8-byte tag load per PRESENT length, all in flight together
            u64 tg[17];
            u32 hs[17];
            {
                u32 h = 0x811C9DC5u;
#pragma unroll
                for (int L = 1; L <= 16; ++L)
                {
                    const int bi = 16 - L; // index of byte i-L+1 inside T
                    h = sfx_hash_step(h, (T[bi >> 2] >> (8 * (bi & 3))) & 0xffu);
                    hs[L] = h;
                    tg[L] = 0;
                    if ((a.lenmask >> L) & 1u) // uniform
                        if (live)
                            tg[L] = a.tags[sfx_slot(h, (u32)L) & a.sfxmask];
                }
            }
            // pass 2: which lengths need a look?  tag hit -> exact compare; occupied slot with another key -> probe on
            u32 look = 0;
#pragma unroll
            for (int L = 1; L <= 16; ++L)
                if ((a.lenmask >> L) & 1u)
                    look |= (tg[L] != 0ull && (u64)L <= i + 1) ? (1u << L) : 0u;
            // rare part, longest first: resolve the looked-at lengths exactly (linear probing + 16-byte compare)
            u32 c = 0, okmask = 0;
            u64 cps = 0; // 4 bits of copies per validated length would not fit: copies are re-read at emission
            for (u32 rest = look; rest;)
            {
                const u32 L = 31u - (u32)__builtin_clz(rest);
                rest &= ~(1u << L);
                const u32 bi = 16u - L;
                u32 V[4];
#pragma unroll
                for (int w = 0; w < 4; ++w)
                {
                    const int lo = (int)bi - 4 * w;
                    V[w] = lo <= 0 ? T[w] : lo >= 4 ? 0u : (T[w] & (0xffffffffu << (8 * lo)));
                }
                // recompute the hash of this length (hs[] is indexed statically only)
                u32 h = 0x811C9DC5u;
                for (u32 k = 0; k < L; ++k)
                    h = sfx_hash_step(h, (T[(15 - k) >> 2] >> (8 * ((15 - k) & 3))) & 0xffu);
                u32 copies = 0;
                for (u32 sl = sfx_slot(h, L);; ++sl)
                {
                    const u64 tv = a.tags[sl & a.sfxmask];
                    if (tv == 0ull)
                        break;
                    if ((u32)(tv >> 32) == h && (u32)(tv & 0xffu) == L)
                    {
                        const uint4 by = a.sfx[2 * (sl & a.sfxmask)];
                        if (by.x == V[0] && by.y == V[1] && by.z == V[2] && by.w == V[3])
                        {
                            copies = (u32)(tv >> 8) & 0xffffffu;
                            break;
                        }
                    }
                }
                if (copies)
                {
                    const u64 st = i + 1 - (u64)L;
                    bool ok = st >= a.own_lo && st < a.own_hi;
                    if (ok && ww)
                    {
                        if (st > 0 && ac_wordc(a.text[st - 1]))
                            ok = false;
                        else if (i + 1 < a.text_len && ac_wordc(a.text[i + 1]))
                            ok = false;
                    }
                    if (ok)
                    {
                        c += copies;
                        okmask |= 1u << L;
                    }
                }
            }
            (void)cps;
            (void)hs;
            if (!__ballot(c != 0u))
                continue;
            u32 incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const u32 t = __shfl_up(incl, o);
                if (lane >= (u32)o)
                    incl += t;
            }
            u32 at = wcnt + incl - c;
            wcnt += __shfl(incl, 63);
            if (c && (do_stage || do_final))
            {
                for (u32 rest = okmask; rest;) // longest first (aho_corasick.c:353-431)
                {
                    const u32 L = 31u - (u32)__builtin_clz(rest);
                    rest &= ~(1u << L);
                    // copies of this (validated) pattern: find its slot again
                    u32 h = 0x811C9DC5u;
                    for (u32 k = 0; k < L; ++k)
                        h = sfx_hash_step(h, (T[(15 - k) >> 2] >> (8 * ((15 - k) & 3))) & 0xffu);
                    const u32 bi = 16u - L;
                    u32 V[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                    {
                        const int lo = (int)bi - 4 * w;
                        V[w] = lo <= 0 ? T[w] : lo >= 4 ? 0u : (T[w] & (0xffffffffu << (8 * lo)));
                    }
                    u32 copies = 0;
                    for (u32 sl = sfx_slot(h, L);; ++sl)
                    {
                        const u64 tv = a.tags[sl & a.sfxmask];
                        if (tv == 0ull)
                            break;
                        if ((u32)(tv >> 32) == h && (u32)(tv & 0xffu) == L)
                        {
                            const uint4 by = a.sfx[2 * (sl & a.sfxmask)];
                            if (by.x == V[0] && by.y == V[1] && by.z == V[2] && by.w == V[3])
                            {
                                copies = (u32)(tv >> 8) & 0xffffffu;
                                break;
                            }
                        }
                    }
                    for (u32 q = 0; q < copies; ++q, ++at)
                    {
                        const u64 st = i + 1 - (u64)L + a.global_base;
                        if (do_stage)
                        {
                            if (at < a.stage_cap)
                                slot[at] = (st << 11) | L;
                        }
                        else if (fbase + at < a.pos_cap)
                        {
                            const u64 en = st + (u64)L;
                            *reinterpret_cast<uint4 *>(a.positions + 2 * (fbase + at)) =
                                make_uint4((u32)st, (u32)(st >> 32), (u32)en, (u32)(en >> 32));
                        }
                    }
                }
            }
        }
        acc_total += wcnt;
        if (want_pos && !emit_final && lane == 0)
        {
            a.unitinfo[unit] = (u64)wcnt | (wcnt ? (kLnHead | kLnTail) : 0ull);
            if (wcnt > a.stage_cap)
            {
                atomicAdd(&a.ctr->overflow_units, 1ull);
                atomicMax(&a.ctr->max_unit_count, (u64)wcnt);
            }
        }
    }
    if (lane == 0 && acc_total && !emit_final)
        atomicAdd(&a.ctr->total, acc_total);
}


template <bool CI, int CLS>
static hipError_t ac_filter_launch2(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ac_filter_kernel<CI, CLS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)
    {
        fail("hipFuncSetAttribute(ac_filter_kernel, %u B LDS) failed: %s", lds, hipGetErrorString(e));
        return e;
    }
    hipLaunchKernelGGL((ac_filter_kernel<CI, CLS>), dim3(grid), dim3(kAcBlock), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess)
        fail("ac_filter_kernel<%d,%d> launch failed: %s (grid %u, lds %u)", (int)CI, CLS, hipGetErrorString(e), grid, lds);
    return e;
}
hipError_t ac_filter_launch(const AcArgs &a, u32 grid, u32 lds, hipStream_t st)
{
    const bool only4 = a.has4 && !a.has1 && !a.has2 && !a.has3;
    if (a.flags & F_CI)
        return only4 ? ac_filter_launch2<true, 8>(a, grid, lds, st) : ac_filter_launch2<true, 15>(a, grid, lds, st);
    return only4 ? ac_filter_launch2<false, 8>(a, grid, lds, st) : ac_filter_launch2<false, 15>(a, grid, lds, st);
}
hipError_t ac_verify_launch(const AcArgs &a, u32 grid, hipStream_t st)
{
    const bool only4 = a.has4 && !a.has1 && !a.has2 && !a.has3, ci = a.flags & F_CI;
    if (a.sfx && !g_ac_force_walk)
    { // every pattern <= 16 bytes: independent per-length probes instead of the trie walk
        if (ci) hipLaunchKernelGGL((ac_verify16_kernel<true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((ac_verify16_kernel<false>), dim3(grid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (ci && only4) hipLaunchKernelGGL((ac_verify_kernel<true, true>), dim3(grid), dim3(256), 0, st, a);
    else if (ci) hipLaunchKernelGGL((ac_verify_kernel<true, false>), dim3(grid), dim3(256), 0, st, a);
    else if (only4) hipLaunchKernelGGL((ac_verify_kernel<false, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((ac_verify_kernel<false, false>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}


} // namespace kg
