// kg_scan.hip — krep_gpu_scan_device[_ex|_seq](): the device-resident scan with every reference return-value convention
// (verdict_for), the single-literal pass and its post-passes (lit_pass), the match-set families (greedy walks, end-of-text
// replay of the block-structured -c bodies, kg_greedy.hip / kg_tail.hip), how a text may be split (kg::split_mode, the boundary
// record kg::fold_carry) and the multi-pattern drivers around kg_ac.hip.  Host logic only; kernels live in kg_literal.hip /
// kg_single.hip / kg_ac.hip / kg_post.hip / kg_greedy.hip / kg_tail.hip / kg_format.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_internal.h"
#include "kg_plan.h"
#include "kg_replay.h"
#include "kg_ac_tables.h"

using namespace kg;

#define HIPCHK(x)                                                                             \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------ reference return-value conventions
// What the reference function returns / stores given the number of emitted matches (`total`, after
// greedy selection and -w) or distinct lines.  One place for all the max_count corner cases.
struct Verdict { uint64_t ret, store; };
static Verdict verdict_for(int algo, const krep_gpu_plan *pl, uint64_t total, uint64_t lines, bool have_result)
{
    const size_t maxc = pl->max_count;
    const bool store = pl->track && have_result;
    Verdict v{0, 0};
    switch (algo)
    {
    case KREP_RA_MEMCHR: // krep.c:3897, :3955, :3976
    case KREP_RA_KMP:    // krep.c:1634, :1696, :1717
    case KREP_RA_AHO_CORASICK: // aho_corasick.c:316
    case KREP_RA_SSE42: // :4713 then the pre-increment checks :4778/:4804
        if (maxc == 0)
            return v;
        break;
    case KREP_RA_NEON: // :4516 then the pre-increment checks :4582/:4616; count-only with max_count == 0 is resolved by
                       // the caller (the tail call's BMH convention, scan_literal)
        if (maxc == 0)
            return v;
        break;
    default: // BMH :1266, memchr_short :4376, AVX2 :4887, AVX-512 :5119
        if (maxc == 0)
        {
            if (pl->lines || pl->track)
                return v;
            v.ret = total > 0 ? 1 : 0; // count-only: the first hit makes 1 >= 0 true (krep.c:1355-1367)
            return v;
        }
    }
    if (pl->lines)
    {
        v.ret = std::min<uint64_t>(lines, maxc);
        return v;
    }
    v.ret = std::min<uint64_t>(total, maxc);
    if (store)
    {
        v.store = v.ret;
        if (algo == KREP_RA_KMP && maxc != SIZE_MAX && total > maxc)
            v.store = v.ret + 1; // krep.c:1717-1724 stores the (max_count+1)-th match before breaking
    }
    return v;
}

// ------------------------------------------------------------------------------------ device scan: single literal
namespace {
// the device buffer being scanned: a slice [global_base, global_base + text_len) of a text of global_len bytes
struct Window
{
    const uint8_t *d_text;
    size_t text_len;          // bytes readable in the buffer
    size_t own_lo, own_hi;    // buffer-relative ownership window
    size_t global_base;       // offset of buffer byte 0 in the whole text (added to reported offsets)
    size_t global_len;        // length of the whole text (== global_base + text_len for the buffer that holds its end)
};
// one pass of the literal kernel family + its ordering post-pass
struct LitPass
{
    bool ww = false, lines = false;
    bool first_byte = false;  // candidate pass of memchr_short -o: scan for pattern[0] only (starts clipped to n - m)
    size_t own_lo = 0, own_hi = 0;
    uint64_t excl_lo = 0, excl_hi = 0; // buffer-relative exclusion window (simd_avx512_search's unexamined block)
    uint64_t ww_exempt = ~0ull;        // buffer-relative start whose left-neighbour test is skipped
    enum Sink { COUNT, RECORDS, OCC } sink = COUNT;
    uint64_t *d_out = nullptr;         // RECORDS: final match_position_t buffer
    uint64_t out_cap = 0;
    PostScratch *post = nullptr;
    hipEvent_t ev_end = nullptr;       // recorded right behind the last kernel of the pass (before the counter read-back)
};
struct LitResult
{
    uint64_t total = 0, lines = 0;
    unsigned long long summary = 0;
    uint64_t n_units = 0, unit_bytes = 0, anchor = 0;
};
} // namespace

// hits per 32-KiB unit from which a 2..8-byte literal with records takes the one-pass kernel (KREP_GPU_FUSEDK_MIN: measurement aid)
static double fusedk_min_hits_per_unit()
{
    static const double v = [] {
        const char *e = getenv("KREP_GPU_FUSEDK_MIN");
        return e ? atof(e) : 24.0;
    }();
    return v;
}

static int lit_pass(krep_gpu_plan *pl, const Window &w, const LitPass &ps, hipStream_t st, LitResult *res)
{
    *res = LitResult{};
    const uint32_t m_scan = ps.first_byte ? 1u : pl->m;
    size_t own_hi = std::min(ps.own_hi, w.text_len);
    if (ps.first_byte)
        own_hi = std::min<size_t>(own_hi, w.text_len - pl->m + 1);
    const uint64_t hi_match = std::min<uint64_t>(own_hi, w.text_len - m_scan + 1);
    if (hi_match <= ps.own_lo)
        return 0;
    PostScratch &post = *ps.post;

    LitArgs a{};
    a.text = w.d_text;
    a.text_len = w.text_len;
    a.own_lo = ps.own_lo;
    a.own_hi = own_hi;
    a.anchor = ps.own_lo & ~(uint64_t)15;
    {
        // big tiles (128 KiB) once there are enough of them to fill the chip several times over
        const uint64_t span = hi_match - a.anchor;
        a.rounds = span >= ((uint64_t)pl->num_cu * 16 * kRoundsBig * kSegBytes * kWavesPerBlk) ? kRoundsBig : 1;
        const int fr = g_force_rounds.load(std::memory_order_relaxed);
        if (fr == 1 || fr == kRoundsBig)
            a.rounds = (uint32_t)fr;
        const uint64_t tile_bytes = (uint64_t)a.rounds * kSegBytes * kWavesPerBlk;
        a.num_tiles = (span + tile_bytes - 1) / tile_bytes;
    }
    a.global_base = w.global_base;
    a.ww_exempt_left = ps.ww_exempt;
    a.excl_lo = ps.excl_lo;
    a.excl_hi = ps.excl_hi;
    a.m = m_scan;
    if (ps.first_byte)
    {
        const uint8_t b = pl->pat_folded[0];
        a.p0 = 0x01010101u * b;
        a.k0 = 0xffu;
        a.l0 = (!pl->cs && b >= 'a' && b <= 'z') ? 0x20202020u : 0u;
    }
    else
    {
        a.p0 = pl->p0; a.p1 = pl->p1; a.k0 = pl->k0; a.k1 = pl->k1;
        a.l0 = pl->l0; a.l1 = pl->l1;
        a.p2 = pl->p2; a.p3 = pl->p3; a.k2 = pl->k2; a.k3 = pl->k3; a.l2 = pl->l2; a.l3 = pl->l3;
    }
    a.pat = pl->d_pat;
    a.pat_chunks = pl->d_pat_chunks;
    a.n_chunks = ps.first_byte ? 0 : pl->n_chunks;
    a.ctr = pl->d_ctr;
    a.flags = (pl->cs ? 0 : F_CI) | (ps.ww ? F_WW : 0) | (ps.lines ? F_LINES : 0) | (ps.sink != LitPass::COUNT ? F_POS : 0);
    {
        // kg_literal_dma.hip's prefilter: only for a first byte that text rarely holds (the blank and the dozen most frequent letters
        // of running text, either case, would pass nearly every 16-byte lane: the test would be paid for nothing)
        // (case-sensitive: the lower-case ones only — a capital first letter is rare)
        const uint32_t b0 = (a.p0 & 0xffu) | (pl->cs ? 0u : 0x20u);
        static const char common[] = "etaoinshrdlu";
        bool rare = b0 != ' ' && b0 != '\n';
        for (const char *q = common; *q; ++q)
            rare = rare && b0 != (uint32_t)*q;
        a.prefilter = (rare && !ps.first_byte && !getenv("KREP_GPU_LIT_NO_PREFILTER")) ? 0x01010101u * b0 : 0u;
    }
    const bool chain = (a.flags & (F_POS | F_LINES)) != 0;
    const uint64_t n_units = a.num_tiles * kWavesPerBlk;
    // Dealing: from ~24 GiB on every wave draws tickets of 8 units (>= 24 tickets per resident wave, ~26 fetch-adds/us on the
    // ticket word); below that the static interleaved deal (upt = 0) is faster — measured at 8 GiB: 1.38 vs 1.41 ms with
    // offsets, 1.29 vs 1.37 ms counting; at 16 GiB 2.58 vs 2.63 ms; at 2 GiB 0.37 vs 0.48 ms (the ticket word becomes the limit
    // of a short scan); at 32 GiB tickets win: 5.20 vs 5.31 ms, and 7.1 vs 7.9 ms on the single-byte workload
    // (the LDS-DMA kernel, kg_literal_dma.hip, wants tickets — a ticket is what it streams through without a seam — and is ahead with them
    //  from ~8 GiB on: 8 GiB 1.339 against 1.362 ms for the register kernel's static deal, 16 GiB 2.610 against 2.639; round 6)
    bool dma_shape = m_scan >= 2 && m_scan <= 8 && !ps.lines && !ps.first_byte && a.prefilter != 0u && !getenv("KREP_GPU_LIT_NO_DMA");
    auto tickets_for = [&](bool dma) { return (a.rounds == kRoundsBig && n_units / ((uint64_t)pl->num_cu * 16) >= (dma ? 64u : 192u)) ? 8u : 0u; };
    a.upt = tickets_for(dma_shape);
    // ... and wants a TEXT in which the prefilter's byte is rare: a 1-KiB cell that holds it pays the full compare, which two workgroups per CU
    // do not hide.  32 GiB, offsets produced, by the share of cells that hold the byte (profiles/r06_ldsdma_byte_rate.txt): 10 % 5.08 ms,
    // 27 % 5.08, 46 % 5.23, 68 % 5.75, 96 % 6.49 — against 5.19-5.20 ms of the register kernel throughout.  The static rule above only bars
    // the letters that running text is made of; what THIS text holds is sampled once (two 2-MiB windows, one wave per cell) before the plan's
    // first such launch, and counted by every launch of the kernel (Counters::candidates).  A plan barred by one text looks again at the next.
    static const double kDmaMaxPass = [] { const char *e = getenv("KREP_GPU_LIT_DMA_MAX_PASS"); return e ? atof(e) : 0.35; }();
    const bool dma_keep = getenv("KREP_GPU_LIT_DMA_KEEP") != nullptr; // measurement aid: no look, no bar
    if (dma_shape && (a.upt >= 4 || getenv("KREP_GPU_LIT_DMA_ALL")) && a.rounds == kRoundsBig && !dma_keep && hi_match - a.anchor >= (64ull << 20) &&
        (!pl->dma_look_done || (pl->dma_off && (pl->dma_off_text != (const void *)w.d_text || pl->dma_off_len != w.text_len))))
    {
        pl->dma_look_done = true;
        const uint64_t span = hi_match - a.anchor, win = 2ull << 20;
        HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
        for (int q = 0; q < 2; ++q)
            HIPCHK(launch_dma_byte_look(w.d_text, (a.anchor + span / 4 * (2 * q + 1)) & ~(uint64_t)15, (uint32_t)(win / kCellBytes), a.prefilter,
                                        !pl->cs, &pl->d_ctr->candidates, st));
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        pl->dma_pass_rate = (double)pl->h_ctr->candidates / (double)(2 * win / kCellBytes);
        pl->dma_off = pl->dma_pass_rate > kDmaMaxPass;
        pl->dma_off_text = w.d_text;
        pl->dma_off_len = w.text_len;
        if (getenv("KREP_GPU_DEBUG"))
            fprintf(stderr, "krep-gpu: literal scan: the prefilter's byte 0x%02x occurs in %.1f %% of the sampled 1-KiB cells -> %s kernel\n", a.prefilter & 0xffu,
                    100.0 * pl->dma_pass_rate, pl->dma_off ? "register" : "LDS-DMA");
    }
    if (pl->dma_off && !dma_keep)
    {
        a.prefilter = 0u; // (kg_literal_dma.hip literal_dma_eligible: the register kernel)
        dma_shape = false;
        a.upt = tickets_for(false);
    }
    if (const char *e = getenv("KREP_GPU_LIT_UPT")) // measurement aid; only the values the kernel's parked-store bookkeeping
    {                                               // is built for (kPark = 240 must be a multiple)
        const int v = atoi(e);
        if (v == 0 || v == 1 || v == 2 || v == 4 || v == 8)
            a.upt = (uint32_t)v;
    }
    // Staging slot per unit.  Single byte: 512 offsets (1 KiB) per 32 KiB unit, ~1.5x BASELINE's 1 % density.  Sparse kinds:
    // 16 offsets = ONE 32-byte slot per 32 KiB unit (BASELINE's 1e-4/B puts 3.3 hits in a unit); units that hold more take the
    // emit-mode re-scan, and a scan in which more than 1 in 64 units did raises the plan's slot to 64 for its next scans.
    // Why so small: every unit's slot is a store into a different place, and the 4096 units in flight walk through the
    // staging array one slot each per step — with 128-byte slots through 128 pages of 4 KiB per step.  On boxes where the
    // driver backs the scratch with small page-table fragments the offsets-producing scan ran 0.3-0.5 ms slower in every
    // second process (store translation misses, coupled to the loads through the shared in-order vmcnt); 32-byte slots
    // cut that to < 0.1 ms (same process, 32 GiB: 5.75/5.48/5.82/5.52 ms with 64 entries, 5.47/5.43/5.52/5.41 with 16).
    a.stage_cap = 0;
    if (a.flags & F_POS)
        a.stage_cap = a.rounds == kRoundsBig ? (m_scan == 1 ? 512u : pl->sparse_cap) : (m_scan == 1 ? 256u : 32u);
    const int fsc = g_force_stage_cap.load(std::memory_order_relaxed);
    if (fsc && (a.flags & F_POS))
        a.stage_cap = (uint32_t)fsc;
    const uint32_t grid = (uint32_t)pl->num_cu; // the launchers size the grid: resident blocks of the chosen variant x CUs
    const uint64_t unit_bytes = (uint64_t)a.rounds * kSegBytes, origin = a.anchor + w.global_base;
    res->n_units = n_units;
    res->unit_bytes = unit_bytes;
    res->anchor = a.anchor;
    // ---- single byte with records (memchr_search, BASELINE config 3): ONE pass, the records written by the scanning waves
    // at their final index (kg_single.hip) — no staging, no info words, no post-pass.  A scan too dense for the rings of its
    // shape is counted but not recorded: the count picks the shape that holds it (up to ~20 % hits) and the scan runs again in
    // that shape — the plan keeps it; beyond that the two-pass kernels below take this scan and the plan's later ones.
    // The same kernel takes a literal of 2..8 bytes (its MULTI instantiations) once a two-pass scan of the plan has counted a
    // density at which the staging slots of the sparse kinds overflow (`-i sh`: 35 hits per unit; ` a`: 180): pl->fusedk_on.
    // ---- the FIRST records scan of a plan on a large text: one cheap look before the full launch (round 6).  A plan
    // used to learn the density of its text from whole scans — the two-pass road first, the one-pass writers and their ring shape
    // from the third or fourth scan on (`-i Sh`: 4222 / 4886 / 5026 / 5097 GB/s in four consecutive scans), which a CLI process,
    // making ONE scan, never reached.  Two 2-MiB windows are counted (~0.1 ms next to >= 0.16 ms of scan) and the road, the ring shape
    // and the staging slot are chosen from that; every later scan re-evaluates them as before.
    if (ps.sink == LitPass::RECORDS && !pl->first_look_done && a.rounds == kRoundsBig && fsc == 0 && !ps.first_byte && !ps.lines &&
        ps.excl_lo == ps.excl_hi && hi_match - a.anchor >= (1ull << 30) && !getenv("KREP_GPU_NO_FIRST_LOOK"))
    {
        pl->first_look_done = true;
        const uint64_t span = hi_match - a.anchor, win = 2ull << 20, tile_bytes = (uint64_t)kRoundsBig * kSegBytes * kWavesPerBlk;
        HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
        uint64_t looked = 0;
        for (int q = 0; q < 2; ++q)
        {
            LitArgs sm = a;
            sm.flags &= ~(uint32_t)(F_POS | F_LINES);
            sm.stage_cap = 0;
            sm.upt = 0;
            sm.emit_mode = 0;
            sm.own_lo = (a.anchor + span / 4 * (2 * q + 1)) & ~(uint64_t)15;
            sm.own_hi = std::min<uint64_t>(hi_match, sm.own_lo + win);
            sm.anchor = sm.own_lo;
            sm.num_tiles = (sm.own_hi - sm.anchor + tile_bytes - 1) / tile_bytes;
            looked += sm.own_hi - sm.own_lo;
            HIPCHK(launch_literal(sm, grid, st));
        }
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const double density = looked ? (double)pl->h_ctr->total / (double)looked : 0.0, per_unit = density * 32768.0;
        if (m_scan == 1)
        {
            if (density > single_fused_max_density(kFusedShapeMax))
                pl->fused1_ok = false;
            else
            {
                int shape = 0;
                while (shape < kFusedShapeMax && 1.05 * density > single_fused_max_density(shape))
                    ++shape;
                pl->fused1_shape = shape;
            }
        }
        else if (m_scan <= 8 && !pl->fusedk_never && per_unit >= fusedk_min_hits_per_unit() &&
                 density <= single_fused_max_density(kFusedShapeMax))
        {
            int shape = 0;
            while (shape < kFusedShapeMax && 1.05 * density > single_fused_max_density(shape))
                ++shape;
            pl->fusedk_on = true;
            pl->fusedk_shape = shape;
        }
        else if (per_unit > 6.0) // the staging road with slots that hold what a unit holds (what a whole scan used to find out)
            pl->sparse_cap = std::max<uint32_t>(pl->sparse_cap, per_unit > 100.0 ? 256u : per_unit > 40.0 ? 128u : 64u);
        if (getenv("KREP_GPU_DEBUG"))
            fprintf(stderr, "krep-gpu: first look: %.2f hits per 32-KiB unit in %llu sampled bytes -> one-pass %d (shape %d / %d), staging slot %u\n", per_unit,
                    (unsigned long long)looked, (int)(m_scan == 1 ? pl->fused1_ok : pl->fusedk_on), pl->fused1_shape, pl->fusedk_shape, pl->sparse_cap);
        if ((a.flags & F_POS) && !fsc)
            a.stage_cap = a.rounds == kRoundsBig ? (m_scan == 1 ? 512u : pl->sparse_cap) : a.stage_cap;
    }
    const bool fusedk = m_scan >= 2 && m_scan <= 8 && pl->fusedk_on;
    // (-w: the 2..8-byte instantiations test it in registers since round 6; the single-byte ones do not build the window it needs)
    if ((m_scan == 1 ? pl->fused1_ok : fusedk) && ps.sink == LitPass::RECORDS && !(ps.ww && m_scan == 1) && !ps.lines && !ps.first_byte &&
        a.rounds == kRoundsBig && fsc == 0 && ps.excl_lo == ps.excl_hi && w.text_len >= 2 * (size_t)kSegBytes &&
        !getenv("KREP_GPU_NO_FUSED1"))
    {
        a.positions = ps.d_out;
        a.pos_cap = ps.out_cap;
        int &plan_shape = m_scan == 1 ? pl->fused1_shape : pl->fusedk_shape;
        for (;;)
        {
            const int shape = plan_shape;
            const uint64_t n_tk = single_fused_tickets(n_units, shape);
            if (n_tk > post.tk_cap)
            {
                if (post.d_tk) (void)hipFree(post.d_tk);
                post.d_tk = nullptr;
                post.tk_cap = 0;
                HIPCHK(hipMalloc(&post.d_tk, single_fused_scratch_words(n_tk) * sizeof(unsigned long long)));
                post.tk_cap = n_tk;
            }
            HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
            HIPCHK(hipMemsetAsync(post.d_tk, 0, single_fused_scratch_words(n_tk) * sizeof(unsigned long long), st));
            HIPCHK(launch_single_fused(a, post.d_tk, post.d_tk + n_tk, n_tk, grid, shape, st));
            g_fused1_launches.fetch_add(1);
            if (ps.ev_end) HIPCHK(hipEventRecord(ps.ev_end, st));
            HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (!pl->h_ctr->overflow_units)
            {
                res->total = pl->h_ctr->total;
                res->summary = res->total ? (kLnHead | kLnTail) : 0;
                // the decision is re-evaluated by every scan: a text half as dense as the next smaller shape holds
                // sends the plan's following scans back to it (larger tickets, more waves per CU)
                const double density = (double)res->total / (double)(hi_match - a.anchor);
                if (shape > 0 && density < 0.5 * single_fused_max_density(shape - 1))
                    plan_shape = shape - 1;
                if (m_scan != 1 && density * 32768.0 < 0.5 * fusedk_min_hits_per_unit())
                    pl->fusedk_on = false; // (a sparser text: the staging road is faster there)
                return 0;
            }
            // the scan counted every ticket (the resolver's running sum): the density chooses the shape — unless the spin-limit
            // safety net fired (then the total is not to be trusted and the two-pass kernels take over)
            const double density = (double)pl->h_ctr->total / (double)(hi_match - a.anchor);
            int next = shape + 1;
            while (next <= kFusedShapeMax && density > single_fused_max_density(next))
                ++next;
            if (next > kFusedShapeMax || pl->h_ctr->total == 0)
                break;
            plan_shape = next;
        }
        if (m_scan == 1)
            pl->fused1_ok = false;
        else
            pl->fusedk_on = false, pl->fusedk_never = true;
        g_fused1_failovers.fetch_add(1);
    }
    if (chain)
    {
        if (post_reserve(post, n_units, (n_units * a.stage_cap + 3) / 4)) // 16-bit staging entries
            return 2;
        a.unitinfo = post.d_unitinfo;
        a.stage = (uint64_t *)post.d_stage;
        a.offsets = (const uint64_t *)post.d_offsets;
    }
    HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
    if (ps.sink != LitPass::OCC)
    {
        a.positions = ps.d_out;
        a.pos_cap = ps.sink == LitPass::RECORDS ? ps.out_cap : 0;
        HIPCHK(launch_literal(a, grid, st));
        if (chain && post_order(post, n_units, a.stage_cap, m_scan, origin, unit_bytes, ps.lines, ps.d_out, a.pos_cap, pl->d_ctr,
                                pl->num_cu, st))
            return 2;
        if (ps.ev_end) HIPCHK(hipEventRecord(ps.ev_end, st));
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (literal_dma_eligible(a) && hi_match - a.anchor >= (64ull << 20))
        {
            // what the LDS-DMA kernel itself counted on this text: its later scans go by it
            pl->dma_pass_rate = (double)pl->h_ctr->candidates / ((double)(hi_match - a.anchor) / kCellBytes);
            if (pl->dma_pass_rate > kDmaMaxPass && !dma_keep)
            {
                pl->dma_off = true;
                pl->dma_off_text = w.d_text;
                pl->dma_off_len = w.text_len;
                if (getenv("KREP_GPU_DEBUG"))
                    fprintf(stderr, "krep-gpu: literal scan: the prefilter passed %.1f %% of the 1-KiB cells -> the register kernel from the next scan on\n", 100.0 * pl->dma_pass_rate);
            }
        }
        if ((a.flags & F_POS) && m_scan != 1 && a.rounds == kRoundsBig && pl->h_ctr->overflow_units * 64 > n_units && !fsc)
        {
            // a dense input: the next scans of this plan stage 64 hits per unit — and if more than 1 unit in 64 overflows that as
            // well (`-i sh`: 36 hits per unit on average, clustered), as many as the fullest unit held, up to 256: re-scanning
            // 5 % of the units cold cost m = 2 -i a fifth of its time
            uint32_t want = 64;
            if (pl->sparse_cap >= 64)
                for (want = 128; want < 256 && want < pl->h_ctr->max_unit_count; want *= 2) {}
            pl->sparse_cap = std::max(pl->sparse_cap, want);
        }
        if ((a.flags & F_POS) && m_scan >= 2 && m_scan <= 8 && a.rounds == kRoundsBig && !fsc && !pl->fusedk_never &&
            ps.sink == LitPass::RECORDS && !ps.lines && hi_match > a.anchor)
        {
            // dense enough for the one-pass record writer (kg_single.hip, MULTI)?  Its shape from the density just counted.
            const double density = (double)pl->h_ctr->total / (double)(hi_match - a.anchor);
            if (density * 32768.0 >= fusedk_min_hits_per_unit() && density <= single_fused_max_density(kFusedShapeMax))
            {
                int shape = 0;
                while (shape < kFusedShapeMax && density > single_fused_max_density(shape))
                    ++shape;
                pl->fusedk_on = true;
                pl->fusedk_shape = shape;
            }
        }
        if (getenv("KREP_GPU_DEBUG") && (a.flags & F_POS))
            fprintf(stderr, "krep-gpu: literal scan: %llu of %llu units overflowed a %u-entry slot (fullest: %llu); one-pass next: %d (shape %d)\n",
                    pl->h_ctr->overflow_units, (unsigned long long)n_units, a.stage_cap, pl->h_ctr->max_unit_count, (int)pl->fusedk_on,
                    pl->fusedk_shape);
        if ((a.flags & F_POS) && pl->h_ctr->overflow_units)
        {
            // some units held more hits than their staging slot: re-scan exactly those, writing in place
            LitArgs e = a;
            e.emit_mode = 1;
            HIPCHK(hipMemsetAsync(&pl->d_ctr->ticket, 0, sizeof(unsigned long long), st));
            HIPCHK(launch_literal(e, grid, st));
            if (ps.ev_end) HIPCHK(hipEventRecord(ps.ev_end, st));
        }
    }
    else
    {
        // all owned hits into post.d_occ (sized after the count is known): input of the sequential-family walks
        HIPCHK(launch_literal(a, grid, st));
        if (post_offsets_pass(post, n_units, false, pl->d_ctr, st))
            return 2;
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const uint64_t n_occ = pl->h_ctr->total;
        if (n_occ > post.occ_cap)
        {
            if (post.d_occ) (void)hipFree(post.d_occ);
            post.d_occ = nullptr;
            post.occ_cap = 0;
            HIPCHK(hipMalloc(&post.d_occ, n_occ * 2 * sizeof(uint64_t)));
            post.occ_cap = n_occ;
        }
        if (n_occ)
        {
            if (post_gather_pass(post, n_units, a.stage_cap, m_scan, origin, unit_bytes, post.d_occ, n_occ, pl->num_cu, st))
                return 2;
            if (pl->h_ctr->overflow_units)
            {
                LitArgs e = a;
                e.emit_mode = 1;
                e.positions = post.d_occ;
                e.pos_cap = n_occ;
                HIPCHK(hipMemsetAsync(&pl->d_ctr->ticket, 0, sizeof(unsigned long long), st));
                HIPCHK(launch_literal(e, grid, st));
            }
        }
    }
    res->total = pl->h_ctr->total;
    res->lines = pl->h_ctr->lines;
    if (!pl->fused1_ok && m_scan == 1 && hi_match > a.anchor &&
        (double)pl->h_ctr->total / (double)(hi_match - a.anchor) < 0.5 * single_fused_max_density(kFusedShapeMax))
        pl->fused1_ok = true; // (a later, sparser text of the same plan takes the one-pass kernel again)
    if (pl->fusedk_never && m_scan >= 2 && m_scan <= 8 && hi_match > a.anchor &&
        (double)pl->h_ctr->total / (double)(hi_match - a.anchor) < 0.5 * single_fused_max_density(kFusedShapeMax))
        pl->fusedk_never = false; // (the same for the 2..8-byte instantiations: one overflowing text no longer bars them for good)
    res->summary = chain ? pl->h_ctr->summary : (res->total ? (kLnHead | kLnTail) : 0);
    return 0;
}

// ---- match-set family of the reference algorithm being reproduced -----------------------------------------------
namespace {
struct Family
{
    bool greedy = false;    // greedy leftmost non-overlapping occurrences (SSE4.2 / KMP; BMH under -o)
    bool mshort_o = false;  // memchr_short_search under -o: candidate walk
    bool need_walk = false; // a sequential pass over the ordered list is required (kg_greedy.hip)
    bool replay = false;    // -c through a block-structured function: end-of-text replay (kg_replay.h)
    bool neon_zero = false; // neon_search, count-only, max_count == 0 (tail-call convention)
    bool nlwalk = false;    // -c through simd_sse42_search / kmp_search with a '\n' inside the pattern (kg_greedy.hip (3))
    bool whole_text() const { return need_walk || replay || neon_zero || nlwalk; } // cannot be scanned in independent pieces
};
Family family_of(int algo, bool only_matching, bool lines, bool ww, bool track, size_t maxc, bool has_border, uint32_t m,
                 bool pat_has_newline = false)
{
    Family f;
    if (lines && pat_has_newline && (algo == KREP_RA_SSE42 || algo == KREP_RA_KMP))
    {
        f.nlwalk = true; // the line jump lands inside the match: a chain over every visited match, nothing else applies
        return f;
    }
    f.greedy = (algo == KREP_RA_SSE42 || algo == KREP_RA_KMP);
    if (only_matching)
    {
        if (algo == KREP_RA_SSE42)
            f.greedy = false; // -o: advance = index + 1 (krep.c:4842), in every mode
        else if (algo == KREP_RA_BMH && !lines)
            f.greedy = true; // -o without -c: i += pattern_len after a hit (krep.c:1371)
        else if (algo == KREP_RA_MEMCHR_SHORT)
            f.mshort_o = true; // -o: advance = candidate + pattern_len, also after a FAILED candidate (krep.c:4495)
    }
    f.need_walk = (f.greedy && has_border && m > 1) || f.mshort_o;
    // without -w a line holds a greedy hit iff it holds any occurrence (a line's first occurrence heads a cluster),
    // so plain -c needs no selection pass
    if (f.need_walk && !f.mshort_o && lines && !ww)
        f.need_walk = false;
    f.replay = lines && (algo == KREP_RA_AVX512 || algo == KREP_RA_NEON || (algo == KREP_RA_AVX2 && ww));
    f.neon_zero = algo == KREP_RA_NEON && maxc == 0 && !lines && !track;
    return f;
}
} // namespace
namespace kg {
// How may the text be cut?  kSplitPieces: independent pieces (start-offset ownership + halo) whose results merge.
// kSplitChain: pieces in text order, each taking the boundary record of the one before it (the greedy / -o walks: where the
// reference's scan stands; -c through simd_avx512_search / simd_avx2_search -w: the line-skip history the end-of-text replay
// needs, for neon_search with its grid origin; multi-pattern -c with a '\n' inside a pattern: newlines so far and the line of
// the last match — krep_gpu_seq_carry_t).  kSplitWhole: one window only — the newline-pattern -c walk of kg_greedy.hip (3),
// neon_search's max_count == 0 corner.
int split_mode(const search_params_t *p, const krep_gpu_config_t &c, size_t text_len)
{
    if (!p || p->use_regex || p->num_patterns == 0)
        return kSplitWhole;
    if (p->num_patterns > 1)
    {
        if (!p->count_lines_mode)
            return kSplitPieces;
        for (size_t i = 0; i < p->num_patterns; ++i)
            if (p->pattern_lens[i] && memchr(p->patterns[i], '\n', p->pattern_lens[i]))
                return kSplitChain; // emission-order line changes: pieces owned by END, chained by (newlines, last line) — round 5
        return kSplitPieces;
    }
    const char *pat = p->patterns && p->pattern_lens ? p->patterns[0] : p->pattern;
    const size_t m = p->patterns && p->pattern_lens ? p->pattern_lens[0] : p->pattern_len;
    if (!pat || m == 0)
        return kSplitPieces;
    search_params_t q = *p;
    q.pattern = pat;
    q.pattern_len = m;
    const int algo = mirror_effective(mirror_top(&q, c), &q, text_len);
    std::vector<uint8_t> f((const uint8_t *)pat, (const uint8_t *)pat + m);
    if (!p->case_sensitive)
        for (auto &b : f)
            b = lo8(b);
    const Family fam = family_of(algo, c.only_matching != 0, p->count_lines_mode, p->whole_word, p->track_positions, p->max_count,
                                 pattern_has_border(f.data(), m), (uint32_t)m, memchr(pat, '\n', m) != nullptr);
    if (fam.neon_zero || (fam.mshort_o && p->count_lines_mode))
        return kSplitWhole; // (the second: a class krep's main() never produces, krep.c:3811-3814 — one window, no boundary record)
    return (fam.need_walk || fam.replay || fam.nlwalk) ? kSplitChain : kSplitPieces;
}
bool shardable(const search_params_t *p, const krep_gpu_config_t &c, size_t text_len) { return split_mode(p, c, text_len) != kSplitWhole; }
// The left fold of the boundary record (include/krep_gpu.h, krep_gpu_seq_carry_t): used by a piece that knows its predecessor's
// record, and by the host after shards were scanned out of order.
krep_gpu_seq_carry_t fold_carry(const krep_gpu_seq_carry_t &in, const krep_gpu_seq_carry_t &pc)
{
    krep_gpu_seq_carry_t o = pc; // keeps the piece's local_* fields
    o.resume = std::max<uint64_t>(in.resume, pc.resume);
    // multi-pattern -c with a newline inside a pattern: newlines so far, line of the last match's start (include/krep_gpu.h)
    o.nl_before = in.nl_before + pc.local_nl;
    o.last_line = pc.local_last ? in.nl_before + 1ull + (pc.local_last - (1ull << 62)) : in.last_line;
    if (!pc.local_q1)
    {
        o.q1 = in.q1;
        o.nl1 = in.nl1 ? in.nl1 : in.q1 ? pc.local_first_nl1 : 0;
        o.g0 = in.g0;
        return o;
    }
    o.q1 = pc.local_q1;
    o.nl1 = pc.local_nl1;
    switch (pc.local_g0_kind)
    {
    case 1: o.g0 = pc.local_g0; break;
    case 2: o.g0 = in.q1 ? (in.nl1 ? in.nl1 : pc.local_first_nl1) : 0; break; // nl1 is stored + 1: it IS the next line start
    case 3: o.g0 = in.q1 ? (in.nl1 ? in.nl1 : in.g0) : 0; break;
    default: o.g0 = 0;
    }
    return o;
}
} // namespace kg
extern "C" void krep_gpu_debug_fold_carry(const krep_gpu_seq_carry_t *in, const krep_gpu_seq_carry_t *piece, krep_gpu_seq_carry_t *out)
{
    if (in && piece && out)
        *out = kg::fold_carry(*in, *piece);
}
extern "C" int krep_gpu_split_mode(const search_params_t *p, size_t text_len)
{
    return kg::split_mode(p, kg::current_config(), text_len);
}

// Where the reference's block loop stands when it enters the last kReplayWindow bytes (kg_replay.h): `cur`, and for
// neon_search whether the (unterminated) line holding `cur` is already counted.  Uses the per-unit info words the
// canonical -c pass over [0, X) just left in pl->post.
// last accepted occurrence with (buffer-relative) start < limit among the starts the canonical -c pass `lr` covered: the last
// unit reporting hits in its info word, re-scanned with records (pl->aux)
static int last_accepted_before(krep_gpu_plan *pl, const Window &w, const LitResult &lr, uint64_t own_lo, uint64_t limit,
                                hipStream_t st, bool *found, uint64_t *q)
{
    unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
    *found = false;
    if (limit <= lr.anchor || lr.n_units == 0)
        return 0;
    uint64_t lim_units = std::min<uint64_t>(lr.n_units, (limit - lr.anchor + lr.unit_bytes - 1) / lr.unit_bytes);
    while (lim_units)
    {
        uint64_t up1 = 0;
        if (tail_last_hit(pl->post.d_unitinfo, lim_units, d_slot, h_slot, st, &up1))
            return 2;
        if (!up1)
            return 0;
        const uint64_t u = up1 - 1, ulo = lr.anchor + u * lr.unit_bytes;
        LitPass ps;
        ps.ww = pl->ww;
        ps.own_lo = std::max<uint64_t>(ulo, own_lo);
        ps.own_hi = std::min<uint64_t>(ulo + lr.unit_bytes, limit);
        ps.sink = LitPass::OCC;
        ps.post = &pl->aux;
        LitResult r2;
        if (lit_pass(pl, w, ps, st, &r2))
            return 2;
        if (r2.total)
        {
            uint64_t rec[2];
            HIPCHK(hipMemcpy(rec, pl->aux.d_occ + 2 * (r2.total - 1), sizeof rec, hipMemcpyDeviceToHost));
            *found = true;
            *q = rec[0] - w.global_base;
            return 0;
        }
        lim_units = u; // every hit of that unit starts at or after `limit`: look further left
    }
    return 0;
}

// The end-of-text replay of a PIECE: where the reference's block loop stands when it enters the last kReplayWindow bytes follows
// from the folded boundary record alone (global offsets, stored + 1); `text_g` is indexed with global offsets.
// (`text_g` is a DEVICE address rebased so that it can be indexed with global offsets; the first global offset really held is text_lo.
//  The rebasing is integer arithmetic on the address — no pointer outside an object is formed on the host)
static const uint8_t *rebased(const void *d_held, uint64_t held_from)
{
    return reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(d_held) - (uintptr_t)held_from);
}
static int replay_from_record(krep_gpu_plan *pl, int algo, const krep_gpu_seq_carry_t &co, const uint8_t *text_g, uint64_t text_lo, uint64_t G,
                              hipStream_t st, uint64_t *extra)
{
    *extra = 0;
    const uint64_t X = G > kReplayWindow ? G - kReplayWindow : 0, B = algo == KREP_RA_AVX512 ? 64 : algo == KREP_RA_AVX2 ? 32 : 16;
    uint64_t cur;
    int open = 0;
    if (!co.q1)
        cur = (X / B) * B; // the block grid never left offset 0
    else if (co.nl1)
        cur = co.nl1 <= X ? co.nl1 + ((X - co.nl1) / B) * B : co.nl1; // restarted at the line start behind q
    else if (algo == KREP_RA_NEON)
    {
        // no restart on an unterminated line: the previous counted line's grid (a grid origin behind X — a text shorter than the
        // replay window — is where the loop stands; ADVICE r05: the unsigned difference wrapped)
        cur = co.g0 <= X ? co.g0 + ((X - co.g0) / B) * B : co.g0;
        open = 1;
    }
    else
        cur = G; // unterminated line counted: the clamped advance ended the scan (krep.c:5006-5008, :5211-5213)
    ReplayIn r{};
    r.algo = algo; r.m = pl->m; r.ww = pl->ww; r.n = G; r.cur = cur; r.open = open;
    r.text = text_g;
    r.pat = pl->d_pat;
    if (cur < G && cur + 1 < text_lo + 1) // (the replay reads global offsets >= cur - 1: they have to lie in what the caller holds)
        return kg::fail("end-of-text replay: the loop stands at %llu, in front of the bytes held (from %llu)", (unsigned long long)cur, (unsigned long long)text_lo);
    if (cur < G && tail_run_replay(r, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0], st, extra))
        return 2;
    return 0;
}

static int replay_entry(krep_gpu_plan *pl, int algo, const Window &w, const LitResult &lr, uint64_t X, hipStream_t st,
                        uint64_t *cur_out, int *open_out)
{
    const uint64_t n = w.text_len, B = algo == KREP_RA_AVX512 ? 64 : algo == KREP_RA_AVX2 ? 32 : 16;
    unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
    auto last_accepted_before = [&](uint64_t limit, bool *found, uint64_t *q) -> int {
        return ::last_accepted_before(pl, w, lr, 0, limit, st, found, q);
    };
    bool have_q = false;
    uint64_t q = 0;
    if (last_accepted_before(X, &have_q, &q))
        return 2;
    *open_out = 0;
    if (!have_q)
    {
        *cur_out = (X / B) * B; // the block grid never left offset 0
        return 0;
    }
    uint64_t nl = n;
    if (tail_find_next_newline(w.d_text, q, n, d_slot, h_slot, st, &nl))
        return 2;
    if (nl < n)
    {
        const uint64_t nls = nl + 1; // the loop restarted here after counting q's line
        *cur_out = nls <= X ? nls + ((X - nls) / B) * B : nls;
        return 0;
    }
    if (algo != KREP_RA_NEON)
    {
        *cur_out = n; // unterminated line counted: the clamped advance ends the scan (krep.c:5006-5008, :5211-5213)
        return 0;
    }
    // neon_search does not restart on an unterminated line: the grid is still the one set by the previous counted line
    uint64_t lsp1 = 0;
    if (tail_find_prev_newline(w.d_text, q, d_slot, h_slot, st, &lsp1))
        return 2;
    const uint64_t ls = lsp1; // start of q's line (0 when no '\n' precedes it)
    bool have_q2 = false;
    uint64_t q2 = 0, grid0 = 0;
    if (last_accepted_before(ls, &have_q2, &q2))
        return 2;
    if (have_q2)
    {
        uint64_t nl2 = n;
        if (tail_find_next_newline(w.d_text, q2, n, d_slot, h_slot, st, &nl2))
            return 2;
        grid0 = nl2 + 1; // <= ls
    }
    *cur_out = grid0 + ((X - grid0) / B) * B;
    *open_out = 1;
    return 0;
}

static int scan_literal(krep_gpu_plan *pl, int algo, const Window &w, match_position_t *d_pos, uint64_t cap, hipStream_t st,
                        int time_it, const krep_gpu_seq_carry_t *carry_in, krep_gpu_seq_carry_t *carry_out, krep_gpu_scan_out_t *out)
{
    const uint32_t m = pl->m;
    memset(out, 0, sizeof *out);
    if (carry_out)
        *carry_out = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
    // the reference function sees the WHOLE text: its length decides the delegation and every early-out
    if (m == 0 || w.global_len < m || w.text_len < m || w.own_lo >= w.own_hi)
        return 0;
    const size_t own_hi = std::min(w.own_hi, w.text_len);
    const bool whole = w.global_base == 0 && w.own_lo == 0 && own_hi + m > w.text_len && w.global_len == w.text_len;

    const Family fam = family_of(algo, pl->only_matching, pl->lines, pl->ww, pl->track, pl->max_count, pl->has_border, m,
                                 pl->has_newline);
    const bool mshort_o = fam.mshort_o, need_walk = fam.need_walk, replay = fam.replay;
    if (fam.nlwalk)
    {
        // -c through simd_sse42_search / kmp_search with a newline inside the pattern: all occurrences, the line number of every
        // start, the -w verdicts — then ONE thread walks the list the way the reference's loop moves (kg_greedy.hip (3)).
        // Pieces (round 5): a piece owns the occurrences that START in it; what couples it to the text in front of it is where
        // the reference's scan stands (resume), the line it counted last and the newlines so far (krep_gpu_seq_carry_t).
        const krep_gpu_seq_carry_t in = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
        krep_gpu_seq_carry_t local{};
        if (!whole && !carry_in && w.global_base + w.own_lo != 0)
            return kg::fail("-c through %s with a newline inside the pattern is a chain over every counted match: scan the whole "
                            "text in one window, or its pieces in text order through krep_gpu_scan_device_seq()", krep_gpu_algorithm_name(algo));
        if (pl->max_count == 0)
            return 0; // krep.c:4713, :1634
        HIPCHK(hipSetDevice(pl->device));
        if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
        unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
        uint64_t nl_halo = 0, nl_own = 0;
        if (tail_count_newlines(w.d_text, 0, w.own_lo, d_slot, h_slot, st, &nl_halo) ||
            tail_count_newlines(w.d_text, w.own_lo, own_hi, d_slot, h_slot, st, &nl_own))
            return 2;
        local.local_nl = nl_own;
        LitPass ps;
        ps.own_lo = w.own_lo; ps.own_hi = own_hi; ps.sink = LitPass::OCC; ps.post = &pl->post;
        LitResult lr0;
        if (lit_pass(pl, w, ps, st, &lr0))
            return 2;
        uint64_t lines_counted = 0, cp_out = in.resume, seen_out = in.last_line ? in.last_line : ~0ull;
        if (lr0.total)
        {
            if (lr0.total > pl->nl_cap)
            {
                if (pl->d_nl_rec) (void)hipFree(pl->d_nl_rec);
                if (pl->d_nl_ln) (void)hipFree(pl->d_nl_ln);
                pl->d_nl_rec = nullptr; pl->d_nl_ln = nullptr; pl->nl_cap = 0;
                const uint64_t want_n = lr0.total + lr0.total / 4 + 1024;
                HIPCHK(hipMalloc(&pl->d_nl_rec, want_n * sizeof(match_position_t)));
                HIPCHK(hipMalloc(&pl->d_nl_ln, want_n * sizeof(uint64_t)));
                pl->nl_cap = want_n;
            }
            if (krep_gpu_line_numbers_ex(w.d_text, w.text_len, w.global_base, (const match_position_t *)pl->post.d_occ, lr0.total, pl->d_nl_ln, st))
                return 2;
            const uint32_t k0 = (uint32_t)((const uint8_t *)memchr(pl->pats[0].data(), '\n', m) - pl->pats[0].data());
            // buffer-relative line numbers (1 + newlines in [0, start)) -> global ones: + newlines in front of the buffer
            const uint64_t line_off = in.nl_before - nl_halo;
            if (post_nlwalk(pl->post, w.d_text, w.text_len, algo == KREP_RA_KMP ? kNlWalkKmp : kNlWalkSse42, m, k0, pl->ww,
                            pl->only_matching, pl->max_count == SIZE_MAX ? ~0ull : (uint64_t)pl->max_count, lr0.total, pl->d_nl_ln,
                            pl->d_ctr, pl->h_ctr, st, &lines_counted, w.global_base, w.global_len, line_off, in.resume,
                            in.last_line ? in.last_line : ~0ull, &cp_out, &seen_out))
                return 2;
        }
        if (time_it)
        {
            HIPCHK(hipEventRecord(pl->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
            out->kernel_ms = ms;
        }
        if (carry_out)
        {
            // (the line counted last, relative to own_lo and biased like the multi-pattern record's, so that fold_carry() re-bases it)
            if (seen_out != ~0ull && seen_out != (in.last_line ? in.last_line : ~0ull))
                local.local_last = (1ull << 62) + (seen_out - (in.nl_before + 1ull));
            krep_gpu_seq_carry_t o = kg::fold_carry(in, local);
            if (!local.local_last)
                o.last_line = in.last_line;
            o.resume = std::max<uint64_t>(in.resume, cp_out);
            *carry_out = o;
        }
        out->total_matches = lines_counted;
        out->line_count = lines_counted;
        out->has_newline = whole ? 0 : 1; // (pieces: krep_gpu_combine_line_counts adds the counts up, the record has settled the cuts)
        out->head_line_hit = out->tail_line_hit = whole ? lines_counted != 0 : 0;
        out->count = lines_counted; // the walk applies max_count the way the functions do (a break before the increment)
        return 0;
    }
    // the walks couple neighbouring matches: a window that does not start the text needs the boundary record of the text
    // in front of it (where the reference's scan stands: krep_gpu_seq_carry_t::resume)
    if (need_walk && !whole && !carry_in && w.global_base + w.own_lo != 0)
        return kg::fail("the greedy / only-matching families couple neighbouring matches: scan the whole text in one window, or "
                        "its pieces in text order through krep_gpu_scan_device_seq() (%s, pattern length %u)",
                        krep_gpu_algorithm_name(algo), m);

    // ---- where the block-structured functions put their quirks: positions in the WHOLE text, translated to the buffer
    uint64_t excl_lo = 0, excl_hi = 0, ww_exempt = ~0ull;
    {
        const uint64_t G = w.global_len, base = w.global_base;
        auto to_local = [&](uint64_t g) -> uint64_t { return g >= base && g - base < w.text_len ? g - base : ~0ull; };
        if (algo == KREP_RA_AVX512 && !pl->lines && G >= 64 && (G % 64) < (uint64_t)m - 1)
        { // krep.c:5171: the last full 64-byte block is stepped over unexamined when remaining < (m-1)+64
            const uint64_t ghi = G - G % 64, glo = ghi - 64;
            if (ghi > base && glo < base + w.text_len)
            {
                excl_lo = glo > base ? glo - base : 0;
                excl_hi = std::min<uint64_t>(ghi - base, w.text_len);
            }
        }
        if (pl->ww && !pl->lines)
        { // the scalar tail call of the block functions sees the tail as its own text: no left context at its first byte
            const uint64_t B = algo == KREP_RA_AVX2 ? 32 : algo == KREP_RA_AVX512 ? 64 : algo == KREP_RA_NEON ? 16 : 0;
            if (B && (G % B) >= m)
                ww_exempt = to_local(G - G % B);
        }
    }

    const uint64_t maxc = pl->max_count;
    uint64_t want = 0; // records the caller can use
    if (d_pos && cap && !pl->lines)
    {
        want = maxc;
        if ((algo == KREP_RA_KMP || algo == KREP_RA_MEMCHR) && maxc != SIZE_MAX)
            want = maxc + 1; // KMP stores one more (krep.c:1717); the memchr batch quirk needs the next record too
        want = std::min<uint64_t>(want, cap);
    }

    HIPCHK(hipSetDevice(pl->device));
    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    uint64_t total = 0, lines = 0;
    unsigned long long summary = 0;
    LitResult lr;
    bool ev1_recorded = false;
    if (replay)
    {
        // canonical -c over the starts before the last kReplayWindow bytes, then the reference's own walk over the rest
        const uint64_t G = w.global_len, X = G > kReplayWindow ? G - kReplayWindow : 0;
        if (!whole)
        {
            // ---- a PIECE of the text (krep_gpu_scan_device_seq).  Only the piece that ends the text runs the replay; what it
            // needs from the text in front of it is the line-skip history {last accepted occurrence q, first '\n' behind it},
            // which every piece extends (krep_gpu_seq_carry_t) — the piece's own contribution is reported next to the folded
            // state, so that shards scanned out of order can be folded afterwards.
            const uint64_t base = w.global_base, B = algo == KREP_RA_AVX512 ? 64 : algo == KREP_RA_AVX2 ? 32 : 16;
            const bool final_piece = base + own_hi >= G;
            if (base + w.own_lo != 0 && !carry_in)
                return kg::fail("-c through %s restarts its block grid at every counted line: scan the whole text in one window, "
                                "or its pieces in text order through krep_gpu_scan_device_seq()", krep_gpu_algorithm_name(algo));
            if (final_piece && (base + w.text_len != G || X < base + w.own_lo + B))
                return kg::fail("-c through %s: the piece that ends the text must hold its last %llu bytes", krep_gpu_algorithm_name(algo),
                                (unsigned long long)(kReplayWindow + B));
            const uint64_t lim = final_piece ? X - base : own_hi; // the canonical pass covers the starts in [own_lo, lim)
            const uint64_t nl_end = final_piece ? w.text_len : own_hi; // where this piece's newline searches stop
            unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
            if (lim > w.own_lo)
            {
                LitPass ps;
                ps.ww = pl->ww; ps.lines = true; ps.own_lo = w.own_lo; ps.own_hi = lim; ps.post = &pl->post;
                if (lit_pass(pl, w, ps, st, &lr))
                    return 2;
            }
            const krep_gpu_seq_carry_t in = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
            krep_gpu_seq_carry_t pc{}; // this piece's own contribution
            bool have_q = false;
            uint64_t q = 0;
            if (lim > w.own_lo && last_accepted_before(pl, w, lr, w.own_lo, lim, st, &have_q, &q))
                return 2;
            {
                uint64_t nl = nl_end; // first newline of the piece (the record of a piece without an occurrence; neon_search's kind 2)
                if ((!have_q || algo == KREP_RA_NEON) && tail_find_next_newline(w.d_text, w.own_lo, nl_end, d_slot, h_slot, st, &nl))
                    return 2;
                pc.local_first_nl1 = nl < nl_end ? base + nl + 1 : 0;
            }
            if (have_q)
            {
                uint64_t nl = nl_end;
                if (tail_find_next_newline(w.d_text, q, nl_end, d_slot, h_slot, st, &nl))
                    return 2;
                pc.local_q1 = base + q + 1;
                pc.local_nl1 = nl < nl_end ? base + nl + 1 : 0;
                if (algo == KREP_RA_NEON)
                {
                    // the grid origin in effect for q's line: (first newline behind the last accepted occurrence on an earlier
                    // line) + 1 — from this piece if that occurrence lies in it, else from the record in front of it
                    uint64_t lsp1 = 0;
                    if (tail_find_prev_newline(w.d_text, q, d_slot, h_slot, st, &lsp1))
                        return 2;
                    if (lsp1 <= w.own_lo) // no newline in [own_lo, q): q's line started in front of this piece
                        pc.local_g0_kind = 3;
                    else
                    {
                        bool have_q2 = false;
                        uint64_t q2 = 0;
                        if (last_accepted_before(pl, w, lr, w.own_lo, lsp1, st, &have_q2, &q2))
                            return 2;
                        if (!have_q2)
                            pc.local_g0_kind = 2;
                        else
                        {
                            uint64_t nl2 = nl_end;
                            if (tail_find_next_newline(w.d_text, q2, nl_end, d_slot, h_slot, st, &nl2))
                                return 2;
                            pc.local_g0_kind = 1;
                            pc.local_g0 = base + nl2 + 1; // nl2 < lsp1 <= q: it exists
                        }
                    }
                }
            }
            if (final_piece)
                pc.local_lines = lr.lines + 1; // (what krep_gpu_replay_tail() starts from when the record turns out different)
            const krep_gpu_seq_carry_t co = kg::fold_carry(in, pc);
            if (carry_out)
                *carry_out = co;
            total = lr.total; lines = lr.lines; summary = lr.summary;
            if (final_piece)
            {
                uint64_t extra = 0;
                if (replay_from_record(pl, algo, co, rebased(w.d_text, base), base ? base + 1 : 0, G, st, &extra)) // (global offsets >= cur - 1 >= base: inside the buffer)
                    return 2;
                // the lines the replay counts are new ones (it starts behind the line of q), and the line open at the piece's
                // start can only be among them when nothing in front of the piece counted it: the canonical head bit stands
                lines = lr.lines + extra;
                total = lines;
            }
        }
        else
        {
            uint64_t cur = 0, extra = 0;
            int open = 0;
            if (X)
            {
                LitPass ps;
                ps.ww = pl->ww; ps.lines = true; ps.own_lo = 0; ps.own_hi = X; ps.post = &pl->post;
                if (lit_pass(pl, w, ps, st, &lr))
                    return 2;
                if (replay_entry(pl, algo, w, lr, X, st, &cur, &open))
                    return 2;
            }
            ReplayIn r{};
            r.algo = algo; r.m = m; r.ww = pl->ww; r.n = G; r.cur = cur; r.open = open;
            r.text = w.d_text; r.pat = pl->d_pat;
            if (cur < G && tail_run_replay(r, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0], st, &extra))
                return 2;
            lines = lr.lines + extra;
            total = lines; // occurrence totals are not defined by a -c scan
            summary = lines ? (kLnHead | kLnTail) : 0;
        }
    }
    else if (!need_walk)
    {
        if (fam.neon_zero)
        {
            // count-only with max_count == 0: the first body hit returns 0 (krep.c:4616); with no body hit the tail call's
            // BMH returns 1 on its first hit (krep.c:1355-1367)
            const uint64_t G = w.global_len, T = G - G % 16;
            if (!whole)
                return kg::fail("neon_search with max_count == 0 needs the whole text in one window");
            LitPass ps;
            ps.ww = pl->ww; ps.own_lo = 0; ps.own_hi = T; ps.ww_exempt = ww_exempt; ps.post = &pl->post;
            if (lit_pass(pl, w, ps, st, &lr))
                return 2;
            uint64_t ret = 0;
            if (lr.total == 0 && G - T >= m)
            {
                ps.own_lo = T; ps.own_hi = G;
                if (lit_pass(pl, w, ps, st, &lr))
                    return 2;
                ret = lr.total ? 1 : 0;
            }
            if (time_it) HIPCHK(hipEventRecord(pl->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
            out->count = ret;
            out->total_matches = lr.total;
            return 0;
        }
        LitPass ps;
        ps.ww = pl->ww; ps.lines = pl->lines; ps.own_lo = w.own_lo; ps.own_hi = own_hi;
        ps.excl_lo = excl_lo; ps.excl_hi = excl_hi; ps.ww_exempt = ww_exempt;
        ps.sink = want ? LitPass::RECORDS : LitPass::COUNT;
        ps.d_out = (uint64_t *)d_pos; ps.out_cap = want; ps.post = &pl->post;
        ps.ev_end = time_it ? pl->ev1 : nullptr;
        if (lit_pass(pl, w, ps, st, &lr))
            return 2;
        ev1_recorded = time_it != 0 && lr.n_units != 0;
        total = lr.total; lines = lr.lines; summary = lr.summary;
    }
    else
    {
        // sequential families on the ordered list (kg_greedy.hip): greedy non-overlapping selection (SSE4.2 / KMP, and BMH
        // under -o) over all occurrences, or memchr_short's -o walk over the first-byte candidates.  A window inside the text
        // starts where the reference's scan stands (carry_in->resume): starts in front of that point are consumed, the first
        // one at or behind it is looked at afresh — exactly what the reference's loop does after `advance`.
        const bool ww_first = pl->ww && algo == KREP_RA_BMH; // BMH -o: a -w rejected hit does not consume (krep.c:1323-1329)
        const uint64_t resume_in = carry_in ? carry_in->resume : 0;
        size_t lo_eff = w.own_lo;
        if (resume_in > w.global_base + w.own_lo)
            lo_eff = (size_t)std::min<uint64_t>(own_hi, resume_in - w.global_base);
        uint64_t resume_out = 0;
        // ---- a pattern of ONE repeated byte (`  `, `--`, `aa`), counted: inside a run of R such bytes the kept matches start at every
        // m-th byte — no list of all occurrences (kg_runs.hip; round 6: `-c -o '  '` materialised 188 M occurrences at 0.65 TB/s)
        bool runs_done = false;
        if (!mshort_o && !pl->ww && !pl->lines && want == 0 && m >= 2 && !getenv("KREP_GPU_NO_RUNS"))
        {
            bool uniform = true;
            for (uint32_t i = 1; i < m; ++i)
                uniform = uniform && pl->pat_folded[i] == pl->pat_folded[0];
            if (uniform)
            {
                uint64_t end_p1 = 0;
                const int rc = runs_count_greedy(w.d_text, w.text_len, lo_eff, own_hi, m, pl->pat_folded[0], !pl->cs, pl->num_cu, &pl->d_ctr->pad[1],
                                                 &pl->h_ctr->pad[1], st, &total, &end_p1);
                if (rc == 2)
                    return 2;
                if (rc == 0)
                {
                    runs_done = true;
                    resume_out = end_p1 ? w.global_base + end_p1 : 0;
                }
            }
        }
        LitPass ps;
        ps.own_lo = lo_eff; ps.own_hi = own_hi; ps.sink = LitPass::OCC; ps.post = &pl->post;
        ps.first_byte = mshort_o;
        ps.ww = ww_first;
        if (!runs_done && lit_pass(pl, w, ps, st, &lr))
            return 2;
        WalkSpec ws{};
        ws.mode = mshort_o ? (pl->lines ? kWalkShortOLines : kWalkShortO) : kWalkGreedy;
        if (ws.mode == kWalkShortOLines && !whole)
            return kg::fail("-c -o through memchr_short_search jumps to the next line start after every counted line: scan the whole "
                            "text in one window");
        ws.m = m;
        ws.ww = pl->ww && !ww_first;
        ws.lines = pl->lines;
        ws.ci = !pl->cs;
        ws.b1 = m > 1 ? pl->pat_folded[1] : 0;
        ws.b2 = m > 2 ? pl->pat_folded[2] : 0;
        if (lr.total)
        {
            HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
            int rc = post_walk(pl->post, w.d_text, w.text_len, w.global_base, ws, lr.total, (uint64_t *)d_pos, want, pl->d_ctr,
                               pl->h_ctr, st, &total, &lines, &resume_out);
            if (rc)
                return rc;
        }
        if (carry_out)
            carry_out->resume = std::max(resume_in, resume_out);
        summary = total ? (kLnHead | kLnTail) : 0;
        if (pl->lines && !whole)
        {
            // line bits of the owned window for the left-to-right fold over the pieces (krep_gpu_combine_line_counts): is there
            // a '\n' in [own_lo, own_hi), does a survivor start at or before the first / behind the last one
            unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
            uint64_t first_nl = own_hi, last_p1 = 0;
            if (tail_find_next_newline(w.d_text, w.own_lo, own_hi, d_slot, h_slot, st, &first_nl))
                return 2;
            summary = 0;
            if (first_nl < own_hi)
            {
                if (tail_find_prev_newline(w.d_text, own_hi, d_slot, h_slot, st, &last_p1))
                    return 2;
                summary |= kLnNl;
            }
            if (total)
            {
                uint64_t s_first[2], s_last[2]; // the survivors were compacted into post.d_surv for the line count
                HIPCHK(hipMemcpy(s_first, pl->post.d_surv, sizeof s_first, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(s_last, pl->post.d_surv + 2 * (total - 1), sizeof s_last, hipMemcpyDeviceToHost));
                const uint64_t f = s_first[0] - w.global_base, l = s_last[0] - w.global_base;
                if (!(summary & kLnNl))
                    summary |= kLnHead | kLnTail;
                else
                    summary |= (f <= first_nl ? kLnHead : 0) | (l >= last_p1 ? kLnTail : 0);
            }
        }
    }
    if (time_it)
    {
        if (!ev1_recorded) HIPCHK(hipEventRecord(pl->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
        out->kernel_ms = ms;
    }
    else
        HIPCHK(hipStreamSynchronize(st));
    out->total_matches = total;
    out->line_count = lines;
    out->has_newline = (summary & kLnNl) != 0;
    out->head_line_hit = (summary & kLnHead) != 0;
    out->tail_line_hit = (summary & kLnTail) != 0;
    Verdict v = verdict_for(algo, pl, total, lines, d_pos != nullptr);
    out->count = v.ret;
    if (d_pos && cap && pl->track && !pl->lines)
    {
        // records the reference semantics need (memchr: one more than max_count, the host fixes the order)
        const uint64_t needed = (algo == KREP_RA_MEMCHR && maxc != SIZE_MAX && maxc != 0) ? std::min<uint64_t>(total, maxc + 1)
                                                                                           : v.store;
        out->overflow = needed > cap;
        out->stored = std::min<uint64_t>(needed, std::min<uint64_t>(total, want));
    }
    return 0;
}

// aho_corasick_search -c when a pattern contains '\n': matches are visited in emission order (end ascending, longest
// first) and the counter is bumped whenever the line of a match START differs from the line of the previously counted
// one (aho_corasick.c:383-396) — with a newline inside a pattern a later match may start on an EARLIER line, so a line
// can be counted more than once.  Reproduced literally: the ordered match list, the line number of every start
// (kg_format.hip), the number of changes along the list.
// Pieces (round 5): a piece owns the matches that END in [own_lo, own_hi) — the pieces' lists, concatenated in text order, ARE
// the emission order — and what couples it to the text in front of it is two numbers (krep_gpu_seq_carry_t): the newlines so
// far (its line numbers are global) and the line of the last match's start (its first match counts only on another line).
// The buffer must hold the longest pattern's length in front of own_lo (every piece of run_pieces does).
static int scan_ac_newline_lines(krep_gpu_plan *pl, const Window &w, hipStream_t st, int time_it, const krep_gpu_seq_carry_t *carry_in,
                                 krep_gpu_seq_carry_t *carry_out, krep_gpu_scan_out_t *out)
{
    const krep_gpu_seq_carry_t in = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
    krep_gpu_seq_carry_t local{};
    const size_t own_hi = std::min(w.own_hi, w.text_len);
    auto leave = [&]() {
        if (carry_out)
            *carry_out = kg::fold_carry(in, local);
    };
    if (pl->max_count == 0 || w.own_lo >= own_hi) // aho_corasick.c:316
    {
        leave();
        return 0;
    }
    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    unsigned long long *d_slot = &pl->d_ctr->pad[0], *h_slot = &pl->h_ctr->pad[0];
    krep_gpu_scan_out_t o1;
    int rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base, nullptr, 0,
                     pl->ww, false, false, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1, 2);
    if (rc)
        return rc;
    const uint64_t total = o1.total_matches;
    uint64_t changes = 0, nl_halo = 0, nl_own = 0;
    if (tail_count_newlines(w.d_text, 0, w.own_lo, d_slot, h_slot, st, &nl_halo) || tail_count_newlines(w.d_text, w.own_lo, own_hi, d_slot, h_slot, st, &nl_own))
        return 2;
    local.local_nl = nl_own;
    if (total)
    {
        if (total > pl->nl_cap) // grow-only scratch of the plan (no allocation per call)
        {
            if (pl->d_nl_rec) (void)hipFree(pl->d_nl_rec);
            if (pl->d_nl_ln) (void)hipFree(pl->d_nl_ln);
            pl->d_nl_rec = nullptr;
            pl->d_nl_ln = nullptr;
            pl->nl_cap = 0;
            const uint64_t want = total + total / 4 + 1024;
            HIPCHK(hipMalloc(&pl->d_nl_rec, want * sizeof(match_position_t)));
            HIPCHK(hipMalloc(&pl->d_nl_ln, want * sizeof(uint64_t)));
            pl->nl_cap = want;
        }
        if (!pl->d_nl_ln) // (the list road of plain -c sizes only the records)
            HIPCHK(hipMalloc(&pl->d_nl_ln, pl->nl_cap * sizeof(uint64_t)));
        match_position_t *d_rec = pl->d_nl_rec;
        uint64_t *d_ln = pl->d_nl_ln;
        rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base, d_rec, total,
                     pl->ww, false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1, 2);
        if (!rc)
            rc = krep_gpu_line_numbers_ex(w.d_text, w.text_len, w.global_base, d_rec, total, d_ln, st); // 1 + newlines in [buffer start, start)
        if (!rc)
            rc = tail_count_changes(d_ln, total, d_slot, h_slot, st, &changes);
        if (rc)
            return rc;
        uint64_t ends[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(&ends[0], d_ln, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&ends[1], d_ln + (total - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // line of a start relative to own_lo (negative for a start in the halo): (ln - 1) - newlines of the halo
        const uint64_t bias = 1ull << 62;
        const uint64_t first_rel = bias + (ends[0] - 1ull) - nl_halo, last_rel = bias + (ends[1] - 1ull) - nl_halo;
        if (in.last_line && in.nl_before + 1ull + (first_rel - bias) == in.last_line)
            changes -= 1; // the first match of this piece starts on the line of the text's last match so far: not counted again
        local.local_last = last_rel;
    }
    if (time_it)
    {
        HIPCHK(hipEventRecord(pl->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
        out->kernel_ms = ms;
    }
    leave();
    out->total_matches = total;
    out->line_count = changes;
    out->has_newline = 1; // (krep_gpu_combine_line_counts then adds the pieces' counts up: nothing merges across a cut that the record has not settled)
    out->head_line_hit = out->tail_line_hit = 0;
    out->count = std::min<uint64_t>(changes, pl->max_count);
    return 0;
}

// aho_corasick_search with count_lines_mode, no newline inside any pattern, on a large sparse text: the matches are scanned as
// RECORDS by the fast kernel (staging + post-pass into the plan's grow-only list) and the lines are counted ON THE LIST — two
// neighbours of the end-ordered list lie on different lines iff the gap between them holds a '\n' (kg_tail.hip,
// tail_line_gaps).  The in-kernel line bookkeeping of kg_ac.hip (exact newline mask of every 1-KiB cell, hit and newline
// bitmaps per unit, a line pass over all 16 cells of every unit) runs at 0.43 of the HBM roofline on BASELINE config 4; this
// road costs the records scan (0.61) plus ~2 cache lines per match.  The line summary of the window (has_newline / head / tail,
// krep_gpu_combine_line_counts) comes from two early-exit newline sweeps and the first and last record.  Matches are owned by
// their END (last byte in [own_lo, own_hi)) on both roads, so neighbouring pieces may take different roads (ac_scan).
// Returns 1 when the text turns out to be too dense for the list (the caller takes the in-kernel road), 2 on error.
constexpr size_t kAcLinesOnListMin = (size_t)32 << 20;
// ---- a dictionary with SHORT patterns beside many longer ones on a word-like text (round 6) ----------------------------------
// The anchored scan (kg_ac_anchor.hip) is what makes a word dictionary fast on word text — 20x — and it is refused when the dictionary
// holds a 1..3-byte pattern: the pair filter's wildcard entries for such a pattern pass a 2-gram's worth of positions, and the exact
// anchor buckets are keyed on four bytes.  One `the` or `of` in a list of a thousand words sent the whole scan back to 0.027 of the
// roofline.  Here the two parts are dictionaries of their own: the patterns of >= 4 bytes (anchored), the 1..3-byte ones (the tiny
// register-compare kernel when they qualify, the general kernel otherwise); their counts add, and their record lists — each in
// aho_corasick_search's order, END ascending and longest first (aho_corasick.c:383-437), patterns of different lengths never produce the
// same record — are merged by two stable radix sorts, by start and then by END.  Decided once per plan, on its first text of >= 1 MiB,
// by sampling that text with the long part (ac_anchor_prepare): split only where the long part anchors.  Returns 1: not applicable.
// (a merged list beyond this — a dense short part — is not worth two sorts with 32 bytes of scratch per record; $KREP_GPU_AC_SPLIT_MAX: the tests' way to the fallback)
static const uint64_t kAcSplitMaxRecords = [] { const char *e = getenv("KREP_GPU_AC_SPLIT_MAX"); return e && atoll(e) > 0 ? (uint64_t)atoll(e) : (1ull << 28); }();
// the decision (once per plan, on its first text of >= 1 MiB): 2 = split, 1 = one dictionary; 0 = not decided yet (the text is too small to sample)
static int ac_split_decide(krep_gpu_plan *pl, const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, hipStream_t st)
{
    kg::AcTables *t = pl->ac;
    if (pl->ac_split)
        return pl->ac_split;
    if (!t || t->has_empty || !(t->has1 || t->has2 || t->has3) || pl->max_count == 0 || getenv("KREP_GPU_AC_NO_SPLIT"))
        return pl->ac_split = 1;
    const size_t hi = std::min(own_hi, text_len);
    if (text_len < (1u << 20) || hi <= own_lo || hi - own_lo < (1u << 19))
        return 0; // (too small to sample: decided by a later text)
    std::vector<const char *> pl_long, pl_short;
    std::vector<size_t> ln_long, ln_short;
    for (size_t i = 0; i < pl->sp.num_patterns; ++i)
    {
        const bool is_long = pl->sp.pattern_lens[i] >= 4;
        (is_long ? pl_long : pl_short).push_back(pl->sp.patterns[i]);
        (is_long ? ln_long : ln_short).push_back(pl->sp.pattern_lens[i]);
    }
    pl->ac_split = 1;
    if (pl_long.size() < 8 || pl_short.empty())
        return 1;
    search_params_t sub = pl->sp;
    sub.patterns = pl_long.data();
    sub.pattern_lens = ln_long.data();
    sub.num_patterns = pl_long.size();
    sub.pattern = nullptr;
    sub.pattern_len = 0;
    pl->ac_long = kg::ac_build(sub, pl->device);
    if (!pl->ac_long)
    {
        krep_gpu_clear_error();
        return 1;
    }
    if (hipSetDevice(pl->device) != hipSuccess || kg::ac_anchor_prepare(pl->ac_long, d_text, text_len, own_lo, hi, st) == 2)
        (void)hipGetLastError();
    if (pl->ac_long->anch_state != 2)
    { // the long part gains nothing from anchors on this text: one scan of the whole dictionary, as always
        kg::ac_free(pl->ac_long);
        pl->ac_long = nullptr;
        return 1;
    }
    sub.patterns = pl_short.data();
    sub.pattern_lens = ln_short.data();
    sub.num_patterns = pl_short.size();
    pl->ac_short = kg::ac_build(sub, pl->device);
    if (!pl->ac_short)
    {
        krep_gpu_clear_error();
        kg::ac_free(pl->ac_long);
        pl->ac_long = nullptr;
        return 1;
    }
    if (getenv("KREP_GPU_DEBUG"))
        fprintf(stderr, "krep-gpu: multi-pattern scan split: %zu patterns of >= 4 bytes (anchored) + %zu of 1..3 bytes, lists merged by (end, start)\n", pl_long.size(),
                pl_short.size());
    return pl->ac_split = 2;
}
static int scan_ac_split(krep_gpu_plan *pl, const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, size_t global_base,
                         match_position_t *d_pos, uint64_t cap, hipStream_t st, int time_it, krep_gpu_scan_out_t *out)
{
    if (ac_split_decide(pl, d_text, text_len, own_lo, own_hi, st) != 2)
        return 1;
    // Under a max_count smaller than the caller's list the result is the first max_count records of the MERGED list: either part may
    // contribute all of them, so both parts' first max_count records go to a scratch list of the plan, are merged there, and the first
    // max_count are copied out.  Without such a limit the caller's list holds everything (or overflows, as always).
    const bool want = d_pos && pl->track;
    const bool limited = want && (uint64_t)pl->max_count < cap;
    match_position_t *dst = d_pos;
    uint64_t room = cap;
    if (limited)
    {
        const uint64_t need = 2 * (uint64_t)pl->max_count;
        if (need > pl->split_cap)
        {
            if (pl->d_split_rec) (void)hipFree(pl->d_split_rec);
            pl->d_split_rec = nullptr;
            pl->split_cap = 0;
            if (hipMalloc(&pl->d_split_rec, need * sizeof(match_position_t)) != hipSuccess)
            {
                (void)hipGetLastError();
                return 1; // (no room for the scratch list: one scan of the whole dictionary)
            }
            pl->split_cap = need;
        }
        dst = pl->d_split_rec;
        room = need;
    }
    krep_gpu_scan_out_t o1, o2;
    int rc = kg::ac_scan(pl->ac_long, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, d_text, text_len, own_lo, own_hi, global_base, want ? dst : nullptr,
                         want ? (limited ? (uint64_t)pl->max_count : room) : 0, pl->ww, false, pl->track, pl->max_count, st, time_it, pl->ev0, pl->ev1, &o1);
    if (rc)
        return rc;
    const uint64_t s1 = want ? o1.stored : 0;
    match_position_t *d2 = (want && room > s1) ? dst + s1 : nullptr; // (no room behind the first list: the second part is counted, and the sum overflows)
    rc = kg::ac_scan(pl->ac_short, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, d_text, text_len, own_lo, own_hi, global_base, d2,
                     d2 ? (limited ? (uint64_t)pl->max_count : room - s1) : 0, pl->ww, false, pl->track, pl->max_count, st, time_it, pl->ev0, pl->ev1, &o2);
    if (rc)
        return rc;
    memset(out, 0, sizeof *out);
    const uint64_t total = o1.total_matches + o2.total_matches;
    out->total_matches = total;
    out->count = std::min<uint64_t>(total, (uint64_t)pl->max_count);
    out->head_line_hit = out->tail_line_hit = total != 0;
    out->kernel_ms = o1.kernel_ms + o2.kernel_ms;
    if (want)
    {
        const uint64_t s2 = d2 ? o2.stored : 0, n = s1 + s2;
        out->overflow = out->count > cap;
        if (!out->overflow)
        {
            // (start, then END: records with one END come out longest first — the smaller start)
            if (s1 && s2 &&
                (n > kAcSplitMaxRecords || kg::order_records(dst, n, global_base + text_len + 1, st, false) ||
                 kg::order_records(dst, n, global_base + text_len + 1, st, true)))
            {
                // (the sort's scratch — 32 bytes per record — did not fit, or the list exceeds the device sort's 2^31-1 items: a dictionary whose
                //  short part is dense.  One scan of the whole dictionary, from now on.)
                krep_gpu_clear_error();
                pl->ac_split = 1;
                return 1;
            }
            out->stored = std::min<uint64_t>(n, out->count);
            if (limited && out->stored)
            {
                HIPCHK(hipMemcpyAsync(d_pos, dst, out->stored * sizeof(match_position_t), hipMemcpyDeviceToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
            }
        }
    }
    // (a later text on which the long part no longer anchors — its decision follows the text, kg_ac.hip — ends the split)
    if (pl->ac_long->anch_state == 1)
        pl->ac_split = 1;
    return 0;
}

static int scan_ac_lines_on_list(krep_gpu_plan *pl, const Window &w, hipStream_t st, int time_it, krep_gpu_scan_out_t *out)
{
    memset(out, 0, sizeof *out);
    if (pl->max_count == 0) // aho_corasick.c:316
        return 0;
    const size_t own_hi = std::min(w.own_hi, w.text_len);
    if (w.own_lo >= own_hi)
        return 0;
    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    // More matches than this and the list is no shortcut: a record costs its 16 bytes plus ~2 random cache lines of gap test
    // (~48 ps), the in-kernel line pass 0.1-0.13 ps per byte of text — break-even at one match per ~400 bytes (measured at
    // 32 GiB, profiles/r04_dictionaries.txt: `xq zj`, one per 500 bytes, 3.1 TB/s on the list against 2.2 in the kernel;
    // `er th an`, one per 300, 1.8 against 2.0; `a Sherlock`, one per 28, 0.65 against 2.4).  The plan remembers.
    // (a tiny dictionary counts its lines in registers since round 5 — 3.4-4.4 TB/s whatever the density: the list is ahead only
    //  below one match per ~1000 bytes; profiles/r05_dictionaries.txt: `he she hers`, one per 811, 3.34 on the list against 3.44)
    const uint64_t dense = w.text_len / (ac_counts_lines_in_registers(pl->ac) ? 1000 : 400) + 4096;
    if (pl->lines_list_off)
        return 1;
    if (pl->nl_cap == 0)
    {
        const uint64_t want = std::max<uint64_t>(w.text_len / 1024, 1u << 16);
        HIPCHK(hipMalloc(&pl->d_nl_rec, want * sizeof(match_position_t)));
        pl->nl_cap = want;
    }
    krep_gpu_scan_out_t o1;
    for (int attempt = 0;; ++attempt)
    {
        int rc = 0;
        bool merged = false; // (the list below is the two parts', merged)
        if (ac_split_decide(pl, w.d_text, w.text_len, w.own_lo, own_hi, st) == 2)
        {
            // (a dictionary with short patterns on word-like text, scan_ac_split: the two parts' END-owned lists, merged into the order the
            //  one list would have — END ascending, longest first —, the line gaps counted on the merged list below)
            krep_gpu_scan_out_t oa, ob;
            rc = ac_scan(pl->ac_long, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base, pl->d_nl_rec, pl->nl_cap,
                         pl->ww, false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &oa, 2);
            if (rc)
                return rc;
            const uint64_t sa = oa.stored;
            match_position_t *d2 = pl->nl_cap > sa ? pl->d_nl_rec + sa : nullptr;
            rc = ac_scan(pl->ac_short, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base, d2, d2 ? pl->nl_cap - sa : 0,
                         pl->ww, false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &ob, 2);
            if (rc)
                return rc;
            memset(&o1, 0, sizeof o1);
            o1.total_matches = oa.total_matches + ob.total_matches;
            o1.overflow = oa.overflow || ob.overflow || !d2 || o1.total_matches > pl->nl_cap;
            o1.line_count = ~0ull; // (counted on the merged list below)
            if (!o1.overflow && oa.stored && ob.stored &&
                (o1.total_matches > kAcSplitMaxRecords || kg::order_records(pl->d_nl_rec, o1.total_matches, w.global_base + w.text_len + 1, st, false) ||
                 kg::order_records(pl->d_nl_rec, o1.total_matches, w.global_base + w.text_len + 1, st, true)))
            { // (no room for the sort's scratch: one scan of the whole dictionary, from now on — see scan_ac_split)
                krep_gpu_clear_error();
                pl->ac_split = 1;
            }
            else
            {
                merged = true;
                if (pl->ac_long->anch_state == 1)
                    pl->ac_split = 1;
            }
        }
        if (!merged)
            rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base,
                         pl->d_nl_rec, pl->nl_cap, pl->ww, false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1, true);
        if (rc)
            return rc;
        if (o1.total_matches > dense)
        {
            pl->lines_list_off = true;
            return 1;
        }
        if (!o1.overflow || attempt == 1)
            break;
        if (pl->d_nl_rec) (void)hipFree(pl->d_nl_rec);
        if (pl->d_nl_ln) (void)hipFree(pl->d_nl_ln); // (the other user of the list sizes both)
        pl->d_nl_rec = nullptr;
        pl->d_nl_ln = nullptr;
        pl->nl_cap = 0;
        const uint64_t want = o1.total_matches + o1.total_matches / 8 + 1024;
        HIPCHK(hipMalloc(&pl->d_nl_rec, want * sizeof(match_position_t)));
        pl->nl_cap = want;
    }
    const uint64_t total = o1.total_matches;
    uint64_t lines = o1.line_count; // counted behind the post-pass on the stream, unless some unit overflowed its staging slot
    if (total && lines == ~0ull && tail_count_line_gaps(w.d_text, w.text_len, w.global_base, (const uint64_t *)pl->d_nl_rec, total, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0],
                                      st, &lines))
        return 2;
    if (time_it) HIPCHK(hipEventRecord(pl->ev1, st));
    // the window's line summary: first and last '\n' of the owned bytes, first and last record
    uint64_t first_nl = own_hi, last_nl1 = 0;
    if (tail_find_next_newline(w.d_text, w.own_lo, own_hi, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0], st, &first_nl))
        return 2;
    const bool has_nl = first_nl < own_hi;
    if (has_nl && tail_find_prev_newline(w.d_text, own_hi, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0], st, &last_nl1))
        return 2;
    bool head = total != 0, tail = total != 0;
    if (total && has_nl)
    {
        match_position_t ends[2];
        HIPCHK(hipMemcpyAsync(&ends[0], pl->d_nl_rec, sizeof(match_position_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&ends[1], pl->d_nl_rec + (total - 1), sizeof(match_position_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // a match before the first / behind the last newline exists iff the first / last record of the end-ordered list is one
        // (no match holds a newline, so none can reach across it)
        head = ends[0].start_offset - w.global_base < first_nl;
        tail = ends[1].start_offset - w.global_base >= last_nl1;
    }
    if (time_it)
    {
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
        out->kernel_ms = ms;
    }
    out->total_matches = total;
    out->line_count = lines;
    out->has_newline = has_nl;
    out->head_line_hit = head;
    out->tail_line_hit = tail;
    out->count = std::min<uint64_t>(lines, pl->max_count);
    return 0;
}

static int scan_device_impl(krep_gpu_plan_t *pl, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                            size_t global_base, size_t global_len, match_position_t *d_positions, uint64_t position_capacity,
                            void *stream, int time_it, const krep_gpu_seq_carry_t *carry_in, krep_gpu_seq_carry_t *carry_out,
                            krep_gpu_scan_out_t *out)
{
    krep_gpu_scan_out_t tmp;
    if (!out)
        out = &tmp;
    memset(out, 0, sizeof *out);
    if (carry_out)
        *carry_out = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
    if (!pl || (!d_text && text_len))
        return kg::fail("scan_device: bad arguments");
    if (global_len == 0)
        global_len = global_base + text_len;
    if (global_len < global_base + text_len)
        return kg::fail("scan_device: global_len %zu is shorter than global_base + text_len", global_len);
    {
        // hipGetLastError() is sticky per thread: an earlier, deliberately ignored failure (e.g. a hipFree in a
        // destructor) must not be mistaken for a failure of the launches below
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("KREP_GPU_DEBUG"))
            fprintf(stderr, "krep-gpu: (debug) cleared stale HIP error: %s\n", hipGetErrorString(stale));
    }
    hipStream_t st = (hipStream_t)stream;
    if (pl->unsupported)
        return kg::fail("%s", pl->unsupported);
    if (kg::inject(3))
        return kg::fail("injected failure: kernel launch");
    if (pl->ref_algo == KREP_RA_AHO_CORASICK && pl->lines && pl->ac_has_newline)
    {
        Window w{(const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, global_len};
        return scan_ac_newline_lines(pl, w, st, time_it, carry_in, carry_out, out);
    }
    if (pl->ref_algo == KREP_RA_AHO_CORASICK && pl->lines && text_len >= kAcLinesOnListMin && !getenv("KREP_GPU_AC_LINES_INKERNEL"))
    {
        // (small texts keep the in-kernel road: the list road ends with a few host round trips, ~0.1 ms)
        Window w{(const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, global_len};
        const int rc = scan_ac_lines_on_list(pl, w, st, time_it, out);
        if (rc != 1)
            return rc;
    }
    if (pl->ref_algo == KREP_RA_AHO_CORASICK && !pl->lines)
    {
        const int rc = scan_ac_split(pl, (const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, d_positions, position_capacity, st, time_it, out);
        if (rc != 1)
            return rc;
    }
    if (pl->ref_algo == KREP_RA_AHO_CORASICK)
    {
        const int rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, (const uint8_t *)d_text, text_len, own_lo, own_hi,
                               global_base, d_positions, position_capacity, pl->ww, pl->lines, pl->track, pl->max_count, st, time_it,
                               pl->ev0, pl->ev1, out);
        // the in-kernel road knows the match count too: a text at half the list's break-even density re-opens the list road for
        // the plan's next pieces (the decision was one-way until round 5)
        if (!rc && pl->lines && pl->lines_list_off && out->total_matches < text_len / (ac_counts_lines_in_registers(pl->ac) ? 2000 : 800))
            pl->lines_list_off = false;
        return rc;
    }
    if (pl->sp.num_patterns != 1)
        return kg::fail("scan_device: no pattern");
    const int algo = mirror_effective(pl->ref_algo, &pl->sp, global_len);
    Window w{(const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, global_len};
    return scan_literal(pl, algo, w, d_positions, position_capacity, st, time_it, carry_in, carry_out, out);
}
extern "C" int krep_gpu_scan_device_ex(krep_gpu_plan_t *pl, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                                       size_t global_base, size_t global_len, match_position_t *d_positions,
                                       uint64_t position_capacity, void *stream, int time_it, krep_gpu_scan_out_t *out)
{
    return scan_device_impl(pl, d_text, text_len, own_lo, own_hi, global_base, global_len, d_positions, position_capacity, stream,
                            time_it, nullptr, nullptr, out);
}
// The pieces of one text, scanned in text order: each call takes the boundary record the previous one left (NULL for the
// piece that starts the text) and leaves its own.  For the families without a sequential dependency the record passes through.
extern "C" int krep_gpu_scan_device_seq(krep_gpu_plan_t *pl, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                                        size_t global_base, size_t global_len, match_position_t *d_positions,
                                        uint64_t position_capacity, void *stream, int time_it,
                                        const krep_gpu_seq_carry_t *carry_in, krep_gpu_seq_carry_t *carry_out,
                                        krep_gpu_scan_out_t *out)
{
    krep_gpu_seq_carry_t zero{};
    return scan_device_impl(pl, d_text, text_len, own_lo, own_hi, global_base, global_len, d_positions, position_capacity, stream,
                            time_it, carry_in ? carry_in : &zero, carry_out, out);
}
// The end-of-text replay alone (include/krep_gpu.h): the piece's canonical count and own contribution are in `piece`, the true
// record of the text in front of it in `carry_true`; only the last bytes of the text are read.
extern "C" int krep_gpu_replay_tail(krep_gpu_plan_t *pl, const void *d_tail, size_t tail_len, size_t global_len, void *stream,
                                    const krep_gpu_seq_carry_t *carry_true, const krep_gpu_seq_carry_t *piece,
                                    krep_gpu_seq_carry_t *carry_out, uint64_t *lines)
{
    if (!pl || !d_tail || !piece || !lines)
        return kg::fail("krep_gpu_replay_tail: NULL argument");
    if (pl->unsupported)
        return kg::fail("%s", pl->unsupported);
    if (pl->ref_algo == KREP_RA_AHO_CORASICK || pl->sp.num_patterns != 1)
        return kg::fail("krep_gpu_replay_tail: not a single-literal plan");
    const int algo = mirror_effective(pl->ref_algo, &pl->sp, global_len);
    const Family fam = family_of(algo, pl->only_matching, pl->lines, pl->ww, pl->track, pl->max_count, pl->has_border, pl->m, pl->has_newline);
    if (!fam.replay)
        return kg::fail("krep_gpu_replay_tail: %s under these parameters has no end-of-text replay", krep_gpu_algorithm_name(algo));
    if (!piece->local_lines)
        return kg::fail("krep_gpu_replay_tail: the record is not that of the piece that ends the text");
    const uint64_t G = global_len, need = std::min<uint64_t>(G, 512);
    if (tail_len > G || tail_len < need)
        return kg::fail("krep_gpu_replay_tail: the tail must hold the last %llu bytes of the text", (unsigned long long)need);
    const krep_gpu_seq_carry_t zero{};
    const krep_gpu_seq_carry_t co = kg::fold_carry(carry_true ? *carry_true : zero, *piece);
    uint64_t extra = 0;
    hipStream_t st = (hipStream_t)stream;
    // (the replay reads global offsets >= cur - 1 with cur >= the block in front of the last kReplayWindow bytes: > G - 512)
    if (replay_from_record(pl, algo, co, rebased(d_tail, G - tail_len), G - tail_len ? G - tail_len + 1 : 0, G, st, &extra))
        return 2;
    HIPCHK(hipStreamSynchronize(st));
    if (carry_out)
        *carry_out = co;
    *lines = (piece->local_lines - 1) + extra;
    return 0;
}
extern "C" int krep_gpu_scan_device(krep_gpu_plan_t *pl, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                                    size_t global_base, match_position_t *d_positions, uint64_t position_capacity,
                                    void *stream, int time_it, krep_gpu_scan_out_t *out)
{
    return krep_gpu_scan_device_ex(pl, d_text, text_len, own_lo, own_hi, global_base, 0, d_positions, position_capacity, stream,
                                   time_it, out);
}

extern "C" uint64_t krep_gpu_combine_line_counts(const krep_gpu_scan_out_t *s, int n)
{
    uint64_t total = 0;
    bool open = false; // the line entering the current shard already holds a match
    for (int i = 0; i < n; ++i)
    {
        total += s[i].line_count;
        if (open && s[i].head_line_hit)
            total -= 1;
        open = s[i].has_newline ? (s[i].tail_line_hit != 0) : (open || s[i].head_line_hit != 0);
    }
    return total;
}

