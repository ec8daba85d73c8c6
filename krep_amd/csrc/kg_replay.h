// kg_replay.h — the END of a -c scan through the block-structured reference functions, replayed exactly.
//
// simd_avx2_search (krep.c:4914-5056), simd_avx512_search (:5145-5257) and neon_search (:4534-4650) walk the
// text in 32 / 64 / 16-byte blocks.  In -c mode a newly counted line makes them restart the block grid at the
// next line start (:5000-5013, :5203-5218, :4590-4611), so where the LAST blocks fall — and with them
//   * the block simd_avx512_search steps over unexamined when fewer than (m-1)+64 bytes remain (:5171),
//   * the first byte of the scalar tail call, which has no left neighbour for -w (:5059-5097, :5260-5283, :4653-4690),
//   * neon_search's second count of an unterminated last line by its tail call (fresh last_counted_line_start),
// depends on the line-skip history.  Everywhere else the count is the canonical "distinct lines holding an accepted
// occurrence", which the scan kernel produces.  Only the last kReplayWindow bytes can differ: for an occurrence that
// starts earlier, every block position examines it and counts its line exactly once.  The host therefore
//   1. runs the canonical -c scan over the starts in [0, n - kReplayWindow),
//   2. derives the block-loop position `cur` at which the reference enters that window (last accepted occurrence
//      before it -> next line start -> whole blocks; kg_tail.hip),
//   3. runs replay_lines() below from `cur` — as a one-thread kernel on the device (kg_tail.hip), reading the
//      window straight from HBM.
// The function is __host__ __device__ so that the CPU test-suite can pin the same source against the oracle
// (krep_gpu_debug_replay_host(), tests/test_replay_cpu.py).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KG_HD __host__ __device__
#else
#define KG_HD
#endif

namespace kg {

constexpr uint64_t kReplayWindow = 256; // >= (m-1) + 64 + 63 for m <= 64: any earlier occurrence lies in an examined block

// values of enum krep_ref_algo (include/krep_gpu.h), repeated here to keep the header free-standing
constexpr int kRpAvx2 = 6, kRpAvx512 = 7, kRpNeon = 8;

struct ReplayIn
{
    int algo;            // kRpAvx2 / kRpAvx512 / kRpNeon
    uint32_t m;          // pattern length (AVX2 17..32, AVX-512 33..64, NEON 2..16)
    int ww;              // -w
    uint64_t n;          // text length
    uint64_t cur;        // block-loop position on entry (current_pos - text_start)
    int open;            // NEON: the line holding `cur` is already counted and has no '\n' up to the end of the text
    const uint8_t *text; // text[i] valid for i in [cur - 1 (when cur > 0), n)
    const uint8_t *pat;  // m bytes (these functions are case-sensitive only)
};

KG_HD inline bool rp_wordc(uint8_t c)
{
    return (uint32_t)(c - '0') < 10u || (uint32_t)((c | 0x20u) - 'a') < 26u || c == '_';
}
KG_HD inline bool rp_match(const uint8_t *t, const uint8_t *p, uint32_t m)
{
    for (uint32_t k = 0; k < m; ++k)
        if (t[k] != p[k])
            return false;
    return true;
}
// is_whole_word_match (krep.h:312-319) on the view text[lo, n): no left neighbour at `lo`
KG_HD inline bool rp_ww(const uint8_t *text, uint64_t lo, uint64_t n, uint64_t at, uint32_t m)
{
    if (at > lo && rp_wordc(text[at - 1]))
        return false;
    if (at + m < n && rp_wordc(text[at + m]))
        return false;
    return true;
}
KG_HD inline uint64_t rp_next_nl(const uint8_t *text, uint64_t from, uint64_t n)
{
    while (from < n && text[from] != '\n')
        ++from;
    return from;
}
// boyer_moore_search in -c mode on text[lo, n) AS ITS OWN TEXT (krep.c:1331-1351): it counts a line, jumps to the next
// line start and goes on, so every accepted occurrence it reaches opens a new line
KG_HD inline uint64_t rp_tail_lines(const uint8_t *text, uint64_t lo, uint64_t n, const uint8_t *pat, uint32_t m, int ww)
{
    uint64_t cnt = 0, i = lo;
    while (i + m <= n)
    {
        if (rp_match(text + i, pat, m) && (!ww || rp_ww(text, lo, n, i, m)))
        {
            ++cnt;
            const uint64_t nl = rp_next_nl(text, i, n);
            if (nl >= n)
                break;
            i = nl + 1;
        }
        else
            ++i;
    }
    return cnt;
}

// lines the reference counts from block-loop position r.cur to the end of the text (max_count not applied)
KG_HD inline uint64_t replay_lines(const ReplayIn &r)
{
    const uint32_t B = r.algo == kRpAvx512 ? 64u : r.algo == kRpAvx2 ? 32u : 16u;
    const uint64_t n = r.n;
    const uint32_t m = r.m;
    uint64_t cur = r.cur, cnt = 0;
    bool open = r.open != 0;
    while (n - cur >= B)
    {
        bool restarted = false;
        const bool examined = !(r.algo == kRpAvx512 && n - cur < (uint64_t)(m - 1) + 64u); // krep.c:5171
        for (uint32_t idx = 0; examined && idx < B; ++idx)
        {
            const uint64_t at = cur + idx;
            if (at + m > n)
                break; // AVX2: the zero-padded last-byte lane cannot match (:4945-4951); NEON :4562
            if (!rp_match(r.text + at, r.pat, m))
                continue;
            if (r.ww && !rp_ww(r.text, 0, n, at, m))
                continue;
            if (open)
                continue; // neon_search on the already counted, unterminated last line: same line_start
            ++cnt;
            const uint64_t nl = rp_next_nl(r.text, at, n); // == find_line_end(line_start): no '\n' in [line_start, at)
            if (nl < n)
            {
                cur = nl + 1;
                restarted = true;
                break;
            }
            if (r.algo == kRpNeon)
            {
                open = true; // no skip on an unterminated line (:4590): the block loop simply continues
                continue;
            }
            cur = n; // AVX2 / AVX-512: the advance is clamped to the remaining length (:5006-5008, :5211-5213)
            restarted = true;
            break;
        }
        if (!restarted)
            cur += B;
    }
    if (n - cur >= m)
        cnt += rp_tail_lines(r.text, cur, n, r.pat, m, r.ww);
    return cnt;
}

} // namespace kg
