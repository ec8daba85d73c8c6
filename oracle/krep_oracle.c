/* krep_oracle.c — TEST INFRASTRUCTURE ONLY (see krep_oracle.h).
 *
 * A plain-C restatement of what each function on krep's literal-scan hot path RETURNS — count,
 * match list, emission order and the max_count / -w / -c corner behaviour — written from the
 * reference's observable control flow, one function per reference function.  The product
 * (krep_amd/) never links this file; tests compare the HIP path against it, and it is itself
 * pinned against the compiled reference in oracle/_ref/ and the reference's known-answer vectors.
 *
 * Citations are into /root/reference (krep v2.2.0).
 */
#include "krep_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- globals (krep.c:117-120) */
static int g_only_matching = 0, g_force_no_simd = 0, g_algo_override = KREP_ALGO_AUTO;
void ko_set_only_matching(int on) { g_only_matching = on != 0; }
void ko_set_force_no_simd(int on) { g_force_no_simd = on != 0; }
void ko_set_algo_override(int a) { g_algo_override = a; }

/* ---------------------------------------------------------------- byte classes
 * lower_table is tolower() evaluated in a constructor, i.e. before main() can call setlocale()
 * (krep.c:125-134) => always the "C" locale: only 'A'..'Z' fold.  is_word_char = isalnum || '_'
 * (krep.h:298-301), ASCII under the C locale. */
static inline unsigned char lo(unsigned char c) { return (c >= 'A' && c <= 'Z') ? (unsigned char)(c + 32) : c; }
static inline unsigned char up(unsigned char c) { return (c >= 'a' && c <= 'z') ? (unsigned char)(c - 32) : c; }
static inline bool wordc(unsigned char c)
{
    return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_';
}

/* ---------------------------------------------------------------- result container (krep.c:139-251) */
match_result_t *ko_result_init(uint64_t cap)
{
    match_result_t *r = malloc(sizeof *r);
    if (!r)
        return NULL;
    if (cap == 0)
        cap = 16;
    else if (cap > SIZE_MAX / sizeof(match_position_t))
    {
        free(r);
        return NULL;
    }
    r->positions = malloc(cap * sizeof(match_position_t));
    if (!r->positions)
    {
        free(r);
        return NULL;
    }
    r->count = 0;
    r->capacity = cap;
    return r;
}
bool ko_result_add(match_result_t *r, size_t s, size_t e)
{
    if (!r)
        return false;
    if (r->count >= r->capacity)
    {
        uint64_t ncap = r->capacity ? r->capacity * 2 : 16; /* doubling, krep.c:217 */
        match_position_t *np = realloc(r->positions, ncap * sizeof *np);
        if (!np)
            return false;
        r->positions = np;
        r->capacity = ncap;
    }
    r->positions[r->count].start_offset = s;
    r->positions[r->count].end_offset = e;
    r->count++;
    return true;
}
void ko_result_free(match_result_t *r)
{
    if (!r)
        return;
    free(r->positions);
    free(r);
}

/* ---------------------------------------------------------------- line / word helpers */
size_t ko_line_start(const char *t, size_t n, size_t pos) /* find_line_start, krep.c:363-398 */
{
    if (pos > n)
        pos = n;
    while (pos > 0 && t[pos - 1] != '\n')
        pos--;
    return pos;
}
size_t ko_line_end(const char *t, size_t n, size_t pos) /* find_line_end, krep.c:401-415 */
{
    if (pos >= n)
        return n;
    const char *q = memchr(t + pos, '\n', n - pos);
    return q ? (size_t)(q - t) : n;
}
bool ko_whole_word(const char *t, size_t n, size_t s, size_t e) /* is_whole_word_match, krep.h:312-319 */
{
    if (s > 0 && wordc((unsigned char)t[s - 1]))
        return false;
    if (e < n && wordc((unsigned char)t[e]))
        return false;
    return true;
}
/* faster line_start for the hot loops: identical value, memrchr like the reference */
static inline size_t lstart(const char *t, size_t n, size_t pos)
{
    if (pos > n)
        pos = n;
    if (pos == 0)
        return 0;
    const char *q = memrchr(t, '\n', pos);
    return q ? (size_t)(q - t) + 1 : 0;
}
static inline size_t next_line(const char *t, size_t n, size_t ls)
{
    size_t le = ko_line_end(t, n, ls);
    return le < n ? le + 1 : n;
}
static inline bool eq_at(const unsigned char *t, const unsigned char *p, size_t m, bool cs)
{
    if (cs)
        return memcmp(t, p, m) == 0;
    for (size_t k = 0; k < m; k++) /* memory_equals_case_insensitive, krep.c:1198 */
        if (lo(t[k]) != lo(p[k]))
            return false;
    return true;
}

/* ================================================================ boyer_moore_search, krep.c:1260-1385
 * Horspool skip table exactly as prepare_bad_char_table() (krep.c:1213-1253): default m; for
 * pattern[i], i < m-1, the minimum of m-1-i; under -i both the folded byte and toupper() of the
 * pattern byte get the entry; the table is indexed with the RAW text byte. */
static void horspool_table(const unsigned char *p, size_t m, bool cs, int *sh)
{
    for (int c = 0; c < 256; c++)
        sh[c] = (int)m;
    for (size_t i = 0; i + 1 < m; i++)
    {
        int d = (int)(m - 1 - i);
        if (cs)
        {
            if (d < sh[p[i]])
                sh[p[i]] = d;
        }
        else
        {
            unsigned char l = lo(p[i]), u = up(p[i]);
            if (d < sh[l])
                sh[l] = d;
            if (u != l && d < sh[u])
                sh[u] = d;
        }
    }
}

uint64_t ko_boyer_moore_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions)) /* krep.c:1266 */
        return 0;
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    const bool cs = p->case_sensitive, lines = p->count_lines_mode;
    if (m == 0 || n < m)
        return 0;
    int sh[256];
    horspool_table(pat, m, cs, sh);
    const unsigned char last = cs ? pat[m - 1] : lo(pat[m - 1]);
    uint64_t cnt = 0;
    size_t seen_line = SIZE_MAX, i = 0;
    const size_t lim = n - m + 1;
    while (i < lim)
    {
        const unsigned char tl = t[i + m - 1];
        if ((cs ? tl : lo(tl)) == last && eq_at(t + i, pat, m - 1, cs))
        {
            if (p->whole_word && !ko_whole_word(text, n, i, i + m)) /* :1323-1329 */
            {
                i += sh[tl];
                continue;
            }
            bool bumped = false;
            if (lines) /* :1331-1351 */
            {
                size_t ls = lstart(text, n, i);
                if (ls != seen_line)
                {
                    cnt++;
                    seen_line = ls;
                    bumped = true;
                    if (cnt >= maxc)
                        break;
                    size_t nx = next_line(text, n, ls);
                    if (nx > i)
                    {
                        i = nx;
                        continue;
                    }
                }
            }
            else /* :1353-1364 */
            {
                cnt++;
                bumped = true;
                if (p->track_positions && res && cnt <= maxc)
                    ko_result_add(res, i, i + m);
            }
            if (bumped && cnt >= maxc)
                break;
            i += (g_only_matching && !lines) ? m : (size_t)sh[tl]; /* :1371-1374 */
            continue;
        }
        i += sh[tl];
    }
    return cnt;
}

/* ================================================================ kmp_search, krep.c:1628-1767 */
uint64_t ko_kmp_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->max_count == 0)
        return 0;
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    const bool cs = p->case_sensitive;
    if (m == 0 || n < m)
        return 0;
    int *fail = malloc(m * sizeof(int)); /* compute_lps_array, krep.c:1585-1623 */
    if (!fail)
        return 0;
    fail[0] = 0;
    for (size_t i = 1, k = 0; i < m;)
    {
        unsigned char a = cs ? pat[i] : lo(pat[i]), b = cs ? pat[k] : lo(pat[k]);
        if (a == b)
            fail[i++] = (int)++k;
        else if (k)
            k = (size_t)fail[k - 1];
        else
            fail[i++] = 0;
    }
    uint64_t cnt = 0;
    size_t i = 0, j = 0, seen_line = SIZE_MAX;
    while (i < n)
    {
        unsigned char ct = cs ? t[i] : lo(t[i]), cp = cs ? pat[j] : lo(pat[j]);
        if (cp == ct)
        {
            i++;
            j++;
        }
        if (j == m)
        {
            size_t s = i - j;
            if (p->whole_word && !ko_whole_word(text, n, s, s + m)) /* :1684-1688 */
            {
                j = 0;
                continue;
            }
            if (p->count_lines_mode) /* :1690-1713 */
            {
                size_t ls = lstart(text, n, s);
                if (ls != seen_line)
                {
                    if (maxc != SIZE_MAX && cnt >= maxc)
                        break;
                    cnt++;
                    seen_line = ls;
                    i = next_line(text, n, ls);
                    j = 0;
                    continue;
                }
                j = 0;
            }
            else /* :1714-1743: the (max_count+1)-th match is stored before the break */
            {
                if (maxc != SIZE_MAX && cnt >= maxc)
                {
                    if (p->track_positions && res)
                        ko_result_add(res, s, s + m);
                    break;
                }
                cnt++;
                if (p->track_positions && res)
                    ko_result_add(res, s, s + m);
                i = s + m; /* greedy non-overlapping */
                j = 0;
            }
        }
        else if (i < n && cp != ct)
        {
            if (j)
                j = (size_t)fail[j - 1];
            else
                i++;
        }
    }
    free(fail);
    return cnt;
}

/* ================================================================ memchr_search, krep.c:3891-4041 */
#define KO_MEMCHR_BATCH 4096 /* MEMCHR_BUFFER_SIZE, krep.c:3910 */
uint64_t ko_memchr_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->max_count == 0)
        return 0;
    const unsigned char c0 = (unsigned char)p->pattern[0];
    /* the "other case" byte, krep.c:3904: islower ? toupper : tolower; 0 when case-sensitive */
    const unsigned char c1 = p->case_sensitive ? 0 : ((c0 >= 'a' && c0 <= 'z') ? up(c0) : lo(c0));
    const bool two = !p->case_sensitive && c1 != c0;
    const size_t maxc = p->max_count;
    const bool store = p->track_positions && res;
    match_position_t *batch = malloc(KO_MEMCHR_BATCH * sizeof *batch);
    size_t nb = 0, pos = 0, seen_line = SIZE_MAX;
    uint64_t cnt = 0;
    while (pos < n)
    {
        const char *f = memchr(text + pos, c0, n - pos);
        if (two)
        {
            const char *g = memchr(text + pos, c1, n - pos);
            if (!f || (g && g < f))
                f = g;
        }
        if (!f)
            break;
        size_t at = (size_t)(f - text);
        if (p->whole_word && !ko_whole_word(text, n, at, at + 1))
        {
            pos = at + 1;
            continue;
        }
        if (p->count_lines_mode) /* :3949-3972 */
        {
            size_t ls = lstart(text, n, at);
            if (ls != seen_line)
            {
                if (maxc != SIZE_MAX && cnt >= maxc)
                    break;
                cnt++;
                seen_line = ls;
                pos = next_line(text, n, ls);
            }
            else
                pos = at + 1;
            continue;
        }
        if (maxc != SIZE_MAX && cnt >= maxc) /* :3976-3992: one extra record, batched or direct */
        {
            if (store)
            {
                if (nb < KO_MEMCHR_BATCH)
                {
                    batch[nb].start_offset = at;
                    batch[nb].end_offset = at + 1;
                    nb++;
                }
                else
                    ko_result_add(res, at, at + 1);
            }
            break;
        }
        cnt++;
        if (store) /* :3997-4020 */
        {
            if (nb == KO_MEMCHR_BATCH)
            {
                for (size_t k = 0; k < nb; k++)
                    ko_result_add(res, batch[k].start_offset, batch[k].end_offset);
                nb = 0;
            }
            batch[nb].start_offset = at;
            batch[nb].end_offset = at + 1;
            nb++;
        }
        pos = at + 1;
    }
    if (store && nb) /* final flush capped by max_count, :4026-4038 */
    {
        uint64_t have = res->count;
        uint64_t room = (maxc == SIZE_MAX) ? nb : (have >= maxc ? 0 : maxc - have);
        size_t lim = nb < room ? nb : (size_t)room;
        for (size_t k = 0; k < lim; k++)
            ko_result_add(res, batch[k].start_offset, batch[k].end_offset);
    }
    free(batch);
    return cnt;
}

/* ================================================================ memchr_short_search, krep.c:4371-4503 */
uint64_t ko_memchr_short_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions))
        return 0;
    const size_t m = p->pattern_len, maxc = p->max_count;
    const unsigned char *pat = (const unsigned char *)p->pattern;
    const bool cs = p->case_sensitive;
    if (m < 2 || m > 3 || n < m)
        return 0;
    const unsigned char f0 = cs ? pat[0] : lo(pat[0]);
    size_t cur = 0, rem = n, seen_line = SIZE_MAX;
    uint64_t cnt = 0;
    while (rem >= m)
    {
        /* first-byte scan over the rem-m+1 feasible starts, :4399-4414 */
        size_t span = rem - m + 1, k = span;
        if (cs)
        {
            const char *q = memchr(text + cur, f0, span);
            if (q)
                k = (size_t)(q - (text + cur));
        }
        else
        {
            for (size_t z = 0; z < span; z++)
                if (lo((unsigned char)text[cur + z]) == f0)
                {
                    k = z;
                    break;
                }
        }
        if (k == span)
            break;
        size_t at = cur + k;
        if (eq_at((const unsigned char *)text + at + 1, pat + 1, m - 1, cs))
        {
            if (p->whole_word && !ko_whole_word(text, n, at, at + m)) /* :4441-4446 */
            {
                rem -= k + 1;
                cur = at + 1;
                continue;
            }
            bool bumped = false;
            if (p->count_lines_mode)
            {
                size_t ls = lstart(text, n, at);
                if (ls != seen_line)
                {
                    cnt++;
                    seen_line = ls;
                    bumped = true;
                    if (cnt >= maxc)
                        break;
                    size_t nx = next_line(text, n, ls);
                    if (nx > cur)
                    {
                        cur = nx;
                        rem = n - nx;
                        continue;
                    }
                }
            }
            else
            {
                cnt++;
                bumped = true;
                if (p->track_positions && res && cnt <= maxc)
                    ko_result_add(res, at, at + m);
            }
            if (bumped && cnt >= maxc)
                break;
        }
        /* :4495 — note: with -o this also jumps pattern_len after a FAILED candidate */
        size_t adv = k + (g_only_matching ? m : 1);
        if (adv > rem)
            break;
        cur += adv;
        rem -= adv;
    }
    return cnt;
}

/* ================================================================ simd_sse42_search, krep.c:4702-4869
 * _mm_cmpestri(EQUAL_ORDERED) over 16-byte windows that advance by 16-m+1 on a miss never skips
 * an occurrence, and `index < chunk_len-m+1` (:4761) rejects the partial tail matches the
 * instruction reports, so the window machinery reduces to "leftmost occurrence at or after
 * cur".  What remains is the policy: advance index+m after a hit (greedy non-overlapping,
 * :4839-4848; +1 under -o), pre-increment max_count checks, and the -c line skip. */
static size_t first_occ(const unsigned char *t, size_t n, const unsigned char *pat, size_t m, size_t from)
{
    if (n < m)
        return SIZE_MAX;
    for (size_t i = from; i + m <= n;)
    {
        const unsigned char *q = memchr(t + i, pat[0], n - m + 1 - i);
        if (!q)
            return SIZE_MAX;
        i = (size_t)(q - t);
        if (memcmp(q, pat, m) == 0)
            return i;
        i++;
    }
    return SIZE_MAX;
}
uint64_t ko_sse42_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->pattern_len == 0 || p->pattern_len > 16 || !p->case_sensitive || n < p->pattern_len)
        return ko_boyer_moore_search(p, text, n, res); /* :4708-4712 */
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions))
        return 0;
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    const size_t step = 17 - m; /* a full 16-byte window without a hit advances 16 - m + 1 (:4858) */
    uint64_t cnt = 0;
    size_t cp = 0, seen_line = SIZE_MAX; /* cp = current_pos - text_start */
    while (n - cp >= m)
    {
        size_t at = first_occ(t, n, pat, m, cp);
        if (at == SIZE_MAX)
            break;
        /* The window in which the reference's loop reports `at`: windows start at cp + j*step while >= 16
         * bytes remain; the first window with fewer than 16 bytes left covers the whole rest (:4739-4750).
         * Only the -c skip below depends on it: it adds (line_end + 1 - match) to the WINDOW start instead
         * of to the match (:4787-4793), so the scan resumes `index` bytes before the next line. */
        size_t j = (at - cp) / step;
        if (n - cp < 16)
            j = 0;
        else
        {
            size_t jt = (n - 16 - cp) / step + 1; /* first window with < 16 bytes remaining */
            if (jt < j)
                j = jt;
        }
        const size_t win = cp + j * step;
        if (!p->whole_word || ko_whole_word(text, n, at, at + m))
        {
            bool bumped = false;
            if (p->count_lines_mode) /* :4771-4800 */
            {
                size_t ls = lstart(text, n, at);
                if (ls != seen_line)
                {
                    if (cnt >= maxc)
                        break;
                    cnt++;
                    seen_line = ls;
                    bumped = true;
                    size_t le = ko_line_end(text, n, ls);
                    if (le < n) /* advance = le + 1 - at is always > 0 */
                    {
                        cp = win + (le + 1 - at);
                        continue;
                    }
                }
            }
            else /* :4801-4828 */
            {
                if (cnt >= maxc)
                    break;
                cnt++;
                bumped = true;
                if (p->track_positions && res && cnt <= maxc)
                    ko_result_add(res, at, at + m);
            }
            if (bumped && cnt >= maxc)
                break;
        }
        cp = at + (g_only_matching ? 1 : m); /* window start + index + m, clamped to the end (:4839-4851) */
        if (cp > n)
            cp = n;
    }
    return cnt;
}

/* ================================================================ simd_avx2_search, krep.c:4877-5101
 * <=16 B -> SSE4.2 (:4892); 17..32 B: 32-byte blocks, candidates where first AND last pattern
 * byte match (:4936-4967), verified by memcmp, ALL occurrences; tail (<32 B) -> BMH on the slice
 * with offsets fixed up afterwards (:5059-5097). */
uint64_t ko_avx2_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->pattern_len == 0 || p->pattern_len > 32 || !p->case_sensitive || n < p->pattern_len)
        return ko_boyer_moore_search(p, text, n, res);
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions))
        return 0;
    if (p->pattern_len <= 16)
        return ko_sse42_search(p, text, n, res);
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    uint64_t cnt = 0;
    size_t cur = 0, rem = n, seen_line = SIZE_MAX;
    while (rem >= 32)
    {
        bool skipped = false;
        for (size_t idx = 0; idx < 32; idx++)
        {
            if (t[cur + idx] != pat[0])
                continue;
            /* last-byte lane: bytes past the end read as 0 (zero-padded vector, :4945-4951) */
            size_t lp = cur + idx + m - 1;
            unsigned char lb = lp < n ? t[lp] : 0;
            if (lb != pat[m - 1])
                continue;
            if (cur + idx + m > n) /* the reference memcmp would read out of bounds here (UB) */
                continue;
            if (memcmp(t + cur + idx, pat, m) != 0)
                continue;
            size_t at = cur + idx;
            if (p->whole_word && !ko_whole_word(text, n, at, at + m))
                continue;
            bool bumped = false;
            if (p->count_lines_mode) /* :4989-5018 */
            {
                size_t ls = lstart(text, n, at);
                if (ls != seen_line)
                {
                    cnt++;
                    seen_line = ls;
                    bumped = true;
                    if (cnt >= maxc)
                        return cnt;
                    size_t nx = next_line(text, n, ls);
                    if (nx > cur)
                    {
                        size_t adv = nx - cur;
                        if (adv > rem)
                            adv = rem;
                        cur += adv;
                        rem -= adv;
                        skipped = true;
                        break;
                    }
                }
            }
            else
            {
                cnt++;
                bumped = true;
                if (p->track_positions && res && cnt <= maxc)
                    ko_result_add(res, at, at + m);
            }
            if (bumped && cnt >= maxc)
                return cnt;
        }
        if (skipped)
            continue;
        cur += 32;
        rem -= 32;
    }
    if (rem >= m) /* tail, :5059-5097 */
    {
        search_params_t tp = *p;
        if (maxc != SIZE_MAX)
            tp.max_count = cnt >= maxc ? 0 : maxc - (size_t)cnt;
        uint64_t tc = ko_boyer_moore_search(&tp, text + cur, rem, res);
        if (res && p->track_positions && tc > 0)
        {
            uint64_t from = cnt; /* "assuming BM added sequentially", :5077 */
            if (from > res->count)
                from = res->count;
            for (uint64_t k = from; k < res->count; k++)
            {
                res->positions[k].start_offset += cur;
                res->positions[k].end_offset += cur;
            }
        }
        cnt += tc;
        if (maxc != SIZE_MAX && cnt > maxc)
            cnt = maxc;
    }
    return cnt;
}

/* ================================================================ simd_avx512_search, krep.c:5108-5286
 * <=32 B -> AVX2 (:5123); 33..64 B: 64-byte blocks; a block is examined only when
 * remaining >= (m-1)+64 (:5171) — a block with 64 <= remaining < m-1+64 is stepped over
 * unexamined (reference bug, reproduced); tail (<64 B) -> simd_avx2_search, which for m > 32 is BMH. */
uint64_t ko_avx512_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->pattern_len == 0 || p->pattern_len > 64 || !p->case_sensitive || n < p->pattern_len)
        return ko_avx2_search(p, text, n, res);
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions))
        return 0;
    if (p->pattern_len <= 32)
        return ko_avx2_search(p, text, n, res);
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    uint64_t cnt = 0;
    size_t cur = 0, rem = n, seen_line = SIZE_MAX;
    while (rem >= 64)
    {
        bool skipped = false;
        if (rem >= (m - 1) + 64)
        {
            for (size_t idx = 0; idx < 64; idx++)
            {
                size_t at = cur + idx;
                if (t[at] != pat[0] || t[at + m - 1] != pat[m - 1] || memcmp(t + at, pat, m) != 0)
                    continue;
                if (p->whole_word && !ko_whole_word(text, n, at, at + m))
                    continue;
                bool bumped = false;
                if (p->count_lines_mode)
                {
                    size_t ls = lstart(text, n, at);
                    if (ls != seen_line)
                    {
                        cnt++;
                        seen_line = ls;
                        bumped = true;
                        if (cnt >= maxc)
                            return cnt;
                        size_t nx = next_line(text, n, ls);
                        if (nx > cur)
                        {
                            size_t adv = nx - cur;
                            if (adv > rem)
                                adv = rem;
                            cur += adv;
                            rem -= adv;
                            skipped = true;
                            break;
                        }
                    }
                }
                else
                {
                    cnt++;
                    bumped = true;
                    if (p->track_positions && res && cnt <= maxc)
                        ko_result_add(res, at, at + m);
                }
                if (bumped && cnt >= maxc)
                    return cnt;
            }
            if (skipped)
                continue;
        }
        cur += 64;
        rem -= 64;
    }
    if (rem >= m) /* :5260-5283 */
    {
        search_params_t tp = *p;
        if (maxc != SIZE_MAX)
            tp.max_count = cnt >= maxc ? 0 : maxc - (size_t)cnt;
        uint64_t tc = ko_avx2_search(&tp, text + cur, rem, res);
        if (res && p->track_positions && tc > 0)
        {
            uint64_t from = res->count >= tc ? res->count - tc : 0;
            for (uint64_t k = 0; k < tc && from + k < res->count; k++)
            {
                res->positions[from + k].start_offset += cur;
                res->positions[from + k].end_offset += cur;
            }
        }
        cnt += tc;
    }
    return cnt;
}

/* ================================================================ neon_search, krep.c:4506-4694
 * arm64 builds (krep.c:69-74); selected for case-sensitive patterns of 2..16 bytes (krep.c:1821, :1852).
 * 16-byte blocks: every offset whose byte equals pattern[0] and that leaves >= m bytes (:4562) is
 * verified with memcmp — ALL occurrences, any overlap.  max_count is tested BEFORE the increment
 * (:4582, :4616).  -c: after a newly counted line the block loop restarts at the next line start, but
 * only when the line is terminated (:4590-4611); on an unterminated last line the loop simply goes on
 * (same line => nothing more is counted).  Tail (< 16 B) -> BMH on the slice as its own text (:4653-4690):
 * no left neighbour for -w at its first byte, a fresh last-counted-line. */
uint64_t ko_neon_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (p->pattern_len == 0 || !p->case_sensitive || n < p->pattern_len)
        return ko_boyer_moore_search(p, text, n, res);
    if (p->max_count == 0 && (p->count_lines_mode || p->track_positions))
        return 0;
    const unsigned char *t = (const unsigned char *)text, *pat = (const unsigned char *)p->pattern;
    const size_t m = p->pattern_len, maxc = p->max_count;
    uint64_t cnt = 0;
    size_t cur = 0, rem = n, seen_line = SIZE_MAX;
    while (rem >= 16)
    {
        bool restarted = false;
        for (size_t idx = 0; idx < 16 && !restarted; idx++)
        {
            if (t[cur + idx] != pat[0] || rem - idx < m)
                continue;
            if (memcmp(t + cur + idx, pat, m) != 0)
                continue;
            const size_t at = cur + idx;
            if (p->whole_word && !ko_whole_word(text, n, at, at + m))
                continue;
            bool bumped = false;
            if (p->count_lines_mode)
            {
                size_t ls = lstart(text, n, at);
                if (ls != seen_line)
                {
                    if (cnt >= maxc)
                        return cnt;
                    cnt++;
                    seen_line = ls;
                    bumped = true;
                    size_t le = ko_line_end(text, n, ls);
                    if (le < n && le + 1 > cur)
                    {
                        size_t adv = le + 1 - cur;
                        if (adv > rem)
                            adv = rem;
                        cur += adv;
                        rem -= adv;
                        restarted = true; /* goto next_chunk: the max_count test below is skipped, :4609 */
                        continue;
                    }
                }
            }
            else
            {
                if (cnt >= maxc)
                    return cnt;
                cnt++;
                bumped = true;
                if (p->track_positions && res && cnt <= maxc)
                    ko_result_add(res, at, at + m);
            }
            if (bumped && cnt >= maxc)
                return cnt;
        }
        if (restarted)
            continue;
        cur += 16;
        rem -= 16;
    }
    if (rem >= m)
    {
        search_params_t tp = *p;
        if (maxc != SIZE_MAX)
            tp.max_count = cnt >= maxc ? 0 : maxc - (size_t)cnt;
        uint64_t tc = ko_boyer_moore_search(&tp, text + cur, rem, res);
        if (res && p->track_positions && tc > 0 && res->count >= tc) /* :4676-4685 */
        {
            uint64_t from = res->count - tc;
            for (uint64_t k = 0; k < tc; k++)
            {
                res->positions[from + k].start_offset += cur;
                res->positions[from + k].end_offset += cur;
            }
        }
        cnt += tc;
    }
    return cnt;
}

/* ================================================================ Aho-Corasick, aho_corasick.c
 * Array automaton instead of 2 KiB pointer nodes: edges in an open-addressing map keyed by
 * (state << 8 | byte); per-state fail link and output list (pattern indices in insertion order,
 * aho_corasick.c:180).  No dictionary-suffix links, like the reference: the search walks the
 * whole fail chain after every byte (aho_corasick.c:353-431). */
struct ac_trie
{
    uint32_t nstates, cap_states;
    uint32_t *fail;
    uint32_t *out_head; /* index into out_pat/out_next, or UINT32_MAX */
    uint32_t *out_tail;
    uint32_t *out_pat, *out_next;
    uint32_t nouts, cap_outs;
    uint64_t *ekey; /* edge map */
    uint32_t *eval;
    uint64_t emask;
    uint32_t nedges;
    bool cs;
};
static uint32_t edge_get(const struct ac_trie *a, uint32_t s, unsigned char c)
{
    uint64_t key = ((uint64_t)s << 8) | c, h = (key * 0x9E3779B97F4A7C15ull) >> 20;
    for (;; h++)
    {
        uint64_t k = a->ekey[h & a->emask];
        if (k == UINT64_MAX)
            return UINT32_MAX;
        if (k == key)
            return a->eval[h & a->emask];
    }
}
static void edge_put_raw(uint64_t *ek, uint32_t *ev, uint64_t mask, uint64_t key, uint32_t v)
{
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 20;
    while (ek[h & mask] != UINT64_MAX)
        h++;
    ek[h & mask] = key;
    ev[h & mask] = v;
}
static void edge_put(struct ac_trie *a, uint32_t s, unsigned char c, uint32_t v)
{
    if ((uint64_t)(a->nedges + 1) * 2 > a->emask + 1)
    {
        uint64_t nm = (a->emask + 1) * 2 - 1;
        uint64_t *nk = malloc((nm + 1) * sizeof *nk);
        uint32_t *nv = malloc((nm + 1) * sizeof *nv);
        memset(nk, 0xff, (nm + 1) * sizeof *nk);
        for (uint64_t i = 0; i <= a->emask; i++)
            if (a->ekey[i] != UINT64_MAX)
                edge_put_raw(nk, nv, nm, a->ekey[i], a->eval[i]);
        free(a->ekey);
        free(a->eval);
        a->ekey = nk;
        a->eval = nv;
        a->emask = nm;
    }
    edge_put_raw(a->ekey, a->eval, a->emask, ((uint64_t)s << 8) | c, v);
    a->nedges++;
}
static uint32_t new_state(struct ac_trie *a)
{
    if (a->nstates == a->cap_states)
    {
        a->cap_states *= 2;
        a->fail = realloc(a->fail, a->cap_states * sizeof(uint32_t));
        a->out_head = realloc(a->out_head, a->cap_states * sizeof(uint32_t));
        a->out_tail = realloc(a->out_tail, a->cap_states * sizeof(uint32_t));
    }
    a->fail[a->nstates] = 0;
    a->out_head[a->nstates] = a->out_tail[a->nstates] = UINT32_MAX;
    return a->nstates++;
}
static void add_output(struct ac_trie *a, uint32_t s, uint32_t pat)
{
    if (a->nouts == a->cap_outs)
    {
        a->cap_outs *= 2;
        a->out_pat = realloc(a->out_pat, a->cap_outs * sizeof(uint32_t));
        a->out_next = realloc(a->out_next, a->cap_outs * sizeof(uint32_t));
    }
    uint32_t id = a->nouts++;
    a->out_pat[id] = pat;
    a->out_next[id] = UINT32_MAX;
    if (a->out_head[s] == UINT32_MAX)
        a->out_head[s] = id;
    else
        a->out_next[a->out_tail[s]] = id;
    a->out_tail[s] = id;
}

ac_trie_t *ko_ac_trie_build(const search_params_t *p) /* ac_trie_build, aho_corasick.c:111-271 */
{
    if (!p || p->num_patterns == 0)
        return NULL;
    struct ac_trie *a = calloc(1, sizeof *a);
    a->cap_states = 64;
    a->fail = malloc(64 * sizeof(uint32_t));
    a->out_head = malloc(64 * sizeof(uint32_t));
    a->out_tail = malloc(64 * sizeof(uint32_t));
    a->cap_outs = 64;
    a->out_pat = malloc(64 * sizeof(uint32_t));
    a->out_next = malloc(64 * sizeof(uint32_t));
    a->emask = 1023;
    a->ekey = malloc(1024 * sizeof(uint64_t));
    a->eval = malloc(1024 * sizeof(uint32_t));
    memset(a->ekey, 0xff, 1024 * sizeof(uint64_t));
    a->cs = p->case_sensitive;
    new_state(a); /* root = 0, fails to itself */
    for (size_t k = 0; k < p->num_patterns; k++)
    {
        const unsigned char *s = (const unsigned char *)p->patterns[k];
        size_t len = p->pattern_lens[k];
        uint32_t st = 0;
        for (size_t i = 0; i < len; i++) /* len == 0 => output on the root, :144-152 */
        {
            unsigned char c = a->cs ? s[i] : lo(s[i]);
            uint32_t nx = edge_get(a, st, c);
            if (nx == UINT32_MAX)
            {
                nx = new_state(a);
                edge_put(a, st, c, nx);
            }
            st = nx;
        }
        add_output(a, st, (uint32_t)k);
    }
    /* BFS fail links, :228-267 */
    uint32_t *q = malloc(a->nstates * sizeof(uint32_t));
    uint32_t qh = 0, qt = 0;
    for (int c = 0; c < 256; c++)
    {
        uint32_t ch = edge_get(a, 0, (unsigned char)c);
        if (ch != UINT32_MAX)
        {
            a->fail[ch] = 0;
            q[qt++] = ch;
        }
    }
    while (qh < qt)
    {
        uint32_t cur = q[qh++];
        for (int c = 0; c < 256; c++)
        {
            uint32_t ch = edge_get(a, cur, (unsigned char)c);
            if (ch == UINT32_MAX)
                continue;
            q[qt++] = ch;
            uint32_t f = a->fail[cur];
            while (f != 0 && edge_get(a, f, (unsigned char)c) == UINT32_MAX)
                f = a->fail[f];
            uint32_t g = edge_get(a, f, (unsigned char)c);
            a->fail[ch] = (g != UINT32_MAX) ? g : 0;
        }
    }
    free(q);
    return (ac_trie_t *)a;
}
void ko_ac_trie_free(ac_trie_t *t)
{
    struct ac_trie *a = (struct ac_trie *)t;
    if (!a)
        return;
    free(a->fail);
    free(a->out_head);
    free(a->out_tail);
    free(a->out_pat);
    free(a->out_next);
    free(a->ekey);
    free(a->eval);
    free(a);
}
uint64_t ko_ac_num_states(const ac_trie_t *t) { return t ? ((const struct ac_trie *)t)->nstates : 0; }

uint64_t ko_aho_corasick_search(const search_params_t *p, const char *text, size_t n, match_result_t *res)
{
    if (!p || !p->ac_trie || !text) /* aho_corasick.c:306 */
        return 0;
    if (p->max_count == 0)
        return 0;
    const struct ac_trie *a = (const struct ac_trie *)p->ac_trie;
    const size_t maxc = p->max_count;
    uint64_t found = 0;
    size_t seen_line = SIZE_MAX;
    uint32_t st = 0;
    for (size_t i = 0; i < n; i++)
    {
        unsigned char c = p->case_sensitive ? (unsigned char)text[i] : lo((unsigned char)text[i]);
        uint32_t nx;
        while ((nx = edge_get(a, st, c)) == UINT32_MAX && st != 0) /* goto/fail, :338-349 */
            st = a->fail[st];
        if (nx != UINT32_MAX)
            st = nx;
        /* emit along the whole fail chain, longest first, :353-431 */
        for (uint32_t o = st; o != 0; o = a->fail[o])
        {
            for (uint32_t e = a->out_head[o]; e != UINT32_MAX; e = a->out_next[e])
            {
                if (found >= maxc)
                    return found;
                size_t len = p->pattern_lens[a->out_pat[e]];
                if (len == 0)
                    continue;
                size_t s = i + 1 - len, en = i + 1;
                if (p->whole_word && !ko_whole_word(text, n, s, en))
                    continue;
                if (p->count_lines_mode) /* dedupe on the last counted line only, no skip, :383-403 */
                {
                    size_t ls = lstart(text, n, s);
                    if (ls != seen_line)
                    {
                        found++;
                        seen_line = ls;
                        if (found >= maxc)
                            return found;
                    }
                }
                else
                {
                    found++;
                    if (p->track_positions && res)
                        ko_result_add(res, s, en);
                    if (found >= maxc)
                        return found;
                }
            }
            if (found >= maxc)
                return found;
        }
    }
    if (n == 0 && a->out_head[0] != UINT32_MAX) /* empty pattern on empty text, :441-463 */
    {
        for (uint32_t e = a->out_head[0]; e != UINT32_MAX; e = a->out_next[e])
            if (p->pattern_lens[a->out_pat[e]] == 0)
            {
                if (found < maxc)
                {
                    found++;
                    if (p->track_positions && res)
                        ko_result_add(res, 0, 0);
                }
                break;
            }
    }
    return found;
}

/* ================================================================ select_search_algorithm, krep.c:1771-1914 */
static bool repetitive(const char *s, size_t m) /* is_repetitive_pattern, krep.c:1873-1914 */
{
    if (m < 3)
        return false;
    size_t run = 0;
    char prev = s[0];
    for (size_t i = 1; i < m; i++)
    {
        if (s[i] == prev)
        {
            if (++run >= m / 2)
                return true;
        }
        else
        {
            run = 0;
            prev = s[i];
        }
    }
    for (size_t per = 2; per <= m / 2; per++)
    {
        bool ok = true;
        for (size_t i = per; i < m && ok; i++)
            ok = s[i] == s[i % per];
        if (ok)
            return true;
    }
    return false;
}
int ko_select(const search_params_t *p, int simd)
{
    if (p->use_regex)
        return KREP_RA_REGEX;
    if (p->num_patterns > 1)
        return KREP_RA_AHO_CORASICK;
    if (g_algo_override == KREP_ALGO_BM)
        return KREP_RA_BMH;
    if (g_algo_override == KREP_ALGO_KMP)
        return KREP_RA_KMP;
    const size_t simd_max = simd == KREP_REF_AVX512 ? 64 : simd == KREP_REF_AVX2 ? 32
                          : (simd == KREP_REF_SSE42 || simd == KREP_REF_NEON)    ? 16
                                                                                 : 0; /* :101-113 */
    const int top = simd == KREP_REF_AVX512 ? KREP_RA_AVX512 : simd == KREP_REF_AVX2 ? KREP_RA_AVX2
                  : simd == KREP_REF_SSE42                                           ? KREP_RA_SSE42
                  : simd == KREP_REF_NEON                                            ? KREP_RA_NEON
                                                                                     : KREP_RA_NONE;
    const size_t m = p->pattern_len;
    bool can = !g_force_no_simd && simd_max > 0 && m <= simd_max;
    if (m == 1)
        return KREP_RA_MEMCHR;
    if (m < 4)
        return (can && p->case_sensitive && top != KREP_RA_NONE) ? top : KREP_RA_MEMCHR_SHORT;
    if (can)
    {
        if (simd == KREP_REF_AVX512 && m <= 64 && p->case_sensitive)
            return KREP_RA_AVX512;
        if ((simd == KREP_REF_AVX512 || simd == KREP_REF_AVX2) && m <= 32)
            return KREP_RA_AVX2; /* also under -i: falls to BMH inside, :4883 */
        if (simd == KREP_REF_SSE42 && m <= 16 && p->case_sensitive)
            return KREP_RA_SSE42;
        if (simd == KREP_REF_NEON && p->case_sensitive)
            return KREP_RA_NEON;
    }
    if (m < 8 && repetitive(p->pattern, m))
        return KREP_RA_KMP;
    return KREP_RA_BMH;
}
uint64_t ko_run(int algo, const search_params_t *p, const char *text, size_t n, match_result_t *r)
{
    switch (algo)
    {
    case KREP_RA_BMH: return ko_boyer_moore_search(p, text, n, r);
    case KREP_RA_KMP: return ko_kmp_search(p, text, n, r);
    case KREP_RA_MEMCHR: return ko_memchr_search(p, text, n, r);
    case KREP_RA_MEMCHR_SHORT: return ko_memchr_short_search(p, text, n, r);
    case KREP_RA_SSE42: return ko_sse42_search(p, text, n, r);
    case KREP_RA_AVX2: return ko_avx2_search(p, text, n, r);
    case KREP_RA_AVX512: return ko_avx512_search(p, text, n, r);
    case KREP_RA_NEON: return ko_neon_search(p, text, n, r);
    case KREP_RA_AHO_CORASICK: return ko_aho_corasick_search(p, text, n, r);
    default: return 0;
    }
}
uint64_t ko_search(const search_params_t *p, const char *text, size_t n, match_result_t *r, int simd)
{
    return ko_run(ko_select(p, simd), p, text, n, r);
}
