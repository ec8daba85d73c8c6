// kg_mirror.hip — the twin of select_search_algorithm() (krep.c:1771-1914): which reference function a given build would end
// up running for given parameters (hence which match-set family and which return-value quirks are reproduced), what the
// backend does not take (kg::unsupported_reason), and the match_result_t container (krep.c:139-251 contract).  Host logic only.
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

using namespace kg;


// The predicate behind "KMP instead of BMH" on builds without a usable SIMD body (is_repetitive_pattern(), krep.c:1873-1914),
// stated as the PROPERTY it computes: a pattern of m >= 3 bytes is repetitive iff it holds a run of more than m/2 equal bytes,
// or it has a period p with 2 <= p <= m/2 (s[i] == s[i - p] for every i >= p).  The periods of a string are exactly m - b over
// its borders b, so they are read off the border chain of the prefix function instead of being tried one by one.
static bool repetitive_pattern(const char *s, size_t m)
{
    if (m < 3)
        return false;
    size_t longest = 1;
    for (size_t lo = 0; lo < m;)
    {
        size_t hi = lo + 1;
        while (hi < m && s[hi] == s[lo])
            ++hi;
        longest = std::max(longest, hi - lo);
        lo = hi;
    }
    if (longest > m / 2)
        return true;
    std::vector<size_t> border(m + 1, 0); // border[k]: longest proper border of s[0, k)
    for (size_t k = 1, b = 0; k < m; ++k)
    {
        while (b && s[k] != s[b])
            b = border[b];
        b += s[k] == s[b];
        border[k + 1] = b;
    }
    for (size_t b = border[m]; b; b = border[b])
    {
        const size_t per = m - b;
        if (per >= 2 && per <= m / 2)
            return true;
    }
    return false;
}

namespace kg {
// The function pointer select_search_algorithm() would return (krep.c:1771-1870) ...
int mirror_top(const search_params_t *p, const krep_gpu_config_t &c)
{
    if (p->use_regex)
        return KREP_RA_REGEX;
    if (p->num_patterns > 1)
        return KREP_RA_AHO_CORASICK;
    if (c.algo_override == KREP_ALGO_BM)
        return KREP_RA_BMH;
    if (c.algo_override == KREP_ALGO_KMP)
        return KREP_RA_KMP;
    const int simd = c.reference_simd;
    const size_t simd_max = simd == KREP_REF_AVX512 ? 64 : simd == KREP_REF_AVX2 ? 32
                          : (simd == KREP_REF_SSE42 || simd == KREP_REF_NEON)    ? 16 : 0; // krep.c:101-113
    const int top = simd == KREP_REF_AVX512 ? KREP_RA_AVX512 : simd == KREP_REF_AVX2 ? KREP_RA_AVX2
                  : simd == KREP_REF_SSE42 ? KREP_RA_SSE42 : simd == KREP_REF_NEON ? KREP_RA_NEON : KREP_RA_NONE;
    const size_t m = p->pattern_len;
    const bool can = !c.force_no_simd && simd_max > 0 && m <= simd_max;
    if (m == 1)
        return KREP_RA_MEMCHR;
    if (m < 4)
        return (can && p->case_sensitive && top != KREP_RA_NONE) ? top : KREP_RA_MEMCHR_SHORT;
    if (can)
    {
        if (simd == KREP_REF_AVX512 && m <= 64 && p->case_sensitive)
            return KREP_RA_AVX512;
        if ((simd == KREP_REF_AVX512 || simd == KREP_REF_AVX2) && m <= 32)
            return KREP_RA_AVX2;
        if (simd == KREP_REF_SSE42 && m <= 16 && p->case_sensitive)
            return KREP_RA_SSE42;
        if (simd == KREP_REF_NEON && p->case_sensitive)
            return KREP_RA_NEON;
    }
    if (m < 8 && repetitive_pattern(p->pattern, m))
        return KREP_RA_KMP;
    return KREP_RA_BMH;
}
// ... and the function that ends up doing the work after the internal delegation chain
// (krep.c:4512-4515, :4708-4712, :4883-4896, :5114-5126).
int mirror_effective(int top, const search_params_t *p, size_t text_len)
{
    const size_t m = p->pattern_len;
    int a = top;
    if (a == KREP_RA_AVX512)
    {
        if (m == 0 || m > 64 || !p->case_sensitive || text_len < m || m <= 32)
            a = KREP_RA_AVX2;
    }
    if (a == KREP_RA_AVX2)
    {
        if (m == 0 || m > 32 || !p->case_sensitive || text_len < m)
            a = KREP_RA_BMH;
        else if (m <= 16)
            a = KREP_RA_SSE42;
    }
    if (a == KREP_RA_SSE42)
    {
        if (m == 0 || m > 16 || !p->case_sensitive || text_len < m)
            a = KREP_RA_BMH;
    }
    if (a == KREP_RA_NEON && (!p->case_sensitive || m == 0 || text_len < m))
        a = KREP_RA_BMH;
    return a;
}
} // namespace kg
extern "C" int krep_gpu_mirror_select(const search_params_t *p, size_t text_len)
{
    if (!p)
        return KREP_RA_NONE;
    return mirror_effective(mirror_top(p, kg::current_config()), p, text_len);
}
extern "C" const char *krep_gpu_algorithm_name(int a)
{
    switch (a) // get_algorithm_name(), krep.c:1964-1996
    {
    case KREP_RA_BMH: return "Boyer-Moore-Horspool";
    case KREP_RA_KMP: return "Knuth-Morris-Pratt";
    case KREP_RA_REGEX: return "Regex";
    case KREP_RA_AHO_CORASICK: return "Aho-Corasick";
    case KREP_RA_MEMCHR: return "memchr";
    case KREP_RA_MEMCHR_SHORT: return "memchr-short";
    case KREP_RA_SSE42: return "SSE4.2";
    case KREP_RA_AVX2: return "AVX2";
    case KREP_RA_AVX512: return "AVX-512";
    case KREP_RA_NEON: return "NEON";
    default: return "Unknown";
    }
}

// ------------------------------------------------------------------------------------ what is accelerated
// ONE input class is not taken; for it krep_gpu_can_accelerate() says 0, krep_gpu_select_search_algorithm() returns NULL (the
// caller keeps its CPU function pointer, exactly like the regex case) and an operator called with it anyway takes the failure road:
//  * memchr_short_search in -c mode while the file-static only_matching is set: main() never produces that
//    combination (krep.c:3811-3814 clears count_lines_mode under -o), so it has no reference behaviour to pin.
// (Round 3: -c through simd_sse42_search / kmp_search with a '\n' inside the pattern — refused until then — is reproduced by a
//  walk over the ordered occurrence list, kg_greedy.hip (3).)
namespace kg {
const char *unsupported_reason(const search_params_t *p, const krep_gpu_config_t &c)
{
    if (!p)
        return "NULL params";
    if (p->use_regex)
        return "regex search is not part of the accelerated path (keep krep's regex_search)";
    if (p->num_patterns > 1)
        return (p->patterns && p->pattern_lens) ? nullptr : "several patterns announced but patterns / pattern_lens are NULL";
    if (!p->pattern && !(p->num_patterns == 1 && p->patterns && p->pattern_lens && p->patterns[0]))
        return "no pattern";
    search_params_t q = *p; // legacy callers fill only pattern / pattern_len (test/test_krep.c:233-235); others only the arrays
    if (p->num_patterns == 1 && p->patterns && p->pattern_lens && p->patterns[0])
    {
        q.pattern = p->patterns[0];
        q.pattern_len = p->pattern_lens[0];
    }
    p = &q;
    // (round 5: count_lines_mode AND only_matching through memchr_short_search — a combination krep's main() never produces,
    //  krep.c:3811-3814 — is taken too: kg_greedy.hip, kWalkShortOLines; nothing on this path is refused any more)
    return nullptr;
}
} // namespace kg
extern "C" int krep_gpu_can_accelerate(const search_params_t *p)
{
    const krep_gpu_config_t c = kg::current_config();
    return kg::unsupported_reason(p, c) == nullptr && kg::device_unusable(c.device) == nullptr ? 1 : 0;
}
// (krep_gpu_worthwhile: kg_cost.hip)

// ------------------------------------------------------------------------------------ result container (krep.c:139-251 contract)
extern "C" match_result_t *krep_gpu_match_result_init(uint64_t cap)
{
    match_result_t *r = (match_result_t *)malloc(sizeof *r);
    if (!r)
        return nullptr;
    if (cap == 0)
        cap = 16;
    if (cap > SIZE_MAX / sizeof(match_position_t))
    {
        free(r);
        return nullptr;
    }
    r->positions = (match_position_t *)malloc(cap * sizeof(match_position_t));
    if (!r->positions)
    {
        free(r);
        return nullptr;
    }
    r->count = 0;
    r->capacity = cap;
    return r;
}
extern "C" void krep_gpu_match_result_free(match_result_t *r)
{
    if (!r)
        return;
    free(r->positions);
    free(r);
}
namespace kg {
// make room for `extra` more records (malloc family, so the reference's match_result_free works)
bool result_reserve(match_result_t *r, uint64_t extra)
{
    const uint64_t need = r->count + extra;
    if (need <= r->capacity && r->positions)
        return true;
    uint64_t cap = r->capacity ? r->capacity : 16;
    while (cap < need)
        cap *= 2; // same doubling policy as match_result_add (krep.c:217)
    match_position_t *np = (match_position_t *)realloc(r->positions, cap * sizeof(match_position_t));
    if (!np)
        return false;
    r->positions = np;
    r->capacity = cap;
    return true;
}
} // namespace kg

