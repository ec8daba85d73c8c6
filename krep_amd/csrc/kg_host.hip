// kg_host.hip — C-ABI of the krep-gpu backend (include/krep_gpu.h): plans, device scans, the
// search_func_t operators, search_buffer(), the reference-algorithm mirror and the generators.
// Host logic only; the scan kernels live in kg_literal.hip / kg_ac.hip / kg_post.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_synth.h"
#include "kg_internal.h"

using namespace kg;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
namespace kg {
int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    fprintf(stderr, "krep-gpu: %s\n", buf);
    return 2;
}
} // namespace kg
#define HIPCHK(x)                                                                             \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *krep_gpu_last_error(void) { return g_err.c_str(); }
extern "C" void krep_gpu_clear_error(void) { g_err.clear(); }
extern "C" const char *krep_gpu_version(void) { return "krep-gpu 0.1 (gfx950)"; }
extern "C" int krep_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// ------------------------------------------------------------------------------------ mirror of the reference's globals
static int g_simd = KREP_REF_AVX2, g_only_matching = 0, g_no_simd = 0, g_algo_override = KREP_ALGO_AUTO;
extern "C" void krep_gpu_set_reference_simd(int l) { g_simd = l; }
extern "C" int krep_gpu_get_reference_simd(void) { return g_simd; }
extern "C" void krep_gpu_set_only_matching(int on) { g_only_matching = on != 0; }
static int g_result_order = 0;
extern "C" void krep_gpu_set_result_order(int by_start) { g_result_order = by_start != 0; }
namespace kg { int current_only_matching() { return g_only_matching; } }
namespace kg { int current_result_order() { return g_result_order; } }
extern "C" void krep_gpu_set_force_no_simd(int on) { g_no_simd = on != 0; }
extern "C" void krep_gpu_set_algo_override(int a) { g_algo_override = a; }
static int g_force_rounds = 0; // test hook: 0 = auto, 1 / 4 = force the tile shape
static int g_force_stage_cap = 0; // test hook: staging records per unit (0 = auto)
namespace kg { extern int g_ac_force_stage_cap; }
extern "C" void krep_gpu_debug_force_stage_cap(int c) { g_force_stage_cap = c; kg::g_ac_force_stage_cap = c; }
extern "C" void krep_gpu_debug_force_rounds(int r) { g_force_rounds = r; }

static inline uint8_t lo8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// is_repetitive_pattern(), krep.c:1873-1914 (decides KMP vs BMH on builds without SIMD)
static bool repetitive_pattern(const char *s, size_t m)
{
    if (m < 3)
        return false;
    size_t run = 0;
    char prev = s[0];
    for (size_t i = 1; i < m; ++i)
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
    for (size_t per = 2; per <= m / 2; ++per)
    {
        bool ok = true;
        for (size_t i = per; i < m && ok; ++i)
            ok = s[i] == s[i % per];
        if (ok)
            return true;
    }
    return false;
}

// The function pointer select_search_algorithm() would return (krep.c:1771-1870) ...
static int mirror_top(const search_params_t *p)
{
    if (p->use_regex)
        return KREP_RA_REGEX;
    if (p->num_patterns > 1)
        return KREP_RA_AHO_CORASICK;
    if (g_algo_override == KREP_ALGO_BM)
        return KREP_RA_BMH;
    if (g_algo_override == KREP_ALGO_KMP)
        return KREP_RA_KMP;
    const size_t simd_max = g_simd == KREP_REF_AVX512 ? 64 : g_simd == KREP_REF_AVX2 ? 32
                          : (g_simd == KREP_REF_SSE42 || g_simd == KREP_REF_NEON)    ? 16 : 0;
    const int top = g_simd == KREP_REF_AVX512 ? KREP_RA_AVX512 : g_simd == KREP_REF_AVX2 ? KREP_RA_AVX2
                  : g_simd == KREP_REF_SSE42 ? KREP_RA_SSE42 : g_simd == KREP_REF_NEON ? KREP_RA_NEON : KREP_RA_NONE;
    const size_t m = p->pattern_len;
    const bool can = !g_no_simd && simd_max > 0 && m <= simd_max;
    if (m == 1)
        return KREP_RA_MEMCHR;
    if (m < 4)
        return (can && p->case_sensitive && top != KREP_RA_NONE) ? top : KREP_RA_MEMCHR_SHORT;
    if (can)
    {
        if (g_simd == KREP_REF_AVX512 && m <= 64 && p->case_sensitive)
            return KREP_RA_AVX512;
        if ((g_simd == KREP_REF_AVX512 || g_simd == KREP_REF_AVX2) && m <= 32)
            return KREP_RA_AVX2;
        if (g_simd == KREP_REF_SSE42 && m <= 16 && p->case_sensitive)
            return KREP_RA_SSE42;
        if (g_simd == KREP_REF_NEON && p->case_sensitive)
            return KREP_RA_NEON;
    }
    if (m < 8 && repetitive_pattern(p->pattern, m))
        return KREP_RA_KMP;
    return KREP_RA_BMH;
}
// ... and the function that ends up doing the work after the internal delegation chain
// (krep.c:4708-4712, :4883-4896, :5114-5126; neon_search falls back like SSE4.2).
static int mirror_effective(int top, const search_params_t *p, size_t text_len)
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
    if (a == KREP_RA_NEON && (!p->case_sensitive || m == 0))
        a = KREP_RA_BMH;
    return a;
}
extern "C" int krep_gpu_mirror_select(const search_params_t *p, size_t text_len)
{
    if (!p)
        return KREP_RA_NONE;
    return mirror_effective(mirror_top(p), p, text_len);
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
// make room for `extra` more records (malloc family, so the reference's match_result_free works)
static bool result_reserve(match_result_t *r, uint64_t extra)
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

// ------------------------------------------------------------------------------------ plans
struct krep_gpu_plan
{
    int device = 0;
    int ref_algo = KREP_RA_NONE; // top-level selection (delegation resolved per text length)
    bool only_matching = false;
    bool cs = true, ww = false, lines = false, track = false;
    size_t max_count = SIZE_MAX;
    std::vector<std::vector<uint8_t>> pats; // as given
    // single literal
    uint32_t m = 0, p0 = 0, p1 = 0, k0 = 0, k1 = 0, l0 = 0, l1 = 0;
    uint32_t p2 = 0, p3 = 0, k2 = 0, k3 = 0, l2 = 0, l3 = 0; // bytes 8..15
    std::vector<uint8_t> pat_folded; // folded when !cs
    bool has_border = false;         // a proper prefix is also a suffix => all-occurrences != greedy
    bool has_newline = false;
    uint8_t *d_pat = nullptr;
    // workspace
    Counters *d_ctr = nullptr, *h_ctr = nullptr;
    int num_cu = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // multi-pattern
    AcTables *ac = nullptr;
    // scratch for the greedy (family N) post-pass
    PostScratch post;
    search_params_t sp{}; // shallow copy with patterns pointing into `pats`
    std::vector<const char *> pat_ptrs;
    std::vector<size_t> pat_lens;
};

static bool pattern_has_border(const std::vector<uint8_t> &p)
{
    const size_t m = p.size();
    for (size_t k = 1; k < m; ++k)
        if (memcmp(p.data(), p.data() + k, m - k) == 0)
            return true;
    return false;
}

extern "C" krep_gpu_plan_t *krep_gpu_plan_create(const search_params_t *p, int only_matching, int device)
{
    if (!p)
    {
        kg::fail("plan_create: NULL params");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        kg::fail("no HIP device available (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev)
    {
        kg::fail("device %d out of range (have %d)", device, ndev);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess)
    {
        kg::fail("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    auto *pl = new krep_gpu_plan();
    pl->device = device;
    pl->only_matching = only_matching != 0;
    pl->cs = p->case_sensitive;
    pl->ww = p->whole_word;
    pl->lines = p->count_lines_mode;
    pl->track = p->track_positions;
    pl->max_count = p->max_count;
    if (p->num_patterns >= 1 && p->patterns && p->pattern_lens)
        for (size_t i = 0; i < p->num_patterns; ++i)
            pl->pats.emplace_back((const uint8_t *)p->patterns[i], (const uint8_t *)p->patterns[i] + p->pattern_lens[i]);
    else if (p->pattern)
        pl->pats.emplace_back((const uint8_t *)p->pattern, (const uint8_t *)p->pattern + p->pattern_len);
    for (auto &v : pl->pats)
    {
        pl->pat_ptrs.push_back((const char *)v.data());
        pl->pat_lens.push_back(v.size());
    }
    pl->sp = *p;
    pl->sp.patterns = pl->pat_ptrs.data();
    pl->sp.pattern_lens = pl->pat_lens.data();
    pl->sp.num_patterns = pl->pats.size();
    if (!pl->pats.empty())
    {
        pl->sp.pattern = pl->pat_ptrs[0];
        pl->sp.pattern_len = pl->pat_lens[0];
    }
    pl->ref_algo = mirror_top(&pl->sp);

    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        pl->num_cu = prop.multiProcessorCount;
    bool ok = hipMalloc(&pl->d_ctr, sizeof(Counters)) == hipSuccess &&
              hipHostMalloc(&pl->h_ctr, sizeof(Counters)) == hipSuccess &&
              hipEventCreate(&pl->ev0) == hipSuccess && hipEventCreate(&pl->ev1) == hipSuccess;
    if (ok && pl->sp.num_patterns == 1 && pl->pats[0].size() >= 1)
    {
        const auto &raw = pl->pats[0];
        pl->m = (uint32_t)raw.size();
        pl->pat_folded = raw;
        if (!pl->cs)
            for (auto &c : pl->pat_folded)
                c = lo8(c);
        pl->has_border = pattern_has_border(pl->pat_folded);
        pl->has_newline = memchr(raw.data(), '\n', raw.size()) != nullptr;
        uint8_t w[8] = {0}, k[8] = {0};
        for (uint32_t i = 0; i < 8 && i < pl->m; ++i)
        {
            w[i] = pl->pat_folded[i];
            k[i] = 0xff;
        }
        memcpy(&pl->p0, w, 4);
        memcpy(&pl->p1, w + 4, 4);
        memcpy(&pl->k0, k, 4);
        memcpy(&pl->k1, k + 4, 4);
        if (pl->m == 1)
            pl->p0 = 0x01010101u * w[0];
        {
            uint8_t w2[8] = {0}, kk[8] = {0}, ll[8] = {0};
            for (uint32_t i = 8; i < 16 && i < pl->m; ++i)
            {
                w2[i - 8] = pl->pat_folded[i];
                kk[i - 8] = 0xff;
                ll[i - 8] = (!pl->cs && w2[i - 8] >= 'a' && w2[i - 8] <= 'z') ? 0x20 : 0;
            }
            memcpy(&pl->p2, w2, 4); memcpy(&pl->p3, w2 + 4, 4);
            memcpy(&pl->k2, kk, 4); memcpy(&pl->k3, kk + 4, 4);
            memcpy(&pl->l2, ll, 4); memcpy(&pl->l3, ll + 4, 4);
        }
        if (!pl->cs)
        { // letter lanes of the first 8 (folded) pattern bytes
            uint8_t l[8] = {0};
            for (uint32_t i = 0; i < 8 && i < pl->m; ++i)
                l[i] = (w[i] >= 'a' && w[i] <= 'z') ? 0x20 : 0;
            memcpy(&pl->l0, l, 4);
            memcpy(&pl->l1, l + 4, 4);
            if (pl->m == 1)
                pl->l0 = 0x01010101u * l[0];
        }
        ok = hipMalloc(&pl->d_pat, pl->m) == hipSuccess &&
             hipMemcpy(pl->d_pat, pl->pat_folded.data(), pl->m, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && pl->ref_algo == KREP_RA_AHO_CORASICK)
    {
        pl->ac = ac_build(pl->sp, device);
        ok = pl->ac != nullptr;
    }
    if (!ok)
    {
        if (g_err.empty())
            kg::fail("plan_create: device allocation failed");
        krep_gpu_plan_destroy(pl);
        return nullptr;
    }
    return pl;
}

#define DBGFREE(x)                                                                      \
    do                                                                                  \
    {                                                                                   \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess && getenv("KREP_GPU_DEBUG"))                               \
            fprintf(stderr, "krep-gpu: (debug) %s -> %s\n", #x, hipGetErrorString(e_)); \
    } while (0)
extern "C" void krep_gpu_plan_destroy(krep_gpu_plan_t *pl)
{
    if (!pl)
        return;
    (void)hipSetDevice(pl->device);
    if (pl->d_pat) DBGFREE(hipFree(pl->d_pat));
    if (pl->d_ctr) DBGFREE(hipFree(pl->d_ctr));
    if (pl->h_ctr) DBGFREE(hipHostFree(pl->h_ctr));
    if (pl->ev0) DBGFREE(hipEventDestroy(pl->ev0));
    if (pl->ev1) DBGFREE(hipEventDestroy(pl->ev1));
    if (pl->ac) ac_free(pl->ac);
    post_free(pl->post);
    delete pl;
}
extern "C" int krep_gpu_plan_ref_algo(const krep_gpu_plan_t *pl) { return pl ? pl->ref_algo : KREP_RA_NONE; }

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
        if (maxc == 0)
            return v;
        break;
    case KREP_RA_SSE42: // :4713 then the pre-increment checks :4778/:4804
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

// ------------------------------------------------------------------------------------ device scan
static int scan_literal(krep_gpu_plan *pl, int algo, const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                        size_t global_base, match_position_t *d_pos, uint64_t cap, hipStream_t st, int time_it,
                        krep_gpu_scan_out_t *out)
{
    const uint32_t m = pl->m;
    memset(out, 0, sizeof *out);
    if (m == 0 || text_len < m || own_lo >= own_hi)
        return 0;
    if (own_hi > text_len)
        own_hi = text_len;
    const uint64_t hi_match = std::min<uint64_t>(own_hi, text_len - m + 1);
    if (hi_match <= own_lo)
        return 0;

    // match-set family of the reference algorithm being reproduced
    bool greedy = (algo == KREP_RA_SSE42 || algo == KREP_RA_KMP);
    if (pl->only_matching && !pl->lines)
    { // -o inverts BMH and SSE4.2 (krep.c:1371, :4842); memchr_short's -o quirk is not reproduced
        if (algo == KREP_RA_BMH)
            greedy = true;
        else if (algo == KREP_RA_SSE42)
            greedy = false;
        else if (algo == KREP_RA_MEMCHR_SHORT)
            return kg::fail("memchr_short_search with -o (krep.c:4495 skips after failed candidates) is not supported");
    }
    bool need_post = greedy && pl->has_border && m > 1;
    if (need_post && pl->has_newline && pl->lines)
        return kg::fail("greedy (SSE4.2/KMP) line counting with a pattern containing a newline is not supported");
    // without -w a line holds a greedy hit iff it holds any occurrence (a line's first occurrence heads a
    // cluster), so plain -c needs no selection pass
    if (need_post && pl->lines && !pl->ww)
        need_post = false;

    LitArgs a{};
    a.text = d_text;
    a.text_len = text_len;
    a.own_lo = own_lo;
    a.own_hi = own_hi;
    a.anchor = own_lo & ~(uint64_t)15;
    {
        // big tiles (128 KiB) once there are enough of them to fill the chip several times over
        const uint64_t span = hi_match - a.anchor;
        a.rounds = span >= ((uint64_t)pl->num_cu * 16 * kRoundsBig * kSegBytes * kWavesPerBlk) ? kRoundsBig : 1;
        if (g_force_rounds == 1 || g_force_rounds == kRoundsBig)
            a.rounds = (uint32_t)g_force_rounds;
        const uint64_t tile_bytes = (uint64_t)a.rounds * kSegBytes * kWavesPerBlk;
        a.num_tiles = (span + tile_bytes - 1) / tile_bytes;
    }
    a.global_base = global_base;
    a.ww_exempt_left = ~0ull;
    a.excl_lo = a.excl_hi = 0; // empty
    if (algo == KREP_RA_AVX512 && !pl->lines && text_len >= 64 && (text_len % 64) < (uint64_t)m - 1)
    { // krep.c:5171: the last full 64-byte block is stepped over unexamined when remaining < (m-1)+64
        a.excl_hi = text_len - text_len % 64;
        a.excl_lo = a.excl_hi - 64;
    }
    if (pl->ww && !pl->lines)
    { // the BMH tail call of the AVX paths sees the tail as its own text: no left context at its first byte
        if (algo == KREP_RA_AVX2 && (text_len % 32) >= m)
            a.ww_exempt_left = text_len - text_len % 32;
        if (algo == KREP_RA_AVX512 && (text_len % 64) >= m)
            a.ww_exempt_left = text_len - text_len % 64;
    }
    a.m = m;
    a.p0 = pl->p0; a.p1 = pl->p1; a.k0 = pl->k0; a.k1 = pl->k1;
    a.l0 = pl->l0; a.l1 = pl->l1;
    a.p2 = pl->p2; a.p3 = pl->p3; a.k2 = pl->k2; a.k3 = pl->k3; a.l2 = pl->l2; a.l3 = pl->l3;
    a.pat = pl->d_pat;
    a.ctr = pl->d_ctr;

    const uint64_t maxc = pl->max_count;
    uint64_t want = 0; // records the caller can use
    if (d_pos && cap)
    {
        want = maxc;
        if (algo == KREP_RA_KMP && maxc != SIZE_MAX)
            want = maxc + 1;
        // memchr batch quirk needs the (max_count+1)-th match as well
        if (algo == KREP_RA_MEMCHR && maxc != SIZE_MAX)
            want = maxc + 1;
        want = std::min<uint64_t>(want, cap);
    }

    a.flags = (pl->cs ? 0 : F_CI);
    if (!need_post)
    {
        if (pl->ww) a.flags |= F_WW;
        if (pl->lines) a.flags |= F_LINES;
        if (want) a.flags |= F_POS;
        a.positions = (uint64_t *)d_pos;
        a.pos_cap = want;
    }
    else
    {
        // all occurrences first (no -w, no lines); greedy selection, -w and lines in the post-pass
        a.flags |= F_POS;
    }

    HIPCHK(hipSetDevice(pl->device));
    const bool chain = (a.flags & (F_POS | F_LINES)) != 0;
    const uint64_t n_units = a.num_tiles * kWavesPerBlk;
    // staging slot per unit: sized for ~4x BASELINE's densities (1e-4/B literal, 1e-2/B single byte);
    // denser units take the emit-mode re-scan
    a.stage_cap = 0;
    if (a.flags & F_POS)
        a.stage_cap = a.rounds == kRoundsBig ? (m == 1 ? 512u : 64u) : (m == 1 ? 256u : 32u);
    if (g_force_stage_cap && (a.flags & F_POS))
        a.stage_cap = (uint32_t)g_force_stage_cap;
    if (chain)
    {
        if (post_reserve(pl->post, n_units, (n_units * a.stage_cap + 3) / 4)) // 16-bit staging entries
            return 2;
        a.unitinfo = pl->post.d_unitinfo;
        a.stage = (uint64_t *)pl->post.d_stage;
        a.offsets = (const uint64_t *)pl->post.d_offsets;
    }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(a.num_tiles, (uint64_t)pl->num_cu * 8);

    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    uint64_t total = 0, lines = 0;
    unsigned long long summary = 0;
    if (!need_post)
    {
        HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
        HIPCHK(launch_literal(a, grid, st));
        const uint64_t unit_bytes = (uint64_t)a.rounds * kSegBytes, origin = a.anchor + global_base;
        if (chain && post_order(pl->post, n_units, a.stage_cap, m, origin, unit_bytes, pl->lines, (uint64_t *)d_pos, want, pl->d_ctr,
                                pl->num_cu, st))
            return 2;
        if (time_it) HIPCHK(hipEventRecord(pl->ev1, st));
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if ((a.flags & F_POS) && pl->h_ctr->overflow_units)
        {
            // some units held more hits than their staging slot: re-scan exactly those, writing in place
            LitArgs e = a;
            e.emit_mode = 1;
            HIPCHK(hipMemsetAsync(&pl->d_ctr->ticket, 0, sizeof(unsigned long long), st));
            HIPCHK(launch_literal(e, grid, st));
            if (time_it) HIPCHK(hipEventRecord(pl->ev1, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        total = pl->h_ctr->total;
        lines = pl->h_ctr->lines;
        summary = pl->h_ctr->summary;
        if (!chain)
            summary = total ? (kLnHead | kLnTail) : 0;
    }
    else
    {
        // family N with a bordered pattern: all occurrences first, then the greedy selection (kg_greedy.hip)
        const bool ww_first = pl->ww && algo == KREP_RA_BMH; // BMH -o: a -w rejected hit does not consume (krep.c:1323-1329)
        if (ww_first) a.flags |= F_WW;
        HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
        HIPCHK(launch_literal(a, grid, st));
        if (post_offsets_pass(pl->post, n_units, false, pl->d_ctr, st))
            return 2;
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const uint64_t n_occ = pl->h_ctr->total;
        if (n_occ > pl->post.occ_cap)
        {
            if (pl->post.d_occ) (void)hipFree(pl->post.d_occ);
            pl->post.d_occ = nullptr;
            pl->post.occ_cap = 0;
            HIPCHK(hipMalloc(&pl->post.d_occ, n_occ * 2 * sizeof(uint64_t)));
            pl->post.occ_cap = n_occ;
        }
        if (n_occ)
        {
            if (post_gather_pass(pl->post, n_units, a.stage_cap, m, a.anchor + global_base, (uint64_t)a.rounds * kSegBytes,
                                 pl->post.d_occ, n_occ, pl->num_cu, st))
                return 2;
            if (pl->h_ctr->overflow_units)
            {
                LitArgs e = a;
                e.emit_mode = 1;
                e.positions = pl->post.d_occ;
                e.pos_cap = n_occ;
                HIPCHK(hipMemsetAsync(&pl->d_ctr->ticket, 0, sizeof(unsigned long long), st));
                HIPCHK(launch_literal(e, grid, st));
            }
            HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
            int rc = post_greedy(pl->post, d_text, text_len, global_base, m, pl->ww && !ww_first, pl->lines, n_occ,
                                 (uint64_t *)d_pos, want, pl->d_ctr, pl->h_ctr, st, &total, &lines);
            if (rc)
                return rc;
        }
        if (time_it) HIPCHK(hipEventRecord(pl->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        summary = total ? (kLnHead | kLnTail) : 0; // shard line bits are not produced on this path
    }
    if (time_it)
    {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
        out->kernel_ms = ms;
    }
    out->total_matches = total;
    out->line_count = lines;
    out->has_newline = (summary & kLnNl) != 0;
    out->head_line_hit = (summary & kLnHead) != 0;
    out->tail_line_hit = (summary & kLnTail) != 0;
    Verdict v = verdict_for(algo, pl, total, lines, d_pos != nullptr);
    out->count = v.ret;
    if (d_pos && cap && pl->track)
    {
        // records the reference semantics need (memchr: one more than max_count, the host fixes the order)
        const uint64_t needed = (algo == KREP_RA_MEMCHR && maxc != SIZE_MAX && maxc != 0) ? std::min<uint64_t>(total, maxc + 1)
                                                                                           : v.store;
        out->overflow = needed > cap;
        out->stored = std::min<uint64_t>(needed, std::min<uint64_t>(total, want));
    }
    return 0;
}

extern "C" int krep_gpu_scan_device(krep_gpu_plan_t *pl, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                                    size_t global_base, match_position_t *d_positions, uint64_t position_capacity,
                                    void *stream, int time_it, krep_gpu_scan_out_t *out)
{
    krep_gpu_scan_out_t tmp;
    if (!out)
        out = &tmp;
    memset(out, 0, sizeof *out);
    if (!pl || (!d_text && text_len))
        return kg::fail("scan_device: bad arguments");
    {
        // hipGetLastError() is sticky per thread: an earlier, deliberately ignored failure (e.g. a hipFree in a
        // destructor) must not be mistaken for a failure of the launches below
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("KREP_GPU_DEBUG"))
            fprintf(stderr, "krep-gpu: (debug) cleared stale HIP error: %s\n", hipGetErrorString(stale));
    }
    hipStream_t st = (hipStream_t)stream;
    if (pl->ref_algo == KREP_RA_REGEX)
        return kg::fail("regex search is not part of the accelerated path");
    if (pl->ref_algo == KREP_RA_AHO_CORASICK)
        return ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, (const uint8_t *)d_text,
                       text_len, own_lo, own_hi, global_base, d_positions, position_capacity, pl->ww, pl->lines, pl->track,
                       pl->max_count, st, time_it, pl->ev0, pl->ev1, out);
    if (pl->sp.num_patterns != 1)
        return kg::fail("scan_device: no pattern");
    const int algo = mirror_effective(pl->ref_algo, &pl->sp, text_len);
    return scan_literal(pl, algo, (const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, d_positions,
                        position_capacity, st, time_it, out);
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

// ------------------------------------------------------------------------------------ host-buffer operators
namespace {
struct DevBuf
{
    uint8_t *p = nullptr;
    size_t cap = 0;
    int dev = -1;
};
thread_local DevBuf tl_text, tl_pos;
int ensure(DevBuf &b, size_t n, int dev)
{
    if (b.dev == dev && b.cap >= n && b.p)
        return 0;
    if (b.p)
    {
        (void)hipSetDevice(b.dev);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(hipSetDevice(dev));
    size_t want = std::max<size_t>(n, 1 << 20);
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    b.dev = dev;
    return 0;
}
} // namespace

// Host buffer -> HBM through two pinned staging buffers: the CPU copy of chunk k+1 overlaps the DMA of chunk k
// (the reference mmaps with MAP_POPULATE, krep.c:2630-2726; a pageable hipMemcpy stages serially).
// PCIe-bound by construction (<= ~55 GB/s); this rate is reported separately and is never the roofline number.
namespace {
struct Stager
{
    static constexpr size_t kChunk = 32u << 20;
    uint8_t *pin[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int dev = -1;
    void release()
    {
        if (dev < 0)
            return;
        (void)hipSetDevice(dev);
        for (int i = 0; i < 2; ++i)
        {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            pin[i] = nullptr;
            done[i] = nullptr;
        }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        dev = -1;
    }
    int init(int device)
    {
        if (dev == device && pin[0])
            return 0;
        release();
        HIPCHK(hipSetDevice(device));
        for (int i = 0; i < 2; ++i)
        {
            HIPCHK(hipHostMalloc(&pin[i], kChunk));
            HIPCHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        }
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        dev = device;
        return 0;
    }
    int copy(uint8_t *d_dst, const char *src, size_t len)
    {
        if (len < (4u << 20)) // small buffers: one synchronous copy is cheaper than the pipeline
        {
            HIPCHK(hipMemcpy(d_dst, src, len, hipMemcpyHostToDevice));
            return 0;
        }
        size_t off = 0;
        for (int k = 0; off < len; ++k, off += kChunk)
        {
            const int b = k & 1;
            const size_t n = std::min(kChunk, len - off);
            if (k >= 2)
                HIPCHK(hipEventSynchronize(done[b])); // the DMA that last used this staging buffer
            {
                // the staging copy is the bottleneck of the host path (one core ~12 GB/s): split it over 4 threads
                constexpr int kT = 4;
                std::thread th[kT - 1];
                const size_t part = (n + kT - 1) / kT;
                for (int q = 1; q < kT; ++q)
                {
                    const size_t o = (size_t)q * part;
                    if (o < n)
                        th[q - 1] = std::thread([=] { memcpy(pin[b] + o, src + off + o, std::min(part, n - o)); });
                }
                memcpy(pin[b], src + off, std::min(part, n));
                for (auto &t : th)
                    if (t.joinable())
                        t.join();
            }
            HIPCHK(hipMemcpyAsync(d_dst + off, pin[b], n, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(done[b], st));
        }
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
};
thread_local Stager tl_stager;
} // namespace

namespace kg {
// pinned, double-buffered host -> device copy on the calling thread's staging buffers (also used per shard thread)
int stage_to_device(uint8_t *d_dst, const char *src, size_t len, int device)
{
    if (tl_stager.init(device))
        return 2;
    return tl_stager.copy(d_dst, src, len);
}
void stage_release() { tl_stager.release(); } // short-lived shard threads give their pinned buffers back
} // namespace kg

// memchr_search's final flush (krep.c:3976-3991 + :4026-4038): when max_count is a multiple of the
// 4096-entry batch and more matches exist, the (max_count+1)-th record is stored FIRST (in front of
// the last batch) and the max_count-th is dropped.
static void memchr_batch_quirk(match_position_t *recs, uint64_t have, size_t maxc)
{
    if (maxc == SIZE_MAX || maxc == 0 || have <= maxc || (maxc % 4096) != 0)
        return;
    const uint64_t f = maxc - 4096;
    match_position_t extra = recs[maxc];
    memmove(recs + f + 1, recs + f, 4095 * sizeof(match_position_t));
    recs[f] = extra;
}

// One cached plan per calling thread: the CLI calls the operator once per file (krep.c:1950) with the same params,
// and building a plan costs device allocations + (multi-pattern) table construction.  Keyed by every field the scan
// depends on, including the mirrored globals.
namespace {
struct PlanKey
{
    std::vector<std::vector<uint8_t>> pats;
    bool cs, lines, track, ww;
    size_t max_count;
    int simd, only_matching, no_simd, algo;
    bool operator==(const PlanKey &o) const
    {
        return pats == o.pats && cs == o.cs && lines == o.lines && track == o.track && ww == o.ww && max_count == o.max_count &&
               simd == o.simd && only_matching == o.only_matching && no_simd == o.no_simd && algo == o.algo;
    }
};
struct PlanCache
{
    PlanKey key;
    krep_gpu_plan_t *plan = nullptr;
    ~PlanCache()
    {
        // process teardown: the HIP runtime may already be gone, so the plan is deliberately not destroyed here
    }
};
thread_local PlanCache tl_plan;
} // namespace

static krep_gpu_plan_t *cached_plan(const search_params_t *p)
{
    PlanKey k;
    if (p->num_patterns >= 1 && p->patterns && p->pattern_lens)
        for (size_t i = 0; i < p->num_patterns; ++i)
            k.pats.emplace_back((const uint8_t *)p->patterns[i], (const uint8_t *)p->patterns[i] + p->pattern_lens[i]);
    else if (p->pattern)
        k.pats.emplace_back((const uint8_t *)p->pattern, (const uint8_t *)p->pattern + p->pattern_len);
    k.cs = p->case_sensitive; k.lines = p->count_lines_mode; k.track = p->track_positions; k.ww = p->whole_word;
    k.max_count = p->max_count;
    k.simd = g_simd; k.only_matching = g_only_matching; k.no_simd = g_no_simd; k.algo = g_algo_override;
    if (tl_plan.plan && tl_plan.key == k)
        return tl_plan.plan;
    if (tl_plan.plan)
    {
        krep_gpu_plan_destroy(tl_plan.plan);
        tl_plan.plan = nullptr;
    }
    tl_plan.plan = krep_gpu_plan_create(p, g_only_matching, 0);
    tl_plan.key = std::move(k);
    return tl_plan.plan;
}

static uint64_t run_host_operator(const search_params_t *params, const char *text, size_t text_len, match_result_t *result,
                                  int *status)
{
    if (status)
        *status = 2;
    if (!params || (!text && text_len))
    {
        kg::fail("NULL params/text");
        return 0;
    }
    krep_gpu_plan_t *pl = cached_plan(params);
    if (!pl)
        return 0;
    uint64_t ret = 0;
    do
    {
        if (pl->ref_algo == KREP_RA_AHO_CORASICK && !params->ac_trie)
        { // aho_corasick.c:306: no trie, no matches
            if (status) *status = 0;
            break;
        }
        if (ensure(tl_text, text_len + 64, 0))
            break;
        if (text_len && (tl_stager.init(0) || tl_stager.copy(tl_text.p, text, text_len)))
        {
            if (g_err.empty())
                kg::fail("H2D copy failed");
            break;
        }
        const bool want_pos = params->track_positions && result != nullptr && !params->count_lines_mode;
        uint64_t cap = 0;
        if (want_pos)
        {
            cap = std::max<uint64_t>(1u << 16, text_len / 64);
            if (params->max_count != SIZE_MAX)
                cap = std::min<uint64_t>(cap, (uint64_t)params->max_count + 1);
            cap = std::max<uint64_t>(cap, 1);
        }
        krep_gpu_scan_out_t so;
        int rc = 0;
        for (int attempt = 0; attempt < 2; ++attempt)
        {
            if (cap && ensure(tl_pos, cap * sizeof(match_position_t), 0))
            {
                rc = 2;
                break;
            }
            rc = krep_gpu_scan_device(pl, tl_text.p, text_len, 0, text_len, 0, cap ? (match_position_t *)tl_pos.p : nullptr, cap,
                                      nullptr, 0, &so);
            if (rc || !so.overflow)
                break;
            cap = so.total_matches + 1; // exact size, second and last pass
        }
        if (rc)
            break;
        ret = so.count;
        if (want_pos && so.stored)
        {
            if (g_result_order && pl->ref_algo == KREP_RA_AHO_CORASICK &&
                krep_gpu_order_by_start((match_position_t *)tl_pos.p, so.stored, text_len, nullptr))
            {
                ret = 0;
                break;
            }
            std::vector<match_position_t> tmp(so.stored);
            if (hipMemcpy(tmp.data(), tl_pos.p, so.stored * sizeof(match_position_t), hipMemcpyDeviceToHost) != hipSuccess)
            {
                kg::fail("D2H copy failed");
                ret = 0;
                break;
            }
            uint64_t n = so.stored;
            const int algo = pl->ref_algo == KREP_RA_AHO_CORASICK ? KREP_RA_AHO_CORASICK
                                                                   : mirror_effective(pl->ref_algo, &pl->sp, text_len);
            if (algo == KREP_RA_MEMCHR && params->max_count != SIZE_MAX)
            {
                memchr_batch_quirk(tmp.data(), n, params->max_count);
                n = std::min<uint64_t>(n, params->max_count);
            }
            if (!result_reserve(result, n))
            {
                kg::fail("out of memory growing match_result_t");
                ret = 0;
                break;
            }
            memcpy(result->positions + result->count, tmp.data(), n * sizeof(match_position_t));
            result->count += n;
        }
        if (status)
            *status = 0;
    } while (0);
    return ret; // the plan stays in this thread's one-entry cache
}

extern "C" uint64_t krep_gpu_literal_search(const search_params_t *params, const char *text, size_t len, match_result_t *result)
{
    return run_host_operator(params, text, len, result, nullptr);
}
extern "C" uint64_t krep_gpu_aho_corasick_search(const search_params_t *params, const char *text, size_t len,
                                                 match_result_t *result)
{
    return run_host_operator(params, text, len, result, nullptr);
}
extern "C" search_func_t krep_gpu_select_search_algorithm(const search_params_t *params)
{
    if (!params || params->use_regex)
        return nullptr;
    return params->num_patterns > 1 ? krep_gpu_aho_corasick_search : krep_gpu_literal_search;
}

// search_string()'s validation and verdict (krep.c:2013-2049, :2166-2199), minus strlen and printing
extern "C" int search_buffer(const search_params_t *params, const char *buf, size_t len, int only_matching, int num_gpus,
                             match_result_t *out, uint64_t *count_out)
{
    if (count_out)
        *count_out = 0;
    if (!params || params->num_patterns == 0)
        return kg::fail("Error: No pattern specified.");
    if (!buf && len)
        return kg::fail("Error: NULL text in search_buffer.");
    if (params->use_regex)
        return kg::fail("regex search is not accelerated; keep krep's regex_search for it");
    for (size_t i = 0; i < params->num_patterns; ++i)
    {
        if (params->pattern_lens[i] == 0)
        {
            if (params->num_patterns > 1)
                return kg::fail("Error: Empty pattern provided for literal search with multiple patterns.");
        }
        else if (params->pattern_lens[i] > 1024) // MAX_PATTERN_LENGTH, krep.c:77
            return kg::fail("Error: Pattern too long (max 1024).");
    }
    const int saved = g_only_matching;
    g_only_matching = only_matching != 0;
    int st = 2;
    uint64_t n = 0;
    if (num_gpus > 1)
        n = multi_gpu_search(params, buf, len, num_gpus, out, &st);
    else
    {
        search_params_t local = *params;
        static int dummy_trie;
        if (local.num_patterns > 1 && !local.ac_trie)
            local.ac_trie = (ac_trie_t *)&dummy_trie; // search_string builds the trie itself (krep.c:2067-2078)
        n = run_host_operator(&local, buf, len, out, &st);
    }
    g_only_matching = saved;
    if (st)
        return 2;
    const size_t maxc = params->max_count;
    if (maxc != SIZE_MAX && n > maxc)
        n = maxc;
    if (out && maxc != SIZE_MAX && out->count > maxc)
        out->count = maxc;
    bool found;
    if (params->count_lines_mode || params->count_matches_mode)
        found = n > 0;
    else
    {
        found = out && out->count > 0;
        if (found)
            n = out->count;
        else if (!out)
            found = n > 0;
    }
    if (count_out)
        *count_out = n;
    return found ? 0 : 1;
}

// ------------------------------------------------------------------------------------ generators
__global__ void synth_kernel(uint8_t *dst, size_t len, size_t goff, int kind, uint64_t seed, const uint8_t *plant,
                             uint64_t plen, uint64_t period)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < len; i += stride)
    {
        uint32_t w[4] = {0, 0, 0, 0};
        const size_t n = len - i < 16 ? len - i : 16;
        for (size_t b = 0; b < n; ++b)
            w[b >> 2] |= (uint32_t)synth_byte(goff + i + b, kind, seed, plant, plen, period) << (8 * (b & 3));
        if (n == 16 && (((uintptr_t)(dst + i)) & 15) == 0)
            *reinterpret_cast<uint4 *>(dst + i) = make_uint4(w[0], w[1], w[2], w[3]);
        else
            for (size_t b = 0; b < n; ++b)
                dst[i + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
    }
}

extern "C" int krep_gpu_generate(void *d_dst, size_t len, size_t global_off, int kind, uint64_t seed, const void *plant,
                                 size_t plant_len, uint64_t period, void *stream)
{
    if (!len)
        return 0;
    if ((kind == 2 || kind == 3 || kind == 4) && (!plant || !plant_len))
        return kg::fail("generate: kind %d needs a plant", kind);
    if ((kind == 2) && period < plant_len)
        return kg::fail("generate: period < plant length");
    hipStream_t st = (hipStream_t)stream;
    uint8_t *d_plant = nullptr;
    if (plant_len)
    {
        HIPCHK(hipMalloc(&d_plant, plant_len));
        HIPCHK(hipMemcpyAsync(d_plant, plant, plant_len, hipMemcpyHostToDevice, st));
    }
    const uint64_t plen = (kind == 4) ? 0 : plant_len;
    hipLaunchKernelGGL(synth_kernel, dim3(256 * 16), dim3(256), 0, st, (uint8_t *)d_dst, len, global_off, kind, seed, d_plant,
                       plen, period ? period : 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (d_plant) (void)hipFree(d_plant);
    return 0;
}
extern "C" void krep_gpu_generate_host(void *dst, size_t len, size_t global_off, int kind, uint64_t seed, const void *plant,
                                       size_t plant_len, uint64_t period)
{
    uint8_t *d = (uint8_t *)dst;
    const uint64_t plen = (kind == 4) ? 0 : plant_len;
    for (size_t i = 0; i < len; ++i)
        d[i] = synth_byte(global_off + i, kind, seed, (const uint8_t *)plant, plen, period ? period : 1);
}
