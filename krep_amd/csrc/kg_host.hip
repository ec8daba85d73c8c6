// kg_host.hip — core of the C-ABI of the krep-gpu backend (include/krep_gpu.h): configuration (the mirrored
// reference globals), the select_search_algorithm() mirror, the result container, plans, and the device-resident
// scan krep_gpu_scan_device[_ex]() with every reference return-value convention.
// Host logic only; kernels live in kg_literal.hip / kg_ac.hip / kg_post.hip / kg_greedy.hip / kg_tail.hip.
// The host-buffer operators (search_func_t entry points, search_buffer, streaming ingest) are in kg_ops.hip.
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
bool have_error() { return !g_err.empty(); }
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
extern "C" const char *krep_gpu_version(void) { return "krep-gpu 0.3 (gfx950)"; }
extern "C" int krep_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// ------------------------------------------------------------------------------------ configuration
// The reference decides its algorithm — hence the match-set family — from compile-time SIMD macros and three
// file-static globals (krep.c:47-74, :117-120).  Here they are one explicit krep_gpu_config_t.  The setters below write
// PROCESS-WIDE defaults (relaxed atomics: the reference's globals are process-wide too, set once by main() before the
// pool threads start); krep_gpu_set_thread_config() overrides them for the calling thread; plans and search_buffer_ex()
// carry their configuration explicitly.  Nothing on a scan path writes any of this.
static std::atomic<int> g_simd{KREP_REF_AVX2}, g_only_matching{0}, g_no_simd{0}, g_algo_override{KREP_ALGO_AUTO},
    g_result_order{0}, g_device{-1};
static std::atomic<size_t> g_stream_chunk{0};
static std::atomic<int> g_num_gpus{INT32_MIN};            // INT32_MIN: not set -> $KREP_GPU_NUM, else 1
static std::atomic<size_t> g_min_bytes{SIZE_MAX};         // SIZE_MAX: not set -> $KREP_GPU_MIN_BYTES, else 1 MiB
static thread_local bool tl_cfg_set = false;
static thread_local krep_gpu_config_t tl_cfg;

static int env_device()
{
    const char *e = getenv("KREP_GPU_DEVICE");
    return e && *e ? atoi(e) : 0;
}
static int env_num_gpus()
{
    const char *e = getenv("KREP_GPU_NUM");
    return e && *e ? atoi(e) : 1;
}
static size_t env_min_bytes()
{
    const char *e = getenv("KREP_GPU_MIN_BYTES");
    return e && *e ? (size_t)strtoull(e, nullptr, 0) : ((size_t)1 << 20);
}
extern "C" void krep_gpu_config_default(krep_gpu_config_t *c)
{
    if (!c)
        return;
    c->reference_simd = g_simd.load(std::memory_order_relaxed);
    c->only_matching = g_only_matching.load(std::memory_order_relaxed);
    c->force_no_simd = g_no_simd.load(std::memory_order_relaxed);
    c->algo_override = g_algo_override.load(std::memory_order_relaxed);
    c->result_order = g_result_order.load(std::memory_order_relaxed);
    const int d = g_device.load(std::memory_order_relaxed);
    c->device = d >= 0 ? d : env_device();
    c->stream_chunk_bytes = g_stream_chunk.load(std::memory_order_relaxed);
    const int ng = g_num_gpus.load(std::memory_order_relaxed);
    c->num_gpus = ng != INT32_MIN ? ng : env_num_gpus();
    const size_t mb = g_min_bytes.load(std::memory_order_relaxed);
    c->min_text_bytes = mb != SIZE_MAX ? mb : env_min_bytes();
}
extern "C" void krep_gpu_set_thread_config(const krep_gpu_config_t *c)
{
    tl_cfg_set = c != nullptr;
    if (c)
        tl_cfg = *c;
}
namespace kg {
krep_gpu_config_t current_config()
{
    if (tl_cfg_set)
        return tl_cfg;
    krep_gpu_config_t c;
    krep_gpu_config_default(&c);
    return c;
}
} // namespace kg
extern "C" void krep_gpu_set_reference_simd(int l) { g_simd.store(l, std::memory_order_relaxed); }
extern "C" int krep_gpu_get_reference_simd(void) { return kg::current_config().reference_simd; }
extern "C" void krep_gpu_set_only_matching(int on) { g_only_matching.store(on != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_result_order(int by_start) { g_result_order.store(by_start != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_force_no_simd(int on) { g_no_simd.store(on != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_algo_override(int a) { g_algo_override.store(a, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_device(int d) { g_device.store(d, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_stream_chunk(size_t bytes) { g_stream_chunk.store(bytes, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_num_gpus(int n) { g_num_gpus.store(n, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_min_text_bytes(size_t b) { g_min_bytes.store(b, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------ availability
// "Is there a device this library can run on" is asked by the SELECTOR, before any operator is handed out (SURVEY §8b:
// a backend must be able to fail BEFORE producing output): device count, range of the configured device, gfx950, and one
// probe kernel of this code object launched and read back.  Once per device and process.
__global__ void kg_probe_kernel(unsigned *out) { *out = 0x950u; }
namespace {
struct Avail
{
    std::mutex mu;
    std::vector<int> state;           // per device: 0 unknown, 1 usable, 2 not usable
    std::vector<std::string> reason;  // why not
};
Avail &avail()
{
    static Avail *a = new Avail(); // leaked: may be consulted from atexit paths
    return *a;
}
thread_local std::string tl_unavail;
bool probe_device(int device, std::string &why)
{
    int prev = -1;
    (void)hipGetDevice(&prev);
    auto done = [&](bool ok) {
        if (prev >= 0)
            (void)hipSetDevice(prev);
        (void)hipGetLastError();
        return ok;
    };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    {
        why = "hipGetDeviceProperties failed";
        return done(false);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        why = std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only";
        return done(false);
    }
    unsigned *d = nullptr, h = 0;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, sizeof(unsigned)) != hipSuccess)
    {
        why = "cannot allocate on the device";
        return done(false);
    }
    hipLaunchKernelGGL(kg_probe_kernel, dim3(1), dim3(1), 0, nullptr, d);
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h == 0x950u;
    (void)hipFree(d);
    if (!ok)
        why = "the gfx950 code object of this library does not run on the device";
    return done(ok);
}
} // namespace
namespace kg {
// NULL = usable; otherwise the reason (valid until the calling thread asks again)
const char *device_unusable(int device)
{
    if (const char *e = getenv("KREP_GPU_DISABLE"))
        if (*e && *e != '0')
            return "disabled by KREP_GPU_DISABLE";
    if (const char *e = getenv("KREP_GPU_ASSUME_AVAILABLE")) // test hook: skip the probe, so that a box WITHOUT a device
        if (*e && *e != '0')                                  // reaches the operators and exercises their run-time failure paths
            return nullptr;
    int ndev = 0;
    const auto t_first = std::chrono::steady_clock::now(); // (the process's first HIP call starts the runtime: kg_cost.hip wants to know)
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        (void)hipGetLastError();
        return "no HIP device available";
    }
    if (device < 0 || device >= ndev)
    {
        tl_unavail = "device " + std::to_string(device) + " out of range (have " + std::to_string(ndev) + ")";
        return tl_unavail.c_str();
    }
    Avail &a = avail();
    std::lock_guard<std::mutex> lk(a.mu);
    if ((size_t)ndev > a.state.size())
    {
        a.state.resize((size_t)ndev, 0);
        a.reason.resize((size_t)ndev);
    }
    if (a.state[device] == 0)
    {
        a.state[device] = probe_device(device, a.reason[device]) ? 1 : 2;
        if (a.state[device] == 1)
            cost_note_device_init(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count());
    }
    if (a.state[device] == 1)
        return nullptr;
    tl_unavail = a.reason[device];
    return tl_unavail.c_str();
}
// ---- failure injection (test hook): the failure paths of the operators must be reachable on a healthy box
static std::atomic<int> g_inject{-1};
bool inject(int kind)
{
    int v = g_inject.load(std::memory_order_relaxed);
    if (v < 0)
    {
        const char *e = getenv("KREP_GPU_INJECT_FAILURE");
        v = e && *e ? atoi(e) : 0;
        g_inject.store(v, std::memory_order_relaxed);
    }
    return v == kind;
}
} // namespace kg
extern "C" void krep_gpu_debug_inject_failure(int kind) { kg::g_inject.store(kind < 0 ? 0 : kind, std::memory_order_relaxed); }
extern "C" int krep_gpu_available(void) { return kg::device_unusable(kg::current_config().device) == nullptr ? 1 : 0; }
extern "C" const char *krep_gpu_unavailable_reason(void)
{
    const char *r = kg::device_unusable(kg::current_config().device);
    return r ? r : "";
}

static std::atomic<int> g_force_rounds{0};    // test hook: 0 = auto, 1 / 4 = force the tile shape
static std::atomic<int> g_force_stage_cap{0}; // test hook: staging records per unit (0 = auto)
namespace kg { extern int g_ac_force_stage_cap; }
extern "C" void krep_gpu_debug_force_stage_cap(int c) { g_force_stage_cap.store(c); kg::g_ac_force_stage_cap = c; }
extern "C" void krep_gpu_debug_force_rounds(int r) { g_force_rounds.store(r); }
namespace kg { extern int g_s1_force_grid; }
static std::atomic<uint64_t> g_fused1_failovers{0}; // one-pass single-byte scans that handed over to the two-pass kernels
extern "C" void krep_gpu_debug_force_single_grid(int blocks) { kg::g_s1_force_grid = blocks < 0 ? 0 : blocks; }
extern "C" uint64_t krep_gpu_debug_single_failovers(void) { return g_fused1_failovers.load(); }

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
static bool pattern_has_border(const uint8_t *p, size_t m)
{
    for (size_t k = 1; k < m; ++k)
        if (memcmp(p, p + k, m - k) == 0)
            return true;
    return false;
}
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
    const int top = mirror_top(p, c);
    const int eff = mirror_effective(top, p, SIZE_MAX / 2);
    if (p->count_lines_mode && c.only_matching && eff == KREP_RA_MEMCHR_SHORT)
        return "memchr_short_search with count_lines_mode AND only_matching is not accelerated (unreachable from the "
               "reference CLI, krep.c:3811-3814)";
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

// ------------------------------------------------------------------------------------ plans
extern "C" krep_gpu_plan_t *krep_gpu_plan_create_ex(const search_params_t *p, const krep_gpu_config_t *cfg_in)
{
    if (!p)
    {
        kg::fail("plan_create: NULL params");
        return nullptr;
    }
    const krep_gpu_config_t cfg = cfg_in ? *cfg_in : kg::current_config();
    const int device = cfg.device;
    if (const char *why = kg::device_unusable(device))
    {
        kg::fail("%s", why);
        return nullptr;
    }
    if (kg::inject(1))
    {
        kg::fail("injected failure: device allocation (plan)");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess)
    {
        kg::fail("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    auto *pl = new krep_gpu_plan();
    pl->cfg = cfg;
    pl->device = device;
    pl->only_matching = cfg.only_matching != 0;
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
    pl->ref_algo = mirror_top(&pl->sp, cfg);
    if (const char *why = kg::unsupported_reason(&pl->sp, cfg))
        pl->unsupported = why;

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
        pl->has_border = pattern_has_border(pl->pat_folded.data(), pl->pat_folded.size());
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
        if (ok && pl->m > 8)
        {
            pl->n_chunks = (pl->m - 8 + 7) / 8;
            std::vector<unsigned long long> ch(2 * pl->n_chunks, 0ull); // pattern words, then their letter masks (-i)
            for (uint32_t k = 0; k < pl->n_chunks; ++k)
            {
                const uint8_t *src = pl->pat_folded.data() + std::min<uint32_t>(8 + 8 * k, pl->m - 8);
                memcpy(&ch[k], src, 8);
                uint8_t l[8];
                for (int b = 0; b < 8; ++b)
                    l[b] = (!pl->cs && src[b] >= 'a' && src[b] <= 'z') ? 0x20 : 0;
                memcpy(&ch[pl->n_chunks + k], l, 8);
            }
            ok = hipMalloc(&pl->d_pat_chunks, ch.size() * 8) == hipSuccess &&
                 hipMemcpy(pl->d_pat_chunks, ch.data(), ch.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    if (ok && pl->ref_algo == KREP_RA_AHO_CORASICK)
    {
        for (auto &v : pl->pats)
            if (!v.empty() && memchr(v.data(), '\n', v.size()))
                pl->ac_has_newline = true;
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
extern "C" krep_gpu_plan_t *krep_gpu_plan_create(const search_params_t *p, int only_matching, int device)
{
    krep_gpu_config_t c = kg::current_config();
    c.only_matching = only_matching != 0;
    c.device = device;
    return krep_gpu_plan_create_ex(p, &c);
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
    if (pl->d_pat_chunks) DBGFREE(hipFree(pl->d_pat_chunks));
    if (pl->d_ctr) DBGFREE(hipFree(pl->d_ctr));
    if (pl->h_ctr) DBGFREE(hipHostFree(pl->h_ctr));
    if (pl->ev0) DBGFREE(hipEventDestroy(pl->ev0));
    if (pl->ev1) DBGFREE(hipEventDestroy(pl->ev1));
    if (pl->ac) ac_free(pl->ac);
    if (pl->d_nl_rec) DBGFREE(hipFree(pl->d_nl_rec));
    if (pl->d_nl_ln) DBGFREE(hipFree(pl->d_nl_ln));
    post_free(pl->post);
    post_free(pl->aux);
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
    const bool chain = (a.flags & (F_POS | F_LINES)) != 0;
    const uint64_t n_units = a.num_tiles * kWavesPerBlk;
    // Dealing: from ~24 GiB on every wave draws tickets of 8 units (>= 24 tickets per resident wave, ~26 fetch-adds/us on the
    // ticket word); below that the static interleaved deal (upt = 0) is faster — measured at 8 GiB: 1.38 vs 1.41 ms with
    // offsets, 1.29 vs 1.37 ms counting; at 16 GiB 2.58 vs 2.63 ms; at 2 GiB 0.37 vs 0.48 ms (the ticket word becomes the limit
    // of a short scan); at 32 GiB tickets win: 5.20 vs 5.31 ms, and 7.1 vs 7.9 ms on the single-byte workload
    a.upt = (a.rounds == kRoundsBig && n_units / ((uint64_t)pl->num_cu * 16) >= 192) ? 8 : 0;
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
    // at their final index (kg_single.hip) — no staging, no info words, no post-pass.  Too dense for its LDS rings (> ~1.5 %
    // hits): counted but not recorded; the two-pass kernels below take this scan and the plan's later ones.
    if (m_scan == 1 && ps.sink == LitPass::RECORDS && !ps.ww && !ps.lines && !ps.first_byte && a.rounds == kRoundsBig && fsc == 0 &&
        ps.excl_lo == ps.excl_hi && pl->fused1_ok && w.text_len >= 2 * (size_t)kSegBytes && !getenv("KREP_GPU_NO_FUSED1"))
    {
        const uint64_t n_tk = single_fused_tickets(n_units);
        if (n_tk > post.tk_cap)
        {
            if (post.d_tk) (void)hipFree(post.d_tk);
            post.d_tk = nullptr;
            post.tk_cap = 0;
            HIPCHK(hipMalloc(&post.d_tk, single_fused_scratch_words(n_tk) * sizeof(unsigned long long)));
            post.tk_cap = n_tk;
        }
        a.positions = ps.d_out;
        a.pos_cap = ps.out_cap;
        HIPCHK(hipMemsetAsync(pl->d_ctr, 0, sizeof(Counters), st));
        HIPCHK(hipMemsetAsync(post.d_tk, 0, single_fused_scratch_words(n_tk) * sizeof(unsigned long long), st));
        HIPCHK(launch_single_fused(a, post.d_tk, post.d_tk + n_tk, n_tk, grid, st));
        if (ps.ev_end) HIPCHK(hipEventRecord(ps.ev_end, st));
        HIPCHK(hipMemcpyAsync(pl->h_ctr, pl->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (!pl->h_ctr->overflow_units)
        {
            res->total = pl->h_ctr->total;
            res->summary = res->total ? (kLnHead | kLnTail) : 0;
            return 0;
        }
        pl->fused1_ok = false;
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
        if ((a.flags & F_POS) && m_scan != 1 && a.rounds == kRoundsBig && pl->h_ctr->overflow_units * 64 > n_units)
            pl->sparse_cap = 64; // a dense input: the next scans of this plan stage 64 hits per unit
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
// needs, for neon_search with its grid origin — krep_gpu_seq_carry_t).  kSplitWhole: one window only — the newline-pattern -c
// walk of kg_greedy.hip (3), neon_search's max_count == 0 corner, multi-pattern -c with a '\n' inside a pattern.
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
                return kSplitWhole;
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
    if (fam.neon_zero || fam.nlwalk)
        return kSplitWhole;
    return (fam.need_walk || fam.replay) ? kSplitChain : kSplitPieces;
}
bool shardable(const search_params_t *p, const krep_gpu_config_t &c, size_t text_len) { return split_mode(p, c, text_len) != kSplitWhole; }
// The left fold of the boundary record (include/krep_gpu.h, krep_gpu_seq_carry_t): used by a piece that knows its predecessor's
// record, and by the host after shards were scanned out of order.
krep_gpu_seq_carry_t fold_carry(const krep_gpu_seq_carry_t &in, const krep_gpu_seq_carry_t &pc)
{
    krep_gpu_seq_carry_t o = pc; // keeps the piece's local_* fields
    o.resume = std::max<uint64_t>(in.resume, pc.resume);
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
        // start, the -w verdicts — then ONE thread walks the list the way the reference's loop moves (kg_greedy.hip (3))
        if (!whole)
            return kg::fail("-c through %s with a newline inside the pattern is a chain over every counted match: scan the whole "
                            "text in one window", krep_gpu_algorithm_name(algo));
        if (pl->max_count == 0)
            return 0; // krep.c:4713, :1634
        HIPCHK(hipSetDevice(pl->device));
        if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
        LitPass ps;
        ps.own_lo = 0; ps.own_hi = own_hi; ps.sink = LitPass::OCC; ps.post = &pl->post;
        LitResult lr0;
        if (lit_pass(pl, w, ps, st, &lr0))
            return 2;
        uint64_t lines_counted = 0;
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
            if (krep_gpu_line_numbers(w.d_text, w.text_len, (const match_position_t *)pl->post.d_occ, lr0.total, pl->d_nl_ln, st))
                return 2;
            const uint32_t k0 = (uint32_t)((const uint8_t *)memchr(pl->pats[0].data(), '\n', m) - pl->pats[0].data());
            if (post_nlwalk(pl->post, w.d_text, w.text_len, algo == KREP_RA_KMP ? kNlWalkKmp : kNlWalkSse42, m, k0, pl->ww,
                            pl->only_matching, pl->max_count == SIZE_MAX ? ~0ull : (uint64_t)pl->max_count, lr0.total, pl->d_nl_ln,
                            pl->d_ctr, pl->h_ctr, st, &lines_counted))
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
        out->total_matches = lines_counted;
        out->line_count = lines_counted;
        out->head_line_hit = out->tail_line_hit = lines_counted != 0;
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
            const krep_gpu_seq_carry_t co = kg::fold_carry(in, pc);
            if (carry_out)
                *carry_out = co;
            total = lr.total; lines = lr.lines; summary = lr.summary;
            if (final_piece)
            {
                // where the reference's block loop stands when it enters the last kReplayWindow bytes (replay_entry, global offsets)
                uint64_t cur, extra = 0;
                int open = 0;
                if (!co.q1)
                    cur = (X / B) * B; // the block grid never left offset 0
                else if (co.nl1)
                    cur = co.nl1 <= X ? co.nl1 + ((X - co.nl1) / B) * B : co.nl1; // restarted at the line start behind q
                else if (algo == KREP_RA_NEON)
                {
                    cur = co.g0 + ((X - co.g0) / B) * B; // no restart on an unterminated line: the previous counted line's grid
                    open = 1;
                }
                else
                    cur = G; // unterminated line counted: the clamped advance ended the scan (krep.c:5006-5008, :5211-5213)
                ReplayIn r{};
                r.algo = algo; r.m = m; r.ww = pl->ww; r.n = G; r.cur = cur; r.open = open;
                r.text = w.d_text - base; // indexed with global offsets >= cur - 1 >= base: inside the buffer
                r.pat = pl->d_pat;
                if (cur < G && tail_run_replay(r, d_slot, h_slot, st, &extra))
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
        LitPass ps;
        ps.own_lo = lo_eff; ps.own_hi = own_hi; ps.sink = LitPass::OCC; ps.post = &pl->post;
        ps.first_byte = mshort_o;
        ps.ww = ww_first;
        if (lit_pass(pl, w, ps, st, &lr))
            return 2;
        uint64_t resume_out = 0;
        WalkSpec ws{};
        ws.mode = mshort_o ? kWalkShortO : kWalkGreedy;
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
static int scan_ac_newline_lines(krep_gpu_plan *pl, const Window &w, hipStream_t st, int time_it, krep_gpu_scan_out_t *out)
{
    if (!(w.global_base == 0 && w.own_lo == 0 && w.own_hi >= w.text_len && w.global_len == w.text_len))
        return kg::fail("multi-pattern -c with a newline inside a pattern counts emission-order line changes: scan the whole "
                        "text in one window");
    if (pl->max_count == 0) // aho_corasick.c:316
        return 0;
    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    krep_gpu_scan_out_t o1;
    int rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, 0, w.text_len, 0, nullptr, 0, pl->ww,
                     false, false, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1);
    if (rc)
        return rc;
    const uint64_t total = o1.total_matches;
    uint64_t changes = 0;
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
        match_position_t *d_rec = pl->d_nl_rec;
        uint64_t *d_ln = pl->d_nl_ln;
        rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, 0, w.text_len, 0, d_rec, total, pl->ww,
                     false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1);
        if (!rc)
            rc = krep_gpu_line_numbers(w.d_text, w.text_len, d_rec, total, d_ln, st);
        if (!rc)
            rc = tail_count_changes(d_ln, total, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0], st, &changes);
        if (rc)
            return rc;
    }
    if (time_it)
    {
        HIPCHK(hipEventRecord(pl->ev1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, pl->ev0, pl->ev1));
        out->kernel_ms = ms;
    }
    out->total_matches = total;
    out->line_count = changes;
    out->head_line_hit = out->tail_line_hit = changes != 0;
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
// START here, by END in the in-kernel road: the same lines either way (a match without '\n' starts and ends on one line).
// Returns 1 when the text turns out to be too dense for the list (the caller takes the in-kernel road), 2 on error.
constexpr size_t kAcLinesOnListMin = (size_t)32 << 20;
static int scan_ac_lines_on_list(krep_gpu_plan *pl, const Window &w, hipStream_t st, int time_it, krep_gpu_scan_out_t *out)
{
    memset(out, 0, sizeof *out);
    if (pl->max_count == 0) // aho_corasick.c:316
        return 0;
    const size_t own_hi = std::min(w.own_hi, w.text_len);
    if (w.own_lo >= own_hi)
        return 0;
    if (time_it) HIPCHK(hipEventRecord(pl->ev0, st));
    const uint64_t dense = w.text_len / 16 + 4096; // more matches than this: 16 B of record per 16 B of text is no shortcut
    if (pl->nl_cap == 0)
    {
        const uint64_t want = std::max<uint64_t>(w.text_len / 1024, 1u << 16);
        HIPCHK(hipMalloc(&pl->d_nl_rec, want * sizeof(match_position_t)));
        pl->nl_cap = want;
    }
    krep_gpu_scan_out_t o1;
    for (int attempt = 0;; ++attempt)
    {
        const int rc = ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, w.d_text, w.text_len, w.own_lo, own_hi, w.global_base,
                               pl->d_nl_rec, pl->nl_cap, pl->ww, false, true, SIZE_MAX, st, 0, pl->ev0, pl->ev1, &o1, true);
        if (rc)
            return rc;
        if (o1.total_matches > dense)
            return 1;
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
    if (total && lines == ~0ull && tail_count_line_gaps(w.d_text, w.global_base, (const uint64_t *)pl->d_nl_rec, total, &pl->d_ctr->pad[0], &pl->h_ctr->pad[0],
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
        return scan_ac_newline_lines(pl, w, st, time_it, out);
    }
    if (pl->ref_algo == KREP_RA_AHO_CORASICK && pl->lines && text_len >= kAcLinesOnListMin && !getenv("KREP_GPU_AC_LINES_INKERNEL"))
    {
        // (small texts keep the in-kernel road: the list road ends with a few host round trips, ~0.1 ms)
        Window w{(const uint8_t *)d_text, text_len, own_lo, own_hi, global_base, global_len};
        const int rc = scan_ac_lines_on_list(pl, w, st, time_it, out);
        if (rc != 1)
            return rc;
    }
    if (pl->ref_algo == KREP_RA_AHO_CORASICK)
        return ac_scan(pl->ac, pl->d_ctr, pl->h_ctr, pl->post, pl->num_cu, (const uint8_t *)d_text,
                       text_len, own_lo, own_hi, global_base, d_positions, position_capacity, pl->ww, pl->lines, pl->track,
                       pl->max_count, st, time_it, pl->ev0, pl->ev1, out);
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
