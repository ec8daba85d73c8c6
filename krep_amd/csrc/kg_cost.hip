// kg_cost.hip — should THIS text go to the GPU?  krep_gpu_worthwhile() and the cost model behind it.
//
// The reference has a size policy of its own: search_file() handles small files on one thread and scales its thread count
// with the file size (krep.c:2404-2420, :2729-2770).  A drop-in backend needs the same kind of answer, per text: the host
// path of this library is PCIe-bound (54.7 GB/s measured end to end, profiles/r03_host_path.txt) and a process's first device
// call pays the HIP runtime start (0.3-0.6 s), while the reference's SIMD literal functions reach 150 GB/s in-process on the
// 256 threads of the GPU box's host — for a single literal in host memory the CPU function is the faster one, for a
// 1000-pattern dictionary (aho_corasick_search: 2.2 GB/s on the same 256 threads) the GPU wins from a few MiB on.
//   t_gpu = (device not yet initialised in this process ? init : 0) + launch + bytes / host_path_rate
//   t_cpu = bytes / min(threads x per-thread rate of the function select_search_algorithm() would run, its memory-bound cap)
//   GPU iff size >= min_text_bytes and t_gpu < t_cpu
// `threads` is what search_file() would use (krep.c:2748-2759: min(cores, size / 4 MiB), at least 1) unless the caller says
// otherwise.  Rates: defaults measured on the GPU box of this repository (bench.py cpu_baseline, profiles/), each overridable
// ($KREP_GPU_COST, krep_gpu_set_cost_rates), and the two GPU-side figures are CALIBRATED ONCE PER PROCESS by the library
// itself: the device-start time by the availability probe, the host-path rate by the first large operator call.
// $KREP_GPU_COST_MODEL=0 (or rates.enabled = 0) keeps the round-3 rule: size alone.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/krep_gpu.h"
#include "kg_internal.h"

namespace {
std::mutex g_mu;
krep_gpu_cost_rates_t g_rates;
bool g_rates_init = false;
std::atomic<double> g_meas_host_gbps{0.0}; // calibrated: the host path of this process (0 = not measured yet)
std::atomic<double> g_meas_init_ms{-1.0};  // what the first device call of this process cost (-1 = none yet): only its PRESENCE is used —
                                           // a process whose device is up no longer pays r.gpu_init_ms (the figure itself is a rate default)

krep_gpu_cost_rates_t defaults()
{
    krep_gpu_cost_rates_t r;
    r.enabled = 1;
    r.gpu_host_path_gbps = 50.0; // 54.7 measured for 4 GiB (profiles/r03_host_path.txt); small texts see less
    r.gpu_launch_us = 100.0;     // one operator call with a cached plan: 50-90 us
    r.gpu_init_ms = 450.0;       // HIP runtime + device start of a fresh process
    r.cpu_memchr_gbps = 12.0; r.cpu_memchr_cap_gbps = 180.0; // memchr_search (glibc AVX2 memchr)
    r.cpu_simd_gbps = 6.0;    r.cpu_simd_cap_gbps = 150.0;   // simd_sse42 / avx2 / avx512 / neon, memchr_short_search
    r.cpu_scalar_gbps = 1.5;  r.cpu_scalar_cap_gbps = 100.0; // boyer_moore_search, kmp_search
    r.cpu_ac_gbps = 0.4;      r.cpu_ac_cap_gbps = 100.0;     // aho_corasick_search while its trie fits the caches ...
    r.cpu_ac_cache_bytes = 1 << 20;                          // ... (2 KiB per state, aho_corasick.h) and, beyond that,
    r.cpu_ac_exponent = 1.3;                                 // x (cache / trie bytes)^1.3: 1000 patterns, 8605 states -> 9 MB/s per thread
    return r;
}
void parse_env(krep_gpu_cost_rates_t &r)
{
    if (const char *e = getenv("KREP_GPU_COST_MODEL"))
        if (*e == '0')
            r.enabled = 0;
    const char *e = getenv("KREP_GPU_COST"); // "host=50,launch=100,init=450,memchr=12:180,simd=6:150,scalar=1.5:100,ac=0.4:100"
    if (!e)
        return;
    char buf[512];
    strncpy(buf, e, sizeof buf - 1);
    buf[sizeof buf - 1] = 0;
    for (char *tok = strtok(buf, ","); tok; tok = strtok(nullptr, ","))
    {
        char *eq = strchr(tok, '=');
        if (!eq)
            continue;
        *eq = 0;
        double a = atof(eq + 1), b = 0;
        if (char *c = strchr(eq + 1, ':'))
            b = atof(c + 1);
        if (!strcmp(tok, "host")) r.gpu_host_path_gbps = a;
        else if (!strcmp(tok, "launch")) r.gpu_launch_us = a;
        else if (!strcmp(tok, "init")) r.gpu_init_ms = a;
        else if (!strcmp(tok, "memchr")) { r.cpu_memchr_gbps = a; if (b > 0) r.cpu_memchr_cap_gbps = b; }
        else if (!strcmp(tok, "simd")) { r.cpu_simd_gbps = a; if (b > 0) r.cpu_simd_cap_gbps = b; }
        else if (!strcmp(tok, "scalar")) { r.cpu_scalar_gbps = a; if (b > 0) r.cpu_scalar_cap_gbps = b; }
        else if (!strcmp(tok, "ac")) { r.cpu_ac_gbps = a; if (b > 0) r.cpu_ac_cap_gbps = b; }
    }
}
krep_gpu_cost_rates_t rates()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_rates_init)
    {
        g_rates = defaults();
        parse_env(g_rates);
        g_rates_init = true;
    }
    return g_rates;
}
} // namespace

namespace kg {
void cost_note_device_init(double ms) // the availability probe: what the first device call of this process took
{
    double none = -1.0;
    g_meas_init_ms.compare_exchange_strong(none, ms);
}
void cost_note_host_path(size_t bytes, double seconds) // an operator call that went through the staging ring
{
    if (bytes < ((size_t)64 << 20) || seconds <= 0)
        return; // small calls are launch-bound: they say nothing about the rate
    // The first large call of a process also pays for the plan / table build, the multi-GiB arena allocation, the pinned staging
    // ring and the copy-thread probe: several times below the real PCIe rate, and averaged in it would bias every later
    // krep_gpu_worthwhile() towards the CPU.  It is not a sample.  Setup can only make a call SLOWER, never faster:
    // a later sample above the running figure replaces it, one below it is averaged in.
    static std::atomic<int> n_seen{0};
    if (n_seen.fetch_add(1) == 0)
        return;
    const double gbps = (double)bytes / seconds / 1e9, old = g_meas_host_gbps.load();
    g_meas_host_gbps.store(old > 0 ? (gbps > old ? gbps : 0.5 * old + 0.5 * gbps) : gbps);
}
} // namespace kg

extern "C" void krep_gpu_get_cost_rates(krep_gpu_cost_rates_t *out)
{
    if (out)
        *out = rates();
}
extern "C" void krep_gpu_set_cost_rates(const krep_gpu_cost_rates_t *in)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (in)
    {
        g_rates = *in;
        g_rates_init = true;
    }
    else
        g_rates_init = false; // back to the defaults + environment
}

extern "C" int krep_gpu_cost_estimate(const search_params_t *p, size_t text_len, int cpu_threads, krep_gpu_cost_t *out)
{
    if (!p || !out)
        return 2;
    const krep_gpu_cost_rates_t r = rates();
    const krep_gpu_config_t c = kg::current_config();
    memset(out, 0, sizeof *out);
    // ---- the CPU side: which function, on how many threads
    int threads = cpu_threads;
    if (threads <= 0)
    { // search_file()'s own policy (krep.c:2729-2759)
        long cores = sysconf(_SC_NPROCESSORS_ONLN);
        if (cores < 1)
            cores = 1;
        const size_t by_size = text_len / ((size_t)4 << 20);
        threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)cores, by_size));
    }
    search_params_t q = *p; // legacy callers fill only pattern / pattern_len
    if (p->num_patterns == 1 && p->patterns && p->pattern_lens && p->patterns[0])
    {
        q.pattern = p->patterns[0];
        q.pattern_len = p->pattern_lens[0];
    }
    const int algo = p->use_regex ? KREP_RA_REGEX : kg::mirror_effective(kg::mirror_top(&q, c), &q, text_len);
    double per = r.cpu_scalar_gbps, cap = r.cpu_scalar_cap_gbps;
    switch (algo)
    {
    case KREP_RA_MEMCHR: per = r.cpu_memchr_gbps; cap = r.cpu_memchr_cap_gbps; break;
    case KREP_RA_MEMCHR_SHORT: case KREP_RA_SSE42: case KREP_RA_AVX2: case KREP_RA_AVX512: case KREP_RA_NEON:
        per = r.cpu_simd_gbps; cap = r.cpu_simd_cap_gbps; break;
    case KREP_RA_AHO_CORASICK:
    {
        // the automaton's size decides: aho_corasick.c keeps 256 child pointers per state (2 KiB); states <= sum of lengths
        double states = 1;
        if (p->pattern_lens)
            for (size_t i = 0; i < p->num_patterns; ++i)
                states += (double)p->pattern_lens[i];
        const double trie = states * 2048.0;
        per = r.cpu_ac_gbps * (trie > r.cpu_ac_cache_bytes ? std::pow(r.cpu_ac_cache_bytes / trie, r.cpu_ac_exponent) : 1.0);
        cap = r.cpu_ac_cap_gbps;
        break;
    }
    default: break;
    }
    const double cpu_gbps = std::max(1e-6, std::min(threads * per, cap));
    out->cpu_threads = threads;
    out->cpu_algo = algo;
    out->cpu_seconds = (double)text_len / (cpu_gbps * 1e9);
    // ---- the GPU side: the calibrated figures of this process where they exist
    const double init_ms = g_meas_init_ms.load();
    out->device_ready = init_ms >= 0;
    const double host = g_meas_host_gbps.load() > 0 ? g_meas_host_gbps.load() : r.gpu_host_path_gbps;
    out->gpu_seconds = (out->device_ready ? 0.0 : r.gpu_init_ms * 1e-3) + r.gpu_launch_us * 1e-6 + (double)text_len / (std::max(1e-6, host) * 1e9);
    out->gpu_host_path_gbps = host;
    return 0;
}

extern "C" int krep_gpu_worthwhile_ex(const search_params_t *p, size_t text_len, int cpu_threads)
{
    const krep_gpu_config_t c = kg::current_config();
    // size first: no device (and no HIP runtime) is touched for a small text
    if (text_len < c.min_text_bytes || kg::unsupported_reason(p, c) != nullptr)
        return 0;
    if (rates().enabled)
    {
        krep_gpu_cost_t e;
        if (krep_gpu_cost_estimate(p, text_len, cpu_threads, &e) == 0 && !(e.gpu_seconds < e.cpu_seconds))
            return 0; // the CPU function is expected to be faster: the device is not even asked
    }
    return kg::device_unusable(c.device) == nullptr ? 1 : 0;
}
extern "C" int krep_gpu_worthwhile(const search_params_t *p, size_t text_len) { return krep_gpu_worthwhile_ex(p, text_len, 0); }
