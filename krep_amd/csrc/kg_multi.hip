// kg_multi.hip — search_buffer(num_gpus > 1): one host buffer sharded by contiguous chunk over the
// devices of THIS process (the CLI drop-in path; the one-process-per-GPU RCCL path is bench.py +
// krep_amd/shard.py).  This replaces the reference's CPU chunk decomposition
// (krep.c:2816-2905: chunk = ceil(N/T), overlap = max_pattern_len-1, concatenate, qsort) with
// START-OFFSET OWNERSHIP: shard g reports a match iff its start lies in [lo_g, hi_g); it sees
// `ctx` bytes of context on both sides so that matches straddling hi_g complete, -w sees both
// neighbours and the multi-pattern -c (owned by END index) sees matches that begin before lo_g.
// Unlike the reference's chunking this reproduces the SINGLE-chunk result exactly: no duplicate
// multi-pattern matches in the overlap and no double-counted lines.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_internal.h"

namespace kg {

namespace {
struct Shard
{
    size_t lo = 0, hi = 0;     // owned window in the global text
    size_t b0 = 0, b1 = 0;     // bytes staged: [b0, b1) superset of [lo, hi)
    int device = 0;
    int rc = 0;
    krep_gpu_scan_out_t out{};
    std::vector<match_position_t> recs;
    std::string err;
};

void run_shard(Shard *sh, const search_params_t *params, const char *buf, size_t len, bool want_pos, int only_matching)
{
    sh->rc = 2;
    search_params_t local = *params;
    local.max_count = SIZE_MAX; // prefix-ordered truncation happens after the shards are merged
    krep_gpu_plan_t *pl = krep_gpu_plan_create(&local, only_matching, sh->device);
    if (!pl)
    {
        sh->err = krep_gpu_last_error();
        return;
    }
    uint8_t *d_text = nullptr;
    match_position_t *d_pos = nullptr;
    do
    {
        const size_t nb = sh->b1 - sh->b0;
        if (hipSetDevice(sh->device) != hipSuccess || hipMalloc(&d_text, nb + 64) != hipSuccess ||
            stage_to_device(d_text, buf + sh->b0, nb, sh->device) != 0)
        {
            sh->err = "shard staging failed";
            break;
        }
        // the global text ends at `len`: the last shard's buffer ends there too, so "end of text" is exact
        (void)len;
        uint64_t cap = want_pos ? std::max<uint64_t>(1u << 16, nb / 64) : 0;
        int rc = 0;
        for (int attempt = 0; attempt < 2; ++attempt)
        {
            if (cap)
            {
                if (d_pos) (void)hipFree(d_pos);
                d_pos = nullptr;
                if (hipMalloc(&d_pos, cap * sizeof(match_position_t)) != hipSuccess)
                {
                    rc = 2;
                    sh->err = "position buffer allocation failed";
                    break;
                }
            }
            rc = krep_gpu_scan_device(pl, d_text, nb, sh->lo - sh->b0, sh->hi - sh->b0, sh->b0, d_pos, cap, nullptr, 0, &sh->out);
            if (rc || !sh->out.overflow)
                break;
            cap = sh->out.total_matches + 1;
        }
        if (rc)
        {
            if (sh->err.empty())
                sh->err = krep_gpu_last_error();
            break;
        }
        if (want_pos && sh->out.stored)
        {
            sh->recs.resize(sh->out.stored);
            if (hipMemcpy(sh->recs.data(), d_pos, sh->out.stored * sizeof(match_position_t), hipMemcpyDeviceToHost) != hipSuccess)
            {
                sh->err = "D2H copy failed";
                break;
            }
        }
        sh->rc = 0;
    } while (0);
    if (d_text) (void)hipFree(d_text);
    if (d_pos) (void)hipFree(d_pos);
    krep_gpu_plan_destroy(pl);
    stage_release();
}
} // namespace

uint64_t multi_gpu_search(const search_params_t *params, const char *buf, size_t len, int num_gpus, match_result_t *out,
                          int *status)
{
    if (status)
        *status = 2;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        fail("no HIP device available (this library has no CPU fallback)");
        return 0;
    }
    size_t lmax = 1;
    for (size_t i = 0; i < params->num_patterns; ++i)
        lmax = std::max(lmax, params->pattern_lens[i]);
    const int algo = krep_gpu_mirror_select(params, len);
    // the greedy family couples neighbouring occurrences (a cluster may straddle a shard boundary): keep
    // bordered patterns of that family on one shard
    bool single = false;
    if ((algo == KREP_RA_SSE42 || algo == KREP_RA_KMP) && params->num_patterns == 1)
    {
        const char *p = params->pattern;
        const size_t m = params->pattern_len;
        for (size_t k = 1; k < m && !single; ++k)
        {
            bool same = true;
            for (size_t j = 0; j + k < m && same; ++j)
            {
                unsigned char x = (unsigned char)p[j], y = (unsigned char)p[j + k];
                if (!params->case_sensitive)
                {
                    if (x >= 'A' && x <= 'Z') x += 32;
                    if (y >= 'A' && y <= 'Z') y += 32;
                }
                same = x == y;
            }
            single = same;
        }
    }
    int G = single ? 1 : std::max(1, num_gpus);
    if ((size_t)G > len / 4096 + 1)
        G = (int)(len / 4096 + 1);
    const size_t ctx = lmax + 1;
    std::vector<Shard> sh((size_t)G);
    const size_t chunk = (len + (size_t)G - 1) / (size_t)G;
    for (int g = 0; g < G; ++g)
    {
        sh[g].lo = std::min(len, (size_t)g * chunk);
        sh[g].hi = std::min(len, sh[g].lo + chunk);
        sh[g].b0 = sh[g].lo > ctx ? sh[g].lo - ctx : 0;
        sh[g].b1 = std::min(len, sh[g].hi + ctx);
        sh[g].device = g % ndev;
    }
    const bool want_pos = params->track_positions && out != nullptr && !params->count_lines_mode;
    // A shard buffer that stops short of the global end must not treat its end as "end of text" for -w.
    // krep_gpu_scan_device() takes text_len = staged bytes; with ctx = Lmax+1 bytes of right context every
    // owned match has its right neighbour inside the buffer, so the only "end of text" it can see is real.
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
        th.emplace_back(run_shard, &sh[g], params, buf, len, want_pos, current_only_matching());
    for (auto &t : th)
        t.join();
    for (int g = 0; g < G; ++g)
        if (sh[g].rc)
        {
            fail("shard %d on device %d failed: %s", g, sh[g].device, sh[g].err.c_str());
            return 0;
        }
    uint64_t total = 0;
    std::vector<krep_gpu_scan_out_t> outs;
    for (auto &s : sh)
    {
        total += s.out.total_matches;
        outs.push_back(s.out);
    }
    const uint64_t lines = krep_gpu_combine_line_counts(outs.data(), G);
    const size_t maxc = params->max_count;
    uint64_t ret;
    if (maxc == 0)
        ret = 0;
    else
        ret = std::min<uint64_t>(params->count_lines_mode ? lines : total, maxc);
    if (want_pos && ret)
    {
        std::vector<match_position_t> all;
        all.reserve((size_t)total);
        if (params->num_patterns > 1)
        {
            // each shard list is in the reference's (end, start) order; merge them into the global order
            for (auto &s : sh)
                all.insert(all.end(), s.recs.begin(), s.recs.end());
            std::stable_sort(all.begin(), all.end(), [](const match_position_t &a, const match_position_t &b) {
                return a.end_offset != b.end_offset ? a.end_offset < b.end_offset : a.start_offset < b.start_offset;
            });
        }
        else
            for (auto &s : sh)
                all.insert(all.end(), s.recs.begin(), s.recs.end());
        uint64_t n = std::min<uint64_t>(all.size(), ret);
        if (algo == KREP_RA_KMP && maxc != SIZE_MAX && all.size() > maxc)
            n = maxc + 1; // krep.c:1717-1724
        if (algo == KREP_RA_MEMCHR && maxc != SIZE_MAX && maxc % 4096 == 0 && all.size() > maxc)
        { // memchr_search's final-flush order (krep.c:3976-3991, :4026-4038)
            const uint64_t f = maxc - 4096;
            const match_position_t extra = all[maxc];
            memmove(&all[f + 1], &all[f], 4095 * sizeof(match_position_t));
            all[f] = extra;
        }
        if (params->num_patterns > 1 && current_result_order()) // the formatter's order (krep.c:420-434)
            std::sort(all.begin(), all.begin() + n, [](const match_position_t &a, const match_position_t &b) {
                return a.start_offset != b.start_offset ? a.start_offset < b.start_offset : a.end_offset < b.end_offset;
            });
        const uint64_t need = out->count + n;
        if (need > out->capacity || !out->positions)
        {
            uint64_t cap = out->capacity ? out->capacity : 16;
            while (cap < need)
                cap *= 2;
            match_position_t *np = (match_position_t *)realloc(out->positions, cap * sizeof(match_position_t));
            if (!np)
            {
                fail("out of memory growing match_result_t");
                return 0;
            }
            out->positions = np;
            out->capacity = cap;
        }
        memcpy(out->positions + out->count, all.data(), n * sizeof(match_position_t));
        out->count += n;
    }
    if (status)
        *status = 0;
    return ret;
}

} // namespace kg
