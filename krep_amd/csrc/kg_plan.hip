// kg_plan.hip — krep_gpu_plan_t: everything a scan needs that depends only on the parameters (folded pattern words and masks,
// the multi-pattern tables of kg_ac.hip, counters, events, scratch), built once and reused (kg_ops.hip caches plans per
// device; the CLI calls its operator once per file with the same parameters, krep.c:1950).  Host logic only.
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
        pl->pat_ptrs.push_back(v.empty() ? "" : (const char *)v.data()); // (an empty pattern is a pattern: the reference returns 0
                                                                          //  for it, krep.c:1278, :1646 — not "no pattern")
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
        if (!kg::have_error())
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
    if (pl->ac_long) ac_free(pl->ac_long);
    if (pl->ac_short) ac_free(pl->ac_short);
    if (pl->d_split_rec) DBGFREE(hipFree(pl->d_split_rec));
    if (pl->d_nl_rec) DBGFREE(hipFree(pl->d_nl_rec));
    if (pl->d_nl_ln) DBGFREE(hipFree(pl->d_nl_ln));
    post_free(pl->post);
    post_free(pl->aux);
    delete pl;
}
extern "C" int krep_gpu_plan_ref_algo(const krep_gpu_plan_t *pl) { return pl ? pl->ref_algo : KREP_RA_NONE; }
extern "C" int krep_gpu_debug_split_state(const krep_gpu_plan_t *pl) { return pl ? pl->ac_split : -1; }
extern "C" int krep_gpu_debug_literal_dma_state(const krep_gpu_plan_t *pl, int *looked, int *barred, double *pass_rate)
{
    if (!pl)
        return 2;
    if (looked) *looked = pl->dma_look_done;
    if (barred) *barred = pl->dma_off;
    if (pass_rate) *pass_rate = pl->dma_pass_rate;
    return 0;
}
extern "C" int krep_gpu_debug_anchor_measured(const krep_gpu_plan_t *pl, double *measured, int *resamples)
{
    if (!pl || !pl->ac)
        return 2;
    if (measured) *measured = pl->ac->anch_measured;
    if (resamples) *resamples = pl->ac->anch_resamples;
    return 0;
}
extern "C" int krep_gpu_debug_anchor_info(const krep_gpu_plan_t *pl, int *state, uint32_t *moved, double *rate_end_grams, double *rate_anchors)
{
    if (!pl || !pl->ac)
        return 2;
    if (state) *state = pl->ac->anch_state;
    if (moved) *moved = pl->ac->anch_moved;
    if (rate_end_grams) *rate_end_grams = pl->ac->anch_rate0;
    if (rate_anchors) *rate_anchors = pl->ac->anch_rate;
    return 0;
}
