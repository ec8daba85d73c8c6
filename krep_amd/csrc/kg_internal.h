// kg_internal.h — internal (C++) interfaces between the translation units of libkrep_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/krep_gpu.h"
#include "kg_common.h"

namespace kg {

int fail(const char *fmt, ...); // records krep_gpu_last_error(), prints "krep-gpu: ..." and returns 2

// kg_literal.hip
hipError_t launch_literal(const LitArgs &a, uint32_t grid, hipStream_t st);

// kg_post.hip — greedy non-overlapping selection (simd_sse42_search / kmp_search family) on the
// ordered occurrence list, then -w, line bookkeeping and compaction
struct PostScratch
{
    uint64_t *d_occ = nullptr;  // all-occurrence records (2 x u64 each)
    uint64_t occ_cap = 0;
    uint8_t *d_keep = nullptr;
    uint64_t *d_blocksum = nullptr;
    uint64_t keep_cap = 0;
    unsigned long long *d_status = nullptr;
    uint64_t status_cap = 0;
};
void post_free(PostScratch &s);
int post_greedy_scan(PostScratch &s, LitArgs a, uint32_t grid, Counters *d_ctr, Counters *h_ctr, bool ww, bool lines,
                     uint64_t *d_pos, uint64_t want, hipStream_t st, hipEvent_t ev_end, uint64_t *total, uint64_t *nlines,
                     unsigned long long *summary);

// kg_ac.hip — multi-pattern scan
struct AcTables;
AcTables *ac_build(const search_params_t &sp, int device);
void ac_free(AcTables *t);
int ac_scan(AcTables *t, Counters *d_ctr, Counters *h_ctr, unsigned long long **d_status, size_t *status_cap, int num_cu,
            const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, size_t global_base,
            match_position_t *d_pos, uint64_t cap, bool ww, bool lines, bool track, size_t max_count, hipStream_t st,
            int time_it, hipEvent_t ev0, hipEvent_t ev1, krep_gpu_scan_out_t *out);

// kg_multi.hip — one process driving several devices (search_buffer(num_gpus > 1))
uint64_t multi_gpu_search(const search_params_t *params, const char *buf, size_t len, int num_gpus, match_result_t *out,
                          int *status);

} // namespace kg
