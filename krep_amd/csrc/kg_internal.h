// kg_internal.h — internal (C++) interfaces between the translation units of libkrep_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/krep_gpu.h"
#include "kg_common.h"

namespace kg {

int fail(const char *fmt, ...); // records krep_gpu_last_error(), prints "krep-gpu: ..." and returns 2

// kg_literal.hip
hipError_t launch_literal(const LitArgs &a, uint32_t grid, hipStream_t st);

// kg_post.hip — ordering post-pass shared by the literal and Aho-Corasick scans
struct PostScratch
{
    unsigned long long *d_unitinfo = nullptr; // [units] info words written by the scan kernel
    unsigned long long *d_offsets = nullptr;  // [units] exclusive global index of each unit's first hit
    unsigned long long *d_blk = nullptr;      // [2 * units/1024] block sums / line carries
    uint64_t units_cap = 0;
    unsigned long long *d_stage = nullptr;    // [units * stage_cap] staged, unit-ordered hit words
    uint64_t stage_cap_words = 0;
    uint64_t *d_occ = nullptr;                // all-occurrence records for the greedy (family N) filter
    uint64_t occ_cap = 0;                     // in records
    uint8_t *d_keep = nullptr;                // greedy survivor flags
    unsigned long long *d_gblk = nullptr;     // compaction block counts
    unsigned long long *d_surv = nullptr;     // compacted survivors (line counting)
    uint64_t keep_cap = 0;
};
void post_free(PostScratch &s);
int post_reserve(PostScratch &s, uint64_t n_units, uint64_t stage_words);
// K1..K4 on `st`: offsets, distinct-line total (ctr->lines), line summary (ctr->summary), gather into d_pos
// fixed_len > 0: 16-bit unit-relative staging, record start = origin + unit * unit_bytes + offset
int post_order(PostScratch &s, uint64_t n_units, uint32_t stage_cap, uint32_t fixed_len, uint64_t origin, uint64_t unit_bytes,
               bool want_lines, uint64_t *d_pos, uint64_t pos_cap, Counters *d_ctr, int num_cu, hipStream_t st);
int post_offsets_pass(PostScratch &s, uint64_t n_units, bool want_lines, Counters *d_ctr, hipStream_t st);
int post_gather_pass(PostScratch &s, uint64_t n_units, uint32_t stage_cap, uint32_t fixed_len, uint64_t origin,
                     uint64_t unit_bytes, uint64_t *d_pos, uint64_t pos_cap, int num_cu, hipStream_t st);
// greedy non-overlapping selection (simd_sse42_search / kmp_search family) on the ordered occurrence list
int post_greedy(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint64_t global_base, uint32_t m, bool ww,
                bool lines, uint64_t n_occ, uint64_t *d_pos, uint64_t want, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
                uint64_t *total, uint64_t *nlines);

// kg_ac.hip — multi-pattern scan
struct AcTables;
AcTables *ac_build(const search_params_t &sp, int device);
void ac_free(AcTables *t);
int ac_scan(AcTables *t, Counters *d_ctr, Counters *h_ctr, PostScratch &post, int num_cu,
            const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, size_t global_base,
            match_position_t *d_pos, uint64_t cap, bool ww, bool lines, bool track, size_t max_count, hipStream_t st,
            int time_it, hipEvent_t ev0, hipEvent_t ev1, krep_gpu_scan_out_t *out);

int current_only_matching();
int current_result_order(); // krep_gpu_set_result_order(): 1 = hand records back in (start, end) order
int stage_to_device(uint8_t *d_dst, const char *src, size_t len, int device); // kg_host.hip: pinned double-buffered H2D
void stage_release(); // the mirrored file-static `only_matching` (krep.c:117)

// kg_multi.hip — one process driving several devices (search_buffer(num_gpus > 1))
uint64_t multi_gpu_search(const search_params_t *params, const char *buf, size_t len, int num_gpus, match_result_t *out,
                          int *status);

} // namespace kg
