// kg_internal.h — internal (C++) interfaces between the translation units of libkrep_gpu.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <vector>

#include "../../include/krep_gpu.h"
#include "../../include/krep_gpu_debug.h" // (test hooks, exported by the same library)
#include "kg_common.h"

namespace kg {

int fail(const char *fmt, ...); // records krep_gpu_last_error(), prints "krep-gpu: ..." and returns 2
const char *device_unusable(int device); // NULL: a gfx950 device this code object runs on; else why not
bool inject(int kind);                   // test hook: is failure `kind` being injected (krep_gpu_debug_inject_failure)

// kg_literal.hip
hipError_t launch_literal(const LitArgs &a, uint32_t num_cu, hipStream_t st); // grid = resident blocks of the variant x CUs
// kg_runs.hip: the greedy families on a pattern of one repeated byte, counted without a list (1: a run too long for its look-back)
int runs_count_greedy(const uint8_t *d_text, uint64_t text_len, uint64_t lo, uint64_t hi, uint32_t m, uint8_t byte, bool ci, int num_cu,
                      unsigned long long *d_slots, unsigned long long *h_slots, hipStream_t st, uint64_t *total, uint64_t *end_p1);
bool literal_dma_eligible(const LitArgs &a);                                       // kg_literal_dma.hip: 2..8-byte patterns, 32-KiB units, no -c
hipError_t launch_literal_dma(const LitArgs &a, uint32_t num_cu, hipStream_t st);
hipError_t launch_dma_byte_look(const uint8_t *text, uint64_t lo, uint32_t n_cells, uint32_t prefilter, bool ci, unsigned long long *out, hipStream_t st); // 1-KiB cells of a sample that hold the prefilter's byte
extern std::atomic<uint64_t> g_runs_launches;                                      // launches of run_count_kernel (kg_runs.hip, test hook)
extern std::atomic<uint64_t> g_lit_dma_launches;                                   // launches of lit_scan_dma (test hook)

// kg_single.hip — single byte with records in one pass (counts resolved by one wave, records written a ticket later)
uint64_t single_fused_tickets(uint64_t n_units, int shape);
uint64_t single_fused_scratch_words(uint64_t n_tickets);
constexpr int kFusedShapeMax = 5;
double single_fused_max_density(int shape); // hits per byte a shape's rings are sure to hold (shapes 0..5: ~1.2 / 3.7 / 5 / 7.5 / 10 / 20 %)
hipError_t launch_single_fused(const LitArgs &a, unsigned long long *d_agg, unsigned long long *d_pref, uint64_t n_tickets,
                               uint32_t num_cu, int shape, hipStream_t st);

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
    unsigned long long *d_tk = nullptr;       // kg_single.hip: per-ticket hit counts | their exclusive prefixes
    uint64_t tk_cap = 0;                      // in tickets
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
// kg_greedy.hip — the sequential match-set families, walked cluster by cluster on the ordered list in s.d_occ
constexpr uint32_t kWalkGreedy = 0; // simd_sse42_search / kmp_search (and BMH under -o): greedy non-overlapping occurrences
constexpr uint32_t kWalkShortO = 1; // memchr_short_search under -o: first-byte candidates, m skipped after a failed one too
constexpr uint32_t kWalkShortOLines = 2; // ... with -c as well (krep.c:4449-4470): an accepted match counts its line and sends the
                                         // walk to the next line start — every visited accepted candidate is one counted line
struct WalkSpec
{
    uint32_t mode, m;
    bool ww, lines, ci;
    uint8_t b1, b2; // kWalkShortO: (folded) pattern bytes 1 and 2
};
int post_walk(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const WalkSpec &ws,
              uint64_t n_occ, uint64_t *d_pos, uint64_t want, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
              uint64_t *total, uint64_t *nlines, uint64_t *resume);

constexpr uint32_t kNlWalkSse42 = 0, kNlWalkKmp = 1; // post_nlwalk modes
int post_nlwalk(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint32_t mode, uint32_t m, uint32_t k0, bool ww, bool om,
                uint64_t maxc, uint64_t n_occ, const uint64_t *d_lineno, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
                uint64_t *count, uint64_t global_base, uint64_t global_len, uint64_t line_off, uint64_t cp_in, uint64_t seen_in,
                uint64_t *cp_out, uint64_t *seen_out);

// kg_tail.hip — end-of-text replay of the block-structured -c paths (kg_replay.h)
struct ReplayIn;
int tail_last_hit(const unsigned long long *d_unitinfo, uint64_t limit_units, unsigned long long *d_slot,
                  unsigned long long *h_slot, hipStream_t st, uint64_t *unit_plus1);
int tail_find_next_newline(const uint8_t *d_text, uint64_t from, uint64_t n, unsigned long long *d_slot,
                           unsigned long long *h_slot, hipStream_t st, uint64_t *pos);
int tail_find_prev_newline(const uint8_t *d_text, uint64_t before, unsigned long long *d_slot, unsigned long long *h_slot,
                           hipStream_t st, uint64_t *pos_plus1);
int tail_count_changes(const uint64_t *d_v, uint64_t n, unsigned long long *d_slot, unsigned long long *h_slot, hipStream_t st,
                       uint64_t *changes);
int tail_count_line_gaps(const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const uint64_t *d_rec, uint64_t n, unsigned long long *d_slot,
                         unsigned long long *h_slot, hipStream_t st, uint64_t *lines);
int tail_launch_line_gaps(const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const uint64_t *d_rec, const unsigned long long *d_n,
                          const unsigned long long *d_skip_if, uint64_t cap, unsigned long long *d_out, hipStream_t st);
// '\n' bytes of d_text[lo, hi) -> *count (kg_tail.hip; results through the plan's pinned counter block)
int tail_count_newlines(const uint8_t *d_text, uint64_t lo, uint64_t hi, unsigned long long *d_slot, unsigned long long *h_slot,
                        hipStream_t st, uint64_t *count);
int tail_run_replay(const ReplayIn &r, unsigned long long *d_slot, unsigned long long *h_slot, hipStream_t st, uint64_t *lines);

// kg_ac.hip — multi-pattern scan
int order_records(match_position_t *d_positions, uint64_t n, size_t max_offset, hipStream_t st, bool by_end); // kg_format.hip: stable radix sort of a record list by start / by end
struct AcTables;
AcTables *ac_build(const search_params_t &sp, int device);
void ac_free(AcTables *t);
bool ac_counts_lines_in_registers(const AcTables *t); // a tiny dictionary: its in-kernel -c road keeps nothing in LDS (kg_ac_tiny.hip)
int ac_scan(AcTables *t, Counters *d_ctr, Counters *h_ctr, PostScratch &post, int num_cu,
            const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, size_t global_base,
            match_position_t *d_pos, uint64_t cap, bool ww, bool lines, bool track, size_t max_count, hipStream_t st,
            int time_it, hipEvent_t ev0, hipEvent_t ev1, krep_gpu_scan_out_t *out, int list_mode = 0);

// kg_config.hip — test hooks shared with the scan drivers
extern std::atomic<int> g_force_rounds, g_force_stage_cap;   // krep_gpu_debug_force_rounds / _stage_cap
extern std::atomic<uint64_t> g_tiny_launches;                // launches of ac_tiny_kernel (kg_ac_tiny.hip)
extern std::atomic<uint64_t> g_tiny_dense_launches;          // ... of its DENSE one-pass flavour
extern std::atomic<uint64_t> g_ac_anchored_launches;         // launches of ac_scan_kernel<.., ANCH> (kg_ac.hip, kg_ac_anchor.hip)
extern std::atomic<uint64_t> g_fused1_launches;              // launches of single_fused (kg_single.hip)
extern std::atomic<uint64_t> g_fused1_failovers;             // one-pass single-byte scans that handed over to the two-pass kernels
inline uint8_t lo8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; } // lower_table, krep.c:125-134
inline bool pattern_has_border(const uint8_t *p, size_t m)   // a proper prefix is also a suffix: all occurrences != greedy
{
    for (size_t k = 1; k < m; ++k)
        if (memcmp(p, p + k, m - k) == 0)
            return true;
    return false;
}

// kg_config.hip / kg_mirror.hip — configuration (explicit; see krep_gpu_config_t), selector mirror, result container
krep_gpu_config_t current_config(); // the calling thread's override, else the process-wide defaults
int mirror_top(const search_params_t *p, const krep_gpu_config_t &c);
int mirror_effective(int top, const search_params_t *p, size_t text_len);
const char *unsupported_reason(const search_params_t *p, const krep_gpu_config_t &c); // NULL = accelerated
constexpr int kSplitWhole = 0, kSplitPieces = 1, kSplitChain = 2; // enum krep_gpu_split (include/krep_gpu.h)
int split_mode(const search_params_t *p, const krep_gpu_config_t &c, size_t text_len);
bool shardable(const search_params_t *p, const krep_gpu_config_t &c, size_t text_len); // split_mode != whole
// a piece's own contribution (the local_* fields of `piece`) folded onto the record of the text in front of it
krep_gpu_seq_carry_t fold_carry(const krep_gpu_seq_carry_t &in, const krep_gpu_seq_carry_t &piece);
bool result_reserve(match_result_t *r, uint64_t extra);
bool have_error();

// kg_format.hip
void format_release(); // frees the formatter scratch of every device

// kg_comm.hip — the RCCL all-reduce of the per-shard counters (one process driving several devices)
int allreduce_across_devices(const std::vector<int> &devs, std::vector<std::vector<unsigned long long>> &vecs);
int comm_warmup(const std::vector<int> &devs);       // creates the communicator of a device list ahead of the first search
int comm_clique_ranks(const std::vector<int> &devs); // ranks of the cached communicator over `devs` (0: none)

// kg_cost.hip — krep_gpu_worthwhile()'s cost model: what this process has measured about its own GPU side
void cost_note_device_init(double ms);                  // the first device call of the process (availability probe)
void cost_note_host_path(size_t bytes, double seconds); // a host-buffer operator call that went through the staging ring

// kg_ops.hip — host-buffer side
void memchr_batch_quirk(match_position_t *recs, uint64_t have, size_t maxc);

} // namespace kg
