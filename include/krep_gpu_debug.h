/* krep_gpu_debug.h — TEST HOOKS of libkrep_gpu.so.  Not part of the drop-in boundary (include/krep_gpu.h): nothing a host
 * program needs is declared here.  The symbols are exported by the same library so that the test-suite (tests/, through
 * krep_amd/engine.py) can force rare paths — tile shapes, starved grids, injected failures — and read counters that say which
 * kernel answered.  Round 6: moved out of krep_gpu.h. */
#ifndef KREP_GPU_DEBUG_H
#define KREP_GPU_DEBUG_H
#include "krep_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hook (host only, no GPU needed): the end-of-text replay of the block-structured -c paths
 * (krep_amd/csrc/kg_replay.h) run on host memory, so that the CPU test-suite can pin it against the oracle */
uint64_t krep_gpu_debug_replay_host(int krep_ref_algo, const void *text, size_t n, const void *pattern, uint32_t m,
                                    int whole_word, size_t cur, int open_line);
/* test hook: force the kernel tile shape (0 = auto, 1 = 32 KiB tiles, 4 = 128 KiB tiles) */
void krep_gpu_debug_force_rounds(int rounds);
/* test hook: staging records per scan unit (0 = auto); small values exercise the emit-mode re-scan */
void krep_gpu_debug_force_stage_cap(int records);
/* test hooks of the one-pass single-byte kernel (kg_single.hip): at most `blocks` workgroups (0 = auto) — a starved grid, as
 * on a shared or partitioned device; and how many of its scans handed over to the two-pass kernels (ring overflow on a
 * dense text, or the spin-limit safety net) since the process started */
void krep_gpu_debug_force_single_grid(int blocks);
uint64_t krep_gpu_debug_single_failovers(void);
/* ... and how many launches of that kernel there were (its 2..8-byte instantiations included: a plan takes them for records of a
 * dense short literal once a two-pass scan has counted the density) */
uint64_t krep_gpu_debug_single_launches(void);
/* test hook: launches of the tiny-dictionary kernel (kg_ac_tiny.hip: every pattern <= 4 bytes, compared in registers) since the
 * process started; $KREP_GPU_AC_NO_TINY=1 (read when a plan is built) keeps such dictionaries on the general kernel */
uint64_t krep_gpu_debug_tiny_launches(void);
/* ... and those of its one-pass record writer in the DENSE flavour (16-bit ring entries, tickets sized by the counted density) */
uint64_t krep_gpu_debug_tiny_dense_launches(void);
/* test hook: make the next operator calls fail at a chosen point — 0 off, 1 device allocation, 2 host->device copy,
 * 3 kernel launch, 4 device->host copy of the records.  Also read from $KREP_GPU_INJECT_FAILURE. */
void krep_gpu_debug_inject_failure(int kind);
/* test hook: pieces the multi-shard operators scanned AGAIN because their boundary record turned out different (rescans) and
 * end pieces that only re-ran the end-of-text replay (replays), since the process started */
void krep_gpu_debug_chain_fixups(uint64_t *rescans, uint64_t *replays);
/* test hook (host only, no GPU needed): the left fold of the boundary record exactly as the library applies it — a piece's
 * own contribution (its local_* fields) onto the record of the text in front of it — so that the CPU test-suite can pin the
 * chained-pieces algebra against the oracle (tests/test_replay_cpu.py) */
void krep_gpu_debug_fold_carry(const krep_gpu_seq_carry_t *in, const krep_gpu_seq_carry_t *piece, krep_gpu_seq_carry_t *out);
/* test hook: launches of the multi-pattern kernel's ANCHORED instantiation (kg_ac_anchor.hip: anchor grams chosen by rarity in a
 * sample of the text) since the process started; $KREP_GPU_AC_ANCHOR=1 anchors every eligible dictionary whatever the estimated
 * gain, $KREP_GPU_AC_NO_ANCHOR=1 none (both read when a dictionary meets its first text of >= 1 MiB) */
uint64_t krep_gpu_debug_anchored_launches(void);
/* test hook: launches of the LDS-DMA literal kernel (kg_literal_dma.hip: 2..8-byte patterns on 32-KiB units without -c) since the
 * process started; $KREP_GPU_LIT_NO_DMA=1 keeps such scans on the register-load kernel (kg_literal.hip) */
uint64_t krep_gpu_debug_literal_dma_launches(void);
/* what chose between that kernel and the register kernel for `plan`: has the text been sampled, is the kernel barred for it (the prefilter's
 * byte occurs in more than 35 % of its 1-KiB cells; $KREP_GPU_LIT_DMA_MAX_PASS), the share last measured (sample or launch).
 * $KREP_GPU_LIT_DMA_KEEP=1: no sample, no bar (measurement aid) */
int krep_gpu_debug_literal_dma_state(const krep_gpu_plan_t *plan, int *looked, int *barred, double *pass_rate);
/* test hook: launches of the run-length kernel (kg_runs.hip: the greedy families on a pattern of one repeated byte, count-only);
 * $KREP_GPU_NO_RUNS=1 keeps such scans on the list road */
uint64_t krep_gpu_debug_runs_launches(void);
/* a multi-pattern plan whose dictionary holds 1..3-byte patterns beside >= 8 longer ones: 0 not decided yet, 1 scanned as one dictionary, 2 split
 * (the long part anchored, the short part on its own, the record lists merged: kg_scan.hip scan_ac_split); $KREP_GPU_AC_NO_SPLIT=1: never */
int krep_gpu_debug_split_state(const krep_gpu_plan_t *plan);
/* what that decision was for `plan`: state 0 not taken yet / 1 end grams kept / 2 anchored; patterns moved off their end; the
 * estimated candidates per tested position with the end grams and with the anchors.  Returns 0, or 2 for a single-literal plan. */
int krep_gpu_debug_anchor_info(const krep_gpu_plan_t *plan, int *state, uint32_t *moved, double *rate_end_grams, double *rate_anchors);
/* what the last general-kernel scan of `plan` MEASURED (candidates per tested position, counted in the kernel), and how many times a
 * measurement that contradicted the estimate re-opened the decision (at most 3; $KREP_GPU_AC_NO_RESAMPLE=1: never) */
int krep_gpu_debug_anchor_measured(const krep_gpu_plan_t *plan, double *measured, int *resamples);

#ifdef __cplusplus
}
#endif
#endif
