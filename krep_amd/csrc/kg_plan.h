// kg_plan.h — the plan object behind krep_gpu_plan_t (shared by kg_plan.hip, kg_scan.hip and kg_ops.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_internal.h"

struct krep_gpu_plan
{
    krep_gpu_config_t cfg{};     // the configuration this plan was compiled for (explicit; no global is read after creation)
    int device = 0;
    int ref_algo = KREP_RA_NONE; // top-level selection (delegation resolved per text length)
    const char *unsupported = nullptr; // non-NULL: why scans with this plan are refused (kg::unsupported_reason)
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
    unsigned long long *d_pat_chunks = nullptr; // see LitArgs::pat_chunks
    uint32_t n_chunks = 0;
    // workspace
    kg::Counters *d_ctr = nullptr, *h_ctr = nullptr;
    int num_cu = 256;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // multi-pattern
    kg::AcTables *ac = nullptr;
    // a dictionary with 1..3-byte patterns beside many longer ones, on a text where the longer ones gain from anchors (kg_scan.hip
    // scan_ac_split, round 6): the two parts as dictionaries of their own, scanned one after the other, their record lists merged
    kg::AcTables *ac_long = nullptr, *ac_short = nullptr;
    int ac_split = 0; // 0: not decided, 1: the whole dictionary in one scan, 2: split
    match_position_t *d_split_rec = nullptr; // under max_count: both parts' first max_count records, merged here before the first max_count go out
    uint64_t split_cap = 0;
    bool ac_has_newline = false; // some pattern of the set contains '\n' (-c then counts emission-order line changes)
    kg::PostScratch post; // ordering post-pass / sequential-family scratch of the main pass
    uint32_t sparse_cap = 16; // staging entries per 32 KiB unit of the sparse literal kinds: 16, raised to 64 after a scan whose
                              // units overflowed (see lit_pass); one 32-byte slot per unit keeps the store stream dense
    uint32_t ac_cap = 16;     // the same for the multi-pattern scan (16 KiB units)
    // lit_pass, the LDS-DMA literal kernel (kg_literal_dma.hip): chosen by what the TEXT holds — the share of 1-KiB cells its prefilter lets
    // through, sampled before the plan's first eligible launch and counted by every launch of the kernel itself
    bool dma_look_done = false;   // the sample has been taken
    bool dma_off = false;         // too many cells pass: the register kernel takes this plan's scans (looked at again when the text changes)
    const void *dma_off_text = nullptr; // ... the text that said so
    size_t dma_off_len = 0;
    double dma_pass_rate = 0;     // last share measured (sample or launch)
    bool first_look_done = false; // lit_pass: the density of the plan's first large text has been sampled (road / ring shape / slot from it)
    bool fused1_ok = true;    // single byte with records: the one-pass kernel (kg_single.hip) until a scan proves too dense for it
    bool fusedk_on = false;   // a 2..8-byte literal with records: the same kernel's MULTI instantiations, switched on by a two-pass scan that
    bool fusedk_never = false; // ... counted a density its staging slots do not hold (lit_pass); never again once a ring overflowed beyond the last shape
    int fusedk_shape = 0;
    int fused1_shape = 0;     // ... its ticket / ring shape (0..5: <= ~1.2 / 3.7 / 5 / 7.5 / 10 / 20 % hits, kg_single.hip), raised by the density a scan counted
    match_position_t *d_nl_rec = nullptr; // multi-pattern -c with a newline inside a pattern (scan_ac_newline_lines): the
    uint64_t *d_nl_ln = nullptr;          // ordered record list and the line number of every start; grow-only
    uint64_t nl_cap = 0;
    bool lines_list_off = false; // multi-pattern -c: a scan found the text too dense for the record-list road (kg_scan.hip)
    kg::PostScratch aux;  // small auxiliary passes that must not disturb `post` (end-of-text replay)
    search_params_t sp{}; // shallow copy with patterns pointing into `pats`
    std::vector<const char *> pat_ptrs;
    std::vector<size_t> pat_lens;
};
