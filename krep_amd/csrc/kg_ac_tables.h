// kg_ac_tables.h — the device-resident tables of one dictionary (built on the host by kg_ac_build.hip, read by the scan drivers of
// kg_ac.hip): class-filter tables for LDS, the chain-compressed 4-gram buckets, exact bitmaps of the 1-3-byte patterns, the
// reversed trie's edge table, and what the scans have learnt about the dictionary's density.
#pragma once
#include <vector>
#include "kg_ac_common.h"

namespace kg {

struct AcTables
{
    int device = 0;
    u32 npat = 0, lmin = 0, lmax = 0;
    bool ci = false, has_nl = false, has_empty = false;
    u32 has1 = 0, has2 = 0, has3 = 0, has4 = 0;
    bool short_dup = false;     // a 1-3-byte pattern occurs more than once (the bitmaps cannot count copies)
    u32 *d_filterx20 = nullptr, *d_filterx19 = nullptr; // exact-class filter tables (2^20 bits; 2^19 for -c)
    u32 *d_filters20 = nullptr, *d_filters19 = nullptr; // the same for the stride-2 filter (nullptr: stride 2 not worth it)
    u32 *d_s1 = nullptr, *d_s2 = nullptr, *d_s3 = nullptr; // exact bitmaps of the 1-/2-/3-byte patterns
    uint2 *d_edges = nullptr;
    u32 emask = 0;
    u32 *d_copies = nullptr;
    u32 nnodes = 0;
    uint2 *d_gram4 = nullptr;
    uint4 *d_g4x = nullptr; // chain-compressed entries, same slots as d_gram4
    u32 g4mask = 0;
    u32 g4x_mode = 0, g4x_mask = 0, g4x_mul = 0;
    u32 *d_redo = nullptr; // emit mode: the overflowed units of the scan just made, as a list (ac_scan)
    uint64_t redo_cap = 0;
    u32 stage_cap = 16; // staged matches per unit (16, raised to 64 by a scan whose units overflowed; see ac_scan)
    AcTiny tiny{};      // ok: the dictionary runs in kg_ac_tiny.hip (every pattern <= 4 bytes, few of them)
    u32 tiny_dense_upt = 0;    // ... in its DENSE flavour, with tickets of this many units (0: not; set from the density a scan counted)
    bool tiny_dense_ok = true; // ... until a scan proves too dense for that as well (re-evaluated)
    bool tiny_fused_ok = true; // ... its records in ONE pass (FUSED) — until a scan proves too dense for the rings (re-evaluated)
    // a dictionary of 2..4 distinct single bytes: with records it is the one-pass single-byte scan with a needle SET
    // (kg_single.hip: records at their final index, nothing staged) — until a scan proves too dense for its largest rings
    u32 set_n = 0;
    uint8_t set_b[4] = {0, 0, 0, 0};
    bool set_ok = true;
    int set_shape = 0;
    // ---- anchored scan (kg_ac_anchor.hip): decided ONCE per dictionary, on the first text of >= 1 MiB it scans, from a 4-gram
    // class histogram of a sample of that text
    std::vector<std::vector<uint8_t>> pats_h; // the patterns (folded under -i), kept for that decision
    int anch_state = 0;                       // 0: not decided yet, 1: end grams stay (nothing to gain / not eligible), 2: anchored
    u32 *d_filtera20 = nullptr;               // pair-layout class table (2^20 bits) of the anchor grams
    uint4 *d_anch = nullptr;                  // buckets of two {exact anchor gram, 1 << 31 | offset mask}
    u32 anch_mask = 0, anch_mul = 0;
    u32 anch_five = 0;                        // the anchor table is indexed with five classes (6-byte windows)
    unsigned short *d_xlen = nullptr;         // stage 3's exact dictionary (AcArgs::xlen / xtab): dictionaries of 4..16-byte patterns
    uint4 *d_xtab = nullptr;
    u32 xmask = 0, xmul = 0;
    int anch_resamples = 0;                   // decisions repeated because a later text's measured candidates contradicted the estimate (at most kAnchResamples)
    double anch_measured = 0;                 // candidates per tested position the last general-kernel scan counted (Counters::candidates)
    double anch_rate0 = 0, anch_rate = 0;     // estimated candidates per tested position: end grams / anchor grams (diagnostic)
    u32 anch_moved = 0;                       // patterns whose anchor is not their end
};
// kg_ac_anchor.hip
constexpr int kAnchResamples = 3; // (a session that alternates kinds of text settles on the last decision after three repeats)
int ac_anchor_prepare(AcTables *t, const uint8_t *d_text, size_t text_len, size_t own_lo, size_t own_hi, hipStream_t st);
void ac_anchor_free(AcTables *t);

} // namespace kg
