// kg_common.h — shared host/device definitions of the krep-gpu scan engine (gfx950 only).
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace kg {

// ---- tile geometry ------------------------------------------------------------------------
// A workgroup = 256 threads = 4 wave64.  One "cell" = what one wave reads with ONE
// global_load_dwordx4: 64 lanes x 16 B = 1 KiB contiguous (fully coalesced).  A wave owns
// CELLS consecutive cells (all loads issued back-to-back => CELLS x 16 B in flight per lane),
// = one 8 KiB load ROUND.  A chain UNIT is what one wave scans before it publishes its aggregate:
// `rounds` rounds (1 for small inputs, kRoundsBig = 4 -> 32 KiB otherwise), hit masks kept in
// registers.  A workgroup TILE = 4 units (128 KiB) and is what one ticket hands out, so the single
// ticket word sees 32 GiB / 128 KiB = 262144 fetch-adds per scan (~46/us at 6 TB/s; one word
// saturates at ~88/us, MI355X_MICROARCH.md "dequeue").
constexpr int      kWave        = 64;
constexpr int      kBlock       = 256;
constexpr int      kWavesPerBlk = kBlock / kWave;
constexpr int      kCells       = 8;
constexpr uint32_t kCellBytes   = kWave * 16;                               // 1 KiB
constexpr uint32_t kSegBytes    = kCellBytes * kCells;                      // per wave: 8 KiB
constexpr int      kRoundsBig   = 4;                                        // 32 KiB per wave unit, 128 KiB per tile

// ---- per-unit info word written by the scan kernel, consumed by the post-pass (kg_post.hip) ---
//   [63]    has_nl : the unit's owned bytes contain a '\n'
//   [62]    head   : a match starts before the first '\n' (== any match when !has_nl)
//   [61]    tail   : a match starts after the last '\n'   (== any match when !has_nl)
//   [60:32] lines  : distinct lines holding a match start, counted as if no line were open on entry
//   [31:0]  count  : matches starting in the unit
constexpr uint64_t kLnNl = 1ull << 63, kLnHead = 1ull << 62, kLnTail = 1ull << 61;
constexpr int      kUiLineShift = 32;
constexpr uint64_t kUiLineMask = (1ull << 29) - 1;
constexpr uint64_t kUiCountMask = 0xffffffffull;

// flags
enum : uint32_t {
    F_CI        = 1u << 0,   // fold A-Z (C-locale lower_table, krep.c:125-134)
    F_WW        = 1u << 1,   // -w: is_whole_word_match (krep.h:312-319)
    F_POS       = 1u << 2,   // write ordered match_position_t records
    F_LINES     = 1u << 3,   // line bookkeeping (count distinct lines holding a match start)
};

// Counters block in device memory (zeroed before each launch, read back after).
struct Counters {
    unsigned long long total;        // matches (after -w / ownership filtering)
    unsigned long long lines;        // distinct lines with a match start (F_LINES)
    unsigned long long ticket;       // dynamic tile id dispenser
    unsigned long long summary;      // line bits (kLnNl|kLnHead|kLnTail) of the whole owned window
    unsigned long long overflow_units; // units whose hit count exceeded the staging capacity
    unsigned long long max_unit_count; // largest per-unit hit count seen
    unsigned long long pad[4];       // scratch slots of the small tail kernels (pad[0]: one value; pad[1..3]: the newline-pattern walk)
    unsigned long long candidates;   // multi-pattern scan: tested positions its LDS filter passed (what its time follows; round 6)
};

// Parameters of a literal scan launch.
struct LitArgs {
    const uint8_t *text;          // device pointer to byte 0 of the buffer
    uint64_t text_len;            // readable bytes
    uint64_t own_lo, own_hi;      // matches are reported iff own_lo <= start < own_hi
    uint64_t anchor;              // tile grid origin (<= own_lo, 16-byte aligned)
    uint64_t num_tiles;           // workgroup tiles of 4 x rounds x 8 KiB
    uint32_t rounds;              // 1 or kRoundsBig
    uint64_t global_base;         // added to reported offsets
    uint64_t excl_lo, excl_hi;    // starts in [excl_lo, excl_hi) are NOT reported (simd_avx512_search's unexamined block)
    uint64_t ww_exempt_left;      // a start offset whose LEFT neighbour test is skipped (AVX tail quirk) or ~0
    uint32_t m;                   // pattern length (1..1024)
    uint32_t flags;
    uint32_t p0, p1, k0, k1;      // first <=8 pattern bytes (folded when F_CI) and their byte masks
    uint32_t p2, p3, k2, k3, l2, l3; // pattern bytes 8..15 (m = 9..16 verify in registers), their byte and letter masks
    uint32_t l0, l1;              // F_CI: 0x20 in the byte lanes where the (folded) pattern holds a letter — (x | l) == p
    uint32_t set_n;               // kg_single.hip: a byte SET of set_n (2..4) needles instead of the one byte p0 (0 = no set)
    uint32_t set_p[4], set_l[4];  // ... their splats and (F_CI) letter masks
                                  // is the C-locale case-insensitive compare of that byte (one OR instead of folding the text)
    const uint8_t *pat;           // device copy of the (folded) pattern, for m > 8
    const unsigned long long *pat_chunks; // m > 8: (folded) pattern bytes [q_k, q_k + 8), q_k = min(8 + 8k, m - 8), 8-byte aligned
    uint32_t n_chunks;            //   ceil((m - 8) / 8) of them: scalar loads for the verify (an unaligned pattern read is a vector load);
                                  //   followed by n_chunks letter masks (0x20 where the folded pattern holds a letter, F_CI)
    unsigned long long *unitinfo; // [num_tiles * 4] per-unit info words (F_POS | F_LINES)
    Counters *ctr;
    uint64_t *stage;              // [num_tiles * 4 * stage_cap] ordered start offsets per unit (F_POS, scan mode)
    uint32_t stage_cap;           // staging records per unit
    uint32_t upt;                 // units per wave ticket (1..8, by text size; divides 256)
    uint32_t emit_mode;           // 0: scan (count + stage); 1: re-scan the overflowed units and write records
    const uint64_t *offsets;      // [units] exclusive global index of each unit's first match (emit mode)
    uint64_t *positions;          // match_position_t records (2 x u64) or nullptr
    uint64_t pos_cap;             // min(capacity, max_count)
    uint32_t prefilter;           // kg_literal_dma.hip: the pattern's first byte in every byte lane (| 0x20 under F_CI) when that byte is
                                  //   rare in text — a cell none of whose bytes equals it is skipped after one zero-byte test per dword
                                  //   (0: no prefilter, all 16 windows are compared)
};

} // namespace kg
