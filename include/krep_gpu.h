/* krep_gpu.h — C-ABI of the MI355X-native literal-scan backend for krep.
 *
 * This header is the drop-in boundary.  Everything here is plain C (pointers + sizes, no C++ and
 * no torch types).  It sits exactly where krep's own algorithm functions sit: behind the uniform
 * operator signature `search_func_t` (reference krep.h:98-101), which search_chunk_thread()
 * (krep.c:1950) and search_string() (krep.c:2169) call, and which select_search_algorithm()
 * (krep.c:1771) returns.  The structs below are layout-identical to the reference's
 * (krep.h:49-60 match_position_t / match_result_t, krep.h:65-94 search_params_t) so a caller built
 * against krep.h can pass its own objects straight through.  If krep.h was included first its
 * definitions are used and ours are skipped.
 *
 * Error behaviour: the search_func_t signature has no in-band error channel (the reference's functions
 * print to stderr and return a partial count, e.g. krep.c:1359-1362), and a backend that answers 0
 * when it could not look would turn "failed" into "no match".  So (SURVEY §8b "Errors"):
 *   - without a usable gfx950 device krep_gpu_available() is 0, krep_gpu_can_accelerate() is 0 and
 *     krep_gpu_select_search_algorithm() returns NULL: the caller keeps the CPU function pointer the
 *     reference's own select_search_algorithm() gives it (krep.c:1944-1948);
 *   - an operator that fails at run time (allocation, copy, launch, device lost) appends NOTHING to
 *     `result`, prints "krep-gpu: ..." to stderr, sets krep_gpu_last_error() and
 *     krep_gpu_last_status() != 0, and — when the host registered its CPU selector with
 *     krep_gpu_set_cpu_fallback() — re-runs the search through the host's own CPU function and returns
 *     THAT result (status KREP_GPU_FELL_BACK).  Without a registered selector it returns 0 with status
 *     KREP_GPU_FAILED, which the caller must test (the error_flag path of krep.c:2940-2947).
 * search_buffer() returns 0 match / 1 no match / 2 error exactly like search_file()/search_string()
 * (krep.h:159,168).  The library itself contains no CPU implementation of any search: a fallback is
 * always the CALLER's function.
 */
#ifndef KREP_GPU_H
#define KREP_GPU_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef KREP_H /* ---- ABI twins of the reference types (krep.h:49-101) ---- */
typedef struct
{
    size_t start_offset; /* first byte of the match, relative to text_start           */
    size_t end_offset;   /* one past the last byte (exclusive)                         */
} match_position_t;      /* 16 bytes — also the record format the kernels write in HBM */

typedef struct match_result_t
{
    match_position_t *positions; /* malloc-family block: match_result_free() (krep.c:244) frees it */
    uint64_t count;
    uint64_t capacity;
} match_result_t;

struct ac_trie;
typedef struct ac_trie ac_trie_t;

typedef struct search_params
{
    const char *pattern; /* valid when num_patterns == 1 (krep.c:3736-3740) */
    size_t pattern_len;
    const char **patterns;
    size_t *pattern_lens;
    size_t num_patterns;
    bool case_sensitive;
    bool use_regex;
    bool count_lines_mode;
    bool count_matches_mode;
    bool track_positions;
    bool whole_word;
    const void *compiled_regex; /* const regex_t* in krep.h; never dereferenced here */
    ac_trie_t *ac_trie;         /* opaque; only tested against NULL (aho_corasick.c:306) */
    size_t max_count;
} search_params_t;

typedef uint64_t (*search_func_t)(const search_params_t *params, const char *text_start,
                                  size_t text_len, match_result_t *result);
#endif /* KREP_H */

/* ------------------------------------------------------------------------------------------------
 * Which reference build are we a drop-in for?  The reference picks its algorithm — and with it the
 * match-set family (all occurrences vs. greedy non-overlapping) — from compile-time SIMD macros
 * (krep.c:47-74, :1798-1869) and three file-static globals that no external backend can see
 * (krep.c:117-120).  They are made explicit here.  Defaults: AVX2 build, no overrides.
 * ---------------------------------------------------------------------------------------------- */
enum krep_ref_simd
{
    KREP_REF_SCALAR = 0, /* no -msse4.2: BMH / KMP / memchr only                   */
    KREP_REF_SSE42 = 1,  /* Makefile:40                                            */
    KREP_REF_AVX2 = 2,   /* Makefile:37                                            */
    KREP_REF_AVX512 = 3, /* Makefile:35                                            */
    KREP_REF_NEON = 4    /* Makefile:44-48 (arm64)                                 */
};
enum krep_ref_algo_override
{
    KREP_ALGO_AUTO = 0,
    KREP_ALGO_BM = 1, /* --algo=bm  (krep.c:1788) */
    KREP_ALGO_KMP = 2 /* --algo=kmp (krep.c:1790) */
};
/* The reference algorithm whose exact output (count, positions, order, max_count quirks) a scan
 * reproduces.  krep_gpu_mirror_select() is the twin of select_search_algorithm() incl. the
 * delegation chain AVX-512 -> AVX2 -> SSE4.2 -> BMH (krep.c:4708-4712, :4883-4896, :5114-5126). */
enum krep_ref_algo
{
    KREP_RA_NONE = 0,
    KREP_RA_BMH = 1,          /* boyer_moore_search   krep.c:1260 */
    KREP_RA_KMP = 2,          /* kmp_search           krep.c:1628 */
    KREP_RA_MEMCHR = 3,       /* memchr_search        krep.c:3891 */
    KREP_RA_MEMCHR_SHORT = 4, /* memchr_short_search  krep.c:4371 */
    KREP_RA_SSE42 = 5,        /* simd_sse42_search    krep.c:4702 */
    KREP_RA_AVX2 = 6,         /* simd_avx2_search     krep.c:4877 (17..32 B body) */
    KREP_RA_AVX512 = 7,       /* simd_avx512_search   krep.c:5108 (33..64 B body) */
    KREP_RA_NEON = 8,         /* neon_search          krep.c:4506 */
    KREP_RA_AHO_CORASICK = 9, /* aho_corasick_search  aho_corasick.c:299 */
    KREP_RA_REGEX = 10        /* regex_search — out of scope, never executed here */
};

/* All of it in one explicit object.  Plans and search_buffer_ex() carry their configuration; the setters below write the
 * PROCESS-WIDE defaults (like the reference's globals, set once by main() before the pool threads start; relaxed atomics),
 * krep_gpu_set_thread_config() overrides them for the calling thread (NULL: back to the defaults).  No scan path writes
 * any of this, so concurrent calls with different configurations do not interfere. */
typedef struct krep_gpu_config
{
    int reference_simd;        /* enum krep_ref_simd: which reference BUILD is reproduced (krep.c:47-74)              */
    int only_matching;         /* static only_matching (-o), krep.c:117                                                 */
    int force_no_simd;         /* static force_no_simd, krep.c:118                                                      */
    int algo_override;         /* enum krep_ref_algo_override (--algo), krep.c:120                                      */
    int result_order;          /* 1: multi-pattern records come back in (start, end) order (see below)                  */
    int device;                /* HIP device the host-buffer operators use (default: $KREP_GPU_DEVICE, else 0)          */
    size_t stream_chunk_bytes; /* piece size of the streamed host path (0 = default 128 MiB); texts > 2 pieces stream   */
    int num_gpus;              /* devices the search_func_t operators shard a text over (default: $KREP_GPU_NUM, else 1;
                                  0 or less = all visible devices); shards meet in one RCCL all-reduce of their counters  */
    size_t min_text_bytes;     /* krep_gpu_worthwhile(): texts shorter than this stay on the CPU function (default:
                                  $KREP_GPU_MIN_BYTES, else 1 MiB — one operator call costs 50-80 us)       */
} krep_gpu_config_t;
void krep_gpu_config_default(krep_gpu_config_t *out);           /* the current process-wide defaults */
void krep_gpu_set_thread_config(const krep_gpu_config_t *cfg);  /* calling thread only; NULL clears   */
void krep_gpu_set_device(int device);                           /* process-wide default device         */
void krep_gpu_set_stream_chunk(size_t bytes);                   /* process-wide default piece size     */
void krep_gpu_set_num_gpus(int n);                              /* process-wide default shard count    */
void krep_gpu_set_min_text_bytes(size_t bytes);                 /* process-wide default size threshold */
/* frees every per-device context of the host path (device buffers, pinned staging ring, cached plans) */
void krep_gpu_release_device_resources(void);

void krep_gpu_set_reference_simd(int krep_ref_simd_level);   /* compile-time SIMD macros, krep.c:47-74 */
void krep_gpu_set_only_matching(int on);                     /* static only_matching,   krep.c:117   */
void krep_gpu_set_force_no_simd(int on);                     /* static force_no_simd,   krep.c:118   */
void krep_gpu_set_algo_override(int krep_ref_algo_override); /* static algo_override,   krep.c:120   */
int krep_gpu_get_reference_simd(void);
/* Twin of select_search_algorithm() (krep.c:1771): the algorithm the reference build would END UP
 * executing for `params` on a text of `text_len` bytes (text_len matters: the SIMD functions fall back
 * to BMH when text_len < pattern_len, and so on). */
int krep_gpu_mirror_select(const search_params_t *params, size_t text_len);
const char *krep_gpu_algorithm_name(int krep_ref_algo); /* twin of get_algorithm_name(), krep.c:1964 */

/* ------------------------------------------------------------------------------------------------
 * Availability, failure status and the CPU fallback (SURVEY §8b "Errors", §5 "Failure detection").
 * ---------------------------------------------------------------------------------------------- */
/* 1 when the configured device exists, is a gfx950 and a probe kernel of this library's code object ran on it
 * (checked once per device and process); 0 otherwise, with the reason in krep_gpu_unavailable_reason().
 * $KREP_GPU_DISABLE=1 forces 0. */
int krep_gpu_available(void);
const char *krep_gpu_unavailable_reason(void); /* "" when available */
/* The host's own selector of CPU functions — for the reference CLI a wrapper of select_search_algorithm() (krep.c:1771)
 * that skips the GPU branch.  When registered, an operator that fails at run time calls
 * select_cpu(params)(params, text, len, result) itself and returns that function's result (sorted into (start, end)
 * order when result_order is set, so the promise of krep_gpu_set_result_order() holds either way). NULL unregisters. */
typedef search_func_t (*krep_gpu_cpu_select_t)(const search_params_t *params);
void krep_gpu_set_cpu_fallback(krep_gpu_cpu_select_t select_cpu);
/* Outcome of the calling thread's LAST operator / search_buffer call. */
enum krep_gpu_status
{
    KREP_GPU_OK = 0,        /* the scan ran on the GPU; the return value and `result` are the backend's             */
    KREP_GPU_FELL_BACK = 1, /* the GPU path failed; the registered CPU function produced the return value / result */
    KREP_GPU_FAILED = 2     /* the GPU path failed and no CPU function was available: the returned 0 means
                               "could not look", NOT "no match" — nothing was appended to `result`                 */
};
int krep_gpu_last_status(void);
/* Should a caller hand THIS text to the backend?  search_file() asks this once per file (krep.c:2404-2420 already special-cases
 * small files; krep.c:2729-2770 scales its thread count with the size).  1 iff
 *   text_len >= min_text_bytes  &&  krep_gpu_can_accelerate(params)  &&  the cost model expects the GPU to be faster  &&
 *   krep_gpu_available()   (asked last: a text that stays on the CPU never starts the HIP runtime).
 * The cost model (krep_amd/csrc/kg_cost.hip):
 *   t_gpu = (device not yet initialised in this process ? gpu_init_ms : 0) + gpu_launch_us + text_len / host-path rate
 *   t_cpu = text_len / min(cpu_threads x per-thread rate of the function select_search_algorithm() would run, its cap)
 * cpu_threads <= 0: what search_file() itself would use, min(cores, text_len / 4 MiB), at least 1 (krep.c:2748-2759).
 * The host-path rate is calibrated per process by the library itself (large operator calls after the first one, which also pays
 * for tables, arena and staging ring); the device-start term is dropped once the availability probe has run in this process (its
 * measured time is not used as a value).  Everything is overridable: krep_gpu_set_cost_rates(), $KREP_GPU_COST
 * ("host=50,launch=100,init=450,memchr=12:180,simd=6:150,scalar=1.5:100,ac=0.4:100": GB/s per thread : cap).
 * rates.enabled = 0 or $KREP_GPU_COST_MODEL=0: size and input class alone decide (the round-3 rule). */
typedef struct krep_gpu_cost_rates
{
    int enabled;
    double gpu_host_path_gbps, gpu_launch_us, gpu_init_ms;
    double cpu_memchr_gbps, cpu_memchr_cap_gbps;  /* memchr_search                                                  */
    double cpu_simd_gbps, cpu_simd_cap_gbps;      /* simd_sse42 / avx2 / avx512 / neon_search, memchr_short_search  */
    double cpu_scalar_gbps, cpu_scalar_cap_gbps;  /* boyer_moore_search, kmp_search                                 */
    double cpu_ac_gbps, cpu_ac_cap_gbps;          /* aho_corasick_search while its automaton fits the caches ...    */
    double cpu_ac_cache_bytes, cpu_ac_exponent;   /* ... and x (cache / automaton bytes)^exponent beyond (2 KiB per state) */
} krep_gpu_cost_rates_t;
typedef struct krep_gpu_cost
{
    double gpu_seconds, cpu_seconds;  /* the two estimates                                                  */
    double gpu_host_path_gbps;        /* the host-path rate used (calibrated when this process has measured one) */
    int cpu_threads, cpu_algo;        /* threads assumed; enum krep_ref_algo of the CPU function             */
    int device_ready;                 /* 1: this process has already started its device (no init in t_gpu)  */
} krep_gpu_cost_t;
int krep_gpu_worthwhile(const search_params_t *params, size_t text_len);
int krep_gpu_worthwhile_ex(const search_params_t *params, size_t text_len, int cpu_threads);
int krep_gpu_cost_estimate(const search_params_t *params, size_t text_len, int cpu_threads, krep_gpu_cost_t *out); /* 0 ok */
void krep_gpu_get_cost_rates(krep_gpu_cost_rates_t *out);
void krep_gpu_set_cost_rates(const krep_gpu_cost_rates_t *rates); /* NULL: back to the defaults + $KREP_GPU_COST */

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (search_func_t-compatible).  `text_start` is a HOST pointer, exactly
 * as in the reference; the bytes are staged to HBM through pinned buffers, scanned by the HIP
 * kernels, and the results appended to `result` with the match_result_add() contract
 * (krep.c:175-241: malloc/realloc'd block, offsets relative to text_start, end exclusive).
 * Return value = match count, or distinct-line count when params->count_lines_mode.
 * ---------------------------------------------------------------------------------------------- */
uint64_t krep_gpu_literal_search(const search_params_t *params, const char *text_start,
                                 size_t text_len, match_result_t *result);
uint64_t krep_gpu_aho_corasick_search(const search_params_t *params, const char *text_start,
                                      size_t text_len, match_result_t *result);
/* Drop-in for select_search_algorithm(): returns one of the two functions above, or NULL when the backend does not
 * take the search — no usable device (krep_gpu_available() == 0), regex_search, and the input classes
 * krep_gpu_can_accelerate() names — so that the caller keeps the CPU function pointer the reference's own
 * select_search_algorithm() gives it.  (-c through simd_sse42_search / kmp_search with a newline inside the pattern, refused
 * until round 3, is reproduced: one device thread walks the ordered occurrence list the way the reference's loop moves.) */
search_func_t krep_gpu_select_search_algorithm(const search_params_t *params);
/* 1 when the backend takes the search for `params` under the current configuration, 0 when not:
 *   - no usable gfx950 device (krep_gpu_available() == 0);
 *   - no pattern at all (num_patterns == 0 and pattern == NULL);
 *   - use_regex.
 * (count_lines_mode together with only_matching through memchr_short_search — a combination krep's main() never produces,
 * krep.c:3811-3814 — was refused until round 5 and is reproduced now: one window, krep_gpu_split_mode() = WHOLE.)
 * An operator called with such params anyway treats it like a run-time failure (see the top of this header): the
 * registered CPU function answers, or status KREP_GPU_FAILED; nothing is silently approximated. */
int krep_gpu_can_accelerate(const search_params_t *params);

/* The in-memory twin of search_file()/search_string() that BASELINE.json calls search_buffer():
 * validation as krep.c:2013-2049 (no patterns -> 2; empty pattern among several -> 2; pattern
 * longer than 1024 -> 2), algorithm selection as krep.c:2166, single-chunk semantics, clamp to
 * max_count as krep.c:2176-2184.  Nothing is printed.  num_gpus > 1 shards the buffer by contiguous
 * chunk across that many devices of this process with start-offset ownership.
 * Returns 0 = match found, 1 = none, 2 = error. */
int search_buffer(const search_params_t *params, const char *buf, size_t len, int only_matching,
                  int num_gpus, match_result_t *out /* nullable */, uint64_t *count_out /* nullable */);
/* The same with the whole configuration explicit (cfg == NULL: the calling thread's current configuration).
 * Re-entrant: any number of threads may call it concurrently with different configurations. */
int search_buffer_ex(const search_params_t *params, const char *buf, size_t len, const krep_gpu_config_t *cfg,
                     int num_gpus, match_result_t *out /* nullable */, uint64_t *count_out /* nullable */);

/* match_result_t helpers with the reference's allocation contract (krep.c:139-251), exported so a
 * caller without krep.c can own the results. */
match_result_t *krep_gpu_match_result_init(uint64_t initial_capacity);
void krep_gpu_match_result_free(match_result_t *r);

/* ------------------------------------------------------------------------------------------------
 * Device-resident path (the one the roofline number is measured on): the haystack already lives
 * in HBM.  A plan holds the compiled pattern set (pattern words / Aho-Corasick filter + trie
 * tables) on one device; a scan runs it over a device buffer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct krep_gpu_plan krep_gpu_plan_t;

/* per-scan output; counters are exact even when the position buffer is too small */
typedef struct krep_gpu_scan_out
{
    uint64_t count;          /* what the reference function would RETURN for this shard           */
    uint64_t stored;         /* number of match_position_t records written to d_positions         */
    uint64_t total_matches;  /* uncapped number of emitted matches (before max_count)             */
    uint64_t line_count;     /* distinct lines holding a match start (count_lines_mode), shard-local */
    uint8_t head_line_hit;   /* a match starts before the first '\n' of the owned window           */
    uint8_t tail_line_hit;   /* a match starts after the last '\n' of the owned window             */
    uint8_t has_newline;     /* the owned window contains at least one '\n'                        */
    uint8_t overflow;        /* 1: more matches than position capacity (re-run with more)          */
    float kernel_ms;         /* hipEvent time of the scan kernels on `stream` (0 if not timed)     */
} krep_gpu_scan_out_t;

/* A plan is used by ONE thread at a time: it owns scratch buffers, events and what its scans have learnt about the text's density
 * (staging-slot size, ring shape of the one-pass kernels, which -c road pays off) — re-evaluated by every scan, not synchronised.
 * Concurrent scans take one plan each (the host operators keep a small plan cache per device behind the device's mutex). */
krep_gpu_plan_t *krep_gpu_plan_create(const search_params_t *params, int only_matching, int device);
krep_gpu_plan_t *krep_gpu_plan_create_ex(const search_params_t *params, const krep_gpu_config_t *cfg /* NULL = current */);
void krep_gpu_plan_destroy(krep_gpu_plan_t *plan);
int krep_gpu_plan_ref_algo(const krep_gpu_plan_t *plan); /* enum krep_ref_algo this plan reproduces */

/* Scan d_text[0 .. text_len) and report the matches whose START lies in [own_lo, own_hi)
 * (start-offset ownership; bytes outside the window are context: the tail halo completes matches
 * that begin inside, the byte before own_lo serves -w and line bookkeeping).  Offsets written are
 * (offset within d_text) + global_base.  d_positions may be NULL (count only) and receives at most
 * `position_capacity` records, in the reference's emission order.  `stream` is a hipStream_t (NULL =
 * default stream); with time_it != 0 the kernels are bracketed by hipEvents on that stream and the
 * call synchronises.  The kernels move 16 bytes per lane from d_text + a multiple of 16: a 16-byte aligned d_text (any
 * hipMalloc block) is the fast case; a slice at an odd offset is scanned correctly (gfx950 serves unaligned vector loads;
 * tests/test_gpu_fullsize.py scans such slices), an aligned block plus an ownership window is the better way to say it.
 * Reads: d_text is never written, and no byte outside [0, text_len) can influence a result; the list-based -c of the
 * multi-pattern scan reads whole ALIGNED 16-byte granules and so may touch up to 15 bytes behind text_len inside the granule
 * that holds the last byte (the same page: safe for any allocation, worth knowing for a slice that ends an allocation).
 * Placement: WHERE the driver puts a large allocation moves a 32-GiB read stream by 2-3 % and a scan that also writes GBs of
 * records by ~10 % (two modes, one per allocation) — the library takes the buffers as the caller made them;
 * krep_gpu_alloc_placed() below makes them with the placement drawn for.
 * Returns 0 on success, non-zero on error (krep_gpu_last_error()). */
int krep_gpu_scan_device(krep_gpu_plan_t *plan, const void *d_text, size_t text_len, size_t own_lo,
                         size_t own_hi, size_t global_base, match_position_t *d_positions,
                         uint64_t position_capacity, void *stream, int time_it,
                         krep_gpu_scan_out_t *out);

/* Device memory for a text and the records of its scans whose PLACEMENT has been drawn for (kg_place.hip).  Allocates up to `tries`
 * (1..8) candidate blocks of text_bytes (+ 64 bytes of slack) followed by record_bytes on `device`, times the single-byte workload of
 * BASELINE config 3 on each (the generator's 1 %-density text in the candidate's text area; counting only, and with the records written
 * into the candidate's record area), keeps the candidate whose record-writing scan ran fastest — the first one that runs within 1.32x of
 * its own counting scan is taken at once — and frees the others.  *d_text receives the block (hipMalloc alignment), *d_records (may be NULL
 * when record_bytes == 0) the record area inside the same block, 256-byte aligned; `info` (may be NULL) what was drawn.  The text area
 * comes back holding the probe's bytes.  tries <= 1, a text of less than 1 GiB or a record area of less than 16 bytes per 64 bytes of the
 * first GiB: one plain allocation, nothing timed.  Cost per draw at 32 GiB + 8 GiB: the allocation + ~50 ms of probe scans.
 * Free with krep_gpu_free_placed(device, *d_text).  Returns 0, or 2 (krep_gpu_last_error()). */
typedef struct krep_gpu_placement
{
    uint32_t tries;          /* candidates drawn                                  */
    uint32_t kept;           /* index of the one returned                         */
    float count_only_ms[8];  /* per draw: kernel time of the counting scan        */
    float records_ms[8];     /* per draw: median of three record-writing scans    */
} krep_gpu_placement_t;
int krep_gpu_alloc_placed(int device, size_t text_bytes, size_t record_bytes, int tries, void **d_text, void **d_records,
                          krep_gpu_placement_t *info);
int krep_gpu_free_placed(int device, void *d_text);

/* The same for a device buffer that is a SLICE [global_base, global_base + text_len) of a text of global_len bytes
 * (global_len == 0: the buffer ends the text).  The reference functions place some of their behaviour by the length of
 * the WHOLE text (the block simd_avx512_search leaves unexamined, krep.c:5171; the first byte of the scalar tail calls,
 * which has no left neighbour for -w, krep.c:5059-5097): with global_len those land where the reference puts them, in
 * whichever shard holds them.  The sequential match-set families (greedy SSE4.2/KMP selection of a bordered pattern,
 * -o through BMH / memchr_short; -c through the AVX-512 / AVX2 -w / NEON block loops) accept a window inside the text only
 * through krep_gpu_scan_device_seq() below; krep_gpu_split_mode() names the few classes that need the whole text in one window. */
int krep_gpu_scan_device_ex(krep_gpu_plan_t *plan, const void *d_text, size_t text_len, size_t own_lo,
                            size_t own_hi, size_t global_base, size_t global_len, match_position_t *d_positions,
                            uint64_t position_capacity, void *stream, int time_it, krep_gpu_scan_out_t *out);

/* ---- the sequential match-set families in pieces ("one boundary record per shard ... one exchange step") ----
 * The greedy left-to-right selection of simd_sse42_search / kmp_search (krep.c:4839-4848, :1741), boyer_moore_search under
 * -o (:1371) and memchr_short_search's -o walk (:4495) couple a match to the one before it; across a cut of the text the
 * whole coupling is ONE number: where the reference's scan stands when it enters the right-hand piece. */
typedef struct krep_gpu_seq_carry
{
    uint64_t resume;      /* greedy / -o walks: global offset from which the reference's scan continues — starts in front of
                             it are consumed, the first occurrence / candidate at or behind it is looked at afresh (0: nothing
                             consumed)                                                                                      */
    /* -c through simd_avx512_search / simd_avx2_search -w / neon_search (krep.c:5203-5218, :5000-5013, :4590-4611): the line-skip history that decides
     * where the block grid stands when it enters the last 256 bytes of the text (the end-of-text replay, kg_replay.h).
     * All offsets are global and stored + 1 (0 = none).                                                                    */
    uint64_t q1;          /* start of the last accepted occurrence in the text so far                                        */
    uint64_t nl1;         /* first '\n' at or behind it                                                                      */
    uint64_t local_q1;    /* the same for THIS piece alone: a caller that scanned its pieces out of order (shards on       */
    uint64_t local_nl1;   /* different devices) folds them afterwards: q1 = local_q1 ? local_q1 : in.q1,                   */
    uint64_t local_first_nl1; /* nl1 = local_q1 ? local_nl1 : in.nl1 ? in.nl1 : in.q1 ? local_first_nl1 : 0                 */
    /* neon_search only (krep.c:4590-4611: no restart on an unterminated line, so its block grid is still the one the previous
     * counted line set): g0 = the grid origin in effect for q's line — (first '\n' behind the last accepted occurrence on an
     * EARLIER line) + 1, or 0.  local_g0_kind: 0 no occurrence in this piece; 1 g0 = local_g0; 2 q's line starts in this piece
     * but the earlier occurrence lies in front of it: g0 = in.q1 ? (in.nl1 ? in.nl1 : local_first_nl1) : 0; 3 q's line started
     * in front of this piece: g0 = in.q1 ? (in.nl1 ? in.nl1 : in.g0) : 0.                                                    */
    uint64_t g0, local_g0, local_g0_kind;
    /* -c with a '\n' inside a pattern.  Multi-pattern (aho_corasick.c:383-396: the counter is bumped whenever the line of a match
     * START differs from the line of the previously counted one, matches visited in emission order — end ascending): a piece
     * owns the matches that END in it, its list continues its predecessor's, and the coupling is two numbers.  Through
     * simd_sse42_search / kmp_search (krep.c:4785-4795, :1703-1707): a piece owns the occurrences that START in it; the
     * coupling is `resume` above (where the reference's scan stands) plus the same two numbers:                           */
    uint64_t nl_before;   /* '\n' bytes of the text in front of this piece's own_hi (carry_out) / own_lo (carry_in)         */
    uint64_t last_line;   /* 1-based line number of the START of the last match of the text so far (0: no match yet)        */
    uint64_t local_nl;    /* this piece alone: '\n' bytes in [own_lo, own_hi) ...                                           */
    uint64_t local_last;  /* ... and the line of its last match's start RELATIVE to own_lo, biased by 2^62 (0: no match):   */
                          /* a caller folding out-of-order pieces: nl_before = in.nl_before + local_nl,                      */
                          /* last_line = local_last ? in.nl_before + 1 + (local_last - 2^62) : in.last_line                  */
    uint64_t local_lines; /* -c through the block loops, the piece that ENDS the text: its canonical line count in front of the   */
                          /* end-of-text replay, stored + 1 (0: not that piece) — what krep_gpu_replay_tail() starts from         */
} krep_gpu_seq_carry_t;
/* krep_gpu_scan_device_ex() for the pieces of one text IN TEXT ORDER: carry_in = the record the previous piece left
 * (NULL: nothing in front of this window is consumed — the piece that starts the text, or an optimistic first pass of a
 * shard whose left neighbour is still running: compare its assumption with the neighbour's carry_out afterwards and re-run
 * the piece if they differ), carry_out (nullable) = this piece's record.  Families without a sequential dependency pass
 * the record through unchanged.  For the -c block-loop family only the piece that ENDS the text depends on the record (it
 * must be at least 4 KiB long); the others only extend it. */
int krep_gpu_scan_device_seq(krep_gpu_plan_t *plan, const void *d_text, size_t text_len, size_t own_lo, size_t own_hi,
                             size_t global_base, size_t global_len, match_position_t *d_positions, uint64_t position_capacity,
                             void *stream, int time_it, const krep_gpu_seq_carry_t *carry_in, krep_gpu_seq_carry_t *carry_out,
                             krep_gpu_scan_out_t *out);
/* The end-of-text replay ALONE (-c through the block loops).  The piece that ends the text was scanned with a guessed record
 * (its shard started before its left neighbours had finished); `carry_true` is what the text in front of it really leaves,
 * `piece` the record that piece produced (its local_* fields: its own line-skip contribution and local_lines).  d_tail holds the
 * last tail_len bytes of the text (at least 512, or the whole text) — nothing else of the piece is read again.  Returns in
 * *lines what krep_gpu_scan_device_seq() would have reported as line_count / count for that piece had it been given carry_true,
 * and in *carry_out the record it would have left.  (the multi-shard operators re-staged and re-scanned that piece.) */
int krep_gpu_replay_tail(krep_gpu_plan_t *plan, const void *d_tail, size_t tail_len, size_t global_len, void *stream,
                         const krep_gpu_seq_carry_t *carry_true, const krep_gpu_seq_carry_t *piece, krep_gpu_seq_carry_t *carry_out,
                         uint64_t *lines);
/* How a text of text_len bytes may be cut for `params` under the current configuration. */
enum krep_gpu_split
{
    KREP_GPU_SPLIT_WHOLE = 0,  /* one window only: neon_search's max_count == 0 corner, and -c with -o through
                                  memchr_short_search (never produced by krep's main())                                    */
    KREP_GPU_SPLIT_PIECES = 1, /* independent pieces: start-offset ownership + halo, results concatenate / merge          */
    KREP_GPU_SPLIT_CHAIN = 2   /* pieces in text order through krep_gpu_scan_device_seq() (round 5: also -c with a newline
                                  inside a pattern, multi-pattern and through simd_sse42_search / kmp_search)             */
};
int krep_gpu_split_mode(const search_params_t *params, size_t text_len);

/* Deterministic synthetic haystacks, generated directly in HBM by a counter-based
 * PRNG so that any [global_off, global_off+len) slice is reproducible on any rank.
 * kind: 2 = background a-z/space/newline + planted 8-byte literal every `period` bytes (cfg 2/5),
 *       3 = background + target byte with probability 1/100 (cfg 3),
 *       4 = background + planted dictionary words (cfg 4; `plant`/`plant_len` = packed patterns),
 *       5 = word text: `period`-byte lines ('\n'-terminated) of words drawn from the packed list `plant` with p(rank) ~ 1/rank,
 *           single blanks between them (the natural-language-like text of cfg 1 / the reference's own benchmark corpus). */
int krep_gpu_generate(void *d_dst, size_t len, size_t global_off, int kind, uint64_t seed,
                      const void *plant, size_t plant_len, uint64_t period, void *stream);
/* Host twin of the generator (same bytes), for parity tests and the CPU baseline. */
void krep_gpu_generate_host(void *dst, size_t len, size_t global_off, int kind, uint64_t seed,
                            const void *plant, size_t plant_len, uint64_t period);

/* Combine per-shard line bookkeeping left-to-right (the one "exchange step" of the multi-GPU path):
 * returns the global distinct-line count given shard outputs in shard order. */
uint64_t krep_gpu_combine_line_counts(const krep_gpu_scan_out_t *shards, int n);

/* ---- the collective of the multi-GPU path: per-shard counters meet in ONE RCCL all-reduce over xGMI -------
 * One process driving several devices (search_buffer(num_gpus > 1), the operators with cfg.num_gpus > 1) does this
 * internally (ncclCommInitAll over the devices used).  One process PER GPU (bench.py under torch.distributed.run) uses the
 * rank-level calls below: rank 0 draws an id, the host program's own bootstrap carries its 128 bytes to the other ranks,
 * every rank joins, and each scan ends with one all-reduce of its counters {matches, lines, ...}.  librccl is opened on
 * first use (dlopen); all return 0, or 2 with krep_gpu_last_error() set. */
#define KREP_GPU_COMM_ID_BYTES 128
int krep_gpu_comm_unique_id(void *id128);
int krep_gpu_comm_init_rank(const void *id128, int nranks, int rank, int device);
int krep_gpu_comm_allreduce_u64(uint64_t *values, int n);                         /* host values, in place; synchronous  */
int krep_gpu_comm_allreduce_device_u64(void *d_values, int n, void *stream);      /* device-resident, in place, async    */
void krep_gpu_comm_destroy(void);
uint64_t krep_gpu_rccl_calls(void); /* collectives this process has issued through RCCL (diagnostic / self-test) */
int krep_gpu_rccl_version(void);    /* ncclGetVersion(), 0 when librccl cannot be loaded */
/* Where the calling thread's LAST sharded host search (search_buffer(num_gpus > 1), the operators with num_gpus > 1) ran:
 * logical shards, the distinct physical devices they were placed on, the ranks of the communicator the counters met in
 * (0: a single device, nothing to reduce) and how they met (1 = the RCCL all-reduce, 2 = RCCL could not be used and the
 * host summed the slots it already held — the search result is the same, krep_gpu_last_error() names the reason).
 * The multi-GPU self-test (tests/test_gpu_multi.py) asserts on it; a search that did not shard leaves shards = 1. */
typedef struct krep_gpu_shard_info
{
    int shards;          /* logical shards of the text                                  */
    int devices_used;    /* distinct physical devices                                   */
    int device_ids[16];  /* the first 16 of them, in shard order                        */
    int comm_ranks;      /* ranks of the communicator (ncclCommInitAll over the devices) */
    int reduced_by;      /* 0 nothing to reduce, 1 RCCL, 2 host sum after an RCCL failure */
} krep_gpu_shard_info_t;
void krep_gpu_last_shard_info(krep_gpu_shard_info_t *out);

/* ---- formatter-side post-processing in HBM (what search_file() does on one host thread after the scan) --------------
 * krep_gpu_order_by_start: the (start, end) order of compare_match_positions (krep.c:420-434) that search_file()
 *   establishes with qsort() before printing (krep.c:3018-3023).  Input: n records ascending in `end` (the order of
 *   every operator of this library); one stable device radix sort keyed on `start`, in place.
 * krep_gpu_line_numbers: the 1-based line number of every record's start, as print_matching_items() derives them by
 *   counting newlines (krep.c:589-668); d_lines[n] on the device.
 * krep_gpu_set_result_order(1): krep_gpu_aho_corasick_search() hands back its records already in (start, end) order
 *   (sorted on the device before the copy to the host), so the caller may skip its qsort().  Default 0: the
 *   reference's emission order (end ascending, longest first).
 * All return 0, or 2 with krep_gpu_last_error() set. */
/* max_offset: upper bound of every start offset in the list (the WHOLE text's length when the records carry a
 * global_base) — the radix key is sized from it.  Limit: n <= 2^31 - 1 records (34 GB of match_position_t; the device
 * radix sort counts its items in an int) — a longer list is refused with an error, never truncated. */
int krep_gpu_order_by_start(match_position_t *d_positions, uint64_t n, size_t max_offset, void *stream);
/* records must be relative to d_text[0]; use the _ex form (records minus global_base) for shard lists.  A record outside
 * [global_base, global_base + text_len] gets line number 0. */
int krep_gpu_line_numbers(const void *d_text, size_t text_len, const match_position_t *d_positions, uint64_t n,
                          uint64_t *d_lines, void *stream);
int krep_gpu_line_numbers_ex(const void *d_text, size_t text_len, size_t global_base, const match_position_t *d_positions,
                             uint64_t n, uint64_t *d_lines, void *stream);
void krep_gpu_set_result_order(int by_start);

int krep_gpu_device_count(void);
const char *krep_gpu_last_error(void); /* "" when the last call on this thread succeeded */
void krep_gpu_clear_error(void);
const char *krep_gpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KREP_GPU_H */
