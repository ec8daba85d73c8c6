/* krep_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of the reference's literal-scan hot path, used as the parity checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under krep_amd/ may
 * include, link or call this.  Parity is PINNED: tests/test_oracle_*.py check every function here
 * against (a) the known-answer vectors of the reference's own tests (tests/golden/reference_kat.json)
 * and (b) the unmodified reference compiled into oracle/_ref/ (differential, offsets included).
 *
 * Types come from include/krep_gpu.h (layout-identical to the reference's krep.h).
 */
#ifndef KREP_ORACLE_H
#define KREP_ORACLE_H
#include "../include/krep_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the reference's file-static globals (krep.c:117-120), explicit */
void ko_set_only_matching(int on);
void ko_set_force_no_simd(int on);
void ko_set_algo_override(int krep_ref_algo_override);

/* select_search_algorithm() (krep.c:1771-1870) for a build at `simd` (enum krep_ref_simd):
 * returns the enum krep_ref_algo of the function pointer the reference would return. */
int ko_select(const search_params_t *p, int simd);
/* run the function named by `algo` (with the reference's internal delegation chain) */
uint64_t ko_run(int algo, const search_params_t *p, const char *text, size_t n, match_result_t *r);
/* ko_select + ko_run */
uint64_t ko_search(const search_params_t *p, const char *text, size_t n, match_result_t *r, int simd);

/* the individual operators, search_func_t-shaped (krep.h:98-101) */
uint64_t ko_boyer_moore_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_kmp_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_memchr_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_memchr_short_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_sse42_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_avx2_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_avx512_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_neon_search(const search_params_t *, const char *, size_t, match_result_t *);
uint64_t ko_aho_corasick_search(const search_params_t *, const char *, size_t, match_result_t *);

/* Aho-Corasick automaton (aho_corasick.c:111-271); the handle goes into params->ac_trie */
ac_trie_t *ko_ac_trie_build(const search_params_t *p);
void ko_ac_trie_free(ac_trie_t *t);
uint64_t ko_ac_num_states(const ac_trie_t *t);

/* result container (krep.c:139-251) */
match_result_t *ko_result_init(uint64_t cap);
bool ko_result_add(match_result_t *r, size_t s, size_t e);
void ko_result_free(match_result_t *r);

/* line helpers (krep.c:363-415) and -w predicate (krep.h:298-319) */
size_t ko_line_start(const char *text, size_t n, size_t pos);
size_t ko_line_end(const char *text, size_t n, size_t pos);
bool ko_whole_word(const char *text, size_t n, size_t s, size_t e);

#ifdef __cplusplus
}
#endif
#endif
