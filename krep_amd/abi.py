"""ctypes mirror of include/krep_gpu.h (== reference krep.h:49-101 layouts).

Only plumbing: struct layouts, a `make_params()` factory that reads like the reference tests'
`create_literal_params()` (test/test_krep.c:239-249), and numpy views of match_result_t.
No search logic lives here.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

SIZE_MAX = (1 << 64) - 1

# enum krep_ref_simd
REF_SCALAR, REF_SSE42, REF_AVX2, REF_AVX512, REF_NEON = 0, 1, 2, 3, 4
# enum krep_ref_algo
(RA_NONE, RA_BMH, RA_KMP, RA_MEMCHR, RA_MEMCHR_SHORT, RA_SSE42, RA_AVX2, RA_AVX512, RA_NEON,
 RA_AHO_CORASICK, RA_REGEX) = range(11)
ALGO_AUTO, ALGO_BM, ALGO_KMP = 0, 1, 2

RA_NAMES = {RA_BMH: "boyer_moore_search", RA_KMP: "kmp_search", RA_MEMCHR: "memchr_search",
            RA_MEMCHR_SHORT: "memchr_short_search", RA_SSE42: "simd_sse42_search",
            RA_AVX2: "simd_avx2_search", RA_AVX512: "simd_avx512_search", RA_NEON: "neon_search",
            RA_AHO_CORASICK: "aho_corasick_search"}


class MatchPosition(C.Structure):
    _fields_ = [("start_offset", C.c_size_t), ("end_offset", C.c_size_t)]


class MatchResult(C.Structure):
    _fields_ = [("positions", C.POINTER(MatchPosition)), ("count", C.c_uint64),
                ("capacity", C.c_uint64)]


class SearchParams(C.Structure):
    _fields_ = [
        ("pattern", C.c_char_p), ("pattern_len", C.c_size_t),
        ("patterns", C.POINTER(C.c_char_p)), ("pattern_lens", C.POINTER(C.c_size_t)),
        ("num_patterns", C.c_size_t),
        ("case_sensitive", C.c_bool), ("use_regex", C.c_bool), ("count_lines_mode", C.c_bool),
        ("count_matches_mode", C.c_bool), ("track_positions", C.c_bool), ("whole_word", C.c_bool),
        ("compiled_regex", C.c_void_p), ("ac_trie", C.c_void_p), ("max_count", C.c_size_t),
    ]


assert C.sizeof(SearchParams) == 72 and C.sizeof(MatchPosition) == 16 and C.sizeof(MatchResult) == 24


class ScanOut(C.Structure):
    _fields_ = [("count", C.c_uint64), ("stored", C.c_uint64), ("total_matches", C.c_uint64),
                ("line_count", C.c_uint64), ("head_line_hit", C.c_uint8), ("tail_line_hit", C.c_uint8),
                ("has_newline", C.c_uint8), ("overflow", C.c_uint8), ("kernel_ms", C.c_float)]


class Config(C.Structure):
    """krep_gpu_config_t: the reference's build level and file-static option globals, explicit."""
    _fields_ = [("reference_simd", C.c_int), ("only_matching", C.c_int), ("force_no_simd", C.c_int),
                ("algo_override", C.c_int), ("result_order", C.c_int), ("device", C.c_int),
                ("stream_chunk_bytes", C.c_size_t), ("num_gpus", C.c_int), ("min_text_bytes", C.c_size_t)]


class CostRates(C.Structure):
    """krep_gpu_cost_rates_t"""
    _fields_ = [("enabled", C.c_int), ("gpu_host_path_gbps", C.c_double), ("gpu_launch_us", C.c_double), ("gpu_init_ms", C.c_double),
                ("cpu_memchr_gbps", C.c_double), ("cpu_memchr_cap_gbps", C.c_double), ("cpu_simd_gbps", C.c_double),
                ("cpu_simd_cap_gbps", C.c_double), ("cpu_scalar_gbps", C.c_double), ("cpu_scalar_cap_gbps", C.c_double),
                ("cpu_ac_gbps", C.c_double), ("cpu_ac_cap_gbps", C.c_double), ("cpu_ac_cache_bytes", C.c_double),
                ("cpu_ac_exponent", C.c_double)]


class Cost(C.Structure):
    """krep_gpu_cost_t"""
    _fields_ = [("gpu_seconds", C.c_double), ("cpu_seconds", C.c_double), ("gpu_host_path_gbps", C.c_double),
                ("cpu_threads", C.c_int), ("cpu_algo", C.c_int), ("device_ready", C.c_int)]


class ShardInfo(C.Structure):
    """krep_gpu_shard_info_t: where the calling thread's last sharded host search ran."""
    _fields_ = [("shards", C.c_int), ("devices_used", C.c_int), ("device_ids", C.c_int * 16), ("comm_ranks", C.c_int),
                ("reduced_by", C.c_int)]


STATUS_OK, STATUS_FELL_BACK, STATUS_FAILED = 0, 1, 2
SPLIT_WHOLE, SPLIT_PIECES, SPLIT_CHAIN = 0, 1, 2


class SeqCarry(C.Structure):
    """krep_gpu_seq_carry_t: the boundary record of the sequential match-set families."""
    _fields_ = [("resume", C.c_uint64), ("q1", C.c_uint64), ("nl1", C.c_uint64), ("local_q1", C.c_uint64),
                ("local_nl1", C.c_uint64), ("local_first_nl1", C.c_uint64), ("g0", C.c_uint64), ("local_g0", C.c_uint64),
                ("local_g0_kind", C.c_uint64), ("nl_before", C.c_uint64), ("last_line", C.c_uint64), ("local_nl", C.c_uint64),
                ("local_last", C.c_uint64), ("local_lines", C.c_uint64)]


SEARCH_FUNC = C.CFUNCTYPE(C.c_uint64, C.POINTER(SearchParams), C.c_char_p, C.c_size_t,
                          C.POINTER(MatchResult))


class Placement(C.Structure):
    """krep_gpu_placement_t (include/krep_gpu.h): what krep_gpu_alloc_placed() drew"""
    _fields_ = [("tries", C.c_uint32), ("kept", C.c_uint32), ("count_only_ms", C.c_float * 8), ("records_ms", C.c_float * 8)]


class Params:
    """Owns the Python-side buffers a search_params_t points into."""

    def __init__(self, patterns: Sequence[bytes], *, case_sensitive=True, count_lines=False,
                 only_match=False, whole_word=False, max_count=SIZE_MAX, track_positions=None):
        pats = [bytes(p) for p in patterns]
        self._pats = pats
        n = len(pats)
        self._arr = (C.c_char_p * max(n, 1))(*pats) if n else (C.c_char_p * 1)()
        self._lens = (C.c_size_t * max(n, 1))(*[len(p) for p in pats]) if n else (C.c_size_t * 1)()
        s = SearchParams()
        s.patterns = C.cast(self._arr, C.POINTER(C.c_char_p)) if n else None
        s.pattern_lens = C.cast(self._lens, C.POINTER(C.c_size_t)) if n else None
        s.num_patterns = n
        if n:
            # legacy single-pattern fields (test_krep.c:233-235); c_char_p keeps embedded NULs
            # because pattern_len carries the length
            s.pattern = pats[0]
            s.pattern_len = len(pats[0])
        s.case_sensitive = case_sensitive
        s.use_regex = False
        # create_base_params(), test/test_krep.c:225-229
        s.count_lines_mode = bool(count_lines and not only_match)
        s.count_matches_mode = bool(count_lines and only_match)
        s.track_positions = (not (count_lines and not only_match)) if track_positions is None \
            else bool(track_positions)
        s.whole_word = whole_word
        s.max_count = max_count
        self.s = s

    @property
    def ref(self):
        return C.byref(self.s)


def make_params(pattern, **kw) -> Params:
    if isinstance(pattern, (bytes, bytearray)):
        return Params([bytes(pattern)], **kw)
    return Params(list(pattern), **kw)


def result_positions(res_ptr) -> np.ndarray:
    """(count, 2) uint64 copy of a match_result_t*'s positions."""
    r = res_ptr.contents if hasattr(res_ptr, "contents") else res_ptr
    n = int(r.count)
    if n == 0:
        return np.zeros((0, 2), dtype=np.uint64)
    buf = C.cast(r.positions, C.POINTER(C.c_uint64 * (2 * n))).contents
    return np.frombuffer(buf, dtype=np.uint64).reshape(n, 2).copy()


def as_char_p(buf) -> C.c_char_p:
    """Pointer to a bytes / numpy uint8 buffer without copying (bytes) or via ctypes (ndarray)."""
    if isinstance(buf, np.ndarray):
        assert buf.dtype == np.uint8 and buf.flags["C_CONTIGUOUS"]
        return C.cast(buf.ctypes.data, C.c_char_p)
    return C.c_char_p(bytes(buf)) if not isinstance(buf, bytes) else C.c_char_p(buf)
