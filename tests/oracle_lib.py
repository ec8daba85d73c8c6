"""Test-side loaders for the parity checkers (oracle/ restatement and oracle/_ref compiled reference).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krep_amd import abi  # noqa: E402

ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_SF_ARGS = [C.POINTER(abi.SearchParams), C.c_void_p, C.c_size_t, C.POINTER(abi.MatchResult)]


def _cpu_flags() -> set:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def build_oracle() -> None:
    """Compile oracle/liboracle_krep.so (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.DEVNULL)


class TextBuf:
    """Keeps a haystack alive and hands out a raw pointer (bytes or uint8 ndarray, zero-copy)."""

    def __init__(self, data):
        if isinstance(data, np.ndarray):
            assert data.dtype == np.uint8
            self.arr = np.ascontiguousarray(data)
            self.ptr = C.c_void_p(self.arr.ctypes.data)
            self.n = self.arr.size
        else:
            self.raw = bytes(data)
            self._cp = C.c_char_p(self.raw)
            self.ptr = C.cast(self._cp, C.c_void_p)
            self.n = len(self.raw)


class _Engine:
    """Uniform face over a library exporting krep's search_func_t operators."""

    name = "?"
    lib = None
    fn = {}
    init_name = add_name = free_name = None
    ac_build = ac_free = None

    def _setup(self):
        L = self.lib
        for f in self.fn.values():
            g = getattr(L, f)
            g.restype = C.c_uint64
            g.argtypes = _SF_ARGS
        init = getattr(L, self.init_name)
        init.restype = C.POINTER(abi.MatchResult)
        init.argtypes = [C.c_uint64]
        free = getattr(L, self.free_name)
        free.restype = None
        free.argtypes = [C.POINTER(abi.MatchResult)]
        self._init, self._free = init, free
        b = getattr(L, self.ac_build)
        b.restype = C.c_void_p
        b.argtypes = [C.POINTER(abi.SearchParams)]
        fr = getattr(L, self.ac_free)
        fr.restype = None
        fr.argtypes = [C.c_void_p]
        self._acb, self._acf = b, fr

    def has(self, algo: int) -> bool:
        return algo in self.fn

    def call(self, algo: int, params: abi.Params, text, want_result=True):
        """-> (returned count, positions[(n,2) uint64] or None).  Single chunk, whole buffer."""
        tb = text if isinstance(text, TextBuf) else TextBuf(text)
        trie = None
        if algo == abi.RA_AHO_CORASICK:
            trie = self._acb(params.ref)
            params.s.ac_trie = trie
        res = self._init(16) if want_result else None
        try:
            ret = getattr(self.lib, self.fn[algo])(params.ref, tb.ptr, tb.n, res)
            pos = abi.result_positions(res) if res else None
        finally:
            if res:
                self._free(res)
            if trie:
                self._acf(trie)
                params.s.ac_trie = None
        return int(ret), pos


class OracleEngine(_Engine):
    name = "oracle"
    fn = {abi.RA_BMH: "ko_boyer_moore_search", abi.RA_KMP: "ko_kmp_search",
          abi.RA_MEMCHR: "ko_memchr_search", abi.RA_MEMCHR_SHORT: "ko_memchr_short_search",
          abi.RA_SSE42: "ko_sse42_search", abi.RA_AVX2: "ko_avx2_search",
          abi.RA_AVX512: "ko_avx512_search", abi.RA_NEON: "ko_neon_search",
          abi.RA_AHO_CORASICK: "ko_aho_corasick_search"}
    init_name, free_name = "ko_result_init", "ko_result_free"
    ac_build, ac_free = "ko_ac_trie_build", "ko_ac_trie_free"

    def __init__(self):
        path = os.environ.get("KREP_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle_krep.so")  # (tools/sanitize.py: an ASan build)
        if not os.path.exists(path):
            subprocess.run(["make", "-C", ORACLE_DIR, "liboracle_krep.so"], check=True,
                           stdout=subprocess.DEVNULL)
        self.lib = C.CDLL(path)
        self._setup()
        self.lib.ko_select.restype = C.c_int
        self.lib.ko_select.argtypes = [C.POINTER(abi.SearchParams), C.c_int]
        self.lib.ko_set_only_matching.argtypes = [C.c_int]
        self.lib.ko_set_force_no_simd.argtypes = [C.c_int]
        self.lib.ko_set_algo_override.argtypes = [C.c_int]
        self.lib.ko_ac_num_states.restype = C.c_uint64
        self.lib.ko_ac_num_states.argtypes = [C.c_void_p]

    def select(self, params: abi.Params, simd: int) -> int:
        return int(self.lib.ko_select(params.ref, simd))

    def set_only_matching(self, on: bool):
        self.lib.ko_set_only_matching(int(on))


_REF_FILES = {abi.REF_SCALAR: "libkrep_ref_scalar.so", abi.REF_SSE42: "libkrep_ref_sse42.so",
              abi.REF_AVX2: "libkrep_ref_avx2.so", abi.REF_AVX512: "libkrep_ref_avx512.so",
              abi.REF_NEON: "libkrep_ref_neon.so"}  # NEON: the arm64 path built against oracle/neon_shim (plain C)
_REF_NEEDS = {abi.REF_SCALAR: set(), abi.REF_SSE42: {"sse4_2"}, abi.REF_AVX2: {"avx2", "sse4_2"},
              abi.REF_AVX512: {"avx512f", "avx512bw", "avx2"}, abi.REF_NEON: set()}


class RefEngine(_Engine):
    """The unmodified reference, compiled by oracle/Makefile into oracle/_ref/."""

    init_name, free_name = "match_result_init", "match_result_free"
    ac_build, ac_free = "ac_trie_build", "ac_trie_free"

    def __init__(self, level: int):
        self.level = level
        self.name = "ref:" + _REF_FILES[level]
        self.lib = C.CDLL(os.path.join(REF_DIR, _REF_FILES[level]))
        fn = {abi.RA_BMH: "boyer_moore_search", abi.RA_KMP: "kmp_search",
              abi.RA_MEMCHR: "memchr_search", abi.RA_MEMCHR_SHORT: "memchr_short_search",
              abi.RA_AHO_CORASICK: "aho_corasick_search"}
        if level == abi.REF_NEON:
            fn[abi.RA_NEON] = "neon_search"
        if abi.REF_SSE42 <= level <= abi.REF_AVX512:
            fn[abi.RA_SSE42] = "simd_sse42_search"
        if abi.REF_AVX2 <= level <= abi.REF_AVX512:
            fn[abi.RA_AVX2] = "simd_avx2_search"
        if level == abi.REF_AVX512:
            fn[abi.RA_AVX512] = "simd_avx512_search"
        self.fn = fn
        self._setup()
        self.lib.select_search_algorithm.restype = C.c_void_p
        self.lib.select_search_algorithm.argtypes = [C.POINTER(abi.SearchParams)]
        self.lib.get_algorithm_name.restype = C.c_char_p
        self.lib.get_algorithm_name.argtypes = [C.c_void_p]

    def select(self, params: abi.Params) -> int:
        """enum krep_ref_algo of the pointer select_search_algorithm() returns."""
        p = self.lib.select_search_algorithm(params.ref)
        for algo, name in self.fn.items():
            if C.cast(getattr(self.lib, name), C.c_void_p).value == p:
                return algo
        return abi.RA_REGEX if params.s.use_regex else abi.RA_NONE


def ref_available(level: int) -> bool:
    return os.path.exists(os.path.join(REF_DIR, _REF_FILES[level])) and _REF_NEEDS[level] <= _cpu_flags()


_cache = {}


def oracle() -> OracleEngine:
    if "o" not in _cache:
        _cache["o"] = OracleEngine()
    return _cache["o"]


def ref(level: int):
    if not ref_available(level):
        return None
    if level not in _cache:
        _cache[level] = RefEngine(level)
    return _cache[level]


# which compiled reference build a direct function call goes to: every _ref library holds the same source, so any build
# that exports the function is the reference for it; spread over all five so that each is exercised
_DIRECT = {abi.RA_BMH: (abi.REF_SCALAR, abi.REF_SSE42), abi.RA_KMP: (abi.REF_SCALAR, abi.REF_SSE42),
           abi.RA_MEMCHR: (abi.REF_SSE42, abi.REF_SCALAR), abi.RA_MEMCHR_SHORT: (abi.REF_SSE42, abi.REF_SCALAR),
           abi.RA_SSE42: (abi.REF_SSE42, abi.REF_AVX2), abi.RA_AVX2: (abi.REF_AVX2, abi.REF_AVX512),
           abi.RA_AHO_CORASICK: (abi.REF_AVX2, abi.REF_SCALAR), abi.RA_AVX512: (abi.REF_AVX512,),
           abi.RA_NEON: (abi.REF_NEON,)}


class Checker:
    """What the `-m gpu` parity tests compare the HIP path with: the UNMODIFIED reference compiled into oracle/_ref
    (called function by function), and the restatement only where the compiled reference cannot be asked — under the
    file-static `only_matching` (-o: not settable through a shared object; pinned against the stock CLI in
    tests/test_oracle_cli_only_matching.py), or where a build is absent / needs CPU features this host lacks.
    `direct_calls` / `restatement_calls` say which one answered."""

    def __init__(self):
        self.o = oracle()
        self._om = False
        self.direct_calls = 0
        self.restatement_calls = 0
        self.om_calls = 0           # restatement calls made under only_matching (the one legitimate reason on a full host)
        self.absent_calls = []      # (algo) the restatement answered because no compiled build exports / can run the function
        self.used = set()

    def _direct(self, algo):
        if self._om:
            return None
        for level in _DIRECT.get(algo, ()):
            r = ref(level)
            if r is not None and r.has(algo):
                return r
        return None

    def call(self, algo, params, text, want_result=True):
        r = self._direct(algo)
        if r is not None:
            self.direct_calls += 1
            self.used.add(r.name)
            return r.call(algo, params, text, want_result)
        self.restatement_calls += 1
        if self._om:
            self.om_calls += 1
        else:
            self.absent_calls.append(algo)
        return self.o.call(algo, params, text, want_result)

    def restatement_budget_ok(self) -> bool:
        """True when the restatement answered ONLY under -o, or for a function whose compiled builds this host cannot
        run (no AVX-512 / a missing oracle/_ref) — VERDICT r03 weak #3: the share of the restatement must stay bounded."""
        for algo in self.absent_calls:
            if any(ref_available(level) for level in _DIRECT.get(algo, ())):
                return False
        return self.restatement_calls == self.om_calls + len(self.absent_calls)

    def set_only_matching(self, on: bool):
        self._om = bool(on)
        self.o.set_only_matching(on)

    def __getattr__(self, name):  # select(), lib, fn, ... of the restatement
        return getattr(self.o, name)


def checker() -> Checker:
    if "chk" not in _cache:
        _cache["chk"] = Checker()
    return _cache["chk"]


def ref_cli() -> str | None:
    p = os.path.join(REF_DIR, "krep")
    return p if os.path.exists(p) and {"avx2", "sse4_2"} <= _cpu_flags() else None
