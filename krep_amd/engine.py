"""ctypes binding of include/krep_gpu.h.  Plumbing only — every search runs in libkrep_gpu.so."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libkrep_gpu.so")


class KrepGpuError(RuntimeError):
    pass


class Engine:
    """Thin face over the C-ABI.  Host-buffer operators mirror krep's search_func_t; the device path
    takes raw device pointers (e.g. torch tensors' data_ptr())."""

    def __init__(self, path: str = LIB_PATH):
        path = os.environ.get("KREP_GPU_LIB", path)  # development aid: A/B two builds of the library in one GPU session
        if not os.path.exists(path):
            raise KrepGpuError(f"{path} is missing: build it with `python -m krep_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # When torch is used in the same process (tests, bench.py: device memory + torch.distributed), its
        # bundled HIP runtime must be the one the process binds first; loading ours first makes torch see
        # "No HIP GPUs".  Importing torch here is plumbing only — nothing in the library needs it.
        if not os.environ.get("KREP_GPU_NO_TORCH"):  # (tools/notorch_bench.py: the library on the system HIP runtime alone)
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = self.lib = C.CDLL(path)
        sf = [C.POINTER(abi.SearchParams), C.c_void_p, C.c_size_t, C.POINTER(abi.MatchResult)]
        for n in ("krep_gpu_literal_search", "krep_gpu_aho_corasick_search"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = sf
        L.krep_gpu_select_search_algorithm.restype = C.c_void_p
        L.krep_gpu_select_search_algorithm.argtypes = [C.POINTER(abi.SearchParams)]
        L.search_buffer.restype = C.c_int
        L.search_buffer.argtypes = [C.POINTER(abi.SearchParams), C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                    C.POINTER(abi.MatchResult), C.POINTER(C.c_uint64)]
        L.search_buffer_ex.restype = C.c_int
        L.search_buffer_ex.argtypes = [C.POINTER(abi.SearchParams), C.c_void_p, C.c_size_t, C.POINTER(abi.Config), C.c_int,
                                       C.POINTER(abi.MatchResult), C.POINTER(C.c_uint64)]
        L.krep_gpu_can_accelerate.restype = C.c_int
        L.krep_gpu_can_accelerate.argtypes = [C.POINTER(abi.SearchParams)]
        L.krep_gpu_config_default.restype = None
        L.krep_gpu_config_default.argtypes = [C.POINTER(abi.Config)]
        L.krep_gpu_set_thread_config.restype = None
        L.krep_gpu_set_thread_config.argtypes = [C.POINTER(abi.Config)]
        L.krep_gpu_set_stream_chunk.restype = None
        L.krep_gpu_set_stream_chunk.argtypes = [C.c_size_t]
        L.krep_gpu_release_device_resources.restype = None
        L.krep_gpu_plan_create_ex.restype = C.c_void_p
        L.krep_gpu_plan_create_ex.argtypes = [C.POINTER(abi.SearchParams), C.POINTER(abi.Config)]
        L.krep_gpu_scan_device_ex.restype = C.c_int
        L.krep_gpu_scan_device_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                              C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(abi.ScanOut)]
        L.krep_gpu_scan_device_seq.restype = C.c_int
        L.krep_gpu_scan_device_seq.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                               C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(abi.SeqCarry),
                                               C.POINTER(abi.SeqCarry), C.POINTER(abi.ScanOut)]
        L.krep_gpu_split_mode.restype = C.c_int
        L.krep_gpu_split_mode.argtypes = [C.POINTER(abi.SearchParams), C.c_size_t]
        L.krep_gpu_match_result_init.restype = C.POINTER(abi.MatchResult)
        L.krep_gpu_match_result_init.argtypes = [C.c_uint64]
        L.krep_gpu_match_result_free.restype = None
        L.krep_gpu_match_result_free.argtypes = [C.POINTER(abi.MatchResult)]
        L.krep_gpu_plan_create.restype = C.c_void_p
        L.krep_gpu_plan_create.argtypes = [C.POINTER(abi.SearchParams), C.c_int, C.c_int]
        L.krep_gpu_plan_destroy.restype = None
        L.krep_gpu_plan_destroy.argtypes = [C.c_void_p]
        L.krep_gpu_plan_ref_algo.restype = C.c_int
        L.krep_gpu_plan_ref_algo.argtypes = [C.c_void_p]
        L.krep_gpu_scan_device.restype = C.c_int
        L.krep_gpu_scan_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(abi.ScanOut)]
        L.krep_gpu_generate.restype = C.c_int
        L.krep_gpu_generate.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint64, C.c_void_p,
                                        C.c_size_t, C.c_uint64, C.c_void_p]
        L.krep_gpu_generate_host.restype = None
        L.krep_gpu_generate_host.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint64, C.c_void_p,
                                             C.c_size_t, C.c_uint64]
        L.krep_gpu_combine_line_counts.restype = C.c_uint64
        L.krep_gpu_combine_line_counts.argtypes = [C.POINTER(abi.ScanOut), C.c_int]
        L.krep_gpu_mirror_select.restype = C.c_int
        L.krep_gpu_mirror_select.argtypes = [C.POINTER(abi.SearchParams), C.c_size_t]
        L.krep_gpu_algorithm_name.restype = C.c_char_p
        L.krep_gpu_algorithm_name.argtypes = [C.c_int]
        L.krep_gpu_order_by_start.restype = C.c_int
        L.krep_gpu_order_by_start.argtypes = [C.c_void_p, C.c_uint64, C.c_size_t, C.c_void_p]
        L.krep_gpu_line_numbers.restype = C.c_int
        L.krep_gpu_line_numbers.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        for n in ("krep_gpu_set_reference_simd", "krep_gpu_set_only_matching", "krep_gpu_set_force_no_simd",
                  "krep_gpu_set_algo_override", "krep_gpu_debug_force_rounds", "krep_gpu_debug_force_stage_cap",
                  "krep_gpu_set_result_order", "krep_gpu_set_device", "krep_gpu_set_num_gpus", "krep_gpu_debug_inject_failure"):
            getattr(L, n).restype = None
            getattr(L, n).argtypes = [C.c_int]
        L.krep_gpu_get_reference_simd.restype = C.c_int
        L.krep_gpu_comm_unique_id.restype = C.c_int
        L.krep_gpu_comm_unique_id.argtypes = [C.c_void_p]
        L.krep_gpu_comm_init_rank.restype = C.c_int
        L.krep_gpu_comm_init_rank.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.krep_gpu_comm_allreduce_u64.restype = C.c_int
        L.krep_gpu_comm_allreduce_u64.argtypes = [C.POINTER(C.c_uint64), C.c_int]
        L.krep_gpu_comm_allreduce_device_u64.restype = C.c_int
        L.krep_gpu_comm_allreduce_device_u64.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.krep_gpu_comm_destroy.restype = None
        L.krep_gpu_rccl_calls.restype = C.c_uint64
        L.krep_gpu_rccl_version.restype = C.c_int
        L.krep_gpu_debug_force_single_grid.restype = None
        L.krep_gpu_debug_force_single_grid.argtypes = [C.c_int]
        L.krep_gpu_debug_single_failovers.restype = C.c_uint64
        L.krep_gpu_debug_single_launches.restype = C.c_uint64
        L.krep_gpu_debug_tiny_launches.restype = C.c_uint64
        L.krep_gpu_debug_chain_fixups.restype = None
        L.krep_gpu_debug_chain_fixups.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.krep_gpu_debug_tiny_dense_launches.restype = C.c_uint64
        if hasattr(L, "krep_gpu_debug_anchor_info"):  # (an older build of the library as an A/B partner, tools/ab_bench.py, lacks the round-6 hooks)
            L.krep_gpu_debug_anchored_launches.restype = C.c_uint64
            L.krep_gpu_debug_literal_dma_launches.restype = C.c_uint64
            L.krep_gpu_debug_runs_launches.restype = C.c_uint64
            L.krep_gpu_debug_anchor_info.restype = C.c_int
            L.krep_gpu_debug_anchor_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                                     C.POINTER(C.c_double)]
        if hasattr(L, "krep_gpu_alloc_placed"):
            L.krep_gpu_alloc_placed.restype = C.c_int
            L.krep_gpu_alloc_placed.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(abi.Placement)]
            L.krep_gpu_free_placed.restype = C.c_int
            L.krep_gpu_free_placed.argtypes = [C.c_int, C.c_void_p]
        if hasattr(L, "krep_gpu_debug_literal_dma_state"):
            L.krep_gpu_debug_literal_dma_state.restype = C.c_int
            L.krep_gpu_debug_literal_dma_state.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        if hasattr(L, "krep_gpu_debug_anchor_measured"):
            L.krep_gpu_debug_anchor_measured.restype = C.c_int
            L.krep_gpu_debug_anchor_measured.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.krep_gpu_last_shard_info.restype = None
        L.krep_gpu_last_shard_info.argtypes = [C.POINTER(abi.ShardInfo)]
        L.krep_gpu_available.restype = C.c_int
        L.krep_gpu_unavailable_reason.restype = C.c_char_p
        L.krep_gpu_last_status.restype = C.c_int
        L.krep_gpu_set_cpu_fallback.restype = None
        L.krep_gpu_set_cpu_fallback.argtypes = [C.c_void_p]
        L.krep_gpu_worthwhile.restype = C.c_int
        L.krep_gpu_worthwhile.argtypes = [C.POINTER(abi.SearchParams), C.c_size_t]
        L.krep_gpu_worthwhile_ex.restype = C.c_int
        L.krep_gpu_worthwhile_ex.argtypes = [C.POINTER(abi.SearchParams), C.c_size_t, C.c_int]
        L.krep_gpu_cost_estimate.restype = C.c_int
        L.krep_gpu_cost_estimate.argtypes = [C.POINTER(abi.SearchParams), C.c_size_t, C.c_int, C.POINTER(abi.Cost)]
        L.krep_gpu_get_cost_rates.restype = None
        L.krep_gpu_get_cost_rates.argtypes = [C.POINTER(abi.CostRates)]
        L.krep_gpu_set_cost_rates.restype = None
        L.krep_gpu_set_cost_rates.argtypes = [C.POINTER(abi.CostRates)]
        L.krep_gpu_set_min_text_bytes.restype = None
        L.krep_gpu_set_min_text_bytes.argtypes = [C.c_size_t]
        L.krep_gpu_device_count.restype = C.c_int
        L.krep_gpu_last_error.restype = C.c_char_p
        L.krep_gpu_clear_error.restype = None
        L.krep_gpu_version.restype = C.c_char_p

    # ---- configuration of the reference mirror ----
    def set_reference_simd(self, level: int):
        self.lib.krep_gpu_set_reference_simd(level)

    def set_only_matching(self, on: bool):
        self.lib.krep_gpu_set_only_matching(int(on))

    def set_force_no_simd(self, on: bool):
        self.lib.krep_gpu_set_force_no_simd(int(on))

    def set_algo_override(self, a: int):
        self.lib.krep_gpu_set_algo_override(a)

    def force_rounds(self, r: int):
        self.lib.krep_gpu_debug_force_rounds(r)

    def force_single_grid(self, blocks: int):
        self.lib.krep_gpu_debug_force_single_grid(blocks)

    def single_failovers(self) -> int:
        return int(self.lib.krep_gpu_debug_single_failovers())

    def single_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_single_launches())

    def chain_fixups(self):
        """(pieces scanned again, end pieces that only re-ran the end-of-text replay) since the process started"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.lib.krep_gpu_debug_chain_fixups(C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def tiny_dense_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_tiny_dense_launches())

    def runs_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_runs_launches())

    def literal_dma_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_literal_dma_launches())

    def alloc_placed(self, text_bytes: int, record_bytes: int, tries: int = 3, device: int = 0):
        """krep_gpu_alloc_placed(): (d_text, d_records, abi.Placement) — one block, its placement drawn for; free_placed(d_text)"""
        t, r, info = C.c_void_p(0), C.c_void_p(0), abi.Placement()
        if self.lib.krep_gpu_alloc_placed(device, text_bytes, record_bytes, tries, C.byref(t), C.byref(r), C.byref(info)):
            raise KrepGpuError("krep_gpu_alloc_placed failed: " + self.last_error())
        return int(t.value), int(r.value or 0), info

    def free_placed(self, d_text: int, device: int = 0):
        if self.lib.krep_gpu_free_placed(device, C.c_void_p(d_text)):
            raise KrepGpuError("krep_gpu_free_placed failed: " + self.last_error())

    def anchored_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_anchored_launches())

    def tiny_launches(self) -> int:
        return int(self.lib.krep_gpu_debug_tiny_launches())

    def force_stage_cap(self, c: int):
        self.lib.krep_gpu_debug_force_stage_cap(c)

    def default_config(self) -> abi.Config:
        c = abi.Config()
        self.lib.krep_gpu_config_default(C.byref(c))
        return c

    def set_thread_config(self, cfg: "abi.Config | None"):
        self.lib.krep_gpu_set_thread_config(C.byref(cfg) if cfg is not None else None)

    def set_stream_chunk(self, nbytes: int):
        self.lib.krep_gpu_set_stream_chunk(nbytes)

    def can_accelerate(self, params: abi.Params) -> bool:
        return bool(self.lib.krep_gpu_can_accelerate(params.ref))

    def select(self, params: abi.Params):
        """krep_gpu_select_search_algorithm(): the operator's address, or None (the caller keeps its CPU function)."""
        return self.lib.krep_gpu_select_search_algorithm(params.ref)

    # ---- availability, failure status, CPU fallback (SURVEY §8b "Errors") ----
    def available(self) -> bool:
        return bool(self.lib.krep_gpu_available())

    def unavailable_reason(self) -> str:
        return (self.lib.krep_gpu_unavailable_reason() or b"").decode()

    def last_status(self) -> int:
        return int(self.lib.krep_gpu_last_status())

    def set_cpu_fallback(self, selector_address):
        """selector_address: address of a `search_func_t (*)(const search_params_t *)` (or None to unregister)."""
        self.lib.krep_gpu_set_cpu_fallback(C.c_void_p(selector_address) if selector_address else None)

    def worthwhile(self, params: abi.Params, text_len: int) -> bool:
        return bool(self.lib.krep_gpu_worthwhile(params.ref, text_len))

    def worthwhile_ex(self, params: abi.Params, text_len: int, cpu_threads: int) -> bool:
        return bool(self.lib.krep_gpu_worthwhile_ex(params.ref, text_len, cpu_threads))

    def cost_estimate(self, params: abi.Params, text_len: int, cpu_threads: int = 0) -> "abi.Cost":
        c = abi.Cost()
        if self.lib.krep_gpu_cost_estimate(params.ref, text_len, cpu_threads, C.byref(c)):
            raise KrepGpuError("krep_gpu_cost_estimate failed")
        return c

    def cost_rates(self) -> "abi.CostRates":
        r = abi.CostRates()
        self.lib.krep_gpu_get_cost_rates(C.byref(r))
        return r

    def set_cost_rates(self, rates: "abi.CostRates | None"):
        self.lib.krep_gpu_set_cost_rates(C.byref(rates) if rates is not None else None)

    def inject_failure(self, kind: int):
        self.lib.krep_gpu_debug_inject_failure(kind)

    def set_num_gpus(self, n: int):
        self.lib.krep_gpu_set_num_gpus(n)

    # ---- the collective of the multi-GPU path, C level (kg_comm.hip: RCCL, dlopen'd on first use) ----
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        if self.lib.krep_gpu_comm_unique_id(buf):
            raise KrepGpuError("krep_gpu_comm_unique_id failed: " + self.last_error())
        return buf.raw

    def comm_init_rank(self, id128: bytes, nranks: int, rank: int, device: int):
        assert len(id128) == 128
        buf = C.create_string_buffer(id128, 128)
        if self.lib.krep_gpu_comm_init_rank(buf, nranks, rank, device):
            raise KrepGpuError("krep_gpu_comm_init_rank failed: " + self.last_error())

    def comm_allreduce(self, values):
        """ONE ncclAllReduce(uint64, sum) of the rank's counters; returns the summed values."""
        arr = (C.c_uint64 * len(values))(*[int(v) for v in values])
        if self.lib.krep_gpu_comm_allreduce_u64(arr, len(values)):
            raise KrepGpuError("krep_gpu_comm_allreduce_u64 failed: " + self.last_error())
        return [int(v) for v in arr]

    def comm_destroy(self):
        self.lib.krep_gpu_comm_destroy()

    def rccl_calls(self) -> int:
        return int(self.lib.krep_gpu_rccl_calls())

    def rccl_version(self) -> int:
        return int(self.lib.krep_gpu_rccl_version())

    def last_shard_info(self) -> "abi.ShardInfo":
        """Shards / physical devices / communicator ranks of the calling thread's last sharded host search."""
        info = abi.ShardInfo()
        self.lib.krep_gpu_last_shard_info(C.byref(info))
        return info

    def split_mode(self, params: abi.Params, text_len: int) -> int:
        return int(self.lib.krep_gpu_split_mode(params.ref, text_len))

    def release_device_resources(self):
        self.lib.krep_gpu_release_device_resources()

    def mirror_select(self, params: abi.Params, text_len: int) -> int:
        return int(self.lib.krep_gpu_mirror_select(params.ref, text_len))

    def last_error(self) -> str:
        return (self.lib.krep_gpu_last_error() or b"").decode()

    def device_count(self) -> int:
        return int(self.lib.krep_gpu_device_count())

    # ---- formatter-side post-processing on the device (krep.c:3018-3023, :589-668) ----
    def set_result_order(self, by_start: bool):
        self.lib.krep_gpu_set_result_order(int(by_start))

    def order_by_start(self, d_positions: int, n: int, text_len: int, stream: int = 0):
        if self.lib.krep_gpu_order_by_start(C.c_void_p(d_positions), n, text_len, C.c_void_p(stream)):
            raise KrepGpuError("krep_gpu_order_by_start failed: " + self.last_error())

    def line_numbers(self, d_text: int, text_len: int, d_positions: int, n: int, d_lines: int, stream: int = 0):
        if self.lib.krep_gpu_line_numbers(C.c_void_p(d_text), text_len, C.c_void_p(d_positions), n, C.c_void_p(d_lines),
                                          C.c_void_p(stream)):
            raise KrepGpuError("krep_gpu_line_numbers failed: " + self.last_error())

    # ---- search_func_t-shaped operators on host buffers ----
    def _ptr(self, text):
        if isinstance(text, np.ndarray):
            assert text.dtype == np.uint8 and text.flags["C_CONTIGUOUS"]
            return C.c_void_p(text.ctypes.data), text.size, text
        raw = bytes(text)
        cp = C.c_char_p(raw)
        return C.cast(cp, C.c_void_p), len(raw), (raw, cp)

    def search(self, params: abi.Params, text, want_result=True):
        """krep_gpu_select_search_algorithm(params)(params, text, len, result) -> (ret, positions)."""
        ptr, n, keep = self._ptr(text)
        fn = self.lib.krep_gpu_aho_corasick_search if params.s.num_patterns > 1 else self.lib.krep_gpu_literal_search
        res = self.lib.krep_gpu_match_result_init(16) if want_result else None
        dummy = C.c_int(0)
        if params.s.num_patterns > 1 and not params.s.ac_trie:
            params.s.ac_trie = C.cast(C.pointer(dummy), C.c_void_p)  # "caller pre-built the trie" (krep.c:2528)
        try:
            self.lib.krep_gpu_clear_error()
            ret = fn(params.ref, ptr, n, res)
            if self.last_status() == abi.STATUS_FAILED:
                raise KrepGpuError(self.last_error() or "krep-gpu operator failed")
            pos = abi.result_positions(res) if res else None
        finally:
            if res:
                self.lib.krep_gpu_match_result_free(res)
            params.s.ac_trie = None
        del keep
        return int(ret), pos

    def search_buffer(self, params: abi.Params, text, only_matching=False, num_gpus=1, want_result=True, cfg=None):
        ptr, n, keep = self._ptr(text)
        res = self.lib.krep_gpu_match_result_init(16) if want_result else None
        cnt = C.c_uint64(0)
        try:
            if cfg is not None:
                rc = self.lib.search_buffer_ex(params.ref, ptr, n, C.byref(cfg), num_gpus, res, C.byref(cnt))
            else:
                rc = self.lib.search_buffer(params.ref, ptr, n, int(only_matching), num_gpus, res, C.byref(cnt))
            pos = abi.result_positions(res) if res else None
        finally:
            if res:
                self.lib.krep_gpu_match_result_free(res)
        del keep
        return int(rc), int(cnt.value), pos

    # ---- device-resident path ----
    def plan(self, params: abi.Params, only_matching=False, device=0) -> "Plan":
        h = self.lib.krep_gpu_plan_create(params.ref, int(only_matching), device)
        if not h:
            raise KrepGpuError("krep_gpu_plan_create failed: " + self.last_error())
        return Plan(self, h, params)

    def generate(self, d_ptr: int, length: int, global_off: int, kind: int, seed: int, plant: bytes = b"",
                 period: int = 0, stream: int = 0):
        rc = self.lib.krep_gpu_generate(C.c_void_p(d_ptr), length, global_off, kind, seed, plant, len(plant), period,
                                        C.c_void_p(stream))
        if rc:
            raise KrepGpuError("krep_gpu_generate failed: " + self.last_error())

    def generate_host(self, length: int, global_off: int, kind: int, seed: int, plant: bytes = b"",
                      period: int = 0) -> np.ndarray:
        out = np.empty(length, dtype=np.uint8)
        self.lib.krep_gpu_generate_host(C.c_void_p(out.ctypes.data), length, global_off, kind, seed, plant, len(plant),
                                        period)
        return out


class Plan:
    def __init__(self, eng: Engine, handle, params):
        self.eng, self.h, self.params = eng, handle, params

    @property
    def ref_algo(self) -> int:
        return int(self.eng.lib.krep_gpu_plan_ref_algo(self.h))

    def scan(self, d_text: int, text_len: int, own_lo=0, own_hi=None, global_base=0, d_positions: int = 0,
             capacity: int = 0, stream: int = 0, time_it=False, global_len=0) -> abi.ScanOut:
        out = abi.ScanOut()
        rc = self.eng.lib.krep_gpu_scan_device_ex(self.h, C.c_void_p(d_text), text_len, own_lo,
                                                  text_len if own_hi is None else own_hi, global_base, global_len,
                                                  C.c_void_p(d_positions) if d_positions else None, capacity,
                                                  C.c_void_p(stream) if stream else None, int(time_it), C.byref(out))
        if rc:
            raise KrepGpuError("krep_gpu_scan_device failed: " + self.eng.last_error())
        return out

    def anchor_info(self):
        """(state 0 undecided / 1 end grams / 2 anchored, patterns moved, est. candidate rate end grams, ... anchors) — multi-pattern plans"""
        st, mv, r0, r1 = C.c_int(0), C.c_uint32(0), C.c_double(0), C.c_double(0)
        if self.eng.lib.krep_gpu_debug_anchor_info(self.h, C.byref(st), C.byref(mv), C.byref(r0), C.byref(r1)):
            return None
        return int(st.value), int(mv.value), float(r0.value), float(r1.value)

    def literal_dma_state(self):
        """(text sampled?, LDS-DMA kernel barred for it?, share of 1-KiB cells its prefilter passed) — kg_scan.hip lit_pass"""
        a, b, r = C.c_int(0), C.c_int(0), C.c_double(0)
        if self.eng.lib.krep_gpu_debug_literal_dma_state(self.h, C.byref(a), C.byref(b), C.byref(r)):
            return None
        return bool(a.value), bool(b.value), float(r.value)

    def split_state(self) -> int:
        """0 undecided / 1 one dictionary / 2 split into a >= 4-byte part and a 1..3-byte part (kg_scan.hip scan_ac_split)"""
        f = self.eng.lib.krep_gpu_debug_split_state
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def anchor_measured(self):
        """(candidates per tested position the last general-kernel scan counted, decisions re-opened so far)"""
        m, r = C.c_double(0), C.c_int(0)
        if self.eng.lib.krep_gpu_debug_anchor_measured(self.h, C.byref(m), C.byref(r)):
            return None
        return float(m.value), int(r.value)

    def scan_seq(self, d_text: int, text_len: int, own_lo, own_hi, global_base=0, d_positions: int = 0, capacity: int = 0,
                 global_len=0, carry_in: "abi.SeqCarry | None" = None):
        """krep_gpu_scan_device_seq(): one piece of a text scanned in text order -> (ScanOut, SeqCarry it leaves)."""
        out, cout = abi.ScanOut(), abi.SeqCarry()
        rc = self.eng.lib.krep_gpu_scan_device_seq(self.h, C.c_void_p(d_text), text_len, own_lo, own_hi, global_base, global_len,
                                                   C.c_void_p(d_positions) if d_positions else None, capacity, None, 0,
                                                   C.byref(carry_in) if carry_in is not None else None, C.byref(cout),
                                                   C.byref(out))
        if rc:
            raise KrepGpuError("krep_gpu_scan_device_seq failed: " + self.eng.last_error())
        return out, cout

    def close(self):
        if self.h:
            self.eng.lib.krep_gpu_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_engine = None


def load() -> Engine:
    global _engine
    if _engine is None:
        _engine = Engine()
    return _engine
