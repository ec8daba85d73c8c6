#!/usr/bin/env python3
"""Builds the reference krep CLI with the MI355X backend wired in — the "drops into the existing CLI" proof.

Nothing of the reference is stored in this repository: the script reads /root/reference/krep.c, applies the
small edits described in INTEGRATION.md to a TEMPORARY copy (regex anchors, no context lines kept
here), compiles it together with the untouched aho_corasick.c and links libkrep_gpu.so.  Output:
oracle/_ref/krep_gpu_cli (git-ignored; travels to the GPU box with the other prebuilt checker binaries).

    KREP_GPU=1 oracle/_ref/krep_gpu_cli -c Sherlock file      # scan on the GPU (CPU function if there is none / it fails)
             oracle/_ref/krep_gpu_cli -c Sherlock file      # unchanged CPU path
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("KREP_REF", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref", "krep_gpu_cli")


def insert_after(src: str, pattern: str, text: str, flags=re.M) -> str:
    m = re.search(pattern, src, flags=flags)
    if not m:
        raise SystemExit(f"anchor not found: {pattern}")
    return src[:m.end()] + text + src[m.end():]


def main():
    if not os.path.exists(os.path.join(REF, "krep.c")):
        print(f"{REF}/krep.c not found - nothing to do")
        return 0
    src = open(os.path.join(REF, "krep.c")).read()
    # 1. the backend's header, after krep.h (it reuses krep's own types when KREP_H is defined)
    src = insert_after(src, r'^#include "aho_corasick\.h".*$', '\n#ifdef KREP_WITH_GPU\n#include "krep_gpu.h"\n#endif\n')
    # 2. the switch, next to the other file-static option globals (krep.c:117-120), and two selectors built on the
    #    reference's own select_search_algorithm():
    #      krep_cpu_select()  — the GPU branch bypassed: what the backend calls when an operator fails at run time
    #                           (krep_gpu_set_cpu_fallback) and what steps 7/8 re-select with;
    #      krep_select_for()  — per text: the GPU operator only when the text is worth a device round trip
    #                           (krep_gpu_worthwhile: device usable, input class reproduced, size >= threshold).
    globals_ = r"""
#ifdef KREP_WITH_GPU
static bool use_gpu = false;             /* KREP_GPU=1 */
static __thread bool gpu_bypass = false; /* this thread is selecting a CPU function */
static search_func_t krep_cpu_select(const search_params_t *p)
{
    const bool saved = gpu_bypass;
    gpu_bypass = true;
    search_func_t f = select_search_algorithm(p);
    gpu_bypass = saved;
    return f;
}
static __thread int krep_cpu_threads_hint = 0; /* threads the CPU path would use for this text (0: search_file()'s own policy) */
static search_func_t krep_select_for(const search_params_t *p, size_t text_len)
{
    /* size, input class, the backend's cost model (GPU iff init + launch + bytes / PCIe rate beats bytes / the CPU function's
       rate on `threads` threads) and the device, in that order */
    if (!use_gpu || p->use_regex || !krep_gpu_worthwhile_ex(p, text_len, krep_cpu_threads_hint))
        return krep_cpu_select(p);
    return select_search_algorithm(p);
}
static bool krep_is_gpu_fn(search_func_t f) { return f == krep_gpu_literal_search || f == krep_gpu_aho_corasick_search; }
#endif
"""
    src = insert_after(src, r'^static bool force_no_simd = false;.*$', globals_)
    # 3. select_search_algorithm(): hand out the GPU operator (krep.c:1771)
    gpu_select = r"""
#ifdef KREP_WITH_GPU
    if (use_gpu && !gpu_bypass && !params->use_regex)
    {
        krep_gpu_set_reference_simd(KREP_USE_AVX512 ? KREP_REF_AVX512 : KREP_USE_AVX2 ? KREP_REF_AVX2
                                    : KREP_USE_SSE42 ? KREP_REF_SSE42 : KREP_USE_NEON ? KREP_REF_NEON : KREP_REF_SCALAR);
        krep_gpu_set_only_matching(only_matching);
        krep_gpu_set_result_order(1); /* records come back in (start, end) order: step 6 skips the qsort */
        krep_gpu_set_force_no_simd(force_no_simd);
        krep_gpu_set_algo_override(!algo_override || !strcmp(algo_override, "auto") ? KREP_ALGO_AUTO
                                   : !strcmp(algo_override, "bm") ? KREP_ALGO_BM
                                   : !strcmp(algo_override, "kmp") ? KREP_ALGO_KMP : KREP_ALGO_AUTO);
        /* NULL: no usable device, or an input class the backend leaves to the CPU -> the CPU function below */
        search_func_t gpu_fn = krep_gpu_select_search_algorithm(params);
        if (gpu_fn)
            return gpu_fn;
    }
#endif
"""
    src = insert_after(src, r'^search_func_t select_search_algorithm\(const search_params_t \*params\)\s*\{', gpu_select)
    # 4. per file (krep.c:2729-2744, :2849): a file that goes to the GPU is ONE chunk — the backend shards internally with
    #    start-offset ownership; a file below the size threshold keeps the CPU function and the reference's own chunking
    #    (krep.c:2404-2420 special-cases small files too: `krep -r` must not pay a device round trip per small file)
    one_chunk = r"""
#ifdef KREP_WITH_GPU
    krep_cpu_threads_hint = actual_thread_count; /* what the CPU path would run this file on (krep.c:2729-2744) */
    if (krep_is_gpu_fn(krep_select_for(&current_params, file_size)))
        actual_thread_count = 1;
#endif
"""
    src = insert_after(src, r'^\s*if \(actual_thread_count <= 0\)\s*\n\s*actual_thread_count = 1;', one_chunk)
    pat = r'search_func_t preselected_algo = select_search_algorithm\(&current_params\);'
    if not re.search(pat, src):
        raise SystemExit("anchor not found: preselected_algo")
    src = re.sub(pat, '\n#ifdef KREP_WITH_GPU\n    search_func_t preselected_algo = krep_select_for(&current_params, file_size);\n#else\n'
                      '    search_func_t preselected_algo = select_search_algorithm(&current_params);\n#endif\n', src, count=1)
    #    ... and search_string() (krep.c:2166) — `krep -s`, stdin
    pat = r'search_func_t search_algo = select_search_algorithm\(&current_params\);'
    if not re.search(pat, src):
        raise SystemExit("anchor not found: search_string's selection")
    src = re.sub(pat, '\n#ifdef KREP_WITH_GPU\n    krep_cpu_threads_hint = 1; /* search_string() is single-threaded */\n    search_func_t search_algo = krep_select_for(&current_params, text_len);\n#else\n'
                      '    search_func_t search_algo = select_search_algorithm(&current_params);\n#endif\n', src, count=1)
    # 5. the CLI switch: environment variable, read at the top of main() (krep.c:3451).  No device is touched here — a run
    #    over small files never initialises the HIP runtime: krep_gpu_worthwhile() looks at the size first, and without a
    #    usable device it (and the selector) say no, so every path is the reference's own.  The backend learns the CPU
    #    selector so that a run-time failure of an operator is answered by the CPU function instead of "no match".
    main_switch = r"""
#ifdef KREP_WITH_GPU
    use_gpu = getenv("KREP_GPU") != NULL;
    if (use_gpu && !getenv("KREP_GPU_NO_FALLBACK_HOOK")) /* (test switch: exercises steps 7/8 instead) */
        krep_gpu_set_cpu_fallback(krep_cpu_select);
#endif
"""
    src = insert_after(src, r'^int main\(int argc, char \*argv\[\]\)\s*\{', main_switch)
    # 6. the records of the GPU operators arrive in compare_match_positions order (sorted in HBM): no host qsort
    #    (krep.c:3020-3023).  Only when a GPU operator really produced them (or its registered CPU fallback, which the
    #    backend sorts): after step 7's re-run the records are a CPU function's and are sorted here as always.
    pat = r'if \(global_matches->count > 1\)(\s*\{\s*qsort\(global_matches->positions)'
    if not re.search(pat, src):
        raise SystemExit("anchor not found: qsort of the global match list")
    src = re.sub(pat, r'if (global_matches->count > 1\n#ifdef KREP_WITH_GPU\n            && !(krep_is_gpu_fn(preselected_algo) && krep_gpu_last_status() != KREP_GPU_FAILED)\n#endif\n            )\1', src, count=1)
    # 7. search_chunk_thread() (krep.c:1944-1956): an operator that could not look (status KREP_GPU_FAILED: no fallback was
    #    registered, or the registered one could not be used) must never be reported as "no match" — re-run the chunk with
    #    the CPU pointer, the reference's own `if (!search_algo) search_algo = select_search_algorithm(...)` idiom.
    rerun = r"""
#ifdef KREP_WITH_GPU
    if (krep_is_gpu_fn(search_algo) && krep_gpu_last_status() == KREP_GPU_FAILED)
    {
        if (local_result)
            local_result->count = 0;
        search_algo = krep_cpu_select(data->params);
        count_result = search_algo(data->params, data->chunk_start, data->chunk_len, local_result);
    }
#endif
"""
    src = insert_after(src, r'count_result = search_algo\(data->params,\s*data->chunk_start,\s*data->chunk_len,\s*local_result\);[^\n]*', rerun)
    # 8. search_string() (krep.c:2169): the same re-run for `krep -s` / stdin
    rerun_s = r"""
#ifdef KREP_WITH_GPU
    if (krep_is_gpu_fn(search_algo) && krep_gpu_last_status() == KREP_GPU_FAILED)
    {
        if (matches)
            matches->count = 0;
        search_algo = krep_cpu_select(&current_params);
        final_count = search_algo(&current_params, text, text_len, matches);
    }
#endif
"""
    src = insert_after(src, r'final_count = search_algo\(&current_params, text, text_len, matches\);', rerun_s)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        patched = os.path.join(td, "krep_gpu_patched.c")
        open(patched, "w").write(src)
        if os.environ.get("KREP_GPU_KEEP_PATCHED"):
            open(os.environ["KREP_GPU_KEEP_PATCHED"], "w").write(src)
        cmd = ["gcc", "-O2", "-std=c11", "-D_GNU_SOURCE", "-D_DEFAULT_SOURCE", "-pthread", "-w", "-mavx2", "-msse4.2",
               "-DKREP_WITH_GPU", f"-I{REF}", f"-I{os.path.join(ROOT, 'include')}", patched,
               os.path.join(REF, "aho_corasick.c"), f"-L{os.path.join(ROOT, 'krep_amd', 'lib')}", "-lkrep_gpu",
               "-Wl,-rpath,$ORIGIN/../../krep_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", OUT]
        subprocess.run(cmd, check=True)
    print("built", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
