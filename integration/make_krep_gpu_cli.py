#!/usr/bin/env python3
"""Builds the reference krep CLI with the MI355X backend wired in — the "drops into the existing CLI" proof.

Nothing of the reference is stored in this repository: the script reads /root/reference/krep.c, applies the
six small edits described in INTEGRATION.md to a TEMPORARY copy (regex anchors, no context lines kept
here), compiles it together with the untouched aho_corasick.c and links libkrep_gpu.so.  Output:
oracle/_ref/krep_gpu_cli (git-ignored; travels to the GPU box with the other prebuilt checker binaries).

    KREP_GPU=1 oracle/_ref/krep_gpu_cli -c Sherlock file      # scan on the GPU
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


def insert_after(src: str, pattern: str, text: str, count=1) -> str:
    m = re.search(pattern, src, flags=re.M)
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
    # 2. the switch, next to the other file-static option globals (krep.c:117-120)
    src = insert_after(src, r'^static bool force_no_simd = false;.*$', '\nstatic bool use_gpu = false; /* KREP_GPU=1 */\n')
    # 3. select_search_algorithm(): hand out the GPU operator (krep.c:1771)
    gpu_select = r'''
#ifdef KREP_WITH_GPU
    if (use_gpu && !params->use_regex)
    {
        krep_gpu_set_reference_simd(KREP_USE_AVX512 ? KREP_REF_AVX512 : KREP_USE_AVX2 ? KREP_REF_AVX2
                                    : KREP_USE_SSE42 ? KREP_REF_SSE42 : KREP_USE_NEON ? KREP_REF_NEON : KREP_REF_SCALAR);
        krep_gpu_set_only_matching(only_matching);
        krep_gpu_set_result_order(1); /* records come back in (start, end) order: step 6 skips the qsort */
        krep_gpu_set_force_no_simd(force_no_simd);
        krep_gpu_set_algo_override(!algo_override || !strcmp(algo_override, "auto") ? KREP_ALGO_AUTO
                                   : !strcmp(algo_override, "bm") ? KREP_ALGO_BM
                                   : !strcmp(algo_override, "kmp") ? KREP_ALGO_KMP : KREP_ALGO_AUTO);
        search_func_t gpu_fn = krep_gpu_select_search_algorithm(params); /* NULL: not accelerated -> CPU function below */
        if (gpu_fn)
            return gpu_fn;
    }
#endif
'''
    src = insert_after(src, r'^search_func_t select_search_algorithm\(const search_params_t \*params\)\s*\{', gpu_select)
    # 4. one chunk when the GPU is on: the backend shards internally with start-offset ownership (krep.c:2729-2744)
    one_chunk = '\n#ifdef KREP_WITH_GPU\n    if (use_gpu && !params->use_regex)\n        actual_thread_count = 1;\n#endif\n'
    src = insert_after(src, r'^\s*if \(actual_thread_count <= 0\)\s*\n\s*actual_thread_count = 1;', one_chunk)
    # 5. the CLI switch: environment variable, read at the top of main() (krep.c:3451)
    src = insert_after(src, r'^int main\(int argc, char \*argv\[\]\)\s*\{', '\n#ifdef KREP_WITH_GPU\n    use_gpu = getenv("KREP_GPU") != NULL;\n#endif\n')
    # 6. the records of the GPU operators arrive in compare_match_positions order (sorted in HBM): no host qsort
    #    (krep.c:3020-3023).  Only when a GPU operator really was selected: for the input classes the backend leaves to the
    #    CPU (krep_gpu_can_accelerate() == 0) the selector above falls through to the reference's own function.
    pat = r'if \(global_matches->count > 1\)(\s*\{\s*qsort\(global_matches->positions)'
    if not re.search(pat, src):
        raise SystemExit("anchor not found: qsort of the global match list")
    src = re.sub(pat, r'if (global_matches->count > 1\n#ifdef KREP_WITH_GPU\n            && preselected_algo != krep_gpu_literal_search && preselected_algo != krep_gpu_aho_corasick_search\n#endif\n            )\1', src, count=1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        patched = os.path.join(td, "krep_gpu_patched.c")
        open(patched, "w").write(src)
        cmd = ["gcc", "-O2", "-std=c11", "-D_GNU_SOURCE", "-D_DEFAULT_SOURCE", "-pthread", "-w", "-mavx2", "-msse4.2",
               "-DKREP_WITH_GPU", f"-I{REF}", f"-I{os.path.join(ROOT, 'include')}", patched,
               os.path.join(REF, "aho_corasick.c"), f"-L{os.path.join(ROOT, 'krep_amd', 'lib')}", "-lkrep_gpu",
               "-Wl,-rpath,$ORIGIN/../../krep_amd/lib", "-Wl,-rpath,/opt/rocm/lib", "-o", OUT]
        subprocess.run(cmd, check=True)
    print("built", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
