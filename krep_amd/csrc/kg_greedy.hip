// kg_greedy.hip — the reference's SEQUENTIAL match-set families, resolved on the ordered list the scan produced.
//
// (1) Greedy leftmost non-overlapping selection:
// simd_sse42_search (krep.c:4839-4848: advance = index + pattern_len after a hit) and kmp_search
// (krep.c:1741: i = match_start + pattern_len) report the GREEDY subset of the occurrences: scan left
// to right, after taking a hit at s the next candidate must start at >= s + m.  For a pattern without
// a border (no proper prefix that is also a suffix) two occurrences can never overlap and the subset
// is everything — the scan kernel's output is used as is.  For bordered patterns ("aa", "abab") this
// pass runs on the occurrence list the scan produced (O(matches), the haystack is not re-read):
//   g_walk    : occurrences split into clusters at gaps >= m; the first thread of a cluster walks it
//   g_ww      : -w applied AFTER the selection, as the reference does (a rejected hit still consumes)
//   g_count/g_scan/g_scatter : order-preserving compaction into the caller's match_position_t buffer
//   g_lines   : distinct lines among the survivors ('\n' between consecutive survivors)
// boyer_moore_search under -o (krep.c:1371: i += pattern_len after a hit) is the same selection, with -w applied BEFORE it
// (a rejected hit shifts by the bad-character table and consumes nothing, krep.c:1323-1329): the scan filters, g_walk selects.
//
// (2) memchr_short_search under -o (krep.c:4396-4500): the walk is over the FIRST-BYTE candidates (memchr / the folded
// scalar loop), and `advance = (candidate - current) + (only_matching ? pattern_len : 1)` (:4495) also runs after a
// candidate whose remaining bytes did NOT match — the next pattern_len - 1 positions are never examined.  A full match that
// fails -w resumes one byte further (:4441-4446).  The list is the candidate list of a one-byte scan; the cluster head walks
// it, classifies every visited candidate against the text (bytes 1..m-1, -w) and keeps the accepted matches.  A candidate
// at least m bytes behind its predecessor is always visited (the resume point never passes candidate + m), so clusters
// split at gaps >= m exactly as in (1).
#include <hip/hip_runtime.h>
#include <algorithm>
#include "kg_common.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

constexpr int kGB = 256;            // threads per block
constexpr int kGPer = 4;            // elements per thread in the compaction kernels
constexpr int kGBlockElems = kGB * kGPer;

__device__ __forceinline__ bool g_wordc(u32 c) { return (c - '0' < 10u) || ((c | 0x20u) - 'a' < 26u) || c == '_'; }

__device__ __forceinline__ uint8_t g_fold(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// mode kWalkGreedy: element = occurrence, a visited element is kept and consumes m.
// mode kWalkShortO: element = first-byte candidate (record start), classified on the fly.
__global__ __launch_bounds__(kGB) void g_walk(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text,
                                              u64 text_len, WalkSpec ws, uint8_t *__restrict__ keep)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i >= n)
        return;
    const u32 m = ws.m;
    const u64 s = occ[2 * i];
    if (i != 0 && s - occ[2 * (i - 1)] < m)
        return; // not the head of a cluster
    u64 cur = s, prev = s; // cur: first start the walk will look at next
    for (u64 j = i; j < n; ++j)
    {
        const u64 sj = occ[2 * j];
        if (j != i && sj - prev >= m)
            break; // next cluster: its own head thread takes over
        prev = sj;
        if (sj < cur)
        {
            keep[j] = 0;
            continue;
        }
        if (ws.mode == kWalkGreedy)
        {
            keep[j] = 1;
            cur = sj + m;
            continue;
        }
        const u64 p = sj - base; // offset inside the device buffer; p + m <= text_len by construction of the list
        uint8_t c1 = text[p + 1], c2 = m > 2 ? text[p + 2] : 0;
        if (ws.ci)
        {
            c1 = g_fold(c1);
            c2 = g_fold(c2);
        }
        const bool full = c1 == ws.b1 && (m < 3 || c2 == ws.b2);
        if (!full)
        {
            keep[j] = 0;
            cur = sj + m; // krep.c:4495
            continue;
        }
        bool ok = true;
        if (ws.ww)
        {
            if (p > 0 && g_wordc(text[p - 1]))
                ok = false;
            else if (p + m < text_len && g_wordc(text[p + m]))
                ok = false;
        }
        keep[j] = ok ? 1 : 0;
        cur = ok ? sj + m : sj + 1; // krep.c:4441-4446: a -w rejected match resumes one byte further
    }
}

__global__ __launch_bounds__(kGB) void g_ww(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text,
                                            u64 text_len, u32 m, uint8_t *__restrict__ keep)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i >= n || !keep[i])
        return;
    const u64 p = occ[2 * i] - base; // offset inside the device buffer
    bool ok = true;
    if (p > 0 && g_wordc(text[p - 1]))
        ok = false;
    else if (p + m < text_len && g_wordc(text[p + m]))
        ok = false;
    if (!ok)
        keep[i] = 0;
}

__global__ __launch_bounds__(kGB) void g_count(const uint8_t *__restrict__ keep, u64 n, u64 *__restrict__ blk)
{
    __shared__ u32 s[4];
    const u64 i0 = ((u64)blockIdx.x * kGB + threadIdx.x) * kGPer;
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < kGPer; ++k)
        c += (i0 + k < n && keep[i0 + k]) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0)
        s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        blk[blockIdx.x] = (u64)s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of the block counts by one wave; total -> ctr->total
__global__ __launch_bounds__(64) void g_scan(u64 nb, u64 *__restrict__ blk, Counters *ctr)
{
    const u32 lane = threadIdx.x;
    u64 run = 0;
    for (u64 b0 = 0; b0 < nb; b0 += 64)
    {
        const u64 b = b0 + lane;
        const u64 v = b < nb ? blk[b] : 0ull;
        u64 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const u64 t = __shfl_up(incl, o);
            if (lane >= (u32)o)
                incl += t;
        }
        if (b < nb)
            blk[b] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0)
        ctr->total = run;
}

// set_len != 0: records are rewritten as {start, start + set_len} (the candidate list of kWalkShortO holds one-byte records)
__global__ __launch_bounds__(kGB) void g_scatter(const u64 *__restrict__ occ, const uint8_t *__restrict__ keep, u64 n,
                                                 const u64 *__restrict__ blk, u64 *__restrict__ out, u64 cap, u32 set_len)
{
    __shared__ u32 s[4];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 i0 = ((u64)blockIdx.x * kGB + threadIdx.x) * kGPer;
    u32 c = 0;
    bool k[kGPer];
#pragma unroll
    for (int q = 0; q < kGPer; ++q)
    {
        k[q] = i0 + q < n && keep[i0 + q];
        c += k[q] ? 1u : 0u;
    }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const u32 t = __shfl_up(incl, o);
        if (lane >= (u32)o)
            incl += t;
    }
    if (lane == 63)
        s[wave] = incl;
    __syncthreads();
    u64 idx = blk[blockIdx.x] + (incl - c);
    for (u32 w = 0; w < wave; ++w)
        idx += s[w];
#pragma unroll
    for (int q = 0; q < kGPer; ++q)
        if (k[q])
        {
            if (idx < cap)
            {
                const u64 st = occ[2 * (i0 + q)];
                out[2 * idx] = st;
                out[2 * idx + 1] = set_len ? st + set_len : occ[2 * (i0 + q) + 1];
            }
            ++idx;
        }
}

// survivors are in `lst` (records): count those whose line differs from the previous survivor's
__global__ __launch_bounds__(kGB) void g_lines(const u64 *__restrict__ lst, u64 n, u64 base, const uint8_t *__restrict__ text,
                                               Counters *ctr)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    u32 first = 0;
    if (i < n)
    {
        if (i == 0)
            first = 1;
        else
        {
            const u64 a = lst[2 * (i - 1)] - base, b = lst[2 * i] - base;
            for (u64 p = b; p > a;) // a '\n' in [a, b) separates the two line starts
            {
                --p;
                if (text[p] == '\n')
                {
                    first = 1;
                    break;
                }
            }
        }
    }
    u64 bal = __ballot(first);
    if ((threadIdx.x & 63) == 0 && bal)
        atomicAdd(&ctr->lines, (u64)__popcll(bal));
}

#define GCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// occ: n_occ records already in s.d_occ.  Results: *total kept matches; records into d_pos (<= want);
// with ws.lines the distinct-line count (the survivors are compacted into scratch for that).
int post_walk(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const WalkSpec &ws,
              uint64_t n_occ, uint64_t *d_pos, uint64_t want, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
              uint64_t *total, uint64_t *nlines)
{
    *total = 0;
    *nlines = 0;
    if (n_occ == 0)
        return 0;
    const u64 nb = (n_occ + kGBlockElems - 1) / kGBlockElems;
    if (n_occ > s.keep_cap)
    {
        if (s.d_keep) (void)hipFree(s.d_keep);
        if (s.d_gblk) (void)hipFree(s.d_gblk);
        if (s.d_surv) (void)hipFree(s.d_surv);
        s.d_keep = nullptr; s.d_gblk = nullptr; s.d_surv = nullptr; s.keep_cap = 0;
        GCHK(hipMalloc(&s.d_keep, n_occ));
        GCHK(hipMalloc(&s.d_gblk, nb * sizeof(u64)));
        GCHK(hipMalloc(&s.d_surv, n_occ * 2 * sizeof(u64)));
        s.keep_cap = n_occ;
    }
    const u64 *occ = (const u64 *)s.d_occ;
    const u32 g1 = (u32)((n_occ + kGB - 1) / kGB);
    const u32 set_len = ws.mode == kWalkShortO ? ws.m : 0u;
    GCHK(hipMemsetAsync(s.d_keep, 0, n_occ, st));
    hipLaunchKernelGGL(g_walk, dim3(g1), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws, s.d_keep);
    if (ws.ww && ws.mode == kWalkGreedy) // -w AFTER the selection: a rejected hit still consumed (krep.c:4767, :1684)
        hipLaunchKernelGGL(g_ww, dim3(g1), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws.m, s.d_keep);
    hipLaunchKernelGGL(g_count, dim3((u32)nb), dim3(kGB), 0, st, (const uint8_t *)s.d_keep, (u64)n_occ, (u64 *)s.d_gblk);
    hipLaunchKernelGGL(g_scan, dim3(1), dim3(64), 0, st, (u64)nb, (u64 *)s.d_gblk, d_ctr);
    if (ws.lines)
    {
        hipLaunchKernelGGL(g_scatter, dim3((u32)nb), dim3(kGB), 0, st, occ, (const uint8_t *)s.d_keep, (u64)n_occ,
                           (const u64 *)s.d_gblk, (u64 *)s.d_surv, (u64)n_occ, set_len);
        GCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        const u64 nsurv = h_ctr->total;
        if (nsurv)
        {
            GCHK(hipMemsetAsync(&d_ctr->lines, 0, sizeof(u64), st));
            hipLaunchKernelGGL(g_lines, dim3((u32)((nsurv + kGB - 1) / kGB)), dim3(kGB), 0, st, (const u64 *)s.d_surv, (u64)nsurv,
                               (u64)global_base, d_text, d_ctr);
        }
    }
    else if (d_pos && want)
        hipLaunchKernelGGL(g_scatter, dim3((u32)nb), dim3(kGB), 0, st, occ, (const uint8_t *)s.d_keep, (u64)n_occ,
                           (const u64 *)s.d_gblk, (u64 *)d_pos, (u64)want, set_len);
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    *total = h_ctr->total;
    *nlines = ws.lines ? h_ctr->lines : 0;
    return 0;
}

} // namespace kg
