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
// (3) -c through simd_sse42_search / kmp_search with a newline inside the pattern: g_nlwalk below.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
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

// What visiting element j means: is it kept, and where does the walk look next (`consume` bytes behind its start).
// kWalkGreedy: element = occurrence; kept, consumes m.  kWalkShortO: element = first-byte candidate, classified against the text.
struct Visit { bool keep; u64 consume; }; // (64 bits: a line jump may pass more than 4 GiB)
__device__ __forceinline__ Visit g_visit(const WalkSpec &ws, u64 sj, u64 base, const uint8_t *__restrict__ text, u64 text_len)
{
    const u32 m = ws.m;
    if (ws.mode == kWalkGreedy)
        return Visit{true, m};
    const u64 p = sj - base; // offset inside the device buffer; p + m <= text_len by construction of the list
    uint8_t c1 = text[p + 1], c2 = m > 2 ? text[p + 2] : 0;
    if (ws.ci)
    {
        c1 = g_fold(c1);
        c2 = g_fold(c2);
    }
    const bool full = c1 == ws.b1 && (m < 3 || c2 == ws.b2);
    if (!full)
        return Visit{false, m}; // krep.c:4495: pattern_len is skipped after a failed candidate as well
    bool ok = true;
    if (ws.ww)
    {
        if (p > 0 && g_wordc(text[p - 1]))
            ok = false;
        else if (p + m < text_len && g_wordc(text[p + m]))
            ok = false;
    }
    if (ok && ws.mode == kWalkShortOLines)
    {
        // -c: the line is counted and the scan continues at the start of the next line (krep.c:4460-4468: find_line_end from the
        // line's start = the first '\n' at or behind the match start — there is none between the two)
        // (eight bytes per step until a word holds the '\n': every candidate of the list pays this walk in g_next's parallel form,
        //  visited or not, and a text with very long lines made it a byte loop per candidate — ADVICE r05)
        u64 q = p;
        struct __attribute__((packed)) U64p { u64 v; };
        while (q + 8 <= text_len)
        {
            const u64 y = reinterpret_cast<const U64p *>(text + q)->v ^ 0x0a0a0a0a0a0a0a0aull;
            if ((y - 0x0101010101010101ull) & ~y & 0x8080808080808080ull)
                break;
            q += 8;
        }
        while (q < text_len && text[q] != '\n')
            ++q;
        const u64 next = q < text_len ? q + 1 : text_len;
        return Visit{true, next - p};
    }
    return Visit{ok, ok ? m : 1u}; // krep.c:4441-4446: a -w rejected match resumes one byte further
}

// flags of one list element after the walk
constexpr uint8_t kKeep = 1;    // reported (after -w)
constexpr uint8_t kVisited = 2; // the reference's scan stood on it (kept or not): it moved the resume point

constexpr u64 kWalkBound = 4096; // elements one head thread walks before the pass is handed to the parallel form below

// The first thread of a cluster walks it (clusters split at gaps >= m: an element that far behind its predecessor is always
// visited).  Typical clusters hold a handful of elements; a walk that reaches kWalkBound elements (text like "aaaa...") raises
// *too_long and the host reruns the pass with g_next / g_jump.
__global__ __launch_bounds__(kGB) void g_walk(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text,
                                              u64 text_len, WalkSpec ws, uint8_t *__restrict__ keep, u32 *too_long)
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
        if (j - i >= kWalkBound)
        {
            *too_long = 1u;
            return;
        }
        prev = sj;
        if (sj < cur)
        {
            keep[j] = 0;
            continue;
        }
        const Visit v = g_visit(ws, sj, base, text, text_len);
        keep[j] = (v.keep ? kKeep : 0) | kVisited;
        cur = sj + v.consume;
    }
}

// ---- the same walk without a serial chain (giant clusters): pointer jumping ---------------------------------------------------
// Visiting element i sends the walk to nxt[i] = the first element starting at or behind s_i + consume_i — a function of element
// i alone.  The visited set is what cluster heads reach through nxt; log2(n) rounds of pointer doubling mark it:
//   round k: every visited i marks jump[i] (its 2^k-th successor) visited; then jump[i] <- jump[jump[i]].
__global__ __launch_bounds__(kGB) void g_next(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text,
                                              u64 text_len, WalkSpec ws, u64 *__restrict__ jump, uint8_t *__restrict__ visited,
                                              uint8_t *__restrict__ accept)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i >= n)
        return;
    const u64 s = occ[2 * i];
    const Visit v = g_visit(ws, s, base, text, text_len);
    accept[i] = v.keep ? 1 : 0;
    const u64 target = s + v.consume;
    u64 lo = i + 1, hi = n; // first j > i with start >= target (starts ascend strictly)
    while (lo < hi)
    {
        const u64 mid = lo + (hi - lo) / 2;
        if (occ[2 * mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    jump[i] = lo; // n = end of the list
    // (-c -o through memchr_short_search: a line jump passes any number of candidates — only the list's first one is a head)
    visited[i] = (i == 0 || (ws.mode != kWalkShortOLines && s - occ[2 * (i - 1)] >= ws.m)) ? 1 : 0;
}
__global__ __launch_bounds__(kGB) void g_jump_mark(const u64 *__restrict__ jump, u64 n, const uint8_t *__restrict__ vin,
                                                   uint8_t *__restrict__ vout)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i < n && vin[i])
    {
        vout[i] = 1;
        const u64 j = jump[i];
        if (j < n)
            vout[j] = 1;
    }
}
__global__ __launch_bounds__(kGB) void g_jump_double(const u64 *__restrict__ jin, u64 n, u64 *__restrict__ jout)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i < n)
    {
        const u64 j = jin[i];
        jout[i] = j < n ? jin[j] : n;
    }
}
__global__ __launch_bounds__(kGB) void g_keep_from(const uint8_t *__restrict__ visited, const uint8_t *__restrict__ accept, u64 n,
                                                   uint8_t *__restrict__ keep)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i < n)
        keep[i] = (uint8_t)(((visited[i] && accept[i]) ? kKeep : 0) | (visited[i] ? kVisited : 0));
}

// Where the reference's scan stands behind the list: start + consume of the LAST visited element — the boundary record a
// following piece of the text resumes from (krep.c:4839-4848, :1741, :1371, :4495).  The elements behind the last visited one
// all start in front of that point, so there are fewer than m of them: one thread walks back.
__global__ void g_resume(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text, u64 text_len, WalkSpec ws,
                         const uint8_t *__restrict__ keep, u64 *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    u64 r = 0;
    for (u64 j = n; j-- > 0;)
        if (keep[j] & kVisited)
        {
            const u64 sj = occ[2 * j];
            r = sj + g_visit(ws, sj, base, text, text_len).consume;
            break;
        }
    *out = r;
}

__global__ __launch_bounds__(kGB) void g_ww(const u64 *__restrict__ occ, u64 n, u64 base, const uint8_t *__restrict__ text,
                                            u64 text_len, u32 m, uint8_t *__restrict__ keep)
{
    const u64 i = (u64)blockIdx.x * kGB + threadIdx.x;
    if (i >= n || !(keep[i] & kKeep))
        return;
    const u64 p = occ[2 * i] - base; // offset inside the device buffer
    bool ok = true;
    if (p > 0 && g_wordc(text[p - 1]))
        ok = false;
    else if (p + m < text_len && g_wordc(text[p + m]))
        ok = false;
    if (!ok)
        keep[i] &= (uint8_t)~kKeep; // still visited: a rejected hit consumed (krep.c:4767, :1684)
}

__global__ __launch_bounds__(kGB) void g_count(const uint8_t *__restrict__ keep, u64 n, u64 *__restrict__ blk)
{
    __shared__ u32 s[4];
    const u64 i0 = ((u64)blockIdx.x * kGB + threadIdx.x) * kGPer;
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < kGPer; ++k)
        c += (i0 + k < n && (keep[i0 + k] & kKeep)) ? 1u : 0u;
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
        k[q] = i0 + q < n && (keep[i0 + q] & kKeep);
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


// ---- (3) -c through simd_sse42_search / kmp_search with a '\n' INSIDE the pattern -------------------------------------------
// After counting a line both functions jump towards the next line start — but with a newline inside the pattern that point
// lies INSIDE the match, and simd_sse42_search adds the jump to its 16-byte WINDOW start instead of to the match
// (krep.c:4787-4793), so where the scan resumes depends on the phase of its window grid (a full window without a hit
// advances 17 - m bytes, :4858), which every earlier match has shifted.  A chain over all visited matches: one thread walks
// the ordered occurrence list.  Everything it needs per occurrence is precomputed in parallel: the line number of its start
// (kg_format.hip), the -w verdict (g_ww), and the first '\n' at or behind the start of its line, which is start + k0 (k0 = offset
// of the first newline in the pattern: the line start has no newline up to the match, and the match spells the pattern).
// Visits are ~2 per counted line (the jump skips the rest of the line), found by galloping from the current list index.
//   kNlWalkSse42: cp = window + (le + 1 - at) after a newly counted line, else at + m (at + 1 under -o);  window = cp + j * (17 - m),
//             j = (at - cp) / (17 - m) capped at the first window with fewer than 16 bytes left (krep.c:4739-4750)
//   kNlWalkKmp  : cp = le + 1 after a newly counted line (krep.c:1703-1707), else at + m; a -w rejected match: at + m (:1684-1688)
struct NlWalkSpec
{
    u32 mode, m, k0; // kNlWalkSse42 / kNlWalkKmp; pattern length; offset of the first '\n' in the pattern
    bool ww, om;     // -w; only_matching (simd_sse42_search advances by 1 instead of m)
    u64 n, maxc;     // length of the WHOLE text; max_count (SIZE_MAX = unlimited)
    // a PIECE of the text (round 5): the occurrence list is the piece's (global starts, buffer-relative line numbers);
    u64 line_off;    // global line number = lineno + line_off
    u64 cp_in;       // where the reference's scan stands when it enters the piece (global; 0 at the start of the text)
    u64 seen_in;     // global line number of the last counted match (~0: none)
};

__global__ void g_nlwalk(const u64 *__restrict__ occ, u64 n_occ, const u64 *__restrict__ lineno, const uint8_t *__restrict__ keep,
                         NlWalkSpec ws, u64 *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    const u64 n = ws.n, m = ws.m, step = 17ull - m;
    u64 cp = ws.cp_in, cnt = 0, seen = ws.seen_in, i = 0;
    while (cp <= n && n - cp >= m)
    {
        // first occurrence with start >= cp: gallop from the current index, then bisect
        if (i < n_occ && occ[2 * i] < cp)
        {
            u64 lo = i, hop = 1;
            while (lo + hop < n_occ && occ[2 * (lo + hop)] < cp)
            {
                lo += hop;
                hop <<= 1;
            }
            u64 hi = lo + hop < n_occ ? lo + hop : n_occ; // occ[lo] < cp <= occ[hi] (or hi == n_occ)
            ++lo;
            while (lo < hi)
            {
                const u64 mid = lo + (hi - lo) / 2;
                if (occ[2 * mid] < cp)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            i = lo;
        }
        if (i >= n_occ)
            break; // (a piece: the scan goes on in the next one, from cp)
        const u64 at = occ[2 * i], ln = lineno[i] + ws.line_off; // (the records carry global offsets)
        const bool ok = !ws.ww || (keep[i] & kKeep);
        if (ws.mode == kNlWalkKmp)
        {
            if (ok && ln != seen)
            {
                if (ws.maxc != ~0ull && cnt >= ws.maxc)
                    break;
                ++cnt;
                seen = ln;
                const u64 le = at + ws.k0;
                cp = le < n ? le + 1 : n;
            }
            else
                cp = at + m;
            continue;
        }
        u64 j = (at - cp) / step;
        if (n - cp < 16)
            j = 0;
        else
        {
            const u64 jt = (n - 16 - cp) / step + 1;
            j = jt < j ? jt : j;
        }
        const u64 win = cp + j * step;
        if (ok && ln != seen)
        {
            if (cnt >= ws.maxc)
                break;
            ++cnt;
            seen = ln;
            const u64 le = at + ws.k0;
            if (le < n)
            {
                cp = win + (le + 1 - at);
                continue;
            }
            if (cnt >= ws.maxc)
                break;
        }
        cp = at + (ws.om ? 1 : m);
        if (cp > n)
            cp = n;
    }
    out[0] = cnt;
    out[1] = cp;   // where the scan stands when it leaves the piece
    out[2] = seen; // ... and the line it counted last
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
// *resume (may be NULL): start + consume of the last element the walk visited (0 for an empty list).
int post_walk(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const WalkSpec &ws,
              uint64_t n_occ, uint64_t *d_pos, uint64_t want, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
              uint64_t *total, uint64_t *nlines, uint64_t *resume)
{
    *total = 0;
    *nlines = 0;
    if (resume)
        *resume = 0;
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
    const u32 set_len = ws.mode != kWalkGreedy ? ws.m : 0u;
    const bool chain_only = ws.mode == kWalkShortOLines; // no cluster structure: the parallel form from the list's first element
    GCHK(hipMemsetAsync(s.d_keep, 0, n_occ, st));
    GCHK(hipMemsetAsync(&d_ctr->pad[1], 0, sizeof(u64), st));
    if (!chain_only)
    {
        hipLaunchKernelGGL(g_walk, dim3(g1), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws, s.d_keep,
                           (u32 *)&d_ctr->pad[1]);
        GCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
    }
    if (chain_only || h_ctr->pad[1] || getenv("KREP_GPU_FORCE_POINTER_JUMPING"))
    {
        // a giant cluster: redo the pass in its parallel form (pointer jumping, ceil(log2 n) rounds)
        u64 *jmp = nullptr;
        uint8_t *flags = nullptr; // visited A | visited B | accept
        GCHK(hipMalloc(&jmp, 2 * n_occ * sizeof(u64)));
        if (hipMalloc(&flags, 3 * n_occ) != hipSuccess)
        {
            (void)hipFree(jmp);
            return fail("pointer-jumping scratch allocation failed");
        }
        u64 *ja = jmp, *jb = jmp + n_occ;
        uint8_t *va = flags, *vb = flags + n_occ, *acc = flags + 2 * n_occ;
        hipLaunchKernelGGL(g_next, dim3(g1), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws, ja, va, acc);
        for (u64 span = 1; span < n_occ; span <<= 1)
        {
            (void)hipMemcpyAsync(vb, va, n_occ, hipMemcpyDeviceToDevice, st);
            hipLaunchKernelGGL(g_jump_mark, dim3(g1), dim3(kGB), 0, st, (const u64 *)ja, (u64)n_occ, (const uint8_t *)va, vb);
            hipLaunchKernelGGL(g_jump_double, dim3(g1), dim3(kGB), 0, st, (const u64 *)ja, (u64)n_occ, jb);
            std::swap(ja, jb);
            std::swap(va, vb);
        }
        hipLaunchKernelGGL(g_keep_from, dim3(g1), dim3(kGB), 0, st, (const uint8_t *)va, (const uint8_t *)acc, (u64)n_occ, s.d_keep);
        const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);
        (void)hipFree(jmp);
        (void)hipFree(flags);
        if (e1 != hipSuccess || e2 != hipSuccess)
            return fail("pointer-jumping pass failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    GCHK(hipMemsetAsync(d_ctr, 0, sizeof(Counters), st));
    if (resume)
        hipLaunchKernelGGL(g_resume, dim3(1), dim3(64), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws,
                           (const uint8_t *)s.d_keep, (u64 *)&d_ctr->pad[1]);
    if (ws.ww && ws.mode == kWalkGreedy) // -w AFTER the selection: a rejected hit still consumed (krep.c:4767, :1684)
        hipLaunchKernelGGL(g_ww, dim3(g1), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, ws.m, s.d_keep);
    hipLaunchKernelGGL(g_count, dim3((u32)nb), dim3(kGB), 0, st, (const uint8_t *)s.d_keep, (u64)n_occ, (u64 *)s.d_gblk);
    hipLaunchKernelGGL(g_scan, dim3(1), dim3(64), 0, st, (u64)nb, (u64 *)s.d_gblk, d_ctr);
    if (ws.lines && !chain_only) // (kWalkShortOLines: every kept element IS a counted line — the count is the answer)
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
    *nlines = chain_only ? h_ctr->total : (ws.lines ? h_ctr->lines : 0);
    if (resume)
        *resume = h_ctr->pad[1];
    return 0;
}

// occ: n_occ ALL-occurrence records in s.d_occ (whole text, global_base 0); d_lineno[n_occ] = line number of every start.
int post_nlwalk(PostScratch &s, const uint8_t *d_text, uint64_t text_len, uint32_t mode, uint32_t m, uint32_t k0, bool ww, bool om,
                uint64_t maxc, uint64_t n_occ, const uint64_t *d_lineno, Counters *d_ctr, Counters *h_ctr, hipStream_t st,
                uint64_t *count, uint64_t global_base, uint64_t global_len, uint64_t line_off, uint64_t cp_in, uint64_t seen_in,
                uint64_t *cp_out, uint64_t *seen_out)
{
    *count = 0;
    *cp_out = cp_in;
    *seen_out = seen_in;
    if (n_occ == 0)
        return 0;
    if (n_occ > s.keep_cap)
    {
        const u64 nb = (n_occ + kGBlockElems - 1) / kGBlockElems;
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
    GCHK(hipMemsetAsync(s.d_keep, kKeep, n_occ, st));
    if (ww)
        hipLaunchKernelGGL(g_ww, dim3((u32)((n_occ + kGB - 1) / kGB)), dim3(kGB), 0, st, occ, (u64)n_occ, (u64)global_base, d_text, (u64)text_len, m,
                           s.d_keep);
    NlWalkSpec ws{};
    ws.mode = mode; ws.m = m; ws.k0 = k0; ws.ww = ww; ws.om = om; ws.n = global_len; ws.maxc = maxc;
    ws.line_off = line_off; ws.cp_in = cp_in; ws.seen_in = seen_in;
    hipLaunchKernelGGL(g_nlwalk, dim3(1), dim3(64), 0, st, occ, (u64)n_occ, (const u64 *)d_lineno, (const uint8_t *)s.d_keep, ws,
                       (u64 *)&d_ctr->pad[1]);
    GCHK(hipGetLastError());
    GCHK(hipMemcpyAsync(h_ctr, d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    *count = h_ctr->pad[1];
    *cp_out = h_ctr->pad[2];
    *seen_out = h_ctr->pad[3];
    return 0;
}

} // namespace kg
