// kg_tail.hip — device helpers for the end-of-text replay of the block-structured -c paths (kg_replay.h):
//   tail_last_hit_unit : the last scan unit (below a limit) that reported an accepted occurrence (per-unit info words)
//   tail_next_newline / tail_prev_newline : first '\n' at or after / last '\n' before a position (early-exit sweeps)
//   tail_replay        : replay_lines() as a one-thread kernel over the last <= 320 bytes of the text in HBM
// None of this is on the bandwidth path: a -c scan through simd_avx512_search / neon_search (or simd_avx2_search
// with -w) pays four tiny launches after the canonical scan.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "kg_common.h"
#include "kg_internal.h"
#include "kg_replay.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

#define TCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// out[0] = 1 + index of the last unit u < limit whose info word reports a hit (0 = none)
__global__ __launch_bounds__(256) void tail_last_hit_unit(const u64 *__restrict__ info, u64 limit, u64 *out)
{
    u64 best = 0;
    for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < limit; u += (u64)gridDim.x * blockDim.x)
        if (info[u] & kUiCountMask)
            best = u + 1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        const u64 v = __shfl_xor(best, o);
        best = v > best ? v : best;
    }
    if ((threadIdx.x & 63) == 0 && best)
        atomicMax(out, best);
}

constexpr u32 kNlChunk = 64 * 1024; // bytes one workgroup inspects per step

// The sweeps read 16 bytes per lane (a wave: 1 KiB per step, aligned) and leave a chunk as soon as ANY lane of the wave has
// found its newline: the first version tested one byte per thread and step and left a chunk thread by thread — 1024 workgroups
// each read their whole 64 KiB before the first result was visible (0.24 ms per sweep at 32 GiB for a newline 80 bytes away).
__device__ __forceinline__ u32 nl_mask16(const uint4 v) // bit k: byte k of the 16 is '\n'
{
    auto m4 = [](u32 x) -> u32 {
        const u32 y = x ^ 0x0a0a0a0au;
        const u32 z = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu); // 0x80 in every byte equal to '\n'
        return (((z >> 7) * 0x00204081u) >> 21) & 0xfu;
    };
    return m4(v.x) | (m4(v.y) << 4) | (m4(v.z) << 8) | (m4(v.w) << 12);
}

// out[0] = min index >= from holding '\n' (initialised to n by the host).  Chunks are visited in ascending order by block
// index; a block stops as soon as an earlier chunk has produced a result.  `text` is 16-byte aligned (a device allocation).
__global__ __launch_bounds__(256) void tail_next_newline(const uint8_t *__restrict__ text, u64 from, u64 n, u64 *out)
{
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 base = from & ~15ull;
    for (u64 c = blockIdx.x;; c += gridDim.x)
    {
        const u64 lo = base + c * kNlChunk;
        if (lo >= n || __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lo)
            return;
        const u64 hi = lo + kNlChunk < n ? lo + kNlChunk : n;
        for (u64 p = lo + (u64)wave * 1024u; p < hi; p += 4096u) // (the four waves interleave 1-KiB steps)
        {
            const u64 q = p + (u64)lane * 16u;
            u32 m = 0;
            if (q + 16 <= n)
                m = nl_mask16(*reinterpret_cast<const uint4 *>(text + q));
            else
                for (u32 k = 0; k < 16u && q + k < n; ++k)
                    m |= text[q + k] == '\n' ? 1u << k : 0u;
            if (q < from) // (the first vector may start in front of `from`)
                m &= from - q < 16 ? ~0u << (u32)(from - q) : 0u;
            const u64 any = __ballot(m != 0u);
            if (any)
            {
                const u32 first = (u32)__builtin_ctzll(any);
                const u32 fm = __shfl(m, first);
                if (lane == 0)
                    atomicMin(out, p + (u64)first * 16u + (u64)__builtin_ctz(fm));
                return; // ascending: nothing later in this chunk (or in this block's later chunks) can be smaller
            }
        }
    }
}

// out[0] = 1 + max index < before holding '\n' (0 = none; initialised to 0 by the host); chunks descend from `before`
__global__ __launch_bounds__(256) void tail_prev_newline(const uint8_t *__restrict__ text, u64 before, u64 cmax, u64 *out)
{
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 top = (before + 15ull) & ~15ull; // the vectors are aligned; bytes at or behind `before` are masked off
    for (u64 c = blockIdx.x;; c += gridDim.x)
    {
        if (c >= cmax || c * kNlChunk >= top) // (cmax: the one-workgroup probe looks at the last chunk only)
            return;
        const u64 hi = top - c * kNlChunk; // exclusive, 16-byte aligned
        if (__hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > hi)
            return;
        const u64 lo = hi > kNlChunk ? hi - kNlChunk : 0;
        for (u64 d = (u64)wave * 1024u; lo + d < hi; d += 4096u) // descending 1-KiB steps
        {
            const u64 pe = hi - d;                          // the step covers [pe - 1024, pe)
            const u64 step_lo = pe > 1024u ? pe - 1024u : 0u;
            const u64 q = step_lo + (u64)lane * 16u;
            u32 m = 0;
            if (q < pe && q + 16 <= pe)
            {
                if (q + 16 <= before)
                    m = nl_mask16(*reinterpret_cast<const uint4 *>(text + q));
                else
                    for (u32 k = 0; k < 16u && q + k < before; ++k)
                        m |= text[q + k] == '\n' ? 1u << k : 0u;
            }
            const u64 any = __ballot(m != 0u);
            if (any)
            {
                const u32 last = 63u - (u32)__builtin_clzll(any);
                const u32 lm = __shfl(m, last);
                if (lane == 0)
                    atomicMax(out, step_lo + (u64)last * 16u + (u64)(31 - __builtin_clz(lm)) + 1ull);
                return;
            }
            if (step_lo == 0)
                break;
        }
    }
}

// out[0] += number of i with v[i] != v[i-1] (i == 0 counts): aho_corasick_search's -c counter is bumped whenever the
// line of the current match differs from the line of the PREVIOUS counted match (aho_corasick.c:383-396)
__global__ __launch_bounds__(256) void tail_changes(const u64 *__restrict__ v, u64 n, u64 *out)
{
    u32 c = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        c += (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(out, (u64)c);
}

// out[0] += number of records i whose match lies on another line than record i - 1's (i == 0 counts): aho_corasick_search -c
// for patterns WITHOUT a newline (aho_corasick.c:383-396) on the ordered record list.  Records ascend in `end`, and a match
// holds no newline, so two neighbours share a line iff they overlap or the gap [end of the earlier, start of the later) holds
// no '\n'.  The gap is read with an early exit: ~80 bytes on text with lines, the whole gap (each text byte at most once
// over all threads) on text without.  rec offsets are global: base = global offset of text[0].
__device__ __forceinline__ bool tail_gap_has_newline(const uint8_t *__restrict__ t, u64 lo, u64 hi)
{
    // aligned 16-byte vectors only, four in flight per step (a 64-byte line: with 80-byte lines the first step decides more than
    // half of the gaps, and a step is one memory round trip however wide it is); the bytes in front of lo / behind hi are masked
    // out of the test (reading them is safe: they share an aligned 16-byte granule with a byte of the gap)
    // 0x80 in every '\n' byte, exactly (the borrow-based (y - 0x01..) & ~y & 0x80.. test is exact only for "any zero byte": a real
    // '\n' below a 0x0b byte flags that one too, and the window mask behind this test can cut the real one away — ADVICE r04)
    auto z = [](u32 x) -> u32 { const u32 y = x ^ 0x0a0a0a0au; return ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu); };
    const size_t mis = ((size_t)(t + lo)) & 15u;
    const uint8_t *p = t + lo - mis; // aligned
    const u64 total = hi - lo + mis; // bytes from p to hi
    for (u64 off = 0; off < total; off += 64)
    {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            v[q] = off + 16u * q < total ? *reinterpret_cast<const uint4 *>(p + off + 16u * q) : make_uint4(0, 0, 0, 0);
        // byte index (relative to p + off) window that counts: [first, last)
        const u64 first = off == 0 ? (u64)mis : 0ull, last = total - off < 64 ? total - off : 64ull;
        u32 any = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w)
        {
            const uint4 &vv = v[w >> 2];
            const u32 m = z((w & 3) == 0 ? vv.x : (w & 3) == 1 ? vv.y : (w & 3) == 2 ? vv.z : vv.w);
            const u64 b0 = 4ull * w; // bytes [b0, b0 + 4) of the 64
            u32 keep = 0xffffffffu;
            if (first > b0)
                keep = first - b0 >= 4 ? 0u : keep << (8 * (u32)(first - b0));
            if (last < b0 + 4)
                keep = last <= b0 ? 0u : keep & (0xffffffffu >> (8 * (u32)(b0 + 4 - last)));
            any |= m & keep;
        }
        if (any)
            return true;
    }
    return false;
}
// n_dev != NULL: the number of records is read on the device (min(*n_dev, n)): the launch needs no host round trip behind
// the scan that produced the list
// skip_if != NULL: the launch does nothing when *skip_if is non-zero — the list is not complete yet when a unit overflowed its
// staging slot (its records are written by the emit-mode launch that follows; until then their slots hold whatever the buffer
// held: offsets far outside the text, r04's latent fault, found by tests/test_gpu_ac.py in round 5).  text_len bounds every gap
// that is read, whatever the records say.
__global__ __launch_bounds__(256) void tail_line_gaps(const uint8_t *__restrict__ text, u64 text_len, const u64 *__restrict__ rec, u64 n,
                                                      const u64 *n_dev, const u64 *skip_if, u64 base, u64 *out)
{
    if (skip_if && *skip_if)
        return;
    if (n_dev)
    {
        const u64 nd = *n_dev;
        n = nd < n ? nd : n;
    }
    u32 c = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
    {
        if (i == 0)
        {
            ++c;
            continue;
        }
        const u64 pe = rec[2 * i - 1] - base, cs = rec[2 * i] - base; // end of the previous match (exclusive), start of this one
        if (cs > pe && cs <= text_len && tail_gap_has_newline(text, pe, cs))
            ++c;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(out, (u64)c);
}

// out[0] += number of '\n' bytes in text[lo, hi): aligned 16-byte vectors, the ragged ends bytewise
__global__ __launch_bounds__(256) void tail_newlines(const uint8_t *__restrict__ text, u64 lo, u64 hi, u64 *out)
{
    const u64 alo = (lo + 15ull) & ~15ull, ahi = hi & ~15ull; // (text itself is 16-byte aligned: a device allocation or a piece of one)
    u32 c = 0;
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x, nth = (u64)gridDim.x * blockDim.x;
    if (alo < ahi)
    {
        const size_t mis = ((size_t)text) & 15u; // a buffer that is not 16-byte aligned: count everything bytewise below
        if (mis == 0)
            for (u64 q = alo + tid * 16ull; q < ahi; q += nth * 16ull)
                c += (u32)__popc(nl_mask16(*reinterpret_cast<const uint4 *>(text + q)));
        else
            for (u64 q = alo + tid; q < ahi; q += nth)
                c += text[q] == '\n' ? 1u : 0u;
    }
    const u64 head_end = alo < hi ? alo : hi, tail_beg = (ahi > alo ? ahi : head_end);
    for (u64 q = lo + tid; q < head_end; q += nth)
        c += text[q] == '\n' ? 1u : 0u;
    for (u64 q = tail_beg + tid; q < hi; q += nth)
        c += text[q] == '\n' ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c)
        atomicAdd(out, (u64)c);
}

__global__ void tail_replay(ReplayIn r, u64 *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[0] = replay_lines(r);
}

// ---- host drivers (results come back through the plan's pinned counter block) -------------------------------------
int tail_last_hit(const unsigned long long *d_unitinfo, uint64_t limit_units, unsigned long long *d_slot,
                  unsigned long long *h_slot, hipStream_t st, uint64_t *unit_plus1)
{
    TCHK(hipMemsetAsync(d_slot, 0, sizeof(u64), st));
    if (limit_units)
    {
        const u32 grid = (u32)std::min<u64>((limit_units + 255) / 256, 1024);
        hipLaunchKernelGGL(tail_last_hit_unit, dim3(grid), dim3(256), 0, st, (const u64 *)d_unitinfo, (u64)limit_units, (u64 *)d_slot);
        TCHK(hipGetLastError());
    }
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    *unit_plus1 = *h_slot;
    return 0;
}

int tail_find_next_newline(const uint8_t *d_text, uint64_t from, uint64_t n, unsigned long long *d_slot,
                           unsigned long long *h_slot, hipStream_t st, uint64_t *pos)
{
    *pos = n;
    if (from >= n)
        return 0;
    *h_slot = n; // pinned: the sentinel "no newline"
    TCHK(hipMemcpyAsync(d_slot, h_slot, sizeof(u64), hipMemcpyHostToDevice, st));
    // one workgroup on the first chunk (a line ends within a few hundred bytes in any text worth counting lines in); the wide
    // sweep only when that chunk holds none
    const u64 chunks = (n - (from & ~15ull) + kNlChunk - 1) / kNlChunk;
    hipLaunchKernelGGL(tail_next_newline, dim3(1), dim3(256), 0, st, d_text, (u64)from, (u64)std::min<u64>(n, (from & ~15ull) + kNlChunk), (u64 *)d_slot);
    TCHK(hipGetLastError());
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    if (*h_slot == n && chunks > 1)
    {
        hipLaunchKernelGGL(tail_next_newline, dim3((u32)std::min<u64>(chunks - 1, 1024)), dim3(256), 0, st, d_text,
                           (u64)((from & ~15ull) + kNlChunk), (u64)n, (u64 *)d_slot);
        TCHK(hipGetLastError());
        TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
        TCHK(hipStreamSynchronize(st));
    }
    *pos = *h_slot;
    return 0;
}

int tail_find_prev_newline(const uint8_t *d_text, uint64_t before, unsigned long long *d_slot, unsigned long long *h_slot,
                           hipStream_t st, uint64_t *pos_plus1)
{
    *pos_plus1 = 0;
    if (before == 0)
        return 0;
    TCHK(hipMemsetAsync(d_slot, 0, sizeof(u64), st));
    const u64 chunks = (((before + 15ull) & ~15ull) + kNlChunk - 1) / kNlChunk;
    hipLaunchKernelGGL(tail_prev_newline, dim3(1), dim3(256), 0, st, d_text, (u64)before, (u64)1, (u64 *)d_slot); // (the last chunk first)
    TCHK(hipGetLastError());
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    if (*h_slot == 0 && chunks > 1)
    {
        hipLaunchKernelGGL(tail_prev_newline, dim3((u32)std::min<u64>(chunks, 1024)), dim3(256), 0, st, d_text, (u64)before, ~0ull, (u64 *)d_slot);
        TCHK(hipGetLastError());
        TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
        TCHK(hipStreamSynchronize(st));
    }
    *pos_plus1 = *h_slot;
    return 0;
}

int tail_count_changes(const uint64_t *d_v, uint64_t n, unsigned long long *d_slot, unsigned long long *h_slot, hipStream_t st,
                       uint64_t *changes)
{
    TCHK(hipMemsetAsync(d_slot, 0, sizeof(u64), st));
    if (n)
    {
        const u32 grid = (u32)std::min<u64>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(tail_changes, dim3(grid), dim3(256), 0, st, (const u64 *)d_v, (u64)n, (u64 *)d_slot);
        TCHK(hipGetLastError());
    }
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    *changes = *h_slot;
    return 0;
}

int tail_count_line_gaps(const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const uint64_t *d_rec, uint64_t n, unsigned long long *d_slot,
                         unsigned long long *h_slot, hipStream_t st, uint64_t *lines)
{
    TCHK(hipMemsetAsync(d_slot, 0, sizeof(u64), st));
    if (n)
    {
        const u32 grid = (u32)std::min<u64>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(tail_line_gaps, dim3(grid), dim3(256), 0, st, d_text, (u64)text_len, (const u64 *)d_rec, (u64)n, (const u64 *)nullptr,
                           (const u64 *)nullptr, (u64)global_base, (u64 *)d_slot);
        TCHK(hipGetLastError());
    }
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    *lines = *h_slot;
    return 0;
}

// the same without a host round trip: at most `cap` records, their number read from *d_n on the device, the count ADDED to *d_out
int tail_launch_line_gaps(const uint8_t *d_text, uint64_t text_len, uint64_t global_base, const uint64_t *d_rec, const unsigned long long *d_n,
                          const unsigned long long *d_skip_if, uint64_t cap, unsigned long long *d_out, hipStream_t st)
{
    if (!cap)
        return 0;
    const u32 grid = (u32)std::min<u64>((cap + 255) / 256, 8192);
    hipLaunchKernelGGL(tail_line_gaps, dim3(grid), dim3(256), 0, st, d_text, (u64)text_len, (const u64 *)d_rec, (u64)cap, (const u64 *)d_n,
                       (const u64 *)d_skip_if, (u64)global_base, (u64 *)d_out);
    TCHK(hipGetLastError());
    return 0;
}

int tail_count_newlines(const uint8_t *d_text, uint64_t lo, uint64_t hi, unsigned long long *d_slot, unsigned long long *h_slot,
                        hipStream_t st, uint64_t *count)
{
    *count = 0;
    if (lo >= hi)
        return 0;
    TCHK(hipMemsetAsync(d_slot, 0, sizeof(u64), st));
    const u32 grid = (u32)std::min<u64>(((hi - lo) / 16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(tail_newlines, dim3(grid), dim3(256), 0, st, d_text, (u64)lo, (u64)hi, (u64 *)d_slot);
    TCHK(hipGetLastError());
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    *count = *h_slot;
    return 0;
}

int tail_run_replay(const ReplayIn &r, unsigned long long *d_slot, unsigned long long *h_slot, hipStream_t st, uint64_t *lines)
{
    hipLaunchKernelGGL(tail_replay, dim3(1), dim3(64), 0, st, r, (u64 *)d_slot);
    TCHK(hipGetLastError());
    TCHK(hipMemcpyAsync(h_slot, d_slot, sizeof(u64), hipMemcpyDeviceToHost, st));
    TCHK(hipStreamSynchronize(st));
    *lines = *h_slot;
    return 0;
}

} // namespace kg

// The same replay_lines() on host memory: lets the CPU test-suite pin the replay (and the window decomposition
// around it) against the oracle without a GPU.  Not used by any product path.
extern "C" uint64_t krep_gpu_debug_replay_host(int algo, const void *text, size_t n, const void *pat, uint32_t m, int ww,
                                               size_t cur, int open)
{
    kg::ReplayIn r{};
    r.algo = algo;
    r.m = m;
    r.ww = ww;
    r.n = n;
    r.cur = cur;
    r.open = open;
    r.text = (const uint8_t *)text;
    r.pat = (const uint8_t *)pat;
    return kg::replay_lines(r);
}
