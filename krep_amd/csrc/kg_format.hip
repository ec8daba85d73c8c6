// kg_format.hip — device-side post-processing of a match list for the reference's output formatter
// (SURVEY.md §8f-4).  After the scan the reference's host code
//   * qsort()s all records by (start, end) on one thread before printing (krep.c:3018-3023, comparator :420-434) —
//     the multi-pattern scan emits them in (end, longest-first) order (aho_corasick.c:353-431);
//   * derives the line number of every printed match by re-counting newlines (krep.c:589-668).
// Both are data-parallel; these entry points do them in HBM so that the host tail is a plain copy.
//
// Ordering: the list is already ascending in `end`, so ONE stable sort keyed on `start` yields the lexicographic
// (start, end) order.  The sort itself is the vendor's device radix sort (hipCUB, header-only in ROCm; it counts items in an
// int, hence the documented 2^31-1 record limit of krep_gpu_order_by_start) — a bandwidth-bound primitive off the hot path;
// the split/merge kernels, the prefix sum and the line-number kernels are ours.  Scratch is one grow-only buffer per device.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <mutex>

#include "../../include/krep_gpu.h"
#include "kg_internal.h"

namespace kg {

using u32 = uint32_t;
using u64 = unsigned long long;

#define FCHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            rc = fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            goto done;                                                                         \
        }                                                                                      \
    } while (0)

__global__ void fmt_split(const u64 *__restrict__ rec, u64 n, u64 *__restrict__ starts, u64 *__restrict__ ends)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const uint4 r = *reinterpret_cast<const uint4 *>(rec + 2 * i);
        starts[i] = ((u64)r.y << 32) | r.x;
        ends[i] = ((u64)r.w << 32) | r.z;
    }
}

__global__ void fmt_merge(const u64 *__restrict__ starts, const u64 *__restrict__ ends, u64 n, u64 *__restrict__ rec)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const u64 s = starts[i], e = ends[i];
        *reinterpret_cast<uint4 *>(rec + 2 * i) = make_uint4((u32)s, (u32)(s >> 32), (u32)e, (u32)(e >> 32));
    }
}

constexpr u32 kLineBlock = 4096; // newline counts are kept per 4 KiB of text

// one wave per 4 KiB block: 4 x (64 lanes x 16 B), SWAR byte-equality + popcount
__global__ __launch_bounds__(256) void fmt_count_newlines(const uint8_t *__restrict__ text, u64 text_len, u64 nblocks,
                                                          u64 *__restrict__ counts)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 wid = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (u64)gridDim.x * (blockDim.x >> 6);
    for (u64 b = wid; b < nblocks; b += nw)
    {
        u32 c = 0;
        const u64 base = b * kLineBlock;
        for (u32 it = 0; it < kLineBlock / 1024; ++it)
        {
            const u64 off = base + (u64)it * 1024 + (u64)lane * 16;
            if (off + 16 <= text_len && ((reinterpret_cast<size_t>(text) + off) & 15u) == 0)
            {
                const uint4 v = *reinterpret_cast<const uint4 *>(text + off);
                const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                    const u32 y = w[q] ^ 0x0a0a0a0au;
                    c += (u32)__popc(~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu));
                }
            }
            else
                for (u32 q = 0; q < 16; ++q)
                    if (off + q < text_len && text[off + q] == '\n')
                        ++c;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            c += __shfl_xor(c, o);
        if (lane == 0)
            counts[b] = c;
    }
}

// line number (1-based) of text[start]: newlines in the blocks before + newlines of the own block before `start`
__global__ void fmt_line_numbers(const uint8_t *__restrict__ text, u64 text_len, u64 base, const u64 *__restrict__ rec, u64 n,
                                 const u64 *__restrict__ block_prefix, u64 *__restrict__ lines)
{
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const u64 g = rec[2 * i];
    if (g < base || g - base > text_len)
    {
        lines[i] = 0; // a record outside the buffer (wrong global_base): flagged, never read out of bounds
        return;
    }
    const u64 s = g - base, b = s / kLineBlock;
    u64 ln = 1 + block_prefix[b];
    auto nl4 = [](u32 w) -> u32 { // newlines among the 4 bytes of w (exact SWAR byte equality)
        const u32 y = w ^ 0x0a0a0a0au;
        return (u32)__popc(~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y | 0x7f7f7f7fu));
    };
    u64 p = b * kLineBlock;
    if ((reinterpret_cast<size_t>(text) & 15u) == 0) // 16 bytes at a time while they lie wholly before `s`
        for (; p + 16 <= s; p += 16)
        {
            const uint4 v = *reinterpret_cast<const uint4 *>(text + p);
            ln += nl4(v.x) + nl4(v.y) + nl4(v.z) + nl4(v.w);
        }
    for (; p < s; ++p)
        ln += text[p] == '\n';
    lines[i] = ln;
}

// ---- exclusive prefix sum over the per-block newline counts (hand-written: three tiny kernels, 64-bit sizes) -----------
constexpr u32 kScanBlock = 256, kScanPer = 8, kScanElems = kScanBlock * kScanPer; // 2048 counts per workgroup
__global__ __launch_bounds__(kScanBlock) void fmt_scan_local(const u64 *__restrict__ in, u64 n, u64 *__restrict__ out,
                                                            u64 *__restrict__ block_sum)
{
    __shared__ u64 s_w[kScanBlock / 64];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u64 i0 = (u64)blockIdx.x * kScanElems + (u64)threadIdx.x * kScanPer;
    u64 v[kScanPer], t = 0;
#pragma unroll
    for (u32 k = 0; k < kScanPer; ++k)
    {
        v[k] = i0 + k < n ? in[i0 + k] : 0ull;
        t += v[k];
    }
    u64 incl = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
    {
        const u64 up = __shfl_up(incl, o);
        if (lane >= (u32)o)
            incl += up;
    }
    if (lane == 63)
        s_w[wave] = incl;
    __syncthreads();
    u64 e = incl - t;
    for (u32 w = 0; w < wave; ++w)
        e += s_w[w];
#pragma unroll
    for (u32 k = 0; k < kScanPer; ++k)
    {
        if (i0 + k < n)
            out[i0 + k] = e;
        e += v[k];
    }
    if (threadIdx.x == kScanBlock - 1)
        block_sum[blockIdx.x] = e;
}
__global__ __launch_bounds__(64) void fmt_scan_sums(u64 *__restrict__ block_sum, u64 nb)
{
    const u32 lane = threadIdx.x;
    u64 run = 0;
    for (u64 b0 = 0; b0 < nb; b0 += 64)
    {
        const u64 v = b0 + lane < nb ? block_sum[b0 + lane] : 0ull;
        u64 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const u64 up = __shfl_up(incl, o);
            if (lane >= (u32)o)
                incl += up;
        }
        if (b0 + lane < nb)
            block_sum[b0 + lane] = run + incl - v;
        run += __shfl(incl, 63);
    }
}
__global__ __launch_bounds__(kScanBlock) void fmt_scan_add(u64 *__restrict__ out, u64 n, const u64 *__restrict__ block_off)
{
    const u64 add = block_off[blockIdx.x];
    const u64 i0 = (u64)blockIdx.x * kScanElems + (u64)threadIdx.x * kScanPer;
#pragma unroll
    for (u32 k = 0; k < kScanPer; ++k)
        if (i0 + k < n)
            out[i0 + k] += add;
}

// ---- scratch: one grow-only buffer per device, reused across calls (ADVICE / VERDICT r02: no hipMalloc per call) ------
namespace {
struct FmtScratch
{
    void *p = nullptr;
    size_t cap = 0;
};
std::mutex g_fmt_mu; // held for the whole call: the entry points synchronise before they return
FmtScratch g_fmt[64];
int fmt_reserve(size_t bytes, void **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return fail("format scratch: bad device");
    FmtScratch &s = g_fmt[dev];
    if (s.cap < bytes)
    {
        if (s.p) (void)hipFree(s.p);
        s.p = nullptr;
        s.cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&s.p, want) != hipSuccess)
        {
            (void)hipGetLastError();
            return fail("format scratch: allocation of %zu bytes failed", want);
        }
        s.cap = want;
    }
    *out = s.p;
    return 0;
}
} // namespace
void format_release()
{
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < 64; ++d)
        if (g_fmt[d].p)
        {
            (void)hipSetDevice(d);
            (void)hipFree(g_fmt[d].p);
            g_fmt[d] = FmtScratch{};
        }
    if (prev >= 0)
        (void)hipSetDevice(prev);
}

} // namespace kg

using namespace kg;

// max_offset: an upper bound of every start offset in the list — the length of the WHOLE text when the records carry a
// global_base (the radix key is sized from it)
// stable sort of a record list keyed on `start` (by_end = false) or on `end` (true); max_offset bounds the key
namespace kg {
int order_records(match_position_t *d_positions, uint64_t n, size_t max_offset, hipStream_t st, bool by_end);
} // namespace kg
extern "C" int krep_gpu_order_by_start(match_position_t *d_positions, uint64_t n, size_t max_offset, void *stream)
{
    return kg::order_records(d_positions, n, max_offset, (hipStream_t)stream, false);
}
namespace kg {
int order_records(match_position_t *d_positions, uint64_t n, size_t max_offset, hipStream_t st, bool by_end)
{
    const size_t text_len = max_offset;
    if (n < 2)
        return 0;
    if (n > 0x7fffffffull) // the vendor's device radix sort counts its items in an int: 2^31-1 records = 34 GB of match_position_t
        return fail("krep_gpu_order_by_start: %llu records exceed the device sort's 2^31-1 item limit", (unsigned long long)n);
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    int rc = 0;
    int bits = 1;
    while (bits < 64 && ((u64)text_len >> bits))
        ++bits;
    {
        size_t tmp_bytes = 0;
        u64 *nil = nullptr;
        FCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, nil, nil, nil, nil, (int)n, 0, bits, st));
        const size_t keys = 4 * n * sizeof(u64);
        void *base = nullptr;
        if (fmt_reserve(keys + tmp_bytes + 256, &base))
            return 2;
        u64 *k0 = (u64 *)base, *k1 = k0 + n, *v0 = k0 + 2 * n, *v1 = k0 + 3 * n;
        void *tmp = (uint8_t *)base + ((keys + 255) & ~(size_t)255);
        const u32 grid = (u32)((n + 255) / 256);
        // (keyed on `end`: the two columns change roles on the way in and on the way out)
        hipLaunchKernelGGL(fmt_split, dim3(grid), dim3(256), 0, st, (const u64 *)d_positions, (u64)n, by_end ? v0 : k0, by_end ? k0 : v0);
        FCHK(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, k0, k1, v0, v1, (int)n, 0, bits, st));
        hipLaunchKernelGGL(fmt_merge, dim3(grid), dim3(256), 0, st, (const u64 *)(by_end ? v1 : k1), (const u64 *)(by_end ? k1 : v1), (u64)n, (u64 *)d_positions);
        FCHK(hipGetLastError());
        FCHK(hipStreamSynchronize(st));
    }
done:
    return rc;
}
} // namespace kg

extern "C" int krep_gpu_line_numbers(const void *d_text, size_t text_len, const match_position_t *d_positions, uint64_t n,
                                     uint64_t *d_lines, void *stream)
{
    return krep_gpu_line_numbers_ex(d_text, text_len, 0, d_positions, n, d_lines, stream);
}
extern "C" int krep_gpu_line_numbers_ex(const void *d_text, size_t text_len, size_t global_base, const match_position_t *d_positions,
                                        uint64_t n, uint64_t *d_lines, void *stream)
{
    if (!n)
        return 0;
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    int rc = 0;
    const u64 nblocks = ((u64)text_len + kLineBlock - 1) / kLineBlock + 1; // + 1: a match may start at text_len
    const u64 nsb = (nblocks + kScanElems - 1) / kScanElems;
    if (nsb > 0x7fffffffull)
        return fail("krep_gpu_line_numbers: text too long"); // > 2^54 bytes
    {
        void *base = nullptr;
        if (fmt_reserve((2 * nblocks + nsb) * sizeof(u64), &base))
            return 2;
        u64 *cnt = (u64 *)base, *pre = cnt + nblocks, *sums = pre + nblocks;
        FCHK(hipMemsetAsync(cnt, 0, nblocks * sizeof(u64), st));
        const u32 grid = (u32)std::min<u64>((nblocks + 3) / 4, 256u * 32u);
        hipLaunchKernelGGL(fmt_count_newlines, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_text, (u64)text_len,
                           (u64)(nblocks - 1), cnt);
        hipLaunchKernelGGL(fmt_scan_local, dim3((u32)nsb), dim3(kScanBlock), 0, st, (const u64 *)cnt, (u64)nblocks, pre, sums);
        hipLaunchKernelGGL(fmt_scan_sums, dim3(1), dim3(64), 0, st, sums, (u64)nsb);
        hipLaunchKernelGGL(fmt_scan_add, dim3((u32)nsb), dim3(kScanBlock), 0, st, pre, (u64)nblocks, (const u64 *)sums);
        hipLaunchKernelGGL(fmt_line_numbers, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_text, (u64)text_len,
                           (u64)global_base, (const u64 *)d_positions, (u64)n, (const u64 *)pre, (u64 *)d_lines);
        FCHK(hipGetLastError());
        FCHK(hipStreamSynchronize(st));
    }
done:
    return rc;
}
