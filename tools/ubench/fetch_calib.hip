// fetch_calib.hip — what does one unit of rocprofv3's FETCH_SIZE stand for, per access pattern, on this ROCm / gfx950?
// (development probe; not product code).  MI355X_MICROARCH.md says FETCH_SIZE under-reports 16 B/lane streams by 2x; the
// multi-pattern verifier does NOT stream: it gathers one 16-byte text window and one 32-byte table entry per candidate.
// Doubling FETCH_SIZE for those reads would over-count them.  Three kernels with exactly known traffic:
//   calib_stream16 : every lane 16 B, wave-coalesced, each byte of N read once            -> N bytes
//   calib_gather16 : every lane 16 B from its OWN 128-byte line (stride 128 B)            -> lanes x 128 B (or 64 B) of lines
//   calib_gather4  : every lane  4 B from its own 128-byte line                           -> the same lines
// run:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o pmc -- /tmp/fc   and read FETCH_SIZE per kernel:
// KiB reported / (accesses) tells the unit per access; tools/collect_profiles.py turns that into the correction factors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void calib_stream16(const uint8_t *__restrict__ p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i + 16 <= n; i += (size_t)gridDim.x * 256 * 16)
    {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + i));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather16(const uint8_t *__restrict__ p, size_t n_lines, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t l = (size_t)blockIdx.x * 256 + threadIdx.x; l < n_lines; l += (size_t)gridDim.x * 256)
    {
        const size_t line = (l * 2654435761ull) % n_lines; // scattered: no two lanes of a wave share a line
        const u32x4 v = *reinterpret_cast<const u32x4 *>(p + line * 128 + 48);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_gather4(const uint8_t *__restrict__ p, size_t n_lines, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t l = (size_t)blockIdx.x * 256 + threadIdx.x; l < n_lines; l += (size_t)gridDim.x * 256)
    {
        const size_t line = (l * 2654435761ull) % n_lines;
        acc ^= *reinterpret_cast<const uint32_t *>(p + line * 128 + 52);
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

int main()
{
    const size_t n = (size_t)8 << 30, n_lines = n / 128; // 8 GiB: far beyond every cache (n_lines is a power of two; the
    uint8_t *p;                                           // multiplier is odd, so the line map is a bijection)
    uint32_t *out;
    CHK(hipMalloc(&p, n));
    CHK(hipMemset(p, 1, n));
    CHK(hipMalloc(&out, 4));
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(calib_stream16, dim3(4096), dim3(256), 0, nullptr, p, n, out);
    hipLaunchKernelGGL(calib_gather16, dim3(4096), dim3(256), 0, nullptr, p, n_lines, out);
    hipLaunchKernelGGL(calib_gather4, dim3(4096), dim3(256), 0, nullptr, p, n_lines, out);
    CHK(hipDeviceSynchronize());
    printf("stream16: %zu bytes read once; gather16 / gather4: %zu accesses, one per 128-byte line\n", n, n_lines);
    return 0;
}
