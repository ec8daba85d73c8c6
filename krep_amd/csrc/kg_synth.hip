// kg_synth.hip — deterministic synthetic haystacks in HBM and on the host; see kg_synth.h.
#include <hip/hip_runtime.h>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_synth.h"
#include "kg_internal.h"

using namespace kg;

#define HIPCHK(x)                                                                             \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------ generators
__global__ void synth_kernel(uint8_t *dst, size_t len, size_t goff, int kind, uint64_t seed, const uint8_t *plant,
                             uint64_t plen, uint64_t period)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < len; i += stride)
    {
        uint32_t w[4] = {0, 0, 0, 0};
        const size_t n = len - i < 16 ? len - i : 16;
        for (size_t b = 0; b < n; ++b)
            w[b >> 2] |= (uint32_t)synth_byte(goff + i + b, kind, seed, plant, plen, period) << (8 * (b & 3));
        if (n == 16 && (((uintptr_t)(dst + i)) & 15) == 0)
            *reinterpret_cast<uint4 *>(dst + i) = make_uint4(w[0], w[1], w[2], w[3]);
        else
            for (size_t b = 0; b < n; ++b)
                dst[i + b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
    }
}

extern "C" int krep_gpu_generate(void *d_dst, size_t len, size_t global_off, int kind, uint64_t seed, const void *plant,
                                 size_t plant_len, uint64_t period, void *stream)
{
    if (!len)
        return 0;
    if ((kind == 2 || kind == 3 || kind == 4 || kind == 5) && (!plant || !plant_len))
        return kg::fail("generate: kind %d needs a plant", kind);
    if ((kind == 2) && period < plant_len)
        return kg::fail("generate: period < plant length");
    if (kind == 5 && period < 2)
        return kg::fail("generate: word text needs a line length (period) >= 2");
    hipStream_t st = (hipStream_t)stream;
    uint8_t *d_plant = nullptr;
    if (plant_len)
    {
        HIPCHK(hipMalloc(&d_plant, plant_len));
        HIPCHK(hipMemcpyAsync(d_plant, plant, plant_len, hipMemcpyHostToDevice, st));
    }
    const uint64_t plen = (kind == 4 || kind == 5) ? 0 : plant_len;
    hipLaunchKernelGGL(synth_kernel, dim3(256 * 16), dim3(256), 0, st, (uint8_t *)d_dst, len, global_off, kind, seed, d_plant,
                       plen, period ? period : 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (d_plant) (void)hipFree(d_plant);
    return 0;
}
extern "C" void krep_gpu_generate_host(void *dst, size_t len, size_t global_off, int kind, uint64_t seed, const void *plant,
                                       size_t plant_len, uint64_t period)
{
    uint8_t *d = (uint8_t *)dst;
    const uint64_t plen = (kind == 4 || kind == 5) ? 0 : plant_len;
    for (size_t i = 0; i < len; ++i)
        d[i] = synth_byte(global_off + i, kind, seed, (const uint8_t *)plant, plen, period ? period : 1);
}
