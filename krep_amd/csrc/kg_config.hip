// kg_config.hip — the part of the C-ABI (include/krep_gpu.h) that holds no search logic: the error channel, the explicit
// configuration object behind the reference's compile-time SIMD macros and file-static option globals (krep.c:47-74,
// :117-120), device availability (asked by the selector BEFORE an operator is handed out), failure injection and the other
// test hooks.  Host logic only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_internal.h"
#include "kg_plan.h"
#include "kg_replay.h"

using namespace kg;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
namespace kg {
int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    fprintf(stderr, "krep-gpu: %s\n", buf);
    return 2;
}
bool have_error() { return !g_err.empty(); }
} // namespace kg
#define HIPCHK(x)                                                                             \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *krep_gpu_last_error(void) { return g_err.c_str(); }
extern "C" void krep_gpu_clear_error(void) { g_err.clear(); }
extern "C" const char *krep_gpu_version(void) { return "krep-gpu 0.3 (gfx950)"; }
extern "C" int krep_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// ------------------------------------------------------------------------------------ configuration
// The reference decides its algorithm — hence the match-set family — from compile-time SIMD macros and three
// file-static globals (krep.c:47-74, :117-120).  Here they are one explicit krep_gpu_config_t.  The setters below write
// PROCESS-WIDE defaults (relaxed atomics: the reference's globals are process-wide too, set once by main() before the
// pool threads start); krep_gpu_set_thread_config() overrides them for the calling thread; plans and search_buffer_ex()
// carry their configuration explicitly.  Nothing on a scan path writes any of this.
static std::atomic<int> g_simd{KREP_REF_AVX2}, g_only_matching{0}, g_no_simd{0}, g_algo_override{KREP_ALGO_AUTO},
    g_result_order{0}, g_device{-1};
static std::atomic<size_t> g_stream_chunk{0};
static std::atomic<int> g_num_gpus{INT32_MIN};            // INT32_MIN: not set -> $KREP_GPU_NUM, else 1
static std::atomic<size_t> g_min_bytes{SIZE_MAX};         // SIZE_MAX: not set -> $KREP_GPU_MIN_BYTES, else 1 MiB
static thread_local bool tl_cfg_set = false;
static thread_local krep_gpu_config_t tl_cfg;

static int env_device()
{
    const char *e = getenv("KREP_GPU_DEVICE");
    return e && *e ? atoi(e) : 0;
}
static int env_num_gpus()
{
    const char *e = getenv("KREP_GPU_NUM");
    return e && *e ? atoi(e) : 1;
}
static size_t env_min_bytes()
{
    const char *e = getenv("KREP_GPU_MIN_BYTES");
    return e && *e ? (size_t)strtoull(e, nullptr, 0) : ((size_t)1 << 20);
}
extern "C" void krep_gpu_config_default(krep_gpu_config_t *c)
{
    if (!c)
        return;
    c->reference_simd = g_simd.load(std::memory_order_relaxed);
    c->only_matching = g_only_matching.load(std::memory_order_relaxed);
    c->force_no_simd = g_no_simd.load(std::memory_order_relaxed);
    c->algo_override = g_algo_override.load(std::memory_order_relaxed);
    c->result_order = g_result_order.load(std::memory_order_relaxed);
    const int d = g_device.load(std::memory_order_relaxed);
    c->device = d >= 0 ? d : env_device();
    c->stream_chunk_bytes = g_stream_chunk.load(std::memory_order_relaxed);
    const int ng = g_num_gpus.load(std::memory_order_relaxed);
    c->num_gpus = ng != INT32_MIN ? ng : env_num_gpus();
    const size_t mb = g_min_bytes.load(std::memory_order_relaxed);
    c->min_text_bytes = mb != SIZE_MAX ? mb : env_min_bytes();
}
extern "C" void krep_gpu_set_thread_config(const krep_gpu_config_t *c)
{
    tl_cfg_set = c != nullptr;
    if (c)
        tl_cfg = *c;
}
namespace kg {
krep_gpu_config_t current_config()
{
    if (tl_cfg_set)
        return tl_cfg;
    krep_gpu_config_t c;
    krep_gpu_config_default(&c);
    return c;
}
} // namespace kg
extern "C" void krep_gpu_set_reference_simd(int l) { g_simd.store(l, std::memory_order_relaxed); }
extern "C" int krep_gpu_get_reference_simd(void) { return kg::current_config().reference_simd; }
extern "C" void krep_gpu_set_only_matching(int on) { g_only_matching.store(on != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_result_order(int by_start) { g_result_order.store(by_start != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_force_no_simd(int on) { g_no_simd.store(on != 0, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_algo_override(int a) { g_algo_override.store(a, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_device(int d) { g_device.store(d, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_stream_chunk(size_t bytes) { g_stream_chunk.store(bytes, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_num_gpus(int n) { g_num_gpus.store(n, std::memory_order_relaxed); }
extern "C" void krep_gpu_set_min_text_bytes(size_t b) { g_min_bytes.store(b, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------ availability
// "Is there a device this library can run on" is asked by the SELECTOR, before any operator is handed out (SURVEY §8b:
// a backend must be able to fail BEFORE producing output): device count, range of the configured device, gfx950, and one
// probe kernel of this code object launched and read back.  Once per device and process.
__global__ void kg_probe_kernel(unsigned *out) { *out = 0x950u; }
namespace {
struct Avail
{
    std::mutex mu;
    std::vector<int> state;           // per device: 0 unknown, 1 usable, 2 not usable
    std::vector<std::string> reason;  // why not
};
Avail &avail()
{
    static Avail *a = new Avail(); // leaked: may be consulted from atexit paths
    return *a;
}
thread_local std::string tl_unavail;
bool probe_device(int device, std::string &why)
{
    int prev = -1;
    (void)hipGetDevice(&prev);
    auto done = [&](bool ok) {
        if (prev >= 0)
            (void)hipSetDevice(prev);
        (void)hipGetLastError();
        return ok;
    };
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
    {
        why = "hipGetDeviceProperties failed";
        return done(false);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        why = std::string("device is ") + prop.gcnArchName + ", this library holds gfx950 code only";
        return done(false);
    }
    unsigned *d = nullptr, h = 0;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, sizeof(unsigned)) != hipSuccess)
    {
        why = "cannot allocate on the device";
        return done(false);
    }
    hipLaunchKernelGGL(kg_probe_kernel, dim3(1), dim3(1), 0, nullptr, d);
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h == 0x950u;
    (void)hipFree(d);
    if (!ok)
        why = "the gfx950 code object of this library does not run on the device";
    return done(ok);
}
} // namespace
namespace kg {
// NULL = usable; otherwise the reason (valid until the calling thread asks again)
const char *device_unusable(int device)
{
    if (const char *e = getenv("KREP_GPU_DISABLE"))
        if (*e && *e != '0')
            return "disabled by KREP_GPU_DISABLE";
    if (const char *e = getenv("KREP_GPU_ASSUME_AVAILABLE")) // test hook: skip the probe, so that a box WITHOUT a device
        if (*e && *e != '0')                                  // reaches the operators and exercises their run-time failure paths
            return nullptr;
    int ndev = 0;
    const auto t_first = std::chrono::steady_clock::now(); // (the process's first HIP call starts the runtime: kg_cost.hip wants to know)
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
        (void)hipGetLastError();
        return "no HIP device available";
    }
    if (device < 0 || device >= ndev)
    {
        tl_unavail = "device " + std::to_string(device) + " out of range (have " + std::to_string(ndev) + ")";
        return tl_unavail.c_str();
    }
    Avail &a = avail();
    std::lock_guard<std::mutex> lk(a.mu);
    if ((size_t)ndev > a.state.size())
    {
        a.state.resize((size_t)ndev, 0);
        a.reason.resize((size_t)ndev);
    }
    if (a.state[device] == 0)
    {
        a.state[device] = probe_device(device, a.reason[device]) ? 1 : 2;
        if (a.state[device] == 1)
            cost_note_device_init(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_first).count());
    }
    if (a.state[device] == 1)
        return nullptr;
    tl_unavail = a.reason[device];
    return tl_unavail.c_str();
}
// ---- failure injection (test hook): the failure paths of the operators must be reachable on a healthy box
static std::atomic<int> g_inject{-1};
bool inject(int kind)
{
    int v = g_inject.load(std::memory_order_relaxed);
    if (v < 0)
    {
        const char *e = getenv("KREP_GPU_INJECT_FAILURE");
        v = e && *e ? atoi(e) : 0;
        g_inject.store(v, std::memory_order_relaxed);
    }
    return v == kind;
}
} // namespace kg
extern "C" void krep_gpu_debug_inject_failure(int kind) { kg::g_inject.store(kind < 0 ? 0 : kind, std::memory_order_relaxed); }
extern "C" int krep_gpu_available(void) { return kg::device_unusable(kg::current_config().device) == nullptr ? 1 : 0; }
extern "C" const char *krep_gpu_unavailable_reason(void)
{
    const char *r = kg::device_unusable(kg::current_config().device);
    return r ? r : "";
}

namespace kg {
std::atomic<int> g_force_rounds{0};    // test hook: 0 = auto, 1 / 4 = force the tile shape
std::atomic<int> g_force_stage_cap{0}; // test hook: staging records per unit (0 = auto)
extern int g_ac_force_stage_cap;
}
extern "C" void krep_gpu_debug_force_stage_cap(int c) { g_force_stage_cap.store(c); kg::g_ac_force_stage_cap = c; }
extern "C" void krep_gpu_debug_force_rounds(int r) { g_force_rounds.store(r); }
namespace kg {
extern int g_s1_force_grid;
std::atomic<uint64_t> g_fused1_failovers{0}; // one-pass single-byte scans that handed over to the two-pass kernels
}
extern "C" void krep_gpu_debug_force_single_grid(int blocks) { kg::g_s1_force_grid = blocks < 0 ? 0 : blocks; }
extern "C" uint64_t krep_gpu_debug_single_failovers(void) { return g_fused1_failovers.load(); }
namespace kg {
std::atomic<uint64_t> g_fused1_launches{0};
}
extern "C" uint64_t krep_gpu_debug_single_launches(void) { return kg::g_fused1_launches.load(); }
namespace kg {
std::atomic<uint64_t> g_tiny_launches{0}; // launches of ac_tiny_kernel
}
extern "C" uint64_t krep_gpu_debug_tiny_launches(void) { return kg::g_tiny_launches.load(); }
namespace kg {
std::atomic<uint64_t> g_tiny_dense_launches{0}; // ... of its DENSE one-pass flavour
}
extern "C" uint64_t krep_gpu_debug_tiny_dense_launches(void) { return kg::g_tiny_dense_launches.load(); }
namespace kg {
std::atomic<uint64_t> g_ac_anchored_launches{0}; // launches of the multi-pattern kernel's anchored instantiation
}
extern "C" uint64_t krep_gpu_debug_anchored_launches(void) { return kg::g_ac_anchored_launches.load(); }
extern "C" uint64_t krep_gpu_debug_literal_dma_launches(void) { return kg::g_lit_dma_launches.load(); }
extern "C" uint64_t krep_gpu_debug_runs_launches(void) { return kg::g_runs_launches.load(); }

