// kg_ops.hip — the host-buffer side of the C-ABI: the search_func_t operators, search_buffer[_ex](), the drop-in for
// select_search_algorithm(), and the executor that brings a host buffer through HBM:
//   * ONE PIECE (the whole text staged, then scanned) for small inputs and for the three input classes that need the whole
//     text in one window (kg::split_mode() == whole);
//   * PIECES otherwise — contiguous chunks with start-offset ownership and Lmax+1 bytes of halo, spread over the requested
//     devices (search_buffer(num_gpus > 1), the operators with cfg.num_gpus > 1: the reference's chunk loop,
//     krep.c:2816-2905, without its double counting; the shards' counters meet in ONE RCCL all-reduce, kg_comm.hip) and, per
//     device, STREAMED: piece k+1 is copied through the pinned staging ring (mmap'd / pageable source -> pinned -> DMA) while
//     piece k is scanned, two device buffers per device, so a haystack larger than HBM works and the scan time hides under
//     the PCIe time (the reference's counterpart is mmap + MAP_POPULATE, krep.c:2630-2726).  The sequential
//     match-set families run as CHAINED pieces: each takes the boundary record of the one before it (krep_gpu_seq_carry_t).
//   * FAILURE: an attempt that fails appends nothing; the host's registered CPU function answers (run_with_fallback).
// Every device has ONE context (buffers, staging ring, a small plan cache) behind a mutex: the operators are re-entrant
// from any number of threads (SURVEY §8b "Threading", krep.c:1950), calls on one device serialise.  The configuration
// travels explicitly (krep_gpu_config_t); nothing here writes a global.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <cctype>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_common.h"
#include "kg_internal.h"
#include "kg_plan.h"

using namespace kg;

#define HIPCHK(x)                                                                             \
    do                                                                                        \
    {                                                                                         \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            return kg::fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace {
thread_local krep_gpu_shard_info_t tl_shards{1, 1, {0}, 0, 0}; // krep_gpu_last_shard_info()
// ---------------------------------------------------------------------------------------------- device buffers
struct DevBuf
{
    uint8_t *p = nullptr;
    size_t cap = 0;
    int dev = -1;
    int ensure(size_t n, int device)
    {
        if (dev == device && cap >= n && p)
            return 0;
        release();
        HIPCHK(hipSetDevice(device));
        const size_t want = std::max<size_t>(n, 1 << 20);
        if (kg::inject(1))
            return kg::fail("injected failure: device allocation of %zu bytes", want);
        if (hipMalloc(&p, want) != hipSuccess)
        {
            p = nullptr;
            (void)hipGetLastError();
            return kg::fail("hipMalloc of %zu bytes failed on device %d", want, device);
        }
        cap = want;
        dev = device;
        return 0;
    }
    void release()
    {
        if (p)
        {
            (void)hipSetDevice(dev);
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        dev = -1;
    }
};

// ONE allocation per device for the host path: the text buffer(s) and, right behind them, the record buffer.  Where the driver
// places a buffer physically moves the offsets-producing scans by several per cent, and freeing / re-allocating buffers between
// calls is what churns that placement (profiles/r03_box_to_box.txt: the record buffer re-allocated inside one
// process draws 6.58 ... 7.42 ms for the single-byte workload).  Round 3 kept that discipline in bench.py only; here the CLI
// and every host-buffer caller get it: the arena grows (rarely: 1.5x steps), it is never freed and re-made per call, and the
// record area always sits behind the text.  A list that outgrows its area (a second, exact-size pass) takes a buffer of its
// own for that call: the staged text must not move.
struct Arena
{
    uint8_t *base = nullptr;
    size_t cap = 0, text_each = 0, pos_bytes = 0;
    int ntext = 0, dev = -1;
    DevBuf pos_extra; // only while a record list does not fit pos_bytes
    static size_t pos_for(size_t text_bytes) { return std::max<size_t>((size_t)1 << 20, text_bytes / 4 + 4096); } // 16 B per 64 B of text
    // `count` text buffers of `each` bytes (64-byte multiples) + the record area: keeps what is there when it is large enough
    int ensure(size_t each, int count, int device)
    {
        each = (each + 255) & ~(size_t)255;
        if (base && dev == device && ntext >= count && text_each >= each)
            return 0;
        const int n = std::max(count, ntext);
        size_t want_each = std::max(each, text_each);
        if (base && dev == device)
            want_each = std::max(want_each, text_each + text_each / 2); // grow in steps: a run of growing files re-allocates log times
        const size_t want_pos = pos_for(want_each), total = (size_t)n * want_each + want_pos;
        release();
        HIPCHK(hipSetDevice(device));
        if (kg::inject(1))
            return kg::fail("injected failure: device allocation of %zu bytes", total);
        // $KREP_GPU_PLACE_TRIES=k (2..8): an arena of >= 4 GiB is drawn for (kg_place.hip krep_gpu_alloc_placed: up to k candidates, the
        // single-byte workload timed on each, the fastest kept).  Off by default: the host path stages its text over PCIe at <= 55 GB/s, next
        // to which the 2-10 % of a scan that placement moves are not visible, and every draw costs an allocation + ~50 ms of probe scans.
        static const int place_tries = [] { const char *e = getenv("KREP_GPU_PLACE_TRIES"); return e ? atoi(e) : 1; }();
        if (place_tries > 1 && total >= ((size_t)4 << 30))
        {
            void *t = nullptr, *r = nullptr;
            // (the arena's layout: n text buffers in a row, the record area behind them — the probe treats the row as one text)
            if (krep_gpu_alloc_placed(device, (size_t)n * want_each - 64, want_pos, place_tries, &t, &r, nullptr) == 0)
            {
                base = (uint8_t *)t;
                cap = total;
                ntext = n;
                text_each = want_each;
                pos_bytes = want_pos;
                dev = device;
                return 0;
            }
            krep_gpu_clear_error(); // (the plain allocation below is tried before anything is reported)
        }
        if (hipMalloc(&base, total) != hipSuccess)
        {
            base = nullptr;
            (void)hipGetLastError();
            // not enough room for the growth step: exactly what was asked for
            want_each = each;
            const size_t exact = (size_t)count * each + pos_for(each);
            if (hipMalloc(&base, exact) != hipSuccess)
            {
                base = nullptr;
                (void)hipGetLastError();
                return kg::fail("hipMalloc of %zu bytes failed on device %d", exact, device);
            }
            cap = exact;
            ntext = count;
            text_each = each;
            pos_bytes = pos_for(each);
            dev = device;
            return 0;
        }
        cap = total;
        ntext = n;
        text_each = want_each;
        pos_bytes = want_pos;
        dev = device;
        return 0;
    }
    uint8_t *text(int i) const { return base + (size_t)i * text_each; }
    // the record area for `bytes` of records: behind the text when it fits, a buffer of its own otherwise
    uint8_t *pos(size_t bytes, int device)
    {
        if (base && bytes <= pos_bytes)
            return base + (size_t)ntext * text_each;
        return pos_extra.ensure(bytes, device) ? nullptr : pos_extra.p;
    }
    void release()
    {
        if (base)
        {
            (void)hipSetDevice(dev);
            (void)hipFree(base);
        }
        base = nullptr;
        cap = text_each = pos_bytes = 0;
        ntext = 0;
        dev = -1;
        pos_extra.release();
    }
};

// Host buffer -> HBM through two pinned staging buffers: the CPU copy of chunk k+1 overlaps the DMA of chunk k
// (a pageable hipMemcpy stages serially).  PCIe-bound by construction (<= ~55 GB/s); this rate is reported separately and
// is never the roofline number.
struct Stager
{
    static constexpr size_t kChunk = 32u << 20;
    static constexpr size_t kSlack = 1u << 20; // a tail of up to this much rides with the last full chunk: a piece's few halo bytes
                                               // as a chunk of their own would leave the CPU nothing to do while a DMA still owns
                                               // the other staging buffer (measured: one DMA time, 0.4 ms, lost per piece)
    uint8_t *pin[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int dev = -1;
    bool ready = false; // every resource below exists (set last by init(); a half-built stager is torn down)
    void release()
    {
        ready = false;
        if (dev < 0)
            return;
        (void)hipSetDevice(dev);
        for (int i = 0; i < 2; ++i)
        {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            pin[i] = nullptr;
            done[i] = nullptr;
        }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        dev = -1;
        seq = 0;
    }
    int init(int device)
    {
        if (dev == device && ready)
            return 0;
        release();
        HIPCHK(hipSetDevice(device));
        dev = device;
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipHostMalloc(&pin[i], kChunk + kSlack) == hipSuccess &&
                 hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        if (!ok)
        {
            (void)hipGetLastError();
            release(); // a recoverable out-of-memory must not leave NULL buffers behind for the next call
            return kg::fail("cannot create the pinned staging ring on device %d", device);
        }
        ready = true;
        return 0;
    }
    uint64_t seq = 0; // staging chunks issued so far: the ring runs on across calls (no drain between the pieces of a stream)
    // queues the whole copy on the staging stream and returns when the last CPU-side chunk copy has been handed to the DMA
    // engine; `after` (may be NULL) is recorded behind the last DMA.  The two pinned buffers alternate across calls, so
    // piece k+1's first chunk is staged while piece k's last DMA is still running.
    int copy_async(uint8_t *d_dst, const char *src, size_t len, hipEvent_t after)
    {
        if (!ready)
            return kg::fail("staging ring not initialised");
        if (kg::inject(2))
            return kg::fail("injected failure: host->device copy");
        HIPCHK(hipSetDevice(dev));
        for (size_t off = 0, n = 0; off < len; off += n, ++seq)
        {
            const int b = (int)(seq & 1);
            n = len - off <= kChunk + kSlack ? len - off : kChunk;
            if (seq >= 2)
                HIPCHK(hipEventSynchronize(done[b])); // the DMA that last used this staging buffer
            {
                // The staging copy feeds a DMA engine that takes 57.6 GB/s from pinned memory on this part
                // (tools/ubench/host_register.hip, profiles/r04_ingest.txt); one host thread copies 12-30 GB/s depending on the
                // host, so the copy is split over just enough helper threads to stay ahead of the DMA — measured once per
                // process on the first full chunk (round 3 always used 4).  Zero-copy ingest (hipHostRegister of the caller's
                // mapping, krep.c:2630-2726) was measured and is NOT used: pinning 2 GiB costs 90-106 ms, the ring moves them
                // in 39 ms.  $KREP_GPU_COPY_THREADS overrides.
                static std::atomic<int> threads{0};
                int kT = threads.load(std::memory_order_relaxed);
                if (kT == 0)
                {
                    const char *e = getenv("KREP_GPU_COPY_THREADS");
                    kT = e && *e ? std::min(8, std::max(1, atoi(e))) : 0;
                    if (kT == 0 && n >= ((size_t)8 << 20))
                    {
                        const size_t probe = (size_t)4 << 20; // (this part of the chunk is copied again below: a one-off 4 MiB)
                        const auto t0 = std::chrono::steady_clock::now();
                        memcpy(pin[b], src + off, probe);
                        const double gbps = probe / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
                        kT = std::min(4, std::max(2, (int)std::ceil(64.0 / std::max(1.0, gbps))));
                    }
                    if (kT)
                        threads.store(kT, std::memory_order_relaxed);
                    else
                        kT = 2; // a small first chunk: decide later
                }
                std::thread th[8];
                const size_t part = (n + (size_t)kT - 1) / (size_t)kT;
                for (int q = 1; q < kT; ++q)
                {
                    const size_t o = (size_t)q * part;
                    if (o < n)
                        th[q - 1] = std::thread([=] { memcpy(pin[b] + o, src + off + o, std::min(part, n - o)); });
                }
                memcpy(pin[b], src + off, std::min(part, n));
                for (auto &t : th)
                    if (t.joinable())
                        t.join();
            }
            HIPCHK(hipMemcpyAsync(d_dst + off, pin[b], n, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(done[b], st));
        }
        if (after)
            HIPCHK(hipEventRecord(after, st));
        return 0;
    }
    int copy(uint8_t *d_dst, const char *src, size_t len)
    {
        if (!ready)
            return kg::fail("staging ring not initialised");
        if (kg::inject(2))
            return kg::fail("injected failure: host->device copy");
        HIPCHK(hipSetDevice(dev));
        if (len < (4u << 20)) // small buffers: one synchronous copy is cheaper than the pipeline
        {
            HIPCHK(hipMemcpy(d_dst, src, len, hipMemcpyHostToDevice));
            return 0;
        }
        if (copy_async(d_dst, src, len, nullptr))
            return 2;
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
};

// ---------------------------------------------------------------------------------------------- plan cache
// The CLI calls the operator once per file (krep.c:1950) with the same params, and building a plan costs device
// allocations + (multi-pattern) table construction.  Keyed by every field the scan depends on, configuration included.
struct PlanKey
{
    std::vector<std::vector<uint8_t>> pats;
    bool cs = true, lines = false, track = false, ww = false, regex = false;
    size_t max_count = SIZE_MAX;
    int simd = 0, only_matching = 0, no_simd = 0, algo = 0, device = 0;
    bool operator==(const PlanKey &o) const
    {
        return pats == o.pats && cs == o.cs && lines == o.lines && track == o.track && ww == o.ww && regex == o.regex &&
               max_count == o.max_count && simd == o.simd && only_matching == o.only_matching && no_simd == o.no_simd &&
               algo == o.algo && device == o.device;
    }
};
PlanKey key_of(const search_params_t *p, const krep_gpu_config_t &c, int device)
{
    PlanKey k;
    if (p->num_patterns >= 1 && p->patterns && p->pattern_lens)
        for (size_t i = 0; i < p->num_patterns; ++i)
            k.pats.emplace_back((const uint8_t *)p->patterns[i], (const uint8_t *)p->patterns[i] + p->pattern_lens[i]);
    else if (p->pattern)
        k.pats.emplace_back((const uint8_t *)p->pattern, (const uint8_t *)p->pattern + p->pattern_len);
    k.cs = p->case_sensitive; k.lines = p->count_lines_mode; k.track = p->track_positions; k.ww = p->whole_word;
    k.regex = p->use_regex;
    k.max_count = p->max_count;
    k.simd = c.reference_simd; k.only_matching = c.only_matching; k.no_simd = c.force_no_simd; k.algo = c.algo_override;
    k.device = device;
    return k;
}

// ---------------------------------------------------------------------------------------------- per-device context
struct DeviceCtx
{
    std::mutex mu; // one host-buffer operation at a time per device
    int device = 0;
    struct Entry
    {
        PlanKey key;
        krep_gpu_plan_t *plan = nullptr;
        uint64_t tick = 0;
    };
    std::vector<Entry> plans; // small LRU
    uint64_t tick = 0;
    Arena mem; // text buffer(s) + record area, one allocation
    Stager stager;

    krep_gpu_plan_t *plan_for(const search_params_t *p, const krep_gpu_config_t &c)
    {
        PlanKey k = key_of(p, c, device);
        for (auto &e : plans)
            if (e.key == k)
            {
                e.tick = ++tick;
                return e.plan;
            }
        krep_gpu_config_t cc = c;
        cc.device = device;
        krep_gpu_plan_t *pl = krep_gpu_plan_create_ex(p, &cc);
        if (!pl)
            return nullptr;
        constexpr size_t kMaxPlans = 4;
        if (plans.size() >= kMaxPlans)
        {
            size_t victim = 0;
            for (size_t i = 1; i < plans.size(); ++i)
                if (plans[i].tick < plans[victim].tick)
                    victim = i;
            krep_gpu_plan_destroy(plans[victim].plan);
            plans.erase(plans.begin() + (long)victim);
        }
        plans.push_back(Entry{std::move(k), pl, ++tick});
        return pl;
    }
    void release()
    {
        for (auto &e : plans)
            krep_gpu_plan_destroy(e.plan);
        plans.clear();
        mem.release();
        stager.release();
    }
};
std::mutex g_ctx_mu;
std::vector<std::unique_ptr<DeviceCtx>> *g_ctx = nullptr; // leaked at process exit on purpose: the HIP runtime may be gone by then
DeviceCtx *ctx_for(int device)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_ctx)
        g_ctx = new std::vector<std::unique_ptr<DeviceCtx>>();
    if ((size_t)device >= g_ctx->size())
        g_ctx->resize((size_t)device + 1);
    if (!(*g_ctx)[device])
    {
        (*g_ctx)[device].reset(new DeviceCtx());
        (*g_ctx)[device]->device = device;
    }
    return (*g_ctx)[device].get();
}
} // namespace

extern "C" void krep_gpu_release_device_resources(void)
{
    std::vector<DeviceCtx *> all;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        if (g_ctx)
            for (auto &c : *g_ctx)
                if (c)
                    all.push_back(c.get());
    }
    for (DeviceCtx *c : all)
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->release();
    }
    kg::format_release();
}

namespace kg {
// memchr_search's final flush (krep.c:3976-3991 + :4026-4038): when max_count is a multiple of the
// 4096-entry batch and more matches exist, the (max_count+1)-th record is stored FIRST (in front of
// the last batch) and the max_count-th is dropped.
void memchr_batch_quirk(match_position_t *recs, uint64_t have, size_t maxc)
{
    if (maxc == SIZE_MAX || maxc == 0 || have <= maxc || (maxc % 4096) != 0)
        return;
    const uint64_t f = maxc - 4096;
    match_position_t extra = recs[maxc];
    memmove(recs + f + 1, recs + f, 4095 * sizeof(match_position_t));
    recs[f] = extra;
}
} // namespace kg

// ------------------------------------------------------------------------------------------------ one piece
// The whole text in one device buffer: every reference convention (max_count corners, batch quirk, sequential families,
// end-of-text replay) is applied by krep_gpu_scan_device_ex() itself.
static int run_whole(DeviceCtx &cx, const search_params_t *params, const krep_gpu_config_t &cfg, const char *text, size_t text_len,
                     match_result_t *result, uint64_t *ret_out)
{
    *ret_out = 0;
    krep_gpu_plan_t *pl = cx.plan_for(params, cfg);
    if (!pl)
        return 2;
    if (pl->ref_algo == KREP_RA_AHO_CORASICK && !params->ac_trie)
        return 0; // aho_corasick.c:306: no trie, no matches
    if (cx.mem.ensure(text_len + 64, 1, cx.device))
        return 2;
    uint8_t *const d_text = cx.mem.text(0);
    if (text_len && (cx.stager.init(cx.device) || cx.stager.copy(d_text, text, text_len)))
    {
        if (!kg::have_error())
            kg::fail("H2D copy failed");
        return 2;
    }
    const bool want_pos = params->track_positions && result != nullptr && !params->count_lines_mode;
    uint64_t cap = 0;
    if (want_pos)
    {
        cap = std::max<uint64_t>(1u << 16, text_len / 64);
        if (params->max_count != SIZE_MAX)
            cap = std::min<uint64_t>(cap, (uint64_t)params->max_count + 1);
        cap = std::max<uint64_t>(cap, 1);
    }
    krep_gpu_scan_out_t so;
    match_position_t *d_pos = nullptr;
    for (int attempt = 0;; ++attempt)
    {
        d_pos = cap ? (match_position_t *)cx.mem.pos(cap * sizeof(match_position_t), cx.device) : nullptr;
        if (cap && !d_pos)
            return 2;
        if (krep_gpu_scan_device_ex(pl, d_text, text_len, 0, text_len, 0, text_len, d_pos, cap, nullptr, 0, &so))
            return 2;
        if (!so.overflow || attempt == 1)
            break;
        cap = so.total_matches + 1; // exact size, second and last pass
    }
    *ret_out = so.count;
    if (want_pos && so.stored)
    {
        if (cfg.result_order && pl->ref_algo == KREP_RA_AHO_CORASICK &&
            krep_gpu_order_by_start(d_pos, so.stored, text_len, nullptr))
            return 2;
        std::vector<match_position_t> tmp(so.stored);
        if (kg::inject(4) || hipMemcpy(tmp.data(), d_pos, so.stored * sizeof(match_position_t), hipMemcpyDeviceToHost) != hipSuccess)
            return kg::fail("D2H copy of the records failed");
        uint64_t n = so.stored;
        const int algo = pl->ref_algo == KREP_RA_AHO_CORASICK ? KREP_RA_AHO_CORASICK : mirror_effective(pl->ref_algo, &pl->sp, text_len);
        if (algo == KREP_RA_MEMCHR && params->max_count != SIZE_MAX)
        {
            memchr_batch_quirk(tmp.data(), n, params->max_count);
            n = std::min<uint64_t>(n, params->max_count);
            if (cfg.result_order) // the caller skips its qsort: hand the (one displaced) record back in file order
                std::sort(tmp.begin(), tmp.begin() + (long)n, [](const match_position_t &a, const match_position_t &b) {
                    return a.start_offset < b.start_offset;
                });
        }
        if (!result_reserve(result, n))
            return kg::fail("out of memory growing match_result_t");
        memcpy(result->positions + result->count, tmp.data(), n * sizeof(match_position_t));
        result->count += n;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ pieces
static std::atomic<uint64_t> g_chain_rescans{0}, g_chain_replays{0}; // chain fix-ups since the process started (test hook)
extern "C" void krep_gpu_debug_chain_fixups(uint64_t *rescans, uint64_t *replays)
{
    if (rescans) *rescans = g_chain_rescans.load();
    if (replays) *replays = g_chain_replays.load();
}
namespace {
struct Piece
{
    size_t lo = 0, hi = 0; // owned window in the text
    size_t b0 = 0, b1 = 0; // bytes staged: [b0, b1) superset of [lo, hi)
    int device = 0;
    int shard = 0; // logical shard (one per requested GPU) this piece belongs to
    krep_gpu_scan_out_t out{};
    std::vector<match_position_t> recs;
    // kSplitChain: the boundary record this piece was scanned with and the one it leaves
    krep_gpu_seq_carry_t carry_used{}, carry_out{};
};
inline bool rec_less(const match_position_t &a, const match_position_t &b) // emission order of aho_corasick_search
{
    return a.end_offset != b.end_offset ? a.end_offset < b.end_offset : a.start_offset < b.start_offset;
}

// all pieces of one device, in text order; piece k+1 is staged by a helper thread while piece k is scanned
struct DeviceRun
{
    DeviceCtx *cx = nullptr;
    std::vector<Piece *> pieces;
    const search_params_t *params = nullptr;
    krep_gpu_config_t cfg{};
    const char *buf = nullptr;
    size_t len = 0;
    bool want_pos = false;
    bool chain = false; // kSplitChain: pieces in text order, each with its predecessor's boundary record
    int rc = 0;
    std::string err;
};

// one piece, resident in d_text: scan (second pass with an exact buffer when the position buffer was too small), records to
// the host.  carry_in == NULL: nothing in front of the piece is consumed.
int scan_one_piece(DeviceCtx &cx, krep_gpu_plan_t *pl, const uint8_t *d_text, Piece *p, size_t global_len, bool want_pos,
                   const krep_gpu_seq_carry_t *carry_in)
{
    const size_t nb = p->b1 - p->b0;
    uint64_t cap = want_pos ? std::max<uint64_t>(1u << 16, nb / 64) : 0;
    p->carry_used = carry_in ? *carry_in : krep_gpu_seq_carry_t{};
    p->recs.clear();
    match_position_t *d_pos = nullptr;
    for (int attempt = 0;; ++attempt)
    {
        d_pos = cap ? (match_position_t *)cx.mem.pos(cap * sizeof(match_position_t), cx.device) : nullptr;
        if (cap && !d_pos)
            return 2;
        if (krep_gpu_scan_device_seq(pl, d_text, nb, p->lo - p->b0, p->hi - p->b0, p->b0, global_len, d_pos, cap, nullptr, 0, carry_in,
                                     &p->carry_out, &p->out))
            return 2;
        if (!p->out.overflow || attempt == 1)
            break;
        cap = p->out.total_matches + 1;
    }
    if (want_pos && p->out.stored)
    {
        p->recs.resize(p->out.stored);
        if (kg::inject(4) ||
            hipMemcpy(p->recs.data(), d_pos, p->out.stored * sizeof(match_position_t), hipMemcpyDeviceToHost) != hipSuccess)
            return kg::fail("D2H copy of the records failed");
    }
    return 0;
}

// A worker thread of a multi-device run goes to the CPUs of its device's NUMA node before it touches anything: the pinned
// staging ring it allocates, the staging copies (its helper threads inherit the mask) and the record lists it fills then live
// next to the PCIe root the DMA goes through — on a two-socket node half the devices are a socket away from a thread the OS
// placed at random, and a copy across the socket link feeds the DMA engine at a fraction of the local rate.  Only OUR threads
// are moved (the single-device paths run on the caller's thread and leave its affinity alone); every step is optional: no
// sysfs entry, one node, or a failing call leave the thread where it is.
static void bind_thread_near_device(int device)
{
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, device) != hipSuccess)
    {
        (void)hipGetLastError();
        return;
    }
    for (char *c = bdf; *c; ++c)
        *c = (char)tolower(*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (!f)
        return;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0)
        return;
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f)
        return;
    char list[4096] = {0};
    const bool ok = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!ok)
        return;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = 0;
    for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n"))
    {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k == 1)
            b = a;
        if (k >= 1)
            for (int c = a; c <= b && c < CPU_SETSIZE; ++c, ++n)
                CPU_SET(c, &set);
    }
    if (n)
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

void run_device(DeviceRun *dr)
{
    DeviceCtx &cx = *dr->cx;
    std::lock_guard<std::mutex> lk(cx.mu);
    dr->rc = 2;
    search_params_t local = *dr->params;
    local.max_count = SIZE_MAX; // prefix-ordered truncation happens after the pieces are merged
    krep_gpu_plan_t *pl = cx.plan_for(&local, dr->cfg);
    if (!pl || hipSetDevice(cx.device) != hipSuccess || cx.stager.init(cx.device))
    {
        dr->err = krep_gpu_last_error();
        return;
    }
    const size_t np = dr->pieces.size();
    size_t maxb = 0;
    for (Piece *p : dr->pieces)
        maxb = std::max(maxb, p->b1 - p->b0);
    if (cx.mem.ensure(maxb + 64, np > 1 ? 2 : 1, cx.device))
    {
        dr->err = krep_gpu_last_error();
        return;
    }
    // producer: queues the staging of piece k into buffer k & 1 once piece k-2 has been consumed; ready[k & 1] fires when
    // its last DMA is done.  The staging ring never drains between pieces.
    hipEvent_t ready[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i)
        if (hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) != hipSuccess)
        {
            dr->err = "event creation failed";
            return;
        }
    std::mutex m;
    std::condition_variable cv;
    size_t staged = 0, consumed = 0;
    bool stage_failed = false, stop = false;
    std::string stage_err;
    std::thread producer([&] {
        for (size_t k = 0; k < np; ++k)
        {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || k < consumed + 2; });
                if (stop)
                    return;
            }
            Piece *p = dr->pieces[k];
            const int rc = cx.stager.copy_async(cx.mem.text((int)(k & 1)), dr->buf + p->b0, p->b1 - p->b0, ready[k & 1]);
            std::lock_guard<std::mutex> l(m);
            if (rc)
            {
                stage_failed = true;
                stage_err = krep_gpu_last_error(); // thread-local of the producer: hand it over
                cv.notify_all();
                return;
            }
            staged = k + 1;
            cv.notify_all();
        }
    });
    auto finish = [&](bool ok) {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
        }
        cv.notify_all();
        producer.join();
        (void)hipStreamSynchronize(cx.stager.st);
        for (int i = 0; i < 2; ++i)
            (void)hipEventDestroy(ready[i]);
        if (ok)
            dr->rc = 0;
    };
    for (size_t k = 0; k < np; ++k)
    {
        {
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return stage_failed || staged > k; });
            if (stage_failed)
            {
                dr->err = "staging failed: " + stage_err;
                l.unlock();
                finish(false);
                return;
            }
        }
        if (hipEventSynchronize(ready[k & 1]) != hipSuccess) // the piece's last DMA
        {
            dr->err = "staging DMA failed";
            finish(false);
            return;
        }
        Piece *p = dr->pieces[k];
        // chain: inside a shard every piece takes its predecessor's boundary record; the FIRST piece of a shard is scanned
        // optimistically — its left neighbour is another device's work, still running — and checked by run_pieces afterwards
        // (one rule for every layout: shards that share a device on a small box take the same road as eight devices)
        const krep_gpu_seq_carry_t *cin = nullptr;
        if (dr->chain && k > 0 && dr->pieces[k - 1]->shard == p->shard)
            cin = &dr->pieces[k - 1]->carry_out;
        if (scan_one_piece(cx, pl, cx.mem.text((int)(k & 1)), p, dr->len, dr->want_pos, cin))
        {
            dr->err = krep_gpu_last_error();
            finish(false);
            return;
        }
        {
            std::lock_guard<std::mutex> l(m);
            consumed = k + 1;
        }
        cv.notify_all();
    }
    finish(true);
}
} // namespace

constexpr size_t kChainTailPiece = 256u << 10; // the piece that holds the end of the text, chained multi-shard runs (run_pieces)

static int run_pieces(const search_params_t *params, const krep_gpu_config_t &cfg, const char *buf, size_t len, int num_gpus,
                      size_t chunk, match_result_t *out, uint64_t *ret_out)
{
    *ret_out = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return kg::fail("no HIP device available");
    size_t lmax = 1;
    for (size_t i = 0; i < params->num_patterns; ++i)
        lmax = std::max(lmax, params->pattern_lens[i]);
    const int algo = params->num_patterns > 1 ? KREP_RA_AHO_CORASICK : mirror_effective(mirror_top(params, cfg), params, len);
    int G = std::max(1, num_gpus);
    if ((size_t)G > len / 4096 + 1)
        G = (int)(len / 4096 + 1);
    const size_t ctx = lmax + 1; // pattern_len-1 to complete straddling matches, +1 for -w, +1 slack
    const size_t share = (len + (size_t)G - 1) / (size_t)G;
    std::vector<Piece> pcs;
    for (int g = 0; g < G; ++g)
    {
        const size_t glo = std::min(len, (size_t)g * share), ghi = std::min(len, glo + share);
        if (ghi <= glo && !(len == 0 && g == 0))
            continue;
        const size_t step = chunk ? std::max<size_t>(chunk, 4096) : std::max<size_t>(ghi - glo, 1);
        size_t lo = glo;
        do
        {
            Piece p;
            p.lo = lo;
            p.hi = std::min(ghi, lo + step);
            p.b0 = p.lo > ctx ? p.lo - ctx : 0;
            p.b1 = std::min(len, p.hi + ctx);
            p.device = (cfg.device + g) % ndev;
            p.shard = g;
            pcs.push_back(std::move(p));
            lo += step;
        } while (lo < ghi);
    }
    if (pcs.size() >= 2 && pcs.back().hi - pcs.back().lo < 4096)
    { // a sliver at the end joins its neighbour (the piece that ends the text must hold the end-of-text replay window)
        Piece &prev = pcs[pcs.size() - 2];
        prev.hi = pcs.back().hi;
        prev.b1 = pcs.back().b1;
        pcs.pop_back();
    }
    const bool want_pos = params->track_positions && out != nullptr && !params->count_lines_mode;
    const bool chain = kg::split_mode(params, cfg, len) == kSplitChain;
    if (chain && G > 1 && !pcs.empty() && pcs.back().hi == len && pcs.back().hi - pcs.back().lo > 2 * kChainTailPiece)
    {
        // Sequential families over several shards: the piece that ENDS the text is the only one whose result depends on the
        // line-skip history of everything in front of it (the end-of-text replay, kg_replay.h) — and every shard starts from a
        // zero record, so whenever an earlier shard held an accepted occurrence that piece is staged and scanned again below.
        // Its last 256 KiB become a piece of their own (same device, scanned behind its neighbour with that neighbour's record):
        // the repeat then costs 256 KiB, not a shard.
        Piece t = pcs.back();
        Piece &big = pcs.back();
        big.hi = len - kChainTailPiece;
        big.b1 = std::min(len, big.hi + ctx);
        t.lo = big.hi;
        t.b0 = t.lo > ctx ? t.lo - ctx : 0;
        pcs.push_back(std::move(t));
    }
    // one worker per PHYSICAL device (several logical shards may share one on a small box)
    std::vector<DeviceRun> runs;
    for (Piece &p : pcs)
    {
        DeviceRun *dr = nullptr;
        for (auto &r : runs)
            if (r.cx->device == p.device)
                dr = &r;
        if (!dr)
        {
            runs.emplace_back();
            dr = &runs.back();
            dr->cx = ctx_for(p.device);
            dr->params = params;
            dr->cfg = cfg;
            dr->buf = buf;
            dr->len = len;
            dr->want_pos = want_pos;
            dr->chain = chain;
        }
        dr->pieces.push_back(&p);
    }
    if (runs.size() == 1)
        run_device(&runs[0]);
    else
    {
        std::vector<std::thread> th;
        for (auto &r : runs)
            th.emplace_back([](DeviceRun *d) {
                if (!getenv("KREP_GPU_NO_NUMA_BIND"))
                    bind_thread_near_device(d->cx->device);
                run_device(d);
            }, &r);
        for (auto &t : th)
            t.join();
    }
    for (auto &r : runs)
        if (r.rc)
            return kg::fail("device %d failed: %s", r.cx->device, r.err.c_str());
    if (chain)
    {
        // The one exchange step of the sequential families: walk the pieces in text order and compare what each
        // was scanned with against what its left neighbour really left.  They differ only where a cluster of overlapping
        // occurrences straddles a cut between two devices (the last consumed occurrence reaches < m bytes into the next
        // shard); that piece is staged and scanned again with the true record — and, should its own record change, the one
        // behind it.
        krep_gpu_seq_carry_t tc{}; // what the text in front of the current piece REALLY leaves
        // -c with a newline inside a pattern (the emission-order line changes of aho_corasick_search, the window-grid walk of
        // simd_sse42_search / kmp_search): the piece's count depends on the EXACT record in front of it
        bool nl_chain = false;
        if (params->count_lines_mode)
            for (size_t i = 0; i < params->num_patterns; ++i)
                nl_chain = nl_chain || (params->pattern_lens[i] && memchr(params->patterns[i], '\n', params->pattern_lens[i]));
        for (Piece &p : pcs)
        {
            // walks: the piece's own list depends on where the scan stands at its start; block-loop -c: only the piece that
            // ends the text depends on the line-skip history
            // (multi-pattern -c with a newline inside a pattern: every piece's count depends on the newlines and the last match's
            //  line in front of it — for every other family these two fields stay 0)
            const bool stale = std::max<uint64_t>(p.carry_used.resume, p.lo) != std::max<uint64_t>(tc.resume, p.lo) ||
                               (p.hi == len && (p.carry_used.q1 != tc.q1 || p.carry_used.nl1 != tc.nl1 || p.carry_used.g0 != tc.g0)) ||
                               p.carry_used.nl_before != tc.nl_before || p.carry_used.last_line != tc.last_line ||
                               (nl_chain && p.carry_used.resume != tc.resume);
            // ... and when the line-skip history is ALL that differs for the piece that ends the text (the rule: every shard
            // starts from a zero record, so this is every multi-shard -c search through the block loops with an occurrence in an
            // earlier shard), only the end-of-text replay runs again — on the last 512 bytes of the text, with the piece's own
            // canonical count and contribution as they stand in its record (krep_gpu_replay_tail)
            const bool replay_only = stale && p.hi == len && p.carry_out.local_lines != 0 && !nl_chain &&
                                     std::max<uint64_t>(p.carry_used.resume, p.lo) == std::max<uint64_t>(tc.resume, p.lo) &&
                                     p.carry_used.nl_before == tc.nl_before && p.carry_used.last_line == tc.last_line &&
                                     !getenv("KREP_GPU_NO_REPLAY_FIXUP");
            if (replay_only)
            {
                DeviceCtx &cx = *ctx_for(p.device);
                std::lock_guard<std::mutex> lk(cx.mu);
                search_params_t local = *params;
                local.max_count = SIZE_MAX;
                krep_gpu_plan_t *pl = cx.plan_for(&local, cfg);
                const size_t tail = std::min<size_t>(len, 512);
                uint64_t lines = 0;
                krep_gpu_seq_carry_t co{};
                if (!pl || cx.mem.ensure(tail + 64, 1, cx.device) || cx.stager.init(cx.device) ||
                    cx.stager.copy(cx.mem.text(0), buf + (len - tail), tail) ||
                    krep_gpu_replay_tail(pl, cx.mem.text(0), tail, len, nullptr, &tc, &p.carry_out, &co, &lines))
                    return 2;
                p.carry_used = tc;
                p.carry_out = co;
                p.out.line_count = p.out.count = p.out.total_matches = lines; // (a -c scan: the count IS the line count; max_count is applied to the fold)
                g_chain_replays.fetch_add(1, std::memory_order_relaxed);
            }
            else if (stale)
            {
                DeviceCtx &cx = *ctx_for(p.device);
                std::lock_guard<std::mutex> lk(cx.mu);
                search_params_t local = *params;
                local.max_count = SIZE_MAX;
                krep_gpu_plan_t *pl = cx.plan_for(&local, cfg);
                if (!pl || cx.mem.ensure(p.b1 - p.b0 + 64, 1, cx.device) || cx.stager.init(cx.device) ||
                    cx.stager.copy(cx.mem.text(0), buf + p.b0, p.b1 - p.b0) || scan_one_piece(cx, pl, cx.mem.text(0), &p, len, want_pos, &tc))
                    return 2;
                g_chain_rescans.fetch_add(1, std::memory_order_relaxed);
            }
            // fold this piece's own contribution onto the true record (for a piece scanned with the true record this
            // reproduces its carry_out)
            tc = kg::fold_carry(tc, p.carry_out);
        }
    }
    // ---- the shards' counters meet: per logical shard one slot {matches, lines, head, tail, has_nl}, each
    // device fills the slots of its own shards, ONE RCCL all-reduce (uint64 sum over xGMI) makes every device hold all of them
    // — the sum doubles as the all-gather the left-to-right line fold needs.  The reference's counterpart is the host loop
    // over thread_args[] (krep.c:2930-3016).  A single shard has nothing to reduce.
    constexpr size_t kSlot = 5;
    std::vector<std::vector<unsigned long long>> vecs(runs.size(), std::vector<unsigned long long>(kSlot * (size_t)G, 0ull));
    for (size_t r = 0; r < runs.size(); ++r)
        for (int g = 0; g < G; ++g)
        {
            std::vector<krep_gpu_scan_out_t> outs;
            for (Piece *p : runs[r].pieces)
                if (p->shard == g)
                    outs.push_back(p->out);
            if (outs.empty())
                continue;
            unsigned long long *slot = &vecs[r][kSlot * (size_t)g];
            for (auto &o : outs)
                slot[0] += o.total_matches;
            slot[1] = krep_gpu_combine_line_counts(outs.data(), (int)outs.size());
            for (size_t i = 0; i < outs.size(); ++i) // a match before the shard's first newline
            {
                slot[2] |= outs[i].head_line_hit;
                if (outs[i].has_newline)
                    break;
            }
            for (size_t i = outs.size(); i-- > 0;) // ... after its last
            {
                slot[3] |= outs[i].tail_line_hit;
                if (outs[i].has_newline)
                    break;
            }
            for (auto &o : outs)
                slot[4] |= o.has_newline;
        }
    tl_shards = krep_gpu_shard_info_t{G, (int)runs.size(), {0}, 0, 0};
    for (size_t r = 0; r < runs.size() && r < 16; ++r)
        tl_shards.device_ids[r] = runs[r].cx->device;
    if (G > 1)
    {
        std::vector<int> devs;
        for (auto &r : runs)
            devs.push_back(r.cx->device);
        if (kg::allreduce_across_devices(devs, vecs) == 0)
        {
            tl_shards.reduced_by = 1;
            tl_shards.comm_ranks = kg::comm_clique_ranks(devs);
        }
        else
        {
            // This process already holds every shard's slots; the collective is how they are MEANT to meet, not a
            // reason to throw a finished search away: when librccl cannot be loaded or a communicator cannot be created the
            // host adds the slot vectors itself.  The reason stays in krep_gpu_last_error(), reported once.
            static std::atomic<bool> told{false};
            if (!told.exchange(true))
                fprintf(stderr, "krep-gpu: RCCL unavailable for the %d-shard count reduction; summing on the host\n", G);
            for (size_t r = 1; r < vecs.size(); ++r)
                for (size_t i = 0; i < vecs[0].size(); ++i)
                    vecs[0][i] += vecs[r][i];
            tl_shards.reduced_by = 2;
        }
    }
    uint64_t total = 0;
    std::vector<krep_gpu_scan_out_t> shard_outs((size_t)G);
    for (int g = 0; g < G; ++g)
    {
        const unsigned long long *slot = &vecs[0][kSlot * (size_t)g];
        total += slot[0];
        shard_outs[g].line_count = slot[1];
        shard_outs[g].head_line_hit = slot[2] != 0;
        shard_outs[g].tail_line_hit = slot[3] != 0;
        shard_outs[g].has_newline = slot[4] != 0;
    }
    const uint64_t lines = krep_gpu_combine_line_counts(shard_outs.data(), G);
    const size_t maxc = params->max_count;
    uint64_t ret;
    if (maxc == 0)
        ret = (algo == KREP_RA_BMH || algo == KREP_RA_MEMCHR_SHORT || algo == KREP_RA_AVX2 || algo == KREP_RA_AVX512) &&
                      !params->count_lines_mode && !params->track_positions
                  ? (total > 0 ? 1 : 0) // count-only: the first hit makes 1 >= 0 true (krep.c:1355-1367)
                  : 0;
    else
        ret = std::min<uint64_t>(params->count_lines_mode ? lines : total, maxc);
    *ret_out = ret;
    if (want_pos && ret && maxc == SIZE_MAX)
    {
        // No truncation: the piece lists go STRAIGHT into the caller's block, in parallel (at BASELINE config 3 density that is
        // 5.5 GB; one thread copying it twice — into a scratch vector, then into the result — cost seconds).
        std::vector<size_t> off(pcs.size() + 1, 0);
        for (size_t i = 0; i < pcs.size(); ++i)
            off[i + 1] = off[i] + pcs[i].recs.size();
        const size_t tot = off.back();
        if (!result_reserve(out, tot))
            return kg::fail("out of memory growing match_result_t");
        match_position_t *dst = out->positions + out->count;
        {
            std::atomic<size_t> next{0};
            auto work = [&] {
                for (size_t i; (i = next.fetch_add(1)) < pcs.size();)
                {
                    if (!pcs[i].recs.empty())
                        memcpy(dst + off[i], pcs[i].recs.data(), pcs[i].recs.size() * sizeof(match_position_t));
                    std::vector<match_position_t>().swap(pcs[i].recs);
                }
            };
            const size_t nt = tot * sizeof(match_position_t) < (64u << 20) ? 1 : std::min<size_t>(8, pcs.size());
            std::vector<std::thread> th;
            for (size_t t = 1; t < nt; ++t)
                th.emplace_back(work);
            work();
            for (auto &t : th)
                t.join();
        }
        if (params->num_patterns > 1)
            for (size_t i = 1; i < pcs.size(); ++i)
            {
                // every piece list is in the reference's (end, start) order and owns its matches by START: only a suffix of
                // what precedes a cut and a prefix of what follows it can interleave — merge exactly that zone
                match_position_t *first_new = dst + off[i], *end = dst + off[i + 1];
                if (first_new == dst || first_new == end)
                    continue;
                match_position_t *zone_lo = std::upper_bound(dst, first_new, *first_new, rec_less);
                match_position_t *zone_hi = std::upper_bound(first_new, end, *(first_new - 1), rec_less);
                std::inplace_merge(zone_lo, first_new, zone_hi, rec_less);
            }
        if (params->num_patterns > 1 && cfg.result_order) // the formatter's order (krep.c:420-434)
            std::sort(dst, dst + tot, [](const match_position_t &a, const match_position_t &b) {
                return a.start_offset != b.start_offset ? a.start_offset < b.start_offset : a.end_offset < b.end_offset;
            });
        out->count += tot;
    }
    else if (want_pos && ret)
    {
        std::vector<match_position_t> all;
        all.reserve((size_t)total);
        for (auto &p : pcs)
        {
            const size_t old = all.size();
            all.insert(all.end(), p.recs.begin(), p.recs.end());
            if (params->num_patterns > 1 && old && all.size() > old)
            {
                const auto first_new = all.begin() + (long)old;
                const auto zone_lo = std::upper_bound(all.begin(), first_new, *first_new, rec_less);
                const auto zone_hi = std::upper_bound(first_new, all.end(), *(first_new - 1), rec_less);
                std::inplace_merge(zone_lo, first_new, zone_hi, rec_less);
            }
            std::vector<match_position_t>().swap(p.recs);
        }
        uint64_t n = std::min<uint64_t>(all.size(), ret);
        if (algo == KREP_RA_KMP && maxc != SIZE_MAX && all.size() > maxc)
            n = maxc + 1; // krep.c:1717-1724
        if (algo == KREP_RA_MEMCHR)
            memchr_batch_quirk(all.data(), all.size(), maxc);
        if ((params->num_patterns > 1 || algo == KREP_RA_MEMCHR) && cfg.result_order) // the formatter's order (krep.c:420-434)
            std::sort(all.begin(), all.begin() + (long)n, [](const match_position_t &a, const match_position_t &b) {
                return a.start_offset != b.start_offset ? a.start_offset < b.start_offset : a.end_offset < b.end_offset;
            });
        if (!result_reserve(out, n))
            return kg::fail("out of memory growing match_result_t");
        memcpy(out->positions + out->count, all.data(), n * sizeof(match_position_t));
        out->count += n;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ dispatcher
namespace {
// One view of the caller's params for everything below: legacy callers fill only pattern / pattern_len (the reference's own
// tests, test/test_krep.c:233-235), the CLI fills the arrays as well.  After this, num_patterns >= 1, patterns / pattern_lens
// are valid and pattern / pattern_len name the first one (run_pieces read pattern_lens[] of the raw struct).
struct NormParams
{
    search_params_t sp{};
    const char *one_pat = nullptr;
    size_t one_len = 0;
    bool valid = false;
    explicit NormParams(const search_params_t *p)
    {
        sp = *p;
        if (p->num_patterns > 1)
        {
            valid = p->patterns && p->pattern_lens;
            return;
        }
        if (p->num_patterns == 1 && p->patterns && p->pattern_lens && p->patterns[0])
        {
            one_pat = p->patterns[0];
            one_len = p->pattern_lens[0];
        }
        else if (p->pattern)
        {
            one_pat = p->pattern;
            one_len = p->pattern_len;
        }
        else
            return;
        sp.pattern = one_pat;
        sp.pattern_len = one_len;
        sp.patterns = &one_pat;
        sp.pattern_lens = &one_len;
        sp.num_patterns = 1;
        valid = true;
    }
    NormParams(const NormParams &) = delete;
    NormParams &operator=(const NormParams &) = delete;
};

// restores the calling thread's current HIP device: the operators are called on the HOST's threads (krep.c:1950), and a
// library must not leave their device changed
struct DeviceGuard
{
    int prev = -1;
    DeviceGuard() { (void)hipGetDevice(&prev); (void)hipGetLastError(); }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};
thread_local int tl_status = KREP_GPU_OK;
std::atomic<krep_gpu_cpu_select_t> g_cpu_select{nullptr};
} // namespace

extern "C" int krep_gpu_last_status(void) { return tl_status; }
extern "C" void krep_gpu_last_shard_info(krep_gpu_shard_info_t *out)
{
    if (out)
        *out = tl_shards;
}
extern "C" void krep_gpu_set_cpu_fallback(krep_gpu_cpu_select_t f) { g_cpu_select.store(f); }

// the GPU attempt: *status 0 = the return value and `result` are good; 2 = failed, nothing appended
static uint64_t run_host_operator(const search_params_t *raw, const char *text, size_t text_len, match_result_t *result,
                                  const krep_gpu_config_t &cfg, int num_gpus, int *status)
{
    *status = 2;
    if (!raw || (!text && text_len))
    {
        kg::fail("NULL params/text");
        return 0;
    }
    NormParams np(raw);
    if (!np.valid)
    {
        kg::fail("no pattern");
        return 0;
    }
    const search_params_t *params = &np.sp;
    if (const char *why = kg::unsupported_reason(params, cfg))
    {
        kg::fail("%s", why);
        return 0;
    }
    if (const char *why = kg::device_unusable(cfg.device))
    {
        kg::fail("%s", why);
        return 0;
    }
    if (params->num_patterns > 1 && !params->ac_trie)
    { // aho_corasick.c:306: no trie, no matches
        *status = 0;
        return 0;
    }
    DeviceGuard guard;
    int ndev = 1;
    (void)hipGetDeviceCount(&ndev);
    if (num_gpus <= 0)
        num_gpus = ndev; // "all visible devices"
    // streaming threshold: pieces of `chunk` bytes once the text is larger than two of them
    const size_t chunk = cfg.stream_chunk_bytes ? cfg.stream_chunk_bytes : ((size_t)128 << 20);
    const bool can_split = kg::shardable(params, cfg, text_len);
    const bool split = can_split && (num_gpus > 1 || text_len > 2 * chunk);
    uint64_t ret = 0;
    int rc;
    const uint64_t count0 = result ? result->count : 0;
    tl_shards = krep_gpu_shard_info_t{1, 1, {cfg.device}, 0, 0};
    if (!split)
    {
        DeviceCtx *cx = ctx_for(cfg.device);
        std::lock_guard<std::mutex> lk(cx->mu);
        rc = run_whole(*cx, params, cfg, text, text_len, result, &ret);
    }
    else
        rc = run_pieces(params, cfg, text, text_len, num_gpus, text_len > 2 * chunk ? chunk : 0, result, &ret);
    if (rc)
    {
        if (result)
            result->count = count0; // a failed attempt leaves the caller's list as it found it
        return 0;
    }
    *status = 0;
    return ret;
}

// GPU attempt, then — on ANY failure (no device, refused input class, allocation, copy, launch) — the host's own CPU
// function, if it registered its selector.  A 0 that means "could not look" is never returned with status OK
// (SURVEY §8b "Errors"; the reference's fallback idiom krep.c:1944-1948, its error channel krep.c:2940-2947).
static uint64_t run_with_fallback(const search_params_t *params, const char *text, size_t text_len, match_result_t *result,
                                  const krep_gpu_config_t &cfg, int num_gpus, int *status_out, bool allow_fallback = true)
{
    int st = 2;
    krep_gpu_clear_error();
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t n = run_host_operator(params, text, text_len, result, cfg, num_gpus, &st);
    if (st == 0)
    {
        kg::cost_note_host_path(text_len, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        tl_status = KREP_GPU_OK;
        if (status_out) *status_out = 0;
        return n;
    }
    const krep_gpu_cpu_select_t sel = g_cpu_select.load();
    search_func_t cpu = (sel && params && allow_fallback) ? sel(params) : nullptr;
    if (cpu == krep_gpu_literal_search || cpu == krep_gpu_aho_corasick_search)
        cpu = nullptr; // a selector that hands our own operators back would recurse
    if (!cpu)
    {
        tl_status = KREP_GPU_FAILED;
        if (status_out) *status_out = 2;
        return 0;
    }
    static std::atomic<bool> warned{false};
    if (!warned.exchange(true))
        fprintf(stderr, "krep-gpu: falling back to the CPU function for this search (reported once)\n");
    const uint64_t count0 = result ? result->count : 0;
    const uint64_t r = cpu(params, text, text_len, result);
    if (result && cfg.result_order && result->count > count0 + 1)
        // krep_gpu_set_result_order(1) promises the formatter's (start, end) order (krep.c:420-434) whoever produced the records
        std::stable_sort(result->positions + count0, result->positions + result->count,
                         [](const match_position_t &a, const match_position_t &b) {
                             return a.start_offset != b.start_offset ? a.start_offset < b.start_offset : a.end_offset < b.end_offset;
                         });
    tl_status = KREP_GPU_FELL_BACK;
    if (status_out) *status_out = 0;
    return r;
}

extern "C" uint64_t krep_gpu_literal_search(const search_params_t *params, const char *text, size_t len, match_result_t *result)
{
    const krep_gpu_config_t cfg = kg::current_config();
    return run_with_fallback(params, text, len, result, cfg, cfg.num_gpus, nullptr);
}
extern "C" uint64_t krep_gpu_aho_corasick_search(const search_params_t *params, const char *text, size_t len,
                                                 match_result_t *result)
{
    const krep_gpu_config_t cfg = kg::current_config();
    return run_with_fallback(params, text, len, result, cfg, cfg.num_gpus, nullptr);
}
extern "C" search_func_t krep_gpu_select_search_algorithm(const search_params_t *params)
{
    if (!params)
        return nullptr;
    const krep_gpu_config_t cfg = kg::current_config();
    // NULL: the caller keeps the CPU function pointer select_search_algorithm() gives it — for an input class that is not
    // reproduced, and when there is no device this library can run on (asked HERE, before any operator is handed out)
    if (kg::unsupported_reason(params, cfg) || kg::device_unusable(cfg.device))
        return nullptr;
    if (cfg.num_gpus != 1)
    {
        // a sharding host: create the devices' communicator NOW — the selector runs before the host has written a byte of
        // output, so RCCL's first-communicator banner (muted by redirecting fd 1 for that moment, kg_comm.hip) cannot swallow
        // output of another host thread later.  A failure here is not an error: run_pieces() reports and sums on the host.
        DeviceGuard guard;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 1)
        {
            const int g = cfg.num_gpus <= 0 ? ndev : std::min(cfg.num_gpus, ndev);
            std::vector<int> devs;
            for (int i = 0; i < g; ++i)
                devs.push_back((cfg.device + i) % ndev);
            if (devs.size() > 1 && kg::comm_warmup(devs))
                krep_gpu_clear_error();
        }
    }
    return params->num_patterns > 1 ? krep_gpu_aho_corasick_search : krep_gpu_literal_search;
}

// search_string()'s validation and verdict (krep.c:2013-2049, :2166-2199), minus strlen and printing
extern "C" int search_buffer_ex(const search_params_t *params, const char *buf, size_t len, const krep_gpu_config_t *cfg_in,
                                int num_gpus, match_result_t *out, uint64_t *count_out)
{
    if (count_out)
        *count_out = 0;
    tl_status = KREP_GPU_FAILED; // until the search below says otherwise
    if (!params || params->num_patterns == 0 || !params->patterns || !params->pattern_lens)
        return kg::fail("Error: No pattern specified.");
    if (!buf && len)
        return kg::fail("Error: NULL text in search_buffer.");
    if (params->use_regex)
        return kg::fail("regex search is not accelerated; keep krep's regex_search for it");
    for (size_t i = 0; i < params->num_patterns; ++i)
    {
        if (params->pattern_lens[i] == 0)
        {
            if (params->num_patterns > 1)
                return kg::fail("Error: Empty pattern provided for literal search with multiple patterns.");
        }
        else if (params->pattern_lens[i] > 1024) // MAX_PATTERN_LENGTH, krep.c:77
            return kg::fail("Error: Pattern too long (max 1024).");
    }
    const krep_gpu_config_t cfg = cfg_in ? *cfg_in : kg::current_config();
    int st = 2;
    search_params_t local = *params;
    static int dummy_trie;
    const bool no_trie = local.num_patterns > 1 && !local.ac_trie;
    if (no_trie)
        local.ac_trie = (ac_trie_t *)&dummy_trie; // search_string builds the trie itself (krep.c:2067-2078)
    // (a CPU function could not use that placeholder: the fallback needs the caller's real trie)
    uint64_t n = run_with_fallback(&local, buf, len, out, cfg, num_gpus, &st, !no_trie);
    if (st)
        return 2;
    const size_t maxc = params->max_count;
    if (maxc != SIZE_MAX && n > maxc)
        n = maxc;
    if (out && maxc != SIZE_MAX && out->count > maxc)
        out->count = maxc;
    bool found;
    if (params->count_lines_mode || params->count_matches_mode)
        found = n > 0;
    else
    {
        found = out && out->count > 0;
        if (found)
            n = out->count;
        else if (!out)
            found = n > 0;
    }
    if (count_out)
        *count_out = n;
    return found ? 0 : 1;
}
extern "C" int search_buffer(const search_params_t *params, const char *buf, size_t len, int only_matching, int num_gpus,
                             match_result_t *out, uint64_t *count_out)
{
    krep_gpu_config_t cfg = kg::current_config();
    cfg.only_matching = only_matching != 0;
    return search_buffer_ex(params, buf, len, &cfg, num_gpus, out, count_out);
}
