// kg_tickets.h — the ticket -> resolver -> deferred-store scheme of the one-pass record writers (kg_single.hip: a single byte or a
// byte set; kg_ac_tiny.hip FUSED: the register-compare dictionary kernel), device side.
//
// A scanning wave draws tickets from ONE counter (so the drawn tickets are always a prefix of the ticket space), ranks the matches
// of a ticket into an LDS ring, publishes the ticket's count (`agg[t] = count | kTkReady`) and goes on; ONE resolver wave — the first
// wave 0 of any block to claim the role, a wave that RUNS whatever part of the grid is resident — turns the counts into exclusive
// prefixes (`pref[t] = first record index | kTkReady`), ticket by ticket as far as the run of published counts extends; a scanning
// wave picks its previous ticket's prefix up one ticket later and writes that ticket's records at their final index.  Progress
// without residency assumptions: with m the smallest unpublished ticket, every prefix up to m is published; m's holder is
// scanning it or waits for the prefix of an EARLIER ticket (published), so it goes on; no wait is circular (kg_single.hip header).
#pragma once
#include <hip/hip_runtime.h>
#include "kg_common.h"

namespace kg {

constexpr unsigned long long kTkReady = 1ull << 63;
constexpr uint32_t kTkSpinLimit = 1u << 24;  // ~0.25 us per spin: seconds — only a logic error gets there (see the safety nets)
constexpr uint32_t kTkResolveChunk = 8;      // tickets per resolver lane and pass (512 per pass)

// The resolver wave's whole life: returns when every ticket has its prefix (ctr->total = the grand total), or — safety net, never
// expected — after flagging the scan as failed-over (ctr->overflow_units) and releasing every waiter with a made-up prefix.
__device__ __forceinline__ void tk_resolve(unsigned long long *__restrict__ agg, unsigned long long *__restrict__ pref,
                                           const unsigned long long n_tickets, Counters *ctr, const uint32_t lane)
{
    typedef unsigned long long u64;
    typedef uint32_t u32;
    // Window of 64 x kTkResolveChunk tickets from `base`; every pass publishes the prefixes of the leading run of ready
    // tickets and moves the window behind it.  In the steady state the scanners are far ahead and a pass takes the whole
    // window; what matters is that the prefix of ticket p never waits for a ticket BEHIND p.
    u64 running = 0, base = 0;
    u32 spins = 0;
    while (base < n_tickets)
    {
        const u64 mine = base + (u64)lane * kTkResolveChunk;
        u64 v[kTkResolveChunk];
        u32 lead = 0, nvalid = 0;
        bool run = true;
#pragma unroll
        for (u32 k = 0; k < kTkResolveChunk; ++k)
        {
            const bool valid = mine + k < n_tickets;
            v[k] = valid ? __hip_atomic_load(&agg[mine + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            nvalid += valid ? 1u : 0u;
            run = run && valid && (v[k] & kTkReady);
            lead += run ? 1u : 0u;
        }
        const u64 open = __ballot(lead != nvalid);                 // lanes whose chunk holds a count that has not arrived
        const u32 f = open ? (u32)__builtin_ctzll(open) : 64u;     // the first of them: the ready run ends inside its chunk
        const u32 take = lane < f ? nvalid : (lane == f ? lead : 0u);
        u64 s = 0;
#pragma unroll
        for (u32 k = 0; k < kTkResolveChunk; ++k)
            s += k < take ? (v[k] & ~kTkReady) : 0ull;
        u64 incl = s;
        u32 tincl = take;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const u64 up = __shfl_up(incl, o);
            const u32 tup = __shfl_up(tincl, o);
            if (lane >= (u32)o)
            {
                incl += up;
                tincl += tup;
            }
        }
        u64 e = running + incl - s;
#pragma unroll
        for (u32 k = 0; k < kTkResolveChunk; ++k)
        {
            if (k < take)
                __hip_atomic_store(&pref[mine + k], e | kTkReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            e += k < take ? (v[k] & ~kTkReady) : 0ull;
        }
        const u32 published = __shfl(tincl, 63);
        running += __shfl(incl, 63);
        base += published;
        if (published)
            spins = 0;
        else
        {
            if (++spins > kTkSpinLimit)
            {
                // safety net (never expected): a count that does not arrive within seconds must not hang the device.  Flag the
                // scan as failed-over (the host re-runs the two-pass kernels) and release every waiter with a made-up prefix.
                if (lane == 0)
                    atomicAdd(&ctr->overflow_units, 1ull);
                for (u64 t = base + lane; t < n_tickets; t += 64)
                    __hip_atomic_store(&pref[t], kTkReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    if (lane == 0)
        ctr->total = running;
}

// the first record index of ticket t, once the resolver has published it (lane 0 spins, the value is broadcast)
__device__ __forceinline__ unsigned long long tk_wait_prefix(const unsigned long long *__restrict__ pref, const unsigned long long t,
                                                              Counters *ctr, const uint32_t lane)
{
    typedef unsigned long long u64;
    u64 p = 0;
    if (lane == 0)
    {
        for (uint32_t spins = 0;; ++spins)
        {
            p = __hip_atomic_load(&pref[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p & kTkReady)
                break;
            if (spins > 2u * kTkSpinLimit) // safety net, as in the resolver: flag the scan, go on with a made-up prefix
            {
                atomicAdd(&ctr->overflow_units, 1ull);
                p = kTkReady;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return (((u64)hi << 32) | lo) & ~kTkReady;
}

} // namespace kg
