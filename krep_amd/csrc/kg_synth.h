// kg_synth.h — deterministic synthetic haystacks, identical on host and device.
//
// Every byte is a pure function of (seed, global byte index), so any [off, off+len) slice can be
// produced independently on any rank (256 GiB sharded over 8 GPUs without a host copy).
//   background : '\n' with p = 3/256 (lines of ~85 B), ' ' with p = 40/256, else 'a'..'z'
//   kind 2     : + the plant literal once per `period`-byte stride at a hashed offset inside the
//                stride, + one occurrence straddling every GiB boundary (start = B-3)
//   kind 3     : + the target byte plant[0] with probability 1/100 (u32 % 100 == 0)
//   kind 4     : + one dictionary word per stride (plant = packed dictionary, see DictView)
//   kind 5     : WORD TEXT (round 6; SURVEY.md §8d cfg 1's "ASCII lines, words from a list"; the reference's only published
//                benchmark runs on a natural-language corpus, test/benchmark_krep_vs_rg.sh:4): lines of exactly `period` bytes
//                (the last one '\n'), each a pure function of (seed, line index): words of the packed list `plant` drawn with a
//                Zipf-like law (an octave of ranks uniformly, a rank inside it uniformly: p(rank) ~ 1/rank, integer arithmetic
//                only), single blanks between them, blanks behind the last word that fits
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define KG_HD __host__ __device__ __forceinline__
#else
#define KG_HD static inline
#endif

namespace kg {

KG_HD uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

KG_HD uint8_t background_byte(uint64_t seed, uint64_t g)
{
    const uint64_t r = splitmix64(seed ^ ((g >> 3) * 0xD1342543DE82EF95ull));
    const uint32_t v = (uint32_t)(r >> (8 * (g & 7))) & 0xffu;
    if (v < 3u)
        return (uint8_t)'\n';
    if (v < 43u)
        return (uint8_t)' ';
    return (uint8_t)('a' + (v - 43u) % 26u);
}

constexpr uint64_t kGiB = 1ull << 30;

// packed dictionary for kind 4: [u32 n][n x {u32 off, u32 len}][bytes...]
struct DictView
{
    const uint8_t *base;
    KG_HD uint32_t n() const { return *(const uint32_t *)base; }
    KG_HD uint32_t off(uint32_t i) const { return ((const uint32_t *)base)[1 + 2 * i]; }
    KG_HD uint32_t len(uint32_t i) const { return ((const uint32_t *)base)[2 + 2 * i]; }
};

// start offset of the plant inside stride k (kind 2 / 4)
KG_HD uint64_t plant_offset(uint64_t seed, uint64_t k, uint64_t period, uint64_t plen)
{
    return splitmix64(seed ^ 0xA5A5A5A5DEADBEEFull ^ (k * 0x9FB21C651E98DF25ull)) % (period - plen + 1);
}
// kind 2: does the stride plant get suppressed because it would touch a GiB-boundary plant?
KG_HD bool near_gib(uint64_t s, uint64_t plen)
{
    // the boundary plant occupies [B-3, B-3+plen) for B = j GiB, j >= 1; keep a margin of plen
    const uint64_t B = ((s + kGiB / 2) / kGiB) * kGiB; // nearest boundary
    if (B == 0)
        return false;
    return (s + 2 * plen + 3 > B) && (s < B + 2 * plen);
}

// kind 5: index of the i-th word of line k in a list of n words — octave o of ranks [2^o, 2^(o+1)) uniformly among the
// floor(log2 n) whole octaves, a rank inside it uniformly: every octave carries the same weight, i.e. p(rank) ~ 1 / rank
KG_HD uint32_t word_text_draw(uint64_t seed, uint64_t k, uint32_t i, uint32_t n)
{
    if (n < 2)
        return 0;
    const uint32_t octaves = 31u - (uint32_t)__builtin_clz(n); // 2^octaves <= n
    const uint64_t r = splitmix64(seed ^ 0x7F4A7C15D1B54A33ull ^ (k * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)i * 0xC2B2AE3D27D4EB4Full));
    const uint32_t o = (uint32_t)((r >> 40) % octaves);
    return (1u << o) + ((uint32_t)r & ((1u << o) - 1u)) - 1u; // rank - 1
}
KG_HD uint8_t word_text_byte(uint64_t g, uint64_t seed, const uint8_t *plant, uint64_t period)
{
    const uint64_t k = g / period, j = g % period;
    if (j == period - 1)
        return (uint8_t)'\n';
    DictView dv{plant};
    const uint32_t n = dv.n();
    uint64_t pos = 0;
    for (uint32_t i = 0; pos <= j; ++i)
    {
        const uint32_t w = word_text_draw(seed, k, i, n);
        const uint64_t wl = dv.len(w);
        if (!wl || pos + wl > period - 1)
            break; // the line's remainder stays blank
        if (j < pos + wl)
            return plant[dv.off(w) + (j - pos)];
        pos += wl + 1; // the blank behind the word
    }
    return (uint8_t)' ';
}

KG_HD uint8_t synth_byte(uint64_t g, int kind, uint64_t seed, const uint8_t *plant, uint64_t plen, uint64_t period)
{
    if (kind == 2)
    {
        // GiB-boundary plants first
        const uint64_t B = ((g + 3) / kGiB) * kGiB; // candidate boundary with B-3 <= g
        if (B >= kGiB && g + 3 >= B && g + 3 - B < plen)
            return plant[g + 3 - B];
        const uint64_t k = g / period, s = k * period + plant_offset(seed, k, period, plen);
        if (g >= s && g - s < plen && !near_gib(s, plen))
            return plant[g - s];
        return background_byte(seed, g);
    }
    if (kind == 3)
    {
        const uint64_t r = splitmix64(seed ^ 0x5851F42D4C957F2Dull ^ ((g >> 1) * 0xC2B2AE3D27D4EB4Full));
        const uint32_t u = (uint32_t)(r >> (32 * (g & 1)));
        if (u % 100u == 0u)
            return plant[0];
        return background_byte(seed, g);
    }
    if (kind == 4)
    {
        DictView dv{plant};
        const uint64_t k = g / period;
        const uint32_t w = (uint32_t)(splitmix64(seed ^ 0x1234567ull ^ (k * 0xD6E8FEB86659FD93ull)) % dv.n());
        const uint64_t wl = dv.len(w);
        if (wl && wl <= period)
        {
            const uint64_t s = k * period + plant_offset(seed, k, period, wl);
            if (g >= s && g - s < wl)
                return plant[dv.off(w) + (g - s)];
        }
        return background_byte(seed, g);
    }
    if (kind == 5)
        return word_text_byte(g, seed, plant, period);
    return background_byte(seed, g);
}

} // namespace kg
