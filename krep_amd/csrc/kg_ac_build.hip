// kg_ac_build.hip — host side of the multi-pattern scan: ac_build() turns a dictionary into the tables of kg_ac_tables.h
// (what ac_trie_build and its helpers are to aho_corasick_search, /root/reference/aho_corasick.c:74-297 — here a reversed trie,
// an exact-class 4-gram filter for LDS in two layouts, chain-compressed 4-gram buckets, exact bitmaps of the short patterns and
// the register-compare description of a tiny dictionary), ac_free() releases them.  No kernel in this file (round 5: split out
// of kg_ac.hip, which keeps the scan kernel and its drivers).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/krep_gpu.h"
#include "kg_ac_common.h"
#include "kg_ac_tables.h"
#include "kg_internal.h"

namespace kg {

#define ACHK(x)                                                                                \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__);       \
            goto bad;                                                                          \
        }                                                                                      \
    } while (0)

static inline uint8_t ac_lo8(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

AcTables *ac_build(const search_params_t &sp, int device)
{
    auto *t = new AcTables();
    t->device = device;
    t->ci = !sp.case_sensitive;
    t->npat = (u32)sp.num_patterns;
    if (sp.num_patterns > 4095)
    {
        fail("too many patterns (%zu > 4095; the reference CLI accepts 1024)", (size_t)sp.num_patterns);
        delete t;
        return nullptr;
    }
    std::vector<std::vector<uint8_t>> pats;
    for (size_t i = 0; i < sp.num_patterns; ++i)
    {
        std::vector<uint8_t> p((const uint8_t *)sp.patterns[i], (const uint8_t *)sp.patterns[i] + sp.pattern_lens[i]);
        if (t->ci)
            for (auto &c : p)
                c = ac_lo8(c); // the trie is built on folded bytes (aho_corasick.c:161)
        if (p.empty())
        {
            t->has_empty = true; // only ever matches the empty text (aho_corasick.c:441-463)
            continue;
        }
        if (p.size() > 1024)
        {
            fail("pattern %zu longer than 1024 bytes", i);
            delete t;
            return nullptr;
        }
        if (memchr(p.data(), '\n', p.size()))
            t->has_nl = true;
        t->lmax = std::max<u32>(t->lmax, (u32)p.size());
        t->lmin = t->lmin ? std::min<u32>(t->lmin, (u32)p.size()) : (u32)p.size();
        pats.push_back(std::move(p));
    }
    t->pats_h = pats; // (the anchor decision of kg_ac_anchor.hip, taken when the first large text arrives)
    // ---- exact bitmaps of the short patterns (verifier) ----
    std::vector<u32> S1, S2, S3;
    for (auto &p : pats)
    {
        const size_t n = p.size();
        auto setbit = [&](std::vector<u32> &v, u32 words, u32 key) {
            if (v.empty())
                v.assign(words, 0);
            if ((v[key >> 5] >> (key & 31)) & 1u)
                t->short_dup = true;
            v[key >> 5] |= 1u << (key & 31);
        };
        if (n == 1) { t->has1 = 1; setbit(S1, kS1Words, p[0]); }
        else if (n == 2) { t->has2 = 1; setbit(S2, kS2Words, (u32)p[0] | ((u32)p[1] << 8)); }
        else if (n == 3) { t->has3 = 1; setbit(S3, kS3Words, (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16)); }
        else t->has4 = 1;
    }
    // ---- tiny dictionary?  (kg_ac_tiny.hip: compared in registers, no tables)
    {
        AcTiny &td = t->tiny;
        // ... and worth it: with a 1- or 2-byte pattern the general kernel reads two or three LDS tables per position (3.3 TB/s
        // on `he she hers`, 1.4 on `e t`); a dictionary of 3- and 4-byte patterns only runs there at 5.2-5.9 TB/s, faster than
        // the register compare (profiles/r04_dictionaries.txt)
        // Lengths: 1..4, or 1..3 beside ONE length of 5..8 (`-e a -e Sherlock`), which then takes the place of length 4
        bool ok = !pats.empty() && !t->has_empty && t->lmax <= 8 && t->lmin <= 2 && !getenv("KREP_GPU_AC_NO_TINY");
        const u32 llong = t->lmax > 4 ? t->lmax : 0u;
        bool five = false; // 4-byte patterns beside the long length: the long ones become a fifth class (AcTiny::five)
        for (auto &q : pats)
            five = five || (llong && q.size() == 4);
        for (size_t i = 0; ok && i < pats.size(); ++i)
        {
            for (size_t k = 0; k < i; ++k)
                if (pats[k] == pats[i])
                    ok = false; // a duplicate reports twice (aho_corasick.c:383-437): the masks cannot count copies
            const size_t L = pats[i].size();
            if (llong && L > 4 && L != llong)
                ok = false; // a second length beyond 4 bytes
            if (five && L > 4)
            {
                if (!ok || td.n5 >= kTinyPer)
                {
                    ok = false;
                    break;
                }
                const u32 p = td.n5++;
                for (size_t s = 0; s < 4; ++s) // its LAST four bytes
                {
                    const uint8_t c = pats[i][L - 1 - s];
                    td.pk5[p] |= (u32)c << (8 * s);
                    td.lf5[p] |= (t->ci && c >= 'a' && c <= 'z') ? 1u << s : 0u;
                }
                for (size_t s = 0; s < L - 4; ++s) // ... and its first L - 4, byte s = s places before the end of that part
                {
                    const uint8_t c = pats[i][L - 5 - s];
                    td.pk2[p] |= (u32)c << (8 * s);
                    td.lf2[p] |= (t->ci && c >= 'a' && c <= 'z') ? 1u << s : 0u;
                }
                continue;
            }
            const size_t cls = L > 4 ? 4 : L; // the class a pattern is compared and reported in
            if (!ok || td.n[cls - 1] >= kTinyPer)
            {
                ok = false;
                break;
            }
            const u32 p = td.n[cls - 1]++;
            for (size_t s = 0; s < cls; ++s) // (a long pattern: its LAST four bytes)
            {
                const uint8_t c = pats[i][L - 1 - s];
                td.pk[cls - 1][p] |= (u32)c << (8 * s);
                td.lf[cls - 1][p] |= (t->ci && c >= 'a' && c <= 'z') ? 1u << s : 0u;
            }
            for (size_t s = 0; L > 4 && s < L - 4; ++s) // ... and its first L - 4, byte s = s places before the end of that part
            {
                const uint8_t c = pats[i][L - 5 - s];
                td.pk2[p] |= (u32)c << (8 * s);
                td.lf2[p] |= (t->ci && c >= 'a' && c <= 'z') ? 1u << s : 0u;
            }
        }
        td.llong = ok ? llong : 0u;
        td.five = (ok && five) ? 1u : 0u;
        td.lmax = t->lmax;
        for (int L = 0; L < 4; ++L)
            td.ncls += td.n[L] ? 1u : 0u;
        td.ok = ok ? 1u : 0u;
        if (ok && t->lmax == 1 && pats.size() >= 2 && pats.size() <= 4)
        {
            t->set_n = (u32)pats.size();
            for (size_t i = 0; i < pats.size(); ++i)
                t->set_b[i] = pats[i][0];
        }
    }
    // ---- reversed trie ----
    std::unordered_map<u32, u32> edge; // key = node << 8 | byte
    std::vector<u32> copies(1, 0);
    for (auto &p : pats)
    {
        u32 node = 0;
        for (size_t k = p.size(); k-- > 0;)
        {
            const u32 key = (node << 8) | p[k];
            auto it = edge.find(key);
            if (it == edge.end())
            {
                const u32 nn = (u32)copies.size();
                copies.push_back(0);
                edge.emplace(key, nn);
                node = nn;
            }
            else
                node = it->second;
        }
        copies[node]++;
    }
    t->nnodes = (u32)copies.size();
    if (t->nnodes >= (1u << 23))
    {
        fail("pattern set too large (%u trie nodes)", t->nnodes);
        delete t;
        return nullptr;
    }
    u32 cap = 1024;
    while (cap < edge.size() * 2 + 16)
        cap <<= 1;
    t->emask = cap - 1;
    std::vector<uint2> tab(cap, make_uint2(0xffffffffu, 0u));
    for (auto &kv : edge)
    {
        u32 h = (kv.first * kHashMul) >> 7;
        while (tab[h & t->emask].x != 0xffffffffu)
            ++h;
        tab[h & t->emask] = make_uint2(kv.first, kv.second | (copies[kv.second] ? 0x80000000u : 0u));
    }
    // exact 4-gram -> depth-4 node (used when every pattern has >= 4 bytes)
    std::vector<uint2> g4;
    std::vector<uint4> g4x;
    {
        struct Item { u32 node, depth, gram; };
        std::vector<std::vector<std::pair<u32, u32>>> kids(t->nnodes); // node -> (byte, child)
        for (auto &kv : edge)
            kids[kv.first >> 8].push_back({kv.first & 255u, kv.second});
        std::vector<Item> st{{0u, 0u, 0u}}, d4;
        while (!st.empty())
        {
            Item it = st.back();
            st.pop_back();
            if (it.depth == 4)
            {
                d4.push_back(it);
                continue;
            }
            for (auto &bc : kids[it.node])
                // depth-1 byte is text[i] (top byte of E), depth-4 byte is text[i-3] (low byte)
                st.push_back({bc.second, it.depth + 1, it.gram | (bc.first << (8 * (3 - it.depth)))});
        }
        // sparse on purpose (load <= 1/16 while the table stays <= 8 MiB): a probe sequence is a chain of DEPENDENT L2
        // round trips that the whole 64-candidate batch waits for; at load 1/2 the longest of ~80 probes took 4-5
        // steps and the verify stage 3x as long (2.30 -> 3.34 TB/s on config 4 with the stride-2 filter)
        u32 gcap = 1024;
        while (gcap < d4.size() * 16 + 16 && (size_t)gcap * 2 * 32 <= (8u << 20))
            gcap <<= 1;
        while (gcap < d4.size() * 2 + 16)
            gcap <<= 1;
        t->g4mask = gcap - 1;
        g4.assign(gcap, make_uint2(0u, 0u));
        g4x.assign(2 * (size_t)gcap, make_uint4(0u, 0u, 0u, 0u));
        for (auto &it : d4)
        {
            u32 h = (it.gram * kHashMul) >> 9;
            while (g4[h & t->g4mask].y != 0u)
                ++h;
            const u32 child = it.node | (copies[it.node] ? 0x80000000u : 0u);
            g4[h & t->g4mask] = make_uint2(it.gram, child);
            // the unary chain below the depth-4 node (ac_walk_fast): <= 12 bytes, depths 5..16
            u32 node = it.node, clen = 0, endmask = 0;
            bool simple = copies[node] <= 1;
            uint8_t cb[12] = {0};
            while (clen < 12 && kids[node].size() == 1)
            {
                const u32 byte = kids[node][0].first;
                node = kids[node][0].second;
                cb[11 - clen] = (uint8_t)byte; // depth 5 + clen <-> window byte 16 - depth
                if (copies[node])
                {
                    endmask |= 1u << clen;
                    if (copies[node] != 1)
                        simple = false;
                }
                ++clen;
            }
            const u32 info = clen | (simple ? kG4Simple : 0u) | (!kids[node].empty() ? kG4Cont : 0u);
            auto word = [&](int w) { return (u32)cb[4 * w] | ((u32)cb[4 * w + 1] << 8) | ((u32)cb[4 * w + 2] << 16) | ((u32)cb[4 * w + 3] << 24); };
            g4x[2 * (size_t)(h & t->g4mask)] = make_uint4(it.gram, child, info, endmask);
            g4x[2 * (size_t)(h & t->g4mask) + 1] = make_uint4(word(0), word(1), word(2), 0u);
        }
        // Preferred layout: buckets of two entries (one 64-byte line) with NO overfull bucket, so that a probe is one
        // round trip without a loop; searched over a few multipliers and sizes up to 8 MiB.  Linear probing (above)
        // stays as the fallback for dictionaries too large for that (the chance that three of n keys share one of nb
        // buckets is ~ n^3 / (6 nb^2)).
        static const u32 muls[] = {0x9E3779B1u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Cu};
        // (KREP_GPU_AC_LINEAR=1: test hook, keeps the linear-probing layout)
        for (u32 nb = 1024; nb <= (8u << 20) / 64 && !t->g4x_mode && !d4.empty() && !getenv("KREP_GPU_AC_LINEAR"); nb <<= 1)
        {
            if ((u64)nb * 2 < d4.size())
                continue;
            for (u32 mul : muls)
            {
                std::vector<uint8_t> fill(nb, 0);
                bool ok = true;
                for (auto &it : d4)
                    if (++fill[((it.gram * mul) >> 9) & (nb - 1)] > 2)
                    {
                        ok = false;
                        break;
                    }
                if (!ok)
                    continue;
                std::vector<uint4> bk(4 * (size_t)nb, make_uint4(0u, 0u, 0u, 0u));
                std::fill(fill.begin(), fill.end(), 0);
                for (u32 slot = 0; slot <= t->g4mask; ++slot)
                {
                    const uint4 e0 = g4x[2 * (size_t)slot], e1 = g4x[2 * (size_t)slot + 1];
                    if (e0.y == 0u)
                        continue;
                    const u32 b = ((e0.x * mul) >> 9) & (nb - 1);
                    const u32 way = fill[b]++;
                    bk[4 * (size_t)b + 2 * way] = e0;
                    bk[4 * (size_t)b + 2 * way + 1] = e1;
                }
                g4x.swap(bk);
                t->g4x_mode = 1;
                t->g4x_mask = nb - 1;
                t->g4x_mul = mul;
                break;
            }
        }
    }
    ACHK(hipMalloc(&t->d_gram4, g4.size() * sizeof(uint2)));
    ACHK(hipMemcpy(t->d_gram4, g4.data(), g4.size() * sizeof(uint2), hipMemcpyHostToDevice));
    ACHK(hipMalloc(&t->d_g4x, g4x.size() * sizeof(uint4)));
    ACHK(hipMemcpy(t->d_g4x, g4x.data(), g4x.size() * sizeof(uint4), hipMemcpyHostToDevice));
    {
        // ---- filter: exact-class table over the last 4 bytes; a pattern shorter than 4 sets every class of the
        //      bytes in front of it (32 / 1024 / 32768 entries) ----
        std::vector<u32> X20((1u << kXBitsBig) / 32, 0), X19((1u << kXBitsLines) / 32, 0), S20, S19;
        // pair == true: the slot layout of the stride-2 kernel without -c (ac_pair_slot)
        auto expand = [&](std::vector<u32> &T20, std::vector<u32> *T19, const uint8_t *last, size_t known, bool pair) {
            // the `known` (<= 4) classes next to the tested position are fixed (last[0..known), text order), the rest free
            u32 fixed = 0;
            for (size_t q = 0; q < known; ++q)
                fixed |= ((u32)last[q] & 31u) << (5 * (4 - known + q));
            const u32 nfree = 1u << (5 * (4 - known));
            for (u32 f = 0; f < nfree; ++f)
            {
                const u32 x = fixed | f;
                if (pair)
                {
                    u32 dw, bit;
                    ac_pair_slot(x, dw, bit);
                    T20[dw] |= 1u << bit;
                }
                else
                    T20[x >> 5] |= 1u << (x & 31);
                if (T19 && pair)
                {
                    u32 dw, bit;
                    ac_pair_slot(x, dw, bit);
                    (*T19)[dw & ((1u << (kXBitsLines - 5)) - 1u)] |= 1u << bit; // the kernel masks the byte address with 0xfffc
                }
                else if (T19)
                {
                    const u32 y = x & ((1u << kXBitsLines) - 1u);
                    (*T19)[y >> 5] |= 1u << (y & 31);
                }
            }
        };
        for (auto &p : pats)
        {
            const size_t n = p.size(), known = std::min<size_t>(n, 4);
            expand(X20, &X19, p.data() + (n - known), known, false);
        }
        ACHK(hipMalloc(&t->d_filterx20, X20.size() * sizeof(u32)));
        ACHK(hipMemcpy(t->d_filterx20, X20.data(), X20.size() * sizeof(u32), hipMemcpyHostToDevice));
        ACHK(hipMalloc(&t->d_filterx19, X19.size() * sizeof(u32)));
        ACHK(hipMemcpy(t->d_filterx19, X19.data(), X19.size() * sizeof(u32), hipMemcpyHostToDevice));
        // stride-2 table: final gram (match ends at the tested position) + the gram one byte earlier (the match ends
        // one later: its last byte is not part of the tested gram, one class fewer is known)
        if (!t->has1)
        {
            S20.assign(X20.size(), 0);
            S19.assign(X19.size(), 0);
            for (auto &p : pats)
            {
                const size_t n = p.size(), k4 = std::min<size_t>(n, 4), known = std::min<size_t>(n - 1, 4);
                expand(S20, &S19, p.data() + (n - k4), k4, true);                // the match ends at the tested position
                expand(S20, &S19, p.data() + (n - 1 - known), known, true);      // ... one byte behind it
            }
            u64 e1 = 0, e2 = 0;
            for (size_t w = 0; w < X20.size(); ++w)
            {
                e1 += (u64)__builtin_popcount(X20[w]);
                e2 += (u64)__builtin_popcount(S20[w]);
            }
            // worth it while the denser table keeps the candidate volume in the same range (per byte e2 / 2^21 against
            // e1 / 2^20, two ends to verify per candidate) or small in absolute terms (<= 0.4 % of the even positions,
            // the rate of BASELINE config 4).  KREP_GPU_AC_STRIDE1=1 forces the one-position filter.
            const bool force2 = getenv("KREP_GPU_AC_STRIDE2") != nullptr; // (measurement hook: the stride-2 filter whatever its density)
            if ((((e2 <= 6 * e1 + 64 || e2 <= 4096) && e2 < (1u << kXBitsBig) / 64) || force2) && !getenv("KREP_GPU_AC_STRIDE1"))
            {
                ACHK(hipMalloc(&t->d_filters20, S20.size() * sizeof(u32)));
                ACHK(hipMemcpy(t->d_filters20, S20.data(), S20.size() * sizeof(u32), hipMemcpyHostToDevice));
                ACHK(hipMalloc(&t->d_filters19, S19.size() * sizeof(u32)));
                ACHK(hipMemcpy(t->d_filters19, S19.data(), S19.size() * sizeof(u32), hipMemcpyHostToDevice));
            }
        }
        if (!S1.empty())
        {
            ACHK(hipMalloc(&t->d_s1, S1.size() * sizeof(u32)));
            ACHK(hipMemcpy(t->d_s1, S1.data(), S1.size() * sizeof(u32), hipMemcpyHostToDevice));
        }
        if (!S2.empty())
        {
            ACHK(hipMalloc(&t->d_s2, S2.size() * sizeof(u32)));
            ACHK(hipMemcpy(t->d_s2, S2.data(), S2.size() * sizeof(u32), hipMemcpyHostToDevice));
        }
        if (!S3.empty())
        {
            ACHK(hipMalloc(&t->d_s3, S3.size() * sizeof(u32)));
            ACHK(hipMemcpy(t->d_s3, S3.data(), S3.size() * sizeof(u32), hipMemcpyHostToDevice));
        }
    }
    ACHK(hipMalloc(&t->d_edges, tab.size() * sizeof(uint2)));
    ACHK(hipMemcpy(t->d_edges, tab.data(), tab.size() * sizeof(uint2), hipMemcpyHostToDevice));
    ACHK(hipMalloc(&t->d_copies, copies.size() * sizeof(u32)));
    ACHK(hipMemcpy(t->d_copies, copies.data(), copies.size() * sizeof(u32), hipMemcpyHostToDevice));
    return t;
bad:
    ac_free(t);
    return nullptr;
}

void ac_free(AcTables *t)
{
    if (!t)
        return;
    (void)hipSetDevice(t->device);
    if (t->d_redo) (void)hipFree(t->d_redo);
    if (t->d_s1) (void)hipFree(t->d_s1);
    if (t->d_s2) (void)hipFree(t->d_s2);
    if (t->d_s3) (void)hipFree(t->d_s3);
    if (t->d_filterx20) (void)hipFree(t->d_filterx20);
    if (t->d_filterx19) (void)hipFree(t->d_filterx19);
    if (t->d_filters20) (void)hipFree(t->d_filters20);
    if (t->d_filters19) (void)hipFree(t->d_filters19);
    if (t->d_edges) (void)hipFree(t->d_edges);
    if (t->d_copies) (void)hipFree(t->d_copies);
    if (t->d_gram4) (void)hipFree(t->d_gram4);
    if (t->d_g4x) (void)hipFree(t->d_g4x);
    ac_anchor_free(t);
    delete t;
}

} // namespace kg
