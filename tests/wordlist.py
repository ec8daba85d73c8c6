"""A deterministic natural-language-LIKE word list and dictionaries drawn from it, for the word-text haystacks of generator
kind 5 (krep_amd/csrc/kg_synth.h).  TEST / BENCH SUPPORT: imported by tests/ and tools/ only.

Why: the reference's only published benchmark runs on a natural-language corpus (test/benchmark_krep_vs_rg.sh:4,
README.md:171-178) and SURVEY.md §8d cfg 1 describes "ASCII lines, words from a small list"; the i.i.d. letter texts of kinds
2-4 have no common suffixes, no frequent words and no repeated grams — exactly what a gram filter is sensitive to.  No corpus
is available offline, so the list is synthesised: syllables (onset + vowel + coda) and the usual English-like affixes
(-ing, -tion, -ment, -ed, -ly, -er, -ness, -able, un-, re-, ...), shorter words at the frequent ranks, the text generator
draws them with p(rank) ~ 1 / rank.
"""
from __future__ import annotations

import random
import struct

ONSETS = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "st", "tr", "ch", "sh",
          "th", "pr", "br", "cl", "gr", "pl", "sp", "fr", "wh", "qu"]
VOWELS = ["a", "e", "i", "o", "u", "ea", "ou", "ai", "ee", "oo", "io", "ie"]
CODAS = ["", "", "", "n", "r", "s", "t", "l", "m", "d", "nd", "st", "ng", "ck", "ll", "rt", "nt", "ss"]
SUFFIXES = ["", "", "", "", "s", "ed", "ing", "ly", "er", "tion", "ment", "ness", "able", "ers", "ings", "ation", "ity", "ous",
            "ive", "al", "ful", "less", "est"]
PREFIXES = ["", "", "", "", "", "", "un", "re", "in", "dis", "pre", "over", "con", "de"]
FUNCTION_WORDS = ["the", "of", "and", "a", "to", "in", "is", "that", "it", "was", "he", "for", "on", "as", "with", "his", "be",
                  "at", "by", "i", "this", "had", "not", "are", "but", "from", "or", "have", "an", "they", "which", "one"]


def word_list(n: int = 65536, seed: int = 20260930) -> list[bytes]:
    """n distinct lowercase words, rank order = list order (rank 1 first): function words, then synthesised words whose
    syllable count grows with the rank (frequent words are short)."""
    rng = random.Random(seed)
    seen, out = set(), []
    for w in FUNCTION_WORDS:
        if w not in seen and len(out) < n:
            seen.add(w)
            out.append(w)
    while len(out) < n:
        r = len(out)
        nsyl = 1 if r < 200 else rng.choice([1, 2]) if r < 2000 else rng.choice([1, 2, 2, 2, 3]) if r < 20000 else rng.choice([2, 2, 2, 3, 3])
        stem = "".join(rng.choice(ONSETS) + rng.choice(VOWELS) + rng.choice(CODAS) for _ in range(nsyl))
        w = (rng.choice(PREFIXES) if r >= 500 else "") + stem + (rng.choice(SUFFIXES) if r >= 100 else "")
        if 1 <= len(w) <= 16 and w not in seen:
            seen.add(w)
            out.append(w)
    return [w.encode() for w in out]


def pack(words) -> bytes:
    """The packed-dictionary blob the generator takes as `plant` ([u32 n][n x {u32 off, u32 len}][bytes], kg_synth.h DictView)."""
    head = struct.pack("<I", len(words))
    off, body = 4 + 8 * len(words), bytearray()
    for w in words:
        head += struct.pack("<II", off + len(body), len(w))
        body += w
    return head + bytes(body)


def dictionary(words, kind: str, n: int = 1000, seed: int = 7, min_len: int = 4, max_len: int = 16) -> list[bytes]:
    """n distinct patterns of min_len..max_len bytes from the list.  kind:
      'rare'    ranks in the upper half of the list only (each ~1e-6 of the word occurrences: the grep -f of unusual terms),
      'uniform' ranks drawn uniformly over the whole list beyond the 256 most frequent words,
      'common'  ranks drawn log-uniformly beyond the 64 most frequent words (many frequent words: a dense result list)."""
    rng = random.Random(seed)
    nw = len(words)
    picked, seen = [], set()
    guard = 0
    while len(picked) < n and guard < 100 * n:
        guard += 1
        if kind == "rare":
            r = rng.randrange(nw // 2, nw)
        elif kind == "uniform":
            r = rng.randrange(256, nw)
        elif kind == "common":
            lo, hi = 6, nw.bit_length() - 1
            o = rng.randrange(lo, hi)
            r = (1 << o) + rng.randrange(1 << o)
        else:
            raise ValueError(kind)
        w = words[r]
        if min_len <= len(w) <= max_len and w not in seen:
            seen.add(w)
            picked.append(w)
    return picked
