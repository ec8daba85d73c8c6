"""CPU side of the word-text workload (no GPU): the generator's host twin is slice-reproducible, its lines have the stated
shape, and the dictionaries drawn from the list do occur — pinned through the compiled reference / the oracle restatement."""
import collections

import numpy as np

import oracle_lib as ol
import wordlist
from krep_amd import abi

SEED, LINE = 20260930, 80


def _host(n, off, blob):
    import krep_amd
    return krep_amd.load().generate_host(n, off, 5, SEED, blob, LINE)


def test_word_text_shape_and_slices():
    w = wordlist.word_list()
    assert len(w) == 65536 and len(set(w)) == 65536 and all(1 <= len(x) <= 16 for x in w)
    blob = wordlist.pack(w)
    text = _host(3277 * LINE, 0, blob)  # whole lines: the last token is a whole word
    assert np.all(text[LINE - 1::LINE] == 10) and int((text == 10).sum()) == 3277
    for off, n in ((0, 500), (79, 3), (80, 80), (12345, 7777)):
        assert np.array_equal(_host(n, off, blob), text[off:off + n])
    toks = bytes(text).split()
    known = set(w)
    assert all(t in known for t in toks)
    top = collections.Counter(toks).most_common(1)[0]
    assert top[0] == b"the" and 0.04 < top[1] / len(toks) < 0.10  # rank 1 carries one octave = 1/16 of the draws
    assert 5.0 < sum(map(len, toks)) / len(toks) < 9.0


def test_reference_finds_dictionary_words_on_word_text():
    w = wordlist.word_list()
    blob = wordlist.pack(w)
    text = _host(1 << 19, 0, blob)
    o = ol.checker()
    for kind in ("rare", "uniform", "common"):
        pats = wordlist.dictionary(w, kind)
        assert len(pats) == 1000 and len(set(pats)) == 1000 and all(4 <= len(p) <= 16 for p in pats)
        n, pos = o.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)
        assert n == len(pos)
        raw = bytes(text)
        assert all(raw[s:e] in set(pats) for s, e in pos[:2000])
        if kind == "common":
            assert n > 1000
