"""Device-side post-processing for the reference's formatter: the (start, end) order that
search_file() establishes with qsort (krep.c:3018-3023, comparator :420-434) and the line numbers that
print_matching_items() derives by counting newlines (krep.c:589-668)."""
import numpy as np
import pytest

import cases
from krep_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import krep_amd
    e = krep_amd.load()
    assert e.device_count() >= 1
    return e


def test_order_by_start_and_line_numbers_on_the_device(gpu):
    import torch
    import bench
    pats = bench.ac_patterns()
    n = (48 << 20) + 123
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    gpu.generate(buf.data_ptr(), n, 0, 4, bench.SEED, bench.pack_dict(pats), 4096)
    cap = n // 500
    pos = torch.empty(2 * cap, dtype=torch.int64, device="cuda")
    out = gpu.plan(abi.Params(pats)).scan(buf.data_ptr(), n, 0, n, 0, pos.data_ptr(), cap)
    assert not out.overflow and out.stored > 10000
    m = out.stored
    emitted = pos[: 2 * m].view(-1, 2).cpu().numpy().astype(np.uint64)
    gpu.order_by_start(pos.data_ptr(), m, n)
    got = pos[: 2 * m].view(-1, 2).cpu().numpy().astype(np.uint64)
    want = emitted[np.lexsort((emitted[:, 1], emitted[:, 0]))]          # qsort by (start, end)
    assert np.array_equal(got, want)
    lines = torch.empty(m, dtype=torch.int64, device="cuda")
    gpu.line_numbers(buf.data_ptr(), n, pos.data_ptr(), m, lines.data_ptr())
    text = buf[:n].cpu().numpy()
    nl = np.flatnonzero(text == 10)
    want_lines = 1 + np.searchsorted(nl, got[:, 0].astype(np.int64), side="left")   # newlines strictly before start
    assert np.array_equal(lines.cpu().numpy(), want_lines)


def test_line_numbers_edge_cases(gpu):
    import torch
    text = np.frombuffer(b"\n\nab\nabab\n" + b"x" * 5000 + b"\nab", dtype=np.uint8).copy()
    buf = torch.from_numpy(text).cuda()
    ret, pos = gpu.search(abi.Params([b"ab"]), text)
    d_pos = torch.from_numpy(pos.astype(np.int64)).cuda().contiguous()
    lines = torch.empty(len(pos), dtype=torch.int64, device="cuda")
    gpu.line_numbers(buf.data_ptr(), text.size, d_pos.data_ptr(), len(pos), lines.data_ptr())
    want = [1 + int((text[: int(s)] == 10).sum()) for s in pos[:, 0]]
    assert lines.cpu().tolist() == want == [3, 4, 4, 6]


def test_host_operator_hands_back_sorted_records(gpu, oracle_engine):
    rng = np.random.RandomState(77)
    alpha = b"abcd \n"
    text = cases.rand_text(rng, 300_000, alpha)
    pats = [cases.pick_pattern(rng, text, m, alpha) for m in (2, 3, 5, 5, 9, 14, 30)]
    want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)[1]
    want = want[np.lexsort((want[:, 1], want[:, 0]))]
    try:
        gpu.set_result_order(True)
        ret, got = gpu.search(abi.Params(pats), text)
    finally:
        gpu.set_result_order(False)
    assert ret == len(want) and np.array_equal(got, want)


def test_search_buffer_shards_hand_back_sorted_records(gpu, oracle_engine):
    rng = np.random.RandomState(78)
    alpha = b"abc \n"
    text = cases.rand_text(rng, 200_000, alpha)
    pats = [cases.pick_pattern(rng, text, m, alpha) for m in (2, 4, 6, 11)]
    want = oracle_engine.call(abi.RA_AHO_CORASICK, abi.Params(pats), text)[1]
    want = want[np.lexsort((want[:, 1], want[:, 0]))]
    try:
        gpu.set_result_order(True)
        for shards in (1, 3):
            rc, cnt, got = gpu.search_buffer(abi.Params(pats), text, num_gpus=shards)
            assert rc == 0 and np.array_equal(got, want), shards
    finally:
        gpu.set_result_order(False)
