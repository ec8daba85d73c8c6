"""A MODEL of the ticket -> resolver -> deferred-store protocol of the one-pass record writers (krep_amd/csrc/kg_tickets.h,
kg_single.hip, kg_ac_tiny.hip FUSED), exhaustively scheduled at random on the CPU: it checks the DESIGN of the protocol — no
schedule and no residency pattern may leave a wave waiting for ever, every ticket is scanned once, every prefix is the sum of the
counts in front of it — not the HIP code (the GPU suite does that: the starved-grid tests of tests/test_gpu_literal.py and
tests/test_gpu_ac_tiny.py).  What is modelled, step by step as the kernels do it:
  * ONE ticket counter; a wave draws one ticket (per-wave mode) or thread 0 of a workgroup draws four consecutive ones behind a
    barrier (per-workgroup mode, kg_single.hip BDRAW); the resolver's own workgroup always draws per wave;
  * the resolver is the first wave 0 to claim the role; it publishes prefixes as far as the run of published counts extends;
  * a wave scans its ticket, publishes the count, draws the NEXT ticket, and only then waits for the prefix of the ticket it
    parked before (the flush) — it waits while holding a drawn, unscanned ticket;
  * only R workgroups are resident at a time (a starved or shared device): a workgroup becomes resident when another one has
    ended, and a resident workgroup is never preempted."""
import random

import pytest

WAVES = 4


class Wave:
    def __init__(self, blk, w):
        self.blk, self.w = blk, w
        self.state = "start"
        self.t = None          # ticket being scanned / about to be scanned
        self.tn = None         # ticket drawn next
        self.pend = None       # ticket whose records wait in the ring
        self.left = 0          # scan steps left
        self.resolver = False
        self.ndraw = 0         # per-workgroup draws this wave has taken part in
        self.arrived = 0       # ... and the one it has arrived at the barrier of


def run(n_tickets, n_blocks, resident, per_workgroup, seed, scan_steps=3):
    rng = random.Random(seed)
    counter = 0
    claimed = False
    agg = [None] * n_tickets      # published counts
    pref = [None] * n_tickets     # published prefixes
    counts = [rng.randrange(0, 9) for _ in range(n_tickets)]
    scanned = [0] * n_tickets
    flushed = {}
    res_base, res_run = 0, 0
    blocks = [[Wave(b, w) for w in range(WAVES)] for b in range(n_blocks)]
    blk_mode = [None] * n_blocks  # per block: True = draws as one
    blk_base = [dict() for _ in range(n_blocks)]  # per block: its k-th draw's base ticket
    waiting = list(range(n_blocks))
    res = []                      # resident blocks

    def done(b):
        return all(x.state == "done" for x in blocks[b])

    def draw_one():
        nonlocal counter
        t = counter
        counter += 1
        return t if t < n_tickets else None

    def step(x):
        """advance wave x by one step if it can; returns True on progress"""
        nonlocal claimed, counter, res_base, res_run
        b = x.blk
        if x.state == "start":
            if x.w == 0 and not claimed:
                claimed = True
                x.resolver = True
            if per_workgroup:
                # the block learns whether it holds the resolver behind a barrier; modelled as: decided when wave 0 has started
                if x.w != 0 and blocks[b][0].state == "start":
                    return False
                if blk_mode[b] is None:
                    blk_mode[b] = not blocks[b][0].resolver
            else:
                blk_mode[b] = False
            x.state = "resolve" if x.resolver else "draw_first"
            return True
        if x.state == "resolve":
            moved = False
            while res_base < n_tickets and agg[res_base] is not None:
                pref[res_base] = res_run
                res_run += agg[res_base]
                res_base += 1
                moved = True
            if res_base >= n_tickets:
                x.state = "done"
                return True
            return moved
        if x.state in ("draw_first", "draw_next"):
            if blk_mode[b]:
                # barrier A + draw by thread 0 + barrier B: the wave's k-th draw happens when every live wave of the workgroup has
                # arrived at ITS k-th draw (a wave that comes back early waits for the stragglers of the round before)
                k = x.ndraw + 1
                if x.arrived != k:
                    x.arrived = k
                    return True
                if any(y.arrived < k for y in blocks[b] if y.state != "done"):
                    return False
                if k not in blk_base[b]:
                    blk_base[b][k] = counter
                    counter += WAVES
                base = blk_base[b][k]
                t = base + x.w
                drawn = t if t < n_tickets else None
                blk_live = base < n_tickets
                x.ndraw = k
            else:
                drawn = draw_one()
                blk_live = drawn is not None
            if x.state == "draw_first":
                x.t = drawn
                x.alive = blk_live
                x.left = rng.randrange(1, scan_steps + 3)
                x.state = "scan" if blk_live else "final"
            else:
                x.tn = drawn
                x.tn_alive = blk_live
                x.state = "flush"
            return True
        if x.state == "scan":
            if x.t is None:       # (per-workgroup draws: this wave got no ticket from the block's last draw)
                x.state = "draw_next"
                return True
            x.left -= 1
            if x.left > 0:
                return True
            scanned[x.t] += 1
            agg[x.t] = counts[x.t]  # publish BEFORE waiting for anything
            x.state = "draw_next"
            return True
        if x.state == "flush":
            if x.pend is not None:
                if pref[x.pend] is None:
                    return False    # waits for the resolver — while holding tn
                flushed[x.pend] = pref[x.pend]
            x.pend = x.t
            x.t = x.tn
            x.left = rng.randrange(1, scan_steps + 3)
            x.state = "scan" if x.tn_alive else "final"
            return True
        if x.state == "final":
            if x.pend is not None:
                if pref[x.pend] is None:
                    return False
                flushed[x.pend] = pref[x.pend]
                x.pend = None
            x.state = "done"
            return True
        return False

    for _ in range(200000):
        res = [b for b in res if not done(b)]
        while len(res) < resident and waiting:
            res.append(waiting.pop(0))
        if not res:
            break
        cand = [x for b in res for x in blocks[b] if x.state != "done"]
        rng.shuffle(cand)
        if not any(step(x) for x in cand):
            raise AssertionError(f"no wave can move: tickets={n_tickets} blocks={n_blocks} resident={resident} "
                                 f"per_workgroup={per_workgroup} seed={seed} states={[(x.blk, x.w, x.state, x.t, x.pend) for x in cand]}")
    else:
        raise AssertionError("did not finish")
    assert scanned == [1] * n_tickets
    run_sum = 0
    for t in range(n_tickets):
        assert pref[t] == run_sum, t
        run_sum += counts[t]
        assert flushed.get(t) == pref[t], t


@pytest.mark.parametrize("per_workgroup", [False, True])
def test_no_schedule_leaves_a_wave_waiting(per_workgroup):
    n = 0
    for seed in range(600):
        rng = random.Random(10_000 + seed)
        n_tickets = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 33, 64, 130])
        n_blocks = rng.choice([1, 2, 3, 5, 9, 17])
        resident = rng.choice([1, 1, 2, 3, n_blocks])
        run(n_tickets, n_blocks, min(resident, n_blocks), per_workgroup, seed)
        n += 1
    assert n == 600
