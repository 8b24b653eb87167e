"""Model test of the fused kernels' three-level prefix (lb_exclusive, hrt_kernels.hip): the status
words of the chunks / groups / supergroups are delivered in adversarial orders and every chunk's
exclusive prefix must still come out right.  The model transcribes the kernel's loop statement by
statement (one `poll` = one iteration of its for(;;)); `fixed=False` is round 3's condition for
forming sum_c, which published a supergroup total without the last group's chunk counts when
super[sg-1] arrived late -- the test shows the model sees that bug."""
import random

import pytest

DONE = 1 << 31


class Words:
    def __init__(self, nchunks):
        self.chunk = [0] * (nchunks + 64)
        self.group = [0] * (nchunks // 64 + 64)
        self.sup = [0] * 128


class Chunk:
    """the state one workgroup's polling wave carries across iterations"""

    def __init__(self, W, chunk, c, fixed=True):
        self.W, self.chunk, self.c, self.fixed = W, chunk, c, fixed
        self.g_owed = (chunk & 63) == 63
        self.s_owed = (chunk & 4095) == 4095
        self.result = None
        W.chunk[chunk] = DONE | c     # published before the prefix is asked for

    def poll(self):
        W, chunk, c = self.W, self.chunk, self.c
        g, sg = chunk >> 6, chunk >> 12
        cw = [W.chunk[(g << 6) + l] if (g << 6) + l < chunk else DONE for l in range(64)]
        gw = [W.group[(sg << 6) + l] if (sg << 6) + l < g else DONE for l in range(64)]
        s0 = [W.sup[l] if l < sg else DONE for l in range(64)]
        s1 = [W.sup[l + 64] if l + 64 < sg else DONE for l in range(64)]
        ok_c = all(x & DONE for x in cw)
        ok_g = all(x & DONE for x in gw)
        ok_s = all((a & b) & DONE for a, b in zip(s0, s1))
        sum_c = sum_g = 0
        need = (self.g_owed or (ok_g and (self.s_owed or ok_s))) if self.fixed else (self.g_owed or (ok_g and ok_s))
        if ok_c and need:
            sum_c = sum(x & ~DONE for x in cw)
        if ok_c and self.g_owed:
            W.group[g] = DONE | (sum_c + c)
            self.g_owed = False
        if ok_c and ok_g and (self.s_owed or ok_s):
            sum_g = sum(x & ~DONE for x in gw)
        if ok_c and ok_g and self.s_owed:
            W.sup[sg] = DONE | (sum_g + sum_c + c)
            self.s_owed = False
        if ok_c and ok_g and ok_s:
            self.result = sum_c + sum_g + sum((a & ~DONE) + (b & ~DONE) for a, b in zip(s0, s1))
        return self.result is not None


def run(nchunks, order, fixed, seed):
    rnd = random.Random(seed)
    counts = [rnd.randrange(0, 257) for _ in range(nchunks)]
    W = Words(nchunks)
    live, got = [], {}
    # chunks become resident in index order (the kernel's progress assumption); `order` picks who polls
    # in a round, in which order (a subset starves the others for that round)
    nxt, rounds = 0, 0
    while len(got) < nchunks:
        while nxt < nchunks and len(live) < 4200:
            live.append(Chunk(W, nxt, counts[nxt], fixed))
            nxt += 1
        for ch in order(list(live), rnd, rounds):
            if ch.poll():
                got[ch.chunk] = ch.result
                live.remove(ch)
        rounds += 1
        assert rounds < 100000
    pre, acc = [], 0
    for c in counts:
        pre.append(acc)
        acc += c
    return [got[i] for i in range(nchunks)], pre


def late_first(live, rnd, r):      # the highest resident chunk polls first: closers run before their prefix exists
    return sorted(live, key=lambda ch: -ch.chunk)


def closers_first(live, rnd, r):   # group / supergroup closers first, then high to low
    return sorted(live, key=lambda ch: (-((ch.chunk & 4095) == 4095), -((ch.chunk & 63) == 63), -ch.chunk))


def shuffled(live, rnd, r):
    rnd.shuffle(live)
    return live[: max(1, len(live) // 3)]


def starve_supergroup0(live, rnd, r):
    """ADVICE r03's interleaving: the chunk that closes group 127 AND supergroup 1 polls once before the
    group words of its supergroup exist, the group closers then publish them, it polls again -- all while
    chunk 4095 (which owes super[0]) is starved"""
    closer = [ch for ch in live if ch.chunk == 8191]
    if not closer:   # (8191 not resident yet: everyone but 4095 polls)
        return sorted((ch for ch in live if ch.chunk != 4095), key=lambda ch: ch.chunk)
    if not closer[0].s_owed:
        return sorted(live, key=lambda ch: ch.chunk)
    if closer[0].g_owed:
        return closer
    groups = [ch for ch in live if (ch.chunk & 63) == 63 and ch.chunk > 4095 and ch.g_owed]
    return groups if groups else closer


@pytest.mark.parametrize("order", [late_first, closers_first, shuffled, starve_supergroup0])
def test_prefix_is_exact_in_any_delivery_order(order):
    got, want = run(2 * 4096 + 200, order, True, 1)
    assert got == want


def test_model_sees_the_round3_bug():
    """the old condition stores super[1] without the chunk counts of its last group"""
    got, want = run(2 * 4096 + 200, starve_supergroup0, False, 1)
    assert got != want
