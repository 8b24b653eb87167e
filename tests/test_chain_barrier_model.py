"""Model test of the chain kernel's grid barrier (chain_barrier, hrt_kernels.hip): flags instead of atomics.
Workgroup j stores a tag into word j of the previous launch's (dead) status words; the first workgroup of every
64 waits for its group's words and stores the group's word; everybody waits for the group words.  The model
transcribes the kernel's steps (one `step` = one store or one poll of its loops) and a scheduler interleaves
the workgroups adversarially.  What must hold: nobody leaves a barrier before everybody has arrived at it;
the leftovers of the words' earlier lives -- status words (bit 31 set), zeros, the tags of the roll call and of
earlier barriers on the same words -- never count as an arrival; a grid that is not complete (a workgroup that
never gets a slot: shared GPU) ends in a timeout for everybody, not in a hang."""
import random

import pytest

DONE = 1 << 31
TAG = 1 << 30


class WG:
    """one workgroup's wave 0 at a barrier; `poll_budget` = hrt_ktune.lb_max_polls"""

    def __init__(self, j, G, chunk, group, tag, err, poll_budget):
        self.j, self.G, self.chunk, self.group, self.tag, self.err = j, G, chunk, group, tag, err
        self.left, self.polls, self.state, self.ok = False, 0, "store", None
        self.budget = poll_budget

    def _give_up(self):
        self.err[0] |= 0x200
        self.ok, self.left = False, True

    def step(self):
        j, G, tag = self.j, self.G, self.tag
        if self.state == "store":
            self.chunk[j] = tag
            self.state = "collect" if j % 64 == 0 else "wait"
            return
        if self.state == "collect":   # the group's first workgroup: all words of its group
            if all(self.chunk[j + l] == tag for l in range(64) if j + l < G):
                self.group[j >> 6] = tag
                self.state = "wait"
                return
        elif self.state == "wait":
            if all(self.group[l] == tag for l in range((G + 63) >> 6)):
                self.ok, self.left = True, True
                return
        if self.err[0] & 0x300:
            self.ok, self.left = False, True
            return
        self.polls += 1
        if self.polls > self.budget:
            self._give_up()


def barrier(G, chunk, group, tag, rnd, absent=(), budget=10 ** 9, order="random"):
    err = [0]
    wgs = [WG(j, G, chunk, group, tag, err, budget) for j in range(G) if j not in absent]
    arrived = set()
    live = list(wgs)
    while live:
        if order == "leaders_first":      # the collectors poll long before the others have stored
            w = min(live, key=lambda w: (w.j % 64 != 0, rnd.random()))
            if w.j % 64 == 0 and w.state != "store" and rnd.random() < 0.5:
                w = rnd.choice(live)
        elif order == "descending":       # the last workgroup arrives first
            w = max(live, key=lambda w: (w.state == "store", w.j)) if rnd.random() < 0.7 else rnd.choice(live)
        else:
            w = rnd.choice(live)
        before = w.state
        w.step()
        if before == "store":
            arrived.add(w.j)
        if w.left:
            if w.ok:   # the property: a workgroup passes only when EVERY workgroup of the grid has arrived
                assert len(arrived) == G and not absent, (w.j, len(arrived), G)
            live.remove(w)
    return [w.ok for w in wgs], err[0]


@pytest.mark.parametrize("G", [1, 2, 63, 64, 65, 200, 1024])
@pytest.mark.parametrize("order", ["random", "leaders_first", "descending"])
def test_nobody_passes_before_everybody_arrived(G, order):
    rnd = random.Random(G * 7 + len(order))
    # the words' earlier life: status words of a finished launch, zeros, and an older tag
    chunk = [rnd.choice([0, DONE | rnd.randrange(1025), TAG | 0xFFFF]) for _ in range(G + 64)]
    group = [rnd.choice([0, DONE | rnd.randrange(70000), TAG | 0xFFFF]) for _ in range(G // 64 + 64)]
    ok, err = barrier(G, chunk, group, TAG | 1, rnd, order=order)
    assert all(ok) and err == 0


def test_roll_call_and_later_barriers_share_the_words():
    """roll call (tag | 0xffff) and the barrier behind the first bounce use the SAME words (launch b0 - 1's);
    the barrier behind bounce b uses launch b - 1's, where the roll call's and older tags may still lie"""
    rnd = random.Random(5)
    G = 300
    chunk = [DONE | rnd.randrange(1025) for _ in range(G + 64)]
    group = [DONE | rnd.randrange(70000) for _ in range(G // 64 + 64)]
    for tag in (TAG | 0xFFFF, TAG | 1, TAG | 2, TAG | 3):
        ok, err = barrier(G, chunk, group, tag, rnd, order="leaders_first")
        assert all(ok) and err == 0


@pytest.mark.parametrize("G,absent", [(64, {63}), (130, {0}), (130, {129}), (700, {511, 699})])
def test_an_incomplete_grid_times_out_for_everybody(G, absent):
    """a workgroup that never gets a slot (the GPU is shared): every resident workgroup leaves with a timeout
    (HRT_ERR_CHAIN_TIMEOUT in the error word), nobody passes, nobody spins for ever"""
    rnd = random.Random(len(absent) + G)
    chunk, group = [0] * (G + 64), [0] * (G // 64 + 64)
    ok, err = barrier(G, chunk, group, TAG | 0xFFFF, rnd, absent=absent, budget=256)
    assert not any(ok) and (err & 0x200)
