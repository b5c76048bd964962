"""A discrete-event MODEL of the loop kernel's inter-workgroup exchange (csrc/wrnn_loop.hip: tag-free slots pre-filled with a
sentinel, a ring of 4 slots by step, each producer re-arming its own words of slot (t+3) % 4 at its last stage of step t after
draining its stores) -- the delicate part of the design, checked here under adversarial timing instead of by argument alone:

* stores become visible after a random delay and NOT in program order (write-through acknowledgements race); only a drain
  (`s_waitcnt vmcnt(0)`) makes a workgroup wait until all its earlier stores are visible;
* every stage of every workgroup takes a random time, so roles and workgroups drift as far apart as the data dependencies allow.

Checked: a consumer whose poll finds no sentinel reads the data of ITS step in every word (never a stale ring turn), a re-arm
never lands on top of data that is still to be read or newer than the re-arm (which would dead-lock the consumers: also
detected as "no progress"), and all workgroups finish.  The model follows the kernel's phase order per role (DESIGN.md
section 3); it is a model of the protocol, not of the HIP code -- the GPU tests cover that.  `late_publish=True` models the
drafted fused stages (drafts/README.md): role B's last publication of a step leaves after its ring-hygiene point."""
import heapq
import random

import pytest

RING = 4
SENT = None


class Sim:
    def __init__(self, seed, n_wg=3, steps=40, late_publish=False, no_drain=False, rearm_ahead=3):
        self.rng = random.Random(seed)
        self.n_wg, self.steps, self.late, self.no_drain, self.rearm_ahead = n_wg, steps, late_publish, no_drain, rearm_ahead
        # mem[layer][slot][producer wg] = tag (step) or SENT; layers published by role A: h1 x1 y1, by role B: h2 x2 y2 xt
        self.mem = {l: [[SENT] * n_wg for _ in range(RING)] for l in ('h1', 'x1', 'y1', 'h2', 'x2', 'y2', 'xt')}
        self.now = 0.0
        self.events = []            # (time, seq, fn)
        self.seq = 0
        self.pending = {}           # wg id -> number of stores not yet visible
        self.violations = []
        self.done = 0

    def at(self, dt, fn):
        self.seq += 1
        heapq.heappush(self.events, (self.now + dt, self.seq, fn))

    # ---- memory ------------------------------------------------------------------------------------------------------
    def store(self, who, layer, slot, j, value):
        self.pending[who] = self.pending.get(who, 0) + 1

        def land():
            old = self.mem[layer][slot][j]
            if value is SENT and old is not SENT and old >= self.rearm_step.get((who, layer, slot), -1):
                # a re-arm must only ever replace the data of the ring turn it was issued for (or older)
                self.violations.append(f're-arm of {layer}[{slot}][{j}] landed on data of step {old}')
            self.mem[layer][slot][j] = value
            self.pending[who] -= 1
        # (a rare very late acknowledgement: longer than three whole steps, which only the drain makes harmless)
        self.at(self.rng.choice([0.1, 0.5, 1.0, 3.0, 8.0, 8.0, 60.0]), land)

    # ---- a workgroup = a generator of (kind, payload) actions --------------------------------------------------------
    def run(self):
        self.rearm_step = {}
        procs = []
        for role in 'AB':
            for j in range(self.n_wg):
                procs.append(self.program(role, j))
        for p in procs:
            self.resume(p)
        idle_limit = 200000
        n = 0
        while self.events and n < idle_limit:
            self.now, _, fn = heapq.heappop(self.events)
            fn()
            n += 1
        if self.done != len(procs):
            self.violations.append(f'no progress: {self.done} of {len(procs)} workgroups finished')
        return self.violations

    def resume(self, p):
        try:
            kind, arg = next(p)
        except StopIteration:
            self.done += 1
            return
        if kind == 'work':                                   # a stage's compute: random duration
            self.at(self.rng.choice([0.2, 1.0, 2.0, 6.0]) * arg, lambda: self.resume(p))
        elif kind == 'poll':                                 # re-read until no word is the sentinel
            layer, slot, t = arg

            def poll():
                words = self.mem[layer][slot]
                if any(wd is SENT for wd in words):
                    self.at(0.3, poll)
                    return
                for j, wd in enumerate(words):
                    if wd != t:
                        self.violations.append(f'{layer}[{slot}][{j}] read as step {wd} while polling for step {t}')
                self.resume(p)
            poll()
        elif kind == 'drain':                                # s_waitcnt vmcnt(0)
            who = arg

            def drain():
                if self.pending.get(who, 0) > 0 and not self.no_drain:
                    self.at(0.2, drain)
                else:
                    self.resume(p)
            drain()

    def program(self, role, j):
        who = (role, j)
        mine = ('h1', 'x1', 'y1') if role == 'A' else ('h2', 'x2', 'y2', 'xt')

        def publish(layer, t):
            self.store(who, layer, t % RING, j, t)

        def hygiene(t):
            slot = (t + self.rearm_ahead) % RING
            for layer in mine:
                self.rearm_step[(who, layer, slot)] = t + self.rearm_ahead - RING + 1      # first step whose data must survive it
                self.store(who, layer, slot, j, SENT)

        for t in range(self.steps):
            s = t % RING
            if role == 'A':
                if t > 0:
                    yield ('poll', ('xt', (t - 1) % RING, t - 1))      # x_{t-1} (here always handed over by role B: the harder case)
                yield ('work', 1.0)
                publish('h1', t); publish('x1', t)
                yield ('poll', ('h1', s, t)); yield ('work', 1.0)      # gh1(t+1)
                yield ('poll', ('x2', s, t)); yield ('work', 0.4)
                publish('y1', t)
                yield ('poll', ('y2', s, t))
                yield ('drain', who); hygiene(t)                       # role A's last stage of the step
                yield ('work', 0.4)                                    # fc3 (the sampling itself is role B's in this model)
            else:
                yield ('poll', ('x1', s, t)); yield ('work', 1.0)
                publish('h2', t); publish('x2', t)
                yield ('poll', ('h2', s, t)); yield ('work', 1.0)      # gh2(t+1)
                yield ('poll', ('y1', s, t))
                if self.late:                                          # fused draft: hygiene first, y2 of this step leaves after it
                    yield ('drain', who); hygiene(t)
                    yield ('work', 0.4); publish('y2', t)
                else:
                    yield ('work', 0.4); publish('y2', t)
                    yield ('drain', who); hygiene(t)
                yield ('poll', ('y2', s, t)); yield ('work', 0.4)      # fc3 + sampling on role B
                publish('xt', t)


@pytest.mark.parametrize('late_publish', [False, True])
def test_exchange_ring_is_safe_under_adversarial_timing(late_publish):
    for seed in range(60):
        v = Sim(seed, n_wg=3, steps=40, late_publish=late_publish).run()
        assert not v, (seed, v[:3])


def test_model_detects_a_broken_protocol():
    """The model is not vacuous: re-arming the slot of the NEXT step (instead of the one three steps ahead), or dropping the
    drain, is caught for some timing."""
    assert any(Sim(seed, rearm_ahead=1).run() for seed in range(20))
    assert any(Sim(seed, no_drain=True, steps=80).run() for seed in range(60))
