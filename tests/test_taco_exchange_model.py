"""A discrete-event MODEL of wrnn_taco_resident_kernel's exchange (csrc/wrnn_taco.hip): every layer's output vector lives in TWO
buffers by step parity, an entry is an 8-byte word {value, tag = step + 1} written by ONE store, and a consumer workgroup polls the
entries it stages until every tag is the one it expects -- no flag barrier, no re-arm.  The kernel header argues that a buffer
written at step t is overwritten at step t + 2 only after every reader of step t (and of step t + 1's cross-step reads) is done,
BECAUSE every wave owns a unit of both LSTM layers (512 units = 512 waves): the chain of step t + 1 cannot pass those layers
before every workgroup has.  Checked here under adversarial timing instead of by argument alone:

* a store becomes visible after a random delay, out of program order;
* every layer of every workgroup takes a random time -- now and then a stall longer than two whole steps -- so workgroups drift as
  far apart as the data dependencies allow.

Checked: a poll that succeeds has read the value of ITS step in every entry (values are (layer, step, row)); all workgroups
finish (an entry overwritten before a late reader saw it would leave that reader spinning for a tag that never comes back:
"no progress").  The negative controls show the model can see what it is there for: ONE buffer per vector instead of two, or
a workgroup that owns no row of the LSTM layers (the kernel run with fewer than 512 units' worth of waves), dead-lock or read
the wrong step.  A model of the protocol, not of the HIP code -- tests/test_gpu_config3.py covers that."""
import heapq
import random

import pytest

# layers of a decoder step in program order: (name, producers) -- 'all' = every workgroup owns rows, 'some' = only the first
# workgroups do (prenet, GRU, query, context, mel rows: fewer rows than waves)
LAYERS = [('pre1', 'some'), ('pre2', 'some'), ('attn_h', 'some'), ('pq', 'some'), ('s', 'some'), ('ctx', 'some'), ('x', 'all'),
          ('h1x2', 'all'), ('h2x3', 'all'), ('mel', 'some')]
# what a layer stages: (vector, 0 = this step / 1 = the previous step's)
READS = {'pre1': [('mel', 1)], 'pre2': [('pre1', 0)], 'attn_h': [('ctx', 1), ('attn_h', 1), ('pre2', 0)], 'pq': [('attn_h', 0)],
         's': [('pq', 0)], 'ctx': [('s', 0)], 'x': [('ctx', 0)], 'h1x2': [('x', 0), ('h1x2', 1)], 'h2x3': [('h1x2', 0), ('h2x3', 1)],
         'mel': [('h2x3', 0)]}


class Sim:
    def __init__(self, seed, n_wg=4, steps=30, buffers=2, lstm_everywhere=True):
        self.rng = random.Random(seed)
        self.n_wg, self.steps, self.nbuf, self.lstm_everywhere = n_wg, steps, buffers, lstm_everywhere
        self.rows = {}
        for name, who in LAYERS:
            if who == 'all' and lstm_everywhere:
                self.rows[name] = list(range(n_wg))                      # one row per workgroup
            else:
                self.rows[name] = list(range(max(1, n_wg // 2)))         # only the first workgroups own rows
        # mem[vector][buffer][row] = (tag, value)
        self.mem = {name: [[(0, None)] * len(self.rows[name]) for _ in range(buffers)] for name, _ in LAYERS}
        self.now, self.events, self.seq = 0.0, [], 0
        self.violations, self.done = [], 0

    def at(self, dt, fn):
        self.seq += 1
        heapq.heappush(self.events, (self.now + dt, self.seq, fn))

    def store(self, vec, buf, row, tag, value):
        def land():
            self.mem[vec][buf][row] = (tag, value)
        self.at(self.rng.choice([0.1, 0.3, 1.0, 2.0, 6.0]), land)

    def program(self, wg):
        for step in range(self.steps):
            for name, _ in LAYERS:
                for vec, back in READS[name]:
                    src = step - back
                    if src < 0:
                        continue                                          # step 0: zeros, nothing to poll
                    yield ('poll', vec, src % self.nbuf, src + 1, src)
                # (a rare long stall -- a pre-empted or throttled CU: longer than two whole steps of the others)
                yield ('work', self.rng.choice([0.2, 0.5, 1.0, 4.0]) if self.rng.random() > 0.004 else 150.0)
                if wg in self.rows[name]:
                    yield ('store', name, step % self.nbuf, self.rows[name].index(wg), step + 1, (name, step, wg))
        self.done += 1

    def run(self):
        procs = [self.program(wg) for wg in range(self.n_wg)]
        blocked = {}

        def advance(i):
            for act in procs[i]:
                if act[0] == 'work':
                    self.at(act[1], lambda i=i: advance(i))
                    return
                if act[0] == 'store':
                    self.store(*act[1:])
                    continue
                blocked[i] = act
                self.at(0.0, lambda i=i: poll(i))
                return

        def poll(i):
            _, vec, buf, tag, src = blocked[i]
            entries = self.mem[vec][buf]
            if all(t == tag for t, _ in entries):
                for row, (_, val) in enumerate(entries):
                    if val != (vec, src, self.rows[vec][row]):
                        self.violations.append(f'workgroup {i} staged {val} for {vec} of step {src}')
                del blocked[i]
                advance(i)
            else:
                self.at(0.4, lambda i=i: poll(i))

        for i in range(self.n_wg):
            self.at(self.rng.random(), lambda i=i: advance(i))
        guard = 0
        while self.events and guard < 2_000_000:
            guard += 1
            self.now, _, fn = heapq.heappop(self.events)
            if self.now > 40.0 * self.steps * len(LAYERS) + 2000.0:
                break                                                     # no progress: pollers spinning for a tag that is gone
            fn()
        return self.done == self.n_wg and not self.violations


@pytest.mark.parametrize('seed', range(12))
def test_tagged_parity_buffers_deliver_the_right_step_and_finish(seed):
    sim = Sim(seed)
    assert sim.run(), (sim.done, sim.violations[:3])


def test_the_model_sees_a_single_buffer_fail():
    """ONE buffer per vector: a fast producer of step t + 1 overwrites entries a slow consumer of step t has not staged yet."""
    bad = 0
    for seed in range(12):
        sim = Sim(seed, buffers=1)
        bad += not sim.run()
    assert bad >= 6, bad


def test_the_model_sees_workgroups_without_lstm_rows_fail():
    """The kernel's reason for exactly 128 workgroups: a workgroup that owns no LSTM unit is not held back by the chain and its
    late reads meet entries of two steps later."""
    bad = 0
    for seed in range(24):
        sim = Sim(seed, n_wg=6, steps=40, lstm_everywhere=False)
        bad += not sim.run()
    assert bad >= 1, bad
