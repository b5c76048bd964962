"""A discrete-event MODEL of wrnn_sparse_kernel's exchange (csrc/wrnn_sparse.hip, round 5): one workgroup per CU, 8 rnn1 + 8 rnn2 workgroups
per cluster (here: n + n), ONE group per cluster, every stage in one instruction stream per workgroup:

    rnn1 j:  drain | x_{t-1} (tagged) -> cell -> publish x1, h1 | poll h1(t) -> gh (registers) | poll x2(t) -> publish y1 |
             poll y1(t), RE-ARM own words of entry t + 2, publish y2 | poll cI(t + 1) -> W_ih . cI
    rnn2 j:  drain | poll x1(t) -> cell -> publish x2, h2 | poll x2(t) -> publish y1 | poll y1(t), RE-ARM, publish y2 |
             j = 0: poll y2(t) -> sample -> x_t as a tagged word in entry t % 2 | poll h2(t) -> gh | form cI(t + 2)

Sentinel layers (h1 x1 h2 x2 y1 y2): four ring entries, re-armed two steps ahead after the last poll of the step, drained at the top of the
next step.  cI: no sentinel inside a launch, cI(t + 2) formed by the rnn2 workgroups at the END of their step t, covered by the same drain, gathered by rnn1 at the end of step t + 1.  x_t: a
tagged word, two entries, never re-armed.  Checked under adversarial timing (the engine of tests/test_duo_exchange_model.py: stores land
after random delays, out of order, now and then later than ten whole steps -- only a drain waits for them): whatever a consumer accepts
carries ITS step in every word, no re-arm lands on data still to be read, no tagged word is overwritten before it was read, everybody
finishes.  The broken variants show the model is not vacuous.  A model of the protocol, not of the HIP code (tests/test_gpu_parity.py)."""
from test_duo_exchange_model import DuoSim, RING, SENT


class SparseSim(DuoSim):
    def __init__(self, seed, n=3, steps=24, ahead=2, drain=True, rearm_site='fc2', cond_lead=1):
        super().__init__(seed, n_wg=n, slots=1, steps=steps)
        self.n, self.ahead, self.drain, self.rearm_site, self.cond_lead = n, ahead, drain, rearm_site, cond_lead
        ring = lambda producers, entries: [[[SENT] * producers for _ in range(entries)]]          # [slot 0][entry][producer]
        self.mem = {l: ring(n, RING) for l in ('h1', 'x1', 'h2', 'x2', 'cI')}
        self.mem['y1'], self.mem['y2'] = ring(2 * n, RING), ring(2 * n, RING)
        self.mem['xt'] = ring(1, 2)

    def run(self):
        procs = [self.program(role, j) for role in ('A', 'B') for j in range(self.n)]
        for p in procs:
            self.resume(p)
        import heapq
        while self.events and self.now < 60000.0:
            self.now, _, fn = heapq.heappop(self.events)
            fn()
        if self.done != len(procs):
            self.violations.append(f'no progress: {self.done} of {len(procs)} workgroups finished')
        return self.violations

    def program(self, role, j):
        who, n, steps = (role, j), self.n, self.steps
        a = role == 'A'
        yj = j if a else n + j                                # this workgroup's words of the y layers
        mine = ('h1', 'x1') if a else ('h2', 'x2')

        def publish(layer, t, idx):
            self.store(who, layer, 0, t % RING, idx, t)

        def rearm(t):
            for layer, idx in ((mine[0], j), (mine[1], j), ('y1', yj), ('y2', yj)):
                self.store(who, layer, 0, (t + self.ahead) % RING, idx, SENT, rearm_turn=t + self.ahead - RING + 1)

        def form(tt):                                         # (rnn2 workgroups form cI: they wait for x1 anyway)
            if not a and tt < steps:
                self.store(who, 'cI', 0, tt % RING, j, tt)

        if a:
            yield ('poll', ('cI', 0, 0)); yield ('work', 0.5)                     # front half of step 0
        else:
            form(0); form(1)
        for t in range(steps):
            if self.drain:
                yield ('drain', who)
            if self.rearm_site == 'top':
                rearm(t)
            if a:
                if t > 0:
                    yield ('tag', ('xt', 0, 0, t - 1))
                yield ('work', 0.3)
                publish('x1', t, j); publish('h1', t, j)
                yield ('poll', ('h1', 0, t)); yield ('work', 0.5)                  # gh(t + 1), kept in registers
                yield ('poll', ('x2', 0, t)); yield ('work', 0.4); publish('y1', t, yj)
                yield ('poll', ('y1', 0, t))
                if self.rearm_site == 'fc2':
                    rearm(t)
                yield ('work', 0.4); publish('y2', t, yj)
                if t + 1 < steps:
                    yield ('poll', ('cI', 0, t + 1)); yield ('work', 0.5)          # (no sentinel from step 2 on: the words must be step t + 1's)
            else:
                yield ('poll', ('x1', 0, t)); yield ('work', 0.6)
                publish('x2', t, j); publish('h2', t, j)
                yield ('poll', ('x2', 0, t)); yield ('work', 0.4); publish('y1', t, yj)
                yield ('poll', ('y1', 0, t))
                if self.rearm_site == 'fc2':
                    rearm(t)
                yield ('work', 0.4); publish('y2', t, yj)
                if j == 0:
                    yield ('poll', ('y2', 0, t)); yield ('work', 0.8)
                    self.store(who, 'xt', 0, t % 2, 0, t)
                yield ('poll', ('h2', 0, t)); yield ('work', 0.5)                  # gh(t + 1): behind fc2 (and the sampling), in the wait for x1(t + 1)
                form(t + 1 + self.cond_lead)                                       # cI(t + 2) at the end of step t


def test_sparse_exchange_is_safe_under_adversarial_timing():
    for seed in range(60):
        for n in (1, 2, 3):
            v = SparseSim(seed, n=n, steps=24).run()
            assert not v, (seed, n, v[:3])


def test_sparse_model_detects_the_shortcuts():
    def broken(**kw):
        return any(SparseSim(seed, steps=30, **kw).run() for seed in range(80))
    assert broken(drain=False)                # a late re-arm (or a late cI) lands on / hides newer data
    assert broken(ahead=1)                    # re-armed one ahead: the publication of the next step can overtake the re-arm
    assert broken(cond_lead=0)                # cI formed only when it is needed: the readers (no sentinel to poll) take the entry's old content


def test_other_safe_distances():
    """Also safe (not what the kernel does): three ahead, or the re-arm at the top of the step -- every workgroup polls x2 and y1 of EVERY
    workgroup in every step, so when one has finished step t - 1 nobody still reads data of step t - 2."""
    for seed in range(30):
        for kw in (dict(ahead=3), dict(rearm_site='top')):
            v = SparseSim(seed, steps=24, **kw).run()
            assert not v, (seed, kw, v[:3])


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: BLOCK-SPARSE fc1 / fc2 (wrnn_sparse_kernel<NBP, FCS = true>).  A workgroup's four waves no longer meet in a dense fc stage: waves 0-1
# own ONE fc1 row block each and gather the surviving columns of x2, waves 2-3 one fc2 row block each and gather y1 -- a SUBSET of the
# producers, fixed by the weights, possibly leaving some producer unread by everybody.  "Every workgroup polls x2 and y1 of EVERY workgroup in
# every step" is gone, and with it round 5's skew bound.  What replaces it:
#   * EVERY wave waits for the tagged word x_{t-1} at the top of step t (rnn2's waves too: with an empty gather a wave has no other input and would
#     run free -- a fully pruned block row; the word is there long before x1(t) is, so this costs the chain nothing);
#   * a workgroup barrier at the top of every step, behind the drain (all four waves: the model's two halves);
#   * the dense fc3 of the sampling workgroup still reads ALL of y2(t): x_t exists => every wave 2-3 of every workgroup has published y2(t) =>
#     has passed the top of step t => EVERY wave of every workgroup has finished step t - 1 and drained it;
#   * a wave in step t (behind its x_{t-1} / x1(t) poll) therefore knows that everybody has finished step t - 2: re-arming entry t + 2 is safe;
#     a consumer that polls entry t + 2 in step t + 2 knows that the producer passed the top of step t + 1, i.e. drained the re-arm;
#   * cI is formed THREE steps ahead (cond_lead = 2): its readers, at the end of step t + 2, know the forming workgroup passed the top of step
#     t + 1 (its drain); two ahead needed "y1(t + 1) needed every x2(t + 1)", which a sparse gather does not give.
# Model: one process per HALF workgroup (q = 0: the fc1 tile, q = 1: the fc2 tile; both run the gate stages of their waves).
# ---------------------------------------------------------------------------------------------------------------------------------
class SparseFcSim(DuoSim):
    def __init__(self, seed, n=2, steps=24, ahead=2, drain=True, barrier=True, cond_lead=2, density=0.5, tag_all=True):
        super().__init__(seed, n_wg=n, slots=1, steps=steps)
        self.n, self.ahead, self.drain, self.use_barrier, self.cond_lead, self.tag_all = n, ahead, drain, barrier, cond_lead, tag_all
        ring = lambda producers, entries: [[[SENT] * producers for _ in range(entries)]]
        self.mem = {l: ring(2 * n, RING) for l in ('h1', 'x1', 'h2', 'x2', 'cI')}      # producers: the half workgroups of one layer
        self.mem['y1'], self.mem['y2'] = ring(2 * n, RING), ring(2 * n, RING)         # ... the q = 0 / q = 1 halves of all 2 n workgroups
        self.mem['xt'] = ring(1, 2)
        self.bar = {}
        pat = __import__('random').Random(1000 + seed)
        self.subset = lambda key, m: self._subset(pat, key, m, density)
        self._sub = {}

    def _subset(self, pat, key, m, density):
        if key not in self._sub:
            self._sub[key] = [k for k in range(m) if pat.random() < density]
        return self._sub[key]

    def resume(self, p):
        try:
            kind, arg = next(p)
        except StopIteration:
            self.done += 1
            return
        if kind == 'pollsub':                                # a gather: only the surviving columns' producers are looked at
            layer, t, idxs = arg

            def poll():
                words = self.mem[layer][0][t % RING]
                if any(words[k] is SENT for k in idxs):
                    self.at(0.5, poll)
                    return
                for k in idxs:
                    if words[k] != t:
                        self.violations.append(f'{layer}[{k}] read as step {words[k]} while gathering step {t}')
                self.resume(p)
            poll()
        elif kind == 'barrier':
            key, count = arg
            waiting = self.bar.setdefault(key, [])
            waiting.append(p)
            if len(waiting) == count:
                self.bar[key] = []
                for q in waiting:
                    self.at(0.0, (lambda q=q: self.resume(q)))
        else:
            def again():
                yield (kind, arg)
                yield from p
            DuoSim.resume(self, again())

    def run(self):
        procs = [self.program(role, j, q) for role in ('A', 'B') for j in range(self.n) for q in (0, 1)]
        for p in procs:
            self.resume(p)
        while self.events and self.now < 60000.0:
            self.now, _, fn = heapq.heappop(self.events)
            fn()
        if self.done != len(procs):
            self.violations.append(f'no progress: {self.done} of {len(procs)} half workgroups finished')
        return self.violations

    def program(self, role, j, q):
        who, n, steps = (role, j, q), self.n, self.steps
        a = role == 'A'
        me = 2 * j + q                                        # this half's words of its layer's h / x (and of cI)
        wgi = j if a else n + j                               # the workgroup's words of y1 (q = 0) / y2 (q = 1)
        mine = ('h1', 'x1') if a else ('h2', 'x2')
        fcl = 'y1' if q == 0 else 'y2'

        def publish(layer, t, idx):
            self.store(who, layer, 0, t % RING, idx, t)

        def rearm(t):
            for layer, idx in ((mine[0], me), (mine[1], me), (fcl, wgi)):
                self.store(who, layer, 0, (t + self.ahead) % RING, idx, SENT, rearm_turn=t + self.ahead - RING + 1)

        def form(tt):
            if not a and tt < steps:
                self.store(who, 'cI', 0, tt % RING, me, tt)

        sub = lambda layer, m=2 * n: self.subset((who, layer), m)
        if a:
            yield ('pollsub', ('cI', 0, sub('cI'))); yield ('work', 0.5)
        else:
            for tt in range(1 + self.cond_lead):
                form(tt)
        for t in range(steps):
            if self.drain:
                yield ('drain', who)
            if self.use_barrier:
                yield ('barrier', ((role, j, 'top'), 2))
            if t > 0 and (a or self.tag_all):                 # (rnn2's waves look at the x_{t-1} word too: a wave whose gathers are EMPTY has no other input)
                yield ('tag', ('xt', 0, 0, t - 1))
            if a:
                yield ('work', 0.3)
                publish('x1', t, me); publish('h1', t, me)
                yield ('pollsub', ('h1', t, sub('h1'))); yield ('work', 0.5)
            else:
                yield ('pollsub', ('x1', t, sub('x1'))); yield ('work', 0.6)
                publish('x2', t, me); publish('h2', t, me)
            if q == 0:
                yield ('pollsub', ('x2', t, sub('x2'))); yield ('work', 0.3); publish('y1', t, wgi)
            else:
                yield ('pollsub', ('y1', t, sub('y1'))); yield ('work', 0.3); publish('y2', t, wgi)
            rearm(t)
            if a:
                if t + 1 < steps:
                    yield ('pollsub', ('cI', t + 1, sub('cI'))); yield ('work', 0.5)
            else:
                if j == 0:                                    # the sampling workgroup: dense fc3 reads ALL of y2(t); its waves meet in an LDS barrier
                    yield ('poll', ('y2', 0, t)); yield ('work', 0.4)
                    yield ('barrier', ((role, j, 'smp'), 2))
                    if q == 1:
                        yield ('work', 0.4)
                        self.store(who, 'xt', 0, t % 2, 0, t)
                yield ('pollsub', ('h2', t, sub('h2'))); yield ('work', 0.5)
                form(t + 1 + self.cond_lead)


import heapq


def test_sparse_fc_exchange_is_safe_under_adversarial_timing():
    for seed in range(60):
        for n in (1, 2, 3):
            for density in (0.15, 0.5, 1.0):
                v = SparseFcSim(seed, n=n, steps=24, density=density).run()
                assert not v, (seed, n, density, v[:3])


def test_sparse_fc_model_detects_the_shortcuts():
    def broken(n=3, density=0.3, **kw):
        return any(SparseFcSim(seed, n=n, steps=30, density=density, **kw).run() for seed in range(150))
    assert broken(n=2, density=0.15, tag_all=False)              # a wave with an empty gather runs ahead of the ring
    assert broken(barrier=False)              # no step barrier: a half workgroup nobody gathers from drifts, and a re-arm (or new data) lands under its reads
    assert broken(cond_lead=1)                # cI two ahead, round 5's distance: its drain is no longer implied by what the reader has seen
    assert broken(drain=False)
    assert broken(ahead=1)
