"""A discrete-event MODEL of wrnn_chain_kernel's exchange (csrc/wrnn_chain.hip, round 5): one workgroup per CU, n rnn1 + n rnn2 workgroups per
cluster (the kernel: 32 + 32), up to four slots (groups of segments) per cluster, every stage in ONE instruction stream per workgroup as a
loop over the slots:

    rnn1 j:  drain | per slot: x_{t-1} (tagged words) -> cell -> publish x1, h1, RE-ARM own words of h1, x1 (RAW: and of its logit rows) in
             entry t + 2 | per slot: poll h1(t) -> gh (stays in LDS) | per slot: poll cI(t + 1) -> W_ih . cI | form cI(t + 2) of every slot |
             MOL: workgroup i: poll y2(t) of slot i -> fc3 + sampling -> x_t as tagged words in entry t % 2
             RAW: per slot: poll y2(t) -> publish its 16 logit rows; workgroup spp * i + q: poll ALL logit rows of slot i -> sample ITS segments
    rnn2 j:  drain | per slot: poll x1(t) -> cell -> publish x2, h2 | per slot: poll x2(t) -> publish y1 | per slot: poll y1(t) -> publish y2,
             RE-ARM own words of h2, x2, y1, y2 in entry t + 2 | per slot: poll h2(t) -> gh

Sentinel layers: four ring entries, re-armed two steps ahead, drained at the top of the next step.  cI: no sentinel inside a launch (formed
at the end of step t for step t + 2 and covered by the same drain; gathered at the end of step t + 1, behind the poll of h1(t + 1) of EVERY
rnn1 workgroup).  x_t: tagged words, two entries, never re-armed.  Checked under adversarial timing with the engine of
tests/test_duo_exchange_model.py (stores land after random delays, out of order, now and then later than ten whole steps -- only a drain
waits for them): whatever a consumer accepts carries ITS step in every word, no re-arm lands on data still to be read, no tagged word is
overwritten before every reader has seen it, everybody finishes.  The broken variants show that the model is not vacuous.  A model of the
protocol, not of the HIP code (tests/test_gpu_parity.py covers that)."""
import heapq

from test_duo_exchange_model import DuoSim, RING, SENT


class ChainSim(DuoSim):
    def __init__(self, seed, n=3, slots=2, steps=24, raw=False, spp=1, ahead=2, drain=True, cond_lead=2, front_behind_gh=True, xt_entries=2):
        spp = spp if raw else 1
        assert spp * slots <= n
        super().__init__(seed, n_wg=n, slots=slots, steps=steps)
        self.n, self.G, self.raw_mode, self.spp = n, slots, raw, spp
        self.ahead, self.drain, self.cond_lead, self.front_behind_gh, self.xt_entries = ahead, drain, cond_lead, front_behind_gh, xt_entries
        ring = lambda producers, entries: [[[SENT] * producers for _ in range(entries)] for _ in range(slots)]
        self.mem = {l: ring(n, RING) for l in ('h1', 'x1', 'cI', 'h2', 'x2', 'y1', 'y2', 'lg')}
        self.mem['xt'] = ring(spp, xt_entries)

    def run(self):
        procs = [self.program(role, j) for role in ('A', 'B') for j in range(self.n)]
        for p in procs:
            self.resume(p)
        while self.events and self.now < 60000.0:
            self.now, _, fn = heapq.heappop(self.events)
            fn()
        if self.done != len(procs):
            self.violations.append(f'no progress: {self.done} of {len(procs)} workgroups finished')
        return self.violations

    def resume(self, p):
        try:
            kind, arg = next(p)
        except StopIteration:
            self.done += 1
            return
        if kind == 'tagall':                                 # x_t: re-read until EVERY word of the entry carries step t's tag
            slot, t = arg

            def poll():
                words = self.mem['xt'][slot][t % self.xt_entries]
                if any(wd is not SENT and wd > t for wd in words):
                    self.violations.append(f'xt[{slot}] of step {t} was overwritten by step {max(wd for wd in words if wd is not SENT)} before it was read')
                    self.resume(p)
                elif any(wd != t for wd in words):
                    self.at(0.5, poll)
                else:
                    self.resume(p)
            poll()
            return
        # (everything else: the parent's engine; re-inject the event we consumed)
        def again():
            yield (kind, arg)
            yield from p
        super().resume(again())

    def program(self, role, j):
        who, n, G, steps = (role, j), self.n, self.G, self.steps
        a = role == 'A'

        def publish(layer, i, t, idx=None):
            self.store(who, layer, i, t % RING, j if idx is None else idx, t)

        def rearm(layers, i, t):
            for layer in layers:
                self.store(who, layer, i, (t + self.ahead) % RING, j, SENT, rearm_turn=t + self.ahead - RING + 1)

        if a:
            def form(tt):
                if tt < steps:
                    for i in range(G):
                        self.store(who, 'cI', i, tt % RING, j, tt)
            for tt in range(self.cond_lead):
                form(tt)
            for i in range(G):
                yield ('poll', ('cI', i, 0)); yield ('work', 0.5)
            for t in range(steps):
                if self.drain:
                    yield ('drain', who)
                for i in range(G):                           # back: the chain comes in here
                    if t > 0:
                        yield ('tagall', (i, t - 1))
                    yield ('work', 0.3)
                    publish('x1', i, t); publish('h1', i, t)
                    rearm(('h1', 'x1') + (('lg',) if self.raw_mode else ()), i, t)
                if self.front_behind_gh:
                    for i in range(G):
                        yield ('poll', ('h1', i, t)); yield ('work', 1.0)
                if t + 1 < steps:
                    for i in range(G):
                        yield ('poll', ('cI', i, t + 1)); yield ('work', 1.0)
                if not self.front_behind_gh:
                    for i in range(G):
                        yield ('poll', ('h1', i, t)); yield ('work', 1.0)
                form(t + self.cond_lead)
                if not self.raw_mode:
                    if j < G:                                # the sampler of slot j
                        yield ('poll', ('y2', j, t)); yield ('work', 0.8)
                        self.store(who, 'xt', j, t % self.xt_entries, 0, t)
                else:
                    for i in range(G):                       # its 16 logit rows of every slot
                        yield ('poll', ('y2', i, t)); yield ('work', 0.4)
                        publish('lg', i, t)
                    if j < self.spp * G:                     # sampler j % spp of slot j // spp
                        slot = j // self.spp
                        yield ('poll', ('lg', slot, t)); yield ('work', 0.6)
                        self.store(who, 'xt', slot, t % self.xt_entries, j % self.spp, t)
        else:
            for t in range(steps):
                if self.drain:
                    yield ('drain', who)
                for i in range(G):
                    yield ('poll', ('x1', i, t)); yield ('work', 1.0)
                    publish('x2', i, t); publish('h2', i, t)
                for i in range(G):
                    yield ('poll', ('x2', i, t)); yield ('work', 0.4)
                    publish('y1', i, t)
                for i in range(G):
                    yield ('poll', ('y1', i, t)); yield ('work', 0.4)
                    publish('y2', i, t)
                    rearm(('h2', 'x2', 'y1', 'y2'), i, t)
                for i in range(G):
                    yield ('poll', ('h2', i, t)); yield ('work', 1.0)


def test_chain_exchange_is_safe_under_adversarial_timing():
    for seed in range(30):
        for n, slots in ((3, 1), (3, 2), (4, 3), (4, 4)):
            v = ChainSim(seed, n=n, slots=slots, steps=24).run()
            assert not v, (seed, n, slots, v[:3])


def test_chain_exchange_raw_form_is_safe_under_adversarial_timing():
    """9-bit RAW: the logit rows (a seventh sentinel layer, published by the rnn1 workgroups behind their poll of y2, re-armed with h1 and x1)
    and several sampling workgroups per slot, each with its own tagged x_t words."""
    for seed in range(30):
        for n, slots, spp in ((4, 1, 4), (4, 2, 2), (6, 3, 2), (5, 1, 1)):
            v = ChainSim(seed, n=n, slots=slots, steps=24, raw=True, spp=spp).run()
            assert not v, (seed, n, slots, spp, v[:3])


def test_one_x_t_entry_would_do():
    """Also safe (not what the kernel does): ONE entry of tagged x_t words -- x_t(t + 1) needs x1(t + 1) of every rnn1 workgroup, each of which
    has read x_t(t) by then; the kernel's second entry is margin, not a requirement."""
    for seed in range(20):
        for kw in (dict(n=3, slots=2), dict(n=4, slots=1, raw=True, spp=4)):
            v = ChainSim(seed, steps=24, xt_entries=1, **kw).run()
            assert not v, (seed, kw, v[:3])


def test_chain_model_detects_the_shortcuts():
    def broken(**kw):
        return any(ChainSim(seed, steps=30, **kw).run() for seed in range(60))
    assert broken(drain=False)                               # a late re-arm / cI store lands on the next turn's data
    assert broken(ahead=1)                                   # one ahead: the entry's readers of step t - 3 are done, its NEXT writer is not held back
    assert broken(cond_lead=1)                               # cI(t + 1) formed at the end of step t: nothing orders it before its gather
    assert broken(front_behind_gh=False, slots=1)            # the gather of cI(t + 1) in front of the poll of h1(t): no dependency on the other workgroups' drain
