"""A discrete-event MODEL of wrnn_duo_kernel's exchange as it stands after round 4 (round 6's MOL stage order: `order6`) (csrc/wrnn_duo.hip, header "Ring discipline"): four roles
per unit block -- rnn1 / rnn2 x {ih, hh} --, several slots (groups of segments) in flight, a ring of FOUR entries per layer, and three
different ways a consumer learns that a value is there:

* sentinel layers (h, x = residual sum, y; x_t): every word is the sentinel until its step's value lands; the producer re-arms its own
  words -- ih workgroups TWO steps ahead, after the last poll of their step, with the drain at the top of the NEXT step; the sampler
  THREE ahead with the drain in front of the re-arm;
* gh: one tagged word per (unit block, slot), two ring entries, never re-armed; an hh workgroup runs the gh stage of the LAST slot at the
  top of the next step (behind its sampling stage: `gh_shift`);
* cI: NO sentinel and no re-arm inside a launch -- rnn1's hh workgroup forms cI(t + 2) at the top of its step t and drains before it
  publishes anything of that step; the readers' own inputs already depend on that publication.  (The first two steps of a launch are
  polled: the buffer starts sentinel-filled.)

The arguments for those distances are statements about how far workgroups can drift apart.  Here they are checked under adversarial
timing instead: stores become visible after random delays, out of program order, now and then later than several whole steps (only a
drain waits for them); every stage takes a random time.  Checked: whatever a consumer accepts carries the tag of ITS step in every word,
a re-arm never lands on data that is still to be read or that is newer than the turn it was issued for (a dead-lock, also caught as "no
progress"), nobody overwrites a gh word that is still to be read, everybody finishes.  `raw=True` adds what 9-bit RAW adds (round 5's form): a
sixth sentinel layer -- the logit rows every rnn2 hh workgroup publishes for every slot behind its poll of y2, re-armed THREE ahead behind the
step's last y2 poll and a drain -- and `spp` sampling workgroups per slot (the kernel: four), each polling all logit rows of its slot and
publishing / re-arming ITS OWN x_t words.  The broken variants at the end show that the model
is not vacuous: each of the shortcuts the header argues against is caught for some timing.  A model of the protocol, not of the HIP code
-- tests/test_gpu_parity.py covers that."""
import heapq
import random

RING, GHRING = 4, 2
SENT = None


class DuoSim:
    def __init__(self, seed, n_wg=3, slots=2, steps=24, ahead_ih=2, ahead_hh=3, cond_ahead=2, cond_drain=True, ih_drain=True,
                 ih_drain_at_rearm=False, gh_shift=True, raw=False, spp=1, ahead_lg=3, lg_drain=True, order6=False):
        assert spp * slots <= n_wg                          # rnn2's hh workgroup j samples slot j (RAW: slot j // spp, as its sampler j % spp)
        self.raw, self.spp, self.ahead_lg, self.lg_drain = raw, (spp if raw else 1), ahead_lg, lg_drain
        self.rng = random.Random(seed)
        self.n_wg, self.G, self.steps = n_wg, slots, steps
        self.ahead_ih, self.ahead_hh, self.cond_ahead, self.cond_drain = ahead_ih, ahead_hh, cond_ahead, cond_drain
        self.ih_drain, self.ih_drain_at_rearm = ih_drain, ih_drain_at_rearm
        self.order6 = order6 and not raw                    # round 6 (MOL): gh(0) .. gh(n - 1) in slot order, the sampler of slot s samples right behind gh(s)
        self.gh_shift = gh_shift and slots >= 2 and not self.order6     # hh workgroups: the last slot's gh stage runs at the top of the next step
        ring = lambda n: [[[SENT] * n_wg for _ in range(n)] for _ in range(slots)]          # [slot][entry][producer]
        self.mem = {l: ring(RING) for l in ('h1', 'x1', 'y1', 'h2', 'x2', 'y2', 'cI')}
        self.mem['gh1'], self.mem['gh2'] = ring(GHRING), ring(GHRING)
        self.mem['xt'] = [[[SENT] * self.spp for _ in range(RING)] for _ in range(slots)]   # one producer per sampler of the slot (MOL: one)
        self.mem['lg'] = ring(RING)                                                         # RAW: the logit rows of every rnn2 hh workgroup
        self.now, self.events, self.seq = 0.0, [], 0
        self.pending, self.violations, self.done = {}, [], 0
        self.rearm_turn = {}

    def at(self, dt, fn):
        self.seq += 1
        heapq.heappush(self.events, (self.now + dt, self.seq, fn))

    def store(self, who, layer, slot, entry, j, value, rearm_turn=None):
        self.pending[who] = self.pending.get(who, 0) + 1

        def land():
            old = self.mem[layer][slot][entry][j]
            if value is SENT and old is not SENT and old >= rearm_turn:
                self.violations.append(f're-arm of {layer}[{slot}][{entry}][{j}] (for step {rearm_turn}) landed on data of step {old}')
            self.mem[layer][slot][entry][j] = value
            self.pending[who] -= 1
        # (now and then an acknowledgement that takes longer than ten whole steps: only a drain makes that harmless)
        self.at(self.rng.uniform(300.0, 600.0) if self.rng.random() < 0.006 else self.rng.choice([0.1, 0.5, 1.0, 3.0, 8.0]), land)

    def run(self):
        procs = [self.program(role, j) for role in ('Aih', 'Ahh', 'Bih', 'Bhh') for j in range(self.n_wg)]
        for p in procs:
            self.resume(p)
        while self.events and self.now < 60000.0:             # (a dead-locked model polls for ever: the clock is the limit)
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
        if kind == 'work':
            self.at(self.rng.choice([0.2, 1.0, 2.0, 6.0]) * arg, lambda: self.resume(p))
        elif kind == 'poll':                                 # sentinel layer: re-read until no word is the sentinel, then every word must be step t's
            layer, slot, t = arg

            def poll():
                words = self.mem[layer][slot][t % RING]
                if any(wd is SENT for wd in words):
                    self.at(0.5, poll)
                    return
                for j, wd in enumerate(words):
                    if wd != t:
                        self.violations.append(f'{layer}[{slot}][{j}] read as step {wd} while polling for step {t}')
                self.resume(p)
            poll()
        elif kind == 'tag':                                  # gh: re-read the one word until it carries step t's tag
            layer, slot, j, t = arg

            def poll():
                wd = self.mem[layer][slot][t % GHRING][j]
                if wd is not SENT and wd > t:
                    self.violations.append(f'{layer}[{slot}][{j}] of step {t} was overwritten by step {wd} before it was read')
                    self.resume(p)
                elif wd != t:
                    self.at(0.5, poll)
                else:
                    self.resume(p)
            poll()
        elif kind == 'drain':
            who = arg

            def drain():
                if self.pending.get(who, 0) > 0:
                    self.at(0.2, drain)
                else:
                    self.resume(p)
            drain()

    def program(self, role, j):
        who, G, n = (role, j), self.G, self.n_wg

        def word(layer):
            return j if layer != 'xt' else j % self.spp

        def publish(layer, i, t):
            self.store(who, layer, i, t % RING, word(layer), t)

        def rearm(layers, t, ahead, slots):
            for layer in layers:
                for i in slots:
                    self.store(who, layer, i, (t + ahead) % RING, word(layer), SENT, rearm_turn=t + ahead - RING + 1)

        if role in ('Aih', 'Bih'):
            a = role == 'Aih'
            operand, gh, mine = ('cI', 'gh1', ('h1', 'x1', 'y1')) if a else ('x1', 'gh2', ('h2', 'x2', 'y2'))
            for t in range(self.steps):
                for i in range(G):                           # gate stages
                    yield ('poll', (operand, i, t))          # (cI: a poll finds no sentinel from step 2 on -- the words must then be step t's)
                    if i == 0 and self.ih_drain and not self.ih_drain_at_rearm:
                        yield ('drain', who)                 # last step's re-arm stores are out before anything of this step is published
                    yield ('work', 1.0)
                    if t > 0:
                        yield ('tag', (gh, i, j, t))
                        if a:
                            yield ('poll', ('xt', i, t - 1))
                    publish(mine[1], i, t); publish(mine[0], i, t)
                for i in range(G):                           # fc stages
                    yield ('poll', ('x2' if a else 'y1', i, t))
                    if i == G - 1:                           # after the last poll of the step: re-arm, all slots
                        if self.ih_drain and self.ih_drain_at_rearm:
                            yield ('drain', who)
                        rearm(mine, t, self.ahead_ih, range(G))
                    yield ('work', 0.4)
                    publish(mine[2], i, t)
        elif role == 'Ahh':
            def form(tt):
                if tt < self.steps:
                    for i in range(G):
                        self.store(who, 'cI', i, tt % RING, j, tt)
            for tt in range(self.cond_ahead):
                form(tt)
            def gh_stages(t):                                # (slot, step) of the gh stages of iteration t
                if not self.gh_shift:
                    return [(i, t) for i in range(G)] if t < self.steps else []
                return ([(G - 1, t - 1)] if t > 0 else []) + ([(i, t) for i in range(G - 1)] if t < self.steps else [])
            for t in range(self.steps + 1):
                if t < self.steps:
                    form(t + self.cond_ahead)
                    if self.cond_drain:
                        yield ('drain', who)
                for i, tt in gh_stages(t):
                    yield ('poll', ('h1', i, tt)); yield ('work', 1.0)
                    if tt + 1 < self.steps:
                        self.store(who, 'gh1', i, (tt + 1) % GHRING, j, tt + 1)
        else:
            def gh_stages(t):
                if not self.gh_shift:
                    return [(i, t) for i in range(G)] if t < self.steps else []
                return ([(G - 1, t - 1)] if t > 0 else []) + ([(i, t) for i in range(G - 1)] if t < self.steps else [])
            for t in range(self.steps + 1):
                for i, tt in gh_stages(t):
                    yield ('poll', ('h2', i, tt)); yield ('work', 1.0)
                    if tt + 1 < self.steps:
                        self.store(who, 'gh2', i, (tt + 1) % GHRING, j, tt + 1)
                    if self.order6 and i == j and j < G:     # round 6: the sampling stage of slot j right behind gh(j)
                        yield ('poll', ('y2', j, t))
                        yield ('drain', who)
                        rearm(('xt',), t, self.ahead_hh, [j])
                        yield ('work', 0.6)
                        publish('xt', j, t)
                if t == self.steps or self.order6:
                    if t == self.steps:
                        break
                    continue
                if self.raw:
                    for i in range(G):                       # logits stages: this workgroup's rows of fc3 for every slot
                        yield ('poll', ('y2', i, t))
                        if i == G - 1:                       # behind the last y2 poll of the step: drain, re-arm the own rows of every slot
                            if self.lg_drain:
                                yield ('drain', who)
                            rearm(('lg',), t, self.ahead_lg, range(G))
                        yield ('work', 0.4)
                        publish('lg', i, t)
                    if j < self.spp * G:                     # sampler j % spp of slot j // spp: its own x_t words
                        slot = j // self.spp
                        yield ('poll', ('lg', slot, t))
                        yield ('drain', who)
                        rearm(('xt',), t, self.ahead_hh, [slot])
                        yield ('work', 0.6)
                        publish('xt', slot, t)
                elif j < G:                                  # the sampler of slot j
                    yield ('poll', ('y2', j, t))
                    yield ('drain', who)
                    rearm(('xt',), t, self.ahead_hh, [j])
                    yield ('work', 0.6)
                    publish('xt', j, t)


def test_duo_exchange_is_safe_under_adversarial_timing():
    for seed in range(40):
        for slots in (1, 2, 3):
            v = DuoSim(seed, n_wg=3, slots=slots, steps=24).run()
            assert not v, (seed, slots, v[:3])


def test_duo_exchange_round6_order_is_safe_under_adversarial_timing():
    """MOL since round 6 (csrc/wrnn_duo.hip, duo_hh's step loop): no gh stage deferred across the step boundary, sampling behind gh(my_slot)."""
    for seed in range(40):
        for slots in (1, 2, 3):
            v = DuoSim(seed, n_wg=3, slots=slots, steps=24, order6=True).run()
            assert not v, (seed, slots, v[:3])
    assert any(DuoSim(seed, steps=30, order6=True, ahead_hh=1).run() for seed in range(60))      # (not vacuous: x_t re-armed one ahead is caught)


def test_duo_exchange_raw_form_is_safe_under_adversarial_timing():
    """9-bit RAW: the logits layer and several sampling workgroups per slot (csrc/wrnn_duo.hip: kind-2 and kind-4 stages)."""
    for seed in range(30):
        for n_wg, slots, spp in ((4, 2, 2), (4, 1, 4), (3, 3, 1), (6, 3, 2)):
            v = DuoSim(seed, n_wg=n_wg, slots=slots, steps=24, raw=True, spp=spp).run()
            assert not v, (seed, n_wg, slots, spp, v[:3])


def test_other_safe_distances():
    """Also safe (not what the kernel does): ih layers re-armed three ahead -- the re-arm site lies behind the fc stage's poll of the
    other layer's residual sum, which needed every hh workgroup's gh of this step, i.e. every reader is past the data of step t - 1 -- with
    the drain at either place; the sampler's x_t two ahead (its drain precedes its publication in every step)."""
    for seed in range(25):
        for kw in (dict(ahead_ih=3), dict(ahead_ih=3, ih_drain_at_rearm=True), dict(ahead_hh=2), dict(gh_shift=False)):
            v = DuoSim(seed, steps=24, **kw).run()
            assert not v, (seed, kw, v[:3])


def test_duo_model_detects_the_shortcuts():
    def broken(**kw):
        return any(DuoSim(seed, steps=30, **kw).run() for seed in range(60))
    # ih layers two ahead but drained only at the re-arm site (round 3's place): a consumer of step t + 2 may not see the re-arm yet
    assert broken(ih_drain_at_rearm=True)
    assert broken(ih_drain=False)
    assert broken(ahead_ih=1)
    # cI without a sentinel needs the two steps of lead (with one slot in flight: the deferred gh stage of a deeper pipeline happens to
    # cover one step of it) AND the drain before the step's publications
    assert broken(cond_ahead=1, slots=1)
    assert broken(cond_drain=False)
    # the sampler's x_t one ahead: its publication of this step can overtake the re-arm
    assert broken(ahead_hh=1)
    # RAW: the same for the logit rows; and their re-arm without the drain in front of it
    # (seen on a workgroup that does not sample: a sampler's own drain, in front of its x_t re-arm, happens to cover the logit rows too)
    assert broken(raw=True, n_wg=6, slots=2, spp=2, ahead_lg=1)
    assert broken(raw=True, n_wg=4, slots=2, spp=2, ahead_hh=1)
