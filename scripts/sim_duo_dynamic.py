#!/usr/bin/env python
"""Time-stepped model of ONE cluster of wrnn_duo_kernel (MOL) at depth G: what a READINESS-DRIVEN stage order in the ih workgroups would buy
over the static order with blocking waits (round 6).  CPU only.  Servers: A / B = the ih workgroups of rnn1 / rnn2 (their 32 unit blocks move
together), AH = rnn1's hh workgroups, BH_s = rnn2's hh workgroup that samples slot s (it also runs the gh stage of every slot).  A stage is
split in its FRONT (operand -> MFMA tiles -> partial tiles in LDS) and its BACK half (barrier, pointwise, publish); the kernel runs the back
half of stage k right behind the front of stage k, the loads of stage k + 1 issued in between (folded into the durations).

Durations (us) are fitted to the measured steps at depth 1 / 2 / 3 / 4 / 8 = 12.3 / 16.4 / 19.4 / 22.7 / 37.4 (profiles/r06v_*): see FIT.

    python scripts/sim_duo_dynamic.py
"""
import sys

DT = 0.01

P = dict(
    Fg=1.95, Kg=0.80,          # gates: front (issue, check, build, 96 MFMAs, partial tiles) | back (barrier, reduce, GRU cell, publish x, h)
    Ff=1.20, Kf=0.45,          # fc1 / fc2
    GH=2.30,                   # an hh workgroup's gh stage, front + back
    Fs=1.20, Ks=0.80,          # fc3 (two tiles) | MoL sampling + publish x_t
    H=0.80, Hl=0.60,           # publish -> usable by a polling consumer: across XCDs | inside one L2
    NP=2,                      # partial-tile sets in LDS: fronts that may be pending without their back half
)


class Sim:
    def __init__(self, G, policy_a='static', policy_b='static', hh_policy='static', p=None, steps=40):
        self.G, self.p, self.steps = G, dict(P, **(p or {})), steps
        self.ready = {}                              # (name, slot, step) -> time usable at the consumer
        for i in range(G):
            self.ready[('xt', i, -1)] = 0.0
            self.ready[('gh1', i, 0)] = 0.0
            self.ready[('gh2', i, 0)] = 0.0
        self.gh2_parts = {}
        self.pol = {'A': policy_a, 'B': policy_b}
        self.hh_policy = hh_policy
        self.done_t = {}

    # ---- ih server ---------------------------------------------------------------------------------------------------------------------
    def ih_inputs(self, srv, kind, i, t):
        if kind == 'Fg':
            return [] if srv == 'A' else [('x1', i, t)]
        if kind == 'Kg':
            return [('xt', i, t - 1), ('gh1', i, t)] if srv == 'A' else [('gh2', i, t)]
        if kind == 'Ff':
            return [('x2', i, t)] if srv == 'A' else [('y1', i, t)]
        return []

    def ih_outputs(self, srv, kind, i, t, end):
        p = self.p
        if kind == 'Kg':
            if srv == 'A':
                self.ready[('x1', i, t)] = end + p['H']
                self.ready[('h1', i, t)] = end + p['Hl']
            else:
                self.ready[('x2', i, t)] = end + p['H']
                self.ready[('h2', i, t)] = end + p['Hl']
        elif kind == 'Kf':
            if srv == 'A':
                self.ready[('y1', i, t)] = end + p['H']
            else:
                self.ready[('y2', i, t)] = end + p['Hl']

    def is_ready(self, names, now):
        return all(self.ready.get(n, 1e30) <= now for n in names)

    def run(self):
        G, p = self.G, self.p
        # per ih server: per step the set of tasks not yet started; pointers
        st = {}
        for srv in ('A', 'B'):
            st[srv] = dict(busy=0.0, t=0, nFg=0, nFf=0, pendK=[], cur=None, order_pos=0)
        hh = {}
        for s in ['AH'] + [('BH', k) for k in range(G)]:
            hh[s] = dict(busy=0.0, prog=None, pos=0, cur=None)
        # hh programs (static, blocking): MOL order of round 6: gh(0..my_slot) | sample | gh(my_slot + 1 ..)
        def hh_prog(s):
            out = []
            for t in range(self.steps):
                for i in range(G):
                    out.append(('GH', i, t))
                    if s != 'AH' and i == s[1]:
                        out.append(('S', i, t))
            return out
        for s in hh:
            hh[s]['prog'] = hh_prog(s)
        now = 0.0
        tmax = self.steps * 80.0
        while now < tmax:
            # ---------------- ih servers
            for srv in ('A', 'B'):
                S = st[srv]
                if S['cur'] is not None and now >= S['busy']:
                    kind, i, t = S['cur']
                    self.ih_outputs(srv, kind, i, t, S['busy'])
                    if kind in ('Fg', 'Ff'):
                        S['pendK'].append(('Kg' if kind == 'Fg' else 'Kf', i, t))
                    if kind == 'Kf' and i == G - 1:
                        pass
                    S['cur'] = None
                if S['cur'] is None and S['t'] < self.steps:
                    task = self.choose(srv, S, now)
                    if task is not None:
                        kind, i, t = task
                        S['cur'] = task
                        S['busy'] = now + p[kind]
                        if kind == 'Fg':
                            S['nFg'] += 1
                        elif kind == 'Ff':
                            S['nFf'] += 1
                        else:
                            S['pendK'].remove(task)
                            if kind == 'Kf' and i == G - 1:      # the step's last task
                                S['t'] += 1
                                S['nFg'] = S['nFf'] = 0
            # ---------------- hh servers (static order, blocking; look-ahead folded into Hl)
            for s, S in hh.items():
                if S['cur'] is not None and now >= S['busy']:
                    kind, i, t = S['cur']
                    if kind == 'GH':
                        if s == 'AH':
                            self.ready[('gh1', i, t + 1)] = S['busy'] + p['Hl']
                        else:
                            parts = self.gh2_parts.setdefault((i, t + 1), [])
                            parts.append(S['busy'] + p['Hl'])
                            if len(parts) == G:
                                self.ready[('gh2', i, t + 1)] = max(parts)
                    else:
                        self.ready[('xt', i, t)] = S['busy'] + p['H']
                        self.done_t[(i, t)] = S['busy']
                    S['cur'] = None
                    S['pos'] += 1
                if S['cur'] is None and S['pos'] < len(S['prog']):
                    kind, i, t = S['prog'][S['pos']]
                    need = [('h1' if s == 'AH' else 'h2', i, t)] if kind == 'GH' else [('y2', i, t)]
                    if self.is_ready(need, now):
                        S['cur'] = (kind, i, t)
                        S['busy'] = now + (p['GH'] if kind == 'GH' else p['Fs'] + p['Ks'])
            if all(st[s]['t'] >= self.steps for s in st):
                break
            now += DT
        a, b = self.steps // 3, self.steps - 3
        return (self.done_t[(0, b)] - self.done_t[(0, a)]) / (b - a)

    def choose(self, srv, S, now):
        G, t, pol = self.G, S['t'], self.pol[srv]
        NP = self.p['NP']
        pend = S['pendK']
        if pol == 'static':
            # kernel order: Fg0 Kg0 Fg1 Kg1 .. Ff0 Kf0 ..; every wait blocks
            if pend:
                k = pend[0]
                return k if self.is_ready(self.ih_inputs(srv, *k), now) else None
            if S['nFg'] < G:
                k = ('Fg', S['nFg'], t)
            else:
                k = ('Ff', S['nFf'], t)
            return k if self.is_ready(self.ih_inputs(srv, *k), now) else None
        if pol == 'defer1':
            # as static, but a back half whose inputs are not there is deferred behind the NEXT front (if that one is ready and a partial-tile set is free)
            if pend:
                k = pend[0]
                if self.is_ready(self.ih_inputs(srv, *k), now):
                    return k
                if len(pend) >= NP:
                    return None
            if S['nFg'] < G:
                k = ('Fg', S['nFg'], t)
            elif S['nFf'] < G:
                k = ('Ff', S['nFf'], t)
            else:
                return None
            return k if self.is_ready(self.ih_inputs(srv, *k), now) else None
        if pol == 'dyn':
            # readiness-driven: oldest ready back half > ready fc front > ready gates front
            for k in pend:
                if self.is_ready(self.ih_inputs(srv, *k), now):
                    return k
            if len(pend) >= NP:
                return None
            cands = []
            if S['nFf'] < G and S['nFf'] < S['nFg']:
                cands.append(('Ff', S['nFf'], t))
            if S['nFg'] < G:
                cands.append(('Fg', S['nFg'], t))
            for k in cands:
                # an fc front needs its own slot's gates back half published (data dependence takes care of it)
                if self.is_ready(self.ih_inputs(srv, *k), now):
                    return k
            return None
        raise ValueError(pol)


MEASURED = {1: 12.3, 2: 16.4, 3: 19.4, 4: 22.7, 8: 37.4}

if __name__ == '__main__':
    print('depth | measured | static (the kernel) | A defers a back half behind the next front | A and B defer | A and B readiness-driven | same, 4 partial-tile sets')
    for G in (1, 2, 3, 4, 6, 8):
        r = [Sim(G).run(), Sim(G, 'defer1', 'static').run(), Sim(G, 'defer1', 'defer1').run(), Sim(G, 'dyn', 'dyn').run(), Sim(G, 'dyn', 'dyn', p=dict(NP=4)).run()]
        print(f'{G}: {MEASURED.get(G, float("nan")):6.1f} | ' + ' | '.join(f'{x:6.2f}' for x in r))
