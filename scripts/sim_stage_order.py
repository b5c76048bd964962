#!/usr/bin/env python
"""A small discrete-event model of ONE cluster of wrnn_duo_kernel at depth G: which static order of the stages bounds a step, and what
another order would buy.  CPU only; durations from the phase clocks of the final round-4 kernel (profiles/r04o_phase_clocks.log, depth 8 =
the least waiting: cycles / 2.4 GHz) and the hop latency of DESIGN.md section 9 item 2.  Servers: the ih workgroups of rnn1 and rnn2 (all
32 unit blocks move together: one server each) and rnn2's hh workgroups that sample (one server per slot; each also runs the gh stage of
every slot).  rnn1's hh workgroups are off every chain and left out.

    python scripts/sim_stage_order.py            # prints us per step for depth 1..8 and a few orders
"""

HOP = 1.2            # publish -> the consumer's poll sees it (us)
A_G_MFMA, A_G_BACK = 2.35, 1.15      # rnn1 gates: front (issue, cI load, 96 MFMAs + partial tiles) | back half (reduce, pointwise, publish x1, h1)
B_G_MFMA, B_G_BACK = 2.45, 1.15      # rnn2 gates (the front includes the L2 latency of the x1 fragments: no look-ahead in the ih role)
FC_MFMA, FC_BACK = 1.5, 0.55         # fc1 / fc2
GH = 2.5                             # an hh workgroup's gh stage (operand already requested one stage ahead)
SAMPLE = 2.2                         # fc3 (two tiles) + MoL sampling + publish x_t


def simulate(G, steps=60, a_order=None, b_order=None, s_order=None):
    """orders: lists of (kind, slot) per step; kind 'g' / 'f' for the ih servers; for sampler s: ('h', slot) and ('s', s)."""
    a_order = a_order or [('g', i) for i in range(G)] + [('f', i) for i in range(G)]
    b_order = b_order or [('g', i) for i in range(G)] + [('f', i) for i in range(G)]
    INF = float('inf')
    t_xt = {(i, -1): 0.0 for i in range(G)}          # x_t(slot, step) available at the consumer
    t_x1, t_x2, t_y1, t_y2, t_h2 = {}, {}, {}, {}, {}
    free = {'A': 0.0, 'B': 0.0, **{('S', s): 0.0 for s in range(G)}}
    prog = {'A': [(t, k, i) for t in range(steps) for k, i in a_order], 'B': [(t, k, i) for t in range(steps) for k, i in b_order]}
    for s in range(G):
        so = (s_order(s, G) if s_order else [('h', i) for i in range(G)] + [('s', s)])
        prog[('S', s)] = [(t, k, i) for t in range(steps) for k, i in so]
    pos = {k: 0 for k in prog}
    done_step = {}
    progress = True
    while progress:
        progress = False
        for srv in prog:
            while pos[srv] < len(prog[srv]):
                t, k, i = prog[srv][pos[srv]]
                now = free[srv]
                if srv == 'A' and k == 'g':
                    need = t_xt.get((i, t - 1))
                    if need is None:
                        break
                    end_front = now + A_G_MFMA                      # cI is always there
                    end = max(end_front, need) + A_G_BACK           # the back half waits for x_{t-1}
                    t_x1[(i, t)] = end + HOP
                elif srv == 'A':
                    need = t_x2.get((i, t))
                    if need is None:
                        break
                    end = max(now, need) + FC_MFMA + FC_BACK
                    t_y1[(i, t)] = end + HOP
                elif srv == 'B' and k == 'g':
                    need = t_x1.get((i, t))
                    if need is None:
                        break
                    end = max(now, need) + B_G_MFMA + B_G_BACK
                    t_x2[(i, t)] = end + HOP
                    t_h2[(i, t)] = end + HOP
                elif srv == 'B':
                    need = t_y1.get((i, t))
                    if need is None:
                        break
                    end = max(now, need) + FC_MFMA + FC_BACK
                    t_y2[(i, t)] = end + HOP
                elif k == 'h':
                    need = t_h2.get((i, t))
                    if need is None:
                        break
                    end = max(now, need - 0.5) + GH                 # (look-ahead: the fragments were requested a stage earlier)
                else:
                    need = t_y2.get((i, t))
                    if need is None:
                        break
                    end = max(now, need - 0.5) + SAMPLE
                    t_xt[(i, t)] = end + HOP
                    done_step[(i, t)] = end
                free[srv] = end
                pos[srv] += 1
                progress = True
    a, b = steps // 3, steps - 2
    return (done_step[(0, b)] - done_step[(0, a)]) / (b - a)


def simulate_v2(G, steps=60, hh_order=None, bs_order=None, look=0.5):
    """The OTHER cut of the four roles (not built): the fc stages on the hh workgroups (A-hh: gh1 + fc1, B-hh: gh2 + fc2), the sampling on rnn2's ih
    workgroups (B-ih s: gates + sample(s)), rnn1's ih workgroups: gates only.  Busy time per slot-step 3.5 | 3.6 (+ 2.2 once) | 4.55 | 4.55 us instead
    of 5.55 | 5.65 | 2.5 | 2.5 (+ 2.2).  hh_order: the hh workgroups' stage order, [('h' | 'f', slot)]."""
    ev = {}   # (name, slot, step) -> time available at consumer
    for i in range(G): ev[('xt', i, -1)] = 0.0; ev[('h1', i, -1)] = 0.0; ev[('h2', i, -1)] = 0.0
    def ord_hh(): return hh_order(G) if hh_order else [('h', i) for i in range(G)] + [('f', i) for i in range(G)]
    prog = {'A': [(t,'g',i) for t in range(steps) for i in range(G)]}
    for s in range(G):
        o = bs_order(s, G) if bs_order else [('g', i) for i in range(G)] + [('s', s)]
        prog[('B', s)] = [(t,k,i) for t in range(steps) for k,i in o]
    prog['AH'] = [(t,k,i) for t in range(steps) for k,i in ord_hh()]
    prog['BH'] = [(t,k,i) for t in range(steps) for k,i in ord_hh()]
    free = {k: 0.0 for k in prog}; pos = {k: 0 for k in prog}
    done = {}
    # x2 of a slot needs ALL B-ih servers' gates of that slot: track per server and take max
    bdone = {}
    progress = True
    while progress:
        progress = False
        for srv in prog:
            while pos[srv] < len(prog[srv]):
                t,k,i = prog[srv][pos[srv]]; now = free[srv]
                if srv == 'A':
                    need = ev.get(('xt', i, t-1)); gh = ev.get(('gh1', i, t)) if t > 0 else 0.0
                    if need is None or gh is None: break
                    end = max(now + A_G_MFMA, need, gh) + A_G_BACK
                    ev[('x1', i, t)] = end + HOP; ev[('h1', i, t)] = end + HOP
                elif isinstance(srv, tuple) and k == 'g':
                    need = ev.get(('x1', i, t)); gh = ev.get(('gh2', i, t)) if t > 0 else 0.0
                    if need is None or gh is None: break
                    end = max(max(now, need) + B_G_MFMA, gh) + B_G_BACK
                    bdone[(srv, i, t)] = end + HOP
                    if all((('B', s), i, t) in bdone for s in range(G)):
                        v = max(bdone[(('B', s), i, t)] for s in range(G))
                        ev[('x2', i, t)] = v; ev[('h2', i, t)] = v
                elif isinstance(srv, tuple):
                    need = ev.get(('y2', i, t))
                    if need is None: break
                    end = max(now, need - look) + SAMPLE
                    ev[('xt', i, t)] = end + HOP; done[(i, t)] = end
                elif srv == 'AH' and k == 'h':
                    need = ev.get(('h1', i, t))
                    if need is None: break
                    end = max(now, need - look) + GH
                    ev[('gh1', i, t+1)] = end + 0.7
                elif srv == 'AH':
                    need = ev.get(('x2', i, t))
                    if need is None: break
                    end = max(now, need - look) + FC_MFMA + FC_BACK
                    ev[('y1', i, t)] = end + HOP
                elif srv == 'BH' and k == 'h':
                    need = ev.get(('h2', i, t))
                    if need is None: break
                    end = max(now, need - look) + GH
                    ev[('gh2', i, t+1)] = end + 0.7
                else:
                    need = ev.get(('y1', i, t))
                    if need is None: break
                    end = max(now, need - look) + FC_MFMA + FC_BACK
                    ev[('y2', i, t)] = end + 0.7
                free[srv] = end; pos[srv] += 1; progress = True
    a, b = steps//3, steps-2
    return (done[(0,b)] - done[(0,a)])/(b-a)

def sampler_early(s, G):
    """sample before the last gh stage (when the slot is not the last one)"""
    o = [('h', i) for i in range(G)]
    return o[:-1] + [('s', s), o[-1]] if G > 1 else o + [('s', s)]


def interleaved(G, lag):
    """gates of slot i followed by the fc stage of slot i - lag"""
    o = []
    for i in range(G + lag):
        if i < G:
            o.append(('g', i))
        if 0 <= i - lag < G:
            o.append(('f', i - lag))
    return o


if __name__ == '__main__':
    # (the kernel's order since the end of round 4 is the second column: the last slot's gh stage deferred behind the sampling stage;
    # measured: 24.8 -> 23.7 us at depth 4, 41.2 -> 40.5 at depth 8, profiles/r04p_probe.json)
    print('depth: us per step (model) for gh(0..n-1) | sample | the last gh stage behind the sampling stage | ih stages interleaved with lag 1, 2, 3 | fc stages on the hh workgroups, sampling on B-ih (not built)')
    for G in range(1, 9):
        base = simulate(G)
        early = simulate(G, s_order=sampler_early)
        inter = [simulate(G, a_order=interleaved(G, l), b_order=interleaved(G, l), s_order=sampler_early) if l < G or G == 1 else float('nan') for l in (1, 2, 3)]
        fc_first = simulate_v2(G, hh_order=lambda n: [('f', i) for i in range(n)] + [('h', i) for i in range(n)])
        print(f'{G}: {base:6.2f} | {early:6.2f} | ' + ' '.join(f'{x:6.2f}' for x in inter) + f' | {fc_first:6.2f}')
    # Round 5's reading of the last column: with four slots in flight a step is as much the latency of a slot's chain behind the other slots'
    # stages as it is the busiest workgroup's work (no single duration moves it: -20 % on any one of them is -0 .. -3 %), so re-balancing the
    # roles buys ~3 % at depth 4 (22.3 us with the best of the 70 interleavings of gh and fc stages) and 18 % at depth 8 -- not what the
    # 256-segment workload needs.
