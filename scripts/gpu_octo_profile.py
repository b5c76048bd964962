#!/usr/bin/env python
"""Per-phase shader clocks of wrnn_octo_kernel (wrnn_options.phase_clocks): python scripts/gpu_octo_profile.py [--depth 4 --B 256]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--depth', type=int, default=4); ap.add_argument('--B', type=int, default=256); ap.add_argument('--T', type=int, default=480)
ap.add_argument('--tuning', type=int, default=0); ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'octo_phase_clocks.json'))
a = ap.parse_args()
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode='MOL'), 'MOL', device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='octo', depth=a.depth, tuning=a.tuning)
plain = eng.last_loop_ms()
pc = torch.zeros(256, 32, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='octo', depth=a.depth, tuning=a.tuning, phase_clocks=pc)
prof_ms = eng.last_loop_ms()
v = pc.cpu().numpy().astype(np.float64)            # [block & 255][matrix wave 0: 0-7 | service wave 0: 16-31]
# block b runs on XCD b % 8; cluster = XCD pair; even XCD of a pair = rnn1, odd = rnn2
rnn1 = [b for b in range(256) if (b % 8) % 2 == 0]
rnn2 = [b for b in range(256) if (b % 8) % 2 == 1]
slots = a.B // 16 // 4
ss = a.T * slots                                    # slot-steps per cluster
res = {'plain_ms': plain, 'profiled_ms': prof_ms, 'us_per_step': plain * 1e3 / a.T, 'us_per_step_profiled': prof_ms * 1e3 / a.T, 'depth': a.depth, 'B': a.B, 'T': a.T}
mn = ['operand wait', 'lds read+check+request', 'buffer wait', 'mfma+partials (gates, gh)', 'mfma+partials (fc)']
sn = ['gates: wait block', 'gates: partial sums', 'gates: wait x / input word', 'gates: cell+publish', 'gh: wait block', 'gh: work', 'fc: wait block', 'fc: work',
      'pre-loads+loop', 'cI + drain', 'sample: wait y2', 'sample: rest', 're-arm']
for name, rows in (('rnn1', rnn1), ('rnn2', rnn2), ('rnn2 samplers', [b for b in rnn2 if b // 8 < slots]), ('rnn2 others', [b for b in rnn2 if b // 8 >= slots])):
    m = v[rows, 0:7].mean(axis=0)
    per = m[:5] / ss
    print(f'{name:14s} matrix : ' + ' | '.join(f'{n} {x:6.0f}' for n, x in zip(mn, per)) + f' | total {per.sum():6.0f} clocks per slot-step; polled {m[6] / max(m[5], 1):.3f} of {m[5] / ss:.1f} blocks')
    res[name + ' matrix'] = {n: round(float(x), 1) for n, x in zip(mn, per)}
    q = v[rows, 16:29].mean(axis=0) / ss
    print(f'{name:14s} service: ' + ' | '.join(f'{n} {x:6.0f}' for n, x in zip(sn, q)) + f' | total {q.sum():6.0f}')
    res[name + ' service'] = {n: round(float(x), 1) for n, x in zip(sn, q)}
print(json.dumps({k: res[k] for k in ('plain_ms', 'profiled_ms', 'us_per_step', 'us_per_step_profiled')}))
json.dump(res, open(a.out, 'w'), indent=1)
