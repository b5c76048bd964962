#!/usr/bin/env python
"""Per-phase shader clocks of wrnn_duo_kernel (wrnn_options.phase_clocks): python scripts/gpu_duo_profile.py [--depth 8 --B 512]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--depth', type=int, default=8); ap.add_argument('--B', type=int, default=512); ap.add_argument('--T', type=int, default=480)
ap.add_argument('--tuning', type=int, default=0); ap.add_argument('--so', default=None); ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'duo_phase_clocks.json'))
a = ap.parse_args()
if a.so:
    import wavernn_amd._lib as _L
    _L.SO_PATH = os.path.abspath(a.so)
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode='MOL'), 'MOL', device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='duo', depth=a.depth, tuning=a.tuning)
plain = eng.last_loop_ms()
pc = torch.zeros(256, 32, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='duo', depth=a.depth, tuning=a.tuning, phase_clocks=pc)
prof_ms = eng.last_loop_ms()
v = pc.cpu().numpy().astype(np.float64)            # [block & 255][ih 0-15 | hh 16-31]
names = ['issue', 'barrier', 'back half', 'operand wait', 'build+lookahead', 'mfma+partials', 'stages', 'polled']
groups = a.B // 16 // 4
gs = a.T * groups                                   # group-steps per cluster = stages of a kind per workgroup
res = {'plain_ms': plain, 'profiled_ms': prof_ms, 'us_per_step': plain * 1e3 / a.T, 'depth': a.depth, 'B': a.B, 'T': a.T}
for name, lo, rows in (('A-ih gates', 0, slice(0, 256, 2)), ('A-ih fc', 8, slice(0, 256, 2)), ('B-ih gates', 0, slice(1, 256, 2)), ('B-ih fc', 8, slice(1, 256, 2)),
                       ('A-hh gh', 16, slice(0, 256, 2)), ('A-hh sample', 24, slice(0, 256, 2)), ('B-hh gh', 16, slice(1, 256, 2))):
    m = v[rows, lo:lo + 8].mean(axis=0)
    per = m[:6] / gs
    res[name] = {n: round(float(x), 1) for n, x in zip(names[:6], per)}
    res[name]['total cycles per group-step'] = round(float(per.sum()), 1)
    res[name]['polled fraction'] = round(float(m[7] / max(m[6], 1)), 4)
    res[name]['slot6 per group-step (DUO_PROF_SPLIT builds: cycles of hygiene + xi)'] = round(float(m[6] / gs), 1)
    print(f'{name:12s}', ' '.join(f'{n}={x:7.0f}' for n, x in zip(names[:6], per)), f'| total {per.sum():7.0f} cycles per group-step; polled {m[7] / max(m[6], 1):.3f}; slot6 {m[6] / gs:7.0f}')
print(json.dumps({k: res[k] for k in ('plain_ms', 'profiled_ms', 'us_per_step')}))
json.dump(res, open(a.out, 'w'), indent=1)
