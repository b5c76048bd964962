#!/usr/bin/env python
"""Time line of the two workgroups of ONE CU of wrnn_duo_kernel over three steps (profiling build, wrnn_options.tuning bit 14; csrc/wrnn_duo.hip "TRACE"):
    python scripts/gpu_duo_trace.py [--depth 8 --B 512 --J 5 --tuning 0]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--depth', type=int, default=8); ap.add_argument('--B', type=int, default=512); ap.add_argument('--T', type=int, default=600)
ap.add_argument('--J', type=int, default=5); ap.add_argument('--tuning', type=lambda v: int(v, 0), default=0); ap.add_argument('--so', default=None)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'duo_trace.txt'))
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
pc = torch.zeros(256 * 32 + 512 * 512, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='duo', depth=a.depth, tuning=a.tuning | 16384, phase_clocks=pc, slab_steps=a.T)
ms = eng.last_loop_ms()
v = pc.cpu().numpy()[256 * 32:].reshape(512, 512)
SEG = ['top/issue', 'barrier', 'back', 'operand', 'build', 'mfma', '', '']
rows = []
for role, b, kinds in (('A-ih', 8 * a.J, ('gates', 'fc')), ('A-hh', 8 * (32 + a.J), ('gh', 'fc/smp')), ('B-ih', 8 * a.J + 1, ('gates', 'fc')), ('B-hh', 8 * (32 + a.J) + 1, ('gh', 'fc/smp'))):
    n = int(v[b, 0])
    for e in v[b, 1:1 + n]:
        e = int(e) & ((1 << 64) - 1)
        rows.append((e >> 16, role, (e >> 8) & 15, kinds[(e & 255) >> 3], SEG[e & 7]))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(a.out, 'w') as f:
    f.write(f'# wrnn_duo_kernel trace: depth {a.depth}, B {a.B}, tuning {a.tuning:#x}, {ms * 1e3 / a.T:.2f} us per step (profiled); end-of-segment events, shader clocks from the first\n')
    last = {}
    for t, role, slot, kind, seg in rows:
        d = t - last.get(role, t)
        last[role] = t
        col = ['A-ih', 'A-hh', 'B-ih', 'B-hh'].index(role)
        f.write(f'{t - t0:8d} ' + ' ' * (30 * col) + f'{role} {kind}[{slot}] {seg} (+{d})\n')
print(open(a.out).read()[:6000])
