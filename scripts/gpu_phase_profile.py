#!/usr/bin/env python
"""Per-phase shader-clock breakdown of the role-split loop kernel (wrnn_options.phase_clocks, MOL) on one MI355X.

    python scripts/gpu_phase_profile.py [--B 64,128,256] [--T 1000] [--out gpurun_out/phase.json]

For every segment count: the un-instrumented loop time, the instrumented one, and per (role, phase) the mean shader cycles per
stage spent in: issue | barrier wait of the previous stage's back half | its pointwise + publish | load wait / poll | operand
build | MFMA + partial tiles, plus the fraction of stages whose first check still found a sentinel.
"""
import argparse, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--B', default='64,128,256')
ap.add_argument('--T', type=int, default=1000)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'phase.json'))
args = ap.parse_args()
dev = torch.device('cuda', 0)
hop, T = 275, args.T
eng = LoopEngine(random_state_dict(0, mode='MOL'), 'MOL', device=dev)
rs = np.random.RandomState(3)
NAMES = ['issue', 'barrier_wait', 'pointwise_publish', 'load_wait', 'operand_build', 'mfma_partials']
PHASES = {0: ['P1 ih1', 'P2 hh1', 'P3 fc1', 'P5 fc3'], 1: ['P2 ih2', 'P3 hh2', 'P4 fc2', 'P5 fc3']}
res = []
for B in [int(x) for x in args.B.split(',')]:
    stride = 64
    L = (B * stride + T + hop - 1) // hop * hop
    mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
    aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
    noise = torch.empty(T, 11 * B, device=dev).uniform_(1e-5, 1 - 1e-5)
    eng.run(mels_up, aux, B, T, stride, noise, hop)
    eng.run(mels_up, aux, B, T, stride, noise, hop)
    ms_plain, info = eng.last_loop_ms(), eng.last_run_info()
    clk = torch.zeros(256, 32, dtype=torch.int64, device=dev)
    eng.run(mels_up, aux, B, T, stride, noise, hop, phase_clocks=clk)
    ms_prof = eng.last_loop_ms()
    c = clk.cpu().numpy().astype(np.float64)
    nblk = info['clusters'] * 64
    per_xcd = nblk // 8
    xpc = 8 // info['clusters']
    row = dict(B=B, T=T, info=info, ms=round(ms_plain, 3), ms_instrumented=round(ms_prof, 3), us_per_step=round(ms_plain * 1e3 / T / max(info['rounds'], 1), 3),
               roles={})
    for role in (0, 1):
        blocks = [b for b in range(nblk) if (((b % 8) % xpc) * per_xcd + b // 8) % 2 == role]
        m = c[blocks].mean(axis=0).reshape(4, 8)
        stages = np.maximum(m[:, 6], 1.0)                      # polled stages (role A phase 0 polls nothing)
        nst = T * max(info['depth'], 1) * max(info['rounds'], 1)
        d = {}
        for ph in range(4):
            d[PHASES[role][ph]] = {n: round(m[ph, k] / nst, 1) for k, n in enumerate(NAMES)}
            d[PHASES[role][ph]]['first_check_missed'] = round(m[ph, 7] / stages[ph], 3)
        d['cycles_per_group_step'] = round(m[:, :6].sum() / nst, 1)
        row['roles']['A' if role == 0 else 'B'] = d
    res.append(row)
    print(json.dumps(row), flush=True)
json.dump(res, open(args.out, 'w'), indent=1)
