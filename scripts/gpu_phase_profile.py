#!/usr/bin/env python
"""Per-phase shader-clock breakdown of the pipelined loop kernel (WRNN_PROF=1 builds) on one MI355X.

    python scripts/gpu_phase_profile.py [--T 1500] [--out gpurun_out/phases.json]
"""
import argparse, json, os, sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict

PHASES = ['S1 barrier-in', 'S1 mfma', 'S1 barrier', 'S1 pointwise+publish', 'sweep (S2-S5)', 'barrier after sweep',
          'mfma (S2-S4)', 'barrier after mfma', 'pointwise+publish+gh (S2-S4) + cond loads', 'fc3 row dot (S5)',
          'S6 cI/noise issue', 'S6 logits poll', 'S6 barrier', 'S6 sample', 'S6 barrier2', 'S6 xi write']

ap = argparse.ArgumentParser()
ap.add_argument('--T', type=int, default=1500)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'phases.json'))
ap.add_argument('--cases', default='1:64,2:128,3:180,2:24')
args = ap.parse_args()
dev = torch.device('cuda', 0)
T, hop = args.T, 275
sd = random_state_dict(0, mode='MOL')
eng = LoopEngine(sd, 'MOL', device=dev)
rs = np.random.RandomState(3)
res = []
for case in args.cases.split(','):
    G, B = (int(x) for x in case.split(':'))
    stride = 64
    L = (B * stride + T + hop - 1) // hop * hop
    mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
    aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
    noise = torch.empty(T, 11 * B, device=dev).uniform_(1e-5, 1 - 1e-5)
    os.environ['WRNN_PIPE_G'] = str(G)
    os.environ.pop('WRNN_PROF', None)
    eng.run(mels_up, aux, B, T, stride, noise, hop, algo='pipe')
    ms_plain = eng.last_loop_ms()
    os.environ['WRNN_PROF'] = '1'
    eng.run(mels_up, aux, B, T, stride, noise, hop, algo='pipe')
    ms = eng.last_loop_ms()
    prof = eng.read_profile().astype(np.float64)
    os.environ.pop('WRNN_PROF', None)
    act = prof[prof.sum(1) > 0]
    tot = act.sum(1)
    ghz = tot.mean() / (ms * 1e-3) / 1e9
    u, ncl, depth = eng.last_loop_split()
    rows = 15 if depth >= 3 else 16
    groups = -(-B // rows)
    rounds = max(1, -(-groups // (ncl * depth)))
    gsteps = T * rounds * min(depth, max(1, groups // ncl))       # group-steps each workgroup executed (approx.)
    fc3 = act[act[:, 9] > 0]
    nof = act[act[:, 9] == 0]
    entry = dict(G=G, B=B, T=T, ms_plain=round(ms_plain, 3), ms_prof=round(ms, 3), workgroups=int(act.shape[0]),
                 effective_GHz=round(ghz, 3), us_per_step=round(ms * 1e3 / (T * rounds), 3), group_steps_per_wg=gsteps,
                 phases_us_per_group_step={n: round(float(act[:, k].mean()) / ghz / 1e3 / gsteps, 3) for k, n in enumerate(PHASES)},
                 fc3_wgs_sweep_us=round(float(fc3[:, 4].mean()) / ghz / 1e3 / gsteps, 3) if len(fc3) else None,
                 other_wgs_sweep_us=round(float(nof[:, 4].mean()) / ghz / 1e3 / gsteps, 3) if len(nof) else None,
                 other_wgs_logit_poll_us=round(float(nof[:, 11].mean()) / ghz / 1e3 / gsteps, 3) if len(nof) else None)
    res.append(entry)
    print(json.dumps(entry, indent=1), flush=True)
json.dump(res, open(args.out, 'w'), indent=1)
