#!/usr/bin/env python
"""The benchmarked workload (16 x 641-frame utterances = 256 segments x 12,100 steps through generate_corpus: HIP pre-loop kernels, slabs, the mel
formed in the loop or materialised) under a list of wrnn_options.tuning values, loop-kernel time per step -- what scripts/gpu_tuning_sweep.py
measures on synthetic conditioning, on the real path:   python scripts/gpu_corpus_ab.py --tunings 0,0x100000 [--mode RAW] [--utterances 16]"""
import argparse, sys, os, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--tunings', default='0'); ap.add_argument('--mode', default='MOL'); ap.add_argument('--utterances', type=int, default=16)
ap.add_argument('--mel', default='default,materialised'); ap.add_argument('--so', default=None); ap.add_argument('--prune', type=float, default=0.0); ap.add_argument('--prune-linear', action='store_true')
a = ap.parse_args()
if a.so:
    from wavernn_amd import _lib as _L
    _L.SO_PATH = os.path.abspath(a.so)
from wavernn_amd.model import WaveRNN
from wavernn_amd.batch import generate_corpus
from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
dev = torch.device('cuda', 0)
sd = random_state_dict(0, mode=a.mode)
if a.prune > 0:
    from wavernn_amd.prune import block_prune_state_dict
    sd, _ = block_prune_state_dict(sd, a.prune, (16, 1), linear=a.prune_linear)
m = WaveRNN(**SHIPPED, mode=a.mode); m.num_params = lambda *x, **k: 0
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True); m = m.to(dev).eval()
mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0) for u in range(a.utterances)]
ref = {}
for mel in a.mel.split(','):
    m.mel_in_loop = None if mel == 'default' else False
    for tun in [int(x, 0) for x in a.tunings.split(',')]:
        e = m._loop_engine()
        if not hasattr(e, '_orig_rs'):
            e._orig_rs = e.run_segments
        e.run_segments = (lambda *x, _t=tun, **k: e._orig_rs(*x, tuning=_t, **k))
        best = 1e9
        for _ in range(2):
            segs, plan = generate_corpus(m, mels, 11000, 550, True, [77 + u for u in range(a.utterances)], noise_source='device', finish='own', check=False, return_segments=True)
            torch.cuda.synchronize()
            best = min(best, e.last_loop_ms())
        o = np.asarray(segs, dtype=np.float32)
        ref.setdefault(mel, o)
        print(json.dumps(dict(mel=mel, tuning=tun, loop_ms=round(best, 2), us_per_step=round(best * 1e3 / plan.T / max(1, e.last_run_info()['rounds']), 3), kernel=e.last_run_info()['kernel'],
                              dev_vs_first=float(np.abs(o - ref[mel]).max()))), flush=True)
