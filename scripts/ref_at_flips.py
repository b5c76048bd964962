#!/usr/bin/env python
"""Where the GPU's RAW class indices part ways with the C oracle (scripts/gpu_raw_flips.py), what does the REFERENCE ITSELF say?
Runs the unmodified reference `WaveRNN.generate()` (PyTorch CPU, /root/reference: build container only) on those utterances and
compares its per-segment samples with the oracle's cached outputs.  usage: python scripts/ref_at_flips.py 34:2:7399 46:14:3015"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import make_golden as MG                      # (imports the reference; nothing is written into its tree)
import torch
from helpers import oracle_utterance

for spec in sys.argv[1:]:
    u, seg, step = (int(x) for x in spec.split(':'))
    sd_np = MG.random_state_dict(0, mode='RAW')
    model = MG.WaveRNN(**MG.SHIPPED, mode='RAW')
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}, strict=True)
    mel = MG.random_mel(1234 + u, 641)
    cap, real_stack = {}, torch.stack

    def stack(tensors, *a, **k):
        r = real_stack(tensors, *a, **k)
        cap['raw'] = r
        return r
    torch.stack = stack
    try:
        torch.manual_seed(77 + u)
        model.generate(torch.tensor(mel).unsqueeze(0), '/tmp/_flip.wav', True, 11000, 550, True)
    finally:
        torch.stack = real_stack
    raw = cap['raw'].transpose(0, 1).contiguous().numpy()
    ref = oracle_utterance('RAW', 0, 0.0, 1234 + u, 77 + u, 641, want_cond=False)['ref']
    d = np.argwhere(raw != ref)
    print(f'utterance {u}: reference vs C oracle: {len(d)} differing samples of {raw.size}; first {d[:3].tolist()}', flush=True)
    to_idx = lambda x: int(round((float(x) + 1) * 511 / 2))
    print(f'  at (segment {seg}, step {step}): reference class {to_idx(raw[seg, step])}, oracle class {to_idx(ref[seg, step])}', flush=True)
    np.save(f'/tmp/ref_raw_u{u}.npy', raw)
