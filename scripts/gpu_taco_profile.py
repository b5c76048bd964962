"""Per-layer shader clocks of the register-resident Tacotron decoder kernel (variant 3) and the decode time of each variant.
Usage (GPU box): python scripts/gpu_taco_profile.py [--steps 800]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
from wavernn_amd.synthetic import random_tacotron_state_dict
from wavernn_amd.tacotron import TacotronInference, text_to_ids
ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=800); ap.add_argument('--so', default=None); a = ap.parse_args()
if a.so:
    from wavernn_amd import _lib as _L
    _L.SO_PATH = os.path.abspath(a.so)
dev = torch.device('cuda', 0)
shapes = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'tacotron_shapes.json')))
tts = TacotronInference(random_tacotron_state_dict(3, shapes), device=dev)
ids = text_to_ids('Scientists at the CERN laboratory say they have discovered a new particle.')
with torch.no_grad():
    seq, seq_proj = tts.encode(ids)
out = {}
for variant in (1, 2, 3):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if seq is not None:
            mel, sc = tts._decode_kernel(seq, seq_proj, a.steps, variant)
        else:
            tts.generate(ids, steps=a.steps, kernel=True, kernel_variant=variant)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f'variant{variant}_ms'] = round(dt * 1e3, 2)
    out[f'variant{variant}_us_per_step'] = round(dt * 1e6 / a.steps, 2)
ws = tts._last_taco_ws
prof = ws[-192:].view(torch.int64).cpu().numpy()
steps = max(int(prof[20]), 1)
out['clocks_per_step'] = {f'L{k // 2 + 1}_{"wait" if k % 2 == 0 else "work"}': round(float(prof[k]) / steps, 1) for k in range(20)}
out['clocks_per_step_total'] = round(float(prof[:20].sum()) / steps, 1)
out['steps'] = steps
print(json.dumps(out, indent=1))
