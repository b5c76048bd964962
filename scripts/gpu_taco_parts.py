"""Where config 3's Tacotron stage spends its time: encoder, decoder kernel, post-net CBHG (conv bank / highways / biGRU), copies."""
import json, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from wavernn_amd.synthetic import random_tacotron_state_dict
from wavernn_amd.tacotron import TacotronInference, text_to_ids
dev = torch.device('cuda', 0)
shapes = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'tacotron_shapes.json')))
tts = TacotronInference(random_tacotron_state_dict(3, shapes), device=dev)
ids = text_to_ids('Scientists at the CERN laboratory say they have discovered a new particle.')
def T(f, reps=3):
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return r, round(dt * 1e3, 2)
out = {}
with torch.no_grad():
    (seq, seq_proj), out['encode_ms'] = T(lambda: tts.encode(ids))
    (mel, sc), out['decode_kernel_ms'] = T(lambda: tts._decode_kernel(seq, seq_proj, 800, 0))
    post, out['postnet_cbhg_ms'] = T(lambda: tts._cbhg(mel, 'postnet', tts._post_k))
    lin, out['post_proj_ms'] = T(lambda: F.linear(post, tts.p['post_proj.weight']).transpose(1, 2)[0])
    _, out['to_numpy_ms'] = T(lambda: (mel[0].cpu().numpy(), lin.cpu().numpy(), sc.cpu().numpy()))
    _, out['generate_ms'] = T(lambda: tts.generate(ids, steps=800, kernel=True))
    tts._bigru_kernel = True
    _, out['encode_bigru_kernel_ms'] = T(lambda: tts.encode(ids))
    _, out['postnet_cbhg_bigru_kernel_ms'] = T(lambda: tts._cbhg(mel, 'postnet', tts._post_k))
    tts._bigru_kernel = False
    # the biGRU of the post-net alone
    p = tts.p; q = 'postnet.rnn.'
    flat = [p[q + n_] for n_ in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse', 'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse')]
    x = torch.randn(1, 800, flat[0].shape[1], device=dev); hx = torch.zeros(2, 1, flat[1].shape[1], device=dev)
    _, out['postnet_bigru_800_ms'] = T(lambda: torch._VF.gru(x, hx, flat, True, 1, 0.0, False, True, True))
    out['gru_shapes'] = [list(f.shape) for f in flat[:2]]
print(json.dumps(out, indent=1))
