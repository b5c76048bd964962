"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (CPU) in the build container.

The reference (fatchord/WaveRNN, mounted read-only at /root/reference) has no tests or golden vectors,
so the fixtures pinning `oracle/` are outputs of the reference's own `WaveRNN.generate()` under a
fixed seed.  /root/reference does not exist on the GPU box: this script runs only here, the .npz
fixtures are committed.  Shims: stub `librosa`, alias `np.cumproduct`,
configure `hparams`.  Nothing is written into the reference tree.

    python scripts/make_golden.py
"""
import os, sys, types
import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = os.environ.get('WRNN_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
lib = types.ModuleType('librosa'); lib.output = types.SimpleNamespace(write_wav=lambda *a, **k: None)
sys.modules['librosa'] = lib
if not hasattr(np, 'cumproduct'):
    np.cumproduct = np.cumprod
import torch
from utils import hparams as hp
hp.configure(os.path.join(REF, 'hparams.py'))
from models.fatchord_version import WaveRNN          # the reference implementation
WaveRNN.gen_display = lambda self, *a, **k: None

from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED

OUT = os.path.join(REPO, 'tests', 'golden')

CASES = [
    # name, mode, weight seed, mel seed, frames, batched, target, overlap, mu_law, sample seed
    dict(name='raw_unbatched_24f', mode='RAW', wseed=11, mseed=101, frames=24, batched=False, target=11000, overlap=550, mu_law=True, seed=77),
    dict(name='raw_batched_60f', mode='RAW', wseed=11, mseed=102, frames=60, batched=True, target=1100, overlap=55, mu_law=True, seed=78),
    dict(name='mol_unbatched_24f', mode='MOL', wseed=12, mseed=103, frames=24, batched=False, target=11000, overlap=550, mu_law=True, seed=79),
    dict(name='mol_batched_100f', mode='MOL', wseed=12, mseed=104, frames=100, batched=True, target=1100, overlap=55, mu_law=True, seed=80),
    dict(name='mol_batched_ragged_53f', mode='MOL', wseed=13, mseed=105, frames=53, batched=True, target=2000, overlap=100, mu_law=False, seed=81),
    # BASELINE config 2 at its stated inputs: weight seed 0, mel seed 1234, N=481 -> B=12, T=12100, sample seed 77
    dict(name='mol_batched_481f', mode='MOL', wseed=0, mseed=1234, frames=481, batched=True, target=11000, overlap=550, mu_law=True, seed=77),
    dict(name='raw_batched_481f', mode='RAW', wseed=0, mseed=1234, frames=481, batched=True, target=11000, overlap=550, mu_law=True, seed=77),
    # BASELINE config 3, vocoder side (gen_tacotron.py:139-166): the mel comes from the reference's own Tacotron.generate
    # (seeded random init, steps=800 -> (80,800)), rescaled (m+4)/8 and clipped (:144-145); L = 220,000 = 19*11,550 + 550
    # exactly -> B = 19 with NO padded fold.  The mel is stored in the fixture (Tacotron is upstream of the path).
    dict(name='mol_tacotron_800f', mode='MOL', wseed=0, mseed=None, frames=800, batched=True, target=11000, overlap=550, mu_law=True, seed=77,
         tts_seed=3),
    # Round 6: the two utterances of the 64-utterance RAW flip-rate measurement (scripts/gpu_raw_flips.py: weight seed 0, mel seed 1234 + u,
    # sample seed 77 + u, 641 frames -> 16 segments x 12,100 steps) in which a kernel's class indices parted ways with the C ORACLE -- u = 34
    # (segment 2, step 7,399: the oracle itself is the odd one out there) and u = 46 (segment 14, step 3,015).  `compact`: only the
    # reference's class indices are stored ([B, T] uint16; the float sample is 2 idx / 511 - 1).
    dict(name='raw_flip_u34', mode='RAW', wseed=0, mseed=1268, frames=641, batched=True, target=11000, overlap=550, mu_law=True, seed=111, compact=True),
    dict(name='raw_flip_u46', mode='RAW', wseed=0, mseed=1280, frames=641, batched=True, target=11000, overlap=550, mu_law=True, seed=123, compact=True),
]


def tacotron_mel(c):
    """gen_tacotron.py:97-145 with a seeded random-init Tacotron: first line of sentences.txt -> text_to_sequence ->
    Tacotron.generate(x, steps=frames) -> (m + 4) / 8, clip to [0, 1].  Stubs: unidecode, inflect."""
    import time
    un = types.ModuleType('unidecode'); un.unidecode = lambda s: s
    sys.modules.setdefault('unidecode', un)
    inf = types.ModuleType('inflect'); inf.engine = lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'number')
    sys.modules.setdefault('inflect', inf)
    from models.tacotron import Tacotron
    from utils.text.symbols import symbols
    from utils.text import text_to_sequence
    torch.manual_seed(c['tts_seed'])
    tts = Tacotron(embed_dims=hp.tts_embed_dims, num_chars=len(symbols), encoder_dims=hp.tts_encoder_dims,
                   decoder_dims=hp.tts_decoder_dims, n_mels=hp.num_mels, fft_bins=hp.num_mels, postnet_dims=hp.tts_postnet_dims,
                   encoder_K=hp.tts_encoder_K, lstm_dims=hp.tts_lstm_dims, postnet_K=hp.tts_postnet_K,
                   num_highways=hp.tts_num_highways, dropout=hp.tts_dropout, stop_threshold=hp.tts_stop_threshold)
    with open(os.path.join(REF, 'sentences.txt')) as f:
        x = text_to_sequence(f.readline().strip(), hp.tts_cleaner_names)
    t0 = time.perf_counter()
    with torch.no_grad():
        _, m, _ = tts.generate(x, steps=c['frames'])
    print(f'reference Tacotron.generate: {len(x)} symbols -> mel {m.shape} in {time.perf_counter() - t0:.1f} s (CPU, {torch.get_num_threads()} threads)')
    m = (m + 4) / 8
    np.clip(m, 0, 1, out=m)
    assert m.shape == (80, c['frames']), m.shape
    return np.ascontiguousarray(m, dtype=np.float32)


def run_case(c):
    sd_np = random_state_dict(c['wseed'], mode=c['mode'])
    model = WaveRNN(**SHIPPED, mode=c['mode'])
    missing = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}, strict=True)
    mel = tacotron_mel(c) if c['mseed'] is None else random_mel(c['mseed'], c['frames'])
    cap = {}
    real_stack = torch.stack

    def stack(tensors, *a, **k):
        r = real_stack(tensors, *a, **k)
        cap['raw'] = r
        return r

    # conditioning tensors the reference feeds the loop (for the upsample/fold oracle checks)
    real_up = model.upsample.forward

    def up(m):
        a, b = real_up(m)
        cap['mels_up'], cap['aux_up'] = a.detach().clone(), b.detach().clone()
        return a, b
    model.upsample.forward = up
    torch.stack = stack
    try:
        torch.manual_seed(c['seed'])
        import time
        t0 = time.perf_counter()
        out = model.generate(torch.tensor(mel).unsqueeze(0), '/tmp/_golden.wav', c['batched'], c['target'],
                             c['overlap'], c['mu_law'])
        c = dict(c, ref_cpu_seconds=round(time.perf_counter() - t0, 2), ref_cpu_threads=torch.get_num_threads())
    finally:
        torch.stack = real_stack
    raw = cap['raw'].transpose(0, 1).contiguous().numpy()       # (B,T) float32, pre-decode (:243)
    if c.get('compact'):
        idx = np.rint((raw.astype(np.float64) + 1.0) * 511.0 / 2.0).astype(np.uint16)
        assert np.array_equal(idx.astype(np.float32) * np.float32(2) / np.float32(511) - np.float32(1), raw)
        np.savez_compressed(os.path.join(OUT, c['name'] + '.npz'), config=np.array(repr(c)), cls=idx, n_classes=np.int64(512))
        print(c['name'], 'class indices', idx.shape, 'min', idx.min(), 'max', idx.max())
        return
    mels_up = cap['mels_up'][0].numpy()
    aux_up = cap['aux_up'][0].numpy()
    # keep fixtures small: conditioning is stored strided (every 97th upsampled sample) + full aux frames
    extra = {'mel': mel.astype(np.float32)} if c['mseed'] is None else {}
    np.savez_compressed(os.path.join(OUT, c['name'] + '.npz'), **extra,
                        config=np.array(repr(c)), out=out.astype(np.float64), raw=raw.astype(np.float32),
                        mels_up_strided=mels_up[::97].astype(np.float32), aux_up_strided=aux_up[::97].astype(np.float32),
                        L=np.int64(mels_up.shape[0]))
    print(c['name'], 'raw', raw.shape, 'out', out.shape, 'absmax', np.abs(out).max())


def rng_kats():
    """Known-answer vectors of torch's CPU generator for the RNG oracle."""
    torch.manual_seed(5)
    u = torch.empty(16).uniform_(0, 1).numpy()
    torch.manual_seed(5)
    u2 = torch.empty(16).uniform_(1e-5, 1 - 1e-5).numpy()
    torch.manual_seed(5)
    e = torch.empty(16).exponential_(1).numpy()
    torch.manual_seed(9)
    _ = torch.nn.GRUCell(512, 512); _ = torch.nn.GRUCell(544, 512)
    after = torch.empty(8).uniform_(1e-5, 1 - 1e-5).numpy()
    torch.manual_seed(21)
    mix = [torch.empty(1, 3, 10).uniform_(1e-5, 1 - 1e-5).numpy().ravel(), torch.empty(1, 3).uniform_(1e-5, 1 - 1e-5).numpy().ravel(),
           torch.empty(1, 3, 10).uniform_(1e-5, 1 - 1e-5).numpy().ravel(), torch.empty(1, 3).uniform_(1e-5, 1 - 1e-5).numpy().ravel()]
    torch.manual_seed(22)
    ex = torch.empty(3, 512).exponential_(1).numpy()
    np.savez_compressed(os.path.join(OUT, 'rng_kats.npz'), uniform01_seed5=u, uniform_mol_seed5=u2, exp_seed5=e,
                        after_grucell_ctors_seed9=after, mol_two_steps_seed21=np.concatenate(mix), exp_3x512_seed22=ex)
    print('rng kats written')


def tacotron_shapes():
    """Key / shape / dtype table of the reference Tacotron's state dict (hparams.py tts_*): lets the GPU box build a random-init
    Tacotron state dict of the right architecture without the reference (tests/test_gpu_config3.py)."""
    import json
    c = dict(mseed=None, tts_seed=3, frames=1)
    un = types.ModuleType('unidecode'); un.unidecode = lambda s: s
    sys.modules.setdefault('unidecode', un)
    inf = types.ModuleType('inflect'); inf.engine = lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'number')
    sys.modules.setdefault('inflect', inf)
    from models.tacotron import Tacotron
    from utils.text.symbols import symbols
    tts = Tacotron(embed_dims=hp.tts_embed_dims, num_chars=len(symbols), encoder_dims=hp.tts_encoder_dims,
                   decoder_dims=hp.tts_decoder_dims, n_mels=hp.num_mels, fft_bins=hp.num_mels, postnet_dims=hp.tts_postnet_dims,
                   encoder_K=hp.tts_encoder_K, lstm_dims=hp.tts_lstm_dims, postnet_K=hp.tts_postnet_K,
                   num_highways=hp.tts_num_highways, dropout=hp.tts_dropout, stop_threshold=hp.tts_stop_threshold)
    table = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in tts.state_dict().items()]
    json.dump(table, open(os.path.join(OUT, 'tacotron_shapes.json'), 'w'))
    print('tacotron shapes written:', len(table), 'tensors')


def tacotron_decoder_golden(steps=200, wseed=3):
    """What the reference's own `Tacotron.generate` (models/tacotron.py:370-430) returns for the weights the GPU tests use:
    `random_tacotron_state_dict(3, shapes)` loaded into the REFERENCE class, first line of sentences.txt, `steps` decoder steps.
    tests/test_gpu_config3.py compares both decoder kernels and the CBHG GRU kernel's post-net to these arrays on the GPU box
    (SURVEY.md section 8 row f3); tests/test_oracle_golden.py compares the CPU mirror to them here."""
    import json
    un = types.ModuleType('unidecode'); un.unidecode = lambda s: s
    sys.modules.setdefault('unidecode', un)
    inf = types.ModuleType('inflect'); inf.engine = lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'number')
    sys.modules.setdefault('inflect', inf)
    from models.tacotron import Tacotron
    from utils.text.symbols import symbols
    from utils.text import text_to_sequence
    from wavernn_amd.synthetic import random_tacotron_state_dict
    tts = Tacotron(embed_dims=hp.tts_embed_dims, num_chars=len(symbols), encoder_dims=hp.tts_encoder_dims,
                   decoder_dims=hp.tts_decoder_dims, n_mels=hp.num_mels, fft_bins=hp.num_mels, postnet_dims=hp.tts_postnet_dims,
                   encoder_K=hp.tts_encoder_K, lstm_dims=hp.tts_lstm_dims, postnet_K=hp.tts_postnet_K,
                   num_highways=hp.tts_num_highways, dropout=hp.tts_dropout, stop_threshold=hp.tts_stop_threshold)
    shapes = json.load(open(os.path.join(OUT, 'tacotron_shapes.json')))
    sd = {k: torch.as_tensor(np.array(v)) for k, v in random_tacotron_state_dict(wseed, shapes).items()}
    tts.load_state_dict(sd, strict=True)
    with open(os.path.join(REF, 'sentences.txt')) as f:
        ids = text_to_sequence(f.readline().strip(), hp.tts_cleaner_names)
    with torch.no_grad():
        mel, lin, attn = tts.generate(ids, steps=steps)
    mel, lin, attn = np.asarray(mel, np.float32), np.asarray(lin, np.float32), np.asarray(attn, np.float32)
    assert mel.shape == (80, steps) and lin.shape == (80, steps) and attn.shape == (steps, len(ids)), (mel.shape, lin.shape, attn.shape)
    np.savez_compressed(os.path.join(OUT, f'tacotron_decoder_{steps}f.npz'), mel=mel, linear=lin, attention=attn, ids=np.array(ids, np.int32),
                        config=np.array(repr(dict(wseed=wseed, steps=steps, source='reference models/tacotron.py Tacotron.generate, CPU'))))
    print('tacotron decoder golden:', mel.shape, lin.shape, attn.shape, 'mel absmax', float(np.abs(mel).max()))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    rng_kats()
    tacotron_shapes()
    only = sys.argv[1:]
    if not only or 'tacotron_decoder_200f' in only:
        tacotron_decoder_golden()
    for c in CASES:
        if only and c['name'] not in only:
            continue
        run_case(c)
