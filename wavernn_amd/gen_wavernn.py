"""`python -m wavernn_amd.gen_wavernn --file mel.npy --weights latest_weights.pyt` -- the `--file x.npy` path of the
reference's vocoder CLI (gen_wavernn.py:38-65, :68-142) on the MI355X-native generate path (`--file x.wav` goes through the
librosa-free mel front-end of dsp.py).

Same flags as the reference where they exist (--batched/-b, --unbatched/-u, --target/-t, --overlap/-o, --file/-f,
--weights/-w); the hparams.py machinery is replaced by flags with the shipped defaults (hparams.py:20-60).  Mel
input must be a (80, N) or (1, 80, N) float array in [0, 1] (gen_wavernn.py:50-55 raises ValueError otherwise)."""
import argparse
from pathlib import Path

import numpy as np
import torch

from .model import WaveRNN
from .synthetic import SHIPPED


def gen_from_file(model, load_path: Path, save_path: Path, batched, target, overlap, mu_law=True):
    """gen_wavernn.py:38-65 for a `.npy` mel."""
    suffix = load_path.suffix
    if suffix == '.wav':
        # gen_wavernn.py:44-47: the librosa front-end, restated without librosa (dsp.py; parity unpinned -- see its header)
        from . import dsp
        wav = dsp.load_wav(load_path, model.sample_rate)
        dsp.save_wav(wav, save_path / f'__{load_path.stem}__{model.get_step() // 1000}k_steps_target.wav', model.sample_rate)
        mel = dsp.melspectrogram(wav)
    elif suffix == '.npy':
        mel = np.load(load_path)
        if mel.ndim != 2 or mel.shape[0] != 80:
            raise ValueError(f'Expected a numpy array shaped (n_mels, n_hops), but got {mel.shape}!')
        _max, _min = mel.max(), mel.min()
        if _max >= 1.01 or _min <= -0.01:
            raise ValueError(f'Expected spectrogram range in [0,1] but was instead [{_min}, {_max}]')
    else:
        raise ValueError(f'Expected an extension of .wav or .npy, but got {suffix}!')
    mel = torch.tensor(mel).unsqueeze(0)
    batch_str = f'gen_batched_target{target}_overlap{overlap}' if batched else 'gen_NOT_BATCHED'
    save_str = save_path / f'__{load_path.stem}__{batch_str}.wav'
    out = model.generate(mel, save_str, batched, target, overlap, mu_law)
    return out, save_str


def main(argv=None):
    ap = argparse.ArgumentParser(description='Generate WaveRNN samples on MI355X')
    ap.add_argument('--batched', '-b', dest='batched', action='store_true')
    ap.add_argument('--unbatched', '-u', dest='batched', action='store_false')
    ap.add_argument('--target', '-t', type=int, default=11000)
    ap.add_argument('--overlap', '-o', type=int, default=550)
    ap.add_argument('--file', '-f', type=str, required=True, help='.npy mel spectrogram (80, N) in [0,1], or a .wav at the model sample rate')
    ap.add_argument('--weights', '-w', type=str, help='state-dict .pyt of the reference WaveRNN (random init if omitted)')
    ap.add_argument('--mode', default='MOL', choices=['MOL', 'RAW'])
    ap.add_argument('--output', default='.', help='output directory')
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--device-noise', action='store_true', help='draw sampling noise on the GPU (not comparable with a CPU run)')
    ap.set_defaults(batched=True)
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('wavernn_amd needs a HIP device; there is no CPU path (use the reference for that)')
    model = WaveRNN(**SHIPPED, mode=a.mode).to('cuda')
    if a.weights:
        model.load(a.weights)
    if a.device_noise:
        model.noise_source = 'device'
    if a.seed is not None:
        torch.manual_seed(a.seed)
    out, path = gen_from_file(model, Path(a.file).expanduser().resolve(), Path(a.output), a.batched, a.target, a.overlap)
    n = out.shape[0]
    print(f'{path}: {n} samples ({n / model.sample_rate:.2f} s), loop {model.last_loop_kernel} {model.last_loop_ms:.1f} ms '
          f'= {n / model.last_loop_ms:.1f} kHz')


if __name__ == '__main__':
    main()
