"""`python -m wavernn_amd.gen_tacotron --input_text "..." --tts_weights tts.pyt --voc_weights voc.pyt` -- the `wavernn` vocoder
path of the reference's TTS CLI (gen_tacotron.py:97-166) on one MI355X: Tacotron (functional PyTorch-ROCm restatement with a
the decoder loop as one persistent HIP kernel, `tacotron.py` / csrc/wrnn_taco.hip; `_, m, attention = tts_model.generate(x)` :142 -- the vocoder is fed the SECOND return,
the postnet / post_proj output, 80 bins because fft_bins = hp.num_mels) -> (m + 4) / 8, clip (:143-145) -> the MI355X-native `WaveRNN.generate` (:161-163).

Flags follow the reference where they exist (--input_text/-i, --tts_weights, --batched/-b, --unbatched/-u, --target/-t,
--overlap/-o, --voc_weights); hparams.py is replaced by the shipped defaults.  Without --input_text every line of
--sentences (default: none) is synthesised.  Text cleaning is the basic pipeline (lowercase, whitespace): number and
abbreviation expansion need the reference's text front-end (out of scope)."""
import argparse
import time
from pathlib import Path

import numpy as np
import torch

from .model import WaveRNN
from .synthetic import SHIPPED
from .tacotron import TacotronInference, tacotron_to_wavernn_mel, text_to_ids


def main(argv=None):
    ap = argparse.ArgumentParser(description='TTS Generator (Tacotron -> WaveRNN) on MI355X')
    ap.add_argument('--input_text', '-i', type=str, help='[string] Type in something here and TTS will generate it!')
    ap.add_argument('--sentences', type=str, help='text file, one sentence per line (the reference reads sentences.txt)')
    ap.add_argument('--tts_weights', type=str, required=True, help='[string/path] reference Tacotron state dict (.pyt)')
    ap.add_argument('--voc_weights', type=str, required=True, help='[string/path] reference WaveRNN state dict (.pyt)')
    ap.add_argument('--batched', '-b', dest='batched', action='store_true')
    ap.add_argument('--unbatched', '-u', dest='batched', action='store_false')
    ap.add_argument('--target', '-t', type=int, default=11_000)
    ap.add_argument('--overlap', '-o', type=int, default=550)
    ap.add_argument('--mode', default='MOL', choices=['MOL', 'RAW'])
    ap.add_argument('--steps', type=int, default=2000, help='decoder step limit (Tacotron.generate default)')
    ap.add_argument('--output', default='.', help='output directory')
    ap.set_defaults(batched=True)
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('wavernn_amd needs a HIP device; there is no CPU path (use the reference for that)')
    voc = WaveRNN(**SHIPPED, mode=a.mode).to('cuda')
    voc.load(a.voc_weights)
    tts = TacotronInference(torch.load(a.tts_weights, map_location='cuda'), device='cuda')
    if a.input_text:
        texts = [a.input_text.strip()]
    elif a.sentences:
        texts = [ln.strip() for ln in open(a.sentences) if ln.strip()]
    else:
        raise SystemExit('give --input_text or --sentences')
    out_dir = Path(a.output)
    out_dir.mkdir(parents=True, exist_ok=True)
    v_type = 'wavernn_batched' if a.batched else 'wavernn_unbatched'
    tts_k = int(tts.p['step'].item()) // 1000 if 'step' in tts.p else 0
    for i, text in enumerate(texts, 1):
        print(f'\n| Generating {i}/{len(texts)}')
        t0 = time.perf_counter()
        _, mel, _ = tts.generate(text_to_ids(text), steps=a.steps, kernel=True)   # the POSTNET output (:142)
        t1 = time.perf_counter()
        m = torch.tensor(tacotron_to_wavernn_mel(mel)).unsqueeze(0)
        name = f'__input_{text[:10]}_{v_type}_{tts_k}k.wav' if a.input_text else f'{i}_{v_type}_{tts_k}k.wav'
        wav = voc.generate(m, out_dir / name, a.batched, a.target, a.overlap, True)
        t2 = time.perf_counter()
        print(f'{name}: {mel.shape[1]} frames, {wav.shape[0] / voc.sample_rate:.2f} s of audio; Tacotron {(t1 - t0) * 1e3:.0f} ms, '
              f'vocoder {(t2 - t1) * 1e3:.0f} ms ({voc.last_loop_kernel})')
    print('\n\nDone.\n')


if __name__ == '__main__':
    main()
