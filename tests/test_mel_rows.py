"""SURVEY.md 8 row f1, last part -- the LAST up-sampling stage formed inside the loop (`wrnn_options.mel_stage = 1`, `engine.MelRows`):
the host-side derivation on the CPU.  The kernel replaces `Stretch2d(11)` + the 23-tap conv + the crop (reference
models/fatchord_version.py:73-80, :86-88) by three tap sums per phase applied to the three input rows the taps reach; here that
formula (tests/helpers.py, restating csrc/wrnn_abi.hip / wrnn_ring.h) is checked against the oracle's UpsampleNetwork -- positions, offsets
of concatenated utterances, arbitrary (trained) taps -- and the RAW goldens are re-run through the C oracle with a mel formed that way:
the class indices must not move.  The GPU side of the same statement is tests/test_gpu_parity.py::test_mel_rows_*."""
import numpy as np
import pytest

from helpers import CASES, load_case, mel_rows_formula, oracle_stage2_rows
from oracle import c_oracle as C, wavernn_oracle as O
from wavernn_amd.synthetic import random_mel, random_state_dict

HOP, PAD, F32 = 275, 2, np.float32


_stage2_rows = oracle_stage2_rows


def test_three_row_form_equals_stretch_conv_crop_for_concatenated_utterances():
    sd = dict(random_state_dict(5, mode='MOL'))
    rng = np.random.default_rng(3)
    sd['upsample.up_layers.5.weight'] = (rng.random((1, 1, 1, 23)) / 23).astype(F32)         # trained taps are not a box filter
    frames = [9, 5, 12]
    ups, rows = [], []
    for k, n in enumerate(frames):
        mel = random_mel(40 + k, n)
        m = O.pad_tensor(mel.T[None].astype(F32), PAD, 'both')[0].T
        ups.append(O.upsample_network(sd, m)[0])
        rows.append(_stage2_rows(sd, mel))
        assert ups[-1].shape == (n * HOP, 80) and rows[-1].shape == ((n + 2 * PAD) * 25, 80)
    ups, rows = np.concatenate(ups), np.concatenate(rows)
    # utterance k of the concatenation: un-cropped position = cropped position + indent * (2 k + 1)   (engine.MelRows.seg_off)
    indent = PAD * HOP
    off = np.concatenate([np.full(n * HOP, indent * (2 * k + 1)) for k, n in enumerate(frames)])
    got = mel_rows_formula(rows, sd['upsample.up_layers.5.weight'], np.arange(ups.shape[0]) + off)
    assert np.abs(got - ups).max() <= 1e-6 * max(1.0, np.abs(ups).max()), np.abs(got - ups).max()
    # a wrong utterance offset is not a rounding difference
    bad = mel_rows_formula(rows, sd['upsample.up_layers.5.weight'], np.arange(ups.shape[0]) + indent)
    assert np.abs(bad[frames[0] * HOP:] - ups[frames[0] * HOP:]).max() > 1e-3


@pytest.mark.parametrize('name', [c for c in CASES if c.startswith('raw')] + ['raw_batched_481f'])
def test_raw_goldens_do_not_move_when_the_last_stage_is_formed_from_rows(name):
    """The reference's RAW class indices with the mel of every step formed by the three-row form instead of the 23-tap sum (another
    float32 rounding of the same value, ~1e-7 relative): bit-identical on every golden, BASELINE config 2's full size included."""
    cfg, g = load_case(name)
    sd = random_state_dict(cfg['wseed'], mode=cfg['mode'])
    mel = random_mel(cfg['mseed'], cfg['frames'])
    C.build()
    rows = _stage2_rows(sd, mel)
    L = cfg['frames'] * HOP
    mels_up = mel_rows_formula(rows, sd['upsample.up_layers.5.weight'], np.arange(L) + PAD * HOP)
    m = O.pad_tensor(mel.T[None].astype(F32), PAD, 'both')[0].T
    ref_up, aux_up = O.upsample_network(sd, m)
    assert np.abs(mels_up - ref_up).max() <= 1e-6
    mels, aux = mels_up[None], aux_up[None]
    if cfg['batched']:
        mels, aux = O.fold_with_overlap(mels, cfg['target'], cfg['overlap']), O.fold_with_overlap(aux, cfg['target'], cfg['overlap'])
    B, T, _ = mels.shape
    raw = C.loop(sd, cfg['mode'], mels, aux, O.draw_noise(cfg['seed'], cfg['mode'], B, T))
    assert np.array_equal(raw, g['raw']), f'{np.count_nonzero(raw != g["raw"])} of {raw.size} class indices moved'
