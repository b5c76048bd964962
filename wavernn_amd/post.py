"""Post-loop stage on the device (`wrnn_post_unfold`, include/wavernn_amd.h): float64 gather cast, mu-law expansion,
cross-fade + overlap-add and tail fade of `WaveRNN.generate()` (reference models/fatchord_version.py:245-258, :342-405,
utils/dsp.py:98-103) for any number of utterances in one launch.

Bit-exact with the reference by construction: the transcendental parts (pow of `decode_mu_law`, sqrt / linspace of the
fades) are evaluated here with the reference's numpy expressions and uploaded as float64 tables; the kernel only gathers,
multiplies and adds doubles in the reference's order."""
import numpy as np
import torch

from . import _lib
from . import fold as _fold

_tables = {}


def _fade_tables(overlap, hop, device):
    key = ('fade', overlap, hop, str(device))
    if key not in _tables:
        silence_len = overlap // 2
        fade_len = overlap - silence_len
        t = np.linspace(-1, 1, fade_len, dtype=np.float64)                       # reference :381-392
        fade_in = np.concatenate([np.zeros(silence_len, np.float64), np.sqrt(0.5 * (1 + t))])
        fade_out = np.concatenate([np.ones(silence_len, np.float64), np.sqrt(0.5 * (1 - t))])
        tail = np.linspace(1, 0, 20 * hop)                                       # reference :256
        _tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(x)).to(device) for x in (fade_in, fade_out, tail))
    return _tables[key]


def _mu_law_lut(n_classes, device):
    key = ('lut', n_classes, str(device))
    if key not in _tables:
        idx = np.arange(n_classes, dtype=np.float32)
        y32 = np.float32(2.0) * idx / np.float32(n_classes - 1) - np.float32(1.0)    # the loop's float32 sample (:237)
        lut = _fold.decode_mu_law(y32.astype(np.float64), n_classes, False)           # utils/dsp.py:98-103, float64
        _tables[key] = torch.from_numpy(np.ascontiguousarray(lut, dtype=np.float64)).to(device)
    return _tables[key]


def unfold_on_device(segments, first, folds, wave_lens, overlap, hop, n_classes, mu_law, batched=True):
    """segments: float32 CUDA (n, T).  Utterance u owns rows [first[u], first[u]+folds[u]).  Returns a float64 CUDA tensor
    of all waveforms concatenated and the list of (start, end) slices.  Raises ValueError where the reference's tail
    fade would (wave_len < 20*hop, :258)."""
    dev = segments.device
    tail_len = 20 * hop
    if min(wave_lens) < tail_len:
        raise ValueError(f'operands could not be broadcast together with shapes ({min(wave_lens)},) ({tail_len},)')
    L = _lib.lib()
    n_utt, T = len(wave_lens), int(segments.shape[1])
    off = np.concatenate([[0], np.cumsum(wave_lens)]).astype(np.int64)
    tabs = torch.from_numpy(np.concatenate([np.asarray(first, np.int32), np.asarray(folds, np.int32)])).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    fade_in, fade_out, tail = _fade_tables(overlap if batched else 0, hop, dev)
    lut = _mu_law_lut(n_classes, dev) if mu_law else None
    out = torch.empty(int(off[-1]), dtype=torch.float64, device=dev)
    segments = segments.contiguous()
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = L.wrnn_post_unfold(segments.data_ptr(), T, n_utt, tabs.data_ptr(), tabs.data_ptr() + 4 * n_utt, d_off.data_ptr(),
                            lut.data_ptr() if lut is not None else None, n_classes, fade_in.data_ptr(), fade_out.data_ptr(),
                            overlap if batched else 0, tail.data_ptr(), tail_len, 1 if batched else 0, out.data_ptr(), stream)
    if rc != 0:
        raise _lib.WrnnError(f'wrnn_post_unfold failed (rc={rc}): {L.wrnn_post_last_error().decode()}')
    return out, [(int(off[u]), int(off[u + 1])) for u in range(n_utt)]
