"""The three DSP helpers the generate path touches after the loop (reference utils/dsp.py:8-9,22-23,98-103)."""
import numpy as np
from .fold import decode_mu_law, label_2_float  # noqa: F401  (re-exported)


def save_wav(x, path, sample_rate):
    """Write a float32 WAV (what `librosa.output.write_wav(path, x.astype(np.float32), sr)` produced)."""
    from scipy.io import wavfile
    wavfile.write(str(path), int(sample_rate), np.asarray(x).astype(np.float32))
