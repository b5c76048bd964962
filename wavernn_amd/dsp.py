"""Audio <-> feature helpers either side of the vocoder call (reference utils/dsp.py).

`save_wav` (utils/dsp.py:22-23) is on the hot path's tail and writes float32 WAV through scipy (the reference uses
librosa.output.write_wav, which does the same).  `decode_mu_law` lives in fold.py.

The `.wav` -> mel front-end of `gen_wavernn.py --file x.wav` (utils/dsp.py:18-19 `load_wav`, :70-73 `melspectrogram`, :76-79
`stft`, :38-40 `linear_to_mel`, :47-56 dB / normalise) is librosa code (librosa==0.6.3 in the reference's requirements.txt,
NOT installed here and not vendored in /root/reference).  It is restated below from librosa's published algorithm (centered,
reflect-padded STFT with a periodic Hann window of `win_length` zero-padded to `n_fft`; Slaney mel filter bank, area-normalised,
`fmax = sr/2`; `melspectrogram(S=|D|)` = `mel_basis @ |D|`).  PARITY UNPINNED: there is no reference output or golden vector
for this stage in the reference tree, and the library that defines it cannot be run here; the tests check its defining
properties only (tests/test_host_logic.py).  It is upstream of the path: `generate()` takes the mel.
"""
import numpy as np
from scipy.io import wavfile

from .fold import decode_mu_law, label_2_float  # noqa: F401  (re-exported: utils/dsp.py:8-9, :98-103)

#: the reference's DSP settings (hparams.py:20-30)
HP = dict(sample_rate=22050, n_fft=2048, num_mels=80, hop_length=275, win_length=1100, fmin=40, min_level_db=-100, ref_level_db=20)


def save_wav(x, path, sample_rate=22050):
    wavfile.write(str(path), int(sample_rate), np.asarray(x).astype(np.float32))


def load_wav(path, sample_rate=22050):
    """utils/dsp.py:18-19 for a file that already has the model's sample rate (librosa.load would resample): mono float32 in [-1, 1]."""
    sr, x = wavfile.read(str(path))
    if sr != sample_rate:
        raise ValueError(f'{path}: sample rate {sr}, expected {sample_rate} (resampling needs the reference\'s librosa front-end)')
    if x.dtype.kind in 'iu':
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    x = x.astype(np.float32)
    return x.mean(axis=1) if x.ndim == 2 else x


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sample_rate=22050, n_fft=2048, n_mels=80, fmin=40, fmax=None):
    """Slaney-style triangular mel filter bank, each filter normalised to unit area (librosa.filters.mel, htk=False, norm=1)."""
    fmax = sample_rate / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sample_rate / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]).astype(np.float32)


def stft(y, n_fft=2048, hop_length=275, win_length=1100):
    """Centered STFT (reflect padding of n_fft // 2), periodic Hann window of win_length centred in the n_fft frame."""
    y = np.asarray(y, dtype=np.float32)
    win = np.zeros(n_fft, np.float32)
    lo = (n_fft - win_length) // 2
    win[lo:lo + win_length] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win_length) / win_length)
    yp = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.fft.rfft(win[:, None] * yp[idx], axis=0).astype(np.complex64)


def melspectrogram(y, hp=HP):
    """utils/dsp.py:70-73: normalise(amp_to_db(mel_basis @ |stft(y)|)) -> (num_mels, n_frames) in [0, 1]."""
    D = np.abs(stft(y, hp['n_fft'], hp['hop_length'], hp['win_length']))
    S = mel_basis(hp['sample_rate'], hp['n_fft'], hp['num_mels'], hp['fmin']) @ D
    S = 20 * np.log10(np.maximum(1e-5, S))
    return np.clip((S - hp['min_level_db']) / -hp['min_level_db'], 0, 1).astype(np.float32)
