"""ctypes binding of oracle/wrnn_oracle.c -- TEST INFRASTRUCTURE (tests/, smoke(), bench cpu_baseline only)."""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libwrnn_oracle.so')


def build(force=False):
    src = os.path.join(_HERE, 'wrnn_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', '_build/libwrnn_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


class _W(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ('rnn_dims', 'fc_dims', 'feat_dims', 'aux_dims', 'n_classes')] + \
               [(n, ctypes.c_void_p) for n in ('I_w', 'I_b', 'w_ih1', 'w_hh1', 'b_ih1', 'b_hh1', 'w_ih2', 'w_hh2',
                                               'b_ih2', 'b_hh2', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'fc3_w', 'fc3_b')]


_KEYS = dict(I_w='I.weight', I_b='I.bias', w_ih1='rnn1.weight_ih_l0', w_hh1='rnn1.weight_hh_l0',
             b_ih1='rnn1.bias_ih_l0', b_hh1='rnn1.bias_hh_l0', w_ih2='rnn2.weight_ih_l0', w_hh2='rnn2.weight_hh_l0',
             b_ih2='rnn2.bias_ih_l0', b_hh2='rnn2.bias_hh_l0', fc1_w='fc1.weight', fc1_b='fc1.bias',
             fc2_w='fc2.weight', fc2_b='fc2.bias', fc3_w='fc3.weight', fc3_b='fc3.bias')


def loop(sd, mode, mels, aux, noise, want_logits=False, nthreads=0):
    """C twin of `wavernn_oracle.loop`.  Returns out (B,T) [and logits (T,B,C)].  nthreads = 0: the OpenMP default, capped at 64 threads -- the
    loop synchronises its threads ~10 times per step, so on a 256-thread host all threads are several times SLOWER than 64
    (bench.py's cpu_baseline leg measures 8 / 16 / 32 / 64); the result does not depend on the thread count (rows of a layer
    are split over threads, each row is summed by one thread in a fixed order)."""
    lib = ctypes.CDLL(build())
    if nthreads <= 0 and len(os.sched_getaffinity(0)) > 64:
        nthreads = 64            # (small hosts keep the OpenMP default, incl. whatever torch.set_num_threads / OMP_NUM_THREADS set)
    keep = {k: np.ascontiguousarray(sd[v], dtype=np.float32) for k, v in _KEYS.items()}
    w = _W()
    w.rnn_dims = keep['w_hh1'].shape[1]
    w.fc_dims = keep['fc1_w'].shape[0]
    w.aux_dims = aux.shape[2] // 4
    w.feat_dims = mels.shape[2]
    w.n_classes = keep['fc3_w'].shape[0]
    for k, a in keep.items():
        setattr(w, k, a.ctypes.data)
    B, T, _ = mels.shape
    mels = np.ascontiguousarray(mels, np.float32)
    aux = np.ascontiguousarray(aux, np.float32)
    if mode == 'MOL':
        n1 = np.ascontiguousarray(noise[0], np.float32); n2 = np.ascontiguousarray(noise[1], np.float32)
    else:
        n1 = np.ascontiguousarray(noise, np.float32); n2 = n1
    out = np.zeros((B, T), np.float32)
    lg = np.zeros((T, B, w.n_classes), np.float32) if want_logits else None
    rc = lib.wrnn_oracle_loop(ctypes.byref(w), 1 if mode == 'MOL' else 0, B, T, mels.ctypes.data_as(ctypes.c_void_p),
                              aux.ctypes.data_as(ctypes.c_void_p), n1.ctypes.data_as(ctypes.c_void_p),
                              n2.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                              lg.ctypes.data_as(ctypes.c_void_p) if want_logits else None, nthreads)
    if rc != 0:
        raise RuntimeError(f'wrnn_oracle_loop failed rc={rc}')
    return (out, lg) if want_logits else out
