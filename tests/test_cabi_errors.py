"""CPU tests of the C ABI's error behaviour (include/wavernn_amd.h): every entry point fails with a negative code and a
message -- it never crashes, never computes on the host -- when arguments are bad or no HIP device is present."""
import ctypes

import numpy as np
import pytest
import torch

from wavernn_amd import _lib
from wavernn_amd.engine import LOOP_KEYS
from wavernn_amd.synthetic import random_state_dict

ERR_ARG, ERR_NO_DEVICE = -1, -2


@pytest.fixture(scope='module')
def L():
    return _lib.lib()


def _weights(sd, kind, **over):
    host = {k: np.ascontiguousarray(sd[v], dtype=np.float32) for k, v in LOOP_KEYS.items()}
    w = _lib.Weights()
    w.rnn_dims, w.fc_dims, w.feat_dims, w.aux_dims = 512, 512, 80, 32
    w.n_classes = host['fc3_w'].shape[0]
    w.mode = _lib.MODE_MOL if kind == 'MOL' else _lib.MODE_RAW
    for k, a in host.items():
        setattr(w, k, a.ctypes.data)
    for k, v in over.items():
        setattr(w, k, v)
    return w, host


def test_pack_create_argument_checks(L):
    sd = random_state_dict(1, mode='MOL')
    pack = ctypes.c_void_p()
    assert L.wrnn_pack_create(None, 0, ctypes.byref(pack)) == ERR_ARG
    for over, needle in ((dict(rnn_dims=4096), b'unsupported dims'), (dict(n_classes=31), b'MOL needs n_classes == 30'),
                         (dict(mode=7), b'unknown mode'), (dict(fc2_w=None), b'NULL weight pointer')):
        w, keep = _weights(sd, 'MOL', **over)
        assert L.wrnn_pack_create(ctypes.byref(w), 0, ctypes.byref(pack)) == ERR_ARG
        assert needle in L.wrnn_last_error(), L.wrnn_last_error()
    w, keep = _weights(random_state_dict(1, mode='RAW'), 'RAW', n_classes=1)
    assert L.wrnn_pack_create(ctypes.byref(w), 0, ctypes.byref(pack)) == ERR_ARG


@pytest.mark.skipif(torch.cuda.is_available(), reason='needs a box WITHOUT a GPU')
def test_no_device_is_an_error_not_a_fallback(L):
    sd = random_state_dict(1, mode='MOL')
    w, keep = _weights(sd, 'MOL')
    pack = ctypes.c_void_p()
    assert L.wrnn_pack_create(ctypes.byref(w), 0, ctypes.byref(pack)) == ERR_NO_DEVICE
    assert b'no HIP device' in L.wrnn_last_error()
    assert L.wrnn_selftest(0, 1) == ERR_NO_DEVICE
    pw = _lib.PreWeights()
    pw.feat_dims, pw.compute_dims, pw.res_out_dims, pw.res_blocks, pw.pad = 80, 128, 128, 0, 2
    for i, s in enumerate((5, 5, 11)):
        pw.upsample_factors[i] = s
    z = np.zeros(128 * 400, np.float32)
    for n in ('conv_in_w', 'bn_in', 'res_w', 'res_bn', 'conv_out_w', 'conv_out_b', 'up_w'):
        setattr(pw, n, z.ctypes.data)
    pre = ctypes.c_void_p()
    assert L.wrnn_pre_create(ctypes.byref(pw), 0, ctypes.byref(pre)) == ERR_NO_DEVICE
    pw.compute_dims = 64                                    # (round 5: any UpsampleNetwork dims are taken -- the next check is the device)
    assert L.wrnn_pre_create(ctypes.byref(pw), 0, ctypes.byref(pre)) == ERR_NO_DEVICE
    pw.compute_dims = 0
    assert L.wrnn_pre_create(ctypes.byref(pw), 0, ctypes.byref(pre)) == ERR_ARG
    assert b'compute_dims' in L.wrnn_pre_last_error()
    pw.compute_dims, pw.feat_dims = 4096, 4096              # not even one frame of such a network fits a workgroup's LDS
    assert L.wrnn_pre_create(ctypes.byref(pw), 0, ctypes.byref(pre)) == ERR_ARG
    assert b'too wide' in L.wrnn_pre_last_error()


def test_null_handles_are_rejected(L):
    g = _lib.Geometry(1, 10, 0, 275, 275, 1)
    assert L.wrnn_workspace_bytes(None, ctypes.byref(g), None) == 0
    assert L.wrnn_workspace_bytes_segments(None, 1, 10, 1, None) == 0
    assert L.wrnn_pack_weight_bytes(None) == 0
    assert L.wrnn_generate(None, ctypes.byref(g), None, None, None, None, None, 0, None, None) == ERR_ARG
    assert L.wrnn_generate_segments(None, 1, 10, None, None, 275, 275, 1, None, None, None, None, None, 0, None, None) == ERR_ARG
    assert L.wrnn_status(None, None) == ERR_ARG
    assert L.wrnn_timer_ms(None) < 0 and L.wrnn_timer_launches(None) == 0
    assert L.wrnn_plan_segments(None, 1, 10, None, None) == ERR_ARG
    assert L.wrnn_debug_read_exchange(None, None, 1, 10, 1, None, 0, 0, 0, 0, None) == ERR_ARG
    L.wrnn_timer_destroy(None)
    assert L.wrnn_pre_hop(None) == 0 and L.wrnn_pre_workspace_bytes(None, 10) == 0
    assert L.wrnn_pre_upsample(None, None, 10, None, None, None, 0, None) == ERR_ARG
    L.wrnn_pack_destroy(None)
    L.wrnn_pre_destroy(None)


def test_python_wrapper_mirrors_reference_errors():
    """model-level error behaviour the reference's callers rely on."""
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import SHIPPED
    m = WaveRNN(**SHIPPED, mode='MOL')
    m.num_params(print_out=False)
    with pytest.raises(RuntimeError):                      # no CPU path: loud failure, not a silent fallback
        m.generate(torch.rand(1, 80, 30), '/tmp/x.wav', True, 1100, 55, True)
    with pytest.raises(AttributeError):                    # reference quirk (:104): the RuntimeError is built, not raised;
        WaveRNN(**SHIPPED, mode='XYZ')                     # construction then dies on the missing n_classes, as upstream
    from wavernn_amd import fold as F
    with pytest.raises(ValueError):                        # wave_len < 20*hop (reference :258)
        F.finish_waveform(np.zeros(19 * 275), 19 * 275, 275)
