"""ctypes binding of libwavernn_amd.so (the C ABI declared in include/wavernn_amd.h).

The library holds the HIP kernels; there is NO CPU fallback.  If the shared object is missing, or no HIP
device is present, the product path raises -- loudly -- instead of computing anything on the host.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SO_PATH = os.path.join(CSRC, 'libwavernn_amd.so')

WRNN_OK = 0
ERR_RESIDENCY = -6          # WRNN_ERR_RESIDENCY: the persistent grid cannot be co-resident on this device
MODE_RAW, MODE_MOL = 0, 1
ABI_VERSION = 9
ALGO_AUTO, ALGO_STREAM, ALGO_LOOP, ALGO_SPARSE, ALGO_DUO, ALGO_CHAIN, ALGO_OCTO = 0, 1, 2, 5, 6, 7, 8
ALGOS = {'auto': ALGO_AUTO, 'stream': ALGO_STREAM, 'loop': ALGO_LOOP, 'sparse': ALGO_SPARSE, 'duo': ALGO_DUO, 'chain': ALGO_CHAIN, 'octo': ALGO_OCTO}

#: every symbol include/wavernn_amd.h declares
EXPORTS = ['wrnn_last_error', 'wrnn_abi_version', 'wrnn_device_cus', 'wrnn_pack_create', 'wrnn_pack_destroy',
           'wrnn_pack_weight_bytes', 'wrnn_pack_sparse_blocks', 'wrnn_pack_sparse_fc_blocks', 'wrnn_workspace_bytes', 'wrnn_workspace_bytes_segments', 'wrnn_generate',
           'wrnn_generate_segments', 'wrnn_plan_segments', 'wrnn_status', 'wrnn_timer_create', 'wrnn_timer_destroy', 'wrnn_timer_ms',
           'wrnn_timer_launches', 'wrnn_debug_read_exchange', 'wrnn_selftest', 'wrnn_selftest_metric', 'wrnn_pre_create',
           'wrnn_pre_destroy', 'wrnn_pre_hop', 'wrnn_pre_workspace_bytes', 'wrnn_pre_upsample', 'wrnn_pre_upsample_rows', 'wrnn_pre_last_error',
           'wrnn_post_unfold', 'wrnn_post_last_error', 'wrnn_taco_workspace_bytes', 'wrnn_taco_decode', 'wrnn_taco_status',
           'wrnn_taco_last_error', 'wrnn_bigru']


class Weights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('rnn_dims', 'fc_dims', 'feat_dims', 'aux_dims', 'n_classes', 'mode')] + \
               [(n, ctypes.c_void_p) for n in ('I_w', 'I_b', 'w_ih1', 'w_hh1', 'b_ih1', 'b_hh1', 'w_ih2', 'w_hh2',
                                               'b_ih2', 'b_hh2', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'fc3_w', 'fc3_b')]


class PreWeights(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('feat_dims', 'compute_dims', 'res_out_dims', 'res_blocks', 'pad')] + \
               [('upsample_factors', ctypes.c_int32 * 3)] + \
               [(n, ctypes.c_void_p) for n in ('conv_in_w', 'bn_in', 'res_w', 'res_bn', 'conv_out_w', 'conv_out_b', 'up_w')]


TACO_WEIGHT_FIELDS = ('prenet_fc1_w', 'prenet_fc1_b', 'prenet_fc2_w', 'prenet_fc2_b', 'attn_rnn_w_ih', 'attn_rnn_w_hh', 'attn_rnn_b_ih',
                      'attn_rnn_b_hh', 'attn_W_w', 'attn_W_b', 'attn_conv_w', 'attn_L_w', 'attn_L_b', 'attn_v_w', 'rnn_input_w', 'rnn_input_b',
                      'rnn1_w_ih', 'rnn1_w_hh', 'rnn1_b_ih', 'rnn1_b_hh', 'rnn2_w_ih', 'rnn2_w_hh', 'rnn2_b_ih', 'rnn2_b_hh', 'mel_proj_w')


class TacoWeights(ctypes.Structure):
    """wrnn_taco_weights (the Tacotron decoder kernels)."""
    _fields_ = [('struct_bytes', ctypes.c_uint32)] + \
               [(n, ctypes.c_int32) for n in ('n_mels', 'prenet1', 'prenet2', 'decoder_dims', 'encoder_width', 'lstm_dims', 'attn_filters',
                                              'attn_kernel')] + \
               [(n, ctypes.c_void_p) for n in TACO_WEIGHT_FIELDS]


class TacoCall(ctypes.Structure):
    """wrnn_taco_call."""
    _fields_ = [('struct_bytes', ctypes.c_uint32), ('n', ctypes.c_int32), ('r', ctypes.c_int32), ('max_r', ctypes.c_int32),
                ('max_steps', ctypes.c_int32), ('stop_threshold', ctypes.c_float), ('seq', ctypes.c_void_p), ('seq_proj', ctypes.c_void_p),
                ('mel_out', ctypes.c_void_p), ('scores_out', ctypes.c_void_p), ('steps_done', ctypes.c_void_p),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t), ('stream', ctypes.c_void_p), ('variant', ctypes.c_int32)]


class BigruCall(ctypes.Structure):
    """wrnn_bigru_call."""
    _fields_ = [('struct_bytes', ctypes.c_uint32), ('T', ctypes.c_int32), ('hidden', ctypes.c_int32)] + \
               [(n, ctypes.c_void_p) for n in ('gi_fwd', 'gi_rev', 'w_hh_fwd', 'w_hh_rev', 'b_hh_fwd', 'b_hh_rev', 'out', 'stream')]


class Geometry(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('B', 'T', 'stride', 'L', 'hop', 'n_frames')]


class RunInfo(ctypes.Structure):
    _fields_ = [('kernel', ctypes.c_char_p)] + \
               [(n, ctypes.c_int32) for n in ('units_per_wg', 'clusters', 'depth', 'rounds', 'slab_steps', 'launches')]


class Options(ctypes.Structure):
    """wrnn_options (include/wavernn_amd.h): per-call options; nothing is read from the environment."""
    _fields_ = [(n, ctypes.c_int32) for n in ('struct_bytes', 'algo', 'depth', 'clusters', 'cond_valu', 'slab_steps', 't_begin',
                                              't_end', 'tuning')] + \
               [('force_x', ctypes.c_void_p), ('logits', ctypes.c_void_p), ('phase_clocks', ctypes.c_void_p), ('timer', ctypes.c_void_p),
                ('info', ctypes.POINTER(RunInfo)), ('progress', ctypes.c_void_p), ('progress_user', ctypes.c_void_p),
                ('mel_stage', ctypes.c_int32), ('mel_rows', ctypes.c_int32), ('mel_scale', ctypes.c_int32),
                ('mel_taps', ctypes.c_void_p), ('seg_moff', ctypes.c_void_p)]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_bytes = ctypes.sizeof(Options)


#: wrnn_options.progress: void (*)(int32 steps_done, int32 T, int32 n_segments, void *user)
PROGRESS_FN = ctypes.CFUNCTYPE(None, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p)


class WrnnError(RuntimeError):
    pass


class ResidencyError(WrnnError):
    """WRNN_ERR_RESIDENCY on the FIRST slice of a step-sliced `auto` run: the persistent grid is not co-resident right now.  The
    callers that slice (WaveRNN.generate, generate_corpus) catch it and redo the call unsliced on the stream kernel."""


def build(verbose=False):
    """Compile the HIP sources for gfx950 into csrc/libwavernn_amd.so (hipcc cross-compiles without a GPU)."""
    cmd = [os.path.join(CSRC, 'build.sh')]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise WrnnError('building libwavernn_amd.so failed')
    global _lib
    _lib = None
    return SO_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises WrnnError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise WrnnError(f'{SO_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                        f'(or wavernn_amd/csrc/build.sh).  There is no CPU fallback.')
    # torch bundles its own libamdhip64; load it FIRST so this library binds to the same HIP runtime instance
    # (two HIP runtimes in one process do not see each other's devices / pointers).
    import torch  # noqa: F401
    L = ctypes.CDLL(SO_PATH)
    L.wrnn_last_error.restype = ctypes.c_char_p
    L.wrnn_abi_version.restype = ctypes.c_int
    L.wrnn_device_cus.argtypes = [ctypes.c_int]
    L.wrnn_pack_create.argtypes = [ctypes.POINTER(Weights), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.wrnn_pack_destroy.argtypes = [ctypes.c_void_p]
    L.wrnn_pack_destroy.restype = None
    L.wrnn_pack_weight_bytes.argtypes = [ctypes.c_void_p]
    L.wrnn_pack_weight_bytes.restype = ctypes.c_size_t
    L.wrnn_pack_sparse_blocks.argtypes = [ctypes.c_void_p]
    L.wrnn_pack_sparse_fc_blocks.argtypes = [ctypes.c_void_p]
    L.wrnn_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.POINTER(Geometry), ctypes.POINTER(Options)]
    L.wrnn_workspace_bytes.restype = ctypes.c_size_t
    L.wrnn_generate.argtypes = [ctypes.c_void_p, ctypes.POINTER(Geometry), ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Options),
                                ctypes.c_void_p]
    L.wrnn_workspace_bytes_segments.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.POINTER(Options)]
    L.wrnn_workspace_bytes_segments.restype = ctypes.c_size_t
    L.wrnn_generate_segments.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(Options), ctypes.c_void_p]
    L.wrnn_plan_segments.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Options), ctypes.POINTER(RunInfo)]
    L.wrnn_timer_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.wrnn_timer_destroy.argtypes = [ctypes.c_void_p]
    L.wrnn_timer_destroy.restype = None
    L.wrnn_timer_ms.argtypes = [ctypes.c_void_p]
    L.wrnn_timer_ms.restype = ctypes.c_float
    L.wrnn_timer_launches.argtypes = [ctypes.c_void_p]
    L.wrnn_debug_read_exchange.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.POINTER(Options), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p]
    L.wrnn_pre_create.argtypes = [ctypes.POINTER(PreWeights), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.wrnn_pre_destroy.argtypes = [ctypes.c_void_p]
    L.wrnn_pre_destroy.restype = None
    L.wrnn_pre_hop.argtypes = [ctypes.c_void_p]
    L.wrnn_pre_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.wrnn_pre_workspace_bytes.restype = ctypes.c_size_t
    L.wrnn_pre_upsample.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.wrnn_pre_upsample_rows.argtypes = L.wrnn_pre_upsample.argtypes
    L.wrnn_pre_last_error.restype = ctypes.c_char_p
    L.wrnn_post_unfold.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                   ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.wrnn_post_last_error.restype = ctypes.c_char_p
    L.wrnn_taco_workspace_bytes.restype = ctypes.c_size_t
    L.wrnn_taco_decode.argtypes = [ctypes.c_int, ctypes.POINTER(TacoWeights), ctypes.POINTER(TacoCall)]
    L.wrnn_taco_status.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32 * 4), ctypes.c_void_p]
    L.wrnn_taco_last_error.restype = ctypes.c_char_p
    L.wrnn_bigru.argtypes = [ctypes.c_int, ctypes.POINTER(BigruCall)]
    L.wrnn_status.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.wrnn_selftest.argtypes = [ctypes.c_int, ctypes.c_int]
    L.wrnn_selftest_metric.restype = ctypes.c_float
    _lib = L
    return L


def check(rc, what):
    if rc != WRNN_OK:
        raise WrnnError(f'{what} failed (rc={rc}): {lib().wrnn_last_error().decode()}')
