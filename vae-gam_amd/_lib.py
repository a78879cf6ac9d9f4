"""ctypes binding of the C ABI in include/vaegam.h (libvaegam_hip.so).

The product path loads ONLY the hipcc-built library that sits next to this file and fails
loudly if it is missing -- there is no CPU fallback.  (`tests/emu` builds the same sources
for the host to debug index arithmetic; the test-side helper tests/emu_inject.py swaps that
handle in by monkeypatching this module, the product never looks for it.)
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST be imported before the kernel library is dlopen'ed: torch ships its own
#                      libamdhip64; loading ours first would put two HIP runtimes in the process and every
#                      stream / pointer handed across would be foreign ("no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = 'libvaegam_hip.so'

i32, i64, f32, f64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, i32) for n in ('N', 'CI', 'CO', 'ID', 'IH', 'IW', 'OD', 'OH', 'OW', 'KD', 'KH', 'KW',
                                   'stride', 'pad_d', 'pad_h', 'pad_w', 'relu_in', 'per_group')]


class WgradDesc(ctypes.Structure):
    _fields_ = [(n, i32) for n in ('N', 'CB', 'CA', 'PD', 'PH', 'PW', 'AD', 'AH', 'AW', 'KD', 'KH', 'KW',
                                   'stride', 'pad_d', 'pad_h', 'pad_w', 'pro_on_a', 'relu_in', 'per_group')]


class MmDesc(ctypes.Structure):
    _fields_ = [(n, i32) for n in ('N', 'CI', 'CO', 'ID', 'IH', 'IW', 'OD', 'OH', 'OW', 'nq')] + [('ks', i32 * 4)] + \
               [(n, i32) for n in ('PDT', 'PH', 'PW', 'PD', 'sdi', 'shi', 'swi', 'd0', 'LD', 'cc', 'sdo', 'sho', 'swo')] + \
               [('od0', i32 * 4), ('oh0', i32 * 4), ('ow0', i32 * 4)] + [(n, i32) for n in ('relu_in', 'per_group', 'tpc', 'slack', 'dbuf', 'PHB', 'hlo', 'hhi', 'waves')]


class GainDesc(ctypes.Structure):
    _fields_ = [('C', i32), ('B', i32), ('n', i32), ('hrf_taps', i32), ('jitter_b', f64), ('jitter_ku', f64), ('prior_var', f64)]


class FcDesc(ctypes.Structure):
    _fields_ = [('M', i32), ('N', i32), ('K', i32), ('batch', i32), ('a_sm', i64), ('a_sk', i64), ('a_sb', i64),
                ('b_sk', i64), ('b_sn', i64), ('b_sb', i64), ('c_sm', i64), ('c_sb', i64), ('bias_sb', i64), ('cx_sb', i64),
                ('ksplit', i32), ('flags', i32)]


class FcJob(ctypes.Structure):
    _fields_ = [('d', FcDesc), ('A', vp), ('amask', vp), ('B', vp), ('bias', vp), ('cmask', vp), ('C', vp), ('cx', vp), ('ws', vp)]


FC_A_RELU, FC_A_MASK, FC_B_RELU, FC_B_ONES, FC_C_BIAS, FC_C_RELU, FC_C_MASK, FC_C_ACCUM = 1, 2, 4, 8, 16, 32, 64, 128


_PROTOS = {
    'vg_version': (ctypes.c_int, []),
    'vg_last_error': (ctypes.c_char_p, []),
    'vg_corr3d': (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    'vg_tconv3d_s2': (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]),
    'vg_tconv3d_s2_stats_chunks': (i64, [ctypes.POINTER(ConvDesc), i32]),
    'vg_tconv3d_s2_stats': (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'vg_bn_stats_from_parts': (ctypes.c_int, [vp, i32, i32, i64, f64, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]),
    'vg_wgrad3d_ws_bytes': (i64, [ctypes.POINTER(WgradDesc)]),
    'vg_wgrad3d': (ctypes.c_int, [ctypes.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, i32, vp]),
    'vg_wgrad3d_grouped_ws_bytes': (i64, [ctypes.POINTER(WgradDesc)]),
    'vg_wgrad3d_grouped': (ctypes.c_int, [ctypes.POINTER(WgradDesc), vp, vp, vp, vp, vp, vp, vp]),
    'vg_bn_tconv1_sums': (ctypes.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp, i32, vp]),
    'vg_bn_ws_bytes': (i64, [i32, i32, i64, i32]),
    'vg_bn_stats': (ctypes.c_int, [vp, i32, i32, i64, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]),
    'vg_bn_finalize': (ctypes.c_int, [vp, i32, i32, vp, vp, f32, vp, vp, vp, vp, vp]),
    'vg_bn_bwd_reduce': (ctypes.c_int, [vp, vp, i32, i32, i64, i32, i32, vp, vp, vp, vp, vp]),
    'vg_bn_bwd_apply': (ctypes.c_int, [vp, vp, i32, i32, i64, i32, i32, vp, vp, vp, vp, f64, vp, vp, vp, vp, i32, vp]),
    'vg_bn_tconv1_ws_bytes': (i64, [i32, i32, i32, i32]),
    'vg_bn_bwd_reduce_tconv1': (ctypes.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    'vg_bn_bwd_apply_tconv1': (ctypes.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, f64, vp, vp, i32, vp]),
    'vg_bn_param_grad': (ctypes.c_int, [vp, i32, i32, vp, vp, i32, vp]),
    'vg_data_bn_nshift': (ctypes.c_int, [vp, vp, i32, vp, vp]),
    'vg_data_bn_grads': (ctypes.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    'vg_latent_fwd': (ctypes.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    'vg_latent_bwd': (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    'vg_loss_fwd': (ctypes.c_int, [vp, vp, vp, vp, i32, i32, f64, f64, f64, f64, vp, vp]),
    'vg_loss_bwd': (ctypes.c_int, [vp, i32, i32, f64, f64, f64, f64, vp, vp, vp, vp, vp]),
    'vg_channel_sum': (ctypes.c_int, [vp, i32, i32, i64, vp, vp, i32, vp]),
    'vg_gam_ws_bytes': (i64, [i32, i32, i64]),
    'vg_gam_elbo_fwd': (ctypes.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, vp, vp, vp, vp, vp]),
    'vg_gam_elbo_bwd': (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, vp, vp, vp, vp, vp, i32, vp]),
    'vg_pack_weights': (ctypes.c_int, [vp, vp, vp, i32, i64, vp]),
    'vg_cholesky_f64': (ctypes.c_int, [vp, vp, i32, i32, vp]),
    'vg_conv_mm_stats_chunks': (i64, [ctypes.POINTER(MmDesc), i32]),
    'vg_conv_mm': (ctypes.c_int, [ctypes.POINTER(MmDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    'vg_gather_f32': (ctypes.c_int, [vp, vp, vp, i64, vp]),
    'vg_fc_ws_bytes': (i64, [ctypes.POINTER(FcDesc)]),
    'vg_fc_gemm': (ctypes.c_int, [ctypes.POINTER(FcDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'vg_fc_gemm_jobs': (ctypes.c_int, [ctypes.POINTER(FcJob), i32, vp]),
    'vg_gp_gain_ws_bytes': (i64, [i32, i32, i32]),
    'vg_gp_gain_fwd': (ctypes.c_int, [ctypes.POINTER(GainDesc), vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'vg_gp_gain_bwd': (ctypes.c_int, [ctypes.POINTER(GainDesc), vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]),
    'vg_adam_advance': (ctypes.c_int, [vp, f64, f64, f64, vp]),
    'vg_adam_step': (ctypes.c_int, [vp, vp, vp, vp, i64, i32, f64, f64, f64, vp, vp]),
}
EXPORTS = tuple(_PROTOS)


class VgError(RuntimeError):
    pass


class VgLibrary:
    host_pointers_ok = False        # the hipcc-built library dereferences DEVICE pointers only

    def __init__(self, path):
        if not os.path.exists(path):
            raise VgError('HIP kernel library %s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % path)
        self.path = path
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(self.dll, name)           # AttributeError here = a declared symbol is not exported
            fn.restype, fn.argtypes = res, args

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise VgError('%s failed (status %d): %s' % (name, rc, self.dll.vg_last_error().decode()))

    def size(self, name, *args):
        n = getattr(self.dll, name)(*args)
        if n < 0:
            raise VgError('%s: unsupported configuration: %s' % (name, self.dll.vg_last_error().decode()))
        return int(n)


_LIB = None


def library_path():
    return os.path.join(_HERE, LIB_NAME)          # the one product library; nothing else is ever searched


def get_lib() -> VgLibrary:
    global _LIB
    if _LIB is None:
        _LIB = VgLibrary(library_path())
    return _LIB
