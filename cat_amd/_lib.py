"""ctypes binding of libcat_hip.so (include/cat_hip.h).

The product path has NO fallback: if the shared object is missing the import raises, and every wrapper raises
RuntimeError with cat_hip_last_error() when a kernel entry point reports a failure."""
import ctypes as C
import os

from . import _build

c_f = C.c_float
c_i = C.c_int
c_l = C.c_int64
c_p = C.c_void_p
c_d = C.c_double


class ConvGeom(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'Cin', 'xcs', 'Ho', 'Wo', 'Cout', 'ycs', 'kh', 'kw', 'stride', 'pad',
                                   'pad_mode', 'act')] + [('slope', c_f), ('ycw', c_i), ('wcs', c_i)]


class WgradItem(C.Structure):
    _fields_ = [('g', ConvGeom), ('x', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p), ('accumulate', c_i)]


WGRAD_BATCH_MAX = 8


class NormGeom(C.Structure):
    _fields_ = [('N', c_i), ('HW', c_i), ('C', c_i), ('cs', c_i), ('mode', c_i), ('eps', c_f), ('momentum', c_f),
                ('act', c_i), ('slope', c_f)]


TCONV_MAXSEG = 8


class TSeg(C.Structure):
    _fields_ = [('src', c_p), ('scale', c_p), ('shift', c_p), ('sstride', c_i), ('xcs', c_i), ('c4', c_i), ('cin', c_i), ('ks', c_i), ('padv', c_i), ('act', c_i),
                ('slope', c_f), ('reflect', c_i), ('pack_off', c_i)]


class TConv(C.Structure):
    _fields_ = [('res', c_p), ('stats', c_p)] + [(n, c_i) for n in ('rcs', 'scs', 'N', 'H', 'W', 'Ho', 'Wo', 'Nn', 'ycs', 'ycw', 'nvalid', 'act')] + [('slope', c_f), ('nseg', c_i),
                                                                                            ('seg', TSeg * TCONV_MAXSEG)]


DWMULTI_MAXQ = 64


class DwMulti(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'nq', 'xcs', 'ycs', 'reflect', 'act')] + [('slope', c_f), ('ks', c_i * DWMULTI_MAXQ)]


KSUM_MAXSEG = 4


class KSeg(C.Structure):
    _fields_ = [('src', c_p), ('w', c_p)] + [(n, c_i) for n in ('xcs', 'c4', 'cin', 'ks', 'reflect', 'wcs')]


class KSum(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'Cout', 'ycs', 'ycw', 'rcs', 'act')] + [('slope', c_f), ('nseg', c_i), ('seg', KSeg * KSUM_MAXSEG)]


QCONV_MAXSEG = 8


class QSeg(C.Structure):
    _fields_ = [('src', c_p), ('scale', c_p), ('shift', c_p)] + [(n, c_i) for n in ('sstride', 'xcs', 'c4', 'cin', 'kh', 'kw', 'oy', 'ox', 'act')] + \
               [('slope', c_f), ('reflect', c_i), ('pack_off', c_i)]


class QConv(C.Structure):
    _fields_ = [('res', c_p), ('stats', c_p)] + [(n, c_i) for n in ('rcs', 'scs', 'N', 'H', 'W', 'Ho', 'Wo', 'S', 'OS', 'ncls', 'Nn', 'ycs', 'ycw', 'nvalid',
                                                                    'act')] + [('slope', c_f), ('nseg', c_i), ('seg', QSeg * QCONV_MAXSEG)]


class QPlan(C.Structure):
    _fields_ = [(n, c_i) for n in ('cs', 'nq', 'nsplit', 'th', 'tw', 'tiles')] + [('pack_floats', c_l)]


TNORM_MAXSLICE, DWM_MAXQ, PREP_MAXSRC = 8, 24, 8
DWM_MAXQ_BWD = 18


class NSlice(C.Structure):
    _fields_ = [('c0', c_i), ('c', c_i), ('running_mean', c_p), ('running_var', c_p), ('num_batches', c_p)]


class Stage1Geom(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'xcs', 'cin', 'reflect', 'ycs', 'scs')] + [('col0', c_i * 3), ('width', c_i * 3), ('nvalid', c_i * 3)]


class Stage1WGeom(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'xcs', 'cin', 'reflect', 'act')] + [('slope', c_f), ('ycs', c_i * 3), ('nvalid', c_i * 3)]


class DwmGeom(C.Structure):
    _fields_ = [(n, c_i) for n in ('N', 'H', 'W', 'nq', 'xcs', 'ycs', 'scs', 'sstride', 'reflect', 'act')] + [('slope', c_f), ('ks', c_i * DWM_MAXQ)]


class PrepJob(C.Structure):
    _fields_ = [('srcs', c_p * PREP_MAXSRC), ('dst', c_p)] + [(n, c_i) for n in ('kind', 'nsrc', 'n', 'mode', 'Nn', 'Ck', 'ks', 'wcs', 'wn', 'c4',
                                                                                'nt_total', 'col0', 'cs', 'block0', 'nblocks')]


PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_RELU6 = 0, 1, 2, 3, 4
NORM_INSTANCE, NORM_BATCH = 0, 1
LOSS_L1, LOSS_LSGAN, LOSS_HINGE_D_REAL, LOSS_HINGE_D_FAKE, LOSS_NEG_MEAN, LOSS_MSE, LOSS_BCE_LOGITS, LOSS_MEAN = range(8)

_G, _NG, _TG = C.POINTER(ConvGeom), C.POINTER(NormGeom), C.POINTER(TConv)
# name -> (restype, argtypes); mirrors include/cat_hip.h one to one (tests check every symbol is exported)
SIGNATURES = {
    'cat_hip_last_error': (C.c_char_p, []),
    'cat_hip_version': (c_i, []),
    'cat_conv2d_fwd': (c_i, [_G, c_p, c_p, c_p, c_p, c_p]),
    'cat_conv2d_dgrad': (c_i, [_G, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    'cat_conv2d_dgrad_t_applicable': (c_i, [_G]),
    'cat_conv2d_weight_transpose': (c_i, [_G, c_p, c_p, c_p]),
    'cat_conv2d_dgrad_t': (c_i, [_G, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    'cat_conv2d_wgrad_ws_bytes': (C.c_size_t, [_G]),
    'cat_conv2d_fwd_ws_bytes': (C.c_size_t, [_G]),
    'cat_conv2d_fwd_rect': (c_i, [_G, c_i, c_p, c_p, c_p, c_p, c_p]),
    'cat_pool2d_fwd': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p]),
    'cat_global_avgpool_fwd': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    'cat_resize_bilinear_fwd': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_f, c_f, c_p]),
    'cat_conv2d_fwd_ws': (c_i, [_G, c_p, c_p, c_p, c_p, c_p, c_p]),
    'cat_conv2d_dgrad_ws_bytes': (C.c_size_t, [_G, c_i]),
    'cat_conv2d_dgrad_ws': (c_i, [_G, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    'cat_conv2d_wgrad': (c_i, [_G, c_p, c_p, c_p, c_i, c_p, c_p]),
    'cat_conv2d_wgrad_batch_ws_bytes': (C.c_size_t, [C.POINTER(WgradItem), c_i]),
    'cat_conv2d_wgrad_batch': (c_i, [C.POINTER(WgradItem), c_i, c_p, c_p]),
    'cat_split_bf16': (c_i, [c_p, c_p, c_l, c_p]),
    'cat_conv2d_dgrad_split_applicable': (c_i, [_G]),
    'cat_conv2d_dgrad_split': (c_i, [_G, c_p, c_p, c_p, c_i, c_p]),
    'cat_conv2d_wgrad_split_applicable': (c_i, [_G]),
    'cat_conv2d_wgrad_split_ws_bytes': (C.c_size_t, [_G]),
    'cat_conv2d_wgrad_split': (c_i, [_G, c_p, c_p, c_p, c_i, c_p, c_p]),
    'cat_tconv_pack_floats': (C.c_size_t, [c_i, c_i, c_i]),
    'cat_tconv_pack': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'cat_tconv_fwd': (c_i, [_TG, c_p, c_p, c_p, c_p]),
    'cat_conv2d_ksum_supported': (c_i, [C.POINTER(KSum)]),
    'cat_conv2d_ksum_fwd': (c_i, [C.POINTER(KSum), c_p, c_p, c_p, c_p]),
    'cat_tstage1_supported': (c_i, [c_i, c_i, c_i]),
    'cat_tstage1_fwd': (c_i, [C.POINTER(Stage1Geom), c_p, c_p, c_p, c_p, c_p, c_p]),
    'cat_tstage1_dgrad_supported': (c_i, [c_i, c_i, c_i]),
    'cat_tstage1w_supported': (c_i, [c_i, c_i, c_i]),
    'cat_tstage1w_fwd': (c_i, [C.POINTER(Stage1WGeom), c_p, c_p, c_p, c_p, c_p]),
    'cat_tstage1_dgrad': (c_i, [C.POINTER(Stage1Geom), c_p, c_p, c_p, c_p, c_p]),
    'cat_tnorm_finalize': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p]),
    'cat_tnorm_finalize2': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p]),
    'cat_tnorm_sums': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'cat_tnorm_finalize_sums': (c_i, [c_p, c_d, c_i, c_p, c_p, c_i, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_p]),
    'cat_qconv_min_tiles16': (c_i, [c_i]),
    'cat_qconv_plan': (c_i, [C.POINTER(QConv), C.POINTER(QPlan)]),
    'cat_qconv_pack': (c_i, [C.POINTER(QConv), c_i, c_p, c_p, c_i, C.POINTER(c_i), c_i, c_i, c_i, c_p]),
    'cat_qconv_fwd': (c_i, [C.POINTER(QConv), c_p, c_p, c_p, c_p]),
    'cat_reflect_pad_bwd2': (c_i, [c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_affine_res_fwd': (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    'cat_dwm_fwd': (c_i, [C.POINTER(DwmGeom), c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'cat_dwm_bwd_ws_bytes': (C.c_size_t, [C.POINTER(DwmGeom)]),
    'cat_dwm_bwd': (c_i, [C.POINTER(DwmGeom), c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    'cat_prep_run': (c_i, [c_p, c_i, c_i, c_i, c_p]),
    'cat_dwconv2d_fwd': (c_i, [_G, c_p, c_p, c_p, c_p, c_p]),
    'cat_dwconv2d_multi_fwd': (c_i, [C.POINTER(DwMulti), c_p, c_p, c_p, c_p, c_p]),
    'cat_dwconv2d_dgrad': (c_i, [_G, c_p, c_p, c_p, c_i, c_p]),
    'cat_dwconv2d_wgrad': (c_i, [_G, c_p, c_p, c_p, c_i, c_p, c_p]),
    'cat_dwconv2d_wgrad_ws_bytes': (C.c_size_t, [_G]),
    'cat_reflect_pad_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_replicate_pad_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_replicate_pad_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_channel_sum': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
    'cat_channel_sum_ws_bytes': (C.c_size_t, [c_i, c_i]),
    'cat_norm_ws_bytes': (C.c_size_t, [_NG]),
    'cat_norm_fwd': (c_i, [_NG, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'cat_norm_bwd': (c_i, [_NG, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    'cat_bn_fold': (c_i, [c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p]),
    'cat_affine_act_fwd': (c_i, [c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_p]),
    'cat_act_fwd': (c_i, [c_p, c_p, c_l, c_i, c_f, c_p]),
    'cat_act_bwd': (c_i, [c_p, c_p, c_p, c_l, c_i, c_f, c_p]),
    'cat_add_n': (c_i, [C.POINTER(c_p), c_i, c_p, c_l, c_p]),
    'cat_concat2': (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_l, c_p]),
    'cat_slice_channels': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_l, c_p]),
    'cat_nchw_to_nhwc': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_nhwc_to_nchw': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_ka_ws_bytes': (C.c_size_t, [c_i]),
    'cat_ka_fwd': (c_i, [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p]),
    'cat_ka_bwd': (c_i, [c_p, c_l, c_i, c_p, c_p, c_p, c_p]),
    'cat_loss_ws_bytes': (C.c_size_t, [c_l]),
    'cat_loss_fwd': (c_i, [c_i, c_p, c_p, c_f, c_l, c_i, c_i, c_p, c_p, c_p]),
    'cat_loss_bwd': (c_i, [c_i, c_p, c_p, c_f, c_l, c_i, c_i, c_p, c_f, c_p, c_p]),
    'cat_adam_step': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p]),
    'cat_adam_step_dev': (c_i, [c_p, c_p, c_p, c_p, c_l, c_p, c_f, c_p]),
    'cat_prof_enable': (None, [c_i]),
    'cat_prof_collect': (c_i, []),
    'cat_prof_family': (c_i, [c_i, C.c_char_p, c_i, C.POINTER(c_l), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'cat_prof_family_bytes': (C.c_double, [c_i]),
    'cat_fill': (c_i, [c_p, c_l, c_f, c_p]),
    'cat_axpy': (c_i, [c_p, c_p, c_l, c_f, c_p]),
    'cat_bn_ws_bytes': (C.c_size_t, [c_l, c_i]),
    'cat_bn_stats_fwd': (c_i, [c_p, c_l, c_i, c_i, c_p, c_p, c_p]),
    'cat_bn_finalize': (c_i, [c_p, c_d, c_i, c_i, c_f, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'cat_bn_stats_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    'cat_bn_apply_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_p, c_p, c_p, c_i, c_l, c_i, c_i, c_i, c_f, c_p]),
    'cat_spade_fwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p]),
    'cat_spade_bwd_stats': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    'cat_spade_bwd_apply': (c_i, [c_p, c_p, c_p, c_p, c_d, c_p, c_l, c_i, c_i, c_p]),
    'cat_interp_nearest_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_upsample_nearest_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_avgpool3x3s2_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_avgpool3x3s2_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_maxpool2x2_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_maxpool2x2_bwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_onehot_edges': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cat_spectral_norm_ws_bytes': (C.c_size_t, [c_i, c_i, c_i, c_i]),
    'cat_spectral_norm_fwd': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    'cat_spectral_norm_bwd': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p]),
}

_lib = None


def lib_path():
    """The production library; CAT_LIB=diag (tools/debug/* only) selects the diagnostic build, whose ablation switches make results wrong."""
    if os.environ.get('CAT_LIB') == 'diag':
        import sys
        print('cat_amd: loading the DIAGNOSTIC library (CAT_LIB=diag): timing experiments only, results may be intentionally wrong', file=sys.stderr)
        return _build.DIAG_LIB
    alt = os.environ.get('CAT_LIB', '')
    if alt.endswith('.so'):      # A/B timing of two builds of this library on one box (tools/debug/*): an explicit path, never a fallback
        import sys
        print(f'cat_amd: loading {alt} (CAT_LIB)', file=sys.stderr)
        return alt
    return _build.LIB


def load():
    """dlopen the in-tree library (never builds implicitly on a GPU box: build() is the build step)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f'{path} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(cat_amd has no CPU / eager fallback)')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().cat_hip_last_error().decode()


def call(name, *args):
    """Call an int-returning entry point, raise on failure."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {last_error()}')


def query(name, *args):
    return getattr(load(), name)(*args)
