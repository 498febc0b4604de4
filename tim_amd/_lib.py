"""ctypes binding of libtimhip.so (include/timhip.h).

The product path has NO fallback: if the library is missing or a call fails the
error is raised.  PyTorch is used only for device memory and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TIM_AMD_LIB") or os.path.join(_HERE, "libtimhip.so")   # (TIM_AMD_LIB: an A/B build of the library, tools only)

ABI_VERSION = 6   # include/timhip.h: TIMHIP_VERSION
PREC_BF16, PREC_BF16X3, PREC_FP32, PREC_F16 = 0, 1, 2, 3
PRECISIONS = {"bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "fp32": PREC_FP32, "fp16": PREC_F16}
H16 = (PREC_BF16, PREC_F16)   # 16-bit operand storage (the MFMA GEMM / attention / transposing weight-gradient kernels)

(EPI_STORE_T, EPI_RELU_T, EPI_STORE_F32, EPI_GELU_DROP_T2, EPI_DROP_RES_F32, EPI_ADD_F32,
 EPI_DGELU_T, EPI_DRELU_T, EPI_ATOMIC_F32, EPI_SIGMOID_F32, EPI_DRELU_F32IN_T, EPI_GELU_DROP_G2, EPI_MULAUX_T, EPI_RELU_SPLIT3_T) = range(14)

# dropout site ids (csrc/common.h)
SITE_FEAT_V, SITE_FEAT_A, SITE_SEQ = 1, 2, 3
SITE_L_ATTN, SITE_L_DROP1, SITE_L_FFN, SITE_L_DROP2 = 0, 1, 2, 3


def layer_site(layer, which):
    return 16 + 8 * layer + which


SAVED_QKV, SAVED_O, SAVED_Y1, SAVED_X1T, SAVED_H, SAVED_Y2, SAVED_FFN_KEEP_BITS, SAVED_ATTN_KEEP_BITS = range(8)   # timhip_layer_saved_field

vp, i32, u32, u64, f32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t


class TimDesc(C.Structure):
    _fields_ = [("B", i32), ("S", i32), ("F", i32), ("d", i32), ("E", i32), ("H", i32), ("FF", i32),
                ("precision", i32), ("p_drop", f32), ("seed", u64), ("layer", i32), ("reserved", i32),
                ("grad_scale", vp)]


_LP = ["in_w", "in_wt", "out_w", "out_wt", "l1_w", "l1_wt", "l2_w", "l2_wt",
       "in_b", "out_b", "l1_b", "l2_b", "n1_w", "n1_b", "n2_w", "n2_b"]
_LG = ["in_w", "in_b", "out_w", "out_b", "l1_w", "l1_b", "l2_w", "l2_b", "n1_w", "n1_b", "n2_w", "n2_b", "ln_partials"]


class TimLayerParams(C.Structure):
    _fields_ = [(n, vp) for n in _LP]


class TimLayerGrads(C.Structure):
    _fields_ = [(n, vp) for n in _LG]


DESC_ATTN_FP32, DESC_ATTN_BWD_ONE_KERNEL, DESC_WGRAD_OVERWRITE, DESC_WGRAD_SEPARATE, DESC_OUTPROJ_SPLIT = 1, 2, 4, 8, 16
DESC_INPROJ_SPLIT, DESC_L1_SPLIT, DESC_L2_SPLIT = 32, 64, 128   # TimDesc.reserved flags
DESC_STREAM16, DESC_STREAM16_IN, DESC_STREAM16_OUT = 0x10000, 0x20000, 0x40000   # 16-bit residual gradient stream (fp16 backward)
EUNSUPPORTED = -2                # include/timhip.h: TIMHIP_EUNSUPPORTED
DESC_ATTN_KEEP_BITS = 0x80000   # the saved block carries the layer's attention keep-bits (timhip_attn_keep_bits)


class TimCastItem(C.Structure):
    _fields_ = [("src", vp), ("plain", vp), ("tr", vp), ("rows", i32), ("cols", i32), ("ldp", i32), ("ldt", i32)]


class TimWgradItem(C.Structure):
    _fields_ = [("dY", vp), ("X", vp), ("dW", vp), ("db", vp), ("ldy", i32), ("ldx", i32), ("Nout", i32), ("Kout", i32)]


class TimEpi(C.Structure):
    _fields_ = [("out0", vp), ("out1", vp), ("bias", vp), ("res", vp), ("aux", vp),
                ("ld0", i32), ("ld1", i32), ("ldres", i32), ("ldaux", i32),
                ("p_drop", f32), ("site", u32), ("seed", u64), ("mask", vp), ("ldmask", i32), ("reserved", i32),
                ("ln_stats", vp), ("ln_w", vp), ("ln_b", vp), ("acc_scale", vp), ("a_wrap_k", i32), ("reserved2", i32)]


class TimGemmItem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("lda", i32), ("ldb", i32), ("M", i32), ("N", i32), ("K", i32), ("reserved", i32),
                ("e", TimEpi)]


_SIGS = {
    "timhip_version": (C.c_int, []),
    "timhip_strerror": (C.c_char_p, [C.c_int]),
    "timhip_layer_saved_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_layer_workspace_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_layer_saved_field": (C.c_int, [C.POINTER(TimDesc), i32, C.POINTER(sz), C.POINTER(sz)]),
    "timhip_cast_weight": (C.c_int, [i32, vp, i32, i32, vp, i32, i32, vp]),
    "timhip_cast_weight_both": (C.c_int, [i32, vp, i32, i32, vp, i32, vp, i32, vp]),
    "timhip_cast_weights": (C.c_int, [i32, vp, i32, vp]),
    "timhip_gemm_nt": (C.c_int, [i32, i32, vp, i32, vp, i32, i32, i32, i32, C.POINTER(TimEpi), i32, vp]),
    "timhip_gemm_nt_group": (C.c_int, [i32, i32, vp, i32, vp]),
    "timhip_wgrad_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "timhip_wgrad": (C.c_int, [i32, vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, sz, vp, vp]),
    "timhip_wgrad_group_workspace_bytes": (sz, [i32, vp, i32, i32]),
    "timhip_wgrad_group": (C.c_int, [i32, vp, i32, i32, i32, vp, sz, vp, vp]),
    "timhip_transpose": (C.c_int, [i32, vp, i32, i32, i32, vp, i32, vp]),
    "timhip_colsum": (C.c_int, [i32, vp, i32, i32, i32, vp, vp]),
    "timhip_cast_rows": (C.c_int, [i32, vp, i32, i32, i32, vp, i32, f32, u64, u32, vp, vp]),
    "timhip_dropout_rows_bwd": (C.c_int, [vp, i32, i32, i32, vp, i32, f32, u64, u32, vp]),
    "timhip_layernorm_fwd": (C.c_int, [i32, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, vp, vp]),
    "timhip_layernorm_bwd": (C.c_int, [i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, vp, i32, vp, i32,
                                       f32, u64, u32, vp, vp, vp, vp]),
    "timhip_attention_fwd": (C.c_int, [C.POINTER(TimDesc), vp, vp, vp, vp]),
    "timhip_attention_bwd": (C.c_int, [C.POINTER(TimDesc), vp, vp, vp, vp, vp, vp, sz, vp]),
    "timhip_attention_bwd_workspace_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_time_l1_fwd": (C.c_int, [i32, vp, i32, i32, vp, vp, vp, i32, vp]),
    "timhip_time_l1_fwd_split3": (C.c_int, [i32, vp, i32, i32, vp, vp, vp, i32, vp]),
    "timhip_time_l1_bwd": (C.c_int, [i32, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp]),
    "timhip_dropout_mask": (C.c_int, [u64, u32, f32, i32, i32, vp, vp]),
    "timhip_dropout_salt": (C.c_int, [vp]),
    "timhip_layer_fwd": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), vp, vp, vp, vp, vp, vp, sz, vp]),
    "timhip_layer_fwd_chained": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), C.POINTER(TimLayerParams), vp, vp, vp,
                                           vp, vp, vp]),
    "timhip_layer_bwd": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), vp, vp, vp, vp,
                                   C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_bwd_split": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), vp, vp, vp, vp, vp, vp,
                                         C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_bwd_data_split": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), vp, vp, vp, vp, vp, vp,
                                              C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_dy_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_layer_data_workspace_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_layer_wgrad_workspace_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_layer_bwd_data": (C.c_int, [C.POINTER(TimDesc), C.POINTER(TimLayerParams), vp, vp, vp, vp,
                                        C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_bwd_weights": (C.c_int, [C.POINTER(TimDesc), vp, vp, vp, C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_bwd_weights_pair": (C.c_int, [C.POINTER(TimDesc), vp, vp, vp, C.POINTER(TimLayerGrads), vp, vp, vp,
                                                C.POINTER(TimLayerGrads), vp, sz, vp]),
    "timhip_layer_wgrad_pair_wins": (C.c_int, [C.POINTER(TimDesc)]),
    "timhip_assemble_fwd": (C.c_int, [i32, vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, f32, u64, u32,
                                      vp, vp, vp]),
    "timhip_assemble_bwd": (C.c_int, [vp, i32, i32, i32, vp, i32, i32, f32, u64, u32, vp, vp, vp, vp, vp, vp]),
    "timhip_assemble_fwd_p": (C.c_int, [i32, vp, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, vp, i32, f32, u64, u32, vp, vp, vp]),
    "timhip_assemble_bwd_p": (C.c_int, [vp, i32, i32, i32, vp, i32, i32, f32, u64, u32, vp, vp, vp, i32, vp, vp, i32, vp]),
    "timhip_gather_rows": (C.c_int, [i32, vp, i32, i32, i32, i32, i32, vp, vp]),
    "timhip_ce_mixup_fwd": (C.c_int, [vp, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp, vp]),
    "timhip_ce_mixup_bwd": (C.c_int, [vp, i32, i32, i32, vp, vp, f32, f32, vp, vp, vp, vp, i32, vp]),
    "timhip_focal_loss_fwd": (C.c_int, [vp, vp, i32, i32, vp, vp, f32, f32, vp, vp, vp]),
    "timhip_focal_loss_bwd": (C.c_int, [vp, vp, i32, i32, vp, vp, f32, f32, vp, vp, vp]),
    "timhip_diou_1d": (C.c_int, [vp, vp, i32, vp, f32, vp, vp, vp, vp]),
    "timhip_det_side_loss_fwd": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, f32, vp, vp, vp]),
    "timhip_det_side_loss_bwd": (C.c_int, [vp, vp, vp, i32, i32, vp, vp, vp, f32, f32, f32, f32, f32, vp, vp, vp, vp, vp]),
    "timhip_softnms_1d_workspace_bytes": (C.c_size_t, [C.c_int64, i32]),
    "timhip_softnms_1d": (C.c_int, [vp, vp, vp, vp, i32, f32, f32, f32, i32, vp, vp, vp, vp, C.c_size_t, vp]),
    "timhip_nms_1d": (C.c_int, [vp, vp, vp, i32, f32, vp, vp, vp, vp]),
    "timhip_window_gather": (C.c_int, [vp, i32, i32, vp, vp, i32, vp, i32, vp, vp, vp]),
    "timhip_window_times": (C.c_int, [vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, f32, vp, vp]),
    "timhip_gemm_timing_start": (C.c_int, [i32, C.c_double]),
    "timhip_gemm_timing_stop": (C.c_int, [vp, vp, vp]),
    "timhip_timing_stop_families": (C.c_int, [vp, vp, vp]),
    "timhip_drloc_gather": (C.c_int, [i32, vp, vp, C.c_int64, C.c_int64, i32, i32, i32, vp, vp, i32, vp, i32, vp]),
    "timhip_drloc_scatter_add": (C.c_int, [vp, i32, vp, vp, C.c_int64, C.c_int64, i32, i32, i32, vp, vp, i32, vp]),
    "timhip_scatter_rows_add": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp]),
    "timhip_layer_ln_partial_bytes": (sz, [C.POINTER(TimDesc)]),
    "timhip_ln_partials_reduce": (C.c_int, [vp, i32, i32, i32, vp, vp, vp]),
    "timhip_gather_ranges": (C.c_int, [i32, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "timhip_layernorm_fwd2": (C.c_int, [i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, i32, vp, i32, vp, vp]),
    "timhip_layernorm_bwd2": (C.c_int, [i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]),
    "timhip_cast_rows_pair": (C.c_int, [i32, vp, vp, vp, vp, i32, f32, u64, vp, vp]),
    "timhip_gather_split3_ranges": (C.c_int, [i32, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "timhip_scatter_ranges_add": (C.c_int, [i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "timhip_dx_init": (C.c_int, [i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp]),
    "timhip_sigmoid_bwd_rows": (C.c_int, [i32, vp, vp, i32, i32, vp, i32, vp, vp]),
    "timhip_dx_init_slabs": (C.c_int, [i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp]),
    "timhip_cast_rows_many": (C.c_int, [i32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "timhip_grad_scale": (C.c_int, [vp, vp, i32, f32, vp, vp]),
    "timhip_reload_env": (None, []),
    "timhip_build_flags": (C.c_int, []),
    "timhip_attn_keep_bits": (C.c_int, [C.POINTER(TimDesc), i32, vp, vp]),
    "timhip_gemm_p8_choice": (C.c_int, [i32, i32, i32, i32]),
    "timhip_dp_reduce": (C.c_int, [i32, vp, i32, C.c_longlong, f32, vp, vp]),
    "timhip_split3_many": (C.c_int, [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "timhip_label_queries": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp]),
    "timhip_smooth_one_hot": (C.c_int, [vp, i32, i32, C.c_int64, i32, f32, f32, vp, vp]),
}

_lib = None


class TimHipError(RuntimeError):
    pass


def load():
    """Load libtimhip.so (built by `__graft_entry__.build()` / csrc/Makefile).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TimHipError("libtimhip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (make -C tim_amd/csrc). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.timhip_version.restype = C.c_int
    if lib.timhip_version() != ABI_VERSION:   # (a stale A/B build would read this binding's buffers with yesterday's layout)
        raise TimHipError("%s is ABI version %d, this binding speaks %d: rebuild it (make -C tim_amd/csrc%s)"
                          % (LIB_PATH, lib.timhip_version(), ABI_VERSION, " TUNING=1" if "tuning" in LIB_PATH else ""))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def reload_env():
    """the launchers read their TIMHIP_* A/B knobs once; call this after changing one inside a process (tests, tools)"""
    load().timhip_reload_env()


def tuning_build():
    """True when the loaded library is a TUNING=1 build (carries the measured-slower kernel variants)"""
    return bool(load().timhip_build_flags() & 1)


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = load().timhip_strerror(rc).decode()
        raise TimHipError("libtimhip: %s failed: %s (%d)" % (what, msg, rc))


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def call(name, *args):
    check(getattr(load(), name)(*args), name)
