"""ctypes binding of libd2t.so (include/d2t.h).

The shared library is built in-tree by doc2tex_amd/csrc/build.sh (see
__graft_entry__.build).  There is no CPU or PyTorch fallback: if the library is
missing or no HIP device is visible, every compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libd2t.so")
if os.environ.get("D2T_PROBE_LIB"):  # development tools only: a probe build of the library (csrc/build.sh with D2T_PROBES=1)
    LIB_PATH = os.path.abspath(os.environ["D2T_PROBE_LIB"])

D2T_OK = 0
ENC_RESNET, ENC_HYBRID_VIT, ENC_VGG_BILSTM, ENC_RESNET_BILSTM = 0, 1, 2, 3
DEC_TFM, DEC_ATTN = 0, 1
ATTN_KEYS_ALL_INIT_MEAN, ATTN_KEYS_NOCLS_INIT_CLS, ATTN_KEYS_ALL_INIT_FIRST = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
CONV_FP32, CONV_BF16X3, CONV_FP16X2, CONV_MIXED = 0, 1, 2, 3
ATTN_CELL_LOCATION, ATTN_CELL_BAHDANAU = 0, 1
VIT_POS_SINCOS_PREFIX, VIT_POS_LEARNED_INTERP, VIT_POS_LEARNED_PREFIX = 0, 1, 2


class D2TConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "encoder", "in_channels", "backbone_out", "vit_depth", "vit_heads", "vit_dim", "patch_h", "patch_w",
        "max_h", "max_w", "dec_dim", "dec_heads", "dec_layers", "dec_ff", "vocab", "max_seq_len",
        "decoder", "attn_hidden", "attn_kernel_size", "attn_kernel_dim", "attn_keys", "attn_enc_init",
        "attn_coverage", "bilstm_hidden", "batch_max_length", "gcb", "attn_cell", "attn_onehot", "vit_pos")]


class D2TPrepConfig(C.Structure):  # include/d2t_prep.h d2t_prep_config
    _fields_ = [(n, C.c_int32) for n in ("max_h", "max_w", "min_h", "min_w", "downsample", "variant")] + \
               [("mean", C.c_float), ("std", C.c_float), ("norm_mode", C.c_int32)]


class D2TPrepPlan(C.Structure):  # include/d2t_prep.h d2t_prep_plan
    _fields_ = [(n, C.c_int32) for n in ("src_h", "src_w", "ds_h", "ds_w", "rs_h", "rs_w", "out_h", "out_w",
                                         "min_branch", "status")]


PREP_DEMO, PREP_API = 0, 1
NORM_ALB, NORM_RAW = 0, 1
PREP_OK, PREP_UNBOUND_LOCAL, PREP_FALLBACK = 0, 1, 2
PREP_FLAG_PASTE_MISMATCH = 1
POST_NONE, POST_API, POST_DEMO = 0, 1, 2

_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
# name -> (restype, argtypes); must list every symbol include/d2t.h declares
SIGNATURES = {
    "d2t_create": (_I, [C.POINTER(D2TConfig), C.POINTER(_P)]),
    "d2t_destroy": (None, [_P]),
    "d2t_last_error": (C.c_char_p, [_P]),
    "d2t_device_available": (_I, []),
    "d2t_device_of": (_I, [_P]),
    "d2t_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _P]),
    "d2t_finalize_weights": (_I, [_P, _P]),
    "d2t_encoder_shape": (_I, [_P, _I, _I] + [C.POINTER(_I)] * 6),
    "d2t_encode": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "d2t_decode_greedy": (_I, [_P, _P, _I, _I, _P, _I, _P, _P, C.POINTER(_I), _P]),
    "d2t_decode_attn_greedy": (_I, [_P, _P, _I, _I, _I, _P, _P, C.POINTER(_I), _P]),
    "d2t_decode_greedy_async": (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "d2t_decode_wait": (_I, [_P, _P, _I]),
    "d2t_decode_greedy_submit": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P, _P, C.POINTER(_L)]),
    "d2t_decode_steps": (_I, [_P, _L, C.POINTER(_I), _I, C.POINTER(_I)]),
    "d2t_decode_last_ticket": (_L, [_P]),
    "d2t_decode_query": (_I, [_P, _L]),
    "d2t_decode_wait_ticket": (_I, [_P, _L, _P, _I]),
    "d2t_decode_beam": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(C.c_float), _P]),
    "d2t_decode_beam_batch": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(C.c_float), _P]),
    "d2t_decode_attn_beam_batch": (_I, [_P, _P, _I, _I, _I, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(C.c_float), _P]),
    "d2t_decode_attn_beam": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(C.c_float), _P]),
    "d2t_set_conv_precision": (_I, [_P, _I]),
    "d2t_set_mixed_units": (_I, [_P, _I]),
    "d2t_set_reserved_blocks": (_I, [_P, _I]),
    "d2t_set_decode_chains": (_I, [_P, _I]),
    "d2t_set_conv_kernel": (_I, [_P, _I]),
    "d2t_set_conv_fusion": (_I, [_P, _I, _I]),
    "d2t_set_beam_shared_tile": (_I, [_P, _I]),
    "d2t_set_reserved_cus": (_I, [_P, _I]),
    "d2t_train_forward": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "d2t_train_backward": (_I, [_P, _P, _P]),
    "d2t_train_grad": (_I, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "d2t_reload_weights": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _P]),
    "d2t_train_gather": (_I, [_P, _I, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P, _P]),
    "d2t_read_weight": (_I, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "d2t_train_set_dropout": (_I, [_P, C.c_float, C.c_uint64]),
    "d2t_train_set_teacher_flags": (_I, [_P, C.c_char_p, _I]),
    "d2t_train_mask_count": (_I, [_P]),
    "d2t_train_decision_count": (_I, [_P]),
    "d2t_train_read_decision": (_I, [_P, _I, _P, C.c_int64, C.POINTER(_I), C.POINTER(C.c_int64), _P]),
    "d2t_train_read_mask": (_I, [_P, _I, _P, C.c_int64, _P]),
    "d2t_train_release": (None, [_P]),
    "d2t_profile_enable": (_I, [_P, _I]),
    "d2t_profile_read": (_I, [_P, _I, C.POINTER(_I)] + [C.POINTER(_I)] * 3 + [C.POINTER(C.c_float)]),
    "d2t_op_conv2d": (_I, [_P] * 5 + [_I] * 12 + [_P]),
    "d2t_op_conv2d_bf16x3": (_I, [_P] * 5 + [_I] * 12 + [_P]),
    "d2t_op_conv2d_bf16x3_split": (_I, [_P] * 5 + [_I] * 12 + [_P]),
    "d2t_op_conv2d_bf16x3_split_pool": (_I, [_P] * 4 + [_I] * 12 + [_P]),
    "d2t_op_set_conv_kernel": (_I, [_I, _I]),
    "d2t_op_linear": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "d2t_op_maxpool2x2": (_I, [_P, _P] + [_I] * 8 + [_P]),
    "d2t_op_layernorm": (_I, [_P] * 4 + [_I, _I, C.c_float, _P]),
    "d2t_op_vit_attention": (_I, [_P, _P, _I, _I, _I, _P]),
    "d2t_op_decode_attention": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "d2t_ce_forward": (_I, [_P, _P, _P, _P, _I, _I, _L, _P]),
    "d2t_ce_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _P]),
    "d2t_op_train_conv": (_I, [_P] * 14 + [_I] * 13 + [_P]),
    "d2t_op_train_linear": (_I, [_P] * 10 + [_I] * 5 + [_P]),
    "d2t_op_train_layernorm": (_I, [_P] * 8 + [_I, _I, C.c_float, _P]),
    "d2t_op_train_attention": (_I, [_P] * 7 + [_I] * 6 + [_P]),
    "d2t_op_train_maxpool": (_I, [_P] * 4 + [_I] * 8 + [_P]),
}
# include/d2t_prep.h
SIGNATURES_PREP = {
    "d2t_prep_plan_image": (_I, [C.POINTER(D2TPrepConfig), _I, _I, C.POINTER(D2TPrepPlan)]),
    "d2t_prep_plan_fallback": (_I, [C.POINTER(D2TPrepConfig), _I, _I, C.POINTER(D2TPrepPlan)]),
    "d2t_prep_create": (_I, [C.POINTER(D2TPrepConfig), C.POINTER(_P)]),
    "d2t_prep_destroy": (None, [_P]),
    "d2t_prep_last_error": (C.c_char_p, [_P]),
    "d2t_prep_run": (_I, [_P, _I, C.POINTER(D2TPrepPlan), _P, C.POINTER(_L), _P, _I, _I, _P, _P]),
    "d2t_prep_pad_hist": (_I, [_P, _I, _P, C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), _P, _P]),
    "d2t_prep_pad_bbox": (_I, [_P, _I, _P, C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), _P, _P, _P]),
    "d2t_prep_pad_apply": (_I, [_P, _I, _P, C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _P, _I, _P,
                                C.POINTER(_L), C.POINTER(_I), C.POINTER(_I), _P, _P]),
    "d2t_prep_lanczos_coeffs": (_I, [_I, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "d2t_vocab_create": (_I, [C.POINTER(C.c_char_p), _I, C.POINTER(_P)]),
    "d2t_vocab_destroy": (None, [_P]),
    "d2t_post_decode": (_I, [_P, C.POINTER(_L), _I, _I, C.c_char_p, _I, _I, C.c_char_p, _L, C.POINTER(_L), C.POINTER(_L)]),
    "d2t_post_strip_whitespace": (_I, [C.c_char_p, _I, C.c_char_p, _L]),
}

_lib = None


def load():
    """Load libd2t.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"doc2tex_amd: {LIB_PATH} is not built (run doc2tex_amd/csrc/build.sh or "
            "__graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(SIGNATURES_PREP.items()):
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def require_device():
    lib = load()
    if not lib.d2t_device_available():
        raise RuntimeError("doc2tex_amd: no HIP device visible; the engine has no CPU path")
    return lib


def ptr(t):
    """Raw device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_of(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def check(rc, ctx=None, what=""):
    if rc != D2T_OK:
        msg = load().d2t_last_error(ctx).decode() if ctx else ""
        raise RuntimeError(f"libd2t {what} failed (code {rc}) {msg}")
