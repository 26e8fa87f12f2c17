"""ctypes binding of libfreefine_hip.so (C ABI in include/freefine_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FREEFINE_HIP_LIB") or os.path.join(_HERE, "libfreefine_hip.so")   # env override: A/B builds of the same ABI

FFN_F32, FFN_BF16, FFN_BF16X3, FFN_FP8 = 0, 1, 2, 3
IG_OUT_SILU, IG_OUT_F32, IG_GEGLU, IG_OUT_TRANSPOSED, IG_OUT_PAIR, IG_OUT_GELU, IG_OUT_RELU, IG_OUT_KV64 = 1, 2, 4, 8, 16, 32, 64, 128
ELT_RELU, ELT_ADD = 0, 1
ATT_MAXP, ATT_MAXB = 4, 16
ATT_HEAD_RULE, ATT_UNIFORM_SEL1, ATT_UNIFORM_SEL0 = 1, 2, 4
NORM_SILU, NORM_OUT_PAIR = 1, 2


class IgemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("W", C.c_void_p), ("out", C.c_void_p),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("residual", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("Kpad", C.c_int),
        ("lda", C.c_int), ("ldo", C.c_int), ("ldr", C.c_int), ("ldrb", C.c_int),
        ("rows_per_batch", C.c_int),
        ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("upsample", C.c_int),
        ("flags", C.c_int), ("alpha", C.c_float), ("conv", C.c_int),
        ("splitk", C.c_int), ("ws", C.c_void_p), ("ws_bytes", C.c_long),
        ("a_lo", C.c_int), ("x3", C.c_int), ("f8", C.c_int), ("kv64_from", C.c_int),
    ]


class AttnEntry(C.Structure):
    _fields_ = [
        ("q_row", C.c_int), ("kv_row", C.c_int),
        ("w_const", C.c_float), ("w_slope", C.c_float),
        ("wq", C.c_void_p), ("kmask", C.c_void_p), ("qsel", C.c_void_p),
        ("flags", C.c_int), ("hr_row", C.c_int),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p), ("w_dev", C.c_void_p),
        ("Bo", C.c_int), ("S", C.c_int), ("Sk", C.c_int), ("heads", C.c_int), ("D", C.c_int),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldvt", C.c_int), ("ldo", C.c_int),
        ("scale", C.c_float), ("npass", C.c_int), ("out_pair", C.c_int), ("kv_pair", C.c_int),
        ("e", AttnEntry * (ATT_MAXP * ATT_MAXB)),
    ]


class CtrlStepDesc(C.Structure):
    _fields_ = [
        ("eps", C.c_void_p), ("x", C.c_void_p), ("noise", C.c_void_p), ("m", C.c_void_p), ("om", C.c_void_p),
        ("x_prev", C.c_void_p), ("pred_x0", C.c_void_p),
        ("c_bt", C.c_float), ("c_at", C.c_float), ("c_ap", C.c_float), ("c_dir", C.c_float),
        ("c_dirm", C.c_float * 8), ("stdv", C.c_float * 8), ("row_masked", C.c_int * 8),
        ("rows", C.c_int), ("CHW", C.c_int), ("HW", C.c_int),
    ]


class SplatXform(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("translate", C.c_float * 3), ("rotate", C.c_float * 9), ("scale", C.c_float * 3),
                ("tan_half_fov", C.c_float)]


class PackDesc(C.Structure):
    _fields_ = [
        ("src", C.c_void_p), ("dst", C.c_void_p), ("src_row", C.c_int * 16),
        ("B", C.c_int), ("Cl", C.c_int), ("CP", C.c_int), ("HW", C.c_int),
    ]


# every symbol include/freefine_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long
SYMBOLS = {
    "ffn_version": (_i, []),
    "ffn_last_error": (C.c_char_p, []),
    "ffn_device_info": (_i, [_i, C.c_char_p, _i]),
    "ffn_graph_launch": (_i, [_vp, _vp]),
    "ffn_igemm": (_i, [_vp, _i, C.POINTER(IgemmDesc)]),
    "ffn_conv3x3_n4": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "ffn_split_pair": (_i, [_vp, _vp, _vp, _l, _i, _i]),
    "ffn_groupnorm_f8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _f, _vp, _vp, _vp]),
    "ffn_attn": (_i, [_vp, _i, C.POINTER(AttnDesc)]),
    "ffn_attn_kernel_name": (_i, [_i, C.POINTER(AttnDesc), C.c_char_p, _i]),
    "ffn_igemm_tune": (_i, [_vp, _i, C.POINTER(IgemmDesc)]),
    "ffn_igemm_tune_entry_ints": (_i, []),
    "ffn_igemm_tune_export": (_i, [C.POINTER(C.c_int), _i]),
    "ffn_igemm_tune_import": (_i, [C.POINTER(C.c_int), _i]),
    "ffn_igemm_tune_stamp": (_i, []),
    "ffn_igemm_tune_clear": (_i, []),
    "ffn_igemm_tune_enable": (_i, [_i]),
    "ffn_igemm_num_configs": (_i, []),
    "ffn_igemm_force_config": (_i, [_i]),
    "ffn_igemm_variant": (_i, [C.POINTER(IgemmDesc), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ffn_igemm_kernel_name": (_i, [_i, C.POINTER(IgemmDesc), C.c_char_p, _i]),
    "ffn_attn_presplit": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "ffn_attn_variant": (_i, [_i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ffn_gn_nchunk": (_i, [_i]),
    "ffn_gn_fused": (_i, [_i, _i, _i, _i]),
    "ffn_gn_stats": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "ffn_gn_apply": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "ffn_groupnorm": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "ffn_groupnorm_pair_raw": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "ffn_layernorm": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _f]),
    "ffn_layernorm_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f]),
    "ffn_softmax_rows": (_i, [_vp, _i, _vp, _vp, _l, _i, _f]),
    "ffn_cfg_masked": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _l, _i]),
    "ffn_ddim_inv_step": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _l]),
    "ffn_ddim_ctrl_step": (_i, [_vp, C.POINTER(CtrlStepDesc)]),
    "ffn_pack_nchw": (_i, [_vp, _i, C.POINTER(PackDesc)]),
    "ffn_nhwc_to_nchw_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "ffn_concat": (_i, [_vp, _i, _vp, _vp, _vp, _l, _i, _i]),
    "ffn_timestep_embed": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i]),
    "ffn_transpose": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i]),
    "ffn_cast": (_i, [_vp, _i, _i, _vp, _vp, _l]),
    "ffn_eltwise": (_i, [_vp, _i, _i, _vp, _vp, _vp, _l]),
    "ffn_resize_bilinear": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "ffn_splat_lift": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f]),
    "ffn_splat_project": (_i, [_vp, _vp, _vp, _i, C.POINTER(SplatXform)]),
    "ffn_splat_bin": (_i, [_vp, _i, _vp, _i, _f, _i, _i, _vp, _vp, _vp]),
    "ffn_splat_render": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _vp, _vp]),
    "ffn_image_to_nhwc": (_i, [_vp, _i, _vp, _vp, _l, _i]),
    "ffn_nhwc_to_image": (_i, [_vp, _i, _vp, _vp, _i, _i, _i]),
}

_lib = None


class FreeFineHipError(RuntimeError):
    pass


def load():
    """Load the HIP extension; raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FreeFineHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
        import torch  # noqa: F401  -- load torch's HIP runtime FIRST so this library binds to the same libamdhip64 instance
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ffn_last_error()
        raise FreeFineHipError(f"{what} failed rc={rc}: {msg.decode() if msg else ''}")
