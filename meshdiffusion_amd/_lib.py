"""ctypes binding of libmeshdiffusion_hip.so (the C ABI declared in include/meshdiffusion_hip.h).

There is NO fallback: if the shared library is missing or an entry point fails, the caller
gets an exception.  PyTorch is only used by the callers for device memory and streams; every
argument that crosses this boundary is a raw device pointer or a plain integer/float.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MD_LIB") or os.path.join(_HERE, "libmeshdiffusion_hip.so")   # MD_LIB: an A/B build (build.py MD_LIB_SUFFIX)

# MD_CFG_* (include/meshdiffusion_hip.h)
(CFG_C3_128, CFG_C3_128_K16, CFG_C3_32, CFG_C3_LOW, CFG_C3_S2, CFG_G1_128, CFG_G1_128_LOW, CFG_G1_64_LOW,
 CFG_C3_128_V2, CFG_C3_128_SW, CFG_C3_128_PIPE) = range(11)          # 11..13 reserved (retired experiments)
(CFG_C3_128_FAST, CFG_C5_128_K16, CFG_C5_32_K16, CFG_C3_128_W4, CFG_C3X_32, CFG_C5X_32_K16, CFG_C3X_128_K16, CFG_C5X_128,
 CFG_G1_128_N128) = range(14, 23)
OUT_F32B, OUT_S16B, OUT_NCDHW = 0, 1, 2
PREC_BF16X3, PREC_FP16X2 = 0, 1
A_PACKED, A_S16B = 0, 1
B_S16B, B_F32B_GN = 0, 1

# (NT, KC) of each cfg -- must match csrc/gemm_conv.hip; checked against the library at load.
CFG_NT_KC = {
    CFG_C3_128: (128, 32), CFG_C3_128_K16: (128, 16), CFG_C3_32: (32, 32), CFG_C3_LOW: (128, 32),
    CFG_C3_S2: (128, 32), CFG_G1_128: (128, 32), CFG_G1_128_LOW: (128, 32), CFG_G1_64_LOW: (64, 32),
    CFG_C3_128_V2: (128, 32), CFG_C3_128_SW: (128, 32), CFG_C3_128_PIPE: (128, 32),
    CFG_C3_128_FAST: (128, 32), CFG_C5_128_K16: (128, 16), CFG_C5_32_K16: (32, 16), CFG_C3_128_W4: (128, 32), CFG_C3X_32: (32, 32), CFG_C5X_32_K16: (32, 16), CFG_C3X_128_K16: (128, 16), CFG_C5X_128: (128, 32), CFG_G1_128_N128: (128, 32),
}
# timing-only ablation ids of C3_128_V2 / C3_128_FAST: known to MD_BUILD_ABLATIONS=1 libraries only (tools/bench_conv.py)
ABLATION_CFG_NT_KC = {c: (128, 32) for c in (101, 102, 103, 104, 105, 111, 113, 114, 116, 117, 118, 122)}


class MdGemmConvArgs(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("alpha", C.c_float), ("cfg", C.c_int32), ("batch", C.c_int32),
        ("rows", C.c_int32), ("rows_alloc", C.c_int32), ("kdim", C.c_int32), ("D", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("ups", C.c_int32), ("a_src", C.c_int32),
        ("out_mode", C.c_int32), ("a_rows", C.c_int32), ("a_bstride", C.c_int64),
        ("bias_bstride", C.c_int64), ("res_bstride", C.c_int64), ("b_bstride", C.c_int64), ("partial", C.c_void_p), ("ksplit", C.c_int32), ("prec", C.c_int32),
        ("stats", C.c_void_p), ("stagger", C.c_int32), ("b_mode", C.c_int32), ("b2", C.c_void_p), ("b_ac", C.c_void_p),
        ("b2_bstride", C.c_int64), ("b_split", C.c_int32), ("b_silu", C.c_int32),
    ]


class MdPackJob(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("out", C.c_void_p), ("s_row", C.c_int64), ("s_k", C.c_int64), ("s_tap", C.c_int64),
        ("n_items", C.c_int64), ("block0", C.c_int64), ("rows", C.c_int32), ("kdim", C.c_int32), ("taps", C.c_int32),
        ("nt", C.c_int32), ("kc", C.c_int32), ("prec", C.c_int32), ("flip", C.c_int32), ("kind", C.c_int32),
    ]


PACK_WPK, PACK_WINO, PACK_WINO_F6 = 0, 1, 2
ABI_VERSION = 15     # MD_ABI_VERSION of include/meshdiffusion_hip.h this host code was written against
_P, _I32, _I64, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); exactly the entry points of include/meshdiffusion_hip.h
SIGNATURES = {
    "md_abi_version": (C.c_int, []),
    "md_device_count": (C.c_int, []),
    "md_gemm_conv": (C.c_int, [C.POINTER(MdGemmConvArgs), _P]),
    "md_gemm_conv_partial_bytes": (_I64, [C.POINTER(MdGemmConvArgs)]),
    "md_gemm_conv_cfg_info": (C.c_int, [_I32] + [C.POINTER(C.c_int32)] * 6),
    "md_pack_weights": (C.c_int, [_P, _P, _I32, _I32, _I32, _I64, _I64, _I64, _I32, _I32, _I32, _P]),
    "md_packed_weight_bytes": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "md_gn_stats": (C.c_int, [_P, _P, _I32, _I32, _I64, _I32, _I32, _P]),
    "md_gn_finalize": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _F, _P, _P]),
    "md_gn_apply": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _F, _U64, _P]),
    "md_dropout_scale": (C.c_int, [_P, _I32, _I32, _I64, _I32, _I32, _F, _U64, _P]),
    "md_zero": (C.c_int, [_P, _I64, _P]),
    "md_timestep_embedding": (C.c_int, [_P, _P, _I32, _I32, _P]),
    "md_linear": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "md_ncdhw_to_s16b_xfold": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_fold_dx": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_ncdhw_to_s16b": (C.c_int, [_P, _P, _I32, _I32, _I32, _I64, _P]),
    "md_f32b_to_ncdhw": (C.c_int, [_P, _P, _I32, _I32, _I64, _P]),
    "md_ncdhw_to_f32b": (C.c_int, [_P, _P, _I32, _I32, _I64, _P]),
    "md_s16b_to_ncdhw": (C.c_int, [_P, _P, _I32, _I32, _I64, _P]),
    "md_nin_f32": (C.c_int, [_P, _P, _I32, _I32, _P, _P, _P, _I32, _I32, _I64, _I32, _P]),
    "md_wino_operand_bytes": (_I64, [_I32, _I32, _I32, _I32, _I32]),
    "md_wino_prep": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _F, _U64, _P]),
    "md_wino_weight_bytes": (_I64, [_I32, _I32]),
    "md_wino_pack_weights": (C.c_int, [_P, _P, _I32, _I32, _I64, _I64, _I32, _P]),
    "md_conv3_wino": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_wino_equaliser": (C.c_int, [_P, _P, _P, _I32, _I32, _I64, _I64, _P, _P]),
    "md_wino_equaliser_measured": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I64, _I64, _P, _P]),
    "md_wino_operand_ms": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _I64, _P, _P]),
    "md_wino_prep_f8": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "md_wino_weight_bytes_f8": (_I64, [_I32, _I32]),
    "md_wino_pack_weights_f8": (C.c_int, [_P, _P, _P, _I32, _I32, _I64, _I64, _P]),
    "md_conv3_wino_f8": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_absmax": (C.c_int, [_P, _I64, _P, _P]),
    "md_wino_prep_dual_f6": (C.c_int, [_P, _I32, _P, _P, _P, _F, _P, _I32, _I32, _I32, _I32, _P]),
    "md_conv3_wino_f6_scaled": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _F, _P, _P]),
    "md_wino_prep_f6": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "md_wino_pack_weights_f6": (C.c_int, [_P, _P, _P, _I32, _I32, _I64, _I64, _P]),
    "md_conv3_wino_f6": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_conv3_stem": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_conv3_head": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_pack_batch": (C.c_int, [_P, _I32, _I64, _I32, _P]),
    "md_conv3_s2": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_wino_prep_v2": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _I32, _I32, _I32, _I32, _F, _U64, _P]),
    "md_wino_prep_dual": (C.c_int, [_P, _P, _I32, _I32, _P, _I32, _I32, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P]),
    "md_wgrad_wino_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "md_wgrad_wino": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _P]),
    "md_wgrad_nin_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "md_wgrad_nin": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I64, _I32, _I64, _I64, _P]),
    "md_attn_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "md_softmax_keys": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "md_ancestral_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I64, _P]),
    "md_ddim_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _P, _P, _I32, _I32, _I64, _P]),
    "md_inpaint_blend": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _I64, _I64, _P]),
    "md_inpaint_renoise": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I64, _P]),
    "md_ddpm_perturb": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I64, _P]),
    "md_masked_sq_err": (C.c_int, [_P, _P, _P, _P, _P, _F, _I32, _I32, _I64, _P]),
    "md_grad_sqnorm": (C.c_int, [_P, _I64, _P, _P]),
    "md_adam_ema_step": (C.c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _I32, _F, _P, _F, _P]),
    "md_pb16_bytes": (_I64, [_I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "md_to_pb16": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_wgrad_finish": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _I64, _P]),
    "md_wgrad_workspace_bytes": (_I64, [_I32, _I32, _I32, _I32]),
    "md_wgrad": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I64, _I64,
                           _I64, _P]),
    "md_gn_bwd_stats": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _F, _U64, _P]),
    "md_gn_bwd_finalize": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I64, _P]),
    "md_gn_bwd_apply": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _I32, _F, _U64, _P, _P, _P, _P]),
    "md_channel_sums": (C.c_int, [_P, _P, _I32, _I32, _I64, _P]),
    "md_s16b_transpose": (C.c_int, [_P, _P, _I32, _I32, _I32, _P]),
    "md_softmax_keys_bwd": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _F, _P]),
    "md_grad_resample": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P]),
    "md_marching_tets_workspace_bytes": (_I64, [_I32, _I32, _I32]),
    "md_vertex_normals": (C.c_int, [_P, _P, _I64, _I64, _P, _P, _P]),
    "md_marching_tets": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P]),
}

# include/meshdiffusion_hip_experimental.h: MD_BUILD_ABLATIONS=1 builds only (same header)
ABLATION_SIGNATURES = {"md_wgrad_set_debug": (None, [_I32])}

_lib = None


class MeshDiffusionHipError(RuntimeError):
    pass


def load():
    """dlopen the library (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MeshDiffusionHipError(
            f"{LIB_PATH} not found: run `python -m meshdiffusion_amd.build` (needs hipcc). "
            "There is no CPU fallback for the HIP path.")
    # ORDER MATTERS: PyTorch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Importing torch
    # first makes the dynamic linker satisfy our DT_NEEDED libamdhip64.so.7 with that already-loaded
    # copy, so torch and this library share ONE HIP runtime (streams, events and the null stream mean
    # the same thing on both sides).  Loading us first would pull in /opt/rocm's runtime next to
    # torch's and every launch would fail with hipErrorNoDevice.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in ABLATION_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.md_abi_version() != ABI_VERSION:
        raise MeshDiffusionHipError(f"ABI version mismatch: library {lib.md_abi_version()}, host code {ABI_VERSION}")
    if hasattr(lib, "md_wgrad_set_debug"):       # an ablation build: its extra configuration ids are usable
        CFG_NT_KC.update(ABLATION_CFG_NT_KC)
    for cfg, (nt, kc) in CFG_NT_KC.items():
        v = [C.c_int32() for _ in range(6)]
        lib.md_gemm_conv_cfg_info(cfg, *[C.byref(x) for x in v])
        if (v[0].value, v[1].value) != (nt, kc):
            raise MeshDiffusionHipError(f"cfg {cfg} NT/KC mismatch between _lib.py and the library")
    if torch.cuda.is_available() and lib.md_device_count() != torch.cuda.device_count():
        raise MeshDiffusionHipError("libmeshdiffusion_hip.so is not sharing PyTorch's HIP runtime "
                                    f"({lib.md_device_count()} vs {torch.cuda.device_count()} devices)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise MeshDiffusionHipError(f"{what} failed with code {rc}")


def cfg_info(cfg):
    lib = load()
    v = [C.c_int32() for _ in range(6)]
    check(lib.md_gemm_conv_cfg_info(cfg, *[C.byref(x) for x in v]), "md_gemm_conv_cfg_info")
    keys = ("nt", "kc", "cols", "taps", "lds_bytes", "threads")
    return dict(zip(keys, (x.value for x in v)))
