"""ctypes binding of libnvae_hip.so (include/nvae_hip.h).

The product path has no CPU fallback: importing this module without a built library, or calling
an entry point that fails, raises.  PyTorch is used only as the owner of device memory and streams:
tensors are passed as raw device pointers, the current torch stream as a hipStream_t."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnvae_hip.so")

ABI_VERSION = 6          # must equal nvae_abi_version() of the loaded library (include/nvae_hip.h NVAE_ABI_VERSION)
F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_SWISH, ACT_ELU = 0, 1, 2
OP_AFFINE, OP_SWISH, OP_ELU = 0, 1, 2
HY_LR, HY_BETA, HY_BALANCE, HY_GSCALE, HY_LSCALE, HY_GOOD, HY_OVERFLOW, HY_SIZE = 0, 1, 2, 3, 4, 5, 6, 8
RES_LOSS, RES_BN, RES_RECON, RES_KL, RES_SIZE = 0, 1, 2, 3, 8


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "Hin", "Win", "Cin", "Hout", "Wout", "Cout", "KH", "KW", "stride", "pad_t",
                 "pad_l", "div", "exact", "in_ld", "out_ld", "res_ld")]


class BnBwdFuse(C.Structure):     # NvaeBnBwdFuse
    _fields_ = [("x", C.c_void_p), ("x_ld", C.c_int), ("act", C.c_int), ("frozen", C.c_int),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("partials", C.c_void_p), ("counters", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("k0k1", C.c_void_p)]


class BnFin(C.Structure):        # NvaeBnFin
    _fields_ = [("counter", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("rm", C.c_void_p),
                ("rv", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float), ("scale", C.c_void_p),
                ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p)]


class BnIn(C.Structure):         # NvaeBnIn
    _fields_ = [("slab", C.c_void_p), ("rows", C.c_int), ("momentum", C.c_float), ("eps", C.c_float),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("rm", C.c_void_p), ("rv", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p)]


class ConvPre(C.Structure):      # NvaeConvPre
    _fields_ = [("bn", BnIn), ("act", C.c_int), ("act_out", C.c_void_p), ("act_ld", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("w_off", C.c_longlong), ("wf_off", C.c_longlong), ("wd_off", C.c_longlong),
                ("u_off", C.c_int), ("t_off", C.c_int), ("K", C.c_int), ("Cout", C.c_int),
                ("Cin", C.c_int), ("taps", C.c_int), ("wf_ld", C.c_int), ("wd_ld", C.c_int),
                ("idx", C.c_int), ("blk_off", C.c_int), ("p_off", C.c_longlong)]


_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
_G = C.POINTER(ConvGeom)

# name -> argtypes (the trailing stream argument is appended automatically)
_SIGS = {
    "nvae_conv_gemm_stats_rows": None,
    "nvae_conv_gemm": [_i, _G, _p, _p, _i, _p, _p, _p, _i, _p],
    "nvae_conv_gemm_ex": [_i, _G, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p],
    "nvae_conv_gemm_pre_max_cin": None,
    "nvae_conv_gemm_force_tile": None,
    "nvae_conv_gemm_force_split": None,
    "nvae_conv_set_workspace": None,
    "nvae_conv_img_ok": None,
    "nvae_conv_img_enable": None,
    "nvae_set_deterministic": None,
    "nvae_get_deterministic": None,
    "nvae_conv_gemm_family": None,
    "nvae_conv_halo4_enable": None,
    "nvae_conv_halo_stamps": None,
    "nvae_conv_gemm_bnbwd": [_i, _p, _p, _p, _i, _p, _p, _p, _p],
    "nvae_conv_wgrad_scratch": None,
    "nvae_conv_wgrad_scratch_n": None,
    "nvae_conv_wgrad": [_i, _G, _p, _p, _p, _i, _p, _p, _l],
    "nvae_conv_wgrad_batched": [_i, _G, _i, _p, _p, _p, _i, _p, _p, _l],
    "nvae_conv_direct": [_i, _G, _p, _p, _l, _l, _l, _i, _p, _p, _p, _i],
    "nvae_conv_direct_wgrad": [_i, _G, _p, _p, _p, _i, _p],
    "nvae_colsum": [_i, _p, _l, _i, _i, _p],
    "nvae_dwconv5": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i],
    "nvae_dwconv5_wgrad": [_i, _p, _p, _p, _p, _i, _i, _i, _i],
    "nvae_dwconv5_pre": [_i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    "nvae_dwconv5_bnbwd": [_i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p],
    "nvae_dwconv5_wgrad_pre": [_i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i],
    "nvae_dwconv5_stats": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "nvae_reduce_splits": None,
    "nvae_dwconv5_stats_rows": None,
    "nvae_bn_stats": [_i, _p, _l, _i, _p],
    "nvae_bn_finalize": [_i, _p, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p],
    "nvae_bn_finalize_s": [_i, _p, _i, _l, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p],
    "nvae_bn_eval_prepare": [_p, _p, _p, _p, _i, _f, _p, _p, _p, _p],
    "nvae_bn_apply": [_i, _p, _p, _l, _i, _p, _p, _i],
    "nvae_bn_bwd_reduce": [_i, _p, _p, _l, _i, _p, _p, _i, _p],
    "nvae_bn_apply_fin": [_i, _p, _p, _l, _i, _p, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i],
    "nvae_bn_bwd_apply_fin": [_i, _p, _p, _p, _l, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i],
    "nvae_bn_stats_fin": [_i, _p, _l, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p],
    "nvae_bn_bwd_reduce_fin": [_i, _p, _p, _l, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i],
    "nvae_bn_bwd_finalize": [_i, _p, _l, _i, _p, _p, _p, _p, _p, _p, _i],
    "nvae_bn_bwd_finalize_s": [_i, _p, _i, _l, _i, _p, _p, _p, _p, _p, _p, _i],
    "nvae_bn_bwd_apply": [_i, _p, _p, _p, _l, _i, _p, _p, _p, _i, _i],
    "nvae_se_pool": [_i, _p, _i, _i, _i, _p],
    "nvae_se_gate": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "nvae_se_pool_gate": [_i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "nvae_se_apply": [_i, _p, _p, _p, _i, _i, _i, _p, _f, _f],
    "nvae_se_apply_stats": [_i, _p, _p, _p, _i, _i, _i, _p, _f, _f, _p],
    "nvae_se_bwd_reduce": [_i, _p, _p, _i, _i, _i, _p],
    "nvae_se_gate_bwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p],
    "nvae_se_reduce_gate_bwd": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _p, _p],
    "nvae_se_wgrad": [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p],
    "nvae_se_wgrad_batched": [_i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p],
    "nvae_se_bwd_apply": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _i],
    "nvae_se_bwd_apply_bn": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p, _p, _p, _i, _p],
    "nvae_se_fused_rows": None,
    "nvae_se_set_workspace": None,
    "nvae_se_force_split": None,
    "nvae_se_fused_fwd": [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p],
    "nvae_se_fused_bwd": [_i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _f, _f, _i, _i, _p, _p],
    "nvae_unary_fwd": [_i, _i, _p, _p, _l, _f, _f],
    "nvae_unary_bwd": [_i, _i, _p, _p, _p, _l, _i],
    "nvae_add": [_i, _p, _p, _l, _i],
    "nvae_cast": [_i, _i, _p, _p, _l],
    "nvae_upsample_pool_bwd": [_i, _p, _p, _i, _i, _i, _i, _i, _i],
    "nvae_randn": [_p, _l, C.c_ulonglong, _p],
    "nvae_sampler_fwd": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i],
    "nvae_sampler_bwd": [_i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i],
    "nvae_sampler_bwd_scaled": [_i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _i, _p, _i],
    "nvae_grad_amax": [_i, _p, _l, _p],
    "nvae_grad_rescale": [_i, _p, _l, _p, _p, _i, _i, _f],
    "nvae_grad_merge": [_i, _p, _p, _l, _p, _i, _i, _i],
    "nvae_grad_unscale": [_p, _p, _i, _p],
    "nvae_bernoulli_fwd": [_i, _p, _p, _p, _i, _i, _i, _i, _i],
    "nvae_bernoulli_bwd": [_i, _p, _p, _p, _l, _f, _p],
    "nvae_dmol_fwd": [_p, _i, _p, _p, _i, _i, _i],
    "nvae_dmol_bwd": [_i, _p, _i, _p, _p, _i, _i, _i, _f, _p],
    "nvae_dmol_sample": [_p, _i, _p, _p, _p, _i, _i, _i, _f],
    "nvae_kl_absmean": [_p, _i, _i, _p],
    "nvae_loss_finalize": [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p],
    "nvae_bn_absmax_fwd": [_p, _p, _i, _f, _p, _p],
    "nvae_bn_absmax_bwd": [_p, _p, _p, _p, _i, _f, _p],
    "nvae_adamax": [_p, _p, _p, _p, _l, _p, _f, _f, _f],
    "nvae_grad_guard": [_p, _l, _p],
    "nvae_loss_scale_update": [_p, _f, _f],
    "nvae_sn_power_iter": [_p, _p, _i, _i, _p, _p, _p, _p, _p],
    "nvae_weight_prep": [_i, _p, _p, _i, _i, _p, _p],
}
EXPORTS = ["nvae_last_error", "nvae_abi_version", *_SIGS.keys()]

_lib = None


def load():
    """Load the shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -m nvae_tf_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the NVAE hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.nvae_last_error.restype = C.c_char_p
    lib.nvae_last_error.argtypes = []
    lib.nvae_abi_version.restype = C.c_int
    lib.nvae_abi_version.argtypes = []
    have = lib.nvae_abi_version()
    if have != ABI_VERSION:
        raise RuntimeError(
            f"{LIB_PATH} is stale: it reports ABI version {have}, this package binds version {ABI_VERSION}. "
            "Rebuild with `python -m nvae_tf_amd.build --force`.")
    lib.nvae_reduce_splits.restype = C.c_int
    lib.nvae_reduce_splits.argtypes = [_l, _i]
    lib.nvae_se_fused_rows.restype = C.c_int
    lib.nvae_se_fused_rows.argtypes = [_i, _i, _i]
    lib.nvae_se_set_workspace.restype = C.c_int
    lib.nvae_se_set_workspace.argtypes = [_p, C.c_size_t, _p, _i]
    lib.nvae_se_force_split.restype = C.c_int
    lib.nvae_se_force_split.argtypes = [_i]
    lib.nvae_dwconv5_stats_rows.restype = C.c_int
    lib.nvae_dwconv5_stats_rows.argtypes = [_i, _i, _i, _i, _i]
    lib.nvae_conv_gemm_stats_rows.restype = C.c_int
    lib.nvae_conv_gemm_stats_rows.argtypes = [_i, _G]
    lib.nvae_conv_gemm_force_tile.restype = C.c_int
    lib.nvae_conv_gemm_force_tile.argtypes = [_i]
    lib.nvae_conv_gemm_force_split.restype = C.c_int
    lib.nvae_conv_gemm_force_split.argtypes = [_i]
    lib.nvae_conv_set_workspace.restype = C.c_int
    lib.nvae_conv_set_workspace.argtypes = [_p, C.c_size_t, _p, _i]
    lib.nvae_conv_img_ok.restype = C.c_int
    lib.nvae_conv_img_ok.argtypes = [_i, _G]
    lib.nvae_conv_img_enable.restype = C.c_int
    lib.nvae_conv_img_enable.argtypes = [_i]
    lib.nvae_set_deterministic.restype = C.c_int
    lib.nvae_set_deterministic.argtypes = [_i]
    lib.nvae_get_deterministic.restype = C.c_int
    lib.nvae_get_deterministic.argtypes = []
    lib.nvae_conv_gemm_family.restype = C.c_int
    lib.nvae_conv_gemm_family.argtypes = [_i, C.c_void_p]
    lib.nvae_conv_halo4_enable.restype = C.c_int
    lib.nvae_conv_halo4_enable.argtypes = [_i]
    lib.nvae_conv_halo_stamps.restype = C.c_int
    lib.nvae_conv_halo_stamps.argtypes = [C.c_void_p, _i]
    lib.nvae_conv_gemm_pre_max_cin.restype = C.c_int
    lib.nvae_conv_gemm_pre_max_cin.argtypes = [_i, _G]
    lib.nvae_conv_wgrad_scratch.restype = C.c_long
    lib.nvae_conv_wgrad_scratch.argtypes = [_i, _G]
    lib.nvae_conv_wgrad_scratch_n.restype = C.c_long
    lib.nvae_conv_wgrad_scratch_n.argtypes = [_i, _G, _i]
    for name, sig in _SIGS.items():
        if sig is None:
            continue
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [*sig, _p]
    # NVAE_DETERMINISTIC=1: ordered cross-workgroup sums everywhere (include/nvae_hip.h, nvae_set_deterministic); set before any
    # slab is sized.  Bit-identical steps for identical inputs and state, at a price in speed.
    if os.environ.get("NVAE_DETERMINISTIC", "0") not in ("", "0"):
        lib.nvae_set_deterministic(1)
    _lib = lib
    return lib


_workspace = {}


def ensure_workspace(device) -> None:
    """Register the split-K workspace of the implicit-GEMM kernel (include/nvae_hip.h nvae_conv_set_workspace) for
    this process: 32 MB of partial-tile slabs + 4096 zeroed arrival counters, allocated once and never freed (the
    library keeps the raw pointers).  One process drives one GPU and issues its convolutions on one stream."""
    dev = torch.device(device)
    if dev.type != "cuda" or "ws" in _workspace:
        return
    slab = torch.empty(32 << 20, dtype=torch.uint8, device=dev)
    counters = torch.zeros(4096, dtype=torch.int32, device=dev)
    rc = load().nvae_conv_set_workspace(slab.data_ptr(), slab.numel(), counters.data_ptr(), counters.numel())
    if rc != 0:
        raise RuntimeError(f"nvae_conv_set_workspace failed: {load().nvae_last_error().decode()}")
    se_buf = torch.empty(8 << 20, dtype=torch.uint8, device=dev)       # SE image-split hand-off vectors
    se_counters = torch.zeros(256, dtype=torch.int32, device=dev)
    rc = load().nvae_se_set_workspace(se_buf.data_ptr(), se_buf.numel(), se_counters.data_ptr(), se_counters.numel())
    if rc != 0:
        raise RuntimeError(f"nvae_se_set_workspace failed: {load().nvae_last_error().decode()}")
    _workspace["ws"] = (slab, counters)
    _workspace["se"] = (se_buf, se_counters)


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Invoke an entry point on the current torch stream; raise on a non-zero return code."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {lib.nvae_last_error().decode()}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise ValueError(f"unsupported activation dtype {dt}")
