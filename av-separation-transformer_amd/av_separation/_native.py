"""ctypes binding of libavsep_hip.so (C ABI: include/avsep.h).

The HIP library IS the product path.  There is no CPU or eager-PyTorch fallback anywhere in this
package: if the shared library is missing or an entry point fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(os.path.dirname(_HERE), "lib")
# The developer build of the same sources (make dev): tile overrides, A/B switches, diagnostics and the kernel instances that
# were measured slower.  AVSEP_LIB=dev makes it the library of the whole process (tools/ sweeps); the bit-identity tests
# open it NEXT to the product library with load_dev().
DEV_LIB_PATH = os.path.join(_LIBDIR, "libavsep_hip_dev.so")
_want = os.environ.get("AVSEP_LIB")
LIB_PATH = DEV_LIB_PATH if _want == "dev" else (_want or os.path.join(_LIBDIR, "libavsep_hip.so"))

# every symbol include/avsep.h declares (tests check the library exports all of them)
ABI_SYMBOLS = (
    "avsep_abi_version", "avsep_last_error", "avsep_build_id", "avsep_create", "avsep_create_ex", "avsep_destroy", "avsep_set_weight",
    "avsep_finalize_weights", "avsep_workspace_bytes", "avsep_forward", "avsep_forward_graph",
    "avsep_audio_encoder", "avsep_visual_encoder", "avsep_fusion", "avsep_decoder",
    "avsep_set_debug_taps", "avsep_set_split_precision", "avsep_read_tap", "avsep_profile_begin", "avsep_profile_end", "avsep_op_linear", "avsep_op_linear_split", "avsep_op_split_planes", "avsep_op_linear_planes", "avsep_op_layernorm_planes", "avsep_op_interp_linear_planes", "avsep_op_attention_split_planes", "avsep_op_h2_row_stats", "avsep_op_split_h2", "avsep_op_linear_h2", "avsep_op_interp_linear_h2", "avsep_op_layernorm_h2", "avsep_op_attention_split_h2", "avsep_op_layernorm", "avsep_op_ln_linear",
    "avsep_op_attention", "avsep_op_attention_split", "avsep_op_attention_h2", "avsep_op_interp_linear", "avsep_stft_basis_floats", "avsep_stft_basis", "avsep_op_stft_mag",
    # training ops
    "avsep_op_linear_ex", "avsep_op_attention_train", "avsep_op_attention_bwd", "avsep_op_transpose",
    "avsep_op_transpose_pad", "avsep_op_im2col1d", "avsep_op_col2im1d", "avsep_op_im2col2d", "avsep_op_col2im2d",
    "avsep_op_colreduce_scratch_floats", "avsep_op_colreduce", "avsep_op_bn_train_fwd", "avsep_op_bn_train_bwd",
    "avsep_op_act_fwd", "avsep_op_act_bwd", "avsep_op_mul_mixed", "avsep_op_add_rows", "avsep_read_stamps", "avsep_op_dropout", "avsep_op_dropout_add", "avsep_op_wgrad_scratch_floats", "avsep_op_wgrad", "avsep_op_wgrad_direct_scratch_floats",
    "avsep_op_wgrad_direct", "avsep_op_wgrad_bias_direct_scratch_floats", "avsep_op_wgrad_bias_direct", "avsep_op_transpose_many",
    "avsep_op_bn_stats", "avsep_op_bn_apply", "avsep_op_bn_bwd_sums", "avsep_op_bn_bwd_dx", "avsep_op_avgpool_fwd", "avsep_op_avgpool_bwd",
    "avsep_op_interp_linear_bwd", "avsep_op_layernorm_bwd", "avsep_op_layernorm_bwd_res", "avsep_op_linear_drop", "avsep_op_linear_split_ex", "avsep_op_wgrad_direct_split",
    "avsep_op_relu_dropout_bwd",
)


class AvsepConfig(C.Structure):
    _fields_ = [("freq_bins", C.c_int32), ("d_model", C.c_int32), ("nhead", C.c_int32),
                ("num_encoder_layers", C.c_int32), ("num_fusion_layers", C.c_int32),
                ("num_speakers", C.c_int32)]


_lib = None
_dev = None


def load():
    """Load (once) and type the shared library; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def load_dev():
    """The developer build as a second handle (own static state, own contexts): for op-level tests and tools only."""
    global _dev
    if _dev is None:
        _dev = _open(DEV_LIB_PATH)
    return _dev


def _open(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP extension first "
            "(`python -c 'import __graft_entry__ as g; g.build()'` at the repo root, or "
            "`make -C av-separation-transformer_amd/csrc`).  There is no fallback path.")
    lib = C.CDLL(path)
    p, i, i64, sz, fp = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_void_p
    lib.avsep_abi_version.restype = i
    lib.avsep_last_error.restype = C.c_char_p
    lib.avsep_build_id.restype = C.c_char_p
    lib.avsep_create.argtypes = [C.POINTER(AvsepConfig), C.POINTER(p)]
    lib.avsep_create_ex.argtypes = [C.POINTER(AvsepConfig), C.c_uint32, C.POINTER(p)]
    lib.avsep_destroy.argtypes = [p]
    lib.avsep_destroy.restype = None
    lib.avsep_set_weight.argtypes = [p, C.c_char_p, fp, C.POINTER(i64), i]
    lib.avsep_finalize_weights.argtypes = [p, p]
    lib.avsep_workspace_bytes.argtypes = [p, i, i, i, i, i]
    lib.avsep_workspace_bytes.restype = sz
    fwd = [p, fp, fp, fp, fp, p, sz, i, i, i, i, i, p]
    lib.avsep_forward.argtypes = fwd
    lib.avsep_forward_graph.argtypes = fwd
    lib.avsep_audio_encoder.argtypes = [p, fp, fp, p, sz, i, i, p]
    lib.avsep_visual_encoder.argtypes = [p, fp, fp, p, sz, i, i, i, i, i, p]
    lib.avsep_fusion.argtypes = [p, fp, fp, fp, p, sz, i, i, p]
    lib.avsep_decoder.argtypes = [p, fp, fp, fp, fp, p, sz, i, i, p]
    lib.avsep_set_debug_taps.argtypes = [p, i]
    lib.avsep_read_tap.argtypes = [p, C.c_char_p, fp, i64, p, i, i, i, i, i, p]
    lib.avsep_read_tap.restype = i64
    if hasattr(lib, "avsep_set_schedule"):          # developer build only (chained schedules: measured slower)
        lib.avsep_set_schedule.argtypes = [p, i, i, C.c_float]
        lib.avsep_chain_status.argtypes = [p, p]
        lib.avsep_chain_peek.argtypes = [p, i, C.POINTER(C.c_uint32), i]
    lib.avsep_profile_begin.argtypes = [p]
    lib.avsep_profile_end.argtypes = [p, C.c_char_p, sz]
    lib.avsep_profile_end.restype = i64
    lib.avsep_op_linear.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, p]
    lib.avsep_op_linear_split.argtypes = [fp, fp, fp, fp, fp, i, i, i, i, p]
    lib.avsep_op_split_planes.argtypes = [fp, i, p, i64, i, i, p]
    lib.avsep_op_linear_planes.argtypes = [p, i64, p, i64, fp, fp, fp, p, i64, i, i, i, i, p]
    lib.avsep_op_layernorm_planes.argtypes = [fp, fp, fp, p, i64, i, i, C.c_float, p]
    lib.avsep_op_interp_linear_planes.argtypes = [fp, p, i64, i, i, i, i, p]
    lib.avsep_op_attention_split_planes.argtypes = [fp, i, fp, i, fp, i, p, i64, i, i, i, i, i, p]
    lib.avsep_op_h2_row_stats.argtypes = [fp, i, i, p, fp, p]
    lib.avsep_op_split_h2.argtypes = [fp, i, p, i64, i, i, p, i, p]
    lib.avsep_op_linear_h2.argtypes = [p, i64, p, i64, fp, fp, fp, fp, fp, p, i64, i, i, i, i, i, p]
    lib.avsep_op_interp_linear_h2.argtypes = [fp, p, fp, i64, i, i, i, i, p]
    lib.avsep_op_layernorm_h2.argtypes = [fp, fp, fp, p, i64, i, i, C.c_float, i, p]
    lib.avsep_op_attention_split_h2.argtypes = [fp, i, fp, i, fp, i, p, i64, i, i, i, i, i, i, p]
    lib.avsep_op_attention_h2.argtypes = [fp, i, fp, i, fp, i, fp, i, i, i, i, i, i, i, i, i, p]
    lib.avsep_set_split_precision.argtypes = [p, i]
    lib.avsep_op_layernorm.argtypes = [fp, fp, fp, fp, i, i, C.c_float, p]
    lib.avsep_op_ln_linear.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, i, i, i, C.c_float, i, p]
    lib.avsep_op_attention.argtypes = [fp, i, fp, i, fp, i, fp, i, i, i, i, i, i, p]
    lib.avsep_op_attention_split.argtypes = [fp, i, fp, i, fp, i, fp, i, i, i, i, i, i, p]
    lib.avsep_op_interp_linear.argtypes = [fp, fp, i, i, i, i, p]
    if hasattr(lib, "avsep_op_mask_head"):          # developer build only
        lib.avsep_op_mask_head.argtypes = [fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, i, p]
    if hasattr(lib, "avsep_op_attention_proj"):     # developer build only
        lib.avsep_op_attention_proj.argtypes = [fp, i, fp, i, fp, i, fp, fp, fp, i, i, i, i, i, p]
    if hasattr(lib, "avsep_op_linear_pair"):        # developer build only (paired-launch experiment)
        lib.avsep_op_linear_pair.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, fp, fp, fp, fp, fp, fp, fp, i, i, i, i, C.c_float, p]
        lib.avsep_op_attention_pair.argtypes = [fp, fp, fp, fp, i, i, i, i, fp, fp, fp, fp, i, i, i, i, i, i, p]
    lib.avsep_stft_basis_floats.argtypes = [i]
    lib.avsep_stft_basis_floats.restype = i64
    lib.avsep_stft_basis.argtypes = [fp, i, p]
    lib.avsep_op_stft_mag.argtypes = [fp, fp, fp, i, i, i, i, p]
    f = C.c_float
    lib.avsep_op_linear_ex.argtypes = [fp, i, fp, i, fp, fp, i, i, fp, i, i, i, i, i, p]
    u64 = C.c_uint64
    lib.avsep_op_attention_train.argtypes = [fp, i, fp, i, fp, i, fp, i, fp, i, i, i, i, i, f, f, u64, p]
    lib.avsep_op_attention_bwd.argtypes = [fp, i, fp, i, fp, i, fp, i, fp, i, fp, fp, fp, i, fp, i, fp, i, i, i, i, i, i, f,
                                           f, u64, p]
    lib.avsep_read_stamps.argtypes = [p, C.POINTER(C.c_uint64), i]
    lib.avsep_op_dropout.argtypes = [fp, fp, i64, f, u64, p]
    lib.avsep_op_dropout_add.argtypes = [fp, fp, fp, i64, f, u64, p]
    lib.avsep_op_wgrad_scratch_floats.argtypes = [i, i, i]
    lib.avsep_op_wgrad_scratch_floats.restype = C.c_int64
    lib.avsep_op_wgrad.argtypes = [fp, fp, fp, fp, i, i, i, p]
    lib.avsep_op_wgrad_direct_scratch_floats.argtypes = [i, i, i]
    lib.avsep_op_wgrad_direct_scratch_floats.restype = C.c_int64
    lib.avsep_op_wgrad_direct.argtypes = [fp, i, fp, i, fp, fp, i, i, i, p]
    lib.avsep_op_wgrad_bias_direct_scratch_floats.argtypes = [i, i, i]
    lib.avsep_op_wgrad_bias_direct_scratch_floats.restype = C.c_int64
    lib.avsep_op_wgrad_bias_direct.argtypes = [fp, i, fp, i, fp, fp, i, i, i, p]
    if hasattr(lib, "avsep_op_wgrad_merged"):       # developer build only
        lib.avsep_op_wgrad_tiles.argtypes = [i, i, i]
        lib.avsep_op_wgrad_tiles.restype = C.c_int64
        lib.avsep_op_wgrad_merged.argtypes = [fp, i, fp, i, fp, fp, fp, i, i, i, i, p]
    lib.avsep_op_transpose_many.argtypes = [p, i, i, i, p]
    lib.avsep_op_bn_stats.argtypes = [fp, fp, fp, fp, i, i, p]
    lib.avsep_op_bn_apply.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, i, f, i, p]
    lib.avsep_op_bn_bwd_sums.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, i, i, p]
    lib.avsep_op_bn_bwd_dx.argtypes = [fp, fp, fp, fp, fp, fp, fp, i, i, f, f, p]
    lib.avsep_op_transpose.argtypes = [fp, fp, i, i, i, p]
    lib.avsep_op_transpose_pad.argtypes = [fp, fp, i, i, i, i, p]
    lib.avsep_op_im2col1d.argtypes = [fp, fp, i, i, i, p]
    lib.avsep_op_col2im1d.argtypes = [fp, fp, i, i, i, p]
    lib.avsep_op_im2col2d.argtypes = [fp, fp, i, i, i, i, i, p]
    lib.avsep_op_col2im2d.argtypes = [fp, fp, i, i, i, i, i, p]
    lib.avsep_op_colreduce_scratch_floats.argtypes = [i, i]
    lib.avsep_op_colreduce_scratch_floats.restype = i64
    lib.avsep_op_colreduce.argtypes = [fp, fp, fp, fp, fp, i, i, p]
    lib.avsep_op_bn_train_fwd.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, f, f, i, p]
    lib.avsep_op_bn_train_bwd.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, f, i, p]
    lib.avsep_op_act_fwd.argtypes = [fp, fp, i64, i, p]
    lib.avsep_op_act_bwd.argtypes = [fp, fp, fp, i64, i, p]
    lib.avsep_op_mul_mixed.argtypes = [fp, fp, fp, i64, i, i, i, p]
    lib.avsep_op_add_rows.argtypes = [fp, fp, fp, i64, i, i, p]
    lib.avsep_op_avgpool_fwd.argtypes = [fp, fp, i, i, i, p]
    lib.avsep_op_avgpool_bwd.argtypes = [fp, fp, i, i, i, p]
    lib.avsep_op_interp_linear_bwd.argtypes = [fp, fp, i, i, i, i, p]
    lib.avsep_op_layernorm_bwd.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, i, i, f, p]
    lib.avsep_op_layernorm_bwd_res.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, f, p]
    lib.avsep_op_linear_drop.argtypes = [fp, i, fp, i, fp, fp, i, i, fp, i, i, i, i, f, u64, p]
    lib.avsep_op_linear_split_ex.argtypes = [fp, i, fp, i, fp, fp, i, i, fp, i, i, i, i, i, f, u64, p]
    lib.avsep_op_wgrad_direct_split.argtypes = [fp, i, fp, i, fp, fp, i, i, i, i, p]
    lib.avsep_op_relu_dropout_bwd.argtypes = [fp, fp, fp, i64, f, p]
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)     # AttributeError here = ABI drift between header and library
        if fn.restype is C.c_int and name not in ("avsep_abi_version", "avsep_build_id"):
            fn.restype = i
    if lib.avsep_abi_version() != 1:
        raise RuntimeError(f"{path}: ABI version mismatch")
    return lib


def check(rc, what="avsep call"):
    if rc is not None and rc < 0:
        msg = load().avsep_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else 'unknown error'}")
    return rc
