"""ctypes binding of ``csrc/libnarfs2.so`` (C-ABI: ``include/nar_fs2.h``).

There is no CPU fallback: if the library is missing or a symbol is absent the
import fails loudly.  ``torch`` is imported first on purpose — it brings its own
``libamdhip64.so.7`` and the library must bind to that same HIP runtime so that
torch's device pointers and streams are valid in our launches.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: one HIP runtime per process)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnarfs2.so")


class NsConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_vocab", "max_seq_len", "d_enc", "n_enc_layer", "n_enc_head", "d_dec", "n_dec_layer", "n_dec_head",
        "d_inner", "ffn_k1", "ffn_k2", "vp_filter", "vp_kernel", "n_bins", "n_mel",
        "postnet_dim", "postnet_k", "postnet_n", "pitch_frame_level", "energy_frame_level", "length_regulator",
        "matmul_bf16x3", "row_epilogue", "phase1_packing")]


_P, _I, _F, _Z, _S = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_char_p

# name -> (restype, argtypes); must list every symbol include/nar_fs2.h declares
SIGNATURES = {
    "ns_last_error": (C.c_char_p, []),
    "ns_create": (_I, [C.POINTER(NsConfig), C.POINTER(_P)]),
    "ns_destroy": (None, [_P]),
    "ns_arena_bytes": (_Z, [_P]),
    "ns_bind_arena": (_I, [_P, _P, _Z]),
    "ns_set_weight": (_I, [_P, _S, _P, C.POINTER(C.c_int64), _I]),
    "ns_check_weight": (_I, [_P, _S, C.POINTER(C.c_int64), _I]),
    "ns_finalize_weights": (_I, [_P, _P]),
    "ns_adopt_arena": (_I, [_P]),
    "ns_encoder_ws_bytes": (_Z, [_P, _I, _I]),
    "ns_decoder_ws_bytes": (_Z, [_P, _I, _I, _I]),
    "ns_forward_durations": (_I, [_P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ns_forward_durations_packed": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ns_last_phase1_rows": (C.c_int64, [_P]),
    "ns_upload_lengths": (_I, [_P, _I, _P, _P]),
    "ns_forward_mel": (_I, [_P, _I, _I, _I, _P, _F, _F, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _P]),
    "ns_forward_mel_packed": (_I, [_P, _I, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P, _P, _P]),

    "ns_last_phase2_rows": (C.c_int64, [_P]),
    "ns_plan_gemm": (_I, [_I, _I, _I, _I, C.POINTER(C.c_int32)]),
    "ns_plan_row_tile": (_I, [_I, _I]),
    "ns_plan_row_tile_k": (_I, [_I, _I, _I]),
    "ns_acc_chunk": (_I, []),
    "ns_abi_version": (_I, []),
    "ns_plan_attention_split": (_I, [_I, _I, _I, _I]),
    "ns_op_ws_bytes": (_Z, [_P, _I, _I]),
    "ns_op_mask_from_lengths": (_I, [_P, _I, _I, _P, _P]),
    "ns_op_sinusoid_table": (_I, [_I, _I, _P, _P]),
    "ns_op_txt_encoder": (_I, [_P, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_multi_head_attention": (_I, [_P, _S, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_positionwise_ffn": (_I, [_P, _S, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_fft_block": (_I, [_P, _S, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_variance_predictor": (_I, [_P, _S, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_duration_round": (_I, [_P, _I, _F, _P, _P]),
    "ns_op_duration_scan": (_I, [_P, _I, _I, _P, _P, _P]),
    "ns_op_length_regulate": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "ns_op_variance_embedding": (_I, [_P, _I, _P, _P, _I, _I, _F, _P, _P, _P, _P, _Z, _P]),
    "ns_op_bucketize": (_I, [_P, _I, _P, _I, _P, _P]),
    "ns_op_gaussian_upsampling": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ns_op_mel_decoder": (_I, [_P, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_mel_linear": (_I, [_P, _P, _I, _I, _P, _P]),
    "ns_op_postnet": (_I, [_P, _P, _I, _I, _P, _P, _Z, _P]),
    "ns_op_ffn_conv1": (_I, [_P, _S, _P, _I, _I, _P, _P]),
    "ns_op_attention_core": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "ns_profile_enable": (_I, [_P, _I]),
    "ns_profile_read": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ns_profile_read_slot": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

STATUS_TRUNCATED, STATUS_BAD_TOKEN = 1, 2  # include/nar_fs2.h NS_STATUS_*

_lib = None


def load():
    """Load the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ns_last_error()
        raise RuntimeError(f"{what}: {msg.decode() if msg else 'error'} (rc={rc})")


def ptr(t) -> C.c_void_p:
    """Device (or host) pointer of a contiguous tensor; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor must be contiguous"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
