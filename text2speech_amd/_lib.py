"""ctypes binding of libt2s_hip.so (the C ABI declared in include/t2s_hip.h).

The product path has NO fallback: if the library is missing or an entry point
fails, a ``T2SError`` is raised.  PyTorch is used only for device memory and
streams; every pointer handed to the library is ``tensor.data_ptr()``.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2S_LIB_PATH") or os.path.join(HERE, "libt2s_hip.so")      # (override: diagnostic builds)

c_int, c_float, c_vp, c_long = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_long

# name -> argtypes (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    "t2s_abi_version": [],
    "t2s_operand_format": [],
    "t2s_sizeof_taco_decoder": [],
    "t2s_sizeof_taco_bptt": [],
    "t2s_error_string": [c_int],
    "t2s_last_hip_error": [],
    "t2s_plane_rows": [c_int, c_int],
    "t2s_padded_rows": [c_int],
    "t2s_pack_conv_weight": [c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_vp, c_vp, c_vp, c_int, c_vp],
    "t2s_pack_conv_weight_table": [c_vp, c_int, c_long, c_vp],
    "t2s_weightnorm_small": [c_vp, c_vp, c_int, c_int, c_vp, c_vp],
    "t2s_wg_upsample_squeeze": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_vp, c_vp, c_vp],
    "t2s_wg_audio_squeeze": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_wg_convinv": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_small_logdet_inv": [c_vp, c_int, c_float, c_vp, c_vp, c_vp],
    "t2s_small_logdet_inv_batch": [c_vp, c_int, c_float, c_vp],
    "t2s_small_logdet_inv_batch_host": [c_vp, c_int, c_float, c_vp],
    "t2s_wg_start": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "t2s_wg_in_cond_gate": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_int, c_vp],
    "t2s_wg_res_skip": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int,
                        c_int, c_int, c_vp],
    "t2s_wg_endfold_weights": [c_vp, c_int, c_int, c_vp],
    "t2s_wg_in_cond_gate_fold": [c_vp] * 11 + [c_int] * 10 + [c_vp],
    "t2s_wg_gate_fold_slots": [c_int, c_int, c_int],
    "t2s_wg_gate_tile_rows": [c_int, c_int, c_int],
    "t2s_wg_upsample_basis": [c_vp, c_vp] + [c_int] * 6 + [c_vp, c_vp, c_vp],
    "t2s_wg_compose_cond": [c_vp, c_vp] + [c_int] * 4 + [c_long, c_vp, c_vp, c_vp, c_vp],
    "t2s_wg_melwin_planes": [c_vp] + [c_int] * 5 + [c_vp, c_vp, c_vp],
    "t2s_wg_in_melwin_gate_fold": [c_vp] * 13 + [c_int] * 12 + [c_vp],
    "t2s_wg_res_only": [c_vp] * 7 + [c_int] * 7 + [c_vp],
    "t2s_wg_end_fold_affine": [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_wg_end_affine": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_vp],
    "t2s_conv_bias_act": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_int, c_int, c_int, c_vp],
    "t2s_gemv": [c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_int, c_long, c_vp, c_int, c_long, c_vp, c_int, c_long,
                 c_vp, c_vp, c_vp, c_long, c_long, c_int, c_int, c_int, c_vp, c_long, c_float, c_vp],
    "t2s_transpose": [c_vp, c_vp, c_int, c_int, c_vp],
    "t2s_embed_planes": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "t2s_f32_to_planes": [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "t2s_taco_parse_output": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "t2s_bn_fold": [c_vp, c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_vp, c_vp],
    "t2s_bn_train": [c_vp, c_vp, c_vp, c_float, c_int, c_vp, c_float, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                     c_vp, c_vp, c_vp],
    "t2s_taco_encoder_lstm": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "t2s_taco_encoder_lstm_bwd": [c_vp] * 9 + [c_int] * 4 + [c_vp],
    "t2s_taco_encoder_lstm_split": [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, ctypes.c_uint, c_vp],
    "t2s_taco_lstm_xbuf_bytes": [c_int],
    "t2s_taco_encoder_lstm_bwd_split": [c_vp] * 9 + [c_int] * 4 + [c_vp, ctypes.c_uint, c_vp],
    "t2s_rows_to_planes": [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "t2s_embedding_grad": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "t2s_bn_running_update": [c_vp, c_vp, c_vp, c_vp, c_vp, c_float, ctypes.c_longlong, c_int, c_vp],
    "t2s_zero_fill": [c_vp, ctypes.c_size_t, c_vp],
    "t2s_wg_bwd_pair8_ok": [c_int, c_int, c_int],
    "t2s_taco_attention": [c_vp] * 14 + [c_int] * 7 + [c_vp],
    "t2s_bernoulli_mask": [c_vp, ctypes.c_size_t, ctypes.c_ulonglong, ctypes.c_ulonglong, c_float, c_vp],
    "t2s_taco_decode_steps": [c_vp, c_int, c_int, c_vp],
    "t2s_taco_stop_check": [c_vp, c_int, c_int, c_int, c_int, c_int, c_float, c_vp, c_vp],
    "t2s_rows_to_tm": [c_vp, c_long, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp],
    "t2s_rows_to_tm_batched": [c_vp, c_long, c_long, c_int, c_int, c_int, c_int, c_vp, c_vp, c_long, c_int, c_int, c_int, c_vp],
    "t2s_lstm_cell_bwd": [c_vp, c_long, c_vp, c_long, c_vp, c_long, c_vp, c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int,
                          c_vp],
    "t2s_relu_drop_bwd": [c_vp, c_vp, c_float, ctypes.c_size_t, c_vp, c_vp],
    "t2s_taco_att_bwd": [c_vp, c_vp],
    "t2s_taco_bptt_steps": [c_vp, c_int, c_int, c_vp],
    "t2s_bn_bwd": [c_vp, c_vp, c_vp],
    "t2s_waveglow_loss": [c_vp, ctypes.c_size_t, c_vp, c_vp, c_int, c_vp, c_float, c_vp, c_vp, c_vp, c_vp],
    "t2s_taco_loss": [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp],
    "t2s_stft_transform": [c_vp, c_int, c_int, c_vp, c_int, c_int, c_vp, c_long, c_vp, c_vp, c_vp, c_vp, c_long, c_vp],
    "t2s_mel_from_mag": [c_vp, c_long, c_int, c_int, c_vp, c_int, c_float, c_vp, c_vp],
    "t2s_stft_inverse": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_long, c_vp, c_float, c_vp, c_float, c_vp, c_vp, c_vp,
                         c_vp],
    "t2s_sum_axis0": [c_vp, c_int, c_int, c_vp, c_vp],
    "t2s_add3": [c_vp, c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp],
    "t2s_scale_by_scalar": [c_vp, ctypes.c_size_t, c_vp, c_float, c_vp, c_vp],
    "t2s_planes_to_f32": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp],
    "t2s_wg_in_cond_gate_train": [c_vp] * 13 + [c_int] * 9 + [c_vp],
    "t2s_wg_res_skip_train": [c_vp] * 10 + [c_int] * 8 + [c_vp],
    "t2s_wg_bwd_gate_dgrad": [c_vp] * 11 + [c_int] + [c_vp] * 2 + [c_int] * 8 + [c_vp],
    "t2s_wg_in_cond_gate_fold_train": [c_vp] * 11 + [c_int] + [c_vp] * 2 + [c_int] * 10 + [c_vp],
    "t2s_wg_res_only_train": [c_vp] * 5 + [c_int] + [c_vp] * 4 + [c_int] * 7 + [c_vp],
    "t2s_wg_skip_sum": [c_vp] * 5 + [c_int] * 2 + [c_vp] + [c_int] * 6 + [c_vp],
    "t2s_conv_accumulate": [c_vp] * 5 + [c_int] + [c_vp] * 2 + [c_int] * 11 + [c_vp],
    "t2s_wgrad_gemm": [c_vp] * 6 + [c_int] * 9 + [c_vp],
    "t2s_wgrad_gemm_flat": [c_vp] * 6 + [c_int] * 9 + [c_vp],
    "t2s_wgrad_cl": [c_vp, c_int, c_vp, c_int, c_vp] + [c_int] * 8 + [c_vp],
    "t2s_plane_transpose": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp],
    "t2s_tm_ones_row": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_pack_transposed": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_vp],
    "t2s_weightnorm_scale": [c_vp, c_vp, c_int, c_int, c_vp, c_vp],
    "t2s_wn_backward": [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp,
                        c_vp, c_int, c_vp],
    "t2s_wg_affine_backward": [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_small_wgrad_scratch": [c_int, c_int],
    "t2s_small_wgrad": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                        c_int, c_vp],
    "t2s_rows_sum": [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "t2s_wg_start_dgrad": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "t2s_wg_convinv_wgrad": [c_vp, c_vp, c_vp, c_vp, c_float, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "t2s_wg_upsample_wgrad": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp,
                              c_vp],
    "t2s_adam_table": [c_vp, c_int, c_long, c_float, c_float, c_float, c_float, c_int, c_float, c_float, c_vp],
}
_RESTYPE = {"t2s_error_string": ctypes.c_char_p, "t2s_last_hip_error": ctypes.c_char_p,
            "t2s_small_wgrad_scratch": ctypes.c_long, "t2s_taco_lstm_xbuf_bytes": ctypes.c_long}


class T2SError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once).  Raises T2SError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise T2SError(
            "libt2s_hip.so not found at %s - build it with `python -m text2speech_amd.build` "
            "(there is no CPU fallback on the product path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    diagnostic = bool(os.environ.get("T2S_LIB_PATH"))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if diagnostic:
                # a diagnostic build (explicit T2S_LIB_PATH) may predate the newest entry points: it loads, and CALLING a missing
                # one raises (`call` below).  The shipped library must export every declared symbol.
                continue
            raise T2SError("libt2s_hip.so does not export %s" % name) from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    return c_vp(t.data_ptr())


HOST_TIMES = {}     # T2S_HOST_TIMING=1: seconds the host spent inside each entry point (enqueue cost; diagnostic)
_HOST_TIMING = bool(os.environ.get("T2S_HOST_TIMING"))


def call(name, *args):
    """Call an int-returning entry point; raise on a non-zero code."""
    lib = load()
    if _HOST_TIMING:
        import time
        t0 = time.perf_counter()
        rc = getattr(lib, name)(*args)
        c = HOST_TIMES.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += time.perf_counter() - t0
    else:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise T2SError("%s does not export %s (a diagnostic build older than the entry point?)" % (LIB_PATH, name)) from e
        rc = fn(*args)
    if rc != 0:
        msg = lib.t2s_error_string(rc).decode()
        hip = lib.t2s_last_hip_error().decode()
        raise T2SError("%s failed: %s%s" % (name, msg, (" (" + hip + ")") if hip and rc == -2 else ""))
    return rc


def operand_format():
    """0: split-bf16 planes (the shipped library); 1: split-fp16 planes (diagnostic build, T2S_LIB_PATH=build/f16x3/...)."""
    return load().t2s_operand_format()


def plane_rows(L, halo):
    return load().t2s_plane_rows(L, halo)


def padded_rows(rows):
    return load().t2s_padded_rows(rows)


def current_stream():
    import torch
    return c_vp(torch.cuda.current_stream().cuda_stream)
