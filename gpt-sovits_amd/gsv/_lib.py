"""ctypes binding of libgsv_hip.so (include/gsv.h).  The product path has no CPU or
PyTorch fallback: if the library cannot be loaded, or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgsv_hip.so")

GSV_F32, GSV_F16 = 0, 1


class T2SConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_layer", "dim", "n_head", "ffn_dim", "vocab", "phoneme_vocab", "bert_dim")]


class SamplingParams(C.Structure):
    _fields_ = [("top_k", C.c_int), ("top_p", C.c_float), ("temperature", C.c_float),
                ("repetition_penalty", C.c_float), ("early_stop_num", C.c_int), ("eos_mask_steps", C.c_int),
                ("max_steps", C.c_int), ("seed", C.c_uint64)]


class VitsConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("inter_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers",
                                        "kernel_size", "gin_channels", "n_symbols", "ssl_dim", "n_bins",
                                        "upsample_initial_channel", "n_ups")] + \
               [("up_rates", C.c_int * 8), ("up_kernels", C.c_int * 8), ("n_resblocks", C.c_int),
                ("rb_kernels", C.c_int * 4), ("rb_dilations", (C.c_int * 3) * 4), ("ref_bins", C.c_int),
                ("flavor", C.c_int), ("v2pro", C.c_int)]


class VocoderConfig(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_channels", C.c_int), ("upsample_initial_channel", C.c_int), ("n_ups", C.c_int),
                ("up_rates", C.c_int * 8), ("up_kernels", C.c_int * 8), ("n_resblocks", C.c_int),
                ("rb_kernels", C.c_int * 4), ("rb_dilations", (C.c_int * 3) * 4), ("bias_at_final", C.c_int),
                ("tanh_at_final", C.c_int), ("snake_logscale", C.c_int)]


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "depth", "heads", "dim_head", "ff_mult", "mel_dim", "text_dim", "conv_layers")]


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("res", C.c_void_p),
                ("T_in", C.c_int), ("T_out", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("taps", C.c_int),
                ("stride", C.c_int), ("dil", C.c_int), ("pad", C.c_int), ("pre_act", C.c_int),
                ("pre_slope", C.c_float), ("post_act", C.c_int), ("scale", C.c_float), ("accumulate", C.c_int),
                ("out_f32", C.c_int), ("ups_u", C.c_int), ("ups_pad", C.c_int),
                ("Z", C.c_int), ("xz", C.c_longlong), ("wz", C.c_longlong), ("yz", C.c_longlong),
                ("ldx", C.c_int), ("ldw", C.c_int), ("ldy", C.c_int), ("gate", C.c_void_p), ("bz", C.c_int),
                ("rz", C.c_longlong), ("ldr", C.c_int)]


_SIGS = {
    "gsv_init": (C.c_int, [C.c_int]),
    "gsv_last_error": (C.c_char_p, []),
    "gsv_abi_version": (C.c_int, []),
    "gsv_t2s_create": (C.c_int, [C.POINTER(T2SConfig), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "gsv_t2s_destroy": (None, [C.c_void_p]),
    "gsv_t2s_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "gsv_t2s_finalize": (C.c_int, [C.c_void_p]),
    "gsv_t2s_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p]),
    "gsv_t2s_decode": (C.c_int, [C.c_void_p, C.POINTER(SamplingParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int), C.c_void_p]),
    "gsv_t2s_decode_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "gsv_t2s_set_mega": (C.c_int, [C.c_void_p, C.c_int]),
    "gsv_t2s_engine_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint)]),
    "gsv_t2s_set_debug": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsv_t2s_debug_stall": (C.c_int, [C.c_void_p, C.c_int]),
    "gsv_t2s_debug_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsv_t2s_time_step": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    "gsv_t2s_debug_set_state": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gsv_t2s_step_bytes": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gsv_vits_create": (C.c_int, [C.POINTER(VitsConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "gsv_vits_destroy": (None, [C.c_void_p]),
    "gsv_vits_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "gsv_vits_finalize": (C.c_int, [C.c_void_p]),
    "gsv_vits_set_refer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int,
                                     C.c_void_p]),
    "gsv_vits_set_refer_sv": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p),
                                        C.c_int, C.c_void_p]),
    "gsv_vits_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                  C.c_double, C.c_uint64, C.c_void_p, C.c_void_p]),
    "gsv_vits_encp_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "gsv_vits_decode_encp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]),
    "gsv_vits_extract_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_vits_debug_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                                        C.c_void_p]),
    "gsv_vits_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "gsv_vocoder_create": (C.c_int, [C.POINTER(VocoderConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "gsv_vocoder_destroy": (None, [C.c_void_p]),
    "gsv_vocoder_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "gsv_vocoder_finalize": (C.c_int, [C.c_void_p]),
    "gsv_vocoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_cfm_create": (C.c_int, [C.POINTER(DitConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "gsv_cfm_destroy": (None, [C.c_void_p]),
    "gsv_cfm_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "gsv_cfm_finalize": (C.c_int, [C.c_void_p]),
    "gsv_cfm_inference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]),
    "gsv_sola": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "gsv_postprocess": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_aa_act_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gsv_op_flash_attn64": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gsv_op_flash_rel96": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "gsv_op_conv1d": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "gsv_op_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "gsv_op_magnitude": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_op_conv_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "gsv_op_aff_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]),
    "gsv_op_time_mean": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_op_channel_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p]),
    "gsv_op_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_float, C.c_int, C.c_void_p]),
    "gsv_op_decode_attn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_void_p]),
    "gsv_op_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                C.POINTER(SamplingParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

EXPORTS = tuple(_SIGS)   # every symbol include/gsv.h declares

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python gpt-sovits_amd/gsv/build.py` "
                               "(the gsv hot path has no CPU fallback)")
        # torch first: it ships its own HIP runtime; loading ours before it puts two runtimes in one
        # process and the second one finds no device
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(l, name)
            except AttributeError:   # tests/test_abi.py asserts the full export list
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().gsv_last_error().decode(errors="replace")
        raise RuntimeError(f"libgsv_hip {what} failed (rc={rc}): {msg}")


_inited = set()


def init(device_index: int):
    if device_index not in _inited:
        check(lib().gsv_init(device_index), "gsv_init")
        _inited.add(device_index)


def dtype_code(torch_dtype) -> int:
    import torch
    if torch_dtype == torch.float16:
        return GSV_F16
    if torch_dtype == torch.float32:
        return GSV_F32
    raise ValueError(f"unsupported dtype {torch_dtype}: the engine computes in float16 or float32")
