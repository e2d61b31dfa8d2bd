"""ctypes binding of libssrhip.so (the C ABI declared in include/ssr_hip.h).

The library is built in-tree by ``ssr_eval_amd.build.build()`` (``hipcc --offload-arch=gfx950``).
There is NO fallback: if the shared object is missing or no HIP device is present every entry
point raises.  Nothing here imports ``oracle``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libssrhip.so")

SSR_F32, SSR_F64 = 0, 1
M_LSD, M_LOG_SISPEC, M_SISPEC, M_SSIM, M_ALL = 1, 2, 4, 8, 15
STFT_MAG, STFT_COMPLEX = 1, 2
LOWPASS_SEGMENTS, LOWPASS_FUSED, LOWPASS_CONV = 0, 1, 2
PAD_REFLECT, PAD_CONSTANT = 0, 1
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4      # include/ssr_hip.h

_vp, _i, _i64, _sz, _u = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_uint

# name -> (restype, argtypes): exactly the symbols include/ssr_hip.h declares
SIGNATURES = {
    "ssr_last_error": (C.c_char_p, []),
    "ssr_version": (_i, []),
    "ssr_plan_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "ssr_plan_create_ex": (_i, [_i, _i, _vp, _i, _i, C.POINTER(_vp)]),
    "ssr_plan_destroy": (_i, [_vp]),
    "ssr_plan_query": (_i, [_vp] + [C.POINTER(_i)] * 6),
    "ssr_plan_set_lowpass_engine": (_i, [_vp, _i]),
    "ssr_plan_set_tl_weights": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ssr_tl_weights": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "ssr_tl_weights_ex": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssr_num_frames": (_i64, [_vp, _i64]),
    "ssr_stft": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ssr_magphase": (_i, [_vp, _vp, _i64, C.c_float, _vp, _vp, _vp, _vp]),
    "ssr_pair_metrics_workspace_bytes": (_sz, [_vp, _i, _i, _i64]),
    "ssr_pair_metrics_workspace_bytes_for": (_sz, [_vp, _i, _i, _i64, _u]),
    "ssr_pair_metrics": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _u, _vp, _vp, _sz, _vp]),
    "ssr_pair_metrics_multi_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i64, _u]),
    "ssr_pair_metrics_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _u, _vp, _vp, _sz, _vp]),
    "ssr_pair_metrics_multi_est64_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i64, _u]),
    "ssr_pair_metrics_multi_est64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _u, _vp, _vp, _sz, _vp]),
    "ssr_pair_metrics_est64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _u, _vp, _vp, _sz, _vp]),
    "ssr_pair_metrics_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _u, _vp, _vp, _sz, _vp]),
    "ssr_pair_metrics_stages": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _u, _vp, _vp, _sz, _vp, _i]),
    "ssr_spectrogram_metrics_workspace_bytes": (_sz, [_i, _i, _i]),
    "ssr_spectrogram_metrics": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _u, _vp, _vp, _sz, _vp]),
    "ssr_to_log": (_i, [_vp, _i64, _vp, _vp]),
    "ssr_from_log": (_i, [_vp, _i64, _vp, _vp]),
    "ssr_energy_sums": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "ssr_scale_items": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "ssr_sispec_multichannel": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _sz, _vp]),
    "ssr_ola_workspace_bytes": (_sz, [_vp, _i64]),
    "ssr_fft_lowpass": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "ssr_fft_lowpass_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i64, _vp, _i64, _vp, _sz, _vp]),
    "ssr_istft": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "ssr_resample_plan": (_i, [_i64, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i),
                               C.POINTER(_i), C.POINTER(_i)]),
    "ssr_resample_poly": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ssr_resample_poly_chain": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ssr_resample_poly_mfma": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ssr_resample_poly_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "ssr_resample_sinc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64, _vp, _vp, _i, _i, _i, C.c_double, C.c_double, _i,
                                _vp, _vp]),
    "ssr_pcm16_to_float": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ssr_flac_info": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i)]),
    "ssr_flac_decode_pcm16": (_i, [C.c_char_p, _vp, _i64, _i, C.POINTER(_i64)]),
    "ssr_flac_decode_i32": (_i, [C.c_char_p, _vp, _i64, _i, C.POINTER(_i64)]),
    "ssr_xcorr_workspace_bytes": (_sz, [_i, _i]),
    "ssr_xcorr_argmax": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ssr_sosfiltfilt_workspace_bytes": (_sz, [_i64, _i, _i]),
    "ssr_sosfiltfilt": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ssr_sosfiltfilt_f64": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "ssr_sosfiltfilt_multi_workspace_bytes": (_sz, [_i64, _i, _vp, _i]),
    "ssr_sosfiltfilt_multi": (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _i64, _vp, _sz, _vp]),
    "ssr_comm_unique_id": (_i, [_vp]),
    "ssr_comm_init_rank": (_i, [_vp, _i, _i, C.POINTER(C.c_void_p)]),
    "ssr_comm_destroy": (_i, [_vp]),
    "ssr_allreduce_sums": (_i, [_vp, _i, _vp, _vp]),
}

_lib = None


class SsrHipError(RuntimeError):
    pass


def load():
    """Load libssrhip.so and bind every declared symbol; raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SsrHipError(
            "libssrhip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  ssr_eval_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ssr_last_error()
        raise SsrHipError("libssrhip error %d: %s" % (rc, msg.decode() if msg else "?"))
