"""ctypes binding of libdiffpir_hip.so (include/diffpir_engine.h).

The library is built in-tree (diffpir_amd/csrc/libdiffpir_hip.so) by `__graft_entry__.build()` /
`make -C diffpir_amd/csrc`.  There is NO fallback: if the library is missing, cannot be loaded, or
no gfx950 device is visible, the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFPIR_LIB: developer override used to A/B kernel variants on one GPU box; the default is the in-tree build
LIB_PATH = os.environ.get("DIFFPIR_LIB") or os.path.join(_HERE, "csrc", "libdiffpir_hip.so")
ABI_VERSION = 2      # include/diffpir_engine.h DPIR_ABI_VERSION: bumped whenever a public struct layout changes



class UNetDesc(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("in_channels", C.c_int32), ("model_channels", C.c_int32),
                ("out_channels", C.c_int32), ("num_res_blocks", C.c_int32), ("num_head_channels", C.c_int32),
                ("n_channel_mult", C.c_int32), ("channel_mult", C.c_float * 8),
                ("n_attention_ds", C.c_int32), ("attention_ds", C.c_int32 * 8), ("num_classes", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class Step(C.Structure):
    _fields_ = [("t", C.c_int32), ("last", C.c_int32), ("c1", C.c_float), ("c2", C.c_float), ("tau", C.c_float),
                ("sa_t", C.c_float), ("s1m_t", C.c_float), ("sa_p", C.c_float),
                ("k1", C.c_float), ("q", C.c_float), ("es", C.c_float), ("k2", C.c_float)]


class LoopDesc(C.Structure):
    _fields_ = [("task", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("sf", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("in_iter", C.c_int32), ("gamma", C.c_float),
                ("guidance", C.c_float), ("sa_start", C.c_float), ("s1m_start", C.c_float),
                ("y_dev", C.c_void_p), ("k_dev", C.c_void_p), ("mask_dev", C.c_void_p), ("labels_host", C.c_void_p),
                ("noise_init_dev", C.c_void_p), ("noise_n1_dev", C.c_void_p), ("noise_n2_dev", C.c_void_p),
                ("seed", C.c_uint64), ("image_offset", C.c_int64), ("use_graph", C.c_int32),
                ("skip_dead_final_eval", C.c_int32), ("generate_mode", C.c_int32), ("first_order", C.c_int32),
                ("noise_rp_dev", C.c_void_p), ("ddim_sample", C.c_int32)]


class DpsCoef(C.Structure):
    _fields_ = [("pc1", C.c_float), ("pc2", C.c_float), ("min_log", C.c_float), ("max_log", C.c_float), ("sa_prev", C.c_float), ("s1m_prev", C.c_float)]


class PSampleCoef(C.Structure):
    _fields_ = [("c1", C.c_float), ("c2", C.c_float), ("pc1", C.c_float), ("pc2", C.c_float), ("min_log", C.c_float), ("max_log", C.c_float),
                ("ddim", C.c_int32), ("sa_prev", C.c_float), ("s1m_prev", C.c_float)]


class DegradeDesc(C.Structure):
    _fields_ = [("task", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("sf", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("noise_level_img", C.c_float), ("seed", C.c_uint64), ("image_offset", C.c_int64)]


PROF_CLASSES = 8
PROF_NAMES = ["conv3x3", "conv1x1", "groupnorm_stats", "attention", "fft_prox", "elementwise", "unet_forward", "loop_graph"]

# name -> (restype, argtypes); every symbol declared in include/diffpir_engine.h
SIGNATURES = {
    "dpir_version": (C.c_int, []),
    "dpir_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dpir_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "dpir_destroy": (None, [C.c_void_p]),
    "dpir_last_error": (C.c_char_p, [C.c_void_p]),
    "dpir_sync": (C.c_int, [C.c_void_p]),
    "dpir_stream": (C.c_void_p, [C.c_void_p]),
    "dpir_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dpir_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dpir_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dpir_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dpir_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dpir_load_unet": (C.c_int, [C.c_void_p, C.POINTER(UNetDesc), C.POINTER(Tensor), C.c_int]),
    "dpir_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "dpir_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_model_fn_xstart": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_int]),
    "dpir_enable_grad": (C.c_int, [C.c_void_p, C.c_int]),
    "dpir_unet_vjp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_run_dps_loop": (C.c_int, [C.c_void_p, C.POINTER(LoopDesc), C.POINTER(Step), C.POINTER(DpsCoef), C.c_int, C.c_int, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "dpir_p_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(PSampleCoef), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_eps_from_xstart": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_size_t]),
    "dpir_grad_and_value": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_unet_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dpir_prox_fft_precalc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.POINTER(C.c_void_p)]),
    "dpir_prox_free": (None, [C.c_void_p, C.c_void_p]),
    "dpir_prox_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "dpir_data_solution": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "dpir_prox_fft_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "dpir_set_prox_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "dpir_prox_fft_apply_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "dpir_prox_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]),
    "dpir_resize_down": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dpir_prox_ibp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dpir_bicubic_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dpir_renoise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Step), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_repaint_mix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_finalize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dpir_randn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dpir_ewise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_size_t]),
    "dpir_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dpir_comm_version": (C.c_int, [C.POINTER(C.c_int)]),
    "dpir_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "dpir_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dpir_allgather_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dpir_comm_allreduce_max": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "dpir_comm_barrier": (C.c_int, [C.c_void_p]),
    "dpir_comm_destroy": (C.c_int, [C.c_void_p]),
    "dpir_degrade": (C.c_int, [C.c_void_p, C.POINTER(DegradeDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dpir_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dpir_run_loop": (C.c_int, [C.c_void_p, C.POINTER(LoopDesc), C.POINTER(Step), C.c_int, C.c_void_p, C.c_void_p]),
    "dpir_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dpir_prof_reset": (C.c_int, [C.c_void_p]),
    "dpir_prof_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dpir_graph_cache_size": (C.c_int, [C.c_void_p]),
    "dpir_unet_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "dpir_unet_flops_class": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
}

_lib = None


class EngineLibraryError(RuntimeError):
    pass


DEBUG_LIB_PATH = os.path.join(_HERE, "csrc", "libdiffpir_dbg.so")
_dbg = None


def load_debug():
    """Development probes (include/diffpir_debug.h) live in their own library; tests/ and tools/ only -- the product path never
    loads it.  Binds the probes' signatures."""
    global _dbg
    if _dbg is not None:
        return _dbg
    load()
    if not os.path.exists(DEBUG_LIB_PATH):
        raise EngineLibraryError(f"{DEBUG_LIB_PATH} not found: `make -C diffpir_amd/csrc` builds it next to the product library")
    d = C.CDLL(DEBUG_LIB_PATH)
    d.dpir_debug_conv_bench.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double)]
    d.dpir_debug_conv7_emit_supported.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.POINTER(C.c_int)]
    d.dpir_debug_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_ulonglong)]
    d.dpir_debug_victim_alu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    for fn in (d.dpir_debug_victim_fft_pk, d.dpir_debug_victim_fft_nopk):
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
    d.dpir_debug_conv7_check.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_ulonglong),
                                                                        C.POINTER(C.c_float), C.POINTER(C.c_int)]
    for n in ("dpir_debug_conv_bench", "dpir_debug_victim", "dpir_debug_victim_alu", "dpir_debug_victim_fft_pk", "dpir_debug_victim_fft_nopk",
              "dpir_debug_conv7_check"):
        getattr(d, n).restype = C.c_int
    _dbg = d
    return d


def load():
    """Load the shared library and bind every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C diffpir_amd/csrc`.  There is no CPU fallback.")
    # One ROCm runtime per process.  The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64 / librccl; libdiffpir_hip.so
    # links the same SONAMEs, so whichever is loaded FIRST serves both.  If this library came first (/opt/rocm's runtime) and
    # torch later, torch's librccl would dlopen its own, never-initialised copy of the HSA runtime and RCCL init fails with
    # "no ROCm-capable device is detected" (measured, tests/test_gpu_dist.py).  Every host flow here uses torch as plumbing
    # (checkpoints, torch.distributed), so torch's runtime is loaded first when torch is installed; without torch the system
    # ROCm under /opt/rocm serves the library and its RCCL binding alike.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as ex:  # pragma: no cover
        raise EngineLibraryError(f"cannot load {LIB_PATH}: {ex}") from ex
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as ex:
            raise EngineLibraryError(f"{LIB_PATH} does not export {name}") from ex
        fn.restype = res
        fn.argtypes = args
    if lib.dpir_version() != ABI_VERSION:
        raise EngineLibraryError("ABI version mismatch")
    _lib = lib
    return lib
