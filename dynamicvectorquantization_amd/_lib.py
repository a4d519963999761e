"""ctypes binding of libdvq.so (include/dvq.h) -- the only bridge between the torch-facing
modules of this package and the HIP kernels.

There is NO fallback: if the shared library is missing or an entry point is absent the import
of this module raises.  torch is imported first so that libdvq.so binds to the HIP runtime torch
already loaded (same libamdhip64.so.7 SONAME) and can use torch's streams and device pointers.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# DVQ_LIBRARY: load another build of the same sources instead (tools/ use csrc/libdvq_tuning.so, made by
# `make -C csrc tuning`, which additionally exports the A/B switches dvq_tuning_set / dvq_tuning_buffers)
LIB_PATH = os.environ.get("DVQ_LIBRARY") or os.path.join(CSRC, "libdvq.so")

DVQ_OK = 0
MODE_EXACT = 0
MODE_FILTER = 1
MODE_FILTER_PASS1 = 2   # profiling aid: only the dominant filter kernel
MODE_FILTER_WIDE = 3    # testing aid: force the two-blocks-per-wave pass-1 kernel (D = 256)
FILTER_MODES = (MODE_FILTER, MODE_FILTER_PASS1, MODE_FILTER_WIDE)
MODE_WS_CLEAN = 0x100   # flag OR-ed into a filter mode: the workspace is clean (include/dvq.h), no zeroing kernel is launched
GATE_F32 = 0
GATE_I64 = 1
GATE_ENTROPY = 2        # routed assign only: entropy map + threshold
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2

EXPORTS = (
    "dvq_version", "dvq_last_error_string", "dvq_codebook_prep_bytes", "dvq_codebook_prepare_f32",
    "dvq_vq_assign_workspace_bytes", "dvq_vq_assign_nchw_f32", "dvq_vq_assign_flat_f32", "dvq_vq_assign_fallback_count_offset",
    "dvq_embed_gather_f32",
    "dvq_vq_assign_routed_workspace_bytes", "dvq_vq_assign_routed_dual_f32", "dvq_vq_assign_routed_triple_f32",
    "dvq_vq_assign_routed_fallback_count_offset",
    "dvq_exchange_bytes", "dvq_exchange_pack", "dvq_exchange_unpack", "dvq_debug_filter_scores_f32",
    "dvq_qconv_prep_bytes", "dvq_qconv_prepare_f32", "dvq_qconv_f32", "dvq_qconv_select_f32",
    "dvq_vq_backward_nchw_f32", "dvq_vq_backward_codebook_nchw_f32", "dvq_vq_assign_qconv_f32", "dvq_vq_assign_routed_qconv_dual_f32", "dvq_vq_assign_routed_qconv_triple_f32",
    "dvq_fold_prep_bytes", "dvq_fold_prepare_f32", "dvq_vq_assign_fold_f32", "dvq_vq_assign_routed_fold_dual_f32",
    "dvq_vq_assign_routed_fold_triple_f32", "dvq_debug_fold_scores_f32",
    "dvq_entropy_gate_f32", "dvq_route_select_dual_f32", "dvq_route_select_dual_entropy_f32", "dvq_route_select_triple_f32",
    "dvq_entropy_map_f32", "dvq_ema_accumulate_nchw_f32", "dvq_restart_pick_i64", "dvq_ema_update_f32", "dvq_router_gate_workspace_bytes", "dvq_router_gate_prep_bytes", "dvq_router_gate_prepare_f32", "dvq_router_gate_prepare_norm_f32", "dvq_router_gate_f32", "dvq_permute_dual_count_i64", "dvq_permute_dual_forward_i64", "dvq_permute_dual_backward_i64",
)


class DvqError(RuntimeError):
    pass


def build(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (csrc/Makefile) into csrc/libdvq.so."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        raise DvqError(
            "libdvq.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C %s` (hipcc --offload-arch=gfx950). There is no CPU fallback." % (LIB_PATH, CSRC))
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise DvqError("libdvq.so does not export %s (stale build?)" % name)
    vp, i32, i64, f32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    lib.dvq_version.restype = i32
    lib.dvq_last_error_string.restype = ctypes.c_char_p
    lib.dvq_codebook_prep_bytes.restype = sz
    lib.dvq_codebook_prep_bytes.argtypes = [i32, i32]
    lib.dvq_codebook_prepare_f32.restype = i32
    lib.dvq_codebook_prepare_f32.argtypes = [vp, i32, i32, vp, sz, vp]
    lib.dvq_vq_assign_workspace_bytes.restype = sz
    lib.dvq_vq_assign_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.dvq_vq_assign_fallback_count_offset.restype = sz
    lib.dvq_vq_assign_fallback_count_offset.argtypes = [i32, i32, i32, i32]
    lib.dvq_vq_assign_nchw_f32.restype = i32
    lib.dvq_vq_assign_nchw_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_assign_flat_f32.restype = i32
    lib.dvq_vq_assign_flat_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, f32, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_workspace_bytes.restype = sz
    lib.dvq_vq_assign_routed_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32]
    lib.dvq_vq_assign_routed_fallback_count_offset.restype = sz
    lib.dvq_vq_assign_routed_fallback_count_offset.argtypes = [i32, i32, i32, i32, i32, i32]
    lib.dvq_vq_assign_routed_dual_f32.restype = i32
    lib.dvq_vq_assign_routed_dual_f32.argtypes = [vp, i32, f32, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32,
                                                  vp, vp, vp, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_triple_f32.restype = i32
    lib.dvq_vq_assign_routed_triple_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32,
                                                    vp, vp, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_backward_nchw_f32.restype = i32
    lib.dvq_vq_backward_nchw_f32.argtypes = [vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, vp, vp]
    lib.dvq_vq_backward_codebook_nchw_f32.restype = i32
    lib.dvq_vq_backward_codebook_nchw_f32.argtypes = [vp, vp, vp, vp, vp, f32, i32, i32, i32, i32, vp, vp]
    lib.dvq_vq_assign_qconv_f32.restype = i32
    lib.dvq_vq_assign_qconv_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, i32, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_qconv_dual_f32.restype = i32
    lib.dvq_vq_assign_routed_qconv_dual_f32.argtypes = [vp, i32, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32,
                                                        vp, vp, vp, vp, vp, vp, vp, i32, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_qconv_triple_f32.restype = i32
    lib.dvq_vq_assign_routed_qconv_triple_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32,
                                                          vp, vp, vp, vp, vp, vp, i32, vp, sz, i32, vp]
    lib.dvq_fold_prep_bytes.restype = sz
    lib.dvq_fold_prep_bytes.argtypes = [i32, i32]
    lib.dvq_fold_prepare_f32.restype = i32
    lib.dvq_fold_prepare_f32.argtypes = [vp, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.dvq_vq_assign_fold_f32.restype = i32
    lib.dvq_vq_assign_fold_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_fold_dual_f32.restype = i32
    lib.dvq_vq_assign_routed_fold_dual_f32.argtypes = [vp, i32, f32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                                                       vp, vp, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_vq_assign_routed_fold_triple_f32.restype = i32
    lib.dvq_vq_assign_routed_fold_triple_f32.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32,
                                                         vp, vp, vp, vp, vp, sz, i32, vp]
    lib.dvq_debug_fold_scores_f32.restype = i32
    lib.dvq_debug_fold_scores_f32.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.dvq_qconv_prep_bytes.restype = sz
    lib.dvq_qconv_prep_bytes.argtypes = [i32]
    lib.dvq_qconv_prepare_f32.restype = i32
    lib.dvq_qconv_prepare_f32.argtypes = [vp, vp, i32, vp, sz, vp]
    lib.dvq_qconv_f32.restype = i32
    lib.dvq_qconv_f32.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.dvq_qconv_select_f32.restype = i32
    lib.dvq_qconv_select_f32.argtypes = [i32, vp, i32, f32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    if hasattr(lib, "dvq_tuning_set"):                     # tuning build only
        lib.dvq_tuning_set.restype = i32
        lib.dvq_tuning_set.argtypes = [ctypes.c_char_p, i32]
        lib.dvq_tuning_buffers.restype = i32
        lib.dvq_tuning_buffers.argtypes = [vp, vp]
        # tools/ only: DVQ_TUNE="key=value,..." sets A/B switches of the tuning build for a whole process (e.g. bench.py under
        # tools/archive/ab_lib.sh); the product library exports no such symbol and this branch is not taken
        for kv in filter(None, os.environ.get("DVQ_TUNE", "").split(",")):
            k, v = kv.split("=")
            if lib.dvq_tuning_set(k.strip().encode(), int(v)) != 0:
                raise DvqError("DVQ_TUNE: unknown switch %r" % k)
    lib.dvq_debug_filter_scores_f32.restype = i32
    lib.dvq_debug_filter_scores_f32.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.dvq_exchange_bytes.restype = sz
    lib.dvq_exchange_bytes.argtypes = [i64, i64, i32, i32]
    lib.dvq_exchange_pack.restype = i32
    lib.dvq_exchange_pack.argtypes = [vp, vp, vp, ctypes.c_double, i32, i32, i64, i64, i32, vp, vp]
    lib.dvq_exchange_unpack.restype = i32
    lib.dvq_exchange_unpack.argtypes = [vp, i32, i32, i64, i64, i32, vp, vp, vp, vp]
    lib.dvq_embed_gather_f32.restype = i32
    lib.dvq_embed_gather_f32.argtypes = [vp, i32, i32, vp, i64, vp, vp]
    lib.dvq_entropy_gate_f32.restype = i32
    lib.dvq_entropy_gate_f32.argtypes = [vp, i64, f32, vp, vp]
    lib.dvq_route_select_dual_f32.restype = i32
    lib.dvq_route_select_dual_f32.argtypes = [vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.dvq_route_select_dual_entropy_f32.restype = i32
    lib.dvq_route_select_dual_entropy_f32.argtypes = [vp, f32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.dvq_route_select_triple_f32.restype = i32
    lib.dvq_route_select_triple_f32.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.dvq_ema_accumulate_nchw_f32.restype = i32
    lib.dvq_ema_accumulate_nchw_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp]
    lib.dvq_ema_update_f32.restype = i32
    lib.dvq_ema_update_f32.argtypes = [vp, vp, f32, f32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp]
    lib.dvq_restart_pick_i64.restype = i32
    lib.dvq_restart_pick_i64.argtypes = [ctypes.c_uint64, i64, i32, vp, vp]
    lib.dvq_router_gate_workspace_bytes.restype = sz
    lib.dvq_router_gate_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32]
    lib.dvq_router_gate_prep_bytes.restype = sz
    lib.dvq_router_gate_prep_bytes.argtypes = [i32, i32, i32]
    lib.dvq_router_gate_prepare_f32.restype = i32
    lib.dvq_router_gate_prepare_f32.argtypes = [vp, i32, i32, i32, vp, sz, vp]
    lib.dvq_router_gate_prepare_norm_f32.restype = i32
    lib.dvq_router_gate_prepare_norm_f32.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.dvq_router_gate_f32.restype = i32
    lib.dvq_router_gate_f32.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp,
                                        vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
    lib.dvq_entropy_map_f32.restype = i32
    lib.dvq_entropy_map_f32.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.dvq_permute_dual_count_i64.restype = i32
    lib.dvq_permute_dual_count_i64.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    lib.dvq_permute_dual_forward_i64.restype = i32
    lib.dvq_permute_dual_forward_i64.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, ctypes.POINTER(i64),
                                                 vp, vp, vp, vp, vp, vp, vp]
    lib.dvq_permute_dual_backward_i64.restype = i32
    lib.dvq_permute_dual_backward_i64.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, vp, vp]
    return lib


lib = _load()
# DVQ_MODE_WS_CLEAN exists since ABI 0.5.0 (an older build loaded through DVQ_LIBRARY for an A/B rejects the flag)
HAS_WS_CLEAN = lib.dvq_version() >= 500


def check(rc, what):
    if rc != DVQ_OK:
        msg = lib.dvq_last_error_string().decode("utf-8", "replace")
        raise DvqError("%s failed (rc=%d): %s" % (what, rc, msg))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device):
    """the current HIP stream of `device` as an integer (what the ABI takes as `void *stream`)"""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)                 # 0.2 us; torch.cuda.current_stream(..).cuda_stream builds a Stream object
    return torch.cuda.current_stream(device).cuda_stream


class _NoDeviceSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoDeviceSwitch()


def on_device(device):
    """`with on_device(t.device):` -- torch.cuda.device(device) only when it is not the current device already (the context
    manager costs ~5 us per call: a fifth of a small op's host time, tools/module_overhead.py)"""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_SWITCH
    return torch.cuda.device(device)


def require_cuda_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise DvqError("%s is on %s: the dvq kernels run on the GPU only (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def ptr(t):
    return 0 if t is None else t.data_ptr()
