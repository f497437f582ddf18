"""ctypes binding of libvdqn.so (include/vdqn.h).  There is no CPU fallback: if the HIP library is
missing or cannot be loaded every product entry point raises."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libvdqn binds to the ROCm runtime torch has already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# VDQN_LIB=<name> loads lib/libvdqn_<name>.so (a variant built with VDQN_LIB_OUT=<name>: A/B and diagnostic builds)
LIB_PATH = os.path.join(_HERE, "lib", f"libvdqn{'_' + os.environ['VDQN_LIB'] if os.environ.get('VDQN_LIB') else ''}.so")

VDQN_F32, VDQN_BF16 = 0, 1
ABI_VERSION = 14

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    _fields_ = [("in_", c_vp), ("wt", c_vp), ("bias", c_vp), ("resid", c_vp), ("mask", c_vp), ("out", c_vp),
                ("out_f32", c_vp), ("colsum_part", c_vp),
                ("n_img", c_i32), ("hi", c_i32), ("wi", c_i32), ("ci", c_i32), ("pix_stride", c_i32),
                ("ho", c_i32), ("wo", c_i32), ("co", c_i32), ("ldo", c_i32),
                ("r", c_i32), ("s", c_i32), ("stride", c_i32), ("pad", c_i32),
                ("mode", c_i32), ("relu", c_i32), ("dtype", c_i32),
                ("in2", c_vp), ("wt2", c_vp), ("bias2", c_vp), ("out2", c_vp),
                ("co2", c_i32), ("ldo2", c_i32), ("relu2", c_i32), ("ci2", c_i32),
]


class WgradArgs(C.Structure):
    _fields_ = [("gy", c_vp), ("x", c_vp), ("dw", c_vp), ("dbias", c_vp),
                ("n_img", c_i32), ("hi", c_i32), ("wi", c_i32), ("ci", c_i32), ("pix_stride", c_i32),
                ("ho", c_i32), ("wo", c_i32), ("co", c_i32), ("ldg", c_i32),
                ("r", c_i32), ("s", c_i32), ("stride", c_i32), ("pad", c_i32),
                ("splitk", c_i32), ("dtype", c_i32), ("workspace", c_vp), ("workspace_bytes", c_i64)]


class TdArgs(C.Structure):
    _fields_ = [("q_before", c_vp), ("q_after_online", c_vp), ("q_after_target", c_vp), ("act", c_vp),
                ("rew", c_vp), ("term", c_vp), ("valid", c_vp), ("loss", c_vp), ("dq", c_vp), ("dq_f32", c_vp),
                ("batch", c_i32), ("n_cat", c_i32), ("n_act", c_i32), ("ldq", c_i32),
                ("gamma", c_f32), ("inv_count", c_f32),
                ("clip_rect", c_i32), ("linear", c_i32), ("use_valid", c_i32), ("dtype", c_i32), ("loss_kind", c_i32), ("deterministic", c_i32), ("q_copy", c_vp)]


class NetConfig(C.Structure):
    _fields_ = [("action_dim", c_i32), ("num_classes", c_i32), ("num_frames", c_i32), ("extra_capacity", c_i32),
                ("dtype", c_i32), ("max_batch", c_i32), ("deterministic", c_i32)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("offset", c_i64), ("numel", c_i64), ("ndim", c_i32),
                ("shape", c_i32 * 4), ("kind", c_i32), ("param_id", c_i32), ("stage", c_i32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", c_i64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class StepArgs(C.Structure):
    _fields_ = [("params", c_vp), ("bnstats", c_vp), ("packed_online", c_vp), ("packed_target", c_vp),
                ("before", c_vp), ("after", c_vp), ("src_kind", c_i32), ("batch", c_i32),
                ("act", c_vp), ("rew", c_vp), ("term", c_vp), ("valid", c_vp), ("gt", c_vp),
                ("gamma", c_f32), ("inv_count", c_f32),
                ("clip_rect", c_i32), ("linear", c_i32), ("use_valid", c_i32), ("train_on_ground_truth", c_i32),
                ("value_learning", c_i32),
                ("acts_online", c_vp), ("acts_target", c_vp), ("bwd", c_vp), ("grads", c_vp), ("loss", c_vp),
                ("q_before", c_vp), ("loss_kind", c_i32), ("packed_frames", c_vp), ("acts_samples", c_i32)]


ALLREDUCE_FN = C.CFUNCTYPE(None, c_vp, c_vp, c_i64, c_vp)  # vdqn_allreduce_fn(user, buf, count, stream)

_SIGS = {
    "vdqn_last_error": (C.c_char_p, []),
    "vdqn_abi_version": (C.c_int, []),
    "vdqn_profile_enable": (C.c_int, [C.c_int]),
    "vdqn_profile_collect": (C.c_int, [C.POINTER(ProfEntry), C.c_int]),
    "vdqn_conv2d": (C.c_int, [C.POINTER(ConvArgs), c_vp]),
    "vdqn_abi_struct_size": (c_i32, [c_i32]),
    "vdqn_conv2d_colsum_rows": (c_i32, [C.POINTER(ConvArgs)]),
    "vdqn_conv2d_wgrad": (C.c_int, [C.POINTER(WgradArgs), c_vp]),
    "vdqn_conv2d_wgrad_workspace_bytes": (c_i64, [C.POINTER(WgradArgs)]),
    "vdqn_pack_input": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp]),
    "vdqn_maxpool_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_maxpool_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_stem_conv_pool": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "vdqn_stem_conv_pool_n": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_td_loss": (C.c_int, [C.POINTER(TdArgs), c_vp]),
    "vdqn_gt_loss": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "vdqn_adam": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, C.c_double, C.c_double, C.c_double, C.c_double, c_vp]),
    "vdqn_bn_train_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                    c_f32, c_f32, c_i32, c_vp, c_i64, c_vp]),
    "vdqn_bn_train_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "vdqn_bn_train_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "vdqn_avgpool_fwd": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_avgpool_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_net_create": (C.c_int, [C.POINTER(NetConfig), C.POINTER(c_vp)]),
    "vdqn_net_destroy": (None, [c_vp]),
    "vdqn_stem_wgrad_pool": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    "vdqn_stem_wgrad_pool_workspace_bytes": (c_i64, [c_i32]),
    "vdqn_net_set_overlap": (C.c_int, [c_vp, C.c_int]),
    "vdqn_net_grad_stream": (c_vp, [c_vp]),
    "vdqn_net_set_bn_sync": (C.c_int, [c_vp, c_vp, c_vp, c_i32]),
    "vdqn_net_num_params": (C.c_int, [c_vp]),
    "vdqn_net_param_info": (C.c_int, [c_vp, C.c_int, C.POINTER(ParamInfo)]),
    "vdqn_net_params_numel": (c_i64, [c_vp]),
    "vdqn_net_trainable_numel": (c_i64, [c_vp]),
    "vdqn_net_bnstats_numel": (c_i64, [c_vp]),
    "vdqn_net_stage_range": (C.c_int, [c_vp, C.c_int, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "vdqn_net_packed_bytes": (c_i64, [c_vp]),
    "vdqn_net_acts_bytes": (c_i64, [c_vp, c_i32]),
    "vdqn_net_bwd_bytes": (c_i64, [c_vp, c_i32]),
    "vdqn_net_act_offset": (c_i64, [c_vp, c_i32, C.c_char_p]),
    "vdqn_net_bwd_offset": (c_i64, [c_vp, c_i32, C.c_char_p]),
    "vdqn_net_pack_weights": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "vdqn_net_forward": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vdqn_net_trunk_forward": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "vdqn_softmax_rows": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "vdqn_softmax_ce": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "vdqn_mask_scale": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_i32, c_vp]),
    "vdqn_axpy": (C.c_int, [c_vp, c_vp, c_f32, c_i64, c_vp]),
    "vdqn_net_forward_train": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vdqn_net_td_forward": (C.c_int, [c_vp, C.POINTER(StepArgs), c_vp]),
    "vdqn_net_backward_stage": (C.c_int, [c_vp, C.POINTER(StepArgs), c_i32, c_vp]),
    "vdqn_net_backward_begin": (C.c_int, [c_vp, C.POINTER(StepArgs), c_vp, c_vp]),
    "vdqn_comm_unique_id": (C.c_int, [c_vp]),
    "vdqn_comm_init": (C.c_int, [c_i32, c_i32, c_vp, C.POINTER(c_vp)]),
    "vdqn_allreduce_bucket": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "vdqn_comm_rank": (C.c_int, [c_vp]),
    "vdqn_comm_size": (C.c_int, [c_vp]),
    "vdqn_comm_destroy": (C.c_int, [c_vp]),
    "vdqn_host_gather": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i32]),
}
COMM_UID_BYTES = 128
EXPORTS = tuple(_SIGS)

_lib = None


class VdqnError(RuntimeError):
    pass


def load():
    """Load libvdqn.so; raise (loudly) when it is missing — there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VdqnError(f"{LIB_PATH} is missing: build it with `python -m video_dqn_amd.build` "
                        "(the HIP extension is required; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.vdqn_abi_version() != ABI_VERSION:
        raise VdqnError(f"libvdqn ABI {lib.vdqn_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    # the argument structs of this binding against the library's own sizeof(): a drifted field fails here, not inside a kernel
    for which, st in enumerate((ConvArgs, WgradArgs, TdArgs, NetConfig, ParamInfo, ProfEntry, StepArgs)):
        if lib.vdqn_abi_struct_size(which) != C.sizeof(st):
            raise VdqnError(f"libvdqn: sizeof({st.__name__}) is {lib.vdqn_abi_struct_size(which)} in the library, "
                            f"{C.sizeof(st)} in this binding: include/vdqn.h and video_dqn_amd/_lib.py have drifted apart")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vdqn_last_error()
        raise VdqnError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def profile_enable(on: bool) -> None:
    load().vdqn_profile_enable(int(on))


def profile_collect():
    """-> {kernel: dict(launches, ms, flops, bytes)}; synchronises the recorded events and resets."""
    buf = (ProfEntry * 256)()
    n = load().vdqn_profile_collect(buf, 256)
    return {buf[i].name.decode(): dict(launches=buf[i].launches, ms=buf[i].ms, flops=buf[i].flops, bytes=buf[i].bytes)
            for i in range(n)}
