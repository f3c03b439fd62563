"""ctypes binding of libpyflow_hip.so (the C ABI in include/pyflow_hip.h).

The product path has NO fallback: if the shared library is missing the import of any op raises.
Tensors are torch tensors used purely as device-memory handles (``data_ptr()``); every call is
enqueued on torch's current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libpyflow_hip.so")

_lib = None


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("C", C.c_void_p), ("bias", C.c_void_p),
                ("res", C.c_void_p), ("gate", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldw", C.c_int),
                ("ldc", C.c_int), ("ldr", C.c_int),
                ("strideA", C.c_longlong), ("strideC", C.c_longlong), ("strideR", C.c_longlong),
                ("gate_stride", C.c_int), ("batch", C.c_int), ("gelu_from", C.c_int), ("flags", C.c_int),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong),
                ("qk_rope", C.c_void_p), ("qk_wq", C.c_void_p), ("qk_wk", C.c_void_p),
                ("qk_d", C.c_int), ("qk_q_col0", C.c_int), ("qk_k_col0", C.c_int), ("qk_row0", C.c_int),
                ("qk_eps", C.c_float), ("qk_q_scale", C.c_float), ("qk_head_stride", C.c_int),
                ("A2", C.c_void_p), ("W2", C.c_void_p), ("C2", C.c_void_p), ("bias2", C.c_void_p), ("res2", C.c_void_p),
                ("gate2", C.c_void_p), ("M2", C.c_int),
                ("strideA2", C.c_longlong), ("strideC2", C.c_longlong), ("strideR2", C.c_longlong),
                ("qk_wq2", C.c_void_p), ("qk_wk2", C.c_void_p), ("qk_row0_2", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("Y", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("T", C.c_int), ("H", C.c_int), ("W_", C.c_int),
                ("Hp", C.c_int), ("Wp", C.c_int), ("Cin", C.c_int), ("kt", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("in_base_off", C.c_longlong),
                ("N", C.c_int), ("n_valid", C.c_int),
                ("st", C.c_int), ("sh", C.c_int), ("sw", C.c_int), ("Cg", C.c_int), ("Hop", C.c_int),
                ("Wop", C.c_int), ("Cout_pitch", C.c_int),
                ("out_base_off", C.c_longlong),
                ("flags", C.c_int), ("out_scale", C.c_float), ("out_t_shift", C.c_int),
                ("in_sh", C.c_int), ("in_sw", C.c_int), ("in_st", C.c_int),
                ("gn_stats", C.c_void_p), ("gn_C", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("Vt", C.c_void_p), ("O", C.c_void_p),
                ("ldq", C.c_int), ("ldk", C.c_int), ("ldo", C.c_int),
                ("strideQ", C.c_longlong), ("strideK", C.c_longlong), ("strideO", C.c_longlong),
                ("strideVt_b", C.c_longlong), ("strideVt_h", C.c_longlong),
                ("B", C.c_int), ("H", C.c_int), ("L", C.c_int), ("Lp", C.c_int), ("Lt", C.c_int),
                ("a_lo", C.c_void_p), ("a_hi", C.c_void_p), ("b_hi", C.c_void_p), ("tile_kv_end", C.c_void_p),
                ("scale", C.c_float), ("head_stride_qk", C.c_int), ("q_row_begin", C.c_int), ("q_prescaled", C.c_int),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong),
                ("V", C.c_void_p), ("ldv", C.c_int), ("strideV", C.c_longlong)]


class AttnSmallDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p), ("O", C.c_void_p),
                ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
                ("strideQ", C.c_longlong), ("strideK", C.c_longlong), ("strideV", C.c_longlong),
                ("strideO", C.c_longlong),
                ("B", C.c_int), ("H", C.c_int), ("L", C.c_int),
                ("bias", C.c_void_p), ("key_mask", C.c_void_p), ("causal", C.c_int), ("scale", C.c_float)]


ABI_VERSION = 7          # PF_ABI_VERSION of include/pyflow_hip.h
GEMM_GATE_RES = 1
GEMM_OUT_F32 = 2
GEMM_ACT_QUICK_GELU = 4
GEMM_ACT_GELU_ERF = 8

# every symbol include/pyflow_hip.h declares (tests check the .so exports all of them)
EXPORTS = [
    "pf_last_error", "pf_version", "pf_comm_init_local", "pf_comm_create_window", "pf_comm_attach_windows", "pf_comm_transport", "pf_struct_size", "pf_gemm_bf16", "pf_gemm_set_policy", "pf_gemm_which", "pf_gemm_which_desc", "pf_gemm_workgroups", "pf_gemm_workspace_bytes", "pf_conv3d_bf16", "pf_conv3d_fuses_gn_stats", "pf_conv3d_which", "pf_attention_bf16", "pf_attention_workspace_bytes", "pf_attention_which", "pf_v_transpose",
    "pf_ln_modulate", "pf_qk_norm_rope", "pf_gemv_f32", "pf_timestep_embed", "pf_patchify", "pf_cfg_euler_step",
    "pf_copy_rows", "pf_sp_relayout", "pf_renoise_upsample", "pf_avgpool2",
    "pf_gn_stats", "pf_gn_apply", "pf_softmax_rows", "pf_shift_caches", "pf_latent_to_nhwc", "pf_blend_tiles", "pf_nhwc_to_planar_f32", "pf_to_uint8",
    "pf_embed_rows", "pf_rmsnorm", "pf_glu_mul", "pf_attention_small_bf16", "pf_rgb_to_yuv420",
    "pf_comm_unique_id", "pf_comm_init", "pf_comm_destroy", "pf_comm_rank", "pf_comm_world", "pf_all_to_all_v",
    "pf_halo_send_recv", "pf_all_gather_v", "pf_all_reduce_sum_f32", "pf_broadcast_bytes", "pf_comm_wait",
    "pf_cmdlist_create", "pf_cmdlist_destroy", "pf_cmdlist_clear", "pf_cmdlist_size", "pf_cmdlist_is_graph", "pf_cmdlist_gemm",
    "pf_cmdlist_attention", "pf_cmdlist_ln_modulate", "pf_cmdlist_qk_norm_rope", "pf_cmdlist_v_transpose",
    "pf_cmdlist_sp_relayout", "pf_cmdlist_copy_rows", "pf_cmdlist_all_to_all_v", "pf_cmdlist_comm_wait", "pf_cmdlist_join",
    "pf_cmdlist_run", "pf_cmdlist_instantiate",
]


class PyflowLibraryMissing(RuntimeError):
    pass


def use_lab_library(name="lab"):
    """tools/ and lab/ only: bind a measurement build (`make -C pyramid-flow_amd/csrc variant NAME=... DEFS=...`; `make lab` =
    the same sources with the lab-only switches of pf_gemm_set_policy compiled in) instead of the shipping library.  Call
    before the first load()."""
    global LIB_PATH, _lib
    assert _lib is None, "use_lab_library() must precede the first load()"
    LIB_PATH = os.path.join(os.path.dirname(_HERE), "variants", name, "libpyflow_hip.so")


def load():
    """Load the shared library; raise loudly if it is absent (no CPU / torch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PyflowLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The MI355X path has no fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.pf_last_error.restype = C.c_char_p
    lib.pf_version.restype = C.c_int
    lib.pf_cmdlist_create.restype = C.c_void_p
    lib.pf_attention_workspace_bytes.restype = C.c_longlong
    lib.pf_gemm_workspace_bytes.restype = C.c_longlong
    if lib.pf_version() != ABI_VERSION:
        raise RuntimeError(f"pyflow_hip: {LIB_PATH} has ABI version {lib.pf_version()}, this host expects {ABI_VERSION}: rebuild it")
    for which, cls in enumerate((GemmDesc, ConvDesc, AttnDesc, AttnSmallDesc)):       # struct mirrors vs the compiled layout
        if lib.pf_struct_size(C.c_int(which)) != C.sizeof(cls):
            raise RuntimeError(f"pyflow_hip: {cls.__name__} is {C.sizeof(cls)} bytes here but "
                               f"{lib.pf_struct_size(C.c_int(which))} in {LIB_PATH}: stale library? rebuild it")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("pyflow_hip: " + load().pf_last_error().decode())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
