"""ctypes binding of libnbci.so (the C-ABI declared in include/nbci.h).

The library is built in-tree (llm_bci_amd/csrc/libnbci.so) by `__graft_entry__.build()` or
`make -C llm_bci_amd/csrc`. There is no CPU fallback: if the library is missing, every op
raises `NbciUnavailable` loudly.
"""
import ctypes as C
import os

# PyTorch bundles its own libamdhip64.so.7; it must be the copy that gets loaded (libnbci.so then
# binds to it by SONAME), otherwise two HIP runtimes end up in one process and neither sees the GPU.
import torch  # noqa: F401  (keep this import BEFORE loading libnbci.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBCI_LIB") or os.path.join(_HERE, "csrc", "libnbci.so")  # NBCI_LIB: A/B a variant build

NBCI_F32, NBCI_BF16 = 0, 1
ACT = {"identity": 0, None: 0, "none": 0, "softsign": 1, "gelu": 2, "relu": 3, "tanh": 4}


class NbciUnavailable(RuntimeError):
    pass


class NbciError(RuntimeError):
    pass


class Operand(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int64), ("kmajor", C.c_int32), ("rpb", C.c_int32),
                ("gstride", C.c_int64), ("zs1", C.c_int64), ("zs2", C.c_int64)]


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("in_dtype", C.c_int32),
                ("A", Operand), ("B", Operand),
                ("C", C.c_void_p), ("C2", C.c_void_p), ("ldc", C.c_int64),
                ("czs1", C.c_int64), ("czs2", C.c_int64), ("c_dtype", C.c_int32),
                ("batch", C.c_int32), ("zdiv", C.c_int32), ("splitk", C.c_int32),
                ("alpha", C.c_float), ("beta", C.c_float), ("bias", C.c_void_p),
                ("act", C.c_int32), ("drop_p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32),
                ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("residual_rows", C.c_void_p), ("residual_first", C.c_int32),
                ("gate", C.c_void_p), ("ldg", C.c_int64), ("gate_act", C.c_int32), ("c2_grad", C.c_int32),
                ("colsum", C.c_void_p), ("colsum_rep_stride", C.c_int64), ("colsum_nrep", C.c_int32),
                ("gate_follows_c", C.c_int32), ("residual_dtype", C.c_int32)]


class NDT1Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_channels", "input_dim", "stack_size", "stack_stride", "hidden", "n_layers",
                                         "n_heads", "inter", "vocab", "max_F")] + [
        ("smooth_sd", C.c_float), ("noise", C.c_int32), ("white_noise_sd", C.c_float), ("constant_offset_sd", C.c_float),
        ("embed_act", C.c_int32), ("mlp_act", C.c_int32), ("embed_dropout", C.c_float), ("dropout", C.c_float),
        ("use_rope", C.c_int32), ("rope_theta", C.c_float), ("context_forward", C.c_int32), ("context_backward", C.c_int32),
        ("pos", C.c_int32), ("blank_id", C.c_int32), ("zero_infinity", C.c_int32), ("dtype", C.c_int32),
        ("factors_size", C.c_int32), ("factors_act", C.c_int32), ("factors_bias", C.c_int32), ("adapt_days", C.c_int32),
        ("day_token_days", C.c_int32), ("block_token_blocks", C.c_int32), ("residual_dtype", C.c_int32)]


class NDT1IO(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("S", C.c_int32),
                ("spikes", C.c_void_p), ("spikes_mask", C.c_void_p), ("spikes_timestamp", C.c_void_p),
                ("spikes_lengths", C.c_void_p), ("targets", C.c_void_p), ("targets_lengths", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("train", C.c_int32), ("want_grad", C.c_int32), ("seed", C.c_uint32), ("grad_scale", C.c_float),
                ("preds", C.c_void_p), ("loss", C.c_void_p), ("argmax", C.c_void_p), ("hidden_out", C.c_void_p),
                ("token_mask_out", C.c_void_p), ("d_hidden", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("day_idx", C.c_void_p), ("block_idx", C.c_void_p),
                ("embed_part", C.c_int32), ("aux_stream", C.c_void_p)]


class MaskerDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("N", C.c_int32), ("mode", C.c_int32), ("ratio", C.c_float),
                ("timespan", C.c_int32), ("zero_ratio", C.c_float), ("random_ratio", C.c_float), ("probs", C.c_void_p),
                ("ext_mask", C.c_void_p), ("seed", C.c_uint32), ("site", C.c_uint32), ("in_", C.c_void_p), ("out", C.c_void_p),
                ("mask", C.c_void_p), ("accumulate", C.c_int32), ("scratch", C.c_void_p), ("target_bn", C.c_void_p)]


MASK_MODE = {"temporal": 0, "neuron": 1, "random": 2, "region": 3, "co-smooth": 4, "given": 5, "table_t": 6}
LOSS_KIND = {("poisson_nll", True): 0, ("poisson_nll", False): 1, ("mse", True): 2, ("mse", False): 2}


class ItrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("max_n_bins", "hidden", "n_heads", "n_layers", "max_n_channels", "n_regions", "act",
                                         "dec_act")] + [("embed_dropout", C.c_float), ("dropout", C.c_float)] + [
        (n, C.c_int32) for n in ("use_cls", "mlp_decoder", "loss", "dtype", "residual_dtype", "embed_depth", "emb_mode", "emb_hidden",
                                 "emb_heads", "emb_layers")]


class ItrIO(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("spikes", C.c_void_p), ("masked", C.c_void_p), ("mask", C.c_void_p),
                ("spikes_mask", C.c_void_p), ("spikes_spacestamp", C.c_void_p), ("region_idx", C.c_void_p),
                ("spikes_timestamp", C.c_void_p), ("neuron_depths", C.c_void_p),
                ("train", C.c_int32), ("want_grad", C.c_int32), ("seed", C.c_uint32), ("grad_scale", C.c_float),
                ("preds", C.c_void_p), ("mask_out", C.c_void_p), ("loss", C.c_void_p), ("n_examples", C.c_void_p),
                ("hidden_out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


class PtstConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_input_channels", "context_length", "patch_length", "patch_stride", "num_hidden_layers",
                                         "d_model", "num_attention_heads", "ffn_dim")] + [
        (n, C.c_float) for n in ("norm_eps", "attention_dropout", "positional_dropout", "path_dropout", "ff_dropout")] + [
        ("act", C.c_int32), ("do_mask_input", C.c_int32), ("random_mask_ratio", C.c_double), ("channel_consistent_masking", C.c_int32),
        ("mask_value", C.c_float), ("method", C.c_int32), ("vocab", C.c_int32), ("blank_id", C.c_int32), ("zero_infinity", C.c_int32),
        ("mlp_decoder", C.c_int32), ("dec_act", C.c_int32), ("loss", C.c_int32), ("dtype", C.c_int32), ("fp8_qkv", C.c_int32), ("residual_dtype", C.c_int32)]


class PtstIO(C.Structure):
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("spikes", C.c_void_p), ("spikes_mask", C.c_void_p), ("spikes_lengths", C.c_void_p),
                ("targets", C.c_void_p), ("targets_lengths", C.c_void_p), ("ext_mask", C.c_void_p),
                ("train", C.c_int32), ("want_grad", C.c_int32), ("seed", C.c_uint32), ("grad_scale", C.c_float),
                ("aux", C.c_void_p), ("nbt", C.c_void_p), ("preds", C.c_void_p), ("patch_input", C.c_void_p), ("mask_out", C.c_void_p),
                ("loss", C.c_void_p), ("n_examples", C.c_void_p), ("argmax", C.c_void_p), ("hidden_out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64)]


_SIGNATURES = {
    "nbci_ptst_plan_create": (C.c_int, [C.POINTER(PtstConfig), C.POINTER(C.c_void_p)]),
    "nbci_ptst_plan_destroy": (None, [C.c_void_p]),
    "nbci_ptst_param_count": (C.c_int64, [C.c_void_p]),
    "nbci_ptst_num_params": (C.c_int32, [C.c_void_p]),
    "nbci_ptst_num_segments": (C.c_int32, [C.c_void_p]),
    "nbci_ptst_num_patches": (C.c_int32, [C.c_void_p]),
    "nbci_ptst_aux_floats": (C.c_int64, [C.c_void_p]),
    "nbci_ptst_param_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nbci_ptst_segment_range": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nbci_ptst_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "nbci_ptst_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PtstIO), C.c_void_p]),
    "nbci_ptst_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PtstIO), C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "nbci_masker": (C.c_int, [C.POINTER(MaskerDesc), C.c_void_p]),
    "nbci_attention_small_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_attention_small_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 5 + [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_attention_flash_fwd": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_attention_flash_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_itr_plan_create": (C.c_int, [C.POINTER(ItrConfig), C.POINTER(C.c_void_p)]),
    "nbci_itr_plan_destroy": (None, [C.c_void_p]),
    "nbci_itr_param_count": (C.c_int64, [C.c_void_p]),
    "nbci_itr_num_params": (C.c_int32, [C.c_void_p]),
    "nbci_itr_num_segments": (C.c_int32, [C.c_void_p]),
    "nbci_itr_param_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "nbci_itr_segment_range": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nbci_itr_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "nbci_itr_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ItrIO), C.c_void_p]),
    "nbci_itr_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ItrIO), C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "nbci_ndt1_plan_create": (C.c_int, [C.POINTER(NDT1Config), C.POINTER(C.c_void_p)]),
    "nbci_ndt1_plan_destroy": (None, [C.c_void_p]),
    "nbci_ndt1_param_count": (C.c_int64, [C.c_void_p]),
    "nbci_ndt1_num_params": (C.c_int32, [C.c_void_p]),
    "nbci_ndt1_num_segments": (C.c_int32, [C.c_void_p]),
    "nbci_ndt1_param_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]),
    "nbci_ndt1_segment_range": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nbci_ndt1_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "nbci_ndt1_tokens": (C.c_int32, [C.c_void_p, C.c_int32]),
    "nbci_ndt1_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NDT1IO), C.c_void_p]),
    "nbci_ndt1_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(NDT1IO), C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "nbci_adamw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                             C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "nbci_streamk_timeouts": (C.c_int, [C.POINTER(C.c_int64)]),
    "nbci_stream_order": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nbci_debug_mlp_strip": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), C.c_void_p]),
    "nbci_profile_collect_text": (C.c_int, [C.c_char_p, C.c_int64]),
    "nbci_adamw_lp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "nbci_adamw_zero": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "nbci_smooth_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "nbci_layernorm_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "nbci_layernorm_bwd": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "nbci_layernorm_fwd_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_void_p]),
    "nbci_layernorm_bwd_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                        C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "nbci_softmax_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 7 +
                         [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_softmax_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 +
                         [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_logsoftmax": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "nbci_ctc_workspace_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "nbci_ctc": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                           C.c_float, C.c_void_p]),
    "nbci_cast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "nbci_gemm_grouped": (C.c_int, [C.POINTER(GemmDesc), C.c_int32, C.c_void_p]),
    "nbci_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_float, C.c_uint32, C.c_uint32,
                                                                                               C.c_uint32, C.c_void_p]),
    "nbci_attention_bwd": (C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 +
                           [C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]),
    "nbci_coupler_splice_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.c_int32] * 4 + [C.c_void_p]),
    "nbci_coupler_splice_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "nbci_colsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "nbci_profile_enable": (C.c_int, [C.c_int32]),
    "nbci_debug_gemm_pc": (C.c_int, [C.c_int32]),
    "nbci_debug_gemm_streamk": (C.c_int, [C.c_int32]),
    "nbci_release_scratch": (C.c_int, []),
    "nbci_debug_gemm_grouped_plan": (C.c_int, [C.POINTER(GemmDesc), C.c_int32, C.POINTER(C.c_int32)]),
    "nbci_mx_quantize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "nbci_gemm_fp8": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "nbci_comm_unique_id": (C.c_int, [C.c_void_p]),
    "nbci_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    "nbci_comm_destroy": (None, [C.c_void_p]),
    "nbci_allreduce_bucket": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "nbci_set_available_cus": (C.c_int, [C.c_int32]),
    "nbci_debug_occupy_cus": (C.c_int, [C.c_int32, C.c_int32, C.c_double, C.c_void_p]),
    "nbci_profile_collect": (C.c_int, [C.POINTER(C.c_double)]),
    "nbci_step_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]),
    "nbci_per": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


_lib = None


def lib():
    """Load libnbci.so once; raise NbciUnavailable (never fall back) if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NbciUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C llm_bci_amd/csrc` (needs hipcc). There is no CPU fallback for the product path.")
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    l = C.CDLL(LIB_PATH)
    l.nbci_version.restype = C.c_int
    l.nbci_last_error.restype = C.c_char_p
    l.nbci_gemm.restype = C.c_int
    l.nbci_gemm.argtypes = [C.POINTER(GemmDesc), C.c_void_p]
    _bind_rest(l)
    _lib = l
    return l


def _bind_rest(l):
    """argtypes for the remaining entry points (filled in as they are added)."""
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = restype
        fn.argtypes = argtypes




def check(status, what=""):
    if status != 0:
        msg = lib().nbci_last_error().decode("utf-8", "replace")
        raise NbciError(f"{what} failed with status {status}: {msg}")


def exported_symbols():
    """Names declared in include/nbci.h (used by the CPU test that checks the .so exports them)."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "nbci.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbci_[a-z0-9_]+)\s*\(", text)))
