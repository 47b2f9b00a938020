"""ctypes binding of the C-ABI kernel library (include/lmod_hip.h).

The product path has NO fallback: if ``liblmod_hip.so`` is missing or a symbol is absent, importing
an op raises.  PyTorch is used here only to own device memory and the stream.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMOD_HIP_LIB", os.path.join(_HERE, "_lib", "liblmod_hip.so"))   # override: kernel A/B tuning only

_P, _I, _Q, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float

# name -> argument type string (p pointer, i int, q long long, f float); every function returns int
# and takes the hipStream_t last ('p').  Kept in the same order as include/lmod_hip.h.
SIGNATURES = {
    "lmod_gemm_bf16_nt": "pppp" + "iiiiii" + "iqqq" + "pp" + "iii" + "p",
    "lmod_gemm_bf16_nt_res": "ppppp" + "iiiiiii" + "p",
    "lmod_gemm_qkv_rope_bf16": "pppp" + "iiiiii" + "ppp" + "i" + "p",
    "lmod_gemm_swiglu_bf16": "pppp" + "iiiiiii" + "iqqqq" + "p" + "p",
    "lmod_gemm_swiglu_bwd_bf16": "pppp" + "iiiiiii" + "iqqqq" + "p" + "p",
    "lmod_gemm_wgrad_bf16_nt": "ppp" + "iiiiii" + "i" + "pq" + "p",
    "lmod_gemm_bf16_tn": "ppp" + "iiiiii" + "iqqq" + "p" + "ii" + "p",
    "lmod_transpose_bf16": "pp" + "iiii" + "iqq" + "p" + "p",
    "lmod_rmsnorm_fwd": "pppppp" + "iif" + "p",
    "lmod_rmsnorm_bwd": "pppppp" + "ii" + "p",
    "lmod_rmsnorm_dw": "pppp" + "p" + "iii" + "p",
    "lmod_embed_wgrad": "ppp" + "qi" + "p",
    "lmod_layernorm_fwd": "pppp" + "iif" + "p",
    "lmod_rope": "pppp" + "iiiii" + "p",
    "lmod_swiglu_fwd": "ppp" + "qiiii" + "ip" + "p",
    "lmod_swiglu_bwd": "ppppp" + "qiiiiii" + "ip" + "p",
    "lmod_gelu_fwd": "pp" + "q" + "p",
    "lmod_gelu_bwd": "ppp" + "q" + "p",
    "lmod_add_bf16": "ppp" + "q" + "p",
    "lmod_gather_rows": "pppp" + "qi" + "p",
    "lmod_im2col_patch": "pp" + "iiii" + "p",
    "lmod_vit_embed": "pppp" + "iii" + "p",
    "lmod_adamw_step": "ppppp" + "q" + "fffff" + "i" + "f" + "i" + "p" + "p",
    "lmod_sumsq_f32": "pq" + "pp" + "i" + "p",
    "lmod_clip_coef": "pff" + "pp" + "p",
    "lmod_cast_f32_bf16": "pp" + "qi" + "p",
    "lmod_attn_fwd": "ppppppp" + "iiiii" + "iiii" + "f" + "i" + "p",
    "lmod_attn_bwd": "pppppppppppp" + "iiiii" + "iiiiiiii" + "f" + "i" + "p",
    "lmod_attn_bwd_rope": "pppppppppppp" + "iiiii" + "iiiiiiii" + "f" + "i" + "ppp" + "p",
    "lmod_attn_bwd_nsplit": "iiiiii",                      # host-side query, no stream: call it through load()
    "lmod_attn_bwd_split": "pppppppppppp" + "iiiii" + "iiiiiiii" + "f" + "i" + "ppp" + "pq" + "p",
    "lmod_splice_count": "pp" + "iiii" + "pp" + "p",
    "lmod_splice_fill": "ppp" + "iiii" + "pp" + "pppp" + "p",
    "lmod_lossplan_count": "p" + "iiiii" + "p" + "p",
    "lmod_lossplan_fill": "p" + "iiiii" + "p" + "ppppppp" + "p",
    "lmod_row_argmax_bf16": "pqi" + "pi" + "p",
    "lmod_attn_decode": "ppppp" + "iiiiiiii" + "f" + "p",
    "lmod_moe_router_fwd": "ppp" + "iii" + "p",
    "lmod_moe_gate": "pp" + "iiii" + "pppppppppppppp" + "iQQp" + "p",
    "lmod_moe_combine_fwd": "pppppp" + "ii" + "p",
    "lmod_moe_combine_bwd": "ppppppppp" + "iii" + "p",
    "lmod_moe_gate_bwd": "pppppppppp" + "iii" + "p",
    "lmod_moe_dispatch_bwd": "pppppp" + "iii" + "p",
    "lmod_moe_router_wgrad": "pppp" + "iiii" + "p",
    "lmod_moe_residual_mix_fwd": "pppp" + "pp" + "ii" + "p",
    "lmod_moe_residual_mix_bwd": "pppp" + "ppp" + "ii" + "p",
    "lmod_small_linear_dgrad": "ppp" + "iii" + "p",
    "lmod_rowloss_fwd": "pqi" + "pqi" + "pp" + "i" + "p",
    "lmod_rowloss_bwd": "pqi" + "pqi" + "pp" + "ppppp" + "pq" + "i" + "p",
    "lmod_segment_wsum": "pii" + "pp" + "i" + "pp" + "p",
    "lmod_row_softmax_f32": "pqii" + "pi" + "p",
    "lmod_rowdot_masked": "pp" + "ii" + "p" + "p",
    "lmod_dpo_loss": "pppp" + "iffi" + "ppppp" + "p",
    # collectives (csrc/comm.hip).  The three communicator life-cycle functions take no stream: call them through load().
    "lmod_comm_unique_id": "p",
    "lmod_comm_init": "pp" + "ii",
    "lmod_comm_destroy": "p",
    "lmod_allreduce_grads": "pp" + "qi" + "p",
    "lmod_reduce_scatter_grads": "pp" + "qi" + "p",
    "lmod_allgather_params": "pp" + "qi" + "p",
    "lmod_moe_all_to_all": "ppp" + "pp" + "i" + "p",
}
_CT = {"p": _P, "i": _I, "q": _Q, "f": _F, "Q": ctypes.c_ulonglong}
_ERR = {-1: "LMOD_EINVAL (bad pointer/shape/alignment)", -2: "LMOD_ELAUNCH (HIP launch error)",
        -3: "LMOD_EUNSUPPORTED (outside compiled envelope)"}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP kernels first (python __graft_entry__.py build, or "
            f"`make -C llava-mod_amd/csrc`).  There is no CPU/PyTorch fallback for the product path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud failure
        fn.restype = _I
        fn.argtypes = [_CT[c] for c in sig]
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


# Launch trace (measurement only, off by default): `TRACE = {"names": {...}, "rows": []}` makes `call` bracket every launch of
# the named entry points with two events on the launch stream and keep (name, integer arguments, start, stop).  bench.py uses it
# for ONE extra, untimed step: the in-step aggregate rate of the dominant kernel (sum of flops / sum of durations).
TRACE = None


UNSUPPORTED = -3


def call(name, *args, allow=()):
    """Invoke `name` on the current torch stream; raise on a non-zero status (statuses listed in `allow` are returned instead:
    a caller that has a documented two-step form for LMOD_EUNSUPPORTED shapes passes `allow=(UNSUPPORTED,)`)."""
    lib = load()
    if TRACE is not None and name in TRACE["names"]:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args, stream())
        e1.record()
        TRACE["rows"].append((name, tuple(a if isinstance(a, int) else (0 if a is None else 1) for a in args), e0, e1))
    else:
        rc = getattr(lib, name)(*args, stream())
    if rc != 0 and rc not in allow:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
    return rc
