"""ctypes binding of libmdlm_hip.so (the C-ABI declared in include/mdlm_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build with `python __graft_entry__.py build` (or
`make -C mdlm_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDLM_HIP_LIB: another build of the same library (same-box A/B measurements)
LIB_PATH = os.environ.get('MDLM_HIP_LIB') or os.path.join(_HERE, 'libmdlm_hip.so')

F32, BF16 = 0, 1
EPI_STORE, EPI_BIAS_GELU, EPI_GATE_RESIDUAL, EPI_QKV_ROTARY = 0, 1, 2, 3

_vp, _i64, _i32, _u64, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_double
_f32 = C.c_float

# name -> argtypes; every function returns int.
SIGNATURES = {
  'mdlm_subs_logprobs': [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _i32, _vp],
  'mdlm_subs_row_lse': [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp, _vp],
  'mdlm_subs_argmax': [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _vp],
  'mdlm_ddpm_cache_update': [_vp, _i64, _i32, _vp, _vp, _dbl, _vp, _u64, _u64,
                             _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
  'mdlm_subs_ddpm_update': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _dbl, _vp, _u64,
                            _u64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
  'mdlm_subs_race_update': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _u64,
                            _u64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
  'mdlm_race_key_probe': [_vp, _vp, _vp, _vp, _i64, _vp],
  'mdlm_d3pm_loss': [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp],
  'mdlm_lazy_select': [_vp, _vp, _dbl, _u64, _u64, _i32, _i32, _i32, _i32, _vp, _vp,
                       _vp, _vp, _vp],
  'mdlm_lazy_race': [_vp, _i64, _i32, _vp, _vp, _i32, _u64, _u64, _vp, _i32, _i32, _vp],
  'mdlm_scatter_rows_f32': [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp],
  'mdlm_gather_rows': [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp],
  'mdlm_philox_uniform': [_vp, _i64, _i32, _u64, _u64, _vp],
  'mdlm_q_xt': [_vp, _vp, _vp, _u64, _u64, _vp, _i32, _i32, _i32, _vp],
  'mdlm_subs_nll': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32,
                    _i32, _vp],
  'mdlm_nll_sums': [_vp, _vp, _vp, _i64, _vp],
  'mdlm_embed_rows': [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _vp, _vp],
  'mdlm_timestep_embedding': [_vp, _vp, _i32, _i32, _vp],
  'mdlm_linear_f32': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
  'mdlm_ln_modulate': [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i64, _vp, _vp,
                       _vp],
  'mdlm_masked_row_map': [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
  'mdlm_gemm_bf16': [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32,
                     _i32, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _i32, _vp, _vp,
                     _vp, _vp, _i32, _i32, _vp],
  'mdlm_lse_combine': [_vp, _i64, _i32, _vp, _vp, _vp, _vp],
  'mdlm_attn_fwd': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
  # training step
  'mdlm_attn_fwd_lse': [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
  'mdlm_attn_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
  'mdlm_qkv_grad_pack': [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
  'mdlm_ln_modulate_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i64, _i32, _vp],
  'mdlm_colsum_reduce': [_vp, _vp, _i64, _i32, _i32, _vp],
  'mdlm_colsum_bf16_partial': [_vp, _i64, _vp, _i64, _i32, _vp],
  'mdlm_dropout_gate_residual_fwd': [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i64, _f32, _u64, _u64, _vp],
  'mdlm_dropout_gate_residual_bwd': [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _i64, _f32, _u64, _u64, _vp],
  'mdlm_gelu_fwd': [_vp, _vp, _i64, _vp],
  'mdlm_gelu_bwd': [_vp, _vp, _vp, _i64, _vp],
  'mdlm_transpose_bf16': [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp],
  'mdlm_embed_grad': [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp],
  'mdlm_subs_nll_bwd': [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
  'mdlm_gemm_bf16_splitk': [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _i64, _i32, _vp],
  'mdlm_gemm_bf16_gelu': [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
  'mdlm_gemm_tn_splitk': [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
  'mdlm_splitk_reduce': [_vp, _vp, _i64, _i32, _i64, _vp],
  'mdlm_grad_sumsq': [_vp, _i64, _vp, _vp, _vp],
  'mdlm_grad_sumsq_static': [_vp, _i64, _vp, _vp],
  'mdlm_adamw_ema': [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _vp, _f32, _f32, _vp],
  'mdlm_compact_active': [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
}

# workspace-size queries: host arithmetic, return int64 bytes (-1: bad shape)
SIGNATURES_I64 = {
  'mdlm_attn_bwd_workspace_bytes': [_i32, _i32, _i32],
  'mdlm_grad_sumsq_workspace_bytes': [],
  'mdlm_gemm_bf16_splitk_workspace_bytes': [_i64, _i32, _i32],
  'mdlm_gemm_tn_splitk_workspace_bytes': [_i32, _i32, _i32, _vp],
  'mdlm_embed_grad_workspace_bytes': [_i64, _i32],
  'mdlm_ln_modulate_bwd_workspace_bytes': [_i64, _i32],
  'mdlm_dropout_gate_residual_bwd_workspace_bytes': [_i64, _i32],
  'mdlm_colsum_bf16_partial_workspace_bytes': [_i64, _i32],
  'mdlm_transpose_bf16_workspace_bytes': [_i32, _i32],
  'mdlm_masked_row_map_workspace_bytes': [_i64, _i32],
  'mdlm_gemm_lse_partials_workspace_bytes': [_i64, _i32],
}

_lib = None


def load():
  """Load the HIP library (once).  Raises RuntimeError if it is not built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
      f'{LIB_PATH} is missing: the MI355X HIP extension is not built and '
      'mdlm_amd has no CPU fallback. Run `python __graft_entry__.py build`.')
  lib = C.CDLL(LIB_PATH)
  lib.mdlm_version.restype = C.c_int
  lib.mdlm_version.argtypes = []
  lib.mdlm_last_error_string.restype = C.c_char_p
  lib.mdlm_last_error_string.argtypes = []
  _lib = lib
  return lib


def _fn(lib, name):
  fn = getattr(lib, name)  # AttributeError -> the library lacks the symbol
  if fn.argtypes is None:
    fn.restype = C.c_int
    fn.argtypes = SIGNATURES[name]
  return fn


def check_symbols():
  """Every entry point declared in include/mdlm_hip.h must be exported."""
  lib = load()
  names = list(SIGNATURES) + list(SIGNATURES_I64)
  missing = [n for n in names if not hasattr(lib, n)]
  if missing:
    raise RuntimeError(f'libmdlm_hip.so lacks symbols: {missing}')
  return sorted(names) + ['mdlm_version', 'mdlm_last_error_string']


def call_i64(name, *args):
  """A workspace-size query (no GPU involved): returns its int64 result."""
  lib = load()
  fn = getattr(lib, name)
  if fn.argtypes is None:
    fn.restype = C.c_int64
    fn.argtypes = SIGNATURES_I64[name]
  if name == 'mdlm_gemm_tn_splitk_workspace_bytes' and len(args) == 3:
    args = args + (None,)
  return int(fn(*args))


def call(name, *args):
  lib = load()
  rc = _fn(lib, name)(*args)
  if rc != 0:
    msg = lib.mdlm_last_error_string().decode('utf-8', 'replace')
    raise RuntimeError(f'{name} failed (rc={rc}): {msg}')
