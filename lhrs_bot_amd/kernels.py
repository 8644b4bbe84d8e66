"""Thin host wrappers: torch tensors (device memory + streams only) -> raw pointers -> the C ABI.

No arithmetic happens here and nothing falls back to torch: every function enqueues one or two HIP kernels
of liblhrs_hip.so on the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_QUICK_GELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3
MAP_GELU, MAP_GELU_BWD, MAP_ADD, MAP_QUICK_GELU = 0, 1, 2, 3


def _L():
    return _lib.load()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")


# --------------------------------------------------------------------------------------------- GEMM
_GEMM_WS = {}


def ensure_gemm_workspace(device) -> None:
    """Give the library its GEMM workspace on `device` (caller-owned, 64 MiB + 4 KiB: include/lhrs_hip.h) - once per device and process;
    the towers call this when they are built.  One user inside the library: the split-K launch for the tail rows of a row-split product
    with a long k-loop.  LHRS_GEMM_WORKSPACE=0: nothing is registered (those tail rows then run whole-K small tiles)."""
    dev = torch.device(device)
    if dev.type != "cuda" or os.environ.get("LHRS_GEMM_WORKSPACE", "1") == "0":
        return
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _GEMM_WS:
        return
    with torch.cuda.device(idx):
        n = int(_L().lhrs_gemm_workspace_bytes())
        ws = torch.empty(n, device=f"cuda:{idx}", dtype=torch.uint8)
        torch.cuda.synchronize()
        _lib.check(_L().lhrs_gemm_set_workspace(ws.data_ptr(), n), "gemm_set_workspace")
    _GEMM_WS[idx] = ws


def gemm_set_u4(on: bool) -> None:
    """The four-wave gemm_u4_kernel (csrc/gemm_u4.hip) for the plain long-k products and the fused-epilogue products it covers (default on; LHRS_GEMM_U4=0:
    the 16-wave kernels everywhere - kernel A/B tests)."""
    _L().lhrs_gemm_set_u4(int(bool(on)))


def gemm_u4_takes(M, N, K, lda=None, ldb=None, ldc=None, ldr=0, has_bias=False, act=0, out_f32=False, accumulate=False, alpha=1.0) -> bool:
    """The shape rule of lhrs_gemm_bf16_nt: True when that problem runs on gemm_u4_kernel (a pure function of the arguments)."""
    return bool(_L().lhrs_gemm_u4_takes(int(M), int(N), int(K), int(K if lda is None else lda), int(K if ldb is None else ldb), int(N if ldc is None else ldc),
                                        int(ldr), int(has_bias), int(act), int(out_f32), int(accumulate), float(alpha)))


GEMM_KIND_NAMES = ("gemm_nt_256s_kernel plain", "gemm_nt_256s_kernel SwiGLU-fwd", "gemm_nt_256s_kernel SwiGLU-bwd", "gemm_nt_256s_kernel RoPE", "gemm_nt_144s_kernel plain",
                   "gemm_u4_kernel<0, true> plain + residual", "gemm_u4_kernel<0, false> plain", "gemm_u4_kernel<1, false> SwiGLU-fwd", "gemm_u4_kernel<2, false> SwiGLU-bwd",
                   "gemm_u4_kernel<3, false> RoPE")


class gemm_kernel_census:
    """with gemm_kernel_census() as c: ...  -> c.counts = {kernel instantiation: timed launches} of the persistent GEMM kernels that ran inside the block (the library's live
    profile counters, include/lhrs_hip.h: lhrs_gemm_profile_*): which kernel the shape rules gave each product - the parity tests record it next to their numbers."""

    def __enter__(self):
        _lib.check(_L().lhrs_gemm_profile_stride(1), "gemm_profile_stride")
        _lib.check(_L().lhrs_gemm_profile_enable(20000), "gemm_profile_enable")
        self.counts = {}
        return self

    def __exit__(self, *exc):
        kinds = (ctypes.c_double * 30)()
        torch.cuda.synchronize()
        _lib.check(_L().lhrs_gemm_profile_read_kinds(ctypes.addressof(kinds)), "gemm_profile_read_kinds")
        _L().lhrs_gemm_profile_enable(0)
        self.counts = {GEMM_KIND_NAMES[k]: int(kinds[3 * k]) for k in range(10) if kinds[3 * k] > 0}
        return False


def gemm_u4_nt(a, b, out, residual=None, a2=None, b2=None) -> bool:
    """The raw launch of gemm_u4_kernel (tests, tools): True when launched, False when the problem is not its kind.  a2 / b2: the fused LoRA pair."""
    ldr = residual.stride(0) if residual is not None else 0
    if a2 is not None:
        st = _L().lhrs_gemm_u4_nt_lora(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0),
                                       a2.shape[1], out.data_ptr(), out.stride(0), a.shape[0], b.shape[0], a.shape[1], _p(residual), ldr, _stream())
    else:
        st = _L().lhrs_gemm_u4_nt(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), a.shape[0], b.shape[0], a.shape[1],
                                  _p(residual), ldr, _stream())
    if st < 0:
        _lib.check(st, "gemm_u4_nt")
    return st == 0


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, bias=None, residual=None,
            act: int = ACT_NONE, out_f32: bool = False, accumulate: bool = False, alpha: float = 1.0,
            M: Optional[int] = None, N: Optional[int] = None, K: Optional[int] = None) -> torch.Tensor:
    """out[M,N] = act(alpha * a[M,K] @ b[N,K]^T + bias) + residual.  a/b/out may be strided row views."""
    _req(a, torch.bfloat16, "a"); _req(b, torch.bfloat16, "b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M = a.shape[0] if M is None else M
    N = b.shape[0] if N is None else N
    K = a.shape[1] if K is None else K
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    assert out.stride(1) == 1 and out.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    ldr = residual.stride(0) if residual is not None else 0
    st = _L().lhrs_gemm_bf16_nt(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                _p(bias), _p(residual), ldr, act, int(out_f32), int(accumulate), float(alpha), _stream())
    _lib.check(st, "gemm_bf16_nt")
    return out


def gemm_nt_lora(a, b, a2, b2, out=None, *, bias=None, residual=None, out_f32=False, accumulate=False, alpha=1.0):
    """out = alpha * (a @ b^T + a2 @ b2^T) + bias + residual  (rank-K2 LoRA update fused into the base GEMM's k-loop)."""
    M, K = a.shape
    N, K2 = b.shape[0], a2.shape[1]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    ldr = residual.stride(0) if residual is not None else 0
    st = _L().lhrs_gemm_bf16_nt_lora(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), a2.data_ptr(), a2.stride(0), b2.data_ptr(),
                                     b2.stride(0), K2, out.data_ptr(), out.stride(0), M, N, K, _p(bias), _p(residual), ldr,
                                     int(out_f32), int(accumulate), float(alpha), _stream())
    _lib.check(st, "gemm_bf16_nt_lora")
    return out


def gemm_swiglu_fwd(x, w_gu, ff, a2=None, b2=None):
    """(gu [M, 2ff], act [M, ff]) = LLaMA MLP first half in ONE launch: gu = x @ w_gu^T (+ a2 @ b2^T), act = silu(gate) * up."""
    M, K = x.shape
    assert w_gu.shape[0] == 2 * ff
    gu = torch.empty((M, 2 * ff), device=x.device, dtype=torch.bfloat16)
    act = torch.empty((M, ff), device=x.device, dtype=torch.bfloat16)
    K2 = a2.shape[1] if a2 is not None else 0
    st = _L().lhrs_gemm_swiglu_fwd(x.data_ptr(), x.stride(0), w_gu.data_ptr(), w_gu.stride(0), _p(a2), a2.stride(0) if K2 else 0, _p(b2),
                                   b2.stride(0) if K2 else 0, K2, gu.data_ptr(), gu.stride(0), act.data_ptr(), act.stride(0), M, ff, K,
                                   _stream())
    _lib.check(st, "gemm_swiglu_fwd")
    return gu, act


def gemm_rope_fwd(x, w, cos_t, sin_t, *, pos_mod, pos0=0, rope_cols, head_dim, a2=None, b2=None, out=None):
    """out [M, N] = x @ w^T (+ a2 @ b2^T) with RoPE applied to the heads in columns [0, rope_cols) inside the GEMM epilogue (one launch;
    bit-identical to gemm_nt(+lora) followed by rope_)."""
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=x.device, dtype=torch.bfloat16) if out is None else out
    K2 = a2.shape[1] if a2 is not None else 0
    st = _L().lhrs_gemm_rope_fwd(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _p(a2), a2.stride(0) if K2 else 0, _p(b2),
                                 b2.stride(0) if K2 else 0, K2, out.data_ptr(), out.stride(0), M, N, K, cos_t.data_ptr(), sin_t.data_ptr(),
                                 int(pos_mod), int(pos0), int(rope_cols), int(head_dim), _stream())
    _lib.check(st, "gemm_rope_fwd")
    return out


def gemm_swiglu_bwd(dy, w_down_t, gu, ff, a2=None, b2=None, out=None):
    """dgu [M, 2ff] = swiglu'(gu) * (dy @ w_down_t^T (+ a2 @ b2^T)) in ONE launch; out defaults to gu (in place)."""
    M, K = dy.shape
    assert w_down_t.shape[0] == ff and gu.shape == (M, 2 * ff)
    dgu = gu if out is None else out
    K2 = a2.shape[1] if a2 is not None else 0
    L = _L()
    scratch = None
    if not L.lhrs_gemm_swiglu_fusable(M, ff, K, K, K2):
        scratch = torch.empty((M, ff), device=dy.device, dtype=torch.bfloat16)
    st = L.lhrs_gemm_swiglu_bwd(dy.data_ptr(), dy.stride(0), w_down_t.data_ptr(), w_down_t.stride(0), _p(a2), a2.stride(0) if K2 else 0,
                                _p(b2), b2.stride(0) if K2 else 0, K2, gu.data_ptr(), dgu.data_ptr(), gu.stride(0), _p(scratch), M, ff, K,
                                _stream())
    _lib.check(st, "gemm_swiglu_bwd")
    return dgu


def gemm_fp8_nt(a8, sa, b8, sb, out=None, *, residual=None, alpha=1.0, a2=None, b2=None):
    """out[M, N] bf16 = alpha * (sa[:, None] * sb[None, :] * (a8 @ b8^T) + a2 @ b2^T) (+ residual); a8 [M, K], b8 [N, K] uint8 e4m3 with
    per-row fp32 scales; the optional bf16 pair a2 [M, K2], b2 [N, K2] (LoRA update) is accumulated by the same launch."""
    M, K = a8.shape
    N = b8.shape[0]
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and sa.dtype == torch.float32 and sb.dtype == torch.float32
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=torch.bfloat16)
    ldr = residual.stride(0) if residual is not None else 0
    if a2 is None:
        st = _L().lhrs_gemm_fp8_nt(a8.data_ptr(), a8.stride(0), sa.data_ptr(), b8.data_ptr(), b8.stride(0), sb.data_ptr(), out.data_ptr(),
                                   out.stride(0), M, N, K, _p(residual), ldr, float(alpha), _stream())
    else:
        st = _L().lhrs_gemm_fp8_nt_lora(a8.data_ptr(), a8.stride(0), sa.data_ptr(), b8.data_ptr(), b8.stride(0), sb.data_ptr(), a2.data_ptr(),
                                        a2.stride(0), b2.data_ptr(), b2.stride(0), a2.shape[1], out.data_ptr(), out.stride(0), M, N, K,
                                        _p(residual), ldr, float(alpha), _stream())
    _lib.check(st, "gemm_fp8_nt")
    return out


def gemm_nt_skinny(a, b, alpha=1.0):
    """out[M, N] bf16 = alpha * a[M, K] @ b[N, K]^T for a skinny N (64..384): split-K launch + reduction (LoRA down-projections)."""
    M, K = a.shape
    N = b.shape[0]
    if N > 256 or N % 64 or M < 1024 or K < 1024:  # N = 384 at K = 4096: 54.5 us here against 42.8 us on the plain path (M = 8190)
        return gemm_nt(a, b, alpha=alpha)
    out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    ws = torch.empty(_L().lhrs_gemm_skinny_splits(K, N) * M * N, device=a.device, dtype=torch.float32)
    st = _L().lhrs_gemm_bf16_nt_skinny(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                       float(alpha), ws.data_ptr(), _stream())
    _lib.check(st, "gemm_bf16_nt_skinny")
    return out


def gemm_nt_splitk_f32(a, b, out):
    """out[M, N] fp32 = a[M, K] @ b[N, K]^T for a long K and few output tiles (weight gradients): split-K launch + ordered reduction."""
    M, K = a.shape
    N = b.shape[0]
    L = _L()
    splits = L.lhrs_gemm_splitk_splits(M, N, K)
    ws = torch.empty(splits * M * N, device=a.device, dtype=torch.float32) if splits > 1 else None
    st = L.lhrs_gemm_bf16_nt_splitk_f32(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                        _p(ws), _stream())
    _lib.check(st, "gemm_bf16_nt_splitk_f32")
    return out


def gemm_tn_f32(p, q, out):
    """out[Mo, No] fp32 = p[T, Mo]^T @ q[T, No] - weight gradients dW = dY^T X from the token-major operands as they lie (no transposes).
    Mo, No multiples of 128; row strides free (multiples of 8)."""
    T, Mo = p.shape
    No = q.shape[1]
    L = _L()
    splits = L.lhrs_gemm_tn_splits(T, Mo, No)
    ws = torch.empty(splits * Mo * No, device=p.device, dtype=torch.float32) if splits > 1 else None
    st = L.lhrs_gemm_tn_f32(p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), out.data_ptr(), out.stride(0), T, Mo, No, _p(ws), _stream())
    _lib.check(st, "gemm_tn_f32")
    return out


class BatchedTranspose:
    """out_i = in_i^T for a fixed list of (in, out) bf16 matrix pairs, refreshed by ONE launch (lhrs_transpose_batched)."""

    def __init__(self, pairs):
        rows, tiles = [], 0
        for src, dst in pairs:
            r, c = src.shape
            assert dst.shape == (c, r) and src.stride(1) == 1 and dst.stride(1) == 1
            rows.append([src.data_ptr(), dst.data_ptr(), src.stride(0), dst.stride(0), r, c, tiles])
            tiles += ((r + 63) // 64) * ((c + 63) // 64)
        self.keep = pairs
        self.n, self.tiles = len(rows), tiles
        self.desc = torch.tensor(rows, dtype=torch.int64).to(pairs[0][0].device)

    def run(self):
        _lib.check(_L().lhrs_transpose_batched(self.desc.data_ptr(), self.n, self.tiles, _stream()), "transpose_batched")


def dropout(x, p, seed, out=None):
    """peft lora_dropout on the adapter input: out = x * mask / (1 - p), mask regenerated from (seed, element index)."""
    rows, cols = x.shape
    out = torch.empty((rows, cols), device=x.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_dropout_bf16(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, cols, float(p), int(seed) & 0xFFFFFFFF,
                                      _stream()), "dropout")
    return out


def gemm_nt_dropmask(a, b, p, seed, residual=None, alpha=1.0):
    """out = mask * (alpha * a @ b^T) / (1 - p) + residual with the same counter-based mask as dropout() over the [M, N] result."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    st = _L().lhrs_gemm_bf16_nt_dropmask(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K,
                                         _p(residual), residual.stride(0) if residual is not None else 0, float(alpha), float(p),
                                         int(seed) & 0xFFFFFFFF, _stream())
    _lib.check(st, "gemm_bf16_nt_dropmask")
    return out


def gemm_tn_skinny(p, q, out, accumulate=False):
    """out[KP, N] (+)= p[M, KP]^T @ q[M, N]  (fp32 out; p, q token-major bf16, row strides free)."""
    M, KP = p.shape
    N = q.shape[1]
    ns = _L().lhrs_tn_skinny_splits(M, N)
    part = torch.empty(ns * KP * N, device=p.device, dtype=torch.float32)
    st = _L().lhrs_gemm_tn_skinny(p.data_ptr(), p.stride(0), q.data_ptr(), q.stride(0), out.data_ptr(), out.stride(0), part.data_ptr(),
                                  M, N, KP, int(accumulate), _stream())
    _lib.check(st, "gemm_tn_skinny")
    return out


def blockdiag_mask(g, r, w, active_mask):
    rows, cols = g.shape
    _lib.check(_L().lhrs_blockdiag_mask(g.data_ptr(), g.stride(0), rows, cols, r, w, active_mask, _stream()), "blockdiag_mask")
    return g


# --------------------------------------------------------------------------------------------- norms
def layernorm_fwd(x, gamma, beta, eps=1e-5, save_stats=False, out=None):
    rows, cols = x.shape
    y = torch.empty_like(x) if out is None else out
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    st = _L().lhrs_layernorm_fwd(x.data_ptr(), x.stride(0), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), y.stride(0),
                                 _p(mean), _p(rstd), rows, cols, eps, _stream())
    _lib.check(st, "layernorm_fwd")
    return (y, mean, rstd) if save_stats else y


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, accumulate=False, need_dx=True, add=None, out=None):
    rows, cols = x.shape
    dx = (torch.empty_like(x) if out is None else out) if need_dx else None
    part = None
    if dgamma is not None:
        nblk = _L().lhrs_layernorm_bwd_nblk(rows)
        part = torch.empty(nblk * 2 * cols, device=x.device, dtype=torch.float32)
    st = _L().lhrs_layernorm_bwd(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), gamma.data_ptr(), mean.data_ptr(),
                                 rstd.data_ptr(), _p(add), _p(dx), dx.stride(0) if dx is not None else 0, _p(dgamma), _p(dbeta),
                                 _p(part), int(accumulate), rows, cols, _stream())
    _lib.check(st, "layernorm_bwd")
    return dx


def rmsnorm_fwd(x, w, eps=1e-5, save_rstd=False, out=None):
    rows, cols = x.shape
    y = torch.empty_like(x) if out is None else out
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save_rstd else None
    st = _L().lhrs_rmsnorm_fwd(x.data_ptr(), x.stride(0), w.data_ptr(), y.data_ptr(), y.stride(0), _p(rstd), rows, cols, eps,
                               _stream())
    _lib.check(st, "rmsnorm_fwd")
    return (y, rstd) if save_rstd else y


def rmsnorm_bwd(dy, x, w, rstd=None, add=None, eps=1e-5, out=None):
    rows, cols = x.shape
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x) if out is None else out
    st = _L().lhrs_rmsnorm_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(rstd), _p(add), dx.data_ptr(), rows, cols, eps,
                               _stream())
    _lib.check(st, "rmsnorm_bwd")
    return dx


def rmsnorm_fwd_q(x, w, eps=1e-5, want_bf16=True, out=None, q_out=None):
    """-> (y bf16 or None, (y8 uint8 [rows, cols], scale fp32 [rows])): RMSNorm + per-row e4m3 copy of its output in one pass."""
    rows, cols = x.shape
    y = (torch.empty_like(x) if out is None else out) if want_bf16 else None
    y8, sc = q_out if q_out is not None else (torch.empty((rows, cols), device=x.device, dtype=torch.uint8),
                                              torch.empty(rows, device=x.device, dtype=torch.float32))
    st = _L().lhrs_rmsnorm_fwd_q(x.data_ptr(), x.stride(0), w.data_ptr(), _p(y), y.stride(0) if y is not None else 0, y8.data_ptr(),
                                 sc.data_ptr(), rows, cols, eps, _stream())
    _lib.check(st, "rmsnorm_fwd_q")
    return y, (y8, sc)


def rmsnorm_bwd_q(dy, x, w, rstd=None, add=None, eps=1e-5, out=None):
    """-> (dx bf16, (dx8, scale)): RMSNorm backward (+ residual add) + per-row e4m3 copy of dx."""
    rows, cols = x.shape
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x) if out is None else out
    d8 = torch.empty((rows, cols), device=x.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x.device, dtype=torch.float32)
    st = _L().lhrs_rmsnorm_bwd_q(dy.data_ptr(), x.data_ptr(), w.data_ptr(), _p(rstd), _p(add), dx.data_ptr(), d8.data_ptr(), sc.data_ptr(),
                                 rows, cols, eps, _stream())
    _lib.check(st, "rmsnorm_bwd_q")
    return dx, (d8, sc)


# --------------------------------------------------------------------------------------------- attention
def h2d(t: torch.Tensor, device) -> torch.Tensor:
    """Small host tensor -> device without stalling the host: a pageable source makes the copy wait for everything already queued on
    the stream (the launch queue drains, the GPU then idles until new work arrives); a pinned source with non_blocking=True is just
    another item in the queue."""
    if t.is_cuda:
        return t
    return t.pin_memory().to(device, non_blocking=True)


def make_desc(entries, device) -> torch.Tensor:
    """entries: list of (q_off, q_len, kv_off, kv_len[, kv_rows[, causal_off]]) -> int32 [n, 8] on device."""
    rows = []
    for e in entries:
        e = list(e)
        if len(e) == 4:
            e.append(e[3])
        if len(e) == 5:
            e.append(0)
        rows.append(e + [0, 0])
    return h2d(torch.tensor(rows, dtype=torch.int32), device)


def pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def seq_transpose(x, cols, LT, desc, nseq, which: str):
    """x: [tokens, ld] view (first `cols` columns used) -> [nseq, cols, LT] zero padded.  which: 'q' | 'kv' | 'kv_rows'."""
    out = torch.empty((nseq, cols, LT), device=x.device, dtype=torch.bfloat16)
    use = {"q": 0, "kv": 1, "kv_rows": 2}[which]
    st = _L().lhrs_seq_transpose(x.data_ptr(), x.stride(0), out.data_ptr(), cols, LT, desc.data_ptr(), nseq, use, _stream())
    _lib.check(st, "seq_transpose")
    return out


def attn_fwd(q, k, v, o, lse, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, key_mask=None):
    """key_mask: optional uint8 [nseq, >= kv_len] HF-style attention_mask over the keys (left-padded batched generate)."""
    if key_mask is None:
        st = _L().lhrs_attn_fwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), o.data_ptr(),
                                o.stride(0), _p(lse), desc.data_ptr(), nseq, H, D, max_q, max_kv, LTq, int(causal), float(scale),
                                _stream())
    else:
        assert key_mask.dtype == torch.uint8 and key_mask.dim() == 2 and key_mask.shape[0] == nseq and key_mask.stride(1) == 1
        st = _L().lhrs_attn_fwd_kmask(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), o.data_ptr(),
                                      o.stride(0), _p(lse), desc.data_ptr(), nseq, H, D, max_q, max_kv, LTq, int(causal),
                                      float(scale), key_mask.data_ptr(), key_mask.stride(0), _stream())
    _lib.check(st, "attn_fwd")


def attn_delta(o, dout, delta, desc, nseq, H, D, max_q, LTq):
    st = _L().lhrs_attn_delta(o.data_ptr(), o.stride(0), dout.data_ptr(), dout.stride(0), delta.data_ptr(), desc.data_ptr(),
                              nseq, H, D, max_q, LTq, _stream())
    _lib.check(st, "attn_delta")


def attn_bwd(q, k, v, dout, lse, delta, dq, dk, dv, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, rope=None):
    """rope = (cos_t, sin_t, pos_mod, pos0): q / k were rotated before the scores - dq / dk are returned as gradients of the UN-rotated
    projections (inverse rotation inside the stores of the resident kernels; a separate pass for long sequences), rows = dq.shape[0]."""
    if rope is None:
        st = _L().lhrs_attn_bwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), dout.data_ptr(),
                                dout.stride(0), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dq.stride(0), dk.data_ptr(),
                                dk.stride(0), dv.data_ptr(), dv.stride(0), desc.data_ptr(), nseq, H, D, max_q, max_kv, LTq,
                                int(causal), float(scale), _stream())
    else:
        cos_t, sin_t, pos_mod, pos0 = rope
        st = _L().lhrs_attn_bwd_rope(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), dout.data_ptr(),
                                     dout.stride(0), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dq.stride(0), dk.data_ptr(),
                                     dk.stride(0), dv.data_ptr(), dv.stride(0), desc.data_ptr(), nseq, H, D, max_q, max_kv, LTq,
                                     int(causal), float(scale), cos_t.data_ptr(), sin_t.data_ptr(), int(pos_mod), int(pos0), dq.shape[0],
                                     _stream())
    _lib.check(st, "attn_bwd")


def attn_bwd_o(q, k, v, dout, o, lse, delta, dq, dk, dv, desc, nseq, H, D, max_q, max_kv, LTq, causal, scale, rope=None):
    """attn_delta + attn_bwd in one call: `o` are the forward output rows, `delta` [nseq, H, LTq] fp32 is WRITTEN (the resident dQ kernel
    computes rowsum(dO * O) for the rows it loads anyway; longer sequences launch the delta kernel inside)."""
    cos_t, sin_t, pos_mod, pos0 = rope if rope is not None else (None, None, 1, 0)
    st = _L().lhrs_attn_bwd_o(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), dout.data_ptr(),
                              dout.stride(0), o.data_ptr(), o.stride(0), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dq.stride(0),
                              dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), desc.data_ptr(), nseq, H, D, max_q, max_kv, LTq,
                              int(causal), float(scale), _p(cos_t), _p(sin_t), int(pos_mod), int(pos0), dq.shape[0], _stream())
    _lib.check(st, "attn_bwd_o")


# --------------------------------------------------------------------------------------------- whole decoder layer (module-level ABI)
def llama_layer_forward(x, L, cos_t, sin_t, desc, B, S, LT, heads, ff, eps, h_scratch):
    """One frozen bf16 decoder layer in ONE library call (lhrs_llama_layer_forward): -> (x_out, saved dict like TextModal._layer_fwd's)."""
    M, d = x.shape
    dev, bf = x.device, torch.bfloat16
    qkv = torch.empty((M, 3 * d), device=dev, dtype=bf)
    o = torch.empty((M, d), device=dev, dtype=bf)
    lse = torch.empty((B, heads, LT), device=dev, dtype=torch.float32)
    x_mid = torch.empty((M, d), device=dev, dtype=bf)
    gu = torch.empty((M, 2 * ff), device=dev, dtype=bf)
    act = torch.empty((M, ff), device=dev, dtype=bf)
    x_out = torch.empty((M, d), device=dev, dtype=bf)
    st = _L().lhrs_llama_layer_forward(x.data_ptr(), L["ln1_w"].data_ptr(), L["qkv_w"].data_ptr(), L["o_w"].data_ptr(), L["ln2_w"].data_ptr(),
                                       L["gu_w"].data_ptr(), L["down_w"].data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), desc.data_ptr(), B, S, LT, d,
                                       heads, ff, float(eps), h_scratch.data_ptr(), qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), x_mid.data_ptr(),
                                       gu.data_ptr(), act.data_ptr(), x_out.data_ptr(), _stream())
    _lib.check(st, "llama_layer_forward")
    return x_out, dict(x_in=x, qkv=qkv, o=o, o_full=o, lse=lse, x_mid=x_mid, gu=gu)


def vit_layer_forward(x, L, desc, B, n, LT, heads, ff, h, qkv, o, f):
    """One frozen CLIP ViT encoder layer IN PLACE on x, in one library call (lhrs_vit_layer_forward)."""
    d = x.shape[1]
    st = _L().lhrs_vit_layer_forward(x.data_ptr(), L["ln1_w"].data_ptr(), L["ln1_b"].data_ptr(), L["qkv_w"].data_ptr(), L["qkv_b"].data_ptr(),
                                     L["o_w"].data_ptr(), L["o_b"].data_ptr(), L["ln2_w"].data_ptr(), L["ln2_b"].data_ptr(), L["fc1_w"].data_ptr(),
                                     L["fc1_b"].data_ptr(), L["fc2_w"].data_ptr(), L["fc2_b"].data_ptr(), desc.data_ptr(), B, n, LT, d, heads, ff,
                                     h.data_ptr(), qkv.data_ptr(), o.data_ptr(), f.data_ptr(), _stream())
    _lib.check(st, "vit_layer_forward")
    return x


def llama_layer_backward(dx_out, s, L, cos_t, sin_t, desc, B, S, LT, heads, ff, eps, delta, dqkv):
    """d loss / d x_out -> d loss / d x of one frozen bf16 decoder layer in ONE library call (lhrs_llama_layer_backward); s = the saved dict."""
    M, d = dx_out.shape
    dev, bf = dx_out.device, torch.bfloat16
    dh = torch.empty((M, d), device=dev, dtype=bf)
    d_o = torch.empty((M, d), device=dev, dtype=bf)
    dx_in = torch.empty((M, d), device=dev, dtype=bf)
    lib = _L()
    dact = None if lib.lhrs_gemm_swiglu_fusable(M, ff, d, d, 0) else torch.empty((M, ff), device=dev, dtype=bf)
    st = lib.lhrs_llama_layer_backward(dx_out.data_ptr(), s["x_in"].data_ptr(), s["x_mid"].data_ptr(), s["qkv"].data_ptr(), s["o_full"].data_ptr(),
                                       s["lse"].data_ptr(), s["gu"].data_ptr(), L["ln1_w"].data_ptr(), L["ln2_w"].data_ptr(), L["qkv_wT"].data_ptr(),
                                       L["o_wT"].data_ptr(), L["gu_wT"].data_ptr(), L["down_wT"].data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(),
                                       desc.data_ptr(), B, S, LT, d, heads, ff, float(eps), dh.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(),
                                       delta.data_ptr(), _p(dact), dx_in.data_ptr(), _stream())
    _lib.check(st, "llama_layer_backward")
    return dx_in, dh          # (d loss / d x, d loss / d x_mid)


# --------------------------------------------------------------------------------------------- element-wise
def patchify(rgb, P=14, KP=640):
    B, C, H, W = rgb.shape
    assert C == 3 and H == W
    _req(rgb, torch.float32, "rgb")
    rgb = rgb.contiguous()
    out = torch.empty((B * (H // P) ** 2, KP), device=rgb.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_patchify(rgb.data_ptr(), out.data_ptr(), B, H, P, KP, _stream()), "patchify")
    return out


def vit_assemble(patch, cls, pos, B, NP, dim):
    out = torch.empty((B * (NP + 1), dim), device=patch.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_vit_assemble(patch.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), B, NP, dim, _stream()),
               "vit_assemble")
    return out


def rope_(x, rows, nheads, D, cos_t, sin_t, pos_mod, pos0=0, inverse=False, pos_ids=None):
    st = _L().lhrs_rope(x.data_ptr(), x.stride(0), rows, nheads, D, cos_t.data_ptr(), sin_t.data_ptr(), _p(pos_ids), pos_mod,
                        pos0, int(inverse), _stream())
    _lib.check(st, "rope")


def swiglu_fwd(gate_up, F, out=None):
    rows = gate_up.shape[0]
    act = torch.empty((rows, F), device=gate_up.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_swiglu_fwd(gate_up.data_ptr(), act.data_ptr(), rows, F, _stream()), "swiglu_fwd")
    return act


def swiglu_fwd_q(gate_up, F, want_bf16=False, q_out=None):
    """-> (act bf16 or None, act8 uint8 [rows, F], scale fp32 [rows]): SwiGLU + per-row e4m3 quantisation in one pass."""
    rows = gate_up.shape[0]
    act = torch.empty((rows, F), device=gate_up.device, dtype=torch.bfloat16) if want_bf16 else None
    act8, sc = q_out if q_out is not None else (torch.empty((rows, F), device=gate_up.device, dtype=torch.uint8),
                                                torch.empty(rows, device=gate_up.device, dtype=torch.float32))
    _lib.check(_L().lhrs_swiglu_fwd_q(gate_up.data_ptr(), _p(act), act8.data_ptr(), sc.data_ptr(), rows, F, _stream()), "swiglu_fwd_q")
    return act, act8, sc


def swiglu_bwd_q(dact, gate_up, F, want_bf16=False):
    """-> (dgu bf16 written over gate_up or None, dgu8 uint8 [rows, 2F], scale fp32 [rows])."""
    rows = gate_up.shape[0]
    dgu8 = torch.empty((rows, 2 * F), device=gate_up.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=gate_up.device, dtype=torch.float32)
    _lib.check(_L().lhrs_swiglu_bwd_q(dact.data_ptr(), gate_up.data_ptr(), gate_up.data_ptr() if want_bf16 else None, dgu8.data_ptr(),
                                      sc.data_ptr(), rows, F, _stream()), "swiglu_bwd_q")
    return (gate_up if want_bf16 else None), dgu8, sc


def swiglu_bwd(dact, gate_up, F, out=None):
    rows = gate_up.shape[0]
    dgu = torch.empty_like(gate_up) if out is None else out
    _lib.check(_L().lhrs_swiglu_bwd(dact.data_ptr(), gate_up.data_ptr(), dgu.data_ptr(), rows, F, _stream()), "swiglu_bwd")
    return dgu


def map_(op, a, b=None, out=None):
    out = torch.empty_like(a) if out is None else out
    assert a.is_contiguous() and out.is_contiguous()
    _lib.check(_L().lhrs_map(op, a.data_ptr(), _p(b), out.data_ptr(), a.numel(), _stream()), "map")
    return out


def colsum(x, out, accumulate=False):
    rows, cols = x.shape
    ns = _L().lhrs_colsum_nsplit(rows)
    part = torch.empty(ns * cols, device=x.device, dtype=torch.float32)
    _lib.check(_L().lhrs_colsum(x.data_ptr(), x.stride(0), out.data_ptr(), part.data_ptr(), rows, cols, int(accumulate),
                                _stream()), "colsum")
    return out


def cast_f32_to_bf16(x, out=None):
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "cast")
    return out


def transpose(x, rows_pad=None, out=None):
    """x [rows, cols] (strided rows ok) -> [cols, rows_pad] with zero padding."""
    rows, cols = x.shape
    rows_pad = rows if rows_pad is None else rows_pad
    out = torch.empty((cols, rows_pad), device=x.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_transpose(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, cols, rows_pad, _stream()),
               "transpose")
    return out


# --------------------------------------------------------------------------------------------- pooler layout
def pooler_build(query, img, B, nq=(64, 48, 32), ni=(256, 256, 256)):
    dim = query.shape[1]
    NQ, KV = sum(nq), sum(nq) + sum(ni)
    t = torch.empty((B * NQ, dim), device=query.device, dtype=torch.bfloat16)
    kv = torch.empty((B * KV, dim), device=query.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_pooler_build(query.data_ptr(), img.data_ptr(), t.data_ptr(), kv.data_ptr(), B, *nq, *ni, dim, _stream()),
               "pooler_build")
    return t, kv


def pooler_query_grad(dt0, dkv, dquery, B, nq=(64, 48, 32), ni=(256, 256, 256), accumulate=False):
    dim = dquery.shape[1]
    _lib.check(_L().lhrs_pooler_query_grad(dt0.data_ptr(), _p(dkv), dquery.data_ptr(), B, *nq, *ni, dim, int(accumulate),
                                           _stream()), "pooler_query_grad")
    return dquery


def copy_2d(dst_ptr, dst_pitch, src_ptr, src_pitch, width_bytes, height):
    _lib.check(_L().lhrs_copy_2d(dst_ptr, dst_pitch, src_ptr, src_pitch, width_bytes, height, _stream()), "copy_2d")


# --------------------------------------------------------------------------------------------- token side
def splice_fwd(ids, labels, mask, image, embed, S):
    B, T = ids.shape
    NI, dim = image.shape[1], image.shape[2]
    _req(ids, torch.int64, "input_ids")
    dev = ids.device
    out = torch.empty((B, S, dim), device=dev, dtype=torch.bfloat16)
    out_labels = torch.empty((B, S), device=dev, dtype=torch.int64)
    out_mask = torch.empty((B, S), device=dev, dtype=torch.uint8)
    img_pos = torch.empty(B, device=dev, dtype=torch.int32)
    m8 = None if mask is None else mask.to(torch.uint8).contiguous()
    st = _L().lhrs_splice_fwd(ids.contiguous().data_ptr(), _p(None if labels is None else labels.contiguous()), _p(m8),
                              image.data_ptr(), embed.data_ptr(), out.data_ptr(), out_labels.data_ptr(), out_mask.data_ptr(),
                              img_pos.data_ptr(), B, T, NI, dim, S, embed.shape[0], _stream())
    _lib.check(st, "splice_fwd")
    return out, out_labels, out_mask, img_pos


def splice_bwd(d_embeds, img_pos, NI):
    B, S, dim = d_embeds.shape
    d_image = torch.empty((B, NI, dim), device=d_embeds.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_splice_bwd(d_embeds.data_ptr(), img_pos.data_ptr(), d_image.data_ptr(), B, NI, dim, S, _stream()),
               "splice_bwd")
    return d_image


def splice_map_fwd(ids, src_tok, src_img, image, embed, S):
    """General splice (several <image> placeholders per sample): ids int64 [B, T], src_tok / src_img int32 [B, S] from TextModal.splice_plan_host,
    image bf16 [n_slots, NI, dim] -> embeds bf16 [B, S, dim]."""
    B, T = ids.shape
    dim = image.shape[-1]
    _req(ids, torch.int64, "input_ids")
    _req(src_tok, torch.int32, "src_tok")
    _req(src_img, torch.int32, "src_img")
    out = torch.empty((B, S, dim), device=ids.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_splice_map_fwd(ids.contiguous().data_ptr(), src_tok.contiguous().data_ptr(), src_img.contiguous().data_ptr(), image.data_ptr(),
                                        embed.data_ptr(), out.data_ptr(), B, T, dim, S, embed.shape[0], _stream()), "splice_map_fwd")
    return out


def splice_map_bwd(d_embeds, inv, n_slots, NI):
    """d_image [n_slots, NI, dim] = rows inv[r] of d_embeds [B, S, dim] (inv int32 [n_slots * NI], < 0 -> zeros)."""
    dim = d_embeds.shape[-1]
    _req(inv, torch.int32, "inv")
    d_image = torch.empty((n_slots, NI, dim), device=d_embeds.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_splice_map_bwd(d_embeds.data_ptr(), inv.data_ptr(), d_image.data_ptr(), n_slots * NI, dim, _stream()), "splice_map_bwd")
    return d_image


def gather_rows(src, idx, out=None):
    n, dim = idx.numel(), src.shape[1]
    out = torch.empty((n, dim), device=src.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_gather_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), n, dim,
                                     _stream()), "gather_rows")
    return out


def scatter_rows(src, idx, dst):
    n, dim = idx.numel(), src.shape[1]
    _lib.check(_L().lhrs_scatter_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), dst.data_ptr(), dst.stride(0), n, dim,
                                      _stream()), "scatter_rows")
    return dst


def argmax_rows(logits_f32, out=None):
    n, V = logits_f32.shape
    out = torch.empty(n, device=logits_f32.device, dtype=torch.int64) if out is None else out
    _lib.check(_L().lhrs_argmax_rows(logits_f32.data_ptr(), logits_f32.stride(0), out.data_ptr(), n, V, _stream()), "argmax_rows")
    return out


def cross_entropy(logits, target, want_grad=True, inplace=True):
    n, V = logits.shape
    row_loss = torch.empty(n, device=logits.device, dtype=torch.float32)
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    dl = None
    if want_grad:
        dl = logits if inplace else torch.empty_like(logits)
    st = _L().lhrs_cross_entropy(logits.data_ptr(), logits.stride(0), target.data_ptr(), row_loss.data_ptr(), loss.data_ptr(),
                                 _p(dl), dl.stride(0) if dl is not None else 0, n, V, _stream())
    _lib.check(st, "cross_entropy")
    return loss, dl


# --------------------------------------------------------------------------------------------- decode
def gemv(W, x, out, residual=None, out_f32=False):
    """out[B, N] = x[B, K] @ W[N, K]^T (+ residual): weight-streaming kernels of the decode step (batch 1: VALU dot products, batch 2..16: MFMA with the batch as N)."""
    B, K = x.shape
    N = W.shape[0]
    st = _L().lhrs_gemv_bf16(W.data_ptr(), W.stride(0), x.data_ptr(), x.stride(0), _p(residual),
                             residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), B, N, K, int(out_f32),
                             _stream())
    _lib.check(st, "gemv_bf16")
    return out


PRO_NONE, PRO_RMSNORM, PRO_SWIGLU = 0, 1, 2


class PackedBf16:
    """bf16 weight [N, K] re-tiled by `repack_bf16_mfma` into the operand order of the batched MFMA GEMV (batch >= 2 decode)."""

    def __init__(self, data, N, K):
        self.data, self.N, self.K = data, N, K
        self.shape = (N, K)


def repack_bf16_mfma(W) -> PackedBf16:
    N, K = W.shape
    out = torch.empty((N + 15) // 16 * 16 * K, device=W.device, dtype=torch.bfloat16)
    _lib.check(_L().lhrs_repack_bf16_mfma(W.data_ptr(), W.stride(0), out.data_ptr(), N, K, _stream()), "repack_bf16_mfma")
    return PackedBf16(out, N, K)


def gemv_fused(W, x, out, K, *, wscale=None, prologue=PRO_NONE, norm_w=None, eps=1e-5, residual=None, out_f32=False):
    """out[B, N] = pro(x)[B, K] @ W[N, K]^T (+ residual).  W: bf16 [N, K], (wscale given) e4m3 bytes [N, K] with per-row scales, or a
    PackedBf16 (batch >= 2)."""
    B, N = x.shape[0], W.shape[0]
    if isinstance(W, PackedBf16):
        wp, ldw, fmt = W.data.data_ptr(), 0, 2
    else:
        wp, ldw, fmt = W.data_ptr(), W.stride(0), int(wscale is not None)
    st = _L().lhrs_gemv(wp, ldw, _p(wscale), fmt, x.data_ptr(), x.stride(0), prologue, _p(norm_w),
                        float(eps), _p(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), B, N, K,
                        int(out_f32), _stream())
    _lib.check(st, "gemv")
    return out


class PackedFp8:
    """e4m3 weight [N, K] re-tiled by `repack_fp8_mfma` into the operand order of the MFMA GEMV (decode weight stream)."""

    def __init__(self, data, N, K):
        self.data, self.N, self.K = data, N, K
        self.shape = (N, K)


def repack_fp8_mfma(W8) -> PackedFp8:
    N, K = W8.shape
    out = torch.empty((N + 15) // 16 * 16 * K, device=W8.device, dtype=torch.uint8)
    _lib.check(_L().lhrs_repack_fp8_mfma(W8.data_ptr(), W8.stride(0), out.data_ptr(), N, K, _stream()), "repack_fp8_mfma")
    return PackedFp8(out, N, K)


def _w8(W8):  # -> (ptr, ldw, N, packed)
    if isinstance(W8, PackedFp8):
        return W8.data.data_ptr(), 0, W8.N, 1
    return W8.data_ptr(), W8.stride(0), W8.shape[0], 0


def gemv_fp8_mfma(W8, wscale, x8, xscale, out, *, residual=None, out_f32=False):
    """out[B, N] = xscale[:, None] * wscale[None, :] * (x8 @ W8^T) (+ residual): e4m3 weights and activations on the scaled MFMA.
    W8: uint8 [N, K] rows or a PackedFp8."""
    B, K = x8.shape
    wp, ldw, N, pk = _w8(W8)
    st = _L().lhrs_gemv_fp8_mfma(wp, ldw, wscale.data_ptr(), x8.data_ptr(), x8.stride(0), xscale.data_ptr(),
                                 _p(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), B, N, K,
                                 int(out_f32), pk, _stream())
    _lib.check(st, "gemv_fp8_mfma")
    return out


def gemv_fp8_mfma_fused(W8, wscale, x, out, K, *, prologue=PRO_NONE, norm_w=None, eps=1e-5, residual=None, out_f32=False):
    """batch <= 2: out = (e4m3(pro(x)) @ W8^T) * scales (+ residual), prologue and activation quantisation inside the kernel."""
    B = x.shape[0]
    wp, ldw, N, pk = _w8(W8)
    st = _L().lhrs_gemv_fp8_mfma_fused(wp, ldw, wscale.data_ptr(), x.data_ptr(), x.stride(0), prologue, _p(norm_w),
                                       float(eps), _p(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(),
                                       out.stride(0), B, N, K, int(out_f32), pk, _stream())
    _lib.check(st, "gemv_fp8_mfma_fused")
    return out


def quant_fp8_rows(W, out=None):
    """bf16 [N, K] -> (uint8 e4m3 [N, K], fp32 per-row scale [N]); out = preallocated (W8, scale) for graph-captured callers."""
    N, K = W.shape
    W8, sc = out if out is not None else (torch.empty((N, K), device=W.device, dtype=torch.uint8),
                                          torch.empty(N, device=W.device, dtype=torch.float32))
    _lib.check(_L().lhrs_quant_fp8_rows(W.data_ptr(), W.stride(0), W8.data_ptr(), W8.stride(0), sc.data_ptr(), N, K, _stream()), "quant_fp8_rows")
    return W8, sc


def rope_kv_append(qkv, kc, vc, cos_t, sin_t, pos, B, H, D, max_ctx):
    _lib.check(_L().lhrs_rope_kv_append(qkv.data_ptr(), qkv.stride(0), kc.data_ptr(), vc.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(),
                                        pos.data_ptr(), B, H, D, max_ctx, _stream()), "rope_kv_append")


def decode_attn(qkv, kc, vc, cos_t, sin_t, pos, out, B, H, D, max_ctx, scale, key_mask=None):
    """RoPE(new q, k) + KV append + attention over the cache for one new token per sequence, one launch (lhrs_decode_attn)."""
    st = _L().lhrs_decode_attn(qkv.data_ptr(), qkv.stride(0), kc.data_ptr(), vc.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), pos.data_ptr(),
                               _p(key_mask), key_mask.stride(0) if key_mask is not None else 0, out.data_ptr(), out.stride(0), B, H, D,
                               max_ctx, float(scale), _stream())
    _lib.check(st, "decode_attn")
    return out


def decode_attn_split(qkv, kc, vc, cos_t, sin_t, pos, out, B, H, D, max_ctx, scale, nsplit, part, tickets, key_mask=None, cs=None):
    """decode_attn with the context split over `nsplit` workgroups per head (lhrs_decode_attn_split); part fp32 [B, H, nsplit, 132],
    tickets int32 [B, H] (zero before the first call)."""
    assert part.dtype == torch.float32 and part.numel() >= B * H * nsplit * 132 and tickets.dtype == torch.int32 and tickets.numel() >= B * H
    st = _L().lhrs_decode_attn_split(qkv.data_ptr(), qkv.stride(0), kc.data_ptr(), vc.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(),
                                     pos.data_ptr(), _p(key_mask), key_mask.stride(0) if key_mask is not None else 0, out.data_ptr(),
                                     out.stride(0), B, H, D, max_ctx, float(scale), int(nsplit), part.data_ptr(), tickets.data_ptr(), _p(cs), _stream())
    _lib.check(st, "decode_attn_split")
    return out


def decode_advance(state, desc, pos, B, max_ctx, inc=1, cos_t=None, sin_t=None, cs=None):
    """cs (float32 [B, 128]) with the tables: also leave the cos | sin rows of the new position there (lhrs_decode_advance_cs)"""
    if cs is not None:
        _lib.check(_L().lhrs_decode_advance_cs(state.data_ptr(), desc.data_ptr(), pos.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), cs.data_ptr(),
                                               B, max_ctx, inc, _stream()), "decode_advance_cs")
        return
    _lib.check(_L().lhrs_decode_advance(state.data_ptr(), desc.data_ptr(), pos.data_ptr(), B, max_ctx, inc, _stream()), "decode_advance")


def kv_append(qkv, kc, vc, pos, B, d, max_ctx):
    _lib.check(_L().lhrs_kv_append(qkv.data_ptr(), qkv.stride(0), kc.data_ptr(), vc.data_ptr(), pos.data_ptr(), B, d, max_ctx, _stream()),
               "kv_append")


def decode_emit(next_ids, tok32, out_ids, state, B, max_new):
    _lib.check(_L().lhrs_decode_emit(next_ids.data_ptr(), tok32.data_ptr(), out_ids.data_ptr(), state.data_ptr(), B, max_new, _stream()),
               "decode_emit")


class HipGraph:
    """hipGraph capture / replay of the launches enqueued on the current (non-default) HIP stream."""

    def __init__(self):
        self.exec = None

    def begin(self):
        _lib.check(_L().lhrs_graph_begin(_stream()), "graph_begin")

    def end(self):
        import ctypes
        h = ctypes.c_void_p()
        _lib.check(_L().lhrs_graph_end(_stream(), ctypes.byref(h)), "graph_end")
        self.exec = h.value

    def launch(self):
        _lib.check(_L().lhrs_graph_launch(self.exec, _stream()), "graph_launch")

    def __del__(self):
        try:
            if self.exec:
                _L().lhrs_graph_destroy(self.exec)
        except Exception:
            pass


# --------------------------------------------------------------------------------------------- optimizer
def sqnorm(g, out, accumulate=False):
    nb = _L().lhrs_sqnorm_nblk(g.numel())
    part = torch.empty(nb, device=g.device, dtype=torch.float32)
    _lib.check(_L().lhrs_sqnorm(g.data_ptr(), g.numel(), part.data_ptr(), out.data_ptr(), int(accumulate), _stream()), "sqnorm")
    return out


def accum_f32(y, x, copy_only=False):
    """y = x (copy_only) or y += x on flat fp32 buffers (gradient accumulation)."""
    assert y.dtype == torch.float32 and x.dtype == torch.float32 and y.numel() == x.numel()
    _lib.check(_L().lhrs_accum_f32(y.data_ptr(), x.data_ptr(), y.numel(), int(copy_only), _stream()), "accum_f32")
    return y


def adan_step(p, g, m, v, n, pre, shadow, step, lr, betas=(0.98, 0.92, 0.99), eps=1e-8, wd=0.0, no_prox=True,
              gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    st = _L().lhrs_adan_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n.data_ptr(), pre.data_ptr(), _p(shadow),
                             p.numel(), step, lr, betas[0], betas[1], betas[2], eps, wd, int(no_prox), _p(gnorm_sq), max_norm,
                             grad_scale, _stream())
    _lib.check(st, "adan_step")


def adamw_step(p, g, m, v, shadow, step, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.0, gnorm_sq=None, max_norm=0.0, grad_scale=1.0):
    st = _L().lhrs_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(shadow), p.numel(), step, lr, betas[0],
                              betas[1], eps, wd, _p(gnorm_sq), max_norm, grad_scale, _stream())
    _lib.check(st, "adamw_step")


# --------------------------------------------------------------------------------------------- data boundary (images)
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def image_preprocess(img_u8: torch.Tensor, out: torch.Tensor = None, short_edge: int = 224, crop_round: int = 0, rescale_mode: int = 0,
                     mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """uint8 [H, W, 3] device tensor -> float32 [3, 224, 224]: Pillow-BICUBIC resize of the short edge to `short_edge`, centre crop,
    byte -> float, (x - mean) / std (lhrs_image_preprocess; the defaults are CLIPImageProcessor, bit-exact)."""
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_cuda and img_u8.stride(2) == 1 \
        and img_u8.stride(1) == 3, "image_preprocess: uint8 [H, W, 3] device tensor with packed pixels"
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    if out is None:
        out = torch.empty((3, 224, 224), device=img_u8.device, dtype=torch.float32)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == 3 * 224 * 224
    nbytes = _L().lhrs_image_preprocess_workspace(H, W, short_edge)
    if nbytes < 0:
        raise ValueError(f"image_preprocess: H={H} W={W} short_edge={short_edge} (short_edge must be >= 224)")
    ws = torch.empty(nbytes, device=img_u8.device, dtype=torch.uint8)
    m, sd = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    st = _L().lhrs_image_preprocess(img_u8.data_ptr(), H, W, img_u8.stride(0), out.data_ptr(), ws.data_ptr(), nbytes, short_edge, crop_round,
                                    rescale_mode, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), _stream())
    _lib.check(st, "image_preprocess")
    return out


def clip_preprocess(img_u8: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """uint8 [H, W, 3] device tensor -> float32 [3, 224, 224] pixel_values, bit-exact with CLIPImageProcessor."""
    return image_preprocess(img_u8, out)


# ------------------------------------------------------------------------------------------------ LLM.int8 base (csrc/int8.hip)
def quant_int8_rows(W):
    """bf16 [N, K] -> (int8 [N, K], fp32 [N] dequantisation factor absmax / 127): bitsandbytes' vector-wise weight quantisation."""
    N, K = W.shape
    Q = torch.empty((N, K), device=W.device, dtype=torch.int8)
    sc = torch.empty(N, device=W.device, dtype=torch.float32)
    _lib.check(_L().lhrs_quant_int8_rows(W.data_ptr(), W.stride(0), Q.data_ptr(), Q.stride(0), sc.data_ptr(), N, K, _stream()), "quant_int8_rows")
    return Q, sc


def dequant_int8_rows(Q, sc, out=None):
    N, K = Q.shape
    out = torch.empty((N, K), device=Q.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(_L().lhrs_dequant_int8_rows(Q.data_ptr(), Q.stride(0), sc.data_ptr(), out.data_ptr(), out.stride(0), N, K, _stream()), "dequant_int8_rows")
    return out


# --------------------------------------------------------------------------------------------- 4-bit base storage (bits: 4)
def dynamic_map_8bit() -> torch.Tensor:
    """The sorted 256-entry table of bitsandbytes' signed "dynamic" 8-bit data type (functional.create_dynamic_map(signed=True,
    max_exponent_bits=7, total_bits=8)) that `double_quant` stores the absmax statistics in: for i = 0..6 the 2^i midpoints of
    linspace(0.1, 1, 2^i + 1) scaled by 10^(i - 6), both signs, plus 0 and 1.  fp32 CPU tensor."""
    data = []
    for i in range(7):
        b = torch.linspace(0.1, 1, 2 ** i + 1)
        means = (b[:-1] + b[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data += [0.0, 1.0]
    assert len(data) == 256
    return torch.tensor(sorted(data), dtype=torch.float32)


_DYN_MAP = {}


def _dyn_map(device) -> torch.Tensor:
    key = str(device)
    if key not in _DYN_MAP:
        _DYN_MAP[key] = dynamic_map_8bit().to(device)
    return _DYN_MAP[key]


def quant4_blocks(W, quant_type: str = "nf4", double_quant: bool = True):
    """bitsandbytes quantize_4bit of ONE weight (a contiguous bf16 [N, K] tensor or row range, N*K % 64 == 0) -> state dict:
    packed uint8 [N*K/2], and either absmax fp32 [N*K/64] or (double_quant) its 8-bit form: qabsmax uint8, absmax2 fp32, offset."""
    if quant_type not in ("nf4", "fp4"):
        raise ValueError(f"quant_type {quant_type!r}: 'nf4' or 'fp4' (bitsandbytes bnb_4bit_quant_type)")
    _req(W, torch.bfloat16, "quant4_blocks weight")
    if not W.is_contiguous():
        raise ValueError("quant4_blocks: the weight (row range) must be contiguous")
    n = W.numel()
    packed = torch.empty(n // 2, device=W.device, dtype=torch.uint8)
    absmax = torch.empty(n // 64, device=W.device, dtype=torch.float32)
    _lib.check(_L().lhrs_quant4_blocks(W.data_ptr(), n, int(quant_type == "fp4"), packed.data_ptr(), absmax.data_ptr(), _stream()), "quant4_blocks")
    st = dict(packed=packed, quant_type=quant_type, shape=tuple(W.shape))
    if not double_quant:
        st["absmax"] = absmax
        return st
    nb = absmax.numel()
    offset = float(absmax.double().mean().float())   # the package: absmax.mean() in fp32; the fp64 sum makes the value independent of the reduction order
    rest = absmax - offset
    q = torch.empty(nb, device=W.device, dtype=torch.uint8)
    absmax2 = torch.empty((nb + 255) // 256, device=W.device, dtype=torch.float32)
    _lib.check(_L().lhrs_quant8_dynamic(rest.data_ptr(), nb, _dyn_map(W.device).data_ptr(), q.data_ptr(), absmax2.data_ptr(), _stream()), "quant8_dynamic")
    st.update(qabsmax=q, absmax2=absmax2, offset=offset)
    return st


def absmax_of(st) -> torch.Tensor:
    """The fp32 block absmax a 4-bit state dequantises with (double_quant: code[q] * absmax2 + offset)."""
    if "absmax" in st:
        return st["absmax"]
    q = st["qabsmax"]
    out = torch.empty(q.numel(), device=q.device, dtype=torch.float32)
    _lib.check(_L().lhrs_dequant8_dynamic(q.data_ptr(), st["absmax2"].data_ptr(), q.numel(), _dyn_map(q.device).data_ptr(), float(st["offset"]),
                                          out.data_ptr(), _stream()), "dequant8_dynamic")
    return out


def dequant4_blocks(st, out=None):
    """bitsandbytes dequantize_4bit: table[code] * absmax -> bf16, into `out` (contiguous, same element count) or a new tensor of the stored shape."""
    packed = st["packed"]
    n = packed.numel() * 2
    out = torch.empty(st["shape"], device=packed.device, dtype=torch.bfloat16) if out is None else out
    if out.numel() != n or not out.is_contiguous() or out.dtype != torch.bfloat16:
        raise ValueError("dequant4_blocks: `out` must be a contiguous bf16 tensor with the element count of the state")
    a = absmax_of(st)
    _lib.check(_L().lhrs_dequant4_blocks(packed.data_ptr(), a.data_ptr(), n, int(st["quant_type"] == "fp4"), out.data_ptr(), _stream()), "dequant4_blocks")
    return out


class Int8Workspace:
    """Device scratch of the LLM.int8 activation side: outlier flags per input feature, the compacted outlier column list, meta = [columns
    found by the last call, that count rounded up to 64] and the persistent [N, KP + K] buffers of the dequantised outlier weight columns."""

    def __init__(self, device, kmax: int = 32768, threshold: float = 6.0):
        self.threshold, self.kmax = float(threshold), (kmax + 63) // 64 * 64
        self.flags = torch.zeros(self.kmax, device=device, dtype=torch.int32)
        self.idx = torch.full((self.kmax,), -1, device=device, dtype=torch.int32)
        self.meta = torch.zeros(2, device=device, dtype=torch.int32)
        self._b2 = {}

    def b2(self, N: int, cols: int):
        key = (N, cols)
        if key not in self._b2:
            self._b2[key] = torch.empty((N, cols), device=self.flags.device, dtype=torch.bfloat16)
        return self._b2[key]

    def last_outlier_columns(self):
        """Host read (synchronises): the outlier feature columns of the most recent product, ascending."""
        n = int(self.meta[0].item())
        return self.idx[:n].tolist()


def int8_linear(x, WQ, wscale, ws: Int8Workspace, *, residual=None, a2=None, b2=None, alpha: float = 1.0, out=None):
    """bitsandbytes MatMul8bitLt forward: y = (int8(x') . WQ^T) * sx * wscale + x[:, O] . dequant(WQ)[:, O]^T (+ a2 . b2^T) (+ residual), O = the
    columns of x holding a value >= ws.threshold (however many there are), x' = x with those columns zeroed.  x bf16 [M, K]; WQ int8 [N, K] +
    wscale [N] from quant_int8_rows; (a2 [M, KP], b2 [N, KP]) an optional bf16 pair (LoRA update) riding on the same accumulators."""
    M, K = x.shape
    N = WQ.shape[0]
    KP = a2.shape[1] if a2 is not None else 0
    kpad = (K + 63) // 64 * 64
    assert kpad <= ws.kmax, (K, ws.kmax)
    xq = torch.empty((M, K), device=x.device, dtype=torch.int8)
    sx = torch.empty(M, device=x.device, dtype=torch.float32)
    A2 = torch.empty((M, KP + kpad), device=x.device, dtype=torch.bfloat16)
    B2 = ws.b2(N, KP + kpad)
    if KP:
        A2[:, :KP].copy_(a2)
        B2[:, :KP].copy_(b2)
    L = _L()
    _lib.check(L.lhrs_int8_prepare(x.data_ptr(), x.stride(0), M, K, ws.threshold, WQ.data_ptr(), WQ.stride(0), wscale.data_ptr(), N,
                                   xq.data_ptr(), xq.stride(0), sx.data_ptr(), ws.flags.data_ptr(), ws.idx.data_ptr(), ws.meta.data_ptr(),
                                   A2.data_ptr() + 2 * KP, A2.stride(0), B2.data_ptr() + 2 * KP, B2.stride(0), _stream()), "int8_prepare")
    out = torch.empty((M, N), device=x.device, dtype=torch.bfloat16) if out is None else out
    _lib.check(L.lhrs_gemm_int8_nt(xq.data_ptr(), xq.stride(0), sx.data_ptr(), WQ.data_ptr(), WQ.stride(0), wscale.data_ptr(), A2.data_ptr(),
                                   A2.stride(0), B2.data_ptr(), B2.stride(0), KP, ws.meta.data_ptr() + 4, out.data_ptr(), out.stride(0), M, N, K,
                                   _p(residual), residual.stride(0) if residual is not None else 0, float(alpha), _stream()), "gemm_int8_nt")
    return out
