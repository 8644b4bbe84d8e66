"""TEST INFRASTRUCTURE ONLY - restatement of the reference's `bits: 8` base weights: bitsandbytes LLM.int8().

PARITY UNPINNED: bitsandbytes==0.41.3 (pyproject.toml of the reference) is absent from this image and from /root/reference; the call
site is lhrs/models/text_modal.py:91-131 (`BitsAndBytesConfig(load_in_8bit=True, llm_int8_threshold=6.0, llm_int8_has_fp16_weight=False,
llm_int8_skip_modules=[...lm_head...])`).  What follows restates the published algorithm (Dettmers et al., "LLM.int8(): 8-bit Matrix
Multiplication for Transformers at Scale", 2022, §3 and `bitsandbytes.autograd._functions.MatMul8bitLt` of the 0.41 series):

forward, y = x W^T with W [out, in] frozen:
  * W is stored once as int8 with one absmax scale per OUTPUT ROW:   CB = round(127 * W / SCB[:, None]),  SCB = max_k |W[n, k]|
  * per call, the feature columns k of x [tokens, in] in which ANY |x[t, k]| >= threshold (6.0) are OUTLIERS:
        - outlier part in 16 bit:     y_o = x[:, O] . (CB[:, O] * SCB[:, None] / 127)^T        (the weight columns are DEquantised int8)
        - the rest in int8:           SCA[t] = max_k { |x[t, k]| : |x[t, k]| < threshold }  (bitsandbytes' row statistics skip outlier ENTRIES, not
                                      whole outlier columns);  CA = round(x * (127 / SCA[:, None])) with the outlier COLUMNS zeroed
                                      y_i = (CA . CB^T  in int32) * SCA[:, None] * SCB[None, :] / (127 * 127)
        - y = y_i + y_o   (fp16 in the reference; here the caller's dtype)
backward (frozen int8 weight, `has_fp16_weights=False`):  dx = dy . (CB * SCB[:, None] / 127)  - the DEQUANTISED weight, in 16 bit.

Only `lm_head` is skipped by the call site, so every decoder linear (q, k, v, o, gate, up, down) goes through this; LoRA adapters (peft)
sit on top in 16 bit.  The e4m3 scheme of the engine (`TextModal.quantize_base`) is a DELIBERATE DEVIATION: per-row e4m3 for weights AND
activations / gradients on the block-scaled MFMA, no outlier decomposition; tests/test_fp8_gpu.py measures both against the fp32 oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

THRESHOLD = 6.0


def _codes(x: torch.Tensor, absmax: torch.Tensor) -> torch.Tensor:
    """rint(x * (127 / absmax)) per row in IEEE fp32 (numpy: correctly rounded division and product, ties to even) - torch's vectorised CPU
    division is not correctly rounded (127 / 0.0810546875 comes out one ulp low), which flips exact .5 ties that bf16 data produces often."""
    import numpy as np
    inv = np.float32(127.0) / absmax.detach().numpy().astype(np.float32)
    q = np.rint(x.detach().numpy().astype(np.float32) * inv[:, None])
    return torch.from_numpy(np.clip(q, -127, 127).astype(np.float32))


def quantize_rows_int8(w: torch.Tensor):
    """-> (CB int8 [out, in], SCB fp32 [out]): vector-wise absmax quantisation of the weight rows (round half to even, like torch.round)."""
    scb = w.abs().amax(dim=1).clamp_min(1e-30)
    cb = _codes(w, scb).to(torch.int8)                                                  # x * (127 / absmax): bitsandbytes multiplies by the reciprocal
    return cb, scb


def dequantize_rows_int8(cb: torch.Tensor, scb: torch.Tensor) -> torch.Tensor:
    return cb.float() * (scb[:, None] / 127.0)


class _MatMul8bitLt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cb, scb, threshold):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).float()
        outlier = (x2.abs() >= threshold).any(dim=0)                      # feature dimensions with at least one outlier in this call
        wd = dequantize_rows_int8(cb, scb)
        y = x2.new_zeros((x2.shape[0], cb.shape[0]))
        if bool(outlier.any()):
            y = y + x2[:, outlier] @ wd[:, outlier].t()                   # 16-bit path of the reference (fp32 here: the oracle's dtype)
        sca = x2.abs().masked_fill(x2.abs() >= threshold, 0.0).amax(dim=1).clamp_min(1e-30)
        ca = _codes(x2, sca).masked_fill(outlier[None, :], 0.0)
        acc = ca.double() @ cb.double().t()                               # exact int32 accumulation (|sum| < 2^31 for in <= 2^17)
        y = y + (acc * (sca[:, None].double() * scb[None, :].double() / (127.0 * 127.0))).float()
        ctx.save_for_backward(cb, scb)
        ctx.shape = shape
        return y.reshape(*shape[:-1], cb.shape[0]).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        cb, scb = ctx.saved_tensors
        dx = dy.reshape(-1, dy.shape[-1]).float() @ dequantize_rows_int8(cb, scb)
        return dx.reshape(ctx.shape).to(dy.dtype), None, None, None


class Int8Weight:
    """A frozen weight in LLM.int8 storage; `F.linear(x, Int8Weight)` is spelled `linear(x, w)` below."""

    def __init__(self, w: torch.Tensor):
        self.cb, self.scb = quantize_rows_int8(w.detach().float())
        self.shape = tuple(w.shape)


def linear(x: torch.Tensor, w, threshold: float = THRESHOLD) -> torch.Tensor:
    """Drop-in for F.linear(x, w) inside the oracle: int8 path for `Int8Weight`, plain product otherwise."""
    if isinstance(w, Int8Weight):
        return _MatMul8bitLt.apply(x, w.cb, w.scb, threshold)
    return F.linear(x, w)


def int8_llama_params(p: dict) -> dict:
    """The oracle's LLaMA parameter dict with every decoder linear replaced by its LLM.int8 storage (lm_head, norms, embeddings keep
    their dtype - `llm_int8_skip_modules` of the call site)."""
    out = {k: v for k, v in p.items() if k != "layers"}
    out["layers"] = [{k: (Int8Weight(v) if k in ("qkv_w", "o_w", "gu_w", "down_w") else v) for k, v in L.items()} for L in p["layers"]]
    out["_linear"] = linear
    return out
