"""TEST INFRASTRUCTURE ONLY - CPU restatement of the optimizer update rules on the LHRS-Bot hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

PARITY UNPINNED: timm==0.9.12 (Adan) and deepspeed (global-norm clipping, FusedAdam) are third-party
dependencies pinned in /root/reference/pyproject.toml:14-43 that are NOT importable in the build container and
are not vendored under /root/reference.  The functions below restate their published update rules:
  * Adan, no_prox=True ("adanp"): call site /root/reference/lhrs/optimizer/build_optimizer.py:76-86.
      m <- lerp(m, g, 1-b1); v <- lerp(v, g-g_prev, 1-b2); n <- b3*n + (1-b3)*(g + b2*(g-g_prev))^2
      theta <- theta*(1 - lr*wd) - lr * (m/bc1 + b2*v/bc2) / (sqrt(n)/sqrt(bc3) + eps);  g_prev(step 1) = g
  * global-norm clipping: call site /root/reference/main_pretrain_stage1.py:28-85 (`gradient_clipping`),
      clip_coef = max_norm / (||g|| + 1e-6), applied when < 1.
  * AdamW (decoupled weight decay): cross-checked against torch.optim.AdamW in tests/test_oracle_cpu.py.
"""
import math

import torch


def clip_coef(grad: torch.Tensor, max_norm: float) -> float:
    if max_norm <= 0:
        return 1.0
    c = max_norm / (grad.double().norm().item() + 1e-6)
    return min(c, 1.0)


def adan_step_ref(state, grad, step, lr, betas=(0.98, 0.92, 0.99), eps=1e-8, wd=0.0, no_prox=True, max_norm=0.0):
    b1, b2, b3 = betas
    grad = grad * clip_coef(grad, max_norm)
    if state["pre"] is None:
        state["pre"] = grad.clone()
    diff = grad - state["pre"]
    state["m"] = state["m"] + (grad - state["m"]) * (1 - b1)
    state["v"] = state["v"] + (diff - state["v"]) * (1 - b2)
    upd = grad + b2 * diff
    state["n"] = state["n"] * b3 + upd * upd * (1 - b3)
    bc1, bc2, bc3s = 1 - b1 ** step, 1 - b2 ** step, math.sqrt(1 - b3 ** step)
    denom = state["n"].sqrt() / bc3s + eps
    update = (state["m"] / bc1 + b2 * state["v"] / bc2) / denom
    if no_prox:
        state["p"] = state["p"] * (1 - lr * wd) - lr * update
    else:
        state["p"] = (state["p"] - lr * update) / (1 + lr * wd)
    state["pre"] = grad.clone()
    return state


def adamw_step_ref(state, grad, step, lr, betas=(0.9, 0.95), eps=1e-8, wd=0.0, max_norm=0.0):
    b1, b2 = betas
    grad = grad * clip_coef(grad, max_norm)
    state["m"] = b1 * state["m"] + (1 - b1) * grad
    state["v"] = b2 * state["v"] + (1 - b2) * grad * grad
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = state["v"].sqrt() / math.sqrt(bc2) + eps
    state["p"] = state["p"] * (1 - lr * wd) - (lr / bc1) * (state["m"] / denom)
    return state
