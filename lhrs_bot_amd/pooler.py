"""AttnPooler (the stage-1 trainable projector): forward + hand-written backward on gfx950.

Mirrors /root/reference lhrs/models/common_arch.py: `AttnPooler` (:79-173) and `ResidualAttentionBlock` (:262-333).
The reference loops over the three query groups and runs the six shared layers on each; here the three groups are
packed per sample ([64|48|32] queries, [320|304|288] keys) so that every projection is ONE GEMM over all groups
and the cross-attention is ONE varlen launch (3*B sequences) - 18 sequential block applications become 6.

Parameters live in three flat buffers (fp32 master, bf16 shadow used by the kernels, fp32 gradient) so that the
optimizer and the data-parallel all-reduce each touch one contiguous range.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import kernels as hk

STAGE_NUM = (64, 48, 32)       # common_arch.py:104
SPLIT_PART = (256, 256, 256)   # common_arch.py:105


import os as _os

def _dw_tn_enabled() -> bool:
    """LHRS_DW_TRANSPOSED=1 forces the round-1 path (transposed copies + NT split-K GEMM); read at call time so tests can flip it."""
    return _os.environ.get("LHRS_DW_TRANSPOSED", "0") != "1"


class AttnPooler:
    def __init__(self, num_query=144, num_layers=6, num_attention_heads=16, encoder_hidden_size=1024, hidden_size=1024,
                 output_size=4096, device="cuda", **_unused):
        assert encoder_hidden_size == hidden_size, "in_proj is None in every shipped config (common_arch.py:112-115)"
        assert num_query == sum(STAGE_NUM)
        self.device = torch.device(device)
        hk.ensure_gemm_workspace(self.device)
        self.nq, self.nl, self.heads, self.d, self.out_dim = num_query, num_layers, num_attention_heads, hidden_size, output_size
        d, od = hidden_size, output_size
        # (name, shape, no_decay) in a fixed order; names follow the reference state_dict (checkpoint row f-1)
        spec: List[Tuple[str, Tuple[int, ...]]] = [("query", (num_query, d))]
        for l in range(num_layers):
            b = f"layers.{l}."
            spec += [(b + "ln_1.weight", (d,)), (b + "ln_1.bias", (d,)), (b + "ln_1_kv.weight", (d,)), (b + "ln_1_kv.bias", (d,)),
                     (b + "attn.in_proj_weight", (3 * d, d)), (b + "attn.in_proj_bias", (3 * d,)),
                     (b + "attn.out_proj.weight", (d, d)), (b + "attn.out_proj.bias", (d,)),
                     (b + "ln_2.weight", (d,)), (b + "ln_2.bias", (d,)),
                     (b + "mlp.c_fc.weight", (4 * d, d)), (b + "mlp.c_fc.bias", (4 * d,)),
                     (b + "mlp.c_proj.weight", (d, 4 * d)), (b + "mlp.c_proj.bias", (d,))]
        spec += [("out_proj.weight", (od, d)), ("out_proj.bias", (od,))]
        self.spec = spec
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in spec:
            n = 1
            for s in shape:
                n *= s
            self.offsets[name] = (off, shape)
            off += (n + 63) // 64 * 64  # keep every tensor 256-B aligned inside the flat buffers
        self.numel = off
        self.master = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
        self.grad = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.w = {n: self._view(self.shadow, n) for n, _ in spec}     # bf16 views for the kernels
        self.g = {n: self._view(self.grad, n) for n, _ in spec}       # fp32 gradient views
        self.wT: Dict[str, torch.Tensor] = {}                          # transposed bf16 weights for dX GEMMs
        self.requires_grad = True
        self.initialised = False  # set by init_random / load_params / load_state_dict
        self._desc_cache: Dict[int, torch.Tensor] = {}
        self._ctx = None

    def _view(self, flat, name):
        off, shape = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off: off + n].view(*shape)

    def named_parameters(self):
        return [(n, self._view(self.master, n)) for n, _ in self.spec]

    def num_parameters(self) -> int:
        return sum(self._view(self.master, n).numel() for n, _ in self.spec)

    # ------------------------------------------------------------------ parameters
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """sd uses the reference's AttnPooler keys ('query' may be [1,144,d] as in the reference)."""
        missing = [n for n, _ in self.spec if n not in sd]
        if strict and missing:
            raise KeyError(f"missing pooler keys: {missing[:4]}...")
        for n, _ in self.spec:
            if n in sd:
                self._view(self.master, n).copy_(sd[n].reshape(self.offsets[n][1]).to(self.device, torch.float32))
        self.sync_shadow()
        return missing

    def load_params(self, p: Dict) -> None:
        """Engine/oracle layout (oracle/params.py) -> reference keys."""
        sd = {"query": p["query"], "out_proj.weight": p["out_proj_w"], "out_proj.bias": p["out_proj_b"]}
        m = {"ln1_w": "ln_1.weight", "ln1_b": "ln_1.bias", "ln1kv_w": "ln_1_kv.weight", "ln1kv_b": "ln_1_kv.bias",
             "in_w": "attn.in_proj_weight", "in_b": "attn.in_proj_bias", "out_w": "attn.out_proj.weight",
             "out_b": "attn.out_proj.bias", "ln2_w": "ln_2.weight", "ln2_b": "ln_2.bias", "fc_w": "mlp.c_fc.weight",
             "fc_b": "mlp.c_fc.bias", "proj_w": "mlp.c_proj.weight", "proj_b": "mlp.c_proj.bias"}
        for l, L in enumerate(p["layers"]):
            for k, v in L.items():
                sd[f"layers.{l}.{m[k]}"] = v
        self.load_state_dict(sd)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {n: self._view(self.master, n).detach().cpu().clone() for n, _ in self.spec}
        sd["query"] = sd["query"][None]
        return sd

    def init_random(self, seed: int = 0) -> None:
        """Default init of the reference module: trunc_normal(0.02) queries, nn.Linear / MHA defaults, LN = (1, 0)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for n, shape in self.spec:
            v = self._view(self.master, n)
            if n == "query":
                v.copy_(torch.randn(shape, device=self.device, generator=g).clamp_(-2, 2) * 0.02)
            elif n.endswith("weight") and len(shape) == 1:
                v.fill_(1.0)
            elif n.endswith("bias"):
                v.zero_()
            else:
                bound = (6.0 / (shape[0] + shape[1])) ** 0.5 if "in_proj" in n else (1.0 / shape[1]) ** 0.5
                v.copy_((torch.rand(shape, device=self.device, generator=g) * 2 - 1) * bound)
        self.sync_shadow()
        self.initialised = True

    def sync_shadow(self) -> None:
        """bf16 shadow + transposed copies after an optimizer step / load (HBM-bound, ~0.5 GB of traffic)."""
        self.initialised = True
        hk.cast_f32_to_bf16(self.master, self.shadow)
        self.refresh_transposed()

    def refresh_transposed(self) -> None:
        """Transposed bf16 copies of the 31 weight matrices (dX = dY . W as an NT GEMM), rebuilt after every optimizer step: the first
        call allocates them, every later one is ONE batched launch (lhrs_transpose_batched) over the fixed (weight view, copy) pairs."""
        if getattr(self, "_bt", None) is not None:
            self._bt.run()
            return
        d = self.d
        pairs = []
        for l in range(self.nl):
            b = f"layers.{l}."
            W = self.w[b + "attn.in_proj_weight"]
            for key, src in ((b + "q", W[:d]), (b + "kv", W[d:])):                    # [d, d], [d, 2d]
                self.wT[key] = hk.transpose(src, out=self.wT.get(key))
                pairs.append((src, self.wT[key]))
            for key in ("attn.out_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"):
                self.wT[b + key] = hk.transpose(self.w[b + key], out=self.wT.get(b + key))
                pairs.append((self.w[b + key], self.wT[b + key]))
        self.wT["out_proj.weight"] = hk.transpose(self.w["out_proj.weight"], out=self.wT.get("out_proj.weight"))
        pairs.append((self.w["out_proj.weight"], self.wT["out_proj.weight"]))
        self._bt = hk.BatchedTranspose(pairs)   # self.w are views of the flat bf16 shadow: same storage for the life of the projector

    def _desc(self, B: int) -> torch.Tensor:
        if B not in self._desc_cache:
            NQ, KV = self.nq, self.nq + sum(SPLIT_PART)
            ent = []
            for b in range(B):
                qo, ko = 0, 0
                for nq, ni in zip(STAGE_NUM, SPLIT_PART):
                    ent.append((b * NQ + qo, nq, b * KV + ko, nq + ni))
                    qo += nq
                    ko += nq + ni
            self._desc_cache[B] = hk.make_desc(ent, self.device)
        return self._desc_cache[B]

    # ------------------------------------------------------------------ forward
    def forward(self, image_embs: torch.Tensor, save_ctx: bool = True) -> torch.Tensor:
        """image_embs [B, 768, 1024] bf16 -> [B, 144, 4096] bf16."""
        B, d, H = image_embs.shape[0], self.d, self.heads
        w = self.w
        desc, nseq = self._desc(B), 3 * B
        LTq, LTkv = hk.pad64(max(STAGE_NUM)), hk.pad64(max(q + i for q, i in zip(STAGE_NUM, SPLIT_PART)))
        t, kv = hk.pooler_build(w["query"], image_embs.contiguous(), B, STAGE_NUM, SPLIT_PART)
        M = t.shape[0]
        scale = (d // H) ** -0.5
        layers = []
        for l in range(self.nl):
            b = f"layers.{l}."
            Win, bin_ = w[b + "attn.in_proj_weight"], w[b + "attn.in_proj_bias"]
            kvn, kv_mean, kv_rstd = hk.layernorm_fwd(kv, w[b + "ln_1_kv.weight"], w[b + "ln_1_kv.bias"], save_stats=True)
            tn, t_mean, t_rstd = hk.layernorm_fwd(t, w[b + "ln_1.weight"], w[b + "ln_1.bias"], save_stats=True)
            q = hk.gemm_nt(tn, Win[:d], bias=bin_[:d])
            kvp = hk.gemm_nt(kvn, Win[d:], bias=bin_[d:])                      # [B*912, 2d] = K | V
            o = torch.empty((M, d), device=self.device, dtype=torch.bfloat16)
            lse = torch.empty((nseq, H, LTq), device=self.device, dtype=torch.float32)
            hk.attn_fwd(q, kvp[:, :d], kvp[:, d:], o, lse, desc, nseq, H, d // H, max(STAGE_NUM), LTkv, LTq, False, scale)
            t1 = hk.gemm_nt(o, w[b + "attn.out_proj.weight"], bias=w[b + "attn.out_proj.bias"], residual=t)
            t1n, t1_mean, t1_rstd = hk.layernorm_fwd(t1, w[b + "ln_2.weight"], w[b + "ln_2.bias"], save_stats=True)
            hpre = hk.gemm_nt(t1n, w[b + "mlp.c_fc.weight"], bias=w[b + "mlp.c_fc.bias"])
            hact = hk.map_(hk.MAP_GELU, hpre)
            t2 = hk.gemm_nt(hact, w[b + "mlp.c_proj.weight"], bias=w[b + "mlp.c_proj.bias"], residual=t1)
            if save_ctx:
                layers.append(dict(t=t, tn=tn, t_stats=(t_mean, t_rstd), kvn=kvn, kv_stats=(kv_mean, kv_rstd), q=q, kvp=kvp, o=o,
                                   lse=lse, t1=t1, t1n=t1n, t1_stats=(t1_mean, t1_rstd), hpre=hpre, hact=hact))
            t = t2
        out = hk.gemm_nt(t, w["out_proj.weight"], bias=w["out_proj.bias"])
        if save_ctx:
            self._ctx = dict(B=B, kv=kv, layers=layers, t_final=t, desc=desc, LTq=LTq, LTkv=LTkv)
        return out.view(B, self.nq, self.out_dim)

    __call__ = forward

    # ------------------------------------------------------------------ backward
    def _dw(self, name: str, dy: torch.Tensor, x: torch.Tensor, rows=None) -> None:
        """grad[name] (rows slice) = dy^T @ x, straight from the token-major operands (lhrs_gemm_tn_f32: transposing LDS reads, token range
        split into f32 slabs, ordered sum); LHRS_DW_TRANSPOSED=1 takes the round-1 path (transposed copies + NT split-K GEMM)."""
        g = self.g[name] if rows is None else self.g[name][rows[0]: rows[1]]
        # lhrs_gemm_tn_f32 wants 16-B rows on both operands and on the output slice; anything else (a sliced / offset operand) takes the
        # transposed split-K path instead of failing inside the backward
        tn_ok = (g.shape[0] % 128 == 0 and g.shape[1] % 128 == 0 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0 and
                 x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and g.stride(1) == 1)
        if _dw_tn_enabled() and tn_ok:
            hk.gemm_tn_f32(dy, x, g)
            return
        Mp = hk.pad64(dy.shape[0])
        dyT = hk.transpose(dy, rows_pad=Mp)
        xT = hk.transpose(x, rows_pad=Mp)
        hk.gemm_nt_splitk_f32(dyT, xT, g)  # K = padded token count (B * 912 for the kv projection): split across blocks, ordered sum

    def backward(self, d_out: torch.Tensor, on_ready=None) -> None:
        """d_out [B,144,4096] bf16 -> fills self.grad (fp32).  The ViT is frozen: no image gradient is produced.
        on_ready(key) is called as soon as a gradient range is final ('out_proj', '5'..'0', 'query'): the engine
        launches that range's all-reduce on its comm stream while the rest of the backward runs."""
        c = self._ctx
        assert c is not None, "forward(save_ctx=True) must precede backward"
        B, d, H, g, w, wT = c["B"], self.d, self.heads, self.g, self.w, self.wT
        nseq, desc, LTq, LTkv = 3 * B, c["desc"], c["LTq"], c["LTkv"]
        scale = (d // H) ** -0.5
        d_out = d_out.reshape(B * self.nq, self.out_dim)
        hk.colsum(d_out, g["out_proj.bias"])
        self._dw("out_proj.weight", d_out, c["t_final"])
        dt = hk.gemm_nt(d_out, wT["out_proj.weight"])
        if on_ready:
            on_ready("out_proj")
        dkv_total = None
        for l in reversed(range(self.nl)):
            b = f"layers.{l}."
            s = c["layers"][l]
            # ---- MLP
            hk.colsum(dt, g[b + "mlp.c_proj.bias"])
            self._dw(b + "mlp.c_proj.weight", dt, s["hact"])
            dh = hk.gemm_nt(dt, wT[b + "mlp.c_proj.weight"])
            dh = hk.map_(hk.MAP_GELU_BWD, dh, s["hpre"], out=dh)
            hk.colsum(dh, g[b + "mlp.c_fc.bias"])
            self._dw(b + "mlp.c_fc.weight", dh, s["t1n"])
            dt1n = hk.gemm_nt(dh, wT[b + "mlp.c_fc.weight"])
            dt1 = hk.layernorm_bwd(dt1n, s["t1"], w[b + "ln_2.weight"], *s["t1_stats"], g[b + "ln_2.weight"], g[b + "ln_2.bias"], add=dt)
            # ---- cross-attention
            hk.colsum(dt1, g[b + "attn.out_proj.bias"])
            self._dw(b + "attn.out_proj.weight", dt1, s["o"])
            do = hk.gemm_nt(dt1, wT[b + "attn.out_proj.weight"])
            delta = torch.empty((nseq, H, LTq), device=self.device, dtype=torch.float32)
            hk.attn_delta(s["o"], do, delta, desc, nseq, H, d // H, max(STAGE_NUM), LTq)
            kvp = s["kvp"]
            dq = torch.empty_like(s["q"])
            dkvp = torch.empty_like(kvp)
            hk.attn_bwd(s["q"], kvp[:, :d], kvp[:, d:], do, s["lse"], delta, dq, dkvp[:, :d], dkvp[:, d:], desc, nseq, H, d // H,
                        max(STAGE_NUM), LTkv, LTq, False, scale)
            gb = g[b + "attn.in_proj_bias"]
            hk.colsum(dq, gb[:d])
            hk.colsum(dkvp, gb[d:])
            self._dw(b + "attn.in_proj_weight", dq, s["tn"], rows=(0, d))
            self._dw(b + "attn.in_proj_weight", dkvp, s["kvn"], rows=(d, 3 * d))
            dtn = hk.gemm_nt(dq, wT[b + "q"])
            dt = hk.layernorm_bwd(dtn, s["t"], w[b + "ln_1.weight"], *s["t_stats"], g[b + "ln_1.weight"], g[b + "ln_1.bias"], add=dt1)
            dkvn = hk.gemm_nt(dkvp, wT[b + "kv"])
            dkv_total = hk.layernorm_bwd(dkvn, c["kv"], w[b + "ln_1_kv.weight"], *s["kv_stats"], g[b + "ln_1_kv.weight"],
                                         g[b + "ln_1_kv.bias"], add=dkv_total)
            if on_ready:
                on_ready(str(l))
        hk.pooler_query_grad(dt, dkv_total, g["query"], B, STAGE_NUM, SPLIT_PART)
        if on_ready:
            on_ready("query")
        self._ctx = None
