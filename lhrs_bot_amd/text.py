"""TextModal: image-token splice + LLaMA-2 forward / activation-gradient backward + causal-LM loss on gfx950.

Mirrors /root/reference lhrs/models/text_modal.py: `TextModal.decode` (:258-294), `prepare_inputs_for_multimodal`
(:296-526), `CustomLlamaForCausalLM` (:30-60) over HF LlamaForCausalLM.  Stage 1 trains the projector only, so the
backward pass propagates dX through the frozen decoder (no dW): every linear's backward is one NT GEMM against a
pre-transposed copy of its weight that stays resident in HBM (2 x 13.5 GB of 288 GB).

Activation memory per layer and token: x_in, x_mid, o (3 x 4096), qkv (12288), gate|up (22016) in bf16 = 93 KB;
32 layers x 2184 tokens (B=8, S=273) = 6.5 GB - no gradient checkpointing (the reference needs it on 80 GB parts).
"""
from __future__ import annotations

import math
import os
import types
from typing import Dict, List, Optional

import torch

from . import kernels as hk

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


class SyntheticTokenizer:
    """Stands in for LlamaTokenizerFast when no tokenizer files exist offline (pad = unk = 0, bos = 1, eos = 2).  Text is split on
    whitespace and every word hashed to a stable id in [3, 32000) ("\n" -> 13, "</s>" -> 2): enough to drive the data pipeline, the
    prompt templates and the stopping criteria end to end in synthetic runs; it is NOT the LLaMA vocabulary and decoding is lossy
    (`decode` prints `<id>` for ids it has never encoded)."""
    unk_token_id = pad_token_id = 0
    bos_token_id = 1
    eos_token_id = 2
    model_max_length = 2048
    is_synthetic = True

    def __init__(self):
        self._words = {0: "<unk>", 1: "<s>", 2: "</s>", 13: "\n"}

    def __len__(self):
        return 32000

    def _id(self, w: str) -> int:
        if w == "\n":
            return 13
        if w == "</s>":
            return 2
        h = 2166136261
        for ch in w.encode():
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        i = 14 + h % (32000 - 14)
        self._words.setdefault(i, w)
        return i

    def __call__(self, text, **_kw):
        ids = [self.bos_token_id] + [self._id(w) for w in text.replace("\n", " \n ").split(" ") if w]
        return types.SimpleNamespace(input_ids=ids)

    def decode(self, ids, skip_special_tokens: bool = False, **_kw) -> str:
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        words = [self._words.get(int(i), f"<{int(i)}>") for i in ids if not (skip_special_tokens and int(i) in (0, 1, 2))]
        return " ".join(words).replace(" \n ", "\n")

    def batch_decode(self, rows, skip_special_tokens: bool = False, **_kw):
        return [self.decode(r, skip_special_tokens=skip_special_tokens) for r in rows]


LORA_GROUPS = {  # fused GEMM group -> (sub-projections, in_features, out_features per projection)
    "qkv": (("q", "k", "v"), 4096, 4096), "o": (("o",), 4096, 4096), "gu": (("gate", "up"), 4096, 11008), "down": (("down",), 11008, 4096)}
LORA_ALL = ("q", "k", "v", "o", "gate", "up", "down")  # find_all_linear_names: every nn.Linear except lm_head (text_modal.py:658-667)


def warp_logits(logits: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = None, top_p: Optional[float] = None) -> torch.Tensor:
    """The score processing HF `generate(do_sample=True)` applies before its multinomial draw, in HF's order: TemperatureLogitsWarper ->
    TopKLogitsWarper -> TopPLogitsWarper (transformers generation/logits_process.py; cli_qa.py:176-186 passes temperature, the Llama-2
    generation config top-k / top-p).  fp32 logits [B, V] in, filtered (-inf) logits out; pinned to the HF classes by
    tests/test_host_cpu.py::test_sampling_warpers_match_hf.  Runs on whatever device the HIP-computed logits live on."""
    z = logits / max(float(temperature), 1e-6)
    if top_k:
        k = min(int(top_k), z.shape[-1])
        kth = torch.topk(z, k, dim=-1).values[:, -1:]
        z = z.masked_fill(z < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sz, si = torch.sort(z, descending=False, dim=-1)
        cp = torch.softmax(sz, -1).cumsum(-1)
        rm = cp <= (1 - top_p)
        rm[:, -1] = False  # keep at least the most likely token
        z = z.masked_fill(rm.scatter(1, si, rm), float("-inf"))
    return z


class LoraStore:
    """peft-style LoRA adapters (lora.Linear: y = W x + (alpha/r) B A x; A ~ kaiming-uniform(a=sqrt 5), B = 0) for the fused
    GEMM groups, in three flat buffers (fp32 master, bf16 shadow, fp32 grad) ordered layer-major so that a layer's
    gradients form one contiguous all-reduce bucket.  Per (layer, group) two stacked tensors:
        A  [KP, in]          rows p*r..p*r+r = A_p                      (KP = r * n_proj padded to 64, padding rows stay 0)
        BD [KP, out_total]   block p = B_p^T at rows p*r.., cols p*out.. (block diagonal; off-blocks are masked to 0)
    and two derived bf16 operands rebuilt after every optimizer step: AT = A^T [in, KP], Bfull = BD^T [out_total, KP]."""

    def __init__(self, n_layers, r, alpha, targets, dims, device, seed=0, dropout=0.0):
        self.r, self.s, self.targets, self.device, self.nl = int(r), float(alpha) / float(r), tuple(targets), device, n_layers
        # peft lora_dropout: applied to the adapter input in train mode only; ONE counter-based mask per fused group and forward pass
        # (peft draws one per wrapped nn.Linear: q, k, v - and gate, up - share a mask here because they share the stacked A product)
        self.dropout, self.train_mode, self.drop_seed, self.drop_epoch = float(dropout), True, int(seed) * 2654435761 % (1 << 32), 0
        self.groups = {}
        for gname, (projs, fin, fout) in dims.items():
            mask = sum(1 << i for i, p in enumerate(projs) if p in self.targets)
            if mask:
                self.groups[gname] = dict(projs=projs, fin=fin, fout=fout, mask=mask, KP=(self.r * len(projs) + 63) // 64 * 64,
                                          out_total=fout * len(projs))
        self.offsets, off = {}, 0
        self.layer_range = []
        for l in range(n_layers):
            start = off
            for gname, G in self.groups.items():
                for kind, shape in (("A", (G["KP"], G["fin"])), ("BD", (G["KP"], G["out_total"]))):
                    self.offsets[(l, gname, kind)] = (off, shape)
                    off += shape[0] * shape[1]
            self.layer_range.append((start, off))
        self.numel = off
        self.master = torch.zeros(off, device=device, dtype=torch.float32)
        self.shadow = torch.zeros(off, device=device, dtype=torch.bfloat16)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.derived = {}
        g = torch.Generator(device=device).manual_seed(seed)
        for l in range(n_layers):
            for gname, G in self.groups.items():
                A = self.view(self.master, l, gname, "A")
                bound = 1.0 / math.sqrt(G["fin"])
                for i, p in enumerate(G["projs"]):
                    if (G["mask"] >> i) & 1:
                        A[i * self.r:(i + 1) * self.r].copy_((torch.rand((self.r, G["fin"]), device=device, generator=g) * 2 - 1) * bound)
        self.refresh()

    def view(self, flat, l, gname, kind):
        off, shape = self.offsets[(l, gname, kind)]
        return flat[off: off + shape[0] * shape[1]].view(*shape)

    def num_parameters(self):
        n = 0
        for G in self.groups.values():
            n += bin(G["mask"]).count("1") * self.r * (G["fin"] + G["fout"])
        return n * self.nl

    def active_dropout(self) -> float:
        return self.dropout if self.train_mode else 0.0

    def seed_for(self, l: int, gname: str) -> int:
        gid = list(self.groups).index(gname)
        return (self.drop_seed + self.drop_epoch * 0x85EBCA77 + (l * 8 + gid) * 0x9E3779B1) & 0xFFFFFFFF

    def refresh(self):
        """bf16 shadow + transposed operands after a load / optimizer step (all 8 transposes of every layer in one launch).  Bumps `version`:
        whatever was derived from the adapters (generate()'s merged decode weights) is stale from here on."""
        self.version = getattr(self, "version", 0) + 1
        hk.cast_f32_to_bf16(self.master, self.shadow)
        if getattr(self, "_bt", None) is None:
            pairs = []
            for l in range(self.nl):
                for gname in self.groups:
                    for kind, dk in (("A", "AT"), ("BD", "Bfull")):
                        src = self.view(self.shadow, l, gname, kind)
                        dst = torch.empty((src.shape[1], src.shape[0]), device=src.device, dtype=torch.bfloat16)
                        self.derived[(l, gname, dk)] = dst
                        pairs.append((src, dst))
            self._bt = hk.BatchedTranspose(pairs)
        self._bt.run()

    # peft-layout accessors (A_p [r, in], B_p [out, r]) for tests / checkpoints
    def get_adapter(self, l, proj):
        for gname, G in self.groups.items():
            if proj in G["projs"]:
                i = G["projs"].index(proj)
                A = self.view(self.master, l, gname, "A")[i * self.r:(i + 1) * self.r]
                Bt = self.view(self.master, l, gname, "BD")[i * self.r:(i + 1) * self.r, i * G["fout"]:(i + 1) * G["fout"]]
                return A, Bt.t()
        raise KeyError(proj)

    def set_adapter(self, l, proj, A, B):
        for gname, G in self.groups.items():
            if proj in G["projs"]:
                i = G["projs"].index(proj)
                self.view(self.master, l, gname, "A")[i * self.r:(i + 1) * self.r].copy_(A.to(self.device))
                self.view(self.master, l, gname, "BD")[i * self.r:(i + 1) * self.r, i * G["fout"]:(i + 1) * G["fout"]].copy_(B.t().to(self.device))
                return
        raise KeyError(proj)

    def grad_adapter(self, l, proj):
        for gname, G in self.groups.items():
            if proj in G["projs"]:
                i = G["projs"].index(proj)
                dA = self.view(self.grad, l, gname, "A")[i * self.r:(i + 1) * self.r]
                dBt = self.view(self.grad, l, gname, "BD")[i * self.r:(i + 1) * self.r, i * G["fout"]:(i + 1) * G["fout"]]
                return dA, dBt.t()
        raise KeyError(proj)


class TextModal:
    def __init__(self, config=None, device="cuda", layers=32, dim=4096, ff=11008, heads=32, vocab=32000, eps=1e-5,
                 rope_theta=10000.0, max_pos=4096):  # LLaMA-2 max_position_embeddings; S reaches model_max_length 2048 + 143 image tokens
        self.device = torch.device(device)
        hk.ensure_gemm_workspace(self.device)
        self.nl, self.d, self.ff, self.heads, self.vocab, self.eps = layers, dim, ff, heads, vocab, float(eps)
        self.hd = dim // heads
        self.tokenizer = SyntheticTokenizer()
        self.p: Dict = {}
        inv = 1.0 / (rope_theta ** (torch.arange(0, self.hd, 2).float() / self.hd))
        fr = torch.outer(torch.arange(max_pos).float(), inv)
        # HF casts cos/sin to the activation dtype before use; keep bf16-representable values in an fp32 table
        self.cos = fr.cos().to(torch.bfloat16).float().to(self.device).contiguous()
        self.sin = fr.sin().to(torch.bfloat16).float().to(self.device).contiguous()
        self._ctx = None
        self.lora: Optional[LoraStore] = None
        self.base8 = False  # frozen decoder linears in e4m3 (quantize_base(8, "e4m3"))
        self.base_int8 = False  # frozen decoder linears as LLM.int8 (quantize_base(8, "int8"): the reference's bitsandbytes arithmetic)
        self.base4 = None  # (quant_type, double_quant) once the decoder linears are bitsandbytes 4-bit storage (quantize_base(4, ...))
        self._i8ws = None
        # training forward: run the last decoder layer's post-attention half on the supervised rows only when they are one contiguous
        # range per sequence (_layer_fwd `tail`); LHRS_TAIL_ROWS_ONLY=0 / attribute False: every row, as HF does
        self.tail_rows_only = os.environ.get("LHRS_TAIL_ROWS_ONLY", "1") != "0"
        self.text_encoder = self  # attribute path used by the entry scripts (.text.text_encoder)

    def get_text_encoder(self):
        return self

    # ------------------------------------------------------------------ parameters
    def _finish(self):
        """Build the transposed copies used by the dX GEMMs (weights are frozen: done once)."""
        for L in self.p["layers"]:
            for k in ("qkv_w", "o_w", "gu_w", "down_w"):
                L[k + "T"] = hk.transpose(L[k])
        self.p["lm_headT"] = hk.transpose(self.p["lm_head"])

    def load_params(self, p: Dict) -> None:
        dev, bf = self.device, torch.bfloat16
        self.p = {"embed": p["embed"].to(dev, bf), "norm_w": p["norm_w"].to(dev, bf), "lm_head": p["lm_head"].to(dev, bf),
                  "layers": [{k: v.to(dev, bf).contiguous() for k, v in L.items()} for L in p["layers"]]}
        self.nl = len(self.p["layers"])
        self._merged_cache = None     # merged copies of the previous weights are stale
        self._finish()

    def from_pretrained(self, path: str, n_layers: Optional[int] = None) -> None:
        """CustomLlamaForCausalLM.from_pretrained(config.text.path) (text_modal.py:79-131): HF checkpoint directory ->
        engine layout.  Architecture numbers come from the checkpoint's own config.json, as in the reference."""
        import json
        import os
        from .checkpoint import llama_from_hf, load_hf_dir
        cfg = json.load(open(os.path.join(path, "config.json")))
        assert cfg["hidden_size"] == self.d and cfg["intermediate_size"] == self.ff and cfg["num_attention_heads"] == self.heads, cfg
        assert cfg.get("num_key_value_heads", self.heads) == self.heads, "GQA checkpoints are outside the reference's LLaMA-2-7B path"
        self.eps = float(cfg.get("rms_norm_eps", self.eps))
        self.load_params(llama_from_hf(load_hf_dir(path), n_layers))

    def merge_lora(self) -> None:
        """peft merge_and_unload (UniBind.custom_load_state_dict, stage == 0; lhrs/models/UniBind.py:112-115):
        W <- W + (alpha/r) B A, computed per fused group as one GEMM  W + s * Bfull . AT^T, then the adapters are dropped."""
        lo = self.lora
        if lo is None:
            return
        wname = {"qkv": "qkv_w", "o": "o_w", "gu": "gu_w", "down": "down_w"}
        for li, L in enumerate(self.p["layers"]):
            for gname in lo.groups:
                W = L[wname[gname]]
                hk.gemm_nt(lo.derived[(li, gname, "Bfull")], lo.derived[(li, gname, "AT")], out=W, residual=W, alpha=lo.s)
                L[wname[gname] + "T"] = hk.transpose(W)
            self._drop_derived(L)
        requant = "e4m3" if self.base8 else ("int8" if self.base_int8 else None)
        base4 = getattr(self, "base4", None)
        self.lora, self._merged_cache, self.base8, self.base_int8 = None, None, False, False
        if requant:
            self.quantize_base(8, requant)   # the 8-bit base is a function of the (now merged) weights
        elif base4:
            self.quantize_base(4, quant_type=base4[0], double_quant=base4[1])   # peft re-quantises a merged Linear4bit the same way

    _W4_PARTS = {"qkv_w": 3, "o_w": 1, "gu_w": 2, "down_w": 1}   # reference Linears per fused weight (row-concatenated): 4-bit statistics are per Linear
    DERIVED_SUFFIXES = ("p", "8", "8s", "8p", "i8", "i8s", "q4")   # decode re-tilings and e4m3 copies of a weight `<name>` / `<name>T`, rebuilt lazily from it

    def _drop_derived(self, L) -> None:
        """Forget every tensor that was computed FROM a decoder weight of layer dict `L` (decode re-tilings, e4m3 copies): after the weight
        changes in place (merge_lora, a checkpoint load) they are stale; their builders re-create them on next use."""
        for base in ("qkv_w", "o_w", "gu_w", "down_w", "qkv_wT", "o_wT", "gu_wT", "down_wT"):
            for suf in self.DERIVED_SUFFIXES:
                L.pop(base + suf, None)

    def init_random(self, seed: int = 0) -> None:
        """Random-init LLaMA-2-7B shapes, N(0, 0.02) (HF initializer_range) - no weights exist offline."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        dev, bf, d, ff = self.device, torch.bfloat16, self.d, self.ff

        def rn(*shape, std=0.02, mean=0.0):
            out = torch.empty(*shape, device=dev, dtype=bf)
            rows = shape[0]
            step = max(1, (1 << 26) // max(1, out[0].numel())) if len(shape) > 1 else rows
            for r0 in range(0, rows, step):  # chunked so the fp32 temporary stays small
                r1 = min(rows, r0 + step)
                out[r0:r1] = (torch.randn((r1 - r0,) + tuple(shape[1:]), device=dev, generator=g) * std + mean).to(bf)
            return out

        self.p = {"embed": rn(self.vocab, d), "norm_w": rn(d, std=0.0, mean=1.0), "lm_head": rn(self.vocab, d), "layers": []}
        for _ in range(self.nl):
            self.p["layers"].append({"ln1_w": rn(d, std=0.0, mean=1.0), "qkv_w": rn(3 * d, d), "o_w": rn(d, d),
                                     "ln2_w": rn(d, std=0.0, mean=1.0), "gu_w": rn(2 * ff, d), "down_w": rn(d, ff)})
        self._finish()

    def enable_lora(self, r=128, alpha=256, targets=LORA_ALL, seed=0, dropout=0.0) -> LoraStore:
        """TextModal.__init__ LoRA branch (text_modal.py:133-151): LoraConfig(r, lora_alpha, target_modules=all linears).
        BASELINE config 4 uses r=8 on ("q","k","v","o").  dropout = lora_dropout (0.05 in the stage-2 YAML; train mode only - stage 3
        runs text.eval())."""
        dims = {"qkv": (("q", "k", "v"), self.d, self.d), "o": (("o",), self.d, self.d), "gu": (("gate", "up"), self.d, self.ff),
                "down": (("down",), self.ff, self.d)}
        self.lora = LoraStore(len(self.p["layers"]), r, alpha, targets, dims, self.device, seed, dropout)
        return self.lora

    def _drop(self, rec, gname):
        """(p, seed) of the dropout mask this group's forward used, or None"""
        seed = rec.get("seed_" + gname)
        return None if seed is None else (self.lora.dropout, seed)

    def _q8(self, L, name):
        """(e4m3 weight, per-row scales) of L[name] when the base weights are 8-bit (quantize_base), else None."""
        return (L[name + "8"], L[name + "8s"]) if self.base8 else None

    def _i8(self, L, name):
        """(int8 rows, dequantisation factors) of a decoder weight under the LLM.int8 base, else None"""
        return (L[name + "i8"], L[name + "i8s"]) if self.base_int8 else None

    def _lin(self, li, gname, x, W, residual=None, save=None, q8=None, xq=None, rope=None, i8=None):
        """y = x W^T (+ s (x A^T) B^T when the group carries adapters) (+ residual).  q8 = (W8, scales): the frozen base product runs
        on the e4m3 MFMA path (x quantised per row on the fly), the adapter update stays bf16 and rides on the same accumulators (lhrs_gemm_fp8_nt_lora).
        rope = (pos_mod, pos0): the qkv projection - RoPE of the q / k heads in the bf16 GEMM's epilogue, a separate launch after the e4m3 one."""
        lo = self.lora
        has_lora = lo is not None and gname in lo.groups
        if has_lora:
            xa, pd = x, lo.active_dropout()
            if pd > 0:  # lora_A(dropout(x)): the mask is regenerated from the saved seed in the backward
                seed = lo.seed_for(li, gname)
                xa = hk.dropout(x, pd, seed)
                if save is not None:
                    save["seed_" + gname] = seed
            T = hk.gemm_nt_skinny(xa, lo.view(lo.shadow, li, gname, "A"), alpha=lo.s)          # [M, KP] = s * dropout(x) A^T
            if save is not None:
                save["T_" + gname] = T
        if i8 is not None:  # LLM.int8 base (text_modal.py:91-131 -> bitsandbytes MatMul8bitLt): int8 product + 16-bit outlier columns, adapters on top
            y = hk.int8_linear(x, i8[0], i8[1], self._i8ws, residual=residual, a2=T if has_lora else None,
                               b2=lo.derived[(li, gname, "Bfull")] if has_lora else None)
            if rope is not None:
                hk.rope_(y, y.shape[0], 2 * self.heads, self.hd, self.cos, self.sin, pos_mod=rope[0], pos0=rope[1])
            return y
        if q8 is not None:
            x8, sx = xq if xq is not None else hk.quant_fp8_rows(x)  # xq: the producer already emitted the e4m3 operand
            if has_lora:
                y = hk.gemm_fp8_nt(x8, sx, q8[0], q8[1], residual=residual, a2=T, b2=lo.derived[(li, gname, "Bfull")])
            else:
                y = hk.gemm_fp8_nt(x8, sx, q8[0], q8[1], residual=residual)
            if rope is not None:
                hk.rope_(y, y.shape[0], 2 * self.heads, self.hd, self.cos, self.sin, pos_mod=rope[0], pos0=rope[1])
            return y
        if rope is not None:
            return hk.gemm_rope_fwd(x, W, self.cos, self.sin, pos_mod=rope[0], pos0=rope[1], rope_cols=2 * self.d, head_dim=self.hd,
                                    a2=T if has_lora else None, b2=lo.derived[(li, gname, "Bfull")] if has_lora else None)
        if not has_lora:
            return hk.gemm_nt(x, W, residual=residual)
        return hk.gemm_nt_lora(x, W, T, lo.derived[(li, gname, "Bfull")], residual=residual)

    def _gu_fwd(self, li, h, W, save, q8=None, xq=None, i8=None):
        """gate|up projection with the SwiGLU in the GEMM epilogue (one launch): -> (gu [M, 2ff], act [M, ff])."""
        lo = self.lora
        if i8 is not None:
            gu = self._lin(li, "gu", h, W, save=save, i8=i8)
            return gu, hk.swiglu_fwd(gu, self.ff), None
        if self.base8:  # SwiGLU emits the e4m3 operand of the down projection; bf16 act only if an adapter on `down` needs it
            gu = self._lin(li, "gu", h, W, save=save, q8=q8, xq=xq)
            act, act8, sact = hk.swiglu_fwd_q(gu, self.ff, want_bf16=lo is not None and "down" in lo.groups)
            return gu, act, (act8, sact)
        if lo is None or "gu" not in lo.groups:
            return hk.gemm_swiglu_fwd(h, W, self.ff) + (None,)
        ha, pd = h, lo.active_dropout()
        if pd > 0:
            seed = lo.seed_for(li, "gu")
            ha = hk.dropout(h, pd, seed)
            if save is not None:
                save["seed_gu"] = seed
        T = hk.gemm_nt_skinny(ha, lo.view(lo.shadow, li, "gu", "A"), alpha=lo.s)
        if save is not None:
            save["T_gu"] = T
        return hk.gemm_swiglu_fwd(h, W, self.ff, T, lo.derived[(li, "gu", "Bfull")]) + (None,)

    def _down_bwd(self, li, dy, WT, gu, act, T, q8=None, dyq=None, drop=None):
        """dgu (written over gu) = swiglu'(gu) * d_act with d_act = dy W_down (+ LoRA) never leaving the GEMM epilogue."""
        lo = self.lora
        if drop is not None and q8 is None:  # masked adapter term: unfused sequence
            return hk.swiglu_bwd(self._lin_bwd(li, "down", dy, WT, act, T, drop=drop), gu, self.ff, out=gu)
        if q8 is not None:  # -> (d(gate|up) bf16 over gu or None, its e4m3 operand for the gate|up dX product)
            dact = self._lin_bwd(li, "down", dy, WT, act, T, q8=q8, dyq=dyq, drop=drop)
            dgu, dgu8, sdgu = hk.swiglu_bwd_q(dact, gu, self.ff, want_bf16=lo is not None and "gu" in lo.groups)
            return dgu, (dgu8, sdgu)
        if lo is None or "down" not in lo.groups:
            return hk.gemm_swiglu_bwd(dy, WT, gu, self.ff)
        G = lo.groups["down"]
        U = hk.gemm_nt_skinny(dy, lo.view(lo.shadow, li, "down", "BD"), alpha=lo.s)
        dgu = hk.gemm_swiglu_bwd(dy, WT, gu, self.ff, U, lo.derived[(li, "down", "AT")])
        hk.gemm_tn_skinny(U, act, lo.view(lo.grad, li, "down", "A"))
        dBD = hk.gemm_tn_skinny(T, dy, lo.view(lo.grad, li, "down", "BD"))
        hk.blockdiag_mask(dBD, lo.r, G["fout"], G["mask"])
        return dgu

    def _lin_bwd(self, li, gname, dy, WT, x, T, q8=None, dyq=None, drop=None):
        """dx = dy W (+ s (dy B) A); adapter gradients dA = (s dy B)^T x, dB^T = (s x A^T)^T dy written into lora.grad.
        q8 = (WT8, scales): e4m3 copy of the transposed base weight (per in-feature scales), dy quantised per row on the fly."""
        lo = self.lora
        has_lora = lo is not None and gname in lo.groups
        if q8 is not None:
            dy8, sdy = dyq if dyq is not None else hk.quant_fp8_rows(dy)
            if not has_lora:
                return hk.gemm_fp8_nt(dy8, sdy, q8[0], q8[1])
        elif not has_lora:
            return hk.gemm_nt(dy, WT)
        G = lo.groups[gname]
        U = hk.gemm_nt_skinny(dy, lo.view(lo.shadow, li, gname, "BD"), alpha=lo.s)         # [M, KP] = s * dy B
        if drop is not None:  # dx = dy W + mask * (U A) / (1 - p): the adapter term cannot share the base product's accumulators
            base = hk.gemm_fp8_nt(dy8, sdy, q8[0], q8[1]) if q8 is not None else hk.gemm_nt(dy, WT)
            dx = hk.gemm_nt_dropmask(U, lo.derived[(li, gname, "AT")], drop[0], drop[1], residual=base)
            x = hk.dropout(x, drop[0], drop[1])                                            # dA sees the same masked input
        elif q8 is not None:
            dx = hk.gemm_fp8_nt(dy8, sdy, q8[0], q8[1], a2=U, b2=lo.derived[(li, gname, "AT")])
        else:
            dx = hk.gemm_nt_lora(dy, WT, U, lo.derived[(li, gname, "AT")])
        hk.gemm_tn_skinny(U, x, lo.view(lo.grad, li, gname, "A"))
        dBD = hk.gemm_tn_skinny(T, dy, lo.view(lo.grad, li, gname, "BD"))
        hk.blockdiag_mask(dBD, lo.r, G["fout"], G["mask"])
        return dx

    # ------------------------------------------------------------------ splice
    def prepare_inputs_for_multimodal(self, input_ids, attention_mask, labels, image_embedding):
        """Device-side restatement of text_modal.py:296-526 (tune_im_start off) -> (embeds, labels, mask, img_pos).  One placeholder per sample:
        everything on the device (`lhrs_splice_fwd`).  Several placeholders in a sample (or an image tensor with one slot per placeholder): the
        walk runs on the host (`splice_plan_host`), the device copies rows through its map; the fourth value is then the inverse map."""
        ids = input_ids.to(self.device)
        B, T = ids.shape
        if image_embedding is None:  # text-only turn (text_modal.py:321-339): no <image> token may be present, plain embeddings
            if bool((ids == IMAGE_TOKEN_INDEX).any()):
                raise ValueError("input_ids contain the <image> placeholder but no image embedding was given")
            image_embedding = torch.zeros((B, 1, self.d), device=self.device, dtype=torch.bfloat16)
        NI = image_embedding.shape[1]
        has_img = (ids == IMAGE_TOKEN_INDEX).any(dim=1)
        n_img = (ids == IMAGE_TOKEN_INDEX).sum(dim=1)
        if int(n_img.max()) > 1 or image_embedding.shape[0] != B:   # several placeholders in a sample: host walk + mapped copy (text_modal.py:341-438)
            ids_h, lab_h, msk_h = self._ints_to_host(input_ids, labels, attention_mask)
            plan = self.splice_plan_host(ids_h, lab_h, msk_h, NI)
            self._check_slots(plan, image_embedding)
            embeds = hk.splice_map_fwd(ids, hk.h2d(plan["src_tok"], self.device), hk.h2d(plan["src_img"], self.device), image_embedding.contiguous(),
                                       self.p["embed"], plan["S"])
            return embeds, hk.h2d(plan["labels"], self.device), hk.h2d(plan["mask"], self.device), hk.h2d(plan["inv"], self.device)
        S = T - 1 + NI if bool(has_img.any()) else T
        lab = None if labels is None else labels.to(self.device)
        msk = None if attention_mask is None else attention_mask.to(self.device)
        return hk.splice_fwd(ids, lab, msk, image_embedding.contiguous(), self.p["embed"], S)

    # ------------------------------------------------------------------ forward
    def _layer_fwd(self, L, x, B, S, desc, LT, save, li=0, tail=None):
        """One decoder layer.  tail = (rows int32 [n], desc, max_q): the LAST layer of a training forward whose supervised positions
        form one contiguous range per sequence - only those rows of its output are ever read (final norm -> lm_head -> loss), so the
        attention runs for those queries only (all keys), and o_proj / residual / RMSNorm / MLP run on the n gathered rows instead of
        all B*S.  HF computes every row and the loss ignores them: same loss, same gradients, ~0.4 of a layer's linear work less."""
        d, H, hd, ff = self.d, self.heads, self.hd, self.ff
        M = x.shape[0]
        if self._native_layer(tail):   # frozen bf16 base, no adapters, every row: the whole layer is ONE call into the library (same launches, same order)
            if getattr(self, "_h_scratch", None) is None or self._h_scratch.shape != x.shape:
                self._h_scratch = torch.empty_like(x)
            x_out, rec = hk.llama_layer_forward(x, L, self.cos, self.sin, desc, B, S, LT, H, ff, self.eps, self._h_scratch)
            if save is not None:
                save.append(rec)
            return x_out
        rec = {} if save is not None else None
        lo = self.lora
        hq = None
        if self.base8:  # RMSNorm emits the e4m3 operand of the next GEMM; the bf16 copy only if an adapter reads it
            h, hq = hk.rmsnorm_fwd_q(x, L["ln1_w"], self.eps, want_bf16=lo is not None and "qkv" in lo.groups)
        else:
            h = hk.rmsnorm_fwd(x, L["ln1_w"], self.eps)
        qkv = self._lin(li, "qkv", h, L["qkv_w"], save=rec, q8=self._q8(L, "qkv_w"), xq=hq, rope=(S, 0), i8=self._i8(L, "qkv_w"))
        o = torch.empty((M, d), device=self.device, dtype=torch.bfloat16)
        lse = torch.empty((B, H, LT), device=self.device, dtype=torch.float32)
        o_full, x_res = o, x
        if tail is None:
            hk.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, desc, B, H, hd, S, S, LT, True, 1.0 / math.sqrt(hd))
        else:
            rows, tdesc, max_q = tail
            hk.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, tdesc, B, H, hd, max_q, S, LT, True, 1.0 / math.sqrt(hd))
            o, x_res = hk.gather_rows(o_full, rows), hk.gather_rows(x, rows)   # from here on: n rows
        x_mid = self._lin(li, "o", o, L["o_w"], residual=x_res, save=rec, q8=self._q8(L, "o_w"), i8=self._i8(L, "o_w"))
        if self.base8:
            h, hq = hk.rmsnorm_fwd_q(x_mid, L["ln2_w"], self.eps, want_bf16=lo is not None and "gu" in lo.groups)
        else:
            h = hk.rmsnorm_fwd(x_mid, L["ln2_w"], self.eps, out=h if tail is None else None)
        gu, act, actq = self._gu_fwd(li, h, L["gu_w"], rec, q8=self._q8(L, "gu_w"), xq=hq, i8=self._i8(L, "gu_w"))
        x_out = self._lin(li, "down", act, L["down_w"], residual=x_mid, save=rec, q8=self._q8(L, "down_w"), xq=actq, i8=self._i8(L, "down_w"))
        if save is not None:
            rec.update(x_in=x, qkv=qkv, o=o, o_full=o_full, lse=lse, x_mid=x_mid, gu=gu)
            save.append(rec)
        return x_out

    def _native_layer(self, tail) -> bool:
        """True when a decoder layer can go through the module-level entry points (lhrs_llama_layer_forward / _backward): the frozen bf16
        base without adapters, on every row (the compact last layer keeps the operator path).  LHRS_NATIVE_LAYER=0: operator path (A/B, tests)."""
        return (tail is None and self.lora is None and not self.base8 and not self.base_int8 and self.hd == 128
                and os.environ.get("LHRS_NATIVE_LAYER", "1") != "0")

    def forward_hidden(self, embeds, mask_u8, save_ctx=True, kv_len=None, tail=None):
        """embeds [B,S,d] bf16, mask [B,S] uint8 (right padding) -> final-norm hidden [B*S, d].  kv_len: the per-sequence key counts when
        the caller already has them on the host (saves the device round trip of summing the mask).  tail = [(p0, n)] per sequence: only
        the positions [p0, p0 + n) are needed from the last layer on (see _layer_fwd) -> hidden [sum n, d] of those rows, sequence-major."""
        B, S, d = embeds.shape
        if S > self.cos.shape[0]:
            raise ValueError(f"sequence length {S} exceeds the {self.cos.shape[0]} positions of the RoPE table")
        if kv_len is None:
            kv_len = mask_u8.to(torch.int32).sum(dim=1).tolist() if mask_u8 is not None else [S] * B
        desc = hk.make_desc([(b * S, S, b * S, int(kv_len[b]), S, 0) for b in range(B)], self.device)
        LT = hk.pad64(S)
        saved: Optional[List] = [] if save_ctx else None
        if self.lora is not None:
            self.lora.drop_epoch += 1  # a fresh dropout mask per forward pass
        x = embeds.reshape(B * S, d)
        nl = len(self.p["layers"])
        tail_t = None
        if tail is not None and nl > 0:
            rows = torch.cat([torch.arange(b * S + p0, b * S + p0 + n, dtype=torch.int32) for b, (p0, n) in enumerate(tail)])
            tdesc = hk.make_desc([(b * S + p0, n, b * S, int(kv_len[b]), S, p0) for b, (p0, n) in enumerate(tail)], self.device)
            tail_t = (hk.h2d(rows, self.device), tdesc, max(n for _, n in tail))
        for li, L in enumerate(self.p["layers"]):
            x = self._layer_fwd(L, x, B, S, desc, LT, saved, li, tail=tail_t if li == nl - 1 else None)
        hidden = hk.rmsnorm_fwd(x, self.p["norm_w"], self.eps)
        if save_ctx:
            self._ctx = dict(B=B, S=S, desc=desc, LT=LT, layers=saved, x_last=x, tail=tail_t)
        return hidden

    @staticmethod
    def splice_ints_host(ids, labels, mask, NI):
        """Host mirror of the INTEGER outputs of `lhrs_splice_fwd` (text_modal.py:296-526: spliced length, labels, mask) on CPU tensors -
        what the training step needs on the host anyway (attention descriptor, supervised rows), computed without asking the device.
        -> (S, has_image [B] bool, new_labels [B,S] int64, new_mask [B,S] uint8); tests pin it to the device kernel."""
        B, T = ids.shape
        is_img = ids == IMAGE_TOKEN_INDEX
        n_img = is_img.sum(dim=1)
        if int(n_img.max()) > 1:
            raise ValueError("several <image> placeholders in a sample: use splice_plan_host (the general walk)")
        has = n_img > 0
        S = T - 1 + NI if bool(has.any()) else T
        p = torch.where(has, is_img.to(torch.int64).argmax(dim=1), torch.full((B,), T, dtype=torch.int64))[:, None]
        hasb = has[:, None]
        j = torch.arange(S, dtype=torch.int64)[None, :]
        new_len = torch.where(hasb, torch.full_like(p, T - 1 + NI), torch.full_like(p, T))
        shift = new_len - T
        src_tok = torch.where(~hasb | (j < p), j.expand(B, S), torch.where(j < p + NI, torch.full((B, S), -1, dtype=torch.int64), j - NI + 1))
        src_tok = torch.where(j < new_len, src_tok, torch.full_like(src_tok, -1))
        if labels is None:
            new_labels = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64)
        else:
            new_labels = torch.where(src_tok >= 0, labels.to(torch.int64).gather(1, src_tok.clamp(0, T - 1)), torch.full((B, S), IGNORE_INDEX, dtype=torch.int64))
        m = torch.ones((B, T), dtype=torch.uint8) if mask is None else mask.to(torch.uint8)
        new_mask = torch.where(j < new_len, torch.where(j < shift, torch.ones((B, S), dtype=torch.uint8), m.gather(1, (j - shift).clamp(0, T - 1))),
                               torch.zeros((B, S), dtype=torch.uint8))
        return S, has, new_labels, new_mask

    @staticmethod
    def splice_plan_host(ids, labels, mask, NI, tune_im_start=False):
        """The GENERAL walk of prepare_inputs_for_multimodal (text_modal.py:318-438) on CPU tensors: any number of <image> placeholders per
        sample, each one taking the next image SLOT of a counter that runs over the batch and that a placeholder-free sample advances too
        (`cur_image_idx`, :339, :402).  -> dict(S, n_slots, labels int64 [B,S], mask uint8 [B,S], src_tok int32 [B,S] (token index or -1),
        src_img int32 [B,S] (row of the flattened slots [n_slots * NI] or -1), inv int32 [n_slots * NI] (flat output row b * S + j that copies
        image row r, -1 for a slot nobody took)).  tune_im_start: the integer side of the `tune_pooler and tune_im_start` branch (:353-387) -
        the token map is the plain one, the label kept behind the image is the placeholder's own and the walk resumes two tokens on."""
        B, T = ids.shape
        rows_tok, rows_img, rows_lab = [], [], []
        slot = 0
        for b in range(B):
            cur = ids[b].tolist()
            lab = labels[b].tolist() if labels is not None else None
            tok, img, nl = [], [], []
            base = 0
            if IMAGE_TOKEN_INDEX not in cur:
                tok, img, nl = list(range(T)), [-1] * T, (list(lab) if lab is not None else [])
                slot += 1
            else:
                while IMAGE_TOKEN_INDEX in cur:
                    p = cur.index(IMAGE_TOKEN_INDEX)
                    tok += [base + i for i in range(p)] + [-1] * NI
                    img += [-1] * p + [slot * NI + k for k in range(NI)]
                    step = 1
                    if tune_im_start:
                        if p + 1 >= len(cur):
                            raise ValueError("tune_im_start: every <image> placeholder needs its <im_end> neighbour (cap_dataset.py:875-876)")
                        tok.append(base + p + 1)
                        img.append(-1)
                        step = 2
                    if lab is not None:
                        nl += lab[:p] + [IGNORE_INDEX] * NI + (lab[p:p + 1] if tune_im_start else [])
                        lab = lab[p + step:]
                    cur = cur[p + step:]
                    base += p + step
                    slot += 1
                tok += [base + i for i in range(len(cur))]
                img += [-1] * len(cur)
                if lab is not None:
                    nl += lab
            rows_tok.append(tok); rows_img.append(img); rows_lab.append(nl)
        S = max(len(r) for r in rows_tok)
        src_tok = torch.full((B, S), -1, dtype=torch.int32)
        src_img = torch.full((B, S), -1, dtype=torch.int32)
        new_labels = torch.full((B, S), IGNORE_INDEX, dtype=torch.int64)
        new_mask = torch.zeros((B, S), dtype=torch.uint8)
        inv = torch.full((max(slot, 1) * NI,), -1, dtype=torch.int32)
        m = torch.ones((B, T), dtype=torch.uint8) if mask is None else mask.to(torch.uint8)
        for b in range(B):
            n = len(rows_tok[b])
            src_tok[b, :n] = torch.tensor(rows_tok[b], dtype=torch.int32)
            src_img[b, :n] = torch.tensor(rows_img[b], dtype=torch.int32)
            if labels is not None:
                new_labels[b, :n] = torch.tensor(rows_lab[b], dtype=torch.int64)
            new_mask[b, : n - T] = 1                      # the reference left-extends the mask by the growth of the row (:511-524)
            new_mask[b, n - T: n] = m[b]
            j = torch.nonzero(src_img[b] >= 0).squeeze(1)
            inv[src_img[b, j].long()] = (b * S + j).to(torch.int32)
        return dict(S=S, n_slots=slot, labels=new_labels, mask=new_mask, src_tok=src_tok, src_img=src_img, inv=inv)

    def _splice_general(self, ids_h, image_embedding, NI):
        """True when the batch needs the general walk: a sample with several placeholders, or an image tensor that is not one slot per sample."""
        n_img = (ids_h == IMAGE_TOKEN_INDEX).sum(dim=1)
        return int(n_img.max()) > 1 or image_embedding.shape[0] != ids_h.shape[0]

    def _check_slots(self, plan, image_embedding):
        if plan["n_slots"] > image_embedding.shape[0]:   # the reference: IndexError at image_embedding[cur_image_idx] (text_modal.py:343-345)
            raise IndexError(f"the batch takes {plan['n_slots']} image slots (one per <image> placeholder, one per placeholder-free sample) but "
                             f"image_embedding holds {image_embedding.shape[0]}")

    def _ints_to_host(self, *ts):
        """CPU views of the small integer inputs.  Host tensors (what a DataLoader delivers) pass through; device tensors cost ONE
        synchronising copy for all of them."""
        if not any(t is not None and t.is_cuda for t in ts):
            return list(ts)
        if os.environ.get("LHRS_INTS_PAGEABLE") == "1":   # the pre-round-5 path (A/B of the shared-device NaN hunt): async copies into PAGEABLE host memory
            out = [None if t is None else (t if not t.is_cuda else t.to("cpu", non_blocking=True)) for t in ts]
            torch.cuda.current_stream().synchronize()
            return out
        # device -> PINNED staging buffers of this model, one synchronisation, then plain host copies of them: the runtime does not have to pin / stage / unpin
        # pageable pages around an asynchronous copy at the start of every step.  ONE flat pinned buffer per (argument slot, dtype), grown geometrically to the
        # largest element count seen and viewed per call: ragged batches (every distinct (B, T)) do not accumulate page-locked memory
        cache = self.__dict__.setdefault("_ints_pinned", {})
        stage = []
        for i, t in enumerate(ts):
            if t is None or not t.is_cuda:
                stage.append(None)
                continue
            key = (i, t.dtype)
            n = t.numel()
            if key not in cache or cache[key].numel() < n:
                cache[key] = torch.empty(max(n, 2 * cache[key].numel() if key in cache else n), dtype=t.dtype, pin_memory=True)
            view = cache[key][:n].view(t.shape)
            view.copy_(t, non_blocking=True)
            stage.append(view)
        torch.cuda.current_stream().synchronize()
        return [t if p is None else p.clone() for t, p in zip(ts, stage)]

    def decode(self, input_ids, image_embedding=None, attention_mask=None, labels=None, save_ctx=True, host_ints=None):
        """TextModal.decode: returns the scalar text loss (0-dim fp32 device tensor).  All integer bookkeeping of the step (spliced
        length, key counts, supervised rows, shifted targets) happens on the host from the host copies of ids / labels / mask, so the
        launch queue is never drained in the middle of a step; the device splice only moves embedding rows.  host_ints: those copies
        when the caller fetched them already (UniBind.forward does, before it enqueues the ViT)."""
        if labels is None:
            raise ValueError("decode() computes the training loss: labels are required")
        if save_ctx:
            self._merged_cache = None     # a training step follows: do not pin ~13.5 GB of merged generate() weights through it
        ids_h, lab_h, msk_h = host_ints if host_ints is not None else self._ints_to_host(input_ids, labels, attention_mask)
        B, T = ids_h.shape
        if image_embedding is None:
            if bool((ids_h == IMAGE_TOKEN_INDEX).any()):
                raise ValueError("input_ids contain the <image> placeholder but no image embedding was given")
            image_embedding = torch.zeros((B, 1, self.d), device=self.device, dtype=torch.bfloat16)
        NI = image_embedding.shape[1]
        plan = None
        if self._splice_general(ids_h, image_embedding, NI):
            plan = self.splice_plan_host(ids_h, lab_h, msk_h, NI)
            self._check_slots(plan, image_embedding)
            S, new_labels, new_mask = plan["S"], plan["labels"], plan["mask"]
        else:
            S, _, new_labels, new_mask = self.splice_ints_host(ids_h, lab_h, msk_h, NI)
        # shifted targets: position j predicts label j+1 (HF LlamaForCausalLM.forward); ignore_index rows are skipped
        tgt = torch.full_like(new_labels, IGNORE_INDEX)
        tgt[:, :-1] = new_labels[:, 1:]
        flat = tgt.reshape(-1)
        rows = torch.nonzero(flat != IGNORE_INDEX).squeeze(1)
        if rows.numel() == 0:
            raise ValueError("no valid target token in the micro-batch (loss would be NaN in the reference)")
        rows32 = hk.h2d(rows.to(torch.int32), self.device)
        targets = hk.h2d(flat[rows].to(torch.int32), self.device)
        ids_d = input_ids if input_ids.is_cuda else hk.h2d(ids_h.contiguous(), self.device)
        if plan is None:
            embeds, _, _, img_pos = hk.splice_fwd(ids_d, None, None, image_embedding.contiguous(), self.p["embed"], S)
            img_inv = None
        else:
            embeds = hk.splice_map_fwd(ids_d, hk.h2d(plan["src_tok"], self.device), hk.h2d(plan["src_img"], self.device),
                                       image_embedding.contiguous(), self.p["embed"], S)
            img_pos, img_inv = None, (hk.h2d(plan["inv"], self.device), image_embedding.shape[0])
        # supervised positions of each sequence: when they form ONE contiguous range (stage 1: the caption at the end of the sequence; any
        # single-answer sample) the last decoder layer only has to produce those rows
        tail = None
        if self.tail_rows_only and self.p["layers"]:
            valid = tgt != IGNORE_INDEX
            cnt = valid.sum(dim=1)
            first = valid.to(torch.int64).argmax(dim=1)
            last = S - 1 - valid.flip(1).to(torch.int64).argmax(dim=1)
            if bool((cnt > 0).all()) and bool((last - first + 1 == cnt).all()):
                tail = list(zip(first.tolist(), cnt.tolist()))
        hidden = self.forward_hidden(embeds, None, save_ctx, kv_len=new_mask.to(torch.int32).sum(dim=1).tolist(), tail=tail)
        self.last_hidden = hidden  # final-norm output of this call: [B*S, d], or the supervised rows only in tail mode (tests read it)
        hv = hidden if tail is not None else hk.gather_rows(hidden, rows32)
        logits = hk.gemm_nt(hv, self.p["lm_head"])
        loss, dlogits = hk.cross_entropy(logits, targets, want_grad=save_ctx, inplace=True)
        if save_ctx:
            self._ctx.update(rows=rows32, dlogits=dlogits, img_pos=img_pos, img_inv=img_inv, NI=NI)
        return loss

    __call__ = decode

    # ------------------------------------------------------------------ generate (KV cache)
    def _layer_step(self, L, x, B, S_new, ctx, cache, desc, max_ctx, kmask=None):
        """One decoder layer over S_new new positions per sequence with `ctx` cached positions (prefill: ctx = 0)."""
        d, H, hd, ff = self.d, self.heads, self.hd, self.ff
        M = x.shape[0]
        h = hk.rmsnorm_fwd(x, L["ln1_w"], self.eps)
        qkv = hk.gemm_rope_fwd(h, L["qkv_w"], self.cos, self.sin, pos_mod=S_new, pos0=ctx, rope_cols=2 * d, head_dim=hd)
        kc, vc = cache
        row_b = d * 2
        for b in range(B):  # append the new K / V rows of sequence b at position ctx of its cache
            src = qkv.data_ptr() + b * S_new * 3 * row_b
            hk.copy_2d(kc.data_ptr() + (b * max_ctx + ctx) * row_b, row_b, src + row_b, 3 * row_b, row_b, S_new)
            hk.copy_2d(vc.data_ptr() + (b * max_ctx + ctx) * row_b, row_b, src + 2 * row_b, 3 * row_b, row_b, S_new)
        o = torch.empty((M, d), device=self.device, dtype=torch.bfloat16)
        hk.attn_fwd(qkv[:, :d], kc, vc, o, None, desc, B, H, hd, S_new, 1 << 30, hk.pad64(S_new), True, 1.0 / math.sqrt(hd), key_mask=kmask)
        x = hk.gemm_nt(o, L["o_w"], residual=x)
        h = hk.rmsnorm_fwd(x, L["ln2_w"], self.eps, out=h)
        act = hk.swiglu_fwd(hk.gemm_nt(h, L["gu_w"]), ff)
        return hk.gemm_nt(act, L["down_w"], residual=x)

    def quantize_fp8(self):
        """e4m3 copies (per-output-row scale) of every LLaMA linear for the decode step: 6.7 GB instead of 13.5 GB per token."""
        for L in self.p["layers"]:
            for k in ("qkv_w", "o_w", "gu_w", "down_w"):
                L[k + "8"], L[k + "8s"] = hk.quant_fp8_rows(L[k])
        self.p["lm_head8"], self.p["lm_head8s"] = hk.quant_fp8_rows(self.p["lm_head"])

    def pack_fp8_decode(self):
        """Decode-only copies of the e4m3 weights in the MFMA GEMV's operand order (hk.repack_fp8_mfma): the batch-1 weight stream then
        reads consecutive 1-KiB lines instead of 64-B segments of 16 strided rows (+6.7 GB of HBM next to the row-major copies that
        the prefill / training GEMMs use)."""
        if "qkv_w8" not in self.p["layers"][0]:
            self.quantize_fp8()
        for L in self.p["layers"]:
            for k in ("qkv_w", "o_w", "gu_w", "down_w"):
                L[k + "8p"] = hk.repack_fp8_mfma(L[k + "8"])
        self.p["lm_head8p"] = hk.repack_fp8_mfma(self.p["lm_head8"])

    def pack_bf16_decode(self):
        """Decode-only copies of the bf16 weights in the batched MFMA GEMV's operand order (hk.repack_bf16_mfma; +13.5 GB of HBM): from
        batch 2 the shared weight stream of generate() reads consecutive 1-KiB lines instead of 64-B segments of 16 strided rows."""
        for L in self.p["layers"]:
            for k in ("qkv_w", "o_w", "gu_w", "down_w"):
                L[k + "p"] = hk.repack_bf16_mfma(L[k])
        self.p["lm_headp"] = hk.repack_bf16_mfma(self.p["lm_head"])

    def quantize_base(self, bits: int = 8, scheme: str = "e4m3", quant_type: str = "nf4", double_quant: bool = True):
        """`bits: 8` of Config/multi_modal_stage{2,3}.yaml (text_modal.py:91-131: the reference loads the frozen LLaMA through bitsandbytes
        LLM.int8 for stages 2/3; lm_head stays 16-bit there and here).  Two schemes:

        * "int8" - the reference's arithmetic (what `UniBind.prepare_for_training` picks for the YAML key): every decoder linear is stored as
          int8 rows + absmax factors (vector-wise), the forward is bitsandbytes' MatMul8bitLt - int8 x int8 -> int32 on the MFMA with
          the activation's outlier columns (any |x| >= 6.0) taken out into a 16-bit product (`hk.int8_linear`) -, the backward multiplies
          with the DEquantised weight in 16 bit.  The bf16 weights of this object BECOME the dequantised ones (generate, merges and the dX
          GEMMs read them).  Parity: oracle/int8_oracle.py (bitsandbytes itself is not installed: unpinned against the package).
        * "e4m3" - MI355X-native fast path: OCP e4m3 copies with one fp32 scale per output row - of W for the forward product and of W^T for
          the dX product - on the 2x-rate block-scaled MFMA, activations / gradients quantised per row on the fly, no outlier split.
        LoRA adapters, norms, attention and the loss stay bf16 / fp32 in both.

        `bits: 4` (text_modal.py:91-107 `load_in_4bit`, `quant_type` nf4 | fp4, `double_quant`): bitsandbytes' Linear4bit stores 4-bit codes
        per block of 64 and computes every product on the DEquantised weight in the compute dtype, forward and backward.  Here each
        reference Linear (q, k, v, o, gate, up, down - the statistics of `double_quant` are per Linear) is quantised by `hk.quant4_blocks`,
        the codes and statistics stay beside the weight (`<name>q4`), and the bf16 weight BECOMES `hk.dequant4_blocks` of them: the ordinary
        bf16 kernels then run the arithmetic bitsandbytes runs.  Parity: oracle/nf4_oracle.py (unpinned against the package)."""
        self._merged_cache = None     # generate() must not answer from pre-quantisation merged weights
        if bits == 4:
            for L in self.p["layers"]:
                self._drop_derived(L)
                for k, parts in self._W4_PARTS.items():
                    W = L[k]
                    rows = W.shape[0] // parts
                    states = []
                    for i in range(parts):
                        sub = W[i * rows:(i + 1) * rows]
                        st = hk.quant4_blocks(sub, quant_type, double_quant)
                        hk.dequant4_blocks(st, out=sub)              # from here on the 16-bit weight IS the 4-bit one
                        states.append(st)
                    L[k + "q4"] = states
                    L[k + "T"] = hk.transpose(W)
            self.base8, self.base_int8, self.base4 = False, False, (quant_type, bool(double_quant))
            return self
        if bits not in (8, 16):
            raise NotImplementedError(f"bits={bits}: 16 (bf16), 8 (LLM.int8 / e4m3 base weights) or 4 (nf4 / fp4 storage)")
        if scheme not in ("e4m3", "int8"):
            raise ValueError(f"quantize_base scheme {scheme!r}: 'int8' (LLM.int8, the reference's) or 'e4m3'")
        self.base4 = None
        if bits == 16:
            self.base8 = self.base_int8 = False
            return self
        if scheme == "int8":
            for L in self.p["layers"]:
                self._drop_derived(L)
                for k in ("qkv_w", "o_w", "gu_w", "down_w"):
                    L[k + "i8"], L[k + "i8s"] = hk.quant_int8_rows(L[k])
                    hk.dequant_int8_rows(L[k + "i8"], L[k + "i8s"], out=L[k])     # from here on the 16-bit weight IS the int8 one
                    L[k + "T"] = hk.transpose(L[k])
            self._i8ws = hk.Int8Workspace(self.device, kmax=max(self.d, self.ff))
            self.base8, self.base_int8 = False, True
            return self
        self.quantize_fp8()
        for L in self.p["layers"]:
            for k in ("qkv_wT", "o_wT", "gu_wT", "down_wT"):
                L[k + "8"], L[k + "8s"] = hk.quant_fp8_rows(L[k])
        self.base8, self.base_int8 = True, False
        return self

    def _decode_session(self, B, max_ctx, caches, max_new, weights="bf16", kmask=None):
        """Static buffers + one captured hipGraph for the single-token step (batch <= 16): embedding gather, 32 x [RMSNorm,
        QKV GEMV, RoPE, KV append, attention over the cache, O GEMV + residual, RMSNorm, gate|up GEMV, SwiGLU, down GEMV +
        residual], final norm, lm_head GEMV -> fp32 logits.  Context length / positions live on the device
        (decode_advance), so the graph is captured once and replayed for every token."""
        dev, d, ff, H, hd, V = self.device, self.d, self.ff, self.heads, self.hd, self.vocab
        bf = torch.bfloat16
        s = types.SimpleNamespace()
        s.B, s.max_ctx, s.max_new, s.caches = B, max_ctx, max_new, caches
        s.x, s.x2, s.h, s.o = (torch.zeros((B, d), device=dev, dtype=bf) for _ in range(4))
        s.qkv = torch.zeros((B, 3 * d), device=dev, dtype=bf)
        s.gu = torch.zeros((B, 2 * ff), device=dev, dtype=bf)
        s.act = torch.zeros((B, ff), device=dev, dtype=bf)
        s.logits = torch.zeros((B, V), device=dev, dtype=torch.float32)
        s.next_ids = torch.zeros(B, device=dev, dtype=torch.int64)
        s.tok32 = torch.zeros(B, device=dev, dtype=torch.int32)
        s.out_ids = torch.zeros((B, max_new), device=dev, dtype=torch.int64)
        s.state = torch.zeros(4, device=dev, dtype=torch.int32)
        s.desc = torch.zeros((B, 8), device=dev, dtype=torch.int32)
        s.pos = torch.zeros(B, device=dev, dtype=torch.int32)
        scale = 1.0 / math.sqrt(hd)

        fp8 = weights == "fp8"
        if fp8 and "qkv_w8p" not in self.p["layers"][0]:
            self.pack_fp8_decode()

        packed16 = not fp8 and B >= 2 and d % 128 == 0 and ff % 128 == 0  # batched bf16: the MFMA GEMV on re-tiled weights
        if packed16 and "qkv_wp" not in self.p["layers"][0]:
            self.pack_bf16_decode()

        def W(L, name):  # (weight, per-row scale or None)
            return (L[name + "8p"], L[name + "8s"]) if fp8 else (L[name + "p"] if packed16 else L[name], None)

        batched = B >= 4 and not fp8  # the MFMA weight stream reads x from L2: norm / SwiGLU run once, not once per block (pays from batch 4)
        if batched:
            s.hn = torch.zeros((B, d), device=dev, dtype=bf)
            s.actb = torch.zeros((B, ff), device=dev, dtype=bf)
        if fp8:  # e4m3 weights AND activations on the block-scaled MFMA: static e4m3 operand buffers (graph capture: no allocation)
            s.x8 = (torch.zeros((B, d), device=dev, dtype=torch.uint8), torch.zeros(B, device=dev, dtype=torch.float32))
            s.a8 = (torch.zeros((B, ff), device=dev, dtype=torch.uint8), torch.zeros(B, device=dev, dtype=torch.float32))

        def lin(w, sc, x_in, out, K, pro=hk.PRO_NONE, norm_w=None, residual=None, out_f32=False):
            if fp8 and B <= 2:  # prologue + activation quantisation inside the GEMV: five launches per layer
                hk.gemv_fp8_mfma_fused(w, sc, x_in, out, K, prologue=pro, norm_w=norm_w, eps=self.eps, residual=residual, out_f32=out_f32)
                return
            if fp8:
                if pro == hk.PRO_RMSNORM:
                    _, q = hk.rmsnorm_fwd_q(x_in, norm_w, self.eps, want_bf16=False, q_out=s.x8)
                elif pro == hk.PRO_SWIGLU:
                    _, a8, sa = hk.swiglu_fwd_q(x_in, ff, q_out=s.a8)
                    q = (a8, sa)
                else:
                    q = hk.quant_fp8_rows(x_in, out=s.x8)
                hk.gemv_fp8_mfma(w, sc, q[0], q[1], out, residual=residual, out_f32=out_f32)
                return
            if batched and pro == hk.PRO_RMSNORM:
                hk.rmsnorm_fwd(x_in, norm_w, self.eps, out=s.hn)
                x_in, pro, norm_w = s.hn, hk.PRO_NONE, None
            elif batched and pro == hk.PRO_SWIGLU:
                hk.swiglu_fwd(x_in, ff, out=s.actb)
                x_in, pro = s.actb, hk.PRO_NONE
            hk.gemv_fused(w, x_in, out, K, wscale=sc, prologue=pro, norm_w=norm_w, eps=self.eps, residual=residual, out_f32=out_f32)

        # split-context attention (lhrs_decode_attn_split): 128-key slices, one workgroup each, so that a long context streams through
        # 32 * nsplit CUs; LHRS_DECODE_SPLIT=0 keeps the one-workgroup-per-head kernel (A/B runs)
        nsplit = min(16, -(-max_ctx // 128)) if os.environ.get("LHRS_DECODE_SPLIT", "1") != "0" else 1
        if nsplit > 1 and hd == 128:
            s.attn_part = torch.zeros((B, H, nsplit, 132), device=dev, dtype=torch.float32)
            s.attn_tickets = torch.zeros((B, H), device=dev, dtype=torch.int32)
            s.attn_cs = torch.zeros((B, 128), device=dev, dtype=torch.float32)   # cos | sin of the new position, written by decode_advance

        def enqueue():
            cs = getattr(s, "attn_cs", None)
            hk.decode_advance(s.state, s.desc, s.pos, B, max_ctx, 1, self.cos, self.sin, cs)
            hk.gather_rows(self.p["embed"], s.tok32, out=s.x)
            x, x2 = s.x, s.x2
            for L, (kc, vc) in zip(self.p["layers"], caches):
                lin(*W(L, "qkv_w"), x, s.qkv, d, hk.PRO_RMSNORM, L["ln1_w"])
                if hd == 128 and nsplit > 1:  # RoPE + KV append + attention over the cache in one launch, context split over workgroups
                    hk.decode_attn_split(s.qkv, kc, vc, self.cos, self.sin, s.pos, s.o, B, H, hd, max_ctx, scale, nsplit, s.attn_part,
                                         s.attn_tickets, key_mask=kmask, cs=cs)
                elif hd == 128:
                    hk.decode_attn(s.qkv, kc, vc, self.cos, self.sin, s.pos, s.o, B, H, hd, max_ctx, scale, key_mask=kmask)
                else:
                    hk.rope_kv_append(s.qkv, kc, vc, self.cos, self.sin, s.pos, B, H, hd, max_ctx)
                    hk.attn_fwd(s.qkv[:, :d], kc, vc, s.o, None, s.desc, B, H, hd, 1, 1 << 30, 64, True, scale, key_mask=kmask)
                lin(*W(L, "o_w"), s.o, x2, d, residual=x)
                lin(*W(L, "gu_w"), x2, s.gu, d, hk.PRO_RMSNORM, L["ln2_w"])
                lin(*W(L, "down_w"), s.gu, x, ff, hk.PRO_SWIGLU, residual=x2)
            w, sc = (self.p["lm_head8p"], self.p["lm_head8s"]) if fp8 else (self.p["lm_headp"] if packed16 else self.p["lm_head"], None)
            lin(w, sc, x, s.logits, d, hk.PRO_RMSNORM, self.p["norm_w"], out_f32=True)

        s.enqueue = enqueue
        s.graph = None
        return s

    def _lora_merged_layers(self):
        """Decoder layers with W + s B A in place of every adapted weight (copies; the base stays untouched) for generate() with un-merged
        adapters.  Cached together with the decode re-tilings / e4m3 copies that `_decode_session` hangs onto the dicts, keyed on the adapter
        store's `version` (bumped by every optimizer step and load): an evaluation loop between two optimizer steps merges ONCE."""
        lo = self.lora
        cache = getattr(self, "_merged_cache", None)
        if cache is not None and cache[0] == (id(lo), lo.version):
            return cache[1]
        wname = {"qkv": "qkv_w", "o": "o_w", "gu": "gu_w", "down": "down_w"}
        out = []
        for li, L in enumerate(self.p["layers"]):
            M = {k: L[k] for k in ("ln1_w", "ln2_w", "qkv_w", "o_w", "gu_w", "down_w")}
            for gname in lo.groups:
                W = L[wname[gname]]
                M[wname[gname]] = hk.gemm_nt(lo.derived[(li, gname, "Bfull")], lo.derived[(li, gname, "AT")], residual=W, alpha=lo.s)
            out.append(M)
        self._merged_cache = ((id(lo), lo.version), out)
        return out

    @torch.no_grad()
    def generate(self, input_ids, image_embedding=None, attention_mask=None, do_sample=False, temperature=1.0, top_p="default",
                 top_k="default", max_new_tokens=512, use_cache=True, stopping_criteria=None, streamer=None, eos_token_id="default",
                 return_logits=False, use_graph=True, weights="bf16", **_kw):
        """See `_generate`.  Two things happen here first: (1) `eos_token_id` defaults to the tokenizer's EOS, as HF `generate` stops on
        the generation config's EOS (pass None to disable); (2) if LoRA adapters are attached and not merged, the call runs on merged
        COPIES of the affected weights (`_lora_merged_layers`) and the base weights come back untouched."""
        if eos_token_id == "default":
            eos_token_id = getattr(self.tokenizer, "eos_token_id", None)
        # sampling defaults of the reference's callers: HF GenerationConfig top_k = 50 and the Llama-2 generation_config.json top_p = 0.9
        # apply when the caller (cli_qa.py:176-186 passes only temperature) does not say otherwise
        if top_k == "default":
            top_k = 50 if do_sample else None
        if top_p == "default":
            top_p = 0.9 if do_sample else None
        kw = dict(image_embedding=image_embedding, attention_mask=attention_mask, do_sample=do_sample, temperature=temperature, top_p=top_p,
                  top_k=top_k, max_new_tokens=max_new_tokens, use_cache=use_cache, stopping_criteria=stopping_criteria, streamer=streamer,
                  eos_token_id=eos_token_id, return_logits=return_logits, use_graph=use_graph, weights=weights)
        if self.lora is None:
            return self._generate(input_ids, **kw)
        base_layers, base8, base_i8 = self.p["layers"], self.base8, self.base_int8
        self.p["layers"], self.base8, self.base_int8 = self._lora_merged_layers(), False, False   # merged 16-bit copies: no 8-bit operands of them exist
        try:
            return self._generate(input_ids, **kw)
        finally:
            self.p["layers"], self.base8, self.base_int8 = base_layers, base8, base_i8

    def _generate(self, input_ids, image_embedding=None, attention_mask=None, do_sample=False, temperature=1.0, top_p=None,
                  top_k=None, max_new_tokens=512, use_cache=True, stopping_criteria=None, streamer=None, eos_token_id=None,
                  return_logits=False, use_graph=True, weights="bf16", **_kw):
        """TextModal.generate (text_modal.py:528-627): prefill over the spliced embeddings, then one token at a time with
        a KV cache; returns only the NEW token ids [B, n_new] (HF generate started from inputs_embeds).  Greedy
        (do_sample=False, the evaluation scripts' mode) runs entirely in HIP kernels; with do_sample=True the HIP-computed
        fp32 logits go through HF's temperature / top-k / top-p warpers and one multinomial draw per token.  The
        single-token step is a captured hipGraph over static buffers (batch <= 16; from batch 2 the weight stream runs on MFMA); without eos / stopping criteria / streamer
        the host never synchronises inside the loop."""
        if streamer is not None and getattr(streamer, "skip_prompt", False):
            streamer.put(input_ids.cpu())  # transformers.TextStreamer protocol: the first put() is the prompt, which skip_prompt drops
        embeds, _, mask, _ = self.prepare_inputs_for_multimodal(input_ids, attention_mask, None, image_embedding)
        B, S0, d = embeds.shape
        max_ctx = S0 + max_new_tokens
        if max_ctx > self.cos.shape[0]:
            raise ValueError(f"prompt ({S0}) + max_new_tokens ({max_new_tokens}) exceeds the {self.cos.shape[0]} positions of the RoPE table")
        dev = self.device
        kmask = None
        if mask is not None and not bool(mask.bool().all()):
            # batched evaluation with LEFT-padded prompts (DataCollatorForVGSupervisedDataset -> main_vqa.py:205-214): HF keeps
            # position_ids = arange (CustomLlamaForCausalLM.prepare_inputs_for_generation passes none) and hides the keys whose
            # (spliced) attention_mask is 0; every generated position is visible.
            kmask = torch.ones((B, max_ctx), device=dev, dtype=torch.uint8)
            kmask[:, :S0] = mask.to(device=dev, dtype=torch.uint8)
        caches = [(torch.empty((B * max_ctx, d), device=dev, dtype=torch.bfloat16), torch.empty((B * max_ctx, d), device=dev, dtype=torch.bfloat16))
                  for _ in range(len(self.p["layers"]))]

        def pick(logits):
            if not do_sample:
                return hk.argmax_rows(logits)
            return torch.multinomial(torch.softmax(warp_logits(logits, temperature, top_k, top_p), -1), 1).squeeze(1)

        # ---- prefill (GEMM path) -> logits of the last prompt position -> first new token
        desc = hk.make_desc([(b * S0, S0, b * max_ctx, S0, S0, 0) for b in range(B)], dev)
        x = embeds.reshape(B * S0, d)
        for L, cache in zip(self.p["layers"], caches):
            x = self._layer_step(L, x, B, S0, 0, cache, desc, max_ctx, kmask)
        hn = hk.rmsnorm_fwd(x.view(B, S0, d)[:, -1].contiguous(), self.p["norm_w"], self.eps)
        logits = hk.gemm_nt(hn, self.p["lm_head"], out_f32=True)  # [B, V] fp32 (HF: logits.float())
        all_logits = [logits.clone()] if return_logits else []
        nxt = pick(logits)

        host_checks = eos_token_id is not None or stopping_criteria is not None or streamer is not None
        finished = torch.zeros(B, dtype=torch.bool, device=dev)

        def host_step(ids_so_far, lg):  # -> stop?
            nonlocal finished
            if streamer is not None:
                streamer.put(ids_so_far[:, -1].cpu())
            if eos_token_id is not None:
                finished |= ids_so_far[:, -1] == eos_token_id
                if bool(finished.all()):
                    return True
            return stopping_criteria is not None and any(c(ids_so_far, lg) for c in stopping_criteria)

        n_done = 1
        if B <= 16:
            s = self._decode_session(B, max_ctx, caches, max_new_tokens, weights, kmask)
            s.state[0], s.state[1] = S0, 0
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                s.next_ids.copy_(nxt)
                hk.decode_emit(s.next_ids, s.tok32, s.out_ids, s.state, B, max_new_tokens)
                stop = host_checks and host_step(s.out_ids[:, :1], logits)
                while not stop and n_done < max_new_tokens:
                    if use_graph and s.graph is None and n_done >= 2:  # one eager step first (lazy kernel attributes), then capture
                        g = hk.HipGraph()
                        g.begin()
                        s.enqueue()
                        g.end()
                        s.graph = g
                        s.graph.launch()
                    elif s.graph is not None:
                        s.graph.launch()
                    else:
                        s.enqueue()
                    if return_logits:
                        all_logits.append(s.logits.clone())
                    if do_sample:
                        s.next_ids.copy_(pick(s.logits))
                    else:
                        hk.argmax_rows(s.logits, out=s.next_ids)
                    if eos_token_id is not None:
                        s.next_ids.copy_(torch.where(finished, torch.full_like(s.next_ids, self.tokenizer.pad_token_id), s.next_ids))
                    hk.decode_emit(s.next_ids, s.tok32, s.out_ids, s.state, B, max_new_tokens)
                    n_done += 1
                    if host_checks:
                        stop = host_step(s.out_ids[:, :n_done], s.logits)
            torch.cuda.current_stream().wait_stream(side)
            ids = s.out_ids[:, :n_done].clone()
        else:
            out_ids = [nxt]
            ctx = S0
            stop = host_checks and host_step(torch.stack(out_ids, 1), logits)
            while not stop and n_done < max_new_tokens:
                x = hk.gather_rows(self.p["embed"], out_ids[-1].clamp(0, self.vocab - 1).to(torch.int32))
                desc = hk.make_desc([(b, 1, b * max_ctx, ctx + 1, ctx + 1, ctx) for b in range(B)], dev)
                for L, cache in zip(self.p["layers"], caches):
                    x = self._layer_step(L, x, B, 1, ctx, cache, desc, max_ctx, kmask)
                logits = hk.gemm_nt(hk.rmsnorm_fwd(x, self.p["norm_w"], self.eps), self.p["lm_head"], out_f32=True)
                if return_logits:
                    all_logits.append(logits.clone())
                nxt = pick(logits)
                if eos_token_id is not None:
                    nxt = torch.where(finished, torch.full_like(nxt, self.tokenizer.pad_token_id), nxt)
                out_ids.append(nxt)
                ctx += 1
                n_done += 1
                if host_checks:
                    stop = host_step(torch.stack(out_ids, 1), logits)
            ids = torch.stack(out_ids, 1)
        if streamer is not None:
            streamer.end()
        return (ids, torch.stack(all_logits, 1)) if return_logits else ids

    # ------------------------------------------------------------------ backward (activation gradients only)
    def backward(self, loss_scale: float = 1.0, need_input_grad: bool = True, on_layer_ready=None):
        """d loss / d image_embedding [B, NI, d] bf16 (None when need_input_grad is False).  With LoRA enabled the adapter
        gradients of layer l are final when its backward finishes; on_layer_ready(l) lets the engine all-reduce them."""
        c = self._ctx
        assert c is not None, "decode(save_ctx=True) must precede backward"
        B, S, desc, LT = c["B"], c["S"], c["desc"], c["LT"]
        d, H, hd, ff, p = self.d, self.heads, self.hd, self.ff, self.p
        lo = self.lora
        M = B * S
        dhv = hk.gemm_nt(c["dlogits"], p["lm_headT"], alpha=loss_scale)
        tail = c.get("tail")
        if tail is None:
            dhid = torch.zeros((M, d), device=self.device, dtype=torch.bfloat16)
            hk.scatter_rows(dhv, c["rows"], dhid)
        else:
            dhid = dhv  # x_last holds exactly the supervised rows: the last layer's backward starts compact
        dxq = None
        if self.base8:
            dx, dxq = hk.rmsnorm_bwd_q(dhid, c["x_last"], p["norm_w"], None, eps=self.eps)
        else:
            dx = hk.rmsnorm_bwd(dhid, c["x_last"], p["norm_w"], None, eps=self.eps)
        scale = 1.0 / math.sqrt(hd)
        delta = torch.empty((B, H, LT), device=self.device, dtype=torch.float32)
        dqkv = torch.empty((M, 3 * d), device=self.device, dtype=torch.bfloat16)
        nl = len(p["layers"])
        for li in reversed(range(nl)):
            L, s = p["layers"][li], c["layers"][li]
            tl = tail if li == nl - 1 else None  # tail layer: dx, dgu, dh, dx_mid, do below have n rows until they are scattered back
            if self._native_layer(tl):
                dx, _ = hk.llama_layer_backward(dx, s, L, self.cos, self.sin, desc, B, S, LT, H, ff, self.eps, delta, dqkv)
                s.clear()
                if on_layer_ready is not None:
                    on_layer_ready(li)
                continue
            gu, qkv = s["gu"], s["qkv"]
            dmq = None
            act = hk.swiglu_fwd(gu, ff) if lo is not None and "down" in lo.groups else None      # x of the down projection
            dgu = self._down_bwd(li, dx, L["down_wT"], gu, act, s.get("T_down"), q8=self._q8(L, "down_wT"), dyq=dxq, drop=self._drop(s, "down"))
            dguq = None
            if self.base8:
                dgu, dguq = dgu
            h2 = hk.rmsnorm_fwd(s["x_mid"], L["ln2_w"], self.eps) if lo is not None and "gu" in lo.groups else None
            dh = self._lin_bwd(li, "gu", dgu, L["gu_wT"], h2, s.get("T_gu"), q8=self._q8(L, "gu_wT"), dyq=dguq, drop=self._drop(s, "gu"))
            if self.base8:
                dx_mid, dmq = hk.rmsnorm_bwd_q(dh, s["x_mid"], L["ln2_w"], None, add=dx, eps=self.eps, out=dh)
            else:
                dx_mid = hk.rmsnorm_bwd(dh, s["x_mid"], L["ln2_w"], None, add=dx, eps=self.eps, out=dh)
            do = self._lin_bwd(li, "o", dx_mid, L["o_wT"], s["o"], s.get("T_o"), q8=self._q8(L, "o_wT"), dyq=dmq, drop=self._drop(s, "o"))
            adesc, max_q = desc, S
            if tl is not None:
                rows_t, adesc, max_q = tl
                do = hk.scatter_rows(do, rows_t, torch.empty((M, d), device=self.device, dtype=torch.bfloat16))   # only query rows are read
                dx_mid = hk.scatter_rows(dx_mid, rows_t, torch.zeros((M, d), device=self.device, dtype=torch.bfloat16))  # residual path: 0 elsewhere
                dqkv[:, :d].zero_()                                                                                # dq exists for the query rows only
            # delta = rowsum(dO * O) inside the dQ kernel (one launch and one pass over O / dO less), inverse RoPE in the dq / dk stores
            hk.attn_bwd_o(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], do, s["o_full"], s["lse"], delta, dqkv[:, :d], dqkv[:, d:2 * d],
                          dqkv[:, 2 * d:], adesc, B, H, hd, max_q, S, LT, True, scale, rope=(self.cos, self.sin, S, 0))
            h1 = hk.rmsnorm_fwd(s["x_in"], L["ln1_w"], self.eps) if lo is not None and "qkv" in lo.groups else None
            dh1 = self._lin_bwd(li, "qkv", dqkv, L["qkv_wT"], h1, s.get("T_qkv"), q8=self._q8(L, "qkv_wT"), drop=self._drop(s, "qkv"))
            if self.base8:
                dx, dxq = hk.rmsnorm_bwd_q(dh1, s["x_in"], L["ln1_w"], None, add=dx_mid, eps=self.eps, out=dh1)
            else:
                dx = hk.rmsnorm_bwd(dh1, s["x_in"], L["ln1_w"], None, add=dx_mid, eps=self.eps, out=dh1)
            s.clear()
            if on_layer_ready is not None:
                on_layer_ready(li)
        d_image = None
        if need_input_grad and c.get("img_inv") is not None:      # general splice: rows of the image slots through the inverse map
            inv, n_slots = c["img_inv"]
            if inv.numel() < n_slots * c["NI"]:                    # slots past the last one the batch took: zero gradient
                inv = torch.cat([inv, torch.full((n_slots * c["NI"] - inv.numel(),), -1, device=inv.device, dtype=torch.int32)])
            d_image = hk.splice_map_bwd(dx.view(B, S, d), inv, n_slots, c["NI"])
        elif need_input_grad:
            d_image = hk.splice_bwd(dx.view(B, S, d), c["img_pos"], c["NI"])
        self._ctx = None
        return d_image
