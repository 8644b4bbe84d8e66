"""TEST INFRASTRUCTURE ONLY - deterministic synthetic parameters for the LHRS-Bot hot path.

No weights or tokenizer files exist offline, so parity runs use seeded random parameters in the ENGINE's own
layout.  `make_params` is machine-independent for a given torch build
(per-tensor CPU generators), so the golden fixtures only have to store inputs and expected outputs; the
converters below map the layout onto the state-dict keys of the reference modules and are used by
tests/golden/make_golden.py to load the very same numbers into the imported reference.
"""
from __future__ import annotations

import zlib
from typing import Dict

import torch


def _seed_of(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) + seed * 1000003) % (2 ** 31)


def _rand(name: str, shape, seed: int, std: float = 0.02, mean: float = 0.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_of(name, seed))
    return torch.randn(*shape, generator=g) * std + mean


def make_vit_params(seed=0, layers=24, dim=1024, ff=4096, patch=14, img=224) -> Dict:
    n_tok = (img // patch) ** 2 + 1
    p = {
        "patch_w": _rand("vit.patch_w", (dim, 3, patch, patch), seed),
        "cls": _rand("vit.cls", (dim,), seed),
        "pos": _rand("vit.pos", (n_tok, dim), seed),
        "pre_ln_w": _rand("vit.pre_ln_w", (dim,), seed, 0.05, 1.0),
        "pre_ln_b": _rand("vit.pre_ln_b", (dim,), seed, 0.02),
        "layers": [],
    }
    for l in range(layers):
        n = f"vit.{l}."
        p["layers"].append({
            "ln1_w": _rand(n + "ln1_w", (dim,), seed, 0.05, 1.0), "ln1_b": _rand(n + "ln1_b", (dim,), seed),
            "qkv_w": _rand(n + "qkv_w", (3 * dim, dim), seed), "qkv_b": _rand(n + "qkv_b", (3 * dim,), seed),
            "o_w": _rand(n + "o_w", (dim, dim), seed), "o_b": _rand(n + "o_b", (dim,), seed),
            "ln2_w": _rand(n + "ln2_w", (dim,), seed, 0.05, 1.0), "ln2_b": _rand(n + "ln2_b", (dim,), seed),
            "fc1_w": _rand(n + "fc1_w", (ff, dim), seed), "fc1_b": _rand(n + "fc1_b", (ff,), seed),
            "fc2_w": _rand(n + "fc2_w", (dim, ff), seed), "fc2_b": _rand(n + "fc2_b", (dim,), seed),
        })
    return p


def make_pooler_params(seed=0, layers=6, dim=1024, out_dim=4096, num_query=144) -> Dict:
    p = {"query": _rand("pool.query", (num_query, dim), seed), "layers": [],
         "out_proj_w": _rand("pool.out_proj_w", (out_dim, dim), seed), "out_proj_b": _rand("pool.out_proj_b", (out_dim,), seed)}
    for l in range(layers):
        n = f"pool.{l}."
        p["layers"].append({
            "ln1_w": _rand(n + "ln1_w", (dim,), seed, 0.05, 1.0), "ln1_b": _rand(n + "ln1_b", (dim,), seed),
            "ln1kv_w": _rand(n + "ln1kv_w", (dim,), seed, 0.05, 1.0), "ln1kv_b": _rand(n + "ln1kv_b", (dim,), seed),
            "in_w": _rand(n + "in_w", (3 * dim, dim), seed), "in_b": _rand(n + "in_b", (3 * dim,), seed),
            "out_w": _rand(n + "out_w", (dim, dim), seed), "out_b": _rand(n + "out_b", (dim,), seed),
            "ln2_w": _rand(n + "ln2_w", (dim,), seed, 0.05, 1.0), "ln2_b": _rand(n + "ln2_b", (dim,), seed),
            "fc_w": _rand(n + "fc_w", (4 * dim, dim), seed), "fc_b": _rand(n + "fc_b", (4 * dim,), seed),
            "proj_w": _rand(n + "proj_w", (dim, 4 * dim), seed), "proj_b": _rand(n + "proj_b", (dim,), seed),
        })
    return p


def make_llama_params(seed=0, layers=2, dim=4096, ff=11008, vocab=32000) -> Dict:
    p = {"embed": _rand("llama.embed", (vocab, dim), seed), "layers": [],
         "norm_w": _rand("llama.norm_w", (dim,), seed, 0.05, 1.0), "lm_head": _rand("llama.lm_head", (vocab, dim), seed)}
    for l in range(layers):
        n = f"llama.{l}."
        p["layers"].append({
            "ln1_w": _rand(n + "ln1_w", (dim,), seed, 0.05, 1.0),
            "qkv_w": _rand(n + "qkv_w", (3 * dim, dim), seed),   # rows: q | k | v
            "o_w": _rand(n + "o_w", (dim, dim), seed),
            "ln2_w": _rand(n + "ln2_w", (dim,), seed, 0.05, 1.0),
            "gu_w": _rand(n + "gu_w", (2 * ff, dim), seed),      # rows: gate | up
            "down_w": _rand(n + "down_w", (dim, ff), seed),
        })
    return p


# ---- converters to the reference modules' state-dict keys -------------------------------------------------
def vit_to_hf(p: Dict, prefix: str = "vision_model.") -> Dict[str, torch.Tensor]:
    """-> HF CLIPVisionModel state dict (transformers 4.36.1 keys carry the 'vision_model.' prefix)."""
    dim = p["cls"].numel()
    sd = {
        prefix + "embeddings.class_embedding": p["cls"],
        prefix + "embeddings.patch_embedding.weight": p["patch_w"],
        prefix + "embeddings.position_embedding.weight": p["pos"],
        prefix + "pre_layrnorm.weight": p["pre_ln_w"], prefix + "pre_layrnorm.bias": p["pre_ln_b"],
    }
    for l, L in enumerate(p["layers"]):
        b = f"{prefix}encoder.layers.{l}."
        for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[b + f"self_attn.{nm}.weight"] = L["qkv_w"][i * dim:(i + 1) * dim]
            sd[b + f"self_attn.{nm}.bias"] = L["qkv_b"][i * dim:(i + 1) * dim]
        sd[b + "self_attn.out_proj.weight"] = L["o_w"]; sd[b + "self_attn.out_proj.bias"] = L["o_b"]
        sd[b + "layer_norm1.weight"] = L["ln1_w"]; sd[b + "layer_norm1.bias"] = L["ln1_b"]
        sd[b + "layer_norm2.weight"] = L["ln2_w"]; sd[b + "layer_norm2.bias"] = L["ln2_b"]
        sd[b + "mlp.fc1.weight"] = L["fc1_w"]; sd[b + "mlp.fc1.bias"] = L["fc1_b"]
        sd[b + "mlp.fc2.weight"] = L["fc2_w"]; sd[b + "mlp.fc2.bias"] = L["fc2_b"]
    return sd


def pooler_to_ref(p: Dict) -> Dict[str, torch.Tensor]:
    """-> state dict of lhrs/models/common_arch.py:AttnPooler."""
    sd = {"query": p["query"][None], "out_proj.weight": p["out_proj_w"], "out_proj.bias": p["out_proj_b"]}
    for l, L in enumerate(p["layers"]):
        b = f"layers.{l}."
        sd[b + "ln_1.weight"] = L["ln1_w"]; sd[b + "ln_1.bias"] = L["ln1_b"]
        sd[b + "ln_1_kv.weight"] = L["ln1kv_w"]; sd[b + "ln_1_kv.bias"] = L["ln1kv_b"]
        sd[b + "attn.in_proj_weight"] = L["in_w"]; sd[b + "attn.in_proj_bias"] = L["in_b"]
        sd[b + "attn.out_proj.weight"] = L["out_w"]; sd[b + "attn.out_proj.bias"] = L["out_b"]
        sd[b + "ln_2.weight"] = L["ln2_w"]; sd[b + "ln_2.bias"] = L["ln2_b"]
        sd[b + "mlp.c_fc.weight"] = L["fc_w"]; sd[b + "mlp.c_fc.bias"] = L["fc_b"]
        sd[b + "mlp.c_proj.weight"] = L["proj_w"]; sd[b + "mlp.c_proj.bias"] = L["proj_b"]
    return sd


def llama_to_hf(p: Dict) -> Dict[str, torch.Tensor]:
    """-> HF LlamaForCausalLM state dict."""
    dim = p["norm_w"].numel()
    sd = {"model.embed_tokens.weight": p["embed"], "model.norm.weight": p["norm_w"], "lm_head.weight": p["lm_head"]}
    for l, L in enumerate(p["layers"]):
        b = f"model.layers.{l}."
        ff = L["gu_w"].shape[0] // 2
        for i, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[b + f"self_attn.{nm}.weight"] = L["qkv_w"][i * dim:(i + 1) * dim]
        sd[b + "self_attn.o_proj.weight"] = L["o_w"]
        sd[b + "input_layernorm.weight"] = L["ln1_w"]; sd[b + "post_attention_layernorm.weight"] = L["ln2_w"]
        sd[b + "mlp.gate_proj.weight"] = L["gu_w"][:ff]; sd[b + "mlp.up_proj.weight"] = L["gu_w"][ff:]
        sd[b + "mlp.down_proj.weight"] = L["down_w"]
    return sd


# ------------------------------------------------------------------------------------------------- LoRA adapters (stages 2/3)
LORA_DIMS = {"q": (4096, 4096), "k": (4096, 4096), "v": (4096, 4096), "o": (4096, 4096),        # proj -> (in_features, out_features)
             "gate": (4096, 11008), "up": (4096, 11008), "down": (11008, 4096)}


def make_lora_params(seed: int, layers: int, r: int, alpha: float, targets) -> list:
    """Per layer {"scale": alpha / r, proj: (A [r, in], B [out, r])}: A as peft initialises it (kaiming-uniform(a=sqrt 5) = U(-1/sqrt(in),
    1/sqrt(in)); text_modal.py:133-151), B NON-zero (N(0, 0.02); peft starts it at 0, which would leave dA = 0 untested)."""
    out = []
    for l in range(layers):
        d = {"scale": float(alpha) / float(r)}
        for pr in targets:
            fin, fout = LORA_DIMS[pr]
            g = torch.Generator().manual_seed(_seed_of(f"lora.{l}.{pr}", seed))
            A = (torch.rand(r, fin, generator=g) * 2 - 1) / fin ** 0.5
            B = torch.randn(fout, r, generator=g) * 0.02
            d[pr] = (A, B)
        out.append(d)
    return out


def merged_lora_weights(p: Dict, lora: list) -> Dict:
    """The decoder after peft `merge_and_unload()` (UniBind.custom_load_state_dict, UniBind.py:105-115): W' = W + s * B @ A on every
    adapted projection; a copy of `p` with merged stacked weights (q|k|v and gate|up stay stacked)."""
    q = dict(p)
    q["layers"] = []
    for L, lo in zip(p["layers"], lora):
        M = dict(L)
        dim, ff = L["o_w"].shape[0], L["gu_w"].shape[0] // 2
        delta = lambda pr: lo["scale"] * (lo[pr][1] @ lo[pr][0]) if pr in lo else None  # noqa: E731
        qkv = L["qkv_w"].clone()
        for i, pr in enumerate(("q", "k", "v")):
            if pr in lo:
                qkv[i * dim:(i + 1) * dim] += delta(pr)
        gu = L["gu_w"].clone()
        for i, pr in enumerate(("gate", "up")):
            if pr in lo:
                gu[i * ff:(i + 1) * ff] += delta(pr)
        M.update(qkv_w=qkv, gu_w=gu, o_w=L["o_w"] + delta("o") if "o" in lo else L["o_w"],
                 down_w=L["down_w"] + delta("down") if "down" in lo else L["down_w"])
        q["layers"].append(M)
    return q
