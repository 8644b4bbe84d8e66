"""Checkpoint formats of the reference: key-name maps between the engine's fused parameter layout and

  * HF `CLIPVisionModel` state dicts, as stored in `FINAL.pt["rgb_ckpt"]` under the `encoder.` prefix of `VisionModal`
    (/root/reference lhrs/models/UniBind.py:68-81, 275-302; lhrs/models/rgb_vision_modal.py:154-157);
  * HF `LlamaForCausalLM` checkpoints (config.text.path; `CustomLlamaForCausalLM.from_pretrained`, lhrs/models/text_modal.py:79-131):
    `model.layers.N.self_attn.{q,k,v,o}_proj.weight`, `mlp.{gate,up,down}_proj.weight`, norms, embed_tokens, lm_head;
  * peft LoRA adapter directories `TextLoRA/` (lhrs/models/UniBind.py:74-79, 105-115): `adapter_config.json` +
    `base_model.model.model.layers.N.<module>.lora_{A,B}.weight`.

Pure host-side tensor bookkeeping (no arithmetic): fused q|k|v and gate|up rows are concatenations of the HF tensors.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterable, Optional

import torch

PROJ_MODULE = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj",
               "gate": "mlp.gate_proj", "up": "mlp.up_proj", "down": "mlp.down_proj"}


# ------------------------------------------------------------------------------------------------ CLIP ViT
def _vit_prefix(keys: Iterable[str]) -> str:
    for pre in ("encoder.vision_model.", "vision_model.", "encoder.", ""):
        if any(k == pre + "embeddings.class_embedding" for k in keys):
            return pre
    raise KeyError("no CLIP vision keys found (looked for [encoder.][vision_model.]embeddings.class_embedding)")


def vit_from_hf(sd: Dict[str, torch.Tensor]) -> Dict:
    pre = _vit_prefix(sd.keys())
    g = lambda k: sd[pre + k].float()  # noqa: E731
    p = {"patch_w": g("embeddings.patch_embedding.weight"), "cls": g("embeddings.class_embedding"),
         "pos": g("embeddings.position_embedding.weight"), "pre_ln_w": g("pre_layrnorm.weight"), "pre_ln_b": g("pre_layrnorm.bias"),
         "layers": []}
    l = 0
    while pre + f"encoder.layers.{l}.layer_norm1.weight" in sd:
        b = f"encoder.layers.{l}."
        p["layers"].append({
            "ln1_w": g(b + "layer_norm1.weight"), "ln1_b": g(b + "layer_norm1.bias"),
            "qkv_w": torch.cat([g(b + f"self_attn.{n}_proj.weight") for n in "qkv"], 0),
            "qkv_b": torch.cat([g(b + f"self_attn.{n}_proj.bias") for n in "qkv"], 0),
            "o_w": g(b + "self_attn.out_proj.weight"), "o_b": g(b + "self_attn.out_proj.bias"),
            "ln2_w": g(b + "layer_norm2.weight"), "ln2_b": g(b + "layer_norm2.bias"),
            "fc1_w": g(b + "mlp.fc1.weight"), "fc1_b": g(b + "mlp.fc1.bias"), "fc2_w": g(b + "mlp.fc2.weight"), "fc2_b": g(b + "mlp.fc2.bias")})
        l += 1
    return p


def vit_to_hf(p: Dict, prefix: str = "encoder.vision_model.") -> Dict[str, torch.Tensor]:
    """Engine layout -> the key names of FINAL.pt["rgb_ckpt"] (transformers 4.36.1: `encoder.vision_model.*`)."""
    dim = p["cls"].numel()
    sd = {prefix + "embeddings.class_embedding": p["cls"], prefix + "embeddings.patch_embedding.weight": p["patch_w"],
          prefix + "embeddings.position_embedding.weight": p["pos"], prefix + "pre_layrnorm.weight": p["pre_ln_w"],
          prefix + "pre_layrnorm.bias": p["pre_ln_b"]}
    for l, L in enumerate(p["layers"]):
        b = f"{prefix}encoder.layers.{l}."
        for i, n in enumerate("qkv"):
            sd[b + f"self_attn.{n}_proj.weight"] = L["qkv_w"][i * dim:(i + 1) * dim]
            sd[b + f"self_attn.{n}_proj.bias"] = L["qkv_b"][i * dim:(i + 1) * dim]
        sd[b + "self_attn.out_proj.weight"] = L["o_w"]; sd[b + "self_attn.out_proj.bias"] = L["o_b"]
        sd[b + "layer_norm1.weight"] = L["ln1_w"]; sd[b + "layer_norm1.bias"] = L["ln1_b"]
        sd[b + "layer_norm2.weight"] = L["ln2_w"]; sd[b + "layer_norm2.bias"] = L["ln2_b"]
        sd[b + "mlp.fc1.weight"] = L["fc1_w"]; sd[b + "mlp.fc1.bias"] = L["fc1_b"]
        sd[b + "mlp.fc2.weight"] = L["fc2_w"]; sd[b + "mlp.fc2.bias"] = L["fc2_b"]
    return sd


# ------------------------------------------------------------------------------------------------ LLaMA
def llama_from_hf(sd: Dict[str, torch.Tensor], n_layers: Optional[int] = None) -> Dict:
    g = lambda k: sd[k]  # noqa: E731
    p = {"embed": g("model.embed_tokens.weight"), "norm_w": g("model.norm.weight"), "lm_head": g("lm_head.weight"), "layers": []}
    l = 0
    while f"model.layers.{l}.input_layernorm.weight" in sd and (n_layers is None or l < n_layers):
        b = f"model.layers.{l}."
        p["layers"].append({
            "ln1_w": g(b + "input_layernorm.weight"),
            "qkv_w": torch.cat([g(b + f"self_attn.{n}_proj.weight") for n in "qkv"], 0),
            "o_w": g(b + "self_attn.o_proj.weight"),
            "ln2_w": g(b + "post_attention_layernorm.weight"),
            "gu_w": torch.cat([g(b + "mlp.gate_proj.weight"), g(b + "mlp.up_proj.weight")], 0),
            "down_w": g(b + "mlp.down_proj.weight")})
        l += 1
    return p


def llama_to_hf(p: Dict) -> Dict[str, torch.Tensor]:
    dim = p["norm_w"].numel()
    sd = {"model.embed_tokens.weight": p["embed"], "model.norm.weight": p["norm_w"], "lm_head.weight": p["lm_head"]}
    for l, L in enumerate(p["layers"]):
        b = f"model.layers.{l}."
        ff = L["gu_w"].shape[0] // 2
        for i, n in enumerate("qkv"):
            sd[b + f"self_attn.{n}_proj.weight"] = L["qkv_w"][i * dim:(i + 1) * dim]
        sd[b + "self_attn.o_proj.weight"] = L["o_w"]
        sd[b + "input_layernorm.weight"] = L["ln1_w"]; sd[b + "post_attention_layernorm.weight"] = L["ln2_w"]
        sd[b + "mlp.gate_proj.weight"] = L["gu_w"][:ff]; sd[b + "mlp.up_proj.weight"] = L["gu_w"][ff:]
        sd[b + "mlp.down_proj.weight"] = L["down_w"]
    return sd


def load_hf_dir(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of a HuggingFace checkpoint directory (sharded *.safetensors or pytorch_model*.bin)."""
    sd: Dict[str, torch.Tensor] = {}
    st = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f))
        return sd
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
    for f in bins:
        sd.update(torch.load(f, map_location="cpu"))
    return sd


# ------------------------------------------------------------------------------------------------ peft LoRA
def lora_to_peft(store) -> Dict[str, torch.Tensor]:
    """LoraStore -> the tensors of a peft adapter (`adapter_model.bin` key names of peft 0.7.1)."""
    sd = {}
    for l in range(store.nl):
        for proj in store.targets:
            A, B = store.get_adapter(l, proj)
            base = f"base_model.model.model.layers.{l}.{PROJ_MODULE[proj]}"
            sd[base + ".lora_A.weight"] = A.detach().cpu().clone()
            sd[base + ".lora_B.weight"] = B.detach().cpu().contiguous().clone()
    return sd


def lora_from_peft(store, sd: Dict[str, torch.Tensor]) -> None:
    for l in range(store.nl):
        for proj in store.targets:
            base = f"base_model.model.model.layers.{l}.{PROJ_MODULE[proj]}"
            keyA = next((k for k in (base + ".lora_A.weight", base + ".lora_A.default.weight") if k in sd), None)
            if keyA is None:
                raise KeyError(f"adapter for {base} missing")
            store.set_adapter(l, proj, sd[keyA].float(), sd[keyA.replace("lora_A", "lora_B")].float())
    store.refresh()


def save_peft_dir(store, path: str, base_model: str = "") -> None:
    os.makedirs(path, exist_ok=True)
    torch.save(lora_to_peft(store), os.path.join(path, "adapter_model.bin"))
    cfg = {"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": store.r, "lora_alpha": store.s * store.r, "lora_dropout": float(getattr(store, "dropout", 0.0)),
           "bias": "none", "target_modules": sorted({PROJ_MODULE[t].split(".")[-1] for t in store.targets}),
           "base_model_name_or_path": base_model, "inference_mode": False}
    with open(os.path.join(path, "adapter_config.json"), "w") as f:
        json.dump(cfg, f, indent=1)


def load_peft_dir(path: str):
    cfg = json.load(open(os.path.join(path, "adapter_config.json")))
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "adapter_model.bin"), map_location="cpu")
    inv = {v.split(".")[-1]: k for k, v in PROJ_MODULE.items()}
    targets = tuple(t for t in ("q", "k", "v", "o", "gate", "up", "down") if PROJ_MODULE[t].split(".")[-1] in cfg["target_modules"])
    assert all(m in inv for m in cfg["target_modules"] if m != "lm_head"), cfg["target_modules"]
    return cfg, targets, sd
