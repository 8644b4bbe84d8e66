"""Reference checkpoint formats on the engine: FINAL.pt + TextLoRA/ round trip, HF LLaMA directory loading, LoRA merge."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from lhrs_bot_amd import checkpoint as C  # noqa: E402
from lhrs_bot_amd.unibind import UniBind  # noqa: E402
from oracle import params as OP  # noqa: E402

DEV = "cuda"


def batch(seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 32000, (2, 14), generator=g)
    ids[:, 0] = 1
    ids[:, 1] = -200
    labels = ids.clone()
    labels[:, :2] = -100
    return dict(rgb=torch.randn(2, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))


@pytest.mark.timeout(900)
def test_final_pt_textlora_roundtrip_hf_dir_and_merge(tmp_path):
    from safetensors.torch import save_file

    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=1)}
    # a HuggingFace-style checkpoint directory for the LLaMA (sharded safetensors + config.json), loaded through from_pretrained
    hf = {k: v.contiguous() for k, v in C.llama_to_hf(P["llama"]).items()}
    keys = sorted(hf)
    d = tmp_path / "llama"
    d.mkdir()
    save_file({k: hf[k] for k in keys[: len(keys) // 2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: hf[k] for k in keys[len(keys) // 2:]}, str(d / "model-00002-of-00002.safetensors"))
    json.dump(dict(hidden_size=4096, intermediate_size=11008, num_attention_heads=32, num_hidden_layers=1, rms_norm_eps=1e-5), open(d / "config.json", "w"))
    m1 = UniBind(("rgb", "text"), None, device=DEV, llama_layers=1)
    m1.rgb.load_params(P["vit"]); m1.rgb_pooler.load_params(P["pooler"]); m1.text.from_pretrained(str(d))
    ref = UniBind(("rgb", "text"), None, device=DEV, llama_layers=1).load_params(P)
    assert torch.equal(m1.text.p["layers"][0]["gu_w"], ref.text.p["layers"][0]["gu_w"]) and torch.equal(m1.text.p["lm_head"], ref.text.p["lm_head"])
    # adapters with non-zero B, then FINAL.pt + TextLoRA/
    lora = m1.enable_lora(r=8, alpha=16, targets=("q", "v", "down"))
    g = torch.Generator().manual_seed(1)
    for pr in ("q", "v", "down"):
        A, B = lora.get_adapter(0, pr)
        lora.set_adapter(0, pr, A.cpu(), torch.randn(B.shape, generator=g) * 0.02)
    lora.refresh()
    m1.eval()
    loss1 = m1(batch())["total_loss"].item()
    out = tmp_path / "ck"
    ck = m1.custom_save_checkpoint(str(out))
    assert set(ck["other_ckpt"]) == {"rgb_pooler", "text_proj", "embed_tokens", "lm_head"}
    assert "encoder.vision_model.encoder.layers.23.mlp.fc2.weight" in ck["rgb_ckpt"]      # all 24 layers go back out
    cfg = json.load(open(out / "TextLoRA" / "adapter_config.json"))
    assert cfg["r"] == 8 and sorted(cfg["target_modules"]) == ["down_proj", "q_proj", "v_proj"]
    # load into a fresh model: stage 3 keeps trainable adapters, stage 0 merges them into the base weights
    m2 = UniBind(("rgb", "text"), None, device=DEV, llama_layers=1)
    m2.stage = 3
    m2.rgb_pooler.init_random(5); m2.text.load_params(P["llama"])
    m2.custom_load_state_dict(str(out / "FINAL.pt"))
    assert m2.text.lora is not None and torch.equal(m2.text.lora.master, lora.master) and torch.equal(m2.rgb_pooler.master, m1.rgb_pooler.master)
    m2.eval()
    assert m2(batch())["total_loss"].item() == loss1
    m3 = UniBind(("rgb", "text"), None, device=DEV, llama_layers=1)
    m3.stage = 0
    m3.text.load_params(P["llama"])
    m3.custom_load_state_dict(str(out / "FINAL.pt"))
    assert m3.text.lora is None
    m3.eval()
    assert abs(m3(batch())["total_loss"].item() - loss1) < 2e-3 * loss1     # merged bf16 weights vs fused adapters
