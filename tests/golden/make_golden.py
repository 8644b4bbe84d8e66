"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from /root/reference with the shim
recipe of SURVEY.md Appendix A) on seeded parameters from oracle/params.py.

Run in the build container only (`python tests/golden/make_golden.py`); /root/reference does not exist on the GPU
box, where the tests read the committed .npz files.  Fixtures hold data only (inputs, expected outputs).
The parameters are regenerated from their seeds by the tests, so they are not stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import params as OP  # noqa: E402

torch.set_num_threads(8)


def f16(x):
    return x.detach().to(torch.float16).numpy()


# ------------------------------------------------------------------------------------------- 1. AttnPooler
def golden_pooler():
    spec = importlib.util.spec_from_file_location("common_arch", f"{REF}/lhrs/models/common_arch.py")
    ca = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ca)
    pool = ca.AttnPooler(num_query=144, num_layers=6, num_attention_heads=16, encoder_hidden_size=1024, hidden_size=1024,
                         output_size=4096, norm_layer=ca.LayerNorm)
    p = OP.make_pooler_params(seed=1)
    missing = pool.load_state_dict(OP.pooler_to_ref(p), strict=True)
    print("pooler load:", missing)
    g = torch.Generator().manual_seed(101)
    x = torch.randn(2, 768, 1024, generator=g)
    dout = torch.randn(2, 144, 4096, generator=g) * 0.01
    x.requires_grad_(True)
    out = pool(x)
    out.backward(dout)
    grads = {n: q.grad for n, q in pool.named_parameters()}
    np.savez_compressed(
        os.path.join(HERE, "pooler.npz"), input_seed=np.array(101), out=f16(out), dx_rows=f16(x.grad[:, ::8]),
        grad_names=np.array(list(grads.keys())),
        grad_norms=np.array([grads[n].norm().item() for n in grads], dtype=np.float64),
        g_query=f16(grads["query"][0]), g_out_proj_b=f16(grads["out_proj.bias"]),
        g_l0_in_w_slice=f16(grads["layers.0.attn.in_proj_weight"][::64, ::64]),
        g_l5_fc_w_slice=f16(grads["layers.5.mlp.c_fc.weight"][::64, ::64]),
        g_l3_ln1kv_w=f16(grads["layers.3.ln_1_kv.weight"]),
    )
    print("pooler golden: out", tuple(out.shape), "|out|", out.norm().item())


# ------------------------------------------------------------------------------------------- shims for lhrs.models
def import_reference_models():
    import transformers  # noqa: F401  (must precede the deepspeed stub)
    from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig

    for name, path in [("lhrs", f"{REF}/lhrs"), ("lhrs.Dataset", f"{REF}/lhrs/Dataset")]:
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    class ConfigDict(dict):
        def __init__(self, d=None):
            super().__init__()
            for k, v in (d or {}).items():
                self[k] = ConfigDict(v) if isinstance(v, dict) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        __setattr__ = dict.__setitem__

    mc = types.ModuleType("ml_collections")
    mc.ConfigDict = ConfigDict
    mc.config_dict = types.ModuleType("ml_collections.config_dict")
    mc.config_dict.ConfigDict = ConfigDict
    z = types.ModuleType("deepspeed.utils.zero_to_fp32")
    z.get_fp32_state_dict_from_zero_checkpoint = z.load_state_dict_from_zero_checkpoint = None
    tk = types.ModuleType("transformers.models.llama.tokenization_llama_fast")
    tk.LlamaTokenizerFast = type("LlamaTokenizerFast", (), {})
    pf = types.ModuleType("peft")
    pf.PeftModel = type("PeftModel", (), {})
    sys.modules.update({"ml_collections": mc, "ml_collections.config_dict": mc.config_dict,
                        "deepspeed": types.ModuleType("deepspeed"), "deepspeed.utils": types.ModuleType("deepspeed.utils"),
                        "deepspeed.utils.zero_to_fp32": z, tk.__name__: tk, "peft": pf})
    cfg = ConfigDict(yaml.safe_load(open(f"{REF}/Config/multi_modal_stage1.yaml")))
    cfg.use_checkpoint = False
    cfg.is_distribute = False
    cfg.dtype = "float32"
    cfg.fp16 = False
    return cfg, CLIPVisionConfig, CLIPVisionModel, LlamaConfig


def golden_unibind(n_llama_layers=2):
    cfg, CLIPVisionConfig, CLIPVisionModel, LlamaConfig = import_reference_models()
    CLIPVisionModel.from_pretrained = staticmethod(lambda name, **kw: CLIPVisionModel(CLIPVisionConfig(
        hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
        patch_size=14, hidden_act="quick_gelu")))
    import lhrs.models.text_modal as tm

    tm.CustomLlamaForCausalLM.from_pretrained = staticmethod(lambda path, **kw: tm.CustomLlamaForCausalLM(LlamaConfig(
        vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=n_llama_layers,
        num_attention_heads=32, hidden_act="silu", max_position_embeddings=2048, rms_norm_eps=1e-5, pad_token_id=0,
        bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)))

    class FakeTok:
        unk_token_id = pad_token_id = 0
        bos_token_id = 1
        model_max_length = 2048

        def __len__(self):
            return 32000

    tm.LlamaTokenizerFast.from_pretrained = staticmethod(lambda n: FakeTok())
    from lhrs.models import build_model

    model = build_model(cfg, activate_modal=("rgb", "text"))
    model.prepare_for_training(freeze_vision=True, freeze_text=True, tune_rgb_pooler=True, model_path=None,
                               tune_im_start=False, compute_dtype=torch.float32)
    for q in model.text.parameters():
        q.requires_grad = False  # BASELINE "projector-only"
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=n_llama_layers)}
    enc_keys = model.rgb.encoder.state_dict().keys()
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in enc_keys) else ""
    r = model.rgb.encoder.load_state_dict(OP.vit_to_hf(P["vit"], prefix), strict=False)
    print("vit load: missing", [k for k in r.missing_keys if "post_layernorm" not in k and "position_ids" not in k], "unexpected", r.unexpected_keys)
    print("pooler load:", model.rgb_pooler.load_state_dict(OP.pooler_to_ref(P["pooler"]), strict=True))
    r = model.text.text_encoder.load_state_dict(OP.llama_to_hf(P["llama"]), strict=False)
    print("llama load: missing", [k for k in r.missing_keys if "rotary" not in k], "unexpected", r.unexpected_keys)

    # ---- (a) splice edge cases straight through the reference method (int tensors, bit-exact) ----
    model.text.tune_pooler = False
    cases = {}
    NI = 4

    def run_splice(name, ids, labels, mask):
        img = torch.arange(ids.shape[0] * NI * 4096, dtype=torch.float32).reshape(ids.shape[0], NI, 4096) * 1e-6 + 0.5
        _, new_mask, _, embeds, new_labels = model.text.prepare_inputs_for_multimodal(
            input_ids=ids, attention_mask=mask, labels=labels, past_key_values=None, image_embedding=img)
        # identify every output row: token id it copies, or image row, or pad
        emb = model.text.get_text_encoder().model.embed_tokens.weight
        B, S = embeds.shape[:2]
        kind = torch.full((B, S), -10 ** 9, dtype=torch.int64)
        for b in range(B):
            for j in range(S):
                row = embeds[b, j]
                hit = (img[b] == row).all(-1).nonzero()
                if hit.numel():
                    kind[b, j] = -(1 + int(hit[0]))
                elif row.abs().sum() == 0:
                    kind[b, j] = -10 ** 9
                else:
                    cand = [t for t in range(ids.shape[1]) if ids[b, t] >= 0 and torch.equal(emb[ids[b, t]], row)]
                    kind[b, j] = cand[0] if len(cand) == 1 else -5  # -5: ambiguous (repeated id); resolved by caller
        cases[name + "_ids"] = ids.numpy(); cases[name + "_labels"] = labels.numpy(); cases[name + "_mask"] = mask.numpy()
        cases[name + "_src"] = kind.numpy(); cases[name + "_new_labels"] = new_labels.numpy()
        cases[name + "_new_mask"] = new_mask.numpy()

    g = torch.Generator().manual_seed(7)

    def mk(B, T, img_pos, pad_from=None):
        ids = torch.stack([torch.randperm(30000, generator=g)[:T] + 3 for _ in range(B)])  # unique ids per row
        ids[:, 0] = 1
        for b, p in enumerate(img_pos):
            if p is not None:
                ids[b, p] = -200
        if pad_from is not None:
            for b, pf_ in enumerate(pad_from):
                if pf_ is not None:
                    ids[b, pf_:] = 0
        labels = ids.clone()
        labels[:, :2] = -100
        labels[ids == 0] = -100
        return ids, labels, ids.ne(0)

    run_splice("uniform", *mk(2, 6, [1, 1]))
    run_splice("ragged_pad", *mk(3, 9, [1, 1, 1], pad_from=[None, 5, 7]))
    run_splice("mixed_noimg", *mk(3, 8, [1, None, 3]))
    run_splice("img_last", *mk(2, 5, [4, 1]))
    run_splice("single", *mk(1, 3, [1]))
    np.savez_compressed(os.path.join(HERE, "splice.npz"), n_img_tokens=np.array(NI), **cases)
    print("splice golden:", [k for k in cases if k.endswith("_src")])

    # ---- (b) end to end: UniBind.forward + backward on the projector ----
    gg = torch.Generator().manual_seed(202)
    B, T = 2, 34
    ids = torch.randint(3, 32000, (B, T), generator=gg)
    ids[:, 0] = 1
    ids[:, 1] = -200
    ids[1, 28:] = 0  # right padding on sample 1
    labels = ids.clone()
    labels[:, :2] = -100
    labels[ids == 0] = -100
    batch = dict(rgb=torch.randn(B, 3, 224, 224, generator=gg), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    taps = {}
    h1 = model.rgb_pooler.register_forward_hook(lambda m, i, o: taps.update(vit_taps=i[0].detach(), image=o))
    h2 = model.text.text_encoder.model.register_forward_hook(lambda m, i, o: taps.update(hidden=o[0].detach()))
    out = model(batch)
    taps["image"].retain_grad()
    loss = out["total_loss"]
    loss.backward()
    h1.remove(); h2.remove()
    grads = {n: q.grad for n, q in model.rgb_pooler.named_parameters()}
    print("e2e loss", loss.item(), "text_loss", out["text_loss"].item())
    np.savez_compressed(
        os.path.join(HERE, "unibind_e2e.npz"), rgb=f16(batch["rgb"]), input_ids=ids.numpy(), labels=labels.numpy(),
        attention_mask=batch["attention_mask"].numpy(), loss=np.array(loss.item(), dtype=np.float64),
        vit_taps=f16(taps["vit_taps"][:1, ::2]), image=f16(taps["image"][:, ::2]),
        d_image=taps["image"].grad.detach().numpy().astype(np.float32)[:, ::4],
        hidden_sample=f16(taps["hidden"][:, ::8, :]),
        grad_names=np.array(list(grads.keys())),
        grad_norms=np.array([grads[n].norm().item() for n in grads], dtype=np.float64),
        g_query=grads["query"][0].numpy().astype(np.float32)[::4], g_out_proj_b=grads["out_proj.bias"].numpy().astype(np.float32),
        n_llama_layers=np.array(n_llama_layers),
    )


if __name__ == "__main__":
    golden_pooler()
    golden_unibind()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
