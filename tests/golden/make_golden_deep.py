"""tests/golden/unibind_e2e_8l.npz: the end-to-end forward + backward of the REFERENCE (UniBind.forward, projector-only) with EIGHT
LLaMA-7B-width decoder layers - a depth check on top of the 2-layer fixture of make_golden.py (error accumulation through the
residual stream, RoPE / causal attention at every depth).  rgb / ids are regenerated from seeds by the test.  Build container only
(~14 GB of fp32 parameters twice)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import params as OP  # noqa: E402
import make_golden as MG  # noqa: E402

torch.set_num_threads(8)
NL = 8


def main(NL=NL, T=40, ids_seed=404, rgb_seed=405, name="unibind_e2e_8l.npz", with_hidden=False):
    cfg, CLIPVisionConfig, CLIPVisionModel, LlamaConfig = MG.import_reference_models()
    CLIPVisionModel.from_pretrained = staticmethod(lambda name, **kw: CLIPVisionModel(CLIPVisionConfig(
        hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224,
        patch_size=14, hidden_act="quick_gelu")))
    import lhrs.models.text_modal as tm

    tm.CustomLlamaForCausalLM.from_pretrained = staticmethod(lambda path, **kw: tm.CustomLlamaForCausalLM(LlamaConfig(
        vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=NL, num_attention_heads=32, hidden_act="silu",
        max_position_embeddings=2048, rms_norm_eps=1e-5, pad_token_id=0, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)))

    class FakeTok:
        unk_token_id = pad_token_id = 0
        bos_token_id = 1
        model_max_length = 2048

        def __len__(self):
            return 32000

    tm.LlamaTokenizerFast.from_pretrained = staticmethod(lambda n: FakeTok())
    from lhrs.models import build_model

    model = build_model(cfg, activate_modal=("rgb", "text"))
    model.prepare_for_training(freeze_vision=True, freeze_text=True, tune_rgb_pooler=True, model_path=None, tune_im_start=False,
                               compute_dtype=torch.float32)
    for q in model.text.parameters():
        q.requires_grad = False
    vit, pool = OP.make_vit_params(seed=2), OP.make_pooler_params(seed=1)
    enc_keys = model.rgb.encoder.state_dict().keys()
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in enc_keys) else ""
    model.rgb.encoder.load_state_dict(OP.vit_to_hf(vit, prefix), strict=False)
    model.rgb_pooler.load_state_dict(OP.pooler_to_ref(pool), strict=True)
    llama = OP.make_llama_params(seed=3, layers=NL)
    model.text.text_encoder.load_state_dict(OP.llama_to_hf(llama), strict=False)
    del llama
    model.text.tune_pooler = False

    g = torch.Generator().manual_seed(ids_seed)
    B = 2
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0], ids[:, 1] = 1, -200
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(rgb_seed))
    batch = dict(rgb=rgb, input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    taps = {}
    h1 = model.rgb_pooler.register_forward_hook(lambda m, i, o: taps.update(image=o))
    h2 = model.text.text_encoder.model.register_forward_hook(lambda m, i, o: taps.update(hidden=o[0].detach()))
    out = model(batch)
    h2.remove()
    taps["image"].retain_grad()
    loss = out["total_loss"]
    loss.backward()
    h1.remove()
    grads = {n: q.grad for n, q in model.rgb_pooler.named_parameters()}
    print(f"{NL}-layer T={T} e2e loss", loss.item())
    extra = dict(hidden_sample=taps["hidden"][:, ::8, ::4].numpy().astype(np.float16)) if with_hidden else {}  # final-norm output rows
    np.savez_compressed(
        os.path.join(HERE, name), ids_seed=np.array(ids_seed), rgb_seed=np.array(rgb_seed), input_ids=ids.numpy(), **extra,
        rgb_checksum=np.array(rgb.double().sum().item()), loss=np.array(loss.item(), dtype=np.float64),
        d_image=taps["image"].grad.detach().numpy().astype(np.float32)[:, ::4, ::4],
        grad_names=np.array(list(grads.keys())), grad_norms=np.array([grads[n].norm().item() for n in grads], dtype=np.float64),
        g_out_proj_b=grads["out_proj.bias"].numpy().astype(np.float32), n_llama_layers=np.array(NL))
    print(os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "s273":
        # the HEADLINE shape (BASELINE configs[1]: T = 130 => S = 273) through the reference, 2 LLaMA-7B-width layers
        main(NL=2, T=130, ids_seed=414, rgb_seed=415, name="unibind_e2e_s273.npz", with_hidden=True)
    else:
        main()
