"""Generate tests/golden/batch_generate.npz: the batched, LEFT-padded evaluation path of the reference
(DataCollatorForVGSupervisedDataset -> UniBind.generate(attention_mask=...), main_vqa.py:205-214) run through the REFERENCE's own
modules: UniBind.encode_image, TextModal.prepare_inputs_for_multimodal (mask rule and all) and its text_encoder
(CustomLlamaForCausalLM) with the spliced attention_mask.  HF `generate` itself cannot run under the transformers 5.x of this
container (cache API of the 4.36.1-era prepare_inputs_for_generation), so the decoding loop is restated as what it computes:
step t = one forward over [spliced prompt | forced tokens[:t]] with the mask extended by ones and NO position_ids (the custom
prepare_inputs_for_generation passes none), logits of the last position.  Teacher-forced tokens make the fixture independent
of arg-max ties.  Build container only; the tests read the committed .npz.
"""
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


def main(n_llama_layers=2):
    cfg, CLIPVisionConfig, CLIPVisionModel, LlamaConfig = MG.import_reference_models()
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
    model.eval()
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=n_llama_layers)}
    enc_keys = model.rgb.encoder.state_dict().keys()
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in enc_keys) else ""
    model.rgb.encoder.load_state_dict(OP.vit_to_hf(P["vit"], prefix), strict=False)
    model.rgb_pooler.load_state_dict(OP.pooler_to_ref(P["pooler"]), strict=True)
    model.text.text_encoder.load_state_dict(OP.llama_to_hf(P["llama"]), strict=False)
    model.text.tune_pooler = False

    g = torch.Generator().manual_seed(303)
    B, T, NEW = 3, 11, 4
    lens = [11, 8, 5]  # real prompt lengths; the collator pads on the LEFT with pad_token_id = 0
    ids = torch.zeros((B, T), dtype=torch.int64)
    for b, n in enumerate(lens):
        row = torch.randint(3, 32000, (n,), generator=g)
        row[0] = 1
        row[1] = -200
        ids[b, T - n:] = row
    mask = ids.ne(0)
    rgb = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(304))  # regenerated from the seed by the tests
    forced = torch.randint(3, 32000, (B, NEW), generator=g)
    with torch.no_grad():
        img = model.encode_image(rgb, pool=False)
        _, new_mask, _, embeds, _ = model.text.prepare_inputs_for_multimodal(
            input_ids=ids, attention_mask=mask, labels=None, past_key_values=None, image_embedding=img)
        S0 = embeds.shape[1]
        emb = model.text.get_text_encoder().model.embed_tokens
        full = torch.cat([embeds, emb(forced[:, :-1])], 1)
        full_mask = torch.cat([new_mask, torch.ones((B, NEW - 1), dtype=new_mask.dtype)], 1)
        # one causal forward over the whole teacher-forced sequence == the per-step forwards of the decoding loop
        logits = model.text.text_encoder(inputs_embeds=full, attention_mask=full_mask).logits[:, S0 - 1:].float()
        # and, literally step by step for the first two steps (guards the "one forward" shortcut)
        for t in range(2):
            lt = model.text.text_encoder(inputs_embeds=full[:, :S0 + t], attention_mask=full_mask[:, :S0 + t]).logits[:, -1].float()
            assert torch.allclose(lt, logits[:, t], atol=2e-4, rtol=1e-4), (t, (lt - logits[:, t]).abs().max())
    assert logits.shape == (B, NEW, 32000)
    np.savez_compressed(
        os.path.join(HERE, "batch_generate.npz"), rgb_seed=np.array(304), rgb_checksum=np.array(rgb.double().sum().item()), input_ids=ids.numpy(), attention_mask=mask.numpy(),
        forced_tokens=forced.numpy(), new_mask=new_mask.numpy(), logits_cols=np.arange(0, 32000, 8),
        logits=logits[:, :, ::8].numpy().astype(np.float32), argmax=logits.argmax(-1).numpy(),
        top2_margin=(logits.topk(2, -1).values[..., 0] - logits.topk(2, -1).values[..., 1]).numpy(), n_llama_layers=np.array(n_llama_layers))
    print("batch_generate golden: S0", S0, "new_mask rows", new_mask.int().sum(1).tolist(), "argmax", logits.argmax(-1).tolist())
    print(os.path.getsize(os.path.join(HERE, "batch_generate.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
