#!/usr/bin/env python
"""Single-image VQA chat on the gfx950 engine - the reference's `cli_qa.py` surface (/root/reference cli_qa.py:44-202) kept
flag-compatible (`-c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --image-file <png> --accelerator gpu`):

    build_model -> custom_load_state_dict -> CLIP preprocess (HIP) -> llava_llama_2 prompt -> tokenizer_image_token ->
    KeywordsStoppingCriteria(["</s>"]) -> model.generate(do_sample=True, temperature=0.4, max_new_tokens=512, streamer=...)

The prefill runs on the GEMM path, the per-token step is one captured hipGraph (lhrs_bot_amd/text.py `_decode_session`);
`bits: 8` in the YAML (or `--opts bits 8`) streams e4m3 weights through the MFMA GEMV.

Imports and call order follow /root/reference cli_qa.py:10-22, 84-193 over the `lhrs.*` surface.  Weights / tokenizer come from the paths
in the YAML (`text.path`, `rgb_vision.vit_name`); when they are not on disk the towers are random-initialised and the word-hash
stand-in tokenizer is used (both announced).  `--synthetic-prompt T` is the non-interactive BASELINE.json configs[4] run: one 224x224
image, a T-token prompt of random ids with the `<image>` placeholder, `--max-new-tokens` greedy tokens, one JSON line with the rate.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.CustomTrainer.utils import ConfigArgumentParser, ConfigDict, str2bool  # noqa: E402
from lhrs.Dataset.build_transform import build_vlp_transform  # noqa: E402
from lhrs.Dataset.conversation import SeparatorStyle, default_conversation  # noqa: E402
from lhrs.models import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX,  # noqa: E402,F401
                         build_model, tokenizer_image_token)
from lhrs.utils import KeywordsStoppingCriteria, type_dict  # noqa: E402


def parse_option(args=None):
    p = ConfigArgumentParser()
    p.add_argument("--opts", default=None, nargs="+", help="'KEY VALUE' pairs applied on top of the YAML, e.g. --opts bits 8")
    p.add_argument("--image-file", type=str, help="path to image")
    p.add_argument("--model-path", type=str, default=None, help="checkpoint directory / FINAL.pt written by custom_save_checkpoint")
    p.add_argument("--seed", type=int, default=322)
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--temperature", type=float, default=0.2)
    p.add_argument("--max-new-tokens", type=int, default=512)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--accelerator", default="gpu", type=str, choices=["cpu", "gpu", "mps"])
    p.add_argument("--use-checkpoint", default=False, type=str2bool)
    # knobs of this engine
    p.add_argument("--tokenizer-path", type=str, default=None, help="directory holding the LLaMA-2 tokenizer files")
    p.add_argument("--synthetic-prompt", type=int, default=0, metavar="T", help="non-interactive: T random prompt ids, greedy decode, JSON line")
    p.add_argument("--llama-layers", type=int, default=32)
    cfg = ConfigDict(p.parse_args(wandb=True, args=args))
    opts = cfg.get("opts") or []
    if len(opts) % 2:
        p.error("--opts takes KEY VALUE pairs")
    import yaml
    for k, v in zip(opts[0::2], opts[1::2]):
        cfg[k] = yaml.safe_load(v)
    return cfg


class _Streamer:
    """transformers.TextStreamer(skip_prompt=True, skip_special_tokens=True) behaviour for ids that arrive one step at a time."""

    def __init__(self, tokenizer):
        self.tok, self.ids, self.shown = tokenizer, [], 0

    def put(self, ids):
        self.ids.extend(int(i) for i in ids.reshape(-1).tolist())
        text = self.tok.decode(self.ids, skip_special_tokens=True)
        sys.stdout.write(text[self.shown:])
        sys.stdout.flush()
        self.shown = len(text)

    def end(self):
        sys.stdout.write("\n")
        self.ids, self.shown = [], 0


def load_image(path):
    from PIL import Image  # decode only; resize / crop / normalise run in lhrs_clip_preprocess
    return Image.open(path).convert("RGB")


def main(config):
    if config.accelerator != "gpu" or not torch.cuda.is_available():
        raise RuntimeError("cli_qa.py drives the HIP engine: it needs --accelerator gpu and a visible MI355X (there is no CPU path)")
    device = torch.device("cuda", 0)
    torch.manual_seed(int(config.seed))
    # build_model loads the frozen towers + tokenizer from config.text.path / config.rgb_vision.vit_name (random-init with a warning when
    # the paths are not on disk); --model-path adds the trained projector (FINAL.pt, or the directory holding it) and TextLoRA/
    model = build_model(config, activate_modal=("rgb", "text"))
    vision_processor = build_vlp_transform(config, is_train=False)
    dtype = type_dict[config.get("dtype", "bfloat16")]
    model.to(dtype)
    conv = default_conversation.copy()
    roles = conv.roles
    if config.get("model_path") is not None:
        print(model.custom_load_state_dict(config.model_path))
    if config.get("tokenizer_path"):
        import transformers
        model.text.tokenizer = transformers.AutoTokenizer.from_pretrained(config.tokenizer_path, use_fast=False)
    tokenizer = model.text.tokenizer
    model.eval()
    weights = "fp8" if int(config.get("bits", 16) or 16) == 8 else "bf16"

    if config.get("image_file"):
        image = load_image(config.image_file)
        image_tensor = vision_processor(image, return_tensors="pt")["pixel_values"]
    elif config.synthetic_prompt:
        g = torch.Generator().manual_seed(int(config.seed))
        image = True
        image_tensor = vision_processor(torch.randint(0, 256, (256, 256, 3), generator=g, dtype=torch.uint8))["pixel_values"]
    else:
        image, image_tensor = None, None

    if config.synthetic_prompt:
        T = int(config.synthetic_prompt)
        g = torch.Generator().manual_seed(int(config.seed))
        ids = torch.randint(3, 32000, (1, T), generator=g)
        ids[0, 0], ids[0, 1] = 1, IMAGE_TOKEN_INDEX
        kw = dict(images=image_tensor, do_sample=False, use_cache=True, weights=weights, eos_token_id=None)
        model.generate(ids, max_new_tokens=4, **kw)  # graph capture + allocator warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(ids, max_new_tokens=int(config.max_new_tokens), **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"metric": "cli_qa greedy generate tokens/s (ViT + projector + prefill included)", "value": round(out.shape[1] / dt, 1),
                          "unit": "tokens/s", "new_tokens": int(out.shape[1]), "prompt_positions": T - 1 + 144, "weights": weights,
                          "seconds": round(dt, 3)}))
        return out

    if getattr(tokenizer, "is_synthetic", False):
        print("WARNING: no tokenizer files (config.text.path / --tokenizer-path): the word-hash stand-in is in use, the answers are not text")
    while True:
        try:
            inp = input(f"{roles[0]}: ")
        except EOFError:
            inp = ""
        if not inp:
            print("exit...")
            break
        print(f"{roles[1]}: ", end="")
        if image is not None:  # first message carries the image
            tok = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN if config.get("tune_im_start", False) else DEFAULT_IMAGE_TOKEN
            inp, image = tok + "\n" + inp, None
        conv.append_message(conv.roles[0], inp)
        conv.append_message(conv.roles[1], None)
        prompt = conv.get_prompt()
        input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).to(device)
        stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
        stopping_criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
        with torch.inference_mode():
            output_ids = model.generate(input_ids, images=image_tensor, do_sample=True, max_new_tokens=int(config.max_new_tokens), temperature=0.4,
                                        streamer=_Streamer(tokenizer), use_cache=True, stopping_criteria=[stopping_criteria], weights=weights)
        outputs = tokenizer.decode(output_ids[0]).strip().split("<s>")[-1].strip()
        conv.messages[-1][-1] = outputs
        if config.debug:
            print("\n", {"prompt": prompt, "outputs": outputs}, "\n")
    return conv


if __name__ == "__main__":
    cfg = parse_option()
    cfg.adjust_norm = False
    main(cfg)
