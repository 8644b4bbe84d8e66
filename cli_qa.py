#!/usr/bin/env python
"""Single-image VQA chat on the gfx950 engine - the reference's `cli_qa.py` surface (/root/reference cli_qa.py:44-202) kept
flag-compatible (`-c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --image-file <png> --accelerator gpu`):

    build_model -> custom_load_state_dict -> CLIP preprocess (HIP) -> llava_llama_2 prompt -> tokenizer_image_token ->
    KeywordsStoppingCriteria(["</s>"]) -> model.generate(do_sample=True, temperature=0.4, max_new_tokens=512, streamer=...)

The prefill runs on the GEMM path, the per-token step is one captured hipGraph (lhrs_bot_amd/text.py `_decode_session`);
`bits: 8` in the YAML (or `--opts bits 8`) streams e4m3 weights through the MFMA GEMV.

No tokenizer files exist offline: `--tokenizer-path <dir with tokenizer.model>` enables the interactive loop; without it only
`--synthetic-prompt T` runs (BASELINE.json configs[4]: one 224x224 image, a T-token prompt of random ids with the `<image>`
placeholder, `--max-new-tokens` greedy tokens) and prints one JSON line with the decode rate.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs_bot_amd.data import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, LLAMA_2_ROLES, LLAMA_2_SEP2, CLIPImageProcessorHIP,  # noqa: E402
                               llama_2_prompt, tokenizer_image_token)
from lhrs_bot_amd.eval_utils import KeywordsStoppingCriteria  # noqa: E402
from lhrs_bot_amd.text import DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX  # noqa: E402
from lhrs_bot_amd.trainer import ConfigArgumentParser, ConfigDict, str2bool  # noqa: E402
from lhrs_bot_amd.unibind import build_model  # noqa: E402


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
    model = build_model(config, activate_modal=("rgb", "text"), device=device, llama_layers=int(config.get("llama_layers", 32)))
    if config.get("model_path"):
        print(model.custom_load_state_dict(config.model_path, strict=False))
    else:
        model.init_random(seed=0)  # offline: LLaMA-2 / CLIP weights are not on disk
    model.eval()
    weights = "fp8" if int(config.get("bits", 16) or 16) == 8 else "bf16"
    processor = CLIPImageProcessorHIP(device=device)

    if config.get("image_file"):
        image_tensor = processor(load_image(config.image_file), return_tensors="pt")["pixel_values"]
    elif config.synthetic_prompt:
        g = torch.Generator().manual_seed(int(config.seed))
        image_tensor = processor(torch.randint(0, 256, (256, 256, 3), generator=g, dtype=torch.uint8))["pixel_values"]
    else:
        image_tensor = None

    if config.synthetic_prompt:
        T = int(config.synthetic_prompt)
        g = torch.Generator().manual_seed(int(config.seed))
        ids = torch.randint(3, 32000, (1, T), generator=g)
        ids[0, 0], ids[0, 1] = 1, IMAGE_TOKEN_INDEX
        kw = dict(images=image_tensor, do_sample=False, use_cache=True, weights=weights)
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

    if not config.get("tokenizer_path"):
        raise RuntimeError("the interactive loop needs --tokenizer-path (no tokenizer files ship with this repo); "
                           "use --synthetic-prompt T for the offline decode run")
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(config.tokenizer_path, use_fast=False)
    model.text.tokenizer = tokenizer
    roles, messages, first = LLAMA_2_ROLES, [], image_tensor is not None
    while True:
        try:
            inp = input(f"{roles[0]}: ")
        except EOFError:
            inp = ""
        if not inp:
            print("exit...")
            break
        print(f"{roles[1]}: ", end="")
        if first:  # first message carries the image
            tok = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN if config.get("tune_im_start", False) else DEFAULT_IMAGE_TOKEN
            inp, first = tok + "\n" + inp, False
        messages.append([roles[0], inp])
        messages.append([roles[1], None])
        prompt = llama_2_prompt(messages)
        input_ids = tokenizer_image_token(prompt, tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
        stopping = KeywordsStoppingCriteria([LLAMA_2_SEP2], tokenizer, input_ids)
        output_ids = model.generate(input_ids, images=image_tensor, do_sample=True, max_new_tokens=int(config.max_new_tokens), temperature=0.4,
                                    streamer=_Streamer(tokenizer), use_cache=True, stopping_criteria=[stopping], weights=weights)
        outputs = tokenizer.decode(output_ids[0]).strip().split("<s>")[-1].strip()
        messages[-1][-1] = outputs
        if config.debug:
            print("\n", {"prompt": prompt, "outputs": outputs}, "\n")


if __name__ == "__main__":
    cfg = parse_option()
    cfg.adjust_norm = False
    main(cfg)
