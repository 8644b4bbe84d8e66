"""UniBind: the multimodal module the entry scripts drive, re-built on the gfx950 engine.

Mirrors /root/reference lhrs/models/UniBind.py (`UniBind.__init__` :25-57, `prepare_for_training` :119-176,
`forward` :178-199, `encode_image` :201-212) and lhrs/models/build.py:16-22 (`build_model`).  There is no autograd:
`forward` records what the hand-written backward needs and `backward()` runs LLaMA dX -> splice slice -> AttnPooler
(dW + dX), leaving fp32 gradients in `rgb_pooler.grad`.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .pooler import AttnPooler
from .text import TextModal
from .vision import VisionModal


def _get(cfg, path: str, default):
    cur = cfg
    for k in path.split("."):
        if cur is None:
            return default
        cur = cur.get(k) if isinstance(cur, dict) else getattr(cur, k, None)
    return default if cur is None else cur


class UniBind:
    def __init__(self, activate_modal: Tuple[str, ...] = ("rgb", "text"), config=None, device="cuda",
                 llama_layers: Optional[int] = None, vit_layers: int = 24):
        _lib.load()  # fail loudly, before anything else, if the HIP library is missing
        assert len(activate_modal) > 0, "activate_modal should not be empty"
        self.modal = tuple(activate_modal)
        self.stage = _get(config, "stage", 1)
        self.config = config
        self.bits = int(_get(config, "bits", 16))  # Config/multi_modal_stage{2,3}.yaml: 8 -> 8-bit frozen base (text_modal.py:91-131)
        self.device = torch.device(device)
        if "rgb" in self.modal:
            self.rgb = VisionModal(config, device, layers=vit_layers)
            self.rgb_pooler = AttnPooler(
                num_query=_get(config, "rgb_vision.attn_pooler.num_query", 144),
                num_layers=_get(config, "rgb_vision.attn_pooler.num_layers", 6),
                num_attention_heads=_get(config, "rgb_vision.attn_pooler.num_attn_heads", 16),
                encoder_hidden_size=VisionModal.EMBEDDING_DIM[_get(config, "rgb_vision.arch", "vit_large")],
                hidden_size=VisionModal.EMBEDDING_DIM[_get(config, "rgb_vision.arch", "vit_large")],
                output_size=_get(config, "text.hidden_size", 4096), device=device)
        if "text" in self.modal:
            eps = float(_get(config, "text.rms_norm_eps", 1e-5))  # yaml.safe_load yields the STRING "1e-5" 
            self.text = TextModal(config, device, layers=llama_layers or _get(config, "text.num_hidden_layers", 32),
                                  dim=_get(config, "text.hidden_size", 4096), eps=eps)
        self.training = True
        self._image_embedding = None
        self._log_precision_keys(config)

    def _log_precision_keys(self, config):
        """The YAMLs' precision keys (Config/multi_modal_stage2.yaml:75-80: `dtype`, `bits`, `double_quant`, `quant_type`, `fp16`, `bf16`;
        consumed at text_modal.py:79-131).  One INFO line saying what each means HERE, so that a shipped YAML runs unchanged and nothing is
        dropped silently."""
        import logging
        if config is None:
            return
        log = logging.getLogger("train")
        dt = _get(config, "dtype", None)
        if dt is not None and str(dt) not in ("bfloat16", "bf16"):
            log.info("config dtype=%s: frozen towers are stored and multiplied in bf16 (fp32 accumulation); the key selects no other path", dt)
        if self.bits in (4, 8):
            if self.bits == 8:
                log.info("config bits=8: frozen decoder linears kept as LLM.int8 rows + absmax factors, int8 MFMA + 16-bit outlier columns "
                         "(LHRS_BASE8=e4m3: e4m3 rows on the fp8 MFMA); double_quant=%s / quant_type=%s are bitsandbytes 4-bit options "
                         "(text_modal.py:97-101) and do not apply to the 8-bit base", _get(config, "double_quant", None), _get(config, "quant_type", None))
            else:
                log.info("config bits=4: frozen decoder linears stored as %s codes per block of 64 (double_quant=%s), every product on the "
                         "dequantised bf16 weight - bitsandbytes' Linear4bit arithmetic", _get(config, "quant_type", "nf4"), _get(config, "double_quant", True))

    # ------------------------------------------------------------------ reference surface
    def prepare_for_training(self, freeze_vision=True, freeze_text=True, tune_rgb_pooler=True, model_path=None,
                             tune_im_start=False, compute_dtype=torch.bfloat16):
        if not freeze_vision or tune_im_start:
            raise NotImplementedError("the ViT and the embedding tables stay frozen (every shipped stage: tune_rgb_bk / tune_im_start False)")
        # UniBind.py:119-141 casts the towers to `compute_dtype`.  This engine has ONE arithmetic: bf16 storage / MFMA operands, fp32
        # accumulation, fp32 masters for what trains.  fp16 (the stage-2/3 YAMLs) and fp32 requests are answered with a warning, not an error
        self.compute_dtype_request = compute_dtype
        if compute_dtype not in (torch.bfloat16, None):
            import logging
            logging.getLogger("train").warning("prepare_for_training(compute_dtype=%s): running the bf16 path (fp32 accumulation, fp32 master "
                                               "weights); fp16 range handling / loss scaling is not needed and not run", compute_dtype)
        if not freeze_text and self.text.lora is None and _get(self.config, "lora.enable", False):
            # TextModal.__init__ LoRA branch (text_modal.py:133-151): LoraConfig(r, lora_alpha, lora_dropout) on every linear of the decoder
            self.enable_lora(r=int(_get(self.config, "lora.lora_r", 128)), alpha=float(_get(self.config, "lora.lora_alpha", 256)),
                             dropout=float(_get(self.config, "lora.lora_dropout", 0.0)))
        if not freeze_text and self.text.lora is None:
            raise NotImplementedError("freeze_text=False means LoRA training (freeze_text = not config.lora.enable): call "
                                      "model.enable_lora(...) first; full LLaMA fine-tuning is not on the reference's path")
        self.rgb_pooler.requires_grad = bool(tune_rgb_pooler)
        self.train()
        if self.bits in (4, 8) and not (self.text.base8 or self.text.base_int8 or getattr(self.text, "base4", None)) and self.text.p.get("layers"):
            # the YAML's `bits: 8`: LLM.int8 storage and arithmetic of the (now loaded) frozen decoder linears, as the reference runs stages 2/3;
            # LHRS_BASE8=e4m3 selects the faster MI355X-native 8-bit base instead (a deviation: no outlier decomposition).
            # `bits: 4`: bitsandbytes 4-bit storage with the YAML's quant_type / double_quant (text_modal.py:97-101)
            import os
            self.text.quantize_base(self.bits, os.environ.get("LHRS_BASE8", "int8"), quant_type=str(_get(self.config, "quant_type", "nf4")),
                                    double_quant=bool(_get(self.config, "double_quant", True)))
        # AFTER the base is quantised, as in the reference: TextModal.__init__ loads the decoder through bitsandbytes (text_modal.py:91-131) and
        # only then UniBind.custom_load_state_dict attaches / merges the adapters (UniBind.py:105-115) - a stage-0 merge therefore computes
        # Q(D(Q(W)) + s B A) (merge_lora re-quantises the merged weight), not Q(W + s B A)
        if model_path is not None:
            self.custom_load_state_dict(model_path)

    def train(self):
        self.training = True
        if getattr(self, "text", None) is not None and self.text.lora is not None:
            self.text.lora.train_mode = True
        return self

    def eval(self):
        self.training = False
        if getattr(self, "text", None) is not None and self.text.lora is not None:
            self.text.lora.train_mode = False  # lora_dropout off (peft: nn.Dropout in eval mode)
        return self

    def to(self, *a, **k):
        return self

    def named_parameters(self):
        """(name, fp32 master view) of every TRAINABLE tensor, named as the reference's modules name them (`rgb_pooler.layers.0.ln_1.weight`,
        peft's `text.text_encoder...lora_A.default.weight`): what `build_optimizer` splits into decay / no-decay groups."""
        from .checkpoint import PROJ_MODULE
        out = []
        if hasattr(self, "rgb_pooler") and self.rgb_pooler.requires_grad:
            out += [("rgb_pooler." + n, t) for n, t in self.rgb_pooler.named_parameters()]
        lo = getattr(getattr(self, "text", None), "lora", None)
        if lo is not None:
            for l in range(lo.nl):
                for proj in lo.targets:
                    A, B = lo.get_adapter(l, proj)
                    base = f"text.text_encoder.base_model.model.model.layers.{l}.{PROJ_MODULE[proj]}"
                    out += [(base + ".lora_A.default.weight", A), (base + ".lora_B.default.weight", B)]
        return out

    def parameters(self):
        return [t for _, t in self.named_parameters()]

    def load_base_weights(self, config=None, allow_random: bool = True, seed: int = 0):
        """What the reference does inside `build_model` (VisionModal: `CLIPVisionModel.from_pretrained(config.rgb_vision.vit_name)`,
        rgb_vision_modal.py:130-157; TextModal: `CustomLlamaForCausalLM.from_pretrained(config.text.path)` + `LlamaTokenizerFast`,
        text_modal.py:79-131,191-197): load the frozen towers and the tokenizer from the paths the YAML names.  There is no hub access:
        a path that is not a local directory leaves that tower RANDOM-initialised (LLaMA-2-7B / ViT-L/14 shapes) with a loud warning -
        or raises when `allow_random` is off.  The projector keeps its fresh init (the reference's AttnPooler default init)."""
        import logging
        import os
        from .checkpoint import load_hf_dir
        log = logging.getLogger("train")
        cfg = config if config is not None else self.config
        done = {}
        if hasattr(self, "rgb"):
            vit = _get(cfg, "rgb_vision.vit_name", None)
            if vit and os.path.isdir(str(vit)):
                self.rgb.load_state_dict(load_hf_dir(str(vit)))
                done["rgb"] = str(vit)
            elif not self.rgb.p:
                if not allow_random:
                    raise FileNotFoundError(f"rgb_vision.vit_name = {vit!r} is not a local checkpoint directory")
                log.warning("CLIP ViT weights %r are not on disk: the vision tower is RANDOM-initialised (synthetic run)", vit)
                self.rgb.init_random(seed)
                done["rgb"] = "random"
            if not self.rgb_pooler.initialised:
                self.rgb_pooler.init_random(seed + 1)
        if hasattr(self, "text"):
            path = _get(cfg, "text.path", None)
            if path and os.path.isdir(str(path)):
                self.text.from_pretrained(str(path), n_layers=self.text.nl)
                done["text"] = str(path)
                if any(os.path.exists(os.path.join(str(path), f)) for f in ("tokenizer.model", "tokenizer.json")):
                    import transformers
                    tok = transformers.AutoTokenizer.from_pretrained(str(path))
                    tok.pad_token_id = tok.unk_token_id  # text_modal.py:196-197
                    self.text.tokenizer = tok
            elif not self.text.p:
                if not allow_random:
                    raise FileNotFoundError(f"text.path = {path!r} is not a local checkpoint directory")
                log.warning("LLaMA weights %r are not on disk: the language model is RANDOM-initialised and the tokenizer is the synthetic "
                            "word-hash stand-in (synthetic run)", path)
                self.text.init_random(seed + 2)
                done["text"] = "random"
        self.base_weights = done
        return self

    def _pixels(self, rgb):
        """`batch["rgb"]` as the loaders deliver it: float [B,3,224,224] (already CLIP-normalised) passes through; uint8 HWC pictures
        (a [B,H,W,3] tensor or a list of [H,W,3] of different sizes, decoded by the DataLoader workers) are resized / cropped /
        normalised here by `lhrs_clip_preprocess`, bit-exact with the reference's `CLIPImageProcessor` (lhrs_bot_amd/datasets.py)."""
        is_list = isinstance(rgb, (list, tuple))
        if not is_list and not (torch.is_tensor(rgb) and rgb.dtype == torch.uint8):
            return rgb
        if getattr(self, "_image_processor", None) is None:
            from .data import CLIPImageProcessorHIP
            self._image_processor = CLIPImageProcessorHIP(device=self.device)
        items = list(rgb)
        ready = [torch.is_tensor(x) and x.is_floating_point() for x in items]  # e.g. the zero picture of a text-only sample (stage-3 mixture)
        if not any(ready):
            return self._image_processor.preprocess(items)["pixel_values"]
        out = torch.empty((len(items), 3, 224, 224), device=self.device, dtype=torch.float32)
        todo = [i for i, r in enumerate(ready) if not r]
        if todo:
            out[todo] = self._image_processor.preprocess([items[i] for i in todo])["pixel_values"]
        for i, r in enumerate(ready):
            if r:
                out[i] = items[i].to(self.device, torch.float32)
        return out

    def init_random(self, seed: int = 0):
        if hasattr(self, "rgb"):
            self.rgb.init_random(seed)
            self.rgb_pooler.init_random(seed + 1)
        if hasattr(self, "text"):
            self.text.init_random(seed + 2)
        return self

    def load_params(self, P: Dict):
        """P = {'vit':..., 'pooler':..., 'llama':...} in the engine layout (oracle/params.py)."""
        self.rgb.load_params(P["vit"])
        self.rgb_pooler.load_params(P["pooler"])
        self.text.load_params(P["llama"])
        return self

    def enable_lora(self, r=128, alpha=256, targets=None, seed=0, dropout=0.0):
        """lora.enable / lora_r / lora_alpha of Config/multi_modal_stage2.yaml:81-86 (text_modal.py:133-151)."""
        from .text import LORA_ALL
        return self.text.enable_lora(r=r, alpha=alpha, targets=targets or LORA_ALL, seed=seed, dropout=dropout)

    def encode_image(self, image, pool: bool = False):
        emb = self.rgb_pooler.forward(self.rgb.encode(self._pixels(image)), save_ctx=False)
        return emb.float().mean(dim=1).to(emb.dtype) if pool else emb

    def forward(self, data: Dict) -> Dict[str, torch.Tensor]:
        """UniBind.forward: {"text_loss", "total_loss"} as 0-dim device tensors."""
        pool_grad = self.training and self.rgb_pooler.requires_grad
        grad = pool_grad or (self.training and self.text.lora is not None)
        # host copies of the small integer inputs FIRST: if they live on the device this is the one synchronising copy of the step, and
        # it happens while the queue is empty anyway (step boundary) instead of draining it behind the ViT
        host_ints = self.text._ints_to_host(data["input_ids"], data["labels"], data.get("attention_mask"))
        image_embedding = self.rgb_pooler.forward(self.rgb.encode(self._pixels(data["rgb"])), save_ctx=pool_grad)
        loss = self.text.decode(data["input_ids"], image_embedding=image_embedding, attention_mask=data.get("attention_mask"),
                                labels=data["labels"], save_ctx=grad, host_ints=host_ints)
        return {"text_loss": loss, "total_loss": loss}

    __call__ = forward

    @torch.no_grad()
    def generate(self, input_ids, images=None, do_sample=True, temperature=0.2, max_new_tokens=1024, streamer=None, use_cache=True,
                 stopping_criteria=None, **kwargs):
        """UniBind.generate (lhrs/models/UniBind.py:214-242): encode the image once, then TextModal.generate."""
        assert hasattr(self, "text"), "text modal is not activate"
        image_embedding = self.encode_image(images, pool=False) if images is not None else None  # None: text-only turn
        return self.text.generate(input_ids=input_ids, image_embedding=image_embedding, do_sample=do_sample, temperature=temperature,
                                  max_new_tokens=max_new_tokens, streamer=streamer, use_cache=use_cache,
                                  stopping_criteria=stopping_criteria, **kwargs)

    def backward(self, loss_scale: float = 1.0) -> None:
        d_image = self.text.backward(loss_scale, need_input_grad=self.rgb_pooler.requires_grad)
        if self.rgb_pooler.requires_grad:
            self.rgb_pooler.backward(d_image)

    def custom_save_checkpoint(self, file_name: str):
        """FINAL.pt = {"rgb_ckpt": VisionModal state dict, "other_ckpt": {rgb_pooler, text_proj, embed_tokens, lm_head}} and, from
        stage 2 on, the peft adapter directory TextLoRA/ next to it (lhrs/models/UniBind.py:68-81, 275-302)."""
        import os
        from .checkpoint import save_peft_dir, vit_to_hf
        os.makedirs(file_name, exist_ok=True)
        ckpt = {"rgb_ckpt": vit_to_hf(self.rgb.export_params()),
                "other_ckpt": {"rgb_pooler": self.rgb_pooler.state_dict(), "text_proj": {},
                               "embed_tokens": {"weight": self.text.p["embed"].float().cpu()}, "lm_head": {}}}
        torch.save(ckpt, os.path.join(file_name, "FINAL.pt"))
        if self.text.lora is not None:
            save_peft_dir(self.text.lora, os.path.join(file_name, "TextLoRA"))
        return ckpt

    def custom_load_state_dict(self, state_dict_path: str, strict: bool = False):
        """lhrs/models/UniBind.py:83-117: FINAL.pt (rgb encoder + projector) and a sibling TextLoRA/ adapter, trainable when
        stage > 2 and merged into the base weights when stage == 0 (evaluation)."""
        import os
        from .checkpoint import load_peft_dir, lora_from_peft
        path = state_dict_path
        if os.path.isdir(path):  # the directory custom_save_checkpoint wrote (cli_qa's --model-path help text allows either)
            path = os.path.join(path, "FINAL.pt")
        ckpt = torch.load(path, map_location="cpu")
        if "model" in ckpt:
            ckpt = ckpt["model"]
        if ckpt.get("rgb_ckpt"):
            self.rgb.load_state_dict(ckpt["rgb_ckpt"], strict=strict)
        if "other_ckpt" in ckpt and "rgb_pooler" in ckpt["other_ckpt"]:
            self.rgb_pooler.load_state_dict(ckpt["other_ckpt"]["rgb_pooler"], strict=strict)
        text_path = os.path.join(os.path.dirname(path), "TextLoRA")
        if os.path.isdir(text_path):
            cfg, targets, sd = load_peft_dir(text_path)
            if self.text.lora is None:
                self.text.enable_lora(r=cfg["r"], alpha=cfg["lora_alpha"], targets=targets, dropout=float(cfg.get("lora_dropout", 0.0)))
            lora_from_peft(self.text.lora, sd)
            if self.stage == 0:
                self.text.merge_lora()
        return None


def build_model(config=None, activate_modal=("rgb", "text"), load_weights: Optional[bool] = None, **kw) -> UniBind:
    """lhrs.models.build_model (lhrs/models/build.py:16-22).  With a config (the entry scripts' call) the frozen towers are loaded from
    `config.text.path` / `config.rgb_vision.vit_name` as the reference does at construction (`load_base_weights`); `config=None` (tests,
    tools) returns the empty module for `init_random` / `load_params`.  Engine knobs ride in the config too: `local_rank` picks the
    device, `llama_layers` truncates the decoder (synthetic smoke runs)."""
    if config is not None:
        if "device" not in kw and torch.cuda.is_available():
            kw["device"] = torch.device("cuda", int(_get(config, "local_rank", 0) or 0) if _get(config, "local_rank", 0) not in (None, -1) else 0)
        if "llama_layers" not in kw and _get(config, "llama_layers", None):
            kw["llama_layers"] = int(_get(config, "llama_layers", 32))
    model = UniBind(activate_modal, config, **kw)
    if load_weights if load_weights is not None else config is not None:
        model.load_base_weights(config)
    return model
