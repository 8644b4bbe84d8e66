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

    # ------------------------------------------------------------------ reference surface
    def prepare_for_training(self, freeze_vision=True, freeze_text=True, tune_rgb_pooler=True, model_path=None,
                             tune_im_start=False, compute_dtype=torch.bfloat16):
        if not freeze_vision or tune_im_start:
            raise NotImplementedError("the ViT and the embedding tables stay frozen (every shipped stage: tune_rgb_bk / tune_im_start False)")
        if not freeze_text and self.text.lora is None and _get(self.config, "lora.enable", False):
            # TextModal.__init__ LoRA branch (text_modal.py:133-151): LoraConfig(r, lora_alpha, lora_dropout) on every linear of the decoder
            self.enable_lora(r=int(_get(self.config, "lora.lora_r", 128)), alpha=float(_get(self.config, "lora.lora_alpha", 256)),
                             dropout=float(_get(self.config, "lora.lora_dropout", 0.0)))
        if not freeze_text and self.text.lora is None:
            raise NotImplementedError("freeze_text=False means LoRA training (freeze_text = not config.lora.enable): call "
                                      "model.enable_lora(...) first; full LLaMA fine-tuning is not on the reference's path")
        self.rgb_pooler.requires_grad = bool(tune_rgb_pooler)
        self.train()
        if model_path is not None:
            self.custom_load_state_dict(model_path)
        if self.bits == 8 and not self.text.base8 and self.text.p.get("layers"):
            self.text.quantize_base(8)  # the YAML's `bits: 8`: e4m3 copies of the (now loaded) frozen decoder linears

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
        emb = self.rgb_pooler.forward(self.rgb.encode(image), save_ctx=False)
        return emb.float().mean(dim=1).to(emb.dtype) if pool else emb

    def forward(self, data: Dict) -> Dict[str, torch.Tensor]:
        """UniBind.forward: {"text_loss", "total_loss"} as 0-dim device tensors."""
        pool_grad = self.training and self.rgb_pooler.requires_grad
        grad = pool_grad or (self.training and self.text.lora is not None)
        # host copies of the small integer inputs FIRST: if they live on the device this is the one synchronising copy of the step, and
        # it happens while the queue is empty anyway (step boundary) instead of draining it behind the ViT
        host_ints = self.text._ints_to_host(data["input_ids"], data["labels"], data.get("attention_mask"))
        image_embedding = self.rgb_pooler.forward(self.rgb.encode(data["rgb"]), save_ctx=pool_grad)
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

    def custom_load_state_dict(self, path: str, strict: bool = False):
        """lhrs/models/UniBind.py:83-117: FINAL.pt (rgb encoder + projector) and a sibling TextLoRA/ adapter, trainable when
        stage > 2 and merged into the base weights when stage == 0 (evaluation)."""
        import os
        from .checkpoint import load_peft_dir, lora_from_peft
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
                self.text.enable_lora(r=cfg["r"], alpha=cfg["lora_alpha"], targets=targets)
            lora_from_peft(self.text.lora, sd)
            if self.stage == 0:
                self.text.merge_lora()
        return None


def build_model(config=None, activate_modal=("rgb", "text"), **kw) -> UniBind:
    """lhrs.models.build_model (lhrs/models/build.py:16-22)."""
    return UniBind(activate_modal, config, **kw)
