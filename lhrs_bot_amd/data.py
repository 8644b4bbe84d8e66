"""Stage-1 batch contract: prompt -> token ids with the <image> placeholder, label masking, collation.

Mirrors /root/reference lhrs/Dataset/cap_dataset.py: `tokenizer_image_token` (:1065-1084), `preprocess_plain` (:955-974),
`DataCollatorForSupervisedDataset.__call__` (:775-810) and the plain conversation separator of
lhrs/Dataset/conversation.py:324-331 (`conv_llava_plain`, sep = "\\n").  Pure host-side integer logic; the tokenizer is any
object with `__call__(text).input_ids`, `bos_token_id`, `pad_token_id`, `model_max_length`.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Sequence

import torch

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
PLAIN_SEP = "\n"


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors=None):
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split(DEFAULT_IMAGE_TOKEN)]
    input_ids: List[int] = []
    offset = 0
    if len(chunks) > 0 and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        input_ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    pieces = []
    for i, c in enumerate(chunks):  # chunk, sep, chunk, sep, ..., chunk
        pieces.append(c)
        if i + 1 < len(chunks):
            pieces.append(sep)
    for x in pieces:
        input_ids.extend(x[offset:])
    if return_tensors is not None:
        if return_tensors == "pt":
            return torch.tensor(input_ids, dtype=torch.long)
        raise ValueError(f"Unsupported tensor type: {return_tensors}")
    return input_ids


def preprocess_plain(sources: Sequence[Dict], tokenizer) -> Dict:
    """Stage-1 ("plain") samples: `<image>` + caption + "\\n"; everything up to and including the image token is masked."""
    conversations = []
    for source in sources:
        assert len(source) == 2
        assert DEFAULT_IMAGE_TOKEN in source["Question"]
        source["Question"] = DEFAULT_IMAGE_TOKEN
        conversations.append(source["Question"] + source["Answer"] + PLAIN_SEP)
    input_ids = [tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in conversations]
    targets = copy.deepcopy(input_ids)
    for target, source in zip(targets, sources):
        target[: len(tokenizer_image_token(source["Question"], tokenizer))] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


class DataCollatorForSupervisedDataset:
    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        input_ids, labels = ([inst["text"][k] for inst in instances] for k in ("input_ids", "labels"))
        pad = self.tokenizer.pad_token_id
        input_ids = torch.nn.utils.rnn.pad_sequence(input_ids, batch_first=True, padding_value=pad)
        labels = torch.nn.utils.rnn.pad_sequence(labels, batch_first=True, padding_value=IGNORE_INDEX)
        input_ids = input_ids[:, : self.tokenizer.model_max_length]
        labels = labels[:, : self.tokenizer.model_max_length]
        batch = dict(input_ids=input_ids, labels=labels, attention_mask=input_ids.ne(pad))
        if "rgb" in instances[0]:
            images = [inst["rgb"] for inst in instances]
            same = all(torch.is_tensor(x) and x.shape == images[0].shape for x in images)
            batch["rgb"] = torch.stack(images) if same else images
        if "valid_image" in instances[0]:
            batch["valid_image"] = torch.tensor([inst["valid_image"] for inst in instances])
        return batch


class CLIPImageProcessorHIP:
    """Drop-in for the transform `build_vlp_transform` returns for the ViT arch (lhrs/Dataset/build_transform.py:43-45, HF
    `CLIPImageProcessor`): `.preprocess(images, return_tensors="pt")["pixel_values"]` -> float32 [B, 3, 224, 224] ON THE DEVICE,
    bit-exact with the PIL pipeline (resize short edge 224 BICUBIC, center crop, /255, CLIP mean/std) but computed by
    `lhrs_clip_preprocess`: the DataLoader only has to decode and ship uint8 pixels.  Accepts PIL images, uint8 HWC numpy
    arrays or uint8 HWC tensors (host or device)."""

    crop_size = {"height": 224, "width": 224}
    image_mean = (0.48145466, 0.4578275, 0.40821073)
    image_std = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, device="cuda"):
        from . import _lib
        _lib.load()
        self.device = torch.device(device)

    def _to_u8(self, im) -> torch.Tensor:
        if isinstance(im, torch.Tensor):
            t = im
        else:
            import numpy as np
            if hasattr(im, "convert"):  # PIL: do_convert_rgb
                im = np.asarray(im.convert("RGB"))
            t = torch.from_numpy(np.ascontiguousarray(im))
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H, W, 3] image, got {tuple(t.shape)} {t.dtype}")
        return t.to(self.device).contiguous()

    def preprocess(self, images, return_tensors="pt", **_kw) -> Dict[str, torch.Tensor]:
        from . import kernels as hk
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = torch.empty((len(images), 3, 224, 224), device=self.device, dtype=torch.float32)
        for b, im in enumerate(images):
            hk.clip_preprocess(self._to_u8(im), out=out[b])
        return {"pixel_values": out}

    __call__ = preprocess
