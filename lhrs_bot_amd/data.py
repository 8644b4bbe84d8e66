"""Stage-1 batch contract: prompt -> token ids with the <image> placeholder, label masking, collation.

Mirrors /root/reference lhrs/Dataset/cap_dataset.py: `tokenizer_image_token` (:1065-1084), `preprocess_plain` (:955-974),
`DataCollatorForSupervisedDataset.__call__` (:775-810) and the plain conversation separator of
lhrs/Dataset/conversation.py:324-331 (`conv_llava_plain`, sep = "\\n"); for stages 2/3 and evaluation (§8 f-2/f-3):
`preprocess_multimodal`, `preprocess_llama_2` with the `conv_llava_llama_2` template, `DataCollatorForVGSupervisedDataset`, and the
image transform `CLIPImageProcessorHIP`.  Pure host-side integer logic; the tokenizer is any
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


# ------------------------------------------------------------------------------------------------ stage 2 / 3 (llava_llama_2)
LLAVA_LLAMA_2_SYSTEM = ("You are a helpful language and vision assistant. You are able to understand the visual content that the user "
                        "provides, and assist the user with a variety of tasks using natural language.")
LLAMA_2_ROLES = ("USER", "ASSISTANT")
LLAMA_2_SEP, LLAMA_2_SEP2 = "<s>", "</s>"
DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_start>", "<im_end>"


def llama_2_prompt(messages: Sequence[Sequence], system: str = LLAVA_LLAMA_2_SYSTEM, sep: str = LLAMA_2_SEP, sep2: str = LLAMA_2_SEP2) -> str:
    """Conversation.get_prompt for SeparatorStyle.LLAMA_2 (lhrs/Dataset/conversation.py:72-95), `conv_llava_llama_2` (:300-311)."""
    ret = ""
    for i, (role, message) in enumerate(messages):
        if i == 0:
            assert message, "first message should not be none"
            assert role == LLAMA_2_ROLES[0], "first message should come from user"
        if message:
            if i == 0:
                message = f"<<SYS>>\n{system}\n<</SYS>>\n\n" + message
            if i % 2 == 0:
                ret += sep + f"[INST] {message} [/INST]"
            else:
                ret += " " + message + " " + sep2
    return ret.lstrip(sep)  # str.lstrip(chars): strips any leading run of '<', 's', '>' characters, exactly like the reference


def preprocess_multimodal(sources, tune_im_start: bool = False):
    """lhrs/Dataset/cap_dataset.py:857-885: move `<image>` to the front of the turn that mentions it."""
    if not isinstance(sources, list):
        sources = [sources]
    for idx, source in enumerate(sources):
        for key, value in source.items():
            if value is not None and DEFAULT_IMAGE_TOKEN in value:
                value = value.replace(DEFAULT_IMAGE_TOKEN, "").strip()
                value = (DEFAULT_IMAGE_TOKEN + "\n" + value).strip()
                replace_token = DEFAULT_IMAGE_TOKEN
                if tune_im_start:
                    replace_token = DEFAULT_IM_START_TOKEN + replace_token + DEFAULT_IM_END_TOKEN
                source[key] = value.replace(DEFAULT_IMAGE_TOKEN, replace_token)
        sources[idx] = source
    return sources


def preprocess_llama_2(sources: Sequence[Dict], tokenizer, has_image: bool = False) -> Dict:
    """lhrs/Dataset/cap_dataset.py:888-952.  Restated with its quirks: ALL sources feed ONE conversation (the reference appends the
    prompt outside its loop over sources), only the assistant turns keep their labels, and a length mismatch after masking blanks
    the whole sample (labels all IGNORE_INDEX)."""
    roles = {"Question": LLAMA_2_ROLES[0], "Answer": LLAMA_2_ROLES[1], "value": LLAMA_2_ROLES[1]}
    messages = []
    for i, source in enumerate(sources):
        for j, key in enumerate(source):
            assert roles[key] == LLAMA_2_ROLES[j % 2], f"{i}"
            messages.append([roles[key], source[key]])
    conversations = [llama_2_prompt(messages)]
    if has_image:
        input_ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in conversations], dim=0)
    else:
        input_ids = tokenizer(conversations, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                              truncation=True).input_ids
    targets = input_ids.clone()
    sep = "[/INST] "
    for conversation, target in zip(conversations, targets):
        total_len = int(target.ne(tokenizer.pad_token_id).sum())
        cur_len = 1
        target[:cur_len] = IGNORE_INDEX
        for rou in conversation.split(LLAMA_2_SEP2):
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            parts[0] += sep
            round_len = len(tokenizer_image_token(rou, tokenizer))
            instruction_len = len(tokenizer_image_token(parts[0], tokenizer)) - 2
            target[cur_len: cur_len + instruction_len] = IGNORE_INDEX
            cur_len += round_len
        target[cur_len:] = IGNORE_INDEX
        if cur_len < tokenizer.model_max_length and cur_len != total_len:
            target[:] = IGNORE_INDEX
    return dict(input_ids=input_ids, labels=targets)


def preprocess(sources, tokenizer, has_image: bool = False, sep_style: str = "llama_2") -> Dict:
    """lhrs/Dataset/cap_dataset.py:1051-1062; `sep_style` stands for conversation_lib.default_conversation.sep_style
    ("plain" in stage 1, "llama_2" = conv_llava_llama_2 in stages 2/3)."""
    if sep_style == "plain":
        return preprocess_plain(sources, tokenizer)
    if sep_style == "llama_2":
        return preprocess_llama_2(sources, tokenizer, has_image=has_image)
    raise ValueError(f"Unsupported separator style: {sep_style}")


class DataCollatorForVGSupervisedDataset:
    """Evaluation collator (lhrs/Dataset/cap_dataset.py:811-854): prompts are padded on the LEFT with pad_token_id; returns
    (images, input_ids, targets, filename, attention_mask)."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, instances):
        seqs = [list(inst[1]) if not torch.is_tensor(inst[1]) else inst[1].tolist() for inst in instances]
        n = max(len(s) for s in seqs)
        pad = self.tokenizer.pad_token_id
        input_ids = torch.tensor([[pad] * (n - len(s)) + s for s in seqs])[:, : self.tokenizer.model_max_length]
        images = [inst[0] for inst in instances]
        if all(torch.is_tensor(x) and x.shape == images[0].shape for x in images):
            images = torch.stack(images)
        return images, input_ids, [inst[2] for inst in instances], [inst[3] for inst in instances], input_ids.ne(pad)


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
                im = np.array(im.convert("RGB"))
            t = torch.from_numpy(np.array(im, copy=True))
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
