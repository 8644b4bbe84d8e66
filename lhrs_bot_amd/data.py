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
from typing import Callable, Dict, List, Optional, Sequence, Tuple

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
from . import conversation as conversation_lib  # noqa: E402  (module object: `default_conversation` is re-bound by the datasets)

DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_start>", "<im_end>"
LLAMA_2_ROLES = tuple(conversation_lib.conv_llava_llama_2.roles)
LLAVA_LLAMA_2_SYSTEM = conversation_lib.conv_llava_llama_2.system
_INST_CLOSE = "[/INST] "


def llama_2_prompt(messages: Sequence[Sequence], system: str = LLAVA_LLAMA_2_SYSTEM, sep: str = "<s>", sep2: str = "</s>") -> str:
    """The `conv_llava_llama_2` prompt of a message list (Conversation.get_prompt, lhrs/Dataset/conversation.py:72-95, 300-311)."""
    conv = conversation_lib.Conversation(system=system, roles=LLAMA_2_ROLES, messages=[list(m) for m in messages],
                                         sep_style=conversation_lib.SeparatorStyle.LLAMA_2, sep=sep, sep2=sep2)
    return conv.get_prompt()


def _image_token_first(text: str, image_text: str) -> str:
    """One turn's text with every `<image>` mention collapsed into a single leading "<image>\n" (then spelled as `image_text`)."""
    body = text.replace(DEFAULT_IMAGE_TOKEN, "").strip()
    return (DEFAULT_IMAGE_TOKEN + "\n" + body).strip().replace(DEFAULT_IMAGE_TOKEN, image_text)


def preprocess_multimodal(sources, tune_im_start: bool = False):
    """lhrs/Dataset/cap_dataset.py:857-885: in every turn that mentions the image, the placeholder moves to the front of the turn
    (optionally wrapped in <im_start>/<im_end>).  Edits the dicts in place and returns the list, like the reference."""
    if not isinstance(sources, list):
        sources = [sources]
    image_text = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN if tune_im_start else DEFAULT_IMAGE_TOKEN
    for turn in sources:
        for who in list(turn):
            if turn[who] is not None and DEFAULT_IMAGE_TOKEN in turn[who]:
                turn[who] = _image_token_first(turn[who], image_text)
    return sources


def _supervised_spans(prompt: str, tokenizer, close: str, end_of_round: str, n_tokens: int, count: Callable[[str], int]):
    """Token ranges of `prompt` that keep their labels under the reference's round arithmetic (cap_dataset.py:920-946, 1011-1044).

    The prompt is a sequence of rounds "<instruction><close><answer>" each terminated by `end_of_round`; the reference measures every
    round by tokenising its TEXT separately (so each measurement carries its own BOS) and supervises, inside a round of `n` measured
    tokens whose instruction measures `k`, the tokens [k - 2, n) counted from the round's start; the walk starts at token 1 (after
    BOS) and stops at the first piece that is empty or does not contain exactly one `close`.  -> (spans, walked) where `walked` is the
    position the walk ended at: the reference blanks the whole sample when it differs from the non-pad length."""
    spans, cur = [], 1
    for piece in prompt.split(end_of_round):
        if piece == "" or piece.count(close) != 1:
            break
        instruction = piece[: piece.index(close) + len(close)]
        n, k = count(piece), count(instruction) - 2
        spans.append((cur + max(k, 0), cur + n))
        cur += n
    return [(a, min(b, n_tokens)) for a, b in spans if a < min(b, n_tokens)], cur


def _preprocess_rounds(sources, tokenizer, has_image: bool, conv, close: str) -> Dict:
    """Shared body of the LLAMA_2 and the two-separator (v1) label rules: ONE conversation is rendered from all `sources` (the
    reference appends the prompt outside its loop over sources), tokenised, and only the answer part of each round keeps labels."""
    roles = {"Question": conv.roles[0], "Answer": conv.roles[1], "value": conv.roles[1]}
    conv = conv.copy()
    conv.messages = []
    for i, source in enumerate(sources):
        for j, key in enumerate(source):
            assert roles[key] == conv.roles[j % 2], f"{i}"
            conv.append_message(roles[key], source[key])
    prompt = conv.get_prompt()
    if has_image:
        input_ids = tokenizer_image_token(prompt, tokenizer, return_tensors="pt")[None]
        count = lambda text: len(tokenizer_image_token(text, tokenizer))  # noqa: E731
    else:
        input_ids = tokenizer([prompt], return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length, truncation=True).input_ids
        # the reference's LLAMA_2 rule measures rounds with the image-aware tokeniser even for text-only samples; v1 does not
        count = (lambda text: len(tokenizer_image_token(text, tokenizer))) if conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2 \
            else (lambda text: len(tokenizer(text).input_ids))
    labels = torch.full_like(input_ids, IGNORE_INDEX)
    row = input_ids[0]
    spans, walked = _supervised_spans(prompt, tokenizer, close, conv.sep2, row.numel(), count)
    total = int(row.ne(tokenizer.pad_token_id).sum())
    if not (walked < tokenizer.model_max_length and walked != total):  # otherwise: tokenisation mismatch -> the sample teaches nothing
        for a, b in spans:
            labels[0, a:b] = row[a:b]
    return dict(input_ids=input_ids, labels=labels)


def preprocess_llama_2(sources: Sequence[Dict], tokenizer, has_image: bool = False, conv=None) -> Dict:
    """lhrs/Dataset/cap_dataset.py:888-952 (`conv_llava_llama_2`: rounds close with "[/INST] " and end with "</s>")."""
    conv = conv or conversation_lib.conv_llava_llama_2
    assert conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2
    return _preprocess_rounds(sources, tokenizer, has_image, conv, _INST_CLOSE)


def preprocess_v1(sources: Sequence[Dict], tokenizer, has_image: bool = False, conv=None) -> Dict:
    """lhrs/Dataset/cap_dataset.py:977-1048 (two-separator templates: a round's instruction closes with " ASSISTANT: ")."""
    conv = conv or conversation_lib.conv_templates["v1"]
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    return _preprocess_rounds(sources, tokenizer, has_image, conv, conv.sep + conv.roles[1] + ": ")


def preprocess(sources, tokenizer, has_image: bool = False, sep_style: Optional[str] = None) -> Dict:
    """lhrs/Dataset/cap_dataset.py:1051-1062: dispatch on `conversation.default_conversation` (the datasets bind it to
    `conv_templates[prompt_type]`: "plain" in stage 1, "llava_llama_2" in stages 2/3).  `sep_style` ("plain" / "llama_2") overrides the
    module-level default for callers that do not want global state."""
    conv = conversation_lib.default_conversation
    if sep_style is not None:
        conv = {"plain": conversation_lib.conv_llava_plain, "llama_2": conversation_lib.conv_llava_llama_2}.get(sep_style)
        if conv is None:
            raise ValueError(f"Unsupported separator style: {sep_style}")
    if conv.sep_style == conversation_lib.SeparatorStyle.PLAIN:
        return preprocess_plain(sources, tokenizer)
    if conv.sep_style == conversation_lib.SeparatorStyle.LLAMA_2:
        return preprocess_llama_2(sources, tokenizer, has_image=has_image, conv=conv)
    if conv.version.startswith("v1"):
        return preprocess_v1(sources, tokenizer, has_image=has_image, conv=conv)
    raise ValueError(f"Unsupported separator style: {conv.sep_style}")


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


class _Features(dict):
    """dict with attribute access: `processor(img, return_tensors="pt").pixel_values` (cli_qa.py:118-122) and `[...]` both work."""
    __getattr__ = dict.__getitem__


class DeviceImageTransform:
    """An image transform whose arithmetic runs on the GPU for a whole batch (`lhrs_image_preprocess`): a dataset that is handed one of
    these only DECODES in its workers (PIL -> uint8 HWC tensor); `preprocess` turns decoded pictures into float32 [B, 3, 224, 224]
    pixel values on the device.  Subclasses fix the parameters of the pipeline they stand in for.  Accepts PIL images, uint8 HWC numpy
    arrays or uint8 HWC tensors (host or device)."""

    crop_size = {"height": 224, "width": 224}
    short_edge, crop_round, rescale_mode = 224, 0, 0
    image_mean: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    image_std: Tuple[float, float, float] = (1.0, 1.0, 1.0)

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
        if torch.is_tensor(images) and images.dim() == 4:
            images = list(images)
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = torch.empty((len(images), 3, 224, 224), device=self.device, dtype=torch.float32)
        for b, im in enumerate(images):
            hk.image_preprocess(self._to_u8(im), out=out[b], short_edge=self.short_edge, crop_round=self.crop_round, rescale_mode=self.rescale_mode,
                                mean=self.image_mean, std=self.image_std)
        return _Features(pixel_values=out)

    __call__ = preprocess


class CLIPImageProcessorHIP(DeviceImageTransform):
    """Drop-in for the transform `build_vlp_transform` returns for the ViT arch (lhrs/Dataset/build_transform.py:43-45, HF
    `CLIPImageProcessor`): `.preprocess(images, return_tensors="pt")["pixel_values"]` -> float32 [B, 3, 224, 224] ON THE DEVICE,
    bit-exact with the PIL pipeline (resize short edge 224 BICUBIC, center crop, /255, CLIP mean/std)."""

    image_mean = (0.48145466, 0.4578275, 0.40821073)
    image_std = (0.26862954, 0.26130258, 0.27577711)


class ClsEvalTransformHIP(DeviceImageTransform):
    """`build_cls_transform(config, is_train=False)` (lhrs/Dataset/build_transform.py:27-40), the transform of the zero-shot
    classification loader: torchvision Resize(int(224 / (224 / 256)) = 256, BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize with the
    ImageNet mean / std.  `pixels(images)` -> float32 [B, 3, 224, 224] on the device."""

    short_edge, crop_round, rescale_mode = 256, 1, 1
    image_mean = (0.485, 0.456, 0.406)
    image_std = (0.229, 0.224, 0.225)

    def __init__(self, device="cuda", input_size=(224, 224)):
        size = input_size if isinstance(input_size, (list, tuple)) else (input_size, input_size)
        if tuple(size)[-2:] != (224, 224):
            raise ValueError(f"transform.input_size {input_size}: the device transform crops 224 x 224 (every shipped YAML)")
        super().__init__(device)
