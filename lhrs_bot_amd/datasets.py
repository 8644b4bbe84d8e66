"""Real-data loaders of the training stages: image + annotation directories -> the batch dict the engine steps on.

Replaces /root/reference lhrs/Dataset/cap_dataset.py `CaptionDataset` (:77-182), `CaptionDatasetVQA` (:330-385), `InstructDataset`
(:386-486) and lhrs/Dataset/build_loader.py `build_loader_hepler` (:25-57), `build_vlp_loader` (:60-162), `build_loader` (:202-212),
lhrs/Dataset/build_transform.py `build_vlp_transform` (:43-45), lhrs/CustomTrainer/utils/sampler.py `InfiniteSampler`.

MI355X-first split of the work: DataLoader workers only DECODE (PIL -> uint8 HWC) and tokenise; resize / crop / normalise is the
bit-exact device kernel `lhrs_clip_preprocess` applied to the whole batch after the H2D copy (`CLIPImageProcessorHIP`, run by
`UniBind.forward` / `Trainer.put_input_to_device` when `batch["rgb"]` holds uint8 pixels).  A 224² fp32 tensor per sample never
crosses PCIe; a worker never touches the GPU.

Directory contract (unchanged): `<root>/<NAME>_Image/` holds the pictures of corpus NAME and `<root>/<NAME>.json` its annotations;
the annotation schema is chosen by NAME exactly as the reference does (table `_CAPTION_SCHEMAS` / `_instruct_item`).
RS5M tar shards (the reference reads them through `webdataset`, absent here) are read by `RS5MDataset` on `tarfile` with the same
pipeline stages and defaults.
"""
from __future__ import annotations

import itertools
import json
import logging
import random
import re
from pathlib import Path
from typing import Callable, Dict, Iterator, List, Optional, Tuple

import torch

from . import conversation as conversation_lib
from .data import (DEFAULT_IMAGE_TOKEN, CLIPImageProcessorHIP, DataCollatorForSupervisedDataset, DeviceImageTransform, preprocess,
                   preprocess_multimodal)

logger = logging.getLogger("train")

def valid_path(p) -> bool:
    """An annotation whose picture is missing is skipped, not an error (cap_dataset.py:44-49)."""
    return Path(p).exists()


def pre_caption(caption, max_words: int = 50):
    """Caption normalisation of cap_dataset.py:52-74: lower-case, punctuation `.!"()*#:;~` -> space, runs of whitespace collapsed,
    trimmed, at most `max_words` words.  Conversations (dict / list) pass through untouched."""
    if isinstance(caption, (dict, list)):
        return caption
    text = re.sub(r"\s{2,}", " ", re.sub(r"([.!\"()*#:;~])", " ", caption.lower())).rstrip("\n").strip(" ")
    words = text.split(" ")
    return " ".join(words[:max_words]) if len(words) > max_words else text


# ------------------------------------------------------------------------------------------------ caption corpora (stage 1)
def _rsicd_like(data, img_dir: Path):            # default schema: {"images": [{"filename", "sentences": [{"raw"}]}]}
    for im in data["images"]:
        yield img_dir / im["filename"], im["sentences"][0]["raw"]


def _textrs(data, img_dir: Path):
    for it in data["TextRS"]:
        yield img_dir / (it["image"] + ".png"), it["annotation"]["caption"][0]


def _uavicd(data, img_dir: Path):
    for im in data["images"]:
        yield img_dir / im["SubFolder"] / im["ImageName"], im["Caption"]


def _nwpu(data, img_dir: Path):
    for sub, items in data.items():
        for it in items:
            yield img_dir / sub / it["filename"], it["raw"]


def _llava(data, img_dir: Path):
    for it in data["data"]:
        yield img_dir / it["name"], it["conv"]


# (substring of the *_Image directory stem, reader) - first match wins, in the reference's order
_CAPTION_SCHEMAS: List[Tuple[str, Callable]] = [("TextRS", _textrs), ("UAVICD", _uavicd), ("NWPU", _nwpu), ("LLAVA", _llava)]


def _osm_items(img_dir: Path, ann: Path):
    """OSM captions: a directory `OSMCapAnn/` of json files, images under <country>/<city>/<name>.jpg."""
    for jf in ann.iterdir():
        if jf.is_file() and jf.suffix == ".json":
            for it in json.loads(jf.read_text())["data"]:
                country, city = it["info"]["location"]
                yield img_dir / country / city / (it["name"] + ".jpg"), it["cap"]


class CaptionDataset(torch.utils.data.Dataset):
    """`<root>/*_Image` + sibling json -> samples {"rgb", "text"}.  transform None -> PIL image; `CLIPImageProcessorHIP` -> uint8 HWC
    tensor (the device finishes the transform per batch); any other callable is applied to the PIL image."""

    def __init__(self, root=".data/rsicd", transform=None):
        self.root = Path(root)
        self.transform = transform
        self.img_dir = list(self.root.glob("*_Image"))
        self.json_dir = []
        for d in self.img_dir:
            name = d.stem.split("_Image")[0]
            osm_dir = d.parent / "OSMCapAnn"
            self.json_dir.append(osm_dir if "OSM" in d.stem and osm_dir.exists() else d.parent / (name + ".json"))
        self.img_list: List[Path] = []
        self.cap_list: List = []
        self.load_dataset()
        self.post_process()

    def post_process(self):
        pass

    def _add(self, pairs):
        for path, cap in pairs:
            if valid_path(path):
                self.img_list.append(path)
                self.cap_list.append(cap)

    def load_dataset(self):
        for img_dir, ann in zip(self.img_dir, self.json_dir):
            if "OSM" in img_dir.stem and all(key not in img_dir.stem for key, _ in _CAPTION_SCHEMAS[:3]):
                self._add(_osm_items(img_dir, ann))
                continue
            data = json.loads(ann.read_bytes())
            reader = next((fn for key, fn in _CAPTION_SCHEMAS if key in img_dir.stem), _rsicd_like)
            self._add(reader(data, img_dir))

    def __len__(self) -> int:
        return len(self.cap_list)

    def load_image(self, idx: int):
        from PIL import Image
        img = Image.open(self.img_list[idx]).convert("RGB")
        if self.transform is None:
            return img
        if isinstance(self.transform, DeviceImageTransform):
            import numpy as np
            return torch.from_numpy(np.array(img, copy=True))  # uint8 [H, W, 3]: decoded here, transformed on the device
        return self.transform(img)

    def __getitem__(self, idx: int) -> Dict:
        cap = self.cap_list[idx]
        return dict(rgb=self.load_image(idx), text=cap if isinstance(cap, list) else pre_caption(cap))


def _tokenised(sample: Dict, tokenizer, tune_im_start: bool) -> Dict:
    conv = preprocess(preprocess_multimodal(sample["text"], tune_im_start=tune_im_start), tokenizer, has_image=True)
    sample["text"] = dict(input_ids=conv["input_ids"][0], labels=conv["labels"][0])
    return sample


class CaptionDatasetVQA(CaptionDataset):
    """Stage-1 alignment data: every caption becomes ONE question/answer turn whose question is drawn from 11 fixed instructions
    (python's global `random`, one draw per caption in corpus order - seed it for reproducible epochs), then tokenised with the
    template `prompt_type` names ("plain" in the stage-1 YAML: the question collapses to the bare image token)."""

    QUESTION_TEMPLACES = [q + "\n" + DEFAULT_IMAGE_TOKEN for q in (
        "Describe the image concisely.", "Provide a brief description of the given image.",
        "Offer a succinct explanation of the picture presented.", "Summarize the visual content of the image.",
        "Give a short and clear explanation of the subsequent image.", "Share a concise interpretation of the image provided.",
        "Present a compact description of the photo’s key features.", "Relay a brief, clear account of the picture shown.",
        "Render a clear and concise summary of the photo.", "Write a terse but informative summary of the picture.",
        "Create a compact narrative representing the image presented.")]

    def __init__(self, tokenizer, **kwargs):
        self.tune_im_start = kwargs.pop("tune_im_start", False)
        conversation_lib.default_conversation = conversation_lib.conv_templates[kwargs.pop("prompt_type", "llava_llama_2")]
        self.tokenizer = tokenizer
        super().__init__(**kwargs)

    def post_process(self):
        for i, cap in enumerate(self.cap_list):
            if isinstance(cap, list):
                first = cap[0]
                if isinstance(first, dict) and DEFAULT_IMAGE_TOKEN in first["Question"]:  # already a conversation about the image
                    if "Answer" not in first:
                        first["Answer"] = first.pop("value")
                        self.cap_list[i] = [first]
                    continue
                cap = first
            self.cap_list[i] = [{"Question": random.choice(self.QUESTION_TEMPLACES), "Answer": pre_caption(cap)}]

    def __getitem__(self, idx: int) -> Dict:
        return _tokenised(super().__getitem__(idx), self.tokenizer, self.tune_im_start)


# ------------------------------------------------------------------------------------------------ instruction corpora (stage 2 / 3)
def _instruct_item(dataset_name: str, img_dir: Path, item: Dict) -> Path:
    """Image path of one annotation record; grounding corpora (RSVG / DIOR) also get their single-turn conversation built here."""
    if dataset_name.endswith(("RSVG", "DIOR")):
        item["conv"] = dict(Question=item["question"], Answer=item["answer"])
        return img_dir / (item["img"] if dataset_name.endswith("RSVG") else item["img"] + ".jpg")
    if "METERML" in dataset_name:
        return img_dir / item["name"] / "naip.png"
    if "OSM" in dataset_name:
        return img_dir / (item["filename"] + ".jpg")
    if "name" in item:
        return img_dir / item["name"]
    fn = item["filename"]
    return img_dir / (fn[0] if isinstance(fn, list) else fn)


class InstructDataset(CaptionDataset):
    """Multi-turn instruction data: records {"name"|"filename"|..., "conv": [{"Question","Answer"}, ...]} (optionally under a top-level
    "data" key).  Conversations longer than 10 turns are sub-sampled to 10 (global `random`); the image token is forced to the front
    of the first question and removed everywhere else."""

    def __init__(self, tokenizer, crop_size: int = 224, **kwargs):
        self.tune_im_start = kwargs.pop("tune_im_start", False)
        conversation_lib.default_conversation = conversation_lib.conv_templates[kwargs.pop("prompt_type", "llava_llama_2")]
        self.tokenizer, self.crop_size = tokenizer, crop_size
        super().__init__(**kwargs)

    def load_dataset(self):
        for img_dir, ann in zip(self.img_dir, self.json_dir):
            data = json.loads(ann.read_bytes())
            if isinstance(data, dict) and "data" in data:
                data = data["data"]
            for item in data:
                path = _instruct_item(ann.stem, img_dir, item)
                if valid_path(path):
                    conv = item["conv"]
                    self.img_list.append(path)
                    self.cap_list.append(random.sample(conv, 10) if isinstance(conv, list) and len(conv) > 10 else conv)

    def post_process(self):
        convs, imgs = [], []
        for path, conv in zip(self.img_list, self.cap_list):
            conv = conv if isinstance(conv, list) else [conv]
            if not conv:
                continue
            if DEFAULT_IMAGE_TOKEN not in conv[0]["Question"]:
                conv[0]["Question"] = DEFAULT_IMAGE_TOKEN + conv[0]["Question"]
            for turn in conv[1:]:
                for who in ("Question", "Answer"):
                    turn[who] = turn[who].replace(DEFAULT_IMAGE_TOKEN, "")
            convs.append(conv)
            imgs.append(path)
        self.cap_list, self.img_list = convs, imgs

    def load_image(self, idx: int):
        if idx >= len(self.img_list):
            return torch.zeros(3, self.crop_size, self.crop_size)
        return super().load_image(idx)

    def __getitem__(self, idx: int) -> Dict:
        out = _tokenised(super().__getitem__(idx), self.tokenizer, self.tune_im_start)
        out["valid_image"] = idx < len(self.img_list)
        return out


class InstructDatasetWithTaskId(InstructDataset):
    """Stage-3 mixture (`weight_sample: True` in Config/multi_modal_stage3.yaml; cap_dataset.py:489-577): every sample carries the sampling
    weight of its corpus (first WEIGHT_DICT key found in the annotation file's name, else 0.5); text-only instruction corpora
    (`<root>/*text.json` whose name contains "geosignal": records {instruction, input, output}) are appended AFTER the image samples - they
    have no picture (`load_image` gives zeros, `valid_image` False) and no image token.  Unlike its parent it only ADDS the image token to
    a first question that lacks it; later turns are left alone."""

    WEIGHT_DICT = {"OSM": 0.6, "LLAVA": 1.0, "geosignal": 0.50, "RSITMD": 0.6, "NWPU": 0.6, "DOTA": 0.9, "FAST": 1.0}

    def __init__(self, **kwargs):
        self.sample_weight: List[float] = []
        super().__init__(**kwargs)

    def load_dataset(self):
        corpora = list(zip(self.img_dir, self.json_dir))
        for img_dir, ann in corpora:
            n0 = len(self.img_list)
            self.img_dir, self.json_dir = [img_dir], [ann]
            try:
                super().load_dataset()  # one corpus at a time: same records, same `random.sample` draws, in the same order
            finally:
                self.img_dir, self.json_dir = [c[0] for c in corpora], [c[1] for c in corpora]
            w = next((wt for key, wt in self.WEIGHT_DICT.items() if key in ann.stem), 0.5)
            self.sample_weight += [w] * (len(self.img_list) - n0)

    def post_process(self):
        for i, conv in enumerate(self.cap_list):
            conv = conv if isinstance(conv, list) else [conv]
            if DEFAULT_IMAGE_TOKEN not in conv[0]["Question"]:
                conv[0]["Question"] = DEFAULT_IMAGE_TOKEN + conv[0]["Question"]
            self.cap_list[i] = conv
        self.txt_json_dir = [f for f in self.root.glob("*text.json") if f not in self.json_dir]
        for f in self.txt_json_dir:
            if "geosignal" in f.stem:
                for item in json.loads(f.read_bytes()):
                    self.cap_list.append([{"Question": item["instruction"] + item["input"], "Answer": item["output"]}])
                    self.sample_weight.append(self.WEIGHT_DICT["geosignal"])


# ------------------------------------------------------------------------------------------------ samplers / loaders
class DistributedSamplerWrapper(torch.utils.data.DistributedSampler):
    """Any sampler, sharded over the ranks (lhrs/Dataset/utils.py:7-57): every pass the wrapped sampler's index list is drawn (each rank
    draws its own - the reference does not synchronise that RNG), and the DistributedSampler's epoch-seeded shuffle of POSITIONS decides
    which of them this rank takes."""

    def __init__(self, sampler, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True):
        self.sampler = sampler
        super().__init__(list(range(len(sampler))), num_replicas=num_replicas, rank=rank, shuffle=shuffle)

    def __iter__(self):
        drawn = list(self.sampler)
        return iter([drawn[i] for i in super().__iter__()])



class InfiniteSampler(torch.utils.data.Sampler):
    """Endless index stream for iteration-based training: reshuffled passes over the dataset from ONE seeded generator shared by all
    ranks, rank r taking elements r, r + world, ... of the stream (lhrs/CustomTrainer/utils/sampler.py)."""

    def __init__(self, dataset, shuffle: bool = True, seed: Optional[int] = None):
        dist = torch.distributed
        on = dist.is_available() and dist.is_initialized()
        self.rank, self.world_size = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        if seed is None:  # every rank must walk the same stream: rank 0 draws, everybody else receives
            t = torch.randint(0, 2 ** 31, (1,))
            if on:
                dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
                t = t.to(dev)
                dist.broadcast(t, src=0)
            seed = int(t.item())
        self.seed, self.shuffle, self.size = seed, shuffle, len(dataset)

    def _stream(self) -> Iterator[int]:
        g = torch.Generator().manual_seed(self.seed)
        while True:
            yield from (torch.randperm(self.size, generator=g) if self.shuffle else torch.arange(self.size)).tolist()

    def __iter__(self) -> Iterator[int]:
        return itertools.islice(self._stream(), self.rank, None, self.world_size)

    def __len__(self) -> int:
        return self.size

    def set_epoch(self, epoch: int) -> None:
        pass


# ------------------------------------------------------------------------------------------------ RS5M tar shards (stage 1's real corpus)
# The reference streams RS5M through the `webdataset` package (cap_dataset.py:649-775 `RS5MDataset`, build_loader.py:110-160).  That package is
# not in this image; the format it reads is plain POSIX tar, so the pipeline is restated on `tarfile` with the same stages and the same
# defaults: brace-expanded shard list -> per-epoch shard shuffle (seed 322 + epoch) -> split by rank, then by DataLoader worker -> tar
# members grouped into samples by the key in front of the first dot (no-throw grouping, cap_dataset.py:587-613) -> sample shuffle buffer ->
# decode (image bytes -> pixels, caption bytes -> one question/answer turn, tokenised) -> batches of exactly `batch_size` (partial=False).
_SHARD_SHUFFLE_SIZE, _SHARD_SHUFFLE_INITIAL = 2000, 500          # cap_dataset.py:30-33
_SAMPLE_SHUFFLE_SIZE, _SAMPLE_SHUFFLE_INITIAL = 5000, 1000
RS5M_NUM_SAMPLES = 5070186                                         # build_loader.py:131


def expand_braces(pattern: str) -> List[str]:
    """`{a,b}` alternatives and `{0000..0031}` zero-padded integer ranges, nested left to right (what braceexpand / wds.shardlists.expand_urls
    do for the reference's `{pub11,rs3}-train-{0000..0031}.tar`)."""
    m = re.search(r"\{([^{}]*)\}", pattern)
    if not m:
        return [pattern]
    body, out = m.group(1), []
    rng = re.fullmatch(r"(-?\d+)\.\.(-?\d+)", body)
    if rng:
        a, b = rng.group(1), rng.group(2)
        width = max(len(a), len(b)) if (a.startswith("0") and len(a) > 1) or (b.startswith("0") and len(b) > 1) else 0
        step = 1 if int(b) >= int(a) else -1
        alts = [str(v).zfill(width) for v in range(int(a), int(b) + step, step)]
    else:
        alts = body.split(",")
    for alt in alts:
        out += expand_braces(pattern[: m.start()] + alt + pattern[m.end():])
    return out


def _shuffle_buffer(src: Iterator, bufsize: int, initial: int, rng: random.Random) -> Iterator:
    """webdataset's `_shuffle`: keep up to `bufsize` items, start yielding once `initial` are buffered, each yield swaps a random slot out."""
    initial = min(initial, bufsize)
    buf: List = []
    for item in src:
        buf.append(item)
        if len(buf) < bufsize:
            try:
                buf.append(next(src))
            except StopIteration:
                pass
        if len(buf) >= initial:
            k = rng.randint(0, len(buf) - 1)
            buf[k], buf[-1] = buf[-1], buf[k]
            yield buf.pop()
    while buf:
        k = rng.randint(0, len(buf) - 1)
        buf[k], buf[-1] = buf[-1], buf[k]
        yield buf.pop()


def tar_samples(path: str, suffixes=("img_content", "img_name", "caption")) -> Iterator[Dict]:
    """Members of one tar shard grouped into samples: `<key>.<suffix>` -> {"__key__": key, suffix: bytes}; a sample ends when the key changes
    or a suffix repeats (group_by_keys_nothrow, cap_dataset.py:587-613); samples need a key and at least one field; unreadable shards /
    members are logged and skipped (log_and_continue)."""
    import tarfile
    try:
        tf = tarfile.open(path, "r|*")
    except (OSError, tarfile.TarError) as e:
        logger.warning("Handling shard error (%r). Ignoring.", e)
        return
    cur: Optional[Dict] = None
    try:
        for member in tf:
            if not member.isreg():
                continue
            base = member.name.rsplit("/", 1)[-1]
            if "." not in base or base.startswith("."):
                continue
            prefix_dir = member.name[: len(member.name) - len(base)]
            key, suffix = base.split(".", 1)
            key, suffix = prefix_dir + key, suffix.lower()
            if cur is None or key != cur["__key__"] or suffix in cur:
                if cur is not None and len(cur) > 2:
                    yield cur
                cur = {"__key__": key, "__url__": path}
            if suffixes is None or suffix in suffixes:
                fh = tf.extractfile(member)
                cur[suffix] = fh.read() if fh is not None else b""
    except (OSError, tarfile.TarError, EOFError) as e:
        logger.warning("Handling shard error (%r). Ignoring.", e)
    finally:
        tf.close()
    if cur is not None and len(cur) > 2:
        yield cur


class RS5MDataset(torch.utils.data.IterableDataset):
    """RS5M image / caption tar shards `<root>/{pub11,rs3}-train-{0000..0031}.tar` as an iterable of stage-1 samples {"rgb", "text"} (or of
    collated batches when `batch_size` is given, as the reference's `wds.batched(batch_size, partial=False)` yields them).  A sample is
    the members `<key>.img_content` (encoded picture), `<key>.img_name`, `<key>.caption` (UTF-8 text)."""

    URL = "{pub11,rs3}-train-{0000..0031}.tar"

    def __init__(self, root=".data/RS5M", transform=None, tokenizer=None, batch_size: Optional[int] = None, rank: Optional[int] = None,
                 world_size: Optional[int] = None, seed: int = 322, shards: Optional[List[str]] = None, **kwargs):
        self.tune_im_start = kwargs.pop("tune_im_start", False)
        self.prompt_type = kwargs.pop("prompt_type", "llava_llama_2")
        self.transform, self.tokenizer, self.batch_size, self.seed = transform, tokenizer, batch_size, seed
        self.url = str(Path(root) / self.URL)
        self.shards = list(shards) if shards is not None else expand_braces(self.url)
        dist = torch.distributed
        on = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank() if on else 0)
        self.world = world_size if world_size is not None else (dist.get_world_size() if on else 1)
        self.epoch = -1
        self.worker_batches: Optional[int] = None   # with_epoch(n): items per worker per epoch (None: one pass over the shards)
        self._shared_epoch = None                   # multiprocessing.Value once a loader with workers owns this dataset
        self.collate = DataCollatorForSupervisedDataset(tokenizer=tokenizer) if batch_size else None

    def share_epoch(self) -> None:
        """Keep the epoch in a `multiprocessing.Value` (the reference's SharedEpoch): persistent DataLoader workers hold their own copy
        of this object, so a plain attribute set by `set_epoch` in the trainer process would never reach them."""
        import multiprocessing
        self._shared_epoch = multiprocessing.Value("i", max(self.epoch, 0))

    def set_epoch(self, epoch: int) -> None:
        if self._shared_epoch is not None:
            self._shared_epoch.value = int(epoch)
        self.epoch = int(epoch) - 1   # __iter__ advances it: the reference's SharedEpoch / detshuffle2 contract

    def _decode(self, s: Dict) -> Optional[Dict]:
        import io
        from PIL import Image
        try:
            img = Image.open(io.BytesIO(s["img_content"])).convert("RGB")
            caption = s["caption"].decode("utf-8")
        except Exception as e:  # noqa: BLE001 - a broken member is skipped like the reference's handler does
            logger.warning("Handling sample error (%r). Ignoring.", e)
            return None
        if self.transform is None:
            rgb = img
        elif isinstance(self.transform, DeviceImageTransform):
            import numpy as np
            rgb = torch.from_numpy(np.array(img, copy=True))
        else:
            rgb = self.transform(img)
        conversation_lib.default_conversation = conversation_lib.conv_templates[self.prompt_type]
        turn = [{"Question": random.choice(CaptionDatasetVQA.QUESTION_TEMPLACES), "Answer": pre_caption(caption)}]
        return _tokenised(dict(rgb=rgb, text=turn), self.tokenizer, self.tune_im_start)

    def my_shards(self) -> List[str]:
        """This epoch's shards of THIS rank and THIS DataLoader worker: shuffled identically everywhere (seed + epoch), then every
        world-th shard from `rank` (wds.split_by_node), of those every num_workers-th from the worker id (wds.split_by_worker)."""
        rng = random.Random(self.seed + self.epoch)
        order = list(_shuffle_buffer(iter(self.shards), _SHARD_SHUFFLE_SIZE, _SHARD_SHUFFLE_INITIAL, rng))
        mine = order[self.rank::self.world]
        info = torch.utils.data.get_worker_info()
        return mine if info is None else mine[info.id::info.num_workers]

    def _epoch_value(self) -> int:
        return self._shared_epoch.value if self._shared_epoch is not None else self.epoch

    def _one_pass(self):
        """One walk over this worker's shards of the current epoch: samples, or whole batches when `batch_size` is set."""
        info = torch.utils.data.get_worker_info()
        rng = random.Random((info.seed if info is not None else self.seed * 7919 + self.rank) + self.epoch)
        raw = itertools.chain.from_iterable(tar_samples(p) for p in self.my_shards())
        decoded = (d for d in (self._decode(s) for s in _shuffle_buffer(raw, _SAMPLE_SHUFFLE_SIZE, _SAMPLE_SHUFFLE_INITIAL, rng)
                               if "img_content" in s and "caption" in s) if d is not None)
        if not self.batch_size:
            yield from decoded
            return
        batch: List[Dict] = []
        for d in decoded:
            batch.append(d)
            if len(batch) == self.batch_size:
                yield self.collate(batch)
                batch = []                       # partial=False: a trailing short batch is dropped

    def __iter__(self):
        """`worker_batches` = the reference's `dataset.with_epoch(num_worker_batches)` (build_loader.py:140-142): every worker of every
        rank yields EXACTLY that many items per epoch, walking its shards again when they run out, so that all ranks leave the epoch at
        the same step (a rank that ran dry earlier would leave the others waiting in the gradient all-reduce)."""
        if self._shared_epoch is not None:       # SharedEpoch (cap_dataset.py:523-534): persistent workers see the trainer's set_epoch
            self.epoch = self._shared_epoch.value
        else:
            self.epoch += 1
        if not self.worker_batches:
            yield from self._one_pass()
            return
        n = 0
        while n < self.worker_batches:
            got = 0
            for item in self._one_pass():
                yield item
                got += 1
                n += 1
                if n == self.worker_batches:
                    return
            if got == 0:
                raise RuntimeError(f"RS5M: rank {self.rank} found no complete batch in its shards {self.my_shards()}")


class _SizedLoader(torch.utils.data.DataLoader):
    """DataLoader over the iterable RS5M pipeline that knows its epoch length (`Trainer.epoch_len` is `len(data_loader)`): every one
    of its workers yields exactly `num_batches / num_workers` batches per epoch."""

    num_batches = 0

    def __len__(self) -> int:
        return int(self.num_batches)

    def set_epoch(self, epoch: int) -> None:
        self.dataset.set_epoch(epoch)


def build_rs5m_loader(config, transform, **kwargs):
    """build_loader.py:110-160 for `"RS5M" in config.data_path`: batches are formed inside the dataset (per worker), the loader only
    multiplexes workers; `num_batches` / `num_samples` / `length` follow the reference's rounding over the nominal 5 070 186 samples."""
    import math
    from torch.utils.data import DataLoader
    world = int(config.get("world_size", 1) or 1)
    workers = int(config.get("workers", 0))
    bs = int(config["batch_size"])
    ds = RS5MDataset(root=str(config["data_path"]), transform=transform, batch_size=bs, **kwargs)
    have = sum(Path(p).exists() for p in ds.shards)
    if have == 0:
        raise FileNotFoundError(f"no RS5M shard found under {config['data_path']} (expected {ds.url})")
    ds.shards = [p for p in ds.shards if Path(p).exists()]
    assert len(ds.shards) >= max(1, workers) * world, "number of shards must be >= total workers"
    global_batch = bs * world
    num_samples = int(config.get("rs5m_num_samples", RS5M_NUM_SAMPLES) or RS5M_NUM_SAMPLES)
    num_batches = math.ceil(num_samples / global_batch)
    nw = max(1, workers)
    ds.worker_batches = math.ceil(num_batches / nw)      # with_epoch(num_worker_batches)
    num_batches = ds.worker_batches * nw
    if workers > 0:
        ds.share_epoch()
    loader = _SizedLoader(ds, batch_size=None, shuffle=False, num_workers=workers, persistent_workers=workers > 0)
    loader.num_batches, loader.num_samples = num_batches, num_batches * global_batch
    loader.length = math.ceil(loader.num_samples / global_batch)
    return loader



def build_vlp_transform(config, is_train: bool = True):
    """build_transform.py:43-45: ViT archs use the CLIP image processor - here the device one.  (The convolutional archs' timm /
    torchvision augmentations belong to the dropped Swin / ResNet branches.)"""
    arch = config["rgb_vision"]["arch"] if "rgb_vision" in config else "vit_large"
    if not str(arch).startswith("vit"):
        raise NotImplementedError(f"rgb_vision.arch {arch!r}: only the CLIP ViT branch is on the hot path")
    import os
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))) if torch.cuda.is_available() else torch.device("cpu")
    return CLIPImageProcessorHIP(device=dev)


def build_loader_hepler(config, dataset, collate_fn=None, is_train: bool = True):
    """Sampler policy of build_loader.py:25-57 (name kept, typo included): DistributedSampler(shuffle) when distributed, else the
    InfiniteSampler when asked for, else a shuffling, last-batch-dropping loader for training."""
    from torch.utils.data import DataLoader, DistributedSampler
    if config.get("is_distribute", False):
        sampler = DistributedSampler(dataset, shuffle=True)
    elif config.get("inf_sampler", False) and is_train:
        sampler = InfiniteSampler(dataset, shuffle=True)
    else:
        sampler = None
    plain_train = is_train and sampler is None
    workers = int(config.get("workers", 0))
    return DataLoader(dataset, int(config["batch_size"]), sampler=sampler, num_workers=workers, pin_memory=torch.cuda.is_available(),
                      drop_last=plain_train, shuffle=plain_train, collate_fn=collate_fn, persistent_workers=workers > 0)


def build_vlp_loader(config, is_train: bool = True, **kwargs):
    transform = build_vlp_transform(config, is_train=is_train)
    root = str(config["data_path"])
    stage = int(config.get("stage", 1))
    if is_train and stage == 1 and "RS5M" in root:
        return build_rs5m_loader(config, transform, **kwargs)
    if is_train and stage == 1:
        dataset = CaptionDatasetVQA(root=root, transform=transform, **kwargs)
    elif is_train:
        size = config["transform"]["input_size"][0] if "transform" in config else 224
        if config.get("weight_sample", False):  # build_loader.py:83-109: weighted draw without replacement, sharded over the ranks
            from torch.utils.data import DataLoader, WeightedRandomSampler
            dataset = InstructDatasetWithTaskId(root=root, transform=transform, crop_size=size, **kwargs)
            dist = torch.distributed
            on = dist.is_available() and dist.is_initialized()
            sampler = DistributedSamplerWrapper(WeightedRandomSampler(dataset.sample_weight, num_samples=len(dataset), replacement=False),
                                                num_replicas=dist.get_world_size() if on else 1, rank=dist.get_rank() if on else 0)
            workers = int(config.get("workers", 0))
            logger.info("Build dataset: Train samples = %d (weighted by corpus)", len(dataset))
            return DataLoader(dataset, sampler=sampler, batch_size=int(config["batch_size"]), num_workers=workers,
                              pin_memory=torch.cuda.is_available(), drop_last=True, persistent_workers=workers > 0,
                              collate_fn=DataCollatorForSupervisedDataset(tokenizer=kwargs["tokenizer"]))
        dataset = InstructDataset(root=root, transform=transform, crop_size=size, **kwargs)
    else:
        raise NotImplementedError("evaluation loaders are built by the eval scripts (main_vqa / main_cls / main_vg), outside this path")
    logger.info("Build dataset: Train images = %d", len(dataset))
    loader = build_loader_hepler(config, dataset, is_train=is_train, collate_fn=DataCollatorForSupervisedDataset(tokenizer=kwargs["tokenizer"]))
    logger.info("Build dataloader: Epoch length = %d", len(loader))
    return loader


def build_loader(config, mode: str = "pretrain", is_train: bool = True, **kwargs):
    """lhrs.Dataset.build_loader.build_loader(config, mode="pretrain", tokenizer=..., prompt_type=...) (build_loader.py:202-212)."""
    assert mode in ["pretrain"], "Please choose mode for dataloder from [pretrain]"
    return build_vlp_loader(config, is_train=is_train, **kwargs)
