"""Datasets and loaders of the evaluation callers of `generate`: zero-shot scene classification, RSVQA, visual
grounding, caption evaluation.

Replaces /root/reference lhrs/Dataset/UCM.py, millionaid_eval.py, ImageFolderInstance.py, meterml.py, rsvqa.py (`RSVQA`, `RSVQALR`,
`RSVQAHR`, `RSVQAxBEN`, `DataCollatorForVQASupervisedDataset`), cap_dataset.py `VGEvalDataset` (:186-260) and `CapEvalDataset` (:263-327),
build_transform.py `build_cls_transform` (:9-40) and build_loader.py `build_zero_shot_loader` (:164-199).  Same constructor arguments,
same directory contracts, same per-sample tuples / dicts, same prompt ids (pinned to the reference's classes over synthetic corpora:
tests/golden/eval.json, tests/test_eval_cpu.py).

MI355X-first split of the work, as for the training loaders (datasets.py): a dataset that is handed a `DeviceImageTransform`
(`CLIPImageProcessorHIP`, `ClsEvalTransformHIP`) only decodes in its DataLoader workers and returns uint8 [H, W, 3] tensors; resize /
crop / normalise happens on the GPU for the whole batch (`lhrs_image_preprocess`) - inside `UniBind.generate(images=...)` for the CLIP
pipeline, inside `DeviceBatchLoader` for the classification pipeline.  Any other callable transform is applied to the decoded picture
in the worker, as the reference does.

Not importable here and therefore restated from their documented behaviour (parity unpinned): torchvision's
`ImageFolder` directory scan (ImageFolderInstance) and geopandas' `read_file` (METERML: the GeoJSON `properties` are read with `json`).
"""
from __future__ import annotations

import json
import logging
import os
from glob import glob
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import conversation as conversation_lib
from .data import (CLIPImageProcessorHIP, ClsEvalTransformHIP, DataCollatorForVGSupervisedDataset, DeviceImageTransform,  # noqa: F401
                   preprocess, preprocess_multimodal)
from .datasets import CaptionDataset, build_loader_hepler, valid_path

logger = logging.getLogger("train")


def _decoded(img, transform):
    """A PIL picture -> what the dataset hands out under `transform` (None: the PIL image; device transform: uint8 HWC tensor)."""
    if transform is None:
        return img
    if isinstance(transform, DeviceImageTransform):
        import numpy as np
        return torch.from_numpy(np.array(img.convert("RGB"), copy=True))
    return transform(img)


# ------------------------------------------------------------------------------------------------ scene classification
class _ListFileDataset(torch.utils.data.Dataset):
    """`<root>/<split>.txt` holds "file label" lines (UCM, MillionAID evaluation lists)."""

    def __init__(self, root, split: str, transform: Optional[Callable], return_idx: bool, splits: Sequence[str]):
        assert split in splits, "data split must be " + ", ".join(splits)
        self.root = Path(root)
        self.split, self.transform, self.return_idx = split, transform, return_idx
        self.imgs: List[str] = []
        self.cat_id: List[int] = []
        with open(self.root / (split + ".txt")) as f:
            for line in f.readlines():
                name, idx = line.split(" ")
                self.imgs.append(name)
                self.cat_id.append(int(idx.replace("\n", "")))

    def _path(self, name: str):
        return name

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, item):
        from PIL import Image
        img = Image.open(self._path(self.imgs[item]))
        if self.transform is not None:
            img = _decoded(img, self.transform)
        return (img, self.cat_id[item], item) if self.return_idx else (img, self.cat_id[item])


class UCM(_ListFileDataset):
    """UC-Merced land use, 21 classes (lhrs/Dataset/UCM.py): pictures under `<root>/<img_file_name>/`."""

    CLASS_NAME = ["agricultural", "airplane", "baseballdiamond", "beach", "buildings", "chaparral", "denseresidential", "forest", "freeway", "golfcourse",
                  "harbor", "intersection", "mediumresidential", "mobilehomepark", "overpass", "parkinglot", "river", "runway", "sparseresidential",
                  "storagetanks", "tenniscourt"]

    def __init__(self, root, split: str = "train", transform: Callable = None, img_file_name: str = "img", return_idx: bool = False):
        super().__init__(root, split, transform, return_idx, ("train", "test", "all"))
        self.data_dir = self.root / img_file_name

    def _path(self, name):
        return self.data_dir / name


class MillionAidEval(_ListFileDataset):
    """lhrs/Dataset/millionaid_eval.py: the list file holds the picture paths themselves."""

    def __init__(self, root, split: str = "train", transform: Callable = None, return_idx: bool = False):
        super().__init__(root, split, transform, return_idx, ("train", "test"))


CLASS_NAME_MAP = {"AID": ["Airport", "BareLand", "BaseballField", "Beach", "Bridge", "Center", "Church", "Commercial", "DenseResidential", "Desert", "Farmland",
                          "Forest", "Industrial", "Meadow", "MediumResidential", "Mountain", "Park", "Parking", "Playground", "Pond", "Port", "RailwayStation",
                          "Resort", "River", "School", "SparseResidential", "Square", "Stadium", "StorageTanks", "Viaduct"]}
IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


class ImageFolderInstance(torch.utils.data.Dataset):
    """lhrs/Dataset/ImageFolderInstance.py over torchvision's `ImageFolder` contract: `<root>/<class dir>/**/<picture>`; classes are the
    sorted directory names, samples are listed class by class in sorted walk order, pictures are opened as RGB.  Attributes `classes`,
    `class_to_idx`, `samples`, `imgs`, `targets`, and `CLASS_NAME` (the dataset's published names, e.g. AID's 30)."""

    def __init__(self, dataset_name: str, return_index: bool = True, root=None, transform: Callable = None, target_transform: Callable = None):
        assert dataset_name in CLASS_NAME_MAP, "dataset name must be in {}".format(CLASS_NAME_MAP.keys())
        self.root, self.transform, self.target_transform = str(root), transform, target_transform
        self.classes = sorted(e.name for e in os.scandir(self.root) if e.is_dir())
        if not self.classes:
            raise FileNotFoundError(f"Couldn't find any class folder in {self.root}.")
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples: List[Tuple[str, int]] = []
        empty = []
        for c in self.classes:
            n0 = len(self.samples)
            for d, _, files in sorted(os.walk(os.path.join(self.root, c), followlinks=True)):
                self.samples += [(os.path.join(d, f), self.class_to_idx[c]) for f in sorted(files) if f.lower().endswith(IMG_EXTENSIONS)]
            if len(self.samples) == n0:
                empty.append(c)
        if empty:
            raise FileNotFoundError(f"Found no valid file for the classes {', '.join(empty)}. Supported extensions are: {', '.join(IMG_EXTENSIONS)}")
        self.imgs = self.samples
        self.targets = [t for _, t in self.samples]
        self.CLASS_NAME = CLASS_NAME_MAP[dataset_name]
        self.return_index = return_index

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        from PIL import Image
        path, target = self.samples[index]
        with open(path, "rb") as f:
            img = _decoded(Image.open(f).convert("RGB"), self.transform)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return (img, target, index) if self.return_index else (img, target)


class METERMLDataset(torch.utils.data.Dataset):
    """METER-ML methane-source facilities (lhrs/Dataset/meterml.py): `<root>/<split>.geojson` lists `Image_Folder` ("<dir>/<id>") and
    the label `idx` of every record; pictures are `<root>/<split>_images/<id>/naip.png` (mode naip_rgb) or `sentinel-2-10m.npy`
    (mode s2_rgb: first three bands / 10000, float32)."""

    CLASS_NAME = ["Other", "concentrated animal feeding operations", "landfills", "coal mines", "natural gas processing plants",
                  "refineries and petroleum terminals", "wastewater treatment plants"]
    class_dict = {"Negative": 0, "CAFOs": 1, "Landfills": 2, "Mines": 3, "ProcPlants": 4, "R&Ts": 5, "WWTPs": 6}

    def __init__(self, root, split: str, mode: str, transform: Callable = None):
        assert split.lower() in ["train", "test", "val"]
        assert mode.lower() in ["naip_rgb", "s2_rgb"], "%s is not implemented currently." % (mode.lower())
        import numpy as np
        self.root = Path(root)
        self.split, self.mode, self.transform = split, mode.lower(), transform
        self.img_dir = self.root / (split + "_images")
        props = [f["properties"] for f in json.loads((self.root / (split + ".geojson")).read_text())["features"]]
        self.image_folder = np.array([p["Image_Folder"].split("/")[1] for p in props], dtype=object)
        self.idx = np.array([p["idx"] for p in props])
        assert self.image_folder.size == self.idx.size, "The length of label is not match with those of images folder"

    def __len__(self) -> int:
        return self.image_folder.size

    def __getitem__(self, index):
        folder, label = self.image_folder[index], self.idx[index]
        if self.mode == "naip_rgb":
            from PIL import Image
            img = Image.open(self.img_dir / folder / "naip.png").convert("RGB")
            if self.transform is not None:
                img = _decoded(img, self.transform)
        else:
            import numpy as np
            img = np.load(self.img_dir / folder / "sentinel-2-10m.npy")[:, :, :3].astype(np.float32) / 10000
            if self.transform is not None:
                img = self.transform(img)
        return img, label


def build_cls_transform(config, is_train: bool = True):
    """lhrs/Dataset/build_transform.py:9-40.  Evaluation: Resize(256, BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize(ImageNet) as
    ONE device transform.  Training augmentation (timm `create_transform`: RandAugment, random erasing) feeds classifier fine-tuning,
    which is not on this path (out of scope) - asked for, it fails loudly."""
    if is_train:
        raise NotImplementedError("build_cls_transform(is_train=True) is timm's augmentation pipeline for classifier training - not part of the LHRS-Bot "
                                  "hot path this engine replaces (SURVEY.md §8)")
    size = (config.get("transform") or {}).get("input_size", (224, 224)) if hasattr(config, "get") else (224, 224)
    return ClsEvalTransformHIP(device=torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cuda", input_size=size)


class DeviceBatchLoader:
    """A DataLoader whose batches of decoded pictures (uint8 HWC, stacked or a list) pass through a `DeviceImageTransform` as they are
    fetched: iterating yields `(pixel_values float32 [B, 3, 224, 224] on the device, target tensor)` like the reference's loader does
    (there the transform ran per picture in the workers).  `.dataset`, `len()`, `.sampler`, `.batch_size` are the wrapped loader's."""

    def __init__(self, loader: torch.utils.data.DataLoader, transform: DeviceImageTransform):
        self.loader, self.transform = loader, transform
        self.dataset, self.sampler, self.batch_size = loader.dataset, loader.sampler, loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, *rest in self.loader:
            yield (self.transform.preprocess(images)["pixel_values"], *rest)


def _decoded_collate(batch):
    """default collation for every field but the pictures, which stay a list unless they all have one shape"""
    images = [b[0] for b in batch]
    if all(torch.is_tensor(x) and x.shape == images[0].shape for x in images):
        images = torch.stack(images)
    return (images, *[torch.utils.data.default_collate([b[i] for b in batch]) for i in range(1, len(batch[0]))])


def build_zero_shot_loader(config, mode: str = "zero_shot_cls"):
    """lhrs/Dataset/build_loader.py:164-199: the classification loader named by `config.eval.dataset` (UCM / METERML / an ImageFolder
    dataset such as AID) over `config.data_path`, evaluation transform, sequential order."""
    assert mode in ["zero_shot_cls", "zero_shot_retrieval"], "Please choose mode for dataloder from [zero_shot_cls, zero_shot_retrieval]"
    if mode != "zero_shot_cls":
        raise NotImplementedError("Zero-shot retrieval not implemented")
    transform = build_cls_transform(config, is_train=False)
    name = config["eval"]["dataset"]
    if name == "UCM":
        dataset = UCM(config.data_path, split="all", transform=transform, return_idx=False)
    elif name == "METERML":
        dataset = METERMLDataset(root=config.data_path, split="test", mode="naip_rgb", transform=transform)
    else:
        dataset = ImageFolderInstance(dataset_name=name, return_index=False, root=config.data_path, transform=transform)
    loader = build_loader_hepler(config, dataset, collate_fn=_decoded_collate, is_train=False)
    logger.info(f"Build dataloader: Epoch length = {len(loader)}")
    return DeviceBatchLoader(loader, transform)


# ------------------------------------------------------------------------------------------------ RSVQA
class Compose:
    """lhrs/Dataset/rsvqa.py:18-31: a transform list that maps over sequences element-wise."""

    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = [t(i) for i in x] if isinstance(x, Sequence) else t(x)
        return x


class ToTensor:
    """rsvqa.py:34-59: numpy HWC -> torch CHW without rescaling (uint16 widened to int32)."""

    def __init__(self, permute_dims: bool = True):
        self.permute_dims = permute_dims

    def __call__(self, x):
        import numpy as np
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(x.astype("int32") if x.dtype == "uint16" else x)
        if x.ndim == 2:
            x = x[:, :, None] if self.permute_dims else x[None, :, :]
        if self.permute_dims:
            x = (x.permute((0, 3, 1, 2)) if x.ndim == 4 else x.permute((2, 0, 1))).contiguous()
        return x


def _stem_number(path: str) -> int:
    return int(os.path.splitext(os.path.basename(path))[0])


class RSVQA(torch.utils.data.Dataset):
    """RSVQA test questions as generation prompts (lhrs/Dataset/rsvqa.py:68-196).  `<root>/<prefix>_split_<split>_{questions,answers,
    images}.json`, pictures `<root>/<image_root>/<image id>.tif`.  Of every active image the questions whose type is not "count" /
    "area" are kept (the reference scores presence / comparison / rural-urban only).  A sample: x (picture), question (prompt ids of
    `token_prefix + question` in the `prompt_type` template, answer slot open), answer, type, questions_idx."""

    splits = ["train", "val", "test"]
    prefix = ""
    neglect_question_type = ("count", "area")

    def __init__(self, root: str = "", image_root: str = None, split: str = "train", image_transform=None, text_transform=None, token_prefix: str = "",
                 tokenizer: Callable = None, **kwargs):
        assert split in self.splits
        prompt_type = kwargs.pop("prompt_type", "llava_llama_2")
        conversation_lib.default_conversation = conversation_lib.conv_templates[prompt_type]
        self.root, self.split = root, split
        self.image_transform = image_transform if image_transform is not None else Compose([ToTensor()])
        self.text_transform = text_transform if text_transform is not None else Compose([])
        self.image_root = os.path.join(root, image_root)
        self.token_prefix = token_prefix
        self.tune_im_start = kwargs.pop("tune_im_start", False)
        self.tokenizer = tokenizer
        self.ids, self.paths, self.images, self.questions, self.answers = self.load_files(self.root, self.image_root, self.split, self.prefix)
        self.post_process()

    @staticmethod
    def load_files(root: str, image_root: str, split: str, prefix: str):
        paths = sorted(glob(os.path.join(image_root, "*.tif")), key=_stem_number)
        read = lambda what: json.load(open(os.path.join(root, f"{prefix}_split_{split}_{what}.json")))[what]  # noqa: E731
        questions, answers, images = read("questions"), read("answers"), read("images")
        return [x["id"] for x in images if x["active"]], paths, images, questions, answers

    def post_process(self):
        ids, qids = [], []
        for image_id in self.ids:
            keep = [q for q in self.images[image_id]["questions_ids"] if self.questions[q]["type"].lower() not in self.neglect_question_type]
            qids += keep
            ids += [image_id] * len(keep)
        self.questions_ids, self.ids = qids, ids

    def __len__(self) -> int:
        return len(self.ids)

    def __getitem__(self, idx: int) -> Dict:
        import numpy as np
        from PIL import Image
        x = np.array(Image.open(os.path.join(self.image_root, f"{self.ids[idx]}.tif")))
        if isinstance(self.image_transform, DeviceImageTransform):
            x = torch.from_numpy(x)  # uint8 [H, W, 3]: the device finishes the transform per batch
        else:
            x = self.image_transform(x)
        q = self.questions[self.questions_ids[idx]]
        answer = self.text_transform(self.answers[q["answers_ids"][0]]["answer"])
        turn = dict(Question=self.token_prefix + self.text_transform(q["question"]), Answer=None)
        ids = preprocess(preprocess_multimodal(turn, tune_im_start=self.tune_im_start), self.tokenizer, has_image=True)["input_ids"][0]
        return dict(x=x, question=ids, answer=answer, type=q["type"], questions_idx=self.questions_ids[idx])


class RSVQALR(RSVQA):
    prefix = "LR"

    def __init__(self, root: str = ".data/RSVQA_LR", *args, **kwargs):
        super().__init__(root, *args, **kwargs)


class RSVQAHR(RSVQA):
    prefix = "USGS"

    def __init__(self, root: str = ".data/RSVQA_HR", *args, **kwargs):
        super().__init__(root, *args, **kwargs)


class RSVQAxBEN(RSVQA):
    prefix = "RSVQAxBEN"

    def __init__(self, root: str = ".data/rsvqaxben", *args, **kwargs):
        super().__init__(root, *args, **kwargs)


def left_pad(sequences, pad: int, max_length: int) -> torch.Tensor:
    """Prompts of different lengths -> one [B, n] id matrix padded on the LEFT (generation continues every row at its last column)."""
    rows = [s.tolist() if torch.is_tensor(s) else list(s) for s in sequences]
    n = max(len(r) for r in rows)
    return torch.tensor([[pad] * (n - len(r)) + r for r in rows])[:, :max_length]


class DataCollatorForVQASupervisedDataset:
    """rsvqa.py:208-262: dict(images, questions [B, n] left-padded, attn_mask, targets, types, questions_idx)."""

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, instances: Sequence[Dict]) -> Dict:
        questions = left_pad([i["question"] for i in instances], self.tokenizer.pad_token_id, self.tokenizer.model_max_length)
        images = [i["x"] for i in instances]
        if all(torch.is_tensor(x) and x.shape == images[0].shape for x in images):
            images = torch.stack(images)
        return dict(images=images, questions=questions, attn_mask=questions.ne(self.tokenizer.pad_token_id), targets=[i["answer"] for i in instances],
                    types=[i["type"] for i in instances], questions_idx=[i["questions_idx"] for i in instances])


# ------------------------------------------------------------------------------------------------ visual grounding / caption evaluation
class VGEvalDataset(CaptionDataset):
    """cap_dataset.py:186-260.  `target` is ONE json `{"data": [...]}` whose stem picks the record schema: `*RSVG_test` (img, question,
    answer), `*DIOR_test` (img without ".jpg", question, answer), anything else (name, conv, answer).  The first question gets
    "<image>" in front; a sample is (picture, prompt ids with the answer slot open, target string, file name)."""

    def __init__(self, root=".data/rsicd", target=None, transform=None, tokenizer=None, **kwargs):
        prompt_type = kwargs.pop("prompt_type", "llava_llama_2")
        conversation_lib.default_conversation = conversation_lib.conv_templates[prompt_type]
        self.transform = transform
        self.img_dir, self.json_dir = Path(root), Path(target)
        self.img_list: List[Path] = []
        self.prompt_list: List = []
        self.target_list: List[str] = []
        self.tokenizer = tokenizer
        self.tune_im_start = kwargs.pop("tune_im_start", False)
        self.load_dataset()
        self.post_process()

    def load_dataset(self):
        stem = self.json_dir.stem
        for item in json.loads(self.json_dir.read_bytes())["data"]:
            if stem.endswith("RSVG_test") or stem.endswith("DIOR_test"):
                path = self.img_dir / (item["img"] if stem.endswith("RSVG_test") else item["img"] + ".jpg")
                item["conv"] = dict(Question=item["question"], Answer=None)
            else:
                path = self.img_dir / item["name"]
            if valid_path(path):
                self.img_list.append(path)
                self.prompt_list.append(item["conv"])
                self.target_list.append(item["answer"])

    def post_process(self):
        for i, conv in enumerate(self.prompt_list):
            conv = conv if isinstance(conv, list) else [conv]
            conv[0]["Question"] = "<image>" + conv[0]["Question"]
            self.prompt_list[i] = conv

    def __len__(self):
        return len(self.target_list)

    def __getitem__(self, idx: int):
        prompt = preprocess(preprocess_multimodal(self.prompt_list[idx], tune_im_start=self.tune_im_start), self.tokenizer, has_image=True)
        return self.load_image(idx), prompt["input_ids"][0], self.target_list[idx], self.img_list[idx].name


class CapEvalDataset(CaptionDataset):
    """cap_dataset.py:263-327: ONE image directory + ONE annotation json (schema by directory name, first caption of every picture);
    a sample adds `filename` and `raw_image` (uint8 CHW, the untransformed picture) to CaptionDataset's {"rgb", "text"}."""

    def __init__(self, root=".data/rsicd", target=None, transform=None):
        self.transform = transform
        self.img_dir, self.json_dir = Path(root), Path(target)
        self.img_list, self.cap_list = [], []
        self.load_dataset()
        self.post_process()

    def load_dataset(self):
        from .datasets import _nwpu, _rsicd_like, _textrs, _uavicd
        data = json.loads(self.json_dir.read_bytes())
        stem = self.img_dir.stem
        reader = _textrs if "TextRS" in stem else _uavicd if "UAVICD" in stem else _nwpu if "NWPU" in stem else _rsicd_like
        self._add(reader(data, self.img_dir))

    def __getitem__(self, idx: int) -> Dict:
        import numpy as np
        from PIL import Image
        out = super().__getitem__(idx)
        out["filename"] = self.img_list[idx].name
        out["raw_image"] = torch.from_numpy(np.array(Image.open(self.img_list[idx]).convert("RGB"), copy=True)).permute(2, 0, 1).contiguous()
        return out
