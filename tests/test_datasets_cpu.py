"""CPU: the real-data loaders against tests/golden/datasets.json - the REFERENCE's CaptionDatasetVQA / InstructDataset
run over the same synthetic corpora (tests/dataset_cases.py, make_golden_datasets.py).  Token ids / labels / batch tensors int-exact."""
import json
import os
import random
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import dataset_cases as DC  # noqa: E402

from lhrs_bot_amd import datasets as DS  # noqa: E402
from lhrs_bot_amd.data import DataCollatorForSupervisedDataset  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
Z = json.load(open(os.path.join(G, "datasets.json")))


@pytest.mark.parametrize("case", sorted(Z["cases"]))
def test_dataset_matches_reference_class(case, tmp_path):
    want = Z["cases"][case]
    kw = DC.build_case(str(tmp_path / case), case)
    tok = DC.ToyTok()
    random.seed(Z["seed"])
    ds = getattr(DS, want["cls"])(tokenizer=tok, prompt_type=want["prompt_type"], transform=None, **kw)
    assert len(ds) == want["n"]
    for i, row in enumerate(want["rows"]):
        s = ds[i]
        assert ds.img_list[i].name == row["file"] and list(s["rgb"].size) == row["size"]
        assert s["text"]["input_ids"].tolist() == row["ids"], (case, i)
        assert s["text"]["labels"].tolist() == row["labels"], (case, i)
        if "valid_image" in row:
            assert s["valid_image"] == row["valid_image"]
    inst = [dict(ds[i], rgb=torch.zeros(3, 2, 2)) for i in range(min(4, len(ds)))]
    b = DataCollatorForSupervisedDataset(tok)(inst)
    assert {k: v.tolist() for k, v in b.items() if k != "rgb"} == want["batch"]


def test_pre_caption_rule():
    assert DS.pre_caption("An  Airport (big); two RUNWAYS!\n") == "an airport big two runways"
    assert DS.pre_caption(" ".join(["W"] * 70)).count("w") == 50
    assert DS.pre_caption([{"Question": "q"}]) == [{"Question": "q"}]


def test_build_loader_stage1_batches_uint8_pixels_and_samplers(tmp_path):
    """build_loader(config, mode="pretrain", tokenizer=..., prompt_type=...): the stage-1 loader over a directory.  On a box without a
    GPU the transform object cannot be built (no CPU path) - the datasets are then exercised with an explicit stand-in transform."""
    from lhrs_bot_amd.data import CLIPImageProcessorHIP
    from lhrs_bot_amd.trainer import ConfigDict
    DC.build_case(str(tmp_path / "d"), "rsicd")
    tok = DC.ToyTok()

    class Deferred(CLIPImageProcessorHIP):  # same type => same "decode only" behaviour in the workers, no device needed to construct
        def __init__(self):
            pass

    ds = DS.CaptionDatasetVQA(tokenizer=tok, prompt_type="plain", transform=Deferred(), root=str(tmp_path / "d"))
    s = ds[2]
    assert s["rgb"].dtype == torch.uint8 and tuple(s["rgb"].shape) == (300, 260, 3)
    cfg = ConfigDict(dict(batch_size=2, workers=2, is_distribute=False, inf_sampler=False))
    loader = DS.build_loader_hepler(cfg, ds, collate_fn=DataCollatorForSupervisedDataset(tok), is_train=True)
    assert len(loader) == 3
    batches = list(loader)
    assert len(batches) == 3 and all(set(b) == {"input_ids", "labels", "attention_mask", "rgb"} for b in batches)
    for b in batches:  # pictures of different sizes stay a list of uint8 HWC tensors (the device kernel resizes them)
        assert (isinstance(b["rgb"], list) and all(x.dtype == torch.uint8 for x in b["rgb"])) or b["rgb"].dtype == torch.uint8
        assert b["input_ids"][:, 1].eq(-200).all() and b["labels"][:, :2].eq(-100).all()
    cfg.inf_sampler = True
    inf = DS.build_loader_hepler(cfg, ds, collate_fn=DataCollatorForSupervisedDataset(tok), is_train=True)
    it = iter(inf)
    seen = [next(it)["input_ids"].shape[0] for _ in range(7)]  # more batches than one pass holds: the stream does not end
    assert seen == [2] * 7
    smp = DS.InfiniteSampler(ds, shuffle=True, seed=5)
    first = [i for _, i in zip(range(12), iter(smp))]
    assert sorted(first[:6]) == list(range(6)) and sorted(first[6:]) == list(range(6))
    with pytest.raises(NotImplementedError):
        DS.build_loader(ConfigDict(dict(data_path="/x/RS5M", stage=1, batch_size=2, rgb_vision={"arch": "vit_large"})), mode="pretrain", tokenizer=tok)
