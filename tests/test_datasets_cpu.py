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
    # a root with several corpora is listed in file-system order (the reference does not sort either): match samples by file name
    order = list(range(len(want["rows"])))
    if case == "taskid":
        names = [p.name for p in ds.img_list]
        order = [names.index(r["file"]) if r["file"] is not None else j for j, r in enumerate(want["rows"])]
        assert [ds.sample_weight[i] for i in order] == want["sample_weight"]
    for i, row in zip(order, want["rows"]):
        s = ds[i]
        if row["file"] is None:  # text-only sample of the weighted mixture
            assert i >= len(ds.img_list) and list(s["rgb"].shape) == row["size"] and float(s["rgb"].abs().sum()) == 0.0
        else:
            assert ds.img_list[i].name == row["file"] and list(s["rgb"].size) == row["size"]
        assert s["text"]["input_ids"].tolist() == row["ids"], (case, i)
        assert s["text"]["labels"].tolist() == row["labels"], (case, i)
        if "valid_image" in row:
            assert s["valid_image"] == row["valid_image"]
    inst = [dict(ds[i], rgb=torch.zeros(3, 2, 2)) for i in order[: min(4, len(ds))]]
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
    with pytest.raises(FileNotFoundError, match="no RS5M shard"):       # "RS5M" in the path selects the tar-shard loader (build_loader.py:66-68)
        DS.build_loader(ConfigDict(dict(data_path="/x/RS5M", stage=1, batch_size=2, rgb_vision={"arch": "vit_large"})), mode="pretrain", tokenizer=tok)


def test_weighted_stage3_loader_and_sampler_wrapper(tmp_path):
    """`weight_sample: True` (Config/multi_modal_stage3.yaml): WeightedRandomSampler without replacement = a weighted permutation of all
    samples, sharded by DistributedSamplerWrapper; batches may mix pictures with text-only samples (zero picture, no image token)."""
    from torch.utils.data import WeightedRandomSampler
    from lhrs_bot_amd.trainer import ConfigDict
    DC.build_case(str(tmp_path / "d"), "taskid")
    tok = DC.ToyTok()
    ds = DS.InstructDatasetWithTaskId(tokenizer=tok, prompt_type="llava_llama_2", transform=None, root=str(tmp_path / "d"))
    assert len(ds) == 8 and len(ds.sample_weight) == 8 and sorted(set(ds.sample_weight)) == [0.5, 0.9, 1.0]
    base = WeightedRandomSampler(ds.sample_weight, num_samples=len(ds), replacement=False)
    shards = []
    for r in range(2):
        w = DS.DistributedSamplerWrapper(base, num_replicas=2, rank=r)
        w.set_epoch(3)
        idx = list(w)
        assert len(idx) == 4 and all(0 <= i < 8 for i in idx)
        shards.append(idx)
    one = DS.DistributedSamplerWrapper(base, num_replicas=1, rank=0)
    assert sorted(one) == list(range(8))                       # one rank: every sample exactly once per pass

    class Deferred(DS.CLIPImageProcessorHIP):
        def __init__(self):
            pass

    import lhrs_bot_amd.datasets as mod
    orig = mod.build_vlp_transform
    mod.build_vlp_transform = lambda config, is_train=True: Deferred()
    try:
        cfg = ConfigDict(dict(data_path=str(tmp_path / "d"), stage=3, weight_sample=True, batch_size=3, workers=0, is_distribute=False,
                              transform={"input_size": [224, 224]}, rgb_vision={"arch": "vit_large"}))
        loader = DS.build_loader(cfg, mode="pretrain", tokenizer=tok, prompt_type="llava_llama_2")
        batches = list(loader)
    finally:
        mod.build_vlp_transform = orig
    assert len(batches) == 2 and all(b["input_ids"].shape[0] == 3 and "valid_image" in b for b in batches)   # 8 samples, drop_last
    for b in batches:
        has_img = b["input_ids"].eq(-200).any(dim=1)
        assert torch.equal(has_img, b["valid_image"])


# ------------------------------------------------------------------------------------------------ RS5M tar shards
def _make_rs5m_shards(root, n_shards=4, per_shard=5):
    """`<root>/{pub11,rs3}-train-000i.tar` with the member naming of RS5M (`<key>.img_content`, `<key>.img_name`, `<key>.caption`)."""
    import io
    import tarfile
    from PIL import Image
    os.makedirs(root, exist_ok=True)
    truth = {}
    for src in ("pub11", "rs3"):
        for s in range(n_shards // 2):
            path = os.path.join(root, f"{src}-train-{s:04d}.tar")
            with tarfile.open(path, "w") as tf:
                for i in range(per_shard):
                    key = f"{src}_{s}_{i:03d}"
                    buf = io.BytesIO()
                    Image.new("RGB", (8 + i, 6 + s), color=(10 * i, 20 * s, 7)).save(buf, format="PNG")
                    cap = f"An Aerial (view) of AREA {src} {s} {i}!"
                    members = [(key + ".img_content", buf.getvalue()), (key + ".img_name", (key + ".png").encode()), (key + ".caption", cap.encode())]
                    if i == 2:   # a stray member without the grouping key pattern and a sample without caption: both are skipped, not errors
                        members.append((f"{key}_nocap.img_content", buf.getvalue()))
                    for name, data in members:
                        ti = tarfile.TarInfo(name)
                        ti.size = len(data)
                        tf.addfile(ti, io.BytesIO(data))
                    truth[key] = (8 + i, 6 + s, DS.pre_caption(cap))
    return truth


def test_rs5m_tar_shards_grouping_split_and_batches(tmp_path, monkeypatch):
    """RS5MDataset (cap_dataset.py:649-775 restated on `tarfile`): brace expansion of the shard pattern, key grouping, every sample of
    every shard exactly once per epoch across (rank, worker) splits, captions normalised and tokenised as a one-turn conversation about
    the image, batches of exactly `batch_size` through the supervised collator (partial batches dropped)."""
    from lhrs_bot_amd import conversation as conversation_lib
    monkeypatch.setattr(conversation_lib, "default_conversation", conversation_lib.default_conversation)   # the datasets re-bind this module global
    root = str(tmp_path / "RS5M")
    truth = _make_rs5m_shards(root)
    assert DS.expand_braces("x/{pub11,rs3}-train-{0000..0031}.tar")[33] == "x/rs3-train-0001.tar"
    one = list(DS.tar_samples(os.path.join(root, "pub11-train-0000.tar")))
    assert [s["__key__"] for s in one][:3] == ["pub11_0_000", "pub11_0_001", "pub11_0_002"] and set(one[0]) == {"__key__", "__url__", "img_content", "img_name", "caption"}
    tok = DC.ToyTok()
    shards = [os.path.join(root, f"{s}-train-{i:04d}.tar") for s in ("pub11", "rs3") for i in range(2)]
    seen = []
    for rank in range(2):
        ds = DS.RS5MDataset(root=root, transform=None, tokenizer=tok, prompt_type="plain", rank=rank, world_size=2, shards=shards)
        random.seed(5)
        rows = list(ds)
        assert len(ds.my_shards()) == 2
        for r in rows:
            w, h = r["rgb"].size
            ids = r["text"]["input_ids"].tolist()
            assert ids[0] == tok.bos_token_id and ids[1] == -200 and r["text"]["labels"].tolist()[:2] == [-100, -100]
            seen.append((w, h))
    assert sorted(seen) == sorted((w, h) for w, h, _ in truth.values())          # 20 samples, each once, the caption-less member dropped
    ds = DS.RS5MDataset(root=root, transform=lambda im: torch.zeros(3, 4, 4), tokenizer=tok, prompt_type="plain", batch_size=3, rank=0, world_size=1, shards=shards)
    batches = list(ds)
    assert len(batches) == 20 // 3 and all(b["rgb"].shape == (3, 3, 4, 4) and b["input_ids"].shape[0] == 3 for b in batches)
    e0 = [b["input_ids"].tolist() for b in batches]
    e1 = [b["input_ids"].tolist() for b in ds]                                        # next epoch: another shard / sample order
    assert sorted(map(str, sum(e0, []))) != [] and e0 != e1
    from lhrs_bot_amd.trainer import ConfigDict
    cfg = ConfigDict(dict(data_path=root, batch_size=2, workers=0, world_size=1, stage=1))
    loader = DS.build_rs5m_loader(cfg, lambda im: torch.zeros(3, 4, 4), tokenizer=tok, prompt_type="plain")
    assert loader.num_batches == -(-DS.RS5M_NUM_SAMPLES // 2) and len(loader.dataset.shards) == 4
    b = next(iter(loader))
    assert set(b) >= {"rgb", "input_ids", "labels", "attention_mask"} and b["input_ids"].shape[0] == 2


def test_rs5m_loader_has_an_epoch_length_and_every_rank_yields_exactly_that_many_batches(tmp_path, monkeypatch):
    """`dataset.with_epoch(num_worker_batches)` (build_loader.py:131-142): every worker of every rank yields exactly
    ceil(num_batches / workers) batches per epoch, walking its shards again when they run dry, so `len(loader)` exists (the trainer's
    `epoch_len`) and two ranks whose shards hold different numbers of samples still leave the epoch at the same step."""
    from lhrs_bot_amd import conversation as conversation_lib
    from lhrs_bot_amd.trainer import ConfigDict
    monkeypatch.setattr(conversation_lib, "default_conversation", conversation_lib.default_conversation)
    root = str(tmp_path / "RS5M")
    _make_rs5m_shards(root)                                             # 4 shards of 5 samples (one shard has a caption-less member)
    tok = DC.ToyTok()
    counts = []
    for rank in range(2):
        cfg = ConfigDict(dict(data_path=root, batch_size=2, workers=2, world_size=2, stage=1, rs5m_num_samples=44))
        loader = DS.build_rs5m_loader(cfg, lambda im: torch.zeros(3, 4, 4), tokenizer=tok, prompt_type="plain", rank=rank, world_size=2)
        assert len(loader) == loader.num_batches == 12 and loader.dataset.worker_batches == 6    # ceil(44 / 4) = 11 -> 2 workers x 6
        loader.set_epoch(0)
        n0 = sum(1 for _ in loader)
        loader.set_epoch(1)                                             # persistent workers read the epoch from the shared value
        n1 = sum(1 for _ in loader)
        counts.append((n0, n1))
        del loader
    assert counts == [(12, 12), (12, 12)]                              # each worker holds ONE 5-sample shard = 2 batches per pass: it re-walks it
