"""tests/golden/datasets.json: the REFERENCE's CaptionDatasetVQA / InstructDataset (lhrs/Dataset/cap_dataset.py:330-486) and collator
run over the synthetic corpora of tests/dataset_cases.py with the toy tokenizer: per sample file name, image size, input_ids, labels
(python `random` seeded before each dataset is built: the question templates / long-conversation sub-sampling draw from it).
Build container only; same import shims as make_golden_data.py."""
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: F401,E402  (before the stub modules exist: torch inspects sys.modules while importing)
import transformers  # noqa: F401,E402

for name in ("webdataset", "webdataset.filters", "webdataset.tariterators", "torchvision", "torchvision.transforms", "braceexpand"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["webdataset.filters"]._shuffle = None
for n in ("base_plus_ext", "tar_file_expander", "url_opener", "valid_sample"):
    setattr(sys.modules["webdataset.tariterators"], n, None)
sys.modules["webdataset"].filters = sys.modules["webdataset.filters"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, n):
        return _Any()


sys.modules["torchvision.transforms"].__getattr__ = lambda n: _Any
sys.modules["webdataset"].__getattr__ = lambda n: _Any
sys.modules["braceexpand"].braceexpand = None
for name, path in [("lhrs", f"{REF}/lhrs"), ("lhrs.Dataset", f"{REF}/lhrs/Dataset"), ("lhrs.models", f"{REF}/lhrs/models")]:
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
mm = sys.modules["lhrs.models"]
mm.DEFAULT_IM_END_TOKEN, mm.DEFAULT_IM_START_TOKEN, mm.DEFAULT_IMAGE_PATCH_TOKEN = "<im_end>", "<im_start>", "<im_patch>"
mm.DEFAULT_IMAGE_TOKEN, mm.IGNORE_INDEX, mm.IMAGE_TOKEN_INDEX = "<image>", -100, -200
import lhrs.Dataset.cap_dataset as cd  # noqa: E402

import dataset_cases as DC  # noqa: E402

tok = DC.ToyTok()
out = {"seed": 1234, "cases": {}}
with tempfile.TemporaryDirectory() as tmp:
    for cls_name, cases in (("CaptionDatasetVQA", DC.STAGE1), ("InstructDataset", DC.STAGE2), ("InstructDatasetWithTaskId", DC.STAGE3)):
        for case, prompt in cases:
            kw = DC.build_case(os.path.join(tmp, case), case)
            random.seed(out["seed"])
            ds = getattr(cd, cls_name)(tokenizer=tok, prompt_type=prompt, transform=None, **kw)
            rows = []
            for i in range(len(ds)):
                s = ds[i]
                if i >= len(ds.img_list):  # text-only sample of the weighted mixture: zero picture, no file
                    rows.append({"file": None, "size": list(s["rgb"].shape), "ids": s["text"]["input_ids"].tolist(), "labels": s["text"]["labels"].tolist(),
                                 "valid_image": s["valid_image"]})
                    continue
                rows.append({"file": ds.img_list[i].name, "size": list(s["rgb"].size), "ids": s["text"]["input_ids"].tolist(),
                             "labels": s["text"]["labels"].tolist(), **({"valid_image": s["valid_image"]} if "valid_image" in s else {})})
            # the collated batch of the first (up to) four samples, images replaced by stand-in tensors (PIL images stay a list)
            import torch
            inst = [dict(ds[i], rgb=torch.zeros(3, 2, 2)) for i in range(min(4, len(ds)))]
            b = cd.DataCollatorForSupervisedDataset(tokenizer=tok)(inst)
            out["cases"][case] = {"cls": cls_name, "prompt_type": prompt, "n": len(ds), "rows": rows,
                                  "batch": {k: v.tolist() for k, v in b.items() if k != "rgb"}}
            if hasattr(ds, "sample_weight"):
                out["cases"][case]["sample_weight"] = list(ds.sample_weight)
            print(case, cls_name, prompt, len(ds), [len(r["ids"]) for r in rows])
json.dump(out, open(os.path.join(HERE, "datasets.json"), "w"))
print(os.path.getsize(os.path.join(HERE, "datasets.json")) // 1024, "KiB")
