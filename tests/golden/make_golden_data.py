"""Data-boundary golden: the reference's own tokenizer_image_token / preprocess_plain / DataCollatorForSupervisedDataset
(lhrs/Dataset/cap_dataset.py) driven with a deterministic toy tokenizer.  Build container only (heavy imports are stubbed)."""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import transformers  # noqa: F401

for name in ("webdataset", "webdataset.filters", "webdataset.tariterators", "torchvision", "torchvision.transforms", "braceexpand"):
    m = types.ModuleType(name)
    sys.modules[name] = m
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


sys.modules["torchvision.transforms"].__getattr__ = lambda n: _Any  # module-level annotations / default args only
sys.modules["webdataset"].__getattr__ = lambda n: _Any                # base classes of pipeline stages we never build
sys.modules["braceexpand"].braceexpand = None
for name, path in [("lhrs", f"{REF}/lhrs"), ("lhrs.Dataset", f"{REF}/lhrs/Dataset"), ("lhrs.models", f"{REF}/lhrs/models")]:
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
mm = sys.modules["lhrs.models"]
mm.DEFAULT_IM_END_TOKEN, mm.DEFAULT_IM_START_TOKEN, mm.DEFAULT_IMAGE_PATCH_TOKEN = "<im_end>", "<im_start>", "<im_patch>"
mm.DEFAULT_IMAGE_TOKEN, mm.IGNORE_INDEX, mm.IMAGE_TOKEN_INDEX = "<image>", -100, -200
try:
    import lhrs.Dataset.cap_dataset as cd
except Exception as e:  # show what else needs a stub
    raise SystemExit(f"import failed: {e!r}")

from lhrs_bot_amd.trainer import ConfigDict  # noqa: E402


class ToyTok:
    """whitespace tokenizer: BOS + one id per word (stable hash), '\\n' kept as its own token"""
    bos_token_id, pad_token_id, unk_token_id, model_max_length = 1, 0, 0, 24

    def __call__(self, text):
        ids = [self.bos_token_id]
        for w in text.replace("\n", " \n ").split(" "):
            if w:
                ids.append(13 if w == "\n" else 3 + sum(ord(c) * (i + 7) for i, c in enumerate(w)) % 31000)
        return ConfigDict({"input_ids": ids})


tok = ToyTok()
prompts = ["<image>a river next to farmland\n", "hello <image> world", "<image>", "no image here", "<image>\nDescribe it. <image> again"]
out = {"prompts": prompts, "tit": [cd.tokenizer_image_token(p, tok) for p in prompts]}
sources = [{"Question": "<image>\nWhat is this?", "Answer": "an airport with two runways"}, {"Question": "Look <image>", "Answer": "dense forest"},
           {"Question": "<image>", "Answer": " ".join(["w%d" % i for i in range(40)])}]
import copy
cd.conversation_lib.default_conversation = cd.conversation_lib.conv_templates["plain"]  # what the stage-1 datasets set (cap_dataset.py:196,353,395)
out["plain_sep"] = cd.conversation_lib.default_conversation.sep
pp = cd.preprocess_plain(copy.deepcopy(sources), tok)
out["sources"] = sources
out["pp_ids"] = [t.tolist() for t in pp["input_ids"]]
out["pp_labels"] = [t.tolist() for t in pp["labels"]]
coll = cd.DataCollatorForSupervisedDataset(tokenizer=tok)
inst = [{"text": {"input_ids": pp["input_ids"][i], "labels": pp["labels"][i]}, "rgb": torch.full((3, 2, 2), float(i)), "valid_image": i % 2 == 0} for i in range(3)]
b = coll(inst)
out["coll"] = {k: v.tolist() for k, v in b.items()}
# ---- stages 2/3: llava_llama_2 conversations (preprocess_multimodal + preprocess_llama_2) and the evaluation collator
cd.conversation_lib.default_conversation = cd.conversation_lib.conv_templates["llava_llama_2"]
ToyTok.model_max_length = 512
l2_cases = [
    [{"Question": "What is in the picture? <image>", "Answer": "an airport with two runways"}],
    [{"Question": "<image>\nDescribe it.", "Answer": "dense forest"}, {"Question": "How many roads?", "Answer": "two roads cross"}],
    [{"Question": "no picture here", "Answer": "ok"}],
]
out["l2_sources"] = l2_cases
out["l2"] = []
for src in l2_cases:
    s2 = cd.preprocess_multimodal(copy.deepcopy(src), tune_im_start=False)
    r = cd.preprocess(copy.deepcopy(s2), tok, has_image=True)
    conv = cd.conversation_lib.default_conversation.copy()
    for s in s2:
        for j, k in enumerate(s):
            conv.append_message(conv.roles[j % 2], s[k])
    out["l2"].append({"mm": s2, "prompt": conv.get_prompt(), "ids": r["input_ids"].tolist(), "labels": r["labels"].tolist()})
ToyTok.model_max_length = 40     # shorter than the 2-turn sample: the "cur_len < model_max_length" guard is skipped
r = cd.preprocess(copy.deepcopy(out["l2"][1]["mm"]), tok, has_image=True)
out["l2_short_ctx"] = {"ids": r["input_ids"].tolist(), "labels": r["labels"].tolist()}
ToyTok.model_max_length = 24
vg = cd.DataCollatorForVGSupervisedDataset(tokenizer=tok)
inst = [(torch.full((3, 2, 2), float(i)), list(range(5, 5 + n)), "t%d" % i, "f%d.png" % i) for i, n in enumerate([4, 9, 30])]
images, ids, targets, names, mask = vg(inst)
out["vg"] = {"ids": ids.tolist(), "mask": mask.tolist(), "targets": targets, "names": names, "images_shape": list(images.shape)}
json.dump(out, open(os.path.join(HERE, "data_boundary.json"), "w"))
print("data golden:", [len(x) for x in out["tit"]], [len(x) for x in out["pp_ids"]], list(b.keys()), b["input_ids"].shape)
