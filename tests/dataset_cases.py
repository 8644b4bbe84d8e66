"""Synthetic image + annotation corpora shared by tests/golden/make_golden_datasets.py (which runs the REFERENCE's dataset classes
on them) and the parity tests: directory layout `<root>/<NAME>_Image/` + `<root>/<NAME>.json`, one root per case so that the order
in which a file system lists the corpora cannot matter."""
import json
import os

import numpy as np


class ToyTok:
    """whitespace tokenizer: BOS + one id per word (stable hash), '\\n' kept as its own token"""
    bos_token_id, pad_token_id, unk_token_id, eos_token_id, model_max_length = 1, 0, 0, 2, 512

    def __call__(self, text):
        ids = [self.bos_token_id]
        for w in text.replace("\n", " \n ").split(" "):
            if w:
                ids.append(13 if w == "\n" else 3 + sum(ord(c) * (i + 7) for i, c in enumerate(w)) % 31000)
        return type("Enc", (), {"input_ids": ids})()

    def __len__(self):
        return 32000


def _png(path, seed, h, w):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path)


SIZES = [(240, 320), (224, 224), (300, 260), (256, 256), (230, 410), (512, 384)]
CAPS = ["An airport with two runways; several planes are parked.", "Dense   forest (green) next to a river!", "a small harbour: boats, piers",
        "Residential blocks ~ with # many * trees", "bare land.", " ".join(f"word{i}" for i in range(60))]


def build_case(root, case):
    """-> dataset kwargs.  Cases: caption corpora 'rsicd' (default schema), 'nwpu', 'textrs', 'llava' (stage 1) and instruction
    corpora 'instruct' (name/filename records, a 12-turn conversation), 'rsvg' (grounding records) for stages 2/3."""
    os.makedirs(root, exist_ok=True)
    J = lambda name, obj: json.dump(obj, open(os.path.join(root, name), "w"))  # noqa: E731
    if case == "rsicd":
        for i, (h, w) in enumerate(SIZES):
            _png(os.path.join(root, "RSICD_Image", f"im{i}.png"), i, h, w)
        J("RSICD.json", {"images": [{"filename": f"im{i}.png", "sentences": [{"raw": CAPS[i]}, {"raw": "unused"}]} for i in range(6)]
                         + [{"filename": "missing.png", "sentences": [{"raw": "skipped: no such file"}]}]})
    elif case == "nwpu":
        for i, (h, w) in enumerate(SIZES[:4]):
            _png(os.path.join(root, "NWPU_Image", "ab"[i % 2], f"n{i}.png"), 10 + i, h, w)
        J("NWPU.json", {"a": [{"filename": "n0.png", "raw": CAPS[0]}, {"filename": "n2.png", "raw": CAPS[2]}],
                        "b": [{"filename": "n1.png", "raw": CAPS[1]}, {"filename": "n3.png", "raw": CAPS[3]}]})
    elif case == "textrs":
        for i, (h, w) in enumerate(SIZES[:3]):
            _png(os.path.join(root, "TextRS_Image", f"t{i}.png"), 20 + i, h, w)
        J("TextRS.json", {"TextRS": [{"image": f"t{i}", "annotation": {"caption": [CAPS[i], "second"]}} for i in range(3)]})
    elif case == "llava":
        for i, (h, w) in enumerate(SIZES[:3]):
            _png(os.path.join(root, "LLAVA_Image", f"l{i}.png"), 30 + i, h, w)
        J("LLAVA.json", {"data": [
            {"name": "l0.png", "conv": [{"Question": "What is shown? <image>", "Answer": "a stadium"}]},
            {"name": "l1.png", "conv": [{"Question": "<image>\nDescribe.", "value": "two bridges over a river"}]},
            {"name": "l2.png", "conv": "a plain caption string inside the conversation corpus"}]})
    elif case == "instruct":
        for i, (h, w) in enumerate(SIZES[:4]):
            _png(os.path.join(root, "FAST_Image", f"f{i}.png"), 40 + i, h, w)
        long_conv = [{"Question": f"question {t} <image>" if t == 3 else f"question {t}", "Answer": f"answer {t}"} for t in range(12)]
        J("FAST.json", {"data": [
            {"name": "f0.png", "conv": [{"Question": "How many planes?", "Answer": "three planes"},
                                        {"Question": "Where? <image>", "Answer": "on the <image> apron"}]},
            {"filename": ["f1.png", "other.png"], "conv": {"Question": "<image>\nIs there water?", "Answer": "yes , a lake"}},
            {"name": "f2.png", "conv": long_conv},
            {"name": "f3.png", "conv": []},
            {"name": "gone.png", "conv": [{"Question": "q", "Answer": "a"}]}]})
    elif case == "rsvg":
        for i, (h, w) in enumerate(SIZES[:2]):
            _png(os.path.join(root, "XRSVG_Image", f"g{i}.jpg"), 50 + i, h, w)
        J("XRSVG.json", {"data": [{"img": f"g{i}.jpg", "question": f"[VG] locate the object number {i}", "answer": f"[{i}, 2, 30, 40]"} for i in range(2)]})
    elif case == "taskid":
        for i, (h, w) in enumerate(SIZES[:3]):
            _png(os.path.join(root, "FAST_Image", f"f{i}.png"), 60 + i, h, w)
            _png(os.path.join(root, "XDOTA_Image", f"d{i}.png"), 70 + i, h, w)
        J("FAST.json", {"data": [{"name": f"f{i}.png", "conv": [{"Question": f"what is object {i}?", "Answer": f"thing {i}"},
                                                                   {"Question": "and next? <image>", "Answer": "kept <image> as is"}]} for i in range(3)]})
        J("XDOTA.json", [{"name": f"d{i}.png", "conv": {"Question": f"<image>\ncount the ships {i}", "Answer": str(i)}} for i in range(3)])
        J("geosignal_text.json", [{"instruction": "Explain NDVI. ", "input": "briefly", "output": "a vegetation index"},
                                  {"instruction": "What is SAR?", "input": "", "output": "synthetic aperture radar"}])
    else:
        raise ValueError(case)
    return dict(root=root)


STAGE1 = [("rsicd", "plain"), ("nwpu", "plain"), ("textrs", "llava_llama_2"), ("llava", "llava_llama_2")]
STAGE2 = [("instruct", "llava_llama_2"), ("rsvg", "llava_llama_2")]
STAGE3 = [("taskid", "llava_llama_2")]
