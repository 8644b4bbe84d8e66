"""The evaluation callers of `generate` on the HIP engine: main_cls.py, main_vqa.py, main_vg.py, main_bench_gen.py end to
end over the synthetic corpora of tests/eval_cases.py with a 2-layer random model - what reaches `generate` (device-transformed
pictures, left-padded prompts, masks; int-exact against tests/golden/eval.json), what comes back (decoded, merged, scored) - and the
classification transform kernel bit-exact against its Pillow-backed oracle."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import eval_cases as EC  # noqa: E402
import lhrs_bot_amd.unibind as U  # noqa: E402
from lhrs_bot_amd import conversation as conv_lib  # noqa: E402
from lhrs_bot_amd import evaluation as EV  # noqa: E402
from lhrs_bot_amd.data import ClsEvalTransformHIP  # noqa: E402
from oracle import image_oracle as IO  # noqa: E402

Z = json.load(open(os.path.join(HERE, "golden", "eval.json")))


class Tok(EC.ToyTok):
    """the fixture's whitespace tokenizer + a decoder: id -> "t<id>" (special ids dropped on request)"""

    def decode(self, ids, skip_special_tokens=False):
        ids = [int(i) for i in (ids.tolist() if hasattr(ids, "tolist") else ids)]
        return " ".join(self.words.get(i, f"t{i}") for i in ids if not (skip_special_tokens and i in (0, 1, 2)))

    def batch_decode(self, ids, skip_special_tokens=True):
        return [self.decode(r, skip_special_tokens) for r in ids]

    words = {}


@pytest.fixture(autouse=True)
def _restore_default_conversation():
    keep = conv_lib.default_conversation
    yield
    conv_lib.default_conversation = keep


@pytest.fixture
def spy(monkeypatch):
    """records every UniBind.generate call; `spy.answers` (a list of id rows per call) replaces the model's output when set"""
    import transformers
    tok = Tok()
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: tok))
    orig = U.UniBind.generate
    rec = type("Spy", (), {"calls": [], "answers": None, "tok": tok})()

    def wrapped(self, input_ids=None, **kw):
        out = orig(self, input_ids=input_ids, **kw)
        images = kw.get("images")
        rec.calls.append(dict(input_ids=input_ids.clone(), images=images, kw={k: v for k, v in kw.items() if k != "images"}, out=out.clone()))
        if rec.answers is not None:
            rows = rec.answers[len(rec.calls) - 1]
            out = torch.tensor(rows, device=out.device)
        return out

    monkeypatch.setattr(U.UniBind, "generate", wrapped)
    return rec


def test_cls_eval_transform_bit_exact_vs_pillow_oracle():
    rng = np.random.default_rng(11)
    sizes = [(341, 500), (256, 256), (300, 260), (259, 777), (1000, 343), (257, 258), (224, 224), (240, 320), (2, 900), (1500, 2000)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    t = ClsEvalTransformHIP()
    pv = t.preprocess(imgs)["pixel_values"]
    assert pv.shape == (len(sizes), 3, 224, 224) and pv.dtype == torch.float32 and pv.is_cuda
    for b, im in enumerate(imgs):
        assert np.array_equal(pv[b].cpu().numpy(), IO.cls_eval_preprocess(im)), sizes[b]
    same = torch.from_numpy(np.stack([imgs[1], imgs[1]]))  # a stacked uint8 batch [B, H, W, 3] is accepted as is
    assert torch.equal(t.preprocess(same)["pixel_values"][1], pv[1])
    # and the Pillow call itself, once, so that the oracle's resize is not only checked against its own restatement here
    from PIL import Image
    h, w = sizes[0]
    r = Image.fromarray(imgs[0]).resize((int(256 * w / h), 256), Image.BICUBIC)
    top, left = int(round((r.height - 224) / 2.0)), int(round((r.width - 224) / 2.0))
    c = torch.from_numpy(np.array(r.crop((left, top, left + 224, top + 224)))).permute(2, 0, 1).float().div(255)
    mean, std = (torch.tensor(v)[:, None, None] for v in (ClsEvalTransformHIP.image_mean, ClsEvalTransformHIP.image_std))
    assert torch.equal(pv[0].cpu(), (c - mean) / std)
    with pytest.raises(ValueError):
        ClsEvalTransformHIP(input_size=(192, 192))


def test_main_cls_end_to_end(tmp_path, spy):
    import main_cls
    kw = EC.build_case(str(tmp_path / "ucm"), "ucm")
    cfg = main_cls.parse_option(["--batch-size", "4", "--data-path", kw["root"], "--llama-layers", "2", "--workers", "0", "--output", str(tmp_path / "o"),
                                 "--tokenizer-path", "toy", "--opts", "eval.dataset", "UCM"])
    os.makedirs(cfg.output, exist_ok=True)
    want = Z["cls"]["prompts"]["ucm"]
    # the model "answers" the class names of the first four pictures, then something unrelated and nothing
    names = want["all_classes"]
    spy.tok.words = {**{40000 + i: n for i, n in enumerate(names)}, 41000: "zzz"}
    spy.answers = [[[40000 + 0, 2], [40000 + 1, 2], [40000 + 3, 2], [40000 + 10, 2]], [[40000 + 5, 2], [41000, 2]]]
    res = main_cls.main(cfg)
    assert len(spy.calls) == 2 and [c["input_ids"].shape[0] for c in spy.calls] == [4, 2]
    ids = torch.tensor(want["input_ids"])[0]
    for c in spy.calls:
        assert all(torch.equal(row, ids) for row in c["input_ids"])  # ONE prompt, repeated (int-exact with the reference's)
        assert c["kw"]["do_sample"] is False and c["kw"]["max_new_tokens"] == 20 and c["kw"]["num_beams"] == 1 and c["kw"]["weights"] == "bf16"
        assert c["images"].dtype == torch.float32 and c["images"].is_cuda and tuple(c["images"].shape[1:]) == (3, 224, 224)
        assert c["out"].shape[0] == c["input_ids"].shape[0] and c["out"].shape[1] <= 20  # the real generate ran on the HIP engine
    from PIL import Image
    first = np.array(Image.open(os.path.join(kw["root"], "img", EC.UCM_FILES[0][0])).convert("RGB"))
    assert np.array_equal(spy.calls[0]["images"][0].cpu().numpy(), IO.cls_eval_preprocess(first))
    assert res["trues"] == [c for _, c in EC.UCM_FILES] and res["classes"] == names
    assert res["preds"][:4] == [names[0], names[1], names[3], names[10]] and res["pred_idx"] == [0, 1, 3, 10, 5, 20]
    assert res["mean_per_class_recall"] == pytest.approx(4 / 6)


def test_main_vqa_end_to_end(tmp_path, spy):
    import main_vqa
    kw = EC.build_case(str(tmp_path / "lr"), "rsvqa_lr")
    out = tmp_path / "o"
    os.makedirs(out)
    cfg = main_vqa.parse_option(["--batch-size", "4", "--data-path", kw["image_root"], "--data-target", kw["root"], "--data-type", "LR", "--llama-layers", "2",
                                 "--workers", "0", "--output", str(out), "--tokenizer-path", "toy", "--opts", "prompt_template", "llava_llama_2", "bits", "8"])
    want = Z["datasets"]["rsvqa_lr"]
    spy.tok.words = {40000: "yes", 40001: "no", 40002: "rural", 40003: "It", 40004: "is", 40005: "urban."}
    # targets of the six kept questions: yes, no, rural, no, yes, yes
    spy.answers = [[[40000, 2, 2], [40000, 2, 2], [40003, 40004, 40002], [40001, 2, 2]], [[40001, 2, 2], [40000, 2, 2]]]
    res = main_vqa.main(cfg)
    assert len(spy.calls) == 2
    got_ids = [r for c in spy.calls for r in c["input_ids"].tolist()]
    n = len(want["batch"]["questions"][0])
    # each batch is left-padded to ITS longest prompt; stripping the pads gives the reference's per-sample ids
    assert [[t for t in r if t != 0] for r in got_ids] == [row["ids"] for row in want["rows"]]
    assert spy.calls[1]["input_ids"].shape[1] == n
    for c in spy.calls:
        assert torch.equal(c["kw"]["attention_mask"], c["input_ids"].ne(0)) and c["kw"]["max_new_tokens"] == 50 and c["kw"]["weights"] == "fp8"
        assert c["images"].dtype == torch.uint8 and tuple(c["images"].shape[1:]) == (256, 256, 3)  # decoded only: the device runs the CLIP transform
        assert c["out"].shape[0] == c["input_ids"].shape[0]
    merged = json.load(open(out / "eval_save_file.json"))
    assert [m["question_id"] for m in merged] == want["batch"]["questions_idx"] and [m["target"] for m in merged] == want["batch"]["targets"]
    assert [m["pred"] for m in merged] == ["yes", "yes", "It is rural", "no", "no", "yes"]
    # yes/yes ok, yes/no wrong, "it is rural" -> not equal to "rural" and not a substring OF "rural" -> wrong, no/no ok, no/yes wrong, yes/yes ok
    assert res["total"] == pytest.approx(100.0 * 3 / 6)
    assert res["per_type"] == {"presence": 100.0, "comp": 0.0, "rural_urban": 0.0}


def test_main_vg_end_to_end(tmp_path, spy):
    import main_vg
    kw = EC.build_case(str(tmp_path / "vg"), "vg_rsvg")
    out = tmp_path / "o"
    os.makedirs(out)
    cfg = main_vg.parse_option(["--batch-size", "2", "--data-path", kw["root"], "--data-target", kw["target"], "--llama-layers", "2", "--workers", "0",
                                "--output", str(out), "--tokenizer-path", "toy"])
    want = Z["datasets"]["vg_rsvg"]
    spy.tok.words = {40000: "[0, 20, 60, 90]", 40001: "[10, 20, 61, 80]", 40002: "nothing"}
    spy.answers = [[[40000, 2], [40001, 2]], [[40002, 2]]]
    res = main_vg.main(cfg)
    assert [c["input_ids"].tolist() for c in spy.calls] == [want["batch"]["input_ids"][:2], want["batch"]["input_ids"][2:]]
    for c in spy.calls:
        assert c["kw"]["max_new_tokens"] == 100 and torch.equal(c["kw"]["attention_mask"], c["input_ids"].ne(0))
        assert isinstance(c["images"], (list, torch.Tensor)) and all(x.dtype == torch.uint8 for x in c["images"])  # pictures of different sizes stay a list
    merged = json.load(open(out / "eval_save_file.json"))
    assert [m["filename"] for m in merged] == want["batch"]["filename"] and [m["target"] for m in merged] == want["batch"]["targets"]
    # targets [0,20,60,90] / [10,20,61,90] / [20,20,62,90]: exact hit, IoU 61*52... > 0.5 hit, unparsable
    assert res["total"] == 2 and res["fail"] == 1 and res["accuracy"] == 100.0 and res["accuracy_with_fail"] == pytest.approx(200 / 3)


def test_main_bench_gen_end_to_end(tmp_path, spy):
    import main_bench_gen
    kw = EC.build_case(str(tmp_path / "b"), "bench")
    cfg = main_bench_gen.parse_option(["--data-path", kw["root"], "--data-target", kw["target"], "--llama-layers", "2", "--output", str(tmp_path / "o"), "--tokenizer-path", "toy"])
    spy.tok.words = {40000: "B.", 40001: "C", 40002: "D"}
    spy.answers = [[[40000, 2]], [[40001, 2]], [[40002, 2]]]
    res = main_bench_gen.main(cfg)
    assert len(spy.calls) == 3 and all(c["input_ids"].shape[0] == 1 and c["kw"]["max_new_tokens"] == 10 for c in spy.calls)
    assert all(c["images"].dtype == torch.float32 and tuple(c["images"].shape) == (1, 3, 224, 224) for c in spy.calls)
    assert torch.equal(spy.calls[0]["images"], spy.calls[1]["images"]) and not torch.equal(spy.calls[0]["images"], spy.calls[2]["images"])
    conv = conv_lib.default_conversation.copy()
    conv.append_message(conv.roles[0], EV.bench_question("What is the object?", "A. ship B. plane C. car"))
    conv.append_message(conv.roles[1], None)
    from lhrs_bot_amd.data import tokenizer_image_token
    assert spy.calls[0]["input_ids"][0].tolist() == tokenizer_image_token(conv.get_prompt(), spy.tok, -200)
    # answers B / A / d: "B." right, "C" wrong, "D" right -> identity 100, color 0, count 50 (wrong + right), total 66.67
    assert res == {"total": 66.67, "identity": 100.0, "color": 0.0, "count": 50.0}
