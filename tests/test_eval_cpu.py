"""CPU: datasets, collators and scoring of the evaluation callers against tests/golden/eval.json - the REFERENCE's classes
and functions run over the same synthetic corpora and answer strings (tests/eval_cases.py, tests/golden/make_golden_eval.py).
Prompt ids, targets, batches, class indices and parse results are exact; accuracies are the same floats."""
import json
import logging
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import eval_cases as EC  # noqa: E402

from lhrs_bot_amd import conversation as conv_lib  # noqa: E402
from lhrs_bot_amd import eval_datasets as ED  # noqa: E402
from lhrs_bot_amd import evaluation as EV  # noqa: E402
from lhrs_bot_amd.data import DataCollatorForVGSupervisedDataset, tokenizer_image_token  # noqa: E402

Z = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "eval.json")))
tok = EC.ToyTok()


@pytest.fixture(autouse=True)
def _restore_default_conversation():
    keep = conv_lib.default_conversation
    yield
    conv_lib.default_conversation = keep


def test_classification_datasets(tmp_path):
    kw = EC.build_case(str(tmp_path / "ucm"), "ucm")
    want = Z["datasets"]["ucm"]
    ds = ED.UCM(kw["root"], split="all", transform=None, return_idx=False)
    assert len(ds) == want["n"] and list(ED.UCM.CLASS_NAME) == want["classes"]
    assert [[ds.imgs[i], ds[i][1], list(ds[i][0].size)] for i in range(len(ds))] == want["rows"]
    with open(os.path.join(kw["root"], "test.txt"), "w") as f:
        f.writelines(f"{os.path.join(kw['root'], 'img', n)} {c}\n" for n, c in EC.UCM_FILES[:3])
    ma = ED.MillionAidEval(kw["root"], split="test", transform=None, return_idx=True)
    assert [[os.path.basename(ma.imgs[i]), ma[i][1], ma[i][2], list(ma[i][0].size)] for i in range(len(ma))] == Z["datasets"]["millionaid"]["rows"]
    with pytest.raises(AssertionError):
        ED.UCM(kw["root"], split="val")

    kw = EC.build_case(str(tmp_path / "meterml"), "meterml")
    want = Z["datasets"]["meterml"]
    ds = ED.METERMLDataset(root=kw["root"], split="test", mode="naip_rgb", transform=None)
    assert len(ds) == want["n"] and list(ED.METERMLDataset.CLASS_NAME) == want["classes"]
    assert [[str(ds.image_folder[i]), int(ds[i][1]), list(ds[i][0].size), ds[i][0].mode] for i in range(len(ds))] == want["rows"]


def test_image_folder_scan(tmp_path):
    """torchvision's ImageFolder contract (not importable here: restated): sorted class directories, sorted files, extension filter."""
    kw = EC.build_case(str(tmp_path / "aid"), "aid")
    ds = ED.ImageFolderInstance(dataset_name="AID", return_index=False, root=kw["root"], transform=None)
    assert ds.classes == sorted(EC.AID_CLASSES) and ds.class_to_idx["Pond"] == 3 and len(ds.CLASS_NAME) == 30
    assert [os.path.basename(p) for p, _ in ds.samples] == ["airport_0.png", "airport_1.png", "bareland_0.png", "church_0.png", "church_1.png", "pond_0.png",
                                                           "viaduct_0.png", "viaduct_1.png"]
    assert ds.targets == [0, 0, 1, 2, 2, 3, 4, 4] and ds[5][1] == 3 and ds[5][0].mode == "RGB"
    assert len(ED.ImageFolderInstance(dataset_name="AID", root=kw["root"])[0]) == 3  # return_index
    os.makedirs(tmp_path / "aid" / "Empty")
    with pytest.raises(FileNotFoundError):
        ED.ImageFolderInstance(dataset_name="AID", root=kw["root"])
    with pytest.raises(AssertionError):
        ED.ImageFolderInstance(dataset_name="NoSuchSet", root=kw["root"])


@pytest.mark.parametrize("case,cls", [("rsvqa_lr", "RSVQALR"), ("rsvqa_hr", "RSVQAHR")])
@pytest.mark.parametrize("tune", [False, True])
def test_rsvqa_prompts_and_batch(case, cls, tune, tmp_path):
    want = Z["datasets"][case + ("_im_start" if tune else "")]
    kw = EC.build_case(str(tmp_path / case), case)
    ds = getattr(ED, cls)(root=kw["root"], image_root=kw["image_root"], image_transform=lambda x: x, split="test", token_prefix="<image>[VQA] ",
                          prompt_type="llava_llama_2", tokenizer=tok, tune_im_start=tune)
    assert len(ds) == want["n"]
    for i, row in enumerate(want["rows"]):
        s = ds[i]
        assert s["question"].tolist() == row["ids"], i
        assert (s["answer"], s["type"], s["questions_idx"], ds.ids[i], list(s["x"].shape)) == (row["answer"], row["type"], row["questions_idx"], row["image_id"], row["x_shape"])
    b = ED.DataCollatorForVQASupervisedDataset(tok)([dict(ds[i], x=torch.zeros(2, 2)) for i in range(len(ds))])
    assert b["questions"].tolist() == want["batch"]["questions"] and b["attn_mask"].tolist() == want["batch"]["attn_mask"]
    assert (b["targets"], b["types"], b["questions_idx"]) == (want["batch"]["targets"], want["batch"]["types"], want["batch"]["questions_idx"])
    assert b["images"].shape == (len(ds), 2, 2)
    # default transform (no rescaling, CHW) and the device-transform marker (decode only)
    chw = getattr(ED, cls)(root=kw["root"], image_root=kw["image_root"], split="test", tokenizer=tok)[0]["x"]
    assert chw.dtype == torch.uint8 and tuple(chw.shape) == (3, 256, 256)

    class Deferred(ED.CLIPImageProcessorHIP):
        def __init__(self):
            pass

    hwc = getattr(ED, cls)(root=kw["root"], image_root=kw["image_root"], split="test", tokenizer=tok, image_transform=Deferred())[0]["x"]
    assert hwc.dtype == torch.uint8 and tuple(hwc.shape) == (256, 256, 3) and torch.equal(hwc.permute(2, 0, 1), chw)


@pytest.mark.parametrize("case", ["vg_rsvg", "vg_dior", "vg_other"])
def test_vg_eval_dataset(case, tmp_path):
    want = Z["datasets"][case]
    kw = EC.build_case(str(tmp_path / case), case)
    ds = ED.VGEvalDataset(root=kw["root"], target=kw["target"], transform=None, tokenizer=tok)
    assert len(ds) == want["n"]
    for i, row in enumerate(want["rows"]):
        img, ids, target, name = ds[i]
        assert (ids.tolist(), target, name, list(img.size)) == (row["ids"], row["target"], row["file"], row["size"])
    b = DataCollatorForVGSupervisedDataset(tok)([(torch.zeros(2, 2),) + tuple(ds[i][1:]) for i in range(len(ds))])
    assert b[1].tolist() == want["batch"]["input_ids"] and b[4].tolist() == want["batch"]["attention_mask"]
    assert (b[2], b[3]) == (want["batch"]["targets"], want["batch"]["filename"])


def test_cap_eval_dataset(tmp_path):
    import dataset_cases as DC
    from lhrs_bot_amd.datasets import pre_caption
    DC.build_case(str(tmp_path / "c"), "rsicd")
    ds = ED.CapEvalDataset(root=str(tmp_path / "c" / "RSICD_Image"), target=str(tmp_path / "c" / "RSICD.json"), transform=None)
    assert len(ds) == 6
    s = ds[2]
    assert s["filename"] == "im2.png" and s["text"] == pre_caption(DC.CAPS[2]) and tuple(s["raw_image"].shape) == (3, 300, 260) and s["raw_image"].dtype == torch.uint8
    DC.build_case(str(tmp_path / "n"), "nwpu")
    ds = ED.CapEvalDataset(root=str(tmp_path / "n" / "NWPU_Image"), target=str(tmp_path / "n" / "NWPU.json"))
    assert sorted(p.name for p in ds.img_list) == ["n0.png", "n1.png", "n2.png", "n3.png"]


def test_class_prompt_and_name_matching():
    for label, classes, tune in (("ucm", ED.UCM.CLASS_NAME, False), ("meterml", ED.METERMLDataset.CLASS_NAME, True), ("folder", ["Dense_Residential", "Storage_Tanks", "Pond"], False)):
        want = Z["cls"]["prompts"][label]
        names, turn = EV.class_prompt(classes, tune_im_start=tune)
        conv = conv_lib.default_conversation.copy()
        conv.append_message(conv.roles[0], turn)
        conv.append_message(conv.roles[1], None)
        assert names == want["all_classes"] and conv.get_prompt() == want["prompt"]
        ids = tokenizer_image_token(conv.get_prompt(), tok, -200, return_tensors="pt").unsqueeze(0).repeat(3, 1)
        assert ids.tolist() == want["input_ids"]
    names, _ = EV.class_prompt(ED.UCM.CLASS_NAME)
    idx = EV.classname_2_idx(Z["cls"]["preds"], {c: i for i, c in enumerate(names)})
    assert idx == Z["cls"]["idx"]
    assert EV.balanced_accuracy(Z["cls"]["trues"], idx) == pytest.approx(Z["cls"]["balanced_accuracy"], abs=1e-12)
    from sklearn.metrics import balanced_accuracy_score
    assert EV.balanced_accuracy([0, 0, 1, 2, 2, 2], [0, 1, 1, 2, 0, 2]) == pytest.approx(balanced_accuracy_score([0, 0, 1, 2, 2, 2], [0, 1, 1, 2, 0, 2]), abs=1e-12)


def test_vqa_answer_processor_and_accuracy(caplog):
    proc = EV.EvalAIAnswerProcessor()
    got = [proc(w) for w in Z["vqa"]["words"]]
    assert got == Z["vqa"]["processed"], [(w, a, b) for w, a, b in zip(Z["vqa"]["words"], got, Z["vqa"]["processed"]) if a != b]
    lg = logging.getLogger("train")
    keep = lg.propagate
    lg.propagate = True
    try:
        with caplog.at_level(logging.INFO, logger="train"):
            total, per_type = EV.TextVQAAccuracyEvaluator().eval_pred_list(Z["vqa"]["preds"], return_types=True)
    finally:
        lg.propagate = keep
    assert total == Z["vqa"]["total"]
    assert [r.getMessage() for r in caplog.records if r.name == "train"] == Z["vqa"]["type_lines"]
    assert set(per_type) == {"presence", "comp", "rural_urban", "count", "x"}


def test_grounding_iou_and_parse():
    for (a, b), want in zip(Z["vg"]["boxes"], Z["vg"]["iou"]):
        assert EV.calculate_iou(a, b) == want
    r = EV.score_grounding(Z["vg"]["preds"])
    assert [f"Accuracy: {r['accuracy']}", f"Fail Sample: {r['fail']}", f"Accuracy With Fail Sample: {r['accuracy_with_fail']}"] == Z["vg"]["lines"]


def test_bench_choice_scoring():
    assert [EV.normalize_answer(s) for s in Z["bench"]["normalize_in"]] == Z["bench"]["normalize_out"]
    for decoded, answer, want in Z["bench"]["restated"]:
        assert EV.score_choice(decoded, answer) == want, (decoded, answer)
    q = EV.bench_question("What is it?", "A. x B. y")
    assert q == "<image>\nWhat is it?\nChoices: A. x B. y Answer from the given choices with A., B., C., D., etc."


def test_save_result_merges_and_deduplicates(tmp_path):
    rows = [dict(question_id=1, pred="a"), dict(question_id=2, pred="b"), dict(question_id=1, pred="again")]
    path = EV.save_result(rows, str(tmp_path), "eval_save_file", "question_id")
    assert json.load(open(path)) == rows[:2] and os.path.exists(tmp_path / "eval_save_file_rank0.json")
