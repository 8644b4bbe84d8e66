"""Synthetic corpora of the evaluation callers, shared by tests/golden/make_golden_eval.py (which runs the
REFERENCE's dataset classes / scoring functions on them) and the parity tests.  Layouts follow what the reference's classes read:

  ucm       <root>/img/*.png + <root>/all.txt ("name idx" lines)                      lhrs/Dataset/UCM.py
  aid       <root>/<ClassName>/*.png (ImageFolder)                                    lhrs/Dataset/ImageFolderInstance.py
  meterml   <root>/test.geojson + <root>/test_images/<folder>/naip.png                lhrs/Dataset/meterml.py
  rsvqa_lr  <root>/LR_split_test_{questions,answers,images}.json + <root>/Images_LR/<id>.tif    lhrs/Dataset/rsvqa.py
  rsvqa_hr  same with the USGS prefix
  vg_rsvg / vg_dior / vg_other   <img dir> + <X_RSVG_test | X_DIOR_test | other>.json           lhrs/Dataset/cap_dataset.py:186-260
  bench     <img dir> + qa json {"data": [{filename, qa_pairs}], "qtype": {...}}      main_bench_gen.py:190-200
"""
import json
import os

import numpy as np

from dataset_cases import SIZES, ToyTok, _png  # noqa: F401


def _tif(path, seed, h, w):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path)


UCM_FILES = [("agricultural00.png", 0), ("airplane07.png", 1), ("beach03.png", 3), ("harbor11.png", 10), ("tenniscourt99.png", 20), ("river05.png", 16)]
AID_CLASSES = ["Airport", "BareLand", "Church", "Pond", "Viaduct"]
METERML_ROWS = [("test_images/00aa", 0), ("test_images/01bb", 3), ("test_images/02cc", 6), ("test_images/03dd", 1)]
VQA_QUESTIONS = [  # (image id, type, question, answer)
    (0, "presence", "Is there a road?", "yes"), (0, "count", "How many buildings are there?", "12"), (0, "comp", "Are there more roads than rivers?", "no"),
    (1, "rural_urban", "Is it a rural or an urban area", "rural"), (1, "area", "What is the area covered by water?", "0m2"),
    (2, "presence", "Is a water area present in the image?", "no"), (3, "comp", "Is the number of roads equal to the number of water areas?", "yes"),
    (3, "presence", "Is there a farmland at the top of a road in the image, and is it longer than that?", "yes")]


def build_case(root, case):
    """-> kwargs of the reference's dataset class for this case (paths only)."""
    os.makedirs(root, exist_ok=True)
    J = lambda name, obj: json.dump(obj, open(os.path.join(root, name), "w"))  # noqa: E731
    if case == "ucm":
        for i, (name, _) in enumerate(UCM_FILES):
            _png(os.path.join(root, "img", name), 100 + i, *SIZES[i % len(SIZES)])
        with open(os.path.join(root, "all.txt"), "w") as f:
            f.writelines(f"{n} {c}\n" for n, c in UCM_FILES)
        return dict(root=root)
    if case == "aid":
        k = 0
        for ci, c in enumerate(AID_CLASSES):
            for j in range(2 if ci % 2 == 0 else 1):
                _png(os.path.join(root, c, f"{c.lower()}_{j}.png"), 200 + k, *SIZES[k % len(SIZES)])
                k += 1
        open(os.path.join(root, "Airport", "notes.txt"), "w").write("not an image\n")  # extension filter
        return dict(root=root)
    if case == "meterml":
        feats = []
        for i, (folder, idx) in enumerate(METERML_ROWS):
            _png(os.path.join(root, folder, "naip.png"), 300 + i, *SIZES[i % len(SIZES)])
            feats.append({"type": "Feature", "properties": {"Image_Folder": folder, "idx": idx, "Type": "x"}, "geometry": {"type": "Point", "coordinates": [0.0, float(i)]}})
        J("test.geojson", {"type": "FeatureCollection", "features": feats})
        return dict(root=root)
    if case in ("rsvqa_lr", "rsvqa_hr"):
        prefix = "LR" if case == "rsvqa_lr" else "USGS"
        img_dir = "Images_LR" if case == "rsvqa_lr" else "Data"
        for i in range(5):
            _tif(os.path.join(root, img_dir, f"{i}.tif"), 400 + i, 256, 256)
        questions, answers = [], []
        per_image = {i: [] for i in range(5)}
        for qi, (img, typ, q, a) in enumerate(VQA_QUESTIONS):
            questions.append({"id": qi, "active": True, "img_id": img, "type": typ, "question": q, "answers_ids": [qi]})
            answers.append({"id": qi, "active": True, "question_id": qi, "answer": a})
            per_image[img].append(qi)
        images = [{"id": i, "active": i != 4, "questions_ids": per_image[i]} for i in range(5)]
        J(f"{prefix}_split_test_questions.json", {"questions": questions})
        J(f"{prefix}_split_test_answers.json", {"answers": answers})
        J(f"{prefix}_split_test_images.json", {"images": images})
        return dict(root=root, image_root=img_dir)
    if case in ("vg_rsvg", "vg_dior", "vg_other"):
        img = os.path.join(root, "imgs")
        if case == "vg_rsvg":
            for i in range(3):
                _png(os.path.join(img, f"r{i}.jpg"), 500 + i, *SIZES[i])
            data = [{"img": f"r{i}.jpg", "question": f"[VG] where is the storage tank number {i} ?", "answer": f"[{10 * i}, 20, {60 + i}, 90]"} for i in range(3)]
            data.append({"img": "absent.jpg", "question": "[VG] nothing", "answer": "[0, 0, 1, 1]"})
            name = "Synth_RSVG_test.json"
        elif case == "vg_dior":
            for i in range(2):
                _png(os.path.join(img, f"d{i}.jpg"), 510 + i, *SIZES[i + 2])
            data = [{"img": f"d{i}", "question": f"[VG] the bridge on the left {i}", "answer": f"[{i}, {i}, 50, 50]"} for i in range(2)]
            name = "Synth_DIOR_test.json"
        else:
            for i in range(2):
                _png(os.path.join(img, f"o{i}.png"), 520 + i, *SIZES[i + 1])
            data = [{"name": "o0.png", "conv": [{"Question": "[VG] first round", "Answer": "[1, 2, 3, 4]"}, {"Question": "and the second?", "Answer": None}], "answer": "[5, 6, 7, 8]"},
                    {"name": "o1.png", "conv": {"Question": "[VG] a single round given as a dict", "Answer": None}, "answer": "[9, 9, 99, 99]"}]
            name = "Synth_other.json"
        J(name, {"data": data})
        return dict(root=img, target=os.path.join(root, name))
    if case == "bench":
        img = os.path.join(root, "imgs")
        for i in range(2):
            _png(os.path.join(img, f"b{i}.png"), 600 + i, *SIZES[i])
        J("bench_qa.json", {"qtype": {"1 identity": 0, "2 color": 0, "3 count": 0},
                            "data": [{"filename": "b0.png", "qa_pairs": [{"question": "What is the object?", "choices": "A. ship B. plane C. car", "answer": "B", "type": ["1"]},
                                                                           {"question": "Which colour?", "choices": "A. red B. blue", "answer": "A", "type": ["2", "3"]}]},
                                     {"filename": "b1.png", "qa_pairs": [{"question": "How many?", "choices": "A. one B. two C. three D. four", "answer": "d", "type": ["3"]}]}]})
        return dict(root=img, target=os.path.join(root, "bench_qa.json"))
    raise ValueError(case)


# strings for the scoring functions (what a model might answer; the reference's parsers must be matched on all of them)
CLS_PREDS = ["airplane", " beach ", "a harbor with boats", "tennis", "Dense Residential area", "", "zzz", "storage tanks", "mobilehomepark.", "The image shows a river"]
VQA_PREDS = [("yes", "yes", "presence"), ("Yes.", "yes", "presence"), ("no", "yes", "comp"), ("It is a rural area", "rural", "rural_urban"), ("rural", "rural", "rural_urban"),
             ("two", "2", "count"), ("There are 3", "3", "count"), ("the urban", "urban", "rural_urban"), ("dont know", "don't know", "comp"), ("1,000", "1000", "count"),
             ("a.b", "a.b", "x"), ("what's", "whats", "x"), ("NONE", "0", "count"), ("yes?", ["yes", "yes", "no", "yes"], "presence"), ("no", ["yes", "yes", "no", "yes"], "presence")]
VG_PREDS = [("[10, 20, 60, 90]", "[10, 20, 61, 90]"), ("the object is at [0,0,10,10]", "[50, 50, 80, 80]"), ("no box here", "[1, 2, 3, 4]"), ("[1, 2, 3]", "[1, 2, 3, 4]"),
            ("[1, 2, 3, 4, 5, 6]", "[1, 2, 3, 4]"), ("[1.5, 2.5, 30.25, 40]", "[2, 3, 30, 40]"), ("[a, b]", "[1, 2, 3, 4]"), ("[5, 5, 20, 20] and [30, 30, 60, 60]", "[5, 5, 20, 21] [0, 0, 10, 10]"),
            ("[1, , 3, 4]", "[1, 2, 3, 4]"), ("[10, 10, 20, 20]", "[10, 10, 20, 20]")]
BENCH_OUT = [("B. plane", "B"), ("a", "A"), (" C", "c"), ("The answer is D", "D"), ("d.", "d"), (".", "A"), ("An", "A")]
