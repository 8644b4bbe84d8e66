"""Scoring and driver plumbing of the evaluation callers of `generate`.

What /root/reference main_cls.py, main_vqa.py, main_vg.py and main_bench_gen.py do AROUND `model.generate`: turn the decoded strings into
class indices / normalised answers / boxes, and reduce them to the numbers of the paper's tables.  The generation itself (batched,
left-padded prompts, greedy) is the engine's (`UniBind.generate` -> `TextModal.generate`, lhrs_bot_amd/text.py); this module is host-side
string and integer work, pinned to the reference's functions on a fixed list of answer strings (tests/golden/eval.json,
tests/test_eval_cpu.py).

  classname_2_idx          main_cls.py:35-62     exact class name, else the class sharing the longest common substring with the answer
  EvalAIAnswerProcessor    main_vqa.py:228-470   the VQA-challenge answer normalisation (punctuation, digits, articles, contractions)
  TextVQAAccuracyEvaluator main_vqa.py:473-518   soft accuracy, total and per question type
  calculate_iou, score_grounding   main_vg.py:31-52, 262-313   "[x1, y1, x2, y2]" parsing, Acc@0.5 with and without the failed parses
  normalize_answer, score_choice   main_bench_gen.py:42-58, 256-266   multiple-choice letter comparison
  save_result              main_vqa.py:30-62     per-rank json -> merged json with duplicates removed
"""
from __future__ import annotations

import json
import logging
import os
import re
import string
from collections import defaultdict
from difflib import SequenceMatcher
from typing import Dict, List, Optional, Sequence, Tuple

import torch

logger = logging.getLogger("train")


# ------------------------------------------------------------------------------------------------ scene classification
CLS_TEMPLATE = [lambda c: f"[CLS] Choose the best categories describe the image from: {c}"]  # main_cls.py:32 (c: the class LIST, printed as a list)


def find_index_of_max_similar_substring(given_string: str, string_list: Sequence[str]) -> int:
    """Index of the first string sharing the longest common substring with `given_string`; -1 when nothing is shared at all (which the
    caller then uses as a Python index: the LAST class)."""
    sizes = [SequenceMatcher(None, given_string, s).find_longest_match(0, len(given_string), 0, len(s)).size for s in string_list]
    best = max(sizes, default=0)
    return sizes.index(best) if best > 0 else -1


def classname_2_idx(preds: Sequence[str], classes_to_idx: Dict[str, int]) -> List[int]:
    classes = list(classes_to_idx)
    return [classes_to_idx[p] if p in classes_to_idx else classes_to_idx[classes[find_index_of_max_similar_substring(p, classes)]]
            for p in (q.strip() for q in preds)]


def class_prompt(all_classes: Sequence[str], tune_im_start: bool = False) -> Tuple[List[str], str]:
    """main_cls.py:150-164 -> (class names as the model sees them, the user turn carrying the picture and the class list)."""
    from .data import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN
    names = [c.lower().replace("_", " ") for c in all_classes]
    image = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN if tune_im_start else DEFAULT_IMAGE_TOKEN
    return names, image + "\n" + CLS_TEMPLATE[0](names)


# ------------------------------------------------------------------------------------------------ VQA answer normalisation
def _contractions() -> Dict[str, str]:
    """The apostrophe-less spellings the VQA-challenge normaliser repairs, generated from their grammar instead of listed: a stem plus one
    of n't / 've / 'd / 'll / 's / 're, and the doubled forms n't've / 'd've with the first or the second apostrophe missing.  Words are
    lower-cased and "'s" is split off before the lookup, so spellings with a capital or containing "'s" can never match and are not
    generated.  Pinned entry by entry to the reference's table (tests/test_eval_cpu.py)."""
    t: Dict[str, str] = {}
    nt = "ai are ca could did does do had has have is might must need ought sha should was were wo would".split()
    t.update({s + "nt": s + "n't" for s in nt})
    for s in "could had might should would".split():
        t[s + "nt've"] = t[s + "n'tve"] = s + "n't've"
    t.update({s + "ve": s + "'ve" for s in "could might must should would not they we what where who you".split()})
    t.update({s + "d": s + "'d" for s in "he how it someone something there they where who you".split()})
    for s in "he it she somebody someone something there they we who you".split():
        t[s + "d've"] = t[s + "'dve"] = s + "'d've"
    t.update({s + "ll": s + "'ll" for s in "how it somebody someone something they what who why you".split()})
    t.update({s + "s": s + "'s" for s in "he how somebody someone that there what when where who why".split()})
    t.update({s + "re": s + "'re" for s in "there they what why you".split()})
    t.update({"maam": "ma'am", "oclock": "o'clock", "twas": "'twas", "'ows'at": "'ow's'at", "somebody'd": "somebodyd",  # the last one really is reversed upstream
              "yall": "y'all", "yall'll": "y'all'll", "y'allll": "y'all'll", "yall'd've": "y'all'd've", "y'alld've": "y'all'd've", "y'all'dve": "y'all'd've"})
    return t


class EvalAIAnswerProcessor:
    """main_vqa.py:228-470.  lower-case; drop "," and "?"; split "'s" off; whitespace -> blank; every punctuation mark is deleted when the
    text has it next to a blank (or has a digit,digit comma) and turned into a blank otherwise; full stops not followed by a digit go
    (at most 32 of them: the reference passes `re.UNICODE` where `count` belongs); number words -> digits; articles dropped;
    contractions repaired."""

    CONTRACTIONS = _contractions()
    NUMBER_MAP = dict(zip("none zero one two three four five six seven eight nine ten".split(), "0 0 1 2 3 4 5 6 7 8 9 10".split()))
    ARTICLES = ("a", "an", "the")
    PUNCTUATIONS = list(";/[]\"{}()=+\\_-><@`,?!")
    _FULL_STOP = re.compile(r"\.(?!\d)")
    _DIGIT_COMMA = re.compile(r"(?<=\d)(\,)+(?=\d)")

    def __init__(self, *args, **kwargs):
        pass

    def word_tokenize(self, word: str) -> str:
        return word.lower().replace(",", "").replace("?", "").replace("'s", " 's").strip()

    def process_punctuation(self, text: str) -> str:
        out = text
        comma_between_digits = self._DIGIT_COMMA.search(text) is not None
        for p in self.PUNCTUATIONS:
            out = out.replace(p, "" if (p + " " in text or " " + p in text or comma_between_digits) else " ")
        return self._FULL_STOP.sub("", out, int(re.UNICODE))

    def process_digit_article(self, text: str) -> str:
        words = [self.NUMBER_MAP.get(w, w) for w in text.lower().split()]
        return " ".join(self.CONTRACTIONS.get(w, w) for w in words if w not in self.ARTICLES)

    def __call__(self, item: str) -> str:
        item = self.word_tokenize(item).replace("\n", " ").replace("\t", " ").strip()
        return self.process_digit_article(self.process_punctuation(item))


class TextVQAAccuracyEvaluator:
    """main_vqa.py:473-518.  An entry {"pred", "target", "types"}: `target` a string (RSVQA: one reference answer, score 1 on an exact
    normalised match) or a list of human answers (soft score min(1, matches / 3) averaged over leave-one-out subsets).  A prediction
    scoring 0 still counts as right when its normalised form occurs INSIDE the raw target (substring for a string target, membership
    for a list) - the reference's fallback."""

    def __init__(self):
        self.answer_processor = EvalAIAnswerProcessor()

    def _compute_answer_scores(self, raw_answers) -> Dict[str, float]:
        if not isinstance(raw_answers, list):
            return {raw_answers: 1}
        answers = [self.answer_processor(a) for a in raw_answers]
        scores = {}
        for unique in set(answers):
            accs = [min(1, sum(1 for j, a in enumerate(answers) if j != i and a == unique) / 3) for i in range(len(answers))]
            scores[unique] = sum(accs) / len(accs)
        return scores

    def eval_pred_list(self, pred_list: Sequence[Dict], return_types: bool = False):
        scores, by_type = [], defaultdict(list)
        for entry in pred_list:
            pred = self.answer_processor(entry["pred"])
            score = self._compute_answer_scores(entry["target"]).get(pred, 0.0)
            if score == 0.0 and pred in entry["target"]:
                score = 1.0
            scores.append(score)
            by_type[entry["types"]].append(score)
        per_type = {t: 100.0 * (sum(s) / len(s)) for t, s in by_type.items()}
        for t, v in per_type.items():
            logger.info(f"{t}: {v}")
        accuracy = sum(scores) / len(scores)
        return (accuracy, per_type) if return_types else accuracy


# ------------------------------------------------------------------------------------------------ visual grounding
def calculate_iou(box1, box2) -> float:
    """IoU of two [x1, y1, x2, y2] boxes under the inclusive-pixel convention (+1 on every extent; main_vg.py:31-52)."""
    iw = max(0, min(box1[2], box2[2]) - max(box1[0], box2[0]) + 1)
    ih = max(0, min(box1[3], box2[3]) - max(box1[1], box2[1]) + 1)
    inter = iw * ih
    area = lambda b: (b[2] - b[0] + 1) * (b[3] - b[1] + 1)  # noqa: E731
    return inter / (area(box1) + area(box2) - inter)


_BOX = re.compile(r"\[([0-9., ]+)\]")


def score_grounding(predictions: Sequence[Dict], threshold: float = 0.5) -> Dict[str, float]:
    """main_vg.py:262-313 over entries {"pred", "target", "filename"}: every "[...]" group of digits in the answer is a box candidate,
    paired in order with the target's boxes; candidates with more than four numbers are cut to four, with fewer they count as failures,
    as do answers without any group or with an unparsable one (those skip the whole entry).  -> accuracy (% of paired boxes with
    IoU > threshold), fail (count), accuracy_with_fail (failures added to the denominator), total."""
    hits = total = fail = 0
    for item in predictions:
        groups = _BOX.findall(item["pred"])
        if not groups:
            fail += 1
        try:
            boxes = [[float(v) for v in g.split(",")] for g in groups]
        except ValueError:
            fail += 1
            continue
        targets = [[float(v) for v in g.split(",")] for g in _BOX.findall(item["target"])]
        for box, target in zip(boxes, targets):
            if len(box) < 4:
                fail += 1
                continue
            total += 1
            hits += calculate_iou(box[:4], target) > threshold
    nan = float("nan")
    return dict(accuracy=hits / total * 100 if total else nan, fail=fail, accuracy_with_fail=hits / (total + fail) * 100 if total + fail else nan, total=total)


# ------------------------------------------------------------------------------------------------ multiple choice (LHRS-Bench)
_ARTICLE_WORD = re.compile(r"\b(a|an|the)\b")
_PUNCT = str.maketrans("", "", string.punctuation)


def normalize_answer(s: str) -> str:
    """lower-case, punctuation removed, the words a / an / the blanked, whitespace squeezed (main_bench_gen.py:42-58)."""
    return " ".join(_ARTICLE_WORD.sub(" ", s.lower().translate(_PUNCT)).split())


def score_choice(decoded: str, answer: str) -> int:
    """main_bench_gen.py:256-266: only the FIRST CHARACTER of the decoded text (cut at "<|eot_id|>") is compared with the answer letter,
    both through `normalize_answer` - which turns "a" into "" (an article), so "A" also matches an empty or punctuation-only first
    character.  Kept as the reference scores."""
    head = decoded.split("<|eot_id|>")[0]
    first = head[0].strip() if head else ""
    return int(normalize_answer(first.lower()) == normalize_answer(answer.lower()))


def bench_question(question: str, choices: str, tune_im_start: bool = False) -> str:
    from .data import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN
    image = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN if tune_im_start else DEFAULT_IMAGE_TOKEN
    return image + "\n" + question + "\nChoices: " + choices + " Answer from the given choices with A., B., C., D., etc."


# ------------------------------------------------------------------------------------------------ shared driver pieces
def _dist_on() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def get_rank() -> int:
    return torch.distributed.get_rank() if _dist_on() else 0


def get_world_size() -> int:
    return torch.distributed.get_world_size() if _dist_on() else 1


def is_distributed() -> bool:
    return _dist_on()


def is_main_process() -> bool:
    return get_rank() == 0


def save_result(result: List[Dict], result_dir: str, filename: str, remove_duplicate: str = "") -> str:
    """Every rank writes `<filename>_rank<r>.json`; after a barrier rank 0 concatenates them in rank order, keeps the first entry per
    `remove_duplicate` key (a DistributedSampler pads the last batch with repeats) and writes `<filename>.json`.  -> that path."""
    final = os.path.join(result_dir, "%s.json" % filename)
    with open(os.path.join(result_dir, "%s_rank%d.json" % (filename, get_rank())), "w") as f:
        json.dump(result, f)
    if is_distributed():
        torch.distributed.barrier()
    if is_main_process():
        merged: List[Dict] = []
        for rank in range(get_world_size()):
            with open(os.path.join(result_dir, "%s_rank%d.json" % (filename, rank))) as f:
                merged += json.load(f)
        if remove_duplicate:
            seen, unique = set(), []
            for r in merged:
                if r[remove_duplicate] not in seen:
                    seen.add(r[remove_duplicate])
                    unique.append(r)
            merged = unique
        with open(final, "w") as f:
            json.dump(merged, f)
        logger.info("result file saved to %s" % final)
    return final


def eval_parse_option(args=None, data_target: bool = False, data_type: bool = False):
    """The argument parser the four evaluation scripts share (main_cls.py:65-123; main_vqa / main_vg / main_bench_gen add `--data-target`,
    main_vqa / main_bench_gen `--data-type`), plus this engine's offline knobs."""
    from .trainer import ConfigArgumentParser, ConfigDict, str2bool
    p = ConfigArgumentParser()
    p.add_argument("--opts", default=None, nargs="+", help="Modify config options by adding 'KEY VALUE' pairs.")
    p.add_argument("--batch-size", type=int, help="batch size for single GPU")
    p.add_argument("--data-path", type=str, help="path to dataset")
    if data_target:
        p.add_argument("--data-target", type=str, help="path to dataset annotation file ")
    if data_type:
        p.add_argument("--data-type", type=str, choices=["LR", "HR"], default="HR", help="VQA dataset type")
    p.add_argument("--workers", type=int, default=8, help="workers of dataloader")
    p.add_argument("--model-path", type=str, default=None, help="pretrained checkpoint path")
    p.add_argument("--enable-amp", type=str2bool, default=False, help="mixed precision")
    p.add_argument("--output", default="output", type=str, metavar="PATH", help="root of output folder")
    p.add_argument("--seed", type=int, default=322, help="random seed")
    p.add_argument("--use-checkpoint", action="store_true", help="accepted; nothing is recomputed in evaluation")
    p.add_argument("--gpus", type=int, default=0, help="gpus ID")
    p.add_argument("--inf_sampler", type=str2bool, default=False)
    p.add_argument("--wandb", type=str2bool, default=False, help="wandb logger (not available offline: must stay False)")
    p.add_argument("--entity", type=str, default="pumpkinn")
    p.add_argument("--project", type=str, default="MaskIndexNet")
    p.add_argument("--accelerator", default="gpu", type=str, choices=["cpu", "gpu", "mps"], help="accelerator (the engine is HIP: gpu)")
    p.add_argument("--local_rank", type=int, help="local rank")
    # knobs of this engine
    p.add_argument("--tokenizer-path", type=str, default=None, help="directory holding the LLaMA-2 tokenizer files")
    p.add_argument("--llama-layers", type=int, default=32, help="decoder layers to build (smoke runs)")
    config = ConfigDict(p.parse_args(wandb=True, args=args))
    opts = config.get("opts") or []
    if len(opts) % 2:
        p.error("--opts takes KEY VALUE pairs")
    import yaml
    for k, v in zip(opts[0::2], opts[1::2]):  # dotted keys descend into the YAML's sections: --opts eval.dataset UCM
        node, *rest = config, *k.split(".")
        for part in rest[:-1]:
            node = node.setdefault(part, ConfigDict())
        node[rest[-1]] = yaml.safe_load(v)
    if config.get("batch_size") is None:
        config.batch_size = 1
    return config


def eval_model(config):
    """build_model -> dtype -> `--model-path` -> device check -> eval(), as every evaluation script opens (main_cls.py:126-148)."""
    from .unibind import build_model
    if config.get("accelerator", "gpu") != "gpu" or not torch.cuda.is_available():
        raise RuntimeError("the evaluation scripts drive the HIP engine: they need --accelerator gpu and a visible MI355X (there is no CPU path)")
    logger.info("Creating model")
    model = build_model(config, activate_modal=("rgb", "text"))
    if config.get("tokenizer_path"):
        import transformers
        model.text.tokenizer = transformers.AutoTokenizer.from_pretrained(config.tokenizer_path, use_fast=False)
    return load_for_eval(model, config)


def generation_weights(config) -> str:
    """`bits: 8` in the YAML streams the e4m3 copies of the decoder weights through the MFMA GEMV (the reference loads LLM.int8 there)."""
    return "fp8" if int(config.get("bits", 16) or 16) == 8 else "bf16"


def load_for_eval(model, config):
    """The block every evaluation script repeats (main_cls.py:135-146): `--model-path` -> `custom_load_state_dict`, then eval mode."""
    path = config.get("model_path")
    if path is not None:
        logger.info(f"Loading pretrained checkpoint from {path}")
        msg = model.custom_load_state_dict(path)
        if msg is not None:
            logger.info(f"After loading, missing keys: {msg.missing_keys}, unexpected keys: {msg.unexpected_keys}")
    model.eval()
    return model


def balanced_accuracy(trues: Sequence[int], preds: Sequence[int]) -> float:
    """Mean per-class recall over the classes present in `trues` (sklearn.metrics.balanced_accuracy_score, main_cls.py:216)."""
    hit, cnt = defaultdict(int), defaultdict(int)
    for t, p in zip(trues, preds):
        cnt[int(t)] += 1
        hit[int(t)] += int(int(t) == int(p))
    return sum(hit[c] / cnt[c] for c in cnt) / len(cnt)


def eval_entry(main, config, seed_offset: Optional[int] = None):
    """The `__main__` block the four evaluation scripts share (main_cls.py:222-249): process group, logger, seeds, config dump, main()."""
    import numpy as np
    from .boundary import init_distributed, setup_logger
    config.rank, config.local_rank, config.world_size = init_distributed()
    config.is_distribute = config.world_size > 1
    config.adjust_norm = False
    setup_logger("train", output=config.output, rank=config.rank)
    os.makedirs(config.output, exist_ok=True)
    seed = config.seed + (get_rank() if config.is_distribute else 0)
    torch.manual_seed(seed)
    np.random.seed(seed)
    if config.rank == 0:
        path = os.path.join(config.output, "config.json")
        with open(path, "w") as f:
            json.dump(dict(config), f, indent=4, default=str)
        logger.info(f"Full config saved to {path}")
    if config.get("wandb", False):
        raise SystemExit("--wandb True: wandb is not available offline")
    return main(config)
