#!/usr/bin/env python
"""Zero-shot scene classification by generation on the gfx950 engine - the reference's `main_cls.py` call sequence over the `lhrs.*`
surface (/root/reference main_cls.py:125-220):

    python main_cls.py -c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --data-path <dataset root> --batch-size 8 \\
        --accelerator gpu --opts eval.dataset UCM          # or METERML, or an ImageFolder set such as AID (the YAML's default)

build_model -> build_zero_shot_loader(config, mode="zero_shot_cls") -> ONE prompt for the whole run ("[CLS] Choose the best categories
describe the image from: [class list]" in the default conversation template) repeated over the batch -> greedy `model.generate` ->
`batch_decode` -> `classname_2_idx` -> balanced accuracy (mean per-class recall) and the per-class report.

The loader's workers decode; Resize(256) / CenterCrop(224) / ImageNet normalisation run on the device per batch
(`lhrs_image_preprocess`); every batch is ONE prefill + one captured hipGraph per generated token for all its rows.
"""
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.Dataset.build_loader import build_zero_shot_loader  # noqa: E402
from lhrs.Dataset.conversation import default_conversation  # noqa: E402
from lhrs.models import IMAGE_TOKEN_INDEX, tokenizer_image_token  # noqa: E402
from lhrs_bot_amd.evaluation import (balanced_accuracy, class_prompt, classname_2_idx, eval_entry, eval_model, eval_parse_option,  # noqa: E402
                                     generation_weights)

logger = logging.getLogger("train")


def parse_option(args=None):
    return eval_parse_option(args)


def main(config):
    model = eval_model(config)
    data_loader = build_zero_shot_loader(config, mode="zero_shot_cls")
    dataset = data_loader.dataset
    all_classes, turn = class_prompt(dataset.classes if hasattr(dataset, "classes") else dataset.CLASS_NAME, tune_im_start=config.get("tune_im_start", False))
    classes_2_idx = {name: idx for idx, name in enumerate(all_classes)}
    conv = default_conversation.copy()
    conv.append_message(conv.roles[0], turn)
    conv.append_message(conv.roles[1], None)
    tokenizer = model.text.tokenizer
    input_ids = tokenizer_image_token(conv.get_prompt(), tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).repeat(int(config.batch_size), 1)
    max_new = 20 if config["eval"]["dataset"] != "METERML" else 30
    preds, trues = [], []
    with torch.no_grad():
        for image, target in data_loader:
            output_ids = model.generate(input_ids=input_ids[: image.shape[0]], images=image, do_sample=False, num_beams=1, temperature=1.0, top_p=1.0,
                                        max_new_tokens=max_new, weights=generation_weights(config))
            preds += tokenizer.batch_decode(output_ids, skip_special_tokens=True)
            trues.append(torch.as_tensor(target).cpu())
    pred_idx = classname_2_idx(preds, classes_2_idx)
    trues = torch.cat(trues).tolist()
    score = balanced_accuracy(trues, pred_idx)
    try:
        from sklearn.metrics import classification_report
        logger.info(classification_report(trues, pred_idx, digits=3, labels=list(range(len(all_classes))), target_names=all_classes, zero_division=0))
    except ImportError:  # the report is a log line, the score below is the result
        pass
    logger.info(score)
    return dict(mean_per_class_recall=score, preds=preds, pred_idx=pred_idx, trues=trues, classes=all_classes)


if __name__ == "__main__":
    eval_entry(main, parse_option())
