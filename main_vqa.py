#!/usr/bin/env python
"""RSVQA (LR / HR) evaluation by generation on the gfx950 engine - the reference's `main_vqa.py` call sequence over the `lhrs.*` surface
(/root/reference main_vqa.py:125-225):

    python main_vqa.py -c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --data-path <image dir> --data-target <RSVQA json dir> \\
        --data-type LR --batch-size 8 --accelerator gpu

build_model -> RSVQALR / RSVQAHR(split="test", token_prefix="<image>[VQA] ") + DataCollatorForVQASupervisedDataset (prompts LEFT-padded) ->
greedy batched `model.generate(attention_mask=...)`, 50 new tokens -> per-rank json -> merged, de-duplicated by question id ->
`TextVQAAccuracyEvaluator` (total and per question type).  The workers decode the .tif pictures; the CLIP transform runs on the
device inside `generate` (uint8 batch -> `lhrs_image_preprocess`).
"""
import json
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.CustomTrainer.utils.distribute import is_main_process  # noqa: E402
from lhrs.Dataset import RSVQAHR, RSVQALR, DataCollatorForVQASupervisedDataset  # noqa: E402
from lhrs.Dataset.build_transform import build_vlp_transform  # noqa: E402
from lhrs_bot_amd.evaluation import (TextVQAAccuracyEvaluator, eval_entry, eval_model, eval_parse_option, generation_weights,  # noqa: E402
                                     save_result)

logger = logging.getLogger("train")


def parse_option(args=None):
    return eval_parse_option(args, data_target=True, data_type=True)


def main(config):
    model = eval_model(config)
    tokenizer = model.text.tokenizer
    cls = RSVQAHR if config.data_type == "HR" else RSVQALR
    dataset = cls(root=config.data_target, image_root=config.data_path, image_transform=build_vlp_transform(config, is_train=False), split="test",
                  token_prefix="<image>[VQA] ", prompt_type=config.get("prompt_template", "llava_llama_2"), tokenizer=tokenizer)
    logger.info(f"Data Length: {len(dataset)}")
    data_loader = torch.utils.data.DataLoader(dataset, num_workers=int(config.workers), pin_memory=True, batch_size=int(config.batch_size), shuffle=False,
                                              collate_fn=DataCollatorForVQASupervisedDataset(tokenizer))
    preds = []
    with torch.no_grad():
        for batch in data_loader:
            output_ids = model.generate(input_ids=batch["questions"], images=batch["images"], do_sample=False, num_beams=1, temperature=1.0, top_p=1.0,
                                        attention_mask=batch["attn_mask"], max_new_tokens=50, weights=generation_weights(config))
            outputs = [o.strip() for o in tokenizer.batch_decode(output_ids, skip_special_tokens=True)]
            preds += [dict(pred=p, target=t, types=ty, question_id=q) for p, t, ty, q in zip(outputs, batch["targets"], batch["types"], batch["questions_idx"])]
    save_result(preds, config.output, "eval_save_file", "question_id")
    if not is_main_process():
        return None
    with open(os.path.join(config.output, "eval_save_file.json")) as f:
        predictions = json.load(f)
    total, per_type = TextVQAAccuracyEvaluator().eval_pred_list(predictions, return_types=True)
    logger.info(f"Total: {100.0 * total}")
    return dict(total=100.0 * total, per_type=per_type, predictions=predictions)


if __name__ == "__main__":
    eval_entry(main, parse_option())
