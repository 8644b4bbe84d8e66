#!/usr/bin/env python
"""LHRS-Bench multiple-choice evaluation on the gfx950 engine - the reference's `main_bench_gen.py` call sequence over the `lhrs.*` surface
(/root/reference main_bench_gen.py:157-287):

    python main_bench_gen.py -c Config/multi_modal_eval.yaml --model-path <FINAL.pt dir> --data-path <image dir> --data-target <qa json> \\
        --accelerator gpu

For every picture and every (question, choices, answer, type ids) pair: the question + "Choices: ... Answer from the given choices with
A., B., C., D., etc." in the default conversation template -> greedy `model.generate`, 10 new tokens -> the first character of the answer
against the answer letter (`score_choice`) -> accuracy per question type and in total (percent, 2 decimals).  The picture is encoded by
the CLIP transform on the device; the reference runs one question at a time and so does this script.
"""
import json
import logging
import os
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lhrs.Dataset.build_transform import build_vlp_transform  # noqa: E402
from lhrs.Dataset.conversation import default_conversation  # noqa: E402
from lhrs.models import IMAGE_TOKEN_INDEX, tokenizer_image_token  # noqa: E402
from lhrs_bot_amd.evaluation import bench_question, eval_entry, eval_model, eval_parse_option, generation_weights, score_choice  # noqa: E402

logger = logging.getLogger("train")


def parse_option(args=None):
    return eval_parse_option(args, data_target=True, data_type=True)


def main(config):
    model = eval_model(config)
    vis_transform = build_vlp_transform(config, is_train=False)
    tokenizer = model.text.tokenizer
    image_root = Path(config.data_path)
    qa_data = json.load(open(config.data_target))
    id_2_type = dict(key.split(" ")[:2] for key in qa_data["qtype"])
    gathered = defaultdict(list)
    with torch.no_grad():
        for record in qa_data["data"]:
            from PIL import Image
            image_path = image_root / record["filename"]
            assert image_path.exists(), f"Image not found: {image_path}"
            image_tensor = vis_transform(Image.open(image_path).convert("RGB"), return_tensors="pt")["pixel_values"]
            for qa in record["qa_pairs"]:
                conv = default_conversation.copy()
                conv.append_message(conv.roles[0], bench_question(qa["question"], qa["choices"], tune_im_start=config.get("tune_im_start", False)))
                conv.append_message(conv.roles[1], None)
                input_ids = tokenizer_image_token(conv.get_prompt(), tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
                output_ids = model.generate(input_ids, images=image_tensor, do_sample=False, num_beams=1, temperature=1.0, top_p=1.0, max_new_tokens=10,
                                            weights=generation_weights(config))
                correct = score_choice(tokenizer.batch_decode(output_ids, skip_special_tokens=False)[0], qa["answer"])
                gathered["total"].append(correct)
                for type_id in qa["type"]:
                    gathered[id_2_type[type_id]].append(correct)
    result = {key: round(sum(v) / len(v) * 100, 2) for key, v in gathered.items()}
    for key, acc in result.items():
        if key != "total":
            logger.info(f"Type: {key}, accuracy: {acc}%")
    logger.info(f"Total accuracy: {result['total']}%")
    return result


if __name__ == "__main__":
    eval_entry(main, parse_option())
