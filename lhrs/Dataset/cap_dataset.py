"""lhrs.Dataset.cap_dataset: datasets, collators and the tokenisation rules (cap_dataset.py:77-327, 330-486, 649-775, 775-854, 857-1084)."""
from lhrs_bot_amd import conversation as conversation_lib  # noqa: F401
from lhrs_bot_amd.data import (DataCollatorForSupervisedDataset, DataCollatorForVGSupervisedDataset, preprocess, preprocess_llama_2,  # noqa: F401
                               preprocess_multimodal, preprocess_plain, preprocess_v1, tokenizer_image_token)
from lhrs_bot_amd.datasets import (CaptionDataset, CaptionDatasetVQA, InstructDataset, InstructDatasetWithTaskId, RS5MDataset,  # noqa: F401
                                   pre_caption, valid_path)
from lhrs_bot_amd.eval_datasets import CapEvalDataset, VGEvalDataset  # noqa: F401
