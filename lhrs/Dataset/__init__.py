"""lhrs.Dataset (lhrs/Dataset/__init__.py:1-15): loaders, transforms, prompt templates, training and evaluation datasets."""
from .build_loader import build_loader, build_zero_shot_loader  # noqa: F401
from .build_transform import build_cls_transform, build_vlp_transform  # noqa: F401
from .cap_dataset import (CapEvalDataset, CaptionDataset, CaptionDatasetVQA, DataCollatorForSupervisedDataset,  # noqa: F401
                          DataCollatorForVGSupervisedDataset, InstructDataset, VGEvalDataset, conversation_lib)
from .ImageFolderInstance import CLASS_NAME_MAP, ImageFolderInstance  # noqa: F401
from .rsvqa import RSVQAHR, RSVQALR, DataCollatorForVQASupervisedDataset  # noqa: F401
from .UCM import UCM  # noqa: F401
