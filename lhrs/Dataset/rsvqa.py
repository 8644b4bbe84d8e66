"""lhrs.Dataset.rsvqa (rsvqa.py)."""
from lhrs_bot_amd.eval_datasets import (RSVQA, RSVQAHR, RSVQALR, Compose, DataCollatorForVQASupervisedDataset, RSVQAxBEN, ToTensor)  # noqa: F401
