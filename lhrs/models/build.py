"""lhrs.models.build (lhrs/models/build.py:16-22)."""
from lhrs_bot_amd.unibind import build_model  # noqa: F401

build_vlm_model = build_model
