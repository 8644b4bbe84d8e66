"""lhrs.models (lhrs/models/__init__.py:1-9): the prompt constants, `build_model`, `tokenizer_image_token`."""
from lhrs_bot_amd.text import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_TOKEN,  # noqa: F401
                               IGNORE_INDEX, IMAGE_TOKEN_INDEX)
from lhrs_bot_amd.data import tokenizer_image_token  # noqa: F401
from lhrs_bot_amd.unibind import UniBind, build_model  # noqa: F401

build_vlm_model = build_model  # lhrs/models/build.py: the same factory under its second name
