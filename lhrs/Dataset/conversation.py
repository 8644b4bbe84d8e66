"""lhrs.Dataset.conversation (conversation.py).  NB the datasets re-bind `lhrs_bot_amd.conversation.default_conversation`; read the
current template through `lhrs_bot_amd.conversation.default_conversation`, this name is the import-time default (llava_llama_2) that
cli_qa.py copies."""
from lhrs_bot_amd.conversation import (Conversation, SeparatorStyle, conv_llava_llama_2, conv_llava_plain, conv_llava_v1, conv_templates,  # noqa: F401
                                       conv_vicuna_v1, default_conversation)
