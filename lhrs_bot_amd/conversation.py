"""Prompt templates of the chat / data boundary: what `lhrs.Dataset.conversation` gives the entry scripts.

Surface kept from /root/reference lhrs/Dataset/conversation.py (`SeparatorStyle` :6-13, `Conversation` :16-136, templates :239-395):
`conv.copy()`, `conv.roles`, `conv.messages`, `conv.append_message(role, msg)`, `conv.get_prompt()`, `conv.sep / sep2 / sep_style /
version / system`, the module-level `default_conversation` and `conv_templates[...]` (cli_qa.py:12,90-91,160-169; cap_dataset.py sets
`default_conversation = conv_templates[prompt_type]`).

Built around ONE rendering rule instead of a branch per style: a style is a small table (`_STYLES`) saying how the system text opens
the prompt, how a non-empty turn of parity 0 / 1 is written and what an empty turn (the slot the model is about to fill) leaves
behind.  The image-tuple messages of the reference's gradio demo are outside the hot path and are not modelled (a tuple's first
element is used as the text).  Pinned int-exactly through the tokenised prompts of tests/golden/data_boundary.json and
tests/golden/datasets.json.
"""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import Callable, Dict, List, Optional, Sequence


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()


@dataclasses.dataclass(frozen=True)
class _Style:
    head: Callable[["Conversation"], str]                       # text before the first turn
    turn: Callable[["Conversation", int, str, str], str]        # (conv, index, role, message) -> text of a filled turn
    open_turn: Callable[["Conversation", int, str], str]        # (conv, index, role) -> text of an empty turn
    finish: Callable[["Conversation", str], str] = lambda c, s: s


def _llama2_turn(c: "Conversation", i: int, role: str, msg: str) -> str:
    if i % 2:                                                    # assistant turn: " answer </s>"
        return " " + msg + " " + c.sep2
    if i == 0:                                                   # the system text rides inside the first instruction
        msg = "<<SYS>>\n" + c.system + "\n<</SYS>>\n\n" + msg
    return c.sep + "[INST] " + msg + " [/INST]"


_STYLES: Dict[SeparatorStyle, _Style] = {
    SeparatorStyle.SINGLE: _Style(lambda c: c.system + c.sep, lambda c, i, r, m: r + ": " + m + c.sep, lambda c, i, r: r + ":"),
    SeparatorStyle.TWO: _Style(lambda c: c.system + c.sep, lambda c, i, r, m: r + ": " + m + (c.sep, c.sep2)[i % 2], lambda c, i, r: r + ":"),
    SeparatorStyle.MPT: _Style(lambda c: c.system + c.sep, lambda c, i, r, m: r + m + c.sep, lambda c, i, r: r),
    SeparatorStyle.PLAIN: _Style(lambda c: c.system, lambda c, i, r, m: m + (c.sep, c.sep2)[i % 2], lambda c, i, r: ""),
    # str.lstrip(chars) strips any leading run of '<', 's', '>' characters - kept, it is what the reference's prompts went through
    SeparatorStyle.LLAMA_2: _Style(lambda c: "", _llama2_turn, lambda c, i, r: "", lambda c, s: s.lstrip(c.sep)),
}


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Sequence[str]
    messages: List[List[Optional[str]]]
    offset: int = 0
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def get_prompt(self) -> str:
        st = _STYLES[self.sep_style]
        out = st.head(self)
        for i, (role, msg) in enumerate(self.messages):
            if isinstance(msg, tuple):
                msg = msg[0]
            if self.sep_style == SeparatorStyle.LLAMA_2 and i == 0:
                assert msg, "first message should not be none"
                assert role == self.roles[0], "first message should come from user"
            out += st.turn(self, i, role, msg) if msg else st.open_turn(self, i, role)
        return st.finish(self, out)

    def append_message(self, role, message) -> None:
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return dataclasses.replace(self, messages=[[r, m] for r, m in self.messages])

    def dict(self) -> Dict:
        return dict(system=self.system, roles=self.roles, messages=self.messages, offset=self.offset, sep=self.sep, sep2=self.sep2)


_ASSISTANT_BLURB = ("A chat between a curious human and an artificial intelligence assistant. "
                    "The assistant gives helpful, detailed, and polite answers to the human's questions.")

conv_vicuna_v1 = Conversation(system=_ASSISTANT_BLURB.replace("human", "user"), roles=("USER", "ASSISTANT"), messages=[], version="v1",
                              sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")
conv_llava_v1 = Conversation(system=_ASSISTANT_BLURB, roles=("USER", "ASSISTANT"), messages=[], version="v1", sep_style=SeparatorStyle.TWO,
                             sep=" ", sep2="</s>")
conv_llava_llama_2 = Conversation(
    system=("You are a helpful language and vision assistant. You are able to understand the visual content that the user provides, "
            "and assist the user with a variety of tasks using natural language."),
    roles=("USER", "ASSISTANT"), messages=[], version="llama_v2", sep_style=SeparatorStyle.LLAMA_2, sep="<s>", sep2="</s>")
conv_llava_plain = Conversation(system="", roles=("", ""), messages=[], sep_style=SeparatorStyle.PLAIN, sep="\n")

# the templates the shipped YAMLs name (`prompt_template`: "plain" in stage 1, "llava_llama_2" in stages 2/3 and evaluation) plus the
# two-separator "v1" family `preprocess` also dispatches on
conv_templates: Dict[str, Conversation] = {
    "plain": conv_llava_plain, "v0_plain": conv_llava_plain, "llava_llama_2": conv_llava_llama_2,
    "v1": conv_vicuna_v1, "vicuna_v1": conv_vicuna_v1, "llava_v1": conv_llava_v1,
}
default_conversation = conv_llava_llama_2
