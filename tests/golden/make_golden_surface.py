"""tests/golden/import_surface.json: the Python surface the reference's entry scripts use, read from the
reference's SOURCE with `ast` (nothing is imported or executed, no text is copied - only names):

  * imports:    every `from lhrs... import a, b` of main_pretrain_stage{1,2,3}.py, cli_qa.py and the evaluation scripts
                main_cls.py / main_vqa.py / main_vg.py / main_bench_gen.py
  * calls:      inside each script's `main()` (and its `__main__` block) every call with its positional count and keyword names, in order
  * signatures: parameter names (and which have defaults) of the functions / methods those calls land on, from the lhrs/ sources

tests/test_surface_cpu.py replays all three against this repo's `lhrs` package.  Build container only."""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPTS = ["main_pretrain_stage1.py", "main_pretrain_stage2.py", "main_pretrain_stage3.py", "cli_qa.py",
           "main_cls.py", "main_vqa.py", "main_vg.py", "main_bench_gen.py"]  # the last four: the evaluation callers of generate (§8 f-3)


def dotted(node):
    if isinstance(node, ast.Name):
        return node.id
    if isinstance(node, ast.Attribute):
        base = dotted(node.value)
        return base + "." + node.attr if base else None
    return None


def calls_in(body):
    out = []
    for stmt in body:
        for node in sorted((n for n in ast.walk(stmt) if isinstance(n, ast.Call)), key=lambda n: (n.lineno, n.col_offset)):
            name = dotted(node.func)
            if name:
                out.append({"callee": name, "n_pos": len(node.args), "kw": [k.arg for k in node.keywords if k.arg], "line": node.lineno})
    return out


def script_surface(path):
    tree = ast.parse(open(path).read())
    imports = []
    for n in tree.body:
        if isinstance(n, ast.ImportFrom) and n.module and n.module.split(".")[0] == "lhrs":
            imports.append({"module": n.module, "names": [a.name for a in n.names]})
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    tail = next((n for n in tree.body if isinstance(n, ast.If) and "__name__" in ast.dump(n.test)), None)
    return {"imports": imports, "main_calls": calls_in(main.body), "entry_calls": calls_in(tail.body) if tail else []}


def params(fn):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args]
    n_def = len(a.defaults)
    return {"params": [n for n in names if n != "self"], "required": [n for n in names[: len(names) - n_def] if n != "self"],
            "kwonly": [x.arg for x in a.kwonlyargs], "var_kw": a.kwarg is not None, "var_pos": a.vararg is not None}


def find(path, qual):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    scope = tree.body
    parts = qual.split(".")
    for i, p in enumerate(parts):
        node = next(n for n in scope if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == p)
        if i + 1 < len(parts):
            scope = node.body
    return params(node)


SIGS = {
    "lhrs.models.build_model": ("lhrs/models/build.py", "build_model"),
    "lhrs.models.tokenizer_image_token": ("lhrs/models/text_modal.py", "tokenizer_image_token"),
    "lhrs.optimizer.build_optimizer": ("lhrs/optimizer/build_optimizer.py", "build_optimizer"),
    "lhrs.Dataset.build_loader.build_loader": ("lhrs/Dataset/build_loader.py", "build_loader"),
    "lhrs.Dataset.build_transform.build_vlp_transform": ("lhrs/Dataset/build_transform.py", "build_vlp_transform"),
    "lhrs.CustomTrainer.utils.auto_resume_helper": ("lhrs/CustomTrainer/utils/misc.py", "auto_resume_helper"),
    "lhrs.CustomTrainer.utils.setup_logger": ("lhrs/CustomTrainer/utils/logger.py", "setup_logger"),
    "lhrs.CustomTrainer.utils.str2bool": ("lhrs/CustomTrainer/utils/misc.py", "str2bool"),
    "lhrs.CustomTrainer.deepspeed_init_distributed": ("lhrs/CustomTrainer/utils/distribute.py", "deepspeed_init_distributed"),
    "lhrs.CustomTrainer.EpochBasedTrainer.EpochBasedTrainer": ("lhrs/CustomTrainer/EpochBasedTrainer.py", "EpochBasedTrainer.__init__"),
    "lhrs.CustomTrainer.IterBasedTrainer.IterBasedTrainer": ("lhrs/CustomTrainer/IterBasedTrainer.py", "IterBasedTrainer.__init__"),
    "Trainer.__init__": ("lhrs/CustomTrainer/trainer.py", "Trainer.__init__"),
    "Trainer.train": ("lhrs/CustomTrainer/trainer.py", "Trainer.train"),
    "UniBind.prepare_for_training": ("lhrs/models/UniBind.py", "UniBind.prepare_for_training"),
    "UniBind.generate": ("lhrs/models/UniBind.py", "UniBind.generate"),
    "UniBind.encode_image": ("lhrs/models/UniBind.py", "UniBind.encode_image"),
    "UniBind.custom_save_checkpoint": ("lhrs/models/UniBind.py", "UniBind.custom_save_checkpoint"),
    "UniBind.custom_load_state_dict": ("lhrs/models/UniBind.py", "UniBind.custom_load_state_dict"),
    "lhrs.utils.KeywordsStoppingCriteria": ("lhrs/utils/eval_utils.py", "KeywordsStoppingCriteria.__init__"),
    "Conversation.append_message": ("lhrs/Dataset/conversation.py", "Conversation.append_message"),
    "Conversation.get_prompt": ("lhrs/Dataset/conversation.py", "Conversation.get_prompt"),
    "Conversation.copy": ("lhrs/Dataset/conversation.py", "Conversation.copy"),
    "ConfigArgumentParser.parse_args": ("lhrs/CustomTrainer/utils/config_parser.py", "ConfigArgumentParser.parse_args"),
    # evaluation callers (§8 f-3)
    "lhrs.CustomTrainer.init_distributed": ("lhrs/CustomTrainer/utils/distribute.py", "init_distributed"),
    "lhrs.Dataset.build_loader.build_zero_shot_loader": ("lhrs/Dataset/build_loader.py", "build_zero_shot_loader"),
    "lhrs.Dataset.build_transform.build_cls_transform": ("lhrs/Dataset/build_transform.py", "build_cls_transform"),
    "lhrs.Dataset.UCM.UCM": ("lhrs/Dataset/UCM.py", "UCM.__init__"),
    "lhrs.Dataset.millionaid_eval.MillionAidEval": ("lhrs/Dataset/millionaid_eval.py", "MillionAidEval.__init__"),
    "lhrs.Dataset.ImageFolderInstance.ImageFolderInstance": ("lhrs/Dataset/ImageFolderInstance.py", "ImageFolderInstance.__init__"),
    "lhrs.Dataset.meterml.METERMLDataset": ("lhrs/Dataset/meterml.py", "METERMLDataset.__init__"),
    "lhrs.Dataset.rsvqa.RSVQA": ("lhrs/Dataset/rsvqa.py", "RSVQA.__init__"),
    "lhrs.Dataset.rsvqa.RSVQALR": ("lhrs/Dataset/rsvqa.py", "RSVQALR.__init__"),
    "lhrs.Dataset.rsvqa.RSVQAHR": ("lhrs/Dataset/rsvqa.py", "RSVQAHR.__init__"),
    "lhrs.Dataset.cap_dataset.VGEvalDataset": ("lhrs/Dataset/cap_dataset.py", "VGEvalDataset.__init__"),
    "lhrs.Dataset.cap_dataset.CapEvalDataset": ("lhrs/Dataset/cap_dataset.py", "CapEvalDataset.__init__"),
}

out = {"scripts": {s: script_surface(os.path.join(REF, s)) for s in SCRIPTS},
       "signatures": {k: find(*v) for k, v in SIGS.items()},
       "model_attributes": ["text.tokenizer", "rgb", "rgb_pooler", "text.text_encoder", "stage"]}
json.dump(out, open(os.path.join(HERE, "import_surface.json"), "w"), indent=1)
for s, d in out["scripts"].items():
    print(s, sum(len(i["names"]) for i in d["imports"]), "imported names,", len(d["main_calls"]), "calls in main()")
print(len(out["signatures"]), "signatures;", os.path.getsize(os.path.join(HERE, "import_surface.json")) // 1024, "KiB")
