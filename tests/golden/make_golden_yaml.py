"""tests/golden/yaml_surface.json: the reference's shipped configuration surface as DATA ("Config surface").

  * yaml:       the parsed key/value tree of every Config/multi_modal_{stage1,stage2,stage3,eval}.yaml (yaml.safe_load -> JSON)
  * ds_config:  what the reference's own `build_ds_config` (main_pretrain_stage1.py:28-85; stage 2 / 3 carry the identical function)
                returns for each training YAML merged with the launcher flags of Script/train_stage{1,2,3}.sh (batch size, accumulation
                steps).  The function is lifted out of the script with `ast` and executed here on its own (the script's module level
                imports deepspeed / wandb, absent in this image); nothing of its text is stored - only the returned dicts.

tests/test_surface_cpu.py pushes every tree through this repo's parse_option -> build_ds_config -> initialize and compares the dict;
tests/test_trainer_gpu.py runs the stage-2 / stage-3 drivers from the same trees.  Build container only (reads /root/reference)."""
import ast
import json
import os

import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
LAUNCH = {"stage1": dict(batch_size=8, accumulation_steps=1), "stage2": dict(batch_size=4, accumulation_steps=1),   # Script/train_stage{1,2,3}.sh
          "stage3": dict(batch_size=4, accumulation_steps=1)}


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def reference_build_ds_config():
    tree = ast.parse(open(os.path.join(REF, "main_pretrain_stage1.py")).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "build_ds_config")
    fn.args.args[0].annotation = None
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference build_ds_config>", "exec"), ns)
    return ns["build_ds_config"]


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    return o


def main():
    build = reference_build_ds_config()
    out = {"yaml": {}, "ds_config": {}, "launch": LAUNCH}
    for name in ("stage1", "stage2", "stage3", "eval"):
        tree = yaml.safe_load(open(os.path.join(REF, "Config", f"multi_modal_{name}.yaml")))
        out["yaml"][name] = tree
        if name in LAUNCH:
            out["ds_config"][name] = jsonable(build(AttrDict({**tree, **LAUNCH[name]})))
    with open(os.path.join(HERE, "yaml_surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: sorted(v) for k, v in out["ds_config"].items()})


if __name__ == "__main__":
    main()
