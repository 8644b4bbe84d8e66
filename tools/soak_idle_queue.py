"""Soak runs for the idle-queue NaN of DESIGN.md 7: does a step that starts behind a DRAINED queue ever compute garbage?

  python tools/soak_idle_queue.py trainer [ITERS]     ONE process, the shipped stage-1 driver (main_pretrain_stage1.main, synthetic loader, micro-batch 8, log_period 1:
                                                      LoggerHook calls loss.item() - a full drain - after EVERY iteration); every iteration's loss and gradient norm must be
                                                      finite, and so must the masters at the end
  python tools/soak_idle_queue.py bench1 [STEPS]      ONE process, bench.py's step with LHRS_BENCH_IDLE_START_MS=5 (synchronise + 5 ms sleep in front of every step) and the
                                                      device-side finite trace (LHRS_BENCH_TRACE_FINITE=1)
  python tools/soak_idle_queue.py share8 [LAUNCHES] [STEPS]   the 8-ranks-on-ONE-device plumbing run (LHRS_SHARE_GPU=1, gloo) with the same injection, host-resident integers
                                                      (the shipped configuration), finite trace per rank; counts launches in which any rank went non-finite

Prints one summary line per mode; the logs go to gpurun_out/soak/."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "soak")
os.makedirs(OUT, exist_ok=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "trainer"


def trainer(iters):
    import math
    import torch
    import main_pretrain_stage1 as drv
    from lhrs.CustomTrainer.utils import ConfigDict
    cfg = ConfigDict(dict(stage=1, batch_size=8, data_path="synthetic", epoch_len=iters, output=os.path.join(os.environ.get("TMPDIR", "/tmp"), "soak_trainer_out"), workers=0, inf_sampler=False, prompt_template="plain", gpus=0, local_rank=0, rank=0, world_size=1,
                          optimizer="adanp", lr=2e-4, wd=0.0, max_grad_norm=0.3, epochs=1, llama_layers=int(os.environ.get("SOAK_LAYERS", "4")), seed=322, bf16=True,
                          fp16=False, accumulation_steps=1, tune_rgb_bk=False, tune_rgb_pooler=True, tune_im_start=False, lora=dict(enable=False),
                          schedule=dict(name="cosine", min_lr=0.0, warmup_epochs=1, warmup_method="linear", warmup_factor=0.1),
                          rgb_vision=dict(arch="vit_large", vit_name="openai/clip-vit-large-patch14"), text=dict(path="/nonexistent/Llama-2-7b-chat-hf"),
                          transform=dict(input_size=[224, 224]), log_period=1, is_distribute=False, enable_amp=True, accelerator="gpu", wandb=False))
    os.makedirs(os.path.join(cfg.output, "checkpoints"), exist_ok=True)
    t0 = time.time()
    t = drv.main(cfg)
    bad = [h["iter"] for h in t.history if not (math.isfinite(h["loss"]) and math.isfinite(h["grad_norm"]))]
    masters = all(bool(torch.isfinite(st.master).all()) for st in t.model.stores)
    print(f"soak trainer: {len(t.history)} iterations of main_pretrain_stage1.main (micro-batch 8, {cfg.llama_layers} decoder layers, log_period 1 = loss.item() every iteration, "
          f"ragged synthetic batches) in {time.time() - t0:.0f} s: non-finite iterations {bad[:10]} ({len(bad)}), masters finite at the end: {masters}; "
          f"first / last loss {t.history[0]['loss']:.4f} / {t.history[-1]['loss']:.4f}", flush=True)
    return 0 if not bad and masters else 1


def bench(launches, steps, gpus):
    fails, rows = 0, []
    for i in range(launches):
        env = dict(os.environ, LHRS_BENCH_IDLE_START_MS=os.environ.get("LHRS_BENCH_IDLE_START_MS", "5"), LHRS_BENCH_TRACE_FINITE="1", LHRS_BENCH_NO_SMI="1")
        if gpus > 1:
            env["LHRS_SHARE_GPU"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--llama-layers", "1", "--micro-batch", "2", "--steps", str(steps), "--warmup", "2",
               "--no-extra", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
        traces = re.findall(r"rank (\d+) finite-trace: (\d+) phases checked, (all finite|FIRST non-finite: '[^']*')", r.stderr)
        bad = [t for t in traces if t[2] != "all finite"]
        ok = r.returncode == 0 and len(traces) == gpus and not bad
        fails += 0 if ok else 1
        rows.append(f"launch {i}: rc {r.returncode}, {len(traces)} rank traces, {'all finite' if ok else 'FAIL ' + str(bad or r.stderr[-400:])}")
        open(os.path.join(OUT, f"{'share8' if gpus > 1 else 'bench1'}_log.txt"), "a").write(rows[-1] + "\n")
    phases = sum(int(t[1]) for t in traces) if launches else 0
    print(f"soak {'share8' if gpus > 1 else 'bench1'}: {launches} launch(es) x {gpus} rank(s) x {steps} steps, each step behind synchronise + "
          f"{os.environ.get('LHRS_BENCH_IDLE_START_MS', '5')} ms idle, host-resident integers, device-side finite trace level 1 ({phases} phases in the last launch): "
          f"{fails} launch(es) with a non-finite phase", flush=True)
    return 0 if fails == 0 else 1


if mode == "trainer":
    sys.exit(trainer(int(sys.argv[2]) if len(sys.argv) > 2 else 2000))
elif mode == "bench1":
    sys.exit(bench(1, int(sys.argv[2]) if len(sys.argv) > 2 else 2000, 1))
else:
    sys.exit(bench(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 40, 8))
