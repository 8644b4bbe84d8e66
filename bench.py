#!/usr/bin/env python
"""Stage-1 pretrain throughput of the gfx950 engine (BASELINE.json metric).

One "step" = one full stage-1 optimizer step over one micro-batch of synthetic samples per GPU:
  CLIP ViT-L/14 forward (22 useful layers) -> AttnPooler forward -> splice -> LLaMA2-7B (32 layers) forward -> loss
  -> LLaMA activation-gradient backward -> AttnPooler backward (dW + dX) -> [RCCL gradient all-reduce] -> Adan step.
Workload = BASELINE.json configs[1]: 224x224 images, input_ids = [BOS, <image>, 128 caption tokens] => S = 273,
random-init LLaMA-2-7B / ViT-L/14 shapes (no weights exist offline), bf16 compute, inputs resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (dominant kernel = the bf16 MFMA
GEMM, timed live with HIP events on its launch stream inside the timed region) and, at N=1, `cpu_baseline` (the CPU
oracle timed on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)


def f_alg(S: int) -> float:
    """Algorithmic FLOPs per sample (SURVEY.md §8d / BASELINE.md §2), projector-only, no recompute."""
    return 278.8e9 + 2 * S * 13.214e9 + 1572864.0 * S * S


def make_batch(B, T, device, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    ids[:, 1] = -200
    labels = ids.clone()
    labels[:, :2] = -100
    rgb = torch.randn(B, 3, 224, 224, generator=g)
    return dict(rgb=rgb.to(device), input_ids=ids.to(device), labels=labels.to(device), attention_mask=ids.ne(0).to(device))


def cpu_baseline(S: int, budget_s: float = 25.0):
    """Oracle (CPU restatement of the reference, fp32 torch) on a bounded sample (~10-15 s of CPU work): eight samples through ViT +
    pooler (fwd+bwd) + FOUR LLaMA-7B-width decoder layers (fwd + activation-gradient bwd) + final norm/lm_head/CE, the per-layer time
    extrapolated to 32 layers.  Reported, not optimised-for."""
    from oracle import lhrs_oracle as O
    from oracle import params as OP

    NL = 4  # decoder layers actually timed
    threads = min(32, os.cpu_count() or 1)  # 256 OpenMP threads on these shapes are slower than 32 (measured)
    torch.set_num_threads(threads)
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=NL)}
    for L in [P["pooler"]] + P["pooler"]["layers"]:
        for v in L.values():
            if torch.is_tensor(v):
                v.requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    NB = 8  # samples in the CPU sample
    rgb = torch.randn(NB, 3, 224, 224, generator=g)
    t0 = time.perf_counter()
    with torch.no_grad():
        taps = O.vit_forward(P["vit"], rgb)
    t_vit = time.perf_counter() - t0
    t0 = time.perf_counter()
    img = O.pooler_forward(P["pooler"], taps)
    img.backward(torch.randn(img.shape, generator=g) * 0.01)
    t_pool = time.perf_counter() - t0
    x = torch.randn(NB, S, 4096, generator=g).requires_grad_(True)
    labels = torch.randint(3, 32000, (NB, S), generator=g)
    labels[:, :146] = -100
    t0 = time.perf_counter()
    h = O.llama_hidden(P["llama"], x, None)
    loss = O.causal_lm_loss(P["llama"], h, labels)
    loss.backward()
    t_l1 = time.perf_counter() - t0
    # the same without the decoder layer = norm + lm_head + CE
    x2 = torch.randn(NB, S, 4096, generator=g).requires_grad_(True)
    t0 = time.perf_counter()
    h2 = O._rms(x2, P["llama"]["norm_w"], 1e-5)
    O.causal_lm_loss(P["llama"], h2, labels).backward()
    t_head = time.perf_counter() - t0
    t_layer = max(t_l1 - t_head, 1e-6) / NL
    t_full = t_vit + t_pool + t_head + 32 * t_layer
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": NB / t_full, "unit": "samples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "sample": (f"{NB} samples, S={S}: ViT-L/14 fwd {t_vit:.2f}s + AttnPooler fwd+bwd {t_pool:.2f}s + lm_head/CE fwd+bwd {t_head:.2f}s "
                       f"+ {NL} of 32 LLaMA-7B layers fwd+dX-bwd {NL * t_layer:.2f}s, per-layer time extrapolated x32 (oracle/lhrs_oracle.py, fp32 torch CPU)")}



class _SmiSampler:
    """Background rocm-smi sampling (shader clock, package power) during the timed region, rank 0 only.  Evidence for the DVFS
    ceiling the MFMA-bound step runs under (MI355X caps at 1400 W: random-valued bf16 operands pull sclk from 2.4 GHz to ~1.7 GHz);
    best effort - any failure just leaves the fields null."""

    def __init__(self):
        import threading
        self.samples, self._stop = [], False
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                c = re.search(r'"sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', o)
                p = re.search(r'Graphics Package Power \(W\)"\s*:\s*"([\d.]+)"', o)
                if c:
                    self.samples.append((int(c.group(1)), float(p.group(1)) if p else None))
            except Exception:  # noqa: BLE001
                return
            time.sleep(0.1)

    def start(self):
        try:
            self._th.start()
        except Exception:  # noqa: BLE001
            pass

    def stop(self):
        self._stop = True
        try:
            self._th.join(timeout=6)
        except Exception:  # noqa: BLE001
            pass
        s = self.samples[1:] if len(self.samples) > 2 else self.samples
        if not s:
            return None, None
        pw = [p for _, p in s if p is not None]
        return sum(c for c, _ in s) / len(s), (sum(pw) / len(pw) if pw else None)

def spawn_ranks(n: int):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU, as the reference's
    `deepspeed --num_gpus=8` does (Script/train_stage1.sh:6-17).  Rank 0 prints the JSON line; the exit status is the launcher's."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--micro-batch", type=int, default=30,
                    help="samples per GPU per step (DESIGN.md §4; the reference script uses 8 on 80 GB parts: Script/train_stage1.sh:11)")
    ap.add_argument("--caption-tokens", type=int, default=128)
    ap.add_argument("--llama-layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-policy", type=int, default=None, help="kernel A/B tests only: lhrs_gemm_set_policy value")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2, 3],
                    help="1: projector-only + Adan (the headline metric, BASELINE configs[1..2]); 3: LoRA r=8 on q,k,v,o + AdamW, projector "
                         "frozen (BASELINE configs[3]); 2: LoRA r=128 on every linear + projector, AdamW (Config/multi_modal_stage2.yaml)")
    ap.add_argument("--comm-dtype", default="bfloat16", choices=["float32", "bfloat16"],
                    help="dtype of the gradient all-reduce (N > 1): bfloat16 = 160 MB per step at stage 1, what DeepSpeed bf16 ZeRO-2 moves (SURVEY §8e)")
    ap.add_argument("--bits", type=int, default=16, choices=[16, 8],
                    help="stages 2/3 only: 8 = frozen decoder linears in e4m3 (the reference's `bits: 8` base weights)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus)  # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the LHRS hot path has no CPU fallback (cpu_baseline is only the checker)")
    share = os.environ.get("LHRS_SHARE_GPU") == "1"  # smoke test on a 1-GPU box: every rank on GPU 0 (gloo backend only)
    if world > torch.cuda.device_count() and not share:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible (LHRS_SHARE_GPU=1 runs the ranks on "
                         "one device over gloo: a plumbing test, not a measurement)")
    if share and world > torch.cuda.device_count():
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" = RCCL over xGMI whenever every rank has its own device; gloo only for the shared-device smoke test
        backend = os.environ.get("LHRS_DIST_BACKEND") or ("nccl" if world <= torch.cuda.device_count() else "gloo")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from lhrs_bot_amd import _lib
    from lhrs_bot_amd.engine import LHRSEngine
    from lhrs_bot_amd.unibind import UniBind

    lib = _lib.load()
    if a.gemm_policy is not None:
        lib.lhrs_gemm_set_policy(a.gemm_policy)
    if os.environ.get("LHRS_GEMM_MIN_TILES"):  # kernel A/B tests only
        lib.lhrs_gemm_set_min_tiles(int(os.environ["LHRS_GEMM_MIN_TILES"]))
    B, T = a.micro_batch, a.caption_tokens + 2
    S = T - 1 + 144
    model = UniBind(("rgb", "text"), None, device=dev, llama_layers=a.llama_layers).init_random(seed=0)  # same weights on every rank
    if a.stage == 1:
        model.prepare_for_training()
        engine = LHRSEngine(model, optimizer="adanp", lr=2e-4, weight_decay=0.0, max_grad_norm=0.3, comm_dtype=getattr(torch, a.comm_dtype))
    else:
        if a.stage == 3:
            model.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"))
        else:
            model.enable_lora(r=128, alpha=256, dropout=0.05)  # Config/multi_modal_stage2.yaml:81-86 (train mode: dropout active)
        if a.bits == 8:
            model.text.quantize_base(8)
        model.prepare_for_training(freeze_text=False, tune_rgb_pooler=a.stage == 2)
        engine = LHRSEngine(model, optimizer="adamw", lr=1e-4 if a.stage == 3 else 2e-4, weight_decay=0.0, max_grad_norm=1.0,
                            comm_dtype=getattr(torch, a.comm_dtype))
    batch = make_batch(B, T, dev, seed=322 + rank)  # reference seed convention (main_pretrain_stage1.py:281-287)

    def step():
        out = engine(batch)
        engine.backward(out["total_loss"])
        engine.step()
        return out["total_loss"]

    for _ in range(a.warmup):
        loss = step()
    torch.cuda.synchronize()
    import ctypes
    _lib.check(lib.lhrs_gemm_profile_enable(6000), "gemm_profile_enable")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    smi = _SmiSampler() if rank == 0 else None
    if smi is not None:
        smi.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sclk, watts = smi.stop() if smi is not None else (None, None)
    prof = (ctypes.c_double * 5)()
    _lib.check(lib.lhrs_gemm_profile_read(ctypes.addressof(prof)), "gemm_profile_read")
    lib.lhrs_gemm_profile_enable(0)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())

    if rank == 0:
        sps = world * B * a.steps / dt
        n_samp, ms, fl = prof[0], prof[1], prof[2]
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        scale_layers = a.llama_layers / 32.0
        traffic = None  # HBM-side bytes per launch of the dominant kernel come from the separate rocprofv3 --pmc passes
        tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        if os.path.exists(tpath) and B == 30 and scale_layers == 1.0:
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        res = {
            "metric": ("stage-1 pretrain samples/sec (224^2 image + 128-tok caption)" if a.stage == 1 else
                       f"stage-{a.stage} LoRA train samples/sec (224^2 image + 128-tok sequence)"), "value": round(sps, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.bits == 16 or a.stage == 1 else "e4m3 base weights + bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: stage-1 projector-only, CLIP ViT-L/14@224 + AttnPooler + LLaMA2-7B "
                                   f"({a.llama_layers} layers), S={S}, random-init weights",
                       "micro_batch_per_gpu": B, "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}",
                       "optimizer": "adanp" if a.stage == 1 else "adamw", "stage": a.stage,
                       "lora": None if a.stage == 1 else ("r=8 on q,k,v,o, no dropout (text.eval())" if a.stage == 3 else "r=128 on all 7 linears, lora_dropout 0.05"),
                       "grad_allreduce": a.comm_dtype if world > 1 else "none",
                       "dist_backend": (torch.distributed.get_backend() if world > 1 else None)},
            "loss": round(final_loss, 4),
            "step_mfma_frac": round(sps / world * f_alg(S) / (PEAK_BF16_TFLOPS * 1e12), 4) if scale_layers == 1.0 and a.stage == 1 else None,
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_256r_kernel<ACT, 0> (256x256 tile, 16 waves, BK=64 double-buffered LDS stages via global_load_lds DMA, v_mfma_f32_32x32x16_bf16; the two launches per layer with a fused SwiGLU epilogue are timed by rocprof only)", "achieved": round(ach, 1),
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "launches_timed": int(n_samp), "avg_launch_us": round(1e3 * ms / max(n_samp, 1), 2),
                         "sclk_mhz_during_timed_region": round(sclk) if sclk else None, "package_power_w": round(watts) if watts else None,
                         "frac_of_peak_at_measured_clock": round(ach / (PEAK_BF16_TFLOPS * sclk / 2400.0), 4) if sclk else None,
                         "gemm_flops_share_of_step": round(prof[4] / a.steps / (B * f_alg(S)), 3) if scale_layers == 1.0 else None},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(S)
            except Exception as e:  # the checker must never take the product number down with it
                res["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
