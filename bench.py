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
PEAK_HBM_GBS = 8000.0      # HBM3E peak (spec; ~6.3 TB/s achievable by a streaming copy)


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
    # pixels on the device; the small integer tensors in HOST memory, as the reference's DataLoader / collator delivers them (cap_dataset.py:775-810) and as
    # lhrs_bot_amd.trainer leaves them: the engine plans the step's integers on the host and uploads what the kernels read, so a step starts without a
    # device -> host copy + stream synchronisation.  LHRS_BENCH_HOST_INTS=0: device-resident integers (the pre-round-5 bench; the A/B of DESIGN.md 6)
    if os.environ.get("LHRS_BENCH_HOST_INTS", "1") != "0":
        return dict(rgb=rgb.to(device), input_ids=ids, labels=labels, attention_mask=ids.ne(0))
    return dict(rgb=rgb.to(device), input_ids=ids.to(device), labels=labels.to(device), attention_mask=ids.ne(0).to(device))


def cpu_baseline(S: int, layers: int = 32):
    """The CPU oracle (oracle/lhrs_oracle.py: the restatement of the reference's fp32 PyTorch path that tests/ pin to the imported reference) timed
    on this box's host cores, UN-extrapolated: one whole stage-1 step - ViT-L/14 forward, AttnPooler forward + backward, splice, all 32
    LLaMA-2-7B-width decoder layers forward + activation-gradient backward, lm_head + shifted CE - through `unibind_forward` + autograd.
      * `value`:   the headline shape (S = 273: 128-token captions), 2 samples, one step
      * `config1`: BASELINE configs[0] / SURVEY §8(d) config 1 (B = 4, T = 34 -> S = 177), one step
    32 distinct fp32 weight buffers per decoder tensor (27 GB; 4 seeded layers, each copied 8 times - the values do not matter to a clock, the
    memory traffic does).  One untimed single-sample step on ONE layer warms the thread pool and the allocator.  Reported, not optimised-for."""
    from oracle import lhrs_oracle as O
    from oracle import params as OP

    threads = min(32, os.cpu_count() or 1)  # 256 OpenMP threads on these shapes are slower than 32 (measured in round 1)
    torch.set_num_threads(threads)
    t_build = time.perf_counter()
    P = {"vit": OP.make_vit_params(seed=2), "pooler": OP.make_pooler_params(seed=1), "llama": OP.make_llama_params(seed=3, layers=4)}
    seeded = P["llama"]["layers"]
    P["llama"]["layers"] = [seeded[i % 4] if i < 4 else {k: v.clone() for k, v in seeded[i % 4].items()} for i in range(layers)]
    for L in [P["pooler"]] + P["pooler"]["layers"]:
        for v in L.values():
            if torch.is_tensor(v):
                v.requires_grad_(True)
    t_build = time.perf_counter() - t_build

    def zero():
        for L in [P["pooler"]] + P["pooler"]["layers"]:
            for v in L.values():
                if torch.is_tensor(v):
                    v.grad = None

    def one_step(B, T, layers=None):
        batch = make_batch(B, T, "cpu", seed=322)
        Q = P if layers is None else {**P, "llama": {**P["llama"], "layers": P["llama"]["layers"][:layers]}}
        t0 = time.perf_counter()
        loss = O.unibind_forward(Q, batch)
        loss.backward()
        dt = time.perf_counter() - t0
        zero()
        return dt, float(loss.detach())

    one_step(1, 34, layers=1)                       # warm-up (thread pool, allocator): not timed
    T_head = S - 143
    t_head, loss_head = one_step(2, T_head)
    t_c1, loss_c1 = one_step(4, 34)
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(2 / t_head, 4), "unit": "samples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model,
            "host_logical_cpus": os.cpu_count(),
            "cores_note": "32 OpenMP threads: on these shapes torch's CPU GEMMs are slower with all hardware threads than with 32 (measured in round 1)",
            "sample": (f"ONE full stage-1 step of 2 samples at S={S} through oracle.unibind_forward + autograd, all {layers} decoder layers of this run's model, nothing extrapolated: "
                       f"{t_head:.1f} s (loss {loss_head:.3f}); fp32 torch CPU; building the fp32 weights ({0.81 * layers + 1.5:.0f} GB) took {t_build:.1f} s (not timed)"),
            "config1": {"value": round(4 / t_c1, 4), "unit": "samples/s", "cores": threads,
                        "sample": f"BASELINE configs[0] (B=4, T=34, S=177), ONE full step, all {layers} layers, nothing extrapolated: {t_c1:.1f} s (loss {loss_c1:.3f})"}}



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

LLAMA_LINEAR_PARAMS_PER_LAYER = 12288 * 4096 + 4096 * 4096 + 22016 * 4096 + 4096 * 11008  # qkv | o | gate,up | down


def decode_probe(model, dev, weights="bf16", prompt_tokens=60, new_tokens=256, seed=322):
    """BASELINE configs[4] / SURVEY §8d config 5: one 224x224 image, a 60-token prompt (S0 = 203 positions), greedy generate through the
    captured hipGraph.  `value` = new tokens / wall time of the whole generate() call (ViT + projector + prefill included, as cli_qa.py
    runs it); the roofline leg isolates the per-token step as the difference of two run lengths."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, 32000, (1, prompt_tokens), generator=g)
    ids[0, 0], ids[0, 1] = 1, -200
    rgb = torch.randn(1, 3, 224, 224, generator=g).to(dev)
    kw = dict(images=rgb, do_sample=False, use_cache=True, weights=weights, eos_token_id=None)
    was_training = model.training
    model.eval()

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.generate(ids, max_new_tokens=n, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out.shape[1]

    run(4)  # graph capture, weight repacks, allocator warm-up
    n_short = max(8, new_tokens // 4)
    t_short, _ = run(n_short)
    t_full, n = run(new_tokens)
    if was_training:
        model.train()
    nl = len(model.text.p["layers"])
    esz = 1 if weights == "fp8" else 2
    S0 = prompt_tokens - 1 + 144
    per_tok = (t_full - t_short) / max(1, new_tokens - n_short)
    ctx_mid = S0 + (n_short + new_tokens) / 2.0
    w_bytes = (nl * LLAMA_LINEAR_PARAMS_PER_LAYER + 32000 * 4096) * esz          # every decoder linear + lm_head once per token
    kv_bytes = 2 * nl * ctx_mid * 4096 * 2                                        # K and V of the context so far, bf16
    ach = (w_bytes + kv_bytes) / per_tok / 1e9
    return {"metric": "cli_qa single-image greedy generate tokens/s (ViT + projector + prefill included)", "value": round(n / t_full, 1),
            "unit": "tokens/s", "new_tokens": n, "prompt_positions": S0, "weights": "e4m3 (fp8 MFMA weight stream)" if weights == "fp8" else "bf16",
            "llama_layers": nl, "ms_per_token_step": round(1e3 * per_tok, 4),
            "roofline": {"bound": "hbm", "kernel": "one captured hipGraph per token: 4 weight-streaming GEMVs + attention per layer, lm_head GEMV",
                         "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_token": int(w_bytes + kv_bytes),
                         "timing": f"wall clock (synchronised) of generate() at {n_short} and {new_tokens} new tokens; per-token = difference / {new_tokens - n_short}"}}


def timed_run(engine, batch, steps, warmup, world, lib, on_timed_start=None):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; the GEMM launches inside the timed region are timed live
    with HIP events on their launch stream (lhrs_gemm_profile_*), every step boundary carries an event (median step time)."""
    import ctypes
    from lhrs_bot_amd import _lib

    check = os.environ.get("LHRS_BENCH_CHECK_FINITE") == "1"   # debugging aid: synchronising finite checks after every phase of the step
    nstep = [0]

    def finite(what, *ts):
        for t in ts:
            if not bool(torch.isfinite(t).all()):
                raise SystemExit(f"rank {os.environ.get('RANK', '0')}: non-finite {what} at step {nstep[0]} ({int((~torch.isfinite(t)).sum())} of {t.numel()})")

    # LHRS_BENCH_IDLE_START_MS=x (soak runs, tools/soak_idle_queue.py): every step starts behind a DRAINED queue - device synchronise + x ms of sleep - the trigger
    # round 5 found for the shared-device NaN, forced here without the device -> host copy that used to cause it
    idle_ms = float(os.environ.get("LHRS_BENCH_IDLE_START_MS", "0") or 0)

    def idle_start():
        if idle_ms > 0:
            torch.cuda.synchronize()
            time.sleep(idle_ms * 1e-3)

    def step():
        idle_start()
        out = engine(batch)
        if check:
            finite("loss", out["total_loss"])
        engine.backward(out["total_loss"])
        if check:
            for r in getattr(engine, "reducers", {}).values():
                r.finish()
            finite("reduced gradient", *[st.grad for st in engine.stores])
        engine.step()
        if check:
            finite("master after the update", *[st.master for st in engine.stores])
            nstep[0] += 1
        return out["total_loss"]

    # LHRS_BENCH_TRACE_FINITE=1: a finite-ness flag of every phase of every step, computed ON THE DEVICE (one tiny reduction each, no host synchronisation, so
    # the timing of the step - what a start-up race depends on - is not disturbed) and read back once, behind the timed steps: which rank's which phase went
    # non-finite FIRST (loss -> d loss / d image -> each bucket's LOCAL gradient as it is handed to the collective -> the REDUCED gradient -> norm -> masters)
    # levels: 1 = loss / gradients / masters per step; 2 = + every stage of the forward, host <-> device integer copies, uploads; 3 = + every ViT layer; 4 = + each
    # of the four operators in front of the first ViT layer.  (Levels 1-3 still reproduce the shared-device NaN of DESIGN.md §6; level 4 does not: the extra
    # launches between those operators hide it.)
    tlevel = int(os.environ.get("LHRS_BENCH_TRACE_FINITE", "0") or 0)
    trace = engine._finite_trace = [] if tlevel >= 1 else None
    if trace is not None:
        def mark(label, *ts):
            trace.append((f"step {nstep[0]}: {label}", torch.stack([(~torch.isfinite(t)).any() for t in ts]).any()))
        text_bwd = engine.model.text.backward
        def traced_text_backward(*a, **k):
            d_image = text_bwd(*a, **k)
            if d_image is not None:
                mark("d loss / d image (LLaMA backward output)", d_image)
            return d_image
        engine.model.text.backward = traced_text_backward
        for name, r in getattr(engine, "reducers", {}).items():
            def wrap(r=r, name=name):
                ready0, finish0 = r.ready, r.finish
                def ready(key):
                    if key not in r.skip:
                        s_, e_ = r.buckets[key]
                        mark(f"LOCAL gradient of {name} bucket {key} [{s_}, {e_}) handed to the collective", r.flat[s_:e_])
                    ready0(key)
                def finish():
                    finish0()
                    mark(f"REDUCED gradient of {name}", r.flat)
                r.ready, r.finish = ready, finish
            wrap()
        # inside the forward (the first failures all named ONE rank's loss at step 1): every stage of it, the small integer tensors the step moves between host
        # and device (the batch is the same every step: they must never change), and every pinned-staging upload compared with its host source afterwards
        from lhrs_bot_amd import kernels as hk_
        model_, text_ = engine.model, engine.model.text
        seen_ints, h2d_log = {}, []
        def same_as_first(label, t):
            key = label
            if key not in seen_ints:
                seen_ints[key] = t.clone()
            elif seen_ints[key].shape != t.shape:
                trace.append((f"step {nstep[0]}: {label} CHANGED SHAPE {tuple(seen_ints[key].shape)} -> {tuple(t.shape)}", torch.ones((), dtype=torch.bool, device=batch["rgb"].device)))
            else:
                trace.append((f"step {nstep[0]}: {label} differs from step 0 (same batch every step)", (seen_ints[key].to(t.device) != t).any().to(batch["rgb"].device)))
        ints0, enc0, pool0, fh0, splice0, ce0, h2d0 = text_._ints_to_host, model_.rgb.encode, model_.rgb_pooler.forward, text_.forward_hidden, hk_.splice_fwd, hk_.cross_entropy, hk_.h2d
        def ints_to_host(*ts):
            out = ints0(*ts)
            for i, o in enumerate(out):
                if o is not None:
                    same_as_first(f"host copy #{i} of the batch's integer tensors (_ints_to_host)", o)
            return out
        head = [False]
        pf0, gn0, va0, ln0 = hk_.patchify, hk_.gemm_nt, hk_.vit_assemble, hk_.layernorm_fwd
        def patchify(rgb_, *a, **k):
            if head[0]:
                mark("ViT head: pixel input rgb", rgb_); same_as_first("ViT head: pixel input rgb", rgb_)
            y = pf0(rgb_, *a, **k)
            if head[0]:
                mark("ViT head: patchify output (im2col rows)", y)
            return y
        def gemm_nt(a_, b_, *a, **k):
            if head[0]:
                same_as_first("ViT head: patch embedding weight", b_); mark("ViT head: patchify output as the GEMM reads it", a_)
            y = gn0(a_, b_, *a, **k)
            if head[0]:
                mark("ViT head: patch-embedding GEMM output", y)
            return y
        def vit_assemble(patch, cls, pos, *a, **k):
            if head[0]:
                same_as_first("ViT head: class embedding", cls); same_as_first("ViT head: position embedding", pos)
            y = va0(patch, cls, pos, *a, **k)
            if head[0]:
                mark("ViT head: assembled tokens (cls + patches + pos)", y)
            return y
        def layernorm_fwd(x_, g_, b_, *a, **k):
            y = ln0(x_, g_, b_, *a, **k)
            if head[0]:
                same_as_first("ViT head: pre-LN weight", g_); mark("ViT head: pre-LN output", y)
            return y
        if tlevel >= 4:
            hk_.patchify, hk_.gemm_nt, hk_.vit_assemble, hk_.layernorm_fwd = patchify, gemm_nt, vit_assemble, layernorm_fwd
        def encode(x):
            head[0] = True
            y = enc0(x); mark("ViT taps", y); return y
        def pool_forward(x, *a, **k):
            y = pool0(x, *a, **k); mark("projector output (image embedding)", y); return y
        def splice_fwd(*a, **k):
            out = splice0(*a, **k); mark("spliced input embeddings", out[0]); return out
        def forward_hidden(*a, **k):
            y = fh0(*a, **k); mark("final-norm hidden state (LLaMA forward output)", y); return y
        def cross_entropy(logits, target, *a, **k):
            mark("logits", logits); same_as_first("device targets of the loss", target)
            return ce0(logits, target, *a, **k)
        def h2d(t, device):
            d = h2d0(t, device)
            if not t.is_cuda and len(h2d_log) < 4000:
                h2d_log.append((f"step {nstep[0]}: upload #{len(h2d_log)} {tuple(t.shape)} {t.dtype}", t, d))
            return d
        vit0 = hk_.vit_layer_forward
        vit_calls = [0]
        def vit_layer_forward(x, L, desc, B_, n_, LT, H, ff, h, qkv, o, f):
            li = vit_calls[0] % len(model_.rgb.p["layers"])
            head[0] = False
            if li == 0:   # the scratch buffers as the allocator handed them out (uninitialised): do they hold NaN bit patterns?  (informational: label says so)
                trace.append((f"step {nstep[0]}: [info] recycled scratch of the ViT holds non-finite bit patterns before its first write",
                              torch.stack([(~torch.isfinite(t)).any() for t in (h, qkv, o, f)]).any()))
                mark("ViT input x of layer 0 (patch embedding + pre-LN)", x)
            y = vit0(x, L, desc, B_, n_, LT, H, ff, h, qkv, o, f)
            for nm, t in (("LN output h", h), ("qkv", qkv), ("attention output o", o), ("fc1 output f", f), ("residual stream x", x)):
                mark(f"ViT layer {li}: {nm}", t)
            vit_calls[0] += 1
            return y
        if tlevel >= 3:
            hk_.vit_layer_forward = vit_layer_forward
        if tlevel >= 2:
            text_._ints_to_host, model_.rgb.encode, model_.rgb_pooler.forward, text_.forward_hidden = ints_to_host, encode, pool_forward, forward_hidden
            hk_.splice_fwd, hk_.cross_entropy, hk_.h2d = splice_fwd, cross_entropy, h2d
        engine._h2d_log = h2d_log
        step0 = step
        def step():
            idle_start()
            out = engine(batch)
            mark("loss", out["total_loss"])
            engine.backward(out["total_loss"])
            engine.step()
            mark("gradient norm^2", engine.gnorm_sq)
            mark("masters after the update", *[st.master for st in engine.stores])
            nstep[0] += 1
            return out["total_loss"]
    loss = None
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    for r in getattr(engine, "reducers", {}).values():   # N > 1: time the compute stream spends waiting for the gradient collectives
        r.measure = True
        r.blocked_ms()
    lib.lhrs_gemm_profile_stride(PROFILE_STRIDE)
    _lib.check(lib.lhrs_gemm_profile_enable(int(os.environ.get("LHRS_GEMM_PROFILE_SAMPLES", "16000"))), "gemm_profile_enable")   # 0: A/B of the event overhead
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if on_timed_start is not None:
        on_timed_start()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss = step()
        marks[i + 1].record()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 5)()
    _lib.check(lib.lhrs_gemm_profile_read(ctypes.addressof(prof)), "gemm_profile_read")
    kinds = (ctypes.c_double * 30)()
    _lib.check(lib.lhrs_gemm_profile_read_kinds(ctypes.addressof(kinds)), "gemm_profile_read_kinds")
    lib.lhrs_gemm_profile_enable(0)
    blocked = 0.0
    for r in getattr(engine, "reducers", {}).values():
        blocked += r.blocked_ms()[1]
        r.measure = False
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    return dict(dt=dt, loss=loss, prof=list(prof), kinds=list(kinds), median_ms=median, blocked_ms_per_step=blocked / max(1, steps))


# every 7th launch of each GEMM variant is bracketed by HIP events: timing every launch costs 1.0 % (micro-batch 30) / 2.2 % (micro-batch 8) of
# the step in event-record idle time; 7 is coprime to the launch sequence's periods (2 forward, 3 backward plain launches per layer)
PROFILE_STRIDE = int(os.environ.get("LHRS_GEMM_PROFILE_STRIDE", "7"))

GEMM_KERNEL_DESC = ("gemm_nt_256s_kernel<ACT, 0, K2P> (256x256 tile, 16 waves): BK=64 double-buffered LDS stages via global_load_lds DMA, "
                    "v_mfma_f32_16x16x32_bf16, persistent over tiles; its launches with a fused SwiGLU / RoPE epilogue and the plain launches of the 144-row "
                    "kernel (ViT / projector products) are timed separately under `variants`")
GEMM_U4_DESC = ("gemm_u4_kernel (csrc/gemm_u4.hip; hand-written): 256x256x64 tile, FOUR waves of 128x128, accumulators in named AGPRs, two 64 KiB LDS stages via paced "
                "global_load_lds DMA (one piece per 6 MFMAs), v_mfma_f32_16x16x32_bf16, a workgroup's tiles walked as ONE stream of stages with each finished tile written "
                "out inside the next tile's first stage - every PLAIN long-k product whose tiles fill the chip (a shape rule in lhrs_gemm_bf16_nt: no timing, no vendor "
                "library); its instantiation with / without a residual is named in `kernel_instantiation`, the fused-epilogue instantiations are timed separately under `variants`")
GEMM_144_DESC = ("gemm_nt_144s_kernel<ACT, 0> (144x256 tile, 12 waves, three 50 KiB LDS stages): the plain-epilogue kernel that carries the most time at this "
                 "micro-batch; the 256-row kernel's variants are listed under `variants`")


def plain_products_note(lib):
    return ("hand-written only: a plain long-k product (no bias / activation, bf16 out, K >= 4096, M and N >= 1024) whose 256x256 tiles fill >= 80 % of one round "
            "of the CUs runs the four-wave gemm_u4_kernel, every other product the 16-wave / 144-row / small-tile kernels of csrc/gemm.hip - a pure shape rule "
            "(lhrs_gemm_u4_takes): no first-call timing, no vendor library in the process path, bit-reproducible run to run and rank to rank")


KIND_NAMES = ("gemm_nt_256s_kernel<ACT, 0, K2P> plain (16 waves, 256-row tiles)", "gemm_nt_256s_kernel<0, 1> SwiGLU-fwd epilogue", "gemm_nt_256s_kernel<0, 2> SwiGLU-bwd epilogue",
              "gemm_nt_256s_kernel<0, 3> RoPE epilogue", "gemm_nt_144s_kernel<ACT, 0> plain (12 waves, 144-row tiles)", "gemm_u4_kernel<0, true> plain + residual (four waves)",
              "gemm_u4_kernel<0, false> plain (four waves)", "gemm_u4_kernel<1, false> SwiGLU-fwd epilogue (four waves)", "gemm_u4_kernel<2, false> SwiGLU-bwd epilogue (four waves)",
              "gemm_u4_kernel<3, false> RoPE epilogue (four waves)")
KIND_DESC = {0: GEMM_KERNEL_DESC, 4: GEMM_144_DESC, 5: GEMM_U4_DESC, 6: GEMM_U4_DESC}


def gemm_traffic(dom, B, scale_layers):
    """HBM-side bytes per launch of the dominant kernel: a PMC pass cannot run inside this process (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE are separate profiled runs of
    this same command); the committed summary of that pass on this tree is quoted, with its provenance, when it names the same kernel at the same micro-batch."""
    for name in ("r06_gemm_traffic_b240.json", "r06_gemm_traffic_b120.json", "r06_gemm_traffic.json", "r05_gemm_traffic.json"):   # the committed pass at THIS micro-batch, newest first (the kernel is unchanged since round 5)
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if B == tj.get("micro_batch", 30) and scale_layers == 1.0 and KIND_NAMES[dom].startswith(tj["kernel_prefix"]):
                return int(tj["traffic_bytes_per_launch"]), tj["bench_note"].replace("r05_gemm_traffic.json", name)
        except Exception:  # noqa: BLE001
            pass
    return None, "not measured in this run (PMC passes are separate rocprofv3 runs; no committed pass for this kernel at this micro-batch)"


def roofline_block(prof, kinds, steps, B, S, scale_layers, sclk=None, watts=None):
    """`achieved` is ONE kernel's figure: among the PLAIN-epilogue kernel instantiations (kinds 0, 4, 5, 6: the names a rocprofv3 kernel trace lists) the one that carries
    the most time, so that its `avg_launch_us` can be held against that kernel's average duration in profiles/*_kernel_stats.csv; every instantiation is listed under `variants`."""
    dom = max((0, 4, 5, 6), key=lambda k: kinds[3 * k + 1])
    n_samp, ms, fl = kinds[3 * dom], kinds[3 * dom + 1], kinds[3 * dom + 2]
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    variants = {}
    for k, nm in enumerate(KIND_NAMES):
        n_k, ms_k, fl_k = kinds[3 * k], kinds[3 * k + 1], kinds[3 * k + 2]
        if n_k > 0 and ms_k > 0:
            tf = fl_k / (ms_k * 1e-3) / 1e12
            variants[nm] = {"launches": int(n_k), "avg_launch_us": round(1e3 * ms_k / n_k, 2), "achieved_tflops": round(tf, 1),
                            "frac": round(tf / PEAK_BF16_TFLOPS, 4)}
    all_ms = sum(kinds[3 * k + 1] for k in range(10))
    all_fl = sum(kinds[3 * k + 2] for k in range(10))
    u4_ms = sum(kinds[3 * k + 1] for k in range(5, 10))
    traffic, traffic_note = gemm_traffic(dom, B, scale_layers)
    return {"bound": "mfma", "kernel": KIND_DESC[dom], "kernel_instantiation": KIND_NAMES[dom], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "traffic_note": traffic_note,
            "variants": variants, "all_variants_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 1) if all_ms > 0 else None,
            "hand_written_kernels_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 1) if all_ms > 0 else None,   # every GEMM launch is one of this library's kernels
            "hand_written_share_of_gemm_time": 1.0 if all_ms > 0 else None,                                      # (round 4 had a vendor-library candidate; gone)
            "four_wave_kernel_share_of_gemm_time": round(u4_ms / all_ms, 3) if all_ms > 0 else None,
            "launches_timed": int(n_samp), "timed_every_nth_launch": PROFILE_STRIDE, "avg_launch_us": round(1e3 * ms / max(n_samp, 1), 2),
            "sclk_mhz_during_timed_region": round(sclk) if sclk else None, "package_power_w": round(watts) if watts else None,
            "frac_of_peak_at_measured_clock": round(ach / (PEAK_BF16_TFLOPS * sclk / 2400.0), 4) if sclk else None,
            "gemm_flops_share_of_step": round(prof[4] / steps / (B * f_alg(S)), 3) if scale_layers == 1.0 else None}


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
    ap.add_argument("--micro-batch", type=int, default=240,
                    help="samples per GPU per step.  240 since the end of round 6: 195 GB of saved activations, 245 GB of the 288 GB allocated at the peak "
                         "(config.peak_hbm_allocated_gb) - the memory is there to be used, and every doubling halves the launch boundaries per sample and keeps the tile walks and "
                         "the attention kernels' workgroups on whole rounds of the 256 CUs (M = 65520 = 256 tile rows; 7680 workgroups = 30 rounds): same-box 60 -> 120 +1.2 %, "
                         "120 -> 240 +0.8 %.  Rounds 1-4 quoted 30, rounds 5-6 60 - reported next to the headline as config.micro_batch_120 / _60 / _30; the reference script uses 8 "
                         "on 80 GB parts (Script/train_stage1.sh:11), reported as micro_batch_8 (DESIGN.md §4)")
    ap.add_argument("--caption-tokens", type=int, default=128)
    ap.add_argument("--llama-layers", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the untimed extras (micro-batch-8 step rate, generate tokens/s) after the headline run")
    ap.add_argument("--decode", action="store_true",
                    help="make BASELINE configs[4] the line: cli_qa-shaped greedy generate (1 image, 60-token prompt), tokens/s with an HBM roofline")
    ap.add_argument("--new-tokens", type=int, default=256, help="--decode: tokens generated per step")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="--decode: weight stream (fp8 = e4m3 MFMA weights)")
    ap.add_argument("--gemm-policy", type=int, default=None, help="kernel A/B tests only: lhrs_gemm_set_policy value")
    ap.add_argument("--stage", type=int, default=1, choices=[1, 2, 3],
                    help="1: projector-only + Adan (the headline metric, BASELINE configs[1..2]); 3: LoRA r=8 on q,k,v,o + AdamW, projector "
                         "frozen (BASELINE configs[3]); 2: LoRA r=128 on every linear + projector, AdamW (Config/multi_modal_stage2.yaml)")
    ap.add_argument("--comm-dtype", default="float32", choices=["float32", "bfloat16"],
                    help="dtype of the gradient all-reduce (N > 1): bfloat16 = 160 MB per step at stage 1, what DeepSpeed bf16 ZeRO-2 moves (SURVEY §8e)")
    ap.add_argument("--base8", default="int8", choices=["int8", "e4m3"], help="--bits 8: scheme of the 8-bit frozen base (text.py::quantize_base)")
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
    if os.environ.get("LHRS_GEMM_PERSISTENT"):  # kernel A/B tests only
        lib.lhrs_gemm_set_persistent(int(os.environ["LHRS_GEMM_PERSISTENT"]))
    if os.environ.get("LHRS_GEMM_MIN_TILES"):  # kernel A/B tests only
        lib.lhrs_gemm_set_min_tiles(int(os.environ["LHRS_GEMM_MIN_TILES"]))
    B, T = a.micro_batch, a.caption_tokens + 2
    S = T - 1 + 144
    model = UniBind(("rgb", "text"), None, device=dev, llama_layers=a.llama_layers).init_random(seed=0)  # same weights on every rank
    if a.decode:
        if world > 1:
            raise SystemExit("--decode is a single-sequence latency run (BASELINE configs[4]: 1xMI355X): replicas only, run it with --gpus 1")
        res = None
        for _ in range(max(1, a.steps)):  # a "step" = one whole generate() call; the best of them is reported
            r = decode_probe(model, dev, a.weights, new_tokens=a.new_tokens)
            res = r if res is None or r["value"] > res["value"] else res
        res.update({"n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * res["new_tokens"] / res["value"], 3),
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.weights == "bf16" else "e4m3 weights, bf16 activations",
                    "data": "synthetic", "config": {"workload": "BASELINE configs[4]: cli_qa.py single-image VQA generate, hipGraph-captured decode, "
                                                                f"LLaMA2-7B ({a.llama_layers} layers) random-init, 60-token prompt + 144 image tokens, greedy",
                                                    "batch": 1, "parallelism": "dp1"}})
        print(json.dumps(res), flush=True)
        return
    if a.stage == 1:
        model.prepare_for_training()
        engine = LHRSEngine(model, optimizer="adanp", lr=2e-4, weight_decay=0.0, max_grad_norm=0.3, comm_dtype=getattr(torch, a.comm_dtype))
    else:
        if a.stage == 3:
            model.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"))
        else:
            model.enable_lora(r=128, alpha=256, dropout=0.05)  # Config/multi_modal_stage2.yaml:81-86 (train mode: dropout active)
        if a.bits == 8:
            model.text.quantize_base(8, a.base8)   # "int8": the reference's LLM.int8 arithmetic (what the YAML key selects); "e4m3": the faster MI355X-native base
        model.prepare_for_training(freeze_text=False, tune_rgb_pooler=a.stage == 2)
        engine = LHRSEngine(model, optimizer="adamw", lr=1e-4 if a.stage == 3 else 2e-4, weight_decay=0.0, max_grad_norm=1.0,
                            comm_dtype=getattr(torch, a.comm_dtype))
    batch = make_batch(B, T, dev, seed=322 + rank)  # reference seed convention (main_pretrain_stage1.py:281-287)

    smi = _SmiSampler() if rank == 0 and os.environ.get("LHRS_BENCH_NO_SMI") != "1" else None   # LHRS_BENCH_NO_SMI=1: A/B of the sampler's own cost
    run = timed_run(engine, batch, a.steps, a.warmup, world, lib, on_timed_start=smi.start if smi is not None else None)
    sclk, watts = smi.stop() if smi is not None else (None, None)
    dt, loss, prof, kinds = run["dt"], run["loss"], run["prof"], run["kinds"]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9   # of the headline run (model + saved activations of one step), before the extras
    dp = None
    if getattr(engine, "_finite_trace", None):
        for lab, host_t, dev_t in getattr(engine, "_h2d_log", []):      # every pinned-staging upload of the run against its host source, now that everything has run
            engine._finite_trace.append((f"{lab}: device copy differs from its host source", (host_t.to(dev_t.device) != dev_t).any()))
        flags = torch.stack([f for _, f in engine._finite_trace]).cpu().tolist()
        info = [lab for (lab, _), f in zip(engine._finite_trace, flags) if f and "[info]" in lab]
        bad = [lab for (lab, _), f in zip(engine._finite_trace, flags) if f and "[info]" not in lab]
        print(f"rank {rank} finite-trace: {len(flags)} phases checked, " + (f"FIRST non-finite: {bad[0]!r}; all non-finite: {bad}" if bad else "all finite") + (f"; info: {info[:6]}" if info else ""),
              file=sys.stderr, flush=True)
    if world > 1:
        # self-validating first contact of the N > 1 path (main_pretrain_stage1.py:54-60): after K optimizer steps on DIFFERENT
        # per-rank batches every rank must hold the same trainable masters (same reduced gradients, same update) - compare an fp64 checksum
        # and the sum of |x| of every trainable store across ranks; also gather what each rank's compute stream waited for its collectives
        sums = torch.stack([torch.stack([st.master.double().sum(), st.master.double().abs().sum()]) for st in engine.stores]).flatten()
        mine = torch.cat([sums, torch.tensor([run["blocked_ms_per_step"], 1e3 * run["dt"] / a.steps], device=dev, dtype=torch.float64)])
        allv = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allv, mine)
        allv = torch.stack(allv).cpu()
        n_ck = sums.numel()
        same = bool((allv[:, :n_ck] == allv[0:1, :n_ck]).all())
        dp_error = None
        if not same:   # the line is still printed (with the per-rank checksums and wait times: the record alone must say what happened), then the run fails
            diag = {"loss": final_loss, "gnorm_sq": float(engine.gnorm_sq)}
            for st in engine.stores:
                diag[st.name] = {"master_nonfinite": int((~torch.isfinite(st.master)).sum()), "grad_nonfinite": int((~torch.isfinite(st.grad)).sum()),
                                 **{k: int((~torch.isfinite(v)).sum()) for k, v in engine.state[st.name].items()}}
            print(f"rank {rank} diagnostics: {diag}", file=sys.stderr, flush=True)
            dp_error = f"trainable masters differ across the {world} ranks after {a.steps} steps"
        dp = {"rccl_ranks": torch.distributed.get_world_size(), "backend": torch.distributed.get_backend(),
              "reduce_mode": next(iter(engine.reducers.values())).mode if engine.reducers else None,
              "collectives_per_step": sum(len(r.buckets) for r in engine.reducers.values()),
              "replica_checksum_equal_on_all_ranks": same,
              "master_checksum": [float(x) for x in allv[0, :n_ck]] if same else None,
              "master_checksum_per_rank": None if same else [[float(x) for x in row[:n_ck]] for row in allv],
              "ms_per_step_blocked_on_allreduce_per_rank": [round(float(x), 3) for x in allv[:, n_ck]],
              "ms_per_step_per_rank": [round(float(x), 3) for x in allv[:, n_ck + 1]], "error": dp_error}

    if rank == 0:
        sps = world * B * a.steps / dt
        scale_layers = a.llama_layers / 32.0
        # FLOPs the step EXECUTES: F_alg counts lm_head forward + backward on all S positions; the engine runs it on the
        # rows that have a target only (caption tokens), so the executed share is lower
        tgt_rows = a.caption_tokens
        f_exec = f_alg(S) - 2 * (S - tgt_rows) * 2 * 4096 * 32000
        tail_only = os.environ.get("LHRS_TAIL_ROWS_ONLY", "1") != "0"
        if tail_only:  # last decoder layer: o_proj, gate|up, down (forward and dX) on the supervised rows only
            f_exec -= 2 * (S - tgt_rows) * 2 * (4096 * 4096 + 4096 * 22016 + 11008 * 4096)
        res = {
            "metric": ("stage-1 pretrain samples/sec (224^2 image + 128-tok caption)" if a.stage == 1 else
                       f"stage-{a.stage} LoRA train samples/sec (224^2 image + 128-tok sequence)"), "value": round(sps, 3), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "ms_per_step_median": round(run["median_ms"], 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.bits == 16 or a.stage == 1 else ("int8 base weights (LLM.int8) + bf16" if a.base8 == "int8" else "e4m3 base weights + bf16"), "data": "synthetic",
            "data_note": "ONE synthetic batch per rank (seed 322 + rank), resident in HBM before the timed region and reused by every step: no per-step image host-to-device copy is in the timed region (the integer tensors stay in host memory, as the reference's collator delivers them)",
            "config": {"workload": (("BASELINE configs[1]: stage-1 projector-only" if world == 1 else "BASELINE configs[2]: stage-1 projector-only, DDP") if a.stage == 1
                                    else ("BASELINE configs[3]: stage-3 SFT, LoRA r=8 on q,k,v,o" if a.stage == 3 else "stage-2 (Config/multi_modal_stage2.yaml): LoRA r=128 on all linears + projector"))
                                   + f", CLIP ViT-L/14@224 + AttnPooler + LLaMA2-7B ({a.llama_layers} layers), S={S}, random-init weights",
                       "micro_batch_per_gpu": B, "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}",
                       "optimizer": "adanp" if a.stage == 1 else "adamw", "stage": a.stage,
                       "lora": None if a.stage == 1 else ("r=8 on q,k,v,o, lora_dropout 0" if a.stage == 3 else "r=128 on all 7 linears, lora_dropout 0.05 (train mode)"),
                       "last_layer_rows": "supervised positions only (same loss and gradients; LHRS_TAIL_ROWS_ONLY=0 computes all)" if os.environ.get("LHRS_TAIL_ROWS_ONLY", "1") != "0" else "all",
                       "peak_hbm_allocated_gb": round(peak_gb, 1),
                       "grad_allreduce": a.comm_dtype if world > 1 else "none",
                       "dist_backend": (torch.distributed.get_backend() if world > 1 else None), "data_parallel": dp,
                       "plain_long_k_products": plain_products_note(lib)},
            "loss": round(final_loss, 4),
            "step_mfma_frac": round(sps / world * f_alg(S) / (PEAK_BF16_TFLOPS * 1e12), 4) if scale_layers == 1.0 and a.stage == 1 else None,
            "step_mfma_frac_executed": round(sps / world * f_exec / (PEAK_BF16_TFLOPS * 1e12), 4) if scale_layers == 1.0 and a.stage == 1 else None,
            "roofline": roofline_block(prof, kinds, a.steps, B, S, scale_layers, sclk, watts),
        }
        res["config"]["step_mfma_frac_executed"] = res["step_mfma_frac_executed"]   # hardware utilisation of the whole step (F_alg minus the rows the engine skips)
        if world == 1 and not a.no_extra and a.stage == 1:
            extra = {}
            try:  # the reference script's micro-batch (Script/train_stage1.sh:11), same engine, its own timed region
                if B != 8:
                    torch.cuda.empty_cache()
                    r8 = timed_run(engine, make_batch(8, T, dev, seed=322), 12, 3, 1, lib)
                    sps8 = 8 * 12 / r8["dt"]
                    res["micro_batch_8"] = {
                        "value": round(sps8, 2), "unit": "samples/s", "ms_per_step": round(1e3 * r8["dt"] / 12, 3), "ms_per_step_median": round(r8["median_ms"], 3),
                        "steps": 12, "warmup": 3, "micro_batch_per_gpu": 8, "loss": round(float(r8["loss"].item()), 4),
                        "step_mfma_frac": round(sps8 * f_alg(S) / (PEAK_BF16_TFLOPS * 1e12), 4) if scale_layers == 1.0 else None,
                        "roofline": roofline_block(r8["prof"], r8["kinds"], 12, 8, S, scale_layers),
                        "note": "the reference script's per-GPU batch (an 80 GB-GPU constraint): M = 2184 rows"}
                    res["config"]["micro_batch_8"] = {"value": round(sps8, 2), "unit": "samples/s", "ms_per_step": round(1e3 * r8["dt"] / 12, 3),
                                                      "roofline_frac": res["micro_batch_8"]["roofline"]["frac"],
                                                      "step_mfma_frac": res["micro_batch_8"]["step_mfma_frac"]}
                for bb in (120, 60, 30):   # the micro-batches earlier lines quoted the headline on (rounds 1-4: 30, rounds 5-6: 60, end of round 6: 120): continuity of the series
                    if B == bb or bb > B:
                        continue
                    torch.cuda.empty_cache()   # the headline step's buffers go back to the device before a smaller shape asks for its own
                    rbb = timed_run(engine, make_batch(bb, T, dev, seed=322), 8, 2, 1, lib)
                    spsbb = bb * 8 / rbb["dt"]
                    rfb = roofline_block(rbb["prof"], rbb["kinds"], 8, bb, S, scale_layers)
                    res["config"][f"micro_batch_{bb}"] = {"value": round(spsbb, 2), "unit": "samples/s", "ms_per_step": round(1e3 * rbb["dt"] / 8, 3),
                                                          "roofline_frac": rfb["frac"], "kernel_instantiation": rfb["kernel_instantiation"],
                                                          "step_mfma_frac": round(spsbb * f_alg(S) / (PEAK_BF16_TFLOPS * 1e12), 4) if scale_layers == 1.0 else None,
                                                          "variants": {k: v["frac"] for k, v in rfb["variants"].items()}}
                del engine
                torch.cuda.empty_cache()
                # one image, ~60-token prompt, 512 new tokens, greedy
                extra["generate_bf16"] = decode_probe(model, dev, "bf16", new_tokens=512)
                extra["generate_fp8"] = decode_probe(model, dev, "fp8", new_tokens=512)
                res["config"]["generate_512_new_tokens"] = {k: {"tokens_per_s": extra[k]["value"], "ms_per_token_step": extra[k]["ms_per_token_step"],
                                                                "hbm_roofline_frac": extra[k]["roofline"]["frac"]} for k in ("generate_bf16", "generate_fp8")}
            except Exception as e:  # extras must never take the headline number down
                extra["error"] = f"{type(e).__name__}: {e}"
            try:   # BASELINE configs[3], per-GPU part: LoRA r = 8 on q,k,v,o + AdamW, projector frozen, micro-batch 32 (global batch 256 over 8 GPUs) - LAST: it changes the model
                torch.cuda.empty_cache()
                model.train()
                model.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"))
                model.prepare_for_training(freeze_text=False, tune_rgb_pooler=False)
                eng3 = LHRSEngine(model, optimizer="adamw", lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
                r3 = timed_run(eng3, make_batch(32, T, dev, seed=322), 8, 2, 1, lib)
                sps3 = 32 * 8 / r3["dt"]
                rb3 = roofline_block(r3["prof"], r3["kinds"], 8, 32, S, scale_layers)
                extra["stage3_lora_r8_b32"] = {
                    "metric": "stage-3 LoRA train samples/sec (BASELINE configs[3] per-GPU part: LoRA r=8 on q,k,v,o, AdamW, micro-batch 32 = global batch 256 / 8)",
                    "value": round(sps3, 2), "unit": "samples/s", "ms_per_step": round(1e3 * r3["dt"] / 8, 3), "ms_per_step_median": round(r3["median_ms"], 3),
                    "steps": 8, "warmup": 2, "micro_batch_per_gpu": 32, "loss": round(float(r3["loss"].item()), 4),
                    "dominant_kernel": rb3["kernel_instantiation"], "dominant_kernel_frac": rb3["frac"], "dominant_kernel_avg_launch_us": rb3["avg_launch_us"],
                    "four_wave_kernel_share_of_gemm_time": rb3["four_wave_kernel_share_of_gemm_time"],
                    "lora_pair": "A2.B2^T rides as K2/64 extra stages of the q|k|v (+RoPE) and o (+residual) products' k-loop (gemm_u4_kernel<3,false> / <0,true> / <0,false>): no separate LoRA launch on the forward or dX path",
                    "variants": {k: {"frac": v["frac"], "avg_launch_us": v["avg_launch_us"], "launches": v["launches"]} for k, v in rb3["variants"].items()}}
                res["config"]["stage3_lora_r8_b32"] = {"value": round(sps3, 2), "unit": "samples/s", "ms_per_step": round(1e3 * r3["dt"] / 8, 3),
                                                       "dominant_kernel": rb3["kernel_instantiation"], "dominant_kernel_frac": rb3["frac"]}
                del eng3
            except Exception as e:  # noqa: BLE001
                extra["stage3_error"] = f"{type(e).__name__}: {e}"
            res["extra"] = extra
        if world == 1 and not a.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(S, a.llama_layers)
            except Exception as e:  # the checker must never take the product number down with it
                res["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        if dp is not None and dp.get("error"):
            res["valid"] = False
            res["error"] = dp["error"]
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
        if dp is not None and dp.get("error"):
            raise SystemExit(f"rank {rank}: {dp['error']} (the JSON line above carries the per-rank checksums)")


if __name__ == "__main__":
    main()
