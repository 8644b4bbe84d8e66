"""Training engine: the small surface the entry scripts use from the DeepSpeed engine, on HIP streams + RCCL.

Replaces, for the stage-1 hot path, what /root/reference reaches through `deepspeed.initialize(...)`
(main_pretrain_stage1.py:215-220) and `DeepSpeedHook.after_iter` (lhrs/CustomTrainer/hook/deepspeed_hook.py:4-19):
    loss_dict = engine(batch); engine.backward(loss); engine.step()
plus `optimizer.param_groups` / `optimizer._global_grad_norm` used by the LR hook and the logger.

Data parallelism: every rank holds the frozen ViT + LLaMA and the 80 M trainable projector
parameters.  The projector's fp32 gradients live in ONE flat buffer ordered [query | layer0..5 | out_proj];
backward finishes them in the order out_proj, layer5..0, query, and each finished range is all-reduced (RCCL over
xGMI, via torch.distributed backend "nccl") from a dedicated HIP stream while the remaining pooler backward runs.
ZeRO sharding is dropped on purpose: 80 M fp32 x 6 states = 1.9 GB per GPU of 288 GB.
Averaging is folded into the optimizer kernel (grad_scale = 1/world); the global-norm clip (DeepSpeed
`gradient_clipping`, main_pretrain_stage1.py:28-85) uses a device-resident squared norm: no host sync per step.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import kernels as hk


def cosine_warmup_lr(it: int, base_lr: float, max_iters: int, min_lr: float = 0.0, warmup_iters: int = 0,
                     warmup_ratio: float = 0.1, warmup: Optional[str] = "linear") -> float:
    """CosineAnnealingLrUpdaterHook(by_epoch=False) + linear warm-up of the reference
    (lhrs/CustomTrainer/hook/lr_scheduler_hook.py:80-145, 243-271, 690-705): pure host math."""
    regular = min_lr + 0.5 * (base_lr - min_lr) * (math.cos(math.pi * it / max_iters) + 1)
    if warmup is None or it >= warmup_iters:
        return regular
    if warmup == "constant":
        return regular * warmup_ratio
    if warmup == "linear":
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        return regular * (1 - k)
    if warmup == "exp":
        return regular * warmup_ratio ** (1 - it / warmup_iters)
    raise ValueError(warmup)


class _Optimizer:
    """`optimizer.param_groups[*]["lr"]` / `_global_grad_norm` surface of the reference's optimizer object."""

    def __init__(self, lr, wd, decay_numel, nodecay_numel):
        # build_optimizer.py:18-38 - decay group and no-decay group (1-D tensors and biases)
        self.param_groups: List[Dict] = [dict(lr=lr, initial_lr=lr, weight_decay=wd, numel=decay_numel),
                                         dict(lr=lr, initial_lr=lr, weight_decay=0.0, numel=nodecay_numel)]
        self._global_grad_norm = None


def bucket_ranges(pool):
    """Flat [key, start, end) ranges in the order AttnPooler.backward completes them: out_proj, layers nl-1..0, query."""
    first = {}
    for name, _ in pool.spec:
        key = name.split(".")[1] if name.startswith("layers.") else name.split(".")[0]
        first.setdefault(key, pool.offsets[name][0])
    keys = list(first.keys())  # query, 0..nl-1, out_proj (spec order)
    ends = {k: (first[keys[i + 1]] if i + 1 < len(keys) else pool.numel) for i, k in enumerate(keys)}
    order = ["out_proj"] + [str(l) for l in reversed(range(pool.nl))] + ["query"]
    return [(k, first[k], ends[k]) for k in order]


def merged_buckets(buckets, min_numel: int = 1 << 20):
    """Greedy fold of the signal-ordered buckets: a bucket joins the one signalled right before it while that one is still smaller than
    `min_numel` elements and the two are adjacent in the flat buffer; the fused range is issued at the LAST of its keys (the projector's
    147 k-element `query` would otherwise be a collective of its own - here layer 0 waits the few microseconds for it; r = 8 LoRA layers of
    262 k elements travel four at a time).  -> (buckets, skipped keys)."""
    out, skipped = [], set()
    for k, s, e in buckets:
        if out and (out[-1][2] - out[-1][1] < min_numel or e - s < min_numel // 4) and (out[-1][2] == s or out[-1][1] == e):
            pk, ps, pe = out[-1]
            skipped.add(pk)
            out[-1] = (k, min(ps, s), max(pe, e))
        else:
            out.append((k, s, e))
    return out, skipped


class GradReducer:
    """Bucketed sum all-reduce of ranges of ONE flat gradient buffer, launched as ranges become final.

    Device-agnostic on purpose: on MI355X the collectives are RCCL over xGMI (torch.distributed backend "nccl")
    issued from a dedicated HIP stream that waits on an event recorded by the compute stream; on CPU tensors
    (gloo, used by the world_size-2 tests) the same code runs without streams."""

    def __init__(self, flat_grad: torch.Tensor, buckets, process_group=None, comm_dtype=torch.float32, mode: str = "bucketed"):
        """mode "bucketed": one collective per bucket, issued when the backward signals it (small neighbours merged); "flat": ONE collective over
        the whole buffer when the LAST bucket is signalled (no overlap with the remaining backward, one launch - SURVEY §2.2 C1)."""
        self.flat, self.pg, self.comm_dtype, self.mode = flat_grad, process_group, comm_dtype, mode
        if mode == "flat":
            keys = [k for k, _, _ in buckets]
            self.buckets = {keys[-1]: (min(s for _, s, _ in buckets), max(e for _, _, e in buckets))}
            self.skip = set(keys[:-1])
        else:
            merged, self.skip = merged_buckets(list(buckets))
            self.buckets = {k: (s, e) for k, s, e in merged}
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat_grad.device) if self.cuda else None
        self.pending = []
        self.measure = False     # bench.py: bracket the compute stream's wait in finish() with events (blocked_ms())
        self._blocked = []

    def ready(self, key: str) -> None:
        if key in self.skip:
            return  # travels with a later bucket
        s, e = self.buckets[key]
        buf = self.flat[s:e]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._issue(buf)
        else:
            self._issue(buf)

    def _issue(self, buf):
        if self.comm_dtype == buf.dtype:
            self.pending.append((torch.distributed.all_reduce(buf, group=self.pg, async_op=True), None, buf))
        else:
            low = buf.to(self.comm_dtype)
            self.pending.append((torch.distributed.all_reduce(low, group=self.pg, async_op=True), low, buf))

    def finish(self) -> None:
        """Make the compute stream (or the host, on CPU) wait for every outstanding bucket."""
        if self.cuda:
            # Work.wait() orders the CURRENT stream behind the collective (RCCL runs it on its own stream): wait from the comm stream,
            # so that the widening copy below cannot read a bucket that is still being reduced, then chain the compute stream behind it
            with torch.cuda.stream(self.comm_stream):
                for w, low, buf in self.pending:
                    w.wait()
                    if low is not None:
                        buf.copy_(low)
            if self.measure:   # time the COMPUTE stream spends waiting for the collectives = what the overlap did not hide
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                torch.cuda.current_stream().wait_stream(self.comm_stream)
                e1.record(torch.cuda.current_stream())
                self._blocked.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            for w, low, buf in self.pending:
                w.wait()
                if low is not None:
                    buf.copy_(low)
        self.pending.clear()


    def blocked_ms(self, reset: bool = True):
        """(finish() calls measured, summed ms the compute stream waited in them); synchronises the recorded events."""
        n, ms = len(self._blocked), 0.0
        for e0, e1 in self._blocked:
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        if reset:
            self._blocked = []
        return n, ms


class _TrainStore:
    """Uniform view of one trainable flat parameter set (projector or LoRA adapters) for the optimizer and the reducer."""

    def __init__(self, name, master, grad, shadow, refresh, wd_ranges, buckets, numel_decay, numel_nodecay):
        self.name, self.master, self.grad, self.shadow, self.refresh = name, master, grad, shadow, refresh
        self.wd_ranges, self.buckets = wd_ranges, buckets
        self.numel = master.numel()
        self.numel_decay, self.numel_nodecay = numel_decay, numel_nodecay


def _pooler_wd_ranges(pool):
    out = []
    for name, shape in pool.spec:
        off, _ = pool.offsets[name]
        n = 1
        for s in shape:
            n *= s
        padded = (n + 63) // 64 * 64
        dec = not (len(shape) == 1 or name.endswith(".bias"))  # build_optimizer.py:41-73: 1-D tensors and biases do not decay
        if out and out[-1][2] == dec and out[-1][1] == off:
            out[-1][1] = off + padded
        else:
            out.append([off, off + padded, dec])
    return out


class LHRSEngine:
    def __init__(self, model, optimizer: str = "adanp", lr: float = 2e-4, weight_decay: float = 0.0,
                 max_grad_norm: float = 0.3, betas=None, eps: float = 1e-8, process_group=None, comm_dtype=torch.float32,
                 gradient_accumulation_steps: int = 1, broadcast_trainable: bool = True, reduce_mode: Optional[str] = None):
        self.module = self.model = model
        self.pool = model.rgb_pooler
        self.opt_name = optimizer.lower()
        if self.opt_name not in ("adanp", "adan", "adamw"):
            raise ValueError(f"optimizer {optimizer!r}: the reference builds adanp (stage 1) or adamw (stage 2/3)")
        self.betas = betas or ((0.98, 0.92, 0.99) if self.opt_name.startswith("adan") else (0.9, 0.95))
        self.eps, self.max_grad_norm = eps, float(max_grad_norm or 0.0)
        dev = self.pool.device
        # ---- trainable sets: the projector (stage 1/2) and / or the LoRA adapters (stage 2/3)
        self.stores: List[_TrainStore] = []
        if self.pool.requires_grad:
            nodecay = sum(v.numel() for nme, v in self.pool.named_parameters() if v.dim() == 1 or nme.endswith(".bias"))
            self.stores.append(_TrainStore("rgb_pooler", self.pool.master, self.pool.grad, self.pool.shadow, self.pool.refresh_transposed,
                                           _pooler_wd_ranges(self.pool), bucket_ranges(self.pool), self.pool.num_parameters() - nodecay,
                                           nodecay))
        lora = getattr(model.text, "lora", None)
        if lora is not None:
            lb = [(str(l), s, e_) for l, (s, e_) in enumerate(lora.layer_range)]
            self.stores.append(_TrainStore("lora", lora.master, lora.grad, lora.shadow, lora.refresh, [[0, lora.numel, True]], lb,
                                           lora.num_parameters(), 0))
        if not self.stores:
            raise ValueError("nothing to train: the projector is frozen and no LoRA adapters are enabled")
        self.state = {}
        for st in self.stores:
            z = lambda: torch.zeros(st.numel, device=dev)  # noqa: E731
            self.state[st.name] = dict(exp_avg=z(), exp_avg_sq=z())
            if self.opt_name.startswith("adan"):
                self.state[st.name].update(exp_avg_diff=z(), pre_grad=z())
        self.gnorm_sq = torch.zeros((), device=dev)
        self.global_steps = 0
        # DeepSpeed gradient_accumulation_steps (main_pretrain_stage1.py:61,115): backward() scales the loss by 1/GAS and sums the
        # micro-batch gradients; step() only acts on every GAS-th call; the all-reduce happens once, at the boundary
        self.gas = max(1, int(gradient_accumulation_steps))
        self.micro_steps = 0
        self._acc = {st.name: torch.zeros_like(st.grad) for st in self.stores} if self.gas > 1 else {}
        self.optimizer = _Optimizer(lr, weight_decay, sum(s.numel_decay for s in self.stores), sum(s.numel_nodecay for s in self.stores))
        # ---- data parallel
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(self.pg)
        # LHRS_DP_SINGLE_RANK=1 (tools/dp_overlap_trace.py): keep the bucketed reducer on a 1-rank RCCL group, so the comm stream and its
        # overlap with the pooler backward can be traced on a 1-GPU box; the numbers a rank ends with do not change (sum over one rank)
        import os
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        with_reducer = self.world > 1 or (dist_on and os.environ.get("LHRS_DP_SINGLE_RANK") == "1")
        mode = reduce_mode or os.environ.get("LHRS_DP_REDUCE", "bucketed")   # "flat": one all-reduce per trainable store and step
        self.reducers = {st.name: GradReducer(st.grad, st.buckets, self.pg, comm_dtype, mode) for st in self.stores} if with_reducer else {}
        if self.world > 1:
            self.sync_replicas(broadcast_trainable)

    # ------------------------------------------------------------------ replica consistency at start-up
    # dict keys that hold tensors DERIVED from another entry of the same dict (transposed copies, decode re-tilings, e4m3 copies): skipped by
    # the replica checksum because the tensor they come from is already in it.  An explicit table - a frozen tensor whose name merely ends
    # in one of these letters is NOT skipped
    DERIVED_KEYS = frozenset(b + suf for b in ("qkv_w", "o_w", "gu_w", "down_w") for suf in ("T", "p", "8", "8s", "8p", "T8", "T8s", "i8", "i8s", "q4")) | \
        frozenset(("lm_head8", "lm_head8s", "lm_head8p", "lm_headp"))

    def replica_checksums(self) -> torch.Tensor:
        """fp64 [sum, sum of squares, position-weighted sum] of every frozen tensor group (ViT, LLaMA) and of every trainable master: what
        must be equal on all ranks before the first step.  The third term (element i of a tensor weighted by 1 + (i mod 8191) / 8191) makes
        the check sensitive to permutations.  Pure reductions on the device (the LLaMA pass reads 13.5 GB once)."""
        dev = self.pool.device
        derived = self.DERIVED_KEYS

        def walk(o):
            if torch.is_tensor(o):
                yield o
            elif isinstance(o, dict):
                for k in sorted(o, key=str):
                    if k not in derived:
                        yield from walk(o[k])
            elif isinstance(o, (list, tuple)):
                for v in o:
                    yield from walk(v)

        CH = 8191 * 2048                                     # chunk: a multiple of the ramp period, so every chunk sees the same weights
        ramp = (1.0 + torch.arange(8191, device=dev, dtype=torch.float64) / 8191.0).repeat(2048)

        def three(c):
            c = c.reshape(-1).double()
            return torch.stack((c.sum(), (c * c).sum(), (c * ramp[: c.numel()]).sum()))

        rows = []
        for part in (getattr(getattr(self.model, "rgb", None), "p", None), getattr(getattr(self.model, "text", None), "p", None)):
            acc = torch.zeros(3, dtype=torch.float64, device=dev)
            for t in walk(part or {}):
                if t.is_floating_point():
                    for c in t.reshape(-1).split(CH):            # chunked so the fp64 temporaries stay small
                        acc += three(c)
            rows.append(acc)
        for st in self.stores:
            rows.append(sum(three(c) for c in st.master.reshape(-1).split(CH)))
        return torch.stack(rows)

    def sync_replicas(self, broadcast_trainable: bool = True) -> None:
        """What DeepSpeed does inside `deepspeed.initialize` (main_pretrain_stage1.py:215-220: the engine broadcasts the module's
        parameters from rank 0) plus a hard check the reference lacks: rank 0's trainable masters are broadcast (80 M fp32 = 320 MB over
        xGMI, once), then a checksum of every frozen tensor group and of the masters is compared across ranks (MIN/MAX all-reduce of a
        [groups, 2] fp64 table).  A mismatch - e.g. a checkpoint loaded on one rank only, or ranks seeded differently - is an error."""
        dist = torch.distributed
        if broadcast_trainable:
            for st in self.stores:
                dist.broadcast(st.master, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
                hk.cast_f32_to_bf16(st.master, st.shadow) if st.master.is_cuda else st.shadow.copy_(st.master)
                st.refresh()
        cs = self.replica_checksums()
        if not bool(torch.isfinite(cs).all()):
            raise RuntimeError("non-finite values in this rank's frozen weights or trainable masters (replica checksum is NaN / inf): "
                               "the weights are broken before any rank comparison")
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            names = ["rgb (frozen ViT)", "text (frozen LLaMA)"] + [f"trainable:{st.name}" for st in self.stores]
            bad = [n for n, a, b in zip(names, lo.tolist(), hi.tolist()) if a != b]
            raise RuntimeError(f"data-parallel replicas differ at engine construction in {bad}: every rank must load / seed the same "
                               "weights (rank-0 broadcast covers only the trainable parameters)")
        self.replica_checksum = cs

    # back-compat accessors used by tests / tools (stage-1: the projector's optimizer state)
    @property
    def exp_avg(self):
        return self.state[self.stores[0].name]["exp_avg"]

    # ------------------------------------------------------------------ engine surface
    def __call__(self, batch):
        return self.model(batch)

    def train(self):
        self.model.train()
        return self

    def is_gradient_accumulation_boundary(self) -> bool:
        return (self.micro_steps + 1) % self.gas == 0

    def backward(self, loss=None):
        """Hand-written backward; gradient ranges are all-reduced on the comm stream as they become final (with gradient
        accumulation: once per window, after the last micro-batch has been added)."""
        pool_on = self.pool.requires_grad
        overlap = self.gas == 1
        r_lora = self.reducers.get("lora") if overlap else None
        d_image = self.model.text.backward(loss_scale=1.0 / self.gas, need_input_grad=pool_on,
                                           on_layer_ready=(lambda l: r_lora.ready(str(l))) if r_lora else None)
        if pool_on:
            r_pool = self.reducers.get("rgb_pooler") if overlap else None
            self.pool.backward(d_image, on_ready=r_pool.ready if r_pool else None)
        if self.gas > 1:
            first, last = self.micro_steps % self.gas == 0, self.is_gradient_accumulation_boundary()
            for st in self.stores:
                if not first:
                    hk.accum_f32(st.grad, self._acc[st.name])                # grad = this micro-batch + the window so far
                if not last:
                    hk.accum_f32(self._acc[st.name], st.grad, copy_only=True)
            if last:
                for st in self.stores:
                    r = self.reducers.get(st.name)
                    if r is not None:
                        for key, _, _ in st.buckets:
                            r.ready(key)

    def step(self, lr_kwargs: Optional[Dict] = None):
        self.micro_steps += 1
        if self.micro_steps % self.gas != 0:
            return  # DeepSpeed: engine.step() is a no-op inside an accumulation window
        for r in self.reducers.values():
            r.finish()
        self.global_steps += 1
        step = self.global_steps
        gscale = 1.0 / self.world
        if self.max_grad_norm > 0:
            for i, st in enumerate(self.stores):
                hk.sqnorm(st.grad, self.gnorm_sq, accumulate=i > 0)
        gn = self.gnorm_sq if self.max_grad_norm > 0 else None
        g_dec, g_nodec = self.optimizer.param_groups
        for st in self.stores:
            ranges = st.wd_ranges
            if g_dec["lr"] == g_nodec["lr"] and g_dec["weight_decay"] == g_nodec["weight_decay"]:
                ranges = [[0, st.numel, True]]  # one launch over the whole flat buffer (the shipped YAMLs: wd = 0)
            S = self.state[st.name]
            for start, end, dec in ranges:
                grp = g_dec if dec else g_nodec
                sl = slice(start, end)
                if self.opt_name.startswith("adan"):
                    hk.adan_step(st.master[sl], st.grad[sl], S["exp_avg"][sl], S["exp_avg_diff"][sl], S["exp_avg_sq"][sl],
                                 S["pre_grad"][sl], st.shadow[sl], step, grp["lr"], self.betas, self.eps, grp["weight_decay"],
                                 no_prox=self.opt_name == "adanp", gnorm_sq=gn, max_norm=self.max_grad_norm, grad_scale=gscale)
                else:
                    hk.adamw_step(st.master[sl], st.grad[sl], S["exp_avg"][sl], S["exp_avg_sq"][sl], st.shadow[sl], step, grp["lr"],
                                  self.betas, self.eps, grp["weight_decay"], gnorm_sq=gn, max_norm=self.max_grad_norm,
                                  grad_scale=gscale)
            st.refresh()
        self.optimizer._global_grad_norm = self.gnorm_sq  # device scalar (squared, un-averaged); see grad_norm()

    # ------------------------------------------------------------------ checkpoint (engine.save/load_checkpoint surface)
    def state_dict(self) -> Dict:
        sd = dict(global_steps=self.global_steps, opt=self.opt_name, param_groups=[dict(g) for g in self.optimizer.param_groups],
                  stores={})
        for st in self.stores:
            sd["stores"][st.name] = dict(master=st.master.cpu(), **{k: v.cpu() for k, v in self.state[st.name].items()})
        if "rgb_pooler" in sd["stores"]:
            sd["master"] = sd["stores"]["rgb_pooler"]["master"]  # stage-1 convenience alias
        return sd

    def load_state_dict(self, sd: Dict) -> None:
        assert sd["opt"] == self.opt_name, f"checkpoint optimizer {sd['opt']} != {self.opt_name}"
        self.global_steps = int(sd["global_steps"])
        for st in self.stores:
            rec = sd["stores"][st.name]
            st.master.copy_(rec["master"])
            for k in self.state[st.name]:
                self.state[st.name][k].copy_(rec[k])
            hk.cast_f32_to_bf16(st.master, st.shadow)
            st.refresh()
        for g, s in zip(self.optimizer.param_groups, sd["param_groups"]):
            g.update(s)

    def save_checkpoint(self, save_dir: str, tag: str, client_state: Optional[Dict] = None) -> None:
        import os
        os.makedirs(os.path.join(save_dir, tag), exist_ok=True)
        torch.save(dict(engine=self.state_dict(), client_state=client_state or {}), os.path.join(save_dir, tag, "engine.pt"))

    def load_checkpoint(self, path: str):
        import os
        st = torch.load(os.path.join(path, "engine.pt") if os.path.isdir(path) else path, map_location="cpu")
        self.load_state_dict(st["engine"])
        return path, st.get("client_state", {})

    def grad_norm(self) -> float:
        """Host read of the global gradient norm of the last step (synchronises; for logging only)."""
        return float(self.gnorm_sq.sqrt().item()) / self.world

    def set_lr(self, lr: float):
        for g in self.optimizer.param_groups:
            g["lr"] = lr
