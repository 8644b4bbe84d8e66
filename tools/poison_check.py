"""Uninitialised-read hunt: fill the caching allocator's free memory with NaN bit patterns (0xFFFFFFFF is a NaN as fp32 and as two
bf16), then run stage-1 / stage-3 training steps and check that loss, gradients and masters stay finite and that a poisoned run gives
the same numbers as a clean one.  A fresh box hands out memory that other tenants wrote; `torch.empty` there is not zero.

    python tools/poison_check.py [--micro-batch 2] [--llama-layers 1] [--stage 1] [--steps 3]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch  # noqa: E402


def poison(dev, big_gb: float, pattern: int = -1):
    big = torch.empty(int(big_gb * 2 ** 30) // 4, dtype=torch.int32, device=dev)
    big.fill_(pattern)
    small = [torch.full((n,), pattern, dtype=torch.int32, device=dev) for n in [128] * 3000 + [2048] * 3000 + [32768] * 800 + [200000] * 300]
    torch.cuda.synchronize()
    del big, small


def run(a, dev, do_poison):
    from lhrs_bot_amd.engine import LHRSEngine
    from lhrs_bot_amd.unibind import UniBind

    torch.cuda.empty_cache()
    if do_poison:
        poison(dev, a.poison_gb, a.pattern)
    model = UniBind(("rgb", "text"), None, device=dev, llama_layers=a.llama_layers).init_random(seed=0)
    if a.stage == 1:
        model.prepare_for_training()
        engine = LHRSEngine(model, optimizer="adanp", lr=2e-4, weight_decay=0.0, max_grad_norm=0.3)
    else:
        model.enable_lora(r=8, alpha=16, targets=("q", "k", "v", "o"))
        model.prepare_for_training(freeze_text=False, tune_rgb_pooler=False)
        engine = LHRSEngine(model, optimizer="adamw", lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
    batch = make_batch(a.micro_batch, a.caption_tokens + 2, dev, seed=322)
    rows = []
    for i in range(a.steps):
        if do_poison:
            poison(dev, a.poison_gb / 4, a.pattern)
        out = engine(batch)
        engine.backward(out["total_loss"])
        g = [(st.name, bool(torch.isfinite(st.grad).all()), float(st.grad.double().abs().sum())) for st in engine.stores]
        engine.step()
        m = [(st.name, bool(torch.isfinite(st.master).all()), float(st.master.double().sum())) for st in engine.stores]
        rows.append((float(out["total_loss"]), g, m))
        if not all(x[1] for x in g):
            for st in engine.stores:
                bad = (~torch.isfinite(st.grad)).nonzero().flatten()
                if bad.numel():
                    print(f"  non-finite gradient in store {st.name}: {bad.numel()} entries, first offsets {bad[:8].tolist()}, last {bad[-4:].tolist()}")
                    pool = getattr(engine, "pool", None)
                    if st.name == "rgb_pooler" and pool is not None:
                        for nme, (off, n) in pool.offsets.items():
                            k = int(((bad >= off) & (bad < off + n)).sum())
                            if k:
                                print(f"    {nme}: {k} of {n}")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--micro-batch", type=int, default=2)
    ap.add_argument("--caption-tokens", type=int, default=128)
    ap.add_argument("--llama-layers", type=int, default=1)
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--poison-gb", type=float, default=24.0)
    ap.add_argument("--pattern", type=lambda s: int(s, 0), default=-1)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    clean = run(a, dev, False)
    dirty = run(a, dev, True)
    ok = True
    for i, (c, d) in enumerate(zip(clean, dirty)):
        same = c == d
        ok &= same
        print(f"step {i}: clean loss {c[0]:.6f} poisoned loss {d[0]:.6f}  identical={same}")
        if not same:
            print("   clean   ", c[1], c[2])
            print("   poisoned", d[1], d[2])
    print("POISON CHECK", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
