"""Evidence for SURVEY §8(e) on a 1-GPU box: the bucketed gradient all-reduce runs on its own HIP stream, through RCCL ("nccl" backend, one
rank), WHILE the AttnPooler backward is still producing the next buckets.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dp_trace -- python tools/dp_overlap_trace.py run
    python tools/dp_overlap_trace.py summarize gpurun_out/dp_trace profiles/r02_dp_overlap.json

`run`: stage-1 steps (micro-batch 8, 4 LLaMA layers) with LHRS_DP_SINGLE_RANK=1: same engine path as N > 1 ranks (GradReducer.ready() per finished
range: out_proj, layer 5..0, query; bf16 wire dtype), only the group has one member.  `summarize`: from the kernel trace, the RCCL kernels, the
stream they ran on, and how much of their time ran concurrently with the engine's own kernels on the compute stream."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), LHRS_DP_SINGLE_RANK="1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import bench
    from lhrs_bot_amd.engine import LHRSEngine
    from lhrs_bot_amd.unibind import UniBind
    model = UniBind(("rgb", "text"), None, device="cuda:0", llama_layers=4).init_random(seed=0)
    model.prepare_for_training()
    eng = LHRSEngine(model, optimizer="adanp", lr=2e-4, max_grad_norm=0.3, comm_dtype=torch.bfloat16)
    assert eng.reducers, "the reducer must be active"
    batch = bench.make_batch(8, 130, torch.device("cuda", 0), seed=322)
    for _ in range(6):
        out = eng(batch)
        eng.backward(out["total_loss"])
        eng.step()
    torch.cuda.synchronize()
    print("loss", float(out["total_loss"]))
    torch.distributed.destroy_process_group()


def summarize(src, dst):
    trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(trace)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows))
    comm = [e for e in ev if "nccl" in e[2].lower() or "rccl" in e[2].lower()]  # RCCL device kernels are named ncclDevKernel_* (NOT "rocclr" copies)
    mine = [e for e in ev if "anonymous namespace" in e[2] or "lhrs" in e[2]]
    # keep the last 3 steps (steady state): a step's first kernel is patchify
    marks = [e[0] for e in mine if "patchify_kernel" in e[2]]
    t0 = marks[-3]
    comm = [e for e in comm if e[0] >= t0]
    mine = [e for e in mine if e[1] >= t0]
    overlap = 0
    per = []
    for cs, ce, name, stream in comm:
        ov = sum(max(0, min(ce, e) - max(cs, s)) for s, e, _, _ in mine if e > cs and s < ce)
        with_k = sorted({n.split("(")[0].replace("void (anonymous namespace)::", "")[:48] for s, e, n, _ in mine if e > cs and s < ce})
        overlap += min(ov, ce - cs)
        per.append({"kernel": name[:70], "stream": stream, "us": round((ce - cs) / 1e3, 1), "us_concurrent_with_compute": round(min(ov, ce - cs) / 1e3, 1),
                    "concurrent_kernels": with_k[:6]})
    total = sum(ce - cs for cs, ce, _, _ in comm)
    if not comm:
        out = {"command": "rocprofv3 --kernel-trace --output-format csv -- python tools/dp_overlap_trace.py run", "rccl_kernels": 0,
               "note": "a one-rank RCCL group launches no device kernel for an in-place all-reduce: the reducer's stream / event plumbing runs, but there is "
                       "nothing to overlap - the overlap of communication with the pooler backward can only be traced with >= 2 devices"}
        json.dump(out, open(dst, "w"), indent=1)
        print(json.dumps(out))
        return
    out = {"command": "rocprofv3 --kernel-trace --output-format csv -- python tools/dp_overlap_trace.py run (1 rank, RCCL, micro-batch 8, 4 LLaMA layers, bf16 buckets)",
           "steps_summarised": 3, "rccl_kernels": len(comm), "rccl_streams": sorted({c[3] for c in comm}), "compute_streams": sorted({m[3] for m in mine}),
           "rccl_us_total": round(total / 1e3, 1), "rccl_us_concurrent_with_engine_kernels": round(overlap / 1e3, 1),
           "fraction_overlapped": round(overlap / total, 3) if total else None, "per_collective": per[:24]}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_collective"}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarize(sys.argv[2], sys.argv[3])
