"""Condense rocprofv3 --pmc passes into profiles/: python tools/pmc_summary.py traffic <fetch_dir> <write_dir> <out.json>
                                                  python tools/pmc_summary.py counters <dir> [<dir> ...] <out.csv> <kernel substring>
traffic: mean FETCH_SIZE / WRITE_SIZE (KB) per launch of gemm_nt_256s_kernel<0, 0, false> in a bench.py run, calibrated on kernels of known byte
count in the SAME run (cast_f32_to_bf16: numel * 4 B read, numel * 2 B written) as MI355X_MICROARCH.md's HBM section prescribes."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[-1]
    return list(csv.DictReader(open(f)))


def per_kernel(rows, counter):
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


if sys.argv[1] == "traffic":
    fdir, wdir, out = sys.argv[2:5]
    want = sys.argv[5] if len(sys.argv) > 5 else "gemm_u4_kernel<0, false>"
    fetch, write = per_kernel(load(fdir), "FETCH_SIZE"), per_kernel(load(wdir), "WRITE_SIZE")
    dom = next(k for k in fetch if want in k)
    cast = next((k for k in fetch if "cast_f32_bf16" in k or "cast_f32_to_bf16" in k), None)
    mean = lambda v: sum(v) / len(v)  # noqa: E731
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE  /  --pmc WRITE_SIZE (two separate passes) -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra (default micro-batch)",
           "kernel": dom[:120], "kernel_prefix": want, "micro_batch": int(sys.argv[6]) if len(sys.argv) > 6 else 60, "launches": len(fetch[dom]), "fetch_size_kb_mean": mean(fetch[dom]),
           "write_size_kb_mean": mean(write[dom])}
    if cast:
        res["calibration"] = {"kernel": cast, "fetch_size_kb_mean": mean(fetch[cast]), "write_size_kb_mean": mean(write[cast]),
                              "note": "the projector's fp32 master -> bf16 shadow cast reads 4 B and writes 2 B per parameter (79.9 M padded): FETCH_SIZE reports ~1/2 of the read bytes of a "
                                      "wide streaming read on gfx950, WRITE_SIZE the written bytes - so reads are doubled, writes taken as is"}
    res["traffic_bytes_per_launch"] = int(2 * res["fetch_size_kb_mean"] * 1024 + res["write_size_kb_mean"] * 1024)
    res["bench_note"] = (f"bytes per launch of {want} from profiles/{os.path.basename(out)}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes of "
                         "`bench.py --steps 1 --warmup 1` at the default micro-batch on this tree, mean over the launches of that kernel in the step), FETCH_SIZE doubled as "
                         "MI355X_MICROARCH.md prescribes for gfx950 and calibrated on a kernel of known byte count in the same run; memory-side L2 traffic, Infinity-Cache hits "
                         "included; NOT measured in this process")
    res["note"] = ("memory-side L2 traffic, Infinity-Cache hits included (A + B of a launch fit the 256 MB cache): each XCD's 4 MB L2 streams 12 two-MB operand panels per "
                   "round of 32 tiles and cannot keep them for the next round - the floor of any tile order at 4 MB per XCD is ~2.5x the algorithmic bytes (docs/design_notes_r01_r02.md §4)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])
else:
    *dirs, out, sub = sys.argv[2:]
    tot = collections.OrderedDict()
    n = 0
    for d in dirs:
        rows = [r for r in load(d) if sub in r["Kernel_Name"]]
        for r in rows:
            tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    with open(out, "w") as f:
        f.write(f"# mean per launch over the traced launches of kernels matching '{sub}' (rocprofv3 --kernel-trace --pmc, separate passes)\n")
        for k, v in tot.items():
            f.write(f"{k},{sum(v) / len(v):.0f},launches={len(v)}\n")
    print(open(out).read())
