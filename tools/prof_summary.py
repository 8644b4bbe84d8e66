"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the files kept under profiles/:
    python tools/prof_summary.py <rocprof output dir> <steps> <warmup> <tag>
writes profiles/<tag>_kernel_stats.csv (copy of the stats table) and profiles/<tag>_gemm_launch_summary.json (dominant kernel:
launch counts, average over all launches and over the launches of the timed steps only)."""
import csv, glob, json, os, shutil, sys

src, steps, warmup, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[-1]
trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(trace)))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
import re
# every activation variant of the plain-epilogue persistent kernels: 256-row tiles and (round 3, chosen per launch at small batches) 144-row tiles
# the dominant kernel: every activation / LoRA-pair variant of the plain-epilogue 256-row persistent kernel (bench.py's roofline.achieved); the plain 144-row
# kernel (ViT / projector products; the dominant one at micro-batch 8) is summarised next to it
dom = [r for r in rows if re.search(r"gemm_nt_256s_kernel<\d, 0, (false|true)(, false)?>", r[name_key])]
r144 = [r for r in rows if re.search(r"gemm_nt_144s_kernel<\d, 0>", r[name_key])]
_rope = sorted(int(r["Start_Timestamp"]) for r in rows if re.search(r"gemm_nt_256s_kernel<\d, 3, ", r[name_key]))
T0 = _rope[len(_rope) // (steps + warmup) * warmup] if len(_rope) >= steps + warmup else 0   # first RoPE-epilogue launch of the first timed step
intimed = lambda rs: [r for r in rs if int(r["Start_Timestamp"]) >= T0]
vend = [r for r in rows if "Cijk_" in r[name_key]]   # the vendor library's kernel on the plain long-k products (csrc/vendor.cpp)
tot = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
u4 = [r for r in rows if "gemm_u4_kernel" in r[name_key]]   # the hand-written four-wave kernel on the plain long-k products (csrc/gemm_u4.hip)
u4_dominant = len(u4) > 0 and tot(intimed(u4)) > max(tot(intimed(dom)), tot(intimed(r144)), tot(intimed(vend)))
vendor_dominant = (len(vend) > 0 and tot(intimed(vend)) > max(tot(intimed(dom)), tot(intimed(r144)))) or u4_dominant   # either way: the timed steps are found by time stamp, below
hand = {"gemm_nt_256s_kernel plain launches": {"launches_total": len(dom), "avg_us": (tot(dom) / len(dom) / 1e3 if dom else None)}}
for epi, nm in ((1, "SwiGLU-fwd"), (2, "SwiGLU-bwd"), (3, "RoPE")):
    rs = [r for r in rows if re.search(r"gemm_nt_256s_kernel<\d, %d, " % epi, r[name_key])]
    if rs:
        hand[f"gemm_nt_256s_kernel<0,{epi}> {nm} epilogue"] = {"launches_total": len(rs), "avg_us": tot(rs) / len(rs) / 1e3}
hand["gemm_u4_kernel (plain long-k products)"] = {"launches_total": len(u4), "avg_us": (tot(u4) / len(u4) / 1e3 if u4 else None)}
vendor_share = {"launches_total": len(vend), "avg_us": (tot(vend) / len(vend) / 1e3 if vend else None)}
if u4_dominant:
    r144, dom = dom + r144, u4
elif vendor_dominant:
    r144, dom = dom + r144, vend
if not vendor_dominant and len(r144) and sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in r144) > sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in dom):
    dom, r144 = r144, dom
dom.sort(key=lambda r: int(r["Start_Timestamp"]))
per_step = len(dom) // (steps + warmup)
timed = dom[-per_step * steps:]
if vendor_dominant:
    # the first call of every problem times all the library's candidate algorithms (csrc/gemm.hip): those launches sit in the warm-up step.  The timed
    # steps start at the first RoPE-epilogue launch (layer 0's q|k|v product) after `warmup` steps of 32 decoder layers
    rope = sorted(int(r["Start_Timestamp"]) for r in rows if re.search(r"gemm_nt_256s_kernel<\d, 3, ", r[name_key]))
    t0 = rope[len(rope) // (steps + warmup) * warmup]
    timed = [r for r in dom if int(r["Start_Timestamp"]) >= t0]
    per_step = len(timed) // steps
avg = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / len(rs) / 1e3
n144 = sum("144s" in r[name_key] for r in timed)
by_name = {}
for r in timed:
    by_name.setdefault(r[name_key][:100], []).append(r)
top = max(by_name, key=lambda k: tot(by_name[k])) if vendor_dominant else ""
out = {"kernel": "gemm_u4_kernel (csrc/gemm_u4.hip, hand-written)" + " (the kernel bench.py's roofline.achieved is quoted on)" if u4_dominant else ((top + f" (vendor library; {len(by_name)} of its kernels were chosen by the first-call timing, this one carries {tot(by_name[top]) / max(tot(timed), 1):.0%} of their time)") if vendor_dominant else "gemm_nt_144s_kernel<ACT, 0>" if n144 else "gemm_nt_256s_kernel<ACT, 0, K2P, false>") + " (the kernel bench.py's roofline.achieved is quoted on)",
       "hand_written_gemm_variants": hand, "vendor_library_kernels_all_launches_incl_first_call_timing": vendor_share,
       "other_plain_persistent_kernel": {"launches_total": len(r144), "avg_us_all_launches": (avg(r144) if r144 else None)},
       "timed_launches_on_144_row_tiles": n144, "timed_launches_on_256_row_tiles": len(timed) - n144,
       "command": f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline [+ the flags of the run]",
       "launches_total": len(dom), "launches_per_step": per_step, "avg_us_all_launches": avg(dom), "avg_us_timed_steps_only": avg(timed),
       "note": "the *_kernel_stats.csv average covers warm-up (first-touch) launches too; the timed-steps average is the one bench.py "
               "measures live with HIP events"}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_gemm_launch_summary.json"), "w"), indent=1)
print(json.dumps(out))
