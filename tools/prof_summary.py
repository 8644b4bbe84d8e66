"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the files kept under profiles/:
    python tools/prof_summary.py <rocprof output dir> <steps> <warmup> <tag>
writes profiles/<tag>_kernel_stats.csv (copy of the stats table) and profiles/<tag>_gemm_launch_summary.json: per persistent-GEMM kernel instantiation the launch
count and the average duration over ALL launches and over the launches of the TIMED steps only (what bench.py measures live with HIP events), and which of the
plain-epilogue instantiations carries the most time (the one bench.py's roofline.achieved is quoted on)."""
import csv, glob, json, os, re, shutil, sys

src, steps, warmup, tag = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[-1]
trace = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(trace)))
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
groups = {  # label -> regex on the demangled kernel name
    "gemm_u4_kernel<0, false> plain (four waves)": r"gemm_u4_kernel<0, false>",
    "gemm_u4_kernel<0, true> plain + residual (four waves)": r"gemm_u4_kernel<0, true>",
    "gemm_u4_kernel<1, false> SwiGLU-fwd epilogue (four waves)": r"gemm_u4_kernel<1, false>",
    "gemm_u4_kernel<2, false> SwiGLU-bwd epilogue (four waves)": r"gemm_u4_kernel<2, false>",
    "gemm_u4_kernel<3, false> RoPE epilogue (four waves)": r"gemm_u4_kernel<3, false>",
    "gemm_nt_256s_kernel<ACT, 0, K2P> plain (16 waves)": r"gemm_nt_256s_kernel<\d, 0, (false|true)>",
    "gemm_nt_256s_kernel<0, 1> SwiGLU-fwd epilogue": r"gemm_nt_256s_kernel<\d, 1, ",
    "gemm_nt_256s_kernel<0, 2> SwiGLU-bwd epilogue": r"gemm_nt_256s_kernel<\d, 2, ",
    "gemm_nt_256s_kernel<0, 3> RoPE epilogue": r"gemm_nt_256s_kernel<\d, 3, ",
    "gemm_nt_144s_kernel<ACT, 0> plain (12 waves, 144-row tiles)": r"gemm_nt_144s_kernel<\d, 0>",
}
# the timed steps start at the first q|k|v product (RoPE epilogue, one per decoder layer) after `warmup` steps
rope = sorted(int(r["Start_Timestamp"]) for r in rows if re.search(r"gemm_u4_kernel<3, false>|gemm_nt_256s_kernel<\d, 3, ", r[nk]))
t0 = rope[len(rope) // (steps + warmup) * warmup] if len(rope) >= steps + warmup else 0
out = {"command": f"rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline --no-extra [+ the flags of the run]",
       "kernels": {}}
for label, rx in groups.items():
    rs = [r for r in rows if re.search(rx, r[nk])]
    if not rs:
        continue
    tm = [r for r in rs if int(r["Start_Timestamp"]) >= t0]
    out["kernels"][label] = {"launches_total": len(rs), "avg_us_all_launches": round(sum(map(dur, rs)) / len(rs) / 1e3, 2), "launches_per_timed_step": len(tm) / steps,
                             "avg_us_timed_steps_only": round(sum(map(dur, tm)) / max(1, len(tm)) / 1e3, 2), "ms_per_timed_step": round(sum(map(dur, tm)) / steps / 1e6, 3)}
plain = [k for k in out["kernels"] if " plain" in k]
out["dominant_plain_instantiation"] = max(plain, key=lambda k: out["kernels"][k]["ms_per_timed_step"]) if plain else None
timed_rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
out["all_kernels_ms_per_timed_step"] = round(sum(map(dur, timed_rows)) / steps / 1e6, 3)
out["persistent_gemm_ms_per_timed_step"] = round(sum(v["ms_per_timed_step"] for v in out["kernels"].values()), 3)
out["note"] = ("the *_kernel_stats.csv averages cover warm-up (first-touch) launches too; avg_us_timed_steps_only is the figure bench.py measures live with HIP events "
               "(roofline.avg_launch_us / variants[*].avg_launch_us)")
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_gemm_launch_summary.json"), "w"), indent=1)
print(json.dumps(out)[:3000])
