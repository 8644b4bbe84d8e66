"""Sample rocm-smi (sclk / power) while the bf16 GEMM runs on uniform-random vs all-zero operands: shows the DVFS ceiling the
MFMA-bound kernels run under.  Usage: python tools/clock_probe.py"""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lhrs_bot_amd import _lib, kernels as hk

_lib.load()
m, n, k = 8190, 12288, 4096
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out))
        except Exception as e:  # noqa: BLE001
            samples.append((time.time(), repr(e)))
        time.sleep(0.05)


def run(kind, secs=4.0):
    global samples
    if kind == "zeros":
        a = torch.zeros(m, k, device="cuda", dtype=torch.bfloat16); b = torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
    else:
        a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16); b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): hk.gemm_nt(a, b, out=c)
    torch.cuda.synchronize()
    samples = []
    t0 = time.time(); it = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): hk.gemm_nt(a, b, out=c)
        it += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    tf = 2.0 * m * n * k * it / (e0.elapsed_time(e1) * 1e-3) / 1e12
    sclk, pw = [], []
    for _, o in samples[2:]:
        for mm in re.finditer(r'"sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', o): sclk.append(int(mm.group(1)))
        for mm in re.finditer(r'"(?:Current Socket Graphics Package Power \(W\)|Average Graphics Package Power \(W\))"\s*:\s*"([\d.]+)"', o): pw.append(float(mm.group(1)))
    print(f"{kind:6s} {tf:7.1f} TF  sclk MHz min/mean/max = {min(sclk) if sclk else None}/{sum(sclk)/max(1,len(sclk)):.0f}/{max(sclk) if sclk else None}"
          f"  power W mean = {sum(pw)/max(1,len(pw)):.0f} ({len(samples)} samples)")
    if not sclk and samples:
        print(samples[-1][1][:1500])


th = threading.Thread(target=sampler, daemon=True); th.start()
print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True).stdout[-400:])
run("zeros"); run("rand"); run("zeros"); run("rand")
stop = True
