import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lhrs_bot_amd import kernels as hk
u = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("ULIB", "libu.so")))
u.gemm_u_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
M = 8190
st = torch.cuda.current_stream().cuda_stream
def run_u(a, b, c):
    assert u.gemm_u_launch(a.data_ptr(), b.data_ptr(), c.data_ptr(), a.shape[0], b.shape[0], a.shape[1], st) == 0
for zero in (0, 1):
    for (n, k) in [(4096, 4096), (4096, 11008), (22016, 4096), (4096, 22016), (11008, 4096)]:
        a = (torch.rand(M, k, device="cuda") * 2 - 1).to(torch.bfloat16) * (1 - zero)
        b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16) * (1 - zero)
        c = torch.empty(M, n, device="cuda", dtype=torch.bfloat16); c2 = torch.empty_like(c)
        hk.gemm_set_vendor(False)
        run_u(a, b, c); hk.gemm_nt(a, b, out=c2); torch.cuda.synchronize()
        err = (c.float() - c2.float()).abs().max().item()
        res = []
        for vend, fn in ((False, lambda: run_u(a, b, c)), (False, lambda: hk.gemm_nt(a, b, out=c2)), (True, lambda: hk.gemm_nt(a, b, out=c2))):
            hk.gemm_set_vendor(vend)
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(2.0 * M * n * k / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
        print(f"zero={zero} N={n} K={k}: u {res[0]:7.1f} TF   16-wave {res[1]:7.1f} TF   tuned dispatch (vendor) {res[2]:7.1f} TF   max|u - 16-wave| {err}", flush=True)
