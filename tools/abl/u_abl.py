import ctypes, os, sys, torch
M = 8190
st = torch.cuda.current_stream().cuda_stream
d = os.path.dirname(os.path.abspath(__file__))
for name in ("libu4.so", "libu4_NOBAR.so", "libu4_NOWR.so", "libu4_NOLD.so", "libu4_NOLDWR.so", "libu4_NORD.so", "libu4_ALL.so"):
    u = ctypes.CDLL(os.path.join(d, name))
    u.gemm_u_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    out = []
    for (n, k) in [(4096, 4096), (4096, 22016), (22016, 4096)]:
        a = torch.zeros(M, k, device="cuda", dtype=torch.bfloat16); b = torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
        c = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        fn = lambda: u.gemm_u_launch(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, n, k, st)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"{2.0 * M * n * k / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12:7.1f}")
    print(f"{name:18s} zeros: " + " ".join(out), flush=True)
