"""N launches of a prototype (ULIB=libu5.so) on one problem, for rocprofv3 --pmc passes:  python u_one.py M N K launches"""
import ctypes, os, sys
import torch
u = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("ULIB", "libu5.so")))
u.gemm_u_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
m, n, k, reps = (int(x) for x in sys.argv[1:5])
a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    assert u.gemm_u_launch(a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, st) == 0
torch.cuda.synchronize()
