"""Data-parallel step on the real engine: 2 ranks (gloo backend, both on GPU 0 - the GPU box has one device; the driver's
multi-GPU runs use "nccl" = RCCL) must end with identical parameters, equal to a single-process step on the averaged
gradients of the two micro-batches (DeepSpeed / DDP mean-of-means semantics)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from lhrs_bot_amd.engine import LHRSEngine
from lhrs_bot_amd.unibind import UniBind
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)

def batch(r):
    g = torch.Generator().manual_seed(322 + r)
    ids = torch.randint(3, 32000, (2, 12), generator=g); ids[:, 0] = 1; ids[:, 1] = -200
    labels = ids.clone(); labels[:, :2] = -100
    return dict(rgb=torch.randn(2, 3, 224, 224, generator=g), input_ids=ids, labels=labels, attention_mask=ids.ne(0))

def model():
    m = UniBind(("rgb", "text"), None, device="cuda:0", llama_layers=1).init_random(seed=0)
    m.prepare_for_training()
    return m

# ---- reference: one process, gradients of both micro-batches averaged, one Adan step with the clip
ref = LHRSEngine(model(), optimizer="adanp", lr=1e-3, max_grad_norm=0.3)
assert ref.world == 1
gs = []
for r in range(world):
    out = ref(batch(r)); ref.backward(out["total_loss"]); gs.append(ref.pool.grad.clone())
ref.pool.grad.copy_(sum(gs) / world)
ref.step()
want = ref.pool.master.clone()

# ---- data parallel: this rank's micro-batch only, bucketed all-reduce during backward, averaging folded into the step
torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
eng = LHRSEngine(model(), optimizer="adanp", lr=1e-3, max_grad_norm=0.3)
assert eng.world == world and eng.reducers
out = eng(batch(rank)); eng.backward(out["total_loss"]); eng.step()
torch.cuda.synchronize()
err = (eng.pool.master - want).abs().max().item()
upd = (want - model().rgb_pooler.master).abs().max().item()
assert err <= 1e-6 + 1e-3 * upd, (err, upd)
mine = eng.pool.master.cpu()
other = [torch.empty_like(mine) for _ in range(world)]
torch.distributed.all_gather(other, mine)
assert all(torch.equal(o, other[0]) for o in other), "ranks diverged"
# ---- replica consistency at construction (DeepSpeed broadcasts in deepspeed.initialize; here: rank-0 broadcast of the trainable
# masters + a checksum of everything frozen).  (i) projector seeded differently per rank -> repaired by the broadcast;
# (ii) frozen LLaMA differing on one rank -> hard error on every rank
m2 = UniBind(("rgb", "text"), None, device="cuda:0", llama_layers=1).init_random(seed=0)
m2.rgb_pooler.init_random(seed=100 + rank)
m2.prepare_for_training()
e2 = LHRSEngine(m2, optimizer="adanp")
got = [torch.empty_like(mine) for _ in range(world)]
torch.distributed.all_gather(got, e2.pool.master.cpu())
assert all(torch.equal(o, got[0]) for o in got), "broadcast did not equalise the trainable masters"
assert torch.equal(e2.pool.shadow.float().cpu(), got[0].to(torch.bfloat16).float()), "bf16 shadow not refreshed after the broadcast"
m3 = UniBind(("rgb", "text"), None, device="cuda:0", llama_layers=1).init_random(seed=0)
if rank == 1:
    m3.text.p["layers"][0]["o_w"][5, 7] += 0.5
m3.prepare_for_training()
try:
    LHRSEngine(m3, optimizer="adanp")
    raise SystemExit("replica mismatch in the frozen LLaMA was not detected")
except RuntimeError as e:
    assert "text (frozen LLaMA)" in str(e) and "rgb" not in str(e).split("in [")[1], str(e)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok{rank}"), "w").write(f"{err} {upd}")
'''


@pytest.mark.timeout(900)
def test_two_rank_step_equals_single_process_on_averaged_gradients(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script), ROOT, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port), timeout=800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


RCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from lhrs_bot_amd.engine import GradReducer
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # RCCL, one rank: the collective is a copy
g = torch.Generator().manual_seed(0)
for comm in (torch.bfloat16, torch.float32):
    flat = torch.randn(3 * 1024 * 1024 + 64, generator=g).to(dev)
    want = flat.to(comm).float()                       # what a sum over one rank in the comm dtype leaves behind
    n = flat.numel()
    red = GradReducer(flat, [("a", 0, n // 3), ("b", n // 3, n // 2), ("c", n // 2, n)], None, comm)
    for rep in range(3):                                # buckets become final while "compute" keeps the stream busy
        src = flat.clone()
        for key in ("c", "b", "a"):
            x = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
            red.ready(key)
        red.finish()
        torch.cuda.synchronize()
        assert torch.equal(flat, src.to(comm).float()), (comm, rep)
    assert torch.equal(flat, want)
torch.distributed.barrier(); torch.distributed.destroy_process_group()
open(os.path.join(sys.argv[2], "ok_rccl"), "w").write("ok")
'''


@pytest.mark.timeout(600)
def test_grad_reducer_on_rccl_single_rank(tmp_path):
    """The bucketed reducer through the REAL RCCL backend ("nccl"), one rank on the box's single GPU: async collectives issued from
    the comm stream, bf16 wire dtype with the widening copy, finish() ordering - what the driver's multi-GPU runs execute."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], capture_output=True, text=True, env=env, timeout=500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "ok_rccl").exists()
