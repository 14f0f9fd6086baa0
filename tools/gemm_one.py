"""Run one GEMM shape repeatedly (for rocprofv3 --pmc / --kernel-trace on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-sam-adapter_amd"))
import torch, mmsa
ops = mmsa.ops
M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a = ops.split_planes(torch.randn(M, K, device="cuda"), kpad=K)
w = ops.split_planes(torch.randn(N, K, device="cuda") / K ** 0.5)
out = torch.empty(M, N, device="cuda")
for _ in range(reps):
    ops.gemm(a, w, out)
torch.cuda.synchronize()
