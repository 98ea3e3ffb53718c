"""dxo_vm_commit_state (p += dp, sigma_n <- sigma) timed alone: 10^7 points, d = 6, 1.2 GB per call. Run with DXO_HIP_LIBRARY set to a variant
(scripts/exp/build_variant.py c<name> von_mises.hip -DDXO_COMMIT_X2=0|1 -DDXO_COMMIT_BLOCKS_PER_CU=k)."""
import json
import os
import sys

sys.path.insert(0, ".")
import torch

from dolfinx_external_operator_amd import Context

ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, d = 10_000_000, (int(sys.argv[1]) if len(sys.argv) > 1 else 6)
g = torch.Generator(device="cuda").manual_seed(0)
p = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
dp = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
s = torch.randn(n * d, dtype=torch.float64, device="cuda", generator=g)
sn = torch.zeros(n * d, dtype=torch.float64, device="cuda")
ref = p + dp
ctx.vm_commit_state(d, n, p.data_ptr(), dp.data_ptr(), sn.data_ptr(), s.data_ptr())
torch.cuda.synchronize()
ok = bool(torch.equal(p, ref) and torch.equal(sn, s))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record()
    for _ in range(20):
        ctx.vm_commit_state(d, n, p.data_ptr(), dp.data_ptr(), sn.data_ptr(), s.data_ptr())
    e1.record()
    e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(json.dumps({"lib": os.environ.get("DXO_HIP_LIBRARY", "product"), "ms": round(best, 4), "d": d, "TBps": round(n * (16 * d + 24) / best / 1e9, 2), "bits_ok": ok}))
