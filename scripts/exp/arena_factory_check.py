#!/usr/bin/env python3
"""Does a second kernel-calibrated arena block (the factory's, beside the bench's) run the caller's launches at the rate its calibration
reports? Per process: block A, then block B, each timed with the SAME torch inputs (3 x 12 launches); prints the calibration records
(chosen_GBps = the rate after the rivals were freed, round 6) next to the measured rates. usage: python scripts/exp/arena_factory_check.py"""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

n, d = 10_000_128, 6
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
slab = torch.empty(n * 13, dtype=torch.float64, device=dev)
deps, sn, p = slab[: n * 6].view(n, 6), slab[n * 6: n * 12].view(n, 6), slab[n * 12:]
deps.normal_(0, 3e-3, generator=g); sn.normal_(0, 100.0, generator=g); p.normal_(0, 1e-3, generator=g).abs_()
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))


def rate(out):
    C, s, dp = out
    args = (prm, d, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    ctx.von_mises(*args)
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(12):
            ctx.von_mises(*args)
        e1.record(stream)
        torch.cuda.synchronize()
        res.append(round(448 * n / (e0.elapsed_time(e1) / 12 * 1e-3) / 1e9))
    return res


rows = {}
A = ctx.vm_output_tensors(n, d)
rows["A"] = {"record": {k: A[0].dxo_block.info[k] for k in ("chosen_kind", "chosen_GBps", "rounds")}, "measured": rate(A)}
B = ctx.vm_output_tensors(n, d)
rows["B"] = {"record": {k: B[0].dxo_block.info[k] for k in ("chosen_kind", "chosen_GBps", "rounds")}, "measured": rate(B)}
rows["A_again"] = rate(A)
rows["B_again"] = rate(B)
print(json.dumps(rows), flush=True)
ctx.close()
