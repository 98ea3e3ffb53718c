#!/usr/bin/env python3
"""Launch shape (option blocks_per_cu: 0 one tile per wave, k persistent workgroups per CU) of the other HBM-bound kernels,
outputs in a block calibrated with the kernel itself (dxo_output_alloc_probed, shapes tried by the calibration)."""
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, IsiharaParams  # noqa: E402

dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)


def ev_time(fn, launches=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(launches):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / launches)
    return statistics.median(ts)


def with_shape(fn):
    def launch(ptrs, shape):
        ctx.set_option("blocks_per_cu", shape)
        fn(ptrs)
    return launch


n = 50_000_000
T = torch.rand(n, device=dev, dtype=torch.float64) + 0.5
sg = torch.randn(n, 2, device=dev, dtype=torch.float64)
heat = lambda ptrs: ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), *ptrs)  # noqa: E731
out = ctx.output_tensors_probed((n * 2, n * 2, n * 4), with_shape(heat), bytes_per_launch=88.0 * n, shapes=(0, 16, 32))
info = out[0].dxo_block.info
row = {"kernel": "heat, 5e7 points", "calibration": {k: info[k] for k in ("chosen_kind", "chosen_GBps", "tuned_blocks_per_cu")}}
for bpc in (0, 16, 32):
    ctx.set_option("blocks_per_cu", bpc)
    row[f"GBps_bpc{bpc}"] = round(88 * n / ev_time(lambda: heat([t.data_ptr() for t in out])) / 1e6)
print(json.dumps(row), flush=True)
del out, T, sg
n = 20_000_000
Ft = torch.randn(n, 4, device=dev, dtype=torch.float64) * 0.05 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
iprm = IsiharaParams(0.5, 1.0, 1.0, 1.5)
isi = lambda ptrs: ctx.isihara(iprm, n, MEM_DEVICE, Ft.data_ptr(), *ptrs)  # noqa: E731
out = ctx.output_tensors_probed((n * 16, n * 4), with_shape(isi), bytes_per_launch=192.0 * n, shapes=(0, 16, 32))
info = out[0].dxo_block.info
row = {"kernel": "analytic Isihara, 2e7 points", "calibration": {k: info[k] for k in ("chosen_kind", "chosen_GBps", "tuned_blocks_per_cu")}}
for bpc in (0, 16, 32):
    ctx.set_option("blocks_per_cu", bpc)
    row[f"GBps_bpc{bpc}"] = round(192 * n / ev_time(lambda: isi([t.data_ptr() for t in out])) / 1e6)
print(json.dumps(row), flush=True)
