#!/usr/bin/env python3
"""Soak: dxo_vm_output_alloc / free in a loop (17 candidates of both kinds built and torn down every time): free device
memory must come back, nothing may fault, the kernel into every block kept must stay correct."""
import gc
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, d = 10_000_000, 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
ctx = Context(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
slab = torch.empty(n * 13, dtype=torch.float64, device=dev)
slab[:n * 6].normal_(0.0, 3e-3, generator=g)
slab[n * 6:n * 12].normal_(0.0, 100.0, generator=g)
slab[n * 12:].normal_(0.0, 1e-3, generator=g).abs_()
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ref = None
free0 = torch.cuda.mem_get_info(dev)[0]
for c in range(cycles):
    C, s, dp = ctx.vm_output_tensors(n, d)
    info = C.dxo_block.info
    ctx.von_mises(prm, d, n, MEM_DEVICE, slab.data_ptr(), slab.data_ptr() + n * 48, slab.data_ptr() + n * 96, C.data_ptr(), s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    chk = (float(s.sum()), float(dp.sum()), float(C[::1009].sum()))
    if ref is None:
        ref = chk
    assert chk == ref, (c, chk, ref)
    del C, s, dp
    gc.collect()
    free = torch.cuda.mem_get_info(dev)[0]
    print(f"cycle {c}: kept {info['chosen_kind']} at {info['chosen_GBps']:.0f} GB/s, shape {info['tuned_blocks_per_cu']}; free memory {free / 2**30:.2f} GiB "
          f"(start {free0 / 2**30:.2f})", flush=True)
    assert abs(free - free0) < 64 * 2**20, "device memory did not come back"
print("soak ok")
