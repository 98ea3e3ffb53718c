import gc, sys, json
sys.path.insert(0, ".")
import torch
from dolfinx_external_operator_amd import Context
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
n, d = 10_000_000, 6
for i in range(4):
    t = ctx.vm_output_tensors(n, d)
    info = dict(t[0].dxo_block.info)
    print(i, hex(t[0].data_ptr()), {k: info[k] for k in ("rounds", "chosen_GBps", "calibration_ms", "chosen_kind", "candidates")}, flush=True)
    del t
    gc.collect()
