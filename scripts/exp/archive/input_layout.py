import json, sys, pathlib, statistics
sys.path.insert(0, '/root/repo')
import torch
from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams
dev = torch.device("cuda:0")
n, d = 10_000_000, 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
def timeit(args, reps=5):
    for _ in range(5): ctx.von_mises(prm, d, n, MEM_DEVICE, *args)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(20): ctx.von_mises(prm, d, n, MEM_DEVICE, *args)
        b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    return round(448 * n / statistics.median(ts) / 1e6)
g = torch.Generator(device=dev); g.manual_seed(1)
order = sys.argv[1] if len(sys.argv) > 1 else "inputs_first"
def make_inputs_sep():
    deps = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 3e-3, generator=g)
    sig = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 100.0, generator=g)
    pp = torch.empty(n, dtype=torch.float64, device=dev).normal_(0.0, 1e-3, generator=g).abs_()
    return deps, sig, pp
def make_inputs_slab():
    slab = torch.empty(n * 13, dtype=torch.float64, device=dev)
    slab[:n*6].normal_(0.0, 3e-3, generator=g); slab[n*6:n*12].normal_(0.0, 100.0, generator=g); slab[n*12:].normal_(0.0, 1e-3, generator=g).abs_()
    return slab[:n*6], slab[n*6:n*12], slab[n*12:]
if order == "inputs_first":
    sep = make_inputs_sep(); slab = make_inputs_slab()
outs = ctx.vm_output_tensors(n, d)
info = outs[0].dxo_block.info
print({k: info[k] for k in ("chosen", "chosen_kind", "chosen_GBps", "tuned_blocks_per_cu")}, [k[0] + str(round(v)) for k, v in zip(info["kinds"], info["probe_GBps"])])
if order != "inputs_first":
    sep = make_inputs_sep(); slab = make_inputs_slab()
o = tuple(t.data_ptr() for t in outs)
print("separate torch inputs:", timeit(tuple(t.data_ptr() for t in sep) + o))
print("one-slab torch inputs:", timeit(tuple(t.data_ptr() for t in slab) + o))
raw = ctx.device_alloc(n * 13 * 8)
ctx.copy(raw, slab[0].data_ptr(), n * 6 * 8, 2); ctx.copy(raw + n*48, slab[1].data_ptr(), n * 6 * 8, 2); ctx.copy(raw + n*96, slab[2].data_ptr(), n * 8, 2)
print("hipMalloc slab inputs:", timeit((raw, raw + n * 48, raw + n * 96) + o))
print("separate again:", timeit(tuple(t.data_ptr() for t in sep) + o))
