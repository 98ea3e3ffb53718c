import json, sys
sys.path.insert(0, ".")
import torch
from dolfinx_external_operator_amd import Context
import tools.bench_device_loop as dl
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
r = dl.assign_leg(torch, ctx, stream, 108)
print(json.dumps({k: r[k] for k in ("ms_per_call", "ms_per_call_with_range_check_sync", "last_writer_spot_check")} | {"plan": {k: v for k, v in r["plan"].items() if k != "meaning"}}))
