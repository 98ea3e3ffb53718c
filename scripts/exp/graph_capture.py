#!/usr/bin/env python3
"""Can the device entry points be captured in a HIP graph (torch.cuda.graph)? One Jacobi-CG iteration body on a SMALL mesh
(launch-bound): dxo_tangent_apply_vm + the vector updates, eager against graph replay. usage: python scripts/exp/graph_capture.py [n_side]"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
m = structured_mesh("triangle", (n_side, n_side), 2, distort=0.1, seed=0)
G, d = 2, 4
nn, npts = m.node_x.shape[0], m.num_cells * m.nq
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
g = torch.Generator(device=dev)
g.manual_seed(0)
f64 = dict(dtype=torch.float64, device=dev)
sigma = torch.randn(npts * d, generator=g, **f64) * 100
dp = (torch.randn(npts, generator=g, **f64) * 1e-3).clamp_(min=0.0)
minv = torch.rand(nn * G, generator=g, **f64) + 0.5
x, r, z, pk, Ap = (torch.zeros(nn * G, **f64) for _ in range(5))
r.copy_(torch.randn(nn * G, generator=g, **f64))
z.copy_(minv * r)
pk.copy_(z)
rz = torch.dot(r, z).reshape(1)


def body():
    """one preconditioned CG iteration, every scalar on the device"""
    dm.tangent_apply_vm(prm, sigma.data_ptr(), dp.data_ptr(), pk.data_ptr(), Ap.data_ptr())
    alpha = rz / torch.dot(pk, Ap)
    x.add_(alpha * pk)
    r.sub_(alpha * Ap)
    torch.mul(minv, r, out=z)
    rz_new = torch.dot(r, z).reshape(1)
    pk.mul_(rz_new / rz).add_(z)
    rz.copy_(rz_new)


state0 = [t.clone() for t in (x, r, z, pk, rz)]


def reset():
    for t, s in zip((x, r, z, pk, rz), state0):
        t.copy_(s)


def run(fn, its):
    reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(its):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / its * 1e6, x.clone()


body()            # warm-up: the mesh's transposed dofmap and element-vector buffer are built on the first call (not capturable)
torch.cuda.synchronize()
us_eager, x_eager = run(body, 200)
graph = torch.cuda.CUDAGraph()
reset()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    ctx.set_stream(side.cuda_stream)
    body()
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(graph):
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)     # the capture stream
    body()
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
us_graph, x_graph = run(graph.replay, 200)
print({"points": npts, "dofs": nn * G, "us_per_cg_iteration_eager": round(us_eager, 1), "us_per_cg_iteration_graph": round(us_graph, 1),
       "max_abs_diff": float((x_eager - x_graph).abs().max()), "x_norm": float(x_eager.norm())})
