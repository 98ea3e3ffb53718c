import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np
from dolfinx_external_operator_amd import Context, DeviceMesh, make_von_mises
from tools.synthetic import structured_mesh
ctx = Context(0)
m = structured_mesh("hexahedron", (108, 108, 108), 2, distort=0.2, seed=0)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
n, d = m.num_cells * m.nq, 6
rng = np.random.Generator(np.random.PCG64(3))
Du = rng.normal(0.0, 1e-3, m.node_x.shape[0] * 3)
sigma_n = np.zeros(n * d); p = np.zeros(n)
C = np.zeros(n * d * d)
ext = make_von_mises(sigma_n, p, ctx=ctx, state="resident", outputs=(C, None, None))
op = dm.operand("eps", lambda: Du, lazy=True, snapshot=False)
ctx.set_option("timing", 1)
for i in range(4):
    t0 = time.perf_counter(); lz = op.eval(None); t1 = time.perf_counter()
    out = ext((1,))(lz); t2 = time.perf_counter()
    print(json.dumps({"eval_ms": round((t1 - t0) * 1e3, 2), "call_ms": round((t2 - t1) * 1e3, 2), "lib": {k: round(v, 2) for k, v in ctx.last_timing().items()}}))
