#!/usr/bin/env python3
"""Generate golden vectors + the weight fixture for the ICNN (Isihara, noise=high) hyperelastic operator.

Run ONLY in the build container (needs /root/reference and its data file doc/demo/Isihara_noise=high.pth).
The reference code is not copied: the classes `convexLinear`, `ICNN`, the construction constants, the
H-correction block (:362-381) and `compute_stress_local` / `vectorized_stress_and_tangent` / `dP_dF_impl`
(:429-456) are pulled out of doc/demo/demo_hyperelasticity.py with `ast` and executed with real torch
(`torch.compile` at :416 is a performance-only step and is skipped).

Outputs
  tests/golden/icnn_isihara_weights.npz : the 9 027 fp32 parameters of the state_dict, raw (data, not code)
  tests/golden/icnn_isihara.npz         : F (N,4) fp64 inputs, dP (N,4,4), P (N,4) reference outputs for fp64
                                          input, plus the same for fp32 input (the reference's dtype follows F)
"""
import ast
import os
import pathlib

import numpy as np
import torch

REF = pathlib.Path("/root/reference/doc/demo/demo_hyperelasticity.py")
PTH = REF.parent / "Isihara_noise=high.pth"
OUT = pathlib.Path(__file__).resolve().parent

WANT_CLASSES = {"convexLinear", "ICNN"}
WANT_FUNCS = {"compute_stress_local", "dP_dF_impl"}
WANT_NAMES = {"n_input", "n_output", "n_hidden", "dropout", "F_0", "W_NN_0", "P_NN_0", "H_flat", "H",
              "vectorized_stress_and_tangent"}


def extract():
    tree = ast.parse(REF.read_text())
    pre, post = [], []
    for node in tree.body:
        take = False
        if isinstance(node, ast.ClassDef) and node.name in WANT_CLASSES:
            take = True
        elif isinstance(node, ast.FunctionDef) and node.name in WANT_FUNCS:
            take = True
        elif isinstance(node, ast.Assign):
            for t in node.targets:
                base = t
                while isinstance(base, (ast.Subscript, ast.Attribute)):
                    base = base.value
                if isinstance(base, ast.Name) and base.id in WANT_NAMES:
                    take = True
        if take:
            # everything up to the network definition goes first; the model is built + loaded in between
            (pre if node.lineno < 310 else post).append(node)
    return ast.Module(body=pre, type_ignores=[]), ast.Module(body=post, type_ignores=[])


def main():
    pre, post = extract()
    ns = {"torch": torch, "np": np}
    exec(compile(pre, str(REF), "exec"), ns)
    torch.manual_seed(0)
    model = ns["ICNN"](n_input=ns["n_input"], n_hidden=ns["n_hidden"], n_output=ns["n_output"], dropout=ns["dropout"])  # :307
    state = torch.load(PTH, map_location="cpu")                                                                            # :314
    model.load_state_dict(state)
    model.eval()                                                                                                            # :315
    ns["model"] = model
    exec(compile(post, str(REF), "exec"), ns)   # H correction (:362-381), compute_stress_local ... dP_dF_impl (:429-456)

    weights = {k.replace(".", "__"): v.detach().cpu().numpy() for k, v in state.items()}
    n_par = sum(v.size for v in weights.values())
    np.savez(OUT / "icnn_isihara_weights.npz", **weights)
    print("parameters:", {k: v.shape for k, v in weights.items()}, "total", n_par)

    # F = I + 0.1 N(0,1), rejected unless det F > 0.2; plus hand-picked states
    rng = np.random.Generator(np.random.PCG64(3))
    F = np.empty((0, 4))
    while F.shape[0] < 1500:
        cand = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(2000, 4))
        det = cand[:, 0] * cand[:, 3] - cand[:, 1] * cand[:, 2]
        F = np.concatenate([F, cand[det > 0.2]])
    F = F[:1500]
    F[0] = [1.0, 0.0, 0.0, 1.0]            # undeformed: P must vanish (that is what H is for)
    F[1] = [1.3, 0.0, 0.0, 1.0 / 1.3]      # isochoric stretch
    F[2] = [1.0, 0.4, 0.0, 1.0]            # simple shear
    F[3] = [0.8, 0.0, 0.0, 0.8]            # compression
    F[4] = [1.5, 0.2, -0.1, 1.4]           # large stretch
    with torch.no_grad():
        pass
    dP64, P64 = ns["dP_dF_impl"](F.reshape(-1, 1, 2, 2))                       # fp64 in -> fp64 out (network in fp32, :286)
    F32 = F.astype(np.float32)
    dP32, P32 = ns["dP_dF_impl"](F32.reshape(-1, 1, 2, 2))
    H = ns["H"].detach().cpu().numpy()
    np.savez(OUT / "icnn_isihara.npz", F=F, dP=dP64.reshape(-1, 4, 4), P=P64.reshape(-1, 4),
             dP_f32in=dP32.reshape(-1, 4, 4), P_f32in=P32.reshape(-1, 4), H=H,
             W_at_identity=float(model(torch.tensor([[1.0, 0, 0, 1.0]])).item()))
    print("H_flat", ns["H_flat"].numpy(), "W_NN(I)", float(model(torch.tensor([[1.0, 0, 0, 1.0]])).item()))
    print("P(I) =", P64.reshape(-1, 4)[0], " |dP| max", np.abs(dP64).max(), "dtype", dP64.dtype, P32.dtype)
    sym = dP64.reshape(-1, 4, 4)
    print("tangent asymmetry (fp32 network noise):", np.abs(sym - sym.transpose(0, 2, 1)).max() / np.abs(sym).max())


if __name__ == "__main__":
    os.environ.setdefault("OMP_NUM_THREADS", "8")
    main()
