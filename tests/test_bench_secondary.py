"""tools/bench_secondary.py: the counter-file parser that attributes FETCH_SIZE / WRITE_SIZE dispatches to the legs' rooflines
(not gpu), and the new legs at small sizes on the GPU (every record carries a recomputable roofline)."""
import pathlib

import numpy as np
import pytest

from tools import bench_secondary as bs


def _csv(tmp_path, counter, rows):
    lines = ["Dispatch_Id,Kernel_Name,Grid_Size,Counter_Name,Counter_Value"]
    lines += [f'{i},"{name}",{grid},{counter},{val}' for i, (name, grid, val) in enumerate(rows, 1)]
    f = tmp_path / f"{counter}_counter_collection.csv"
    f.write_text("\n".join(lines))
    return f


def test_counter_parser_attributes_followers_and_halves(tmp_path):
    rows = [("void at::native::vectorized_elementwise_kernel<4, FillFunctor>", 10, 7),
            ("operand_adjoint_c8_mfma<27>(OperandDev, ...)", 100, 1000), ("node_sum<3>(long, ...)", 50, 500),
            ("tangent_apply<3, 27, 8, false, true>(OperandDev, ...)", 100, 3000), ("node_sum<3>(long, ...)", 50, 500),
            ("vm_commit(long, long, double*, ...)", 40, 10), ("vm_commit(long, long, double*, ...)", 40, 10),
            ("vm_field<2, true, 0, 0, 0>(VmConst, ...)", 64, 5), ("vm_field<2, true, 0, 0, 0>(VmConst, ...)", 640, 50),
            ("vm_field<2, true, 0, 0, 1>(VmConst, ...)", 640, 9),
            ("vm_commit(long, long, double*, ...)", 40, 30), ("vm_commit(long, long, double*, ...)", 40, 30)]
    got = bs.parse_counter_csv([_csv(tmp_path, "FETCH_SIZE", rows)], "FETCH_SIZE")
    assert got["operand_adjoint_c8"] == [[100, 1500 * 1024.0]]          # node_sum added to the call that launched it; gather kernels: x1
    assert got["tangent_apply<3, 27, 8, false*>"] == [[100, (2 * 3000 + 500) * 1024.0]]  # streaming kernel x2, its node_sum x1
    assert bs._pick(got["vm_commit("], "first_half") == 20 * 1024.0 and bs._pick(got["vm_commit("], "second_half") == 60 * 1024.0
    assert bs._pick(got["vm_field<2,*, 0>("]) == 100 * 1024.0             # largest grid only (small set-up dispatches ignored)
    assert bs._pick(got["vm_field<2,*, 1>("]) == 18 * 1024.0              # the (sigma, dp)-only launch is its own kernel name
    out = {"device_loop_q2hex": {"calls": {"internal_force": {"roofline": {"algorithmic_bytes_per_launch": 1024.0 * 1000}}}}}
    w = bs.parse_counter_csv([_csv(tmp_path, "WRITE_SIZE", rows)], "WRITE_SIZE")
    bs.apply_traffic(out, {"fetch": got, "write": w})
    r = out["device_loop_q2hex"]["calls"]["internal_force"]["roofline"]
    assert r["traffic"] == (1500 + 1500) * 1024.0 and r["traffic_over_algorithmic"] == pytest.approx(3.0)


def test_every_traffic_key_names_a_registered_leg():
    assert {leg for leg, _ in bs.TRAFFIC_KEYS} <= set(bs.ALL_LEGS)


@pytest.mark.gpu
def test_new_legs_small(ctx):
    import torch

    from dolfinx_external_operator_amd import VmParams

    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    E = 70e3
    prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
    saved = ctx.get_option("placement_mode")
    ctx.set_option("placement_mode", 0)
    bs.QUICK = True
    try:
        out = bs.secondary_block(torch, ctx, stream, prm, n=200_000, cpu=False, field_cells=12,
                                 legs=("von_mises_cfg2_1e6", "device_loop_q2hex", "device_loop_p2tri", "assign_cg"), traffic=False)
    finally:
        bs.QUICK = False
        ctx.set_option("placement_mode", saved)
    for leg, rec in out.items():
        assert "error" not in rec, (leg, rec)
    for leg in ("device_loop_q2hex", "device_loop_p2tri"):
        rec = out[leg]
        assert set(rec["calls"]) == {"von_mises_field_state", "internal_force", "tangent_apply", "tangent_diagonal", "state_commit"}
        for call in rec["calls"].values():
            r = call["roofline"]
            assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / call["ms_per_call"] / 1e6)
        assert 0.0 < rec["plastic_fraction"] < 1.0 and rec["iteration_ms"] > 0
    assert out["assign_cg"]["last_writer_spot_check"] == "ok"
    assert out["von_mises_cfg2_1e6"]["points"] == 1_000_000


def test_fp64_roofline_is_attached_only_at_the_counted_size():
    """tools/bench_device_loop._attach_fp64: the issued-flop figures of profiles/consumer_flop.json price a call only when the leg's mesh has
    the size they were counted on."""
    import json

    from tools import bench_device_loop as dl

    rec = json.loads((bs.ROOT / "profiles" / "consumer_flop.json").read_text())
    pts = rec["points"]["device_loop_q2hex"]

    def leg(points):
        return {"points": points, "calls": {"tangent_apply": {"ms_per_call": 1.0, "roofline": {}}, "internal_force": {"ms_per_call": 0.5, "roofline": {}}},
                "without_tangent_array": {"calls": {"tangent_apply_vm": {"ms_per_call": 1.0, "roofline": {}}}}}

    a = leg(pts)
    dl._attach_fp64(a, "device_loop_q2hex", 3)
    f = a["without_tangent_array"]["calls"]["tangent_apply_vm"]["roofline"]["fp64_valu"]
    assert f["flop_per_launch_issued"] > 1e10 and f["frac_over_call"] == pytest.approx(f["flop_per_launch_issued"] / 1e9 / 78.6)
    assert "fp64_valu" in a["calls"]["internal_force"]["roofline"] and "fp64_valu" in a["calls"]["tangent_apply"]["roofline"]
    b = leg(pts + 8)
    dl._attach_fp64(b, "device_loop_q2hex", 3)
    assert "fp64_valu" not in b["calls"]["tangent_apply"]["roofline"]
