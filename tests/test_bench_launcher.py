"""bench.py --gpus N (N > 1) must start its own ranks from a parent that never touches the GPU (not gpu).

The driver may start the scaling run as plain `python bench.py --gpus N`: the parent then spawns
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child. On this pool a process that has
initialised HIP must never exec / be restarted, so the parent must not import torch at all before it launches.
"""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]

PROBE = r"""
import json, sys, types
sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
import subprocess
calls = []
class FakeDone:
    returncode = 0
    stdout = b'noise from a library\n{"metric": "m", "value": 1.0, "n_gpus": 4}\n'
def fake_run(cmd, **kw):
    calls.append((cmd, kw))
    return FakeDone()
subprocess.run = fake_run
import bench
try:
    bench.main()
except SystemExit as e:
    code = e.code
print(json.dumps({"code": code, "cmd": calls[0][0], "torch_imported": "torch" in sys.modules,
                  "hip_loaded": any("dolfinx_external_operator_amd" in m for m in sys.modules),
                  "env_ipc": calls[0][1]["env"].get("HSA_ENABLE_IPC_MODE_LEGACY")}))
"""


def test_parent_spawns_torchrun_without_importing_torch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert json.loads(lines[0]) == {"metric": "m", "value": 1.0, "n_gpus": 4}      # rank 0's line relayed verbatim
    info = json.loads(lines[-1])
    assert info["code"] == 0
    assert info["torch_imported"] is False and info["hip_loaded"] is False        # no GPU call possible in the parent
    cmd = info["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert info["env_ipc"] == "0"


def test_two_rank_launch_fails_for_lack_of_devices_not_of_a_launcher():
    """On this GPU-less box the ranks start and stop with the device-count message; exit code is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch

    if torch.cuda.device_count() >= 2:
        assert res.returncode == 0, res.stderr[-2000:]
        assert json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 2
        return
    assert res.returncode != 0
    assert "needs 2 MI355X on this node" in res.stderr, res.stderr[-2000:]
    assert "torch.distributed.run --nproc-per-node" not in res.stderr.split("bench: launching")[0]   # no "wrap me in torchrun" message
    assert res.stdout.strip() == ""


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_multi_rank_path_logic_on_one_gpu(ranks):
    """The N > 1 control flow of bench.py (cell-block offsets into full-length arena buffers, the four gather modes,
    the remote-tangent rebuild and its finiteness / symmetry check, max-over-ranks timing) with two and three ranks sharing
    the one GPU of the test box and gloo collectives (`--dry-collective`): RCCL itself needs one device per rank and is
    exercised with a world of one (tests/test_round2_gpu.py) and by the driver's 8-GPU run. The line must carry what the
    first real 8-GPU run will be read by: the ranks, the backend, the per-mode status, the in-place decision, the gather
    check and how many placement candidates were probed for the FULL-length block."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                          "--master-port", str(29571 + ranks), str(ROOT / "bench.py"), "--gpus", str(ranks), "--dry-collective", "--steps", "2",
                          "--warmup", "1", "--nqp", "500000", "--no-probe"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # dry mode: no second line (the library check needs one GPU per rank)
    line = json.loads(lines[-1])
    cfg = line["config"]
    assert line["n_gpus"] == ranks and "dry_collective" in line and cfg["rccl_ranks"] == ranks
    assert set(cfg["gather_modes"]) == {"compact", "compact_pipelined", "compact_direct", "full"}
    assert cfg["points_per_gpu"] % 128 == 0 and line["value"] > 0
    assert cfg["collective_backend"] == "gloo"
    assert cfg["mode_status"] == {m: "timed" for m in ("compact", "compact_pipelined", "compact_direct", "full")}
    assert cfg["gather_in_place"] == {"ok": False, "why": "backend gloo: cloned send buffer"}
    assert line["gather_check"]["status"] == "skipped" and "dry_collective" in line["gather_check"]["why"]
    assert len(lines[-1]) < 6000
    full = json.loads((ROOT / "bench_full.json").read_text())["config"]      # the full record beside bench.py
    assert full["placement_block_bytes"] == ranks * cfg["points_per_gpu"] * 43 * 8
    # the test-sized block is below placement_min_bytes: a plain hipMalloc, nothing probed; a full-size run reports how many
    # candidates of the FULL-length block fitted 60 % of the free memory and were timed
    assert 0 <= cfg.get("placement_candidates_probed", 0) <= full["placement_candidates_requested"]


@pytest.mark.gpu
def test_single_rank_under_torchrun_prints_the_gather_check_line():
    """A world of one under torch.distributed.run with the gather on: RCCL comes up, the in-place form is the one the backend rule
    names (in the line), and the library's own RCCL path is cross-checked AFTER the result
    line: the second, final line repeats the first with `gather_check` filled in."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", "29579", str(ROOT / "bench.py"), "--gpus", "1", "--gather", "1", "--steps", "2", "--warmup", "1",
                          "--nqp", "500000", "--no-probe", "--no-cpu", "--no-e2e", "--no-secondary", "--no-traffic"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2
    first, last = lines
    assert first["gather_check"]["status"] == "pending" and last["gather_check"]["status"] == "ok", last["gather_check"]
    assert first["line"] == "final" and last["line"] == "final+gather_check"
    assert {k: v for k, v in first.items() if k not in ("gather_check", "line")} == {k: v for k, v in last.items() if k not in ("gather_check", "line")}
    assert all(len(ln) < 6000 for ln in res.stdout.splitlines())
    assert last["config"]["collective_backend"] == "nccl" and last["config"]["gather_in_place"]["ok"] is True
    assert last["gather_check"]["compact_replicas_bit_identical"] is True
