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
def test_two_rank_path_logic_on_one_gpu():
    """The N > 1 control flow of bench.py (cell-block offsets into full-length arena buffers, the four gather modes,
    the remote-tangent rebuild and its finiteness / symmetry check, max-over-ranks timing) with two ranks sharing the
    one GPU of the test box and gloo collectives (`--dry-collective`): RCCL itself needs one device per rank and is
    exercised with a world of one (tests/test_round2_gpu.py) and by the driver's 8-GPU run."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29571", str(ROOT / "bench.py"), "--gpus", "2", "--dry-collective", "--steps", "2",
                          "--warmup", "1", "--nqp", "500000", "--no-probe"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "dry_collective" in line and line["config"]["rccl_ranks"] == 2
    assert set(line["config"]["gather_modes"]) == {"compact", "compact_pipelined", "compact_direct", "full"}
    assert line["config"]["points_per_gpu"] % 128 == 0 and line["value"] > 0
