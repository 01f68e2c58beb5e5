"""bench.py's launch contract, the part that needs no GPU: `--gpus N` is a promise about the line's n_gpus."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=120, cwd=ROOT)


def test_world_size_that_is_not_gpus_is_refused():
    r = _run(["--gpus", "8"], WORLD_SIZE="1", RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=1 but --gpus 8" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "2"], WORLD_SIZE="4", RANK="0")
    assert r.returncode == 2 and not r.stdout.strip()
    r = _run(["--gpus", "0"])
    assert r.returncode == 2


def test_plain_launch_with_n_gt_1_goes_through_the_launcher(monkeypatch):
    """Without WORLD_SIZE, --gpus N > 1 re-launches through torch.distributed.run with N ranks on 127.0.0.1 and passes the
    arguments through (the command is inspected, not run: no GPU here)."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr("subprocess.call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main() == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["ADSB_BENCH_SPAWNED"] == "1"


def test_unknown_adsb_environment_variable_is_refused():
    r = _run(["--gpus", "1"], ADSB_SOME_TUNING_KNOB="1")
    assert r.returncode == 2 and "ADSB_SOME_TUNING_KNOB" in r.stderr
