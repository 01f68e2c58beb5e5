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


def test_leg_verdicts_are_the_first_keys_of_config():
    """A record that keeps only the first 23 keys of `config` (the driver's BENCH_rNN.json does, and cuts strings at ~135
    characters) still holds every BASELINE config and every input format: roofline fraction live / alone, bit-match."""
    sys.path.insert(0, ROOT)
    import bench
    flat = {"workload": "w" * 100, "fs": 2e6, "samples_per_gpu_per_step": 1 << 30, "bursts_per_step_rank0": 1, "sharding": "none",
            "rank_sync": None, "launched_by": "plain process", "pipeline": "3", "detect_gap_ms_avg": 0.01, "detect_grid": 1,
            "retries": 0, "longrun_calls": 0, "long_pulses_per_step": 0.0, "env": {}, "numa_node": 1, "local_cpulist": "0-1",
            "bit_match_identical": True, "bit_match_bursts": 1, "roofline_frac_isolated": 0.8, "ms_per_step_untimed_ctx": 1.3}
    for key in ("cfg3", "cfg4", "cfg5", "mag2", "sc16", "sc8", "sc8g", "cu8"):
        flat.update({key + "_frac": 0.5, key + "_frac_isolated": 0.5, key + "_ms": 1.0, key + "_ms_untimed_ctx": 1.0,
                     key + "_identical": True})
    flat.update({"cfg4_8shards_msps": 6e5, "cfg4_stitched_equals_single": True})
    for f in ("fc32", "sc16", "sc8", "cu8"):
        flat.update({"hostfed_%s_pinned_msps" % f: 1.0, "hostfed_%s_pageable_msps" % f: 1.0, "hostfed_%s_pinned_vs_plain_h2d" % f: 0.99})
    out = bench.order_config(dict(flat), 1)
    assert out == flat and len(out) == len(flat)                      # nothing lost, nothing invented
    first = list(out)[:23]
    want = (["workload", "samples_per_gpu_per_step", "cfg4_8shards_msps", "hostfed_fc32_pinned_vs_plain_h2d"]
            + ["cfg%d_%s" % (c, k) for c in (3, 4, 5) for k in ("frac", "frac_isolated", "identical")]
            + ["%s_%s" % (f, k) for f in ("mag2", "sc16", "sc8", "sc8g", "cu8") for k in ("frac", "identical")])
    assert sorted(first) == sorted(want)
    # the diagnostics come behind them
    assert list(out).index("detect_gap_ms_avg") > 22 and list(out).index("local_cpulist") > 22
    # N > 1: the seam verdict and config 4's legs lead
    multi = bench.order_config({"workload": "w", "fs": 2e6, "samples_per_gpu_per_step": 1, "numa_node": 0, "seams_identical": True,
                                "stitch_fallbacks": 0, "cfg4_weak_msps": 1.0, "cfg4_strong_msps": 1.0}, 8)
    assert list(multi)[:6] == ["workload", "samples_per_gpu_per_step", "seams_identical", "stitch_fallbacks", "cfg4_weak_msps", "cfg4_strong_msps"]


def test_headline_workload_string_survives_a_135_character_cut():
    src = open(os.path.join(ROOT, "bench.py")).read()
    line = [l for l in src.splitlines() if '"workload": "synthetic %g Msps %s IQ' in l][0]
    longest = line.split('"workload": "')[1].rsplit('"', 1)[0] % (20, "f32 |IQ|^2", 1000, "mixed-DF 3-25 dB",
                                                                  "AWGN 2e-3", 0.01, 30)
    assert len(longest) < 130, len(longest)
