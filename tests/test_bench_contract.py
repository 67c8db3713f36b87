"""bench.py's stdout contract: ONE line, strict JSON, < 4 KB, carrying the driver's keys + roofline + cpu_baseline
(round 5's line grew to 23 KB and the driver could not parse it).  Also: `bench.py --gpus N` builds the right child command
without touching the GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402  (module level imports the standard library only)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU = ("value", "unit", "cores", "kind", "sample")


def _leg(ms, frac=None, depth=0):
    d = {"rows": 1000000, "ms": ms, "ms_spread": {"min": ms, "median": ms, "max": ms, "reps": 10}, "bound": "hbm",
         "achieved": 1234.5678, "peak": 8000.0, "unit": "GB/s", "shape": "x" * 200, "note": "n" * 600,
         "traffic_source": "t" * 300, "cpu_rows_per_s": 12.3, "cpu_form": "c" * 150}
    if frac is not None:
        d["frac"] = frac
    if depth:
        d["inner_a"] = _leg(ms / 2, 0.5)
        d["inner_b"] = _leg(ms / 3)
    return d


def fat_record(workload="cfg2", n_stages=40):
    """A record at least as large as anything bench.py assembles: long strings everywhere, per-step lists, nested stages."""
    rec = {
        "metric": "OOD scores/sec, LaREM 16-MC PCA-256 (ResNet-18 layer4 latent 512x4x4)", "value": 59047127.04321,
        "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.16940001, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "ms_per_step_stats": {"per_step": [0.1619] * 2000, "how": "h" * 400},
        "clock_ghz_observed": {"before": {"ghz": 2.43}, "after": {"ghz": 2.36}, "how": "w" * 500},
        "config": {"workload": "CIFAR10 ResNet-18 LaREM: 16 MC samples, 512-d latent -> PCA-256, 10000 test images per GPU" + "!" * 300,
                   "images_per_gpu": 10000, "mc_samples": 16, "latent": [512, 4, 4], "pca_components": 256,
                   "row_blocks_per_step": 1, "input_dtype": "f32", "draws": "d" * 200, "input_sets_rotated": 3,
                   "input_bytes_per_set": 337920000, "pipelining": "none (one stream)", "clock_warmup_s": 1.0,
                   "setup": {"mode": "device", "fit_s": 0.951, "what": "s" * 200}, "keep_flag_table": "k" * 100,
                   "gather": "none (1 GPU)", "a": 1, "b": 2, "c": 3, "d": 4},
        "roofline": {"bound": "hbm", "binding_limit": "valu-issue", "bound_frac": 0.85, "kernel": "mc_entropy_kernel" + "<" * 300,
                     "achieved": 3060.8123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.38261234, "traffic": 392115488,
                     "traffic_source": "p" * 400, "algorithmic_bytes_per_launch": 338000000, "avg_launch_ms": 0.11041234,
                     "launches_per_step": 1, "launches_timed": 40, "valu_issue": {"note": "v" * 900},
                     "launch_ms": {"min": 0.1, "max": 0.2, "where": "w" * 300}},
        "api_level": {"entry": "e" * 500, "value": 6.0e7},
        "cpu_baseline": {"value": 83.27, "unit": "images/s", "cores": 1, "kind": "port", "sample": "s" * 700},
        "cpu_baseline_all_cores": {"value": 1286.83, "cores": 16, "sample": "s" * 700},
        "parity": {"max_rel_err_from_device_samples": 5.1e-15, "max_rel_err_from_latents": 5.5e-15, "auroc_gpu": 0.8727099895477295,
                   "auroc_oracle": 0.8727099895477295, "counter_draws": {"note": "n" * 900}},
        "stages": {f"leg_number_{i}_with_a_long_name": _leg(1.0 + i, 0.1 * (i % 9), depth=i % 2) for i in range(n_stages)},
    }
    rec["stages"]["larex_eval"] = {"seconds_device_resident": 1.799, "table": {f"row {i}": [0.1] * 5 for i in range(90)}}
    if workload == "cfg3":
        rec.update(metric="OOD scores/sec, Mahalanobis + Energy + kNN(k=50) on synthetic 1M x 2048 features", unit="rows/s",
                   scaling="strong", dtype="f64 (Mahalanobis) / f32 (Energy, kNN)" + "; " + "z" * 600)
        rec["roofline"] = {"bound": "mfma", "kernel": "knn_dist_bf16_kernel" + " (" + "k" * 300 + ")", "achieved": 1458.58,
                           "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.5834, "traffic": None, "avg_stage_ms": 421.2}
    if workload == "larex_eval":
        rec.update(metric="OOD scores/sec through the evaluation harness loop", unit="rows scored/s", roofline=None)
        rec["cpu_baseline"] = {"value": None, "unit": "s (bounded subset, see sample)", "cores": 16, "kind": "port", "sample": "q" * 400}
        rec["larex_eval"] = {"table": {f"row {i}": [0.1] * 5 for i in range(90)}}
        del rec["stages"], rec["parity"]
    if workload == "baselines_eval":
        rec.update(metric="OOD scores/sec through the baselines harness loop", unit="rows scored/s", roofline=None)
        rec["cpu_baseline"] = {"value": None, "unit": "s (bounded subset, see sample)", "cores": 16, "kind": "port", "sample": "q" * 400}
        rec["baselines_eval"] = {"seconds_per_baseline": {f"baseline {i}": 0.1 for i in range(12)}, "notes": "n" * 20000}
        rec["parity"] = {"max_rel_err": 2.3e-6, **{f"max_rel_err_b{i}": 1e-7 * i for i in range(12)}}
        del rec["stages"]
    return rec


@pytest.mark.parametrize("workload", ["cfg2", "cfg3", "larex_eval", "baselines_eval"])
@pytest.mark.parametrize("n_stages", [0, 12, 400])
def test_contract_line_is_small_strict_json_with_the_required_keys(workload, n_stages):
    full = fat_record(workload, n_stages)
    assert len(json.dumps(full)) > 20000  # the detail record is the large one
    line = bench.contract_line(full)
    assert "\n" not in line
    assert len(line.encode()) < 4096 == bench.CONTRACT_MAX_BYTES

    def no_constants(name):  # NaN / Infinity are not JSON
        raise AssertionError(f"non-standard JSON constant {name} on the contract line")

    rec = json.loads(line, parse_constant=no_constants)
    for k in REQUIRED:
        assert k in rec, k
    assert rec["value"] == pytest.approx(full["value"], rel=1e-6) and rec["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-3)
    assert rec["steps"] == 20 and rec["warmup"] == 5 and rec["n_gpus"] == 1 and rec["vs_baseline"] is None
    assert set(rec["config"]) - {"workload"} and len(rec["config"]) <= 9 and "workload" in rec["config"]
    assert all(not isinstance(v, (dict, list)) for v in rec["config"].values())
    if workload == "baselines_eval":
        assert rec["roofline"] is None and rec["parity"]["max_rel_err"] == pytest.approx(2.3e-6)
    elif workload != "larex_eval":
        for k in ROOFLINE:
            assert k in rec["roofline"], k
        assert rec["roofline"]["bound"] in ("hbm", "mfma")
        assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
        assert set(rec["parity"]) == {"max_rel_err", "auroc_gpu", "auroc_oracle"}
        if n_stages:
            assert all(isinstance(v, list) and len(v) == 2 for v in rec["stages"].values())
            assert rec["stages"]["leg_number_1_with_a_long_name.inner_a"][1] == 0.5
            assert rec.get("stages_truncated", False) == (n_stages == 400)
        if n_stages == 12:
            assert rec["stages"]["larex_eval"] == [1799.0, None]
    else:
        assert rec["roofline"] is None
    for k in CPU:
        assert k in rec["cpu_baseline"], k
    assert len(rec["cpu_baseline"]["sample"]) <= 120
    assert rec["detail"] == bench.DETAIL_FILE


def test_contract_line_drops_non_finite_numbers():
    full = fat_record("cfg2", 2)
    full["roofline"]["achieved"] = float("nan")
    full["value"] = float("inf")
    rec = json.loads(bench.contract_line(full))
    assert rec["roofline"]["achieved"] is None and rec["value"] is None


def test_emit_record_writes_detail_beside_bench_and_one_stdout_line(tmp_path, monkeypatch, capfd):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    full = fat_record("cfg2", 40)
    text = bench.emit_record(full)
    out, err = capfd.readouterr()
    assert out == text + "\n" and out.count("\n") == 1
    assert err.startswith("bench detail: ") and len(err) > 20000
    for d in (tmp_path, tmp_path / "gpurun_out"):
        assert json.load(open(d / bench.DETAIL_FILE)) == full


def test_gpus_n_builds_the_child_command_without_touching_the_gpu(monkeypatch):
    """bench.py --gpus 8 without a launcher starts torch.distributed.run as a CHILD process (never an exec) before any GPU call."""
    calls = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"], calls["kw"] = cmd, env, kw
        return Done()

    import torch

    def no_gpu(*a, **k):
        raise AssertionError("the parent process touched the GPU before launching its ranks")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", no_gpu)
    monkeypatch.setattr(torch.cuda, "set_device", no_gpu)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7  # the child's exit code is handed back
    cmd, env = calls["cmd"], calls["env"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "OMP_NUM_THREADS" in env
    assert "WORLD_SIZE" not in env  # torch.distributed.run sets the rank environment itself


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)
