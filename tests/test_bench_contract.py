"""bench.py's driver contract: `python bench.py --gpus N` starts N ranks by itself (no launcher), the roofline block uses the kernels'
own bytes for `frac`, and the PMC traffic figure is only quoted for the kernel sources it was measured on."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_launches_that_many_ranks(n):
    r, line = run_bench("--gpus", str(n), "--steps", "2", "--warmup", "1", env={"WL_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert line == {"dry_run": True, "n_gpus": n, "steps": 2, "warmup": 1, "size": 512}
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1       # ONE line, from rank 0


def test_world_size_must_match_gpus_flag():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", WL_BENCH_DRY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=e, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r, line = run_bench("--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and line is None and "no CPU fallback" in (r.stderr + r.stdout)


def _fake_prof(a_ms, b_ms):
    z = {"launches": 0, "avg_ms": None, "total_ms": 0.0}
    prof = {k: dict(z) for k in ("gs_sweep", "smooth", "jacobi", "conv_diff", "residual", "bdim", "prolong_increment", "coarse_levels", "mom_step")}
    prof["gsrb_A"] = {"launches": 40, "avg_ms": a_ms, "total_ms": 40 * a_ms}
    prof["gsrb_B"] = {"launches": 40, "avg_ms": b_ms, "total_ms": 40 * b_ms}
    return prof


def test_roofline_frac_is_own_bytes_not_operation_bytes(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    ncell = 512.0**3
    # round-1 kernel pair: A updates x itself (the library reports which kernel applied the V-cycle's x increment: wl_sim_counter "xdefer")
    roof = bench.build_roofline(_fake_prof(0.86, 0.75), ncell, True, 2, 512, use_traffic=False, xdefer=0)
    own = (20.5 + 20.0) * ncell / 1.61e-3 / 1e9
    assert roof["bytes_per_cell"] == 40.5 and abs(roof["achieved"] - own) < 1e-6 * own
    assert abs(roof["frac"] - own / 8000.0) < 1e-9 and 0.40 < roof["frac"] < 0.44          # 0.42, not 0.80
    assert roof["op_equivalent"]["bytes_per_cell"] == 76.5 and 0.78 < roof["op_equivalent"]["frac"] < 0.82
    assert roof["traffic"] is None and roof["traffic_frac"] is None
    # default pair kernels: `x += ω·x_c↓` is applied by kernel B — the pair owns 8 B/cell less, and frac is quoted on what it owns
    roof = bench.build_roofline(_fake_prof(0.49, 0.70), ncell, True, 2, 512, use_traffic=False, xdefer=1)
    assert roof["x_increment_deferred_to_B"] and roof["bytes_per_cell"] == 33.0
    assert roof["kernels"]["A"]["bytes_per_cell"] == 12.5 and roof["kernels"]["B"]["bytes_per_cell"] == 20.5
    assert abs(roof["frac"] - 33.0 * ncell / 1.19e-3 / 1e9 / 8000.0) < 1e-9
    assert roof["op_equivalent"]["bytes_per_cell"] == 76.5                                  # the reference's operations are the same
    # traffic is quoted only when the PMC pass was taken on the same kernel sources
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    csrc = tmp_path / "waterlily.jl_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in bench.KERNEL_CONFIG_FILES:
        (csrc / f).write_text("v1 " + f)
    tj = {"size": 512, "source": "x", "kernel_config_sha": bench.kernel_config_sha(),
          "kernels": {"A": {"hbm_bytes_per_launch": 3.2e9}, "B": {"hbm_bytes_per_launch": 3.5e9}}}
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(tj))
    roof = bench.build_roofline(_fake_prof(0.86, 0.75), ncell, True, 2, 512, use_traffic=True)
    assert roof["traffic"] == 6.7e9 and abs(roof["traffic_frac"] - 6.7e9 / 1.61e-3 / 8e12) < 1e-9
    (csrc / bench.KERNEL_CONFIG_FILES[0]).write_text("v2: the kernel changed after the PMC pass")
    roof = bench.build_roofline(_fake_prof(0.86, 0.75), ncell, True, 2, 512, use_traffic=True)
    assert roof["traffic"] is None and roof["traffic_frac"] is None


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu_prints_n_gpus_2():
    r, line = run_bench("--gpus", "2", "--size", "128", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        env={"WL_BENCH_ONE_GPU": "1", "WL_DIST_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["config"]["transport"] == "CallbackComm" and line["value"] > 0
    assert line["config"]["parallelism"] == "zslab2" and line["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_single_gpu_line_carries_roofline_probe_and_cpu_baseline():
    """the default single-GPU line at a small size: the contract keys, the roofline on the kernels' own bytes, launch and path counters, and the CPU
    baseline (port) with its serial companion"""
    r, line = run_bench("--size", "128", "--steps", "5", "--warmup", "2", "--cpu-budget", "3", timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["dtype"] == "f32" and line["vs_baseline"] is None and line["value"] > 0
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert "probe" not in roof                                        # (round 3: the self-made comparator is gone from the line; tools/probe holds the scans)
    cfg = line["config"]
    assert cfg["launches_per_step"] > 10 and cfg["resjac"] + cfg["resjac_redo"] >= 0 and cfg["mean_pois_n"] >= 1
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["serial"]["cores"] == 1
