"""bench.py --gpus N (N > 1) as a launcher, on the CPU: both launch forms (self-launched, torch.distributed.run), the wall-clock budget,
the tiers, and the JSON line it must always print.  The `uspmv` rank processes are replaced by tests/helpers/fake_uspmv.py
(USPMV_BENCH_EXE) -- what is under test here is the process handling around them; the real harness runs in tests/test_dist_native_gpu.py."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "helpers", "fake_uspmv.py")


def run_bench(mode, launch, extra=(), n=2, budget=40, port=29650):
    env = dict(os.environ, USPMV_BENCH_EXE=FAKE, FAKE_USPMV_MODE=mode)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    base = ["bench.py", "--gpus", str(n), "--steps", "5", "--warmup", "2", "--budget-s", str(budget), *extra]
    if launch == "self":
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port)] + base
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=budget + 120)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly ONE JSON line on stdout, got {len(lines)}:\n{r.stdout}\n{r.stderr[-2000:]}"
    return r.returncode, json.loads(lines[0]), time.time() - t0


@pytest.mark.parametrize("launch", ["self", "torchrun"])
def test_healthy_run_reports_every_rank_and_the_strong_scaling_efficiency(launch):
    rc, d, _ = run_bench("ok", launch, n=4, port=29651)
    assert rc == 0 and d["value"] == 6000.0 and d["n_gpus"] == 4 and d["scaling"] == "strong" and d["y_checked"] is True
    c = d["config"]
    assert c["rccl_nranks"] == 4 and c["exchange"] == "rccl" and [r["rank"] for r in c["per_rank"]] == [0, 1, 2, 3]
    assert all(k in c["per_rank"][2] for k in ("n_local", "n_halo", "local_kernel_ms", "local_kernel_GBs"))
    assert ("itself" in c["launcher"]) == (launch == "self")
    s = d["single_gpu_same_matrix"]
    assert s["ms_per_step"] == 1.6 and "304x304x304" in s["cmd"]
    assert d["speedup_vs_single_gpu"] == pytest.approx(1.6 / 0.25) and d["strong_scaling_efficiency"] == pytest.approx(1.6 / (4 * 0.25))
    assert d["roofline"]["slowest_rank"]["rank"] == 3
    assert d["weak_scaling"]["value"] == 6000.0 and "253x253x1012" in d["weak_scaling"]["workload"]
    assert [t["ok"] for t in d["budget"]["tiers"]] == [True, True] and "fallback_reason" not in d


@pytest.mark.parametrize("launch", ["self", "torchrun"])
def test_hung_children_are_killed_at_the_budget_and_the_line_says_so(launch):
    rc, d, wall = run_bench("hang", launch, budget=30, port=29652)
    assert rc != 0 and d["value"] is None and "killed at the time limit" in d["error"] and "waiting for the other ranks" in d["error"]
    assert wall < 30 + 25, wall                       # the whole run ends with its budget, not with the driver's patience
    assert d["budget"]["used_s"] <= 30 + 10
    # a communicator that never came up: the eager tier is skipped, the host-staged tier is tried while budget is left
    assert not any(t["tier"].startswith("2:") for t in d["budget"]["tiers"])


def test_capture_crash_falls_to_eager_steps_in_fresh_children():
    rc, d, _ = run_bench("capture_crash", "self", extra=["--no-second-line"])
    assert rc == 0 and d["value"] == 6000.0 and "eager steps instead of graph replay" in d["fallback_reason"]
    assert "eager C++ steps" in d["config"]["step"] and [t["ok"] for t in d["budget"]["tiers"]] == [False, True]


@pytest.mark.parametrize("launch", ["self", "torchrun"])
def test_rccl_down_falls_to_the_host_staged_exchange_and_labels_it(launch):
    rc, d, _ = run_bench("rccl_down", launch, extra=["--no-second-line"], port=29653)
    assert rc == 0 and d["config"]["exchange"] == "host" and d["config"]["rccl_nranks"] == 0
    assert "host-staged exchange" in d["config"]["step"] and "staged through host memory instead of RCCL" in d["fallback_reason"]
    assert [t["tier"][0] for t in d["budget"]["tiers"]] == ["1", "3"]


def test_everything_failing_still_prints_the_line_with_the_reason():
    rc, d, wall = run_bench("fail", "self")
    assert rc != 0 and d["value"] is None and "made to fail" not in d["error"] and "rc 1" in d["error"] and wall < 20
    assert "error" in d["single_gpu_same_matrix"]
