"""bench.py's N > 1 control flow on the one-GPU box: `--rehearse-shared-gpu` puts every rank on cuda:0 and moves the
exchange through the library communicator's shared-memory rehearsal transport, so the launcher (bench.py starting its own ranks), the rank processes under torch.distributed.run,
the weak and the carried strong measurement, verification against the simple kernel and the assembly of the ONE JSON
line all run -- everything of an 8-GPU run but RCCL and xGMI (those: tests/test_gpu_comm.py with one rank).  Its first run
found the weak-scaling line referencing a run it had already freed: the command the driver issues for N > 1 crashed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--rehearse-shared-gpu", "--reads", "20000", "--steps", "3", "--warmup", "1", "--verify", "--cpu-sample", "0", "--spinup-ms", "0"]


def line_of(cmd):
    env = dict(os.environ, MSQ_BENCH_WATCHDOG="200")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-500:]
    return json.loads(lines[0])


def check(d, n, scaling):
    assert d["n_gpus"] == n and d["scaling"] == scaling and d["steps"] == 3 and d["value"] > 0
    assert d["verified_against_simple_kernel"] is True
    assert "REHEARSAL" in d["config"]["exchange"] and "rehearsal" in d
    assert d["roofline"]["kernel_ms"] > 0 and d["roofline"]["frac"] > 0
    if scaling == "weak":
        assert d["strong_scaling"]["reads_per_gpu"] == 20000 // n and d["strong_scaling"]["value"] > 0
        assert d["config"]["workload"].startswith("pol_1m: 20000 reads/GPU")


def test_own_launcher_weak_line_carries_the_strong_case():
    d = line_of([sys.executable, "bench.py", "--gpus", "2"] + ARGS)
    check(d, 2, "weak")
    assert "started 2 rank processes itself" in d["config"]["launcher"]


def test_under_torch_distributed_run_as_the_driver_starts_it():
    d = line_of([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29633", "bench.py", "--gpus", "2"] + ARGS)
    check(d, 2, "weak")


def test_strong_scaling_line_and_a_wide_window_over_three_ranks():
    d = line_of([sys.executable, "bench.py", "--gpus", "3", "--scaling", "strong", "--workload", "hiv_mix_200k"] + ARGS)
    check(d, 3, "strong")
