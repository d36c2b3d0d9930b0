"""The vectorised workload generators `bench.py` uses (tools/workloads.py) against the scalar definitions the parity tests
use (tests/signals.py, tests/golden_inputs.py), sample for sample; and bench.py's own launcher: `python bench.py --gpus 2`
with no RANK in the environment must start two ranks itself (gloo here, RCCL on a node) and report n_gpus = 2."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from golden_inputs import cfg1_pcm, cfg2_pcm  # noqa: E402
from signals import xorshift32_noise  # noqa: E402


def test_vectorised_generators_equal_the_scalar_definitions():
    import workloads
    n = 3000
    bank = workloads.cfg2_bank(1022, 4, n)     # crosses stream 1024: phase 2 pi s / 64 wraps, seeds do not
    for i, s in enumerate(range(1022, 1026)):
        assert np.array_equal(bank[i], cfg2_pcm(s, n)), s
    assert np.array_equal(workloads.cfg1_pcm(n), cfg1_pcm(n))
    seeds = [0, 1, 0x9E3779B9, 0xFFFFFFFF]
    got = workloads.xorshift32_noise_bank(seeds, 500, 1e-3)
    for i, seed in enumerate(seeds):
        assert np.array_equal(got[i], xorshift32_noise(seed, 500, 1e-3))


def run_bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_launcher_spawns_the_ranks_itself_gloo_world_size_2():
    r, line = run_bench(["--gpus", "2", "--dry-run", "--streams", "5"], {"OMX_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1   # ONE line, from rank 0
    assert line["n_gpus"] == 2 and line["dry_run"] and line["config"]["workload"] == "cfg5" and line["config"]["gathered_rows"] == 10


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
