"""Shows round 5's bug on the installed runtime (GPU box): the scenario of
tests/test_gpu_train_graph.py::test_split_attention_backward_replays_from_a_graph with the destinations of the split launches
zeroed by hipMemsetAsync (GRL_ZERO_MEMSET=1, what round 5 shipped) and by the zero-fill kernel (default)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for mode in ("1", "0"):
    env = dict(os.environ, GRL_ZERO_MEMSET=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        "tests/test_gpu_train_graph.py::test_split_attention_backward_replays_from_a_graph"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    tail = [l for l in r.stdout.splitlines() if "differs from" in l or "passed" in l or "failed" in l or "isfinite" in l or "AssertionError" in l]
    print(f"GRL_ZERO_MEMSET={mode}: rc {r.returncode}  " + " | ".join(t.strip() for t in tail[-3:]))
