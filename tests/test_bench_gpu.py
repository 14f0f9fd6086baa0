"""bench.py itself on the GPU box, on a small configuration: the eager (`--no-graph`) path verifies the outputs it has just computed
(it probed stale tensors and failed spuriously; the counter-pass scripts tools/pmc_*.sh run exactly this mode)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "vitb512", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline", *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_no_graph_verifies_the_eager_outputs():
    d = _bench("--no-graph", "--chains", "1")
    assert d["config"]["hip_graph"] is False
    assert d["metric"].endswith("ViT-B") and "NOT the BASELINE headline" in d["config"]["workload"]
    assert d["verified"]["replayed_graph_vs_reference_golden_probes_max_rel"] <= 1e-3
    assert d["value"] > 0 and d["vs_baseline"] is None


def test_bench_graph_and_chains_verify_bitwise():
    d = _bench()
    assert d["config"]["hip_graph"] is True
    assert d["verified"]["graph_replay_equals_eager_bitwise"] is True
    assert d["verified"]["replayed_graph_vs_reference_golden_probes_max_rel"] <= 1e-3
    assert d["config"]["attention_blocks"]["f16"] + d["config"]["attention_blocks"]["b3"] == 12
