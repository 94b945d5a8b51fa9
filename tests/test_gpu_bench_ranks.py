"""`bench.py --gpus 2` end to end on the one-GPU box: the launcher starts two ranks (gloo rendezvous, both ranks on the
card -- RCCL refuses two ranks per GPU), the scenes of BASELINE configs[3] are sharded scene-wise
(/root/reference/tbsim/utils/guidance_loss.py:31-37,252-262 fix the scene as the unit), every rank samples its own
scenes, the plans are all-gathered -- and, noise being seeded by GLOBAL scene index, the gathered plans are bit-identical
to the one-rank run's.  The child processes are started before this test touches the GPU-owning state of its own
process (plain subprocesses, no exec)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus):
    env = dict(os.environ, TRACE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "cfg4_filter_only",
           "--scaling", "strong", "--scenes", "2", "--sub-scenes", "8", "--steps", "1", "--warmup", "0",
           "--no-roofline", "--no-cpu-baseline", "--digest"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, p.stdout[-1000:]
    return json.loads(lines[0])


def test_two_rank_bench_equals_the_one_rank_bench():
    one, two = _bench(1), _bench(2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for r in (one, two):
        assert r["value"] > 0 and r["value"] == r["value"] and r["scaling"] == "strong"
        assert r["config"]["plan_rows"] == 16 * 32                   # 8 x 2 scenes x 32 agents, all of them on rank 0
    assert one["config"]["plan_sha1"] == two["config"]["plan_sha1"]
