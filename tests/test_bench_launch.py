"""bench.py's multi-rank path -- what the driver's SCALE run takes -- under test (reference main.py:102-106 is a
commented TODO, SURVEY 8e is the contract): the self-launcher, the pre-flight and the N > 1 JSON fields.
GPU cases run the real thing with two ranks sharing the box's one GPU over gloo (VQA_BENCH_OVERSUBSCRIBE=1: a rehearsal of the
plumbing, not a scaling number); the CPU case needs no GPU at all."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=600):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    # a fresh child: the test process has touched the GPU and must not exec or fork into GPU work itself
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout, cwd=ROOT)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_gpu_prints_the_scale_line():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16", "--no-extras", "--no-cpu-baseline"],
             env={"VQA_BENCH_OVERSUBSCRIBE": "1", "VQA_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "QA-pairs/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["precision"] == "exact" and out["dtype"] == "f32"
    assert abs(out["value"] - 2 * 16 / (out["ms_per_step"] * 1e-3)) < 0.01 * out["value"]       # whole-job rate over both ranks
    assert out["dist_backend"] == "gloo" and out["rccl_world_size"] == 2
    assert [e["rank"] for e in out["ranks"]] == [0, 1] and out["distinct_gpus"] == 1             # (both on this box's one GPU)
    assert abs(out["allreduce_payload_mb"] - 48.7) < 0.2 and out["allreduce_buckets"] >= 3       # 12.18 M fp32 gradients
    ex = out["exchange"]
    assert set(ex["ms_per_step"]) >= {"allreduce", "direct", "none"}
    assert all(ex["ms_per_step"][k] is not None and ex["ms_per_step"][k] > 0 for k in ("allreduce", "direct", "none")), ex
    assert ex["expected_ms"]["ring_one_link_ms"] > 0 and ex["expected_ms"]["one_shot_all_links_ms"] > 0
    assert ex["preflight"]["ranks"] == 2 and ex["world_size"] == 2


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2), "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], timeout=200)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "needs %d visible GPUs, found %d" % (n + 2, n) in r.stderr and not r.stdout.strip()


def test_self_launch_reports_a_rank_that_dies_at_startup(tmp_path, monkeypatch):
    """One rank exits at start-up while the others would sit in the rendezvous: self_launch must stop them and return non-zero
    well inside a minute (it polls ALL ranks), not wait for a rendezvous timeout.  The ranks here are a stub script."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text("import os, sys, time\n"
                    "if os.environ['RANK'] == '1':\n"
                    "    sys.exit(3)\n"
                    "time.sleep(300)\n")
    monkeypatch.setattr(bench, "__file__", str(stub))
    monkeypatch.setenv("VQA_BENCH_OVERSUBSCRIBE", "1")
    t0 = time.time()
    rc = bench.self_launch(argparse.Namespace(gpus=3))
    assert rc != 0 and time.time() - t0 < 60.0
