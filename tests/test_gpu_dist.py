"""GPU: the data-parallel path on RCCL (torch.distributed backend "nccl"), each rank a fresh child process
(tests/_dist_child.py) -- the reference has only a TODO here (main.py:71, :102-106).

world 1 (always, one GPU): an nccl group of size 1 on cuda:0, two `Trainer.step`s of the attention model with the
    `GradReducer` attached, all-reduce and one-shot ("direct") exchange: same losses / parameters as the run without
    a process group (averaging over one rank is the identity), W_b discovered as unused, several buckets.
world 2 (when >= 2 GPUs are visible): half-batch ranks on the HIP co-attention + MLP subgraph; the averaged
    gradients equal the single-process full-batch gradients and the ranks stay in lockstep."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_ranks(tmp_path, mode, world, exchange, tag):
    port = _free_port()
    outs = [str(tmp_path / ("%s_%s_%d.npz" % (tag, exchange, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_child.py"), "--mode", mode, "--world", str(world),
                               "--rank", str(r), "--port", str(port), "--exchange", exchange, "--out", outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    return [dict(np.load(o)) for o in outs]


@pytest.mark.timeout(900)
def test_nccl_world1_trainer_steps(tmp_path):
    ref = _run_ranks(tmp_path, "net", 1, "none", "ref")[0]
    for exchange in ("allreduce", "direct"):
        r = _run_ranks(tmp_path, "net", 1, exchange, "w1")[0]
        assert int(r["n_buckets"][0]) >= 1 and int(r["payload"][0]) > 1e6
        assert np.all(np.isfinite(r["losses"])) and np.allclose(r["losses"], ref["losses"], rtol=1e-5, atol=1e-6), \
            (exchange, r["losses"], ref["losses"])
        for k in ref:
            if k.startswith("p."):
                assert np.allclose(r[k], ref[k], rtol=1e-4, atol=2e-4), (exchange, k)    # two Adam steps of lr 1e-4
        assert not any(k.startswith("p.co_attention.W_b") and not np.array_equal(r[k], ref[k]) for k in ref)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["allreduce", "direct"])
def test_nccl_world2_equals_full_batch(tmp_path, exchange):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    full = _run_ranks(tmp_path, "sub", 1, "none", "full")[0]
    r0, r1 = _run_ranks(tmp_path, "sub", 2, exchange, "w2")
    assert int(r0["n_buckets"][0]) > 1
    for k in full:
        if k.startswith("g."):
            scale = max(1e-6, float(np.abs(full[k]).max()))
            assert np.array_equal(r0[k], r1[k]), k                                       # identical on every rank
            assert np.abs(r0[k] - full[k]).max() <= 2e-5 * scale + 1e-7, k               # == full-batch gradient
        elif k.startswith("p."):
            assert np.array_equal(r0[k], r1[k]), k                                       # lockstep
            if k.endswith("w_v.bias") or k.endswith("w_q.bias"):
                continue                 # analytically zero gradient: Adam turns rounding noise into +-lr steps
            assert np.allclose(r0[k], full[k], atol=2e-5), k
