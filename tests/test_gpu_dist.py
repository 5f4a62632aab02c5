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


def _run_ranks(tmp_path, mode, world, exchange, tag, extra=(), env=None):
    port = _free_port()
    outs = [str(tmp_path / ("%s_%s_%d.npz" % (tag, exchange, r))) for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_child.py"), "--mode", mode, "--world", str(world),
                               "--rank", str(r), "--port", str(port), "--exchange", exchange, "--out", outs[r], *extra],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(os.environ, **(env or {}))) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    return [dict(np.load(o)) for o in outs]


GLOO_ONE_GPU = ("--backend", "gloo", "--same-gpu")


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
@pytest.mark.parametrize("exchange", ["allreduce", "direct", "p2p"])
def test_nccl_world2_equals_full_batch(tmp_path, exchange):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    full = _run_ranks(tmp_path, "sub", 1, "none", "full")[0]
    _check_against_full_batch(full, _run_ranks(tmp_path, "sub", 2, exchange, "w2"))


def _check_against_full_batch(full, ranks):
    r0 = ranks[0]
    assert int(r0["n_buckets"][0]) > 1
    for k in full:
        if k.startswith("g."):
            scale = max(1e-6, float(np.abs(full[k]).max()))
            for r in ranks[1:]:
                assert np.array_equal(r0[k], r[k]), k                                    # identical on every rank
            assert np.abs(r0[k] - full[k]).max() <= 2e-5 * scale + 1e-7, k               # == full-batch gradient
        elif k.startswith("p."):
            for r in ranks[1:]:
                assert np.array_equal(r0[k], r[k]), k                                    # lockstep
            if k.endswith("w_v.bias") or k.endswith("w_q.bias"):
                continue                 # analytically zero gradient: Adam turns rounding noise into +-lr steps
            # three Adam steps of lr 1e-3: an element whose gradient is rounding noise may move by lr per step in either
            # run; everywhere else the parameters agree closely
            diff = np.abs(r0[k] - full[k])
            assert diff.max() <= 3.1e-3 and diff.mean() <= 2e-5, (k, float(diff.max()), float(diff.mean()))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_p2p_exchange_ranks_sharing_one_gpu(tmp_path, world):
    """The one-shot exchange over peer-mapped buckets (exchange="p2p": HIP IPC + csrc/p2p.hip, no collective on the data
    path) with `world` processes on this ONE GPU (gloo for the handles and the phase barriers): the averaged gradients
    equal the single-process full-batch gradients, identical on every rank, three Adam steps in lockstep."""
    full = _run_ranks(tmp_path, "sub", 1, "none", "full")[0]
    ranks = _run_ranks(tmp_path, "sub", world, "p2p", "p2p%d" % world, extra=("--backend", "gloo", "--same-gpu"))
    _check_against_full_batch(full, ranks)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["allreduce", "direct"])
def test_world2_ranks_sharing_one_gpu_equal_full_batch(tmp_path, exchange):
    """The gloo twins of the nccl world-2 cases above, which have never had two GPUs to run on: two ranks on
    this ONE GPU, CUDA buckets -- the hooked bucket path, the all-reduce and the one-shot "direct" exchange (its collectives on
    a host copy of the bucket: gloo has no all-to-all on CUDA tensors; same shard arithmetic, same order of the sums) give
    averaged gradients equal to the single-process full-batch gradients, identical on both ranks, three Adam steps in lockstep."""
    full = _run_ranks(tmp_path, "sub", 1, "none", "full")[0]
    ranks = _run_ranks(tmp_path, "sub", 2, exchange, "g2", extra=GLOO_ONE_GPU)
    assert all(str(r["exchange_used"][0]) == exchange for r in ranks)
    _check_against_full_batch(full, ranks)


@pytest.mark.timeout(900)
def test_p2p_falls_back_to_all_reduce_by_consensus(tmp_path):
    """exchange="p2p" when ONE rank cannot export its buckets (injected by the test child: VQA_TEST_P2P_FAIL_RANK=1 replaces reduce_tensor there): every rank reaches the one
    collective of the set-up, the decision is all-reduced, BOTH ranks continue on the all-reduce and say why -- the
    gradients are still the full-batch gradients (nobody hangs, nobody trains on unreduced gradients)."""
    full = _run_ranks(tmp_path, "sub", 1, "none", "full")[0]
    ranks = _run_ranks(tmp_path, "sub", 2, "p2p", "fb", extra=GLOO_ONE_GPU, env={"VQA_TEST_P2P_FAIL_RANK": "1"})
    for r in ranks:
        assert str(r["exchange_used"][0]) == "allreduce" and "export" in str(r["fallback"][0]), (r["exchange_used"], r["fallback"])
    _check_against_full_batch(full, ranks)


@pytest.mark.parametrize("world,shard", [(1, 1024), (2, 4), (3, 1000), (8, 260096)])
def test_p2p_kernels_in_one_process(world, shard):
    """csrc/p2p.hip by itself: `world` buckets in one process stand for the ranks (the phases of the ranks run one after the
    other, which is what the cross-rank barriers guarantee): every bucket ends as the rank-ordered mean, bit for bit."""
    import ctypes as C
    from vqa_amd import _lib
    lib = _lib.load()
    torch.manual_seed(world)
    bufs = [torch.randn(world * shard, device="cuda") for _ in range(world)]
    want = bufs[0].clone()
    for r in range(1, world):
        want += bufs[r]
    want *= 1.0 / world
    ptrs = (C.c_void_p * world)(*[b.data_ptr() for b in bufs])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for r in range(world):
        _lib.check(lib.coattn_p2p_reduce_scatter(ptrs, world, r, shard, 1.0 / world, stream), "coattn_p2p_reduce_scatter")
    for r in range(world):
        _lib.check(lib.coattn_p2p_all_gather(ptrs, world, r, shard, stream), "coattn_p2p_all_gather")
    torch.cuda.synchronize()
    for r in range(world):
        assert torch.equal(bufs[r], want), r
    assert lib.coattn_p2p_reduce_scatter(ptrs, world, 0, shard + 1, 1.0, stream) != 0        # shard not a multiple of 4
    assert lib.coattn_p2p_reduce_scatter(ptrs, 9, 0, shard, 1.0, stream) != 0
