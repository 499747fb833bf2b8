"""Multi-process tests of the sharded path (cascade_amd/parallel.py, sim.set_shard): world size
2, gloo rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(mode, world=2, timeout=240, backend="gloo"):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CSC_TEST_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert mode + " ok" in out, out


def test_event_exchange_gloo():
    """The per-superstep summary all-gather, and the merge of rank-local conjunction lists
    (ragged, empty, tied in time, overlapping time ranges of successive supersteps)."""
    _run_world("exchange")


@pytest.mark.gpu
def test_sharded_sim_matches_unsharded():
    """Two processes share the queries of one simulation (both on cuda:0, gloo exchange): same
    outcomes, interrupt info, states and conjunction lists as the unsharded simulation; the
    shards' pair counts add up to the full count."""
    _run_world("sim")


@pytest.mark.gpu
def test_sharded_propagation_matches_unsharded():
    """Three processes, each integrating its own range of particles (all on cuda:0, gloo staging
    of the all-gather): same outcomes, states, event lists as the unsharded simulation, for plain
    Kepler with collisions / conjunctions and for J2 + drag with reentries and removals."""
    _run_world("simprop", world=3)


@pytest.mark.gpu
def test_local_error_is_raised_on_every_rank():
    """A non-finite global box in the chunk block of one process only: both processes raise the
    reference's ValueError (src/sim_propagate.cpp:453-456) instead of one hanging in the
    collective."""
    _run_world("simerr", world=2, timeout=120)


def _gpus():
    import torch

    return torch.cuda.device_count()


@pytest.mark.gpu
def test_rccl_exchange_helpers():
    """The exchange helpers of cascade_amd.parallel on device memory through the nccl (RCCL)
    backend, with every GPU there is (one rank on a one-GPU box: RCCL initialises, the device
    tensors, views of the simulation's memory and collectives are the production ones)."""
    _run_world("rcclhelpers", world=max(1, min(_gpus(), 8)), backend="nccl", timeout=180)


@pytest.mark.gpu
def test_sharded_sim_rccl():
    """The production backend: one GPU per rank, RCCL all-gathers on device tensors. Needs at
    least two GPUs; skipped on a one-GPU box (where the gloo tests above cover the logic)."""
    if _gpus() < 2:
        pytest.skip("needs >= 2 GPUs: backend nccl (RCCL) with one GPU per rank")
    world = min(_gpus(), 8)
    _run_world("sim", world=world, backend="nccl")
    _run_world("simprop", world=world, backend="nccl")
    _run_world("simerr", world=2, backend="nccl")
