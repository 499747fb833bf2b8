"""Multi-GPU operation of ``cascade_amd.sim``: one process per GPU, ``torch.distributed``.

Every process holds the same simulation. The broad and narrow phases are sharded over the pair
tree (``csc_set_shard``), and the per-step lists of detected collisions and conjunctions are merged
with one small all-gather (``make_event_exchange``; a second one only when some process found an
event). By default propagation, sort and tree build run redundantly on each GPU (at 1M Kepler
objects they take ~2 ms, less than moving the per-particle results over xGMI would);
``shard(sim, propagation=True)`` shards the integration as well: every process integrates its own
range of particles and ONE all-gather of 80 + 96 C bytes per particle (``make_device_gather``,
in place on the simulation's device buffers) gives everybody the boxes, trajectory records, substep
counts and final states -- worth it when the integration dominates the superstep (DESIGN.md
section 7). Backend "nccl" (RCCL over xGMI) on GPUs, "gloo" in the tests.
"""
import numpy as np


def merge_events(parts_coll, parts_conj):
    """Concatenate per-process event arrays and sort them by (time, i, j): the container order
    the single-process narrow phase produces (libcascade_hip sorts its lists the same way)."""
    coll = np.concatenate(parts_coll) if parts_coll else np.empty(0)
    conj = np.concatenate(parts_conj) if parts_conj else np.empty(0)
    if len(coll) > 1:
        coll = coll[np.lexsort((coll["j"], coll["i"], coll["time"]))]
    if len(conj) > 1:
        conj = conj[np.lexsort((conj["j"], conj["i"], conj["time"]))]
    return coll, conj


def make_event_exchange(group=None, device=None):
    """exchange(coll, conj) -> (coll, conj) over the processes of ``group``."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def exchange(coll, conj):
        counts = torch.tensor([coll.nbytes, conj.nbytes], dtype=torch.int64, device=device)
        all_counts = torch.empty(world * 2, dtype=torch.int64, device=device)  # flat: what gloo accepts too
        dist.all_gather_into_tensor(all_counts, counts, group=group)
        all_counts = all_counts.cpu().numpy().reshape(world, 2)
        if not all_counts.any():
            return coll, conj
        nmax = int(all_counts.sum(axis=1).max())
        payload = np.zeros(nmax, dtype=np.uint8)
        payload[: coll.nbytes] = coll.view(np.uint8).reshape(-1)
        payload[coll.nbytes : coll.nbytes + conj.nbytes] = conj.view(np.uint8).reshape(-1)
        send = torch.from_numpy(payload).to(device)
        recv = torch.empty(world * nmax, dtype=torch.uint8, device=device)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.cpu().numpy().reshape(world, nmax)
        pc, pj = [], []
        for r in range(world):
            nc, nj = int(all_counts[r, 0]), int(all_counts[r, 1])
            pc.append(recv[r, :nc].copy().view(coll.dtype))
            pj.append(recv[r, nc : nc + nj].copy().view(conj.dtype))
        return merge_events(pc, pj)

    return exchange


class _DeviceBytes:
    """A stretch of device memory owned by the simulation, as torch sees it."""

    def __init__(self, address, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(address), False), "version": 2}


def make_device_gather(group=None):
    """gather(send_address, recv_address, nbytes): all-gather of ``nbytes`` bytes of device memory
    per process (what ``sim.set_shard_propagation`` expects). RCCL moves the buffers in place;
    with the gloo backend (the CPU tests) they are staged through host memory."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"

    def gather(send_address, recv_address, nbytes):
        dev = torch.device("cuda", torch.cuda.current_device())
        send = torch.as_tensor(_DeviceBytes(send_address, nbytes), device=dev)
        recv = torch.as_tensor(_DeviceBytes(recv_address, nbytes * world), device=dev)
        if on_device:
            dist.all_gather_into_tensor(recv, send, group=group)
        else:
            out = torch.empty(nbytes * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, send.cpu(), group=group)
            recv.copy_(out)
        torch.cuda.synchronize(dev)

    return gather


def shard(sim_obj, group=None, propagation=False):
    """Shard ``sim_obj`` over the processes of ``group`` (call it on every process, on
    simulations constructed from the same arguments). ``propagation=True`` shards the
    integration too (one all-gather of 80 + 96 C bytes per particle and superstep): worth it when
    the integration dominates the superstep, see DESIGN.md section 7."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world > 1:
        sim_obj.set_shard(rank, world, make_event_exchange(group))
        sim_obj.set_shard_propagation(make_device_gather(group) if propagation else None)
    return sim_obj
