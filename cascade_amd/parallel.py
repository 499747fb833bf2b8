"""Multi-GPU operation of ``cascade_amd.sim``: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

Every process holds the same simulation (same state, same calls). Per superstep
(``shard(sim)``, the partition BASELINE.json's north_star names):

  1. each process integrates ITS range of particles (``csc_propagate_range``);
  2. ONE all-gather of the per-particle records -- substep count, status, event, final state and,
     per chunk, the box and the 64-byte trajectory record: 80 + 96 C bytes -- in place on the
     simulation's device buffers (``make_device_gather``);
  3. with ``n_par_ct >= world`` every process owns a block of the superstep's chunks -- chunks
     are independent once the trajectories are known -- and sorts, builds the trees of, traverses
     and narrow-phases only those: nothing is replicated, and step 2 becomes ONE all-to-all that
     brings a process the per-chunk records of its own chunks only (``make_device_alltoall``). With fewer chunks than processes every
     process sorts and builds the whole tree of every chunk (replicated: cheaper than a distributed
     sort at these sizes, and bit-identical everywhere), expands only its share of each pair tree
     and runs the narrow phase on the pairs it found. Either way Taylor rows are produced on demand
     for the particles of the local candidate pairs;
  4. ONE all-gather of a 48-byte summary per process (``make_summary_exchange``: event counts
     and the earliest local collision), from which every process derives the same outcome.

Conjunction records never travel during the run: ``sim.conjunctions`` holds the records found by
this process; ``gather_conjunctions(sim)`` merges them into the list the unsharded simulation
would hold. ``shard(sim, propagation=False)`` keeps the integration on every process (no step 2).
"""
import numpy as np


def make_summary_exchange(group=None, device=None):
    """exchange(local int64[6]) -> int64[world, 6]: one all-gather on a device tensor (RCCL) --
    or on a host tensor with the gloo backend."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    device = torch.device(device)
    # Persistent buffers: nothing is allocated per superstep. With RCCL the 48 bytes go through a
    # pinned host pair and a device pair that live as long as the exchange does.
    recv = torch.empty(world * 6, dtype=torch.int64, device=device)  # flat: what gloo accepts too
    send = torch.empty(6, dtype=torch.int64, device=device)
    on_device = device.type == "cuda"
    if on_device:
        send_h = torch.empty(6, dtype=torch.int64).pin_memory()
        recv_h = torch.empty(world * 6, dtype=torch.int64).pin_memory()
    else:
        send_h, recv_h = send, recv
    send_np, recv_np = send_h.numpy(), recv_h.numpy().reshape(world, 6)

    def exchange(local):
        send_np[:] = np.asarray(local, dtype=np.int64)
        if on_device:
            send.copy_(send_h, non_blocking=True)
        dist.all_gather_into_tensor(recv, send, group=group)
        if on_device:
            recv_h.copy_(recv, non_blocking=True)
            torch.cuda.current_stream(device).synchronize()
        return recv_np.copy()

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


def make_device_alltoall(group=None):
    """alltoall(send_address, recv_address, send_bytes, recv_bytes_each): process r receives
    ``send_bytes[r]`` consecutive bytes of every process's send buffer (``recv_bytes_each`` from
    each, rank-major). RCCL moves device memory in place; gloo (the CPU tests) stages through
    host memory."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"

    def alltoall(send_address, recv_address, send_bytes, recv_bytes_each):
        dev = torch.device("cuda", torch.cuda.current_device())
        send_bytes = [int(b) for b in send_bytes]
        assert len(send_bytes) == world
        send = torch.as_tensor(_DeviceBytes(send_address, sum(send_bytes)), device=dev)
        recv = torch.as_tensor(_DeviceBytes(recv_address, recv_bytes_each * world), device=dev)
        if on_device:
            dist.all_to_all_single(recv, send, [recv_bytes_each] * world, send_bytes, group=group)
        else:
            out = torch.empty(recv_bytes_each * world, dtype=torch.uint8)
            dist.all_to_all_single(out, send.cpu(), [recv_bytes_each] * world, send_bytes, group=group)
            recv.copy_(out)
        torch.cuda.synchronize(dev)

    return alltoall


def shard(sim_obj, group=None, propagation=True, chunks=True, alltoall=True):
    """Shard ``sim_obj`` over the processes of ``group`` (call it on every process, on
    simulations constructed from the same arguments). ``propagation=False`` keeps the integration
    replicated on every process (no all-gather of the per-particle records). ``chunks`` (with
    the propagation sharded and ``n_par_ct >= world``): every process also owns a block of the
    superstep's chunks and sorts / builds / traverses / narrow-phases only those -- nothing is
    replicated; ``chunks=False`` keeps every chunk on every process with its pair tree sharded.
    ``alltoall`` (with the chunks dealt out): the exchange of step 2 is an all-to-all in which a
    process receives the per-chunk records of ITS chunks only (80 + 96 * own chunks bytes per
    particle instead of 80 + 96 C); ``alltoall=False`` keeps the all-gather."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world > 1:
        sim_obj.set_shard(rank, world, make_summary_exchange(group))
        sim_obj.set_shard_propagation(
            make_device_gather(group) if propagation else None, chunks, make_device_alltoall(group) if propagation and chunks and alltoall else None
        )
    return sim_obj


def make_event_exchange(*_a, **_k):
    """Removed in round 2 (the conjunction records no longer travel during the run)."""
    raise RuntimeError("cascade_amd.parallel.make_event_exchange is gone: shard(sim) sets up make_summary_exchange(); "
                       "read the merged conjunction list with gather_conjunctions(sim)")


def merge_events(*_a, **_k):
    """Removed in round 2."""
    raise RuntimeError("cascade_amd.parallel.merge_events is gone: use merge_conjunctions(parts) / gather_conjunctions(sim)")


def merge_conjunctions(parts):
    """parts: per process, (records, segments) as ``sim.conjunctions`` / ``sim._conj_segments``
    give them. Returns the list of the unsharded simulation: superstep by superstep, the records
    of each sorted by (time, i, j) (the container order of the single-process narrow phase)."""
    chunks = []
    for recs, segs in parts:
        off = 0
        for step, cnt in np.asarray(segs, dtype=np.uint64).reshape(-1, 2):
            chunks.append((int(step), recs[off : off + int(cnt)]))
            off += int(cnt)
        assert off == len(recs), "conjunction segments do not cover the records"
    if not chunks:
        return parts[0][0][:0].copy()
    out = []
    for step in sorted({c[0] for c in chunks}):
        c = np.concatenate([r for s, r in chunks if s == step])
        out.append(c[np.lexsort((c["j"], c["i"], c["time"]))])
    return np.concatenate(out)


def gather_conjunctions(sim_obj, group=None):
    """The conjunctions of a sharded simulation as the unsharded one would hold them (collective:
    call it on every process; host-side object gather, meant for the end of a run, not for the
    superstep loop)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = (np.array(sim_obj.conjunctions), np.array(sim_obj._conj_segments))
    if world == 1:
        return merge_conjunctions([mine])
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    return merge_conjunctions(parts)
