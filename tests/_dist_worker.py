"""Worker of the multi-process tests (started by tests/test_distributed.py with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set). mode "exchange": event-exchange logic only (CPU, gloo).
mode "sim": a sharded simulation on cuda:0 against the unsharded one (gloo for the exchange, all
ranks on the same GPU: what a one-GPU box can run)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402


def main():
    mode = sys.argv[1]
    # CSC_TEST_BACKEND=nccl (RCCL): one GPU per rank; gloo (default): all ranks on cuda:0.
    backend = os.environ.get("CSC_TEST_BACKEND", "gloo")
    dev = 0
    if backend == "nccl":
        import torch

        dev = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    from cascade_amd import _capi

    if mode == "rcclhelpers":
        # The three exchange helpers on DEVICE memory through the nccl (RCCL) backend, with as many
        # ranks as there are: what a one-GPU box can execute of the production path (one rank:
        # every collective returns the rank's own bytes, but tensor types, sizes, the
        # __cuda_array_interface__ views and the stream handling are the ones N ranks use).
        import torch

        from cascade_amd.parallel import make_device_alltoall, make_device_gather, make_summary_exchange

        assert backend == "nccl"
        ex = make_summary_exchange()
        for step in range(3):
            out = ex(np.arange(6, dtype=np.int64) + 10 * rank + step)
            assert out.shape == (world, 6)
            for r in range(world):
                assert np.array_equal(out[r], np.arange(6, dtype=np.int64) + 10 * r + step)
        gather, a2a = make_device_gather(), make_device_alltoall()
        nb = 1 << 20
        src = torch.randint(0, 255, (nb,), dtype=torch.uint8, device="cuda")
        dst = torch.zeros(nb * world, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gather(src.data_ptr(), dst.data_ptr(), nb)
        assert torch.equal(dst[rank * nb : (rank + 1) * nb], src)
        # all-to-all: rank r receives segment r of every sender (ragged segments, one of them empty when world > 1)
        seg = [1000 * (r + 1) if (r != 1 or world == 1) else 0 for r in range(world)]
        send = torch.randint(0, 255, (sum(seg),), dtype=torch.uint8, device="cuda")
        recv = torch.zeros(max(seg[rank], 1) * world, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        if seg[rank]:
            a2a(send.data_ptr(), recv.data_ptr(), seg, seg[rank])
            off = sum(seg[:rank])
            assert torch.equal(recv[rank * seg[rank] : (rank + 1) * seg[rank]], send[off : off + seg[rank]])
        dist.barrier()
        print("rcclhelpers ok")
        dist.destroy_process_group()
        return
    if mode == "exchange":
        from cascade_amd.parallel import make_summary_exchange, merge_conjunctions

        ex = make_summary_exchange()
        for step in range(3 if backend == "gloo" else 0):
            local = np.array([rank * 3 + step, 7 - rank, 100 + rank, rank, rank + 10, 0], dtype=np.int64)
            local[5:6].view(np.float64)[0] = 0.25 * (rank + 1) + step
            out = ex(local)
            assert out.shape == (world, 6)
            for r in range(world):
                assert out[r, 0] == r * 3 + step and out[r, 3] == r and out[r, 5:6].view(np.float64)[0] == 0.25 * (r + 1) + step
        # Merging rank-local conjunction lists: segments of the same superstep are merged by
        # (time, i, j), supersteps stay in order even when their time ranges overlap (the last
        # superstep of a propagate_until keeps records beyond the final time).
        rng = np.random.default_rng(100 + rank)
        recs, segs = [], []
        for step, (t0, cnt) in enumerate([(0.0, 3 + rank), (1.0, 0 if rank == 0 else 2), (0.9, 4)]):
            c = np.zeros(cnt, dtype=_capi.CONJ_DTYPE)
            c["i"], c["j"] = rng.integers(0, 50, cnt), rng.integers(50, 100, cnt)
            c["time"] = np.sort(t0 + rng.random(cnt))
            if step == 2:
                c["time"][:2] = 1.5  # ties in time across the processes
            recs.append(c)
            if cnt:
                segs.append((step + 1, cnt))
        mine = (np.concatenate(recs), np.array(segs, dtype=np.uint64).reshape(-1, 2))
        objs = [None] * world
        dist.all_gather_object(objs, mine)
        merged = merge_conjunctions(objs)
        assert len(merged) == sum(len(o[0]) for o in objs)
        # step-major, then (time, i, j)
        off = 0
        for st in (1, 2, 3):
            cnt = sum(int(c) for o in objs for s, c in o[1] if int(s) == st)
            part = merged[off : off + cnt]
            key = list(zip(part["time"].tolist(), part["i"].tolist(), part["j"].tolist()))
            assert key == sorted(key), (st, rank)
            off += cnt
        assert off == len(merged)
        print("exchange ok", rank)
    elif mode == "sim":
        from cascade_amd import sim
        from cascade_amd.parallel import gather_conjunctions, shard
        from cascade_amd.workloads import random_leo_pop_kep

        def all_collisions(p):
            objs = [None] * world
            dist.all_gather_object(objs, np.array(p.last_collisions))
            c = np.concatenate(objs)
            return c[np.lexsort((c["j"], c["i"], c["time"]))]

        for n, n_par_ct, conj_thresh, scale in [(20000, 1, 0.0, 1.0), (6000, 3, 1e-3, 3e4)]:
            st = random_leo_pop_kep(n, seed=21)
            st[:, 6] *= scale
            full = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj_thresh, device=dev)
            part = shard(sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj_thresh, device=dev), propagation=False)
            import torch

            events = 0
            for _ in range(4):
                of, op = full.step(), part.step()
                assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
                assert np.array_equal(full.state, part.state)
                assert full.time == part.time
                # The shards partition the pair set.
                t = torch.tensor([part.last_bp_pairs], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t)
                assert int(t.item()) == full.last_bp_pairs, (int(t.item()), full.last_bp_pairs)
                assert part.last_bp_pairs < full.last_bp_pairs or full.last_bp_pairs < 100
                assert all_collisions(part).tobytes() == full.last_collisions.tobytes()
                events += len(full.last_collisions) + len(full.last_step_conjunctions)
            assert gather_conjunctions(part).tobytes() == full.conjunctions.tobytes()
            if len(full.conjunctions) > 10:
                assert len(part.conjunctions) < len(full.conjunctions), "conjunction records stay on the rank that found them"
            if scale > 1:
                assert events > 0
        print("sim ok", rank)
    elif mode == "simprop":
        # The propagation sharded too (set_shard_propagation: each rank integrates its range of
        # particles, one all-gather of the per-particle records, Taylor rows on demand) against
        # the unsharded simulation: plain Kepler with collisions and conjunctions, and J2 + drag
        # with reentries and removals; ragged ranges (the particle counts are no multiples of the
        # number of processes).
        import torch

        from cascade_amd import dynamics, outcome, sim
        from cascade_amd.parallel import gather_conjunctions, shard
        from cascade_amd.workloads import leo_si_population, random_leo_pop_kep

        def all_collisions(p):
            objs = [None] * world
            dist.all_gather_object(objs, np.array(p.last_collisions))
            c = np.concatenate(objs)
            return c[np.lexsort((c["j"], c["i"], c["time"]))]

        # (a) chunks dealt to the processes (n_par_ct = 3 and 7: even and ragged blocks), the
        # per-particle records by all-to-all (own chunks only) or by all-gather, (b) every chunk
        # everywhere with its pair tree sharded.
        for n_par_ct, chunks, a2a in [(3, True, True), (7, True, True), (7, True, False), (3, False, False)]:
            st = random_leo_pop_kep(6001, seed=21)
            st[:, 6] *= 3e4
            full = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=1e-3, device=dev)
            part = shard(sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=1e-3, device=dev), propagation=True, chunks=chunks, alltoall=a2a)
            events = 0
            for _ in range(4):
                of, op = full.step(), part.step()
                assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
                assert np.array_equal(full.state, part.state)
                assert full.time == part.time
                assert all_collisions(part).tobytes() == full.last_collisions.tobytes()
                per = -(-6001 // world)
                assert full.last_propagated == 6001 and part.last_propagated == min(per, 6001 - rank * per)
                t = torch.tensor([part.last_bp_pairs], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(t)
                assert int(t.item()) == full.last_bp_pairs, (int(t.item()), full.last_bp_pairs)
                c0, cl = part.last_chunk_block
                if chunks and of.name == "success":
                    per_c = -(-n_par_ct // world)
                    assert (c0, cl) == (min(n_par_ct, rank * per_c), max(0, min(per_c, n_par_ct - rank * per_c))), (rank, c0, cl)
                elif not chunks:
                    assert c0 == 0 and cl == full.last_nchunks
                if a2a:
                    per_c = -(-n_par_ct // world)
                    mine = max(0, min(per_c, n_par_ct - rank * per_c))
                    assert part.last_exchanged_bytes == world * per * (80 + 96 * mine), (rank, part.last_exchanged_bytes)
                else:
                    assert part.last_exchanged_bytes == world * per * (80 + 96 * n_par_ct)
                events += len(full.last_collisions) + len(full.last_step_conjunctions)
            of, op = full.propagate_until(full.time + 1.0), part.propagate_until(part.time + 1.0)
            assert of == op and full.time == part.time and np.array_equal(full.state, part.state)
            assert gather_conjunctions(part).tobytes() == full.conjunctions.tobytes()
            assert events > 0

        Re, GM = 6378136.3, 3.986004407799724e14
        st, pars = leo_si_population(1003, seed=7)
        rng = np.random.default_rng(3)
        for k, i in enumerate(rng.choice(1003, size=8, replace=False)):
            r0, ang = Re + 158e3 + 2.5e3 * k, rng.uniform(0, 2 * np.pi)
            v = np.sqrt(GM / r0)
            st[i, :6] = [r0 * np.cos(ang), r0 * np.sin(ang), 0.0, -v * np.sin(ang), v * np.cos(ang), 0.0]
            pars[i, 0] = 5e-2
        kw = dict(dyn=dynamics.simple_earth(C22S22=False), n_par_ct=8, reentry_radius=Re + 150e3, device=dev)
        full = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
        part = shard(sim(st.copy(), 185.5663, pars=pars.copy(), **kw), propagation=True)
        nre = 0
        for _ in range(25):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state) and full.time == part.time
            assert part.last_propagated <= -(-part.nparts // world)
            if of == outcome.reentry:
                nre += 1
                full.remove_particles([full.interrupt_info])
                part.remove_particles([part.interrupt_info])
        assert nre >= 3

        # Tape dynamics (the reference's default simple_earth: J2 + C22S22 + drag) over a range.
        st, pars = leo_si_population(301, seed=9)
        kw = dict(dyn=dynamics.simple_earth(), n_par_ct=4, reentry_radius=Re + 150e3, device=dev)
        full = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
        part = shard(sim(st.copy(), 185.5663, pars=pars.copy(), **kw), propagation=True)
        for _ in range(3):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state) and full.time == part.time
            assert part.last_propagated <= -(-301 // world)

        # User dynamics (a random right-hand side of tests/test_wave_plan_fuzz.py): every process
        # compiles the tape's straight-line kernel itself (hipRTC) and integrates its range with it.
        from test_wave_plan_fuzz import _random_dynamics

        dyn = _random_dynamics(12)
        npars = dyn.tape["npars"]
        st = random_leo_pop_kep(501, seed=12)
        pars = np.random.default_rng(12).uniform(0.5, 1.5, (501, max(1, npars)))[:, :npars].copy()
        kw = dict(dyn=dyn, n_par_ct=3, conj_thresh=1e-2, min_coll_radius=float("inf"), device=dev)
        if npars:
            kw["pars"] = pars
        full = sim(st.copy(), 0.23, **kw)
        part = shard(sim(st.copy(), 0.23, **kw), propagation=True)
        for _ in range(3):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state) and full.time == part.time
        assert gather_conjunctions(part).tobytes() == full.conjunctions.tobytes()
        print("simprop ok", rank)
    elif mode == "simerr":
        # An error that only ONE process can see -- the global box of chunk 1 overflows f32, the
        # one of chunk 0 does not, and the chunks are dealt to the processes -- must be raised on
        # every process (it travels as a code in the summary exchange); before, the process that
        # threw left the other one waiting in the all-gather forever.
        from cascade_amd import sim
        from cascade_amd.parallel import shard
        from cascade_amd.workloads import random_leo_pop_kep

        st = random_leo_pop_kep(500, seed=4)
        st[0, :6] = [-1.5e38, 0, 0, 0, 0, 0]  # far away, at rest
        st[1, :6] = [1e37, 0, 0, 5e38, 0, 0]  # chunk 0: x up to 1.25e38 (extent 2.75e38 < FLT_MAX), chunk 1: 2.4e38 (3.9e38: inf)
        full = sim(st.copy(), 0.23, n_par_ct=2, device=dev)
        part = shard(sim(st.copy(), 0.23, n_par_ct=2, device=dev), propagation=True)
        msg = "A global bounding box with non-finite boundaries and/or size was generated"
        for s in (full, part):
            before, t0 = np.array(s.state), s.time
            try:
                s.step()
                raise AssertionError("no exception on rank %d" % rank)
            except ValueError as e:
                assert msg in str(e), str(e)
            assert np.array_equal(s.state, before) and s.time == t0
        # ... and the simulation is usable afterwards (same calls on every process).
        st2 = random_leo_pop_kep(500, seed=4)
        full.set_new_state_pars(st2.copy())
        part.set_new_state_pars(st2.copy())
        assert full.step() == part.step() and np.array_equal(full.state, part.state)
        print("simerr ok", rank)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
