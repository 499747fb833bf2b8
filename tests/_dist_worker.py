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
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from cascade_amd import _capi
    from cascade_amd.parallel import make_event_exchange, merge_events

    if mode == "exchange":
        ex = make_event_exchange()
        rng = np.random.default_rng(100 + rank)
        # Step 1: nobody has events. Step 2: ragged lists (rank 0 empty). Step 3: ties in time.
        for step, (nc, nj) in enumerate([(0, 0), (3 * rank, 2 * rank + (rank == 1)), (2, 1)]):
            coll = np.zeros(nc, dtype=_capi.COLL_DTYPE)
            conj = np.zeros(nj, dtype=_capi.CONJ_DTYPE)
            coll["i"], coll["j"] = rng.integers(0, 50, nc), rng.integers(50, 100, nc)
            coll["time"] = rng.random(nc) if step < 2 else 0.5
            conj["i"], conj["j"] = rng.integers(0, 50, nj), rng.integers(50, 100, nj)
            conj["time"], conj["dist"] = rng.random(nj), rng.random(nj)
            conj["state_i"] = rng.random((nj, 6))
            mc, mj = ex(coll, conj)
            # Reference result: gather the raw lists with the object collective and merge.
            objs = [None] * world
            dist.all_gather_object(objs, (coll, conj))
            rc, rj = merge_events([o[0] for o in objs], [o[1] for o in objs])
            assert mc.tobytes() == rc.tobytes() and mj.tobytes() == rj.tobytes(), (step, rank)
            assert np.all(np.diff(mc["time"]) >= 0) and np.all(np.diff(mj["time"]) >= 0)
            assert len(mc) == sum(len(o[0]) for o in objs) and len(mj) == sum(len(o[1]) for o in objs)
        print("exchange ok", rank)
    elif mode == "sim":
        from cascade_amd import sim
        from cascade_amd.parallel import shard
        from cascade_amd.workloads import random_leo_pop_kep

        for n, n_par_ct, conj_thresh, scale in [(20000, 1, 0.0, 1.0), (6000, 3, 1e-3, 3e4)]:
            st = random_leo_pop_kep(n, seed=21)
            st[:, 6] *= scale
            full = sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj_thresh)
            part = shard(sim(st.copy(), 0.23, n_par_ct=n_par_ct, conj_thresh=conj_thresh))
            import torch

            events = 0
            for _ in range(4):
                of, op = full.step(), part.step()
                assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
                assert np.array_equal(full.state, part.state)
                assert full.time == part.time
                # The shards partition the pair set.
                t = torch.tensor([part.last_bp_pairs], dtype=torch.int64)
                dist.all_reduce(t)
                assert int(t.item()) == full.last_bp_pairs, (int(t.item()), full.last_bp_pairs)
                assert part.last_bp_pairs < full.last_bp_pairs or full.last_bp_pairs < 100
                assert part.last_collisions.tobytes() == full.last_collisions.tobytes()
                events += len(full.last_collisions) + len(full.last_step_conjunctions)
            assert part.conjunctions.tobytes() == full.conjunctions.tobytes()
            if scale > 1:
                assert events > 0
        print("sim ok", rank)
    elif mode == "simprop":
        # The propagation sharded too (set_shard_propagation: each rank integrates its range of
        # particles, one all-gather of the per-particle records, Taylor rows on demand) against
        # the unsharded simulation: plain Kepler with collisions and conjunctions, and J2 + drag
        # with reentries and removals; ragged ranges (the particle counts are no multiples of the
        # number of processes).
        from cascade_amd import dynamics, outcome, sim
        from cascade_amd.parallel import shard
        from cascade_amd.workloads import leo_si_population, random_leo_pop_kep

        st = random_leo_pop_kep(6001, seed=21)
        st[:, 6] *= 3e4
        full = sim(st.copy(), 0.23, n_par_ct=3, conj_thresh=1e-3)
        part = shard(sim(st.copy(), 0.23, n_par_ct=3, conj_thresh=1e-3), propagation=True)
        events = 0
        for _ in range(4):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state)
            assert full.time == part.time
            assert part.last_collisions.tobytes() == full.last_collisions.tobytes()
            assert full.last_propagated == 6001 and part.last_propagated == (2001 if rank < 2 else 1999)
            events += len(full.last_collisions) + len(full.last_step_conjunctions)
        of, op = full.propagate_until(full.time + 1.0), part.propagate_until(part.time + 1.0)
        assert of == op and full.time == part.time and np.array_equal(full.state, part.state)
        assert part.conjunctions.tobytes() == full.conjunctions.tobytes()
        assert events > 0

        Re, GM = 6378136.3, 3.986004407799724e14
        st, pars = leo_si_population(1003, seed=7)
        rng = np.random.default_rng(3)
        for k, i in enumerate(rng.choice(1003, size=8, replace=False)):
            r0, ang = Re + 158e3 + 2.5e3 * k, rng.uniform(0, 2 * np.pi)
            v = np.sqrt(GM / r0)
            st[i, :6] = [r0 * np.cos(ang), r0 * np.sin(ang), 0.0, -v * np.sin(ang), v * np.cos(ang), 0.0]
            pars[i, 0] = 5e-2
        kw = dict(dyn=dynamics.simple_earth(C22S22=False), n_par_ct=8, reentry_radius=Re + 150e3)
        full = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
        part = shard(sim(st.copy(), 185.5663, pars=pars.copy(), **kw), propagation=True)
        nre = 0
        for _ in range(25):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state) and full.time == part.time
            assert part.last_propagated <= (part.nparts + 2) // 3
            if of == outcome.reentry:
                nre += 1
                full.remove_particles([full.interrupt_info])
                part.remove_particles([part.interrupt_info])
        assert nre >= 3

        # Tape dynamics (the reference's default simple_earth: J2 + C22S22 + drag) over a range.
        st, pars = leo_si_population(301, seed=9)
        kw = dict(dyn=dynamics.simple_earth(), n_par_ct=4, reentry_radius=Re + 150e3)
        full = sim(st.copy(), 185.5663, pars=pars.copy(), **kw)
        part = shard(sim(st.copy(), 185.5663, pars=pars.copy(), **kw), propagation=True)
        for _ in range(3):
            of, op = full.step(), part.step()
            assert of == op and full.interrupt_info == part.interrupt_info, (rank, of, op)
            assert np.array_equal(full.state, part.state) and full.time == part.time
            assert part.last_propagated <= 101
        print("simprop ok", rank)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
