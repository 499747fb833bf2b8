#!/usr/bin/env python3
"""bench.py -- object-days propagated per second, collision detection included.

Workload (BASELINE.json configs[1]): the random LEO Kepler population of
benchmark/random_leo_pop_kep.cpp scaled to 1,000,000 objects, ct = 0.23, Keplerian dynamics
mu = 1 (1 TU = 806.81 s), n_par_ct = 8 collisional timesteps per superstep (SURVEY 8(d) config 2
lists C in {1, 8}; results do not depend on it; 8 is what lets 8 GPUs take one chunk each, and it
is the faster setting on one GPU too; the C = 1 figure of round 1 is reported beside it as
"n_par_ct_1"). A "step" is one sim.step() (propagate -> AABB -> Morton -> sort -> BVH -> broad
phase -> narrow phase -> outcome over the 8 chunks) with the state already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

ONE experiment at every N: one 1M-object simulation (strong scaling). At N > 1 it is sharded the
way BASELINE.json's north_star says: every rank integrates its own range of objects, one RCCL
exchange moves the per-object records (final states; per-chunk boxes and trajectory records);
every rank takes its block of the superstep's chunks -- chunks are independent -- and sorts,
builds the trees of, traverses and narrow-phases those only, so the exchange is an all-to-all
that brings a rank the per-chunk records of its own chunks (with fewer chunks than ranks: an
all-gather, every chunk everywhere, pair trees sharded); conjunction records stay on the rank that found them,
event counts and the earliest collision are exchanged (cascade_amd/parallel.py). --replicate-propagation keeps the
integration on every rank instead; --mode replicas runs N independent simulations (weak scaling,
no collective; not the default).

Prints ONE JSON line (rank 0): the contract fields, "roofline" (dominant kernel: SURVEY 8(d)'s
algorithmic bytes / its HIP-event duration inside the timed region), "roofline_other_kernels",
"ms_per_step_host_sync" (the default mode of the drop-in class: host arrays authoritative at
every step) and, at N = 1, "cpu_baseline": the CPU oracle (an OpenMP port of the reference's
algorithms) timed on the SAME supersteps from the SAME state, after asserting that it and the GPU
agree on outcomes, pair counts, event lists and final states.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

TU_SECONDS = 806.81  # test/last_bit_flip.cpp:61
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VEC_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak
P = 20  # Taylor order at tol = DBL_EPSILON
# k_bp_tiles issues far more instructions than its bytes suggest: its VALU wave-instruction count per
# launch comes from a PMC run of THIS configuration (profiles/pmc_tiles.sh -> profiles/*_tiles_valu.json);
# one wave64 VALU instruction per 4 clocks and SIMD, 1024 SIMDs at 2.4 GHz (MI355X_MICROARCH.md).
VALU_PEAK_GINST = 1024 * 2.4 / 4  # 614.4 G wave-instructions/s


def taylor_flops(dyn="kepler", p=P):
    """FP64 flops of ONE object-substep: the Taylor recurrences of orders 0..p-1 for the given
    dynamics exactly as cascade_amd/csrc/k_propagate.hpp::dyn_order runs them (= oracle/orc.hpp::
    compute_tcs; an FMA counts 2, a division or square root 1), plus the Horner update of the 7
    state variables (7 * 2p). Step-size control, event polynomials and the AABB
    Horner are not counted."""
    conv = lambda n: 2 * (n + 1)  # noqa: E731  n + 1 FMAs

    def sq(n):  # conv_sq: symmetric half-sum
        return 1 if n == 0 else 2 * ((n + 1) // 2) + 1 + (2 if n % 2 == 0 else 0)

    tot = 0
    for n in range(p):
        f = 3 * sq(n) + 2  # u = x^2 + y^2 + z^2
        f += 4 if n == 0 else 3 * n + 2  # w = u^(-3/2)
        f += 3 * (conv(n) + 1)  # -mu x w ...
        f += 3 * conv(n) + 2 + 3 * n + 1  # r' = (x vx + y vy + z vz) / r
        f += 7  # x^[n+1] = rhs^[n] / (n + 1)
        if dyn in ("j2", "j2drag"):
            f += (2 * n + 2) + conv(n) + sq(n) + (conv(n) + 1) + 2 * (conv(n) + 2) + 3 * (conv(n) + 1)
        if dyn == "j2drag":
            f += 2 * max(n - 1, 0) + 1  # sqrt(u)
            f += 4 * (4 * n + 1) + 8  # four exponentials + density sum
            f += 3 * sq(n) + 2 + 2 * max(n - 1, 0) + 1  # |v|^2, |v|
            f += conv(n) + 1 + 3 * (conv(n) + 1)  # drag force
        tot += f
    return tot + 7 * 2 * p


# Algorithmic bytes per launch, P = 20: what the reference's algorithm has to move per
# unit, not what this implementation happens to fetch.
BYTES = {
    "k_propagate_store": lambda u: u["stored_substeps"] * (1176 + 16) + u["stored_objects"] * 56,
    # K1 without rows + K2 fused: state in, final state + boxes + (new) trajectory records out
    "k_propagate": lambda u: u["n"] * (56 + 48 + 32 * u["nchunks"] + 28 + 64 * u["nchunks"]),
    "k_np_fex": lambda u: u["np_candidates"] * 1008,
    "k_broad_phase": lambda u: u["bp_nodes_visited"] * 44 + u["pairs"] * 8,
    # K6 = 44 B x nodes visited + 8 B x pairs; the dual traversal visits two nodes per node pair
    # and ends in tiles (8 B each) when k_bp_tiles is on, in pairs otherwise.
    "k_bp_dual": lambda u: u["bp_node_pairs"] * 2 * 44 + (u["bp_tiles"] * 8 if u["bp_tiles"] else u["pairs"] * 8),
    # every tile record once, every particle's box + flags once (36 B; the tiles share them); then either
    # every pair out (csc_broad_phase) or, filtered on chip (csc_broad_phase_filtered, what step() runs),
    # every trajectory record once (64 B) and the surviving candidates out (16 B)
    "k_bp_tiles": lambda u: u["bp_tiles"] * 8 + u["n"] * u["nchunks"] * 36
    + (u["n"] * u["nchunks"] * 64 + u["np_candidates"] * 16 if u.get("bp_fused") else u["pairs"] * 8),
    # single-pass rows (every object's substep rows written while it is integrated; the boxes and
    # records are k_chunk_boxes' since round 2: state in, final state out, the rows out)
    "k_propagate_rows": lambda u: u["n"] * (56 + 48 + 28) + u["substeps"] * (1176 + 16),
    # the x, y, z, r series and the end time of every substep in, a box and a record per particle and chunk out
    "k_chunk_boxes": lambda u: u["substeps"] * (4 * 21 * 8 + 16) + u["n"] * u["nchunks"] * (32 + 64),
    # every pair record read once, every trajectory record read once (the pairs of a chunk share them)
    "k_np_prefilter": lambda u: u["pairs"] * 8 + u["n"] * u["nchunks"] * 64,
    "k_rs_scatter": lambda u: u["sort_keys"] * 24,
    "k_rs_hist": lambda u: u["sort_keys"] * 8,
    "k_gather_aabb": lambda u: u["n"] * u["nchunks"] * 2 * (32 + 64),
    "k_bvh_refit": lambda u: u["n"] * u["nchunks"] * 2 * 2 * 32,  # every node's box written once, read once
}


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary under profiles/
    (profiles/summarise.py) whose workload string matches this run; None otherwise."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and kernel in d.get("kernels", {}):
            best = (d["kernels"][kernel], os.path.basename(path))
    return best


def measured_valu(kernel, workload):
    """VALU wave-instructions per launch of `kernel` from the newest profiles/*_valu.json (a PMC
    run, profiles/pmc_tiles.sh) whose workload string matches this run; None otherwise."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and kernel in d.get("kernels", {}):
            best = (d["kernels"][kernel]["valu_per_launch"], os.path.basename(path))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nparts", type=int, default=1_000_000)
    ap.add_argument("--ct", type=float, default=0.23)
    ap.add_argument("--n-par-ct", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-sync-leg", action="store_true")
    ap.add_argument("--no-ct-sweep", action="store_true")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="sharded")
    ap.add_argument("--replicate-propagation", action="store_true",
                    help="N > 1: every rank integrates all objects (no all-gather of the per-object records)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process (which has made no GPU call)
        # starts the N ranks itself, relays rank 0's JSON line and fails if any rank fails.
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # (a line that says n_gpus = N must come from N ranks)
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report one as the other" % (args.gpus, world))
    dist = None
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist_

        # "nccl" is RCCL on ROCm. CSC_DIST_BACKEND=gloo lets a one-GPU box rehearse the
        # multi-process path (all ranks on the same device).
        backend = os.environ.get("CSC_DIST_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        dist_.init_process_group(backend=backend)
        dist = dist_
        tdev = "cuda" if backend == "nccl" else "cpu"
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: the process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    from cascade_amd import _capi, sim
    from cascade_amd.workloads import random_leo_pop_kep

    N = args.nparts
    sharded = world > 1 and args.mode == "sharded"
    replicas = world > 1 and not sharded
    shard_prop = sharded and not args.replicate_propagation
    st = random_leo_pop_kep(N, seed=42 + rank if replicas else 42)
    s = sim(st, args.ct, n_par_ct=args.n_par_ct, device=local_rank)
    if sharded:
        from cascade_amd.parallel import shard

        shard(s, propagation=shard_prop)
    s.host_sync = False  # state stays resident in HBM between steps
    s.upload_state()
    ctx = _capi.Context.from_handle(s._hip_context)  # the simulation's C-ABI context (profiling only)

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        s.step()

    # Per-kernel survey (untimed): HIP events around every launch of 3 supersteps give the kernel
    # table and the dominant kernel. The events themselves cost ~0.8 ms per superstep (~100
    # launches), so the timed region below keeps them only around the dominant kernel.
    ctx.profile(True)
    ctx.profile_reset()
    nsurvey = 3
    for _ in range(nsurvey):
        s.step()
    survey = ctx.profile_get()
    dom = max(survey, key=lambda k: survey[k][1])
    ctx.profile_only(dom)
    ctx.profile_reset()
    # The state the timed region starts from (for the CPU leg, which runs the same supersteps).
    st_w, t_w = None, s.time
    if world == 1 and not args.no_cpu_baseline:
        s.download_state()
        st_w = np.array(s.state)
    pairs = 0
    pairs_by_step = []
    outcomes = {}
    t_sim0 = s.time
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        oc = s.step()
        outcomes[oc.name] = outcomes.get(oc.name, 0) + 1
        pairs += s.last_bp_pairs
        pairs_by_step.append(s.last_bp_pairs)
    barrier()
    wall = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile(False)
    ctx.profile_only(None)
    sim_time = s.time - t_sim0

    wall_max = wall
    tot_obj_days = N * sim_time * TU_SECONDS / 86400.0
    propagated = s.last_propagated
    if dist is not None:
        import torch

        t = torch.tensor([wall], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_max = float(t.item())
        if replicas:
            od = torch.tensor([tot_obj_days], dtype=torch.float64, device=tdev)
            dist.all_reduce(od, op=dist.ReduceOp.SUM)
            tot_obj_days = float(od.item())
        pr = torch.tensor([float(pairs), float(propagated)], dtype=torch.float64, device=tdev)
        gathered = [torch.zeros_like(pr) for _ in range(world)]
        dist.all_gather(gathered, pr)
        per_rank_pairs = [float(g[0].item()) / args.steps for g in gathered]
        per_rank_prop = [int(g[1].item()) for g in gathered]
    else:
        per_rank_pairs = [pairs / args.steps]
        per_rank_prop = [int(propagated)]
    pairs_all = sum(per_rank_pairs)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = tot_obj_days / wall_max
    # kern: per-kernel (launches, ms) scaled to the timed steps: the dominant kernel from the
    # timed region itself, the others from the survey.
    kern = {k: (v[0] * args.steps / nsurvey, v[1] * args.steps / nsurvey) for k, v in survey.items()}
    kern[dom] = prof[dom]
    tot_kernel_ms = sum(v[1] for v in kern.values())
    # Units of one superstep of this rank (populations change slowly; the counts of all steps agree to <0.1 %).
    store_launches = max(kern.get("k_propagate_store", (0, 0))[0] / args.steps, 0)
    units = dict(
        n=int(propagated),
        nchunks=args.n_par_ct,
        substeps=s.last_total_substeps * propagated / N,
        pairs=pairs / args.steps,
        np_candidates=s.last_np_candidates,
        stored_objects=s.last_stored_objects,
        stored_substeps=s.last_stored_objects * max(1.0, s.last_total_substeps / N),
        sort_keys=N * args.n_par_ct,
        bp_node_pairs=s.last_bp_node_pairs,
        bp_tiles=s.last_bp_tiles,
        bp_nodes_visited=0,
        bp_fused=int(os.environ.get("CASCADE_BP_FUSE", "1") != "0"),
    )
    if dom == "k_broad_phase":
        units["bp_nodes_visited"] = s.bp_nodes_visited()
    if replicas:
        workload = "random_leo_pop_kep scaled to %d objects per GPU, %d independent simulations (Kepler mu=1, ct=%g, n_par_ct=%d, seed 42+rank)" % (N, world, args.ct, args.n_par_ct)
    else:
        workload = "random_leo_pop_kep scaled to %d objects, one simulation on %d GPU(s) (Kepler mu=1, ct=%g, n_par_ct=%d, seed 42)" % (N, world, args.ct, args.n_par_ct)

    def roofline_of(name):
        launches, ms = kern[name]
        per_step_launches = launches / args.steps
        avg_ms = ms / launches
        r = dict(bound="hbm", kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None, avg_launch_ms=round(avg_ms, 5))
        alg_bytes = None
        if name in BYTES:
            alg_bytes = BYTES[name](units) / per_step_launches
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            r.update(achieved=round(achieved, 2), frac=round(achieved / HBM_PEAK_GBS, 5), algorithmic_bytes_per_launch=int(alg_bytes))
        if name in ("k_propagate", "k_propagate_rows"):
            # report against the FP64 vector peak AND the HBM figure; the bound that
            # takes longer at its peak is the binding one.
            fl = taylor_flops("kepler") * units["substeps"] / per_step_launches
            tf = fl / (avg_ms * 1e-3) / 1e12
            r["fp64"] = dict(flops_per_object_substep=taylor_flops("kepler"), flops_per_launch=int(fl), achieved_TFLOPs=round(tf, 3), peak_TFLOPs=FP64_VEC_PEAK_TFLOPS, frac=round(tf / FP64_VEC_PEAK_TFLOPS, 5))
            t_f, t_b = fl / (FP64_VEC_PEAK_TFLOPS * 1e12), (alg_bytes or 0) / (HBM_PEAK_GBS * 1e9)
            if t_f >= t_b:
                r.update(bound="fp64_valu", achieved=round(tf, 3), peak=FP64_VEC_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / FP64_VEC_PEAK_TFLOPS, 5), hbm_frac=round(t_b / (avg_ms * 1e-3), 5))
        if name == "k_bp_tiles" and units["bp_tiles"]:
            # SURVEY 8(d) bounds K6 by HBM bytes: that is `frac`. What the kernel is busy with is
            # instruction issue: reported beside it, from a PMC run of the same configuration.
            v = measured_valu(name, workload)
            if v is not None:
                ginst = v[0] / (avg_ms * 1e-3) / 1e9
                r["valu_issue"] = dict(valu_wave_instructions_per_launch=int(v[0]), achieved_Ginst=round(ginst, 1), peak_Ginst=VALU_PEAK_GINST,
                                       frac=round(ginst / VALU_PEAK_GINST, 5), source="SQ_INSTS_VALU per launch, profiles/" + v[1])
        t = measured_traffic(name, workload)
        if t is not None:
            tb = t[0]["traffic_bytes"]
            r.update(traffic=tb, traffic_source="profiles/" + t[1], traffic_GBps=round(tb / (avg_ms * 1e-3) / 1e9, 1), traffic_note=t[0].get("note"))
        return r

    roof = roofline_of(dom)
    roof["units"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in units.items()}
    other_roofs = [roofline_of(k) for k in sorted(kern, key=lambda k: -kern[k][1])[1:6] if k in BYTES]

    if replicas:
        parallelism = "replicas x%d (independent simulations, no collective)" % world
    elif world == 1:
        parallelism = "one simulation, one GPU"
    elif shard_prop:
        if args.n_par_ct >= world:
            parallelism = "one simulation: propagation sharded x%d, the %d chunks dealt to the ranks (sort / BVH / broad / narrow phase of a rank's chunks only: nothing replicated), one RCCL all-to-all brings each rank the per-object records of its chunks, conjunction records rank-local, event counts + earliest collision exchanged" % (world, args.n_par_ct)
        else:
            parallelism = "one simulation: propagation sharded x%d, one RCCL all-gather of the per-object records, replicated sort / BVH, pair tree and narrow phase sharded x%d, conjunction records rank-local, event counts + earliest collision exchanged" % (world, world)
    else:
        parallelism = "one simulation: replicated propagation / sort / BVH, pair tree and narrow phase sharded x%d, conjunction records rank-local, collisions + earliest event exchanged" % world

    out = {
        "metric": "object-days propagated/sec incl. collision detection",
        "value": round(value, 2),
        "unit": "object-days/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall_max / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak" if replicas else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "nparts": N,
            "ct": args.ct,
            "n_par_ct": args.n_par_ct,
            "parallelism": parallelism,
            "backend": backend,
            "ranks": dist.get_world_size() if dist is not None else 1,  # what the process group (RCCL) saw
            "propagated_per_rank": per_rank_prop,
            "exchange_bytes_received_per_step_rank0": int(s.last_exchanged_bytes),
            "bp_pairs_per_step_per_rank": per_rank_pairs,
            "bp_pairs_per_step_all_ranks": pairs_all,
            "outcomes": outcomes,
            "sim_time_advanced_TU": sim_time,
            "state": "resident in HBM between steps (host_sync off)",
        },
        "roofline": roof,
        "roofline_other_kernels": other_roofs,
        "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])},
        "kernel_ms_note": "dominant kernel: HIP events inside the timed region; the others: a 3-superstep survey before it",
        "kernel_ms_total_per_step": round(tot_kernel_ms / args.steps, 4),
    }

    if world == 1 and not args.no_host_sync_leg:
        # The default mode of the drop-in class (cascade.py/core.cpp:199-230 hands out writable views
        # of the host arrays): state up / down around every step. Not part of `value`.
        s.host_sync = True
        s.step()
        k = max(3, min(args.steps, 10))
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(k):
            s.step()
        ctx.sync()
        out["ms_per_step_host_sync"] = round((time.perf_counter() - t1) / k * 1e3, 4)
        s.host_sync = False

    if world == 1 and not args.no_host_sync_leg and args.n_par_ct != 1:
        # Round 1's configuration (n_par_ct = 1) on the same population, for continuity.
        c1 = sim(st, args.ct, n_par_ct=1, device=local_rank)
        c1.host_sync = False
        for _ in range(3):
            c1.step()
        c1x = _capi.Context.from_handle(c1._hip_context)
        c1x.sync()
        t1, ts1, k1 = time.perf_counter(), c1.time, 10
        for _ in range(k1):
            c1.step()
        c1x.sync()
        w1 = time.perf_counter() - t1
        out["n_par_ct_1"] = {"ms_per_step": round(w1 / k1 * 1e3, 4), "value": round(N * (c1.time - ts1) * TU_SECONDS / 86400.0 / w1, 2), "unit": "object-days/s", "steps": k1}
        del c1

    if world == 1 and not args.no_ct_sweep:
        # "also report a tuned ct" (results do not depend on ct,
        # doc/tutorials/quickstart.ipynb:191). Same superstep length (ct x n_par_ct), shorter chunks.
        sweep = [{"ct": args.ct, "n_par_ct": args.n_par_ct, "ms_per_step": out["ms_per_step"], "value": out["value"],
                  "aabb_pairs_per_particle_per_chunk": round(pairs_all / N / args.n_par_ct, 3)}]
        for div in (2, 4):
            ct_k, c_k = args.ct / div, args.n_par_ct * div
            sk = sim(st, ct_k, n_par_ct=c_k, device=local_rank)
            sk.host_sync = False
            for _ in range(3):
                sk.step()
            skx = _capi.Context.from_handle(sk._hip_context)
            skx.sync()
            t1, ts1, k1, pk = time.perf_counter(), sk.time, 6, 0
            for _ in range(k1):
                sk.step()
                pk += sk.last_bp_pairs
            skx.sync()
            w1 = time.perf_counter() - t1
            sweep.append({"ct": ct_k, "n_par_ct": c_k, "ms_per_step": round(w1 / k1 * 1e3, 4),
                          "value": round(N * (sk.time - ts1) * TU_SECONDS / 86400.0 / w1, 2),
                          "aabb_pairs_per_particle_per_chunk": round(pk / k1 / N / c_k, 3)})
            del sk
        out["ct_sweep"] = sweep
        best = max(sweep, key=lambda e: e["value"])
        out["tuned_ct"] = {"ct": best["ct"], "n_par_ct": best["n_par_ct"], "value": best["value"], "unit": "object-days/s",
                           "note": "best of ct_sweep; `value` above stays the ct = %g figure" % args.ct}

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, st_w, t_w, pairs_by_step)

    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(n):
    """One child process per GPU with the torch.distributed environment set (rendezvous on
    127.0.0.1, a free port), the same command line. Rank 0's stdout is relayed (the JSON line); the
    other ranks' output goes to stderr. Returns non-zero if any rank failed, after ending the rest."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading

    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # a rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    reader.join(timeout=10)
    out0 = got[0] if got else ""
    lines = [l for l in (out0 or "").splitlines() if l.strip()]
    if rc == 0 and not any(l.startswith("{") for l in lines):
        rc = 1
    for l in lines:  # (the JSON line on stdout, anything else a library printed there on stderr)
        print(l, file=sys.stdout if l.startswith("{") else sys.stderr)
    if rc != 0:
        print("bench.py: a rank failed (exit code %d)" % rc, file=sys.stderr)
    return rc


def cpu_baseline(args, st_w, t_w, gpu_pairs_by_step):
    """The CPU oracle (oracle/: an OpenMP port of the reference's algorithms, kind "port") on a
    bounded sample of the SAME workload: the first 3 supersteps of the timed region (1 untimed, 2
    timed), from the state the GPU's timed region started from. Before the timing is accepted the
    GPU path (a fresh simulation from that state) and the oracle must agree on every superstep:
    outcome, broad-phase pair count, collision and conjunction lists, and the states to 1e-9."""
    from cascade_amd import sim
    from oracle import orc_py

    orc_py.build()
    n = len(st_w)
    nsteps = 3 if args.n_par_ct < 4 else 2  # (about 2.2 s of CPU per chunk of 1M objects on 16 threads)
    g = sim(st_w.copy(), args.ct, n_par_ct=args.n_par_ct)
    g.time = t_w
    o = orc_py.Sim(st_w.copy(), args.ct, n_par_ct=args.n_par_ct)
    o.time = t_w
    walls, t_sim = [], []
    checked = []
    for k in range(nsteps):
        og = g.step()
        t_sim.append(o.time)
        t0 = time.perf_counter()
        oo = o.step()
        walls.append(time.perf_counter() - t0)
        np_o = int(o.stage_times()["n_bp_pairs"])
        assert og.name == oo, ("outcome", k, og.name, oo)
        assert g.last_bp_pairs == np_o, ("pair count", k, g.last_bp_pairs, np_o)
        if k < len(gpu_pairs_by_step):
            assert gpu_pairs_by_step[k] == np_o, ("pair count of the timed region", k, gpu_pairs_by_step[k], np_o)
        cg, co = g.last_collisions, o.collisions()
        assert len(cg) == len(co) and np.array_equal(cg["i"], co["i"]) and np.array_equal(cg["j"], co["j"]), ("collisions", k)
        jg, jo = g.last_step_conjunctions, o.step_conjunctions()
        assert len(jg) == len(jo) and np.array_equal(jg["i"], jo["i"]) and np.array_equal(jg["j"], jo["j"]), ("conjunctions", k)
        so = o.state
        assert np.all(np.abs(g.state - so) <= 1e-9 * np.maximum(1.0, np.abs(so))), ("state", k)
        checked.append({"outcome": oo, "bp_pairs": np_o, "collisions": len(co), "conjunctions": len(jo)})
    wall = sum(walls[1:])
    od = n * (o.time - t_sim[1]) * TU_SECONDS / 86400.0
    return {
        "value": round(od / wall, 2),
        "unit": "object-days/s",
        "cores": int(orc_py.lib().orc_num_threads()),
        "kind": "port",
        "sample": "supersteps 1-%d of the timed region from the same state (%d objects, ct=%g x n_par_ct=%d): 1 untimed + %d timed, %.3f s of CPU wall" % (nsteps, n, args.ct, args.n_par_ct, nsteps - 1, wall),
        "checked_against_gpu": checked,
        "check": "outcome, broad-phase pair count, collision and conjunction lists equal; states within 1e-9 relative",
        "stage_times_last_step_s": o.stage_times(),
    }


if __name__ == "__main__":
    main()
