"""CPU: libcascade_hip.so loads and exports every symbol declared in include/cascade_hip.h
(no compute calls without a GPU), and the product path fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "cascade_hip.h")).read()
    return sorted(set(re.findall(r"\b(csc_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build_hip()
    from cascade_amd import _capi

    L = _capi.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    # the ctypes table binds exactly the declared set
    assert sorted(_capi.SYMBOLS) == names


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    from cascade_amd import sim

    s = sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23)  # construction and validation need no device
    with pytest.raises(RuntimeError, match="there is no CPU fallback"):
        s.step()


def test_host_validation_messages():
    """Error texts of the reference (test/sim_api.cpp:191-236, cascade.py/test.py:251-375)
    raised before any device work."""
    import numpy as np

    from cascade_amd import sim

    with pytest.raises(ValueError, match="The collisional timestep must be finite and positive, but it is 0 instead"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.0)
    with pytest.raises(ValueError, match="An input state with 7 columns is expected"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0]]), 0.23)
    with pytest.raises(ValueError, match="The number of collisional timesteps to be processed in parallel cannot be zero"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, n_par_ct=0)
    with pytest.raises(ValueError, match="is inside the spherical central body"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, reentry_radius=2.0)
    with pytest.raises(ValueError, match="is outside the exit radius"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, exit_radius=0.5)
    with pytest.raises(ValueError, match="An invalid particle size of -1 was detected for the particle at index 0"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, -1.0]]), 0.23)
    with pytest.raises(ValueError, match="The conjunction threshold value -1 is invalid"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, conj_thresh=-1)
    with pytest.raises(ValueError, match="vector of 3 elements"):
        sim(np.array([[1.0, 0, 0, 0, 1.0, 0, 0.0]]), 0.23, reentry_radius=[1.0, 2.0])


def test_workload_generator():
    from cascade_amd.workloads import random_leo_pop_kep
    import numpy as np

    st = random_leo_pop_kep(1000, 42)
    r = np.linalg.norm(st[:, :3], axis=1)
    assert st.shape == (1000, 7) and r.min() > 0.99 and r.max() < 1.33
    assert np.all(st[:, 6] >= 1.57e-8) and np.all(st[:, 6] <= 1.57e-7)
    assert np.array_equal(st, random_leo_pop_kep(1000, 42))
    # first canonical draw of std::mt19937(42): raw outputs 1608637542, 3421126067
    a0 = (1.3 - 1.02) * ((1608637542.0 + 3421126067.0 * 4294967296.0) / 18446744073709551616.0) + 1.02
    x = np.linalg.norm(st[0, :3])
    assert abs(x - a0) < 0.03 * a0  # r within e of a


def test_import_cascade_alias():
    """`import cascade` (the reference's package name, cascade.py/__init__.py:47-61) gives the
    same classes; the six logger setters of cascade.py/logging.cpp:19-29 exist and are callable."""
    import cascade
    import cascade.dynamics
    import cascade_amd

    assert cascade.sim is cascade_amd.sim and cascade.outcome is cascade_amd.outcome
    assert cascade.dynamics.kepler(1.0).kind == 0
    for lvl in ("trace", "debug", "info", "warn", "err", "critical"):
        getattr(cascade, "set_logger_level_" + lvl)()
    cascade.set_logger_level_info()
    cascade.set_nthreads(3)
    assert cascade.get_nthreads() == 3
    cascade.set_nthreads(0)


def test_refit_handoff_isa():
    """k_bvh_refit (cascade_amd/csrc/k_bvh.hpp) hands child boxes between waves on different XCDs
    with relaxed agent-scope atomics and one s_waitcnt, relying on how gfx950 executes them: the
    box stores and loads must carry sc1 (write-through / L2 bypass), the arrival counter must be a
    returning global atomic, and a `s_waitcnt vmcnt(0)` must sit between the last box store of a
    level and the atomic. Checked on the ISA of the built object, so that a toolchain update that
    lowers them differently is caught here and not by a silently stale parent box."""
    import shutil
    import subprocess
    import tempfile

    import __graft_entry__ as g

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "build", "cascade_hip.o")
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    if not os.path.exists(obj):
        g.build_hip()
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "cascade_hip.o")
        shutil.copy(obj, local)
        subprocess.check_call([objdump, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert dev, os.listdir(tmp)
        asm = subprocess.check_output([objdump, "-d", "--no-show-raw-insn", os.path.join(tmp, dev[0])], text=True)
    lines = asm.splitlines()
    # (the cross-workgroup kernel; k_bvh_refit_local, which runs before it, hands off through LDS)
    start = next(i for i, l in enumerate(lines) if "11k_bvh_refitE" in l and l.rstrip().endswith(">:"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.split("//")[0].strip() for l in lines[start:end]]
    mem = [l for l in body if l.startswith(("global_", "flat_", "buffer_", "s_waitcnt vmcnt"))]
    atom = [i for i, l in enumerate(mem) if l.startswith("global_atomic_add")]
    assert len(atom) == 1 and "sc0" in mem[atom[0]], mem  # one returning atomic: the arrival counter
    stores = [l for l in mem if l.startswith("global_store")]
    assert len(stores) == 8 and all("sc1" in l for l in stores), stores
    after = [l for l in mem[atom[0] + 1 :] if l.startswith("global_load")]
    box_loads = [l for l in after if "sc1" in l]
    assert len(box_loads) == 16, after  # the two child boxes, agent scope
    # The loop closes on the atomic: between the last box store and the atomic (wrapping around)
    # there is a wait for all outstanding vector memory operations.
    assert mem[atom[0] - 1].startswith("s_waitcnt vmcnt(0)"), mem[: atom[0] + 1]
    assert not any(l.startswith(("buffer_wbl2", "buffer_inv")) for l in body), "no cache-wide fences"


def test_host_hash_sees_structured_edits():
    """csc_host_hash is what sim::step() uses to decide that nobody edited the host arrays
    (cascade.py/core.cpp:199-230 hands out writable views). Round 2's hash was a linear sum mod
    2^64: blind to an even number of sign flips (time reversal `state[:, 3:6] *= -1`), to +d / -d
    on two words and, with bad luck, to swapped rows. Every word is avalanched with its position
    now; each of these edits must change BOTH 64-bit lanes."""
    import ctypes as C

    import numpy as np

    import __graft_entry__ as g

    g.build_hip()
    from cascade_amd import _capi

    L = _capi.lib()

    def h(a):
        a = np.ascontiguousarray(a)
        out = (C.c_uint64 * 2)()
        L.csc_host_hash(a.ctypes.data_as(C.c_void_p), a.nbytes, out)
        return (int(out[0]), int(out[1]))

    rng = np.random.default_rng(5)
    for n in (2, 1000, 200000):  # one thread and several (>= 65536 words)
        st = rng.normal(size=(n, 7))
        st[:, 6] = np.abs(st[:, 6])
        base = h(st)
        assert base == h(st.copy()), "deterministic, independent of the address"
        edits = []
        a = st.copy(); a[:, 3:6] *= -1; edits.append(a)  # 3n sign flips: even for every n here
        a = st.copy(); a[0, 0] *= -1; a[0, 1] *= -1; edits.append(a)
        a = st.copy(); a[[0, n - 1]] = a[[n - 1, 0]]; edits.append(a)  # two rows swapped
        a = st.copy().view(np.uint64); a[0, 0] += 1 << 40; a[1, 0] -= 1 << 40; edits.append(a.view(np.float64))
        a = st.copy().view(np.uint64); a[0, 2] ^= 1 << 62; a[n - 1, 2] ^= 1 << 62; edits.append(a.view(np.float64))
        a = st.copy(); a[n // 2, 4] = np.nextafter(a[n // 2, 4], 9.0); edits.append(a)  # one ulp
        seen = {base}
        for e in edits:
            he = h(e)
            assert he[0] != base[0] and he[1] != base[1], (n, he, base)
            seen.add(he)
        assert len(seen) == len(edits) + 1
    # chunks combine by addition: the transfer paths hash piecewise
    assert h(np.zeros(0)) == (0, 0) and h(np.ones(3)) != h(np.ones(4))
    b = np.frombuffer(bytes(range(20)), dtype=np.uint8)  # a tail that is not a multiple of 8
    assert h(b) != h(b[:16]) and h(b) != h(np.concatenate([b[:19], [0]]).astype(np.uint8))
