"""The C++ host (include/cascade/sim.hpp, cascade_amd/libcascade.so) through its own C++ test
programs (tests/cpp/*.cpp, built by __graft_entry__.build() into build/)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(name):
    exe = os.path.join(ROOT, "build", name)
    if not os.path.exists(exe):
        import __graft_entry__ as g

        g.build_hip()
        g.build_host()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checks passed" in r.stdout


def test_sim_api_cpp():
    """Construction, named arguments, error texts, removal: no device needed."""
    _run("test_sim_api")


@pytest.mark.gpu
def test_sim_gpu_cpp():
    """The reference's known-answer tests through cascade::sim::step() on the GPU."""
    _run("test_sim_gpu")
