"""Turns the DATA files that the reference's own tests and benchmark hold into .npz fixtures
(numbers only: no reference source text travels). Run in the build container, which has
/root/reference; the .npz files are committed.

    python tests/golden/make_fixtures.py

  ref_lbf.npz         test/{x,y,z,vx,vy,vz,sizes}_lbf.txt -- 19 114 objects, SI units: the input of
                      test/last_bit_flip.cpp:25-66 (BVH split on the last Morton bit)
  ref_haumea_00.npz   test/haumea_00_{x,y,z,vx,vy,vz}.txt -- 9 983 objects: the input of
                      test/haumea_bug_00.cpp:33-131 (stopping terminal event, then a singularity)
  ref_leo_19647.npz   benchmark/data/test_ic_19647.txt (state, [19647][7]), test_par_19647.txt
                      (BSTAR) and best_fit_density.txt (12 constants): the catalogue and density
                      fit of benchmark/2022_leo_pop.cpp:197-209
"""
import os

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def col(path):
    return np.loadtxt(os.path.join(REF, path), dtype=np.float64)


def main():
    lbf = np.stack([col("test/%s_lbf.txt" % k) for k in ("x", "y", "z", "vx", "vy", "vz", "sizes")], axis=1)
    assert lbf.shape == (19114, 7), lbf.shape
    np.savez_compressed(os.path.join(HERE, "ref_lbf.npz"), state=lbf)

    hau = np.stack([col("test/haumea_00_%s.txt" % k) for k in ("x", "y", "z", "vx", "vy", "vz")], axis=1)
    assert hau.shape == (9983, 6), hau.shape
    np.savez_compressed(os.path.join(HERE, "ref_haumea_00.npz"), posvel=hau)

    ic = col("benchmark/data/test_ic_19647.txt").reshape(-1, 7)
    par = col("benchmark/data/test_par_19647.txt").reshape(-1, 1)
    dens = col("benchmark/data/best_fit_density.txt")
    assert ic.shape == (19647, 7) and par.shape == (19647, 1) and dens.shape == (12,)
    np.savez_compressed(os.path.join(HERE, "ref_leo_19647.npz"), state=ic, bstar=par, best_fit_density=dens)
    for f in ("ref_lbf.npz", "ref_haumea_00.npz", "ref_leo_19647.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
