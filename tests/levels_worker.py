"""Worker of test_sort_free_levels_keep_walls_and_lattices: the device reference filter on a cloud made to put thousands of
EQUAL coordinates around the medians of the box tree's upper levels, compared bit for bit with the oracle.  Run with
LSGPU_GS_DEBUG=1 the library says on stderr when the sort-free levels hand a filter over to the segmented sorts; the test
reads that."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cloud(kind, rng):
    import numpy as np
    if kind == "wall":
        # 400 k points, 6000 of them on the plane x = 0 exactly, which holds the first level's median: six thousand candidates
        # that tie on the cut key (more than k_gs_select keeps in LDS), resolved by the original index
        n, nw = 400000, 6000
        c = np.ones((n, 4), np.float32)
        c[:, 0] = rng.uniform(-50, 50, n); c[:, 1] = rng.uniform(-40, 40, n); c[:, 2] = rng.uniform(-2, 4, n)
        w = rng.choice(n, nw, replace=False)
        c[w, 0] = 0.0; c[w, 1] = rng.uniform(-5, 5, nw).astype(np.float32); c[w, 2] = rng.uniform(0, 3, nw).astype(np.float32)
        return c
    if kind == "lattice":
        # eight distinct x, four distinct y: a fifth of a segment ties at every upper level
        n = 120000
        c = np.ones((n, 4), np.float32)
        c[:, 0] = 10.0 * rng.integers(0, 8, size=n); c[:, 1] = 8.0 * rng.integers(0, 4, size=n); c[:, 2] = rng.normal(size=n)
        return c
    raise SystemExit("unknown cloud " + kind)


def main():
    import numpy as np
    from laser_slam_amd import icp
    from oracle import oracle_py as oracle           # the ctypes binding of oracle/ (test infrastructure: the checker)
    oracle.lib()
    kind = sys.argv[1]
    c = cloud(kind, np.random.default_rng(11))
    of, on = oracle.sampling_surface_normal(c, 10, 1.0, 2)
    with icp.IcpHandle() as h:
        gf, gn = h.filter_reference(c, 10, 1.0, 2)
    print("LEVELS_RESULT " + json.dumps({"points": int(len(c)), "kept": int(len(gf)),
                                         "equal": bool(np.array_equal(gf, of) and np.array_equal(gn, on))}))


if __name__ == "__main__":
    main()
