"""Capture golden vectors of the sampling round's exit from the REFERENCE implementation (build container
only): ``is_valid_NAS201_x`` error codes, ``BasicArchMetricsMeta.compute_validity`` strings and validity
(reference ``analysis/arch_functions.py:85-118``, ``:209-329``) on seeded sampler-like outputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_validity.py

The reference's ``utils.quantize`` is not importable here (``utils.py`` needs torch_scatter); its threshold
``x >= 0.5`` (``utils.py:254-255``) is applied in this script before the reference's validity code runs.
Only inputs and outputs are stored.
"""

from __future__ import annotations

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/NAS-Bench-201"
sys.dont_write_bytecode = True


def sampler_like_outputs(B, seed):
    """[B, 8, 7] floats in the sampler's output scale: mostly valid one-hot cells pushed off 0/1 by noise,
    plus every kind of rule violation and some unstructured rows."""
    rs = np.random.RandomState(seed)
    x = np.zeros((B, 8, 7), np.float32)
    for b in range(B):
        types = [0] + list(rs.randint(2, 7, size=6)) + [1]
        kind = rs.randint(0, 10)
        if kind == 0:
            types[rs.randint(1, 7)] = rs.randint(0, 2)          # start / end type inside the cell
        elif kind == 1:
            types[0 if rs.randint(2) else 7] = rs.randint(2, 7)  # wrong first / last node
        for n, t in enumerate(types):
            x[b, n, t] = 1.0
        if kind == 2:
            x[b, rs.randint(8), rs.randint(7)] = 1.0             # (maybe) a second type on one node
        elif kind == 3:
            x[b, rs.randint(8)] = 0.0                            # a node without a type
        elif kind == 4:
            x[b] = rs.uniform(0, 1, size=(8, 7)) > 0.8
    x = x + rs.uniform(-0.45, 0.45, size=x.shape).astype(np.float32)   # never crosses the 0.5 threshold
    x[0, 1, :] = [0.1, 0.2, 0.5, 0.3, 0.1, 0.0, 0.2]             # exactly 0.5 counts as 1 (>=)
    return x.astype(np.float32)


def main():
    os.chdir(tempfile.mkdtemp(prefix="dnag_golden_"))
    sys.path.insert(0, REF)
    from analysis import arch_functions as AF                   # reference

    for tag, B, seed in [("validity_nb201_a", 512, 21), ("validity_nb201_b", 64, 22)]:
        x = sampler_like_outputs(B, seed)
        q = [(xi >= 0.5).astype(np.float32) for xi in x]
        metrics = AF.BasicArchMetricsMeta()
        valid, validity, valid_str, all_str = metrics.compute_validity(q)
        errs = [AF.is_valid_NAS201_x(qi)[1] for qi in q]
        unique, uniqueness = metrics.compute_uniqueness(valid_str) if valid_str else ([], 0.0)
        path = os.path.join(HERE, tag + ".npz")
        np.savez_compressed(path, seed=seed, x=x, error_types=np.asarray(errs, np.int32), validity=np.float64(validity),
                            all_arch_str=np.asarray([s if s is not None else "" for s in all_str]),
                            n_unique=np.int32(len(unique)), uniqueness=np.float64(uniqueness))
        print(f"{tag}.npz {os.path.getsize(path)/1024:.1f} KiB validity={validity:.3f} unique={len(unique)} errs={np.bincount(np.asarray(errs)+1)}")


if __name__ == "__main__":
    main()
