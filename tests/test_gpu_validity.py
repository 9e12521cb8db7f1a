"""Exit of a sampling round on the device (dnag_nb201_validity through the C ABI) against the golden vectors
captured from the reference's analysis/arch_functions.py (tests/golden/validity_nb201_*.npz) and the oracle:
integer work, so everything must match exactly.  Needs an MI355X."""

import numpy as np
import pytest
import torch

from diffusionnag_amd import arch_functions as AF
from oracle import dnag_oracle as O
from tests.helpers import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["validity_nb201_a", "validity_nb201_b"])
def test_validity_strings_and_error_types_match_reference(name):
    g = golden(name)
    x = torch.from_numpy(g["x"]).to(DEV)
    m = AF.BasicArchMetricsMeta(device=DEV)
    valid, validity, valid_str, all_str = m.compute_validity(x)
    assert (m.last_error_types == g["error_types"]).all()
    assert [s if s is not None else "" for s in all_str] == list(g["all_arch_str"])
    assert validity == float(g["validity"])
    assert len(valid) == len(valid_str) == int(round(validity * len(all_str)))
    unique, uniq = m.compute_uniqueness(valid_str)
    assert len(unique) == int(g["n_unique"]) and uniq == float(g["uniqueness"])
    # the list form that utils.quantize hands over gives the same answer
    q = [(xi >= 0.5).astype(np.float32) for xi in g["x"]]
    assert m.compute_validity(q)[3] == all_str


def test_first_seen_table_accumulates_across_rounds_and_edge_cases():
    g = golden("validity_nb201_a")
    x = torch.from_numpy(g["x"]).to(DEV)
    seen = AF.new_seen_table(DEV)
    assert int((seen != AF.UNSEEN).sum()) == 0
    st0, cd0, c0 = AF.device_validity(x[:200], seen, 0)
    st1, cd1, c1 = AF.device_validity(x[200:], seen, 200)
    status = torch.cat([st0, st1]).cpu().numpy()
    code = torch.cat([cd0, cd1]).cpu().numpy()
    assert (status == g["error_types"]).all()
    assert int(c0[0]) + int(c1[0]) == int((status == -1).sum())
    assert int(c1[1]) == int(g["n_unique"])
    # first occurrence per code, exactly
    first = seen.cpu().numpy()
    want = {}
    for i, c in enumerate(code):
        if c >= 0 and c not in want:
            want[int(c)] = i
    assert {int(c): int(first[c]) for c in np.nonzero(first != AF.UNSEEN)[0]} == want
    # oracle agreement on codes and strings
    q = [(xi >= 0.5).astype(np.float32) for xi in g["x"]]
    errs, strs = O.compute_validity(q)
    assert errs == list(status)
    assert [AF.code_to_string(int(c)) if c >= 0 else None for c in code] == strs
    # empty batch and a batch of one
    st, cd, cnt = AF.device_validity(x[:0])
    assert st.numel() == 0 and int(cnt[0]) == 0
    st, cd, cnt = AF.device_validity(x[3:4])
    assert int(st[0]) == int(g["error_types"][3])


def test_full_size_round_exit():
    """65 536 graphs (BASELINE configs[4] size): the kernel agrees with a vectorised numpy restatement of the rules."""
    rs = np.random.RandomState(0)
    B = 65536
    types = rs.randint(0, 7, size=(B, 8))
    types[:, 0] = np.where(rs.uniform(size=B) < 0.9, 0, types[:, 0])
    types[:, 7] = np.where(rs.uniform(size=B) < 0.9, 1, types[:, 7])
    types[:, 1:7] = np.where(rs.uniform(size=(B, 6)) < 0.97, rs.randint(2, 7, size=(B, 6)), types[:, 1:7])
    x = np.zeros((B, 8, 7), np.float32)
    x[np.arange(B)[:, None], np.arange(8)[None, :], types] = 1.0
    extra = rs.uniform(size=(B, 8, 7)) < 0.002
    x = np.maximum(x, extra.astype(np.float32))
    q = x >= 0.5
    single = (q.sum(-1) == 1).all(-1)
    startend = q[:, 0, 0] & q[:, 7, 1]
    interm = ~(q[:, 1:7, 0] | q[:, 1:7, 1]).any(-1)
    want = np.where(~single, 1, np.where(~startend, 2, np.where(~interm, 3, -1)))
    code = ((q[:, 1:7].argmax(-1) - 2) * (5 ** np.arange(6))[None, :]).sum(-1)
    st, cd, cnt = AF.device_validity(torch.from_numpy(x).to(DEV), AF.new_seen_table(DEV), 0)
    assert (st.cpu().numpy() == want).all()
    assert (cd.cpu().numpy()[want == -1] == code[want == -1]).all() and (cd.cpu().numpy()[want != -1] == -1).all()
    assert int(cnt[0]) == int((want == -1).sum()) and int(cnt[1]) == len(set(code[want == -1]))
