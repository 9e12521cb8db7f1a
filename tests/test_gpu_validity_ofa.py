"""Exit of a MobileNetV3 / OFA sampling round on the device (dnag_ofa_validity through the C ABI) against the oracle's
restatement of MobileNetV3/analysis/arch_functions.py:125-216, :374-429.  PARITY UNPINNED against the reference itself
(its module imports wandb and igraph, which are not installed); integer work, so kernel and oracle must agree exactly.
Needs an MI355X."""

import numpy as np
import pytest
import torch

from diffusionnag_amd import arch_functions as AF
from oracle import dnag_oracle as O
from tests.helpers import ofa_sampler_like_outputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,seed", [(700, 31), (33, 32), (1, 33)])
def test_ofa_validity_matches_oracle(B, seed):
    xs = ofa_sampler_like_outputs(B, seed)
    q = [(xi >= 0.5).astype(np.float32) for xi in xs]
    errs, strs = O.compute_validity_ofa(q)
    m = AF.BasicArchMetricsMetaOFA(device=DEV)
    valid, validity, valid_str, all_str, error_types = m.compute_validity(torch.from_numpy(xs).to(DEV))
    assert all_str == strs
    assert error_types == [e for e in errs if e != -1]
    assert validity == sum(e == -1 for e in errs) / B
    assert list(m.last_counts) == [sum(e == k for e in errs) for k in (-1, 1, 2, 3)]
    for v, qi in zip(valid, [qi for qi, e in zip(q, errs) if e == -1]):
        assert np.array_equal(v.numpy(), qi.astype(np.int64))
    if B > 1:
        assert len(set(errs)) == 4                     # every outcome occurs in the sample
    # the list form that utils.quantize hands over gives the same answer
    assert m.compute_validity(q)[3] == all_str
    # strings of the reference's own helper from the one-hot rows
    for qi, s in zip(q, strs):
        if s is not None:
            assert m.get_string_from_onehot_x(qi) == s


def test_ofa_evaluate_and_uniqueness():
    xs = ofa_sampler_like_outputs(300, 41)
    xs = np.concatenate([xs, xs[:100]])                # duplicates
    m = AF.BasicArchMetricsMetaOFA(device=DEV)
    (validity, uniqueness, novelty, e1, e2, e3), unique, props, all_str = m.evaluate(torch.from_numpy(xs).to(DEV))
    errs, strs = O.compute_validity_ofa([(xi >= 0.5).astype(np.float32) for xi in xs])
    valid_strs = [s for s in strs if s is not None]
    assert validity == len(valid_strs) / len(xs)
    assert len(unique) == len(set(valid_strs)) and uniqueness == len(set(valid_strs)) / len(valid_strs)
    assert (e1, e2, e3) == tuple(sum(e == k for e in errs) / len(xs) for k in (1, 2, 3))
    assert novelty == -1.0 and all_str is None and props["test_acc_list"] == [0]


def test_ofa_empty_batch_and_shape_check():
    m = AF.BasicArchMetricsMetaOFA(device=DEV)
    assert m.compute_validity([]) == ([], 0, [], [], [])
    st, ty, c = AF.device_validity_ofa(torch.zeros(0, 20, 9, device=DEV))
    assert st.numel() == 0 and list(c.cpu().numpy()) == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        AF.device_validity_ofa(torch.zeros(4, 8, 7, device=DEV))
