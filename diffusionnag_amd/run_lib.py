"""Host mirror of the Transfer-NAG generation loop, the caller of the sampling path
(reference ``main_exp/diffusion/run_lib.py:103-152``).

The reference quantises every sample on the host, checks it in Python and keeps a list of strings; here the
round's exit runs on the device (``dnag_nb201_validity``) against a persistent first-occurrence table, and the
loop reads back two integers per round.  Strings are built once, at the end, from the table.
"""

from __future__ import annotations

import random

import numpy as np
import torch

from . import arch_functions as AF
from . import utils


def reset_seed(seed):
    """Reference ``main_exp/diffusion/run_lib.py`` helper: reseed python, numpy and torch."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def generate_archs_meta(config, sampling_fn, score_model, score_ema, meta_surrogate_model, num_samples, args=None, task=None,
                        patient_factor=20, batch_size=256):
    """Same arguments and return value (list of unique valid architecture strings) as the reference.
    ``score_ema`` may be None when the model already holds the EMA weights."""
    mask_s = utils.aug_mask(utils.nb201_adj())[0].to(config.device)                 # :118-124 (default algo 'floyed')
    if score_ema is not None:
        score_ema.copy_to(score_model.parameters())                                  # :126
    score_model.eval()
    meta_surrogate_model.eval()
    c_scale = args.classifier_scale
    num_sampling_rounds = int(np.ceil(num_samples / batch_size) * patient_factor) if num_samples > batch_size else int(patient_factor)
    seen = AF.new_seen_table(config.device)
    rnd, base = 0, 0
    while rnd < num_sampling_rounds:                                                 # :134
        rnd += 1
        sample = sampling_fn(score_model, mask_s, meta_surrogate_model, classifier_scale=c_scale, task=task)
        _, _, counts = AF.device_validity(sample, seen, base)
        base += sample.shape[0]
        c_scale -= args.scale_step                                                   # :145
        seed = int(random.random() * 10000)
        reset_seed(seed)
        if hasattr(sampling_fn, "seed"):
            sampling_fn.seed = seed                                                  # the Philox key follows the reseed
        if int(counts[1]) >= num_samples:                                            # :149 (one 8-byte readback per round)
            break
    first = seen.cpu().numpy()
    order = np.argsort(first[first != AF.UNSEEN], kind="stable")
    codes = np.nonzero(first != AF.UNSEEN)[0][order]                                 # first-occurrence order
    return [AF.code_to_string(int(c)) for c in codes]
