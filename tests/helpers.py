"""Shared test helpers: minimal config objects (the reference's configs are ml_collections
ConfigDicts, which is not installed) and golden-vector loading."""

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Cfg(dict):
    """Stand-in for ml_collections.ConfigDict: attribute access and ``in``."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def score_config(dims, device):
    ge = Cfg(n_layers=dims.n_layers, d_model=dims.d_model, n_head=dims.n_head, d_ff=dims.d_ff, dropout=0.1,
             n_vocab=dims.n_vocab)
    return Cfg(device=torch.device(device),
               data=Cfg(name="ofa" if dims.stagewise_pos else "NASBench201", max_node=dims.max_node,
                        n_vocab=dims.n_vocab, centered=True),
               model=Cfg(name="CATE", nonlinearity="swish", dropout=0.1, pos_enc_type=2, graph_encoder=ge),
               training=Cfg(continuous=True),
               sampling=Cfg(method="pc", predictor="reverse_diffusion", corrector="langevin", snr=0.16,
                            n_steps_each=1, probability_flow=False, noise_removal=True, regress=True, labels="max"))


def predictor_config(d, device):
    ge = Cfg(initial_hidden=d.n_vocab, gcn_hidden=d.gcn_hidden, gcn_layers=d.gcn_layers, linear_hidden=d.linear_hidden)
    return Cfg(device=torch.device(device),
               model=Cfg(name="MetaNeuralPredictor", nonlinearity="swish", input_type="DA", hs=d.hs, nz=d.nz,
                         num_sample=d.num_sample, graph_encoder=ge))


def predictor_cate_config(d, device):
    ge = Cfg(n_layers=d.n_layers, d_model=d.d_model, n_head=d.n_head, d_ff=d.d_ff, dropout=0.1, n_vocab=d.n_vocab)
    return Cfg(device=torch.device(device),
               data=Cfg(name="ofa", max_node=d.max_node, n_vocab=d.n_vocab, centered=True),
               model=Cfg(name="MetaPredictorCATE", nonlinearity="swish", dropout=0.1, input_type="DA", hs=d.hs, nz=d.nz,
                         num_sample=d.num_sample, graph_encoder=ge))


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def ofa_sampler_like_outputs(B, seed):
    """[B, 20, 9] floats in the sampler's output scale: mostly valid OFA encodings (stage depths 2..4, one type per
    used layer slot) pushed off 0/1 by noise that never crosses the 0.5 threshold, plus every rule violation of
    MobileNetV3/analysis/arch_functions.py:374-429 and some unstructured graphs."""
    rs = np.random.RandomState(seed)
    x = np.zeros((B, 20, 9), np.float32)
    for b in range(B):
        kind = rs.randint(0, 10)
        for s in range(5):
            d = rs.randint(2, 5)
            for j in range(d):
                x[b, 4 * s + j, rs.randint(9)] = 1.0
        if kind == 0:
            x[b, rs.randint(20), rs.randint(9)] = 1.0            # (maybe) a second type in one slot
        elif kind == 1:
            x[b, 4 * rs.randint(5) + rs.randint(2)] = 0.0        # a missing first / second layer
        elif kind == 2:
            s = rs.randint(5)
            x[b, 4 * s + 2] = 0.0                                # third layer skipped ...
            x[b, 4 * s + 3, rs.randint(9)] = 1.0                 # ... but a fourth one
        elif kind == 3:
            x[b] = rs.uniform(0, 1, size=(20, 9)) > 0.9
    x = x + rs.uniform(-0.45, 0.45, size=x.shape).astype(np.float32)
    x[0, 0, :] = [0.5, 0.1, 0.2, 0.3, 0.1, 0.0, 0.2, 0.4, 0.49]   # exactly 0.5 counts as 1 (>=)
    return x.astype(np.float32)
