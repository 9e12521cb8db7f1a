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
