"""Host mirror of the reference's ``models/utils.py``: model registry and the score / guidance
function wrappers, same names, arguments and error behaviour (reference ``models/utils.py:8-179``).

The wrappers dispatch to the HIP library; nothing here runs a network in PyTorch.
"""

from __future__ import annotations

import torch

from . import sde_lib

_MODELS = {}


def register_model(cls=None, *, name=None):
    """A decorator for registering model classes (reference ``models/utils.py:8-25``)."""

    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _MODELS:
            raise ValueError(f'Already registered model with name: {local_name}')
        _MODELS[local_name] = cls
        return cls

    return _register if cls is None else _register(cls)


def get_model(name):
    return _MODELS[name]


def create_model(config):
    """Create the model (reference ``models/utils.py:32-37``)."""
    model = get_model(config.model.name)(config)
    return model.to(config.device)


def get_model_fn(model, train=False):
    """Reference ``models/utils.py:40-70``.  Only evaluation mode exists on this path."""

    def model_fn(x, labels, *args, **kwargs):
        if train:
            raise NotImplementedError("training through the HIP path is a 'next' row (SURVEY.md 8f); use train=False")
        model.eval()
        return model(x, labels, *args, **kwargs)

    return model_fn


def get_score_fn(sde, model, train=False, continuous=False):
    """Reference ``models/utils.py:73-123`` with ``continuous=True``.  VPSDE: ``labels = 999 t`` and
    ``score = -model(x, labels, mask) / std``; VESDE: ``labels = sigma(t)`` and the model output is
    the score."""
    model_fn = get_model_fn(model, train=train)
    if isinstance(sde, sde_lib.VPSDE):
        if not continuous:
            # the reference's discrete branch reads a misspelt attribute and raises
            raise NotImplementedError("only continuous=True is supported (as in the reference's shipped configs)")

        def score_fn(x, t, *args, **kwargs):
            labels = t * 999
            score = model_fn(x, labels, *args, **kwargs)
            std = sde.marginal_prob(torch.zeros_like(x), t)[1]
            return -score / std[:, None, None]
    elif isinstance(sde, sde_lib.VESDE):
        if not continuous:
            raise NotImplementedError("only continuous=True is supported (as in the reference's shipped configs)")

        def score_fn(x, t, *args, **kwargs):
            labels = sde.marginal_prob(torch.zeros_like(x), t)[1]          # models/utils.py:109
            return model_fn(x, labels, *args, **kwargs)
    else:
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    return score_fn


def get_logit_fn(sde, classifier, train=False, continuous=False):
    """Reference ``models/utils.py:150-179`` (continuous): the surrogate is conditioned like the score net."""
    classifier_fn = get_model_fn(classifier, train=train)
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)) or not continuous:
        raise NotImplementedError("only a continuously trained VPSDE / VESDE surrogate is supported")

    def logit_fn(x, t, *args, **kwargs):
        return classifier_fn(x, sde.time_cond(t), *args, **kwargs)

    return logit_fn


def get_classifier_grad_fn(sde, classifier, train=False, continuous=False, regress=True, labels='max'):
    """Reference ``models/utils.py:126-147`` with ``regress=True``.  The reference differentiates
    ``+-logit.sum()`` with autograd; the surrogate module exposes the same quantity through the
    hand-derived backward pass of the HIP kernel (``input_grad``)."""
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)) or not continuous:
        raise NotImplementedError("only a continuously trained VPSDE / VESDE surrogate is supported")
    if not regress:
        raise NotImplementedError("classification guidance (regress=False) is not on the hot path")
    assert labels in ['max', 'min']

    def classifier_grad_fn(x, t, *args, **kwargs):
        if train:
            raise NotImplementedError("train=True is not supported on the HIP path")
        classifier.eval()
        grad = classifier.input_grad(x, sde.time_cond(t), *args, **kwargs)
        return grad if labels == 'max' else -grad

    return classifier_grad_fn
