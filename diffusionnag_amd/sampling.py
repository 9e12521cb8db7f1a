"""Host mirror of the reference's ``sampling.py`` factory surface (reference ``sampling.py:14-133``,
``:380-447``, ``:450-579``): the same registries, ``get_sampling_fn`` signature and sampler call
signatures / return values.  The loop itself -- Langevin corrector, reverse-diffusion predictor,
score network, surrogate gradient, per-step noise -- runs in ``dnag_pc_sample`` (``libdnag.so``).
"""

from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib, sde_lib
from .cate import CATE, Workspace
from .digcn_meta import MetaNeuralPredictor
from .predictor_cate import MetaPredictorCATE
from .utils import MetaTestDataset

DATA_PATH = "./data/transfer_nag"      # reference all_path.py

_CORRECTORS = {}
_PREDICTORS = {}


def register_predictor(cls=None, *, name=None):
    """A decorator for registering predictor classes (reference ``sampling.py:18-34``)."""

    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _PREDICTORS:
            raise ValueError(f'Already registered predictor with name: {local_name}')
        _PREDICTORS[local_name] = cls
        return cls

    return _register if cls is None else _register(cls)


def register_corrector(cls=None, *, name=None):
    """A decorator for registering corrector classes (reference ``sampling.py:37-53``)."""

    def _register(cls):
        local_name = cls.__name__ if name is None else name
        if local_name in _CORRECTORS:
            raise ValueError(f'Already registered corrector with name: {local_name}')
        _CORRECTORS[local_name] = cls
        return cls

    return _register if cls is None else _register(cls)


def get_predictor(name):
    return _PREDICTORS[name]


def get_corrector(name):
    return _CORRECTORS[name]


# The registered classes only select which fused update the library runs: the update functions
# themselves (reference ``sampling.py:189-263``) are device code in csrc/dnag_head.hip.h.
@register_predictor(name='reverse_diffusion')
class ReverseDiffusionPredictor:
    """Reference ``sampling.py:203-213``; fused into ``head_out_kernel`` (EXIT_PREDICTOR)."""
    fused = True
    kind = _lib.PREDICTOR_REVERSE_DIFFUSION
    table = "reverse_diffusion"


@register_predictor(name='euler_maruyama')
class EulerMaruyamaPredictor:
    """Reference ``sampling.py:189-200`` (the shipped default); the same exit kernel with the
    Euler-Maruyama scalars of ``sde_lib.schedule_table``."""
    fused = True
    kind = _lib.PREDICTOR_EULER_MARUYAMA
    table = "euler_maruyama"


@register_predictor(name='none')
class NonePredictor:
    """Reference ``sampling.py:216-224``: identity, no score evaluation."""
    fused = True
    kind = _lib.PREDICTOR_NONE
    table = "reverse_diffusion"


@register_corrector(name='langevin')
class LangevinCorrector:
    """Reference ``sampling.py:227-256``; ``head_out_kernel`` (EXIT_CORRECTOR) + ``corrector_update_kernel``."""
    fused = True
    enabled = 1


@register_corrector(name='none')
class NoneCorrector:
    """Reference ``sampling.py:259-267``: identity."""
    fused = True
    enabled = 0


def get_sampling_fn(config, sde, shape, inverse_scaler, eps, conditional=False, data_name='cifar10', num_sample=20,
                    p=1, prod_w=False, weight_ratio_abs=False):
    """Create a sampling function (reference ``sampling.py:64-133``, same arguments).  ``p``, ``prod_w``,
    ``weight_ratio_abs`` and ``config.sampling.{weight_scheduling, weight_ratio, t_spot, t_spot_end}`` are the guidance
    weighting options of the MobileNetV3 tree's factory (``MobileNetV3/sampling.py:72-190``); absent = the
    NAS-Bench-201 tree's behaviour."""
    sampler_name = config.sampling.method
    if sampler_name.lower() == 'pc':
        predictor = get_predictor(config.sampling.predictor.lower())
        corrector = get_corrector(config.sampling.corrector.lower())
        noise = config.sampling.noise if 'noise' in config.sampling else 'philox'
        common = dict(sde=sde, shape=shape, predictor=predictor, corrector=corrector, inverse_scaler=inverse_scaler,
                      snr=config.sampling.snr, n_steps=config.sampling.n_steps_each,
                      probability_flow=config.sampling.probability_flow, continuous=config.training.continuous,
                      denoise=config.sampling.noise_removal, eps=eps, device=config.device, noise=noise)
        if not conditional:
            return get_pc_sampler_nas(**common)
        sc = config.sampling
        weighting = dict(weight_scheduling=bool(sc.weight_scheduling) if 'weight_scheduling' in sc else False,
                         weight_ratio=bool(sc.weight_ratio) if 'weight_ratio' in sc else False,
                         t_spot=float(sc.t_spot) if 't_spot' in sc else 1.,
                         t_spot_end=sc.t_spot_end if 't_spot_end' in sc else None,
                         p=p, prod_w=bool(prod_w), weight_ratio_abs=bool(weight_ratio_abs))
        return get_pc_conditional_sampler_meta_nas(regress=config.sampling.regress, labels=config.sampling.labels,
                                                   data_name=data_name, num_sample=num_sample, weighting=weighting, **common)
    raise NotImplementedError(f"Sampler name {sampler_name} unknown.")


class _PCSampler:
    """State shared by the two sampler callables.  Besides the reference's call signature it has a
    few attributes for tests and benchmarks: ``noise_tape`` (dict of device tensors ``x0 [B,N,V]``,
    ``corrector [S,B,N,V]``, ``predictor [S,B,N,V]`` consumed in the reference's RNG order),
    ``max_steps`` (run only the first steps of the time grid), ``record`` (keep per-step tensors in
    ``last_record``), ``seed`` / ``stream_offset`` (Philox key and first element index of this shard),
    ``langevin_batch`` (``'shard'``: the Langevin step size of ``sampling.py:249-252`` is a mean over this process's
    batch, the default; ``'global'``: over the batches of all ranks of the ``torch.distributed`` group, exchanged per
    corrector update, ``distributed.LangevinExchange``).  With ``n_steps_each = c > 1`` the corrector tape and the
    corrector records hold ``S * c`` entries (entry ``i * c + k`` for update ``k`` of PC step ``i``)."""

    def __init__(self, sde, shape, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow, continuous,
                 denoise, eps, device, noise):
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        if not getattr(predictor, "fused", False):
            raise NotImplementedError(f"predictor {predictor.__name__} is not built")
        if not getattr(corrector, "fused", False):
            raise NotImplementedError(f"corrector {corrector.__name__} is not built (only 'langevin' and 'none')")
        if probability_flow:
            raise NotImplementedError("probability_flow=True is not built")
        if not continuous:
            raise NotImplementedError("only continuous=True is supported (SURVEY.md 5.2-6)")
        if int(n_steps) != n_steps or n_steps < 1:
            raise ValueError(f"n_steps_each={n_steps!r}: expected a positive integer")
        if noise not in ("philox", "torch"):
            raise ValueError(f"config.sampling.noise={noise!r}: expected 'philox' or 'torch'")
        self.sde, self.shape, self.corrector, self.predictor = sde, tuple(shape), corrector, predictor
        self.inverse_scaler, self.snr, self.n_steps, self.denoise = inverse_scaler, snr, n_steps, denoise
        self.eps, self.device, self.noise = eps, torch.device(device), noise
        self.noise_tape: Optional[dict] = None
        self.max_steps: Optional[int] = None
        self.record = False
        self.last_record: Optional[dict] = None
        self.seed = 0
        self.stream_offset = 0
        self.langevin_batch = "shard"
        self.last_exchange = None
        self.weighting: Optional[dict] = None
        self.last_chain: Optional[torch.Tensor] = None
        self._ws = Workspace()
        self._calls = 0

    def _guidance_schedule(self, S, classifier_scale):
        """Per-step guidance weight and on/off flag of the MobileNetV3 sampler's ``total_grad_fn``
        (``MobileNetV3/sampling.py:858-859``, ``:875-878``, ``:902-910``), computed like the reference: python floats
        (double) from the fp32 time grid.  Returns ``(cscale_steps, guide_steps, mode, p, prod_w)`` or None."""
        wo = self.weighting
        if not wo:
            return None
        mode = 1 if wo.get("weight_ratio") else (2 if wo.get("weight_ratio_abs") else 0)
        if wo.get("weight_ratio") and wo.get("weight_ratio_abs"):
            raise AssertionError("weight_ratio and weight_ratio_abs are exclusive (MobileNetV3/sampling.py:889)")
        t_spot = wo.get("t_spot", 1.)
        if not (wo.get("weight_scheduling") or mode or t_spot < 1.):
            return None
        t_end = wo.get("t_spot_end")
        if t_end is None or t_end == 0.:
            t_end = self.eps
        ts = torch.linspace(self.sde.T, self.eps, self.sde.N)[:S]
        cs = np.empty(S, np.float32)
        on = np.ones(S, np.int32)
        for i in range(S):
            t0 = ts[i].item()
            cs[i] = classifier_scale * 0.1 ** t0 if wo.get("weight_scheduling") else classifier_scale
            if t_spot < 1. and not (t0 <= t_spot and bool(ts[i] >= t_end)):
                on[i] = 0
        return cs, on, mode, float(wo.get("p", 1)), int(bool(wo.get("prod_w")))

    # ---------------------------------------------------------------------------------------------
    def _run(self, score_model: CATE, mask, classifier=None, classifier_scale=None, task=None, labels='max', chain=None):
        lib = _lib.load()
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("diffusionnag_amd samplers run on an MI355X only (config.device must be a cuda device)")
        if not isinstance(score_model, CATE):
            raise TypeError("score_model must be diffusionnag_amd.cate.CATE (create it with mutils.create_model)")
        B, N, V = self.shape
        S = self.sde.N if self.max_steps is None else int(self.max_steps)
        sched = self.sde.schedule_table(self.eps, S, self.predictor.table)
        score_model.eval()
        cd = score_model.dims_c()
        if (N, V) != (cd.max_node, cd.n_vocab):
            raise ValueError(f"sampler shape {self.shape} does not match the score network ({cd.max_node}, {cd.n_vocab})")
        m = torch.as_tensor(mask, dtype=torch.float32)
        if m.dim() == 3:
            m = m[0]                                                 # sampling.py:438 / :568
        m = m.to(dev).contiguous()
        args = _lib.PcArgsC()
        keep = [sched, m]
        args.cate_dims = C.pointer(cd)
        args.cate_packed = _lib.ptr(score_model.packed(dev))
        pd = None
        if classifier is not None:
            if not isinstance(classifier, (MetaNeuralPredictor, MetaPredictorCATE)):
                raise TypeError("classifier must be diffusionnag_amd's MetaNeuralPredictor or MetaPredictorCATE")
            if classifier_scale is None:
                raise ValueError("classifier_scale is required for guided sampling")
            classifier.eval()
            classifier.sample_state = True                           # sampling.py:561-562
            classifier.D_mu = None
            d_mu = classifier._dataset_embedding(task.to(dev), dev)
            pd = classifier.dims_c()
            if isinstance(classifier, MetaPredictorCATE):
                args.predcate_dims = C.pointer(pd)
                args.predcate_packed = _lib.ptr(classifier.packed(dev))
            else:
                args.pred_dims = C.pointer(pd)
                args.pred_packed = _lib.ptr(classifier.packed(dev))
            args.d_mu = _lib.ptr(d_mu)
            args.classifier_scale = float(classifier_scale)
            args.label_sign = 1.0 if labels == 'max' else -1.0
            keep += [d_mu]
            gs = self._guidance_schedule(S, float(classifier_scale))
            if gs is not None:
                cs, on, mode, pnorm, prod_w = gs
                args.cscale_steps, args.guide_steps = cs.ctypes.data, on.ctypes.data
                args.weight_ratio, args.weight_ratio_p, args.prod_w = mode, pnorm, prod_w
                keep += [cs, on]
        args.batch, args.n_steps = B, S
        args.mask = _lib.ptr(m)
        args.sched = sched.ctypes.data
        args.snr = float(self.snr)
        args.corrector = int(self.corrector.enabled)
        CS = int(self.n_steps)                                       # corrector updates per PC step (sampling.py:244)
        args.corrector_steps = CS
        args.predictor = int(self.predictor.kind)
        args.score_mode = int(self.sde.score_mode)
        args.prior_scale = float(self.sde.prior_scale)
        # noise ------------------------------------------------------------------------------------
        tape = self.noise_tape
        if tape is None:
            x0 = self.sde.prior_sampling(self.shape).to(dev)         # CPU generator, then H2D (sampling.py:436)
            if self.noise == "torch":
                # the reference draws eps then z with randn_like on the device generator, step by step
                nc = torch.empty(S * CS, B, N, V, device=dev)
                npd = torch.empty(S, B, N, V, device=dev)
                for i in range(S):
                    if self.corrector.enabled:
                        for k in range(CS):
                            nc[i * CS + k] = torch.randn_like(x0)
                    if self.predictor.kind != _lib.PREDICTOR_NONE:
                        npd[i] = torch.randn_like(x0)
                tape = {"x0": x0, "corrector": nc, "predictor": npd}
            else:
                tape = {"x0": x0, "corrector": None, "predictor": None}
        x0 = tape["x0"].to(dev, torch.float32).contiguous()
        nc = tape.get("corrector")
        npd = tape.get("predictor")
        if (nc is None) != (npd is None):
            raise ValueError("noise_tape needs both 'corrector' and 'predictor' or neither")
        if nc is not None:
            nc = nc.to(dev, torch.float32).contiguous()
            npd = npd.to(dev, torch.float32).contiguous()
            if nc.shape[0] < S * CS or npd.shape[0] < S or tuple(nc.shape[1:]) != (B, N, V):
                raise ValueError("noise_tape is shorter than the number of steps or has the wrong shape")
        keep += [x0, nc, npd]
        args.x0, args.noise_corr, args.noise_pred = _lib.ptr(x0), _lib.ptr(nc), _lib.ptr(npd)
        self._calls += 1
        args.seed = int(self.seed) + 0x9E3779B97F4A7C15 * self._calls & 0xFFFFFFFFFFFFFFFF if nc is None else int(self.seed)
        args.stream_offset = int(self.stream_offset)
        # outputs ----------------------------------------------------------------------------------
        out = torch.empty(B, N, V, dtype=torch.float32, device=dev)
        x_last = torch.empty_like(out)
        x_mean = torch.empty_like(out)
        args.x_out, args.x_last, args.x_mean_last = _lib.ptr(out), _lib.ptr(x_last), _lib.ptr(x_mean)
        chain_x = None
        if chain is not None:                                        # MobileNetV3/sampling.py:980-988
            ncs = chain.get("number_chain_steps") or self.sde.N
            kc = chain.get("keep_chain") or B
            assert ncs <= self.sde.N
            chain_x = torch.zeros(ncs, kc, N, V, dtype=torch.float32, device=dev)
            args.chain_x, args.chain_steps, args.chain_keep = _lib.ptr(chain_x), int(ncs), int(kc)
            args.chain_n, args.chain_denoise = int(self.sde.N), int(bool(self.denoise))
        rec = None
        if self.record:
            names = ["rec_score_c", "rec_total_c", "rec_x_corr", "rec_score_p", "rec_total_p", "rec_x_mean", "rec_x_pred"]
            if classifier is not None:
                names += ["rec_cgrad_c", "rec_cgrad_p"]
            per_corr = ("rec_score_c", "rec_total_c", "rec_x_corr", "rec_cgrad_c")
            rec = {n: torch.zeros(S * CS if n in per_corr else S, B, N, V, dtype=torch.float32, device=dev) for n in names}
            rec["rec_step_size"] = torch.zeros(S * CS, 3, dtype=torch.float32, device=dev)
            for n, t in rec.items():
                setattr(args, n, _lib.ptr(t))
        if isinstance(classifier, MetaPredictorCATE):
            nbytes = lib.dnag_workspace_bytes_predcate(C.byref(cd), C.byref(pd), B, S)
        else:
            nbytes = lib.dnag_workspace_bytes(C.byref(cd), C.byref(pd) if pd is not None else None, B, S)
        ws = self._ws.get(nbytes, dev)
        args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        args.stream = _lib.current_stream(dev)
        exch = cb = None
        if self.langevin_batch == "global" and self.corrector.enabled:
            from .distributed import LangevinExchange
            exch = LangevinExchange(lib.dnag_langevin_partials_floats(C.byref(cd), B), dev)

            def _exchange(_user, _stream):          # called by dnag_pc_sample once per corrector update, on this thread
                try:
                    exch()
                    return 0
                except BaseException as e:          # no exception crosses the C ABI
                    exch.error = e
                    return -1
            cb = _lib.LANGEVIN_EXCHANGE_FN(_exchange)
            args.langevin_exchange = C.cast(cb, C.c_void_p)
            args.langevin_local, args.langevin_global = _lib.ptr(exch.local), _lib.ptr(exch.global_)
            args.langevin_world = exch.world
        elif self.langevin_batch not in ("shard", "global"):
            raise ValueError(f"langevin_batch={self.langevin_batch!r}: expected 'shard' or 'global'")
        with torch.cuda.device(dev):
            rc = lib.dnag_pc_sample(C.byref(args))
        if exch is not None and exch.error is not None:
            raise RuntimeError("the Langevin exchange (torch.distributed all_gather) failed inside dnag_pc_sample") from exch.error
        _lib.check(rc, "dnag_pc_sample")
        self.last_exchange = exch
        del cb
        if classifier is not None:
            classifier.sample_state = False                          # sampling.py:576
        self.last_record = rec
        del keep
        # the reference keeps the chain on the host (torch.zeros on the default device, filled from the GPU)
        self.last_chain = None
        if chain_x is not None:
            ncs = chain_x.shape[0]
            hit = sorted({ncs - 1 - (i * ncs) // self.sde.N for i in range(S)})    # slots a truncated run never reaches stay 0
            chain_x[hit] = self.inverse_scaler(chain_x[hit])
            self.last_chain = chain_x.cpu()
        kind = getattr(self.inverse_scaler, "dnag_kind", None)
        if self.denoise and kind == "centered":
            return out                                               # (x_mean + 1) / 2 from the exit kernel
        raw = x_mean if self.denoise else x_last
        return self.inverse_scaler(raw)


class _PCUnconditional(_PCSampler):
    def __call__(self, model, mask):
        """Reference ``pc_sampler(model, mask)`` (``sampling.py:424-445``): returns
        ``(samples, sde.N * (n_steps + 1), None)``."""
        x = self._run(model, mask)
        return x, self.sde.N * (self.n_steps + 1), None


class _PCConditional(_PCSampler):
    def __init__(self, *a, regress=True, labels='max', data_name='cifar10', num_sample=20, weighting=None, **k):
        super().__init__(*a, **k)
        self.weighting = weighting
        if not regress:
            raise NotImplementedError("classification guidance (regress=False) is not built")
        assert labels in ['max', 'min']
        self.labels = labels
        # the reference opens the feature blob here (sampling.py:490-493); it is opened on first use instead
        self.test_dataset = MetaTestDataset(data_path=DATA_PATH, data_name=data_name, num_sample=num_sample)

    def __call__(self, score_model, mask, classifier, classifier_scale=None, task=None, eval_chain=None,
                 keep_chain=None, number_chain_steps=None, sample_bs=None):
        """Reference ``pc_conditional_sampler`` (``sampling.py:534-577``): returns the samples.

        Called with ``eval_chain`` (True or False), it answers like the MobileNetV3 tree's sampler
        (``MobileNetV3/sampling.py:965-1037``): ``(samples, sde.N * (n_steps + 1), chain_x, norms)`` with ``chain_x``
        ``[number_chain_steps, keep_chain, N, V]`` on the host (None unless ``eval_chain``); ``norms`` stands for that
        sampler's four per-step gradient-norm lists, a logging aid that is not recorded here (four empty lists)."""
        if task is None:
            task = self.test_dataset[0]
        chain = dict(number_chain_steps=number_chain_steps, keep_chain=keep_chain) if eval_chain else None
        x = self._run(score_model, mask, classifier, classifier_scale, task, self.labels, chain=chain)
        if eval_chain is None:
            return x
        return x, self.sde.N * (self.n_steps + 1), self.last_chain, ([], [], [], [])


def get_pc_sampler_nas(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False,
                       continuous=False, denoise=True, eps=1e-3, device='cuda', noise='philox'):
    """Reference ``sampling.py:380-447``."""
    return _PCUnconditional(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow,
                            continuous, denoise, eps, device, noise)


def get_pc_conditional_sampler_meta_nas(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1,
                                        probability_flow=False, continuous=False, denoise=True, eps=1e-5,
                                        device='cuda', regress=True, labels='max', data_name='cifar10',
                                        num_sample=20, noise='philox', weighting=None):
    """Reference ``sampling.py:450-579``; ``weighting``: the extra options of ``MobileNetV3/sampling.py:816-823``."""
    return _PCConditional(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps, probability_flow,
                          continuous, denoise, eps, device, noise, regress=regress, labels=labels,
                          data_name=data_name, num_sample=num_sample, weighting=weighting)
