"""Parameter layouts of the two networks on the sampling hot path.

The hot path consumes checkpoints written by the reference, so the *names and
shapes* of the parameters are part of the drop-in boundary:

* score network ``CATE``            -- reference ``NAS-Bench-201/models/cate.py:108-164``
  (+ ``models/transformer.py:166-179`` for the encoder stack),
* surrogate ``MetaNeuralPredictor`` -- reference ``NAS-Bench-201/models/digcn_meta.py:56-107``
  (+ ``models/set_encoder/setenc_models.py:8-38`` for the two set pools).

Each spec entry is ``(dotted_name, shape, init)`` in ``state_dict()`` order, where
``init`` names the distribution the reference's constructor draws from (used only
by :mod:`diffusionnag_amd.synthetic` for random-init benchmarks and fixtures):

``("lin_w", fan_in)`` / ``("lin_b", fan_in)``  U(-1/sqrt(fan_in), 1/sqrt(fan_in))  (torch ``nn.Linear`` default)
``("ones",)`` / ``("zeros",)``                 LayerNorm affine
``("xavier", fan_in, fan_out)``                U(-a, a), a = sqrt(6/(fan_in+fan_out))
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

Spec = List[Tuple[str, Tuple[int, ...], tuple]]


@dataclass(frozen=True)
class CateDims:
    """Shape hyper-parameters of the CATE score network."""
    max_node: int = 8      # N  (NAS-Bench-201: 8, OFA: 20)
    n_vocab: int = 7       # V  (NAS-Bench-201: 7, OFA: 9)
    d_model: int = 64
    n_head: int = 8
    d_ff: int = 128
    n_layers: int = 12
    stagewise_pos: bool = False   # False: PositionalEncoding_Cell, True: _StageWise (OFA)

    @property
    def fdim(self) -> int:        # width of the all-layer concat
        return self.n_layers * self.d_model

    @property
    def hidden(self) -> int:      # head MLP hidden width (2 * fdim)
        return 2 * self.fdim


@dataclass(frozen=True)
class PredictorDims:
    """Shape hyper-parameters of MetaNeuralPredictor."""
    max_node: int = 8
    n_vocab: int = 7          # graph_encoder.initial_hidden
    gcn_hidden: int = 144
    gcn_layers: int = 4
    linear_hidden: int = 128
    hs: int = 32
    nz: int = 56
    num_sample: int = 20
    set_heads: int = 4        # SetPool default num_heads


def _linear(prefix: str, out_f: int, in_f: int, bias: bool = True) -> Spec:
    s: Spec = [(prefix + ".weight", (out_f, in_f), ("lin_w", in_f))]
    if bias:
        s.append((prefix + ".bias", (out_f,), ("lin_b", in_f)))
    return s


def _layernorm(prefix: str, d: int) -> Spec:
    return [(prefix + ".weight", (d,), ("ones",)), (prefix + ".bias", (d,), ("zeros",))]


def cate_param_spec(dims: CateDims = CateDims()) -> Spec:
    """``CATE.state_dict()`` names/shapes in order (reference cate.py:110-126, transformer.py:69-179)."""
    d, ff, V = dims.d_model, dims.d_ff, dims.n_vocab
    s: Spec = []
    s += _linear("opEmb.fc", d, V)
    s += _linear("timeEmb1", 4 * d, d)
    s += _linear("timeEmb2", d, 4 * d)
    for l in range(dims.n_layers):
        p = f"graph_encoder.encoder_f.layers.{l}"
        s += _layernorm(p + ".self_attn.norm", d)
        for j in range(4):  # 0=Q 1=K 2=V 3=O
            s += _linear(p + f".self_attn.attn.linears.{j}", d, d)
        s += _layernorm(p + ".feed_forward.norm", d)
        s += _linear(p + ".feed_forward.feed_forward.w_1", ff, d)
        s += _linear(p + ".feed_forward.feed_forward.w_2", d, ff)
    for l in range(dims.n_layers):
        s += _layernorm(f"graph_encoder.encoder_f.norms.{l}", d)
    s += _linear("final.linears.0", dims.hidden, dims.fdim)
    s += _linear("final.linears.1", dims.hidden, dims.hidden)
    s += _linear("final.linears.2", V, dims.hidden)
    return s


def _mab(prefix: str, dim_q: int, dim_k: int, dim_v: int) -> Spec:
    s: Spec = []
    s += _linear(prefix + ".fc_q", dim_v, dim_q)
    s += _linear(prefix + ".fc_k", dim_v, dim_k)
    s += _linear(prefix + ".fc_v", dim_v, dim_k)
    s += _linear(prefix + ".fc_o", dim_v, dim_v)
    return s


def _setpool(prefix: str, dim_in: int, nz: int) -> Spec:
    """SetPool(mode='sabPF'): enc = SAB,SAB ; dec = PMA(1 seed), Linear (setenc_models.py:12-23)."""
    s: Spec = []
    s += _mab(prefix + ".enc.0.mab", dim_in, dim_in, nz)
    s += _mab(prefix + ".enc.1.mab", nz, nz, nz)
    s.append((prefix + ".dec.0.S", (1, 1, nz), ("xavier", nz, nz)))   # fan of a (1,1,nz) tensor: (1*nz, 1*nz)
    s += _mab(prefix + ".dec.0.mab", nz, nz, nz)
    s += _linear(prefix + ".dec.1", nz, nz)
    return s


def predictor_param_spec(dims: PredictorDims = PredictorDims()) -> Spec:
    """``MetaNeuralPredictor.state_dict()`` names/shapes in order (reference digcn_meta.py:58-107)."""
    g, nz = dims.gcn_hidden, dims.nz
    s: Spec = []
    for l in range(dims.gcn_layers):
        fin = dims.n_vocab if l == 0 else g
        s.append((f"gcn.{l}.weight1", (fin, g), ("xavier", g, fin)))   # [in, out]: torch fan_in = size(1)
        s.append((f"gcn.{l}.weight2", (fin, g), ("xavier", g, fin)))
    s += _linear("fc1", dims.linear_hidden, g, bias=False)
    s += _linear("timeEmb1", 4 * g, g)
    s += _linear("timeEmb2", g, 4 * g)
    s += _setpool("intra_setpool", 512, nz)
    s += _setpool("inter_setpool", nz, nz)
    s += _linear("set_fc.0", nz, 512)            # defined but unused in forward (digcn_meta.py:90-92)
    s += _linear("pred_fc.0", dims.hs, nz + dims.linear_hidden)
    s += _linear("pred_fc.2", 1, dims.hs)
    return s


@dataclass(frozen=True)
class PredictorCateDims:
    """Shape hyper-parameters of MetaPredictorCATE, the OFA surrogate (SURVEY.md section 5.1;
    reference MobileNetV3/configs/tr_meta_surrogate_ofa.py:112-145)."""
    max_node: int = 20
    n_vocab: int = 9
    d_model: int = 64
    n_head: int = 2
    d_ff: int = 32
    n_layers: int = 2
    hs: int = 512
    nz: int = 56
    num_sample: int = 20
    set_heads: int = 4

    @property
    def fdim(self) -> int:
        return self.n_layers * self.d_model

    @property
    def rdim(self) -> int:
        return self.max_node * self.n_vocab


def predictor_cate_param_spec(dims: PredictorCateDims = PredictorCateDims()) -> Spec:
    """``MetaPredictorCATE.state_dict()`` names/shapes in order (reference cate.py:253-305)."""
    d, ff, V, nz = dims.d_model, dims.d_ff, dims.n_vocab, dims.nz
    s: Spec = []
    s += _linear("opEmb.fc", d, V)
    s += _linear("timeEmb1", 4 * d, d)
    s += _linear("timeEmb2", d, 4 * d)
    for l in range(dims.n_layers):
        p = f"graph_encoder.encoder_f.layers.{l}"
        s += _layernorm(p + ".self_attn.norm", d)
        for j in range(4):
            s += _linear(p + f".self_attn.attn.linears.{j}", d, d)
        s += _layernorm(p + ".feed_forward.norm", d)
        s += _linear(p + ".feed_forward.feed_forward.w_1", ff, d)
        s += _linear(p + ".feed_forward.feed_forward.w_2", d, ff)
    for l in range(dims.n_layers):
        s += _layernorm(f"graph_encoder.encoder_f.norms.{l}", d)
    F, R = dims.fdim, dims.rdim
    s += _linear("final.linears.0", 2 * F, F)
    s += _linear("final.linears.1", 2 * F, 2 * F)
    s += _linear("final.linears.2", V, 2 * F)
    s += _linear("regeress.linears.0", 2 * R, R)
    s += _linear("regeress.linears.1", 2 * R, 2 * R)
    s += _setpool("intra_setpool", 512, nz)
    s += _setpool("inter_setpool", nz, nz)
    s += _linear("set_fc.0", nz, 512)
    s += _linear("pred_fc.0", dims.hs, nz + 2 * R)
    s += _linear("pred_fc.2", 1, dims.hs)
    return s


def n_params(spec: Spec) -> int:
    n = 0
    for _, shape, _ in spec:
        k = 1
        for v in shape:
            k *= v
        n += k
    return n
