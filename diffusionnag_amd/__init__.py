"""MI355X-native reverse-diffusion sampling hot path of DiffusionNAG.

Host mirrors of the reference's modules (same names and call signatures):
``sde_lib`` (VPSDE), ``sampling`` (get_sampling_fn, registries), ``mutils`` (create_model,
get_score_fn, get_classifier_grad_fn), ``cate`` (CATE), ``digcn_meta`` (MetaNeuralPredictor), ``predictor_cate`` (MetaPredictorCATE),
``utils`` (quantize, aug_mask, scalers).  All compute goes through ``libdnag.so`` (``_lib``).
"""

__all__ = ["sde_lib", "sampling", "mutils", "cate", "digcn_meta", "predictor_cate", "utils", "specs", "synthetic"]
