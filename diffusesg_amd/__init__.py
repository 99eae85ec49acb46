"""diffusesg_amd: MI355X-native scene-graph diffusion sampler (denoiser forward + EDM reverse loop).

The product is libdsg.so (diffusesg_amd/csrc, include/dsg.h); this package is the thin Python host
side mirroring the reference's `model/` and `runner/mcmc_sampler/` callables."""
from .spec import ModelConfig, tiny_config, vg_config, coco_config  # noqa: F401

__all__ = ["ModelConfig", "tiny_config", "vg_config", "coco_config"]
