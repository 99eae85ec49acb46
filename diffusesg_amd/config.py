"""Build the HIP network / sampler from the reference's YAML schema (R/config/edm_diffuse_sg/*.yaml).

Counterparts of `get_network` (R/utils/learning_utils.py:33-106) and `get_mc_sampler`
(R/utils/sampling_utils.py:8-31) that read the same keys:
  dataset.{name,max_node_num}  mcmc.{name,precond,num_steps,sample_clip.{min,max,scope}}
  model.{name,feature_dims[],depths[],window_size,patch_size}  train.{node_encoding,edge_encoding,self_cond,node_only}
plus the constants the reference hard-codes (num_heads=[3,6,12,24], mlp_ratio=4, noise_emb=512).
CLI overrides of the reference (`--num_steps`, `--node_encoding bits`, ...) map to keyword overrides here.
"""
from __future__ import annotations

from typing import Any, Dict

import yaml

from .spec import ModelConfig, sg_channels


def load_yaml(path_or_text: str) -> Dict[str, Any]:
    if "\n" in path_or_text or ":" in path_or_text and not path_or_text.endswith((".yaml", ".yml")):
        return yaml.safe_load(path_or_text)
    with open(path_or_text) as f:
        return yaml.safe_load(f)


def model_config_from_yaml(cfg: Dict[str, Any], **overrides) -> ModelConfig:
    model, train, ds = cfg["model"], dict(cfg["train"]), cfg["dataset"]
    train.update({k: v for k, v in overrides.items() if k in ("node_encoding", "edge_encoding", "self_cond")})
    if model["name"] not in ("diffuse_sg", "diffuse_sg_hip"):
        raise ValueError(f"Unknown model name {model['name']}")     # learning_utils.py:65
    if cfg["mcmc"]["name"] != "edm":
        raise NotImplementedError("only mcmc.name == 'edm' (sampling_utils.py:14)")
    if train.get("node_only", False):
        raise NotImplementedError("node_only ablation is out of scope")
    if train["node_encoding"] != train["edge_encoding"]:
        raise NotImplementedError("the channel table is keyed on train.node_encoding (sg_utils.py:418)")
    ch = sg_channels(ds["name"], train["node_encoding"])
    depths = tuple(model["depths"])
    return ModelConfig(max_node_num=int(ds["max_node_num"]), c_adj=ch["c_adj"], c_node=ch["c_node"],
                       embed_dim=int(model["feature_dims"][-1]), depths=depths, num_heads=(3, 6, 12, 24)[:len(depths)],
                       window_size=int(model["window_size"]), mlp_ratio=4, self_condition=bool(train["self_cond"]),
                       patch_size=int(model["patch_size"]))


def network_from_yaml(cfg: Dict[str, Any], state_dict=None, device="cuda", **overrides):
    from .model import build_network
    return build_network(model_config_from_yaml(cfg, **overrides), state_dict, device=device)


def sampler_from_yaml(cfg: Dict[str, Any], device="cuda", **overrides):
    from .sampler import NodeAdjEDMSamplerHip
    mc = cfg["mcmc"]
    clip = mc.get("sample_clip", {}) or {}
    flag_clip = clip.get("min") is not None and clip.get("max") is not None
    self_cond = overrides.get("self_cond", cfg["train"]["self_cond"])
    return NodeAdjEDMSamplerHip(num_steps=int(overrides.get("num_steps", mc["num_steps"])),
                                clip_samples=flag_clip, clip_samples_min=clip.get("min"), clip_samples_max=clip.get("max"),
                                clip_samples_scope=clip.get("scope", "x_0"), dev=device, objective="edm",
                                self_condition=bool(self_cond), symmetric_noise=False)
