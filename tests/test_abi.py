"""CPU: the C-ABI library loads and exports every symbol include/dsg.h declares; the host-only schedule helper
matches the reference's sigma tables; the state-dict layout matches the golden generator's check."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from diffusesg_amd import lib, spec
from diffusesg_amd import synth as Y
from util import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "dsg.h")).read()
    declared = set(re.findall(r"\b(dsg_[a-z_0-9]+)\s*\(", hdr)) - {"dsg_handle_s"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    L = lib.load()
    for name in declared:
        assert hasattr(L, name), f"libdsg.so does not export {name}"
    assert b"gfx950" in L.dsg_version()


@pytest.mark.parametrize("T", [50, 100, 256, 1000])
def test_sigma_schedule_matches_reference(T):
    g = load("sampler.npz")
    sg, th, nz, hs = lib.sigma_schedule(lib.make_sampler_cfg(T))
    np.testing.assert_allclose(sg, g[f"sigma_steps_{T}"], rtol=1e-14, atol=0)
    t = sg.astype(np.float32)
    # churn only inside [S_min, S_max]; never when it is off (a fused multiply-add would leave a residue here)
    off = (t < np.float32(0.05)) | (t > np.float32(50))
    assert np.all(nz[off] == 0) and np.all(th[off] == t[off])
    assert np.all(nz[~off] > 0)
    _, th0, nz0, _ = lib.sigma_schedule(lib.make_sampler_cfg(T, S_churn=0.0))
    assert np.all(nz0 == 0) and np.array_equal(th0, t)
    assert hs[-1] == -th[-1]   # t_N = 0 (edm.py:319)


def test_structs_match_header_layout():
    assert C.sizeof(lib.DsgConfig) == 4 * (5 + 8 + 8 + 3)
    assert C.sizeof(lib.DsgSamplerCfg) == 56
    assert C.sizeof(lib.DsgSampleStats) == 24


def test_param_counts_and_flops():
    assert spec.num_parameters(spec.vg_config()) == 35_813_660      # SURVEY §6 (measured from the reference)
    assert spec.num_parameters(spec.coco_config()) == 30_693_965
    assert spec.num_parameters(spec.tiny_config()) == 2_402_756
    assert abs(spec.flops_per_forward(spec.vg_config()) / 13.30e9 - 1) < 0.01
    assert abs(spec.flops_per_forward(spec.coco_config()) / 7.36e9 - 1) < 0.01


def test_channel_table():
    vg, coco = spec.sg_channels("visual_genome", "bits"), spec.sg_channels("coco_stuff", "bits")
    assert (vg["c_adj"], vg["c_node"], vg["in_chans"]) == (6, 12, 30)
    assert (coco["c_adj"], coco["c_node"], coco["in_chans"]) == (3, 12, 27)
    d = spec.sg_channels("visual_genome", "ddpm")
    assert (d["c_adj"], d["c_node"], d["in_chans"]) == (1, 5, 11)
    o = spec.sg_channels("visual_genome", "one_hot")
    assert (o["c_adj"], o["c_node"], o["in_chans"]) == (51, 154, 359)


def test_module_state_dict_keys_match_spec():
    from diffusesg_amd.model import build_network
    for name in ("tiny", "small", "vg"):
        cfg = Y.CONFIGS[name]()
        m = build_network(cfg, device="cpu")
        keys = set(m.state_dict().keys())
        assert keys == {"model." + t.key for t in spec.state_dict_spec(cfg)}
    assert len(build_network(spec.vg_config(), device="cpu").state_dict()) == 247


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffusesg_amd.model import build_network
    from diffusesg_amd import weights as W
    cfg = spec.tiny_config()
    net = build_network(cfg, W.synth_state_dict(cfg, 0), device="cpu")
    flags, adj, node, _, _ = Y.case_inputs(cfg, 2, [8, 5], 1, "x")
    with pytest.raises(lib.DsgError):
        net.model(torch.from_numpy(adj), torch.from_numpy(node), torch.from_numpy(flags), torch.zeros(2))


def test_gelu_table_accuracy():
    """numpy restatement of the table-driven GELU of csrc/kernels.hip (gelu_lut): nodes every 1/64 on [-6,6] holding
    (Phi, phi, -x*phi/2), second-order Taylor from the nearest node.  Max abs error vs an fp64 exact-erf GELU."""
    from scipy.special import erfc
    xs = np.linspace(-6.0, 6.0, 769)
    phi = np.exp(-0.5 * xs * xs) / np.sqrt(2 * np.pi)
    tab = np.stack([0.5 * erfc(-xs / np.sqrt(2)), phi, -0.5 * xs * phi], 1).astype(np.float32)
    x = np.concatenate([np.linspace(-9, 9, 400001), np.random.RandomState(0).randn(200000) * 2]).astype(np.float32)
    t = np.clip(x + np.float32(6), np.float32(0), np.float32(12)).astype(np.float32)
    r = np.rint(t * np.float32(64)).astype(np.float32)
    d = (r * np.float32(-0.015625) + t).astype(np.float32)
    c = tab[r.astype(np.int64)]
    g = (x * (d * (d * c[:, 2] + c[:, 1]) + c[:, 0])).astype(np.float32)
    ref = 0.5 * x.astype(np.float64) * erfc(-x.astype(np.float64) / np.sqrt(2))
    assert np.abs(g - ref).max() < 6e-7      # same level as an fp32 evaluation of the erf formula (4.5e-7)
