"""CPU: host-side logic -- YAML schema, batch sharding, the packed gather over world-size-2 gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusesg_amd import config as dcfg
from diffusesg_amd import dist as ddist
from diffusesg_amd import spec

VG_YAML = """
seed: 1234
dataset: {name: visual_genome, max_node_num: 64, subset: null}
mcmc:
  name: edm
  precond: edm
  sigma_dist: edm
  num_steps: 256
  sample_clip: {min: -1.0, max: 1.0, scope: x_0}
model: {name: diffuse_sg, feature_dims: [96], depths: [1, 1, 3, 1], window_size: 8, patch_size: 1}
test: {batch_size: 0, eval_size: 0}
train: {batch_size: 1000, node_encoding: ddpm, edge_encoding: ddpm, self_cond: true, node_only: false, binary_edge: false}
"""


def test_yaml_schema_to_model_config():
    y = dcfg.load_yaml(VG_YAML)
    c = dcfg.model_config_from_yaml(y, node_encoding="bits", edge_encoding="bits")   # README: --node_encoding bits
    assert c == spec.vg_config()
    d = dcfg.model_config_from_yaml(y)            # yaml default 'ddpm'
    assert (d.c_adj, d.c_node, d.in_chans) == (1, 5, 22)
    y["model"]["name"] = "something_else"
    with pytest.raises(ValueError):
        dcfg.model_config_from_yaml(y)


def test_sampler_from_yaml_host_only():
    y = dcfg.load_yaml(VG_YAML)
    s = dcfg.sampler_from_yaml(y, device="cpu", num_steps=100)
    assert s.num_steps == 100 and s.solver == "heun" and s.clip_samples
    assert abs(float(s.sigma_steps[0]) - 80.0) < 1e-12 and abs(float(s.sigma_steps[-1]) - 0.002) < 1e-12


def test_shard_and_seed():
    assert ddist.shard_batch(2048, 8) == 256 and ddist.shard_batch(67, 8) == 8
    assert ddist.rank_seed(1234, 3) == 1237
    with pytest.raises(ValueError):
        ddist.shard_batch(4, 8)


def test_pack_roundtrip():
    a, n = torch.randn(5, 6, 8, 8), torch.randn(5, 8, 12)
    p = ddist.pack_results(a, n)
    assert p.shape == (5, 6 * 64 + 96) and p.is_contiguous()
    a2, n2 = ddist.unpack_results(p, 6, 8, 12)
    assert torch.equal(a, a2) and torch.equal(n, n2)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = ddist.shard_batch(6, world)
        g = torch.Generator().manual_seed(ddist.rank_seed(1234, rank))
        adj, node = torch.randn(B, 3, 4, 4, generator=g), torch.randn(B, 4, 5, generator=g)
        out = ddist.gather_results(ddist.pack_results(adj, node))
        # the decoded hand-off is int16, which RCCL cannot carry: gather_results moves it as bytes
        from diffusesg_amd import io as dio
        qa = torch.randint(0, 51, (B, 4, 4), generator=g, dtype=torch.int32)
        qn = torch.randint(0, 150, (B, 4), generator=g, dtype=torch.int32)
        fl = torch.ones(B, 4, dtype=torch.bool)
        bb = torch.rand(B, 4, 4, generator=g)
        dec = ddist.gather_results(dio.pack_decoded(qa, qn, bb, fl))
        assert dec.dtype == torch.int16 and dec.shape[0] == world * B
        a2, n2, f2, b2 = dio.unpack_decoded(dec[rank * B:(rank + 1) * B], 4, True)
        assert torch.equal(a2, qa) and torch.equal(n2, qn) and torch.equal(b2, bb)
        q.put((rank, out.numpy(), adj.numpy(), node.numpy()))
    finally:
        dist.destroy_process_group()


def test_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    full0, full1 = res[0][1], res[1][1]
    assert np.array_equal(full0, full1) and full0.shape == (6, 3 * 16 + 20)       # every rank holds all graphs
    for r in range(2):                                                           # rank order, contents intact
        a, n = ddist.unpack_results(torch.from_numpy(full0[3 * r:3 * r + 3]), 3, 4, 5)
        assert np.array_equal(a.numpy(), res[r][2]) and np.array_equal(n.numpy(), res[r][3])
    assert not np.array_equal(res[0][2], res[1][2])                              # seeds differ by rank


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        grads = {"a.weight": torch.randn(7, 5, generator=g), "a.bias": torch.randn(7, generator=g), "b.weight": torch.randn(300, 40, generator=g)}
        mine = {k: v.clone() for k, v in grads.items()}
        out = ddist.all_reduce_mean(grads, bucket_bytes=4096)   # small buckets: several collectives, one tensor larger than a bucket
        assert out is grads
        # the flat form train_step_grads returns (train.GradDict: every gradient a view into one buffer): ONE collective, in place
        from diffusesg_amd.train import GradDict
        fg = GradDict()
        fg.flat = torch.cat([mine[k].reshape(-1) for k in mine]).clone()
        off = 0
        for k, v in mine.items():
            fg[k] = fg.flat[off:off + v.numel()].view(v.shape)
            off += v.numel()
        assert ddist.all_reduce_mean(fg) is fg
        for k in grads:
            assert torch.allclose(fg[k], grads[k], atol=1e-7), k
        q.put((rank, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in mine.items()}))
    finally:
        dist.destroy_process_group()


def test_gradient_all_reduce_mean_world2_gloo():
    """data-parallel training: every rank ends with the mean of the ranks' gradients (what DDP's backward leaves), bucketed"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in res[0][1]:
        mean = (res[0][2][k] + res[1][2][k]) / 2
        assert np.allclose(res[0][1][k], mean, atol=1e-7) and np.array_equal(res[0][1][k], res[1][1][k])
    g = {"w": torch.ones(3)}
    assert ddist.all_reduce_mean(g) is g and torch.equal(g["w"], torch.ones(3))   # identity when not distributed


def reference_style_checkpoint(cfg, path, numpy1_names=False):
    """A file shaped like the one `get_ckpt_data` writes (trainer_utils.py:168-185): 'model' keys carry the precond
    wrapper's 'model.' prefix, `config` is the nested `config.to_dict()` (incl. the torch.device of arg_parser.py:352),
    `train_loss` / `test_loss` are NumPy float64 scalars (np.concatenate(...).mean(), :160-161), one EMA state dict per
    beta, here saved from a DDP-wrapped model ('module.' prefix).  numpy1_names rewrites the pickle to the module path
    NumPy 1.x (the reference's pin) stores for a scalar."""
    import zipfile
    from diffusesg_amd import weights as W
    sd = {k: torch.from_numpy(v) for k, v in W.synth_state_dict(cfg, 0, prefix="model.").items()}
    ema = {"module." + k: v * 0.5 if v.dtype == torch.float32 else v for k, v in sd.items()}
    conf = {"seed": 1234, "dev": torch.device("cpu"), "dataset": {"name": "visual_genome", "max_node_num": 64, "subset": None},
            "mcmc": {"name": "edm", "num_steps": 256, "sample_clip": {"min": -1.0, "max": 1.0, "scope": "x_0"}},
            "model": {"feature_dims": [96], "depths": [1, 1, 3, 1]}, "train": {"lr_init": 2e-4, "self_cond": True, "ema_coef": [0.9, 0.999]}}
    losses = np.array([0.25, 0.75, 0.5])
    torch.save({"model": sd, "config": conf, "epoch": 10, "train_loss": losses.mean(), "test_loss": np.float64(0.2),
                "model_ema_beta_0.9990": ema}, path)
    assert isinstance(losses.mean(), np.float64)
    if numpy1_names:
        src = zipfile.ZipFile(path)
        blobs = {n: src.read(n) for n in src.namelist()}
        src.close()
        with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as out:
            for n, b in blobs.items():
                if n.endswith("data.pkl"):
                    assert b.count(b"numpy._core.multiarray") >= 1
                    b = b.replace(b"numpy._core.multiarray", b"numpy.core.multiarray")
                out.writestr(n, b)
    return sd


@pytest.mark.parametrize("numpy1_names", [False, True])
def test_checkpoint_roundtrip(tmp_path, numpy1_names):
    """the reference's checkpoint layout loads through the no-code loader: NumPy-scalar losses, nested config, DDP
    'module.' keys, EMA copies; a plain weights_only load refuses the same file (that is what round 1 did)"""
    from diffusesg_amd import io as dio
    from diffusesg_amd.model import build_network
    cfg = spec.tiny_config()
    path = str(tmp_path / "tiny_00010.pth")
    sd = reference_style_checkpoint(cfg, path, numpy1_names)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu", weights_only=True)
    ck = dio.load_checkpoint(path)
    assert float(ck["train_loss"]) == 0.5 and float(ck["test_loss"]) == 0.2 and ck["epoch"] == 10
    assert ck["config"]["mcmc"]["sample_clip"]["scope"] == "x_0" and ck["config"]["dataset"]["subset"] is None
    assert dio.ema_weight_keywords(ck) == ["model"]
    assert dio.ema_weight_keywords(ck, "all") == ["model", "model_ema_beta_0.9990"]
    assert dio.ema_weight_keywords(ck, [1.0, 0.999]) == ["model", "model_ema_beta_0.9990"]
    net = build_network(cfg, device="cpu")
    dio.load_model(ck, net, "model")
    assert torch.equal(net.state_dict()["model.norm.weight"], sd["model.norm.weight"])
    dio.load_model(ck, net, "model_ema_beta_0.9990")
    assert torch.allclose(net.state_dict()["model.norm.weight"], sd["model.norm.weight"] * 0.5)
    bad = dict(sd)
    bad.pop("model.norm.bias")
    with pytest.raises(RuntimeError):
        dio.load_model({"model": bad}, net, "model")        # strict=True like the reference


def test_checkpoint_loader_still_refuses_code(tmp_path):
    """the allow-list covers NumPy scalar reconstruction only: a pickle that names anything else is refused"""
    from diffusesg_amd import io as dio
    import argparse
    path = str(tmp_path / "evil.pth")
    torch.save({"model": {}, "oops": argparse.Namespace(a=1)}, path)
    with pytest.raises(Exception):
        dio.load_checkpoint(path)


def test_npz_writer_read_like_the_evaluator(tmp_path):
    """The archive from the consumer's side: R/helper/eval_sg_samples.py:248-253 opens it with a plain `np.load(path)` (no
    pickle), reads samples_x, samples_a, gt_x, gt_a, samples_x_bbox, gt_x_bbox, gt_node_flags and turns each into a tensor with
    `torch.tensor(array)`; shapes / dtypes are what the reference's writer produces from `_decode_*` (sampler_node_adj.py:395-407:
    float32 quantised graphs [B,N,N] / [B,N], float32 bbox [B,N,4], bool flags)."""
    from diffusesg_amd import io as dio
    B, n = 3, 8
    g = torch.Generator().manual_seed(0)
    fl = torch.rand(B, n, generator=g) > 0.3
    qa = torch.randint(0, 51, (B, n, n), generator=g, dtype=torch.int32)
    qn = torch.randint(0, 150, (B, n), generator=g, dtype=torch.int32)
    bb = torch.rand(B, n, 4, generator=g)
    gt_fl = torch.rand(B, n, generator=g) > 0.3
    gt_a = torch.randint(0, 51, (B, n, n), generator=g, dtype=torch.int32)
    gt_x = torch.randint(0, 150, (B, n), generator=g, dtype=torch.int32)
    gt_bb = torch.rand(B, n, 4, generator=g)
    p = str(tmp_path / "final_samples_array_before_eval.npz")
    dio.save_samples_npz(p, samples_node_flags=fl, samples_a=qa, samples_x=qn, raw_a=torch.zeros(B, 6, n, n), raw_x=torch.zeros(B, n, 8),
                         samples_x_bbox=bb, gt_node_flags=gt_fl, gt_a=gt_a, gt_x=gt_x, gt_x_bbox=gt_bb, gt_image_ids=torch.arange(B))
    # --- the consumer's statements ---
    data = np.load(p)
    samples_x, samples_a, gt_x_, gt_a_ = data['samples_x'], data['samples_a'], data['gt_x'], data['gt_a']
    samples_x_bbox, gt_x_bbox, node_flags = data['samples_x_bbox'], data['gt_x_bbox'], data['gt_node_flags']
    tens = [torch.tensor(a) for a in (samples_x, samples_a, gt_x_, gt_a_, samples_x_bbox, gt_x_bbox, node_flags)]
    # --- what it must have got ---
    assert [t.dtype for t in tens] == [torch.float32] * 6 + [torch.bool]
    assert tens[0].shape == (B, n) and tens[1].shape == (B, n, n) and tens[4].shape == (B, n, 4) and tens[6].shape == (B, n)
    assert torch.equal(tens[0], qn.float()) and torch.equal(tens[1], qa.float()) and torch.equal(tens[2], gt_x.float())
    assert torch.equal(tens[3], gt_a.float()) and torch.equal(tens[4], bb) and torch.equal(tens[5], gt_bb) and torch.equal(tens[6], gt_fl)
    assert set(data.files) == {"samples_node_flags", "samples_a", "samples_x", "raw_a", "raw_x", "gt_node_flags", "gt_a", "gt_x",
                               "samples_x_bbox", "gt_x_bbox", "gt_image_ids"}
    assert data["samples_node_flags"].dtype == bool and data["raw_a"].shape == (B, 6, n, n) and data["gt_image_ids"].dtype == np.int64
    # every entry loads without pickle (the reference's None entries would not); a run without bbox channels writes typed empties
    p2 = str(tmp_path / "nobbox.npz")
    dio.save_samples_npz(p2, samples_node_flags=fl, samples_a=qa, samples_x=qn, raw_a=torch.zeros(B, 6, n, n), raw_x=torch.zeros(B, n, 8),
                         gt_node_flags=gt_fl, gt_a=gt_a, gt_x=gt_x)
    d2 = np.load(p2)
    for k in d2.files:
        assert d2[k].dtype != object, k
    assert d2["samples_x_bbox"].shape == (B, n, 0) and d2["gt_x_bbox"].shape == (B, n, 0)
    # the ground truth is not optional, and bbox arrays come in pairs
    with pytest.raises(ValueError):
        dio.save_samples_npz(p2, samples_node_flags=fl, samples_a=qa, samples_x=qn, raw_a=torch.zeros(B, 6, n, n), raw_x=torch.zeros(B, n, 8),
                             gt_node_flags=gt_fl, gt_a=None, gt_x=gt_x)
    with pytest.raises(ValueError):
        dio.save_samples_npz(p2, samples_node_flags=fl, samples_a=qa, samples_x=qn, raw_a=torch.zeros(B, 6, n, n), raw_x=torch.zeros(B, n, 8),
                             samples_x_bbox=bb, gt_node_flags=gt_fl, gt_a=gt_a, gt_x=gt_x)


def test_pack_decoded_roundtrip():
    from diffusesg_amd import io as dio
    B, n = 4, 8
    qa = torch.randint(0, 51, (B, n, n), dtype=torch.int32)
    qn = torch.randint(0, 150, (B, n), dtype=torch.int32)
    fl = torch.rand(B, n) > 0.3
    bb = torch.rand(B, n, 4)
    p = dio.pack_decoded(qa, qn, bb, fl)
    assert p.dtype == torch.int16 and p.shape == (B, n * n + 2 * n + 8 * n)     # int16 ids + fp32 bbox bits, as documented
    vg = dio.pack_decoded(torch.zeros(1, 64, 64, dtype=torch.int32), torch.zeros(1, 64, dtype=torch.int32), torch.zeros(1, 64, 4),
                          torch.ones(1, 64, dtype=torch.bool))
    assert vg.numel() * vg.element_size() == 2 * (4096 + 64 + 64) + 1024          # 9.3 KB per VG graph vs 101 KB raw
    a2, n2, f2, b2 = dio.unpack_decoded(p, n, True)
    assert torch.equal(a2, qa) and torch.equal(n2, qn) and torch.equal(f2, fl) and torch.equal(b2, bb)


def _run_bench(args, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, [json.loads(ln) for ln in lines], p.stderr


def test_bench_self_launches_two_ranks_gloo():
    """`python bench.py --gpus 2` with no torchrun environment: the parent spawns torch.distributed.run (2 ranks, 127.0.0.1),
    the ranks run the bench's own barrier / max-over-ranks timing / packed all-gather sequence over gloo with a stand-in
    step, rank 0's single JSON line is relayed and the exit status is the children's"""
    rc, lines, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "tiny", "--batch", "3", "--selftest-launcher"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    ln = lines[0]
    assert ln["selftest"] and ln["n_gpus"] == 2 and ln["world_size"] == 2 and ln["backend"] == "gloo"
    assert ln["gathered_rows"] == 6 and ln["steps"] == 2 and ln["warmup"] == 1
    # an N-GPU line describes itself: the per-rank spread of network forwards (coin streams differ by rank; `value` is over the
    # max-over-ranks time) and the gathered payload; the stand-in step reports 100 + rank forwards per step
    pr = ln["per_rank"]
    assert pr["net_forwards_per_step_min"] == 100 and pr["net_forwards_per_step_max"] == 101 and pr["net_forwards_per_step_rank0"] == 100
    tiny = 6 * 8 * 8 + 8 * 12   # floats of one tiny-config graph (C_adj N^2 + N C_node)
    assert pr["gather_payload_bytes_per_rank"] == 3 * tiny * 4 and pr["gather_payload_bytes_total"] == 6 * tiny * 4
    assert "cpu_baseline" in ln


def test_bench_launcher_propagates_failure():
    """a rank that dies makes the launcher exit non-zero (here: WORLD_SIZE check fails because torchrun gives 2 ranks to --gpus 3...)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--selftest-launcher"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_bench_single_rank_selftest_needs_no_process_group():
    rc, lines, err = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--config", "tiny", "--batch", "2", "--selftest-launcher"])
    assert rc == 0, err[-2000:]
    assert lines[0]["world_size"] == 1 and lines[0]["gathered_rows"] == 2


def test_checkpoint_writer_layout_matches_reference(tmp_path):
    """io.get_ckpt_data / save_checkpoint write what the reference's trainer writes (trainer_utils.py:168-185): 'model' with the
    precond wrapper's 'model.' prefix, nested config, epoch, NumPy-scalar losses, 'model_ema_beta_{beta:.4f}' per EMA helper -- and
    load_checkpoint / load_model / ema_weight_keywords read it back (CPU only: no library call involved)"""
    from diffusesg_amd import io as dio, synth as Y, weights as W
    from diffusesg_amd.model import build_network
    cfg = Y.CONFIGS["tiny"]()
    model = build_network(cfg, W.synth_state_dict(cfg, 0), device="cuda")   # parameters stay on the CPU until a forward needs the GPU

    class FakeEMA:   # the two attributes of diffusesg_amd.train.EMAHip the writer uses
        beta = 0.999
        shadow = {k: p.detach().clone() * 0.5 for k, p in model.model.named_parameters()}
    path = dio.save_checkpoint(str(tmp_path / "visual_genome_00007.pth"), model, [FakeEMA()], 7, np.float64(1.25), np.float64(1.5),
                               {"model": {"name": "diffuse_sg"}, "train": {"ema_coef": [0.999]}})
    ckp = dio.load_checkpoint(path)
    assert set(ckp) == {"model", "config", "epoch", "train_loss", "test_loss", "model_ema_beta_0.9990"}
    assert all(k.startswith("model.") for k in ckp["model"]) and ckp["epoch"] == 7 and float(ckp["train_loss"]) == 1.25
    assert dio.ema_weight_keywords(ckp, [0.999]) == ["model_ema_beta_0.9990"]
    k0 = "model.patch_embed.proj.weight"
    assert torch.equal(ckp["model_ema_beta_0.9990"][k0], ckp["model"][k0] * 0.5)
    assert torch.equal(ckp["model_ema_beta_0.9990"]["model.down_layers.0.blocks.0.attn.relative_position_index"],
                       ckp["model"]["model.down_layers.0.blocks.0.attn.relative_position_index"])   # buffers come from the online model
    fresh = build_network(cfg, W.synth_state_dict(cfg, 3), device="cuda")
    dio.load_model(ckp, fresh, "model_ema_beta_0.9990")
    assert torch.equal(dict(fresh.model.named_parameters())["patch_embed.proj.weight"], ckp["model"][k0] * 0.5)


def test_gelu_coefficients_in_the_kernel_are_the_fit_scripts():
    """csrc/kernels_common.hip.h::gelu_f evaluates max(x,0) - |x| 2^h(min(|x|,6)) with a degree-6 h fitted by tools/fit_gelu.py:
    the constants in the header are that script's output, and the fp32 evaluation stays within 3e-7 of an fp64 exact-erf GELU"""
    import importlib.util
    import re
    root = os.path.join(os.path.dirname(__file__), "..")
    spec_ = importlib.util.spec_from_file_location("fit_gelu", os.path.join(root, "tools", "fit_gelu.py"))
    fg = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(fg)
    c = fg.fit()
    src = open(os.path.join(root, "diffusesg_amd", "csrc", "kernels_common.hip.h")).read()
    body = src[src.index("__device__ __forceinline__ float gelu_f(float x) {"):]
    body = body[:body.index("}")]
    consts = [float(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+|-?\d\.\d{6,})f", body)]
    assert len(consts) == 7, consts            # Horner order: c6, c5, c4, c3, c2, c1, c0
    np.testing.assert_allclose(consts, c[::-1], rtol=2e-9)
    from scipy.special import erfc
    x = np.linspace(-12, 12, 400001).astype(np.float32)
    ref = x.astype(np.float64) * 0.5 * erfc(-x.astype(np.float64) / np.sqrt(2))
    assert np.abs(fg.gelu_f32(x, np.array(consts[::-1])) - ref).max() < 3e-7


def test_ema_decay_schedule_restated_from_ema_pytorch():
    """EMAHip.get_current_decay with the reference's arguments (update_every=1, update_after_step=0, inv_gamma=1, power=1):
    0 on the first two updates' epochs <= 0, then 1 - 1/(1 + epoch) capped at beta (ema_pytorch is absent: parity unpinned)"""
    from diffusesg_amd.train import EMAHip

    class Net:   # the attributes EMAHip touches without a GPU
        _dev = torch.device("cpu")
        def named_parameters(self):
            return iter([("w", torch.nn.Parameter(torch.ones(3)))])
    e = EMAHip(Net(), beta=0.9)
    seen = []
    for step in range(1, 14):
        e.step = step
        seen.append(e.get_current_decay())
    assert seen[0] == 0.0 and seen[1] == 0.5 and abs(seen[2] - 2.0 / 3.0) < 1e-12
    assert seen[-1] == 0.9 and all(b >= a for a, b in zip(seen, seen[1:]))


def test_exponential_lr_matches_torch():
    """ExponentialLRHip against torch.optim.lr_scheduler.ExponentialLR (what get_optimizer pairs with Adam, learning_utils.py:142) over 12
    epochs, and through a state_dict round trip"""
    from diffusesg_amd.train import ExponentialLRHip

    class _Opt:   # AdamHip's only attribute the scheduler touches
        lr = 2.0e-4
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=2.0e-4)
    tsch = torch.optim.lr_scheduler.ExponentialLR(topt, gamma=0.97)
    o = _Opt()
    sch = ExponentialLRHip(o, gamma=0.97)
    for ep in range(12):
        topt.step()
        tsch.step(); sch.step()
        assert abs(o.lr - topt.param_groups[0]["lr"]) <= 1e-12 * 2.0e-4 + 1e-18, ep
        assert abs(sch.get_last_lr()[0] - tsch.get_last_lr()[0]) <= 1e-15
        if ep == 5:
            o2 = _Opt()
            sch2 = ExponentialLRHip(o2, gamma=0.5)
            sch2.load_state_dict(sch.state_dict())
            assert o2.lr == o.lr and sch2.last_epoch == 6
