"""CPU: the ORACLE (oracle/dsg_ref.c) against the golden vectors produced by the reference's own
Python modules (tools/gen_golden.py).  This is what pins the oracle (prompt ③)."""
import numpy as np
import pytest

from diffusesg_amd import synth as Y
from diffusesg_amd import weights as W
from oracle.oracle import Oracle
from util import FWD_RTOL, assert_close, load


def make_oracle(cfg, seed=0):
    return Oracle(cfg, W.synth_state_dict(cfg, seed))


@pytest.mark.parametrize("name", ["tiny", "small", "nosc", "vg", "coco", "onehot"])
def test_forward_matches_reference(name):
    cfg, flags, adj, node, sc_adj, sc_node = Y.fwd_case(name)
    g = load(f"fwd_{name}.npz")
    orc = make_oracle(cfg)
    oa, on = orc.forward(adj, node, flags, Y.FWD_C_NOISE)
    assert_close(oa, g["nosc_adj_out"], FWD_RTOL, f"{name} adj (self-cond None)")
    assert_close(on, g["nosc_node_out"], FWD_RTOL, f"{name} node (self-cond None)")
    if cfg.self_condition:
        oa, on = orc.forward(adj, node, flags, Y.FWD_C_NOISE, sc_adj, sc_node)
        assert_close(oa, g["sc_adj_out"], FWD_RTOL, f"{name} adj (self-cond)")
        assert_close(on, g["sc_node_out"], FWD_RTOL, f"{name} node (self-cond)")
    # padded rows / columns must be exactly zero
    f = flags.astype(bool)
    assert np.all(on[~f] == 0)
    assert np.all(oa.transpose(0, 2, 3, 1)[~f] == 0) and np.all(oa.transpose(0, 3, 2, 1)[~f] == 0)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_forward_intermediates(name):
    cfg, flags, adj, node, sc_adj, sc_node = Y.fwd_case(name)
    g = load(f"fwd_{name}.npz")
    keys = [k[len("inter/"):] for k in g.files if k.startswith("inter/")]
    taps = {k: int(np.prod(g["inter/" + k].shape[1:])) for k in keys}
    orc = make_oracle(cfg)
    _, _, bufs = orc.forward(adj, node, flags, Y.FWD_C_NOISE, sc_adj, sc_node, taps=taps)
    for k in keys:
        ref = g["inter/" + k]
        got = bufs[k].reshape(ref.shape[0], -1)
        if k == "read_out":  # reference tensor is [B,C,H,W]; the oracle keeps token-major [T,C]
            ref = ref.transpose(0, 2, 3, 1)
        assert_close(got, ref.reshape(ref.shape[0], -1), FWD_RTOL, f"{name} {k}")


@pytest.mark.parametrize("name", ["vg", "coco"])
def test_forward_module_rows_full_size(name):
    """per-module fixtures of the full-size nets: a fixed sample of token rows of every block of every
    (T, C, shift) class, every PatchMerging / PatchBreakup, PatchEmbed and the read-out, hooked from the reference's modules
    (fwd_{vg,coco}.npz 'rows/<tap>', tools/gen_golden.py::gen_forward)"""
    cfg, flags, adj, node, sc_adj, sc_node = Y.fwd_case(name)
    g = load(f"fwd_{name}.npz")
    keys = [k[len("rows/"):] for k in g.files if k.startswith("rows/")]
    shapes = Y.tap_shapes(cfg)
    assert set(keys) == set(shapes), set(keys) ^ set(shapes)
    orc = make_oracle(cfg)
    _, _, bufs = orc.forward(adj, node, flags, Y.FWD_C_NOISE, sc_adj, sc_node, taps={k: shapes[k][0] * shapes[k][1] for k in keys})
    for k in keys:
        Tk, Ck = shapes[k]
        got = bufs[k].reshape(2 * Tk, Ck)[Y.inter_rows(2 * Tk)]
        assert_close(got, g["rows/" + k], FWD_RTOL, f"{name} {k}")


@pytest.mark.parametrize("tag", ["vg_heun6", "vg_euler6", "coco_heun6"])
def test_full_size_short_trajectories_match_reference(tag):
    """6-step sampler runs of the full-size VG / COCO-Stuff networks through the reference's own sampler (traj_big.npz,
    tools/gen_golden.py::gen_big_trajectories): the oracle on the same replayed noise and coins"""
    cfg, T_, solver, churn, flags, ia, inn, na, nn, coins = Y.big_traj_case(tag)
    g = load("traj_big.npz")
    orc = make_oracle(cfg)
    ra, rn = orc.sample(flags, ia, inn, na if churn > 0 else None, nn if churn > 0 else None, coins, num_steps=T_, solver=solver,
                        S_churn=churn)
    assert_close(ra, g[tag + "_adj"], 1e-4, f"{tag} adj")
    assert_close(rn, g[tag + "_node"], 1e-4, f"{tag} node")


@pytest.mark.parametrize("name", ["tiny", "small", "nosc"])
def test_precond_matches_reference(name):
    g = load(f"precond_{name}.npz")
    orc = None
    for si in range(3):
        cfg, sigma, flags, adj, node, sc_adj, sc_node = Y.precond_case(name, si)
        orc = orc or make_oracle(cfg)
        sig = np.full((2,), sigma, np.float32)
        for coin in (0, 1):
            for with_sc in (0, 1):
                da, dn = orc.precond(adj, node, flags, sig, sc_adj if with_sc else None,
                                     sc_node if with_sc else None, coin=bool(coin))
                key = f"s{si}_coin{coin}_sc{with_sc}"
                assert_close(da, g[key + "_adj"], FWD_RTOL, f"{name} {key} adj")
                assert_close(dn, g[key + "_node"], FWD_RTOL, f"{name} {key} node")


@pytest.mark.parametrize("T", [50, 100, 256, 1000])
def test_sigma_schedule(T):
    g = load("sampler.npz")
    np.testing.assert_allclose(Oracle.sigma_steps(T), g[f"sigma_steps_{T}"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("tag,T,solver,churn", Y.SAMPLER_RUNS)
def test_sampler_trajectory(tag, T, solver, churn):
    g = load("sampler.npz")
    cfg = Y.CONFIGS["tiny"]()
    flags, ia, inn, na, nn, coin_vals = Y.sampler_case(cfg, T, 4, Y.SAMPLER_VALID, 3, f"smp/{tag}", solver)
    orc = make_oracle(cfg)
    coins = (coin_vals < 0.5).astype(np.uint8)
    oa, on = orc.sample(flags, ia, inn, na, nn, coins, num_steps=T, solver=solver, S_churn=churn)
    # Trajectory tolerance: the per-forward bar compounded over a few steps (T=8) / stated looser bar (T=50)
    tol = 1e-4 if T <= 8 else 1e-3
    assert_close(oa, g[f"{tag}_adj"], tol, f"{tag} adj")
    assert_close(on, g[f"{tag}_node"], tol, f"{tag} node")
    expected_nfe = int(g[f"{tag}_coins_used"]) + int(coins[: int(g[f"{tag}_coins_used"])].sum())
    assert orc.nfe == expected_nfe


def test_sampler_no_self_cond():
    g = load("sampler.npz")
    cfg = Y.CONFIGS["nosc"]()
    flags, ia, inn, na, nn, _ = Y.sampler_case(cfg, 8, 2, [8, 3], 3, "smp/nosc_t8")
    orc = make_oracle(cfg)
    oa, on = orc.sample(flags, ia, inn, na, nn, None, num_steps=8)
    assert_close(oa, g["nosc_t8_adj"].reshape(oa.shape), 1e-4, "nosc adj")
    assert_close(on, g["nosc_t8_node"].reshape(on.shape), 1e-4, "nosc node")
    assert orc.nfe == 15  # no coin without self-conditioning: exactly 2T-1 forwards


def test_sampler_known_answer():
    """sanity-check mode (edm.py:372-377): with the denoiser replaced by GT the loop must return GT."""
    g = load("sampler.npz")
    cfg = Y.CONFIGS["tiny"]()
    flags, ia, inn, na, nn, _ = Y.sampler_case(cfg, 8, 4, Y.SAMPLER_VALID, 3, "smp/gt")
    gt_adj, gt_node = Y.gt_case(cfg, 4, Y.SAMPLER_VALID)
    orc = make_oracle(cfg)
    oa, on = orc.sample(flags, ia, inn, na, nn, None, gt_adj=gt_adj, gt_node=gt_node, num_steps=8)
    assert np.abs(oa - gt_adj).max() < 1e-6 and np.abs(on - gt_node).max() < 1e-6
    assert_close(oa, g["gt_adj"], 1e-6, "gt adj vs reference run")
    assert orc.nfe == 0


@pytest.mark.parametrize("name", ["vg", "coco"])
def test_decode_bits_matches_reference(name):
    """post-decode of 'bits' samples (sampler_node_adj.py:222-285 around the reference's own bin2dec): BIT-EXACT, incl.
    values outside [-1,1], exact zeros and codes above n_type-1 (clamped)"""
    g = load("decode.npz")
    cfg, flags, adj, node = Y.decode_case(name)
    _, n_adj, n_node, _ = Y.DECODE_CASES[name]
    qa, qn, bb = make_oracle(cfg).decode_bits(adj, node, flags, n_adj, n_node)
    assert np.array_equal(qa, g[f"{name}_q_adj"].astype(np.int32)) and np.array_equal(qn, g[f"{name}_q_node"].astype(np.int32))
    assert np.array_equal(bb, g[f"{name}_bbox"])
    assert (qa == n_adj - 1).sum() > 0 and (qn == n_node - 1).sum() > 0   # the clamp is exercised


@pytest.mark.parametrize("name", sorted(Y.DECODE_ENC_CASES))
def test_decode_one_hot_and_ddpm_match_reference(name):
    """post-decode of 'one_hot' / 'ddpm' samples (and mixed with 'bits'): the oracle against tests/golden/decode_enc.npz, which the
    reference's own attribute_converter produced inside the restated _decode_node / _decode_adj closures (sampler_node_adj.py:
    222-285) -- BIT-EXACT, with several / no positive one_hot channels, exact zeros, and ddpm values on and one ulp either side of
    every interval edge"""
    g = load("decode_enc.npz")
    cfg, flags, adj, node, e_adj, e_node, n_adj, n_node = Y.decode_enc_case(name)
    qa, qn, bb = make_oracle(cfg).decode(adj, node, flags, e_adj, e_node, n_adj, n_node)
    assert np.array_equal(qa, g[f"{name}_q_adj"].astype(np.int32)), f"{name} q_adj: {(qa != g[f'{name}_q_adj']).sum()} differ"
    assert np.array_equal(qn, g[f"{name}_q_node"].astype(np.int32)), f"{name} q_node: {(qn != g[f'{name}_q_node']).sum()} differ"
    assert np.array_equal(bb, g[f"{name}_bbox"])
    assert len(np.unique(qa)) == n_adj   # every edge class occurs


def test_train_forward_matches_reference():
    """forward half of a training / test-loss step (trainer_node_adj.py:96-163 in 'test' mode) against tests/golden/
    train_forward.npz: the objective generator's sigmas / weights / noisy inputs from replayed draws, the preconditioned
    model with per-sample sigmas, the masked sigma-weighted MSE per sample and the bounding-box IoU term"""
    g = load("train_forward.npz")
    cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin = Y.train_case("tiny")
    orc = make_oracle(cfg)
    sig, wts, na, nn = orc.train_inputs(clean_adj, clean_node, flags, rnd, eps_adj, eps_node)
    np.testing.assert_allclose(sig, g["tiny_sigmas"], rtol=1e-6)
    np.testing.assert_allclose(wts, g["tiny_weights"], rtol=2e-6)
    assert_close(na, g["tiny_noisy_adj"], 1e-6, "noisy adj")
    assert_close(nn, g["tiny_noisy_node"], 1e-6, "noisy node")
    pa, pn = orc.precond(na, nn, flags, sig, None, None, coin=coin < 0.5)
    assert_close(pa, g["tiny_pred_adj"], FWD_RTOL, "model output adj")
    assert_close(pn, g["tiny_pred_node"], FWD_RTOL, "model output node")
    la0, ln0 = orc.rainbow_loss(g["tiny_pred_adj"], g["tiny_pred_node"], clean_adj, clean_node, flags, wts)
    np.testing.assert_allclose(la0, g["tiny_loss_adj_noiou"], rtol=1e-5)
    np.testing.assert_allclose(ln0, g["tiny_loss_node_noiou"], rtol=1e-5)
    la, ln = orc.rainbow_loss(g["tiny_pred_adj"], g["tiny_pred_node"], clean_adj, clean_node, flags, wts, iou_w=1.0)
    np.testing.assert_allclose(la, g["tiny_loss_adj"], rtol=1e-5)
    np.testing.assert_allclose(ln, g["tiny_loss_node"], rtol=1e-5, atol=1e-5)
    # end to end through the oracle's own model output: loss = adj.mean() + node.mean() (trainer_node_adj.py:167)
    la, ln = orc.rainbow_loss(pa, pn, clean_adj, clean_node, flags, wts, iou_w=1.0)
    assert abs(float(la.mean() + ln.mean()) - float(g["tiny_loss"])) <= 1e-3 * abs(float(g["tiny_loss"]))


def test_train_backward_head_matches_reference_autograd():
    """first stage of the backward of a training step against tests/golden/train_backward.npz (the reference's own autograd,
    trainer_node_adj.py:96-170 in 'train' mode): dL/d(preconditioned outputs) -- sigma-weighted masked MSE terms plus the IoU term
    through clamp / max / min -- and dL/d(raw network outputs) = c_out(sigma) * that.  The fixture also holds the parameter
    gradients of the whole network (norm + strided sample per tensor): the target of the network backward, which is not built."""
    g = load("train_backward.npz")
    cfg, flags, clean_adj, clean_node, rnd, eps_adj, eps_node, coin = Y.train_case("tiny")
    orc = make_oracle(cfg)
    ga, gn, fa, fn = orc.rainbow_loss_backward(g["tiny_pred_adj"], g["tiny_pred_node"], clean_adj, clean_node, flags, g["tiny_weights"],
                                               iou_w=1.0, sigmas=g["tiny_sigmas"])
    assert_close(ga, g["tiny_grad_pred_adj"].reshape(ga.shape), 1e-5, "dL/dD adj")
    assert_close(gn, g["tiny_grad_pred_node"].reshape(gn.shape), 1e-5, "dL/dD node")
    assert_close(fa, g["tiny_grad_F_adj"].reshape(fa.shape), 1e-5, "dL/dF adj")
    assert_close(fn, g["tiny_grad_F_node"].reshape(fn.shape), 1e-5, "dL/dF node")
    assert np.abs(g["tiny_grad_pred_node"][..., -4:]).max() > 0    # the IoU term contributes
    # the fixture is internally consistent: the recorded total norm is the norm of the recorded per-tensor norms
    assert abs(float(np.sqrt((g["tiny_gparam_norms"] ** 2).sum())) - float(g["tiny_total_grad_norm"])) < 1e-3 * float(g["tiny_total_grad_norm"])
    assert len(g["tiny_gparam_names"]) == len([k for k in g.files if k.startswith("tiny_gparam/")])


@pytest.mark.parametrize("name", ["tiny", "vg", "coco"])
def test_noise_embedding_standalone_matches_reference(name):
    """survey fixture G1: PositionalEmbedding and the map MLP on their own (diffusesg.py:507-513, :768-771) against
    tests/golden/noise_embed.npz, over c_noise = ln(sigma)/4 for sigma from 1e-4 to 80"""
    g = load("noise_embed.npz")
    cfg = Y.CONFIGS[name]()
    pe, emb = make_oracle(cfg).noise_embed(g["c_noise"])
    assert_close(pe, g[f"{name}_pe"], 2e-6, "positional embedding")
    assert_close(emb, g[f"{name}_emb"], 1e-5, "mapped noise embedding")


@pytest.mark.parametrize("iou_type", Y.IOU_TYPES)
def test_bbox_loss_types_match_reference_autograd(iou_type):
    """Every iou_loss_type of the trainer's bounding-box term (R/runner/trainer/trainer_node_adj.py:130-159; the reference's README
    trains with 'giou'): per-sample losses and d(loss_adj.mean() + loss_node.mean())/d(model outputs) against
    tests/golden/iou_losses.npz (tools/gen_golden.py::gen_iou_losses: the imported NodeAdjRainbowLoss + the trainer's block under the
    reference's autograd).  torchvision is absent: its box_iou / generalized_ / distance_ / complete_box_iou_loss are restated in the
    generator from the published source -- parity for those four ops is unpinned against torchvision itself."""
    cfg, flags, pred_adj, pred_node, tgt_adj, tgt_node, wts, sigmas = Y.iou_case()
    g = load("iou_losses.npz")
    ew, nw, iw = float(g["edge_w"]), float(g["node_w"]), float(g["iou_w"])
    orc = make_oracle(cfg)
    la, ln = orc.rainbow_loss(pred_adj, pred_node, tgt_adj, tgt_node, flags, wts, ew, nw, iw, iou_type=iou_type)
    assert_close(la, g[f"{iou_type}_loss_adj"], 2e-6, f"{iou_type} loss_adj")
    assert_close(ln, g[f"{iou_type}_loss_node"], 2e-6, f"{iou_type} loss_node")
    ga, gn, fa, fn = orc.rainbow_loss_backward(pred_adj, pred_node, tgt_adj, tgt_node, flags, wts, ew, nw, iw, sigmas=sigmas, iou_type=iou_type)
    assert_close(ga, g[f"{iou_type}_grad_adj"], 1e-5, f"{iou_type} grad_adj")
    assert_close(gn, g[f"{iou_type}_grad_node"], 1e-5, f"{iou_type} grad_node")
    # the bbox channels on their own (the MSE part would otherwise dominate the scale)
    ref_b = g[f"{iou_type}_grad_node"][..., -4:]
    mse_b = (gn - g["iou_grad_node"] * 0)[..., -4:]
    assert_close(mse_b, ref_b, 1e-5, f"{iou_type} grad bbox channels")
    c_out = sigmas * 0.5 / np.sqrt(sigmas ** 2 + 0.25)
    assert_close(fn, gn * c_out[:, None, None], 1e-6, "dL/dF = c_out dL/dD")
    # the five types really differ
    if iou_type != "iou":
        assert np.abs(g[f"{iou_type}_grad_node"] - g["iou_grad_node"]).max() > 1e-3
