#!/usr/bin/env python3
"""bench.py -- scene-graphs/sec of the MI355X sampling path on BASELINE.json's headline configuration.

One "step" = one full pass of the hot path over one batch: NodeAdjEDMSamplerHip.sample() of B graphs with the
reference sampler settings (Visual-Genome shape: 64 padded / 30 valid nodes, 6 adjacency + 12 node channels,
stochastic Heun, S_churn=40, num_steps T=1000, fp32), from on-device initial noise to the raw generated
(adj, node) resident in HBM, followed (N>1) by the single all-gather of the packed results.  Nothing is skipped:
every preconditioned call, every coin-triggered extra self-conditioning forward, churn noise and masking run.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: this process becomes a LAUNCHER -- before any GPU call it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
as a child process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the child's status.  Launched
by torchrun (RANK/LOCAL_RANK/WORLD_SIZE set) it is a rank.  The W warm-up samples run on a short schedule
(`--warmup-num-steps`, default 20: graph capture, workspace allocation and clocks settle within the first forwards);
the K timed steps are full T-step samples.  Prints ONE JSON line on rank 0 with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2516.6  # same guide: ~2.5 PF dense = 16 x the fp32 matrix rate


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="graphs per GPU per step (configs[1]: 64; configs[3]: 256)")
    ap.add_argument("--num-steps", type=int, default=1000, help="sampler steps T (BASELINE metric: 1000)")
    ap.add_argument("--warmup-num-steps", type=int, default=20, help="sampler steps of the untimed warm-up samples")
    ap.add_argument("--config", default="vg", choices=["vg", "coco", "tiny"])
    ap.add_argument("--valid", type=int, default=None, help="valid nodes per graph (VG: 30)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32-split", "bf16"],
                    help="f32 (default, the BASELINE headline): fp32 MFMA; f32-split: opt-in fp32-accurate GEMMs as six bf16-MFMA "
                         "partial products of hi/mid/lo operand splits; bf16: opt-in bf16-MFMA GEMMs with fp32 accumulate")
    ap.add_argument("--solver", default="heun", choices=["heun", "euler"], help="configs[2] variant 3b: --solver euler --s-churn 0")
    ap.add_argument("--s-churn", type=float, default=40.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=40.0, help="wall-clock bound of the cpu_baseline leg")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="with --gpus 1: create the RCCL process group anyway (world size 1) so that the barrier, the float64 "
                         "max-over-ranks all-reduce and both all-gathers (fp32 raw, int16-as-bytes decoded) run through RCCL on one GPU")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (launcher, rank env, gloo process group, barrier, max-over-ranks "
                         "timing, packed all-gather) with a stand-in step; no GPU, no product kernels, the value is meaningless")
    return ap.parse_args(argv)


def launch_ranks(n_ranks: int, argv) -> int:
    """Parent of a `--gpus N` (N > 1) run started without torchrun.  Touches no GPU API and never re-execs: the ranks are
    child processes of `python -m torch.distributed.run`; their stdout (rank 0's JSON line) is relayed line by line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL / cross-process tensors)
    env.setdefault("OMP_NUM_THREADS", "1")              # the ranks are launch-bound host threads, not OpenMP workers
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    finally:
        rc = proc.wait()
    return rc


def cpu_baseline(cfg, sd, T, valid, budget_s):
    """The ORACLE (oracle/dsg_ref.c, kind 'port') timed on this box's host cores on a bounded sample of the same workload:
    B = cores graphs (the oracle runs one graph per OpenMP thread), the first sampler steps of the same T-step schedule with
    the same coin stream, as many as fit the budget; extrapolated by network forwards (per-forward cost is constant)."""
    import numpy as np
    from diffusesg_amd import synth, weights
    from oracle.oracle import Oracle
    orc = Oracle(cfg, sd)
    cores = os.cpu_count() if not hasattr(os, "sched_getaffinity") else len(os.sched_getaffinity(0))
    B = max(2, min(64, cores))
    # the ranks of a multi-GPU run are started with OMP_NUM_THREADS=1 (launch_ranks); this leg runs on rank 0 while the others wait at
    # the final barrier, and wants the host's cores
    threads = Oracle.set_threads(min(cores, B))
    flags, ia, inn, _, _, cv = synth.sampler_case(cfg, 4, B, valid, 77, "bench/cpu")
    coins_all = (weights.coins(77, "bench/cpu/all", 2 * T - 1) < 0.5).astype(np.uint8)
    nfe_total = (2 * T - 1) + int(coins_all.sum())
    t_begin = time.perf_counter()
    # calibrate on one step, then one run sized to the remaining budget
    n0 = orc.nfe
    t0 = time.perf_counter()
    orc.sample(flags, ia, inn, None, None, coins_all, num_steps=T, max_steps=1)
    elapsed, nfe, steps = time.perf_counter() - t0, orc.nfe - n0, 1
    remaining = budget_s - (time.perf_counter() - t_begin)
    more = int(remaining / (elapsed / nfe) / 3.0)   # ~3 forwards per Heun step (2 precond calls + coins)
    if more >= 2:
        steps = min(64, more)
        n0 = orc.nfe
        t0 = time.perf_counter()
        orc.sample(flags, ia, inn, None, None, coins_all, num_steps=T, max_steps=steps)
        elapsed, nfe = time.perf_counter() - t0, orc.nfe - n0
    per_fwd = elapsed / nfe
    return {"value": B / (per_fwd * nfe_total), "unit": "scene-graphs/s", "cores": int(min(threads, B)), "kind": "port",
            "sample": f"oracle/dsg_ref.c (OpenMP, one graph per thread), same config, B={B}, first {steps} of T={T} Heun steps = "
                      f"{nfe} network forwards of {B} graphs in {elapsed:.1f} s, scaled to {nfe_total} forwards per graph"}


def worker(args):
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    from diffusesg_amd import dist as dsg_dist
    from diffusesg_amd import spec, synth, weights

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    selftest = args.selftest_launcher
    use_pg = world > 1 or args.rehearse_collectives
    if args.rehearse_collectives:
        os.environ["DSG_FORCE_COLLECTIVE"] = "1"   # diffusesg_amd.dist.gather_results: run the all-gather even at world size 1
    if use_pg and "MASTER_ADDR" not in os.environ:   # single-rank rehearsal started without torchrun
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if selftest:
        dev = torch.device("cpu")
        if use_pg:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if use_pg:
            # RCCL prints a version banner to the process's stdout when its first communicator comes up; the contract is ONE JSON
            # line on rank 0's stdout, so file descriptor 1 points at stderr while the group initialises and runs its first collective
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" is RCCL on ROCm
                dist.barrier()
                torch.cuda.synchronize(dev)
            finally:
                sys.stdout.flush()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
    world_seen = dist.get_world_size() if use_pg else 1

    cfg = synth.CONFIGS[args.config]()
    n = cfg.max_node_num
    valid = args.valid if args.valid is not None else {"vg": 30, "coco": 20, "tiny": 8}[args.config]
    T, B = args.num_steps, args.batch
    seed = dsg_dist.rank_seed(args.seed, rank)     # arg_parser.py:293-294: seed += rank
    np.random.seed(seed)                           # the coin stream (np.random.rand, precond.py:90) is part of the workload

    if selftest:
        net = sd = smp = smp_warm = None
        flags = torch.from_numpy(weights.synth_flags(B, n, valid))

        def one_step(k, sampler):
            g = torch.Generator().manual_seed(seed + 1000 * (k + 7))
            oa, on = torch.randn(B, cfg.c_adj, n, n, generator=g), torch.randn(B, n, cfg.c_node, generator=g)
            return dsg_dist.gather_results(dsg_dist.pack_results(oa, on)), {"net_forwards": 100 + rank}   # (stand-in count, differs by rank)
    else:
        from diffusesg_amd.model import build_network
        from diffusesg_amd.sampler import NodeAdjEDMSamplerHip
        sd = weights.synth_state_dict(cfg, 0)
        net = build_network(cfg, sd, device=dev)
        h = net.model._ensure_handle()
        if args.precision == "bf16":
            h.set_option("gemm_bf16", 1)
        if args.precision == "f32-split":
            h.set_option("gemm_split", 1)
        skw = dict(solver=args.solver, S_churn=args.s_churn, S_min=0.05, S_max=50, S_noise=1.003, clip_samples=True,
                   clip_samples_min=-1.0, clip_samples_max=1.0, clip_samples_scope="x_0", self_condition=cfg.self_condition,
                   dev=dev, use_graph=not args.no_graph)
        smp = NodeAdjEDMSamplerHip(num_steps=T, **skw)
        smp_warm = NodeAdjEDMSamplerHip(num_steps=max(1, min(T, args.warmup_num_steps)), **skw)
        flags = torch.from_numpy(weights.synth_flags(B, n, valid)).to(dev)

        def one_step(k, sampler):
            oa, on = sampler.sample(net, flags, num_node_chan=cfg.c_node, num_edge_chan=cfg.c_adj, seed=seed + 1000 * k,
                                    return_device=True)
            packed = dsg_dist.pack_results(oa.reshape(B, cfg.c_adj, n, n), on.reshape(B, n, cfg.c_node))
            return dsg_dist.gather_results(packed), dict(sampler.last_stats)

    def fence():
        if use_pg:
            dist.barrier()
        if not selftest:
            torch.cuda.synchronize(dev)

    def progress(msg):   # stderr only: stdout carries exactly one JSON line
        if rank == 0:
            sys.stderr.write(f"[bench] {msg}\n")
            sys.stderr.flush()

    for k in range(args.warmup):
        one_step(-1 - k, smp_warm)
    fence()
    progress(f"{args.warmup} warm-up sample(s) done; timing {args.steps} step(s) of T={T}, B={B} per GPU, {world} GPU(s)")
    t0 = time.perf_counter()
    nfe = 0
    out = None
    t_last = t0
    for k in range(args.steps):
        out, st = one_step(k, smp)
        nfe += st["net_forwards"]
        # a progress line about once a minute (dsg_sample synchronises once at its start, so the host loop follows the GPU
        # by at most one step); no extra synchronisation is added to the timed region for it
        if time.perf_counter() - t_last > 45.0:
            progress(f"step {k + 1}/{args.steps} enqueued after {time.perf_counter() - t0:.0f} s")
            t_last = time.perf_counter()
    fence()
    elapsed = time.perf_counter() - t0
    progress(f"timed region: {elapsed:.1f} s")
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert out.shape[0] == world_seen * B and bool(torch.isfinite(out).all())
    # what differs by rank in a multi-GPU run: every rank draws its own coin stream (seed + rank), so its number of network forwards
    # differs; `value` is graphs over the MAX-over-ranks time, i.e. the slowest rank's -- record the spread
    nfe_min = nfe_max = nfe
    if use_pg:
        tn = torch.tensor([float(nfe), -float(nfe)], dtype=torch.float64, device=dev)
        dist.all_reduce(tn, op=dist.ReduceOp.MAX)
        nfe_max, nfe_min = int(tn[0].item()), int(-tn[1].item())
    per_rank = {"net_forwards_per_step_min": nfe_min / args.steps, "net_forwards_per_step_max": nfe_max / args.steps,
                "net_forwards_per_step_rank0": nfe / args.steps,
                "gather_payload_bytes_per_rank": int(out.numel() // world_seen * out.element_size()),
                "gather_payload_bytes_total": int(out.numel() * out.element_size())}
    if world > 1:
        # rank order of the gather: every rank drew from its own seed, so the blocks must differ and block r must be rank r's
        mine = out[rank * B:(rank + 1) * B]
        other = out[((rank + 1) % world) * B:((rank + 1) % world + 1) * B]
        assert not torch.equal(mine, other), "two ranks produced identical graphs: seeds are not offset by rank"

    line = None
    if selftest:
        if rank == 0:
            line = {"metric": "launcher self-test (no GPU work)", "value": world * B * args.steps / max(elapsed, 1e-9),
                    "unit": "stand-in steps/s", "n_gpus": world, "world_size": world_seen, "steps": args.steps,
                    "warmup": args.warmup, "selftest": True, "backend": "gloo" if world > 1 else None,
                    "gathered_rows": int(out.shape[0]), "per_rank": per_rank, "cpu_baseline": None}
    else:
        # The tail the reference pays after sampling (sampler_node_adj.py:194-345): decode the 'bits' samples, gather the
        # decoded graphs and bring them to the host.  Reported beside `value`, never inside it.
        from diffusesg_amd import io as dsg_io
        n_adj_type, n_node_type = (51, 150) if args.config == "vg" else (7, 171) if args.config == "coco" else (2 ** cfg.c_adj, 2 ** (cfg.c_node - 4))
        raw_a, raw_x = dsg_dist.unpack_results(out[rank * B:(rank + 1) * B], cfg.c_adj, n, cfg.c_node)
        fence()
        t1 = time.perf_counter()
        qa, qn, bb = dsg_io.decode_bits(net, raw_a, raw_x, flags, n_adj_type, n_node_type, bbox=True)
        dec = dsg_dist.gather_results(dsg_io.pack_decoded(qa, qn, bb, flags))
        dec_host = dec.cpu()
        fence()
        tail = time.perf_counter() - t1
        assert dec_host.shape[0] == world_seen * B
        per_rank["decoded_gather_payload_bytes_per_rank"] = int(dec.numel() // world_seen * dec.element_size())

    if rank == 0 and not selftest:
        graphs = world * B * args.steps
        value = graphs / elapsed
        f_fwd = spec.flops_per_forward(cfg)
        mode = h.precision_mode()     # what the handle actually ran (options or DSG_* environment defaults), not the CLI flag
        # roofline of the dominant kernel (the GEMM): HIP events around every launch of eager forwards on the state the timed
        # run left in the workspace (random-data clocks, not zeros)
        ms, cnt, fl = (C.c_double * 5)(), (C.c_int64 * 5)(), (C.c_double * 5)()
        gemm_ms = C.c_double(0.0)
        iters = 3
        st_ptr = torch.cuda.current_stream(dev).cuda_stream
        h.check(h.L.dsg_profile_forward(h.raw, B, iters, ms, cnt, fl, C.byref(gemm_ms), C.c_void_p(st_ptr)), "dsg_profile_forward")
        # avg_launch_ms: HIP events recorded on the launch stream around every GEMM launch (eager forwards).
        # avg_launch_ms_inkernel: first-block-start -> last-block-end stamps (100 MHz constant clock) of back-to-back
        # launches; consecutive kernels overlap at their tails, so this one reads a few % high.
        # the dominant kernel class: the GEMM, except on the bf16 block pipeline, where a Swin block is two kernels (QKV + attention;
        # proj + MLP) and the second one carries most of the time and of the FLOPs
        bx_pipe = mode == "bf16" and bool(h.get_option("bf16_pipe")) and bool(h.get_option("bf16_proj_mlp")) and cnt[4] > 0
        dk = 4 if bx_pipe else 0
        gemm_avg_ms = ms[dk] / cnt[dk]
        gemm_avg_ms_inkernel = gemm_ms.value / cnt[0] if cnt[0] else None
        achieved = (fl[dk] / cnt[dk]) / (gemm_avg_ms * 1e-3) / 1e12
        clock_ghz = float(h.L.dsg_profile_clock_ghz(h.raw))   # shader clock held during those GEMM launches (in-kernel stamps)
        kinds = ["gemm", "window_attn", "row", "elementwise", "fused_blocks"]
        breakdown = {kinds[i]: {"ms_per_forward": ms[i] / iters, "launches_per_forward": cnt[i] // iters,
                                "tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if fl[i] > 0 and ms[i] > 0 else None} for i in range(5)}
        # HBM traffic of that kernel: PMC counters cannot be read from inside this process; the per-launch figure comes from
        # the committed rocprofv3 --pmc passes over this same command (tools/pmc_traffic.sh -> profiles/*/pmc_traffic.json)
        traffic, traffic_src = None, None
        prof_root = os.path.join(ROOT, "profiles")
        # (file, kernel family, condition): the committed PMC passes exist for the headline workload and for configs[4]'s per-GPU share
        mlp384 = "mlp384d_bx_kernel" if (bx_pipe and h.get_option("bf16_mlp") in (1, 6)) else "mlp384_bx_kernel"
        pmc_sets = [("pmc_traffic.json", "gemm4_f32_kernel", args.config == "vg" and B == 64 and mode == "f32"),
                    ("pmc_traffic_vg_B256.json", "gemm4_f32_kernel", args.config == "vg" and B == 256 and mode == "f32"),
                    ("pmc_traffic_coco_bf16.json", (mlp384, "mlp_bx_kernel") if bx_pipe else "gemm_bx_kernel",
                     args.config == "coco" and B == 512 and mode == "bf16")]
        for rnd in sorted(os.listdir(prof_root), reverse=True) if os.path.isdir(prof_root) else []:
            for fname, fam, cond in pmc_sets:
                pj = os.path.join(prof_root, rnd, fname)
                if cond and traffic is None and os.path.exists(pj):
                    tj = json.load(open(pj))
                    fams = [f for f in ((fam,) if isinstance(fam, str) else fam) if f in tj.get("kernels", {})]
                    if fams and len(fams) == (1 if isinstance(fam, str) else len(fam)):
                        n_l = sum(tj["kernels"][f]["launches"] for f in fams)
                        traffic = sum(tj["kernels"][f]["hbm_bytes_per_launch"] * tj["kernels"][f]["launches"] for f in fams) / n_l
                        traffic_src = f"profiles/{rnd}/{fname} (committed rocprofv3 --pmc passes of this command from an earlier run of this build's kernels, not this process)"
        # peak of the pipe the dominant kernel runs on, in ALGORITHMIC (2*M*N*K) FLOP/s: the split kernel issues six bf16
        # MFMA products per algorithmic product, so its ceiling is the bf16 dense peak / 6
        bf16_kern = ("mlp384d_bx_kernel + mlp_bx_kernel (proj + MLP half of a Swin block; the class's brackets also hold the PatchEmbed and "
                     "read-out launches, 2 of 20 on COCO)" if bx_pipe else
                     ("gemm_bx_kernel" if (mode == "bf16" and h.get_option("bf16_pipe")) else "gemm_bf16_kernel"))
        kern, peak = {"f32": ("gemm4_f32_kernel", PEAK_F32_MFMA_TFLOPS), "bf16": (bf16_kern, PEAK_BF16_MFMA_TFLOPS),
                      "f32-split": ("gemm_split2_kernel", PEAK_BF16_MFMA_TFLOPS / 6.0)}[mode]
        if mode != "f32":
            gemm_avg_ms_inkernel = None
        roofline = {"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak,
                    "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": gemm_avg_ms, "avg_launch_ms_inkernel": gemm_avg_ms_inkernel,
                    "flops_per_launch": fl[dk] / cnt[dk], "launches_per_forward": cnt[dk] // iters,
                    "held_clock_ghz": clock_ghz if mode == "f32" else None,
                    "clock_limited_peak": (clock_ghz * 1024 * 64 / 1e3) if (mode == "f32" and clock_ghz > 0) else None,
                    "forward_breakdown": breakdown}
        # whole-path achieved rate: graphs/s/GPU x forwards per graph x FLOPs per forward
        roofline["whole_path_tflops"] = nfe * B * f_fwd / elapsed / 1e12
        # against the peak of the pipe the mode's GEMMs run on (bf16 mode: most of the path's FLOPs are bf16-MFMA products; the
        # fp32 PatchEmbed / read-out / heads are priced against that peak too, which only understates the fraction)
        roofline["whole_path_frac"] = roofline["whole_path_tflops"] / peak
        # cpu_baseline on rank 0 at every world size: the timed region is over, the other ranks are idle at the final barrier (the
        # leg is bounded by --cpu-budget-s, well inside the process group's timeout)
        cpu = None
        if not args.no_cpu_baseline:
            progress(f"cpu_baseline: timing the oracle on the host cores for up to {args.cpu_budget_s:.0f} s")
            cpu = cpu_baseline(cfg, sd, T, valid, args.cpu_budget_s)
        # BASELINE.json's metric string for the configuration it is quoted on (VG shape: 30 valid nodes, T=1000); any other
        # workload is labelled plainly and described in config.workload
        metric_name = "scene-graphs/sec"
        if args.config == "vg" and T == 1000 and valid == 30 and args.solver == "heun":
            try:
                metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:
                metric_name = "scene-graphs/sec at N=30 nodes, T=1000 DDPM steps, 1/2/4/8 MI355X"
        line = {
            "metric": metric_name, "value": value, "unit": "scene-graphs/s", "n_gpus": world, "world_size": world_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 GEMM operands, f32 accumulate/activations",
                      "f32-split": "f32 (GEMM products as 3-way bf16 operand splits, six partial products, f32 accumulate)"}[mode],
            "data": "synthetic",
            "config": {"workload": f"{args.config}-bits N={n} valid={valid} C_adj={cfg.c_adj} C_node={cfg.c_node} "
                                   f"T={T} {args.solver} S_churn={args.s_churn:g} self_cond={int(cfg.self_condition)}",
                       "batch_per_gpu": B, "global_batch": world * B, "num_steps": T,
                       "warmup_num_steps": max(1, min(T, args.warmup_num_steps)),
                       "net_forwards_per_step": nfe / args.steps, "gflop_per_forward_per_graph": f_fwd / 1e9,
                       "hip_graph": not args.no_graph, "precision_mode": mode,
                       "parallelism": f"batch-sharded x{world}, one all-gather", "process_group": "nccl (RCCL)" if use_pg else None},
            "roofline": roofline, "cpu_baseline": cpu, "per_rank": per_rank,
            "tail": {"what": "on-GPU decode of the bits samples + packed int16 all-gather + D2H of the decoded graphs (once per step)",
                     "ms": 1e3 * tail, "value_with_tail": graphs / (elapsed + args.steps * tail)},
        }
    if line is not None:
        sys.stdout.write(json.dumps(line) + "\n")
        sys.stdout.flush()           # the line is out before any teardown
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    worker(args)


if __name__ == "__main__":
    main()
