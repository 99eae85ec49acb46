"""Child process of tests/test_hip_parity.py::test_rccl_collectives_world1: a world-size-1 RCCL ("nccl") process group on
cuda:0 through which the product's two collectives really run (DSG_FORCE_COLLECTIVE=1 disables their world-size-1 shortcuts):
`dist.gather_results` on both payloads of the sampling tail -- fp32 raw [B, C_adj*N^2 + N*C_node] and the int16-as-bytes decoded
pack -- and `dist.all_reduce_mean` on a gradient dict (R/utils/dist_training.py:170-195 is the reference's gather).  The
rendezvous environment is set before anything touches the GPU; the parent starts this file as a fresh interpreter."""
import os
import socket
import sys

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                  DSG_FORCE_COLLECTIVE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from diffusesg_amd import dist as D, io as IO  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dist.barrier()
    g = torch.Generator(device="cpu").manual_seed(3)
    B, N, Ca, Cn = 16, 64, 6, 12
    adj = torch.randn(B, Ca, N, N, generator=g).to(dev)
    node = torch.randn(B, N, Cn, generator=g).to(dev)
    packed = D.pack_results(adj, node)
    assert packed.shape == (B, Ca * N * N + N * Cn)
    out = D.gather_results(packed)
    assert out.data_ptr() != packed.data_ptr(), "the collective was skipped"
    assert out.dtype == packed.dtype and torch.equal(out.view(torch.uint8), packed.view(torch.uint8))
    ga, gn = D.unpack_results(out, Ca, N, Cn)
    assert torch.equal(ga, adj) and torch.equal(gn, node)
    # the decoded pack: int16 graph codes + fp32 bbox bytes viewed as int16 (io.pack_decoded), moved as uint8 by gather_results
    a_int = torch.randint(0, 51, (B, N, N), generator=g, dtype=torch.int32).to(dev)
    n_int = torch.randint(0, 150, (B, N), generator=g, dtype=torch.int32).to(dev)
    bbox = torch.rand(B, N, 4, generator=g).to(dev)
    nflags = (torch.rand(B, N, generator=g) < 0.5).to(dev)
    dec = IO.pack_decoded(a_int, n_int, bbox, nflags)
    assert dec.dtype == torch.int16
    out16 = D.gather_results(dec)
    assert out16.data_ptr() != dec.data_ptr() and out16.dtype == torch.int16
    assert torch.equal(out16.view(torch.uint8), dec.view(torch.uint8))
    ua, un, uf, ub = IO.unpack_decoded(out16, N, True)
    assert torch.equal(ua, a_int) and torch.equal(un, n_int) and torch.equal(uf, nflags) and torch.equal(ub, bbox)
    # gradient all-reduce (bucketed): sum over one rank / 1 == identity, through RCCL; two buckets forced
    # (at world size 1 the sum is the identity, so equality alone would also hold if no collective ran: count the calls)
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append((t.data_ptr(), t.numel()))
        return real_all_reduce(t, *a, **k)

    dist.all_reduce = counting_all_reduce
    grads = {f"p{i}": torch.randn(n_, generator=g).to(dev) for i, n_ in enumerate((1000, 70000, 3, 512 * 512))}
    ref = {k: v.clone() for k, v in grads.items()}
    D.all_reduce_mean(grads, bucket_bytes=300000)
    assert len(calls) == 2 and sum(c[1] for c in calls) == sum(v.numel() for v in ref.values()), calls   # two buckets forced
    for k in grads:
        assert torch.equal(grads[k], ref[k]), k
    # the flat form train_step_grads returns (what every multi-GPU training step reduces): the dict entries are VIEWS into one flat
    # buffer and ONE collective runs on that buffer, with no packing copy
    from diffusesg_amd.train import GradDict
    fg = GradDict()
    fg.flat = torch.cat([ref[k].reshape(-1) for k in ref]).clone()
    off = 0
    for k, v in ref.items():
        fg[k] = fg.flat[off:off + v.numel()].view_as(v)
        off += v.numel()
    del calls[:]
    D.all_reduce_mean(fg)
    assert calls == [(fg.flat.data_ptr(), fg.flat.numel())], calls
    for k in ref:
        assert torch.equal(fg[k], ref[k]) and fg[k].data_ptr() >= fg.flat.data_ptr(), k
    # ... and a GradDict with no dict entries at all still reduces its flat buffer (the emptiness shortcut used to return first)
    bare = GradDict()
    bare.flat = fg.flat.clone()
    del calls[:]
    D.all_reduce_mean(bare)
    assert calls == [(bare.flat.data_ptr(), bare.flat.numel())] and torch.equal(bare.flat, fg.flat)
    dist.all_reduce = real_all_reduce
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks timing reduction
    assert float(t) == 1.25
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)


if __name__ == "__main__":
    main()
