"""Multi-GPU side of the sampling path: independent samples shard batch-wise, one collective at the end.

Reference: each rank samples its DistributedSampler shard with per-GPU batch `batch_size // world_size`
(R/runner/sampler/sampler_utils.py:32-36), seeds are offset by rank (R/utils/arg_parser.py:293-294) and the
results are collected with 13 `all_gather_into_tensor` calls staged through the host
(R/runner/sampler/sampler_node_adj.py:331-345, R/utils/dist_training.py:170-195).  Here: one process per GPU,
no data-path collective during sampling, and ONE all-gather (RCCL over xGMI with backend "nccl"; gloo in the
CPU tests) of a packed per-rank buffer.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def rank_seed(base_seed: int, rank: int) -> int:
    return int(base_seed) + int(rank)


def shard_batch(total_batch: int, world_size: int) -> int:
    """Per-rank batch, like the reference: batch_size // world_size (remainder samples are not generated)."""
    if world_size < 1 or total_batch < world_size:
        raise ValueError("batch smaller than world size")
    return total_batch // world_size


def pack_results(adj: torch.Tensor, node: torch.Tensor) -> torch.Tensor:
    """[B,C,N,N] + [B,N,Cn] -> one contiguous [B, C*N*N + N*Cn] buffer (the unit of the single all-gather)."""
    B = adj.shape[0]
    return torch.cat([adj.reshape(B, -1), node.reshape(B, -1)], dim=1).contiguous()


def unpack_results(packed: torch.Tensor, c_adj: int, n: int, c_node: int) -> Tuple[torch.Tensor, torch.Tensor]:
    B = packed.shape[0]
    sa = c_adj * n * n
    return packed[:, :sa].reshape(B, c_adj, n, n), packed[:, sa:].reshape(B, n, c_node)


def gather_results(packed: torch.Tensor) -> torch.Tensor:
    """All ranks get [world*B, D] in rank order.  One collective; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return packed
    if dist.get_world_size() == 1 and not os.environ.get("DSG_FORCE_COLLECTIVE"):   # (bench.py --rehearse-collectives sets it)
        return packed
    world = dist.get_world_size()
    if packed.dtype in (torch.int16, torch.uint16):
        # RCCL/NCCL have no 16-bit integer type (ncclDataType: 8/32/64-bit integers and floats only): move the same bytes as uint8
        flat = packed.contiguous().view(torch.uint8)
        out = torch.empty((world * flat.shape[0],) + tuple(flat.shape[1:]), dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(out, flat)
        return out.view(packed.dtype)
    out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return out


def all_reduce_mean(grads, bucket_bytes: int = 256 << 20):
    """Data-parallel gradient averaging of a training iteration (the reference wraps the model in DistributedDataParallel,
    R/utils/dist_training.py, whose backward averages gradients over ranks): the gradient tensors (dict or list, as
    `train_step_grads` returns them) are packed into a few large flat buckets -- xGMI rings are per-link bound, so few large
    all-reduces beat one per tensor -- summed with one `all_reduce` each (RCCL with backend "nccl"; gloo in the CPU test) and
    divided by the world size, in place.  Identity when not distributed."""
    tensors = list(grads.values()) if isinstance(grads, dict) else list(grads)
    flat = getattr(grads, "flat", None)   # looked at BEFORE the emptiness shortcut: a GradDict's payload is its flat buffer
    if not (dist.is_available() and dist.is_initialized()) or (not tensors and flat is None):
        return grads
    if dist.get_world_size() == 1 and not os.environ.get("DSG_FORCE_COLLECTIVE"):   # (the world-size-1 RCCL test sets it)
        return grads
    world = dist.get_world_size()
    if flat is not None:   # train_step_grads' GradDict: every gradient is a view into one flat buffer -- one collective, no packing
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        return grads
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        if size and size + t.numel() * t.element_size() > bucket_bytes:
            flush()
        bucket.append(t)
        size += t.numel() * t.element_size()
    flush()
    return grads
