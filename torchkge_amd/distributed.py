# -*- coding: utf-8 -*-
"""Multi-GPU plumbing of the link-prediction path: one process per GPU,
torch.distributed (backend 'nccl' = RCCL over xGMI on MI355X; 'gloo' in the CPU
tests).  The reference has no distributed code at all; the
partitioning below is the one BASELINE.json's north star names:

  * the entity table is row-sharded: rank p scores candidates
    [p*ceil(N/P), min(N, (p+1)*ceil(N/P))) -- and, after `shard_model_`, HOLDS only those rows of
    every entity-indexed table (relation tables stay replicated); the (2B, K) query rows of a batch
    are then built by the rank that owns each query's entity and summed over the ranks
    (zeros elsewhere: exact), one all-reduce per query matrix;
  * exchange='scores': the partial score tiles S_p (B, N/P) are exchanged.  Since r04 as an ALL-TO-ALL of row
    blocks: every rank receives only the score rows of the B/P queries IT ranks, as P rank-major tiles, and ranks
    them in place (`all_to_all_rows` + kge_filtered_rank_from_tiles) -- 1/P of the bytes of the all-gather and 1/P
    of its ranking work per rank; `all_gather_columns` (every rank gets the full (B, N) matrix and ranks all of
    it) remains for engines without tile ranking and for the drop-in composition;
  * exchange='counts': ranks are sums over candidates, so each rank counts
    `>=` on its shard and ONE all-reduce of 3*B int32 (raw, filter correction,
    found flag) gives bit-identical ranks -- B*12 bytes instead of B*N*4;
  * the true-candidate score is computed by the shard that owns the true
    entity and summed with zeros from the others (x + 0 is exact).
Pure functions of (N, P, p) so P virtual shards can be checked on one device.
"""
import os

import torch
import torch.distributed as dist

# KGE_FORCE_COLLECTIVES=1 (debug): take the sharded code paths and issue the collectives even
# in a world of ONE rank, so the RCCL calls can be exercised on a single-GPU box.
MIN_WORLD = 1 if os.environ.get('KGE_FORCE_COLLECTIVES') == '1' else 2


def multi(world):
    """True when the multi-rank code paths apply to a world of this size."""
    return world >= MIN_WORLD and dist.is_available() and dist.is_initialized()


def backend_name(group=None):
    """'nccl' (= RCCL), 'gloo', ... or None without a process group."""
    if dist.is_available() and dist.is_initialized():
        return str(dist.get_backend(group))
    return None


def world_and_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_size(n, world):
    return (n + world - 1) // world


def shard_range(n, world, rank):
    """Contiguous block partition: [lo, hi) of `n` items owned by `rank`."""
    per = shard_size(n, world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def shard_model_(model, group=None):
    """Row-shard `model`'s entity tables over the process group, in place: this rank keeps rows
    shard_range(n_ent, world, rank) of every entity-indexed table.  Every rank must hold the same
    tables before the call (e.g. broadcast from rank 0).  Returns (lo, hi)."""
    world, rank = world_and_rank(group)
    lo, hi = shard_range(model.n_ent, world, rank)
    model.shard_entities_(lo, hi)
    return lo, hi


def all_reduce_sum(t, group=None):
    """In-place SUM all-reduce (no-op when not distributed)."""
    if dist.is_available() and dist.is_initialized() and multi(dist.get_world_size(group)):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_max(t, group=None):
    """In-place MAX all-reduce (no-op when not distributed)."""
    if dist.is_available() and dist.is_initialized() and multi(dist.get_world_size(group)):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t


def all_gather_columns(local, n_total, group=None):
    """Partial score tiles (B, n_p) of every rank -> (B, n_total): ONE
    all-gather of equal-size (padded) tiles, then a strided copy."""
    world, rank = world_and_rank(group)
    if not multi(world):
        return local
    B = local.shape[0]
    per = shard_size(n_total, world)
    send = local
    if local.shape[1] != per:                      # last shard may be short: pad
        send = local.new_zeros(B, per)
        send[:, :local.shape[1]] = local
    send = send.contiguous()
    gathered = local.new_empty(world, B, per)
    dist.all_gather_into_tensor(gathered.view(world * B, per), send, group=group)
    full = gathered.permute(1, 0, 2).reshape(B, world * per)
    return full[:, :n_total].contiguous()


_A2A_FALLBACK = set()     # (backend, device type) pairs whose all_to_all_single is not implemented (gloo on HIP tensors)


def all_to_all_rows(local, recv, group=None):
    """Score tile (P * m, per) of this rank -- rows [j*m, (j+1)*m) are the queries rank j ranks -- ->
    recv (P, m, per): tile p = the scores rank p computed for THIS rank's m queries (rank-major, no re-layout).
    ONE all-to-all: (P-1)/P of one local tile per rank on the fabric.  Returns True when recv[rank] was filled as
    well, False when the own block was left where the scorer wrote it (local[rank*m : (rank+1)*m]) -- a world of one:
    nothing is exchanged, the rank kernel reads the block in place."""
    world, rank = world_and_rank(group)
    P, m, per = recv.shape
    if not multi(world):
        return False
    assert P == world and local.shape[0] == world * m and local.is_contiguous() and recv.is_contiguous()
    key = (backend_name(group), local.device.type)
    if world == 1:          # (forced collectives on a world of one: nothing to exchange, the own block is ranked in place)
        return False
    if key[0] == 'nccl':
        # RCCL: the plain equal-split all-to-all (the own block travels as a device copy inside the collective: 1/P of
        # one local tile).  A grouped send / recv form with a zero-sized own entry would save that copy, but it is a
        # far less travelled path of the backend and cannot be exercised on the one-GPU boxes this engine is built on.
        dist.all_to_all_single(recv.view(world * m, per), local, group=group)
        return True
    if key not in _A2A_FALLBACK:
        try:
            dist.all_to_all_single(recv.view(world * m, per), local, group=group)
            return True
        except (RuntimeError, NotImplementedError):
            _A2A_FALLBACK.add(key)      # unsupported-op errors are raised before anything is sent, on every rank alike
    # debugging backends only (two gloo ranks sharing one GPU): gather every rank's tile, keep this rank's row block
    gathered = local.new_empty(world, world * m, per)
    dist.all_gather_into_tensor(gathered.view(world * world * m, per), local, group=group)
    recv.copy_(gathered[:, rank * m:(rank + 1) * m])
    return True


def all_gather_blocks(block, out, group=None):
    """Equal-size row blocks (m, d) of every rank -> out (P * m, d), rank-major: ONE all-gather."""
    world, rank = world_and_rank(group)
    if not multi(world):
        out.copy_(block)
        return out
    dist.all_gather_into_tensor(out, block.contiguous(), group=group)
    return out


def all_gather_facts(local, n_total, group=None):
    """(4, n_p) rank rows of every query shard -> (4, n_total)."""
    world, rank = world_and_rank(group)
    if not multi(world):
        return local
    per = shard_size(n_total, world)
    send = local
    if local.shape[1] != per:
        send = local.new_zeros(local.shape[0], per)
        send[:, :local.shape[1]] = local
    send = send.contiguous()
    gathered = local.new_empty(world, local.shape[0], per)
    dist.all_gather_into_tensor(gathered.view(world * local.shape[0], per), send, group=group)
    return gathered.permute(1, 0, 2).reshape(local.shape[0], world * per)[:, :n_total].contiguous()
