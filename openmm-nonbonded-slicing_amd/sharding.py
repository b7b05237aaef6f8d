"""Multi-GPU sharding of ONE SlicedNonbondedForce evaluation.

One process per GPU.  Rank g of G owns the PME charge grids of the subsets J with J % G == g and the direct-space tiles of
the 32-atom i-blocks I with I % G == g (the engine applies the same rules from snb_config.shard_rank/shard_count; block
indices follow the sorted atom order, which every rank derives identically from the same positions); every rank holds all
positions.  With more ranks than grids the ranks that carry a grid are the slow ones: `balance_block_ranges` turns measured
per-rank times into uneven block ranges for `snb_set_shard_blocks` (I % period in [begin, end)), the counterpart of the
load balancing between devices that OpenMM's parallel kernels give the reference.  The only exchange on the path is a sum: one all-reduce of the N x 3 partial forces per step, plus -- on energy
steps -- one all-reduce of the S x 2 raw slice energies.  RCCL ("nccl" backend) on MI355X, gloo in the CPU tests.
"""
from __future__ import annotations


def owned_subsets(n_subsets: int, rank: int, world: int):
    """Subsets whose PME grid lives on `rank` (csrc/engine.hip: `s % shard_count == shard_rank`)."""
    return [s for s in range(n_subsets) if s % world == rank]


def owned_work_items(n_items: int, rank: int, world: int):
    """Direct-space i-blocks of `rank` (csrc/neighbor.hip k_nbBuildTiles: `block % shard_count == shard_rank`; the rule must not depend
    on the order in which a rank's own builder emits its work items, which is not reproducible across GPUs)."""
    return list(range(rank, n_items, world))


def default_block_ranges(world: int):
    """The ownership the engine starts with: rank r owns I % world == r, i.e. range (r, r + 1) of period `world`."""
    return [(r, r + 1) for r in range(world)], world


def balance_block_ranges(direct_ms, other_ms, period: int = 128):
    """Block ranges that equalise direct + other time across ranks.

    direct_ms[r]: time rank r spent in the direct-space kernel with its current share of the blocks (the shares of all ranks sum
    to one, so sum(direct_ms) is the cost c of the whole direct-space pass on one rank); other_ms[r]: everything else on rank r
    (gather, reciprocal path, its share of the rebuild).  Finds the level T with sum_r max(0, T - other_r) = c, gives rank r the
    share max(0, T - other_r) / c, and rounds the shares to `period` slots (largest remainders first).  Returns
    ([(begin, end)] per rank, period); ranges are contiguous, cover [0, period) and may be empty.  Pure host arithmetic, the same
    on every rank when fed the all-gathered times."""
    world = len(direct_ms)
    assert len(other_ms) == world and world >= 1 and period >= world
    c = float(sum(direct_ms))
    if c <= 0.0:
        share = [1.0 / world] * world
    else:
        lo, hi = min(other_ms), max(other_ms) + c
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            if sum(max(0.0, mid - o) for o in other_ms) < c: lo = mid
            else: hi = mid
        share = [max(0.0, hi - o) / c for o in other_ms]
        tot = sum(share); share = [x / tot for x in share]
    slots = [int(x * period) for x in share]
    rest = sorted(range(world), key=lambda r: (share[r] * period - slots[r], -r), reverse=True)
    for r in rest[:period - sum(slots)]:
        slots[r] += 1
    ranges, b = [], 0
    for r in range(world):
        ranges.append((b, b + slots[r])); b += slots[r]
    assert b == period
    return ranges, period


def allreduce_partials(forces, slice_energies=None, group=None):
    """Sum the per-rank partial forces (and raw slice energies) in place.  `forces`/`slice_energies` are torch tensors
    living on the rank's device; with world size 1 this is a no-op."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return forces, slice_energies
    dist.all_reduce(forces, op=dist.ReduceOp.SUM, group=group)
    if slice_energies is not None:
        dist.all_reduce(slice_energies, op=dist.ReduceOp.SUM, group=group)
    return forces, slice_energies
