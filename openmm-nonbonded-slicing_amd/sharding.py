"""Multi-GPU sharding of ONE SlicedNonbondedForce evaluation.

One process per GPU.  Rank g of G owns the PME charge grids of the subsets J with J % G == g and the direct-space tiles of
the 32-atom i-blocks I with I % G == g (the engine applies the same rules from snb_config.shard_rank/shard_count; block
indices follow the sorted atom order, which every rank derives identically from the same positions); every rank holds all
positions.  The only exchange on the path is a sum: one all-reduce of the N x 3 partial forces per step, plus -- on energy
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
