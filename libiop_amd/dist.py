"""Multi-GPU sharding of the prover pipeline (one process per GPU, torch.distributed; RCCL on GPUs, gloo in
the CPU tests).  SURVEY.md §8(e): a codeword over a 2^m-point affine subspace of a polynomial with 2^d
coefficients is 2^(m-d) independent transforms on contiguous blocks, so rank g of N owns the contiguous block
[g * 2^m / N, (g+1) * 2^m / N) of every oracle.  FRI cosets (contiguous, subspace.tcc:73-91) and Merkle leaves
are local to a rank; the only cross-rank data are N sub-tree roots (32 bytes each) per Merkle tree."""
import hashlib

import numpy as np

from . import host


def shard_range(n, rank, world):
    per = n // world
    return rank * per, per


def local_subdomain(basis, shift, rank, world):
    """The rank's block of a domain as an affine subspace: first m - log2(world) basis vectors, shift moved by the
    block's high-bit combination (element_by_index(rank * 2^m / world), subspace.tcc:56-71)."""
    basis = np.asarray(basis, dtype=np.uint64)
    lg = world.bit_length() - 1
    assert (1 << lg) == world and lg <= basis.shape[0]
    m = basis.shape[0]
    s = np.array(shift, dtype=np.uint64).copy()
    for k in range(lg):
        if (rank >> k) & 1:
            s ^= basis[m - lg + k]
    return basis[: m - lg], s


def sharded_lde(lib, torch, d_coeffs, n_coeffs, basis, shift, rank, world):
    """This rank's contiguous block of FFT_over_field_subset(coeffs, domain): no collective."""
    m = np.asarray(basis).shape[0]
    d = 0 if n_coeffs <= 1 else (n_coeffs - 1).bit_length()
    cosets = 1 << (m - d)
    if cosets % world:
        raise ValueError("need at least one coset per rank: 2^(m-d) = %d cosets, %d ranks" % (cosets, world))
    per = cosets // world
    out = torch.empty((per << d, 3), dtype=torch.int64, device=d_coeffs.device)
    lib.additive_LDE_dev(d_coeffs.data_ptr(), n_coeffs, basis, shift, rank * per, per, out.data_ptr())
    return out


def sharded_merkle_root(lib, torch, dist, d_oracles, n_local, coset_size, rank, world):
    """Merkle tree over oracles sharded by contiguous blocks: every rank builds the sub-tree over its leaves; the
    N sub-roots are all-gathered (32 bytes each) and the top log2(N) levels are finished on the host
    (node = H(left || right), blake2b.cpp:28-48).  Returns (global root bytes, local node buffer)."""
    leaves = n_local // coset_size
    dev = d_oracles[0].device
    nodes = torch.empty((2 * leaves - 1, 32), dtype=torch.uint8, device=dev)
    lib.merkle_tree_dev([o.data_ptr() for o in d_oracles], 24, n_local, coset_size, nodes.data_ptr())
    lib.synchronize()
    sub_root = nodes[0].clone()
    if world == 1:
        return bytes(sub_root.cpu().numpy()), nodes
    gathered = [torch.empty_like(sub_root) for _ in range(world)]
    dist.all_gather(gathered, sub_root)
    level = [bytes(g.cpu().numpy()) for g in gathered]
    while len(level) > 1:
        level = [hashlib.blake2b(level[2 * i] + level[2 * i + 1], digest_size=32).digest() for i in range(len(level) // 2)]
    return level[0], nodes


def sharded_fri_fold(lib, torch, d_f_local, basis, shift, coset_size, x_i, rank, world):
    """evaluate_next_f_i_over_entire_domain on this rank's block: cosets are contiguous, so the block folds as
    the affine sub-domain local_subdomain(...) with the same x_i; the result is the rank's block of f_{i+1}."""
    b_loc, s_loc = local_subdomain(basis, shift, rank, world)
    out = torch.empty((d_f_local.shape[0] // coset_size, 3), dtype=torch.int64, device=d_f_local.device)
    lib.fri_fold_dev(d_f_local.data_ptr(), b_loc, s_loc, coset_size, x_i, out.data_ptr())
    return out
