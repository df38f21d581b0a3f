"""Multi-GPU sharding of the prover pipeline (one process per GPU, torch.distributed; RCCL on GPUs, gloo in
the CPU tests).  SURVEY.md §8(e): a codeword over a 2^m-point affine subspace of a polynomial with 2^d
coefficients is 2^(m-d) independent transforms on contiguous blocks, so rank g of N owns the contiguous block
[g * 2^m / N, (g+1) * 2^m / N) of every oracle.  FRI cosets (contiguous, subspace.tcc:73-91) and Merkle leaves
are local to a rank; the only cross-rank data are N sub-tree roots (32 bytes each) per Merkle tree.

Streams: with Library.set_stream(torch.cuda.current_stream().cuda_stream) — what bench.py and the tools do — library kernels, torch
ops and torch.distributed collectives are ordered by that one stream (a collective makes the current stream wait for it), so nothing
here waits on the host: sharded_merkle_root, distributed_fft and the sharded operator sets only enqueue.  When the library runs on
its own stream instead, the same functions fall back to host synchronisation in both directions (_lib_to_torch / _torch_to_lib)."""
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


def sharded_lde_batch(lib, torch, d_coeffs_list, n_coeffs, basis, shift, rank, world):
    """sharded_lde for several polynomials over one domain: phase 1 of the transforms runs once for the batch."""
    m = np.asarray(basis).shape[0]
    d = 0 if n_coeffs <= 1 else int(n_coeffs - 1).bit_length()
    cosets = 1 << (m - d)
    if cosets % world:
        raise ValueError("fewer cosets than ranks: use distributed_fft for this transform")
    per = cosets // world
    outs = [torch.empty((per << d, 3), dtype=torch.int64, device=c.device) for c in d_coeffs_list]
    lib.additive_LDE_batch_dev([c.data_ptr() for c in d_coeffs_list], n_coeffs, basis, shift, rank * per, per, [o.data_ptr() for o in outs])
    return outs


def sharded_merkle_root(lib, torch, dist, d_oracles, n_local, coset_size, rank, world):
    """Merkle tree over oracles sharded by contiguous blocks: every rank builds the sub-tree over its leaves; the
    N sub-roots are all-gathered (32 bytes each) and the top log2(N) levels are finished on the host
    (node = H(left || right), blake2b.cpp:28-48).  Returns (global root bytes, local node buffer)."""
    leaves = n_local // coset_size
    dev = d_oracles[0].device
    nodes = torch.empty((2 * leaves - 1, 32), dtype=torch.uint8, device=dev)
    lib.merkle_tree_dev([o.data_ptr() for o in d_oracles], 24, n_local, coset_size, nodes.data_ptr())
    _lib_to_torch(lib, torch)
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


# ---------------------------------------------------------------------------------------------------------------
# ONE additive FFT sharded across N = 2^r GPUs (the case that does need exchange steps: the polynomial is as long
# as the domain, so the transform does not split into independent cosets).
#
# Gao–Mateer's recursion (fft.tcc:55-96) splits the polynomial r times into 2^r sub-polynomials that are
# interleaved in the coefficient index (sub-polynomial s = index mod 2^r).  Rank rho works on s = rev_r(rho):
#   1. transpose: block-distributed coefficients -> "s-cyclic" layout (ONE all-to-all: RCCL over xGMI on GPUs),
#   2. the top r levels: twist by the rank's slice of the power table and run the Taylor network over the
#      local index bits (iopx_add_taylor_gf192_dev); the r(r+1)/2 network operations that touch the index bits
#      which now identify ranks are XORs of whole (or half) shards exchanged between peer ranks,
#   3. a complete LOCAL FFT of the rank's sub-polynomial over the depth-r recursed domain (existing kernels),
#   4. the last r butterfly levels (fft.tcc:102-120, stride >= shard size): peers exchange shards and apply
#      iopx_add_combine_gf192_dev.
# Rank rho ends with the natural-order block rho of the evaluations (same sharding as sharded_lde).
# ---------------------------------------------------------------------------------------------------------------
def _rev(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


class DistributedFFTPlan:
    """Per-(domain, rank) constants of the sharded transform; build once, reuse for every polynomial."""

    def __init__(self, lib, torch, basis, shift, rank, world, device):
        basis = np.asarray(basis, dtype=np.uint64)
        self.m = m = basis.shape[0]
        self.r = r = world.bit_length() - 1
        if (1 << r) != world or r > m - 1:
            raise ValueError("world size must be a power of two smaller than the domain")
        self.rank, self.world, self.s = rank, world, _rev(rank, r)
        self.n_loc = 1 << (m - r)
        b = [host.gf_from_words(w) for w in basis]
        sh = host.gf_from_words(np.asarray(shift, dtype=np.uint64))
        self.rec, self.rs, self.twist, self.twist_inv = [], [], [], []
        for j in range(r):                      # fft.tcc:57-96 for the top r levels
            beta = b[m - 1 - j]
            binv = host.gf_inv(beta)
            # this rank's slice of the level-j twist: beta^((l * 2^r + s) >> j) = beta^(s >> j) * (beta^(2^(r-j)))^l
            base = beta
            for _ in range(r - j):
                base = host.gf_sq(base)
            init = 1
            e, acc = self.s >> j, beta
            while e:
                if e & 1:
                    init = host.gf_mul(init, acc)
                acc = host.gf_sq(acc)
                e >>= 1
            tab = torch.empty((self.n_loc, 3), dtype=torch.int64, device=device)
            lib.pow_table_dev(tab.data_ptr(), self.n_loc, host.gf_to_words(base), host.gf_to_words(init))
            self.twist.append(tab)
            # the inverse transform multiplies by the inverse powers (fft.tcc:189-199)
            itab = torch.empty((self.n_loc, 3), dtype=torch.int64, device=device)
            lib.pow_table_dev(itab.data_ptr(), self.n_loc, host.gf_to_words(host.gf_inv(base)), host.gf_to_words(host.gf_inv(init)))
            self.twist_inv.append(itab)
            newb = []
            for i in range(m - 1 - j):
                nb = host.gf_mul(b[i], binv)
                newb.append(nb)
                b[i] = host.gf_sq(nb) ^ nb
            self.rec.append(newb)
            ns = host.gf_mul(sh, binv)
            self.rs.append(ns)
            sh = host.gf_sq(ns) ^ ns
            b = b[: m - 1 - j]
        self.local_basis = np.array([host.gf_to_words(v) for v in b], dtype=np.uint64).reshape(-1, 3)
        self.local_shift = host.gf_to_words(sh)
        _lib_to_torch(lib, torch)


def _torch_to_lib(lib, torch, t):
    """Before the library consumes a buffer that torch ops or collectives produced: nothing to do when the library enqueues on
    torch's current stream (stream order; collectives make the current stream wait for them), a host wait otherwise."""
    if t.device.type != "cpu" and not lib.shares_stream_with(torch, t.device):
        torch.cuda.current_stream(t.device).synchronize()


def _lib_to_torch(lib, torch):
    """Before torch ops or collectives consume what library calls produced: stream order when the stream is shared."""
    if not lib.shares_stream_with(torch):
        lib.synchronize()


def _exchange(torch, dist, send_t, peer, recv_like):
    """Symmetric shard exchange with one peer (NCCL/RCCL send+recv pair or gloo)."""
    recv = torch.empty_like(recv_like)
    ops = [dist.P2POp(dist.isend, send_t, peer), dist.P2POp(dist.irecv, recv, peer)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    return recv


def _send(torch, dist, t, peer):
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, peer)]):
        w.wait()


def _recv(torch, dist, like, peer):
    buf = torch.empty_like(like)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, peer)]):
        w.wait()
    return buf


def distributed_fft(lib, torch, dist, plan, d_block):
    """d_block: this rank's contiguous block of the 2^m coefficients ((2^m / N, 3) int64).  Returns the rank's
    contiguous block of additive_FFT(coeffs, domain) (natural order)."""
    r, m, s, world, rank, n_loc = plan.r, plan.m, plan.s, plan.world, plan.rank, plan.n_loc
    # 1. transpose to the s-cyclic layout: coefficient (rank * n_loc + t) goes to the rank of sub-polynomial
    #    t mod 2^r, local slot rank * (n_loc / N) + t // N
    x = d_block.reshape(n_loc // world, world, 3).permute(1, 0, 2).contiguous()     # [s', t // N, :]
    x = x[[_rev(q, r) for q in range(world)]].contiguous()                           # chunk q -> sub-polynomial rev(q)
    S = torch.empty_like(x)
    if world > 1:
        dist.all_to_all_single(S.view(-1), x.view(-1))
    else:
        S.copy_(x)
    S = S.reshape(n_loc, 3)
    peer_of = lambda s2: _rev(s2, r)
    # 2. top r levels
    for j in range(r):
        _torch_to_lib(lib, torch, S)
        lib.taylor_dev(S.data_ptr(), m - r, plan.twist[j].data_ptr())
        _lib_to_torch(lib, torch)
        # network operations on global index bits (k+1, k), k = r-1 .. j
        for k in range(r - 1, j - 1, -1):
            if k + 1 < r:       # both bits select ranks: whole-shard XORs
                hi, lo = (s >> (k + 1)) & 1, (s >> k) & 1
                if (hi, lo) == (1, 1):
                    _send(torch, dist, S, peer_of(s ^ (1 << k)))
                elif (hi, lo) == (1, 0):
                    S ^= _recv(torch, dist, S, peer_of(s | (1 << k)))
                if (hi, lo) == (1, 0):
                    _send(torch, dist, S, peer_of(s ^ (3 << k)))
                elif (hi, lo) == (0, 1):
                    S ^= _recv(torch, dist, S, peer_of(s ^ (3 << k)))
            else:               # bit k selects the rank, bit k+1 is local index bit 0
                bit = (s >> k) & 1
                peer = peer_of(s ^ (1 << k))
                odd = S[1::2]
                if bit == 1:    # (1,1) -> (1,0); then (0,1) += (1,0)
                    _send(torch, dist, odd.contiguous(), peer)
                    S[0::2] ^= _recv(torch, dist, odd.contiguous(), peer)
                else:
                    S[1::2] ^= _recv(torch, dist, odd.contiguous(), peer)
                    _send(torch, dist, S[1::2].contiguous(), peer)
    # 3. local transform of the sub-polynomial over the recursed domain
    loc = torch.empty_like(S)
    _torch_to_lib(lib, torch, S)
    lib.additive_FFT_dev(S.data_ptr(), n_loc, plan.local_basis, plan.local_shift, loc.data_ptr())
    _lib_to_torch(lib, torch)
    # 4. the last r butterfly levels across blocks
    cur = loc
    for t in range(r):
        peer = rank ^ (1 << t)
        other = _exchange(torch, dist, cur, peer, cur)
        upper = (rank >> t) & 1
        a, bb = (other, cur) if upper else (cur, other)
        lvl = r - 1 - t                                   # recursion level that produced these twiddles
        B = np.array([host.gf_to_words(v) for v in plan.rec[lvl]], dtype=np.uint64).reshape(-1, 3)
        out = torch.empty_like(cur)
        _torch_to_lib(lib, torch, other)
        lib.combine_dev(a.data_ptr(), bb.data_ptr(), out.data_ptr(), n_loc, (rank & ((1 << t) - 1)) * n_loc, B,
                        host.gf_to_words(plan.rs[lvl]), upper)
        _lib_to_torch(lib, torch)
        cur = out
    return cur


def distributed_ifft(lib, torch, dist, plan, d_block):
    """The inverse of distributed_fft (additive_IFFT, fft.tcc:126-204, on a codeword as long as its domain — IFFT_over_field_subset on a
    full codeword, fft.tcc:421-433): d_block is this rank's contiguous block of the 2^m evaluations, the result its contiguous block of
    the coefficients.  Every step of the forward transform undone in reverse order: the last r butterfly levels across blocks (peer
    exchange + iopx_add_combine_inv_gf192_dev), a complete local IFFT over the depth-r recursed domain, the top r levels (network
    operations on rank bits as shard exchanges — each is its own inverse in characteristic 2 — then iopx_add_taylor_inv_gf192_dev),
    and the transpose back (ONE all-to-all)."""
    r, m, s, world, rank, n_loc = plan.r, plan.m, plan.s, plan.world, plan.rank, plan.n_loc
    peer_of = lambda s2: _rev(s2, r)
    # 4'. the last r butterfly levels across blocks, outermost first
    cur = d_block.contiguous()
    for t in range(r - 1, -1, -1):
        peer = rank ^ (1 << t)
        _lib_to_torch(lib, torch)
        other = _exchange(torch, dist, cur, peer, cur)
        upper = (rank >> t) & 1
        lo, up = (other, cur) if upper else (cur, other)
        lvl = r - 1 - t
        B = np.array([host.gf_to_words(v) for v in plan.rec[lvl]], dtype=np.uint64).reshape(-1, 3)
        out = torch.empty_like(cur)
        _torch_to_lib(lib, torch, other)
        lib.combine_inv_dev(lo.data_ptr(), up.data_ptr(), out.data_ptr(), n_loc, (rank & ((1 << t) - 1)) * n_loc, B, host.gf_to_words(plan.rs[lvl]), upper)
        _lib_to_torch(lib, torch)
        cur = out
    # 3'. local inverse transform over the recursed domain
    S = torch.empty_like(cur)
    _torch_to_lib(lib, torch, cur)
    lib.additive_IFFT_dev(cur.data_ptr(), plan.local_basis, plan.local_shift, S.data_ptr())
    _lib_to_torch(lib, torch)
    # 2'. top r levels, innermost first; within a level the operations on global index bits (k+1, k) run k = j .. r-1
    for j in range(r - 1, -1, -1):
        for k in range(j, r):
            if k + 1 < r:       # both bits select ranks: (0,1) += (1,0), then (1,0) += (1,1)
                hi, lo_ = (s >> (k + 1)) & 1, (s >> k) & 1
                if (hi, lo_) == (1, 0):
                    _send(torch, dist, S, peer_of(s ^ (3 << k)))
                elif (hi, lo_) == (0, 1):
                    S ^= _recv(torch, dist, S, peer_of(s ^ (3 << k)))
                if (hi, lo_) == (1, 1):
                    _send(torch, dist, S, peer_of(s ^ (1 << k)))
                elif (hi, lo_) == (1, 0):
                    S ^= _recv(torch, dist, S, peer_of(s | (1 << k)))
            else:               # bit k selects the rank, bit k+1 is local index bit 0: (0,1) += (1,0), then (1,0) += (1,1)
                bit = (s >> k) & 1
                peer = peer_of(s ^ (1 << k))
                if bit == 0:
                    _send(torch, dist, S[1::2].contiguous(), peer)                     # this rank's (1,0) quarter to the holder of (0,1)
                    S[1::2] ^= _recv(torch, dist, S[1::2].contiguous(), peer)           # (1,0) += (1,1)
                else:
                    S[0::2] ^= _recv(torch, dist, S[1::2].contiguous(), peer)           # (0,1) += (1,0)
                    _send(torch, dist, S[1::2].contiguous(), peer)
        _torch_to_lib(lib, torch, S)
        lib.taylor_inv_dev(S.data_ptr(), m - r, plan.twist_inv[j].data_ptr())
        _lib_to_torch(lib, torch)
    # 1'. transpose back: s-cyclic -> block distribution
    x = S.reshape(world, n_loc // world, 3)                                              # chunk q: local slots [q n_loc/N, ...) came from rank q
    back = torch.empty_like(x)
    if world > 1:
        dist.all_to_all_single(back.view(-1), x.contiguous().view(-1))
    else:
        back.copy_(x)
    # chunk q received from the rank of sub-polynomial rev(q): its entries are coefficients rank * n_loc + t with t mod N = rev(q)
    back = back[[_rev(q, r) for q in range(world)]]                                      # [s', t // N, :]
    return back.permute(1, 0, 2).contiguous().reshape(n_loc, 3)


# ---------------------------------------------------------------------------------------------------------------
# Multiplicative cosets of the 181-bit prime field (BASELINE config 5), sharded by RESIDUE: rank g of N owns
# the evaluations at positions i = g (mod N) of every codeword, stored as the local array a_g[i'] = codeword[g + N i'].  That
# array is the codeword over the coset (shift * gen^g) * <gen^N> of order n / N in natural order, so
#   * the LDE is one ordinary transform per rank over that coset (no exchange),
#   * FRI / Merkle cosets {j + k n/c} (subgroup.tcc:175-197) stay inside one residue class as long as n/c is a multiple of N:
#     the fold is the ordinary fold of the local coset and produces the rank's residue class of f_{i+1},
#   * leaf l of a tree belongs to rank l mod N, so the rank's leaf digests are an interleaved subset of the heap's leaf level:
#     ONE all-to-all of 32-byte digests (1/18 of the bytes of a 12-oracle leaf) hands every rank a contiguous run of leaves,
#     whose sub-tree it builds; N sub-roots are all-gathered as in the additive case.
# ---------------------------------------------------------------------------------------------------------------
def _fp_pow_int(base, e, p):
    return pow(int(base), int(e), int(p))


def local_coset(log_n, gen_int, shift_int, rank, world, modulus):
    """(log2 of the local order, generator, shift) of rank's residue class, as canonical integers."""
    lg = world.bit_length() - 1
    assert (1 << lg) == world and lg <= log_n
    return log_n - lg, _fp_pow_int(gen_int, world, modulus), int(shift_int) * _fp_pow_int(gen_int, rank, modulus) % modulus


def _mont(la, v):
    return la.edwards_to_montgomery([v])[0]


def sharded_mul_lde(lib, torch, la, d_coeffs, n_coeffs, log_n, gen_int, shift_int, rank, world):
    """The rank's residue class of multiplicative_FFT(coeffs, coset(2^log_n, shift)): n_coeffs <= 2^log_n / world."""
    lg_loc, g_loc, s_loc = local_coset(log_n, gen_int, shift_int, rank, world, la.EDWARDS_FR_MODULUS)
    if n_coeffs > (1 << lg_loc):
        raise ValueError("more coefficients than a residue class holds: this transform needs the exchange steps")
    out = torch.empty((1 << lg_loc, 3), dtype=torch.int64, device=d_coeffs.device)
    lib._check(lib.c.iopx_mul_fft_fp3_dev(d_coeffs.data_ptr(), n_coeffs, lg_loc, la._as_u64(_mont(la, g_loc)).ctypes.data_as(la._u64p),
                                          la._as_u64(_mont(la, s_loc)).ctypes.data_as(la._u64p), out.data_ptr()))
    return out


def sharded_mul_fri_fold(lib, torch, la, d_f_local, log_n, gen_int, shift_int, coset_size, x_i, rank, world):
    """multiplicative_evaluate_next_f_i_over_entire_domain on the rank's residue class of f_i (domain of order 2^log_n): returns
    its residue class of f_{i+1}.  Needs 2^log_n / coset_size to be a multiple of world."""
    if ((1 << log_n) // coset_size) % world:
        raise ValueError("domain too small to stay residue-sharded: gather it first (gather_residues)")
    lg_loc, g_loc, s_loc = local_coset(log_n, gen_int, shift_int, rank, world, la.EDWARDS_FR_MODULUS)
    out = torch.empty((d_f_local.shape[0] // coset_size, 3), dtype=torch.int64, device=d_f_local.device)
    x = np.ascontiguousarray(x_i, dtype=np.uint64)
    lib._check(lib.c.iopx_fri_fold_mul_fp3_dev(d_f_local.data_ptr(), lg_loc, la._as_u64(_mont(la, g_loc)).ctypes.data_as(la._u64p),
                                               la._as_u64(_mont(la, s_loc)).ctypes.data_as(la._u64p), coset_size,
                                               x.ctypes.data_as(la._u64p), out.data_ptr()))
    return out


def sharded_mul_merkle_root(lib, torch, dist, la, d_oracles_local, n_local, coset_size, rank, world):
    """Merkle root over residue-sharded multiplicative oracles (every rank passes its local arrays, n_local = n / world).
    Returns (global root, node buffer of the rank's contiguous sub-tree: leaves [rank * L / N, (rank + 1) * L / N))."""
    leaves_loc = n_local // coset_size                  # = L / N
    dev = d_oracles_local[0].device
    nodes = torch.empty((2 * leaves_loc - 1, 32), dtype=torch.uint8, device=dev)
    # local leaf l' is global leaf rank + N l'
    lib.merkle_leaves_dev([o.data_ptr() for o in d_oracles_local], 24, n_local, coset_size, nodes.data_ptr(),
                          domain_type=la.DOMAIN_MULTIPLICATIVE)
    _lib_to_torch(lib, torch)
    if world > 1:
        if leaves_loc % world:
            raise ValueError("fewer leaves per rank than ranks: gather the oracle first (gather_residues)")
        mine = nodes[leaves_loc - 1:].contiguous()      # (L/N, 32): chunk q holds the leaves that fall into rank q's run
        got = torch.empty_like(mine)
        dist.all_to_all_single(got.view(-1), mine.view(-1))
        # chunk s, entry u is global leaf rank * L/N + s + N u  ->  position s + N u of the run
        run = got.reshape(world, leaves_loc // world, 32).permute(1, 0, 2).contiguous().reshape(leaves_loc, 32)
        nodes[leaves_loc - 1:] = run
        _torch_to_lib(lib, torch, nodes)
    lib.merkle_inner_dev(nodes.data_ptr(), leaves_loc)
    _lib_to_torch(lib, torch)
    sub_root = nodes[0].clone()
    if world == 1:
        return bytes(sub_root.cpu().numpy()), nodes
    gathered = [torch.empty_like(sub_root) for _ in range(world)]
    dist.all_gather(gathered, sub_root)
    level = [bytes(g.cpu().numpy()) for g in gathered]
    while len(level) > 1:
        level = [hashlib.blake2b(level[2 * i] + level[2 * i + 1], digest_size=32).digest() for i in range(len(level) // 2)]
    return level[0], nodes


def gather_residues(torch, dist, d_local, world):
    """All residue classes -> the natural-order array on every rank (the tail of FRI, where a round has fewer cosets than
    ranks, and the host boundary)."""
    if world == 1:
        return d_local
    parts = [torch.empty_like(d_local) for _ in range(world)]
    dist.all_gather(parts, d_local.contiguous())
    return torch.stack(parts, dim=1).reshape(d_local.shape[0] * world, *d_local.shape[1:]).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# FRI commit phase on sharded codewords (the round loop of libiop_amd/fri.py with the sharded primitives above).  Both return
# (roots, final polynomial coefficients as a (k, 3) uint64 array) and are bit-identical to the single-process commit.
# ---------------------------------------------------------------------------------------------------------------
def sharded_fri_commit(lib, torch, dist, d_f_local, basis, shift, localization_parameters, final_degree_bound, rank, world,
                       hashchain=None, domains=None):
    """Additive domains: every rank holds its contiguous block of the codeword.  While a round has at least one coset per
    rank the tree is built from N sub-trees and the fold is local; then the remainder is all-gathered and every rank finishes
    it (identically)."""
    hc = hashchain or host.Blake2bHashchain()
    doms = domains or host.fri_additive_domains(basis, shift, localization_parameters)
    f, w, rk, roots = d_f_local, world, rank, []
    for i, eta in enumerate(localization_parameters):
        b_i, s_i = doms[i]
        cs = 1 << eta
        if w > 1 and f.shape[0] // cs < 2:
            _lib_to_torch(lib, torch)
            parts = [torch.empty_like(f) for _ in range(w)]
            dist.all_gather(parts, f.contiguous())
            f = torch.cat(parts, 0)
            _torch_to_lib(lib, torch, f)
            w, rk = 1, 0
        root, _ = sharded_merkle_root(lib, torch, dist, [f], f.shape[0], cs, rk, w)
        roots.append(root)
        hc.absorb(root)
        hc.absorb(None)
        x = hc.squeeze_gf192(1)[0]
        f = sharded_fri_fold(lib, torch, f, b_i, s_i, cs, x, rk, w)
    if w > 1:
        _lib_to_torch(lib, torch)
        parts = [torch.empty_like(f) for _ in range(w)]
        dist.all_gather(parts, f.contiguous())
        f = torch.cat(parts, 0)
        _torch_to_lib(lib, torch, f)
    b_l, s_l = doms[len(localization_parameters)]
    coeffs = torch.empty_like(f)
    lib.additive_IFFT_dev(f.data_ptr(), b_l, s_l, coeffs.data_ptr())
    _lib_to_torch(lib, torch)
    return roots, coeffs.cpu().numpy().view(np.uint64)[:final_degree_bound].copy()


def sharded_mul_fri_commit(lib, torch, dist, la, fri, d_f_local, log_n, gen_int, shift_int, localization_parameters,
                           final_degree_bound, rank, world, hashchain=None):
    """Multiplicative cosets: every rank holds its residue class.  A round stays sharded while it has at least N^2 cosets (fold
    locality needs a multiple of N, the digest all-to-all a multiple of N per rank)."""
    P = la.EDWARDS_FR_MODULUS
    hc = hashchain or host.Blake2bHashchain()
    f, logn, sh, gi, w, rk, roots = d_f_local, log_n, int(shift_int), int(gen_int), world, rank, []
    for eta in localization_parameters:
        cs = 1 << eta
        if w > 1 and ((1 << logn) // cs) % (w * w):
            _lib_to_torch(lib, torch)
            f = gather_residues(torch, dist, f, w)
            _torch_to_lib(lib, torch, f)
            w, rk = 1, 0
        root, _ = sharded_mul_merkle_root(lib, torch, dist, la, [f], f.shape[0], cs, rk, w)
        roots.append(root)
        hc.absorb(root)
        hc.absorb(None)
        x = fri.squeeze_edwards_fr(hc)
        f = sharded_mul_fri_fold(lib, torch, la, f, logn, gi, sh, cs, x, rk, w)
        logn, sh, gi = logn - eta, pow(sh, cs, P), pow(gi, cs, P)
    if w > 1:
        _lib_to_torch(lib, torch)
        f = gather_residues(torch, dist, f, w)
        _torch_to_lib(lib, torch, f)
    coeffs = torch.empty_like(f)
    lib._check(lib.c.iopx_mul_ifft_fp3_dev(f.data_ptr(), logn, la._as_u64(_mont(la, gi)).ctypes.data_as(la._u64p),
                                           la._as_u64(_mont(la, sh)).ctypes.data_as(la._u64p), coeffs.data_ptr()))
    _lib_to_torch(lib, torch)
    return roots, coeffs.cpu().numpy().view(np.uint64)[:final_degree_bound].copy()


# ---------------------------------------------------------------------------------------------------------------
# The Aurora prover sharded over N = 2^r GPUs by contiguous cosets (BASELINE config 4).
#
# Every vector over the codeword domain L and over the FRI domains L^(i) is block-distributed: rank g holds positions
# [g |D| / N, (g + 1) |D| / N), which is the affine sub-domain span(basis[0 .. m - r)) + element_by_index(g 2^(m-r)).  Hence
#   * a low-degree extension is the rank's coset range of the transform (no exchange; phase 1 on the <= 2^20 coefficients
#     is replicated),
#   * virtual oracles, the LDT combination and the FRI fold are pointwise / coset-local on the sub-domain,
#   * a Merkle tree is N sub-trees whose roots are all-gathered (N x 32 bytes) and finished on the host,
#   * the sumcheck's known-degree IFFT needs the first 2^(log|H| + 1) evaluations only: they sit on rank 0, which interpolates
#     and divides, and h's coefficients are broadcast (|H| x 24 bytes — the one codeword-derived exchange of a proof),
#   * once an FRI domain has fewer than MIN_BLOCK elements per rank it is all-gathered and every rank finishes it,
#   * queried values and authentication paths are collected from their owners at the end (a few hundred KB).
# Everything over the small domains (constraint / variable / summation / input: <= 2^20 elements), the hashchain and the
# proof of work are replicated: every rank derives the same challenges and the same transcript.
# ---------------------------------------------------------------------------------------------------------------
from .domains import DeviceOps, Domain, MerkleTree, ADDITIVE

MIN_BLOCK = 64          # elements per rank below which an FRI domain is gathered


class AuroraShard:
    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world
        self.r = world.bit_length() - 1
        if (1 << self.r) != world:
            raise ValueError("world size must be a power of two")


class ShardedMerkleTree:
    """Global tree = top r levels over N device-resident sub-trees.  The N sub-roots are all-gathered as 32-byte device tensors (one
    collective on the stream) and the top log2 N levels are hashed on the device (iopx_merkle_inner_blake2b_dev over the N sub-roots
    as leaves); only the root is read back, as for a single-GPU tree."""

    def __init__(self, ops, sub_tree, num_leaves_global):
        self.ops, self.sub, self.L = ops, sub_tree, num_leaves_global
        sh, torch, lib = ops.shard, ops.torch, ops.lib
        W = sh.world
        _lib_to_torch(lib, torch)
        sub_root = sub_tree.nodes[0].contiguous()
        self.top_nodes = torch.zeros((2 * W - 1, 32), dtype=torch.uint8, device=sub_root.device)     # heap order: level r = the sub-roots
        if W == 1:
            self.top_nodes[0].copy_(sub_root)
        else:
            parts = [torch.empty_like(sub_root) for _ in range(W)]
            sh.dist.all_gather(parts, sub_root)
            self.top_nodes[W - 1:].copy_(torch.stack(parts))
            _torch_to_lib(lib, torch, self.top_nodes)
            lib.merkle_inner_dev(self.top_nodes.data_ptr(), W)
        self._top_host = None

    def root(self):
        return self.ops.lib.read_digest(self.top_nodes.data_ptr())

    def _top_table(self):
        if self._top_host is None:
            _lib_to_torch(self.ops.lib, self.ops.torch)
            self._top_host = self.top_nodes.cpu().numpy()
        return self._top_host

    def membership_proof(self, leaf_positions):
        """merkle_tree::get_set_membership_proof (merkle_tree.tcc:242-336) over the distributed tree: the index walk on the host (the
        same on every rank); every rank writes the auxiliary nodes it owns into a zeroed (count, 32) device tensor, one all-reduce
        (sum: exactly one owner per row) completes it everywhere; the nodes of the top levels come from the replicated top table."""
        sh, torch = self.ops.shard, self.ops.torch
        idx = membership_proof_node_indices(self.L, leaf_positions)
        if not idx:
            return np.zeros((0, 32), dtype=np.uint8)
        r = sh.r
        rows, local_nodes, top_rows = [], [], []
        for row, node in enumerate(idx):
            depth = (node + 1).bit_length() - 1
            if depth <= r:
                top_rows.append((row, node))
                continue
            j = node - ((1 << depth) - 1)
            owner, loc_depth = j >> (depth - r), depth - r
            if owner == sh.rank:
                rows.append(row)
                local_nodes.append((1 << loc_depth) - 1 + (j & ((1 << loc_depth) - 1)))
        buf = torch.zeros((len(idx), 32), dtype=torch.uint8, device=self.sub.nodes.device)
        if rows:
            sel = self.ops.upload_raw(np.array(local_nodes, dtype=np.int64), torch.int64)
            dst = self.ops.upload_raw(np.array(rows, dtype=np.int64), torch.int64)
            _lib_to_torch(self.ops.lib, torch)
            buf[dst] = self.sub.nodes[sel]
        if sh.world > 1:
            sh.dist.all_reduce(buf)
        out = buf.cpu().numpy()
        if top_rows:
            top = self._top_table()
            for row, node in top_rows:
                out[row] = top[node]
        return out


def membership_proof_node_indices(num_leaves, positions):
    """Heap indices of the auxiliary hashes of merkle_tree::get_set_membership_proof, in the reference's order
    (merkle_tree.tcc:256-336): level by level from the leaves, a left node whose right sibling is not queried takes the
    sibling, a right node takes its left sibling."""
    out = []
    S = sorted(set(int(p) for p in positions))
    if not S:
        return out
    for p in S:
        if p >= num_leaves:
            raise ValueError("All positions must be between 0 and num_leaves-1.")
    S = [p + num_leaves - 1 for p in S]
    while not (len(S) == 1 and S[0] == 0):
        new_S, i = [], 0
        while i < len(S):
            pos, nxt = S[i], i + 1
            new_S.append((pos - 1) // 2)
            if pos % 2 == 0:
                out.append(pos - 1)
            elif nxt == len(S) or S[nxt] != pos + 1:
                out.append(pos + 1)
            else:
                nxt += 1
            i = nxt
        S = new_S
    return out


class ShardedDeviceOps(DeviceOps):
    """DeviceOps with the codeword-domain vectors block-distributed over the ranks of `shard` (additive domains)."""

    def __init__(self, lib, torch, device, field, shard):
        super().__init__(lib, torch, device, field)
        if not field.additive:
            raise ValueError("contiguous-coset sharding is for affine subspaces (multiplicative cosets shard by residue class)")
        self.shard = shard

    # ---- layout ----
    def _is_sharded(self, domain):
        return getattr(domain, "sharded", False)

    def local_size(self, domain):
        return domain.size // self.shard.world if self._is_sharded(domain) else domain.size

    def local_domain(self, domain):
        """The rank's block as an affine subspace (local_subdomain above)."""
        if not self._is_sharded(domain):
            return domain
        b, s = local_subdomain(domain.basis, domain.shift, self.shard.rank, self.shard.world)
        return Domain(self.field, ADDITIVE, basis=b, shift=s)

    def mark_codeword_domain(self, domain):
        if domain.size // self.shard.world < MIN_BLOCK:
            raise ValueError("codeword domain too small for %d ranks" % self.shard.world)
        domain.sharded = True
        return domain

    def mark_fri_domains(self, domains, localization):
        """L^(i+1) stays distributed while it keeps MIN_BLOCK elements per rank (folds are local: cosets are contiguous)."""
        for i in range(1, len(domains)):
            # the last domain carries no oracle: the final polynomial is interpolated from it on every rank, so it is gathered
            domains[i].sharded = (self._is_sharded(domains[i - 1]) and i < len(localization)
                                  and domains[i].size // self.shard.world >= MIN_BLOCK)
        return domains

    def _all_gather(self, d_local):
        parts = [self.torch.empty_like(d_local) for _ in range(self.shard.world)]
        self.shard.dist.all_gather(parts, d_local.contiguous())
        return self.torch.cat(parts, 0)

    # ---- transforms ----
    def FFT(self, d_coeffs, n_coeffs, domain):
        if not self._is_sharded(domain):
            return super().FFT(d_coeffs, n_coeffs, domain)
        return self.FFT_batch([d_coeffs], n_coeffs, domain)[0]

    def FFT_batch(self, d_coeffs_list, n_coeffs, domain):
        if not self._is_sharded(domain):
            return super().FFT_batch(d_coeffs_list, n_coeffs, domain)
        return sharded_lde_batch(self.lib, self.torch, d_coeffs_list, int(n_coeffs), domain.basis, domain.shift, self.shard.rank, self.shard.world)

    def _coset_range(self, codeword_domain, d):
        if not self._is_sharded(codeword_domain):
            return super()._coset_range(codeword_domain, d)
        cosets = 1 << (codeword_domain.dim - d)
        if cosets % self.shard.world:
            raise ValueError("fewer cosets than ranks")
        per = cosets // self.shard.world
        return self.shard.rank * per, per

    def IFFT(self, d_evals, domain):
        if self._is_sharded(domain):
            raise ValueError("inverse transform of a distributed vector: gather it first")
        return super().IFFT(d_evals, domain)

    def IFFT_of_known_degree(self, d_evals, degree, domain):
        """fft.tcc:458-475 needs the first 2^ceil(log2 degree) evaluations: rank 0's head.  Rank 0 interpolates, the
        coefficients are broadcast."""
        if not self._is_sharded(domain):
            return super().IFFT_of_known_degree(d_evals, degree, domain)
        k = max(int(degree) - 1, 0).bit_length()
        if (1 << k) > self.local_size(domain):
            full = self._all_gather(d_evals)
            return super().IFFT(full[: 1 << k], domain.get_subset_of_order(1 << k))
        if self.shard.rank == 0:
            out = super().IFFT(d_evals[: 1 << k], domain.get_subset_of_order(1 << k))
        else:
            out = self.empty(1 << k)
        self.shard.dist.broadcast(out, src=0)
        return out

    # ---- FRI / Merkle ----
    def fold(self, d_f, domain, coset_size, x_i, next_domain=None):
        if not self._is_sharded(domain):
            return super().fold(d_f, domain, coset_size, x_i)
        nxt = super().fold(d_f, self.local_domain(domain), coset_size, x_i)
        if next_domain is not None and self._is_sharded(next_domain):
            return nxt
        return self._all_gather(nxt)

    def merkle_tree(self, d_oracles, domain, coset_size):
        if not self._is_sharded(domain):
            return super().merkle_tree(d_oracles, domain, coset_size)
        loc = self.local_domain(domain)
        return ShardedMerkleTree(self, super().merkle_tree(d_oracles, loc, coset_size), domain.size // coset_size)

    def query_responses(self, d_oracles, domain, positions):
        if not self._is_sharded(domain):
            return super().query_responses(d_oracles, domain, positions)
        block = self.local_size(domain)
        lo = self.shard.rank * block
        owned = [(row, p - lo) for row, p in enumerate(positions) if lo <= p < lo + block]
        return _collect_query_responses(self, d_oracles, positions, owned)

    def solve_pow(self, challenge, pow_bitlen):
        return _sharded_solve_pow(self, challenge, pow_bitlen)

    # ---- pointwise operators: the rank's block is the sub-domain ----
    def rowcheck(self, d_az, d_bz, d_cz, codeword_domain, constraint_domain):
        return super().rowcheck(d_az, d_bz, d_cz, self.local_domain(codeword_domain), constraint_domain)

    def fz(self, d_fw, d_f1v, codeword_domain, input_domain):
        return super().fz(d_fw, d_f1v, self.local_domain(codeword_domain), input_domain)

    def sumcheck_g(self, d_f, d_h, codeword_domain, summation_domain, claimed_sum):
        return super().sumcheck_g(d_f, d_h, self.local_domain(codeword_domain), summation_domain, claimed_sum)

    def ldt_combine(self, d_oracles, degrees, random_coefficients, domain):
        return super().ldt_combine(d_oracles, degrees, random_coefficients, self.local_domain(domain))

    # the holographic (Fractal) prover's domain-dependent steps; its elementwise ones (div, lincomb_affine, rational_combine,
    # lincheck) act on whatever block they are handed
    def domain_offsets(self, domain, point):
        return super().domain_offsets(self.local_domain(domain), point)

    def vanishing_evals(self, vanishing_domain, domain, constant):
        return super().vanishing_evals(vanishing_domain, self.local_domain(domain), constant)

    def rational_sumcheck_constraint(self, d_p, d_N, d_D, codeword_domain, summation_domain, claimed_sum):
        return super().rational_sumcheck_constraint(d_p, d_N, d_D, self.local_domain(codeword_domain), summation_domain, claimed_sum)


class ResidueShardedDeviceOps(DeviceOps):
    """DeviceOps with the codeword-domain vectors distributed over the ranks of `shard` by RESIDUE CLASS (multiplicative cosets): rank
    r of N holds the positions p = r (mod N), at local index p // N — the sub-coset (shift g^r) <g^N>, itself a multiplicative coset, so
    every per-domain operator runs unchanged on the rank's sub-coset.  A Merkle leaf (the coset {j + k n / 2^eta}) lies on rank j mod N
    whole; leaf digests are exchanged once per tree (all-to-all) so that each rank builds a contiguous sub-tree.  FRI cosets are local."""

    def __init__(self, lib, torch, device, field, shard):
        super().__init__(lib, torch, device, field)
        if field.additive:
            raise ValueError("residue-class sharding is for multiplicative cosets (affine subspaces shard by contiguous cosets)")
        self.shard = shard

    # ---- layout ----
    def _is_sharded(self, domain):
        return getattr(domain, "sharded", False)

    def local_size(self, domain):
        return domain.size // self.shard.world if self._is_sharded(domain) else domain.size

    def local_domain(self, domain, rank=None):
        """The rank's sub-coset: order |domain| / N (its default generator is g^N), shift * g^rank."""
        if not self._is_sharded(domain):
            return domain
        f = self.field
        r = self.shard.rank if rank is None else rank
        shift = domain.shift_int * pow(f.to_int(domain.gen), r, f.P) % f.P
        return Domain(f, "multiplicative_coset", shift=shift, log_n=domain.dim - self.shard.r)

    def _can_shard(self, size, coset_size):
        """Enough elements per rank, and at least N leaves per rank so that the digest exchange splits evenly."""
        W = self.shard.world
        return size // W >= MIN_BLOCK and (size // coset_size) % (W * W) == 0

    def mark_codeword_domain(self, domain, coset_size=2):
        if not self._can_shard(domain.size, coset_size):
            raise ValueError("codeword domain too small for %d ranks" % self.shard.world)
        domain.sharded = True
        return domain

    def mark_fri_domains(self, domains, localization):
        """L^(i+1) stays distributed while its own Merkle tree (cosets of 2^eta_(i+1)) can be exchanged evenly."""
        for i in range(1, len(domains)):
            cs = 1 << localization[i] if i < len(localization) else 1
            domains[i].sharded = self._is_sharded(domains[i - 1]) and i < len(localization) and self._can_shard(domains[i].size, cs)
        return domains

    def _gather(self, d_local):
        return gather_residues(self.torch, self.shard.dist, d_local, self.shard.world)

    # ---- transforms ----
    def FFT(self, d_coeffs, n_coeffs, domain):
        if not self._is_sharded(domain):
            return super().FFT(d_coeffs, n_coeffs, domain)
        if int(n_coeffs) > self.local_size(domain):
            raise ValueError("more coefficients than a residue class holds")
        return super().FFT(d_coeffs, n_coeffs, self.local_domain(domain))

    def IFFT(self, d_evals, domain):
        if self._is_sharded(domain):
            raise ValueError("inverse transform of a distributed vector: gather it first")
        return super().IFFT(d_evals, domain)

    def IFFT_of_known_degree(self, d_evals, degree, domain):
        """fft.tcc:435-456 reads every (|domain| / 2^k)-th evaluation: while that stride is a multiple of N they are all rank 0's, at
        stride / N in its sub-coset (whose shift is the domain's) — rank 0 interpolates, the coefficients are broadcast."""
        if not self._is_sharded(domain):
            return super().IFFT_of_known_degree(d_evals, degree, domain)
        k = max(int(degree) - 1, 0).bit_length()
        if (domain.size >> k) % self.shard.world:
            return super().IFFT_of_known_degree(self._gather(d_evals), degree, domain)
        if self.shard.rank == 0:
            out = super().IFFT_of_known_degree(d_evals, degree, self.local_domain(domain, 0))
        else:
            out = self.empty(1 << k)
        self.shard.dist.broadcast(out, src=0)
        return out

    # ---- FRI / Merkle ----
    def fold(self, d_f, domain, coset_size, x_i, next_domain=None):
        if not self._is_sharded(domain):
            return super().fold(d_f, domain, coset_size, x_i)
        nxt = super().fold(d_f, self.local_domain(domain), coset_size, x_i)
        if next_domain is not None and self._is_sharded(next_domain):
            return nxt
        return self._gather(nxt)

    def merkle_tree(self, d_oracles, domain, coset_size):
        if not self._is_sharded(domain):
            return super().merkle_tree(d_oracles, domain, coset_size)
        sh, torch = self.shard, self.torch
        W = sh.world
        n_local = self.local_size(domain)
        leaves_loc = n_local // coset_size
        if leaves_loc % W:
            raise ValueError("fewer leaves per rank than ranks")
        nodes = torch.empty((2 * leaves_loc - 1, 32), dtype=torch.uint8, device=self.device)
        # local leaf l' is global leaf rank + N l' (the sub-coset's own coset structure): digests first, then the exchange
        self.lib.merkle_leaves_dev([t.data_ptr() for t in d_oracles], 24, n_local, coset_size, nodes.data_ptr(), domain_type=domain.domain_type)
        _lib_to_torch(self.lib, torch)
        mine = nodes[leaves_loc - 1:].contiguous()              # chunk q holds the leaves that fall into rank q's contiguous run
        got = torch.empty_like(mine)
        sh.dist.all_to_all_single(got.view(-1), mine.view(-1))
        nodes[leaves_loc - 1:] = got.reshape(W, leaves_loc // W, 32).permute(1, 0, 2).contiguous().reshape(leaves_loc, 32)
        _torch_to_lib(self.lib, torch, nodes)
        self.lib.merkle_inner_dev(nodes.data_ptr(), leaves_loc)
        return ShardedMerkleTree(self, MerkleTree(self.lib, nodes, leaves_loc), domain.size // coset_size)

    def query_responses(self, d_oracles, domain, positions):
        if not self._is_sharded(domain):
            return super().query_responses(d_oracles, domain, positions)
        W, rank = self.shard.world, self.shard.rank
        owned = [(row, p // W) for row, p in enumerate(positions) if p % W == rank]
        return _collect_query_responses(self, d_oracles, positions, owned)

    def solve_pow(self, challenge, pow_bitlen):
        return _sharded_solve_pow(self, challenge, pow_bitlen)

    # ---- pointwise operators: the rank's residue class is a coset in its own right ----
    def rowcheck(self, d_az, d_bz, d_cz, codeword_domain, constraint_domain):
        return super().rowcheck(d_az, d_bz, d_cz, self.local_domain(codeword_domain), constraint_domain)

    def fz(self, d_fw, d_f1v, codeword_domain, input_domain):
        return super().fz(d_fw, d_f1v, self.local_domain(codeword_domain), input_domain)

    def sumcheck_g(self, d_f, d_h, codeword_domain, summation_domain, claimed_sum):
        return super().sumcheck_g(d_f, d_h, self.local_domain(codeword_domain), summation_domain, claimed_sum)

    def ldt_combine(self, d_oracles, degrees, random_coefficients, domain):
        return super().ldt_combine(d_oracles, degrees, random_coefficients, self.local_domain(domain))

    def domain_offsets(self, domain, point):
        return super().domain_offsets(self.local_domain(domain), point)

    def vanishing_evals(self, vanishing_domain, domain, constant):
        return super().vanishing_evals(vanishing_domain, self.local_domain(domain), constant)

    def rational_sumcheck_constraint(self, d_p, d_N, d_D, codeword_domain, summation_domain, claimed_sum):
        return super().rational_sumcheck_constraint(d_p, d_N, d_D, self.local_domain(codeword_domain), summation_domain, claimed_sum)


def _collect_query_responses(ops, d_oracles, positions, owned):
    """values[p][k] = oracle_k[positions[p]] for oracles distributed over the ranks: every rank writes the rows it owns (row, local
    index) into a zeroed (positions, oracles, 3) device tensor, one all-reduce (sum: one owner per row) completes it everywhere."""
    torch = ops.torch
    if not positions:
        return np.zeros((0, len(d_oracles), 3), dtype=np.uint64)
    buf = torch.zeros((len(positions), len(d_oracles), 3), dtype=torch.int64, device=d_oracles[0].device)
    if owned:
        rows = ops.upload_raw(np.array([r for r, _ in owned], dtype=np.int64), torch.int64)
        loc = ops.upload_raw(np.array([l for _, l in owned], dtype=np.int64), torch.int64)
        _lib_to_torch(ops.lib, torch)
        buf[rows] = torch.stack([t[loc] for t in d_oracles], dim=1)
    if ops.shard.world > 1:
        ops.shard.dist.all_reduce(buf)
    return buf.cpu().numpy().view(np.uint64)


def _sharded_solve_pow(ops, challenge, pow_bitlen):
    """pow::solve_pow (bcs/pow.tcc:67-103) split by candidate range: in super-batch s rank r searches candidates
    [(s N + r) B_s, (s N + r + 1) B_s); a min all-reduce of the hits ends the search at the first super-batch that has one, and the
    minimum is the reference's first hit (the candidates of earlier super-batches all failed)."""
    sh, torch, lib = ops.shard, ops.torch, ops.lib
    if sh.world == 1:
        return lib.solve_pow(challenge, pow_bitlen)
    none = (1 << 62)
    first, batch = 0, 1 << 14
    dev = ops.device
    while True:
        hit = lib.pow_search(challenge, pow_bitlen, first + sh.rank * batch, batch)
        t = torch.tensor([none if hit is None else hit], dtype=torch.int64, device=dev)
        sh.dist.all_reduce(t, op=sh.dist.ReduceOp.MIN)
        best = int(t.item())
        if best != none:
            return lib.pow_candidate(challenge, best)
        first += sh.world * batch
        if batch < (1 << 22):
            batch <<= 2


def sharded_ops(lib, torch, device, field, shard):
    """The sharded operator set of a field: contiguous cosets for affine subspaces, residue classes for multiplicative cosets."""
    return (ShardedDeviceOps if field.additive else ResidueShardedDeviceOps)(lib, torch, device, field, shard)


def sharded_aurora_snark_prover(ops, constraint_system, primary_input, parameters, d_assignment, auxiliary_input=None, round_hook=None):
    """aurora_snark_prover with `ops` a ShardedDeviceOps: every rank calls it with the same (replicated) instance and witness
    and returns the same transcript, byte-identical to the single-GPU prover's."""
    from . import aurora
    return aurora.aurora_snark_prover(ops, constraint_system, primary_input, auxiliary_input, parameters, round_hook=round_hook,
                                      d_assignment=d_assignment)


def sharded_fractal_snark_indexer(ops, constraint_system, parameters):
    """fractal_snark_indexer with `ops` a ShardedDeviceOps: each rank holds its block of the twelve index oracles and the sub-tree over
    it (the root is assembled from the ranks' sub-roots); the evaluations over the index domain are replicated."""
    from . import fractal
    return fractal.fractal_snark_indexer(ops, constraint_system, parameters)


def sharded_fractal_snark_prover(ops, index, constraint_system, primary_input, parameters, d_assignment, auxiliary_input=None, round_hook=None):
    """fractal_snark_prover on block-distributed oracles: every rank calls it with the same (replicated) instance and witness and its
    own part of the index, and returns the same transcript, byte-identical to the single-GPU prover's."""
    from . import fractal
    return fractal.fractal_snark_prover(ops, index, constraint_system, primary_input, auxiliary_input, parameters, round_hook=round_hook,
                                        d_assignment=d_assignment)
