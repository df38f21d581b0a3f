"""BCS round driver on device-resident oracles: the prover side of the reference's IOP bookkeeping and BCS transformation.

    libiop/iop/iop.tcc                   registration state machine (domains, oracles, virtual oracles, messages, rounds, queries)
    libiop/bcs/bcs_common.tcc:399-696    one Merkle tree per (round, domain), round parameters, hashchain per round
    libiop/bcs/bcs_prover.tcc            signal_prover_round_done, proof of work, get_transcript

Oracles are device tensors; the only host traffic is 32-byte roots, O(log n) challenges and, at the end, the queried
values and authentication paths.  Protocols (libiop_amd/aurora.py, the FRI-only protocol in libiop_amd/fri.py) register
against this object exactly as the reference's protocols register against bcs_prover.  Non-zk, BLAKE2b digests.
Citations are relative to the reference tree."""
import hashlib

import numpy as np

from . import host


class OracleHandle:
    __slots__ = ("id", "virtual")

    def __init__(self, id, virtual=False):
        self.id, self.virtual = id, virtual


class VirtualOracle:
    """virtual_oracle<FieldT> (iop/oracles.hpp): prover side needs evaluated_contents only."""

    def evaluated_contents(self, constituents):
        raise NotImplementedError


class PositionHandle:
    __slots__ = ("id", "random")

    def __init__(self, id, random):
        self.id, self.random = id, random


class Transcript:
    """bcs_transformation_transcript (bcs/bcs_common.hpp:36-106)."""

    def __init__(self):
        self.prover_messages = []           # (len, 3) uint64 each
        self.MT_roots = []                  # 32 bytes each
        self.query_positions = []           # per tree: sorted positions
        self.MT_leaf_positions = []         # per tree: sorted leaf indices
        self.query_responses = []           # per tree: (positions, oracles, 3) uint64
        self.MT_set_membership_proofs = []  # per tree: (count, 32) uint8 auxiliary hashes
        self.proof_of_work = b""

    def serialize(self):
        """The canonical byte form the parity tests compare (counts and positions as 8-byte little-endian integers, elements and
        digests raw, in the field order of this class; the reference's own serialisation is text, bcs_common.tcc:96-390)."""
        out = bytearray()
        u64 = lambda v: out.extend(int(v).to_bytes(8, "little"))
        u64(len(self.prover_messages))
        for m in self.prover_messages:
            u64(len(m))
            out.extend(np.ascontiguousarray(m, dtype=np.uint64).tobytes())
        u64(len(self.MT_roots))
        for r in self.MT_roots:
            out.extend(r)
        for t in range(len(self.query_positions)):      # one entry per tree: a holographic transcript has more trees than roots
            u64(len(self.query_positions[t]))
            for p in self.query_positions[t]:
                u64(p)
            u64(len(self.MT_leaf_positions[t]))
            for p in self.MT_leaf_positions[t]:
                u64(p)
            resp = self.query_responses[t]
            u64(resp.shape[1] if resp.shape[0] else 0)
            out.extend(np.ascontiguousarray(resp, dtype=np.uint64).tobytes())
            aux = self.MT_set_membership_proofs[t]
            u64(len(aux))
            out.extend(np.ascontiguousarray(aux, dtype=np.uint8).tobytes())
        out.extend(self.proof_of_work)
        return bytes(out)


class ProverIndex:
    """bcs_prover_index (bcs/bcs_common.hpp): the index round's oracles and their Merkle trees, device resident, plus whatever the
    protocol's indexer wants the prover to find next to them (iop_prover_index::all_oracle_evals_ and, here, `extra`)."""

    def __init__(self, oracles, trees, roots, prover_messages, extra=None):
        self.oracles, self.trees, self.roots, self.prover_messages, self.extra = oracles, trees, roots, prover_messages, extra


class BCSProver:
    """bcs_prover<FieldT, binary_hash_digest>; without interactions registered after the index round it is the bcs_indexer
    (bcs/bcs_indexer.tcc): signal_index_submissions_done then Merkleises the submitted index oracles."""

    def __init__(self, ops, pow_work_parameter, index=None):
        self.ops, self.lib, self.field = ops, ops.lib, ops.field
        self.index, self.is_holographic = index, False            # bcs_prover.tcc:12-21
        self.pow_bitlen = pow_work_parameter            # pow_parameters(dim_h + 3, cost 1).pow_bitlen() (pow.tcc:21-32)
        self.hashchain = host.Blake2bHashchain()
        # registrations
        self.domains = []
        self.oracle_regs = []                            # (domain, degree, make_zk)
        self.virtual_regs = []                           # (domain, degree, constituents, contents, cache)
        self.prover_message_sizes, self.verifier_message_sizes = [], []
        self.num_oracles_at_end_of_round, self.num_prover_messages_at_end_of_round = [], []
        self.num_verifier_messages_at_end_of_round = []
        self.round_params = []
        self.from_prover, self.sealed = False, False
        self.num_interaction_rounds = 0
        self.random_position_domains, self.deterministic_positions, self.queries = [], [], []
        # run state
        self.oracles, self.prover_messages = [], []
        self.num_prover_rounds_done, self.processed_MTs = 0, 0
        self.MT_trees, self.MT_roots, self.MT_info = [], [], []
        self.verifier_random_messages = {}
        self.virtual_contents_cache = {}
        self.pow_answer = None
        self.round_hooks = []                            # timing / tracing callbacks: fn(round index)

    # ---- registration (iop.tcc:22-263) ----
    def register_domain(self, domain):
        self.domains.append(domain)
        return len(self.domains) - 1

    def get_domain(self, handle):
        return self.domains[handle]

    def _assert_can_register(self, domain, degree):
        if self.sealed:
            raise AssertionError("attempted to register an oracle after interactive registrations sealed")
        if domain >= len(self.domains):
            raise ValueError("domain not registered")
        if degree >= self.domains[domain].size:
            raise ValueError("attempting to register oracle whose degree exceeds domain size")

    def _update_rounds_and_direction(self, from_prover):
        """iop.tcc:36-63."""
        if self.from_prover == from_prover:
            return
        if self.from_prover:
            self.num_oracles_at_end_of_round.append(len(self.oracle_regs))
            self.num_prover_messages_at_end_of_round.append(len(self.prover_message_sizes))
            self.num_interaction_rounds += 1
        else:
            self.num_verifier_messages_at_end_of_round.append(len(self.verifier_message_sizes))
        self.from_prover = from_prover

    def register_oracle(self, name, domain, degree, make_zk=False):
        self._assert_can_register(domain, degree)
        if make_zk:
            raise ValueError("zero-knowledge oracles are out of scope (salts and masks are not reproducible)")
        self._update_rounds_and_direction(True)
        if self.is_holographic and self.num_interaction_rounds == 0:
            raise ValueError("Cannot register non-index oracles in round 0 of a holographic IOP")
        self.oracle_regs.append((domain, degree, name))
        self.oracles.append(None)
        return OracleHandle(len(self.oracle_regs) - 1)

    def register_index_oracle(self, domain, degree):
        """iop.tcc:106-125."""
        self._assert_can_register(domain, degree)
        if self.num_prover_rounds_done != 0:
            raise ValueError("index oracles must be created in the 0th round")
        self._update_rounds_and_direction(True)
        self.is_holographic = True
        self.oracle_regs.append((domain, degree, "index"))
        self.oracles.append(None)
        return OracleHandle(len(self.oracle_regs) - 1)

    def signal_index_registrations_done(self):
        """iop.tcc:377-386."""
        if not self.is_holographic or self.num_interaction_rounds != 0:
            raise ValueError("Should only be used to end round 0 of a holographic IOP")
        self._update_rounds_and_direction(False)

    def register_virtual_oracle(self, domain, degree, constituents, contents, cache_evaluated_contents=False):
        self._assert_can_register(domain, degree)
        self.virtual_regs.append((domain, degree, list(constituents), contents, cache_evaluated_contents))
        return OracleHandle(len(self.virtual_regs) - 1, virtual=True)

    def register_prover_message(self, size):
        self._update_rounds_and_direction(True)
        self.prover_message_sizes.append(size)
        self.prover_messages.append(None)
        return len(self.prover_message_sizes) - 1

    def register_verifier_random_message(self, size):
        self._update_rounds_and_direction(False)
        self.verifier_message_sizes.append(size)
        return len(self.verifier_message_sizes) - 1

    def set_round_parameters(self, quotient_map_domain):
        """bcs_common.tcc:482-495; round_parameters(domain): the Merkle leaves of the current round hold cosets of |domain|."""
        cur_round = self.num_interaction_rounds
        if self.round_params and cur_round == len(self.round_params) - 1:
            raise AssertionError("Already set round parameters for this round")
        while len(self.round_params) < cur_round:
            self.round_params.append(1)
        self.round_params.append(quotient_map_domain.size)

    def get_round_parameters(self, round):
        return self.round_params[round] if round < len(self.round_params) else 1

    def get_oracle_domain(self, handle):
        return self.virtual_regs[handle.id][0] if handle.virtual else self.oracle_regs[handle.id][0]

    def get_oracle_degree(self, handle):
        return self.virtual_regs[handle.id][1] if handle.virtual else self.oracle_regs[handle.id][1]

    def seal_interaction_registrations(self):
        """iop.tcc:227-251, bcs_common.tcc:423-480."""
        if not self.from_prover:
            raise AssertionError("attempted to seal interaction registrations where verifier sends the last interactive message")
        self.num_oracles_at_end_of_round.append(len(self.oracle_regs))
        self.num_prover_messages_at_end_of_round.append(len(self.prover_message_sizes))
        self.num_interaction_rounds += 1
        self.sealed = True
        for round in range(self.num_interaction_rounds):
            for dom, ids in self.oracles_in_round_by_domain(round):
                self.MT_info.append((round, dom, ids))
                self.MT_trees.append(None)
                self.MT_roots.append(None)

    def register_random_query_position(self, domain):
        self.random_position_domains.append(domain)
        return PositionHandle(len(self.random_position_domains) - 1, True)

    def register_deterministic_query_position(self, seeds, calculator):
        self.deterministic_positions.append((list(seeds), calculator))
        return PositionHandle(len(self.deterministic_positions) - 1, False)

    def register_query(self, oracle, position):
        self.queries.append((oracle, position))
        return len(self.queries) - 1

    def seal_query_registrations(self):
        pass

    def oracles_in_round_by_domain(self, round):
        """iop.tcc:801-820: domain -> oracle ids of the round, domains in handle order (std::map)."""
        begin = 0 if round == 0 else self.num_oracles_at_end_of_round[round - 1]
        mapping = {}
        for oid in range(begin, self.num_oracles_at_end_of_round[round]):
            mapping.setdefault(self.oracle_regs[oid][0], []).append(oid)
        return sorted(mapping.items())

    # ---- proving (iop.tcc:265-433, bcs_prover.tcc:23-98) ----
    def submit_oracle(self, handle, d_contents):
        oid = handle.id
        if self.oracles[oid] is not None:
            raise ValueError("attempted to submit already submitted oracle")
        begin = 0 if self.num_prover_rounds_done == 0 else self.num_oracles_at_end_of_round[self.num_prover_rounds_done - 1]
        if oid < begin:
            raise ValueError("submitting an oracle for a previous round")
        if oid >= self.num_oracles_at_end_of_round[self.num_prover_rounds_done]:
            raise ValueError("submitting an oracle for a future round (did you forget to call signal_prover_round_done?)")
        if self.ops.local_size(self.domains[self.oracle_regs[oid][0]]) != d_contents.shape[0]:
            raise ValueError("oracle evaluations don't match the domain size")
        self.oracles[oid] = d_contents

    def submit_prover_message(self, handle, contents):
        contents = np.ascontiguousarray(contents, dtype=np.uint64).reshape(-1, 3)
        if self.prover_messages[handle] is not None:
            raise ValueError("attempted to submit already submitted prover message")
        if self.prover_message_sizes[handle] != contents.shape[0]:
            raise ValueError("prover message submission does not match its registered size")
        self.prover_messages[handle] = contents

    def submit_prover_index(self, index):
        """iop_protocol::submit_prover_index (iop.tcc:309-341) + bcs_prover::signal_index_submissions_done (bcs_prover.tcc:68-80):
        round 0's oracles, messages and trees come from the index; only the hashchain runs."""
        if self.num_prover_rounds_done != 0:
            raise ValueError("The IOP prover index should only be for round 0")
        count = self.num_oracles_at_end_of_round[0]
        if len(index.oracles) != count:
            raise ValueError("The IOP prover index provided the wrong number of evaluations")
        for oid in range(count):
            self.submit_oracle(OracleHandle(oid), index.oracles[oid])
        for mid in range(self.num_prover_messages_at_end_of_round[0]):
            self.submit_prover_message(mid, index.prover_messages[mid])
        self.signal_index_submissions_done()

    def signal_index_submissions_done(self):
        """bcs_indexer.tcc:17-53 when no index was given (the trees are built here), bcs_prover.tcc:68-80 otherwise."""
        if self.num_prover_rounds_done != 0:
            raise ValueError("Index submissions should be round 0")
        self._finish_round(build_trees=self.index is None)

    def get_prover_index(self, extra=None):
        """bcs_indexer::get_bcs_prover_index (bcs_indexer.tcc:80-103)."""
        k = len(self.oracles_in_round_by_domain(0))
        count = self.num_oracles_at_end_of_round[0]
        return ProverIndex(self.oracles[:count], self.MT_trees[:k], self.MT_roots[:k],
                           self.prover_messages[: self.num_prover_messages_at_end_of_round[0]], extra)

    def get_verifier_index(self):
        """bcs_indexer::get_verifier_index (bcs_indexer.tcc:67-77): the index trees' roots (and the indexed messages)."""
        k = len(self.oracles_in_round_by_domain(0))
        return list(self.MT_roots[:k]), self.prover_messages[: self.num_prover_messages_at_end_of_round[0]]

    def signal_prover_round_done(self):
        self._finish_round(build_trees=True)

    def _finish_round(self, build_trees):
        if self.num_prover_rounds_done >= self.num_interaction_rounds:
            raise AssertionError("attempting to signal end of a round after protocol already finished")
        ended = self.num_prover_rounds_done
        begin = 0 if ended == 0 else self.num_oracles_at_end_of_round[ended - 1]
        for oid in range(begin, self.num_oracles_at_end_of_round[ended]):
            if self.oracles[oid] is None:
                raise AssertionError("signaling end of round without submitting all oracles in the round")
        mbegin = 0 if ended == 0 else self.num_prover_messages_at_end_of_round[ended - 1]
        for mid in range(mbegin, self.num_prover_messages_at_end_of_round[ended]):
            if self.prover_messages[mid] is None:
                raise AssertionError("signaling end of round without submitting all prover messages in the round")
        self.num_prover_rounds_done += 1
        # one tree per (round, domain) over every oracle of that domain, leaves serialised by cosets (bcs_prover.tcc:36-47).
        # Quirk F11: the reference indexes Merkle_trees_[processed_MTs_] for each domain of the round, so two
        # oracle domains in one round would double-construct a tree; every shipped protocol has one.
        mapping = self.oracles_in_round_by_domain(ended)
        if len(mapping) > 1:
            raise AssertionError("more than one oracle domain in a round (bcs_prover.tcc:36-47)")
        cs = self.get_round_parameters(ended)
        roots = []
        for dom, ids in mapping:
            if build_trees:
                tree = self.ops.merkle_tree([self.oracles[i] for i in ids], self.domains[dom], cs)
                root = tree.root()                                      # merkle_tree::get_root
            else:                                                       # "The Merkle trees are already filled in by the preprocessor."
                tree, root = self.index.trees[self.processed_MTs], self.index.roots[self.processed_MTs]
            self.MT_trees[self.processed_MTs] = tree
            self.MT_roots[self.processed_MTs] = root
            roots.append(root)
            self.processed_MTs += 1
        self._run_hashchain_for_round(ended, roots)
        # bcs_prover.tcc:52-59; the indexer's one-round protocol registers no proof of work (bcs_common.tcc:426-431)
        if self.num_prover_rounds_done == self.num_interaction_rounds and not (self.is_holographic and self.num_interaction_rounds == 1):
            challenge = self._squeeze_root_type()
            self.pow_answer = self.ops.solve_pow(challenge, self.pow_bitlen)
        for hook in self.round_hooks:
            hook(ended)

    def _run_hashchain_for_round(self, round, roots):
        """bcs_common.tcc:550-614.  absorb() advances the state without reading its input (quirk F8), so neither the roots nor
        the message bytes need to be on the host for the challenges; the root is read back because it goes into the transcript."""
        for r in roots:
            self.hashchain.absorb(r)
        self.hashchain.absorb(None)                                   # the round's prover messages (their hash is ignored too)
        start = self.num_verifier_messages_at_end_of_round[round]
        end = 0 if round == self.num_interaction_rounds - 1 else self.num_verifier_messages_at_end_of_round[round + 1]
        for i in range(start, end):
            self.verifier_random_messages[i] = self.field.squeeze(self.hashchain, self.verifier_message_sizes[i])

    def _squeeze_root_type(self):
        """blake2b_hashchain::squeeze_root_type (blake2b.tcc:105-110)."""
        x = self.field.squeeze(self.hashchain, 1)
        return hashlib.blake2b(x.tobytes(), digest_size=32).digest()

    def obtain_verifier_random_message(self, handle):
        if handle not in self.verifier_random_messages:
            raise AssertionError("verifier random message not available yet (did you forget to call signal_prover_round_done?)")
        return self.verifier_random_messages[handle]

    def get_oracle_evaluations(self, handle):
        """iop.tcc:630-667: device tensor of the oracle's evaluations; virtual oracles are evaluated from their constituents."""
        if not handle.virtual:
            return self.oracles[handle.id]
        if handle.id in self.virtual_contents_cache:
            return self.virtual_contents_cache[handle.id]
        _, _, constituents, contents, cache = self.virtual_regs[handle.id]
        result = contents.evaluated_contents([self.get_oracle_evaluations(c) for c in constituents])
        if cache:
            self.virtual_contents_cache[handle.id] = result
        return result

    def release(self):
        """Drops every device buffer this prover holds (oracles, cached virtual-oracle contents, Merkle trees) and the registered
        virtual oracles and position calculators, so the memory returns to the allocator when the prover function returns and not
        when a cyclic garbage collection happens to run.  The prover index is not touched: it belongs to the caller."""
        self.oracles, self.virtual_contents_cache, self.MT_trees = [], {}, []
        self.virtual_regs, self.deterministic_positions, self.queries, self.round_hooks = [], [], [], []
        self.index = None

    # ---- transcript (bcs_prover.tcc:136-233) ----
    def _obtain_query_position(self, handle, random_cache, det_cache):
        if handle.random:
            if handle.id not in random_cache:
                n = self.domains[self.random_position_domains[handle.id]].size
                random_cache[handle.id] = self.hashchain.squeeze_query_positions(1, n)[0]      # bcs_common.tcc:536-548
            return random_cache[handle.id]
        if handle.id not in det_cache:
            seeds, calc = self.deterministic_positions[handle.id]
            det_cache[handle.id] = calc([self._obtain_query_position(s, random_cache, det_cache) for s in seeds])
        return det_cache[handle.id]

    def _record(self, handle, position, positions_by_oracle):
        """get_oracle_evaluation_at_point with record = true (iop.tcc:669-714): a query to a virtual oracle touches the same
        position of each of its constituents."""
        if not handle.virtual:
            positions_by_oracle.setdefault(handle.id, set()).add(position)
            return
        for c in self.virtual_regs[handle.id][2]:
            self._record(c, position, positions_by_oracle)

    def get_transcript(self):
        t = Transcript()
        t.prover_messages = [m for m in self.prover_messages]
        t.MT_roots = list(self.MT_roots)
        random_cache, det_cache, positions_by_oracle = {}, {}, {}
        for oracle, position in self.queries:                          # "Make sure all queries are hit", in registration order
            self._record(oracle, self._obtain_query_position(position, random_cache, det_cache), positions_by_oracle)
        for mt, (round, dom, ids) in enumerate(self.MT_info):
            cs = self.get_round_parameters(round)
            domain = self.domains[dom]
            num_leaves = domain.size // cs
            qset, lset = set(), set()
            for oid in ids:
                for pos in positions_by_oracle.get(oid, ()):
                    qset.add(pos)
                    lset.add(pos if cs == 1 else (pos // cs if domain.additive else pos % num_leaves))      # bcs_common.tcc:682-696
            qpos, lpos = sorted(qset), sorted(lset)
            t.query_positions.append(qpos)
            t.MT_leaf_positions.append(lpos)
            t.query_responses.append(self.ops.query_responses([self.oracles[i] for i in ids], domain, qpos))
            t.MT_set_membership_proofs.append(self.MT_trees[mt].membership_proof(lpos))
        if self.is_holographic:                                        # remove_index_info_from_transcript (bcs_prover.tcc:119-134)
            t.prover_messages = t.prover_messages[self.num_prover_messages_at_end_of_round[0]:]
            t.MT_roots = t.MT_roots[len(self.oracles_in_round_by_domain(0)):]
        t.proof_of_work = self.pow_answer
        return t
