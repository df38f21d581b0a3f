"""Fractal preprocessing SNARK on the device path (non-zk, BLAKE2b): indexer and prover as host orchestration over the C ABI,
every vector in HBM.  The classes mirror the reference's composition so that registration order — rounds, Merkle trees, the
hashchain's squeeze order — is the reference's:

    fractal_snark_indexer / fractal_snark_prover / fractal_snark_parameters   libiop/snark/fractal_snark.tcc:7-162
    fractal_iop_parameters, fractal_iop                                       libiop/protocols/fractal_hiop.tcc:5-329
    matrix_indexer                                                            libiop/protocols/encoded/r1cs_rs_iop/fractal_indexer.tcc
    holographic_multi_lincheck (+ its virtual oracle, single_matrix_denominator)   .../encoded/lincheck/holographic_lincheck{,_aux}.tcc
    compute_p_alpha_M                                                         .../encoded/lincheck/common.tcc
    rational_sumcheck_protocol, sumcheck_constraint_oracle                    .../encoded/sumcheck/rational_sumcheck.tcc
    rational_linear_combination, single_boundary_constraint                   .../encoded/common/
    bcs_indexer, bcs_prover's index handling                                  libiop/bcs/bcs_indexer.tcc, bcs_prover.tcc:12-21,68-80,119-134

The encoded witness part (f_w, f_Az, f_Bz, f_Cz, fz, rowcheck), the batch sumcheck over H, the LDT instance reducer and FRI are
libiop_amd/aurora.py's.  Citations are relative to the reference tree."""
import math

import numpy as np

from . import host
from .aurora import BatchSumcheckProtocol, EncodedAuroraProtocol, LDTInstanceReducer, _is_pow2, _log2
from .bcs import BCSProver, VirtualOracle


class FractalParameters:
    """fractal_snark_parameters + fractal_iop_parameters (non-zk, heuristic FRI soundness, optimistic-heuristic LDT-reducer
    soundness; profiling/instrument_fractal_snark.cpp:93-120: RS_extra_dimensions 3, localization 2)."""

    def __init__(self, field, constraint_system, security_parameter=128, RS_extra_dimensions=3, FRI_localization_parameter=2):
        cs = constraint_system
        n = cs.num_constraints()
        if not _is_pow2(n):
            raise ValueError("Fractal requires the number of constraints to be a power of two")
        if n != cs.num_variables + 1:
            raise ValueError("Fractal requires the matrices to be square")
        self.field, self.security_parameter, self.RS_extra_dimensions = field, security_parameter, RS_extra_dimensions
        self.num_constraints, self.num_variables, self.num_inputs = n, cs.num_variables, cs.num_inputs
        max_nonzero = max(int(M.row_ptr[-1]) for M in (cs.A, cs.B, cs.C))                        # fractal_hiop.tcc:28-35
        self.index_domain_dim = _log2(max_nonzero)
        self.matrix_domain_dim = _log2(n)
        self.codeword_domain_dim = _log2(4 << self.index_domain_dim) + RS_extra_dimensions       # :38-39
        self.pow_bits = _log2(n) + 3                                                             # fractal_snark.tcc:90-95
        self.query_soundness_error_bits = security_parameter + 1 - self.pow_bits                 # fractal_hiop.tcc:77-78
        self.interactive_soundness_error_bits = security_parameter + 3
        self.localization_parameters = host.localization_parameter_to_array(FRI_localization_parameter, self.codeword_domain_dim, RS_extra_dimensions)
        fbits = field.soundness_bits
        ceil_div = lambda bits, per: max(1, math.ceil(-bits / per))
        self.holographic_lincheck_repetitions = ceil_div(self.interactive_soundness_error_bits, 1 + self.matrix_domain_dim - fbits)   # holographic_lincheck.tcc:16-36
        H = 1 << self.matrix_domain_dim
        self.max_tested_degree_bound = max(3 * H, H - 1)                                         # r1cs_rs_iop.tcc:56-97, holographic, b = 0
        self.max_constraint_degree_bound = max(4 * H, 2 * H - 1)
        step = 1 << sum(self.localization_parameters)                                           # next_testable_degree_bound (fri_ldt.tcc:148-163)
        rem = self.max_tested_degree_bound % step
        self.max_LDT_tested_degree_bound = self.max_tested_degree_bound if rem == 0 else self.max_tested_degree_bound - rem + step
        codeword_size = 1 << self.codeword_domain_dim
        if self.max_LDT_tested_degree_bound >= codeword_size or self.max_constraint_degree_bound >= codeword_size:
            raise ValueError("degree bounds exceed the codeword domain")
        self.absolute_proximity_parameter = min(codeword_size - self.max_constraint_degree_bound, codeword_size - self.max_LDT_tested_degree_bound) - 1
        self.num_output_LDT_instances = ceil_div(self.interactive_soundness_error_bits, self.codeword_domain_dim - fbits)
        delta = self.absolute_proximity_parameter / codeword_size
        self.fri_query_repetitions = ceil_div(self.query_soundness_error_bits, math.log2(1 - delta))
        per_interaction = math.log2((1 << self.localization_parameters[0]) - 1) - fbits
        self.fri_interactive_repetitions = ceil_div(self.interactive_soundness_error_bits, per_interaction)


# ---- virtual oracles ----
class HolographicMultiLincheckVirtualOracle(VirtualOracle):
    """holographic_multi_lincheck_virtual_oracle (holographic_lincheck_aux.tcc:4-95): p(alpha, x) sum_m r_m f_Mz(x) - f_z(x) t(x)."""

    def __init__(self, ops, codeword_domain, summation_domain, num_matrices):
        self.ops, self.L, self.H, self.num_matrices = ops, codeword_domain, summation_domain, num_matrices
        self.alpha = self.r_Mz = None

    def set_challenge(self, alpha, r_Mz):
        if len(r_Mz) != self.num_matrices:
            raise ValueError("Not enough random linear combination coefficients were provided")
        self.alpha, self.r_Mz = alpha, r_Mz

    def evaluated_contents(self, constituents):
        if len(constituents) != self.num_matrices + 2:
            raise ValueError("multi_lincheck uses more constituent oracles than what was provided.")
        p_alpha_prime = self.ops.lagrange_evals(self.alpha, self.H, self.L)                              # :37-39
        return self.ops.lincheck(constituents[0], constituents[1:-1], self.r_Mz, p_alpha_prime, constituents[-1], constituents[0].shape[0])


class SingleMatrixDenominator(VirtualOracle):
    """single_matrix_denominator (holographic_lincheck_aux.tcc:97-169): (row - row_query)(col - col_query) from (row, col, row*col)."""

    def __init__(self, ops):
        self.ops, self.row_query_point, self.column_query_point = ops, None, None

    def set_challenge(self, row_query_point, column_query_point):
        self.row_query_point, self.column_query_point = row_query_point, column_query_point

    def evaluated_contents(self, constituents):
        if len(constituents) != 3:
            raise ValueError("single_matrix_denominator was expecting row, col, row*col oracles as input")
        f = self.ops.field
        coefficients = np.stack([f.neg(self.column_query_point), f.neg(self.row_query_point), f.one()])
        return self.ops.lincomb_affine(constituents, coefficients, f.mul(self.row_query_point, self.column_query_point), constituents[0].shape[0])


class RationalLinearCombination:
    """rational_linear_combination (common/rational_linear_combination.tcc:136-212): registers the combined numerator and the
    combined denominator as two virtual oracles; one kernel produces both, so the pair is computed on the first request and the
    second is served from it."""

    class _Numerator(VirtualOracle):
        def __init__(self, owner):
            self.owner = owner

        def evaluated_contents(self, constituents):
            n = self.owner.num_rationals
            if len(constituents) != 2 * n:
                raise ValueError("Expected same number of evaluations as in registration.")
            return self.owner._pair(constituents[:n], constituents[n:])[0]

    class _Denominator(VirtualOracle):
        def __init__(self, owner):
            self.owner = owner

        def evaluated_contents(self, constituents):
            if len(constituents) != self.owner.num_rationals:
                raise ValueError("Expected same number of evaluations as in registration.")
            key = tuple(t.data_ptr() for t in constituents)
            if self.owner._last is not None and self.owner._last[0] == key:
                return self.owner._last[2]
            out = constituents[0]
            for t in constituents[1:]:
                out = self.owner.ops.mul(out, t)
            return out

    def __init__(self, IOP, num_rationals, numerator_handles, denominator_handles):
        if len(numerator_handles) != num_rationals or len(denominator_handles) != num_rationals:
            raise ValueError("Rational Linear Combination: #numerator handles passed in != #denominator handles passed in")
        # no reference back to the round driver: it holds the two virtual oracles below, and a cycle would keep the proof's codewords
        # alive until the cyclic collector runs instead of until the prover returns
        self.ops, self.num_rationals, self.coefficients, self._last = IOP.ops, num_rationals, None, None
        domain = IOP.get_oracle_domain(numerator_handles[0])
        denominator_degree = 1 + sum(IOP.get_oracle_degree(h) - 1 for h in denominator_handles)
        self.denominator_handle = IOP.register_virtual_oracle(domain, denominator_degree, denominator_handles, self._Denominator(self))
        numerator_degree = max(IOP.get_oracle_degree(n) + denominator_degree - IOP.get_oracle_degree(d)
                               for n, d in zip(numerator_handles, denominator_handles))
        self.numerator_handle = IOP.register_virtual_oracle(domain, numerator_degree, list(numerator_handles) + list(denominator_handles),
                                                            self._Numerator(self))

    def set_coefficients(self, coefficients):
        if len(coefficients) != self.num_rationals:
            raise ValueError("Expected same number of random coefficients as oracles.")
        self.coefficients, self._last = np.stack(coefficients), None

    def _pair(self, numerators, denominators):
        N, D = self.ops.rational_combine(numerators, denominators, self.coefficients, numerators[0].shape[0])
        self._last = (tuple(t.data_ptr() for t in denominators), N, D, denominators)      # keeps the keyed tensors alive
        return N, D

    def evaluated_contents(self, numerator_evals, denominator_evals):
        """:183-209 — the combined rational function itself (over the index domain)."""
        N, D = self._pair(numerator_evals, denominator_evals)
        self._last = None
        return self.ops.div(N, D)


class SingleBoundaryConstraint(VirtualOracle):
    """single_boundary_constraint (common/boundary_constraint.tcc): (f(x) - claimed_eval) / (x - eval_point)."""

    def __init__(self, ops, codeword_domain):
        self.ops, self.L, self.eval_point, self.oracle_evaluation = ops, codeword_domain, None, None

    def set_evaluation_point_and_eval(self, eval_point, oracle_eval):
        self.eval_point, self.oracle_evaluation = eval_point, oracle_eval

    def evaluated_contents(self, constituents):
        if len(constituents) != 1:
            raise ValueError("Single Boundary Constraint: Expected exactly 1 constituent oracle.")
        f, ops = self.ops.field, self.ops
        if f.element_in_domain(self.L, self.eval_point):
            raise NotImplementedError("the evaluation point lies in the codeword domain")
        numerator = ops.lincomb_affine(constituents, f.neg(f.one()).reshape(1, 3), self.oracle_evaluation, constituents[0].shape[0])   # claimed - f
        return ops.div(numerator, ops.domain_offsets(self.L, self.eval_point))                                                      # / (point - x)


class SumcheckConstraintOracle(VirtualOracle):
    """sumcheck_constraint_oracle (rational_sumcheck.tcc:9-137), constituents (p, N, D)."""

    def __init__(self, ops, summation_domain, codeword_domain):
        self.ops, self.K, self.L, self.claimed_sum = ops, summation_domain, codeword_domain, ops.field.zero()

    def set_claimed_sum(self, claimed_sum):
        self.claimed_sum = claimed_sum

    def evaluated_contents(self, constituents):
        if len(constituents) != 3:
            raise ValueError("sumcheck_constraint_oracle has three constituent oracles")
        return self.ops.rational_sumcheck_constraint(constituents[0], constituents[1], constituents[2], self.L, self.K, self.claimed_sum)


# ---- protocols ----
class RationalSumcheckProtocol:
    """rational_sumcheck_protocol (rational_sumcheck.tcc:139-274)."""

    def __init__(self, IOP, summation_domain_handle, codeword_domain_handle, numerator_degree_bound, denominator_degree_bound):
        self.IOP, self.ops = IOP, IOP.ops
        self.summation_domain_handle, self.codeword_domain_handle = summation_domain_handle, codeword_domain_handle
        self.K, self.L = IOP.get_domain(summation_domain_handle), IOP.get_domain(codeword_domain_handle)
        self.reextended_oracle_degree = self.K.size - 1
        self.constraint_oracle_degree = max(numerator_degree_bound, denominator_degree_bound + self.K.size - 1) - self.K.size
        self.claimed_sum = None

    def register_summation_oracle(self, numerator_handle, denominator_handle):
        self.numerator_handle, self.denominator_handle = numerator_handle, denominator_handle

    def register_proof(self):
        self.reextended_oracle_handle = self.IOP.register_oracle("rational sumcheck reextension", self.codeword_domain_handle,
                                                                 self.reextended_oracle_degree, False)
        self.constraint_oracle = SumcheckConstraintOracle(self.ops, self.K, self.L)
        self.constraint_oracle_handle = self.IOP.register_virtual_oracle(
            self.codeword_domain_handle, self.constraint_oracle_degree,
            [self.reextended_oracle_handle, self.numerator_handle, self.denominator_handle], self.constraint_oracle)

    def calculate_and_submit_proof(self, d_rational_function_over_summation_domain):
        """:222-252 — interpolate over K, take the sum off the polynomial (its constant term times |K| / eps times its top
        coefficient), re-extend the rest."""
        ops, f, n = self.ops, self.ops.field, self.K.size
        coeffs = ops.IFFT(d_rational_function_over_summation_domain, self.K)
        if self.K.additive:
            eps = f.vanishing_derivative(self.K, f.zero(), ops.lib)
            self.claimed_sum = f.mul(eps, ops.download(coeffs[n - 1:n])[0])
            rest = coeffs[: n - 1]
        else:
            self.claimed_sum = f.mul(ops.download(coeffs[:1])[0], f.from_int(n))
            rest = coeffs[1:]
        self.IOP.submit_oracle(self.reextended_oracle_handle, ops.FFT(rest, n - 1, self.L))
        self.constraint_oracle.set_claimed_sum(self.claimed_sum)

    def get_claimed_sum(self):
        return self.claimed_sum

    def get_all_oracle_handles(self):
        return [self.reextended_oracle_handle, self.constraint_oracle_handle]


class HolographicMultiLincheck:
    """holographic_multi_lincheck (holographic_lincheck.tcc:113-580), non-zk."""

    def __init__(self, IOP, codeword_domain_handle, summation_domain_handle, input_variable_dim, transposed_matrices, fz_handle, Mz_handles,
                 repetitions):
        self.IOP, self.ops = IOP, IOP.ops
        self.codeword_domain_handle, self.summation_domain_handle = codeword_domain_handle, summation_domain_handle
        self.input_variable_dim, self.matrices_T, self.repetitions = input_variable_dim, transposed_matrices, repetitions
        self.num_matrices = len(transposed_matrices)
        if self.num_matrices < 1:
            raise ValueError("multi_lincheck expects at least one matrix")
        if len(Mz_handles) != self.num_matrices:
            raise ValueError("inconsistent number of Mz_handles and matrices passed into multi lincheck.")
        self.L, self.H = IOP.get_domain(codeword_domain_handle), IOP.get_domain(summation_domain_handle)
        self.constituent_oracle_handles = [fz_handle] + list(Mz_handles)
        self.lincheck_degree = self.H.size + max(IOP.get_oracle_degree(fz_handle), IOP.get_oracle_degree(Mz_handles[0])) - 1      # :146-150
        self.sumcheck_H = [BatchSumcheckProtocol(IOP, summation_domain_handle, codeword_domain_handle, self.lincheck_degree) for _ in range(repetitions)]
        self.lincheck_oracles = [HolographicMultiLincheckVirtualOracle(self.ops, self.L, self.H, self.num_matrices) for _ in range(repetitions)]
        self.t_boundary_constraint = [SingleBoundaryConstraint(self.ops, self.L) for _ in range(repetitions)]

    def set_index_oracles(self, indexed_domain_handle, indexed_handles, index_evals_over_K):
        """:190-254.  index_evals_over_K[i] = (row, col, val, row*col) of matrix i over the index domain, device resident: the
        reference recomputes them inside calculate_response_beta (:447-458, "TODO: Also index evals over K"); here they are part
        of the prover's index."""
        if len(indexed_handles) != self.num_matrices:
            raise ValueError("Incorrect number of sets of indexed oracles")
        if any(len(hs) != 4 for hs in indexed_handles):
            raise ValueError("Incorrect number of indexed oracles within set")
        IOP = self.IOP
        self.index_domain_handle, self.K, self.index_evals_over_K = indexed_domain_handle, IOP.get_domain(indexed_domain_handle), index_evals_over_K
        single = self.K.size
        combined_numerator_degree = single + (self.num_matrices - 1) * single - (self.num_matrices - 1)
        combined_denominator_degree = self.num_matrices * single - (self.num_matrices - 1)
        self.matrix_denominators, self.matrix_numerator_handles, self.matrix_denominator_handles, self.sumcheck_K = [], [], [], []
        for _ in range(self.repetitions):
            dens = [SingleMatrixDenominator(self.ops) for _ in range(self.num_matrices)]
            self.matrix_denominators.append(dens)
            self.matrix_numerator_handles.append([hs[2] for hs in indexed_handles])                    # val
            # cached: the combined numerator and the combined denominator both read them (prover-side choice, not in the transcript)
            self.matrix_denominator_handles.append([IOP.register_virtual_oracle(self.codeword_domain_handle, single, [hs[0], hs[1], hs[3]], d, True)
                                                    for hs, d in zip(indexed_handles, dens)])           # row, col, row*col
            self.sumcheck_K.append(RationalSumcheckProtocol(IOP, indexed_domain_handle, self.codeword_domain_handle,
                                                            combined_numerator_degree, combined_denominator_degree))

    def register_challenge_alpha(self):                                                                  # :256-265
        self.alpha_handle = [self.IOP.register_verifier_random_message(1) for _ in range(self.repetitions)]
        self.random_coefficient_handle = [self.IOP.register_verifier_random_message(self.num_matrices) for _ in range(self.repetitions)]

    def register_response_alpha(self):                                                                   # :267-300
        self.t_oracle_handle = []
        for r in range(self.repetitions):
            t = self.IOP.register_oracle("lincheck_t", self.codeword_domain_handle, self.H.size, False)
            self.t_oracle_handle.append(t)
            h = self.IOP.register_virtual_oracle(self.codeword_domain_handle, self.lincheck_degree, self.constituent_oracle_handles + [t],
                                                 self.lincheck_oracles[r])
            self.sumcheck_H[r].attach_oracle_for_summing(h)

    def register_challenge_beta(self):                                                                   # :302-310
        self.beta_handle = [self.IOP.register_verifier_random_message(1) for _ in range(self.repetitions)]
        for r in range(self.repetitions):
            self.sumcheck_H[r].register_challenge()

    def register_response_beta(self):                                                                    # :312-366
        IOP = self.IOP
        self.M_at_alpha_beta = [IOP.register_prover_message(1) for _ in range(self.repetitions)]
        self.rational_linear_combination, self.t_boundary_constraint_handle = [], []
        for r in range(self.repetitions):
            rlc = RationalLinearCombination(IOP, self.num_matrices, self.matrix_numerator_handles[r], self.matrix_denominator_handles[r])
            self.rational_linear_combination.append(rlc)
            self.sumcheck_K[r].register_summation_oracle(rlc.numerator_handle, rlc.denominator_handle)
            self.t_boundary_constraint_handle.append(IOP.register_virtual_oracle(self.codeword_domain_handle, self.H.size - 1,
                                                                                 [self.t_oracle_handle[r]], self.t_boundary_constraint[r]))
            self.sumcheck_H[r].register_proof()
            self.sumcheck_K[r].register_proof()

    def calculate_response_alpha(self):                                                                  # :381-417
        IOP, ops = self.IOP, self.ops
        self.r_Mz = [None] * self.repetitions
        for r in range(self.repetitions):
            alpha = IOP.obtain_verifier_random_message(self.alpha_handle[r])[0]
            self.r_Mz[r] = IOP.obtain_verifier_random_message(self.random_coefficient_handle[r])
            p_alpha_over_H = ops.lagrange_evals(alpha, self.H, self.H)                                   # unnormalised (:393-397)
            p_alpha_M_over_H = ops.empty(self.H.size)                                                    # compute_p_alpha_M (common.tcc:5-38)
            for m, MT in enumerate(self.matrices_T):
                ops.spmv(MT, p_alpha_over_H, d_out=p_alpha_M_over_H, scale=self.r_Mz[r][m], accumulate=m > 0)
            t = ops.reextend_packed(p_alpha_M_over_H, 1, self.H, self.L)[0]                              # IFFT over H, FFT over L (:33, :410)
            IOP.submit_oracle(self.t_oracle_handle[r], t)
            self.lincheck_oracles[r].set_challenge(alpha, self.r_Mz[r])

    def _set_rational_linear_combination_coefficients(self):                                             # :480-499
        f, lib = self.ops.field, self.ops.lib
        for r in range(self.repetitions):
            alpha = self.IOP.obtain_verifier_random_message(self.alpha_handle[r])[0]
            beta = self.IOP.obtain_verifier_random_message(self.beta_handle[r])[0]
            shift = f.mul(f.vanishing_eval(self.H, alpha, lib), f.vanishing_eval(self.H, beta, lib))
            self.rational_linear_combination[r].set_coefficients([f.mul(shift, self.r_Mz[r][i]) for i in range(self.num_matrices)])

    def _set_matrix_denominator_challenges(self):                                                        # :501-513
        for r in range(self.repetitions):
            alpha = self.IOP.obtain_verifier_random_message(self.alpha_handle[r])[0]
            beta = self.IOP.obtain_verifier_random_message(self.beta_handle[r])[0]
            for d in self.matrix_denominators[r]:
                d.set_challenge(beta, alpha)

    def calculate_response_beta(self):                                                                   # :430-478
        IOP = self.IOP
        self._set_rational_linear_combination_coefficients()
        self._set_matrix_denominator_challenges()
        for r in range(self.repetitions):
            beta = IOP.obtain_verifier_random_message(self.beta_handle[r])[0]
            numerators = [ev[2] for ev in self.index_evals_over_K]
            denominators = [self.matrix_denominators[r][i].evaluated_contents([ev[0], ev[1], ev[3]]) for i, ev in enumerate(self.index_evals_over_K)]
            combined_rational_over_K = self.rational_linear_combination[r].evaluated_contents(numerators, denominators)
            self.sumcheck_K[r].calculate_and_submit_proof(combined_rational_over_K)
            M_at_alpha_beta = self.sumcheck_K[r].get_claimed_sum()
            IOP.submit_prover_message(self.M_at_alpha_beta[r], M_at_alpha_beta.reshape(1, 3))
            self.t_boundary_constraint[r].set_evaluation_point_and_eval(beta, M_at_alpha_beta)
            self.sumcheck_H[r].calculate_and_submit_proof()

    def get_all_oracle_handles(self):                                                                    # :550-580
        out = []
        for r in range(self.repetitions):
            out += [self.t_oracle_handle[r], self.t_boundary_constraint_handle[r]]
            out += self.sumcheck_H[r].get_all_oracle_handles() + self.sumcheck_K[r].get_all_oracle_handles()
        return out


def matrix_index_over_K(ops, M, K, H, input_variable_dim):
    """matrix_indexer::compute_oracles_over_K (fractal_indexer.tcc:47-121) on the device: gathers of the matrix domain's elements by
    the entries' row / column, the values scaled by 1 / u_H(col, col) = 1 / (DZ_H)(col), padding, and the transposition swap.
    Returns [row, col, val, row*col] over the index domain K."""
    f, torch = ops.field, ops.torch
    nnz = int(M.row_ptr[-1])
    row_index = np.repeat(np.arange(M.rows, dtype=np.int64), np.diff(M.row_ptr))
    col_index = H.reindex_by_subset_array(input_variable_dim, H.size)[M.col.astype(np.int64)]
    H_elements = ops.domain_elements(H)
    row_evals = H_elements[ops.upload_raw(row_index, torch.int64)[:nnz]]
    col_evals = H_elements[ops.upload_raw(col_index, torch.int64)[:nnz]]
    row_times_col = ops.mul(row_evals, col_evals)
    if H.additive:                                            # (DZ_H) is the constant linear coefficient
        scale = f.inv(f.vanishing_derivative(H, f.zero(), ops.lib), ops.lib)
        val_evals = ops.lincomb([M.d_coeff[:nnz]], scale.reshape(1, 3), nnz)
    else:                                                     # (DZ_H)(c) = |H| c^(|H| - 1) = |H| shift^|H| / c on the coset
        scale = f.inv(f.from_int(H.size * pow(H.shift_int, H.size, f.P)))
        val_evals = ops.lincomb([ops.mul(M.d_coeff[:nnz], col_evals)], scale.reshape(1, 3), nnz)
    pad = K.size - nnz
    h0 = H_elements[:1]
    k0 = ops.domain_elements(K)[:1]
    padded = lambda t, fill: t if pad == 0 else torch.cat([t, fill.expand(pad, 3)])
    zero = ops.upload(f.zero().reshape(1, 3))
    rows, cols = padded(row_evals, h0), padded(col_evals, h0)
    vals, rcs = padded(val_evals, zero), padded(row_times_col, ops.mul(k0, k0))
    return [cols.contiguous(), rows.contiguous(), vals.contiguous(), rcs.contiguous()]                   # "We are dealing with the transpose"


class MatrixIndexer:
    """matrix_indexer (fractal_indexer.tcc): the index of M' = M^T scaled by u_H(col, col), four oracles per matrix."""

    def __init__(self, IOP, index_domain_handle, matrix_domain_handle, codeword_domain_handle, input_variable_dim, matrix):
        self.IOP, self.ops, self.matrix, self.input_variable_dim = IOP, IOP.ops, matrix, input_variable_dim
        self.index_domain_handle, self.matrix_domain_handle, self.codeword_domain_handle = index_domain_handle, matrix_domain_handle, codeword_domain_handle
        self.K, self.H, self.L = IOP.get_domain(index_domain_handle), IOP.get_domain(matrix_domain_handle), IOP.get_domain(codeword_domain_handle)

    def register_oracles(self):                                                                          # :29-45: row, col, val, row*col
        bound = self.K.size
        if bound < int(self.matrix.row_ptr[-1]):
            raise AssertionError("index domain smaller than the number of non-zero entries")
        self.handles = [self.IOP.register_index_oracle(self.codeword_domain_handle, bound) for _ in range(4)]
        return self.handles

    def compute_oracles_over_K(self):
        return matrix_index_over_K(self.ops, self.matrix, self.K, self.H, self.input_variable_dim)

    def compute_oracles(self, over_K=None):                                                              # :123-156
        over_K = self.compute_oracles_over_K() if over_K is None else over_K
        packed = self.ops.torch.cat(over_K)
        codewords = self.ops.reextend_packed(packed, 4, self.K, self.L)
        for handle, cw in zip(self.handles, codewords):
            self.IOP.submit_oracle(handle, cw)
        return over_K


class FractalIOP:
    """fractal_iop (fractal_hiop.tcc:218-346)."""

    def __init__(self, IOP, constraint_system, params, index_evals_over_K=None):
        self.IOP, self.params, self.cs = IOP, params, constraint_system
        f, ops = IOP.field, IOP.ops
        index_domain, matrix_domain = f.domain(1 << params.index_domain_dim), f.domain(params.num_constraints)
        codeword_domain_shift = f.domain(1 << params.codeword_domain_dim).element_outside_of_subset()
        codeword_domain = ops.mark_codeword_domain(f.domain(1 << params.codeword_domain_dim, codeword_domain_shift))
        self.index_domain_handle = IOP.register_domain(index_domain)
        self.matrix_domain_handle = IOP.register_domain(matrix_domain)
        self.codeword_domain_handle = IOP.register_domain(codeword_domain)
        self._quotient_map_domain = codeword_domain.get_subset_of_order(1 << params.localization_parameters[0])
        # register_index_oracles (:277-300); libff::log2(num_inputs)
        input_variable_dim = _log2(constraint_system.num_inputs)
        self.matrix_indexers = [MatrixIndexer(IOP, self.index_domain_handle, self.matrix_domain_handle, self.codeword_domain_handle, input_variable_dim, M)
                                for M in (constraint_system.A, constraint_system.B, constraint_system.C)]
        self.indexed_handles = [mi.register_oracles() for mi in self.matrix_indexers]
        IOP.set_round_parameters(self._quotient_map_domain)
        IOP.signal_index_registrations_done()
        # :253-275
        self.protocol = EncodedAuroraProtocol(IOP, self.matrix_domain_handle, self.matrix_domain_handle, self.codeword_domain_handle, constraint_system,
                                              0, holographic=True)
        self.lincheck = HolographicMultiLincheck(IOP, self.codeword_domain_handle, self.matrix_domain_handle, self.protocol.I.dim,
                                                 self.protocol.transposed_matrices, self.protocol.fz_handle, self.protocol.Mz_handles,
                                                 params.holographic_lincheck_repetitions)
        if index_evals_over_K is not None and input_variable_dim != self.protocol.I.dim:
            # Reference quirk F15: the indexer reindexes columns with libff::log2(num_inputs) (fractal_hiop.tcc:279) while the lincheck
            # rebuilds the index evaluations with log2(num_inputs + 1) (holographic_lincheck.tcc:447-458 via r1cs_rs_iop.tcc:352); they
            # differ for num_inputs = 1 only (multiplicative domains), where the reference's own proof is rejected.  Follow it.
            K, Hd = IOP.get_domain(self.index_domain_handle), IOP.get_domain(self.matrix_domain_handle)
            index_evals_over_K = [matrix_index_over_K(ops, M, K, Hd, self.protocol.I.dim) for M in (constraint_system.A, constraint_system.B, constraint_system.C)]
        self.lincheck.set_index_oracles(self.index_domain_handle, self.indexed_handles, index_evals_over_K)
        self.LDT_reducer = LDTInstanceReducer(IOP, self.codeword_domain_handle, params.num_output_LDT_instances, params.max_LDT_tested_degree_bound)
        IOP.set_round_parameters(self._quotient_map_domain)

    def register_interactions(self):                                                                     # :302-325
        IOP, p = self.IOP, self.protocol
        self.lincheck.register_challenge_alpha()
        IOP.set_round_parameters(self._quotient_map_domain)
        self.lincheck.register_response_alpha()
        self.lincheck.register_challenge_beta()
        self.lincheck.register_response_beta()
        IOP.set_round_parameters(self._quotient_map_domain)
        handles = self.lincheck.get_all_oracle_handles() + [p.fw_handle, p.fAz_handle, p.fBz_handle, p.fCz_handle, p.rowcheck_handle]   # r1cs_rs_iop.tcc:650-668
        self.LDT_reducer.register_interactions(handles, self.params.localization_parameters, self.params.fri_interactive_repetitions,
                                               self.params.fri_query_repetitions)

    def register_queries(self):
        self.LDT_reducer.register_queries()

    def produce_index(self):                                                                             # :306-314
        over_K = [mi.compute_oracles() for mi in self.matrix_indexers]
        self.IOP.signal_index_submissions_done()
        return over_K

    def produce_proof(self, primary_input, auxiliary_input, index, d_assignment=None):                   # :316-329, r1cs_rs_iop.tcc:618-627
        IOP = self.IOP
        IOP.submit_prover_index(index)
        self.protocol.submit_witness_oracles(primary_input, auxiliary_input, d_assignment)
        IOP.signal_prover_round_done()
        self.lincheck.calculate_response_alpha()
        IOP.signal_prover_round_done()
        self.lincheck.calculate_response_beta()
        IOP.signal_prover_round_done()
        self.LDT_reducer.calculate_and_submit_proof()


def fractal_snark_indexer(ops, constraint_system, parameters):
    """fractal_snark_indexer (fractal_snark.tcc:114-133): returns (prover index, verifier index) — the twelve index oracles over the
    codeword domain with their Merkle tree (and their evaluations over the index domain) on the device / the tree's root."""
    IOP = BCSProver(ops, parameters.pow_bits)
    full_protocol = FractalIOP(IOP, constraint_system, parameters)
    IOP.seal_interaction_registrations()
    IOP.seal_query_registrations()
    over_K = full_protocol.produce_index()
    return IOP.get_prover_index(extra=over_K), IOP.get_verifier_index()


def fractal_snark_prover(ops, index, constraint_system, primary_input, auxiliary_input, parameters, round_hook=None, d_assignment=None):
    """fractal_snark_prover (fractal_snark.tcc:135-162): returns the Transcript (libiop_amd/bcs.py) without the index's roots."""
    IOP = BCSProver(ops, parameters.pow_bits, index)
    if round_hook is not None:
        IOP.round_hooks.append(round_hook)
    full_protocol = FractalIOP(IOP, constraint_system, parameters, index.extra)
    full_protocol.register_interactions()
    IOP.seal_interaction_registrations()
    full_protocol.register_queries()
    IOP.seal_query_registrations()
    full_protocol.produce_proof(primary_input, auxiliary_input, index, d_assignment)
    transcript = IOP.get_transcript()
    IOP.release()
    return transcript
