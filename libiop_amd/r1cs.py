"""R1CS instances for the device prover: sparse matrices in CSR form resident in HBM, and the synthetic instance the
reference's harnesses prove (libiop/relations/examples/r1cs_examples.tcc:23-78, called as generate_r1cs_example(n, 15, n - 1)
by profiling/instrument_aurora_snark.cpp:108-110), seeded with SplitMix64 instead of libsodium randomness.

Column 0 of a matrix is the constant 1, column j >= 1 is variable j - 1 of z = (primary, auxiliary)
(relations/variable.tcc, r1cs.tcc:236-268).  Citations are relative to the reference tree."""
import numpy as np

import libiop_amd as la


class CSRMatrix:
    """rows x cols sparse matrix, entries on the device: row_ptr (rows + 1, int64), col (int32), coeff ((nnz, 3) elements)."""

    def __init__(self, ops, row_ptr, col, coeff, rows):
        torch = ops.torch
        self.rows = int(rows)
        self.row_ptr, self.col, self.coeff = np.asarray(row_ptr, dtype=np.int64), np.asarray(col, dtype=np.int32), coeff
        self.d_row_ptr = ops.upload_raw(self.row_ptr, torch.int64)
        self.d_col = ops.upload_raw(self.col, torch.int32)
        self.d_coeff = coeff if hasattr(coeff, "data_ptr") else ops.upload(coeff)

    def transposed_onto(self, ops, num_rows_out, out_row_of_col):
        """The transpose with output row out_row_of_col[c] for column c (entries keep their coefficient; their new column is the
        old row): what set_challenge's accumulation p[summation_index(col)] += coeff * alpha^row walks (basic_lincheck_aux.tcc:64-88)."""
        nnz = self.col.shape[0]
        old_row = np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.row_ptr))
        new_row = np.asarray(out_row_of_col, dtype=np.int64)[self.col]
        order = np.argsort(new_row, kind="stable")
        counts = np.bincount(new_row, minlength=num_rows_out)
        row_ptr = np.concatenate([[0], np.cumsum(counts)])
        d_order = ops.upload_raw(order.astype(np.int64), ops.torch.int64)
        coeff = self.d_coeff[d_order] if nnz else self.d_coeff        # gather of 24-byte elements: data movement only
        return CSRMatrix(ops, row_ptr, old_row[order].astype(np.int32), coeff, num_rows_out)


class R1CS:
    """r1cs_constraint_system<FieldT>: A, B, C with num_constraints rows and num_variables + 1 columns."""

    def __init__(self, A, B, C, num_inputs, num_variables):
        self.A, self.B, self.C = A, B, C
        self.num_inputs, self.num_variables = int(num_inputs), int(num_variables)
        self._lincheck_matrices = {}

    def lincheck_matrices(self, ops, num_rows_out, col_to_summation, key):
        """A^T, B^T, C^T with their rows placed at the summation-domain index of each column — the form multi_lincheck's
        set_challenge walks (basic_lincheck_aux.tcc:64-88).  Part of the instance's device-resident representation: built on
        first use for a domain layout (`key`) and kept, like the CSR arrays themselves."""
        if key not in self._lincheck_matrices:
            if callable(col_to_summation):
                col_to_summation = col_to_summation()
            self._lincheck_matrices[key] = [M.transposed_onto(ops, num_rows_out, col_to_summation) for M in (self.A, self.B, self.C)]
        return self._lincheck_matrices[key]

    def cached(self, key, build):
        """Per-instance device-resident helper data that depends on the domain layout only (index permutations): built once."""
        if key not in self._lincheck_matrices:
            self._lincheck_matrices[key] = build()
        return self._lincheck_matrices[key]

    def num_constraints(self):
        return self.A.rows


def _splitmix64(seed, index):
    """SplitMix64 output number `index` of the stream seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (index.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def seeded_elements(field, seed, count):
    """count field elements from the seed: element i takes stream outputs 3 i .. 3 i + 2 as its words (GF(2^192): the raw
    words; the prime field: the 192-bit draw reduced mod p, then Montgomery form)."""
    w = _splitmix64(seed, np.arange(3 * count, dtype=np.uint64)).reshape(count, 3)
    if field.additive:
        return w
    vals = [(int(a) | (int(b) << 64) | (int(c) << 128)) % field.P for a, b, c in w]
    return la.edwards_to_montgomery(vals)


def generate_r1cs_example(ops, num_constraints, num_inputs, num_variables, seed):
    """r1cs_examples.tcc:23-78: constraint i is z[i mod m] * z[(i + 7) mod m] = coef_i * z[(2 i + 1) mod m] with
    coef_i = A B / C (the constant term carries A B when C's variable is zero).  The products and inverses run on the device.
    Returns (R1CS, primary_input, auxiliary_input) with the inputs as host (count, 3) uint64 arrays."""
    if num_inputs > num_variables:
        raise ValueError("Number of inputs can't exceed number of variables.")
    field = ops.field
    z = seeded_elements(field, seed, num_variables)
    i = np.arange(num_constraints, dtype=np.int64)
    a_idx, b_idx, c_idx = i % num_variables, (i + 7) % num_variables, (2 * i + 1) % num_variables
    d_z = ops.upload(z)
    d_idx = lambda idx: ops.upload_raw(idx, ops.torch.int64)
    ab = ops.mul(d_z[d_idx(a_idx)], d_z[d_idx(b_idx)])
    c_inv = ops.inv(d_z)                                           # zero stays zero
    coef = ops.mul(ab, c_inv[d_idx(c_idx)])
    c_zero = ~np.any(z[c_idx] != 0, axis=1)
    if c_zero.any():                                               # C.add_term(0, AB_val)
        coef_h, ab_h = ops.download(coef), ops.download(ab)
        coef_h[c_zero] = ab_h[c_zero]
        coef = ops.upload(coef_h)
    one = field.from_int(1) if not field.additive else np.array([1, 0, 0], dtype=np.uint64)
    ones = np.broadcast_to(one, (num_constraints, 3))
    row_ptr = np.arange(num_constraints + 1, dtype=np.int64)
    A = CSRMatrix(ops, row_ptr, a_idx + 1, ones, num_constraints)
    B = CSRMatrix(ops, row_ptr, b_idx + 1, ones, num_constraints)
    C = CSRMatrix(ops, row_ptr, np.where(c_zero, 0, c_idx + 1), coef, num_constraints)
    return R1CS(A, B, C, num_inputs, num_variables), z[:num_inputs].copy(), z[num_inputs:].copy()
