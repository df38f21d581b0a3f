"""TEST INFRASTRUCTURE: seeded R1CS instances of the shapes a caller builds through r1cs_constraint_system::add_constraint
(/root/reference libiop/relations/r1cs.tcc:151-160, variable.tcc:196-230) — rows are term LISTS of a linear combination: zero to five terms,
index 0 (the constant 1) allowed, an index may repeat inside a row, coefficients are arbitrary field elements in A, B and C, some columns are
hit by a large share of the rows, some rows are empty.  generate_r1cs_example (relations/examples/r1cs_examples.tcc:38-64), the only
instance the reference's harnesses prove, has exactly one unit-coefficient term per row of A and B.

The arithmetic is Python integers only (no oracle, no product code): GF(2^192) = GF(2)[x] / (x^192 + x^7 + x^2 + x + 1) with an element's
words little-endian; the 181-bit prime field with elements in Montgomery form, R = 2^192.  A mistake here cannot make a wrong prover look
right: a generated instance that is not satisfied is rejected by the oracle's own is_satisfied restatement (oracle.r1cs_check_csr) before
any prover sees it."""
import random

import numpy as np

GF192_LOW = (1 << 7) | (1 << 2) | (1 << 1) | 1
MASK192 = (1 << 192) - 1
EDWARDS_R = 1552511030102430251236801561344621993261920897571225601


class GF192:
    name, zero, one = "gf192", 0, 1

    @staticmethod
    def add(a, b):
        return a ^ b

    sub = add

    @staticmethod
    def mul(a, b):
        if a.bit_count() > b.bit_count():
            a, b = b, a
        acc, shift = 0, 0
        while a:
            low = a & -a
            acc ^= b << (low.bit_length() - 1)
            a ^= low
        while acc >> 192:
            hi = acc >> 192
            acc = (acc & MASK192) ^ hi ^ (hi << 1) ^ (hi << 2) ^ (hi << 7)
        return acc

    @staticmethod
    def inv(a):
        """Extended Euclid over GF(2)[x]."""
        assert a
        r0, r1, s0, s1 = (1 << 192) | GF192_LOW, a, 0, 1
        while r1 != 1:
            d = r0.bit_length() - r1.bit_length()
            if d < 0:
                r0, r1, s0, s1, d = r1, r0, s1, s0, -d
            r0 ^= r1 << d
            s0 ^= s1 << d
            if r0.bit_length() < r1.bit_length():
                r0, r1, s0, s1 = r1, r0, s1, s0
        return GF192.mul(s1, 1)           # reduce

    @staticmethod
    def rand(rng):
        return rng.getrandbits(192)

    @staticmethod
    def words(values):
        return np.array([[v & 0xFFFFFFFFFFFFFFFF, (v >> 64) & 0xFFFFFFFFFFFFFFFF, v >> 128] for v in values], dtype=np.uint64).reshape(-1, 3)


class EdwardsFr:
    name, zero, one = "edwards_Fr", 0, 1

    @staticmethod
    def add(a, b):
        return (a + b) % EDWARDS_R

    @staticmethod
    def sub(a, b):
        return (a - b) % EDWARDS_R

    @staticmethod
    def mul(a, b):
        return a * b % EDWARDS_R

    @staticmethod
    def inv(a):
        assert a % EDWARDS_R
        return pow(a, -1, EDWARDS_R)

    @staticmethod
    def rand(rng):
        return rng.getrandbits(200) % EDWARDS_R

    @staticmethod
    def words(values):
        return GF192.words([(v << 192) % EDWARDS_R for v in values])              # Montgomery representatives, R = 2^192


FIELDS = {"gf192": GF192, "edwards_Fr": EdwardsFr}


def _dot(F, row, z):
    acc = F.zero
    for col, coeff in row:
        acc = F.add(acc, F.mul(z[col], coeff))
    return acc


def _csr(F, rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([c for r in rows for c, _ in r], dtype=np.uint32)
    return row_ptr, col, F.words([v for r in rows for _, v in r])


class Instance:
    def __init__(self, F, rows_abc, z, num_inputs):
        self.F, self.rows, self.z, self.num_inputs = F, rows_abc, z, num_inputs      # z = [1, v_1 .. v_m]
        self.num_variables, self.num_constraints = len(z) - 1, len(rows_abc[0])

    @property
    def matrices(self):
        return [_csr(self.F, rows) for rows in self.rows]

    @property
    def assignment(self):
        return self.F.words(self.z[1:])

    def nnz(self):
        return [sum(len(r) for r in rows) for rows in self.rows]

    def violated(self):
        F = self.F
        return [i for i in range(self.num_constraints)
                if F.mul(_dot(F, self.rows[0][i], self.z), _dot(F, self.rows[1][i], self.z)) != _dot(F, self.rows[2][i], self.z)]


def generate(field_name, num_constraints, num_variables, num_inputs, seed, max_nnz=None):
    """A satisfied instance.  max_nnz: the most terms each of A, B, C may hold (Fractal's holographic degree bounds are stated for at most |H|
    non-zero entries per matrix, holographic_lincheck.tcc:72-90); None = one to five terms in nearly every row."""
    F = FIELDS[field_name]
    rng = random.Random(seed)
    z = [F.one] + [F.rand(rng) for _ in range(num_variables)]
    if num_variables > num_inputs + 3:
        z[num_inputs + 2] = F.zero                                                # a variable whose value is zero
    hot = [1 + rng.randrange(max(1, num_inputs)), 1 + num_inputs + rng.randrange(max(1, num_variables - num_inputs))]
    hot = [min(c, num_variables) for c in hot]
    small = [F.one, F.add(F.one, F.one) or 3, F.sub(F.zero, F.one)]
    used = [0, 0, 0]
    dense = max_nnz is None

    def count(q, rows_left, at_least=0):
        if dense:
            return max(at_least, rng.choices([0, 1, 2, 3, 4, 5], [1, 5, 4, 3, 2, 2])[0])
        room = max_nnz - used[q]
        k = rng.choices([0, 1, 2, 3, 5], [9, 8, 3, 1, 1])[0]
        return max(min(k, room), min(at_least, room))

    def term_col(earlier):
        u = rng.random()
        if earlier and u < 0.08:
            return rng.choice(earlier)                                            # the same variable twice in one linear combination
        if u < 0.30:
            return rng.choice(hot)
        if u < 0.40:
            return 0                                                              # the constant 1 (variable.tcc:199-207)
        return 1 + rng.randrange(num_variables)

    def coeff():
        return rng.choice(small) if rng.random() < 0.2 else (F.rand(rng) or F.one)

    def terms(k):
        row = []
        for _ in range(k):
            row.append((term_col([c for c, _ in row]), coeff()))
        return row

    A, B, C = [], [], []
    for i in range(num_constraints):
        left = num_constraints - i
        if not dense and max_nnz - used[2] == 0:
            a_row = []                                                            # C is full: the product must be zero
        else:
            a_row = terms(count(0, left))
        b_row = terms(count(1, left))
        target = F.mul(_dot(F, a_row, z), _dot(F, b_row, z))
        kc = count(2, left, at_least=1 if target != F.zero else 0)
        c_row = terms(max(0, kc - 1))
        rem = F.sub(target, _dot(F, c_row, z))
        if kc >= 1:
            col = term_col([])
            if z[col] == F.zero or rng.random() < 0.3:
                col = 0
            c_row.append((col, F.mul(rem, F.inv(z[col]))))                        # may be a zero coefficient: a stored term all the same
        else:
            assert rem == F.zero
        for q, row in enumerate((a_row, b_row, c_row)):
            used[q] += len(row)
        A.append(a_row); B.append(b_row); C.append(c_row)
    inst = Instance(F, [A, B, C], z, num_inputs)
    assert not inst.violated()
    return inst


def perturbed(inst, kind, seed):
    """An unsatisfied variant: 'constraint' changes one coefficient of C (exactly one violated row, same witness), 'primary' one primary
    input, 'auxiliary' one auxiliary variable (every row that reads it with a non-zero net coefficient breaks)."""
    F, rng = inst.F, random.Random(seed)
    rows = [[list(r) for r in M] for M in inst.rows]
    z = list(inst.z)
    if kind == "constraint":
        candidates = [i for i, r in enumerate(rows[2]) if any(z[c] != F.zero for c, _ in r)]
        i = rng.choice(candidates)
        t = next(t for t, (c, _) in enumerate(rows[2][i]) if z[c] != F.zero)
        c, v = rows[2][i][t]
        rows[2][i][t] = (c, F.add(v, F.one))
    else:
        lo, hi = (1, inst.num_inputs) if kind == "primary" else (inst.num_inputs + 1, inst.num_variables)
        read = sorted({c for M in rows for r in M for c, _ in r if lo <= c <= hi})
        z_col = rng.choice(read)
        z[z_col] = F.add(z[z_col], F.one)
    out = Instance(F, rows, z, inst.num_inputs)
    return out
