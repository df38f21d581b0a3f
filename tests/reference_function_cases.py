"""tests/golden/reference_functions.json: outputs of the REFERENCE'S OWN functions (additive / multiplicative FFT, IFFT, known-degree IFFT, FRI folds,
the LDT combination, BLAKE2b trees by cosets, the proof-of-work grind), computed in the build container by tests/harness/reference_vectors.cpp — libiop's
sources compiled unmodified over a stand-in libff — on seeded inputs, one BLAKE2b-256 digest per case.  The same inputs are rebuilt here from the seeds
and the oracle, the CPU build of the kernels and the HIP kernels are compared with the digests.  (What the digests pin: libiop's loops as libiop's code
runs them.  Not libff's bytes: the shim's field layout is the one the kernels assume.)"""
import hashlib
import json
import os

import numpy as np

import libiop_amd
import oracle
from libiop_amd import domains, r1cs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = 3
GF, FP = domains.GF192(), domains.EdwardsFr()


def entries(cases=None):
    with open(os.path.join(ROOT, "tests", "golden", "reference_functions.json")) as f:
        out = json.load(f)["entries"]
    return [e for e in out if cases is None or e["case"] in cases]


def ident(e):
    return "-".join(str(e[k]) for k in ("case", "field", "set", "m", "log_n", "kind", "ncoeffs", "degree", "oracles", "coset_size", "x_in_domain", "work_parameter") if k in e)


def _digest(a):
    return hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=32).hexdigest()


def _gf(seed, n):
    return r1cs.seeded_elements(GF, seed, n)


def _fp(seed, n):
    return r1cs.seeded_elements(FP, seed, n)


def _subspace(m, kind, seed):
    """reference_vectors.cpp subspace(): 0 standard basis, shift 0; 1 shift x^m; 2 seeded shift; 3 seeded basis and shift"""
    basis = oracle.standard_basis(m, W)
    shift = np.zeros(W, dtype=np.uint64)
    if kind == 1:
        shift = np.array([1 << m, 0, 0], dtype=np.uint64)
    if kind >= 2:
        shift = _gf(seed + 2, 1)[0]
    if kind == 3:
        basis = _gf(seed + 1, m)
    return basis, shift


def _element_by_index(basis, shift, i):
    x = shift.copy()
    for k in range(basis.shape[0]):
        if (i >> k) & 1:
            x ^= basis[k]
    return x


def _coset_shift(kind, seed):
    if kind == 0:
        return libiop_amd.edwards_to_montgomery([1])[0]
    if kind == 1:
        return libiop_amd.edwards_to_montgomery([libiop_amd.EDWARDS_FR_GENERATOR])[0]
    return _fp(seed + 2, 1)[0]


LDT_DEGREES = lambda n: [n // 4, n // 8, n // 4 - 3, 5, n // 4]      # noqa: E731


def compute(e, lib=None):
    """The output of case `e` by the oracle (lib None) or by the kernel library behind the C ABI."""
    case, seed = e["case"], e["seed"]
    if case.startswith("additive"):
        m = e["m"]
        n = 1 << m
        basis, shift = _subspace(m, e["kind"], seed)
        if case == "additive_fft":
            c = _gf(seed, e["ncoeffs"])
            return lib.additive_FFT(c, basis, shift) if lib else oracle.additive_fft(c, basis, shift)
        if case == "additive_ifft":
            v = _gf(seed, n)
            return lib.additive_IFFT(v, basis, shift) if lib else oracle.additive_ifft(v, basis, shift)
        if case == "additive_ifft_known_degree":
            d = e["degree"]
            c = _gf(seed, d)
            if lib:
                return lib.IFFT_of_known_degree(lib.additive_FFT(c, basis, shift), d, basis, shift)
            return oracle.additive_ifft_known_degree(oracle.additive_fft(c, basis, shift), d, basis, shift)
        if case == "additive_fold":
            f = _gf(seed, n)
            x = _element_by_index(basis, shift, n // 3) if e["x_in_domain"] else _gf(seed + 3, 1)[0]
            cs = e["coset_size"]
            return lib.evaluate_next_f_i_over_entire_domain(f, basis, shift, cs, x) if lib else oracle.fri_fold_additive(f, basis, shift, cs, x)
        if case == "additive_ldt_combine":
            degrees = LDT_DEGREES(n)
            coeffs = _gf(seed + 4, 2 * len(degrees))
            evals = [_gf(seed + 10 + k, n) for k in range(len(degrees))]
            return lib.ldt_combine(evals, degrees, coeffs, basis, shift) if lib else oracle.ldt_combine_additive(evals, degrees, coeffs, basis, shift)
    if case.startswith("multiplicative"):
        m = e["m"]
        n = 1 << m
        shift = _coset_shift(e["kind"], seed)
        if case == "multiplicative_fft":
            c = _fp(seed, e["ncoeffs"])
            return lib.multiplicative_FFT(c, m, shift) if lib else oracle.multiplicative_fft(c, n, shift)
        if case == "multiplicative_ifft":
            v = _fp(seed, n)
            return lib.multiplicative_IFFT(v, shift) if lib else oracle.multiplicative_ifft(v, shift)
        if case == "multiplicative_ifft_known_degree":
            d = e["degree"]
            c = _fp(seed, d)
            if lib:
                return lib.multiplicative_IFFT_of_known_degree(lib.multiplicative_FFT(c, m, shift), d, shift)
            return oracle.multiplicative_ifft_known_degree(oracle.multiplicative_fft(c, n, shift), d, shift)
        if case == "multiplicative_fold":
            f, x, cs = _fp(seed, n), _fp(seed + 3, 1)[0], e["coset_size"]
            return lib.multiplicative_evaluate_next_f_i(f, shift, cs, x) if lib else oracle.fri_fold_multiplicative(f, shift, cs, x)
        if case == "multiplicative_ldt_combine":
            degrees = LDT_DEGREES(n)
            coeffs = _fp(seed + 4, 2 * len(degrees))
            evals = [_fp(seed + 10 + k, n) for k in range(len(degrees))]
            if lib:
                return lib.ldt_combine_multiplicative(evals, degrees, coeffs, m, libiop_amd.edwards_subgroup_generator(m), shift)
            return oracle.ldt_combine_fp(evals, degrees, coeffs, n, shift)
    if case == "merkle_root":
        additive = e["field"] == "gf192"
        gen = _gf if additive else _fp
        cols = [gen(seed + k, 1 << e["log_n"]) for k in range(e["oracles"])]
        if lib:
            return lib.merkle_tree(cols, e["coset_size"], libiop_amd.DOMAIN_ADDITIVE if additive else libiop_amd.DOMAIN_MULTIPLICATIVE)[0]
        return oracle.merkle_build(cols, e["coset_size"], additive=additive)[0]
    if case == "merkle_root_zk":
        additive = e["field"] == "gf192"
        gen = _gf if additive else _fp
        n, cs = 1 << e["log_n"], e["coset_size"]
        cols = [gen(seed + k, n) for k in range(e["oracles"])]
        leaves = n // cs
        words = r1cs._splitmix64(seed + 100, np.arange(leaves * e["salt_bytes"] // 8, dtype=np.uint64))       # the interposed randombytes_buf of reference_vectors.cpp
        salts = np.frombuffer(words.astype("<u8").tobytes(), dtype=np.uint8).reshape(leaves, e["salt_bytes"])
        if lib:
            return lib.merkle_tree(cols, cs, libiop_amd.DOMAIN_ADDITIVE if additive else libiop_amd.DOMAIN_MULTIPLICATIVE, salts=salts)[0]
        return oracle.merkle_build(cols, cs, additive=additive, salts=salts)[0]
    if case.startswith("poseidon"):
        import json as _json
        with open(os.path.join(ROOT, "libiop_amd", "data", "poseidon_alt_bn128.json")) as f:
            d = _json.load(f)["sets"][e["set"]]
        params = libiop_amd.PoseidonParams.from_dict(d) if lib else oracle.PoseidonParams(d)

        def bn(sd, count):                  # four stream words per element, reduced mod r, Montgomery form
            w = r1cs._splitmix64(sd, np.arange(4 * count, dtype=np.uint64)).reshape(count, 4)
            return oracle.bn_from_ints([(int(a) | (int(b) << 64) | (int(c) << 128) | (int(dd) << 192)) % oracle.BN128_R for a, b, c, dd in w])
        if case == "poseidon_merkle_root":
            cols = [bn(seed + k, 1 << e["log_n"]) for k in range(e["oracles"])]
            if lib:
                return lib.merkle_tree_poseidon(params, cols, e["coset_size"], libiop_amd.DOMAIN_MULTIPLICATIVE, None)[0]
            return oracle.poseidon_merkle(params, cols, e["coset_size"], False, None)[0]
        if case == "poseidon_pow":
            challenge = bn(seed, 1)[0]
            return lib.solve_pow(challenge, e["bitlen"], params) if lib else oracle.pow_solve_poseidon(params, challenge, e["bitlen"])[0]
    if case == "hashchain":
        # the product's host-side hashchain (libiop_amd/host.py + the field's squeeze of domains.py): the script of reference_vectors.cpp hashchain_case.
        # The oracle's Python binding exposes the binary arm only: it is checked for gf192, the prime field through the provers' transcripts.
        from libiop_amd import host
        field = GF if e["field"] == "gf192" else FP
        if lib is None and e["field"] == "gf192":
            hc = oracle.Hashchain()
            sq = lambda n: hc.squeeze(n, 3)                     # noqa: E731
            absorb = lambda: hc.absorb(b"\0" * 32)              # noqa: E731
            pos = hc.squeeze_query_positions
            root = lambda: hashlib.blake2b(hc.squeeze(1, 3).tobytes(), digest_size=32).digest()      # noqa: E731  (blake2b.tcc:105-110)
        else:
            hc = host.Blake2bHashchain()
            sq = lambda n: field.squeeze(hc, n)                 # noqa: E731
            absorb = lambda: hc.absorb(b"")                     # noqa: E731
            pos = hc.squeeze_query_positions
            root = hc.squeeze_root_type if e["field"] == "gf192" else (lambda: hashlib.blake2b(field.squeeze(hc, 1).tobytes(), digest_size=32).digest())
        out = b""
        out += sq(3).tobytes()
        absorb()
        out += sq(1).tobytes()
        out += sq(40).tobytes()
        absorb()
        out += b"".join(int(p).to_bytes(8, "little") for p in pos(6, 1 << 12))
        out += root()
        out += b"".join(int(p).to_bytes(8, "little") for p in pos(3, 1 << 25))
        out += sq(2).tobytes()
        return np.frombuffer(out, dtype=np.uint8)
    if case == "pow":
        challenge = _gf(seed, 2).tobytes()[:32]
        answer = lib.solve_pow(challenge, e["bitlen"]) if lib else oracle.pow_solve_blake2b(challenge, e["bitlen"])[0]
        return np.frombuffer(bytes(answer), dtype=np.uint8)
    raise KeyError(case)


def check(e, lib=None):
    out = compute(e, lib)
    got = bytes(np.ascontiguousarray(out).tobytes()).hex() if (e["case"].startswith("merkle_root") or e["case"].startswith("poseidon")) else _digest(out)
    assert got == e["digest"], (ident(e), "differs from what libiop's own function produced")
