"""libiop_amd — MI355X (gfx950) prover hot path for libiop-style IOP SNARKs.

Python binding (ctypes) over the C ABI declared in ``include/libiop_amd.h``.  The shared library
``libiop_amd/lib/libiop_amd.so`` is built from the HIP sources in ``libiop_amd/csrc`` by
``libiop_amd.build.build()`` (``hipcc --offload-arch=gfx950``).  There is no CPU fallback: if the
library is missing, or no HIP device is visible, every compute call raises.

Field elements travel as ``numpy.uint64`` arrays of shape ``(count, 3)`` — libff::gf192's raw words.
The function names mirror the reference's operator API for this path:

    additive_FFT / additive_IFFT                         libiop/algebra/fft.hpp:28-38
    IFFT_of_known_degree_over_field_subset (additive)    libiop/algebra/fft.tcc:458-475
    evaluate_next_f_i_over_entire_domain                 libiop/protocols/ldt/fri/fri_aux.hpp:23-28
    merkle_tree::construct_with_leaves_serialized_by_cosets / get_root   libiop/bcs/merkle_tree.hpp:88-104
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libiop_amd.so")

IOPX_OK = 0
IOPX_ERR_INVALID_ARGUMENT = -1
IOPX_ERR_LOGIC = -2
IOPX_ERR_RUNTIME = -3
IOPX_ERR_NO_DEVICE = -4

DOMAIN_ADDITIVE = 0
DOMAIN_MULTIPLICATIVE = 1

# every symbol include/libiop_amd.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = [
    "iopx_version", "iopx_last_error", "iopx_device_count", "iopx_init", "iopx_set_stream", "iopx_use_own_stream", "iopx_synchronize",
    "iopx_malloc", "iopx_free", "iopx_memcpy_h2d", "iopx_memcpy_d2h", "iopx_clear_plans",
    "iopx_add_fft_gf192_dev", "iopx_add_fft_gf192", "iopx_add_lde_gf192_dev", "iopx_add_taylor_gf192_dev", "iopx_gf192_pow_table_dev", "iopx_add_combine_gf192_dev", "iopx_add_ifft_gf192_dev", "iopx_add_ifft_gf192", "iopx_add_ifft_gf192_batch_dev", "iopx_add_lde_gf192_batch_dev",
    "iopx_fri_fold_add_gf192_dev", "iopx_fri_fold_add_gf192",
    "iopx_mul_fft_fp3_dev", "iopx_mul_fft_fp3", "iopx_mul_ifft_fp3_dev", "iopx_mul_ifft_fp3",
    "iopx_mul_ifft_known_degree_fp3_dev", "iopx_fri_fold_mul_fp3_dev", "iopx_fri_fold_mul_fp3",
    "iopx_merkle_blake2b_dev", "iopx_merkle_blake2b", "iopx_merkle_leaves_blake2b_dev", "iopx_merkle_inner_blake2b_dev",
    "iopx_bn128_to_montgomery_dev", "iopx_poseidon_permute_bn128_dev", "iopx_merkle_poseidon_bn128_dev", "iopx_merkle_poseidon_bn128",
    "iopx_pow_solve_blake2b", "iopx_pow_solve_poseidon_bn128", "iopx_ldt_combine_gf192_dev", "iopx_ldt_combine_fp3_dev",
    "iopx_merkle_membership_proof_dev", "iopx_query_responses_dev", "iopx_rowcheck_gf192_dev", "iopx_rowcheck_fp3_dev",
    "iopx_fz_gf192_dev", "iopx_fz_fp3_dev", "iopx_sumcheck_g_gf192_dev", "iopx_sumcheck_g_fp3_dev",
    "iopx_lincheck_gf192_dev", "iopx_lincheck_fp3_dev",
    "iopx_gf192_mul_dev", "iopx_gf192_mul_uniform_dev", "iopx_profile_begin", "iopx_profile_report",
    "iopx_spmv_gf192_dev", "iopx_spmv_fp3_dev", "iopx_poly_div_vanishing_gf192_dev", "iopx_poly_div_vanishing_fp3_dev",
    "iopx_lincomb_gf192_dev", "iopx_lincomb_fp3_dev", "iopx_gf192_add_dev", "iopx_gf192_inv_dev", "iopx_fp3_mul_dev", "iopx_fp3_sub_dev",
    "iopx_fp3_inv_dev", "iopx_fp3_pow_table_dev",
    "iopx_lincomb_affine_gf192_dev", "iopx_lincomb_affine_fp3_dev", "iopx_gf192_div_dev", "iopx_fp3_div_dev", "iopx_domain_offsets_gf192_dev",
    "iopx_domain_offsets_fp3_dev", "iopx_vanishing_evals_gf192_dev", "iopx_vanishing_evals_fp3_dev", "iopx_rational_combine_gf192_dev",
    "iopx_rational_combine_fp3_dev", "iopx_rational_sumcheck_constraint_gf192_dev", "iopx_rational_sumcheck_constraint_fp3_dev",
    "iopx_gf192_vanishing_host", "iopx_gf192_inverse_host",
    "iopx_fp3_subgroup_generator", "iopx_fp3_multiplicative_generator", "iopx_fp3_host_mul", "iopx_fp3_host_pow", "iopx_poseidon_shipped_params", "iopx_fri_domains_gf192", "iopx_add_reextend_gf192_batch_dev", "iopx_add_reextend_lde_gf192_batch_dev", "iopx_defer_downloads_begin", "iopx_defer_downloads_end",
    "iopx_pool_alloc", "iopx_pool_free", "iopx_memcpy_d2d", "iopx_memset_dev", "iopx_upload_small", "iopx_gather_dev", "iopx_scatter_dev", "iopx_gather_stride_dev", "iopx_count_mismatch_dev", "iopx_div_by_vanishing_gf192_dev", "iopx_add_reextend2_gf192_batch_dev",
    "iopx_transfer_stats", "iopx_blake2b_host", "iopx_gf192_host_mul", "iopx_fp3_host_add", "iopx_fp3_host_sub", "iopx_fp3_host_inverse",
    "iopx_fp3_from_uint", "iopx_fp3_modulus", "iopx_pow_search_blake2b", "iopx_pow_search_blake2b_begin", "iopx_pow_search_blake2b_end", "iopx_pow_candidate_blake2b",
    "iopx_side_stream_begin", "iopx_side_stream_end", "iopx_side_stream_join",
    "iopx_aurora_instance_create", "iopx_aurora_example_instance_create", "iopx_aurora_prove", "iopx_aurora_instance_warm", "iopx_aurora_instance_free", "iopx_host_free",
    "iopx_fractal_index", "iopx_fractal_prove",
    "iopx_memcpy_d2h_deferrable", "iopx_comm_rccl_unique_id", "iopx_comm_create_rccl", "iopx_comm_create_callbacks", "iopx_comm_create_replay", "iopx_comm_is_replay", "iopx_cold_stats", "iopx_cold_add", "iopx_gf192_mul_halves_dev", "iopx_mul_fft_fp3_windows_dev", "iopx_set_option", "iopx_clear_option", "iopx_get_option", "iopx_comm_destroy", "iopx_comm_rank",
    "iopx_comm_all_gather_dev", "iopx_comm_all_reduce_u64_dev", "iopx_comm_broadcast_dev", "iopx_comm_all_to_all_dev", "iopx_comm_sendrecv_dev", "iopx_comm_stats", "iopx_comm_bind_transforms", "iopx_add_taylor_inv_gf192_dev", "iopx_add_combine_inv_gf192_dev",
    "iopx_interleave_dev", "iopx_gather_rows_dev", "iopx_fri_snark_prove", "iopx_fri_snark_prove_dist", "iopx_add_fft_gf192_dist_dev", "iopx_add_ifft_gf192_dist_dev", "iopx_aurora_prove_dist", "iopx_fractal_index_dist", "iopx_fractal_prove_dist",
]


class NoDeviceError(RuntimeError):
    pass


def _bind_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's) and looks it up by
    file name, so if this library pulled in the system copy first, a later `import torch` would bring up a second runtime and
    find no GPU.  Loading torch's copy first makes both bind to that one instance, whichever package is imported first; without
    PyTorch installed the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


_u64p = ctypes.POINTER(ctypes.c_uint64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_sz = ctypes.c_size_t
_vp = ctypes.c_void_p


def _as_u64(a, cols=3):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, cols)
    if a.shape[-1] != cols:
        raise ValueError("expected (count, %d) uint64 words" % cols)
    return a


class _PoseidonParamsC(ctypes.Structure):
    _fields_ = [("alpha", _sz), ("full_rounds", _sz), ("partial_rounds", _sz), ("rate", _sz), ("state_size", _sz),
                ("near_mds", ctypes.c_int), ("ark", _u64p), ("mds", _u64p)]


def _ints_to_words4(values):
    return np.array([[(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in values], dtype=np.uint64).reshape(-1, 4)


class PoseidonParams:
    """poseidon_params<FieldT> (libiop/bcs/hashing/poseidon.hpp:20-60) for alt_bn128 Fr, as the C ABI takes it.

    ``PoseidonParams.shipped(name)`` loads one of the sets the reference ships (poseidon.tcc:311-520, chosen by
    hash_enum.tcc:73-110): "starkware_alpha5_t3", "high_alpha17_t3", "high_alpha17_t4"."""

    def __init__(self, alpha, full_rounds, partial_rounds, rate, state_size, near_mds, ark, mds):
        self.alpha, self.full_rounds, self.partial_rounds = int(alpha), int(full_rounds), int(partial_rounds)
        self.rate, self.state_size, self.near_mds = int(rate), int(state_size), bool(near_mds)
        self.ark = _ints_to_words4([v for row in ark for v in row])
        self.mds = _ints_to_words4([v for row in mds for v in row]) if mds else None
        self.c = _PoseidonParamsC(self.alpha, self.full_rounds, self.partial_rounds, self.rate, self.state_size, int(self.near_mds),
                                  self.ark.ctypes.data_as(_u64p), self.mds.ctypes.data_as(_u64p) if self.mds is not None else None)

    @classmethod
    def from_dict(cls, d):
        return cls(d["alpha"], d["full_rounds"], d["partial_rounds"], d["rate"], d["state_size"], d["near_mds"], d["ark"], d.get("mds"))

    @classmethod
    def shipped(cls, name):
        import json
        with open(os.path.join(_HERE, "data", "poseidon_alt_bn128.json")) as f:
            return cls.from_dict(json.load(f)["sets"][name])


class Library:
    """One loaded instance of the C ABI.  ``Library()`` loads the product library."""

    def __init__(self, path=None):
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "libiop_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        _bind_hip_runtime()
        self.c = ctypes.CDLL(path)
        c = self.c
        c.iopx_last_error.restype = ctypes.c_char_p
        c.iopx_malloc.argtypes = [ctypes.POINTER(_vp), _sz]
        c.iopx_free.argtypes = [_vp]
        c.iopx_memcpy_h2d.argtypes = [_vp, _vp, _sz]
        c.iopx_memcpy_d2h.argtypes = [_vp, _vp, _sz]
        c.iopx_set_stream.argtypes = [_vp]
        c.iopx_add_fft_gf192_dev.argtypes = [_vp, _sz, _u64p, _sz, _u64p, _vp]
        c.iopx_add_fft_gf192.argtypes = [_u64p, _sz, _u64p, _sz, _u64p, _u64p]
        c.iopx_add_lde_gf192_dev.argtypes = [_vp, _sz, _u64p, _sz, _u64p, _sz, _sz, _vp]
        c.iopx_add_taylor_gf192_dev.argtypes = [_vp, _sz, _vp]
        c.iopx_add_taylor_inv_gf192_dev.argtypes = [_vp, _sz, _vp]
        c.iopx_add_combine_inv_gf192_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, _u64p, _sz, _u64p, ctypes.c_int]
        c.iopx_gf192_pow_table_dev.argtypes = [_vp, _sz, _u64p, _u64p]
        c.iopx_add_combine_gf192_dev.argtypes = [_vp, _vp, _vp, _sz, _sz, _u64p, _sz, _u64p, ctypes.c_int]
        c.iopx_add_ifft_gf192_dev.argtypes = [_vp, _u64p, _sz, _u64p, _vp]
        c.iopx_add_ifft_gf192.argtypes = [_u64p, _u64p, _sz, _u64p, _u64p]
        c.iopx_add_ifft_gf192_batch_dev.argtypes = [_vp, _sz, _u64p, _sz, _u64p, _vp]
        c.iopx_add_lde_gf192_batch_dev.argtypes = [ctypes.POINTER(_vp), _sz, _sz, _u64p, _sz, _u64p, _sz, _sz, ctypes.POINTER(_vp)]
        c.iopx_fri_fold_add_gf192_dev.argtypes = [_vp, _u64p, _sz, _u64p, _sz, _u64p, _vp]
        c.iopx_fri_fold_add_gf192.argtypes = [_u64p, _u64p, _sz, _u64p, _sz, _u64p, _u64p]
        c.iopx_mul_fft_fp3_dev.argtypes = [_vp, _sz, _sz, _u64p, _u64p, _vp]
        c.iopx_mul_fft_fp3.argtypes = [_u64p, _sz, _sz, _u64p, _u64p, _u64p]
        c.iopx_mul_ifft_fp3_dev.argtypes = [_vp, _sz, _u64p, _u64p, _vp]
        c.iopx_mul_ifft_fp3.argtypes = [_u64p, _sz, _u64p, _u64p, _u64p]
        c.iopx_mul_ifft_known_degree_fp3_dev.argtypes = [_vp, _sz, _sz, _u64p, _u64p, _vp]
        c.iopx_fri_fold_mul_fp3_dev.argtypes = [_vp, _sz, _u64p, _u64p, _sz, _u64p, _vp]
        c.iopx_fri_fold_mul_fp3.argtypes = [_u64p, _sz, _u64p, _u64p, _sz, _u64p, _u64p]
        c.iopx_merkle_blake2b_dev.argtypes = [ctypes.POINTER(_vp), _sz, _sz, _sz, _sz, ctypes.c_int, _vp, _sz, _vp]
        c.iopx_merkle_blake2b.argtypes = [ctypes.POINTER(_vp), _sz, _sz, _sz, _sz, ctypes.c_int, _vp, _sz, _vp]
        c.iopx_merkle_leaves_blake2b_dev.argtypes = [ctypes.POINTER(_vp), _sz, _sz, _sz, _sz, ctypes.c_int, _vp, _sz, _vp]
        c.iopx_merkle_inner_blake2b_dev.argtypes = [_vp, _sz]
        pp = ctypes.POINTER(_PoseidonParamsC)
        c.iopx_bn128_to_montgomery_dev.argtypes = [_vp, _vp, _sz]
        c.iopx_poseidon_permute_bn128_dev.argtypes = [pp, _vp, _sz]
        c.iopx_merkle_poseidon_bn128_dev.argtypes = [pp, ctypes.POINTER(_vp), _sz, _sz, _sz, ctypes.c_int, _vp, _vp]
        c.iopx_merkle_poseidon_bn128.argtypes = [pp, ctypes.POINTER(_vp), _sz, _sz, _sz, ctypes.c_int, _vp, _vp]
        c.iopx_pow_solve_blake2b.argtypes = [_vp, _sz, _vp]
        c.iopx_rowcheck_gf192_dev.argtypes = [_vp, _vp, _vp, _u64p, _sz, _u64p, _sz, _u64p, _vp]
        c.iopx_rowcheck_fp3_dev.argtypes = [_vp, _vp, _vp, _sz, _u64p, _u64p, _sz, _u64p, _vp]
        c.iopx_fz_gf192_dev.argtypes = [_vp, _vp, _u64p, _sz, _u64p, _u64p, _sz, _u64p, _vp]
        c.iopx_fz_fp3_dev.argtypes = [_vp, _vp, _sz, _u64p, _u64p, _sz, _u64p, _vp]
        c.iopx_sumcheck_g_gf192_dev.argtypes = [_vp, _vp, _u64p, _sz, _u64p, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_sumcheck_g_fp3_dev.argtypes = [_vp, _vp, _sz, _u64p, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_lincheck_gf192_dev.argtypes = [_vp, ctypes.POINTER(_vp), _sz, _u64p, _vp, _vp, _sz, _vp]
        c.iopx_lincheck_fp3_dev.argtypes = [_vp, ctypes.POINTER(_vp), _sz, _u64p, _vp, _vp, _sz, _vp]
        c.iopx_merkle_membership_proof_dev.argtypes = [_vp, _sz, ctypes.POINTER(_sz), _sz, _vp, _sz, ctypes.POINTER(_sz)]
        c.iopx_query_responses_dev.argtypes = [ctypes.POINTER(_vp), _sz, _sz, _sz, ctypes.POINTER(_sz), _sz, _vp]
        c.iopx_ldt_combine_gf192_dev.argtypes = [ctypes.POINTER(_vp), _sz, ctypes.POINTER(_sz), _u64p, _u64p, _sz, _u64p, _vp]
        c.iopx_ldt_combine_fp3_dev.argtypes = [ctypes.POINTER(_vp), _sz, ctypes.POINTER(_sz), _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_pow_solve_poseidon_bn128.argtypes = [pp, _vp, _sz, _vp]
        c.iopx_gf192_mul_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_gf192_mul_uniform_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_spmv_gf192_dev.argtypes = [_vp, _vp, _vp, _sz, _vp, _u64p, ctypes.c_int, _vp]
        c.iopx_spmv_fp3_dev.argtypes = [_vp, _vp, _vp, _sz, _vp, _u64p, ctypes.c_int, _vp]
        c.iopx_poly_div_vanishing_gf192_dev.argtypes = [_vp, _sz, _u64p, _sz, _u64p, _vp]
        c.iopx_poly_div_vanishing_fp3_dev.argtypes = [_vp, _sz, _sz, _u64p, _vp]
        c.iopx_lincomb_gf192_dev.argtypes = [ctypes.POINTER(_vp), _sz, _u64p, _sz, _vp]
        c.iopx_lincomb_fp3_dev.argtypes = [ctypes.POINTER(_vp), _sz, _u64p, _sz, _vp]
        c.iopx_gf192_add_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_gf192_inv_dev.argtypes = [_vp, _vp, _sz]
        c.iopx_fp3_mul_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_fp3_sub_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_fp3_inv_dev.argtypes = [_vp, _vp, _sz]
        c.iopx_fp3_pow_table_dev.argtypes = [_vp, _sz, _u64p, _u64p]
        pv = ctypes.POINTER(_vp)
        c.iopx_lincomb_affine_gf192_dev.argtypes = [pv, _sz, _u64p, _u64p, _sz, _vp]
        c.iopx_lincomb_affine_fp3_dev.argtypes = [pv, _sz, _u64p, _u64p, _sz, _vp]
        c.iopx_gf192_div_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_fp3_div_dev.argtypes = [_vp, _vp, _vp, _sz]
        c.iopx_domain_offsets_gf192_dev.argtypes = [_u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_domain_offsets_fp3_dev.argtypes = [_sz, _u64p, _u64p, _u64p, _vp]
        c.iopx_vanishing_evals_gf192_dev.argtypes = [_u64p, _sz, _u64p, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_vanishing_evals_fp3_dev.argtypes = [_sz, _u64p, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_rational_combine_gf192_dev.argtypes = [pv, pv, _sz, _u64p, _sz, _vp, _vp]
        c.iopx_rational_combine_fp3_dev.argtypes = [pv, pv, _sz, _u64p, _sz, _vp, _vp]
        c.iopx_rational_sumcheck_constraint_gf192_dev.argtypes = [_vp, _vp, _vp, _vp, _u64p, _sz, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_rational_sumcheck_constraint_fp3_dev.argtypes = [_vp, _vp, _vp, _sz, _u64p, _u64p, _sz, _u64p, _u64p, _vp]
        c.iopx_gf192_vanishing_host.argtypes = [_u64p, _sz, _u64p, _u64p, _u64p, _u64p]
        c.iopx_gf192_inverse_host.argtypes = [_u64p, _u64p]

    # ---- error translation (the exception types the reference throws) ----
    def _check(self, rc):
        if rc == IOPX_OK:
            return
        msg = (self.c.iopx_last_error() or b"").decode()
        if rc == IOPX_ERR_INVALID_ARGUMENT:
            raise ValueError(msg)                 # std::invalid_argument
        if rc == IOPX_ERR_LOGIC:
            raise AssertionError(msg)             # std::logic_error
        if rc == IOPX_ERR_NO_DEVICE:
            raise NoDeviceError(msg)
        raise RuntimeError(msg)                   # std::runtime_error

    # ---- runtime ----
    def version(self):
        return self.c.iopx_version()

    def device_count(self):
        return self.c.iopx_device_count()

    def init(self, device=0):
        self._check(self.c.iopx_init(device))

    def set_stream(self, hip_stream):
        """All later work goes to this hipStream_t, taken as given: 0 is the legacy default stream (torch's default stream)."""
        self._check(self.c.iopx_set_stream(_vp(hip_stream)))
        self._stream_handle = int(hip_stream or 0)

    def use_own_stream(self):
        self._check(self.c.iopx_use_own_stream())
        self._stream_handle = None

    def shares_stream_with(self, torch, device=None):
        """True when the library enqueues on torch's current stream (set_stream(torch.cuda.current_stream().cuda_stream)): library
        calls, torch ops and torch.distributed collectives are then ordered by the stream and need no host synchronisation."""
        h = getattr(self, "_stream_handle", None)
        return h is not None and torch.cuda.is_available() and int(torch.cuda.current_stream(device).cuda_stream) == h

    def synchronize(self):
        self._check(self.c.iopx_synchronize())

    def clear_plans(self):
        self._check(self.c.iopx_clear_plans())

    def malloc(self, nbytes):
        p = _vp()
        self._check(self.c.iopx_malloc(ctypes.byref(p), nbytes))
        return p.value

    def free(self, dptr):
        self._check(self.c.iopx_free(_vp(dptr)))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._check(self.c.iopx_memcpy_h2d(_vp(dptr), _vp(arr.ctypes.data), arr.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self.c.iopx_memcpy_d2h(_vp(arr.ctypes.data), _vp(dptr), arr.nbytes))

    def read_digest(self, d_nodes, index=0):
        """32-byte node `index` of a device-resident tree, copied on the library's stream (ordered after the kernels that
        produce it — a torch `.cpu()` on another stream is not)."""
        out = np.empty(32, dtype=np.uint8)
        self.d2h(out, int(d_nodes) + 32 * index)
        return bytes(out)

    # ---- host-pointer operators (std::vector in / out, like the reference templates) ----
    def additive_FFT(self, poly_coeffs, basis, shift):
        """additive_FFT(poly_coeffs, affine_subspace(basis, shift)) — fft.tcc:39-124."""
        coeffs, basis, shift = _as_u64(poly_coeffs), _as_u64(basis), _as_u64(shift)
        m = basis.shape[0]
        out = np.empty((1 << m, 3), dtype=np.uint64)
        self._check(self.c.iopx_add_fft_gf192(coeffs.ctypes.data_as(_u64p), coeffs.shape[0], basis.ctypes.data_as(_u64p), m,
                                              shift.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def additive_IFFT(self, evals, basis, shift):
        """additive_IFFT(evals, affine_subspace(basis, shift)) — fft.tcc:126-204."""
        evals, basis, shift = _as_u64(evals), _as_u64(basis), _as_u64(shift)
        m = basis.shape[0]
        if evals.shape[0] != 1 << m:
            raise ValueError("additive IFFT: %d evaluations for a domain of size %d" % (evals.shape[0], 1 << m))
        out = np.empty((1 << m, 3), dtype=np.uint64)
        self._check(self.c.iopx_add_ifft_gf192(evals.ctypes.data_as(_u64p), basis.ctypes.data_as(_u64p), m,
                                               shift.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def IFFT_of_known_degree(self, evals, degree, basis, shift):
        """IFFT_of_known_degree_over_field_subset, additive overload — fft.tcc:458-475."""
        evals, basis = _as_u64(evals), _as_u64(basis)
        k = max(int(degree) - 1, 0).bit_length()
        return self.additive_IFFT(evals[: 1 << k], basis[:k], shift)

    def evaluate_next_f_i_over_entire_domain(self, f_i_evals, basis, shift, coset_size, x_i):
        """fri_aux.tcc:5-34 -> additive_evaluate_next_f_i_over_entire_domain (:36-103)."""
        f, basis, shift, x = _as_u64(f_i_evals), _as_u64(basis), _as_u64(shift), _as_u64(x_i)
        m = basis.shape[0]
        if f.shape[0] != 1 << m:
            raise ValueError("f_i has %d evaluations for a domain of size %d" % (f.shape[0], 1 << m))
        out = np.empty(((1 << m) // max(int(coset_size), 1), 3), dtype=np.uint64)
        self._check(self.c.iopx_fri_fold_add_gf192(f.ctypes.data_as(_u64p), basis.ctypes.data_as(_u64p), m,
                                                   shift.ctypes.data_as(_u64p), int(coset_size), x.ctypes.data_as(_u64p),
                                                   out.ctypes.data_as(_u64p)))
        return out

    # ---- multiplicative cosets over the 181-bit prime field (Montgomery words) ----
    def multiplicative_FFT(self, poly_coeffs, log_n, shift, gen=None):
        """multiplicative_FFT(poly_coeffs, multiplicative_coset(2^log_n, shift)) — fft.tcc:236-317."""
        coeffs, shift = _as_u64(poly_coeffs), _as_u64(shift)
        gen = _as_u64(edwards_subgroup_generator(log_n) if gen is None else gen)
        out = np.empty((1 << log_n, 3), dtype=np.uint64)
        self._check(self.c.iopx_mul_fft_fp3(coeffs.ctypes.data_as(_u64p), coeffs.shape[0], log_n, gen.ctypes.data_as(_u64p),
                                            shift.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def multiplicative_IFFT(self, evals, shift, gen=None):
        """multiplicative_IFFT(evals, coset) — fft.tcc:343-361 over libfqfft iFFT / icosetFFT."""
        evals, shift = _as_u64(evals), _as_u64(shift)
        n = evals.shape[0]
        log_n = n.bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("multiplicative IFFT: %d evaluations is not a power of two" % n)
        gen = _as_u64(edwards_subgroup_generator(log_n) if gen is None else gen)
        out = np.empty_like(evals)
        self._check(self.c.iopx_mul_ifft_fp3(evals.ctypes.data_as(_u64p), log_n, gen.ctypes.data_as(_u64p),
                                             shift.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def multiplicative_IFFT_of_known_degree(self, evals, degree, shift):
        """IFFT_of_known_degree_over_field_subset, multiplicative overload — fft.tcc:435-456."""
        evals, shift = _as_u64(evals), _as_u64(shift)
        n = evals.shape[0]
        log_n = n.bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("multiplicative IFFT of known degree: %d evaluations is not a power of two" % n)
        k = max(int(degree) - 1, 0).bit_length()
        gen = _as_u64(edwards_subgroup_generator(log_n))
        d_in, d_out = self.malloc(evals.nbytes), self.malloc(24 << k)
        try:
            self.h2d(d_in, evals)
            self._check(self.c.iopx_mul_ifft_known_degree_fp3_dev(_vp(d_in), int(degree), log_n, gen.ctypes.data_as(_u64p),
                                                                  shift.ctypes.data_as(_u64p), _vp(d_out)))
            out = np.empty((1 << k, 3), dtype=np.uint64)
            self.d2h(out, d_out)
        finally:
            self.free(d_in)
            self.free(d_out)
        return out

    def multiplicative_evaluate_next_f_i(self, f_i_evals, shift, coset_size, x_i, gen=None):
        """multiplicative_evaluate_next_f_i_over_entire_domain — fri_aux.tcc:106-249."""
        f, shift, x = _as_u64(f_i_evals), _as_u64(shift), _as_u64(x_i)
        n = f.shape[0]
        log_n = n.bit_length() - 1
        if n != 1 << log_n:
            raise ValueError("multiplicative fold: %d evaluations is not a power of two" % n)
        gen = _as_u64(edwards_subgroup_generator(log_n) if gen is None else gen)
        out = np.empty((n // max(int(coset_size), 1), 3), dtype=np.uint64)
        self._check(self.c.iopx_fri_fold_mul_fp3(f.ctypes.data_as(_u64p), log_n, gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p),
                                                 int(coset_size), x.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def merkle_tree(self, oracles, coset_size, domain_type=DOMAIN_ADDITIVE, salts=None):
        """construct_with_leaves_serialized_by_cosets + compute_inner_nodes (merkle_tree.tcc:92-229).
        Returns the (2L-1, 32) uint8 node array in heap order; row 0 is get_root()."""
        oracles = [np.ascontiguousarray(o, dtype=np.uint64) for o in oracles]
        n, w = oracles[0].shape
        for o in oracles:
            if o.shape != (n, w):
                raise AssertionError("Attempting to construct a Merkle tree with a constituent vector of wrong size")
        cs = int(coset_size)
        L = n // cs if cs > 0 else 0
        nodes = np.zeros((max(2 * L - 1, 1), 32), dtype=np.uint8)
        ptrs = (_vp * len(oracles))(*[o.ctypes.data for o in oracles])
        if salts is not None:
            salts = np.ascontiguousarray(salts, dtype=np.uint8)
            if salts.ndim != 2 or salts.shape[0] != L:
                raise ValueError("zk salts: expected one row per leaf (%d), got shape %s" % (L, salts.shape))
            sp, sb = _vp(salts.ctypes.data), salts.shape[1]
        else:
            sp, sb = _vp(0), 0
        self._check(self.c.iopx_merkle_blake2b(ptrs, len(oracles), 8 * w, n, cs, int(domain_type), sp, sb, _vp(nodes.ctypes.data)))
        return nodes

    # ---- Poseidon over alt_bn128 Fr: elements are (count, 4) uint64 Montgomery words ----
    def bn128_to_montgomery(self, values):
        """FieldT(bigint) for a list of Python ints / a (count, 4) array of canonical words."""
        a = _as_u64(values, 4) if isinstance(values, np.ndarray) else _ints_to_words4(values)
        n = a.shape[0]
        d = self.malloc(max(a.nbytes, 8))
        try:
            self.h2d(d, a)
            self._check(self.c.iopx_bn128_to_montgomery_dev(_vp(d), _vp(d), n))
            out = np.empty_like(a)
            self.d2h(out, d)
        finally:
            self.free(d)
        return out

    def poseidon_permute(self, params, states):
        """poseidon::apply_permutation on (count, state_size, 4) Montgomery states."""
        st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, params.state_size, 4).copy()
        d = self.malloc(max(st.nbytes, 8))
        try:
            self.h2d(d, st)
            self._check(self.c.iopx_poseidon_permute_bn128_dev(ctypes.byref(params.c), _vp(d), st.shape[0]))
            self.d2h(st, d)
        finally:
            self.free(d)
        return st

    def merkle_tree_poseidon(self, params, oracles, coset_size, domain_type=DOMAIN_MULTIPLICATIVE, salts=None):
        """merkle_tree<FieldT, FieldT> with the algebraic leaf / two-to-one hashes (algebraic_sponge.tcc:220-265).
        Returns the (2L-1, 4) uint64 node array (Montgomery words), row 0 = get_root()."""
        oracles = [_as_u64(o, 4) for o in oracles]
        n = oracles[0].shape[0]
        for o in oracles:
            if o.shape[0] != n:
                raise AssertionError("Attempting to construct a Merkle tree with a constituent vector of wrong size")
        cs = int(coset_size)
        L = n // cs if cs > 0 else 0
        nodes = np.zeros((max(2 * L - 1, 1), 4), dtype=np.uint64)
        ptrs = (_vp * len(oracles))(*[o.ctypes.data for o in oracles])
        if salts is not None:
            salts = np.ascontiguousarray(salts, dtype=np.uint8).reshape(-1, 32)
            if salts.shape[0] != L:
                raise ValueError("zk salts: expected %d rows of 32 bytes, got %d" % (L, salts.shape[0]))
            sp = _vp(salts.ctypes.data)
        else:
            sp = _vp(0)
        self._check(self.c.iopx_merkle_poseidon_bn128(ctypes.byref(params.c), ptrs, len(oracles), n, cs, int(domain_type), sp, _vp(nodes.ctypes.data)))
        return nodes

    def merkle_tree_poseidon_dev(self, params, d_oracles, n, coset_size, d_nodes, domain_type=DOMAIN_MULTIPLICATIVE, d_salts=0):
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        self._check(self.c.iopx_merkle_poseidon_bn128_dev(ctypes.byref(params.c), ptrs, len(d_oracles), n, int(coset_size), int(domain_type),
                                                          _vp(d_salts), _vp(d_nodes)))

    # ---- transcript extraction (merkle_tree.tcc:242-336, bcs_prover.tcc:187-197) ----
    def get_set_membership_proof_dev(self, d_nodes, num_leaves, positions):
        """Auxiliary hashes of merkle_tree::get_set_membership_proof as a (count, 32) uint8 array; d_nodes on the device."""
        self._refuse_inside_defer_window("get_set_membership_proof_dev")
        pos = (_sz * max(len(positions), 1))(*[int(p) for p in positions])
        cap = max(1, len(positions) * max(1, int(num_leaves).bit_length()))
        out = np.zeros((cap, 32), dtype=np.uint8)
        cnt = _sz(0)
        self._check(self.c.iopx_merkle_membership_proof_dev(_vp(d_nodes), int(num_leaves), pos, len(positions), _vp(out.ctypes.data), cap, ctypes.byref(cnt)))
        return out[:cnt.value].copy()

    def defer_downloads_begin(self):
        """Until defer_downloads_end(), iopx_query_responses_dev / iopx_merkle_membership_proof_dev only queue their read-backs: call them
        through self.c with buffers that stay alive until defer_downloads_end().  The array-returning wrappers get_set_membership_proof_dev /
        query_responses_dev would hand back unfilled arrays and let _end() write into freed memory, so they raise inside a window."""
        self._check(self.c.iopx_defer_downloads_begin())
        self._deferring = True

    def defer_downloads_end(self):
        try:
            self._check(self.c.iopx_defer_downloads_end())
        finally:
            self._deferring = False

    def _refuse_inside_defer_window(self, what):
        if getattr(self, "_deferring", False):
            raise RuntimeError("%s returns its result at once and cannot run inside a defer_downloads window: call the C entry point "
                               "with a buffer that outlives defer_downloads_end()" % what)

    def query_responses_dev(self, d_oracles, elem_bytes, n, positions):
        """values[p][k] = oracle_k[positions[p]] as a (positions, oracles, elem_bytes / 8) uint64 array."""
        self._refuse_inside_defer_window("query_responses_dev")
        pos = (_sz * max(len(positions), 1))(*[int(p) for p in positions])
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        out = np.zeros((len(positions), len(d_oracles), elem_bytes // 8), dtype=np.uint64)
        self._check(self.c.iopx_query_responses_dev(ptrs, len(d_oracles), int(elem_bytes), int(n), pos, len(positions), _vp(out.ctypes.data)))
        return out

    # ---- R1CS row check (rowcheck.tcc:16-88) ----
    def rowcheck_dev(self, d_az, d_bz, d_cz, basis, shift, constraint_dim, constraint_shift, d_out):
        basis, shift, cs = _as_u64(basis), _as_u64(shift), _as_u64(constraint_shift)
        self._check(self.c.iopx_rowcheck_gf192_dev(_vp(d_az), _vp(d_bz), _vp(d_cz), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                   shift.ctypes.data_as(_u64p), int(constraint_dim), cs.ctypes.data_as(_u64p), _vp(d_out)))

    def rowcheck_multiplicative_dev(self, d_az, d_bz, d_cz, log_n, gen, shift, constraint_log_order, constraint_shift, d_out):
        gen, shift, cs = _as_u64(gen), _as_u64(shift), _as_u64(constraint_shift)
        self._check(self.c.iopx_rowcheck_fp3_dev(_vp(d_az), _vp(d_bz), _vp(d_cz), int(log_n), gen.ctypes.data_as(_u64p),
                                                 shift.ctypes.data_as(_u64p), int(constraint_log_order), cs.ctypes.data_as(_u64p), _vp(d_out)))

    def rowcheck(self, az, bz, cz, basis, shift, constraint_dim, constraint_shift):
        """Host-array form of rowcheck_dev."""
        return self._ldt_combine_host([az, bz, cz], lambda d, o: self.rowcheck_dev(d[0], d[1], d[2], basis, shift, constraint_dim, constraint_shift, o))

    def rowcheck_multiplicative(self, az, bz, cz, log_n, gen, shift, constraint_log_order, constraint_shift):
        return self._ldt_combine_host([az, bz, cz], lambda d, o: self.rowcheck_multiplicative_dev(d[0], d[1], d[2], log_n, gen, shift,
                                                                                                 constraint_log_order, constraint_shift, o))

    # ---- fz virtual oracle (r1cs_rs_iop.tcc:181-222) ----
    def fz_dev(self, d_fw, d_f1v, basis, shift, input_basis, input_shift, d_out):
        basis, shift, ish = _as_u64(basis), _as_u64(shift), _as_u64(input_shift)
        ib = np.ascontiguousarray(input_basis, dtype=np.uint64).reshape(-1, 3)
        self._check(self.c.iopx_fz_gf192_dev(_vp(d_fw), _vp(d_f1v), basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p),
                                             ib.ctypes.data_as(_u64p), ib.shape[0], ish.ctypes.data_as(_u64p), _vp(d_out)))

    def fz_multiplicative_dev(self, d_fw, d_f1v, log_n, gen, shift, input_log_order, input_shift, d_out):
        gen, shift, ish = _as_u64(gen), _as_u64(shift), _as_u64(input_shift)
        self._check(self.c.iopx_fz_fp3_dev(_vp(d_fw), _vp(d_f1v), int(log_n), gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p),
                                           int(input_log_order), ish.ctypes.data_as(_u64p), _vp(d_out)))

    def fz(self, fw, f1v, basis, shift, input_basis, input_shift):
        return self._ldt_combine_host([fw, f1v], lambda d, o: self.fz_dev(d[0], d[1], basis, shift, input_basis, input_shift, o))

    def fz_multiplicative(self, fw, f1v, log_n, gen, shift, input_log_order, input_shift):
        return self._ldt_combine_host([fw, f1v], lambda d, o: self.fz_multiplicative_dev(d[0], d[1], log_n, gen, shift, input_log_order, input_shift, o))

    # ---- sumcheck g oracle (sumcheck.tcc:58-119) ----
    def sumcheck_g_dev(self, d_f, d_h, basis, shift, summation_basis, summation_shift, claimed_sum, d_out):
        basis, shift, ssh, mu = _as_u64(basis), _as_u64(shift), _as_u64(summation_shift), _as_u64(claimed_sum)
        sb = np.ascontiguousarray(summation_basis, dtype=np.uint64).reshape(-1, 3)
        self._check(self.c.iopx_sumcheck_g_gf192_dev(_vp(d_f), _vp(d_h), basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p),
                                                     sb.ctypes.data_as(_u64p), sb.shape[0], ssh.ctypes.data_as(_u64p), mu.ctypes.data_as(_u64p), _vp(d_out)))

    def sumcheck_g_multiplicative_dev(self, d_f, d_h, log_n, gen, shift, summation_log_order, summation_shift, claimed_sum, d_out):
        gen, shift, ssh, mu = _as_u64(gen), _as_u64(shift), _as_u64(summation_shift), _as_u64(claimed_sum)
        self._check(self.c.iopx_sumcheck_g_fp3_dev(_vp(d_f), _vp(d_h), int(log_n), gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p),
                                                   int(summation_log_order), ssh.ctypes.data_as(_u64p), mu.ctypes.data_as(_u64p), _vp(d_out)))

    def sumcheck_g(self, f, h, basis, shift, summation_basis, summation_shift, claimed_sum):
        return self._ldt_combine_host([f, h], lambda d, o: self.sumcheck_g_dev(d[0], d[1], basis, shift, summation_basis, summation_shift, claimed_sum, o))

    def sumcheck_g_multiplicative(self, f, h, log_n, gen, shift, summation_log_order, summation_shift, claimed_sum):
        return self._ldt_combine_host([f, h], lambda d, o: self.sumcheck_g_multiplicative_dev(d[0], d[1], log_n, gen, shift, summation_log_order,
                                                                                              summation_shift, claimed_sum, o))

    # ---- lincheck virtual oracle (basic_lincheck_aux.tcc:102-144) ----
    def lincheck_dev(self, d_fz, d_mz, r_mz, d_p1, d_p2, n, d_out, prime_field=False):
        r = _as_u64(r_mz)
        if r.shape[0] != len(d_mz):
            raise ValueError("Not enough random linear combination coefficients were provided")
        ptrs = (_vp * len(d_mz))(*d_mz)
        fn = self.c.iopx_lincheck_fp3_dev if prime_field else self.c.iopx_lincheck_gf192_dev
        self._check(fn(_vp(d_fz), ptrs, len(d_mz), r.ctypes.data_as(_u64p), _vp(d_p1), _vp(d_p2), int(n), _vp(d_out)))

    def lincheck(self, fz, mz, r_mz, p1, p2, prime_field=False):
        n = _as_u64(fz).shape[0]
        return self._ldt_combine_host([fz, p1, p2] + list(mz), lambda d, o: self.lincheck_dev(d[0], d[3:], r_mz, d[1], d[2], n, o, prime_field))

    # ---- LDT reducer (ldt_reducer_aux.tcc:39-131) ----
    def ldt_combine_dev(self, d_oracles, degrees, random_coefficients, basis, shift, d_out):
        """combined_LDT_virtual_oracle::evaluated_contents over the affine subspace (basis, shift); device pointers."""
        basis, shift, rc = _as_u64(basis), _as_u64(shift), _as_u64(random_coefficients)
        if rc.shape[0] != 2 * len(d_oracles):
            raise ValueError("Expected the nunmber of random coefficients to be twice the number of oracles.")
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        deg = (_sz * len(degrees))(*[int(d) for d in degrees])
        self._check(self.c.iopx_ldt_combine_gf192_dev(ptrs, len(d_oracles), deg, rc.ctypes.data_as(_u64p), basis.ctypes.data_as(_u64p),
                                                      basis.shape[0], shift.ctypes.data_as(_u64p), _vp(d_out)))

    def ldt_combine_multiplicative_dev(self, d_oracles, degrees, random_coefficients, log_n, gen, shift, d_out):
        gen, shift, rc = _as_u64(gen), _as_u64(shift), _as_u64(random_coefficients)
        if rc.shape[0] != 2 * len(d_oracles):
            raise ValueError("Expected the nunmber of random coefficients to be twice the number of oracles.")
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        deg = (_sz * len(degrees))(*[int(d) for d in degrees])
        self._check(self.c.iopx_ldt_combine_fp3_dev(ptrs, len(d_oracles), deg, rc.ctypes.data_as(_u64p), int(log_n),
                                                    gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p), _vp(d_out)))

    def _ldt_combine_host(self, evals, call):
        evals = [_as_u64(e) for e in evals]
        n = evals[0].shape[0]
        for e in evals:
            if e.shape[0] != n:
                raise ValueError("Vectors of mismatched size.")
        bufs = [self.malloc(e.nbytes) for e in evals] + [self.malloc(evals[0].nbytes)]
        try:
            for b, e in zip(bufs, evals):
                self.h2d(b, e)
            call(bufs[:-1], bufs[-1])
            out = np.empty_like(evals[0])
            self.d2h(out, bufs[-1])
        finally:
            for b in bufs:
                self.free(b)
        return out

    def ldt_combine(self, evals, degrees, random_coefficients, basis, shift):
        """Host-array form of ldt_combine_dev (copies in and out)."""
        return self._ldt_combine_host(evals, lambda d, o: self.ldt_combine_dev(d, degrees, random_coefficients, basis, shift, o))

    def ldt_combine_multiplicative(self, evals, degrees, random_coefficients, log_n, gen, shift):
        return self._ldt_combine_host(evals, lambda d, o: self.ldt_combine_multiplicative_dev(d, degrees, random_coefficients, log_n, gen, shift, o))

    # ---- proof of work (pow.tcc) ----
    def solve_pow(self, challenge, pow_bitlen, poseidon_params=None):
        """pow::solve_pow: 32-byte challenge -> 32-byte answer (BLAKE2b), or, with `poseidon_params`, a (4,) uint64
        Montgomery challenge -> the Montgomery words of the smallest passing counter."""
        if poseidon_params is None:
            ch = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(challenge))
            out = (ctypes.c_uint8 * 32)()
            self._check(self.c.iopx_pow_solve_blake2b(ctypes.addressof(ch), int(pow_bitlen), ctypes.addressof(out)))
            return bytes(out)
        ch = np.ascontiguousarray(challenge, dtype=np.uint64).reshape(4)
        out = np.empty(4, dtype=np.uint64)
        self._check(self.c.iopx_pow_solve_poseidon_bn128(ctypes.byref(poseidon_params.c), _vp(ch.ctypes.data), int(pow_bitlen), _vp(out.ctypes.data)))
        return out

    # ---- the native Aurora prover (libiop_amd/csrc/prover_capi.hip) ----
    def aurora_example_instance(self, field_code, num_constraints, num_inputs, num_variables, seed):
        """Opaque handle of generate_r1cs_example(n, k, vars) seeded with `seed`, resident in HBM (release with aurora_instance_free)."""
        h = ctypes.c_void_p()
        self.c.iopx_aurora_example_instance_create.argtypes = [ctypes.c_int, _sz, _sz, _sz, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
        self._check(self.c.iopx_aurora_example_instance_create(int(field_code), int(num_constraints), int(num_inputs), int(num_variables), int(seed), ctypes.byref(h)))
        return h

    def aurora_instance(self, field_code, matrices, num_variables, num_inputs, assignment):
        """iopx_aurora_instance_create: an instance from the caller's own constraint system and variable assignment.  matrices = three
        (row_ptr, col, coeff) triples in CSR form (A, B, C; coeff: (entries, 3) uint64 words; column 0 is the constant 1), assignment =
        (num_variables, 3) words, primary inputs first.  Release with aurora_instance_free."""
        class _R1CS(ctypes.Structure):
            _fields_ = [("num_constraints", _sz), ("num_variables", _sz), ("num_inputs", _sz), ("row_ptr", _u64p * 3),
                        ("col", ctypes.POINTER(ctypes.c_uint32) * 3), ("coeff", _u64p * 3)]
        keep, r = [], _R1CS()
        r.num_constraints, r.num_variables, r.num_inputs = len(matrices[0][0]) - 1, int(num_variables), int(num_inputs)
        for q, (row_ptr, col, coeff) in enumerate(matrices):
            rp, cl = np.ascontiguousarray(row_ptr, dtype=np.uint64), np.ascontiguousarray(col, dtype=np.uint32)
            cf = np.ascontiguousarray(coeff, dtype=np.uint64).reshape(-1, 3)
            keep += [rp, cl, cf]
            r.row_ptr[q], r.col[q], r.coeff[q] = rp.ctypes.data_as(_u64p), cl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cf.ctypes.data_as(_u64p)
        z = np.ascontiguousarray(assignment, dtype=np.uint64).reshape(-1, 3)
        h = ctypes.c_void_p()
        self.c.iopx_aurora_instance_create.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        self._check(self.c.iopx_aurora_instance_create(ctypes.byref(r), z.ctypes.data_as(_u64p), int(field_code), ctypes.byref(h)))
        return h

    def aurora_prove(self, instance, security_parameter=128, RS_extra_dimensions=5, FRI_localization_parameter=2):
        """aurora_snark_prover through the C ABI: the canonical transcript bytes."""
        buf, n = ctypes.c_void_p(), _sz(0)
        self.c.iopx_aurora_prove.argtypes = [ctypes.c_void_p, _sz, _sz, _sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_sz)]
        self._check(self.c.iopx_aurora_prove(instance, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter), ctypes.byref(buf), ctypes.byref(n)))
        try:
            return ctypes.string_at(buf, n.value)
        finally:
            self.c.iopx_host_free.argtypes = [ctypes.c_void_p]
            self.c.iopx_host_free(buf)

    def fractal_index(self, instance, security_parameter=128, RS_extra_dimensions=3, FRI_localization_parameter=2):
        """fractal_snark_indexer on the instance: the prover index stays in HBM inside it; returns the verifier index (the tree roots)."""
        roots = (ctypes.c_uint8 * (32 * 8))()
        n = _sz(0)
        self.c.iopx_fractal_index.argtypes = [ctypes.c_void_p, _sz, _sz, _sz, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]
        self._check(self.c.iopx_fractal_index(instance, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter), ctypes.addressof(roots), 8,
                                              ctypes.byref(n)))
        return [bytes(roots[32 * i:32 * (i + 1)]) for i in range(n.value)]

    def fractal_prove(self, instance, security_parameter=128, RS_extra_dimensions=3, FRI_localization_parameter=2):
        buf, n = ctypes.c_void_p(), _sz(0)
        self.c.iopx_fractal_prove.argtypes = [ctypes.c_void_p, _sz, _sz, _sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_sz)]
        self._check(self.c.iopx_fractal_prove(instance, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter), ctypes.byref(buf), ctypes.byref(n)))
        try:
            return ctypes.string_at(buf, n.value)
        finally:
            self.c.iopx_host_free.argtypes = [ctypes.c_void_p]
            self.c.iopx_host_free(buf)

    # ---- multi-GPU: the communicator of include/libiop_amd.h ("multi-GPU") and the provers distributed over it ----
    def comm_rccl_unique_id(self):
        """The 128-byte ncclUniqueId rank 0 creates and hands to the other ranks."""
        buf = (ctypes.c_uint8 * 128)()
        self.c.iopx_comm_rccl_unique_id.argtypes = [ctypes.c_void_p]
        self._check(self.c.iopx_comm_rccl_unique_id(ctypes.addressof(buf)))
        return bytes(buf)

    def comm_create_rccl(self, rank, world, unique_id):
        h = ctypes.c_void_p()
        uid = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self.c.iopx_comm_create_rccl.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        self._check(self.c.iopx_comm_create_rccl(int(rank), int(world), ctypes.addressof(uid), ctypes.byref(h)))
        return h

    def comm_create_rccl_from_torch(self, dist, rank, world, device):
        """One RCCL communicator over the ranks of an initialised torch.distributed group: rank 0's unique id travels through the group
        (a byte tensor broadcast), the collectives of the provers then go through the library's own communicator on the library's stream."""
        import torch
        uid = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            uid = torch.tensor(list(self.comm_rccl_unique_id()), dtype=torch.uint8, device=device)
        if world > 1:
            dist.broadcast(uid, src=0)
        return self.comm_create_rccl(rank, world, bytes(uid.cpu().numpy().tobytes()))

    def comm_create_torch_callbacks(self, dist, rank, world):
        """A communicator whose collectives are forwarded to an initialised torch.distributed group through callbacks on HOST pointers:
        what the CPU test-suite uses (gloo group, kernels compiled for the CPU).  The returned handle keeps the callbacks alive."""
        import torch
        _i, _vp_, _szt = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

        def view(ptr, nbytes):
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))

        def all_gather(user, send, recv, nbytes, stream):
            parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, view(send, nbytes).clone())
            view(recv, nbytes * world).copy_(torch.cat(parts))
            return 0

        def all_reduce_u64(user, buf, count, op, stream):
            t = view(buf, count * 8).view(torch.int64)
            if op == 0:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)                # wrapping int64 sum = wrapping uint64 sum
            else:                                                       # unsigned minimum through the order-preserving map x -> x ^ 2^63
                flipped = t ^ torch.tensor(-(1 << 63), dtype=torch.int64)
                dist.all_reduce(flipped, op=dist.ReduceOp.MIN)
                t.copy_(flipped ^ torch.tensor(-(1 << 63), dtype=torch.int64))
            return 0

        def broadcast(user, buf, nbytes, root, stream):
            dist.broadcast(view(buf, nbytes), src=root)
            return 0

        def all_to_all(user, send, recv, nbytes, stream):
            out = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            src = view(send, nbytes * world)
            ins = [src[q * nbytes:(q + 1) * nbytes].clone() for q in range(world)]
            if dist.get_backend() == "gloo":                            # gloo has no all_to_all for CPU tensors in every build: gather per destination
                for q in range(world):
                    got = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)] if rank == q else None
                    dist.gather(ins[q], got, dst=q)
                    if rank == q:
                        out = got
            else:
                dist.all_to_all(out, ins)
            view(recv, nbytes * world).copy_(torch.cat(out))
            return 0

        def sendrecv(user, send, recv, nbytes, peer, stream):
            r = torch.empty(nbytes, dtype=torch.uint8)
            ops = [dist.P2POp(dist.isend, view(send, nbytes).clone(), peer), dist.P2POp(dist.irecv, r, peer)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            view(recv, nbytes).copy_(r)
            return 0

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a)
                except Exception:                                       # an exception must not unwind through the C caller
                    import traceback
                    traceback.print_exc()
                    return 1
            return wrapped

        types = [ctypes.CFUNCTYPE(_i, _vp_, _vp_, _vp_, _szt, _vp_), ctypes.CFUNCTYPE(_i, _vp_, _vp_, _szt, _i, _vp_), ctypes.CFUNCTYPE(_i, _vp_, _vp_, _szt, _i, _vp_),
                 ctypes.CFUNCTYPE(_i, _vp_, _vp_, _vp_, _szt, _vp_), ctypes.CFUNCTYPE(_i, _vp_, _vp_, _vp_, _szt, _i, _vp_)]
        fns = [t(guard(f)) for t, f in zip(types, (all_gather, all_reduce_u64, broadcast, all_to_all, sendrecv))]

        class Callbacks(ctypes.Structure):
            _fields_ = [("user", _vp_), ("all_gather", types[0]), ("all_reduce_u64", types[1]), ("broadcast", types[2]), ("all_to_all", types[3]), ("sendrecv", types[4])]

        cbs = Callbacks(None, *fns)
        h = ctypes.c_void_p()
        self.c.iopx_comm_create_callbacks.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        self._check(self.c.iopx_comm_create_callbacks(int(rank), int(world), ctypes.addressof(cbs), ctypes.byref(h)))
        self._comm_keepalive = getattr(self, "_comm_keepalive", []) + [(cbs, fns)]
        return h

    def additive_FFT_dist_dev(self, comm, d_block, basis, shift, d_out, inverse=False):
        """One transform as long as its domain across the ranks of `comm` (iopx_add_[i]fft_gf192_dist_dev): this rank's contiguous block in, its block out."""
        basis, shift = _as_u64(basis), _as_u64(shift)
        fn = self.c.iopx_add_ifft_gf192_dist_dev if inverse else self.c.iopx_add_fft_gf192_dist_dev
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _u64p, _sz, _u64p, ctypes.c_void_p]
        self._check(fn(comm, _vp(d_block), basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p), _vp(d_out)))

    def comm_create_replay(self, rank, world):
        """Rank `rank` of `world` played alone on this GPU (collectives completed locally, timing only: the transcript is meaningless)."""
        h = ctypes.c_void_p()
        self.c.iopx_comm_create_replay.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        self._check(self.c.iopx_comm_create_replay(int(rank), int(world), ctypes.byref(h)))
        return h

    def comm_destroy(self, comm):
        self.c.iopx_comm_destroy.argtypes = [ctypes.c_void_p]
        self._check(self.c.iopx_comm_destroy(comm))

    def comm_stats(self, reset=False):
        """(collectives issued, payload bytes sent by this rank) since the last reset."""
        n, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.c.iopx_comm_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        self._check(self.c.iopx_comm_stats(ctypes.byref(n), ctypes.byref(b), 1 if reset else 0))
        return int(n.value), int(b.value)

    def _transcript_call(self, fn, *args):
        buf, n = ctypes.c_void_p(), _sz(0)
        self._check(fn(*args, ctypes.byref(buf), ctypes.byref(n)))
        try:
            return ctypes.string_at(buf, n.value)
        finally:
            self.c.iopx_host_free.argtypes = [ctypes.c_void_p]
            self.c.iopx_host_free(buf)

    def aurora_instance_warm(self, instance, fractal=False, security_parameter=128, RS_extra_dimensions=None, FRI_localization_parameter=2):
        """One-time work of the first proof done ahead of it (pool, plans, tables, transposed matrices): include/libiop_amd.h."""
        rs = RS_extra_dimensions if RS_extra_dimensions is not None else (3 if fractal else 5)
        self.c.iopx_aurora_instance_warm.argtypes = [ctypes.c_void_p, ctypes.c_int, _sz, _sz, _sz]
        self._check(self.c.iopx_aurora_instance_warm(instance, 1 if fractal else 0, int(security_parameter), int(rs), int(FRI_localization_parameter)))

    def aurora_prove_dist(self, instance, comm, security_parameter=128, RS_extra_dimensions=5, FRI_localization_parameter=2):
        """aurora_snark_prover with the codeword-domain vectors distributed over `comm`: every rank calls it and gets the same transcript."""
        self.c.iopx_aurora_prove_dist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _sz, _sz, _sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_sz)]
        return self._transcript_call(self.c.iopx_aurora_prove_dist, instance, comm, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter))

    def fractal_index_dist(self, instance, comm, security_parameter=128, RS_extra_dimensions=3, FRI_localization_parameter=2):
        roots = (ctypes.c_uint8 * (32 * 8))()
        n = _sz(0)
        self.c.iopx_fractal_index_dist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _sz, _sz, _sz, ctypes.c_void_p, _sz, ctypes.POINTER(_sz)]
        self._check(self.c.iopx_fractal_index_dist(instance, comm, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter),
                                                   ctypes.addressof(roots), 8, ctypes.byref(n)))
        return [bytes(roots[32 * i:32 * (i + 1)]) for i in range(n.value)]

    def fractal_prove_dist(self, instance, comm, security_parameter=128, RS_extra_dimensions=3, FRI_localization_parameter=2):
        self.c.iopx_fractal_prove_dist.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _sz, _sz, _sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(_sz)]
        return self._transcript_call(self.c.iopx_fractal_prove_dist, instance, comm, int(security_parameter), int(RS_extra_dimensions), int(FRI_localization_parameter))

    def fri_snark_prove(self, field_code, d_poly_coeffs, n_coeffs, codeword_domain_dim, RS_extra_dimensions, FRI_localization_parameter=2,
                        num_interactive_repetitions=1, num_query_repetitions=10, comm=None):
        """FRI_snark_prover through the C ABI (libiop_amd/cpp/fri.hpp inside the library): the canonical transcript bytes; `comm`: distributed."""
        self.c.iopx_fri_snark_prove_dist.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, _sz, _sz, _sz, _sz, _sz, _sz, ctypes.POINTER(ctypes.c_void_p),
                                                      ctypes.POINTER(_sz)]
        return self._transcript_call(self.c.iopx_fri_snark_prove_dist, int(field_code), comm, _vp(d_poly_coeffs), int(n_coeffs), int(codeword_domain_dim),
                                     int(RS_extra_dimensions), int(FRI_localization_parameter), int(num_interactive_repetitions), int(num_query_repetitions))

    def aurora_instance_free(self, instance):
        self.c.iopx_aurora_instance_free.argtypes = [ctypes.c_void_p]
        self.c.iopx_aurora_instance_free(instance)

    def blake2b_host(self, msg, digest_size=32, key=b""):
        """The library's host-side BLAKE2b (the hashchain of the native prover): no device needed."""
        out = (ctypes.c_uint8 * int(digest_size))()
        m, k = bytes(msg), bytes(key)
        self.c.iopx_blake2b_host.argtypes = [ctypes.c_void_p, _sz, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz]
        self._check(self.c.iopx_blake2b_host(ctypes.addressof(out), int(digest_size), m if m else None, len(m), k if k else None, len(k)))
        return bytes(out)

    def pow_search(self, challenge, pow_bitlen, first, count):
        """The smallest passing proof-of-work candidate index in [first, first + count) or None (candidate 0 = the challenge itself,
        i >= 1 = the challenge with its last word set to i - 1: the reference's search order, pow.tcc:86-112)."""
        ch = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(challenge))
        found = ctypes.c_uint64(0)
        self.c.iopx_pow_search_blake2b.argtypes = [_vp, _sz, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
        self._check(self.c.iopx_pow_search_blake2b(ctypes.addressof(ch), int(pow_bitlen), int(first), int(count), ctypes.byref(found)))
        return None if found.value == 0xFFFFFFFFFFFFFFFF else int(found.value)

    def pow_search_begin(self, challenge, pow_bitlen, first, count):
        """pow_search in two halves: enqueues the batch on the library's stream and returns; pow_search_end waits and reads the result."""
        ch = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(challenge))
        self.c.iopx_pow_search_blake2b_begin.argtypes = [_vp, _sz, ctypes.c_uint64, ctypes.c_uint64]
        self._check(self.c.iopx_pow_search_blake2b_begin(ctypes.addressof(ch), int(pow_bitlen), int(first), int(count)))

    def pow_search_end(self):
        found = ctypes.c_uint64(0)
        self.c.iopx_pow_search_blake2b_end.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
        self._check(self.c.iopx_pow_search_blake2b_end(ctypes.byref(found)))
        return None if found.value == 0xFFFFFFFFFFFFFFFF else int(found.value)

    def pow_candidate(self, challenge, index):
        ch = (ctypes.c_uint8 * 32).from_buffer_copy(bytes(challenge))
        out = (ctypes.c_uint8 * 32)()
        self.c.iopx_pow_candidate_blake2b.argtypes = [_vp, ctypes.c_uint64, _vp]
        self._check(self.c.iopx_pow_candidate_blake2b(ctypes.addressof(ch), int(index), ctypes.addressof(out)))
        return bytes(out)

    # ---- device-pointer operators (integers are raw device addresses, e.g. torch.Tensor.data_ptr()) ----
    def additive_FFT_dev(self, d_coeffs, n_coeffs, basis, shift, d_out):
        basis, shift = _as_u64(basis), _as_u64(shift)
        self._check(self.c.iopx_add_fft_gf192_dev(_vp(d_coeffs), n_coeffs, basis.ctypes.data_as(_u64p), basis.shape[0],
                                                  shift.ctypes.data_as(_u64p), _vp(d_out)))

    def additive_LDE_dev(self, d_coeffs, n_coeffs, basis, shift, coset_begin, coset_count, d_out):
        """Cosets [coset_begin, +coset_count) of the transform (one GPU's shard of a codeword)."""
        basis, shift = _as_u64(basis), _as_u64(shift)
        self._check(self.c.iopx_add_lde_gf192_dev(_vp(d_coeffs), n_coeffs, basis.ctypes.data_as(_u64p), basis.shape[0],
                                                  shift.ctypes.data_as(_u64p), coset_begin, coset_count, _vp(d_out)))

    def additive_IFFT_batch_dev(self, d_evals, batch, basis, shift, d_out):
        """`batch` inverse transforms over one domain, vectors back to back."""
        basis, shift = _as_u64(basis), _as_u64(shift)
        self._check(self.c.iopx_add_ifft_gf192_batch_dev(_vp(d_evals), int(batch), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                         shift.ctypes.data_as(_u64p), _vp(d_out)))

    def additive_LDE_batch_dev(self, d_coeffs, n_coeffs, basis, shift, coset_begin, coset_count, d_outs):
        """Low-degree extensions of several polynomials (lists of device pointers) over one domain; phase 1 runs once."""
        basis, shift = _as_u64(basis), _as_u64(shift)
        cin, cout = (_vp * len(d_coeffs))(*d_coeffs), (_vp * len(d_outs))(*d_outs)
        self._check(self.c.iopx_add_lde_gf192_batch_dev(cin, int(n_coeffs), len(d_coeffs), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                        shift.ctypes.data_as(_u64p), int(coset_begin), int(coset_count), cout))

    def additive_reextend_batch_dev(self, d_evals, batch, basis, d_dim, eval_shift, shift, coset_begin, coset_count, d_outs):
        """FFT_over_field_subset(IFFT_over_field_subset(evals, H), L) for `batch` back-to-back vectors over H = span(basis[:d_dim]) +
        eval_shift onto cosets of L = span(basis) + shift, without materialising the coefficients."""
        basis, shift, es = _as_u64(basis), _as_u64(shift), _as_u64(eval_shift)
        self.c.iopx_add_reextend_gf192_batch_dev.argtypes = [_vp, _sz, _u64p, _sz, _sz, _u64p, _u64p, _sz, _sz, ctypes.POINTER(_vp)]
        cout = (_vp * len(d_outs))(*d_outs)
        self._check(self.c.iopx_add_reextend_gf192_batch_dev(_vp(d_evals), int(batch), basis.ctypes.data_as(_u64p), basis.shape[0], int(d_dim),
                                                             es.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p), int(coset_begin), int(coset_count), cout))

    def additive_reextend2_batch_dev(self, d_evals_a, batch_a, eval_shift_a, d_evals_b, batch_b, eval_shift_b, basis, d_dim, shift, coset_begin, coset_count, d_outs):
        """additive_reextend_batch_dev for two groups of vectors over cosets of the same span with different shifts, in one batch: d_outs holds
        group a's codewords first, then group b's."""
        basis, shift, ea, eb = _as_u64(basis), _as_u64(shift), _as_u64(eval_shift_a), _as_u64(eval_shift_b)
        self.c.iopx_add_reextend2_gf192_batch_dev.argtypes = [_vp, _sz, _u64p, _vp, _sz, _u64p, _u64p, _sz, _sz, _u64p, _sz, _sz, ctypes.POINTER(_vp)]
        cout = (_vp * len(d_outs))(*d_outs)
        self._check(self.c.iopx_add_reextend2_gf192_batch_dev(_vp(d_evals_a), int(batch_a), ea.ctypes.data_as(_u64p), _vp(d_evals_b), int(batch_b), eb.ctypes.data_as(_u64p),
                                                              basis.ctypes.data_as(_u64p), basis.shape[0], int(d_dim), shift.ctypes.data_as(_u64p), int(coset_begin),
                                                              int(coset_count), cout))

    def additive_reextend_lde_batch_dev(self, d_evals, batch, d_coeffs, n_coeffs, basis, d_dim, eval_shift, shift, coset_begin, coset_count, d_outs):
        """additive_reextend_batch_dev plus, in the same batch, the codewords of the polynomials at device pointers d_coeffs (n_coeffs
        coefficients each, at most 2^d_dim): d_outs holds the `batch` re-extensions first, then one codeword per polynomial."""
        basis, shift, es = _as_u64(basis), _as_u64(shift), _as_u64(eval_shift)
        self.c.iopx_add_reextend_lde_gf192_batch_dev.argtypes = [_vp, _sz, ctypes.POINTER(_vp), _sz, _sz, _u64p, _sz, _sz, _u64p, _u64p, _sz, _sz,
                                                                 ctypes.POINTER(_vp)]
        cin = (_vp * max(1, len(d_coeffs)))(*d_coeffs)
        cout = (_vp * len(d_outs))(*d_outs)
        self._check(self.c.iopx_add_reextend_lde_gf192_batch_dev(_vp(d_evals), int(batch), cin, int(n_coeffs), len(d_coeffs), basis.ctypes.data_as(_u64p),
                                                                 basis.shape[0], int(d_dim), es.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p),
                                                                 int(coset_begin), int(coset_count), cout))

    def taylor_dev(self, d_S, log_n, d_twist=0):
        self._check(self.c.iopx_add_taylor_gf192_dev(_vp(d_S), log_n, _vp(d_twist)))

    def taylor_inv_dev(self, d_S, log_n, d_twist=0):
        self._check(self.c.iopx_add_taylor_inv_gf192_dev(_vp(d_S), log_n, _vp(d_twist)))

    def combine_inv_dev(self, d_lo, d_up, d_out, count, index_base, basis, shift_term, upper):
        basis, shift_term = _as_u64(basis), _as_u64(shift_term)
        self._check(self.c.iopx_add_combine_inv_gf192_dev(_vp(d_lo), _vp(d_up), _vp(d_out), count, index_base, basis.ctypes.data_as(_u64p),
                                                          basis.shape[0], shift_term.ctypes.data_as(_u64p), int(upper)))

    def pow_table_dev(self, d_out, count, base, init):
        base, init = _as_u64(base), _as_u64(init)
        self._check(self.c.iopx_gf192_pow_table_dev(_vp(d_out), count, base.ctypes.data_as(_u64p), init.ctypes.data_as(_u64p)))

    def combine_dev(self, d_a, d_b, d_out, count, index_base, basis, shift_term, upper):
        basis, shift_term = _as_u64(basis), _as_u64(shift_term)
        self._check(self.c.iopx_add_combine_gf192_dev(_vp(d_a), _vp(d_b), _vp(d_out), count, index_base, basis.ctypes.data_as(_u64p),
                                                      basis.shape[0], shift_term.ctypes.data_as(_u64p), int(upper)))

    def additive_IFFT_dev(self, d_evals, basis, shift, d_out):
        basis, shift = _as_u64(basis), _as_u64(shift)
        self._check(self.c.iopx_add_ifft_gf192_dev(_vp(d_evals), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                   shift.ctypes.data_as(_u64p), _vp(d_out)))

    def fri_fold_dev(self, d_f, basis, shift, coset_size, x_i, d_next):
        basis, shift, x = _as_u64(basis), _as_u64(shift), _as_u64(x_i)
        self._check(self.c.iopx_fri_fold_add_gf192_dev(_vp(d_f), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                       shift.ctypes.data_as(_u64p), int(coset_size), x.ctypes.data_as(_u64p), _vp(d_next)))

    def merkle_tree_dev(self, d_oracles, elem_bytes, n, coset_size, d_nodes, domain_type=DOMAIN_ADDITIVE, d_salts=0, salt_bytes=0):
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        self._check(self.c.iopx_merkle_blake2b_dev(ptrs, len(d_oracles), elem_bytes, n, int(coset_size), int(domain_type),
                                                   _vp(d_salts), salt_bytes, _vp(d_nodes)))

    def merkle_leaves_dev(self, d_oracles, elem_bytes, n, coset_size, d_nodes, domain_type=DOMAIN_ADDITIVE, d_salts=0, salt_bytes=0):
        """Leaf digests only, into nodes[L-1 .. 2L-2] (merkle_tree.tcc:116-149)."""
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        self._check(self.c.iopx_merkle_leaves_blake2b_dev(ptrs, len(d_oracles), elem_bytes, n, int(coset_size), int(domain_type),
                                                          _vp(d_salts), salt_bytes, _vp(d_nodes)))

    def merkle_inner_dev(self, d_nodes, num_leaves):
        """compute_inner_nodes (merkle_tree.tcc:200-229) over leaf digests already stored in the node array."""
        self._check(self.c.iopx_merkle_inner_blake2b_dev(_vp(d_nodes), int(num_leaves)))

    # ---- encoded Aurora prover: vector-sized steps between the transforms (include/libiop_amd.h) ----
    def spmv_dev(self, d_row_ptr, d_col, d_coeff, rows, d_vec, d_out, scale=None, accumulate=False, prime_field=False):
        """out[r] (+)= scale * sum_t coeff[t] * vec[col[t]] over CSR rows (r1cs.tcc:236-268; basic_lincheck_aux.tcc:64-88)."""
        sc = _as_u64(scale).ctypes.data_as(_u64p) if scale is not None else None
        fn = self.c.iopx_spmv_fp3_dev if prime_field else self.c.iopx_spmv_gf192_dev
        self._check(fn(_vp(d_row_ptr), _vp(d_col), _vp(d_coeff), int(rows), _vp(d_vec), sc, int(bool(accumulate)), _vp(d_out)))

    def poly_div_vanishing_dev(self, d_poly, n_coeffs, basis, shift, d_quotient):
        """Quotient by the vanishing polynomial of the affine subspace (basis, shift): n_coeffs - 2^dim coefficients."""
        shift = _as_u64(shift)
        b = np.ascontiguousarray(basis, dtype=np.uint64).reshape(-1, 3)
        self._check(self.c.iopx_poly_div_vanishing_gf192_dev(_vp(d_poly), int(n_coeffs), b.ctypes.data_as(_u64p), b.shape[0],
                                                             shift.ctypes.data_as(_u64p), _vp(d_quotient)))

    def poly_div_vanishing_multiplicative_dev(self, d_poly, n_coeffs, log_order, shift, d_quotient):
        shift = _as_u64(shift)
        self._check(self.c.iopx_poly_div_vanishing_fp3_dev(_vp(d_poly), int(n_coeffs), int(log_order), shift.ctypes.data_as(_u64p), _vp(d_quotient)))

    def lincomb_dev(self, d_oracles, coefficients, n, d_out, prime_field=False):
        """random_linear_combination_oracle::evaluated_contents: out = sum_i coefficients[i] * oracle_i."""
        co = _as_u64(coefficients)
        if co.shape[0] != len(d_oracles):
            raise ValueError("Random Linear Combination Oracle: Expected same number of random coefficients as oracles.")
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        fn = self.c.iopx_lincomb_fp3_dev if prime_field else self.c.iopx_lincomb_gf192_dev
        self._check(fn(ptrs, len(d_oracles), co.ctypes.data_as(_u64p), int(n), _vp(d_out)))

    def lincomb_affine_dev(self, d_oracles, coefficients, constant, n, d_out, prime_field=False):
        """sum_i coefficients[i] * oracle_i + constant (single_matrix_denominator::evaluated_contents)."""
        co, c0 = _as_u64(coefficients), _as_u64(constant)
        if co.shape[0] != len(d_oracles):
            raise ValueError("Expected same number of coefficients as oracles.")
        ptrs = (_vp * len(d_oracles))(*d_oracles)
        fn = self.c.iopx_lincomb_affine_fp3_dev if prime_field else self.c.iopx_lincomb_affine_gf192_dev
        self._check(fn(ptrs, len(d_oracles), co.ctypes.data_as(_u64p), c0.ctypes.data_as(_u64p), int(n), _vp(d_out)))

    def field_div_dev(self, d_num, d_den, d_out, count, prime_field=False):
        """d_out = d_num / d_den elementwise by batch inversion (d_num None: inverses); zero denominators give zero."""
        fn = self.c.iopx_fp3_div_dev if prime_field else self.c.iopx_gf192_div_dev
        self._check(fn(_vp(d_num) if d_num is not None else None, _vp(d_den), _vp(d_out), int(count)))

    def domain_offsets_dev(self, basis, shift, point, d_out):
        """d_out[j] = point - x_j over the affine subspace (basis, shift)."""
        basis, shift, point = _as_u64(basis).reshape(-1, 3), _as_u64(shift), _as_u64(point)
        self._check(self.c.iopx_domain_offsets_gf192_dev(basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p),
                                                         point.ctypes.data_as(_u64p), _vp(d_out)))

    def domain_offsets_multiplicative_dev(self, log_n, gen, shift, point, d_out):
        gen, shift, point = _as_u64(gen), _as_u64(shift), _as_u64(point)
        self._check(self.c.iopx_domain_offsets_fp3_dev(int(log_n), gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p), point.ctypes.data_as(_u64p),
                                                       _vp(d_out)))

    def vanishing_evals_dev(self, basis, shift, vanishing_basis, vanishing_shift, constant, d_out):
        """d_out[j] = constant - Z_S(x_j), S = span(vanishing_basis) + vanishing_shift, x over the subspace (basis, shift)."""
        basis, vb = _as_u64(basis).reshape(-1, 3), _as_u64(vanishing_basis).reshape(-1, 3)
        shift, vs, c0 = _as_u64(shift), _as_u64(vanishing_shift), _as_u64(constant)
        self._check(self.c.iopx_vanishing_evals_gf192_dev(basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p), vb.ctypes.data_as(_u64p),
                                                          vb.shape[0], vs.ctypes.data_as(_u64p), c0.ctypes.data_as(_u64p), _vp(d_out)))

    def vanishing_evals_multiplicative_dev(self, log_n, gen, shift, vanishing_log_order, vanishing_shift, constant, d_out):
        gen, shift, vs, c0 = _as_u64(gen), _as_u64(shift), _as_u64(vanishing_shift), _as_u64(constant)
        self._check(self.c.iopx_vanishing_evals_fp3_dev(int(log_n), gen.ctypes.data_as(_u64p), shift.ctypes.data_as(_u64p), int(vanishing_log_order),
                                                        vs.ctypes.data_as(_u64p), c0.ctypes.data_as(_u64p), _vp(d_out)))

    def rational_combine_dev(self, d_numerators, d_denominators, coefficients, n, d_numerator_out, d_denominator_out, prime_field=False):
        """combined_numerator / combined_denominator::evaluated_contents for up to 4 rationals."""
        co = _as_u64(coefficients)
        if co.shape[0] != len(d_numerators) or len(d_numerators) != len(d_denominators):
            raise ValueError("Expected same number of random coefficients as oracles.")
        pn, pd = (_vp * len(d_numerators))(*d_numerators), (_vp * len(d_denominators))(*d_denominators)
        fn = self.c.iopx_rational_combine_fp3_dev if prime_field else self.c.iopx_rational_combine_gf192_dev
        self._check(fn(pn, pd, len(d_numerators), co.ctypes.data_as(_u64p), int(n), _vp(d_numerator_out), _vp(d_denominator_out)))

    def rational_sumcheck_constraint_dev(self, d_p, d_N, d_D, d_xinv, basis, shift, summation_dim, summation_shift, claimed_sum, d_out):
        basis, shift, ss, mu = _as_u64(basis).reshape(-1, 3), _as_u64(shift), _as_u64(summation_shift), _as_u64(claimed_sum)
        self._check(self.c.iopx_rational_sumcheck_constraint_gf192_dev(_vp(d_p), _vp(d_N), _vp(d_D), _vp(d_xinv), basis.ctypes.data_as(_u64p), basis.shape[0],
                                                                       shift.ctypes.data_as(_u64p), int(summation_dim), ss.ctypes.data_as(_u64p),
                                                                       mu.ctypes.data_as(_u64p), _vp(d_out)))

    def rational_sumcheck_constraint_multiplicative_dev(self, d_p, d_N, d_D, log_n, gen, shift, summation_log_order, summation_shift, claimed_sum, d_out):
        gen, shift, ss, mu = _as_u64(gen), _as_u64(shift), _as_u64(summation_shift), _as_u64(claimed_sum)
        self._check(self.c.iopx_rational_sumcheck_constraint_fp3_dev(_vp(d_p), _vp(d_N), _vp(d_D), int(log_n), gen.ctypes.data_as(_u64p),
                                                                     shift.ctypes.data_as(_u64p), int(summation_log_order), ss.ctypes.data_as(_u64p),
                                                                     mu.ctypes.data_as(_u64p), _vp(d_out)))

    def gf192_vanishing_host(self, basis, shift, x):
        """(Z_S(x), Z_S's linear coefficient) for S = span(basis) + shift, on the host."""
        basis, shift, x = _as_u64(basis).reshape(-1, 3), _as_u64(shift), _as_u64(x)
        value, lin = np.zeros(3, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
        self._check(self.c.iopx_gf192_vanishing_host(basis.ctypes.data_as(_u64p), basis.shape[0], shift.ctypes.data_as(_u64p), x.ctypes.data_as(_u64p),
                                                     value.ctypes.data_as(_u64p), lin.ctypes.data_as(_u64p)))
        return value, lin

    def gf192_inverse_host(self, x):
        x, out = _as_u64(x), np.zeros(3, dtype=np.uint64)
        self._check(self.c.iopx_gf192_inverse_host(x.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p)))
        return out

    def field_add_dev(self, d_a, d_b, d_out, count):
        self._check(self.c.iopx_gf192_add_dev(_vp(d_a), _vp(d_b), _vp(d_out), int(count)))

    def field_inv_dev(self, d_a, d_out, count, prime_field=False):
        fn = self.c.iopx_fp3_inv_dev if prime_field else self.c.iopx_gf192_inv_dev
        self._check(fn(_vp(d_a), _vp(d_out), int(count)))

    def fp3_mul_dev(self, d_a, d_b, d_out, count):
        self._check(self.c.iopx_fp3_mul_dev(_vp(d_a), _vp(d_b), _vp(d_out), int(count)))

    def fp3_sub_dev(self, d_a, d_b, d_out, count):
        self._check(self.c.iopx_fp3_sub_dev(_vp(d_a), _vp(d_b), _vp(d_out), int(count)))

    def fp3_pow_table_dev(self, d_out, count, base, init):
        base, init = _as_u64(base), _as_u64(init)
        self._check(self.c.iopx_fp3_pow_table_dev(_vp(d_out), int(count), base.ctypes.data_as(_u64p), init.ctypes.data_as(_u64p)))

    def fri_additive_domains(self, basis, shift, localization):
        """FRI_protocol::compute_domains, additive branch (fri_ldt.tcc:310-338): [(basis_i, shift_i)] for L^(0), L^(1), ..."""
        basis, shift = _as_u64(basis), _as_u64(shift)
        m = basis.shape[0]
        dims, d = [], m
        for eta in localization:
            d -= int(eta)
            dims.append(d)
        ob = np.zeros((max(sum(dims), 1), 3), dtype=np.uint64)
        osh = np.zeros((max(len(dims), 1), 3), dtype=np.uint64)
        loc = (_sz * max(len(localization), 1))(*[int(e) for e in localization])
        self.c.iopx_fri_domains_gf192.argtypes = [_u64p, _sz, _u64p, ctypes.POINTER(_sz), _sz, _u64p, _u64p]
        self._check(self.c.iopx_fri_domains_gf192(basis.ctypes.data_as(_u64p), m, shift.ctypes.data_as(_u64p), loc, len(localization),
                                                  ob.ctypes.data_as(_u64p), osh.ctypes.data_as(_u64p)))
        out, off = [(basis, shift.reshape(3))], 0
        for i, dm in enumerate(dims):
            out.append((ob[off:off + dm].copy(), osh[i].copy()))
            off += dm
        return out

    def set_option(self, name, value):
        """A named integer option of the library (schedule switches such as IOPX_HEAD_EVAL, tile geometries): include/libiop_amd.h."""
        self.c.iopx_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        self._check(self.c.iopx_set_option(name.encode(), int(value)))

    def clear_option(self, name):
        self.c.iopx_clear_option.argtypes = [ctypes.c_char_p]
        self._check(self.c.iopx_clear_option(name.encode()))

    def get_option(self, name, default):
        self.c.iopx_get_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
        return int(self.c.iopx_get_option(name.encode(), int(default)))

    def cold_stats(self, reset=False):
        """{label: (count, total_ms)} of one-time host-side costs (pool growth, plans, tables, transpositions) since the last reset."""
        buf = ctypes.create_string_buffer(1 << 14)
        self.c.iopx_cold_stats.argtypes = [ctypes.c_char_p, _sz, ctypes.c_int]
        self._check(self.c.iopx_cold_stats(buf, len(buf), 1 if reset else 0))
        out = {}
        for line in buf.value.decode().splitlines():
            f = line.split()
            out[f[0]] = (int(f[1]), float(f[2]))
        return out

    def profile_begin(self):
        self._check(self.c.iopx_profile_begin())

    def profile_report(self):
        """Returns {kernel: (launches, total_ms, algorithmic_bytes)} for the launches since profile_begin(); the field products each kernel
        reported (0 for kernels that do not) are kept in self.last_profile_products."""
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.c.iopx_profile_report(buf, len(buf)))
        out, products, gaps = {}, {}, {"idle_ms": None, "count": None, "longest": []}
        for line in buf.value.decode().splitlines():
            f = line.split()
            if f[0] == "@idle":                      # idle time between consecutive profiled launches
                gaps["idle_ms"], gaps["count"] = float(f[1]), int(f[2])
                continue
            if f[0] == "@gap":
                gaps["longest"].append({"us": float(f[1]), "after": f[2], "before": f[3]})
                continue
            out[f[0]] = (int(f[1]), float(f[2]), float(f[3]) if len(f) > 3 else 0.0)
            products[f[0]] = float(f[4]) if len(f) > 4 else 0.0
        self.last_profile_products = products
        self.last_profile_gaps = gaps
        return out

    def gf192_mul_halves_dev(self, d_a, d_c2, d_out, d_active, count, lane_mask):
        self.c.iopx_gf192_mul_halves_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _sz, ctypes.c_uint32]
        self._check(self.c.iopx_gf192_mul_halves_dev(_vp(d_a), _vp(d_c2), _vp(d_out), _vp(d_active), count, int(lane_mask)))

    def gf192_mul_dev(self, d_a, d_b, d_out, count):
        self._check(self.c.iopx_gf192_mul_dev(_vp(d_a), _vp(d_b), _vp(d_out), count))

    def gf192_mul(self, a, b):
        """Elementwise a[i] * b[i]; a single-row b selects the wave-uniform multiplier kernel."""
        a, b = _as_u64(a), _as_u64(b)
        n = a.shape[0]
        if b.shape[0] == 1 and n != 1:
            da, db, do = self.malloc(a.nbytes), self.malloc(24), self.malloc(a.nbytes)
            try:
                self.h2d(da, a)
                self.h2d(db, b)
                self._check(self.c.iopx_gf192_mul_uniform_dev(_vp(da), _vp(db), _vp(do), n))
                out = np.empty_like(a)
                self.d2h(out, do)
            finally:
                for p in (da, db, do):
                    self.free(p)
            return out
        da, db, do = self.malloc(a.nbytes), self.malloc(a.nbytes), self.malloc(a.nbytes)
        try:
            self.h2d(da, a)
            self.h2d(db, b)
            self.gf192_mul_dev(da, db, do, n)
            out = np.empty_like(a)
            self.d2h(out, do)
        finally:
            for p in (da, db, do):
                self.free(p)
        return out


_default = None


def lib():
    """The product library (HIP, gfx950).  Raises if it has not been built."""
    global _default
    if _default is None:
        _default = Library()
    return _default


EDWARDS_FR_MODULUS = 1552511030102430251236801561344621993261920897571225601    # libff edwards_Fr, 181 bits
EDWARDS_FR_GENERATOR = 19                                                        # multiplicative_generator


def edwards_to_montgomery(values):
    """canonical ints -> libff Fp_model `mont_repr` words (x * 2^192 mod p), shape (count, 3)."""
    out = np.empty((len(values), 3), dtype=np.uint64)
    for i, v in enumerate(values):
        mv = (int(v) % EDWARDS_FR_MODULUS) * (1 << 192) % EDWARDS_FR_MODULUS
        out[i] = [(mv >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
    return out


def edwards_subgroup_generator(log_n):
    """multiplicative_generator^((p-1)/2^log_n) — multiplicative_subgroup_base::construct_internal, subgroup.tcc:55-59."""
    g = pow(EDWARDS_FR_GENERATOR, (EDWARDS_FR_MODULUS - 1) >> log_n, EDWARDS_FR_MODULUS)
    return edwards_to_montgomery([g])[0]


def standard_basis(m):
    """affine_subspace default basis: element i is FieldT(1ull << i) (subspace.tcc:93-108)."""
    b = np.zeros((m, 3), dtype=np.uint64)
    for i in range(m):
        b[i, 0] = np.uint64(1) << np.uint64(i)
    return b
