// R1CS instances for the device prover (libiop/relations/r1cs.{hpp,tcc}, relations/variable.{hpp,tcc}).
//
// r1cs_constraint_system<FieldT> keeps the reference's interface — constraints  <a, z> * <b, z> = <c, z>  over
// z = (1, primary_input, auxiliary_input), variable index 0 being the constant 1 — and hands the prover what its kernels read:
// the three matrices in CSR form in HBM (create_Az_Bz_Cz_from_variable_assignment, r1cs.tcc:236-268, becomes three sparse
// matrix-vector products) and, for the lincheck, their transposes with rows placed at the summation-domain index of each column
// (multi_lincheck_virtual_oracle::set_challenge walks exactly that, basic_lincheck_aux.tcc:64-88).
#pragma once
#include <numeric>

#include "iop.hpp"

namespace libiop_amd {

template<typename FieldT>
struct linear_term {                       // relations/variable.hpp: coeff * variable(index); index 0 is the constant term
    std::size_t index;
    FieldT coeff;
};
template<typename FieldT> using linear_combination = std::vector<linear_term<FieldT>>;

template<typename FieldT>
struct r1cs_constraint {                   // relations/r1cs.hpp:40-68
    linear_combination<FieldT> a, b, c;
};

template<typename FieldT> using r1cs_primary_input = std::vector<FieldT>;
template<typename FieldT> using r1cs_auxiliary_input = std::vector<FieldT>;

// one sparse matrix: CSR on the host (the statement) and, once prepared, in HBM
template<typename FieldT>
struct sparse_matrix {
    std::size_t rows = 0;
    std::vector<uint64_t> row_ptr{ 0 };
    std::vector<uint32_t> col;
    std::vector<FieldT> coeff;
    mutable device_array<uint64_t> d_row_ptr;
    mutable device_array<uint32_t> d_col;
    mutable device_vector<FieldT> d_coeff;

    void add_row(const linear_combination<FieldT> &lc)
    {
        for (auto &t : lc) { col.push_back((uint32_t)t.index); coeff.push_back(t.coeff); }
        row_ptr.push_back(col.size());
        ++rows;
    }
    void to_device() const
    {
        if (d_row_ptr.size() == row_ptr.size()) return;
        d_row_ptr = device_array<uint64_t>::from_host(row_ptr);
        d_col = device_array<uint32_t>::from_host(col);
        d_coeff = device_vector<FieldT>(device_array<FieldT>::from_host(coeff));
    }
    // out[r] (+)= scale * sum_t coeff[t] * vec[col[t]]   (iopx_spmv_*_dev)
    void times_vector(const device_vector<FieldT> &vec, const device_vector<FieldT> &out, const FieldT *scale = nullptr, bool accumulate = false) const
    {
        if (out.size() != rows) throw std::invalid_argument("sparse_matrix::times_vector: output size != rows");
        auto fn = field_host<FieldT>::additive() ? iopx_spmv_gf192_dev : iopx_spmv_fp3_dev;
        check(fn(d_row_ptr.data(), d_col.data(), d_coeff.words(), rows, vec.words(), scale ? detail::words(scale) : nullptr, accumulate ? 1 : 0, out.words()));
    }
    // The transpose with output row out_row_of_col[c] for column c; an entry's new column is its old row.
    sparse_matrix transposed_onto(std::size_t num_rows_out, const std::vector<std::size_t> &out_row_of_col) const
    {
        sparse_matrix T;
        T.rows = num_rows_out;
        std::vector<uint64_t> counts(num_rows_out + 1, 0);
        for (uint32_t c : col) ++counts[out_row_of_col[c] + 1];
        T.row_ptr.assign(num_rows_out + 1, 0);
        for (std::size_t r = 0; r < num_rows_out; ++r) T.row_ptr[r + 1] = T.row_ptr[r] + counts[r + 1];
        T.col.resize(col.size());
        T.coeff.resize(col.size());
        std::vector<uint64_t> fill(T.row_ptr.begin(), T.row_ptr.end() - 1);
        for (std::size_t r = 0; r < rows; ++r)                                    // stable in the old row, as a stable argsort is
            for (uint64_t t = row_ptr[r]; t < row_ptr[r + 1]; ++t) {
                const uint64_t slot = fill[out_row_of_col[col[t]]]++;
                T.col[slot] = (uint32_t)r;
                T.coeff[slot] = coeff[t];
            }
        T.to_device();
        return T;
    }
};

template<typename FieldT>
class r1cs_constraint_system {             // relations/r1cs.hpp:101-154
public:
    std::size_t primary_input_size_ = 0, auxiliary_input_size_ = 0;
    sparse_matrix<FieldT> A, B, C;
    // per-instance data derived for a domain layout (transposed matrices, index permutations): built once, kept with the instance
    mutable std::map<std::string, std::vector<sparse_matrix<FieldT>>> lincheck_matrix_cache_;
    mutable std::map<std::string, device_array<uint64_t>> index_cache_;

    std::size_t num_inputs() const { return primary_input_size_; }
    std::size_t num_variables() const { return primary_input_size_ + auxiliary_input_size_; }
    std::size_t num_constraints() const { return A.rows; }
    void add_constraint(const r1cs_constraint<FieldT> &c) { A.add_row(c.a); B.add_row(c.b); C.add_row(c.c); }
    void prepare_device() const { A.to_device(); B.to_device(); C.to_device(); }
};

// ---- the synthetic instance the reference's harnesses prove (libiop/relations/examples/r1cs_examples.tcc:23-78, called as
// generate_r1cs_example(n, 15, n - 1) by profiling/instrument_aurora_snark.cpp:108-110), seeded with SplitMix64 instead of
// libsodium randomness so that every implementation derives the same instance from (n, k, seed) ----
inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// element i takes stream outputs 3 i .. 3 i + 2 as its words: GF(2^192) the raw words; the prime field the 192-bit draw reduced
// mod p, in Montgomery form (one Montgomery product with R^2 does both)
template<typename FieldT>
std::vector<FieldT> seeded_elements(uint64_t seed, std::size_t count)
{
    typedef field_host<FieldT> H;
    std::vector<FieldT> out(count);
    const FieldT R2 = H::additive() ? H::one() : H::pow(H::from_uint(2), 192);       // the element 2^192: its stored words are R^2 mod p
    for (std::size_t i = 0; i < count; ++i) {
        uint64_t w[3];
        for (int k = 0; k < 3; ++k) w[k] = splitmix64_at(seed, 3 * i + k);
        out[i] = H::from_words(w);
        if (!H::additive()) out[i] = H::mul(out[i], R2);
    }
    return out;
}

template<typename FieldT>
struct r1cs_example {
    r1cs_constraint_system<FieldT> constraint_system;
    r1cs_primary_input<FieldT> primary_input;
    r1cs_auxiliary_input<FieldT> auxiliary_input;
};

// constraint i is z[i mod m] * z[(i + 7) mod m] = coef_i * z[(2 i + 1) mod m] with coef_i = A B / C (the constant term carries A B
// when C's variable is zero); the products and inverses run on the device (2^20 field inversions on one host core would take minutes)
template<typename FieldT>
r1cs_example<FieldT> generate_r1cs_example(std::size_t num_constraints, std::size_t num_inputs, std::size_t num_variables, uint64_t seed)
{
    typedef field_host<FieldT> H;
    if (num_inputs > num_variables) throw std::invalid_argument("Number of inputs can't exceed number of variables.");
    r1cs_example<FieldT> ex;
    const std::vector<FieldT> z = seeded_elements<FieldT>(seed, num_variables);
    std::vector<uint64_t> a_idx(num_constraints), b_idx(num_constraints), c_idx(num_constraints);
    for (std::size_t i = 0; i < num_constraints; ++i) { a_idx[i] = i % num_variables; b_idx[i] = (i + 7) % num_variables; c_idx[i] = (2 * i + 1) % num_variables; }
    const device_vector<FieldT> d_z(device_array<FieldT>::from_host(z));
    auto gathered = [&](const device_vector<FieldT> &src, const std::vector<uint64_t> &idx) {
        const device_array<uint64_t> d_idx = device_array<uint64_t>::from_host(idx);
        device_vector<FieldT> out(idx.size());
        check(iopx_gather_dev(src.data(), d_idx.data(), idx.size(), sizeof(FieldT), out.data()));
        return out;
    };
    auto product = [&](const device_vector<FieldT> &a, const device_vector<FieldT> &b) {
        device_vector<FieldT> out(a.size());
        check((H::additive() ? iopx_gf192_mul_dev : iopx_fp3_mul_dev)(a.words(), b.words(), out.words(), a.size()));
        return out;
    };
    const device_vector<FieldT> ab = product(gathered(d_z, a_idx), gathered(d_z, b_idx));
    device_vector<FieldT> z_inv(num_variables);
    check((H::additive() ? iopx_gf192_inv_dev : iopx_fp3_inv_dev)(d_z.words(), z_inv.words(), num_variables));      // zero stays zero
    const std::vector<FieldT> coef = product(ab, gathered(z_inv, c_idx)).to_host(), ab_host = ab.to_host();
    r1cs_constraint_system<FieldT> &cs = ex.constraint_system;
    cs.primary_input_size_ = num_inputs;
    cs.auxiliary_input_size_ = num_variables - num_inputs;
    const FieldT one = H::one();
    for (std::size_t i = 0; i < num_constraints; ++i) {
        const bool c_zero = H::is_zero(z[c_idx[i]]);
        cs.add_constraint({ { { a_idx[i] + 1, one } }, { { b_idx[i] + 1, one } }, { c_zero ? linear_term<FieldT>{ 0, ab_host[i] } : linear_term<FieldT>{ c_idx[i] + 1, coef[i] } } });
    }
    ex.primary_input.assign(z.begin(), z.begin() + num_inputs);
    ex.auxiliary_input.assign(z.begin() + num_inputs, z.end());
    return ex;
}

} // namespace libiop_amd
