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

} // namespace libiop_amd
