// TEST INFRASTRUCTURE ONLY (tests/harness): stand-in for libff/algebra/field_utils/field_utils.hpp — the traits and helpers libiop names.
#pragma once
#include <libff/common/utils.hpp>
#include <libff/algebra/field_utils/bigint.hpp>

namespace libff {

enum field_type { multiplicative_field_type = 1, additive_field_type = 2 };

// libff's own enable_if: the disabled arm is void*, so that both overloads of a member pair exist and overload resolution picks by argument type
template<bool B, class T = void> struct enable_if { typedef void *type; };
template<class T> struct enable_if<true, T> { typedef T type; };

template<typename FieldT> struct is_additive { static const bool value = false; };
template<typename FieldT> struct is_multiplicative { static const bool value = false; };

template<typename FieldT>
field_type get_field_type(const typename enable_if<is_multiplicative<FieldT>::value, FieldT>::type) { return multiplicative_field_type; }
template<typename FieldT>
field_type get_field_type(const typename enable_if<is_additive<FieldT>::value, FieldT>::type) { return additive_field_type; }

template<typename FieldT>
std::size_t log_of_field_size_helper(typename enable_if<is_multiplicative<FieldT>::value, FieldT>::type) { return FieldT::ceil_size_in_bits(); }
template<typename FieldT>
std::size_t log_of_field_size_helper(typename enable_if<is_additive<FieldT>::value, FieldT>::type) { return FieldT::extension_degree(); }

template<typename FieldT>
std::size_t soundness_log_of_field_size_helper(typename enable_if<is_multiplicative<FieldT>::value, FieldT>::type) { return FieldT::floor_size_in_bits(); }
template<typename FieldT>
std::size_t soundness_log_of_field_size_helper(typename enable_if<is_additive<FieldT>::value, FieldT>::type) { return FieldT::extension_degree(); }

template<typename FieldT>
std::size_t get_word_of_field_elem(typename enable_if<is_additive<FieldT>::value, FieldT>::type field_elem, std::size_t word) { return field_elem.to_words()[word]; }
template<typename FieldT>
std::size_t get_word_of_field_elem(typename enable_if<is_multiplicative<FieldT>::value, FieldT>::type field_elem, std::size_t word) { return field_elem.as_bigint().data[word]; }

template<typename FieldT>
FieldT power(const FieldT &base, const unsigned long exponent)
{
    FieldT result = FieldT::one();
    bool found_one = false;
    for (long i = 63; i >= 0; --i) {
        if (found_one) result = result * result;
        if ((exponent >> i) & 1) { found_one = true; result = result * base; }
    }
    return result;
}
template<typename FieldT, mp_size_t m>
FieldT power(const FieldT &base, const bigint<m> &exponent)
{
    FieldT result = FieldT::one();
    bool found_one = false;
    for (long i = (long)exponent.max_bits() - 1; i >= 0; --i) {
        if (found_one) result = result * result;
        if (exponent.test_bit((std::size_t)i)) { found_one = true; result = result * base; }
    }
    return result;
}

} // namespace libff
