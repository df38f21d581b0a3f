"""Container-only check of the drop-in boundary's declarations against the reference's own header TEXT.

The reference's domain classes cannot be compiled here (every libiop header includes libff, an empty submodule, and writing stand-in
headers for it is not allowed), so what CAN be checked is checked: for every accessor and free function the forwarding bodies of
libiop_amd/cpp/reference_binding.hpp go through, the declaration in the reference header (return type, value category, constness,
parameter types) is extracted and compared with the mirror's declaration in libiop_amd/cpp/libiop_amd.hpp.  This is the test that would
have caught round 5's `const FieldT &shift()` (the reference returns BY VALUE: subspace.hpp:61, field_subset.hpp:67, subgroup.hpp:102).
It reads /root/reference, which exists only in the build container: skipped elsewhere."""
import os
import re

import pytest

REF = "/root/reference/libiop"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIRROR = os.path.join(ROOT, "libiop_amd", "cpp", "libiop_amd.hpp")
BINDING = os.path.join(ROOT, "libiop_amd", "cpp", "reference_binding.hpp")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _class_body(text, name):
    """Text of `class name ... { ... };` (first definition, brace-matched)."""
    m = re.search(r"\bclass\s+" + name + r"\b[^;{]*\{", text)
    assert m, "class %s not found" % name
    depth, i = 1, m.end()
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end():i - 1]


def _norm_type(t):
    t = re.sub(r"\b(inline|static|virtual|explicit|typename)\b", " ", t)
    t = t.replace("std::", "")
    t = re.sub(r"libff::enable_if<[^>]*<FieldT>::value,\s*FieldT>::type", "FieldT", t)     # the overload-selection wrapper of fft.hpp:62-88
    t = re.sub(r"\s+", " ", t).strip()
    t = re.sub(r"\s*([&*<>,])\s*", r"\1", t)
    return t


def _param_types(params):
    out, depth, cur = [], 0, ""
    for c in params:
        depth += (c == "<") - (c == ">")
        if c == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur)
    types = []
    for p in out:
        p = re.sub(r"=.*$", "", p.strip())                         # default arguments
        p = _norm_type(p)
        m = re.match(r"^(.*?[&*>\s])\s*([A-Za-z_]\w*)$", p)         # drop the parameter name
        p = m.group(1).strip() if m else p
        p = re.sub(r"^const (size_t|bool|FieldT)$", r"\1", p)       # top-level const of a by-value parameter is not part of the signature
        types.append(p)
    return types


def _declaration(body, name):
    """(return type, [parameter types], const?) of the first declaration or in-class definition of `name` in `body`."""
    for m in re.finditer(r"(?:^|[;{}:>])\s*((?:[\w:<>,&*]|\s)+?)\b" + name + r"\s*\(([^()]*(?:\([^()]*\)[^()]*)*)\)\s*(const)?\s*[;{]", body, flags=re.S):
        if m.group(1).strip() not in ("return", "else"):           # a call, not a declaration
            break
    else:
        raise AssertionError("no declaration of %s" % name)
    ret = re.sub(r"^.*\btemplate\s*<[^>]*>", "", m.group(1), flags=re.S)
    return _norm_type(ret), _param_types(m.group(2)), bool(m.group(3))


def _ref(path):
    with open(os.path.join(REF, path)) as f:
        return _strip_comments(f.read())


def _mirror():
    with open(MIRROR) as f:
        return _strip_comments(f.read())


ACCESSORS = [
    # (reference header, reference class, mirror class, member)
    ("algebra/field_subset/subspace.hpp", "linear_subspace", "affine_subspace", "dimension"),
    ("algebra/field_subset/subspace.hpp", "linear_subspace", "affine_subspace", "num_elements"),
    ("algebra/field_subset/subspace.hpp", "linear_subspace", "affine_subspace", "basis"),
    ("algebra/field_subset/subspace.hpp", "affine_subspace", "affine_subspace", "shift"),
    ("algebra/field_subset/subgroup.hpp", "multiplicative_subgroup_base", "multiplicative_coset", "generator"),
    ("algebra/field_subset/subgroup.hpp", "multiplicative_subgroup_base", "multiplicative_coset", "dimension"),
    ("algebra/field_subset/subgroup.hpp", "multiplicative_subgroup_base", "multiplicative_coset", "num_elements"),
    ("algebra/field_subset/subgroup.hpp", "multiplicative_coset", "multiplicative_coset", "shift"),
    ("algebra/field_subset/subgroup.hpp", "multiplicative_coset", "multiplicative_coset", "element_outside_of_subset"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "type"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "dimension"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "num_elements"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "shift"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "generator"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "basis"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "subspace"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "coset"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "element_outside_of_subset"),
    ("algebra/field_subset/field_subset.hpp", "field_subset", "field_subset", "get_subset_of_order"),
    ("bcs/merkle_tree.hpp", "merkle_tree", "merkle_tree", "get_root"),
    ("bcs/merkle_tree.hpp", "merkle_tree", "merkle_tree", "get_set_membership_proof"),
]


@pytest.mark.parametrize("header,ref_class,mirror_class,member", ACCESSORS, ids=lambda v: os.path.basename(v) if "/" in v else v)
def test_member_declarations_match_the_reference(header, ref_class, mirror_class, member):
    ref = _declaration(_class_body(_ref(header), ref_class), member)
    mine = _declaration(_class_body(_mirror(), mirror_class), member)
    assert ref == mine, "%s::%s: reference %s, mirror %s" % (ref_class, member, ref, mine)


FUNCTIONS = [
    ("algebra/fft.hpp", "additive_FFT"), ("algebra/fft.hpp", "additive_IFFT"),
    ("algebra/fft.hpp", "multiplicative_FFT"), ("algebra/fft.hpp", "multiplicative_IFFT"),
    ("algebra/fft.hpp", "FFT_over_field_subset"), ("algebra/fft.hpp", "IFFT_over_field_subset"),
    ("algebra/fft.hpp", "IFFT_of_known_degree_over_field_subset"),
    ("protocols/ldt/fri/fri_aux.hpp", "evaluate_next_f_i_over_entire_domain"),
    ("protocols/ldt/fri/fri_aux.tcc", "additive_evaluate_next_f_i_over_entire_domain"),
    ("protocols/ldt/fri/fri_aux.tcc", "multiplicative_evaluate_next_f_i_over_entire_domain"),
]


@pytest.mark.parametrize("header,name", FUNCTIONS, ids=lambda v: os.path.basename(v) if "/" in v else v)
def test_free_function_signatures_match_the_reference(header, name):
    ref = _declaration(_ref(header), name)
    mine = _declaration(_mirror(), name)
    assert ref == mine, "%s: reference %s, mirror %s" % (name, ref, mine)


def test_construct_with_leaves_serialized_by_cosets_takes_the_references_arguments():
    """The mirror adds one defaulted trailing argument (the position map, for trees over the other domain kind); the reference's
    two arguments come first with the reference's types (merkle_tree.hpp:88-90)."""
    ref = _declaration(_class_body(_ref("bcs/merkle_tree.hpp"), "merkle_tree"), "construct_with_leaves_serialized_by_cosets")
    mine = _declaration(_class_body(_mirror(), "merkle_tree"), "construct_with_leaves_serialized_by_cosets")
    assert ref[0] == mine[0] == "void" and mine[1][:2] == ref[1] and mine[1][2:] == ["int"] and ref[2] == mine[2]


def _first_constructor(body, name):
    m = re.search(r"[;{}:]\s*" + name + r"\s*\(([^()]*)\)\s*[;:{]", body, flags=re.S)
    assert m, "no constructor of %s" % name
    return _param_types(m.group(1))


def test_merkle_constructor_takes_the_references_arguments():
    """merkle_tree.hpp:67-72: (num_leaves, leaf_hasher, node_hasher, digest_len_bytes, make_zk, security_parameter)."""
    ref = _first_constructor(_class_body(_ref("bcs/merkle_tree.hpp"), "merkle_tree"), "merkle_tree")
    mine = _first_constructor(_class_body(_mirror(), "merkle_tree"), "merkle_tree")
    assert len(ref) == 6 and ref == mine, (ref, mine)


def test_the_binding_never_takes_the_address_of_an_accessor_result():
    """shift() / generator() / subspace() / coset() are prvalues in the reference: `&domain.shift()` is ill-formed against it."""
    bad = re.compile(r"(?<!&)&\s*[\w\.\->]*\b(shift|generator|subspace|coset)\(\)")
    for path in [BINDING, MIRROR, os.path.join(ROOT, "INTEGRATION.md")] + [
            os.path.join(ROOT, "libiop_amd", "cpp", h) for h in sorted(os.listdir(os.path.join(ROOT, "libiop_amd", "cpp")))]:
        with open(path) as f:
            text = f.read()
        if path.endswith((".hpp", ".h")):
            text = _strip_comments(text)
        for i, line in enumerate(text.splitlines(), 1):
            assert not bad.search(line), "%s:%d takes the address of a prvalue: %s" % (path, i, line.strip())


def test_binding_header_uses_only_accessors_the_reference_declares():
    """Every `domain.member()` the forwarding bodies call exists, publicly, on the reference class it is documented for."""
    with open(BINDING) as f:
        body = _strip_comments(f.read())
    used = set(re.findall(r"\b(?:domain|f_i_domain|codeword_domain)\.(\w+)\(", body))
    assert used == {"shift", "basis", "dimension", "num_elements", "generator"}, used
    fs = _class_body(_ref("algebra/field_subset/field_subset.hpp"), "field_subset")
    for member in used:
        _declaration(fs, member)                                     # field_subset declares all five (field_subset.hpp:47-68)
