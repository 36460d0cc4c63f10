"""The reference's remaining unit-test literals, held against BOTH the CPU oracle and libnimble_host.so (no GPU).

Each case restates one test of the reference (file:line in its docstring), with the reference's inputs and expected values:
  parse_calls                      /root/reference/src/align.rs:1233-1252
  unmap x3                         src/align.rs:1533-1608
  get_reference_sequence_data x2   src/utils.rs:127-164
  sort_score_vector x6             src/utils.rs:283-360
(the FASTQ reader's error texts of src/parse/fastq.rs:72-83 are in tests/test_host_cpu.py)."""
import importlib
import json

import pytest

from oracle import oracle as ora

nim = importlib.import_module("nimble-aligner_amd")
SEP = "§"


def both(headers, columns, group_on=None, sequence_name_idx=None, sequence_idx=None):
    """the Reference struct as the reference's tests build it by hand, for the oracle and for libnimble_host.so"""
    ni = headers.index("sequence_name") if sequence_name_idx is None else sequence_name_idx
    si = headers.index("sequence") if sequence_idx is None else sequence_idx
    g = ni if group_on is None else group_on
    n0 = len(columns[0])
    if all(len(c) == n0 for c in columns):
        oref = ora.Reference.raw(headers, columns, g, ni, si)
    else:  # columns of different lengths (src/utils.rs:151-161): the equal part first, the longer columns pushed behind
        short = [h for h, c in zip(headers, columns) if len(c) == n0]
        oref = ora.Reference.raw(short, [c for c in columns if len(c) == n0], g, ni, si)
        for h, c in zip(headers, columns):
            if len(c) != n0:
                oref.push_column(h, c)
    return oref, nim.Library.from_table(headers, columns, g, ni, si)


# ---------------- parse_calls ----------------
def test_parse_calls():
    """src/align.rs:1233-1252"""
    calls = ["feat1", "feat2" + SEP + "rev", "feat3", "feat4" + SEP + "rev", "feat4" + SEP + "rev", "feat4"]
    expected = [("feat1", False), ("feat2", True), ("feat3", False), ("feat4", True), ("feat4", True), ("feat4", False)]
    assert ora.parse_calls(calls) == expected
    # the host answers from the tables its coercion reads: a library whose rows carry these names ...
    names = ["feat1", "feat2" + SEP + "rev", "feat3", "feat4" + SEP + "rev", "feat4"]
    _, lib = both(["sequence_name", "sequence"], [names, ["ACGT"] * len(names)])
    assert lib.parse_calls(calls) == expected
    # ... and one that has never seen them (the string rule)
    _, other = both(["sequence_name", "sequence"], [["x"], ["ACGT"]])
    assert other.parse_calls(calls) == expected


# ---------------- unmap ----------------
NAMES3 = ["feature1", "feature2", "feature3"]


def test_unmap():
    """src/align.rs:1533-1556"""
    oref, lib = both(["sequence_name", "sequence"], [NAMES3, ["ACGT"] * 3])
    assert ora.unmap(["feature1", "feature2", "feature3"], oref) == [0, 1, 2]
    assert lib.unmap(["feature1", "feature2", "feature3"]) == [0, 1, 2]


def test_unmap_unorder():
    """src/align.rs:1558-1581"""
    oref, lib = both(["sequence_name", "sequence"], [NAMES3, ["ACGT"] * 3])
    assert ora.unmap(["feature2", "feature1", "feature3"], oref) == [1, 0, 2]
    assert lib.unmap(["feature2", "feature1", "feature3"]) == [1, 0, 2]


def test_process_and_unmap():
    """src/align.rs:1583-1608: names of a class, without group roll-up, and back"""
    oref, lib = both(["sequence_name", "sequence"], [NAMES3, ["ACGT"] * 3])
    cfg = ora.Config()
    feats = ora.process_equivalence_class_to_feature_list([0, 1, 2], oref, cfg, True)
    assert ora.unmap(feats, oref) == [0, 1, 2]
    assert lib.unmap(lib.feature_list([0, 1, 2], True)) == [0, 1, 2]
    assert lib.feature_list([0, 1, 2], True) == feats == NAMES3


def test_unmap_unknown_feature_is_the_reference_panic():
    """src/align.rs:861 `.expect("Feature not found in reference columns")`"""
    oref, lib = both(["sequence_name", "sequence"], [NAMES3, ["ACGT"] * 3])
    with pytest.raises(ora.OracleError, match="Feature not found in reference columns"):
        ora.unmap(["feature9"], oref)
    with pytest.raises(nim.Panic, match="Feature not found in reference columns"):
        lib.unmap(["feature9"])


def test_unmap_takes_the_first_row_of_a_repeated_name():
    """`.position(..)` at src/align.rs:858-861: the first match"""
    names = ["a", "b", "a"]
    oref, lib = both(["sequence_name", "sequence"], [names, ["ACGT"] * 3])
    assert ora.unmap(["a", "b"], oref) == [0, 1] == lib.unmap(["a", "b"])


# ---------------- get_reference_sequence_data ----------------
def test_get_reference_sequence_data():
    """src/utils.rs:127-147"""
    headers = ["id", "sequence_name", "sequence"]
    cols = [["1", "2"], ["gene1", "gene2"], ["ATGC", "CGTA"]]
    oref, lib = both(headers, cols)
    seqs, names = ora.get_reference_sequence_data(oref)
    assert len(seqs) == 2 and names == ["gene1", "gene2"] and seqs == ["ATGC", "CGTA"]
    hseqs, hnames = lib.reference_sequence_data()
    assert hnames == ["gene1", "gene2"] and len(hseqs) == 2
    # the host hands the ASCII column to the index builder, which applies DnaString::from_acgt_bytes itself; rendered the
    # way DnaString::to_string would (2-bit packed and back):
    assert [nim.dna_to_string(s) for s in hseqs] == ["ATGC", "CGTA"]


def test_get_reference_sequence_data_panic_on_missing_name():
    """src/utils.rs:149-164: `#[should_panic(expected = "Error -- could not read library name")]`"""
    headers = ["id", "sequence_name", "sequence"]
    oref, lib = both(headers, [["1"], ["gene1"], ["ATGC", "CGTA"]], group_on=0, sequence_name_idx=1, sequence_idx=2)
    with pytest.raises(ora.OracleError, match="Error -- could not read library name"):
        ora.get_reference_sequence_data(oref)
    with pytest.raises(nim.Panic, match="Error -- could not read library name"):
        lib.reference_sequence_data()


# ---------------- sort_score_vector ----------------
def row(key, n, a, b):
    return ([key], (n, [a], [b]))


SORT_CASES = {
    # src/utils.rs:283-301
    "names": ([row("Charlie", 90, "A", "Fail"), row("Alice", 95, "A", "Pass"), row("Bob", 85, "B", "Pass")],
              [row("Alice", 95, "A", "Pass"), row("Bob", 85, "B", "Pass"), row("Charlie", 90, "A", "Fail")]),
    # :303-308
    "empty": ([], []),
    # :310-315
    "single": ([row("a", 1, "x", "y")], [row("a", 1, "x", "y")]),
    # :317-326
    "sorted": ([row("a", 1, "x", "y"), row("b", 2, "x2", "y2"), row("c", 3, "x3", "y3")],
               [row("a", 1, "x", "y"), row("b", 2, "x2", "y2"), row("c", 3, "x3", "y3")]),
    # :328-342
    "unsorted": ([row("c", 3, "x3", "y3"), row("a", 1, "x", "y"), row("b", 2, "x2", "y2")],
                 [row("a", 1, "x", "y"), row("b", 2, "x2", "y2"), row("c", 3, "x3", "y3")]),
    # :344-358 (equal keys keep their order: sort_by is stable)
    "same_key": ([row("a", 1, "x", "y"), row("a", 2, "x2", "y2"), row("b", 3, "x3", "y3")],
                 [row("a", 1, "x", "y"), row("a", 2, "x2", "y2"), row("b", 3, "x3", "y3")]),
}


@pytest.mark.parametrize("case", sorted(SORT_CASES))
def test_sort_score_vector(case):
    scores, expected = SORT_CASES[case]
    assert ora.sort_score_vector(list(scores)) == expected
    assert nim.sort_score_vector(list(scores)) == expected


def test_sort_score_vector_orders_vectors_not_joined_strings():
    """Vec<String> Ord (src/utils.rs:57): element by element, a shorter vector that is a prefix comes first"""
    scores = [(["a", "b"], 1), (["a"], 2), (["a", "B"], 3), (["A", "z", "z"], 4), (["a!"], 5)]
    expected = [(["A", "z", "z"], 4), (["a"], 2), (["a", "B"], 3), (["a", "b"], 1), (["a!"], 5)]
    assert ora.sort_score_vector(list(scores)) == expected
    assert nim.sort_score_vector(list(scores)) == expected
