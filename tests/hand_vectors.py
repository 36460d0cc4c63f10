"""Hand-derived vectors for the rules of the mapping walk that nothing in the reference's own tests pins.

The walk (debruijn_mapping's map_read_to_nodes_with_mismatch, restated in oracle/nimble_oracle.cpp and again in
csrc/kernels.hip) is known to this repository only through recollection; the reference's 15 literals never exercise
the seed stride, the left extension or a re-seed.  Each case below is small enough to walk ON PAPER, and the expected
(coverage, mismatches, class) is the result of that paper walk, written out in the comment -- not the output of either
implementation.  Both the oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_parity.py) are held to
them.  K = 30.

The graph: two alleles A = P + "A" + Q and B = P + "C" + Q, |P| = 100, |Q| = 80, P and Q random.
  node NP: P            (k-mers at offsets 0..70 of both alleles, colour {A, B}); two right extensions, A and C
  node NA: A[71..130)   (the 30 k-mers that hold A's base 100, colour {A}), 59 bases
  node NB: B[71..130)   (colour {B}), 59 bases
  node NQ: Q            (k-mers at 101..151, colour {A, B}), 80 bases, two left extensions
"""
import numpy as np

K = 30


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def _other(b, k=1):
    return "ACGT"[("ACGT".index(b) + k) % 4]


P = _rand(100, 9001)
Q = _rand(80, 9002)
A = P + "A" + Q
B = P + "C" + Q
SEQS = [A, B]
NAMES = ["A", "B"]


def _subst(s, i):
    return s[:i] + _other(s[i], 2) + s[i + 1:]


CASES = []


def _case(name, read, allowed, coverage, mismatches, cls, why):
    CASES.append(dict(name=name, read=read, allowed=allowed, coverage=coverage, mismatches=mismatches, cls=cls, why=why))


# 1. a fork whose junction base matches neither branch.  read = P[60..100) + "G" + Q[0..50), L = 91, last k-mer 61.
#    Seed at 0 (NP offset 60; 0 < int(0.2 * 91) = 18: no left extension).  Enter NP: +30, then its 10 remaining bases
#    match: coverage 40, position 40.  Next base G is no right extension of NP (A, C): search from 40 at stride 3.
#    40: the k-mer starts with the G, absent.  43: read[43..73) = Q[2..32): NQ offset 2.  Enter NQ: +30 = 70,
#    position 73; 18 bases left, all match: 88.  The junction base and the two bases behind it (41, 42) are never
#    counted.  Nodes NP, NQ: class {A, B}.  The junction is not a compared base, so the budget does not matter.
for nm in (0, 1):
    _case("fork_junction_mismatch_nm%d" % nm, P[60:100] + "G" + Q[0:50], nm, 88, 0, ["A", "B"],
          "dead end at a fork, re-seed at stride 3 behind it")

# 2. a seed only the stride-3 scan reaches.  read = 7 foreign bases + P[10..80), L = 77.  Positions 0, 3, 6 hold foreign
#    bases; 7 and 8 would hit but are never asked; 9: read[9..39) = P[12..42): NP offset 12, 9 < int(15.4) = 15, no left
#    extension.  Enter NP: +30, position 39; 38 bases left of the read, 58 of the unitig: +38 = 68.  The two matching
#    bases at 7 and 8 in front of the seed are not covered.
_foreign = "".join(_other(c) for c in P[3:10])
_case("seed_only_at_stride_3", _foreign + P[10:80], 0, 68, 0, ["A", "B"], "positions 0, 3, 6 miss, 9 hits; 7 and 8 are skipped")

# 3. a late first seed with a mismatch directly to its left.  read = P[20..100) with base 20 (P[40]) replaced, L = 80,
#    left-extension threshold int(16.0) = 16.  Every k-mer at 0, 3, .. 18 holds base 20; 21: read[21..51) = P[41..71): NP
#    offset 41, and 21 >= 16: extend left from read base 20 / unitig base 40.
#    budget 0: that first compared base mismatches: counted as a mismatch, then the extension stops with nothing covered.
#              Forward: +30 (position 51), the unitig's last 29 bases match: 59.  (59, 1)
#    budget 1: the mismatch is tolerated and counts as covered, the 20 bases further left match: 21.  Forward the same:
#              21 + 30 + 29 = 80.  (80, 1)
_r3 = _subst(P[20:100], 20)
_case("late_seed_left_extension_nm0", _r3, 0, 59, 1, ["A", "B"], "left extension stops at its first base; the mismatch is still counted")
_case("late_seed_left_extension_nm1", _r3, 1, 80, 1, ["A", "B"], "left extension runs to the start of the read")

# 4. a fork taken.  read = P[50..100) + "C" + Q[0..40) = B[50..141), L = 91.  Seed at 0 (NP offset 50).  Enter NP: +30, its
#    20 remaining bases: 50, position 50.  Next base C is a right extension: on to NB at offset 0, the position goes back
#    by 29 (the k-mer overlap) and so does the coverage: 21.  Enter NB: +30 = 51, position 51; NB has 29 bases behind its
#    first k-mer, all match: 80, position 80.  Next base Q[29] is NB's one right extension: NQ, back by 29: 51.  Enter NQ:
#    81, position 81; 10 bases left: 91 = L.  Nodes NP {A, B}, NB {B}, NQ {A, B}: class {B}.
_case("fork_taken_along_B", B[50:141], 0, 91, 0, ["B"], "k-mer overlap of 29 is un-counted at every hop")

# 5. a mismatch inside a unitig.  read = Q with base 40 replaced, L = 80, last k-mer 50.  Seed at 0 (NQ offset 0).
#    budget 0: enter NQ: +30, bases 30..39 match (40), base 40 mismatches: counted, compare stops, position 40.  Search
#              from 40: 40 holds the bad base; 43: read[43..73) = Q[43..73): NQ offset 43.  Enter NQ again: +30 = 70,
#              position 73, 7 bases left: 77.  (77, 1); bases 40, 41, 42 are not covered.
#    budget 1: the mismatch is tolerated and covered: 30 + 50 = 80.  (80, 1)
_r5 = _subst(Q, 40)
_case("substitution_inside_unitig_nm0", _r5, 0, 77, 1, ["A", "B"], "re-seed three bases behind the mismatch, same unitig entered twice")
_case("substitution_inside_unitig_nm1", _r5, 1, 80, 1, ["A", "B"], "tolerated mismatch counts as covered")

# 6. shorter than a k-mer: nothing to map
_case("shorter_than_k", P[0:29], 0, None, None, None, "read_length < K")
