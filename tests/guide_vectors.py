"""Known-answer vectors that do NOT depend on this repo's recollection of cutadapt's source.

They restate the worked examples and tables of the published cutadapt user guide
(docs/guide.rst of the ``cutadapt~=5.0`` the reference pins, reference pyproject.toml:17) --
adapter-type tables for ``-a`` / ``-g`` / anchored / non-internal adapters, the error-tolerance
table, the quality-trimming worked example -- plus DNA-alphabet copies of the same shapes, so
the bit-parallel filter and the strip DP see them as well as the raw-alphabet fallback.
Every vector is checked on the C oracle and the string pipeline (CPU suite) and on the HIP
kernel through the C ABI (``-m gpu`` suite); the call sites they stand for are
cutseq/run.py:332-417 (single-end) and 544-723 (paired).

A vector: (kind, adapter, max_error_rate, min_overlap, read, kept) where ``kept`` is the read
after trimming (the guide prints exactly that) and kind names the cutadapt adapter class.
"""

# kind -> (pyref class name, where, remove-before?, rightmost?)
KINDS = {
    "back": ("BackAdapter", "BACK", False, False),                     # -a ADAPTER
    "front": ("FrontAdapter", "FRONT", True, False),                   # -g ADAPTER
    "prefix": ("PrefixAdapter", "PREFIX", True, False),                # -g ^ADAPTER
    "suffix": ("SuffixAdapter", "SUFFIX", False, False),               # -a ADAPTER$
    "back_ni": ("NonInternalBackAdapter", "BACK_NI", False, False),    # -a ADAPTERX
    "front_ni": ("NonInternalFrontAdapter", "FRONT_NI", True, False),  # -g XADAPTER
    "rightmost_front": ("RightmostFrontAdapter", "BACK", True, True),  # -g "ADAPTER;rightmost"
}

GUIDE = [
    # --- guide "Regular 3' adapters" (-a ADAPTER, default error rate 0.1, minimum overlap 3)
    ("back", "ADAPTER", 0.1, 3, "MYSEQUENCEADAPTER", "MYSEQUENCE"),
    ("back", "ADAPTER", 0.1, 3, "MYSEQUENCEADAP", "MYSEQUENCE"),
    ("back", "ADAPTER", 0.1, 3, "MYSEQUENCEADAPTERSOMETHINGELSE", "MYSEQUENCE"),
    ("back", "ADAPTER", 0.1, 3, "MADAPTER", "M"),
    ("back", "ADAPTER", 0.1, 3, "ADAPTERSOMETHING", ""),
    ("back", "ADAPTER", 0.1, 3, "MYSEQUENCE", "MYSEQUENCE"),
    # the guide prints the read part in lower case: output keeps the case of the input
    ("back", "ADAPTER", 0.1, 3, "mysequenceADAPTERsomethingelse", "mysequence"),
    ("back", "ADAPTER", 0.1, 3, "mysequenceADA", "mysequence"),
    ("back", "ADAPTER", 0.1, 3, "mysequenceAD", "mysequenceAD"),  # below the minimum overlap of 3
    # --- guide "Regular 5' adapters" (-g ADAPTER)
    ("front", "ADAPTER", 0.1, 3, "ADAPTERMYSEQUENCE", "MYSEQUENCE"),
    ("front", "ADAPTER", 0.1, 3, "DAPTERMYSEQUENCE", "MYSEQUENCE"),
    ("front", "ADAPTER", 0.1, 3, "TERMYSEQUENCE", "MYSEQUENCE"),
    ("front", "ADAPTER", 0.1, 3, "SOMETHINGADAPTERMYSEQUENCE", "MYSEQUENCE"),
    ("front", "ADAPTER", 0.1, 3, "MYSEQUENCE", "MYSEQUENCE"),
    # --- guide "Anchored 5' adapters" (-g ^ADAPTER): only a full-length occurrence at the very start
    ("prefix", "ADAPTER", 0.1, 7, "ADAPTERMYSEQUENCE", "MYSEQUENCE"),
    ("prefix", "ADAPTER", 0.1, 7, "DAPTERMYSEQUENCE", "DAPTERMYSEQUENCE"),
    ("prefix", "ADAPTER", 0.1, 7, "SOMETHINGADAPTERMYSEQUENCE", "SOMETHINGADAPTERMYSEQUENCE"),
    # --- guide "Anchored 3' adapters" (-a ADAPTER$)
    ("suffix", "ADAPTER", 0.1, 7, "MYSEQUENCEADAPTER", "MYSEQUENCE"),
    ("suffix", "ADAPTER", 0.1, 7, "MYSEQUENCEADAP", "MYSEQUENCEADAP"),
    ("suffix", "ADAPTER", 0.1, 7, "MYSEQUENCEADAPTERSOMETHINGELSE", "MYSEQUENCEADAPTERSOMETHINGELSE"),
    # --- guide "Non-internal 5' and 3' adapters" (-a ADAPTERX / -g XADAPTER)
    ("back_ni", "ADAPTER", 0.1, 3, "MYSEQUENCEADAPTER", "MYSEQUENCE"),
    ("back_ni", "ADAPTER", 0.1, 3, "MYSEQUENCEADAP", "MYSEQUENCE"),
    ("back_ni", "ADAPTER", 0.1, 3, "MYSEQUENCEADAPTERSOMETHINGELSE", "MYSEQUENCEADAPTERSOMETHINGELSE"),
    ("front_ni", "ADAPTER", 0.1, 3, "ADAPTERMYSEQUENCE", "MYSEQUENCE"),
    ("front_ni", "ADAPTER", 0.1, 3, "TERMYSEQUENCE", "MYSEQUENCE"),
    ("front_ni", "ADAPTER", 0.1, 3, "SOMETHINGADAPTERMYSEQUENCE", "SOMETHINGADAPTERMYSEQUENCE"),
    # --- guide "Multiple adapter occurrences within a single read": the leftmost one is used
    ("back", "ADAPTER", 0.1, 3, "cccADAPTERgggggADAPTERttt", "ccc"),
    ("front", "ADAPTER", 0.1, 3, "cccADAPTERgggggADAPTERttt", "gggggADAPTERttt"),
    # ... and ";rightmost" picks the other one (RightmostFrontAdapter, cutseq/run.py:333, 547)
    ("rightmost_front", "ADAPTER", 0.1, 3, "cccADAPTERgggggADAPTERttt", "ttt"),
]

# lower-case adapter copies in the read: matched through sequence.upper() (cs_params.case_rule ==
# CS_CASE_FOLD, SURVEY.md appendix B.1 -- the one rule in this file that is a recollection)
CASE = [
    ("back", "ADAPTER", 0.1, 3, "mysequenceadapter", "mysequence"),
    ("back", "ADAPTER", 0.1, 3, "mysequenceAdApTersomething", "mysequence"),
    ("front", "ADAPTER", 0.1, 3, "somethingadapterMYSEQUENCE", "MYSEQUENCE"),
]

# DNA-alphabet copies of the same shapes (construction-certain: the insert shares no 3-mer with the
# adapter and no error is involved, so no tie-breaking rule takes part)
_AD = "AGATCGGAAGAGCACACGTC"  # TruSeq read-through adapter, p7 of 15 presets (cutseq/adapters.toml)
_P5 = "ACACGACGCTCTTCCGATCT"
_INS = "TTGACCTGAACCTTGGAACCTTGACCTGAA"
DNA = [
    ("back", _AD, 0.2, 3, _INS + _AD, _INS),
    ("back", _AD, 0.2, 3, _INS + _AD + "GGGTTT", _INS),
    ("back", _AD, 0.2, 3, _INS + _AD[:13], _INS),
    ("back", _AD, 0.2, 3, _INS + _AD[:3], _INS),
    ("back", _AD, 0.2, 3, _INS + _AD[:2], _INS + _AD[:2]),
    ("back", _AD, 0.2, 3, _AD + _INS, ""),
    ("back", _AD, 0.2, 3, _INS, _INS),
    ("back", _AD, 0.2, 3, _INS + "TT" + _AD + "CCATT" + _AD + "GG", _INS + "TT"),
    ("rightmost_front", _P5, 0.2, 10, "TT" + _P5 + "GGAGG" + _P5 + _INS, _INS),
    ("rightmost_front", _P5, 0.2, 10, _P5[8:] + _INS, _INS),       # 12-nt adapter tail at the read start
    ("rightmost_front", _P5, 0.2, 10, _P5[14:] + _INS, _P5[14:] + _INS),  # 6 nt: out of reach of min_overlap 10
    ("prefix", "ATCACG", 0.2, 6, "ATCACG" + _INS, _INS),
    ("prefix", "ATCACG", 0.2, 6, "TCACG" + _INS, _INS),               # one deletion: floor(6 * .2) = 1 error
    ("prefix", "ATCACG", 0.2, 6, "GG" + "ATCACG" + _INS, "GG" + "ATCACG" + _INS),
    ("suffix", "ATCACG", 0.2, 6, _INS + "ATCACG", _INS),
    # the insert ends ...GAA: the tail takes the two A's too (longer rows lose on score under both selection rules)
    ("back_ni", "A" * 100, 0.15, 3, _INS + "A" * 30, _INS[:-2]),
    ("back_ni", "A" * 100, 0.15, 3, _INS + "A" * 30 + "CCGGTT" * 3, _INS + "A" * 30 + "CCGGTT" * 3),
    ("front_ni", "T" * 100, 0.15, 3, "T" * 25 + _INS[2:], _INS[2:]),
]

# guide "Error tolerance": "adapter of length 10, rate 0.1: no errors below ten bases, one from ten to
# nineteen, two from twenty" -- the number of allowed errors is floor(matched length * rate)
ERROR_TOLERANCE = [
    (0.1, {0: 0, 9: 0, 10: 1, 19: 1, 20: 2, 29: 2, 30: 3}),
    (0.2, {4: 0, 5: 1, 9: 1, 10: 2, 20: 4}),
]

# guide "Quality trimming algorithm": qualities 42 40 26 27 8 7 11 4 2 3, threshold 10 ->
# subtract 10, partial sums from the end, cut at the minimum: the first four bases stay
QUALITY = [
    ([42, 40, 26, 27, 8, 7, 11, 4, 2, 3], 10, 4),
    ([40] * 10, 10, 10),
    ([2] * 10, 10, 0),
    ([42, 40, 26, 27, 8, 7, 11, 4, 2, 30], 10, 10),  # a good last base shields the tail (BWA: sum < 0 at once)
]


CASE += [("back", _AD, 0.2, 3, (_INS + _AD + "GGG").lower(), _INS.lower()),
         ("back", _AD, 0.2, 3, _INS + _AD.lower()[:9] + _AD[9:], _INS)]


def all_vectors():
    return [("guide",) + v for v in GUIDE] + [("dna",) + v for v in DNA] + [("case",) + v for v in CASE]
