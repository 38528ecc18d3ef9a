"""oracle/marker_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + plain loops; the inputs of the tests are small) of the reference's marker-array post-pass:
  * marker-positions stream (.mps): records  start, end, marker_1 .. marker_k, 0xFFFFFFFFFFFFFFFF  written by
    MarkerPositionsWriter, include/marker_array.hpp:60-136 (text positions [start, end] -> list of packed markers,
    include/marker.hpp:9-52: allele 4 bits | sequence 16 bits | position 46 bits);
  * lookup  rle_window_arr::at(i), include/rle_window_array.hpp:118-131: the list of the record whose interval holds text
    position i (#starts <= i  ==  #ends < i  + 1), else empty;
  * write_marker_array, include/marker_array.hpp:138-174: walk the suffix array in BWT order, group consecutive rows with
    EQUAL marker lists (vec_eq: by content), and for every group with a non-empty list write
    first row, last row, markers.., 0xFFFFFFFFFFFFFFFF;
  * scripts/readable_markers.py: the text form the reference's goldens (tests/data/*.markers) are kept in.
Pinned by tests/test_markers.py against the reference's own tests/data/{single_chrom,mult_chroms}.{sa,markers}."""
import numpy as np

DELIM = np.uint64(0xFFFFFFFFFFFFFFFF)
# include/marker.hpp:9-13 -- as written there: the position mask has 44 bits and the sequence mask covers bits 44..59 while
# the sequence is shifted by 46, so a sequence id keeps its low 14 bits (set_allele clears bits 60..63 afterwards)
ALE_MASK, SEQ_MASK, POS_MASK = 0xF000000000000000, 0x0FFFF00000000000, 0x00000FFFFFFFFFFF
ALE_SHIFT, SEQ_SHIFT = 60, 46
M64 = (1 << 64) - 1


def create_marker(pos, ale, seq):
    """create_marker_t(pos, ale, seqid), include/marker.hpp:43-52: set_pos, set_seq, set_allele in that order"""
    x = int(pos) & POS_MASK
    x = (((int(seq) & 0xFFFF) << SEQ_SHIFT) & M64) | (x & ~SEQ_MASK & M64)
    x = (((int(ale) & 0xF) << ALE_SHIFT) & M64) | (x & ~ALE_MASK & M64)
    return x


def marker_fields(m):
    """get_seq / get_pos / get_allele, include/marker.hpp:19-37 (and scripts/readable_markers.py)"""
    return (m & SEQ_MASK) >> SEQ_SHIFT, m & POS_MASK, (m & ALE_MASK) >> ALE_SHIFT


def mps_parse(words):
    """records of a .mps / .ma stream -> (starts, ends, lists)"""
    words = np.asarray(words, np.uint64)
    starts, ends, lists = [], [], []
    i, n = 0, words.size
    while i < n:
        j = i
        while words[j] != DELIM:
            j += 1
        assert j - i >= 2, "record without keys"
        starts.append(int(words[i])); ends.append(int(words[i + 1])); lists.append(tuple(int(x) for x in words[i + 2:j]))
        i = j + 1
    return starts, ends, lists


def mps_build(starts, ends, lists):
    out = []
    for s, e, l in zip(starts, ends, lists):
        out += [s, e] + list(l) + [int(DELIM)]
    return np.array(out, np.uint64)


def marker_array(mps_words, sa):
    """write_marker_array: the .ma stream for suffix array `sa` (row 0 holds n, src/pfbwt-f.cpp:301)"""
    starts, ends, lists = mps_parse(mps_words)
    st = np.array(starts, np.uint64); en = np.array(ends, np.uint64)
    sa = np.asarray(sa, np.uint64)
    k = np.searchsorted(st, sa, side="right").astype(np.int64) - 1            # last interval that starts at or before sa[i]
    inside = (k >= 0) & (sa <= en[np.maximum(k, 0)]) if st.size else np.zeros(sa.size, bool)
    out, prev, first = [], (), 0
    for i in range(sa.size):
        cur = lists[k[i]] if inside[i] else ()
        if cur != prev:
            if prev:
                out += [first, i - 1] + list(prev) + [int(DELIM)]
            first = i
        prev = cur
    if prev:
        out += [first, sa.size - 1] + list(prev) + [int(DELIM)]
    return np.array(out, np.uint64)


def readable(ma_words):
    """scripts/readable_markers.py: one line per row of every run, 'row seq pos allele' of the LAST marker of the run's list"""
    starts, ends, lists = mps_parse(ma_words)
    lines = []
    for s, e, l in zip(starts, ends, lists):
        m = l[-1] if l else None
        for j in range(s, e + 1):
            lines.append("%d %s %s %s" % ((j,) + (marker_fields(m) if m is not None else (None, None, None))))
    return "\n".join(lines) + ("\n" if lines else "")


def mps_from_golden(sa, markers_text):
    """Inverse of the pipeline for goldens made with --ma_wsize 1 (tests/vcf_to_bwt_test.sh:29: every list holds one marker):
    text position sa[row] carries the marker printed for `row`; maximal runs of consecutive text positions with the same
    marker are the records MarkerPositionsWriter writes (adjacent equal windows are merged, marker_array.hpp:124-127)."""
    sa = np.asarray(sa, np.uint64)
    pos2m = {}
    for line in markers_text.strip().splitlines():
        row, seq, pos, ale = (int(x) for x in line.split())
        pos2m[int(sa[row])] = create_marker(pos, ale, seq)
    starts, ends, lists = [], [], []
    for p in sorted(pos2m):
        if starts and ends[-1] + 1 == p and lists[-1] == (pos2m[p],):
            ends[-1] = p
        else:
            starts.append(p); ends.append(p); lists.append((pos2m[p],))
    return mps_build(starts, ends, lists)
