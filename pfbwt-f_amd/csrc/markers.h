// pfbwt-f_amd/csrc/markers.h -- marker-array post-pass.
//
// Replaces write_marker_array, include/marker_array.hpp:138-174 (driven by src/mps_to_ma.cpp:45-51): the reference reads
// the suffix array back from a file or pipe, looks every value up in an rle_window_arr (two sd_vector rank queries,
// include/rle_window_array.hpp:118-131) and groups consecutive rows with equal marker lists.  Here the suffix array is
// the one the emission left in HBM: one thread per row bisects the sorted interval starts, run heads are compacted, and
// one thread per run writes its record.  Lists are compared by CONTENT in the reference (vec_eq); the host gives every
// distinct list one id, so the device compares ids.
#pragma once
#include "prims.h"

namespace pfp {

constexpr uint32_t MA_NONE = 0xFFFFFFFFu;

// rowlist[i] = id of the marker list of the interval that holds text position SA[i], MA_NONE if there is none
template <typename SAT> __global__ __launch_bounds__(BLOCK) void k_ma_lookup(const SAT *sa, uint64_t nrows, const uint64_t *istart, const uint64_t *iend, const uint32_t *ilist, uint32_t nint, uint32_t *rowlist)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= nrows) return;
    const uint64_t s = (uint64_t)sa[i];
    uint32_t lo = 0, hi = nint;                              // first interval that starts behind s
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (istart[mid] <= s) lo = mid + 1; else hi = mid; }
    rowlist[i] = (lo && s <= iend[lo - 1]) ? ilist[lo - 1] : MA_NONE;      // rle_window_arr::at: #starts <= s == #ends < s + 1
}
__global__ __launch_bounds__(BLOCK) void k_ma_heads(const uint32_t *rowlist, uint64_t nrows, uint32_t *head)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < nrows) head[i] = (i == 0 || rowlist[i] != rowlist[i - 1]) ? 1u : 0u;      // marker_array.hpp:151: !vec_eq(markers, pmarkers)
}
__global__ __launch_bounds__(BLOCK) void k_ma_collect(const uint32_t *rowlist, const uint32_t *head, const uint32_t *pos, uint64_t nrows, uint64_t *hrow, uint32_t *hlist)
{
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < nrows && head[i]) { hrow[pos[i]] = i; hlist[pos[i]] = rowlist[i]; }
}
// words of the record of run h: first row, last row, the markers, the delimiter; runs without markers write nothing (:152)
__global__ __launch_bounds__(BLOCK) void k_ma_lengths(const uint32_t *hlist, const uint32_t *loff, uint64_t nh, unsigned long long *len)
{
    const uint64_t h = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (h < nh) { const uint32_t l = hlist[h]; len[h] = l == MA_NONE ? 0ULL : 3ULL + (loff[l + 1] - loff[l]); }
}
__global__ __launch_bounds__(BLOCK) void k_ma_write(const uint64_t *hrow, const uint32_t *hlist, const unsigned long long *off, const uint32_t *loff, const uint64_t *lvals, uint64_t nh, uint64_t nrows, uint64_t *out)
{
    const uint64_t h = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (h >= nh) return;
    const uint32_t l = hlist[h];
    if (l == MA_NONE) return;
    uint64_t o = off[h];
    out[o++] = hrow[h];
    out[o++] = (h + 1 < nh ? hrow[h + 1] : nrows) - 1;
    for (uint32_t k = loff[l]; k < loff[l + 1]; ++k) out[o++] = lvals[k];
    out[o] = ~0ULL;                                           // delim, marker_array.hpp:143
}

} // namespace pfp
