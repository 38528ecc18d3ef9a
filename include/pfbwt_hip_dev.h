/*
 * include/pfbwt_hip_dev.h -- measurement and development entry points of libpfbwt_hip.so.  NOT part of the
 * drop-in boundary (include/pfbwt_hip.h): nothing here stands in for a reference interface; bench.py and
 * tools/ use it for the per-kernel HIP-event timings behind the roofline figures.
 */
#ifndef PFBWT_HIP_DEV_H
#define PFBWT_HIP_DEV_H
#include "pfbwt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- instrumentation ---------------------------------------------------------------------------- */
/* per-kernel timing with hipEvents on the context's stream (off by default) */
int pfp_profile_enable(pfp_ctx *ctx, int on);
/* time only the launches of one kernel (name as reported by pfp_profile_get); keeps event overhead out of a timed run */
int pfp_profile_select(pfp_ctx *ctx, const char *kernel_name);
int pfp_profile_reset(pfp_ctx *ctx);
/* idx-th record: kernel name, launches, total ms, algorithmic bytes; returns 0 or PFP_E_ARG past the end */
int pfp_profile_get(pfp_ctx *ctx, int idx, const char **name, uint64_t *launches, double *ms, double *bytes);
/* wall-clock milliseconds of the last call of each stage (host timer around a stream sync):
 * [0] parse_finalize [1] parse_bwt [2] bwt_build */
int pfp_stage_ms(pfp_ctx *ctx, double out[3]);
/* Route and tuning switches of ONE context (tests force the routes that only huge inputs take; A/B measurements):
 *   verbose, seg_grid, seg_stage, sort_k (1 | 3), sort_no_table, class_sort_maxrange, dedup_table_log2, no_trigger_table,
 *   emit_chunk_rows, fill_subs, sample_cap (< 0: none), no_runaware, big_group_members (< 0: never), force_wide_rows,
 *   fasta_chunk_bytes, ingest_block_bytes, emit_group_rows (0: no group-stationary emission of the special rows; small: most groups left to the row-wise kernel),
 *   no_slot_records (per-slot fields by two gathers: the route of dictionaries with words of 64 Mbase and more),
 *   dict_text_rounds (-1 auto | 0 rank-based dictionary sort only | 1 text rounds forced), int_key_symbols (2 | 3: parse symbols in the initial sort key),
 *   force_run_round (the run round of the dictionary sort even without a long run).
 *   round 4: parse_rec (-1 auto | 0 never | 1 always: suffix sort of the parse through its level-2 prefix-free parse, csrc/recsort.h), parse_rec_p2, parse_rec_min,
 *   parse_rec_depth, parse_rec_tile_rows, parse_rec_table_log2; dict_rec (-1 | 0 | 1: the same for the dictionary, csrc/dictrec.h), dict_rec_p2;
 *   dedup_variant (1: representatives read by the wave together | 0: by every lane | -1 default: 1 for a collection of >= 8 sequences while its first table lasts), dedup_period (workgroups per sequence for the per-XCD column order of
 *   k_dedup_insert: 0 = estimated from the sequences fed, -1 = text order), dedup_chunk (workgroups per column), dedup_phases (!= 0: stage times from inside the kernel on stderr);
 *   ingest_readers, expand_dma.
 * Returns PFP_E_ARG for an unknown key.  In a process started with PFP_TEST_HOOKS=1 pfp_create presets a new context from the
 * environment variables PFP_<KEY IN UPPER CASE>; without PFP_TEST_HOOKS=1 the environment is ignored (PFP_VERBOSE excepted,
 * which only prints). */
int pfp_debug_set(pfp_ctx *ctx, const char *key, long long value);
/* development aid: sorts n pseudo-random (key, value) pairs with `bits` significant key bits, returns the best
 * wall time of `reps` runs and the number of out-of-order neighbours (must be 0) */
int pfp_debug_sort(pfp_ctx *ctx, uint64_t n, int bits, int reps, double *ms_out, uint32_t *unsorted_pairs);
/* position-weighted checksum of `bytes` bytes of device memory that sit at `global_offset` of a larger logical buffer:
 * out[0..1] = sum over bytes of (byte + 1) * mix_k(global position) mod 2^64.  Checksums of the pieces of a buffer add
 * up (mod 2^64) to the checksum of the whole: tools/big_check_slices.py compares the sliced (multi-GPU) outputs of a
 * 32 Gbase build with the single-context output this way, without moving them off the device. */
int pfp_debug_checksum(pfp_ctx *ctx, const void *d_buf, uint64_t bytes, uint64_t global_offset, uint64_t out[2]);

/* Order check of the run samples against the resident text, at any size: .esa[k] and .ssa[k + 1] are adjacent BWT rows, so
 * T[esa[k].sa ..] < T[ssa[k + 1].sa ..] must hold for all r - 1 pairs (the suffixes are compared byte by byte on the device; the
 * engine's own sort results are not consulted).  Needs the state left by pfp_bwt_build(want_rssa = 1) of a context that still
 * holds its text.  out[0] pairs checked, out[1] order violations, out[2] pairs whose rows are not adjacent, out[3] longest and
 * out[4] sum of the common prefixes. */
int pfp_debug_check_sample_order(pfp_ctx *ctx, uint64_t out[5]);
/* Properties of the full suffix array of the last pfp_bwt_build(want_sa = 1) against the resident text, at any size, without moving
 * it off the device: out[0] rows checked; out[1] values above n; out[2] values that occur twice (with out[1] = 0 and n + 1 rows: the
 * array is a permutation of [0, n]); out[3] rows whose BWT byte is not the text byte in front of SA[row] (0x00 in front of the whole
 * text; row 0 must hold n); out[4] 0x00 bytes in the BWT (must be 1).  What scripts/generate_truth_set.py:92-100 of the reference
 * states about its golden files, checked for a text of any length. */
int pfp_debug_check_sa(pfp_ctx *ctx, uint64_t out[5]);
/* The run samples of a build with want_sa = want_rssa = 1 against its own .bwt and .sa, on the device: run k starts where the BWT
 * byte changes, ends in front of the next start, and carries the SA values of its first and last row (src/pfbwt-f.cpp:304-315, 325-328).
 * out[0] runs checked, out[1] runs with a wrong row, out[2] runs with a wrong value. */
int pfp_debug_check_samples(pfp_ctx *ctx, uint64_t out[3]);
/* sum of the 64-bit little-endian words of a device buffer (out[0]) and of word * (word index + 1) (out[1]), modulo 2^64; a
 * trailing partial word is zero-padded.  bench.py checks the outputs that were streamed to host memory against the
 * device-resident ones with it. */
int pfp_debug_wordsum(pfp_ctx *ctx, const void *d_buf, uint64_t bytes, uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
