// Internal interface between eval.hip (exact kernels, entry points) and eval_filter.hip (the
// fp16 candidate filter and the exact re-scoring of its candidates).
#pragma once
#include "common.h"

namespace rk {

constexpr int kStatSlots = 1024;

// X[n_rows][D] fp32 -> out[n_rows][D] fp16 of x * 2^-e.  per_row_scale: e is the row's own
// exponent (written to row_exp; row_maxabs: n_rows floats of scratch), norms gets the rows'
// Euclidean norms.  Otherwise e is the exponent of the table's largest |x|, and table_stats
// receives [0] the largest row norm, [1] the largest |x| (stats_part: 2 * kStatSlots words of
// scratch, zeroed by the caller).
int half_rows(const float *X, int64_t n_rows, int D, bool per_row_scale, void *out, float *norms,
              float *row_maxabs, int32_t *row_exp, unsigned int *stats_part, float *table_stats,
              hipStream_t st);

// The fused top-K kernel on fp16 rows: per user and item slice the kc best FILTER scores,
// item ids local to this table.
int launch_filter_topk(int kc, int DC, int grid, hipStream_t st, const float *Usplit, int32_t n_users,
                       const float *Isplit, int32_t n_items, const int32_t *mask_ptr,
                       const int32_t *mask_items, int32_t nsplit, float *part_s, int32_t *part_id,
                       const int32_t *heavy_thresh);

// heavy[0..31] (zeroed by the caller): histogram of the users' training-list lengths by bit length;
// heavy[32] receives the length above which a user skips the filter pass: at least a quarter of an
// item per tile of the table (32 at least), and such that at most kHeavyUsers users are above it.
int heavy_threshold(const int32_t *mask_ptr, int32_t n_users, int32_t n_items, int32_t *heavy, hipStream_t st);

// Exact fp32 scores of each user's kc candidates, the best K of them by (score desc, id asc)
// written to out (ids + item_id_offset), and done[user] = 1 iff no item outside the candidates
// can rank among those K (the smallest filter score among the candidates plus the filter's error
// bound is below the K-th exact score, or every item is a candidate).  The other users are counted
// in open_count[0] (zeroed by the caller) and the first open_cap of them listed in open_list.
int refine_candidates(const float *Uemb, const float *Iemb, int32_t n_users, int32_t n_items, int D,
                      const float *cand_s, const int32_t *cand_id, int P, int kc, int K, const int32_t *mask_ptr,
                      const int32_t *heavy_thresh, const float *unorm,
                      const int32_t *uexp, const float *istats, int32_t item_id_offset, float *out_s, int32_t *out_id,
                      int32_t *done, int32_t *open_count, int32_t *open_list, int32_t open_cap,
                      hipStream_t st);

// The threshold form of the filter (topk_threshold.h): per user and item slice the running maxima
// of kStrata strata (strata: threshold_strata_bytes of scratch), thr[user] = the kc-th largest of a
// user's maxima, then every item whose filter score is >= thr[user] as a candidate: cand
// (threshold_cand_bytes) holds (score bits, local item id) pairs, cnt[user * nsplit + slice] their
// number.  Slices are whole groups of four tiles.
int threshold_candidates(int kc, int D, int32_t nsplit, hipStream_t st, const float *Uhalf, int32_t n_users,
                         const float *Ihalf, int32_t n_items, const int32_t *mask_ptr, const int32_t *mask_items,
                         const int32_t *heavy_thresh, float *strata, float *thr, void *cand, int32_t *cnt);
size_t threshold_strata_bytes(int32_t n_users, int32_t nsplit);
size_t threshold_cand_bytes(int32_t n_users, int32_t nsplit);

// refine_candidates for the threshold form: all candidates of a user (at most 64 take part) are
// scored exactly; done[user] = 1 iff thr[user] plus the filter's error bound is below the K-th
// exact score (or every item is a candidate).
int refine_threshold_candidates(const float *Uemb, const float *Iemb, int32_t n_users, int32_t n_items, int D,
                                const void *cand, const int32_t *cnt, int32_t nsplit, const float *thr, int K,
                                const int32_t *mask_ptr, const int32_t *heavy_thresh, const float *unorm,
                                const int32_t *uexp, const float *istats, int32_t item_id_offset, float *out_s,
                                int32_t *out_id, int32_t *done, int32_t *open_count, int32_t *open_list,
                                int32_t open_cap, hipStream_t st);

}  // namespace rk
