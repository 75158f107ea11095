// Host-side launchers of the HIP kernels (one translation unit per stage).
#pragma once
#include "common.h"

namespace pfann {

// ---- mel.hip -------------------------------------------------------------------------
struct MelPlan {
    int seg_len, n_fft, hop, n_mels, n_frames, n_freqs, log2n;
    int power, pad_reflect, log_mode, spec_norm_max;
    float log_eps;
    float *window;        // [n_fft] periodic hann
    float2 *twiddle;      // [n_fft/2] exp(-2*pi*i*j/n_fft)
    int *fb_ptr;          // [n_mels+1] CSR over mel bins
    int *fb_idx;          // [nnz] frequency bin
    float *fb_val;        // [nnz]
    int max_nnz_row, fb_nnz;
};
int launch_melspec(const MelPlan &mp, const float *segs, int64_t B, int64_t seg_stride,
                   const int64_t *starts, int remove_mean, float *out, hipStream_t s);
int launch_resample_to_mono(const int16_t *pcm, int n_ch, const float *K, int old_r, int new_r, int width, const int64_t *plan,
                            int n_pieces, int64_t n_out, float *tmp, float *wav, float *scratch2, hipStream_t s);
int launch_pcm16_to_mono(const int16_t *pcm, int64_t n_frames, int n_ch, float *wav,
                         float *scratch2, hipStream_t s);

// ---- encoder.hip ---------------------------------------------------------------------
int launch_conv_first(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s);
int launch_conv_gemm(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s);
int launch_conv_depthwise(const SubLayer &L, const float *x, float *y, int64_t B, hipStream_t s);
int launch_ln_act(const SubLayer &L, float *xy, int64_t B, int activation, int relu_after_bn,
                  hipStream_t s);
int launch_myg(const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
               int d, int u, int v, int64_t B, float *emb, int normalize, hipStream_t s);
int launch_cl_to_nchw(const float *x, float *y, int64_t B, int C, int HW, hipStream_t s);

// ---- encoder_fused.hip (LayerNorm fused into the GEMM; "fuller" models) -------------------
bool fused_supported(const SubLayer *sub, int n);
int fused_out_slots(const SubLayer &L, int64_t B);
// bytes of split-K scratch sub-layer L needs at batch B under the plan for batch Bp (0: the layer is not split)
size_t splitk_scratch_need(const SubLayer &L, int64_t B, int64_t Bp);
int launch_conv_first_stats(const SubLayer &L, const float *x, float *y, float *part, int64_t B, int act,
                            int after_bn, hipStream_t s);
int launch_conv_first_gram_stats(const SubLayer &L, const float *x, float *part, int64_t B, const float *gram14,
                                 hipStream_t s);
// in_stats: workspace for B x (mean, rstd) of the input, filled here from in_part
int launch_conv_gemm_ln(const SubLayer &L, const SubLayer &Lin, const float *x, const float *in_part, int in_P,
                        float *in_stats, float *y, float *out_part, int64_t B, int act, int after_bn,
                        const SubLayer *Lfirst, int precision, hipStream_t s, float *splitk_scratch = nullptr,
                        size_t splitk_bytes = 0, int *stats_final = nullptr, int64_t Bplan = 0);
int launch_conv_dw_ln(const SubLayer &L, const SubLayer &Lin, const float *x, const float *in_part, int in_P, float *in_stats,
                      float *y, float *out_part, int64_t B, int act, int after_bn, const SubLayer *Lfirst, hipStream_t s);
int launch_ln_apply(const SubLayer &L, const float *z, const float *part, int P, float *out, int64_t B, int act,
                    int after_bn, hipStream_t s);
int launch_myg_ln(const SubLayer &Llast, const float *z, const float *part, int P, int act, int after_bn,
                  const float *w1, const float *b1, const float *w2, const float *b2, int d, int u, int v, int64_t B,
                  float *emb, int normalize, hipStream_t s, int64_t Bplan = 0);
// Bplan (pfann_set_plan_batch): the batch size the kernel VARIANTS (tile size, split-K, small-batch head) are chosen for;
// 0 = the launch's own B.  With a fixed plan a segment's fingerprint has the same bits in every batch it is part of.

// ---- search.hip ----------------------------------------------------------------------
struct SearchWorkspace {
    int64_t cap_q = 0;      // query rows the buffers are sized for
    int cap_c = 0;          // candidate slots per query row
    float *thr = nullptr;   // [cap_q]
    int *cnt = nullptr;     // [cap_q]
    float *cs = nullptr;    // [cap_q][cap_c]
    int64_t *cl = nullptr;  // [cap_q][cap_c]
    int *overflow = nullptr;   // [0] unused, [1] rows left to the big select kernel, [2] rows the wave-per-row select left, [3] spare
    int *left = nullptr;       // [cap_q] the rows counted in overflow[2]: the workgroup selects then walk THIS list with a small grid
    int *row_ovf = nullptr;    // [cap_q] set by a select kernel whose row lost survivors (a sub-list overflowed);
                               // topk_fallback_kernel recomputes those rows exactly ON DEVICE and clears the flag:
                               // no host round trip, no retry loop (allocated zeroed)
    // fp16 pre-filter path (search_f16.hip)
    float *thr_adj = nullptr;  // [cap_q] tau - eps
    float *eps = nullptr;      // [cap_q] per-row bound of |s16 - s|
    void *qh = nullptr;        // [cap_q][d] fp16 query rows
    int64_t qh_elems = 0;
    // two-phase search of a song-sharded job (search_topk phase 1 -> all-reduce MAX of the bounds -> phase 2): what
    // phase 1 left behind, valid for exactly this (q, nq, k)
    const float *bound_q = nullptr;
    int64_t bound_nq = 0;
    int bound_k = 0, bound_G = 0;
    bool bound_valid = false;
};
// db != nullptr: fp32 rows, exact results; dbh != nullptr additionally: fp16 copy of the rows for the pre-filter path
// (batches > 64 rows), xnorm_max = largest row norm of db.
// db == nullptr (fp16-only storage, pfann_db_set_storage): scores are s16 = sum fl16(q_i) * fl16(x_i) accumulated in
// fp32, no fp32 re-scoring.  Fully asynchronous on `s`: no host synchronisation inside.
// phase 0: the whole search.  Song-sharded jobs split it around one collective (pfann_search_bound / _bounded):
//   phase 1: query preparation + sampled group-maximum pass + group select only; lb[m][0..mtop) = the mtop best sampled
//            group maxima of query row m, each lowered to a bound of the TRUE score of its row (-inf padding; all -inf
//            where this path has no sampled threshold).  The k-th largest of the union of all shards' values bounds the
//            k-th best over all shards from below (>= k different real rows reach it);
//   phase 2: lb[m] = that lower bound of the k-th best over ALL shards: the full pass
//            emits only rows that can be in the global top-k, so D / I may hold fewer than k entries (padded) -- the
//            merge of the shards' lists is still the exact global top-k.
int search_topk(const float *db, const void *dbh, float xnorm_max, int64_t n, int d, int64_t label_base,
                const float *q, int64_t nq, int k, float *D, int64_t *I, SearchWorkspace &ws, hipStream_t s,
                int phase = 0, float *lb = nullptr, int mtop = 1);
int topk_merge(const float *S, const int64_t *L, int64_t nq, int m, int k, float *D, int64_t *I,
               hipStream_t s);
// one wavefront per query row (search.hip): k-th largest of the gathered bound candidates; merge of G sorted shard lists
int bound_reduce(const float *cands, int G, int64_t nq, int m, int k, float *lb, hipStream_t s);
int merge_lists(const float *Dl, const int64_t *Il, int G, int64_t nq, int k, float *D, int64_t *I, hipStream_t s);

// ---- rerank.hip ----------------------------------------------------------------------
struct RerankArgs {
    const float *db; const void *dbh;   // fp32 rows, or (db == nullptr) fp16 rows
    int64_t n; int d; int64_t label_base;
    const int64_t *song_pos; int n_songs; int song_lo, song_hi;  // owned songs [lo,hi)
    const float *q; const int64_t *labels; int k;
    const int64_t *qstart; const int32_t *qlen; int64_t nQ;
    int fsm; float alpha; int mode; int only_owned;
    int pmax;               // next pow2 >= max_qlen * k (LDS sizing)
    unsigned long long *gkeys; float *gscore;   // HBM scratch [nQ][pmax] used instead of LDS when pmax > 8192
    int *ncand;             // [nQ] scratch of the phased launch (few queries: scoring spread over the whole GPU)
    int phase;              // 0: whole query in one workgroup; 1: candidates; 2: scores; 3: argmax (1-3 need gkeys)
    pfann_match_result *results; float *song_scores;
    int ss_lo, ss_n;        // the song_scores block of one query covers songs [ss_lo, ss_lo + ss_n): all, or the owned ones
};
int launch_match(const RerankArgs &a, hipStream_t s);
int launch_match_pack(const pfann_match_result *res, int64_t nQ, unsigned long long *keys, hipStream_t s);
int launch_match_pick(const unsigned long long *keys, int G, int64_t nQ, pfann_match_result *out, hipStream_t s);
int launch_song_scores_to_seconds(float *ss, int64_t n_pairs, int fsm, double hop_size, int native_path, hipStream_t s);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device
int ensure_dyn_lds(const void *func, int bytes);

// one per translation unit with device code (pfann_prewarm)
int prewarm_mel();
int prewarm_encoder();
int prewarm_encoder_fused();
int prewarm_search();
int prewarm_search_f16();
int prewarm_rerank();

}  // namespace pfann
