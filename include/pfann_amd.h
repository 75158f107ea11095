/* pfann_amd C ABI -- the drop-in boundary of the MI355X hot path.
 *
 * Plain C: pointers, sizes, opaque handles.  No torch / C++ types cross this boundary.
 * Every function returns 0 (or a value documented below) on success and a negative code
 * on failure; pfann_last_error() returns the message of the last failure on this thread.
 * No exception ever crosses the boundary.  Pointers named *_dev are device (HBM)
 * pointers, e.g. torch.Tensor.data_ptr(); *_host are host pointers; `stream` is a
 * hipStream_t passed as void* (0 = the null stream).
 *
 * Reference interfaces these entry points replace (paths relative to the reference repo):
 *   version, seq_score      cpp/seqscore.cpp:27-43 (bound by ctypes at database.py:15-32,
 *                           called at database.py:178-189)
 *   pfann_melspec           datautil/melspec.py:33-50  MelSpec.forward
 *   pfann_encode            model.py:148-153           FpNetwork.forward(x, norm)
 *   pfann_segment_embed     builder.py:88-100 / matcher.py:110-128 emit loops fused with
 *                           datautil/musicdata.py:82-88 (pad, unfold, mean removal)
 *   pfann_pcm16_to_mono     datautil/musicdata.py:48,72-80
 *   pfann_resample_to_mono  datautil/musicdata.py:28-65 (julius.ResampleFrac, minute-wise) + 72-80
 *   pfann_db_*              database.py:75-109 Database.__init__ (index + song_pos)
 *   pfann_search_topk       database.py:121  index.search(query, top_k)  (exact flat IP)
 *   pfann_match             database.py:117-166 query_embeddings_base (search + rerank)
 */
#ifndef PFANN_AMD_H
#define PFANN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFANN_SEQSCORE_VERSION 20220625002LL

/* ---- the reference's native seam, byte-identical signatures ------------------------ */

/* cpp/seqscore.cpp:27-30.  database.py:30 exits unless this equals 20220625002. */
long long version(void);

/* cpp/seqscore.cpp:32-43.  `index` is a pfann_db* (the reference passes a faiss::Index*;
 * only d + row fetch are used there).  All other pointers are HOST pointers, exactly as
 * database.py:178-189 passes them.  song_scores must arrive zeroed (database.py:176);
 * offsets are written in frames.  Returns the best song id, or -1 with no candidate. */
int seq_score(void *index, const int64_t *song_pos, int n_songs, const float *query,
              int query_len, const int64_t *labels, int top_k, float *song_scores,
              int frame_shift_mul, float score_alpha);

/* ---- errors ------------------------------------------------------------------------- */
const char *pfann_last_error(void);

/* ---- front-end + encoder ------------------------------------------------------------- */
typedef struct pfann_ctx pfann_ctx;

typedef struct pfann_config {
    /* front-end (config keys sample_rate, stft_n, stft_hop, n_mels, segment_size)        */
    int32_t segment_len;    /* samples per segment = int(sample_rate * segment_size)      */
    int32_t stft_n;         /* FFT size, power of two, 64..4096                            */
    int32_t stft_hop;
    int32_t n_mels;
    int32_t power;          /* 2 (default) or 1 (naf_mode), melspec.py:27                  */
    int32_t pad_reflect;    /* 1 reflect (default) / 0 constant zeros, melspec.py:28       */
    int32_t log_mode;       /* 1 natural log (default), 2 log10, 0 none, melspec.py:43-46  */
    int32_t spec_norm_max;  /* 0: L2 normalise (default); 1: 'max' mode, melspec.py:35,48  */
    float   log_eps;        /* 1e-8 (default) or 0.06 (naf_mode), melspec.py:38-41         */
    /* encoder (config key "model") */
    int32_t d, h, u;
    int32_t fuller;         /* 1: full conv2, 0: depthwise, model.py:26-29                 */
    int32_t activation;     /* 0 ReLU, 1 ELU, model.py:7-12                                */
    int32_t relu_after_bn;  /* model.py:58-72                                              */
    int32_t stride_t[8];    /* conv1 stride along T per block (default 2), model.py:83-85  */
    int32_t stride_f[8];    /* conv2 stride along F per block (default 2)                  */
    int32_t max_batch;      /* segments processed per internal pass (workspace size)       */
} pfann_config;

/* Creates a context on HIP device `device`; NULL on failure (see pfann_last_error). */
pfann_ctx *pfann_create(const pfann_config *cfg, int device);
void pfann_destroy(pfann_ctx *ctx);

/* Mel filterbank fb[n_freqs][n_mels] (host, row-major), n_freqs = stft_n/2+1.  The host
 * mirror builds it (pfann_amd/melspec.py) so any bank torchaudio would build can be used. */
int pfann_set_melbank(pfann_ctx *ctx, const float *fb_host, int n_freqs, int n_mels);

/* One tensor of FpNetwork.state_dict() by its reference name (e.g.
 * "f.convs.3.conv2.weight", "g.linear1.bias"), host pointer, PyTorch layout.  The library
 * re-lays it out for its kernels.  Returns -2 for an unknown name, -3 for a size mismatch. */
int pfann_load_weight(pfann_ctx *ctx, const char *name, const float *host, int64_t numel);
/* Number of state_dict tensors still missing (0 = ready). */
int pfann_weights_missing(pfann_ctx *ctx);

/* MelSpec.forward: segs_dev[b*seg_stride + i], i < segment_len  ->  out_dev[B][n_mels][T],
 * T = 1 + segment_len/stft_hop.  remove_mean=1 additionally subtracts each segment's mean
 * first (musicdata.py:88), letting callers pass overlapping windows of one waveform
 * (seg_stride = hop) instead of a materialised unfold. */
int pfann_melspec(pfann_ctx *ctx, const float *segs_dev, int64_t B, int64_t seg_stride,
                  int remove_mean, float *out_dev, void *stream);

/* FpNetwork.forward(x, norm): mel_dev[B][n_mels][T] -> emb_dev[B][d]. */
int pfann_encode(pfann_ctx *ctx, const float *mel_dev, int64_t B, float *emb_dev,
                 int normalize, void *stream);

/* Fused segmenter tail + MelSpec + FpNetwork: B windows of wav_dev at stride seg_stride. */
int pfann_segment_embed(pfann_ctx *ctx, const float *wav_dev, int64_t B, int64_t seg_stride,
                        float *emb_dev, int normalize, void *stream);

/* Same, with explicit window starts: window b = wav_dev[starts_dev[b] .. +segment_len)
 * (many queries concatenated in one buffer; windows never straddle two recordings). */
int pfann_segment_embed_at(pfann_ctx *ctx, const float *wav_dev, const int64_t *starts_dev,
                           int64_t B, float *emb_dev, int normalize, void *stream);

/* int16 interleaved PCM -> float32 mono (x/32768, fake-stereo fix, channel mean). */
int pfann_pcm16_to_mono(pfann_ctx *ctx, const int16_t *pcm_dev, int64_t n_frames, int n_ch,
                        float *wav_dev, void *stream);

/* Many MONO files at the model's rate in one call (the per-song loop of builder.py:75-103 / matcher.py:87-110 spends its
 * host time per file): file i = n_samples[i] int16 samples at host_pcm[i] (pinned host memory uploads asynchronously)
 * is copied to pcm_dev[dst_off[i] ..] and the whole slab pcm_dev[0 .. total) is converted to wav_dev (x/32768) by ONE
 * launch.  Gaps between files (zero padding of short files, slots other paths fill afterwards) are the caller's. */
int pfann_pcm16_files_to_mono(pfann_ctx *ctx, const void *const *host_pcm, const int64_t *n_samples, const int64_t *dst_off,
                              int n_files, int16_t *pcm_dev, int64_t total, float *wav_dev, void *stream);

/* ---- host-side WAV input (no GPU involved): the reference's decode workers (DataLoader(num_workers=4) over
 * MusicDataset, builder.py:66; datautil/audio.py:130-149: `wave` module, 16-bit PCM only) as native threads that fill
 * one pinned slab per launch group.  pfann_wav_probe walks each file's RIFF chunks ("fmt " then "data", other chunks
 * skipped) and reports the frames that are really in the file; pfann_wav_read reads file i's n_frames*n_ch interleaved
 * samples to dst + dst_off[i] (int16 units).  status: 0 ok, or one of the codes below (such a file is what the reference
 * logs as a load error and treats as a 0-segment song, musicdata.py:95-101).  Both return 0, or <0 on bad arguments
 * (-2: a file does not fit dst_cap). */
typedef struct pfann_wav_info {
    int64_t n_frames;      /* per channel */
    int64_t data_pos;      /* byte offset of the samples in the file */
    int32_t n_ch;
    int32_t sample_rate;
    int32_t status;
    int32_t reserved;
} pfann_wav_info;
#define PFANN_WAV_EOPEN (-1)    /* cannot open / stat */
#define PFANN_WAV_EFORMAT (-2)  /* not RIFF/WAVE, or fmt / data chunk missing */
#define PFANN_WAV_ECODEC (-3)   /* not PCM (format tag != 1), or no channels */
#define PFANN_WAV_EWIDTH (-4)   /* not 16-bit samples */
#define PFANN_WAV_EREAD (-5)    /* read error */
int pfann_wav_probe(const char *const *paths, int n, int n_threads, pfann_wav_info *info);
int pfann_wav_read(const char *const *paths, int n, int n_threads, pfann_wav_info *info, const int64_t *dst_off,
                   int16_t *dst, int64_t dst_cap);

/* Files at another sample rate (datautil/musicdata.py:28-65: `julius.ResampleFrac(file_sr, sample_rate)`, applied to 60 s
 * pieces that start every 59 s, half a second dropped at the inner seams), then the same mono conversion.
 *   pcm_dev      int16 interleaved [n_in][n_ch]
 *   kernels_dev  float [new_rate][2*width + old_rate]: the resampler's polyphase filters for the gcd-reduced rates
 *                (pfann_amd/resample.py builds them as julius does)
 *   plan_dev     int64 [n_pieces][5] = {in_start, in_len, out_skip, out_keep, out_off} per piece, in samples
 *   tmp_dev      float [n_ch][n_out] scratch;  wav_dev float [n_out] result.
 * julius is an un-vendored, unpinned dependency of the reference: this path is restated from its published algorithm. */
int pfann_resample_to_mono(pfann_ctx *ctx, const int16_t *pcm_dev, int n_ch, const float *kernels_dev, int old_rate,
                           int new_rate, int width, const int64_t *plan_dev, int n_pieces, int64_t n_out, float *tmp_dev,
                           float *wav_dev, void *stream);

/* Debug/verification taps: copy the activation after sub-layer `idx` (0..15) of the LAST
 * pfann_encode call's first `B` samples to host as NCHW floats.  Returns numel or <0. */
int64_t pfann_debug_activation(pfann_ctx *ctx, int idx, int64_t B, float *host, int64_t cap);
/* Enable keeping those taps (costs one D2D copy per sub-layer; off by default). */
void pfann_debug_keep(pfann_ctx *ctx, int on);

/* Selects the LayerNorm-fused encoder path (default: on whenever the model supports it, i.e.
 * every conv2 is a full conv; off = separate LayerNorm kernels).  Returns the state now in
 * effect (1 fused / 0 unfused).  Both paths are parity-tested. */
int pfann_set_fused_layernorm(pfann_ctx *ctx, int on);

/* Arithmetic of the fused conv GEMMs.  0 (default): fp32 MFMA, bitwise an fmaf chain.  1: every operand as two
 * fp16 terms (x = hi + lo to 2^-22 relative; weights pre-scaled by a power of two), three fp16 MFMAs per product
 * (hi*hi + lo*hi + hi*lo) with fp32 accumulation -- fp32-grade results (embeddings within ~1e-6 of mode 0,
 * far inside the 1e-4 parity bar) at 3/16 of the MFMA cycles.  Needs the fused path.  Returns the mode now in
 * effect.  New capability: the reference computes these convolutions in fp32 (model.py:54-73). */
int pfann_set_encoder_precision(pfann_ctx *ctx, int mode);

/* Makes the HIP runtime initialise `device` and load every code object of this library now (one empty launch per
 * translation unit) instead of at each unit's first real launch: about 0.45 s that the drop-in tools spend on a thread
 * while the interpreter is still importing torch.  Needs no context; safe to call more than once.  0 / -1. */
int pfann_prewarm(int device);

/* Kernel-variant plan of the encoder.  By default every call picks its GEMM tile size, the split-K path and the
 * small-batch head from its OWN batch size, which makes the last bits of a fingerprint depend on the batch it was
 * computed in (different summation orders; all within 5e-6 of the fp32 reference).  pfann_set_plan_batch(ctx, n) with
 * n > 0 makes every later call pick the variants a batch of n segments would get (n < 65 is raised to 65), whatever
 * its own size: a segment then has bit-identical fingerprints in every batch -- the drop-in CLIs set n = their launch
 * group size, which is what makes their outputs byte-identical for any number of ranks and any grouping.  n = 0
 * restores the default.  Returns the value in effect.  New capability (the reference's PyTorch kernels make no such
 * promise either way; SURVEY 8a "batch independence"). */
int64_t pfann_set_plan_batch(pfann_ctx *ctx, int64_t n);

/* Number of internal HIP streams (1..8) a batch is split over inside pfann_encode /
 * pfann_segment_embed*: the MFMA-bound GEMMs of one sub-batch overlap the HBM-bound passes
 * of another.  Work is forked from and joined back into the caller's stream.  Returns n. */
int pfann_set_streams(pfann_ctx *ctx, int n);

/* ---- database: device-resident fingerprints, exact search, sequence match ------------ */
typedef struct pfann_db pfann_db;

pfann_db *pfann_db_create(int d, int device);
void pfann_db_destroy(pfann_db *db);
int pfann_db_dim(pfann_db *db);
int64_t pfann_db_ntotal(pfann_db *db);

/* Loads rows emb[n][d] (host pointer if emb_is_device==0, else device pointer; copied) and
 * the int64 prefix sums song_pos_host[n_songs+1] (database.py:84-86).  `label_base` is
 * added to every label this shard reports (song-sharded multi-GPU: global row id of local
 * row 0); song_pos stays GLOBAL and the shard must start and end on song boundaries. */
int pfann_db_load(pfann_db *db, const float *emb, int emb_is_device, int64_t n,
                  const int64_t *song_pos_host, int n_songs, int64_t label_base);

/* Storage precision of the shard's rows; call BEFORE pfann_db_load.  Returns the mode in effect, <0 on error.
 *   PFANN_DB_F32 (default): fp32 rows (+ an fp16 copy for the pre-filter below); every result is exact fp32.
 *   PFANN_DB_F16: ONLY fp16 rows are kept (n*d*2 bytes, half the HBM footprint and half the bytes per scan
 *     pass).  Search returns the k best s16 = sum_i fl16(q_i)*fl16(x_i) (exact products, fp32 accumulation on
 *     v_mfma_f32_32x32x16_f16) with no fp32 re-scoring; the sequence matcher scores against the stored fp16
 *     rows.  Approximate with respect to the fp32 path.  This goes FURTHER than the reference's only fp16
 *     precedent, faiss' GpuMultipleClonerOptions.useFloat16 (database.py:101-104; cpp/faisscputest.cpp:97-108):
 *     there only the GPU search index is fp16 and the rerank still reconstructs fp32 rows from the CPU index
 *     (database.py:148-152); here the rerank scores -- and so tie / argmax decisions between near-equal
 *     candidates -- also carry fp16 rounding of the database rows (measured: 99.85 % identical decisions on
 *     BASELINE config 5).  Keep PFANN_DB_F32 when the reference's exact scores are wanted.  d % 8 == 0. */
#define PFANN_DB_F32 0
#define PFANN_DB_F16 1
int pfann_db_set_storage(pfann_db *db, int mode);

/* Batches of more than 64 query rows are scanned on the fp16 matrix cores with a rigorous error
 * margin and re-scored in exact fp32 (csrc/search_f16.hip): the result is the exact fp32 top-k
 * either way.  on=0 forces the all-fp32 scan.  Returns 1 if the pre-filter is now in use. */
int pfann_db_set_prefilter(pfann_db *db, int on);

/* Exact inner-product top-k of q_dev[nq][d] over the shard: D_dev[nq][k] descending,
 * I_dev[nq][k] int64 labels (+label_base); unfilled slots D=-FLT_MAX, I=-1.  Asynchronous on `stream`: the
 * call never synchronises with the host (rows whose survivor lists overflow -- thousands of ties at the k-th
 * score -- are recomputed exactly by a device-side fallback kernel). */
int pfann_search_topk(pfann_db *db, const float *q_dev, int64_t nq, int k, float *D_dev,
                      int64_t *I_dev, void *stream);

/* The same search split around ONE collective, for a database sharded over several GPUs (pfann_amd/dist.py; the
 * reference has no counterpart: database.py:101-104 replicates the index).  nq <= 16384 per call.
 *   pfann_search_bound        : query preparation + the sampled pass only; lb_dev[nq][m] = for every query row the m
 *                               best sampled scores of THIS shard (one per group of rows, so m different real rows),
 *                               each lowered to a bound of its exact inner product; -inf padding (all -inf where the
 *                               path taken has no sampled threshold: tiny shards, <= 32 query rows, k > 128).
 *   -- the caller gathers the ranks' values; the k-th largest of their union bounds the k-th best over all shards
 *      from below (dist.py sends m = 2k/ranks + 8 values per row and rank) --
 *   pfann_search_topk_bounded : finishes the search started by pfann_search_bound for the SAME (q_dev, nq, k) on the
 *                               same handle (nothing else may use the handle's search in between), lb_dev[nq] = the
 *                               reduced bound; rows that cannot reach it are not emitted, so D / I may hold fewer than k entries
 *                               (D = -FLT_MAX, I = -1 padding).  pfann_topk_merge of the shards' lists is the exact
 *                               global top-k.  Without a matching pfann_search_bound call it is pfann_search_topk. */
int pfann_search_bound(pfann_db *db, const float *q_dev, int64_t nq, int k, int m, float *lb_dev, void *stream);
int pfann_search_topk_bounded(pfann_db *db, const float *q_dev, int64_t nq, int k, const float *lb_dev,
                              float *D_dev, int64_t *I_dev, void *stream);

/* Exact top-k of arbitrary (score,label) lists: in[nq][m] -> out[nq][k] (merging per-shard
 * top-k lists after an all-gather).  Entries with label<0 are ignored. */
int pfann_topk_merge(pfann_db *db, const float *S_dev, const int64_t *L_dev, int64_t nq, int m,
                     int k, float *D_dev, int64_t *I_dev, void *stream);

/* The two reductions of the sharded search in the layouts the collectives deliver, one wavefront per query row:
 *   pfann_bound_reduce     : cands_dev[n_ranks][nq][m] (the all-gathered pfann_search_bound outputs) -> lb_dev[nq], the k-th
 *                            largest of each row's n_ranks * m values (-inf entries are absent; fewer than k present:
 *                            -FLT_MAX) -- what pfann_search_topk_bounded wants.  n_ranks * m <= 1024.
 *   pfann_topk_merge_lists : D_lists_dev / I_lists_dev[n_lists][nq][k] (every shard's list for these query rows, as the
 *                            all-to-all delivers them; label < 0 = padding) -> the exact top-k of the union, ordered like
 *                            pfann_topk_merge over the shard-major concatenation (ties: lower list first, then list order).
 *                            k <= 128, n_lists * k <= 1024. */
int pfann_bound_reduce(pfann_db *db, const float *cands_dev, int n_ranks, int64_t nq, int m, int k, float *lb_dev,
                       void *stream);
int pfann_topk_merge_lists(pfann_db *db, const float *D_lists_dev, const int64_t *I_lists_dev, int n_lists, int64_t nq,
                           int k, float *D_dev, int64_t *I_dev, void *stream);

/* Result of the sequence matcher for one query. */
typedef struct pfann_match_result {
    int32_t song;        /* best song id (global), -1 if no candidate                     */
    int32_t offset;      /* best offset in sub-query frames (t), python path              */
    int32_t shift;       /* frame shift of the best candidate                             */
    int32_t n_cand;      /* unique candidates scored                                       */
    double  score;       /* python path: fp32 dot / sub_len in double (database.py:157)   */
} pfann_match_result;

/* Candidate generation + sequence score + argmax for nQ queries at once, given labels.
 * Query j owns rows [qstart[j], qstart[j]+qlen[j]) of q_dev / labels_dev[.][k].
 * mode 0 = python path (database.py:129-163), mode 1 = native path (seqscore.cpp:49-135,
 * fp32 divide, offsets t*fsm-shift, score_alpha honoured).
 * results_dev[nQ]; song_scores_dev[nQ][n_songs][2] may be NULL; when given it must be
 * zeroed by the caller and receives (score, offset-in-frames) of songs owned by the shard.
 * only_owned: bit 0 restricts candidates to songs of this shard (multi-GPU rerank); bit 1 (PFANN_MATCH_OWNED_BLOCK,
 * only together with bit 0) makes song_scores_dev a [nQ][owned songs][2] block (pfann_db_owned_songs) instead of
 * [nQ][n_songs][2]: the shard's columns of the matcher's `.bin` matrix and nothing else.
 * max_qlen = largest qlen[j]: candidate lists of up to 8192 (qlen*k) entries are sorted in LDS,
 * longer ones (e.g. a 60 s query at k=100) in a per-query HBM slab the handle grows on demand;
 * a query longer than max_qlen gets song=-2. */
int pfann_match(pfann_db *db, const float *q_dev, const int64_t *labels_dev, int k,
                const int64_t *qstart_dev, const int32_t *qlen_dev, int64_t nQ, int max_qlen,
                int frame_shift_mul, float score_alpha, int mode, int only_owned,
                pfann_match_result *results_dev, float *song_scores_dev, void *stream);

#define PFANN_MATCH_ONLY_OWNED 1
#define PFANN_MATCH_OWNED_BLOCK 2

/* Songs whose rows all live in this shard: [*song_lo, *song_hi) (either pointer may be NULL); returns their number. */
int pfann_db_owned_songs(pfann_db *db, int *song_lo, int *song_hi);

/* pfann_db_load derives the owned songs from the shard's row range; songs WITHOUT rows (unreadable files: builder.py
 * writes a 0 into landmarkKey) that sit at a shard boundary are then ambiguous.  A caller that cut the song list itself
 * (pfann_amd/dist.py: shard_songs) states its cut here, after pfann_db_load: [song_lo, song_hi) must span exactly the
 * shard's rows (-3 otherwise).  Owner-side matching (PFANN_MATCH_ONLY_OWNED) and the owned score block
 * (PFANN_MATCH_OWNED_BLOCK) then use this range. */
int pfann_db_set_owned_songs(pfann_db *db, int song_lo, int song_hi);

/* In place, for n_pairs (score, alignment) pairs of a song_scores block written by pfann_match: the alignment slot
 * goes from fine frames (t * frame_shift_mul - shift) to seconds.  native_path 0: (t - shift / frame_shift_mul) * hop_size
 * computed in double and stored as float32 -- what database.py:148,160 leaves in song_score[:, 1]; native_path 1: the
 * float32 multiply of database.py:193, song_score[:, 1] *= hop_size / frame_shift_mul.  Asynchronous on `stream`. */
int pfann_song_scores_to_seconds(pfann_db *db, float *song_scores_dev, int64_t n_pairs, int frame_shift_mul,
                                 double hop_size, int native_path, void *stream);

/* Song-sharded multi-GPU retrieval, winner selection without the host (no reference counterpart):
 * pfann_match_pack turns this rank's results_dev[nQ] (from pfann_match with only_owned=1, python path) into one
 * 128-bit key per query, keys_dev[nQ][2] = (hi, lo) uint64, whose unsigned lexicographic order is the reference's
 * preference -- higher score first, ties to the smallest (shift, song, offset), the order of its np.unique-sorted
 * candidate list (database.py:129,140,158-163); a query without candidates gets all ones.  After an all-gather of the
 * keys (16 bytes per query and rank), pfann_match_pick reduces keys_dev[n_ranks][nQ][2] to the winners, out_dev[nQ]
 * (n_cand = 0).  Both are asynchronous on `stream`. */
int pfann_match_pack(pfann_db *db, const pfann_match_result *results_dev, int64_t nQ, uint64_t *keys_dev, void *stream);
int pfann_match_pick(pfann_db *db, const uint64_t *keys_dev, int n_ranks, int64_t nQ, pfann_match_result *out_dev,
                     void *stream);

/* Bytes of the shard's fingerprint matrix as stored (n*d*4, or n*d*2 with fp16 storage). */
int64_t pfann_db_bytes(pfann_db *db);

/* Timing hooks for bench.py: HIP events on the caller's stream around a tagged region.
 * With profiling enabled every kernel launch is bracketed by an event pair tagged with
 * the kernel's name ("conv_gemm", "scan_topk", "ln_act", ...); pfann_prof_elapsed_ms sums
 * all completed brackets of that tag since pfann_prof_reset, returning their count in *count. */
void pfann_prof_enable(int on);
/* Launches an empty kernel named pfann_bench_region_marker on `stream`: bench.py brackets its
 * timed region with two of them so a rocprofv3 trace can be cut to exactly that region. */
void pfann_prof_marker(void *stream);
void pfann_prof_reset(void);
double pfann_prof_elapsed_ms(const char *tag, int64_t *count);
/* Sum of the algorithmic work (flops for the MFMA-bound GEMM kernels, HBM bytes for the
 * streaming ones) of all launches recorded under `tag` since the last reset. */
double pfann_prof_work(const char *tag);
/* Comma-separated tags recorded since the last reset; returns their number or -1. */
int pfann_prof_tags(char *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* PFANN_AMD_H */
