/* Oracle (TEST INFRASTRUCTURE): plain-C restatement of the reference's native sequence
 * scorer, cpp/seqscore.cpp:23-136 (`seq_score`), plus an exact flat inner-product top-k
 * (the faiss IndexFlatIP.search semantics the reference relies on at database.py:121).
 *
 * PARITY UNPINNED BY EXECUTION: cpp/seqscore.cpp includes <faiss/Index.h> and links
 * libfaiss; neither exists in this image, so the reference file cannot be built here.
 * This restatement is pinned (tests/test_oracle.py) by the probed outputs SURVEY.md §8c
 * records for the compiled reference and by agreement with oracle/seqscore.py.
 *
 * The "index" is a flat row-major float matrix (what faiss::Index::reconstruct returns
 * row by row, seqscore.cpp:96).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int song, t, shift; } cand_t;

static int cand_cmp(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->song != y->song) return x->song < y->song ? -1 : 1;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    if (x->shift != y->shift) return x->shift < y->shift ? -1 : 1;
    return 0;
}

/* largest s with song_pos[s] <= idx, over song_pos[0..n_songs)  (seqscore.cpp:23-25) */
static int row_to_song(const int64_t *song_pos, int n_songs, int64_t idx) {
    int lo = 0, hi = n_songs;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (song_pos[mid] <= idx) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

/* Same argument meaning and return as seq_score (seqscore.cpp:33-43), with the faiss
 * index replaced by (db, d). */
int oracle_seq_score(const float *db, int d, const int64_t *song_pos, int n_songs,
                     const float *query, int query_len, const int64_t *labels, int top_k,
                     float *song_scores, int frame_shift_mul, float score_alpha) {
    size_t cap = (size_t)query_len * (size_t)top_k, n = 0;
    cand_t *c = (cand_t *)malloc((cap ? cap : 1) * sizeof(cand_t));
    for (int t = 0; t < query_len; t++) {                       /* seqscore.cpp:49-60 */
        int tim = t / frame_shift_mul, shift = t % frame_shift_mul;
        for (int i = 0; i < top_k; i++) {
            int64_t lab = labels[(size_t)t * top_k + i];
            if (lab < 0) continue;
            int s = row_to_song(song_pos, n_songs, lab);
            c[n].song = s; c[n].t = (int)(lab - song_pos[s] - tim); c[n].shift = shift; n++;
        }
    }
    qsort(c, n, sizeof(cand_t), cand_cmp);                      /* seqscore.cpp:61-62 */
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (m == 0 || cand_cmp(&c[m - 1], &c[i]) != 0) c[m++] = c[i];
    n = m;

    float *score = (float *)malloc((n ? n : 1) * sizeof(float));
    float *toff = (float *)malloc((n ? n : 1) * sizeof(float));
    float best = -INFINITY; int best_song = -1;
    for (size_t i = 0; i < n; i++) {                            /* seqscore.cpp:79-125 */
        int s = c[i].song;
        score[i] = 0; toff[i] = 0;
        if (s >= n_songs || s < 0) continue;
        int slen = (int)(song_pos[s + 1] - song_pos[s]);
        int64_t start = song_pos[s];
        int t = c[i].t, shift = c[i].shift;
        float sco = 0;
        int my_len = (query_len - shift + frame_shift_mul - 1) / frame_shift_mul;
        for (int j = 0; j < my_len; j++) {
            int qi = j * frame_shift_mul + shift;
            if (t + j < 0 || t + j >= slen) continue;
            const float *v = db + (size_t)(start + t + j) * d;
            float ip = 0;
            for (int k = 0; k < d; k++) ip += v[k] * query[(size_t)qi * d + k];
            float l2 = 1.0f - 1.0f * ip;
            if (score_alpha == 0.0f) sco += ip;
            else if (score_alpha > 0.0f) sco += expf(-score_alpha * l2 * l2);
        }
        sco /= (float)(my_len > 1 ? my_len : 1);
        score[i] = sco;
        toff[i] = (float)(t * frame_shift_mul - shift);
        /* serial form of the per-thread max + critical-section merge
         * (seqscore.cpp:115-124): highest score, ties -> smaller song id */
        if (sco > best || (sco == best && s < best_song)) { best = sco; best_song = s; }
    }
    for (size_t i = 0; i < n; i++) {                            /* seqscore.cpp:126-133 */
        int s = c[i].song;
        if (s >= n_songs || s < 0) continue;
        if (score[i] > song_scores[s * 2]) {
            song_scores[s * 2] = score[i];
            song_scores[s * 2 + 1] = toff[i];
        }
    }
    free(c); free(score); free(toff);
    return best_song;
}

/* Exact flat inner-product top-k, one query row at a time with a bounded min-heap.
 * D descending, ties -> smaller label; unfilled slots: D=-FLT_MAX, I=-1. */
typedef struct { float s; int64_t i; } hit_t;
static int worse(hit_t a, hit_t b) { return a.s < b.s || (a.s == b.s && a.i > b.i); }
static void sift_down(hit_t *h, int n, int p) {
    for (;;) {
        int l = 2 * p + 1, r = l + 1, m = p;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == p) return;
        hit_t t = h[p]; h[p] = h[m]; h[m] = t; p = m;
    }
}
static int hit_cmp_desc(const void *a, const void *b) {
    hit_t x = *(const hit_t *)a, y = *(const hit_t *)b;
    return worse(x, y) ? 1 : (worse(y, x) ? -1 : 0);
}

void oracle_flat_ip_topk(const float *db, int64_t n, int d, const float *query, int nq, int k,
                         float *D, int64_t *I) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < nq; q++) {
        hit_t *h = (hit_t *)malloc((size_t)k * sizeof(hit_t));
        int cnt = 0;
        const float *qv = query + (size_t)q * d;
        for (int64_t r = 0; r < n; r++) {
            const float *v = db + (size_t)r * d;
            float ip = 0;
            for (int j = 0; j < d; j++) ip += v[j] * qv[j];
            hit_t x = { ip, r };
            if (cnt < k) {
                h[cnt++] = x;
                if (cnt == k) for (int p = k / 2 - 1; p >= 0; p--) sift_down(h, k, p);
            } else if (worse(h[0], x)) { h[0] = x; sift_down(h, k, 0); }
        }
        qsort(h, cnt, sizeof(hit_t), hit_cmp_desc);
        for (int j = 0; j < k; j++) {
            D[(size_t)q * k + j] = j < cnt ? h[j].s : -3.4028234663852886e38f;
            I[(size_t)q * k + j] = j < cnt ? h[j].i : -1;
        }
        free(h);
    }
}
