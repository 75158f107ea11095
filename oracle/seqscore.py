"""Oracle a8-a9, Python path: candidate generation + sequence score + argmax
(reference database.py:117-166 query_embeddings_base), restated over a plain db matrix.
"""
import numpy as np


def song_pos_from_key(landmark_key):
    """int32 seg counts -> int64 prefix sums with leading 0 (database.py:84-86)."""
    return np.pad(np.cumsum(np.asarray(landmark_key), dtype=np.int64), (1, 0))


def query_embeddings_base(query, labels, db, song_pos, hop_size, frame_shift_mul=1):
    """-> (best: float, (song_id, real_time), song_score float32[n_songs,2])."""
    n_songs = song_pos.shape[0] - 1
    best = -1e999
    best_song_t = (-1, 0)
    song_score = np.zeros([n_songs, 2], dtype=np.float32)
    if db.shape[0] == 0:                                       # database.py:126-127
        return best, best_song_t, song_score
    for shift in range(frame_shift_mul):                        # database.py:129
        subquery = query[shift::frame_shift_mul]
        sub_len = subquery.shape[0]
        cands = []
        for t in range(sub_len):                                # database.py:133-138
            lab = labels[t * frame_shift_mul + shift]
            lab = lab[lab != -1]
            sid = np.searchsorted(song_pos, lab, side="right") - 1
            cands.append(np.stack([sid, lab - song_pos[sid] - t], axis=1))
        cands = np.unique(np.concatenate(cands), axis=0)        # sorted (song, offset)
        vec = np.zeros_like(subquery)
        for c in cands:                                         # database.py:143-163
            sid, t = int(c[0]), int(c[1])
            start = int(song_pos[sid])
            slen = int(song_pos[sid + 1]) - start
            real_time = (t - shift / frame_shift_mul) * hop_size
            for i in range(sub_len):
                if t + i < 0 or t + i >= slen:
                    vec[i] = 0.0
                else:
                    vec[i] = db[start + t + i]
            sco = np.dot(vec.flatten(), subquery.flatten()).item() / sub_len
            if sco > song_score[sid, 0]:
                song_score[sid, 0] = sco
                song_score[sid, 1] = real_time
            if sco > best:
                best = sco
                best_song_t = (sid, real_time)
    return best, best_song_t, song_score
