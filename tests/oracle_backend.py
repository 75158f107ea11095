"""Test-only compute backend with the DeviceIndex interface, implemented by the ORACLE
(numpy / C restatement).  It lets the multi-GPU host logic in pfann_amd/dist.py run in
world_size-2 `gloo` tests on CPU.  Lives under tests/: the product never imports it."""
import numpy as np
import torch

from oracle import search as osr
from oracle import seqscore as osq


class OracleIndex:
    def __init__(self, d):
        self.d = d

    def load(self, emb, song_pos, label_base=0):
        self.emb = np.ascontiguousarray(emb, np.float32).reshape(-1, self.d)
        self.song_pos = np.asarray(song_pos, np.int64)
        self.label_base = label_base
        self.ntotal = self.emb.shape[0]
        self.n_songs = self.song_pos.shape[0] - 1
        lo = int(np.searchsorted(self.song_pos, label_base, side="left"))
        hi = lo
        while hi < self.n_songs and self.song_pos[hi + 1] <= label_base + self.ntotal:
            hi += 1
        self.song_lo, self.song_hi = lo, hi

    def search(self, q, k):
        D, I = osr.flat_ip_topk(q.numpy(), self.emb, k)
        I = np.where(I >= 0, I + self.label_base, -1)
        return torch.from_numpy(D), torch.from_numpy(I)

    # ---- two-phase sharded search (pfann_search_bound / pfann_search_topk_bounded): the oracle's candidates are the
    # shard's m best exact scores, the reduced bound the k-th largest of the union, and its bounded search drops every
    # entry below that bound
    BOUND_CHUNK = 7            # small on purpose: the gloo tests walk several chunks

    def search_bound(self, q, k, m=1):
        D, _ = osr.flat_ip_topk(q.numpy(), self.emb, m)            # the shard's m best exact scores: the tightest legal values
        out = np.where(D > -np.finfo(np.float32).max, D, -np.inf).astype(np.float32)
        return torch.from_numpy(np.ascontiguousarray(out[:, :m]))

    def reduce_bound(self, cands, k):
        c = cands.numpy()
        G, nq, m = c.shape
        v = -np.sort(-c.transpose(1, 0, 2).reshape(nq, G * m), axis=1)
        lb = v[:, k - 1] if G * m >= k else np.full(nq, -np.inf, np.float32)
        return torch.from_numpy(np.where(np.isfinite(lb), lb, -np.finfo(np.float32).max).astype(np.float32))

    def search_bounded(self, q, k, lb):
        D, I = self.search(q, k)
        D, I = D.numpy().copy(), I.numpy().copy()
        drop = D < lb.numpy()[:, None]
        D[drop] = -np.finfo(np.float32).max
        I[drop] = -1
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge_topk(self, S, L, k):
        S, L = S.numpy(), L.numpy()
        D = np.full((S.shape[0], k), -np.finfo(np.float32).max, np.float32)
        I = np.full((S.shape[0], k), -1, np.int64)
        for r in range(S.shape[0]):
            ok = np.nonzero(L[r] >= 0)[0]
            o = ok[np.argsort(-S[r][ok], kind="stable")[:k]]
            D[r, :len(o)] = S[r][o]
            I[r, :len(o)] = L[r][o]
        return torch.from_numpy(D), torch.from_numpy(I)

    # ---- winner keys: numpy statement of pfann_match_pack / pfann_match_pick (include/pfann_amd.h) --------------
    @staticmethod
    def _ord64(score):
        u = np.asarray(score, np.float64).view(np.uint64)
        return np.where(u >> np.uint64(63), ~u, u ^ np.uint64(1 << 63))

    def pack_winner_keys(self, res):
        hi = ~self._ord64(res["score"] + 0.0)                       # -0.0 -> +0.0
        lo = (res["shift"].astype(np.uint64) << np.uint64(58)) | (res["song"].astype(np.int64).astype(np.uint64) << np.uint64(28)) | \
            (res["offset"].astype(np.int64) + (1 << 27)).astype(np.uint64)
        none = res["song"] < 0
        hi = np.where(none, ~np.uint64(0), hi)
        lo = np.where(none, ~np.uint64(0), lo)
        refused = res["song"] == -2                                 # a refused query beats every real key
        hi = np.where(refused, np.uint64(0), hi)
        lo = np.where(refused, np.uint64(0), lo)
        return torch.from_numpy(np.stack([hi, lo], 1).view(np.int64).copy())

    def pick_winner(self, all_keys, to_host=True):
        k = all_keys.cpu().numpy().view(np.uint64)                  # [G, nQ, 2]
        G, nQ = k.shape[0], k.shape[1]
        out = np.zeros(nQ, dtype=[("song", "<i4"), ("offset", "<i4"), ("shift", "<i4"), ("n_cand", "<i4"), ("score", "<f8")])
        for j in range(nQ):
            g = min(range(G), key=lambda r: (int(k[r, j, 0]), int(k[r, j, 1])))
            hi, lo = int(k[g, j, 0]), int(k[g, j, 1])
            if hi == (1 << 64) - 1 and lo == (1 << 64) - 1:
                out[j] = (-1, 0, 0, 0, -np.inf)
                continue
            if hi == 0 and lo == 0:
                out[j] = (-2, 0, 0, 0, -np.inf)
                continue
            o = ~np.uint64(hi)
            bits = o ^ np.uint64(1 << 63) if int(o) >> 63 else ~o
            out[j] = ((lo >> 28) & 0x3FFFFFFF, (lo & 0xFFFFFFF) - (1 << 27), lo >> 58, 0, np.array([bits], np.uint64).view(np.float64)[0])
        return out

    def match(self, q, labels, qstart, qlen, fsm=1, alpha=0.0, mode=0, only_owned=False, want_song_scores=False,
              to_host=True, owned_block=False):
        """Python-path oracle restricted to owned songs: labels of other shards' songs are
        dropped before candidate generation, local rows are addressed through label_base."""
        q, labels = q.numpy(), labels.numpy()
        out = np.zeros(len(qlen), dtype=[("song", "<i4"), ("offset", "<i4"), ("shift", "<i4"),
                                         ("n_cand", "<i4"), ("score", "<f8")])
        # full-size db view so the oracle can index rows globally
        full = np.zeros((int(self.song_pos[-1]), self.d), np.float32)
        full[self.label_base:self.label_base + self.ntotal] = self.emb
        blocks = []
        for j in range(len(qlen)):
            sl = slice(int(qstart[j]), int(qstart[j]) + int(qlen[j]))
            lab = labels[sl].copy()
            if only_owned:
                own_lo, own_hi = self.song_pos[self.song_lo], self.song_pos[self.song_hi]
                lab[(lab < own_lo) | (lab >= own_hi)] = -1
            ss = np.zeros((self.n_songs, 2), np.float32)
            if (lab >= 0).any():
                score, (song, sec), ss = osq.query_embeddings_base(q[sl], lab, full, self.song_pos, 1.0, fsm)
            else:
                score, song, sec = -np.inf, -1, 0
            if want_song_scores:            # alignments in FINE FRAMES like pfann_match (hop 1.0 above: seconds == frames / fsm)
                ss = ss.copy()
                ss[:, 1] = np.round(ss[:, 1] * fsm)
                blocks.append(ss[self.song_lo:self.song_hi] if owned_block else ss)
            if song < 0:
                out[j] = (-1, 0, 0, 0, -np.inf)
            else:
                fine = int(round(sec * fsm))
                shift = (-fine) % fsm
                out[j] = (song, (fine + shift) // fsm, shift, 0, score)
        return out, (torch.from_numpy(np.stack(blocks)) if want_song_scores and blocks else None)
