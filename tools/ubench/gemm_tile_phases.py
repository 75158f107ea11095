"""Phase overlap of co-resident workgroups from gemm_tile's TS_OUT dump (tuning aid).
For every CU: fraction of the busy time with 0 / 1 / 2 workgroups inside the K loop."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)
hw = a[:, 4].astype(np.int64)
xcc = (a[:, 5] & np.uint64(0xF)).astype(np.int64)
lds = (a[:, 5] >> np.uint64(32)).astype(np.int64) & 0xFF
cu = (xcc << 16) | (hw & 0xFF00)            # cu_id 11:8, sh_id 12, se_id 15:13
print("tiles", len(a), "CUs", len(np.unique(cu)), "lds_base values", np.unique(lds)[:8], "wave_id values", np.unique(hw & 0xF))
t = a[:, :4].astype(np.int64)
print("cycles per tile: prologue %.0f  loop %.0f  epilogue %.0f  total %.0f" % tuple(
    np.mean(x) for x in (t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0])))
tot = np.zeros(3)
res = np.zeros(3)
for c in np.unique(cu)[:64]:
    m = cu == c
    tt = t[m]
    ev = []
    for s0, l0, l1, e1 in tt:
        ev += [(l0, 1, 0), (l1, -1, 0), (s0, 0, 1), (e1, 0, -1)]
    ev.sort()
    inl = resd = 0
    prev = ev[0][0]
    for when, dl, dr in ev:
        if resd > 0:
            tot[min(inl, 2)] += when - prev
            res[min(resd, 2)] += when - prev
        prev = when
        inl += dl
        resd += dr
print("time with 0/1/2 workgroups in the K loop: %.3f %.3f %.3f" % tuple(tot / tot.sum()))
print("time with 1/2 workgroups resident: %.3f %.3f" % tuple(res[1:] / res.sum()))
# lag between the two co-resident workgroups' loop starts, relative to the tile period
# progress rates: a workgroup's 12 K-tiles = r2 * (cycles with the partner also in its loop) + r1 * (cycles without)
X = []
for c in np.unique(cu)[:128]:
    tt = t[cu == c]
    l0, l1 = tt[:, 1], tt[:, 2]
    for i in range(len(tt)):
        ov = np.clip(np.minimum(l1, l1[i]) - np.maximum(l0, l0[i]), 0, None)
        ov[i] = 0
        x = ov.sum()
        X.append((x, (l1[i] - l0[i]) - x))
X = np.array(X, dtype=np.float64)
nk = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
r, *_ = np.linalg.lstsq(X, np.full(len(X), nk), rcond=None)
print("K-tile rate shared %.3e /cycle (pipe use by both %.3f), alone %.3e /cycle (pipe use %.3f)" % (r[0], 2 * r[0] * 4096, r[1], r[1] * 4096))
print("mean cycles of a loop with partner in loop %.0f, without %.0f" % (X[:, 0].mean(), X[:, 1].mean()))
