#!/usr/bin/env python
"""Seeded synthetic stand-in for the reference's dataset + genquery.py protocol (no datasets exist
here): writes <out>/music/*.wav + <out>/music.txt, and per SNR <out>/query_snr<S>/{*.wav,list.txt,
expected.csv} (columns query,answer,time,snr as genquery.py:139-160 writes them).
    python tools/gen_synth_dataset.py <out dir> [--songs 20] [--queries 40] [--seconds 10] [--snr 0 ...]"""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfann_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--songs", type=int, default=20)
    ap.add_argument("--queries", type=int, default=40)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--song-seconds", type=float, default=30.0)
    ap.add_argument("--snr", type=float, nargs="*", default=[0.0])
    a = ap.parse_args()
    md = os.path.join(a.out, "music")
    os.makedirs(md, exist_ok=True)
    songs, paths = [], []
    for s in range(a.songs):
        pcm = synth.make_song(s, a.song_seconds)
        p = os.path.abspath(os.path.join(md, "song%05d.wav" % s))
        synth.write_wav(p, pcm)
        songs.append(pcm)
        paths.append(p)
    open(os.path.join(a.out, "music.txt"), "w").write("".join(p + "\n" for p in paths))
    for snr in a.snr:
        qd = os.path.join(a.out, "query_snr%g" % snr)
        os.makedirs(qd, exist_ok=True)
        rows, names = [], []
        for j in range(a.queries):
            s = j % a.songs
            q, off = synth.make_query(songs[s], j, a.seconds, snr)
            p = os.path.abspath(os.path.join(qd, "q%05d.wav" % j))
            synth.write_wav(p, q)
            names.append(p)
            rows.append([p, paths[s], off, snr])
        open(os.path.join(qd, "list.txt"), "w").write("".join(p + "\n" for p in names))
        with open(os.path.join(qd, "expected.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["query", "answer", "time", "snr"])
            w.writerows(rows)
    print("wrote", a.songs, "songs and", a.queries, "queries x", len(a.snr), "SNRs under", a.out)


if __name__ == "__main__":
    main()
