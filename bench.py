#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X: query segments/sec (+ top-1
hit-rate) for 10 s @ SNR 0 queries against a 1 M-segment fingerprint database.

One "step" = one pass of the whole hot path over one batch of synthetic queries whose
int16 PCM is already resident in HBM:
    PCM -> mono float -> 1 s windows (0.5 s hop) -> log-mel -> CNN encoder -> unit-norm
    128-d fingerprints -> exact inner-product top-100 over the db -> sequence matcher
    -> (song, offset) decisions on the host.
The database is REAL: every one of the 16,950 synthetic songs (1,000,050 segments) is embedded by the path itself,
through the builder's own loop (pfann_amd.builder.embed_files, host PCM in -> fingerprints in HBM), whose throughput
is reported as `builder`.  `value` is SURVEY 8(d)'s metric: wall time from PCM in (pinned) HOST memory to decisions on
the host, i.e. the H2D of the step's query PCM is inside the timed region; the same step with the PCM already resident in
HBM is reported beside it as `hbm_resident` (about 1 % faster).

    python bench.py --gpus N --steps K --warmup W           (N > 1: starts its own N ranks, one per GPU, RCCL)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...     (the driver's form)

The job is the same at every N (strong scaling, the default): one 1M-segment db, 4096 ten-second queries per step.
N > 1: the db is sharded by songs over the ranks; each rank embeds 1/N of the step's query windows, the fingerprints
and the per-shard top-k lists are exchanged over RCCL, each rank sequence-scores the candidates it owns.  Per N the
line also carries `scan_throughput` (db rows x query rows per second of scan-kernel time, max over ranks) and the
event time of every collective of the exchange protocol.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

SEG_PER_SONG = 59            # 30 s song, 1 s window, 0.5 s hop (reference musicdata.py:87)
QUERY_SEGS = 19              # 10 s query (matcher.py:109)
GEMM_FLOP_PER_SEG = 2 * (291.02e6 - 1.57e6 - 0.037e6)   # the 15 implicit-GEMM convs
ENC_FLOP_PER_SEG = 0.58204e9                             # whole encoder
PEAK_F32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM = 8000.0            # GB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


CLI_LEG_TIMEOUT_S = 1500            # per tool run of the CLI leg (tools/cli_bench.py) when bench.py starts it


def cpu_baseline_leg(args, params, sd, shard, song_pos, q_pcm_mine, res, k, n_rows):
    """The oracle (a CPU restatement of the reference's path: `kind` = "port") timed on this box's host cores on a bounded
    sample of the same queries against the same database, split by the reference's stage names (tools/stat.py:17):
    compute embedding = torch-CPU mel + encoder, search = BLAS sgemm + argpartition top-k (the IndexFlatIP stand-in;
    faiss-cpu is not installable here), rerank = the C restatement of cpp/seqscore.cpp (OpenMP).  The thread count is
    swept on a small probe and the best setting runs the sample; also: encoder rates at the reference's batch sizes
    (builder 32, matcher 16) and the reference's default Python double-loop rerank (database.py:143-163) on 3 queries."""
    import torch
    from threadpoolctl import threadpool_limits
    from oracle import encoder as oe
    from oracle import melspec as om
    from oracle import native, search as osr, segmenter as osg, seqscore as osq
    nq_cpu = min(args.cpu_queries, q_pcm_mine.shape[0])
    db_host = shard.cpu().numpy()
    q_pcm = q_pcm_mine[:nq_cpu].cpu().numpy()
    native.lib()
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()

    def run(js, nthreads):
        """-> (stage seconds dict, decisions) for queries js with `nthreads` torch / BLAS / OpenMP threads"""
        torch.set_num_threads(nthreads)
        st = {"compute embedding": 0.0, "search": 0.0, "rerank": 0.0}
        dec = []
        with threadpool_limits(limits=nthreads):
            for j in js:
                t0 = time.perf_counter()
                segs = osg.segment(osg.pcm_to_mono(q_pcm[j][:, None]), 8000, 4000)
                e = oe.encode(om.melspec(segs, params), sd, params)
                t1 = time.perf_counter()
                Dc, Ic = osr.flat_ip_topk_blas(e, db_host, k)
                t2 = time.perf_counter()
                best, ss = native.seq_score(db_host, song_pos, e, Ic, 1, 0.0)
                t3 = time.perf_counter()
                st["compute embedding"] += t1 - t0
                st["search"] += t2 - t1
                st["rerank"] += t3 - t2
                dec.append((best, int(ss[best, 1]) if best >= 0 else 0, e, Ic))
        return st, dec

    sweep = {}
    # all logical cores only up to 128: on this pool's 256-thread hosts the oracle runs at ~1 segment/s with 256
    # threads (oversubscribed BLAS + OpenMP), a 76 s probe for a number nobody would pick (profiles/r3/bench.json)
    for nt in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu), min(default_threads, 128), min(ncpu, 128)}):
        run([0], nt)                                                       # warm this setting
        stp, _ = run(list(range(min(4, nq_cpu))), nt)
        sweep[nt] = round(min(4, nq_cpu) * QUERY_SEGS / sum(stp.values()), 1)
    best_nt = max(sweep, key=lambda t: sweep[t])
    tc = time.perf_counter()
    st, dec = run(list(range(nq_cpu)), best_nt)
    tcpu = time.perf_counter() - tc
    agree = sum(1 for j, (b, off, _, _) in enumerate(dec) if b == int(res[j]["song"]) and off == int(res[j]["offset"]))
    nseg = nq_cpu * QUERY_SEGS
    # encoder alone at the reference's batch sizes (builder.py:88: 32, matcher.py:110: 16)
    enc = {}
    torch.set_num_threads(best_nt)
    segs = osg.segment(osg.pcm_to_mono(np.concatenate([q_pcm[j] for j in range(min(2, nq_cpu))])[:, None]), 8000, 4000)[:32]
    mel = om.melspec(segs, params)
    for bsz in (16, 32):
        if mel.shape[0] >= bsz:
            oe.encode(mel[:bsz], sd, params)
            t0 = time.perf_counter()
            for _ in range(3):
                oe.encode(mel[:bsz], sd, params)
            enc["batch_%d" % bsz] = round(3 * bsz / (time.perf_counter() - t0), 1)
    # the reference's DEFAULT rerank: the Python double loop over candidates and rows (cpp_accelerate = False)
    npy = min(3, nq_cpu)
    t0 = time.perf_counter()
    py_same = 0
    for j in range(npy):
        _, _, e, Ic = dec[j]
        sco, (sid, sec), _ = osq.query_embeddings_base(e, Ic, db_host, song_pos, 0.5, 1)
        py_same += int(sid == dec[j][0] and sec == dec[j][1] * 0.5)
    t_py = (time.perf_counter() - t0) / max(npy, 1)
    torch.set_num_threads(default_threads)
    # ---- the same oracle on MANY cores: one process per 8 threads (tools/oracle_pool.py; inside one process the oracle
    # anti-scales beyond 8-16 threads, so this is what the host can really do with the reference's algorithm): a larger
    # sample of the same queries, every decision compared with the GPU's
    pool_leg = None
    n_pool = min(args.cpu_pool_queries, q_pcm_mine.shape[0])
    if n_pool > 0:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import oracle_pool
        # 8 processes: measured on this pool's 256-thread hosts with 192 queries (profiles/r4/NOTES.md): 8 / 16 / 24 / 32
        # processes x 8 threads = 300 / 279 / 239 / 214 segments/s -- the host saturates at about 64 busy threads
        procs = max(1, min(int(os.environ.get("PFANN_CPU_POOL_PROCS", "8")), ncpu // 8, n_pool))
        try:
            pool = oracle_pool.run(params, sd, db_host, song_pos, q_pcm_mine[:n_pool].cpu().numpy(), k, workers=procs)
            same = int(np.sum((pool["song"] == res["song"][:n_pool]) & (pool["sec"] == res["offset"][:n_pool] * 0.5)))
            pool_leg = {"value": round(n_pool * QUERY_SEGS / pool["compute_s"], 1), "unit": "segments/s",
                        "processes": pool["workers"], "threads_per_process": pool["threads_per_worker"],
                        "cores": pool["workers"] * pool["threads_per_worker"], "queries": n_pool,
                        "compute_s": round(pool["compute_s"], 2), "wall_s_with_process_startup_and_db_load": round(pool["wall_s"], 2),
                        "stages_cpu_s_summed_over_processes": {kk: round(v, 2) for kk, v in pool["stages_s"].items()},
                        "rerank": "the reference's default Python-path matcher (database.py:143-163 restated)",
                        "identical_song_and_offset_vs_gpu": "%d/%d" % (same, n_pool)}
        except Exception as x:                                  # the single-process figures below still stand
            pool_leg = {"error": repr(x)[:300]}
    single = {"value": round(nseg / tcpu, 2), "cores": best_nt, "queries": nq_cpu}
    if pool_leg and pool_leg.get("value", 0) > single["value"]:
        headline, cores, how = pool_leg["value"], pool_leg["cores"], "%d processes x %d threads" % (pool_leg["processes"], pool_leg["threads_per_process"])
    else:
        headline, cores, how = single["value"], best_nt, "one process, %d threads" % best_nt
    cpu = {"value": headline, "unit": "segments/s", "cores": cores, "kind": "port", "how": how,
           "single_process": single, "multi_process": pool_leg,
           "sample": "the step's own 10 s queries vs the same %d-row db, the oracle's whole path (torch-CPU mel+encoder, BLAS sgemm + "
                     "argpartition top-%d, sequence matcher): `single_process` = %d queries (%d segments) in one process at the "
                     "best thread count of the sweep, C seq_score (OpenMP) as the matcher; `multi_process` = %d queries on one "
                     "oracle process per 8 threads, the reference's Python-path matcher, clock from the first worker's first "
                     "query to the last worker's last (process start-up and each worker's load of the db excluded); `value` = the "
                     "better of the two; host has %d logical cores; torch default %d threads, OMP_NUM_THREADS=%s"
                     % (n_rows, k, nq_cpu, nseg, n_pool, ncpu, default_threads, os.environ.get("OMP_NUM_THREADS")),
           "thread_sweep_segments_per_s": {str(t): v for t, v in sweep.items()},
           "stages_s": {kk: round(v, 3) for kk, v in st.items()},
           "stage_segments_per_s": {kk: round(nseg / v, 1) for kk, v in st.items() if v > 0},
           "encoder_segments_per_s": enc,
           "python_rerank": {"queries": npy, "seconds_per_query": round(t_py, 3), "segments_per_s": round(QUERY_SEGS / t_py, 1),
                             "same_decision_as_c_path": "%d/%d" % (py_same, npy),
                             "what": "database.py:143-163 restated (oracle/seqscore.py): Python loop over <= 1900 candidates x 19 rows"}}
    par = {"queries": nq_cpu, "identical_song_and_offset": int(agree)}
    if pool_leg and "identical_song_and_offset_vs_gpu" in pool_leg:
        par["python_path_pool"] = pool_leg["identical_song_and_offset_vs_gpu"]
    return cpu, par


def self_launch(n, backend):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command (one per GPU, RCCL unless
    PFANN_DIST_BACKEND says otherwise) and return their exit status.  Fails before starting anything when the box does
    not have N devices for an RCCL job."""
    import socket
    import subprocess
    if backend == "nccl" and "PFANN_FORCE_DEVICE" not in os.environ:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            log("bench.py --gpus %d: only %d HIP device(s) visible; an RCCL job needs one device per rank -- refusing to "
                "run fewer ranks than asked for" % (n, have))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)))
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=None,
                    help="10 s queries per step: for the WHOLE job (--scaling strong, the default: 4096 = 512 per GPU at 8 "
                         "GPUs, the same job at every N) or per GPU (--scaling weak: 512)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): the SAME job at every N -- one 1M-segment db, --queries queries per step split "
                         "over the ranks; weak: every rank brings its own --queries queries per step")
    ap.add_argument("--db-songs", type=int, default=16950, help="16950 x 59 = 1,000,050 segments")
    ap.add_argument("--filler-db", action="store_true",
                    help="round-1 style database (48 real songs + seeded unit-norm filler rows): scan-only studies")
    ap.add_argument("--snr", type=float, default=0.0)
    ap.add_argument("--max-batch", type=int, default=9728,
                    help="encoder chunk (segments); 9728 = the whole step in one chunk: 29 GB of activations, and the\n"
                         "small late layers get enough 128x128 tiles to fill the 512 resident workgroups")
    ap.add_argument("--cpu-queries", type=int, default=64, help="bounded sample for the CPU baseline (one process)")
    ap.add_argument("--cpu-pool-queries", type=int, default=256,
                    help="bounded sample for the CPU baseline on many cores (one oracle process per 8 threads); 0: skip")
    ap.add_argument("--no-cli", action="store_true", help="skip the drop-in CLI leg (builder.py / matcher.py from WAV files)")
    ap.add_argument("--cli-songs", type=int, default=10000, help="CLI leg: songs written as WAVs (BASELINE config 2: 10 k)")
    ap.add_argument("--cli-queries", type=int, default=2000, help="CLI leg: 10 s queries written as WAVs")
    ap.add_argument("--encoder-precision", type=int, default=0, choices=[0, 1],
                    help="0: exact fp32 MFMA (default, the headline); 1: opt-in 3-term fp16 split (pfann_set_encoder_precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational fp16-storage run")
    ap.add_argument("--serial", action="store_true", help="time the K batches strictly one at a time instead of two deep")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic: keep the committed PMC figure instead of measuring it in this run (two short "
                         "rocprofv3 --pmc passes of a child process, rank 0 at N = 1, after the timed loop)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, meet in the process group, print who is there (n_gpus, ranks_seen, backend, "
                         "devices) and exit: the launcher's own test, runs without a GPU under PFANN_DIST_BACKEND=gloo")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded protocol (two-phase search, all-to-all, merge, all-gathers, device pick) even at "
                         "world 1: every collective then really goes through the backend (RCCL on one GPU)")
    ap.add_argument("--dump-decisions", default=None, help="write (song, offset, score) per query as .npy (rank 0)")
    args = ap.parse_args()
    if args.queries is None:
        args.queries = 4096 if args.scaling == "strong" else 512

    # ------------------------------------------------------------------ ranks: launch, verify, never fall through
    # `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts its own N ranks, one per GPU, through
    # torch.distributed.run on 127.0.0.1; launched externally (the driver's form), WORLD_SIZE must equal --gpus.  No code
    # path below prints an `n_gpus` different from --gpus.
    backend = os.environ.get("PFANN_DIST_BACKEND", "nccl")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to run (start it as `python bench.py --gpus %d` "
            "or with torch.distributed.run --nproc-per-node %d)" % (world, args.gpus, args.gpus, args.gpus))
        sys.exit(2)

    import torch
    import torch.distributed as dist
    from pfann_amd import lib as plib
    from pfann_amd import synth
    from pfann_amd.database import DeviceIndex
    from pfann_amd.dist import ShardedIndex, all_gather_ragged, shard_songs, split_even
    from pfann_amd.engine import Engine
    from pfann_amd.utils import read_config

    if not args.launch_check:
        plib.require_gpu()
    # PFANN_FORCE_DEVICE / PFANN_DIST_BACKEND=gloo: debugging aid to run the N-rank path on a box
    # with a single GPU (all ranks share it); the driver's multi-GPU runs use neither.
    if "PFANN_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["PFANN_FORCE_DEVICE"])
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        if local_rank >= torch.cuda.device_count():
            log("bench.py: rank %d wants HIP device %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
            sys.exit(2)
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if have_gpu else torch.device("cpu")
    in_group = "WORLD_SIZE" in os.environ            # also at world 1 under a launcher: the RCCL calls then really run
    ranks_info = {"ranks_seen": 1, "backend": None, "devices": [local_rank if have_gpu else -1]}
    if in_group:
        from pfann_amd.dist import quiet_stdout
        with quiet_stdout():                 # gloo announces its connections on stdout; stdout is the ONE JSON line
            # the waits around rank 0's side legs (the CLI leg: two tool runs of up to CLI_LEG_TIMEOUT_S each) must outlast
            # them: a barrier that gives up after the default 30 minutes would take the headline line with it
            import datetime
            long_wait = datetime.timedelta(seconds=2 * CLI_LEG_TIMEOUT_S + 1800)
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(backend, timeout=long_wait)
            # host-side waits that may last a minute (the CLI leg runs on rank 0 only) go through a gloo group: an RCCL
            # barrier would spin a kernel on the other ranks' GPUs for as long
            meta_group = dist.new_group(backend="gloo", timeout=long_wait) if backend != "gloo" else None
            dist.barrier(group=meta_group)   # (gloo connects lazily: make it talk now)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, torch.cuda.current_device() if have_gpu else -1))
        ranks_info = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                      "devices": [dv for _, dv in sorted(seen)]}
        if sorted(r for r, _ in seen) != list(range(world)) or dist.get_world_size() != args.gpus:
            log("bench.py: rank set %r does not cover --gpus %d" % (seen, args.gpus))
            sys.exit(2)
        if backend == "nccl" and len(set(ranks_info["devices"])) != world:
            log("bench.py: %d ranks share HIP devices %r: RCCL needs one device per rank" % (world, ranks_info["devices"]))
            sys.exit(2)
    if args.launch_check:
        # what the CPU test of the launcher reads: the ranks really exist and have met in one process group
        if in_group:
            t = torch.ones(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t)
            ranks_info["all_reduce_of_ones"] = int(t.item())
        if rank == 0:
            print(json.dumps(dict({"launch_check": True, "n_gpus": world}, **ranks_info)), flush=True)
        if in_group:
            dist.destroy_process_group()
        return

    # PFANN_EMULATE_WORLD=N (tuning aid, single process): do rank 0's share of an N-rank job -- 1/N of the db,
    # 1/N of the queries embedded (then tiled to the full batch), the sharded query path with its merge and
    # owned-only rerank -- without the collectives.  Shows how the per-rank work shrinks with N; never a result.
    emu = int(os.environ.get("PFANN_EMULATE_WORLD", "0"))
    if emu > 1:
        assert not in_group, "PFANN_EMULATE_WORLD is a single-process aid"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29611")
        dist.init_process_group("gloo", rank=0, world_size=1)
    params = read_config(os.path.join(REPO, "configs", "default.json"))
    d, k = params["model"]["d"], params["indexer"]["top_k"]
    t_setup = time.time()
    eng = Engine(params, local_rank, max_batch=args.max_batch)
    # seeded weights, output bias calibrated so that an untrained network's fingerprints spread over the sphere
    # (pfann_amd/synth.py: make_state_dict_calibrated); identical constants on every box
    sd = synth.make_state_dict_calibrated(params, seed=123)
    eng.load_state_dict(sd)
    if args.encoder_precision:
        assert eng.set_encoder_precision(args.encoder_precision) == args.encoder_precision

    # ---------------------------------------------------------------- database (untimed)
    from pfann_amd.builder import embed_files
    n_songs = args.db_songs
    song_pos = np.arange(n_songs + 1, dtype=np.int64) * SEG_PER_SONG
    n_rows = int(song_pos[-1])
    s_lo, s_hi = shard_songs(song_pos, emu if emu > 1 else world)[rank]
    r_lo, r_hi = int(song_pos[s_lo]), int(song_pos[s_hi])
    shard = torch.empty((r_hi - r_lo, d), device=dev, dtype=torch.float32)

    class PcmList:                       # what builder.embed_files needs of a MusicDataset: files + load_pcm(i)
        def __init__(self, ids, pcm_host):
            self.files = ["synthetic song %d" % i for i in ids]
            self.pcm = pcm_host

        def load_pcm(self, i):
            return self.pcm[i]           # int16 [n] in pinned HOST memory: the builder uploads it

        def __len__(self):
            return len(self.files)

    builder_s, builder_segs = 0.0, 0
    if args.filler_db:
        gen = torch.Generator(device=dev)
        blk = 1 << 18
        for gb in range(r_lo // blk, (r_hi + blk - 1) // blk):
            b0, b1 = gb * blk, min((gb + 1) * blk, n_rows)
            gen.manual_seed(1234567 + gb)
            x = torch.randn((b1 - b0, d), device=dev, generator=gen)
            x = x / x.norm(dim=1, keepdim=True)
            lo, hi = max(b0, r_lo), min(b1, r_hi)
            shard[lo - r_lo:hi - r_lo] = x[lo - b0:hi - b0]
    # songs per generated chunk: whole launch groups (164 songs = 9676 windows fit --max-batch 9728), four per chunk, so
    # that the builder loop is timed on full groups and fills / drains its pipeline once per 656 songs (315 MB of PCM)
    CH = 4 * max(1, args.max_batch // SEG_PER_SONG)
    real_ids = range(s_lo, s_hi) if not args.filler_db else \
        [int(v) for v in np.unique(np.linspace(0, n_songs - 1, 48).astype(np.int64)) if s_lo <= v < s_hi]
    host_buf = torch.empty((CH, SEG_PER_SONG * 4000 + 4000), dtype=torch.int16).pin_memory()
    for c0 in range(0, len(real_ids), CH):
        ids = list(real_ids[c0:c0 + CH])
        host_buf[:len(ids)].copy_(synth.make_songs_torch(ids, 30.0, device=dev))      # synthesised on the device
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i, n_seg, e in embed_files(eng, PcmList(ids, host_buf), 4000, batch_windows=args.max_batch):
            assert n_seg == SEG_PER_SONG
            shard[int(song_pos[ids[i]]) - r_lo: int(song_pos[ids[i] + 1]) - r_lo] = e
        torch.cuda.synchronize()
        builder_s += time.perf_counter() - tb
        builder_segs += len(ids) * SEG_PER_SONG
    index = DeviceIndex(d, local_rank)
    index.load(shard, song_pos, r_lo, song_range=(s_lo, s_hi))
    use_sharded = world > 1 or emu > 1 or args.force_sharded
    if args.force_sharded and not in_group:
        log("bench.py: --force-sharded needs a process group (start with torch.distributed.run --nproc-per-node 1)")
        sys.exit(2)
    if use_sharded:
        sharded = ShardedIndex(index, song_pos, k, 1, 0.0, always_exchange=args.force_sharded)
        if sharded.xs is not None:               # (exchange stream on: the next batch's log-mel front end starts behind the
            eng.before_front_end = sharded.hold_front_end     # shard scan in flight -- pfann_amd/dist.py: hold_front_end)

    # ----------------------------------------------------------------- queries (untimed)
    Q = args.queries * ((emu if emu > 1 else world) if args.scaling == "weak" else 1)      # queries per step, whole job
    if args.filler_db:
        all_real = np.unique(np.linspace(0, n_songs - 1, 48).astype(np.int64))
        q_song = [int(all_real[j % len(all_real)]) for j in range(Q)]
    else:
        q_song = [int((j * 7919 + 13) % n_songs) for j in range(Q)]        # spread over the whole db
    # every rank synthesises only its own slice of the step's queries; the crop offsets (for the hit-rate) are gathered
    my_q = split_even(Q, emu if emu > 1 else world)[rank]
    q_counts = [(hi - lo) * QUERY_SEGS for lo, hi in split_even(Q, world)]
    q_pcm_t, q_off_t = [], []
    for c0 in range(my_q[0], my_q[1], CH):
        c1 = min(c0 + CH, my_q[1])
        ids = q_song[c0:c1]
        qp, qo = synth.make_queries_torch(synth.make_songs_torch(ids, 30.0, device=dev), list(range(c0, c1)), 10.0, args.snr)
        q_pcm_t.append(qp)
        q_off_t.append(qo)
    q_pcm_mine = torch.cat(q_pcm_t)                                         # [my queries, 80000] int16 on the device
    q_off_mine = torch.cat(q_off_t)
    if in_group:
        q_off = all_gather_ragged(q_off_mine.reshape(-1, 1), [hi - lo for lo, hi in split_even(Q, world)]).reshape(-1).cpu().numpy()
    else:
        q_off = np.full(Q, 1e9)                                             # emulation: only rank 0's slice is known
        q_off[my_q[0]:my_q[1]] = q_off_mine.cpu().numpy()
    q_len = q_pcm_mine.shape[1]
    pcm_dev = q_pcm_mine.reshape(-1).contiguous()                           # resident in HBM
    pcm_host = pcm_dev.cpu().pin_memory()                                   # the same bytes as a host hand-over
    starts = (np.arange(my_q[1] - my_q[0], dtype=np.int64)[:, None] * q_len +
              np.arange(QUERY_SEGS, dtype=np.int64)[None, :] * 4000).reshape(-1)
    starts_dev = torch.as_tensor(starts).to(dev)
    qstart = np.arange(Q, dtype=np.int64) * QUERY_SEGS
    qlen = np.full(Q, QUERY_SEGS, dtype=np.int32)
    torch.cuda.synchronize()
    log("[rank %d] setup %.1fs: shard rows %d (songs %d..%d, %d embedded by the builder loop in %.2f s), %d queries/step" %
        (rank, time.time() - t_setup, r_hi - r_lo, s_lo, s_hi, builder_segs // SEG_PER_SONG, builder_s, Q))

    cur_index = [index]

    def make_workload(counts):
        """counts[r] = queries rank r brings per step (the first counts[r] of the ones it synthesised)"""
        n_me, tot = counts[rank], sum(counts)
        return {"pcm_dev": pcm_dev[: n_me * q_len], "pcm_host": pcm_host[: n_me * q_len],
                "starts": starts_dev[: n_me * QUERY_SEGS].contiguous(), "q_counts": [c * QUERY_SEGS for c in counts],
                "qstart": np.arange(tot, dtype=np.int64) * QUERY_SEGS, "qlen": np.full(tot, QUERY_SEGS, dtype=np.int32), "Q": tot}
    main_w = make_workload([hi - lo for lo, hi in split_even(Q, world)])
    if emu > 1:
        main_w.update(qstart=qstart, qlen=qlen, Q=Q)

    def step(from_host=True, w=main_w):
        wav = eng.pcm16_to_mono(w["pcm_host"].to(dev, non_blocking=True) if from_host else w["pcm_dev"])
        emb = eng.embed_windows(wav, w["starts"])
        if emu > 1:
            emb = emb.repeat(emu, 1)[: Q * QUERY_SEGS].contiguous()
            return sharded.query_batch(emb, w["qstart"], w["qlen"]), emb
        if use_sharded:
            with sharded.exchange(emb):             # (PFANN_EXCHANGE_STREAM=1: on the exchange stream; else a no-op)
                emb = sharded._timed("emb_allgather", all_gather_ragged, emb, w["q_counts"])
                return sharded.query_batch(emb, w["qstart"], w["qlen"]), emb
        D, I = cur_index[0].search(emb, k)
        res, _ = cur_index[0].match(emb, I, w["qstart"], w["qlen"])
        return res, emb

    def fence():
        if in_group:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed loop, two batches deep, as a serving loop (and the matcher CLI, pfann_amd/matcher.py) runs the
    # path: the H2D of batch i+1 goes on a side stream under the kernels of batch i, and the decisions of batch i are read
    # on the host after batch i+1 has been launched.  Every batch still does everything (H2D of its PCM, all kernels, D2H
    # of its decisions) inside the timed region; only the host's waits are overlapped.  --serial: one batch at a time.
    class TwoDeep:
        def __init__(self, w):
            self.w = w
            self.cs = torch.cuda.Stream(device=dev)
            self.bufs = [torch.empty_like(w["pcm_dev"]) for _ in range(2)]
            self.up = [torch.cuda.Event() for _ in range(2)]         # upload of the buffer complete
            self.used = [torch.cuda.Event() for _ in range(2)]       # buffer consumed (mono conversion issued behind it)

        def upload(self, i):
            with torch.cuda.stream(self.cs):
                self.bufs[i & 1].copy_(self.w["pcm_host"], non_blocking=True)
                self.up[i & 1].record(self.cs)

        def launch(self, i):
            w = self.w
            torch.cuda.current_stream().wait_event(self.up[i & 1])
            wav = eng.pcm16_to_mono(self.bufs[i & 1])
            self.used[i & 1].record()
            e = eng.embed_windows(wav, w["starts"])
            if use_sharded:
                # the exchange of batch i (fingerprint all-gather, shard search, list exchange, merge, owner-side matcher,
                # winner pick): with PFANN_EXCHANGE_STREAM=1 on a stream of its own, so that batch i+1's encoder -- issued
                # on this stream by the next launch() -- does not queue behind the collectives
                with sharded.exchange(e):
                    e = sharded._timed("emb_allgather", all_gather_ragged, e, w["q_counts"])
                    return sharded.query_batch(e, w["qstart"], w["qlen"], to_host=False), e
            D, I = cur_index[0].search(e, k)
            r, _ = cur_index[0].match(e, I, w["qstart"], w["qlen"], to_host=False)
            return r, e

        def read_back(self, pend):
            if use_sharded:
                with sharded.on_exchange_stream():  # (the results were produced there; no wait for the next batch's encoder)
                    return index.results_to_host(pend)
            return index.results_to_host(pend)

        def run(self, n):
            self.upload(0)
            pend = res = e = None
            for i in range(n):
                r, e = self.launch(i)
                if i + 1 < n:
                    if i >= 1:
                        self.cs.wait_event(self.used[(i + 1) & 1])   # batch i-1 has consumed the buffer batch i+1 lands in
                    self.upload(i + 1)
                if pend is not None:
                    res = self.read_back(pend)
                pend = r
            if pend is not None:
                res = self.read_back(pend)
            return res, e

    def max_over_ranks(sec):
        if not in_group:
            return sec
        t = torch.tensor([sec], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # what has been imported and built so far goes to the garbage collector's permanent generation: a full collection over
    # torch's million objects takes ~50 ms and would land in whichever timed step crosses the allocation threshold
    import gc
    gc.freeze()
    for _ in range(args.warmup):
        step()
    two_deep = None if (args.serial or emu > 1) else TwoDeep(main_w)
    if two_deep is not None:
        two_deep.run(2)                  # untimed: the side stream, its buffers and events exist before the clock starts
    lib = plib.load()
    prof = not args.no_prof
    fence()
    if prof:
        lib.pfann_prof_reset()
        lib.pfann_prof_enable(1)
    fence()
    if use_sharded and prof:
        sharded.timing = {}
    lib.pfann_prof_marker(None)
    t0 = time.perf_counter()
    if two_deep is not None:
        res, emb = two_deep.run(args.steps)
    else:
        for _ in range(args.steps):
            res, emb = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.pfann_prof_marker(None)
    if prof:
        lib.pfann_prof_enable(0)
    elapsed = max_over_ranks(elapsed)
    n_seg = Q * QUERY_SEGS
    value = n_seg * args.steps / elapsed
    collectives = None
    my_collectives = {}
    if use_sharded and prof:
        # event time of every collective of the exchange protocol (pfann_amd/dist.py: ShardedIndex._timed), per step,
        # the slowest rank's: bound all-gather, all-to-all of the shard lists, merge kernel, merged-slice all-gather,
        # winner-key all-gather, and the ragged all-gather of the step's fingerprints
        mine = sharded.timing_ms()
        sharded.timing = None
        my_collectives = {nm: ms / args.steps for nm, ms in mine.items()}
        collectives = {nm: round(max_over_ranks(ms / args.steps), 4) for nm, ms in sorted(mine.items())}
        collectives["unit"] = "ms per step, max over ranks"

    # ---- the same K batches one at a time (H2D, kernels, D2H, host wait; then the next): what `value` was through round 3's
    # first half, kept for comparison
    serial = None
    if two_deep is not None:
        fence()
        tp = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = max_over_ranks(time.perf_counter() - tp)
        serial = {"value": round(n_seg * args.steps / el, 1), "unit": "segments/s", "ms_per_step": round(1e3 * el / args.steps, 3),
                  "what": "the same batches strictly one at a time: the host waits for batch i's decisions before it starts "
                          "the H2D of batch i+1"}

    # ---- the same step with the query PCM already resident in HBM (reported beside `value`)
    pcie = None
    if emu <= 1:
        step(False)
        fence()
        tp = time.perf_counter()
        for _ in range(args.steps):
            step(False)
        fence()
        el = max_over_ranks(time.perf_counter() - tp)
        pcie = {"value": round(n_seg * args.steps / el, 1), "unit": "segments/s", "ms_per_step": round(1e3 * el / args.steps, 3),
                "what": "same step with the int16 query PCM already resident in HBM (`value` includes the H2D of %.1f MB per "
                        "step and rank from pinned host memory)" % (pcm_host.numel() * 2 / 1e6)}

    # ---- N > 1, default (weak) mode: the same job in the OTHER scaling mode as a side figure, so SCALE runs stay
    # comparable with round 1's strong-scaling numbers: --queries queries for the WHOLE job, split over the ranks
    other_mode = None
    if world > 1 and args.scaling == "strong" and emu <= 1 and Q // 8 >= 1:
        # the weak-scaling side figure: every rank brings Q/8 queries of its own (512 at the default 4096): equals the
        # headline job at N = 8 and is 1/8 ... 1/2 of it below
        per = min(Q // 8, Q // world)            # (every rank synthesised at least Q // world queries)
        sw = make_workload([per] * world)
        step(True, sw)
        fence()
        tp = time.perf_counter()
        for _ in range(args.steps):
            step(True, sw)
        fence()
        el = max_over_ranks(time.perf_counter() - tp)
        other_mode = {"scaling": "weak", "value": round(sw["Q"] * QUERY_SEGS * args.steps / el, 1), "unit": "segments/s",
                      "ms_per_step": round(1e3 * el / args.steps, 3), "queries_per_step": sw["Q"],
                      "what": "weak scaling: %d queries per step and rank (%d per step for the job) against the same 1 M-segment "
                              "db" % (per, sw["Q"])}
    elif world > 1 and args.scaling == "weak" and emu <= 1:
        sw = make_workload([hi - lo for lo, hi in split_even(args.queries, world)])
        step(True, sw)
        fence()
        tp = time.perf_counter()
        for _ in range(args.steps):
            step(True, sw)
        fence()
        el = max_over_ranks(time.perf_counter() - tp)
        other_mode = {"scaling": "strong", "value": round(sw["Q"] * QUERY_SEGS * args.steps / el, 1), "unit": "segments/s",
                      "ms_per_step": round(1e3 * el / args.steps, 3), "queries_per_step": sw["Q"],
                      "what": "the same 1 M-segment job with a FIXED %d queries per step split over the %d ranks (each rank "
                              "encodes %d segments per step)" % (sw["Q"], world, sw["q_counts"][0])}

    # ---- the reference's own native seam (cpp/seqscore.cpp:32-43 via database.py:178-189): host pointers in, best
    # song out; 200 calls, median and p95, once here and once more after the fp16-storage leg below
    seam_before = seam_after = None

    def seam_leg():
        import ctypes
        from oracle import native
        q1 = np.ascontiguousarray(emb[:QUERY_SEGS].cpu().numpy())
        _, I1 = index.search(emb[:QUERY_SEGS].contiguous(), k)
        lab1 = np.ascontiguousarray(I1.cpu().numpy())
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        ss = np.zeros((n_songs, 2), np.float32)
        args_c = (index.handle, song_pos.ctypes.data_as(i64p), n_songs, q1.ctypes.data_as(f32p), QUERY_SEGS,
                  lab1.ctypes.data_as(i64p), k, ss.ctypes.data_as(f32p), 1, 0.0)
        torch.cuda.synchronize()
        for _ in range(20):
            ss[:] = 0
            b_gpu = lib.seq_score(*args_c)
        ts = []
        for _ in range(200):
            ss[:] = 0                                                 # database.py:176: the caller zeroes the block
            t1 = time.perf_counter()
            lib.seq_score(*args_c)
            ts.append(time.perf_counter() - t1)
        ss_gpu = ss.copy()
        ts = np.sort(np.asarray(ts)) * 1e6
        db_host = shard.cpu().numpy()
        b_cpu, ss_cpu = native.seq_score(db_host, song_pos, q1, lab1, 1, 0.0)
        t1 = time.perf_counter()
        for _ in range(5):
            native.seq_score(db_host, song_pos, q1, lab1, 1, 0.0)
        seam_cpu_us = 1e6 * (time.perf_counter() - t1) / 5
        return {"gpu_call_us_median": round(float(ts[len(ts) // 2]), 1), "gpu_call_us_p95": round(float(ts[int(len(ts) * 0.95)]), 1),
                "gpu_call_us_min": round(float(ts[0]), 1), "calls": len(ts),
                "cpu_oracle_call_us": round(seam_cpu_us, 1),
                "same_best_song": bool(b_gpu == b_cpu), "max_abs_score_diff": float(np.abs(ss_gpu - ss_cpu)[:, 0].max()),
                "what": "seq_score(index, song_pos, n_songs=%d, query[19x128], labels[19x100], song_scores, 1, 0) with host "
                        "pointers, as database.py:178-189 calls it (the caller's zeroing of song_scores outside the clock); pinned "
                        "staging + private stream in the handle; cpu_oracle = the C restatement of cpp/seqscore.cpp, OpenMP" % n_songs}
    if rank == 0 and world == 1 and not use_sharded and emu <= 1 and not args.no_cpu_baseline:
        seam_before = seam_leg()

    # ---- informational: fp16-only storage of the same db (BASELINE config 5's "fp16 embeddings"; never `value`)
    alt = None
    if world == 1 and emu <= 1 and not use_sharded and not args.no_alt:
        idx16 = DeviceIndex(d, local_rank, storage="f16")
        idx16.load(shard, song_pos, r_lo, song_range=(s_lo, s_hi))
        cur_index[0] = idx16
        step()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            res16, _ = step()
        torch.cuda.synchronize()
        el = (time.perf_counter() - ta) / args.steps
        cur_index[0] = index
        same = int(np.sum((res16["song"] == res["song"]) & (res16["offset"] == res["offset"])))
        hit16 = int(np.sum(res16["song"] == np.asarray(q_song)))
        alt = {"fp16_db": {
            "what": "pfann_db_set_storage(PFANN_DB_F16): only fp16 rows kept (%d MB instead of %d MB), fp16-MFMA scan without "
                    "fp32 re-scoring, sequence scores over the stored rows; approximate like faiss useFloat16 "
                    "(database.py:101-104); NOT the headline value" % ((r_hi - r_lo) * d * 2 >> 20, (r_hi - r_lo) * d * 4 >> 20),
            "value": round(n_seg / el, 1), "unit": "segments/s", "ms_per_step": round(1e3 * el, 3),
            "decisions_identical_to_fp32_db": "%d/%d" % (same, Q), "top1_hit_rate": round(hit16 / Q, 4)}}
        del idx16
        # ---- encoder-only rates of the other model families and of the unfused fallback (the rates DESIGN.md quotes):
        # this step's windows (PCM resident in HBM) -> fingerprints, 3 passes each; never `value`
        enc_alt = {}
        wav_alt = eng.pcm16_to_mono(pcm_dev)
        for name, cfg_file, fused in (("seg.json (depthwise conv2, 116 MMAC/segment)", "seg.json", True),
                                      ("n640d64.json (d = 64, depthwise, 36 MMAC/segment)", "n640d64.json", True),
                                      ("default.json, LayerNorm NOT fused (pfann_set_fused_layernorm(0): conv_gemm_kernel + ln_act_kernel)",
                                       "default.json", False)):
            pa = read_config(os.path.join(REPO, "configs", cfg_file))
            ea = Engine(pa, local_rank, max_batch=4864)
            ea.load_state_dict(synth.make_state_dict(pa, seed=123))
            if not fused:
                assert ea.set_fused_layernorm(False) is False
            ea.embed_windows(wav_alt, starts_dev)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(3):
                ea.embed_windows(wav_alt, starts_dev)
            torch.cuda.synchronize()
            enc_alt[name] = {"segments_per_s": round(3 * starts_dev.shape[0] / (time.perf_counter() - ta), 1),
                             "windows_per_pass": int(starts_dev.shape[0]), "max_batch": 4864}
            del ea
        alt["encoder_only"] = enc_alt
    if seam_before is not None:
        seam_after = seam_leg()

    # ------------------------------------------------------------ per-kernel event times
    kernels = {}
    if prof:
        import ctypes
        buf = ctypes.create_string_buffer(4096)
        lib.pfann_prof_tags(buf, 4096)
        for tag in buf.value.decode().split(","):
            if not tag:
                continue
            cnt = ctypes.c_int64(0)
            ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(cnt))
            work = lib.pfann_prof_work(tag.encode())
            kernels[tag] = {"ms_per_step": ms / args.steps, "launches_per_step": cnt.value / args.steps,
                            "avg_us": 1e3 * ms / max(cnt.value, 1), "work_per_launch": work / max(cnt.value, 1)}
    # ---- scan throughput (north_star: ">= 6x scan throughput at 8 GPUs"): db rows x query rows per second of SCAN time, the
    # scan being every kernel of the exact top-k search (sampled pass, full pass, selects, fallback, and the sharded
    # path's bound / merge kernels); per rank from the kernel tags of the timed loop, the slowest rank's time counts
    scan_throughput = None
    if prof:
        SCAN_TAGS = [t for t in kernels if t.startswith("scan_topk") or t.startswith("topk_")]
        scan_ms = sum(kernels[t]["ms_per_step"] for t in SCAN_TAGS)
        if in_group:
            allms = [None] * world
            dist.all_gather_object(allms, scan_ms)
        else:
            allms = [scan_ms]
        slowest = max(allms)
        if slowest > 0:
            scan_throughput = {"value": float("%.4g" % (float(n_rows) * Q * QUERY_SEGS / (slowest * 1e-3))),
                               "unit": "db rows x query rows / s of scan-kernel time (slowest rank)",
                               "db_rows": n_rows, "query_rows_per_step": Q * QUERY_SEGS,
                               "scan_ms_per_step_per_rank": [round(v, 4) for v in allms], "tags": sorted(SCAN_TAGS),
                               "TFLOPs_equivalent": float("%.4g" % (2.0 * d * n_rows * Q * QUERY_SEGS / (slowest * 1e-3) / 1e12))}
    # ---- where a rank's step goes (N > 1: what to read off the first curve measured on real xGMI):
    # event time of this rank's kernels by stage + event time of its collectives, against the step's wall time.  One
    # stream (the default): the stages run back to back, so `unattributed` is launch gaps and waiting -- for the host, or
    # inside a collective for a slower rank (a rank that arrives early at a collective sees its waiting as collective
    # time: compare min and max over ranks).  PFANN_EXCHANGE_STREAM=1: the exchange overlaps the next batch's encoder
    # and `unattributed` goes NEGATIVE by the overlap won.
    critical_path = None
    if prof:
        def stage_of(tag):
            if tag.startswith("scan_topk") or tag.startswith("topk_"):
                return "scan"
            if tag.startswith("seq_match") or tag.startswith("match_") or tag.startswith("song_scores"):
                return "matcher"
            return "encoder"
        mine_cp = {"encoder": 0.0, "scan": 0.0, "matcher": 0.0}
        for tag, kv in kernels.items():
            mine_cp[stage_of(tag)] += kv["ms_per_step"]
        mine_cp["collectives"] = sum(ms for nm, ms in my_collectives.items() if nm != "merge")    # (merge is a kernel: in `scan`)
        step_ms = 1e3 * elapsed / args.steps
        mine_cp["unattributed"] = step_ms - sum(mine_cp.values())
        if in_group:
            allcp = [None] * world
            dist.all_gather_object(allcp, mine_cp)
        else:
            allcp = [mine_cp]
        keys = ("encoder", "scan", "collectives", "matcher", "unattributed")
        critical_path = {"unit": "ms per step", "step_ms": round(step_ms, 3),
                         "max_over_ranks": {kk: round(max(c[kk] for c in allcp), 3) for kk in keys},
                         "min_over_ranks": {kk: round(min(c[kk] for c in allcp), 3) for kk in keys},
                         "per_rank": [{kk: round(c[kk], 3) for kk in keys} for c in allcp],
                         "exchange_stream": bool(use_sharded and sharded.xs is not None),
                         "what": "HIP-event time of each rank's kernels by stage (encoder = PCM conversion, log-mel, convs, head; "
                                 "scan = every kernel of the exact top-k search incl. the sharded path's bound / merge kernels; "
                                 "matcher = sequence match, key pack / pick) and of its collectives (fingerprint all-gather, bound "
                                 "all-gather, list all-to-all, merged-slice all-gather, key all-gather), per step of the timed loop; "
                                 "unattributed = step wall time minus their sum (negative when the exchange stream overlaps them)"}
    # roofline of the dominant kernel (most time in the timed region)
    ROOF = {"conv_gemm_ln_128": ("pfann::conv_gemm_ln_w22_kernel / conv_gemm_ln_kernel<128,128,64,32,...> (the 15 implicit-GEMM "
                                 "convs, 128x128 tiles, LayerNorm+ReLU of the input fused into the A-loader, LN statistics of the "
                                 "output in the epilogue; the stride-2 layers -- 95 % of the flops -- on the five-blocks-per-output-"
                                 "pair kernel, incl. the one with the first conv folded in)", "mfma"),
            "conv_gemm_ln_64": ("pfann::conv_gemm_ln_kernel<64,64,32,32>", "mfma"),
            "conv_first_stats": ("pfann::conv_first_stats_kernel", "hbm"),
            "conv_gemm_128": ("pfann::conv_gemm_kernel<128,128,64,64>", "mfma"),
            "conv_gemm_64": ("pfann::conv_gemm_kernel<64,64,32,32>", "mfma"),
            "scan_topk": ("pfann::scan_emit_kernel (full-db pass)", "mfma"),
            "scan_topk_f16": ("pfann::scan_f16_qres_kernel (fp16 pre-filter pass; exact fp32 re-scoring follows)", "mfma16"),
            "ln_act": ("pfann::ln_act_kernel", "hbm"), "conv_first": ("pfann::conv_first_kernel", "hbm")}
    for tag, kv in kernels.items():
        if tag in ROOF and kv["work_per_launch"] > 0:
            rate = kv["work_per_launch"] / (kv["avg_us"] * 1e-6)
            if ROOF[tag][1] == "mfma16":
                kv["TFLOPs"] = rate / 1e12
                kv["frac_of_peak"] = rate / 1e12 / 2500.0
            elif ROOF[tag][1] == "mfma":
                kv["TFLOPs"] = rate / 1e12
                kv["frac_of_peak"] = rate / 1e12 / PEAK_F32_MFMA
            else:
                kv["GBps"] = rate / 1e9
                kv["frac_of_peak"] = rate / 1e9 / PEAK_HBM
    roofline = None
    cands = [t for t in kernels if t in ROOF and ROOF[t][1] != "mfma16" and kernels[t]["work_per_launch"] > 0]
    if cands:
        dom = max(cands, key=lambda t: kernels[t]["ms_per_step"])
        kv = kernels[dom]
        mf = ROOF[dom][1] == "mfma"
        ach = kv["TFLOPs"] if mf else kv["GBps"]
        peak = PEAK_F32_MFMA if mf else PEAK_HBM
        roofline = {"kernel": ROOF[dom][0], "bound": ROOF[dom][1], "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s" if mf else "GB/s", "frac": round(ach / peak, 4), "traffic": None,
                    "avg_launch_us": round(kv["avg_us"], 1), "launches_per_step": kv["launches_per_step"],
                    "algorithmic_work_per_launch": kv["work_per_launch"],
                    "flops_counted": "ALGORITHMIC: 2*M*N*K_live of the convolutions (SURVEY 8d); the five-block kernel executes "
                                     "5/6 of them on its layers, so the MFMA pipe itself is used at about 5/6 of `frac` there",
                    "share_of_step_time": round(kv["ms_per_step"] / (1e3 * elapsed / args.steps), 3)}
        if dom == "conv_gemm_ln_128":
            # the same launches priced two other ways: in MFMA work actually EXECUTED (the five-block kernel runs 5/6 of its
            # layers' multiplies: 10 of the 15 launches, 95.4 % of the flops at the default model's shapes), and the
            # whole step against SURVEY 8(d)'s whole-encoder count (0.58204 GFLOP per segment, dead taps included)
            roofline["frac_in_executed_flops"] = round(ach / peak * (5.0 / 6.0 * 0.954 + 0.046), 4)
            roofline["end_to_end_mfma_frac_survey_8d"] = round(value * ENC_FLOP_PER_SEG / (world * PEAK_F32_MFMA * 1e12), 4)
    # HBM-side traffic per launch of the dominant kernel, from the committed PMC passes of this
    # same command (tools/profile_bench.sh -> tools/make_traffic_json.py -> profiles/traffic.json)
    if roofline is not None:
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            meta = tj.pop("_meta", {})
            import glob
            import hashlib
            hh = hashlib.sha256()
            for f in sorted(glob.glob(os.path.join(REPO, "pfann_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "pfann_amd", "csrc", "*.h"))):
                hh.update(open(f, "rb").read())
            # NOT measured in this run (hardware counters need the rocprofv3 wrapper): a committed observation
            roofline["traffic_from_committed_profile"] = True
            roofline["traffic_profile"] = dict(meta, kernels_unchanged_since_profile=(meta.get("csrc_sha16") == hh.hexdigest()[:16])
                                               if meta.get("csrc_sha16") else None)
            key = ROOF[dom][0].split("<")[0].split(" ")[0]
            tmpl = ROOF[dom][0].split(" ")[0].replace(",...>", "").replace(",", ", ")
            if "tag:" + dom in tj:
                roofline["traffic"] = tj["tag:" + dom]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "%s PMC FETCH_SIZE x2 + WRITE_SIZE, launch-weighted over the kernels of tag %s" % (
                    tj["tag:" + dom]["source"], dom)
                raise KeyError("done")
            names = sorted(tj, key=lambda nm: "all instantiations" not in nm)     # prefer the combined record
            for name in names:
                rec = tj[name]
                if name.replace("void ", "").startswith(tmpl.split(">")[0]) or (key in name and "<" not in tmpl):
                    roofline["traffic"] = rec["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = "%s PMC FETCH_SIZE x2 + WRITE_SIZE (%s)" % (rec["source"], name[:60])
                    break
        except (OSError, ValueError, KeyError):
            pass
    # ... and MEASURED in this run where that is possible (rank 0, N = 1, the default workload's dominant kernel): two short
    # rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of a child that pushes one launch group through the encoder
    # (tools/live_traffic.py).  The GPU is idle here: the timed loop is over.  Any failure keeps the committed figure.
    if roofline is not None and dom == "conv_gemm_ln_128" and world == 1 and not in_group and have_gpu and not args.no_live_traffic \
            and not args.encoder_precision:
        roofline["traffic_live"] = False
        try:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import live_traffic
            lt = live_traffic.measure(windows=args.max_batch, device=local_rank)
            if roofline.get("traffic") is not None:
                roofline["traffic_committed_profile_value"] = roofline["traffic"]
            roofline["traffic"] = lt["hbm_bytes_per_launch"]
            roofline["traffic_live"] = True
            roofline["traffic_from_committed_profile"] = False
            roofline["traffic_source"] = "measured in this run: " + lt["command"] + "; " + lt["correction"]
            roofline["traffic_measurement"] = {kk: lt[kk] for kk in ("fetch_size_kb_per_launch", "write_size_kb_per_launch", "dispatches",
                                                                      "windows_per_group", "seconds")}
            log("bench.py: live HBM traffic of the conv GEMMs: %.3f GB per launch (%d dispatches, %.0f s)"
                % (lt["hbm_bytes_per_launch"] / 1e9, lt["dispatches"], lt["seconds"]))
        except Exception as x:                       # noqa: BLE001 -- a side leg must never take the headline line with it
            roofline["traffic_live_error"] = ("%s: %s" % (type(x).__name__, x))[:300]
            log("bench.py: live traffic leg failed, committed figure kept:", roofline["traffic_live_error"])
    for stag in ("scan_topk", "scan_topk_f16"):
        if stag not in kernels:
            continue
        ks = kernels[stag]
        gbs = (r_hi - r_lo) * d * 4 / (ks["avg_us"] * 1e-6) / 1e9
        ks.update({"db_GBps": gbs, "db_hbm_frac": gbs / PEAK_HBM})

    # ---------------------------------- throughput vs launch-group size (round 6)
    # the whole step -- PCM resident in HBM -> mono -> log-mel -> encoder -> exact top-k -> candidates -> sequence score ->
    # decisions on the host -- for ONE launch group of n queries (19 n windows), with the plan the library picks for that
    # batch on its own (no pfann_set_plan_batch), one group at a time (the host waits for each group's decisions: the
    # latency a service with n concurrent queries sees).  Every figure elsewhere is at 9728-window groups or at one query.
    batch_curve = None
    if prof and world == 1 and emu <= 1 and not use_sharded and not args.no_alt:
        import ctypes
        batch_curve = {"what": "whole step for one launch group of n ten-second queries against the %d-row db, default plan of each "
                               "batch size, PCM resident in HBM, decisions read back on the host after every group" % n_rows,
                       "points": []}

        def stage_of_tag(tag):
            if tag.startswith("scan_topk") or tag.startswith("topk_"):
                return "scan"
            if tag.startswith("seq_match") or tag.startswith("match_") or tag.startswith("song_scores"):
                return "matcher"
            return "encoder"
        max_q = min(Q, args.max_batch // QUERY_SEGS)
        for nq_c in (1, 4, 16, 64, 256, 512):
            if nq_c > max_q:
                continue
            nw = nq_c * QUERY_SEGS
            pcm_c = pcm_dev[: nq_c * q_len].contiguous()
            st_c = starts_dev[:nw].contiguous()
            qs_c, ql_c = np.arange(nq_c, dtype=np.int64) * QUERY_SEGS, np.full(nq_c, QUERY_SEGS, np.int32)

            def group():
                e_c = eng.embed_windows(eng.pcm16_to_mono(pcm_c), st_c)
                D_c, I_c = index.search(e_c, k)
                return index.match(e_c, I_c, qs_c, ql_c)[0]
            for _ in range(3):
                r_c = group()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            group()
            torch.cuda.synchronize()
            reps = int(max(3, min(200, 0.4 / max(time.perf_counter() - t1, 1e-5))))
            t1 = time.perf_counter()
            for _ in range(reps):
                r_c = group()
            torch.cuda.synchronize()
            el_c = (time.perf_counter() - t1) / reps
            lib.pfann_prof_reset()
            lib.pfann_prof_enable(1)
            for _ in range(3):
                group()
            torch.cuda.synchronize()
            lib.pfann_prof_enable(0)
            st_ms = {"encoder": 0.0, "scan": 0.0, "matcher": 0.0}
            bufc = ctypes.create_string_buffer(4096)
            lib.pfann_prof_tags(bufc, 4096)
            for tag in bufc.value.decode().split(","):
                if tag:
                    cc = ctypes.c_int64(0)
                    st_ms[stage_of_tag(tag)] += lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(cc)) / 3
            same_c = int(np.sum((r_c["song"] == res["song"][:nq_c]) & (r_c["offset"] == res["offset"][:nq_c])))
            batch_curve["points"].append({"queries": nq_c, "windows": nw, "ms_per_group": round(1e3 * el_c, 4),
                                          "segments_per_s": round(nw / el_c, 1),
                                          "kernel_ms_by_stage": {kk: round(v, 4) for kk, v in st_ms.items()},
                                          "decisions_identical_to_the_timed_step": "%d/%d" % (same_c, nq_c)})
        top = batch_curve["points"][-1]["segments_per_s"] if batch_curve["points"] else None
        for pt in batch_curve["points"]:
            pt["fraction_of_largest_group_rate"] = round(pt["segments_per_s"] / top, 4)
        log("batch curve:", [(pt["windows"], pt["segments_per_s"]) for pt in batch_curve["points"]])
        lib.pfann_prof_reset()

    # ---------------------------------- single-query scan regime (HBM-bound; SURVEY §8d note)
    single = None
    if prof and world == 1:
        import ctypes
        q19 = emb[:QUERY_SEGS].contiguous()
        if os.environ.get("PFANN_BENCH_SLEEP"):
            time.sleep(float(os.environ["PFANN_BENCH_SLEEP"]))
        if os.environ.get("PFANN_BENCH_FREE"):
            del eng
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        for _ in range(30):                    # short kernels separated by host syncs: let the clocks settle
            index.search(q19, k)
        lib.pfann_prof_reset()
        lib.pfann_prof_enable(1)
        t1 = time.perf_counter()
        for _ in range(50):
            index.search(q19, k)
        torch.cuda.synchronize()
        call_us = 1e6 * (time.perf_counter() - t1) / 50
        lib.pfann_prof_enable(0)
        cnt = ctypes.c_int64(0)
        ms = lib.pfann_prof_elapsed_ms(b"scan_topk", ctypes.byref(cnt))
        us = 1e3 * ms / max(cnt.value, 1)
        # default: the pass streams the shard's fp16 copy as a pre-filter (exact fp32 re-scoring in the select);
        # PFANN_SMALL_F32=1: the fp32 rows themselves
        elt = 4 if os.environ.get("PFANN_SMALL_F32") else 2
        gbs = (r_hi - r_lo) * d * elt / (us * 1e-6) / 1e9
        call_kernels = {}
        buf1 = ctypes.create_string_buffer(4096)
        lib.pfann_prof_tags(buf1, 4096)
        for tag in buf1.value.decode().split(","):
            if tag:
                c1 = ctypes.c_int64(0)
                t_ms = lib.pfann_prof_elapsed_ms(tag.encode(), ctypes.byref(c1))
                call_kernels[tag] = round(1e3 * t_ms / 50, 1)               # us per search call
        log("single-query search call, us per kernel tag:", call_kernels)
        # the same call without the profiling events around every launch
        for _ in range(10):
            index.search(q19, k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(200):
            index.search(q19, k)
        torch.cuda.synchronize()
        call_us = 1e6 * (time.perf_counter() - t1) / 200
        single = {"kernel": "pfann::scan_small_kernel<128,%d,0> (one 19-row query vs the whole shard, full pass%s)"
                            % (elt, " over the fp16 copy; exact fp32 re-scoring of the survivors follows" if elt == 2 else ""),
                  "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM, "unit": "GB/s",
                  "frac": round(gbs / PEAK_HBM, 4), "pass_us": round(us, 1),
                  "algorithmic_bytes": (r_hi - r_lo) * d * elt, "search_call_us": round(call_us, 1),
                  "search_call_kernels_us": call_kernels}
        # the whole path for ONE 10 s query (PCM in HBM -> decision on the host), one db pass per query
        n1 = QUERY_SEGS
        pcm1 = pcm_dev[:q_len].contiguous()
        st1 = starts_dev[:n1].contiguous()
        qs1, ql1 = np.zeros(1, np.int64), np.full(1, n1, np.int32)

        def one():
            e1 = eng.embed_windows(eng.pcm16_to_mono(pcm1), st1)
            D1, I1 = index.search(e1, k)
            return index.match(e1, I1, qs1, ql1)[0]
        for _ in range(3):
            r1 = one()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            r1 = one()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / 20
        single["one_query_end_to_end"] = {"ms": round(1e3 * lat, 3), "segments_per_s": round(n1 / lat, 1),
                                          "same_decision_as_batched": bool(int(r1[0]["song"]) == int(res[0]["song"]) and
                                                                           int(r1[0]["offset"]) == int(res[0]["offset"]))}

    # ------------------------------------------------------------------------- hit-rate
    hits = near = exact = 0
    hit_js = range(Q) if emu <= 1 else range(my_q[0], my_q[1])        # emulation: only rank 0's queries are real
    for j in hit_js:
        if int(res[j]["song"]) == q_song[j]:
            hits += 1
            tm = int(res[j]["offset"]) * 0.5
            near += abs(tm - float(q_off[j])) <= 0.5
            exact += abs(tm - float(q_off[j])) <= 0.25
    # ------------------------------------------- CPU baseline + decision parity (rank 0)
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not use_sharded and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline_leg(args, params, sd, shard, song_pos, q_pcm_mine, res, k, n_rows)
    # N > 1 lines carry the N = 1 run's cpu_baseline (the contract times it on rank 0 at N = 1 only: at N > 1 no rank holds the
    # whole database): the driver runs N = 1, 2, 4, 8 back to back on one node, so the N = 1 run leaves its figure in the
    # system temp dir; without one, the committed record of this round's N = 1 run is quoted, and says so
    carry = os.path.join(tempfile.gettempdir(), "pfann_bench_cpu_baseline_n1.json")
    # the carry is only quoted by a run of the SAME bench.py and library (hashes), the same job and sample sizes, on the same
    # host, and at most six hours later: a stale file from another commit / box / workload is skipped, not passed off
    import hashlib
    import socket

    def _sha16(path):
        try:
            return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
        except OSError:
            return None
    job = {"db_songs": n_songs, "snr": args.snr, "queries": Q, "max_batch": args.max_batch,
           "cpu_queries": args.cpu_queries, "cpu_pool_queries": args.cpu_pool_queries,
           "bench_py_sha16": _sha16(os.path.abspath(__file__)),
           "lib_sha16": _sha16(os.path.join(REPO, "pfann_amd", "libpfann_amd.so")), "host": socket.gethostname()}
    CARRY_MAX_AGE_S = 6 * 3600
    if rank == 0 and cpu is not None and world == 1:
        try:
            json.dump({"job": job, "written_unix": time.time(), "cpu_baseline": cpu}, open(carry, "w"))
        except OSError:
            pass
    elif rank == 0 and cpu is None and world > 1 and not args.no_cpu_baseline:
        import glob
        committed = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "bench.json")), reverse=True)
        for src, label in [(carry, "the N = 1 run of this series on this node")] + \
                [(f, os.path.relpath(f, REPO) + " (committed N = 1 run of the default job, another box)") for f in committed]:
            try:
                got = json.load(open(src))
                if src == carry and (got.get("job") != job or not (0 <= time.time() - float(got.get("written_unix", 0)) <= CARRY_MAX_AGE_S)):
                    continue                       # (another commit, workload, host or day left it: not this series' baseline)
                got = got.get("cpu_baseline")
                if got and "value" in got:
                    cpu = dict(got, carried_from=label, measured_in_this_run=False)
                    break
            except (OSError, ValueError):
                continue
    if seam_before is not None:
        seam_info = dict(seam_after, before_fp16_leg={kk: seam_before[kk] for kk in ("gpu_call_us_median", "gpu_call_us_p95")})
    else:
        seam_info = None

    # ------------------------------------------- the drop-in CLIs from WAV files on disk (rank 0, N = 1)
    cli = None
    if rank == 0 and not args.force_sharded and emu <= 1 and not args.no_cli:
        # N > 1: the tools start their own N ranks (PFANN_GPUS = --gpus) while this job's other ranks wait at the barrier
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import cli_bench
        try:
            cli = cli_bench.run(args.cli_songs, args.cli_queries, args.snr, device=local_rank, log=log, gpus=world,
                                tool_timeout_s=CLI_LEG_TIMEOUT_S)
            if world > 1:
                cli["gpus_shared_with_the_bench_ranks"] = ("the tools' %d ranks ran on the GPUs where this job's ranks still held "
                                                           "their workspaces and database shards (idle, waiting at a barrier)" % world)
            if "builder" in cli:
                cli["cli_builder_segments_per_s"] = cli["builder"]["segments_per_s"]
                cli["cli_matcher_segments_per_s"] = cli["matcher"]["segments_per_s"]
                cli["library_step_segments_per_s"] = round(value, 1)
        except Exception as x:                                          # never lose the headline line to the side leg
            cli = {"error": repr(x)[:500]}

    if in_group and not args.no_cli:
        dist.barrier(group=meta_group)       # (the CLI leg ran on rank 0)
    if rank == 0 and args.dump_decisions:
        np.save(args.dump_decisions, np.stack([res["song"].astype(np.float64), res["offset"].astype(np.float64),
                                               res["score"].astype(np.float64)], 1))
    if rank == 0:
        out = {
            "metric": "query segments/sec, 10 s @ SNR %g queries vs %s-segment db (exact flat IP top-100 + sequence match)"
                      % (args.snr, "1M" if n_rows == 1000050 else str(n_rows)),
            "value": round(value, 1), "unit": "segments/s", "n_gpus": world, "steps": args.steps,
            "ranks_seen": ranks_info["ranks_seen"], "backend": ranks_info["backend"], "devices": ranks_info["devices"],
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "dtype_note": "all results exact fp32 (fp32 MFMA encoder; batched scan pre-filtered on fp16 MFMA with a rigorous "
                          "margin, then re-scored in fp32)",
            "data": "synthetic",
            "config": {"workload": ("%d-segment db = %d real synthetic 30 s songs x %d segments, every one embedded by the hot path "
                                    "(builder loop)" % (n_rows, n_songs, SEG_PER_SONG) if not args.filler_db else
                                    "%d-segment db (%d songs x %d segs, 48 real synthetic songs + unit-norm filler rows)"
                                    % (n_rows, n_songs, SEG_PER_SONG)) +
                                   "; %d x 10 s queries/step at SNR %g dB (%d segments), configs/default.json encoder "
                                   "(d=128,h=1024,u=32,fuller; seeded weights with calibrated output bias), top_k=100" %
                                   (Q, args.snr, n_seg),
                       "db_rows": n_rows, "queries_per_step": Q, "queries_per_step_per_gpu": Q // world,
                       "segments_per_step": n_seg,
                       "parallelism": "song-sharded db x%d" % world, "max_batch": args.max_batch},
            "top1_hit_rate": round(hits / len(hit_js), 4), "top1_near_0.5s": round(near / len(hit_js), 4),
            "top1_exact_0.25s": round(exact / len(hit_js), 4), "hit_rates_cover": "%d of %d queries" % (len(hit_js), Q),
            "roofline": roofline, "single_query_scan_roofline": single, "cpu_baseline": cpu, "oracle_decision_parity": parity,
            "builder": None if not builder_segs else {
                "value": round(builder_segs / builder_s, 1), "unit": "segments/s", "segments": builder_segs,
                "seconds": round(builder_s, 3),
                "what": "pfann_amd.builder.embed_files over this rank's songs: int16 PCM in pinned host memory -> H2D -> mono -> "
                        "windows -> log-mel -> encoder -> unit-norm fingerprints in HBM (builder.py:75-103's loop), "
                        "%d windows per launch group" % args.max_batch},
            "value_includes": "H2D of the query PCM from pinned host memory (SURVEY 8d: PCM-in-host-memory to decisions)",
            "step_overlap": ("none: one batch at a time" if two_deep is None else
                             "two batches deep: the H2D of batch i+1 runs on a side stream under the kernels of batch i and the "
                             "host reads batch i's decisions after launching batch i+1; all K batches' H2D, kernels and D2H lie "
                             "inside the timed region"),
            "serial": serial,
            "hbm_resident": pcie, "seq_score_seam": seam_info, "cli": cli,
            "alt_modes": alt, "batch_curve": batch_curve, "other_scaling_mode": other_mode, "scan_throughput": scan_throughput, "critical_path": critical_path, "collectives": collectives,
            "kernels": {t: {kk: (float("%.4g" % vv) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                        for t, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])},
        }
        print(json.dumps(out), flush=True)
    if in_group:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
