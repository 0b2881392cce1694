#!/usr/bin/env python3
"""bench.py -- frames/s scored (MFCC + GMM) on MI355X.

HEADLINE = BASELINE.json configs[2], the largest single-GPU configuration: 16 kHz synthetic PCM ->
13 MFCC (25/10 ms frames, FFT 2048, 50 filters) + CMVN + delta + delta-delta = 39 dims, scored
against a 512-mixture diagonal UBM + 200 speaker GMMs MAP-adapted from it (201 models x 512
mixtures per frame), 10 M frames per GPU.  Models are SURVEY.md 8d's synthetic ones (UBM: mu ~ N(0,1),
sigma ~ U(0.2,1.5), w ~ Dirichlet(1); speakers: means moved by alpha_k N(0, 0.3^2), sigma and weights
shared -- what gmmubm.cc:40-81 produces), written through the reference's 6-digit text format.  (A set
trained by EM + MAP on the device from this audio is the `trained_ubm_map` block.)

One step = one pass of the hot path over the batch, inputs already resident in HBM:
int16 PCM -> MFCC -> CMVN -> delta/delta-delta -> all 201 models -> per-utterance sums + argmax
copied back to the host.

`--gpus N`: one rank per GPU, utterances shard by rank, models replicated, no data-path collective
(SURVEY.md 8e); per-GPU work is fixed ("weak").  Launched by torch.distributed.run the ranks come
with RANK/LOCAL_RANK/WORLD_SIZE set; launched plainly with --gpus N > 1 this script spawns the N
ranks itself.  The barrier / max-over-ranks / gather of the per-rank figures go over the package's own
Unix-domain-socket rendezvous (speaker-recognition_amd/rendezvous.py): nothing here imports torch
(`--rendezvous gloo` keeps torch.distributed as the carrier).  Every rank pins itself to the cores of its
GPU's NUMA node.  At N > 1 the line also carries the weak-scaling efficiency against rank 0 running the same
step ALONE in the same run, the configs[3] strong-scaling split, and the ONE-process path over the N devices
(sr_multi_predict_pcm from page-locked host PCM).

Beside the headline the full record (bench_blocks.json) carries a `configs` block (N = 1 only, outside the timed region):
configs[0] end to end through the CLI surface with the reference's C++ + numpy MFCC timed beside it, configs[1], one rank's
configs[3] shard in full, configs[4] latencies, the 256-mixture x 39-dim point of BASELINE.json's north_star on the matrix
cores and on north_star's literal vector-ALU path, the feature stage in both precision modes, the headline from HOST PCM,
serving-size batches, a set trained on the device, a speaker's MAP enrolment from 512- / 2048-mixture UBMs, the reference's published
EM benchmark, the legacy ABI's per-speaker loop --
each with its own roofline and a parity sample checked by oracle/parity_check.py on all host cores (per frame on the device's
own features AND end to end from PCM against the float64 feature oracle).

oracle/ is never imported by this process: the CPU baseline (oracle/cpu_baseline.py) and the parity
samples (oracle/parity_check.py) run as subprocesses, outside every timed region, as checkers.

Prints ONE compact JSON line (<= 1500 characters: headline + roofline + cpu_baseline + parity figures) as the LAST line of
stdout on rank 0; the full record with every secondary block goes to bench_blocks.json (--blocks-out) and to stderr.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 16000
MFCC_KW = dict(win_length_ms=25, win_shift_ms=10)      # cfg-0 framing; FFT 2048 / 50 filters / 13 ceps defaults
ND = 2                                                  # 13 -> 39 dims
DIM = 39
FRAMES_PER_UTT = 1000
MFCC_PRECISION = 2                                      # sr_set_option("mfcc_precision"): 2 = float64 spectrum / ln / DCT (the library's default), 0 = fp32
CFG2_SPEAKERS, CFG2_MIX, CFG2_UTTS = 200, 512, 10000
CFG1_MODELS, CFG1_MIX, CFG1_UTTS = 100, 64, 1000
AUDIO_SEED, MODEL_SEED = 2000, 7
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA16_PEAK_TFLOPS = 2500.0    # dense bf16 / fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3       # fp32 vector peak == fp32 dense MFMA peak
FP64_PEAK_TFLOPS = 78.6        # fp64 vector peak
MFCC_FLOPS_PER_FRAME = 2.5 * 2048 * 11 + 3 * 1025 + 2 * 1989 + 50 + 2 * 13 * 50     # ~64.7 kflop
MFCC_BYTES_PER_FRAME = 2 * 160 + 4 * 13                                            # 372 B


# ------------------------------------------------------------------ multi-rank launch
def spawn_ranks(args):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks ourselves."""
    from speaker_recognition_amd import _lib
    n = args.gpus
    have = _lib.device_count()
    if args.device_override < 0 and have < n:
        sys.exit("bench.py: --gpus %d but only %d GPU(s) visible (use --device-override D to stack ranks on one device for testing)" % (n, have))
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    nonce = os.urandom(8).hex()                               # the job's own rendezvous name (two launches of one user never meet)
    for r in range(n):
        env = dict(os.environ, SR_RDZV_NONCE=nonce, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = None if r == 0 else subprocess.DEVNULL          # rank 0 prints the JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    for p in procs:
        rc = max(rc, p.wait())
    sys.exit(rc)


# ------------------------------------------------------------------ workload construction
def base_clips(n_speakers, n_samples):
    """One unique clip per synthetic speaker (the same on every rank)."""
    from speaker_recognition_amd import synth
    return [synth.synth_speech(s, n_samples / FS + 0.01, FS, seed=AUDIO_SEED + s)[:n_samples] for s in range(n_speakers)]


def make_pcm(base, n_utt, rank):
    """Concatenated int16 PCM of n_utt utterances: speaker u % S's clip under a per-utterance gain, so
    that no two utterances are byte-identical.  -> (cat, offsets)"""
    n = len(base[0])
    rng = np.random.default_rng(AUDIO_SEED + 7919 * (rank + 1))
    cat = np.empty(n_utt * n, dtype=np.int16)
    for u in range(n_utt):
        gain = 0.5 + 0.5 * rng.random()
        np.rint(base[u % len(base)] * gain, out=cat[u * n:(u + 1) * n], casting="unsafe")
    return cat, np.arange(n_utt + 1, dtype=np.int64) * n


def build_workload(rank, n_utt, frames_per_utt):
    """The configs[1] audio as a list of clips (tests/test_gpu_mfcc.py, scripts/time_mfcc.py) + its models."""
    from speaker_recognition_amd import synth
    L, shift = int(MFCC_KW["win_length_ms"] / 1000 * FS), int(MFCC_KW["win_shift_ms"] / 1000 * FS)
    n_samples = (frames_per_utt + ND - 1) * shift + L
    cat, off = make_pcm(base_clips(min(CFG1_MODELS, n_utt), n_samples), n_utt, rank)
    clips = [cat[off[u]:off[u + 1]] for u in range(n_utt)]
    return clips, [synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s) for s in range(CFG1_MODELS)]


def train_cfg2_models(ex, base, n_mix, em_iters=12):
    """UBM by EM on the speakers' own features, speakers by means-only MAP (gmmubm.cc:29-81), on the
    device, from two differently scaled renditions of every speaker's clip."""
    from speaker_recognition_amd.core import Batch
    from speaker_recognition_amd.pygmm import GMM
    rend = []
    for s, c in enumerate(base):
        rend.append(c)
        rend.append(np.rint(c * 0.7).astype(np.int16))
    fb = ex.extract_batch(Batch.from_pcm(rend), nd=ND)
    X = fb.download()
    off = fb.offsets()
    ubm = GMM(nr_mixture=n_mix, nr_iteration=em_iters, init_with_kmeans=1, seed=MODEL_SEED)
    ubm.fit(X[::2])
    spk = []
    for s in range(len(base)):
        g = GMM(nr_mixture=n_mix, nr_iteration=2)
        g.fit(X[off[2 * s]:off[2 * s + 2]], ubm=ubm)
        spk.append(g)
    return ubm, spk


# ------------------------------------------------------------------ rooflines
def executed_over_algorithmic(kname, S, K, D):
    """16-bit MFMA flops the scoring kernel executes per algorithmic flop S*K*(4D+6), from its name."""
    alg = float(S) * K * (4 * D + 6)
    tiles = (K + 31) // 32
    mfma_flops = 2.0 * 32 * 32 * 16 / 32.0          # per frame column: one 32x32x16 MFMA covers 32 frames
    if "h2s" in kname or "h2p" in kname or "h2m" in kname:
        kq, kl = (int(v) for v in kname.split("<")[1].split(">")[0].split(",")[:2])     # <KQF,KLF,waves=N>
        blocks = (S + 14) // 15
        if "models split" in kname:     # a workgroup per (tile, block), its four waves each forming the quadratic half themselves
            return blocks * tiles * (4 * kq + 15 * kl) * mfma_flops / alg
        if "h2p" in kname:      # the pipelined shape runs a block's images in stages of 4 and skips the stages that hold phantom models only
            last = S - 15 * (blocks - 1)
            images = 16 * (blocks - 1) + 4 * ((1 + last + 3) // 4)
            return tiles * (blocks * kq + (images - blocks) * kl) * mfma_flops / alg
        return blocks * tiles * (kq + 15 * kl) * mfma_flops / alg
    if "bx3_shared" in kname:
        kq, kl = (int(v) for v in kname.split("<")[1].split(">")[0].split(","))
        blocks = (S + 14) // 15
        return blocks * tiles * 6 * (kq + 15 * kl) * mfma_flops / alg
    if "split_kernel<f16x2" in kname or "split_kernel<bf16x3" in kname or "splitp_kernel<f16x2" in kname:
        ks = int(kname.split("<")[1].split(",")[1])
        prod = 3 if "f16x2" in kname else 6
        return S * tiles * prod * ks * mfma_flops / alg
    return None


def score_roofline(kname, n_frames, S, K, D, avg_s, hbm_measured):
    flops = float(n_frames) * S * K * (4 * D + 6)
    byts = float(n_frames) * 4 * D                         # fp32 frame read once
    ach = flops / avg_s / 1e12 if avg_s > 0 else 0.0
    ratio = executed_over_algorithmic(kname, S, K, D)
    if ratio:
        peak = MFMA16_PEAK_TFLOPS
        note = ("compute-bound (%.0f flop/B vs machine balance ~20).  achieved = ALGORITHMIC flops, S*K*(4D+6) per frame "
                "(SURVEY.md 8d), / HIP-event launch time; peak = the dense 16-bit MFMA peak the work runs on; frac = "
                "frac_algorithmic = achieved / peak.  The kernel evaluates each fp32 product as exact 16-bit part products on "
                "the matrix cores (fp32 accumulate) and executes %.2fx the algorithmic flops: frac_executed_mfma = that "
                "rate / peak = how busy the matrix pipe is.  fp32 MFMA / vector peak for scale: %.1f TFLOP/s."
                % (flops / byts, ratio, FP32_PEAK_TFLOPS))
    else:
        peak, note = FP32_PEAK_TFLOPS, "compute-bound; fp32 engine: peak = fp32 MFMA = fp32 vector peak"
    r = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak if peak else 0.0,
         "frac_algorithmic": ach / peak if peak else 0.0, "frac_executed_mfma": (ach * ratio / peak) if ratio else None,
         "note": note, "avg_launch_ms": 1e3 * avg_s,
         # long grids of the shared-sigma kernel are cut into several launches per pass (HISTORY.md 2.1 point 4): rocprofv3's
         # per-call average is avg_launch_ms / launches_per_pass
         "launches_per_pass": int(kname.split("[")[1].split()[0]) if "launches per pass" in kname else 1,
         "traffic": None, "traffic_note": "not collected for this block (the headline's is: rocprofv3 --pmc passes of this script, separate runs)",
         "algorithmic_bytes": byts,
         "achieved_over_fp32_peak": ach / FP32_PEAK_TFLOPS,
         "hbm": {"achieved_GBps": byts / avg_s / 1e9 if avg_s > 0 else 0.0, "peak_GBps": HBM_PEAK_GBS,
                 "frac": byts / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else 0.0,
                 "measured_copy_ceiling_GBps": hbm_measured,
                 "note": "north_star's HBM roofline: the path moves 4D B per frame against S*K*(4D+6) flops, so HBM "
                         "is not the active bound for any K*S > ~5 (SURVEY.md 7)"}}
    if ratio:
        r["executed_16bit_tflops"] = ach * ratio
        r["executed_over_algorithmic"] = ratio
    return r


def mfcc_roofline(n_raw_frames, avg_s, hbm_measured, precision=2):
    fl = MFCC_FLOPS_PER_FRAME * n_raw_frames
    by = MFCC_BYTES_PER_FRAME * n_raw_frames
    peak = FP64_PEAK_TFLOPS if precision == 2 else FP32_PEAK_TFLOPS
    return {"kernel": "mfcc_frames_fft2048_f64_kernel" if precision == 2 else "mfcc_frames_fft2048_kernel",
            "bound": "valu (%s vector ALU: an FFT is not a contraction) + LDS exchanges" % ("fp64" if precision == 2 else "fp32"),
            "achieved": fl / avg_s / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / avg_s / 1e12 / peak,
            "avg_launch_ms": 1e3 * avg_s, "flops_per_frame": MFCC_FLOPS_PER_FRAME, "bytes_per_frame": MFCC_BYTES_PER_FRAME,
            "hbm": {"achieved_GBps": by / avg_s / 1e9, "peak_GBps": HBM_PEAK_GBS, "frac": by / avg_s / 1e9 / HBM_PEAK_GBS,
                    "measured_copy_ceiling_GBps": hbm_measured},
            "note": "float64 spectrum, ln and DCT (mfcc_precision 2, the default): DESIGN.md 3.1, profiles/r05_mfcc_f64_notes.txt"
                    if precision == 2 else "fp32 throughout (mfcc_precision 0): DESIGN.md 3.2"}


# ------------------------------------------------------------------ HBM traffic of the dominant kernel (PMC)
def measure_traffic(kernel_substr, extra_args):
    """HBM-side bytes per scoring pass of the dominant kernel from the L2's fabric counters, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
    (they do not fit one), kernel trace only beside them; both are in KiB; on gfx950 FETCH_SIZE tallies a wide coalesced
    read at half its bytes -> x2; WRITE_SIZE is uncalibrated and taken as is.  Each pass is a run of THIS script
    (1 warm-up + 1 step, headline only) outside any timed region.  -> (bytes per pass or None, detail dict)"""
    import csv
    import glob
    import shutil
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    detail = {}
    total = 0.0
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
               "--no-config-blocks", "--no-traffic"] + extra_args
        try:
            run = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        except Exception as e:
            return None, {"error": "%s pass: %s: %s" % (counter, type(e).__name__, e)}
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if run.returncode != 0 or not files:
            return None, {"error": "%s pass failed (rc %d): %s" % (counter, run.returncode, (run.stderr or run.stdout)[-300:])}
        kib, n = 0.0, 0
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    kib += float(row["Counter_Value"])
                    n += 1
        shutil.rmtree(d, ignore_errors=True)
        if n == 0:
            return None, {"error": "no dispatch of %s in the %s pass" % (kernel_substr, counter)}
        passes = 2.0                                       # 1 warm-up + 1 step, each one scoring pass
        b = kib * 1024.0 / passes * (2.0 if counter == "FETCH_SIZE" else 1.0)
        detail[counter + "_bytes_per_pass"] = b
        detail[counter + "_dispatches"] = n
        total += b
    detail["method"] = ("rocprofv3 --kernel-trace --pmc <counter> of `bench.py --steps 1 --warmup 1` (headline only), one run per "
                        "counter; KiB x 1024, FETCH_SIZE x 2 (gfx950 tallies wide reads at half), WRITE_SIZE as reported; per "
                        "scoring pass (all launches of the kernel in it)")
    return total, detail



def kernel_times(_lib, steps):
    out = {}
    for name, kind in (("mfcc_frames", _lib.T_MFCC), ("cmvn_delta", _lib.T_CMVN), ("gmm_score", _lib.T_SCORE),
                       ("gmm_score_ref_prepass", _lib.T_SCORE_REF), ("finalize", _lib.T_FINALIZE)):
        ms, n = _lib.profile_get(kind)
        out[name] = {"ms_per_step": ms / max(1, steps), "launches": n}
    return out


def timed(fn, warmup, steps, barrier=None):
    for _ in range(warmup):
        fn()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    if barrier:
        barrier()
    return time.perf_counter() - t0, r


# ------------------------------------------------------------------ secondary configs (rank 0, N = 1)
def block_cfg1(_lib, ex, base, hbm, preq):
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    raw = [synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s) for s in range(CFG1_MODELS)]
    ms = ModelSet([GMM.from_arrays(*m) for m in raw])
    cat, off = make_pcm(base[:CFG1_MODELS], CFG1_UTTS, 0)
    pcm = Batch.from_pcm((cat, off))
    n_frames = CFG1_UTTS * FRAMES_PER_UTT
    step = lambda: ex.predict_batch(ms, pcm, nd=ND)
    step(); step()
    _lib.synchronize()
    _lib.profile_reset()
    el, (sums, arg) = timed(step, 0, 10, _lib.synchronize)
    kt = kernel_times(_lib, 10)
    kname = _lib.last_score_kernel()
    # parity samples for the pool of host cores (oracle subprocess): the first 200 utterances x ALL 100 models -- per frame on
    # their own batch (features = the device's own MFCC output), per utterance against the sums of the TIMED 1 M-frame pass --
    # and the feature stage of the same 200 utterances against the float64 restatement of MFCC.py
    n200 = min(200, CFG1_UTTS)
    fb = ex.extract_batch(Batch.from_pcm((cat[:off[n200]], off[:n200 + 1])), nd=ND)
    X200 = fb.download()
    s200, a200, f200 = ms.score(fb, frame_ll=True)
    preq["configs[1]"] = {"models": raw, "X": X200, "offsets": fb.offsets(), "device_sums": sums[:n200], "device_frame_ll": f200}
    preq["mfcc_configs[1]_audio"] = {"kind": "mfcc", "pcm": cat[:off[n200]], "sample_offsets": off[:n200 + 1], "fs": FS, "mfcc_kw": MFCC_KW,
                                     "nd": ND, "device_feats": X200, "offsets": fb.offsets()}
    # END TO END (north_star's criterion): the first 64 utterances' per-frame log-likelihoods computed from PCM by the device against the
    # float64 feature oracle -> the reference's arithmetic, all 100 models
    n64 = min(64, n200)
    fo = fb.offsets()
    preq["configs[1]_from_pcm"] = {"kind": "pcm_ll", "pcm": cat[:off[n64]], "sample_offsets": off[:n64 + 1], "fs": FS, "mfcc_kw": MFCC_KW, "nd": ND,
                                   "models": raw, "device_frame_ll": f200[:, :fo[n64]], "offsets": fo[:n64 + 1], "device_sums": s200[:n64]}
    small_vs_timed = float(np.max(np.abs(s200 - sums[:n200]) / np.maximum(1.0, np.abs(s200))))
    return {"workload": "BASELINE.json configs[1]: 39-dim MFCC+delta+delta-delta, 100 speaker GMMs x 64 mixtures, %d utterances x %d frames" % (CFG1_UTTS, FRAMES_PER_UTT),
            "frames_per_s": n_frames * 10 / el, "ms_per_step": 1e3 * el / 10,
            "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kt.items()},
            "sum_of_kernels_ms": sum(v["ms_per_step"] for v in kt.values()),
            "roofline": score_roofline(kname, n_frames, CFG1_MODELS, CFG1_MIX, DIM, kt["gmm_score"]["ms_per_step"] * 1e-3, hbm),
            "mfcc_roofline": mfcc_roofline(CFG1_UTTS * (FRAMES_PER_UTT + ND), kt["mfcc_frames"]["ms_per_step"] * 1e-3, hbm),
            "parity": {"small_batch_sums_vs_timed_pass_max_rel": small_vs_timed}}


def block_legacy(_lib):
    """The reference's own calling pattern through the ten legacy symbols (gmmset.py:95-99, pygmm.py:120-132): one
    `score_all(gmm_s, double **X, n, dim, concurrency)` per speaker and utterance, rows handed over as a pointer array
    of float64 rows.  The library keeps the last uploaded utterance on the device (fp32, keyed by a hash of its
    contents), so the S calls of an utterance upload it once.  Timed beside the fused call on the same work."""
    import ctypes as C
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    raw = [synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s) for s in range(CFG1_MODELS)]
    gm = [GMM.from_arrays(*m) for m in raw]
    utts = [np.ascontiguousarray(synth.draw_frames(raw[u], FRAMES_PER_UTT, 9000 + u), dtype=np.float64) for u in range(8)]
    L = C.CDLL(_lib.LIB_PATH)                       # the bare C ABI, bound the way the reference binds it
    pp = C.POINTER(C.POINTER(C.c_double))
    L.score_all.restype = C.c_double
    L.score_all.argtypes = [C.c_void_p, pp, C.c_int, C.c_int, C.c_int]

    def rows(X):
        arr = (C.POINTER(C.c_double) * len(X))()
        base = X.ctypes.data
        for i in range(len(X)):
            arr[i] = C.cast(base + i * X.shape[1] * 8, C.POINTER(C.c_double))
        return arr
    ptrs = [rows(X) for X in utts]

    def legacy():
        out = np.empty((len(utts), len(gm)))
        for u, X in enumerate(utts):
            for s, g in enumerate(gm):
                out[u, s] = L.score_all(g.gmm, ptrs[u], len(X), DIM, 1)
        return out
    legacy()
    t0 = time.perf_counter()
    got = legacy()
    t_legacy = time.perf_counter() - t0
    ms = ModelSet(gm)
    fb = Batch.from_features([u.astype(np.float32) for u in utts])
    ms.score(fb)
    t0 = time.perf_counter()
    sums, arg = ms.score(fb)
    t_fused = time.perf_counter() - t0
    n = len(utts) * FRAMES_PER_UTT
    return {"workload": "%d utterances x %d frames x %d dims against %d models x %d mixtures: one legacy score_all(double **) call per "
                        "(utterance, speaker) as gmmset.py:95-99, vs one fused sr_score_batch_set call" % (len(utts), FRAMES_PER_UTT, DIM, len(gm), CFG1_MIX),
            "legacy_calls": len(utts) * len(gm), "legacy_s": t_legacy, "legacy_us_per_call": 1e6 * t_legacy / (len(utts) * len(gm)),
            "legacy_frames_per_s_all_models": n / t_legacy, "fused_s": t_fused, "fused_frames_per_s_all_models": n / t_fused,
            "argmax_equal": bool(np.array_equal(np.argmax(got, axis=1), arg)),
            "max_rel_sum_diff_legacy_vs_fused": float(np.max(np.abs(got - sums) / np.maximum(1.0, np.abs(sums))))}


CFG3_S, CFG3_K, CFG3_TOTAL_FRAMES, CFG3_DISTINCT_UTTS = 1000, 2048, 100_000_000, 1250


def cfg3_models():
    from speaker_recognition_amd import synth
    ubm = synth.synth_gmm(CFG3_K, DIM, 99)
    w, mean, sigma = ubm
    alpha = ((w * 40.0 * CFG3_K) / (w * 40.0 * CFG3_K + 16.0))[:, None]
    spk = []
    for s in range(CFG3_S):
        rng = np.random.default_rng(500 + s)
        spk.append((w, mean + alpha * 0.3 * rng.standard_normal(mean.shape), sigma))
    return ubm, spk


def block_cfg3(_lib, hbm, preq, world=1, rank=0, barrier=None, total_frames=CFG3_TOTAL_FRAMES):
    """BASELINE.json configs[3] -- 2048-mixture UBM + 1000 MAP speakers, 100 M frames sharded by utterance over the GPUs of a
    node -- as ONE rank sees it: the whole model set and its share of the frames.  At world = 1 the rank takes the shard of
    an 8-GPU job (12.5 M frames, in full); at world = N > 1 the N ranks split the 100 M frames (strong scaling: the caller
    takes the max over ranks).  Features are drawn from the models (SURVEY.md 8d, 0.1 % outlier frames at +60):
    CFG3_DISTINCT_UTTS distinct utterances, repeated to the shard's size on the host (the kernels' time does not depend on
    the values; 2 GB per 12.5 M frames do not fit any cache)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    S, K, T = CFG3_S, CFG3_K, FRAMES_PER_UTT
    t_block = time.perf_counter()
    n_ranks = world if world > 1 else 8
    U = total_frames // T // n_ranks
    ubm, spk = cfg3_models()
    t0 = time.perf_counter()
    ms = ModelSet([GMM.from_arrays(*m) for m in [ubm] + spk])
    t_pack = time.perf_counter() - t0
    distinct = min(U, CFG3_DISTINCT_UTTS)
    first = rank * U                                     # this rank's utterances of the job: speaker = utterance % S
    base = np.concatenate([synth.draw_frames(spk[(first + u) % S], T, 9000 + first + u, outlier_frac=0.001) for u in range(distinct)])
    reps = (U + distinct - 1) // distinct
    X = np.tile(base, (reps, 1))[:U * T]
    feats = Batch.from_features(X, np.arange(U + 1, dtype=np.int64) * T)
    del X
    step = lambda: ms.score(feats)
    step()
    _lib.profile_reset()
    el, (sums, arg) = timed(step, 0, 1, barrier or _lib.synchronize)
    ms_k, n_k = _lib.profile_get(_lib.T_SCORE)
    ms_r, _ = _lib.profile_get(_lib.T_SCORE_REF)
    kname = _lib.last_score_kernel()
    n = U * T
    out = {"workload": "BASELINE.json configs[3]: 2048-mixture UBM + 1000 MAP speakers (all 1001 models), rank %d of %d: %d utterances x %d "
                       "frames = %d of the job's 100 M frames, IN FULL (features drawn from the models, 0.1 %% outliers; %d distinct "
                       "utterances repeated to the shard's size)" % (rank, n_ranks, U, T, n, distinct),
           "frames": n, "frames_per_s": n / el, "s_per_pass": el, "model_pack_upload_s": t_pack, "block_wall_s": time.perf_counter() - t_block,
           "roofline": score_roofline(kname, n, S + 1, K, DIM, (ms_k + ms_r) / max(1, n_k) * 1e-3, hbm),
           "parity": {"own_speaker_wins": bool(np.array_equal(np.argmax(sums[:, 1:], axis=1), (first + np.arange(U) % distinct) % S))}}
    if preq is not None:
        # parity sample for the pool of host cores: 20 utterances x ALL 1001 models, per frame (their own small batch) and
        # per utterance (the sums of the full shard's pass)
        n20 = 20
        sub = Batch.from_features(base[:n20 * T], np.arange(n20 + 1, dtype=np.int64) * T)
        s20, a20, f20 = ms.score(sub, frame_ll=True)
        preq["configs[3]_rank_shard"] = {"models_recipe": {"kind": "cfg3", "K": K, "dim": DIM, "S": S, "ubm_seed": 99, "spk_seed": 500},
                                         "X": base[:n20 * T], "offsets": np.arange(n20 + 1) * T, "device_sums": sums[:n20],
                                         "device_frame_ll": f20}
        out["parity"]["small_batch_sums_equal_full_pass"] = float(np.max(np.abs(s20 - sums[:n20]) / np.maximum(1.0, np.abs(s20))))
    return out


def block_published_em(_lib):
    """The reference's only PUBLISHED benchmark (doc/Final-Report-Complete/result.tex:41-49, img/time-comp.pdf): EM training,
    10 iterations, 256 mixtures, 13-dim MFCC, 512 k frames -- reference C++ ~475 s at 1 thread, ~125 s at 8, ~70 s at 16
    (hardware unstated); here through the same entry point (train_model: random-frame start, threshold off so that all 10
    iterations run), frames drawn from a 256-mixture model."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    n, K, D, iters = 512000, 256, 13, 10
    true = synth.synth_gmm(K, D, 5)
    X = synth.draw_frames(true, n, 11)
    GMM(K, nr_iteration=2, threshold=-1.0, seed=3).fit(X[:4000])          # warm-up
    _lib.profile_reset()
    g = GMM(K, nr_iteration=iters, threshold=-1.0, seed=3)
    t0 = time.perf_counter()
    it = g.fit(X)
    dt = time.perf_counter() - t0
    ms_e, n_e = _lib.profile_get(_lib.T_ESTEP)
    ms_s, n_s = _lib.profile_get(_lib.T_SCORE)
    gk = GMM(K, nr_iteration=iters, threshold=-1.0, seed=3, init_with_kmeans=1)
    t0 = time.perf_counter()
    gk.fit(X)
    dt_k = time.perf_counter() - t0
    ll = g.score_all(X[:50000]) / 50000
    flops = 4.0 * K * n * D * it                          # the statistics pass alone: 2 FMAs per (frame, mixture, dim)
    return {"workload": "reference_published_em: %d EM iterations, %d mixtures, %d dims, %d frames (result.tex:41-49)" % (it, K, D, n),
            "seconds": dt, "seconds_per_iteration": dt / it, "seconds_with_kmeans_ii_start": dt_k,
            "estep_stats_kernel_ms_total": ms_e, "score_kernel_ms_total": ms_s,
            "estep_stats_tflops": flops / (ms_e * 1e-3) / 1e12 if ms_e > 0 else None,
            "mean_ll_after": ll, "mean_ll_generating_model": GMM.from_arrays(*true).score_all(X[:50000]) / 50000,
            "reference_published_seconds": {"c++ 1 thread": 475, "c++ 8 threads": 125, "c++ 16 threads": 70, "scikit-learn": 2400},
            "vs_reference_16_threads": 70.0 / dt, "published_hardware": "unstated (2013 laptop/desktop class): not a like-for-like number"}


_POOL_SET = None


def _pool_predict_one(x):
    return _POOL_SET.predict_one(x)


def block_logged_predict(_lib):
    """The reference's own logged prediction run (log/final/final-log/nperson-newg-mix-t5.log:85-90, driver src/test/test-nperson.py:133-146):
    80 speakers, 50 test fragments of 5 s each (8 kHz, 32 / 16 ms frames: 311 frames), 32-mixture models on 34-dim features -- 378 s through
    pygmm on a multiprocessing.Pool, 53.6 s with scikit-learn (nperson-sklearn-mix-t5.log:85-90), hardware unstated.  Here: the same shape
    through the same surface (GMMSet.predict on a list of float64 host feature matrices: conversion, upload, one fused pass, argmax
    back), features drawn from the models, every call from HOST memory; and the reference's per-utterance loop (predict_one: a launch per utterance) beside it."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.gmmset import GMMSet
    from speaker_recognition_amd.pygmm import GMM
    S, K, D, per, T = 80, 32, 34, 50, 311
    raw = [synth.synth_gmm(K, D, 300 + s) for s in range(S)]
    gs = GMMSet(gmm_order=K)
    for s, m in enumerate(raw):
        gs._append(s, GMM.from_arrays(*m))
    utts, truth = [], []
    for s in range(S):
        X = synth.draw_frames(raw[s], per * T, 900 + s)
        for u in range(per):
            utts.append(X[u * T:(u + 1) * T].astype(np.float64))      # (what feature extraction returns: float64, MFCC.py:69-79)
            truth.append(s)
    gs.predict(utts[:64])                                             # warm-up: packed set, code objects
    t0 = time.perf_counter()
    pred = gs.predict(utts)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    one = [gs.predict_one(x) for x in utts[:400]]
    dt_one = (time.perf_counter() - t0) * (len(utts) / 400.0)
    kernel = _lib.last_score_kernel()
    # ... and as the reference's driver itself runs it (test-nperson.py:133-146): a multiprocessing.Pool of FORKED workers, one
    # predict_one per task -- forked after this process used the GPU, so every worker is served by its own helper process
    # (csrc/fork_proxy.cpp), one conversation and one fused pass per utterance
    pool_s, pool_ok, workers = None, None, 8
    try:
        import multiprocessing
        global _POOL_SET
        _POOL_SET = gs
        pool = multiprocessing.get_context("fork").Pool(workers)
        try:
            t0 = time.perf_counter()
            res = [pool.apply_async(_pool_predict_one, (x,)) for x in utts]
            pool.close()
            got = [r.get(timeout=240) for r in res]
            pool_s = time.perf_counter() - t0
            pool_ok = bool(got == pred)
        finally:
            pool.terminate()
    except Exception as e:
        pool_ok = "%s: %s" % (type(e).__name__, e)
    return {"workload": "reference's logged predict run: %d speakers x %d fragments x %d frames, %d mixtures x %d dims (nperson-newg-mix-t5.log:85-90)"
                        % (S, per, T, K, D),
            "frames": len(utts) * T, "seconds_batch": dt, "frames_per_s_all_models": len(utts) * T / dt,
            "seconds_per_utterance_loop": dt_one, "correct": int(sum(int(p == t) for p, t in zip(pred, truth))), "of": len(utts),
            "loop_agrees_with_batch": bool(one == pred[:400]), "kernel": kernel,
            "seconds_forked_pool": pool_s, "forked_pool_workers": workers, "forked_pool_agrees_with_batch": pool_ok,
            "reference_logged_seconds": {"pygmm + multiprocessing": 378.2, "scikit-learn + multiprocessing": 53.6},
            "vs_reference_pygmm": 378.2 / dt, "published_hardware": "unstated: not a like-for-like number"}


def block_multi_slot(_lib, ex, base):
    """The one-process multi-GPU path (sr_multi_predict_pcm: a host thread + model replica per slot, PCM from HOST memory
    in every call, uploaded and scored in up to 8 pieces without a host wait in between) on the configs[1] workload with 1 and
    2 slots on this one GPU, beside the resident-PCM step of the same work."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet, MultiPredictor
    from speaker_recognition_amd.pygmm import GMM
    raw = [synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s) for s in range(CFG1_MODELS)]
    gm = [GMM.from_arrays(*m) for m in raw]
    cat, off = make_pcm(base[:CFG1_MODELS], CFG1_UTTS, 0)
    ms = ModelSet(gm)
    pcm = Batch.from_pcm((cat, off))
    step = lambda: ex.predict_batch(ms, pcm, nd=ND)
    step(); step()
    el_res, (sums, arg) = timed(step, 0, 10, _lib.synchronize)
    out = {"workload": "configs[1] (%d utterances x %d frames, 100 x 64 mixtures): sr_multi_predict_pcm from host PCM "
                       "(%.0f MB per call) vs the resident-PCM step" % (CFG1_UTTS, FRAMES_PER_UTT, cat.nbytes / 1e6),
           "resident_pcm_ms_per_step": 1e3 * el_res / 10}
    for pinned in (False, True):
        if pinned:
            _lib.host_register(cat)            # what a serving loop does once with its PCM ring: the copy engines read it in place
        for slots, merge in ((1, 1), (2, 1), (2, 0)):
            # merge = 1 (the default): slots that share a device are one queue on it; 0: a thread and a share per slot
            _lib.set_option("multi_merge_same_device", merge)
            mp_ = MultiPredictor(gm, FS, n_slots=slots, **MFCC_KW)
            f = lambda: mp_.predict_concat(cat, off, nd=ND)
            first = []
            for _ in range(4):                 # a predictor's first calls are not its steady state (buffers and tables on the first; the
                t0 = time.perf_counter()       # runtime still runs the uploads behind each piece's kernels on the second and third)
                f()
                first.append(1e3 * (time.perf_counter() - t0))
            el, (s2, a2) = timed(f, 0, 10)
            out["slots_%d%s_%s" % (slots, "" if merge else "_a_thread_each", "caller_memory_page_locked" if pinned else "pageable_caller_memory")] = {
                "ms_per_call": 1e3 * el / 10, "first_four_calls_ms": first, "over_resident": (el / 10) / (el_res / 10), "argmax_equal": bool(np.array_equal(a2, arg)),
                "sums_bit_identical": bool(np.array_equal(s2, sums)), "slot_seconds": [float(v) for v in mp_.slot_seconds]}
            del mp_
        _lib.set_option("multi_merge_same_device", 1)
        if pinned:
            _lib.host_unregister(cat)
    out["pcie_floor_ms"] = cat.nbytes / 55e9 * 1e3
    out["note"] = ("every call moves the PCM host -> device; a slot uploads its utterances in up to 8 pieces on the copy stream and enqueues every "
                   "piece's kernels + result copies behind its upload event, one host wait at the end; pcie_floor_ms = the PCM at 55 GB/s; "
                   "floor of the whole call ~ max(copy, kernels) + the first piece; slots that share a device (both of slots_2 on this one-GPU box) are "
                   "one queue on it unless multi_merge_same_device = 0 (slots_2_a_thread_each_*)")
    return out


def block_serving_small(_lib, ex, base, ms, S, K, hbm):
    """The serving shape between a 61-frame window and a 10 M-frame batch: 1, 8 and 64 short utterances (300 frames = 3 s) per
    call against the whole 201 x 512 set -- one utterance against a 200-speaker set is what gui/interface.py:85-94 does per
    decision.  PCM from HOST memory in every call (in-place update of a device batch), per-call latency at the host,
    the scoring kernel's HIP-event time and its roofline fraction."""
    from speaker_recognition_amd.core import Batch
    L, shift = ex.FRAME_LEN, ex.FRAME_SHIFT
    T = 300
    n_samples = (T + ND - 1) * shift + L
    out = {"workload": "1 / 8 / 64 utterances x %d frames (16 kHz, %d samples each) per call against %d models x %d mixtures x %d dims, PCM from "
                       "host memory in every call" % (T, n_samples, S, K, DIM)}
    for U in (1, 8, 64):
        clips = [np.ascontiguousarray(base[u % len(base)][:n_samples]) for u in range(U)]
        cat = np.concatenate(clips)
        batch = Batch.from_pcm(clips)
        lat = []
        out_bufs = (np.zeros((U, S)), np.full(U, -1, np.int32))
        _lib.profile_enable(False)                          # (the HIP-event pairs around every kernel are ~20 us of a 140 us decision)
        try:
            for i in range(230):
                t0 = time.perf_counter()
                batch.update_pcm(cat)
                sums, arg = ex.predict_batch(ms, batch, nd=ND, out=out_bufs)
                lat.append((time.perf_counter() - t0) * 1e3)
        finally:
            _lib.profile_enable(True)
        _lib.profile_reset()
        for i in range(30):                                 # ... and the kernels' own times from a few calls with them
            batch.update_pcm(cat)
            sums, arg = ex.predict_batch(ms, batch, nd=ND, out=out_bufs)
        ms_k, n_k = _lib.profile_get(_lib.T_SCORE)
        ms_r, _ = _lib.profile_get(_lib.T_SCORE_REF)
        ms_m, _ = _lib.profile_get(_lib.T_MFCC)
        lat = np.array(lat[30:])
        kname = _lib.last_score_kernel()
        rf = score_roofline(kname, U * T, S, K, DIM, (ms_k + ms_r) / max(1, n_k) * 1e-3, hbm)
        out["utterances_%d" % U] = {"latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))},
                                    "decisions_per_s": U / (float(np.median(lat)) * 1e-3),
                                    "scoring_kernel_ms": (ms_k + ms_r) / max(1, n_k), "mfcc_kernel_ms": ms_m / max(1, n_k),
                                    "kernel": kname.split(" (")[0], "frac_algorithmic": rf["frac"], "frac_executed_mfma": rf["frac_executed_mfma"],
                                    "algorithmic_tflops": rf["achieved"], "all_finite": bool(np.all(np.isfinite(sums)))}
    # ... and one utterance per call whose LENGTH changes from call to call (2.5 .. 3.5 s: what a real stream of decisions looks like):
    # new layout in the same device batch (Batch.reset_pcm), the last decision checked against a fresh batch's
    clip = np.ascontiguousarray(base[0])
    rng = np.random.default_rng(1)
    lens = rng.integers(min(len(clip), int(2.5 * FS)), min(len(clip), int(3.5 * FS)) + 1, 230)
    batch = Batch.from_pcm([clip[:lens[0]]])
    lat = []
    _lib.profile_enable(False)
    try:
        for n in lens:
            t0 = time.perf_counter()
            batch.reset_pcm([clip[:n]])
            sums, arg = ex.predict_batch(ms, batch, nd=ND)
            lat.append((time.perf_counter() - t0) * 1e3)
    finally:
        _lib.profile_enable(True)
    fresh = ex.predict_batch(ms, Batch.from_pcm([clip[:lens[-1]]]), nd=ND)
    lat = np.array(lat[30:])
    out["utterances_1_varying_length"] = {"seconds_per_utterance": [2.5, 3.5], "latency_ms": {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99))},
                                          "last_decision_equals_fresh_batch": bool(np.array_equal(fresh[0], sums) and np.array_equal(fresh[1], arg))}
    return out


def block_one_process_multi(_lib, base, n_devices, device_override):
    """N > 1 only: sr_multi_predict_pcm with one slot per device (a host thread + model replica each, pinned to its GPU's NUMA
    node), the configs[1] workload PER DEVICE (weak scaling: N x 1000 utterances), PCM from page-locked host memory in every
    call -- beside the N-process number of the headline.  With --device-override the slots stack on one device (code path only)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import MultiPredictor
    from speaker_recognition_amd.pygmm import GMM
    gm = [GMM.from_arrays(*synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s)) for s in range(CFG1_MODELS)]
    out = {"workload": "configs[1] per device (1000 utterances x 1000 frames each, 100 x 64 mixtures), one process, one slot per device, "
                       "page-locked host PCM -> device in every call"}
    per = {}
    for n in sorted({1, n_devices}):
        cat, off = make_pcm(base[:CFG1_MODELS], CFG1_UTTS * n, 0)
        _lib.host_register(cat)
        try:
            mp_ = MultiPredictor(gm, FS, n_slots=n, **MFCC_KW)
            f = lambda: mp_.predict_concat(cat, off, nd=ND)
            f(); f()
            el, (s2, a2) = timed(f, 0, 5)
            per[n] = CFG1_UTTS * n * FRAMES_PER_UTT * 5 / el
            out["slots_%d" % n] = {"ms_per_call": 1e3 * el / 5, "frames_per_s": per[n], "slot_devices": mp_.slot_devices(),
                                   "slot_numa_nodes": mp_.slot_numa_nodes(), "slot_seconds": [float(v) for v in mp_.slot_seconds],
                                   "all_utterances_decided": bool(np.all(a2 >= 0))}
            del mp_
        finally:
            _lib.host_unregister(cat)
    if n_devices > 1 and 1 in per:
        out["scaling_efficiency_vs_one_slot"] = per[n_devices] / (n_devices * per[1])
    return out


def block_point256(_lib, hbm, preq):
    """north_star's '256 mixtures x 39-dim' point: ONE 256-mixture model, 2 M frames (100 distinct utterances drawn from the
    model with 0.1 % outlier frames, SURVEY.md 8d, each 20 times in the batch)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    m = synth.synth_gmm(256, DIM, 77)
    ms = ModelSet([GMM.from_arrays(*m)])
    utts = [synth.draw_frames(m, 1000, 100 + u, outlier_frac=0.001) for u in range(100)]
    feats = Batch.from_features([utts[u % 100] for u in range(2000)])
    step = lambda: ms.score(feats)
    step()
    _lib.profile_reset()
    el, (sums, arg) = timed(step, 0, 5, _lib.synchronize)
    ms_k, n_k = _lib.profile_get(_lib.T_SCORE)
    kname = _lib.last_score_kernel()
    # parity: the WHOLE block (its 100 distinct utterances), per frame and per utterance
    s100, a100, f100 = ms.score(Batch.from_features(utts), frame_ll=True)
    preq["north_star_256x39"] = {"models": [m], "X": np.concatenate(utts), "offsets": np.arange(101) * 1000, "device_sums": sums[:100, :1],
                                 "device_frame_ll": f100}
    return {"workload": "north_star point: 1 model x 256 mixtures x 39 dims, 2 M frames resident",
            "frames_per_s": 2e6 * 5 / el,
            "roofline": score_roofline(kname, 2000000, 1, 256, DIM, ms_k / max(1, n_k) * 1e-3, hbm),
            "parity": None}


def block_vector_alu(_lib, hbm):
    """north_star's LITERAL path -- no matrix cores: the vector-ALU engine (`score_engine` 1: gmm_score_kernel<D,F,PK>, mixture
    parameters staged in LDS, wavefront log-sum-exp) at the 256 x 39 point and on configs[1]'s shape, beside the matrix-core engines
    that the dispatcher picks: frames/s, fraction of the fp32 vector peak, achieved HBM GB/s and its fraction."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    out = {}
    shapes = (("256x39", [synth.synth_gmm(256, DIM, 77)], 2000), ("configs[1]", [synth.synth_gmm(CFG1_MIX, DIM, MODEL_SEED + s) for s in range(CFG1_MODELS)], 1000))
    for name, raw, n_utt in shapes:
        ms = ModelSet([GMM.from_arrays(*m) for m in raw])
        utts = [synth.draw_frames(raw[u % len(raw)], 1000, 100 + u, outlier_frac=0.001) for u in range(50)]
        feats = Batch.from_features([utts[u % 50] for u in range(n_utt)])
        n, S, K = n_utt * 1000, len(raw), len(raw[0][0])
        row = {"frames": n, "models": S, "mixtures": K}
        want = None
        for label, eng in (("vector_alu", 1), ("auto", 0)):
            _lib.set_option("score_engine", eng)
            try:
                ms.score(feats)
                _lib.profile_reset()
                el, (sums, arg) = timed(lambda: ms.score(feats), 0, 3, _lib.synchronize)
                t_ms, n_k = _lib.profile_get(_lib.T_SCORE)
                avg = t_ms / max(1, n_k) * 1e-3
                flops = float(n) * S * K * (4 * DIM + 6)
                byts = float(n) * 4 * DIM
                row[label] = {"kernel": _lib.last_score_kernel()[:64], "kernel_ms": 1e3 * avg, "frames_per_s": n / avg,
                              "algorithmic_tflops": flops / avg / 1e12, "frac_fp32_valu_peak": flops / avg / 1e12 / FP32_PEAK_TFLOPS,
                              "hbm_GBps": byts / avg / 1e9, "hbm_frac_of_8TBps": byts / avg / 1e9 / HBM_PEAK_GBS}
                if want is None:
                    want = sums
                else:
                    row["engines_max_rel_sum_diff"] = float(np.max(np.abs(sums - want) / np.maximum(1.0, np.abs(want))))
            finally:
                _lib.set_option("score_engine", 0)
        out[name] = row
    out["note"] = ("BASELINE.json north_star asks for this path (LDS-staged parameters, wavefront log-sum-exp, no MFMA) and for >= 60 % of the HBM "
                   "roofline at 256 x 39: the point has 40 kflop per 156-byte frame, so at 100 % of the fp32 vector peak (157.3 TFLOP/s) it moves "
                   "613 GB/s = 7.7 % of HBM; the active bound is the ALU, and the matrix-core engines exist because of it")
    return out


def block_mfcc_precision(_lib, base, preq):
    """The feature stage in both precision modes on the configs[1] audio (1000 utterances x 1002 raw frames): kernel time, and 100
    utterances of each mode's output for the float64 oracle (SURVEY.md 8d gate: 1e-3 max / 1e-5 mean after CMVN).  What the float64
    spectrum costs and buys, in the driver-visible record."""
    from speaker_recognition_amd.core import Batch, MfccExtractor
    cat, off = make_pcm(base[:CFG1_MODELS], CFG1_UTTS, 0)
    pcm = Batch.from_pcm((cat, off))
    out = {"workload": "configs[1] audio: %d utterances x %d frames, 25/10 ms, FFT 2048, 13 MFCC + delta + delta-delta" % (CFG1_UTTS, FRAMES_PER_UTT)}
    n100 = 100
    for mode, name in ((0, "fp32"), (2, "float64_spectrum")):
        _lib.set_option("mfcc_precision", mode)
        try:
            ex = MfccExtractor(FS, **MFCC_KW)
            ex.extract_batch(pcm, nd=ND)
            ts = []
            for _ in range(5):
                _lib.profile_reset()
                ex.extract_batch(pcm, nd=ND)
                _lib.synchronize()
                ts.append(_lib.profile_get(_lib.T_MFCC)[0])
            fb = ex.extract_batch(Batch.from_pcm((cat[:off[n100]], off[:n100 + 1])), nd=ND)
            preq["mfcc_mode_" + name] = {"kind": "mfcc", "pcm": cat[:off[n100]], "sample_offsets": off[:n100 + 1], "fs": FS, "mfcc_kw": MFCC_KW,
                                         "nd": ND, "device_feats": fb.download(), "offsets": fb.offsets()}
            out[name] = {"mfcc_kernel_ms": float(np.median(ts)), "frames_per_s": CFG1_UTTS * (FRAMES_PER_UTT + ND) / (float(np.median(ts)) * 1e-3)}
        finally:
            _lib.set_option("mfcc_precision", 2)
    out["float64_over_fp32_time"] = out["float64_spectrum"]["mfcc_kernel_ms"] / out["fp32"]["mfcc_kernel_ms"]
    return out


def block_cfg2_from_host(_lib, gm, cat, off, sums_resident, arg_resident):
    """The headline's step with the PCM handed over in HOST memory on every call (what the C-ABI boundary does for a caller that owns
    its audio): sr_multi_predict_pcm on one slot, page-locked caller memory, uploads in pieces overlapped with the kernels."""
    from speaker_recognition_amd.core import MultiPredictor
    _lib.host_register(cat)
    try:
        mp_ = MultiPredictor(gm, FS, n_slots=1, **MFCC_KW)
        f = lambda: mp_.predict_concat(cat, off, nd=ND)
        first = []
        for _ in range(2):                                  # the first two calls of a shape are not the steady state: buffers, tables, and a
            t0 = time.perf_counter()                        # second call whose uploads the runtime still runs behind each piece's kernels
            f()
            first.append(1e3 * (time.perf_counter() - t0))
        el, (s2, a2) = timed(f, 0, 3)
        del mp_
    finally:
        _lib.host_unregister(cat)
    return {"workload": "configs[2] headline step from page-locked HOST PCM (%.2f GB per call), one slot" % (cat.nbytes / 1e9),
            "ms_per_step": 1e3 * el / 3, "first_two_calls_ms": first, "pcie_floor_ms": cat.nbytes / 55e9 * 1e3,
            "argmax_equal_resident": bool(np.array_equal(a2, arg_resident)), "sums_bit_identical_resident": bool(np.array_equal(s2, sums_resident))}


def block_cfg0(_lib):
    """BASELINE configs[0] end to end (src/speaker-recognition.py:52-90): 10 speakers x 30 s synthetic 16 kHz WAV each for enrolment and
    for prediction, 25/10 ms frames, 13 MFCC, one 16-mixture diagonal GMM per speaker.  Device: this package's ModelInterface (WAV ->
    MFCC kernels -> EM on the device -> fused scoring), timed here; CPU: oracle/cfg0_baseline.py in its own process on the same files."""
    from scipy.io import wavfile
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.cli import read_wav
    from speaker_recognition_amd.interface import ModelInterface
    kw, K, spk = dict(win_length_ms=25, win_shift_ms=10), 16, [3 * i for i in range(10)]
    tmp = tempfile.mkdtemp()
    enroll, test = [], []
    for sp in spk:
        e, t = os.path.join(tmp, "enroll_%d.wav" % sp), os.path.join(tmp, "test_%d.wav" % sp)
        wavfile.write(e, FS, synth.synth_speech(sp, 30.0, FS, seed=1000 + sp))
        wavfile.write(t, FS, synth.synth_speech(sp, 30.0, FS, seed=2000 + sp))
        enroll.append((str(sp), e))
        test.append((str(sp), t))
    m = ModelInterface(gmm_order=K, feature_kwargs=kw, lpc=False, gmm_kwargs={"seed": 1}, verbose=False)
    m.enroll("warm", *read_wav(enroll[0][1]))         # code objects + workspaces, not timed
    m.train()
    m = ModelInterface(gmm_order=K, feature_kwargs=kw, lpc=False, gmm_kwargs={"seed": 1}, verbose=False)
    t0 = time.perf_counter()
    for label, f in enroll:
        m.enroll(label, *read_wav(f))
    t1 = time.perf_counter()
    m.train()
    t2 = time.perf_counter()
    pred = m.predict_many([read_wav(f) for _, f in test])
    t3 = time.perf_counter()
    dev = {"enroll_features_s": t1 - t0, "train_s": t2 - t1, "predict_s": t3 - t2, "total_s": t3 - t0,
           "correct": int(sum(p == l for p, (l, _) in zip(pred, test))), "of": len(test)}
    # the same ten fits an iteration per launch (em_stats_engine 3; csrc/em.hip) beside the whole-fit kernel's (csrc/em_small.hip)
    try:
        _lib.set_option("em_stats_engine", 3)
        m3 = ModelInterface(gmm_order=K, feature_kwargs=kw, lpc=False, gmm_kwargs={"seed": 1}, verbose=False)
        for label, f in enroll:
            m3.enroll(label, *read_wav(f))
        t4 = time.perf_counter()
        m3.train()
        dev["train_iteration_per_launch_s"] = time.perf_counter() - t4
    finally:
        _lib.set_option("em_stats_engine", 0)
    cpu = None
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cfg0_baseline.py"), tmp], capture_output=True, text=True, timeout=400)
        cpu = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:
        cpu = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"workload": "BASELINE configs[0]: 10 x 30 s enrol + 10 x 30 s predict, 16 kHz WAV files, 25/10 ms, 13 MFCC, 16 mixtures per speaker",
            "device": dev, "cpu": cpu, "speedup_total": (cpu["total_s"] / dev["total_s"]) if cpu and "total_s" in cpu else None}


def block_stream(_lib):
    """configs[4]: 1 s windows of 8 kHz audio -> (LTSD VAD ->) MFCC -> 256-mixture speaker set -> decision;
    host-observed latency per window, H2D and D2H included."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet, ServingStream
    from speaker_recognition_amd.filters import VAD
    from speaker_recognition_amd.pygmm import GMM
    fs, S, K = 8000, 20, 256
    ex = MfccExtractor(fs)
    models = ModelSet([GMM.from_arrays(*synth.synth_gmm(K, 13, 7 + s)) for s in range(S)])
    audio = synth.synth_speech(3, 40.0, fs)
    win = Batch.from_pcm([audio[:fs]])
    lat = []
    for i in range(330):
        chunk = audio[(i % 39) * fs // 2:(i % 39) * fs // 2 + fs]
        t0 = time.perf_counter()
        win.update_pcm(chunk)
        ex.predict_batch(models, win, nd=0)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat[30:])
    rng = np.random.default_rng(5)
    floor = rng.normal(0, 60, len(audio)).astype(np.int16)
    gate = (np.arange(len(audio)) // (fs * 3 // 2)) % 2 == 0
    scene = (np.where(gate, audio // 2, 0) + floor).astype(np.int16)
    vad = VAD()
    vad.init_noise(fs, rng.normal(0, 60, 3 * fs).astype(np.int16))
    vwin = Batch.from_pcm([scene[:fs]])
    lv = []
    for i in range(330):
        chunk = scene[(i % 70) * fs // 2:(i % 70) * fs // 2 + fs]
        t0 = time.perf_counter()
        voiced, _ = vad.filter(fs, chunk)
        if len(voiced) > len(chunk) / 3 and ex.num_frames(len(voiced)) > 0:
            vwin.reset_pcm([voiced])
            ex.predict_batch(models, vwin, nd=0)
        lv.append((time.perf_counter() - t0) * 1e3)
    lv = np.array(lv[30:])
    st = ServingStream(ex, models, 1024, fs)
    cat = np.stack([audio[(j % 39) * fs // 2:(j % 39) * fs // 2 + fs] for j in range(1024)])
    st.submit(cat)
    t0 = time.perf_counter()
    for i in range(40):
        st.submit(cat)
        st.collect()
    st.collect()
    dt = (time.perf_counter() - t0) / 40
    pc = lambda a, q: float(np.percentile(a, q))
    return {"workload": "BASELINE.json configs[4]: 8 kHz, 1 s windows (61 frames at the reference's 32/16 ms framing), 13 MFCC, "
                        "20 speakers x 256 mixtures; one window per call, H2D + kernels + D2H, 300 windows",
            "mfcc_gmm_decision_latency_ms": {"p50": pc(lat, 50), "p99": pc(lat, 99)},
            "ltsd_vad_mfcc_gmm_decision_latency_ms": {"p50": pc(lv, 50), "p99": pc(lv, 99),
                                                      "note": "VAD = the published LTSD measure (pyssp is third-party and absent: parity unpinned)"},
            "double_buffered_1024_streams": {"tick_ms": dt * 1e3, "windows_per_s": 1024 / dt}}


def block_trained(_lib, ex, base):
    """The enroll path on the device: a 64-mixture UBM by EM and 40 speakers by means-only MAP
    (train_model / train_model_from_ubm, gmm.cc:581-653, gmmubm.cc:29-81) from this bench's own audio, then
    identification of held-out renditions of every speaker's clip."""
    from speaker_recognition_amd.core import Batch, ModelSet
    n_spk, K = 40, 64
    t0 = time.perf_counter()
    ubm, spk = train_cfg2_models(ex, base[:n_spk], K)
    t_train = time.perf_counter() - t0
    ms = ModelSet([ubm] + spk)
    test = [np.rint(base[s] * g).astype(np.int16) for g in (0.55, 0.9) for s in range(n_spk)]
    sums, arg = ex.predict_batch(ms, Batch.from_pcm(test), nd=ND)
    best = np.argmax(sums[:, 1:], axis=1)
    return {"workload": "EM (12 iterations, k-means|| start as the reference's) of a %d-mixture UBM on %d frames + means-only MAP of %d speakers, on the device; "
                        "then %d held-out utterances identified" % (K, n_spk * 1000, n_spk, len(test)),
            "train_s": t_train, "scoring_kernel": _lib.last_score_kernel(), "model_set": ms.info(),
            "identification_accuracy": float(np.mean(best == np.arange(len(test)) % n_spk)),
            "best_speaker_beats_ubm_fraction": float(np.mean(sums[:, 1:].max(axis=1) > sums[:, 0]))}


def block_map_enrolment(_lib):
    """Enrolment as configs[2] / [3] do it (gmmubm.cc:29-81): a speaker's MAP adaptation of a 512- and a 2048-mixture UBM on one
    utterance of 3000 frames x 39, the drop-in defaults (200 iterations, threshold 0.01) -- through the float64 iteration engine
    (csrc/em_f64.hip) and, beside it, an iteration per launch through the scoring engines (csrc/em.hip, em_stats_engine 3)."""
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    out = {"workload": "MAP enrolment of one speaker (3000 frames x 39, drop-in defaults) from a 512- / 2048-mixture UBM: ms per speaker"}
    try:
        for K in (512, 2048):
            ubm_raw = synth.synth_gmm(K, 39, 99)
            ubm = GMM.from_arrays(*ubm_raw)
            spk = [synth.draw_frames(synth.synth_map_speaker(ubm_raw, 500 + s), 3000, 10 + s) for s in range(4)]
            rec = {}
            for eng, name in ((0, "float64_engine_ms"), (3, "iteration_per_launch_ms")):
                _lib.set_option("em_stats_engine", eng)
                GMM(K).fit(spk[0], ubm=ubm)                       # workspaces, code objects
                best, its, means = 1e9, 0, None
                for s in (1, 2, 3):
                    m = GMM(K)
                    t0 = time.perf_counter()
                    its = m.fit(spk[s], ubm=ubm)
                    best = min(best, time.perf_counter() - t0)
                    means = m.params()[1]
                rec[name] = best * 1e3
                rec[name.replace("_ms", "_iterations")] = int(its)
                rec[name.replace("_ms", "_engine")] = int(_lib.last_em_stats_engine())
                rec.setdefault("_means", []).append(means)
            rec["means_max_abs_difference"] = float(np.max(np.abs(rec["_means"][0] - rec["_means"][1])))
            del rec["_means"]
            out["K=%d" % K] = rec
    finally:
        _lib.set_option("em_stats_engine", 0)
    return out


def usable_cores():
    """cores this container may use: affinity mask and cgroup CPU quota (os.cpu_count() reports the machine's)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(round(float(q) / float(p)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_leg(spec):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), json.dumps(spec)],
                         capture_output=True, text=True, timeout=1500)
    if out.returncode != 0:
        return None, "cpu baseline failed: " + out.stderr[-400:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), None


# ------------------------------------------------------------------ the line the driver parses
COMPACT_LIMIT = 1500           # characters; the driver keeps a 2000-character tail of stdout


def _r(v, sig=5):
    """numbers at `sig` significant digits (the full-precision values are in the blocks file)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float("%.*g" % (sig, float(v)))
    except (TypeError, ValueError):
        return None


def compact(result, blocks_file=None):
    """The headline as ONE short JSON object: BASELINE.json's metric with `roofline` and `cpu_baseline`, numbers only --
    every prose note, per-rank list and secondary block stays in the blocks file.  Never longer than COMPACT_LIMIT."""
    rf = result.get("roofline") or {}
    cb = result.get("cpu_baseline") or {}
    par = result.get("parity") or {}
    cfg = result.get("config") or {}
    e2e = par.get("end_to_end_cpu_pool_sample") or {}
    pf = par.get("per_frame_ll_from_pcm") or {}
    mf = par.get("mfcc_configs[2]_audio") or par.get("mfcc_configs[1]_audio") or {}
    line = {
        "metric": result.get("metric"), "value": _r(result.get("value"), 7), "unit": result.get("unit"),
        "n_gpus": result.get("n_gpus"), "steps": result.get("steps"), "warmup": result.get("warmup"),
        "frames_per_s_per_gpu": _r(result.get("frames_per_s_per_gpu"), 7),
        "ms_per_step": _r(result.get("ms_per_step"), 6), "higher_is_better": True, "scaling": result.get("scaling"),
        "vs_baseline": result.get("vs_baseline"), "dtype": str(result.get("dtype") or "").split(" ")[0][:8], "data": result.get("data", "synthetic"),
        "config": {"workload": "%s: 16 kHz PCM -> %s-dim MFCC+d+dd, %s-mix UBM + %s MAP speakers, %s M frames/GPU"
                               % ("configs[2]" if (cfg.get("frames_per_gpu"), cfg.get("models"), cfg.get("mixtures")) ==
                                  (CFG2_UTTS * FRAMES_PER_UTT, CFG2_SPEAKERS + 1, CFG2_MIX) else "configs[2] shape, non-default size",
                                  cfg.get("dim"), cfg.get("mixtures"), (cfg.get("models") or 1) - 1, _r((cfg.get("frames_per_gpu") or 0) / 1e6, 4)),
                   "frames_per_gpu": cfg.get("frames_per_gpu"), "models": cfg.get("models"), "mixtures": cfg.get("mixtures"),
                   "dim": cfg.get("dim"), "sharding": "utterances, no collective"},
        "roofline": {"kernel": str(rf.get("kernel", ""))[:48], "bound": rf.get("bound"), "achieved": _r(rf.get("achieved")),
                     "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _r(rf.get("frac"), 4),
                     "frac_executed_mfma": _r(rf.get("frac_executed_mfma"), 4), "avg_launch_ms": _r(rf.get("avg_launch_ms")),
                     "launches_per_pass": rf.get("launches_per_pass"), "traffic": _r(rf.get("traffic")),
                     "algorithmic_bytes": _r(rf.get("algorithmic_bytes")),
                     "hbm_GBps": _r((rf.get("hbm") or {}).get("achieved_GBps"), 4)},
        "cpu_baseline": ({"error": str(cb.get("error"))[:80]} if "error" in cb else
                         {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                          "sample": ("%s utt x 10.04 s, all %s models: reference C++ score_batch on Pool(%s) + f64 MFCC.py restatement"
                                     % (e2e.get("utterances"), e2e.get("models"), cb.get("cores"))) if cb else "measured at N = 1 only"}),
        "parity": {"argmax_mismatches": e2e.get("argmax_mismatches"), "utt_sum_max_rel": _r(e2e.get("max_rel_sum_diff"), 3),
                   "frame_ll_from_pcm_max_rel": _r(pf.get("max_rel"), 3), "frame_ll_from_pcm_frames": pf.get("frames"),
                   "mfcc_max_abs": _r(mf.get("max_abs_diff_vs_oracle"), 3), "mfcc_mean_abs": _r(mf.get("mean_abs_diff_vs_oracle"), 3),
                   "steps_bit_identical": par.get("last_two_steps_bit_identical")},
        # this box under its power cap (csrc/probe.hip): what the matrix pipe sustains with operands in registers / fed from LDS with
        # random bits, the clock it holds meanwhile, and the scoring kernel's executed rate against the fed figure -- boxes differ by
        # 3-6 %, this is what normalises them
        "clock": {"mfma_sustained_tflops": _r((rf.get("sustained_mfma") or {}).get("executed_tflops"), 4),
                  "mfma_streamed_tflops": _r((rf.get("sustained_mfma_streamed") or {}).get("executed_tflops"), 4),
                  "mhz_sustained": _r((rf.get("sustained_mfma") or {}).get("clock_mhz"), 4),
                  "mhz_streamed": _r((rf.get("sustained_mfma_streamed") or {}).get("clock_mhz"), 4),
                  "kernel_over_streamed": _r((rf.get("sustained_mfma_streamed") or {}).get("frac_executed_of_sustained"), 3)},
        "from_host_pcm_ms_per_step": _r(result.get("from_host_pcm_ms_per_step")),
        "mfcc_ms_per_step": _r((result.get("kernel_ms_per_step") or {}).get("mfcc_frames")),
        "blocks_file": os.path.basename(blocks_file) if blocks_file else None,
    }
    if (result.get("n_gpus") or 1) > 1:
        line["scaling_efficiency_vs_rank0_alone"] = _r(result.get("scaling_efficiency_vs_rank0_alone"), 4)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:                          # cannot happen with the fields above; never let it break the driver
        for k in ("parity", "from_host_pcm_ms_per_step", "mfcc_ms_per_step", "blocks_file", "clock"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text) > COMPACT_LIMIT:                          # still too long: some string came in far wider than any this script writes
        def clip(o, n):
            if isinstance(o, dict):
                return {k: clip(v, n) for k, v in o.items()}
            return o[:n] if isinstance(o, str) else o
        for n in (64, 24, 8):
            text = json.dumps(clip(line, n), separators=(",", ":"))
            if len(text) <= COMPACT_LIMIT:
                break
    return text


def emit(result, blocks_file):
    """full record -> blocks file (+ stderr); compact headline -> the LAST line of stdout"""
    full = json.dumps(result)
    try:
        with open(blocks_file, "w") as fh:
            fh.write(full + "\n")
    except OSError as e:
        print("bench.py: could not write %s: %s" % (blocks_file, e), file=sys.stderr)
        blocks_file = None
    print(full, file=sys.stderr, flush=True)
    print(compact(result, blocks_file), flush=True)


# ------------------------------------------------------------------ main
def peak_rss_mb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=CFG2_UTTS, help="utterances per GPU (default: the configs[2] size, 10 M frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-blocks", action="store_true", help="headline only (PMC / rocprof passes)")
    ap.add_argument("--cfg3-total-frames", type=int, default=CFG3_TOTAL_FRAMES,
                    help="testing only: total frames of the configs[3] job the ranks split (default: its stated 100 M)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc sub-runs that fill roofline.traffic")
    ap.add_argument("--cpu-sample-utts", type=int, default=4, help="utterances of the single-call CPU leg (reference DSO, concurrency = cores)")
    ap.add_argument("--cpu-pool-utts", type=int, default=0,
                    help="utterances of the Pool leg (reference DSO over (utterance, model group) tasks on every usable core, all models); "
                         "0 = 64 when the container may use >= 64 cores, else 32 (a bounded ~20 s of CPU work either way)")
    ap.add_argument("--rendezvous", choices=("socket", "gloo"), default=os.environ.get("SR_RENDEZVOUS", "socket"),
                    help="how the N ranks meet on the host: the package's Unix-domain socket (no torch) or torch.distributed over gloo")
    ap.add_argument("--blocks-out", default=os.path.join(ROOT, "bench_blocks.json"),
                    help="file the FULL record (headline + every secondary block, rooflines, parity samples) is written to; stdout "
                         "carries the compact headline line only")
    ap.add_argument("--device-override", type=int, default=-1,
                    help="testing only: put every rank on this device (N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus) and rank == 0:
        print("bench.py: note: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    grp = None
    if world > 1:
        from speaker_recognition_amd import rendezvous
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)                                    # (gloo announces its connections on stdout: ONE JSON line is the contract)
        try:
            grp = rendezvous.init(args.rendezvous)       # host-side barrier / gather only; the default imports no torch
            grp.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet

    dev = local_rank if args.device_override < 0 else args.device_override
    if dev >= _lib.device_count():
        sys.exit("bench.py: rank %d wants device %d but only %d visible" % (rank, dev, _lib.device_count()))
    _lib.set_device(dev)
    numa_node = _lib.bind_thread_near_device(dev)          # this rank's host thread next to its GPU's PCIe root (-1: platform does not say)
    _lib.set_option("mfcc_precision", MFCC_PRECISION)
    ex = MfccExtractor(FS, **MFCC_KW)
    L, shift = ex.FRAME_LEN, ex.FRAME_SHIFT
    n_samples = (FRAMES_PER_UTT + ND - 1) * shift + L
    base = base_clips(CFG2_SPEAKERS, n_samples)
    from speaker_recognition_amd import synth
    from speaker_recognition_amd.pygmm import GMM
    ubm_raw = synth.synth_gmm(CFG2_MIX, DIM, 99)
    raw_models = [ubm_raw] + [synth.synth_map_speaker(ubm_raw, 500 + s) for s in range(CFG2_SPEAKERS)]
    models = [GMM.from_arrays(*m) for m in raw_models]
    ms = ModelSet(models)
    cat, off = make_pcm(base, args.utts, rank)
    pcm = Batch.from_pcm((cat, off))                      # resident in HBM before the timed region
    n_frames = args.utts * FRAMES_PER_UTT
    S = len(models)

    def barrier():
        _lib.synchronize()
        if grp is not None:
            grp.barrier()

    # the results of consecutive steps go into two page-locked buffers in turn, as a serving loop would keep them (fresh numpy
    # arrays are 16 MB of page faults and a staged copy per step: 0.2 ms of the ~3.3 ms a step spends outside its kernels; the last
    # two steps' results stay side by side for the check below)
    outs = [(np.zeros((args.utts, S), dtype=np.float64), np.full(args.utts, -1, dtype=np.int32)) for _ in range(2)]
    for o in outs:
        _lib.host_register(o[0])
    step = lambda i: ex.predict_batch(ms, pcm, nd=ND, out=outs[i & 1])
    _lib.profile_enable(True)      # HIP-event kernel timers (pre-warms the runtime's event pool once)
    setup_s = time.perf_counter() - t_start        # process start -> workload resident, models packed (per rank; host-side, outside every timed region)
    for i in range(args.warmup):
        step(i)
    # N > 1: rank 0 runs the same step ALONE first (the others wait), so that the line carries its own weak-scaling reference
    alone_rate = None
    if grp is not None:
        barrier()
        if rank == 0:
            k = max(1, min(3, args.steps))
            t0 = time.perf_counter()
            for i in range(k):
                step(i)
            _lib.synchronize()
            alone_rate = n_frames * k / (time.perf_counter() - t0)
        barrier()
    _lib.profile_reset()           # timed region starts with zeroed timers
    fl0 = _lib.flush_stats()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sums, arg = step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    prev = outs[args.steps & 1][0] if args.steps > 1 else None       # the step before the last
    rank_rate = n_frames * args.steps / elapsed
    if grp is not None:
        elapsed = grp.all_max(elapsed)
        info = grp.all_gather({"rate": rank_rate, "device": dev, "numa_node": numa_node, "setup_s": setup_s, "peak_rss_mb": peak_rss_mb()})
        rates = [v["rate"] for v in info]
    else:
        info = [{"rate": rank_rate, "device": dev, "numa_node": numa_node, "setup_s": setup_s, "peak_rss_mb": peak_rss_mb()}]
        rates = [rank_rate]
    kt = kernel_times(_lib, args.steps)
    kname = _lib.last_score_kernel()
    fl1 = _lib.flush_stats()
    # ---- N > 1: north_star's configs[3] split -- the N ranks share the 100 M frames (strong scaling), every rank with
    #      the whole 2048-mixture UBM + 1000 speakers; wall time = the slowest rank's
    strong = None
    one_process = None
    if grp is not None and not args.no_config_blocks:
        # (secondary blocks: whatever goes wrong in them -- on this rank or, seen as a closed connection, on another -- must not
        # take the headline down; the ranks meet again at the barrier below or not at all)
        try:
            del pcm
            mine = block_cfg3(_lib, None, None, world, rank, barrier, args.cfg3_total_frames)
            wall = grp.all_max(mine["s_per_pass"])
            per_rank = grp.all_gather({"frames": mine["frames"], "s_per_pass": mine["s_per_pass"], "own_speaker_wins": mine["parity"]["own_speaker_wins"],
                                       "block_wall_s": mine["block_wall_s"], "peak_rss_mb": peak_rss_mb()})
            strong = {"workload": "BASELINE.json configs[3] at its stated size: 2048-mixture UBM + 1000 MAP speakers, 100 M frames split by utterance "
                                  "over %d ranks (strong scaling; models replicated; no collective on the data path)" % world,
                      "frames_total": sum(p["frames"] for p in per_rank), "stated_frames": CFG3_TOTAL_FRAMES, "n_gpus": world, "wall_s": wall,
                      "frames_per_s": sum(p["frames"] for p in per_rank) / wall, "per_rank": per_rank,
                      "roofline_rank0": mine["roofline"], "scaling": "strong"}
        except Exception as e:
            strong = {"error": "%s: %s" % (type(e).__name__, e)}
        # the ONE-process path over the same N devices (a host thread + model replica per GPU, PCM from page-locked host memory in
        # every call): rank 0 drives it while the other ranks idle at the barrier below
        if rank == 0:
            try:
                one_process = block_one_process_multi(_lib, base, world, args.device_override)
            except Exception as e:
                one_process = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            grp.barrier()
        except Exception:
            pass
    if rank != 0:
        if grp is not None:
            try:
                grp.barrier()
            except Exception:
                pass
        return

    hbm = _lib.hbm_copy_gbps(1 << 30, 10)
    # what the matrix pipe sustains on this box under its power cap: a kernel of v_mfma_f32_32x32x16_f16 chains only (csrc/probe.hip)
    sustained = _lib.mfma_peak_probe(60.0)
    # ... and on chains FED as the scoring kernel feeds them: a fresh A fragment from LDS per MFMA, random operand bits (probe mode 1)
    streamed = _lib.mfma_streamed_probe(60.0)
    kt1 = kt
    score_s = (kt1["gmm_score"]["ms_per_step"] + kt1["gmm_score_ref_prepass"]["ms_per_step"]) * 1e-3
    result = {
        "metric": "frames/sec scored (MFCC+GMM)",
        "value": world * n_frames * args.steps / elapsed,
        "unit": "frames/s",
        "frames_per_s_per_gpu": n_frames * args.steps / elapsed,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 (scoring: fp32 operands split exactly into 2 fp16 parts, 3 part products on the fp16 matrix cores, fp32 accumulate; "
                 "MFCC: %s; CMVN statistics and per-utterance sums in f64)"
                 % ("float64 spectrum / ln / DCT" if MFCC_PRECISION == 2 else "fp32"),
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[2]: 16 kHz synthetic PCM -> 13 MFCC (25/10 ms, FFT 2048, 50 filters) + CMVN + "
                               "delta + delta-delta = 39 dims; 512-mixture diagonal UBM + %d speaker GMMs MAP-style adapted from it (SURVEY.md 8d's "
                               "synthetic models: sigma and weights shared, means moved) = %d models x 512 mixtures per frame; "
                               "%d utterances x %d frames per GPU"
                               % (CFG2_SPEAKERS, S, args.utts, FRAMES_PER_UTT),
                   "frames_per_gpu": n_frames, "models": S, "mixtures": CFG2_MIX, "dim": DIM,
                   "sharding": "utterances/%d ranks, models replicated, no collective" % world},
        "rank_frames_per_s": rates,
        "ranks": info,
        "rank0_alone_frames_per_s": alone_rate,
        # weak scaling: N ranks, each with the N = 1 workload; 1.0 = every rank as fast in company as rank 0 alone in this same run
        "scaling_efficiency_vs_rank0_alone": (world * n_frames * args.steps / elapsed) / (world * alone_rate) if alone_rate else None,
        "rendezvous": args.rendezvous if world > 1 else None,
        "roofline": score_roofline(kname, n_frames, S, CFG2_MIX, DIM, score_s, hbm),
        "mfcc_roofline": mfcc_roofline(args.utts * (FRAMES_PER_UTT + ND), kt1["mfcc_frames"]["ms_per_step"] * 1e-3, hbm),
        "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in kt.items()},
        "kernel_launches_per_step": {k: v["launches"] / max(1, args.steps) for k, v in kt.items()},
        "sum_of_kernels_ms": sum(v["ms_per_step"] for v in kt.values()),
        "hbm_copy_ceiling_GBps": hbm,
        # frames whose log-likelihood sits where the reference's flushes of PARTIAL products decide: noted by the
        # engine per (32-frame tile, model), re-evaluated with the reference's own arithmetic inside the timed step (csrc/gmm_flush.hip)
        "partial_product_path_per_step": {"tile_model_pairs": (fl1[1] - fl0[1]) / max(1, args.steps),
                                          "frames_re_evaluated": (fl1[2] - fl0[2]) / max(1, args.steps)},
        "device": _lib.device_name(),
        "model_set": ms.info(),
        "parity": {"all_sums_finite": bool(np.all(np.isfinite(sums))),
                   "last_two_steps_bit_identical": bool(np.array_equal(prev, sums)) if prev is not None else None},
    }
    rf = result["roofline"]
    rf["sustained_mfma"] = {
        "executed_tflops": sustained[0], "clock_mhz": sustained[1],
        "frac_of_nominal_peak": sustained[0] / MFMA16_PEAK_TFLOPS,
        "frac_executed_of_sustained": (rf["executed_16bit_tflops"] / sustained[0]) if rf.get("executed_16bit_tflops") and sustained[0] > 0 else None,
        "note": "measured in this run: 12 waves per CU issuing nothing but v_mfma_f32_32x32x16_f16 for ~60 ms (sr_mfma_peak_probe). The nominal "
                "peak assumes 2.4 GHz; under the socket power cap a matrix-only kernel holds ~1.55 GHz, so this is the ceiling any MFMA kernel "
                "longer than a few ms has on this box; the scoring kernel's parts cost time in proportion to their energy "
                "(profiles/r03_h2p_parts.txt). `frac` above stays algorithmic flops / nominal peak."}
    rf["sustained_mfma_streamed"] = {
        "executed_tflops": streamed[0], "clock_mhz": streamed[1], "frac_of_nominal_peak": streamed[0] / MFMA16_PEAK_TFLOPS,
        "frac_executed_of_sustained": (rf["executed_16bit_tflops"] / streamed[0]) if rf.get("executed_16bit_tflops") and streamed[0] > 0 else None,
        "note": "the same chains with every MFMA's A fragment re-read from LDS (8 KiB images of random finite fp16 bit patterns), 8 random B "
                "fragments resident, chains of 8 links (sr_mfma_streamed_probe, ~60 ms): the matrix pipe + the LDS feed + operands whose bits "
                "toggle, nothing else -- no LDS-DMA stream, no exponentials, no barriers.  The ceiling a kernel of the scoring kernel's shape "
                "has on this box under its power cap; sustained_mfma above (operands held in registers for the whole launch) is a load no "
                "real kernel presents."}
    if strong is not None:
        result["configs[3]_strong_scaling"] = strong
    if one_process is not None:
        result["one_process_all_devices"] = one_process
    if world == 1 and not args.no_traffic:
        tb, td = measure_traffic(kname.split("<")[0].split()[0], ["--utts", str(args.utts)])
        result["roofline"]["traffic"] = tb
        result["roofline"]["traffic_unit"] = "bytes per scoring pass (HBM side of L2)"
        result["roofline"]["traffic_over_algorithmic"] = (tb / result["roofline"]["algorithmic_bytes"]) if tb else None
        result["roofline"]["traffic_note"] = td
    if world == 1 and not args.no_cpu_baseline:
        tmp = tempfile.mkdtemp()
        sample_models = [0] + list(range(1, 12))           # the UBM + 11 speakers: bounded CPU work
        files = []
        for i in sample_models:
            p = os.path.join(tmp, "m%d.model" % i)
            models[i].dump(p)
            files.append(p)
        all_files = []
        for i in range(S):
            p = os.path.join(tmp, "m%d.model" % i)
            if i not in sample_models:
                models[i].dump(p)
            all_files.append(p)
        spec = dict(fs=FS, mfcc_kw=MFCC_KW, nd=ND, n_utt=args.cpu_sample_utts, seconds=10.04, seed=AUDIO_SEED,
                    n_speakers=CFG2_SPEAKERS, model_files=files, single_core_models=2,
                    pool_utts=args.cpu_pool_utts or (64 if usable_cores() >= 64 else 32), pool_model_files=all_files)
        cb, err = cpu_baseline_leg(spec)
        if cb is None:
            result["cpu_baseline"] = {"error": err}
        else:
            sample = [synth.synth_speech(u % CFG2_SPEAKERS, 10.04, FS, seed=AUDIO_SEED + u) for u in range(spec["n_utt"])]
            sub = ModelSet([GMM.load(f) for f in files])
            dsums, darg = ex.predict_batch(sub, Batch.from_pcm(sample), nd=ND)
            csums = np.array(cb["sums"])
            scale = S / float(len(files))                  # the single-call leg scored len(files) of the S models
            t_total = (cb["t_mfcc_pool_s"] or cb["t_mfcc_s"]) + cb["t_gmm_s"] * scale
            single = {"value": cb["n_frames"] / t_total, "unit": "frames/s", "cores": cb["cores"],
                      "sample": "%d utterances x 10.04 s (%d frames); GMM: ONE score_batch call per model with concurrency = cores (the reference's "
                                "thread pool inside the call, gmm.cc:533-560) on %d of the %d models, scaled linearly to all of them; MFCC: float64 "
                                "numpy restatement of MFCC.py through multiprocessing.Pool(%d)" % (spec["n_utt"], cb["n_frames"], len(files), S, cb["pool_procs"]),
                      "gmm_frames_per_s_all_models": cb["gmm_frames_per_s"] / scale,
                      "gmm_frames_per_s_all_models_concurrency_1": (cb["gmm_frames_per_s_1core"] / scale) if cb["gmm_frames_per_s_1core"] else None,
                      "mfcc_frames_per_s_pool": cb["mfcc_frames_per_s_pool"], "mfcc_frames_per_s_1proc": cb["mfcc_frames_per_s_1proc"]}
            result["parity"].update({
                "cpu_sample_argmax_mismatches": int(np.sum(darg != np.array(cb["argmax"]))),
                "cpu_sample_max_rel_sum_diff": float(np.max(np.abs(dsums - csums) / np.maximum(1.0, np.abs(csums)))),
                "cpu_sample": "%d utterances x %d models vs the reference's C++ scorer on the float64 numpy MFCC" % (spec["n_utt"], len(files))})
            pool = cb.get("pool")
            if pool and "error" not in pool:
                # the reference at its best on this host: its own driver's parallelism (a process pool over utterances,
                # test-gmm.py:128-133), all 201 models, every core busy -- and the end-to-end parity sample that goes with it
                psample = [synth.synth_speech(u % CFG2_SPEAKERS, 10.04, FS, seed=AUDIO_SEED + u) for u in range(pool["utterances"])]
                psums, parg = ex.predict_batch(ms, Batch.from_pcm(psample), nd=ND)
                want = np.array(pool["sums"])
                result["cpu_baseline"] = {
                    "value": pool["frames_per_s"], "unit": "frames/s", "cores": pool["processes"], "cores_busy": pool["processes"], "kind": cb["kind"],
                    "host": cb.get("host"),
                    "sample": "%d utterances x 10.04 s (%d frames) of the same workload, ALL %d models: the reference's compiled C++ score_batch "
                              "(concurrency = 1 per call) over %d (utterance, %d-model group) tasks on a multiprocessing.Pool(%d) -- the parallelism "
                              "of the reference's own drivers (src/test/test-gmm.py:128-133) -- + the float64 numpy restatement of MFCC.py on "
                              "the same pool" % (pool["utterances"], pool["frames"], pool["models"], pool["tasks"], pool["models_per_task"], pool["processes"]),
                    "gmm_frames_per_s_all_models": pool["gmm_frames_per_s_all_models"], "mfcc_frames_per_s": pool["mfcc_frames_per_s"],
                    "t_gmm_pool_s": pool["t_gmm_pool_s"], "t_mfcc_pool_s": pool["t_mfcc_pool_s"],
                    "single_call_concurrency_cores": single}
                result["parity"]["end_to_end_cpu_pool_sample"] = {
                    "utterances": pool["utterances"], "models": pool["models"],
                    "argmax_mismatches": int(np.sum(parg != np.array(pool["argmax"]))),
                    "max_rel_sum_diff": float(np.max(np.abs(psums - want) / np.maximum(1.0, np.abs(want)))),
                    "what": "device: PCM -> MFCC -> CMVN/deltas -> all models; CPU: float64 restatement of MFCC.py -> the reference's compiled "
                            "C++ scorer (its own remez5 exp); per-utterance sums and argmax"}
            else:
                single["kind"] = cb["kind"]
                single["pool_leg"] = pool
                result["cpu_baseline"] = single
    if world == 1 and not args.no_config_blocks:
        blocks, preq = {}, {}
        # headline parity for the pool of host cores: the first 200 utterances x ALL 201 models -- per frame on their own
        # batch (features = the device's own MFCC output) and per utterance against the sums of the timed 10 M-frame pass
        try:
            n200 = min(200, args.utts)
            fb = ex.extract_batch(Batch.from_pcm((cat[:off[n200]], off[:n200 + 1])), nd=ND)
            s200, a200, f200 = ms.score(fb, frame_ll=True)
            preq["configs[2]_headline"] = {"models_recipe": {"kind": "cfg2", "K": CFG2_MIX, "dim": DIM, "S": CFG2_SPEAKERS, "ubm_seed": 99, "spk_seed": 500},
                                           "X": fb.download(), "offsets": fb.offsets(), "device_sums": sums[:n200], "device_frame_ll": f200}
            result["parity"]["small_batch_sums_vs_timed_pass_max_rel"] = float(np.max(np.abs(s200 - sums[:n200]) / np.maximum(1.0, np.abs(s200))))
            # END TO END (north_star's criterion): 64 utterances from PCM, the UBM + 24 speakers (bounded CPU work for the checker)
            n64, fo, idx = min(64, n200), fb.offsets(), list(range(25))
            preq["configs[2]_from_pcm"] = {"kind": "pcm_ll", "pcm": cat[:off[n64]], "sample_offsets": off[:n64 + 1], "fs": FS, "mfcc_kw": MFCC_KW, "nd": ND,
                                           "models_recipe": {"kind": "cfg2", "K": CFG2_MIX, "dim": DIM, "S": CFG2_SPEAKERS, "ubm_seed": 99, "spk_seed": 500},
                                           "model_index": idx, "device_frame_ll": f200[idx][:, :fo[n64]], "offsets": fo[:n64 + 1],
                                           "device_sums": s200[:n64][:, idx]}
            preq["mfcc_configs[2]_audio"] = {"kind": "mfcc", "pcm": cat[:off[n200]], "sample_offsets": off[:n200 + 1], "fs": FS, "mfcc_kw": MFCC_KW,
                                             "nd": ND, "device_feats": preq["configs[2]_headline"]["X"], "offsets": fo}
            del fb, f200
        except Exception as e:
            result["parity"]["headline_sample_error"] = "%s: %s" % (type(e).__name__, e)
        del pcm
        sums_res, arg_res = sums.copy(), arg.copy()
        for name, fn in (("configs[2]_from_host_pcm", lambda: block_cfg2_from_host(_lib, models, cat, off, sums_res, arg_res)),
                         ("serving_small_batch", lambda: block_serving_small(_lib, ex, base, ms, S, CFG2_MIX, hbm)),
                         ("configs[1]", lambda: block_cfg1(_lib, ex, base, hbm, preq)),
                         ("configs[3]_rank_shard", lambda: block_cfg3(_lib, hbm, preq, total_frames=args.cfg3_total_frames)),
                         ("configs[4]_streaming", lambda: block_stream(_lib)),
                         ("trained_ubm_map", lambda: block_trained(_lib, ex, base)),
                         ("map_enrolment", lambda: block_map_enrolment(_lib)),
                         ("reference_published_em", lambda: block_published_em(_lib)),
                         ("reference_logged_predict", lambda: block_logged_predict(_lib)),
                         ("legacy_abi_per_speaker_loop", lambda: block_legacy(_lib)),
                         ("sr_multi_predict_pcm_host_pcm", lambda: block_multi_slot(_lib, ex, base)),
                         ("north_star_256x39", lambda: block_point256(_lib, hbm, preq)),
                         ("north_star_literal_vector_alu", lambda: block_vector_alu(_lib, hbm)),
                         ("mfcc_precision_modes", lambda: block_mfcc_precision(_lib, base, preq)),
                         ("configs[0]_end_to_end", lambda: block_cfg0(_lib))):
            if os.environ.get("SR_BENCH_BLOCKS") and name not in os.environ["SR_BENCH_BLOCKS"].split(","):
                continue                                    # (experiments: a chosen subset of the blocks)
            try:
                blocks[name] = fn()
            except Exception as e:                          # a secondary block must not take the headline down
                blocks[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the parity samples of the blocks, checked by the oracle in its own process (checker only)
        try:
            import pickle
            path = os.path.join(tempfile.mkdtemp(), "parity.pkl")
            pickle.dump(preq, open(path, "wb"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "parity_check.py"), path],
                                 capture_output=True, text=True, timeout=900)
            par = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            for name, v in par.items():
                if isinstance(blocks.get(name), dict):
                    blocks[name]["parity"] = dict(blocks[name].get("parity") or {}, **v)
                elif name == "configs[2]_headline":
                    result["parity"]["headline_200_utterances_x_all_models"] = v
                elif name.startswith("mfcc_mode_"):
                    if isinstance(blocks.get("mfcc_precision_modes"), dict) and name[len("mfcc_mode_"):] in blocks["mfcc_precision_modes"]:
                        blocks["mfcc_precision_modes"][name[len("mfcc_mode_"):]]["vs_float64_oracle"] = {
                            k: v.get(k) for k in ("utterances", "frames", "max_abs_diff_vs_oracle", "mean_abs_diff_vs_oracle")}
                elif name.startswith("mfcc_"):
                    result["parity"][name] = v
                elif name.endswith("_from_pcm"):
                    result["parity"].setdefault("per_frame_ll_from_pcm", {"configs": {}})["configs"][name[:-len("_from_pcm")]] = v
                elif name == "_checker":
                    result["parity"]["checker"] = v
            pf = result["parity"].get("per_frame_ll_from_pcm")
            if pf:
                vs = list(pf["configs"].values())
                pf.update({"max_rel": max(v["max_rel_frame_ll_diff_vs_oracle"] for v in vs), "frames": sum(v["frames"] for v in vs),
                           "frame_model_pairs": sum(v["frames"] * v["models"] for v in vs),
                           "clamp_mismatches": sum(v["frames_with_different_clamp_decision"] for v in vs),
                           "argmax_mismatches": sum(v.get("argmax_mismatches", 0) for v in vs),
                           "gate": "north_star: |d| <= 1e-4 max(1, |LL|) per frame; device PCM -> LL against float64 MFCC.py restatement -> "
                                   "the reference's scoring arithmetic (oracle/parity_check.py, kind pcm_ll)"})
        except Exception as e:
            blocks["parity_error"] = "%s: %s" % (type(e).__name__, e)
        result["configs"] = blocks
        if isinstance(blocks.get("configs[2]_from_host_pcm"), dict):
            result["from_host_pcm_ms_per_step"] = blocks["configs[2]_from_host_pcm"].get("ms_per_step")
    emit(result, args.blocks_out)
    if grp is not None:
        grp.barrier()


if __name__ == "__main__":
    main()
