#!/usr/bin/env python3
"""bench.py -- frames/s scored (MFCC + GMM) on MI355X, BASELINE.json configs[1]:
39-dim MFCC+delta+delta-delta, 64-mixture diagonal GMMs, 100 speakers, 1 M synthetic frames.

One step = one pass of the hot path over the batch, inputs already resident in HBM:
int16 PCM (1000 utterances x ~10 s, 16 kHz) -> MFCC (25 ms / 10 ms frames, reference defaults
otherwise: FFT 2048, 50 filters, 13 ceps) -> CMVN -> delta/delta-delta -> all 100 speaker GMMs
-> per-utterance sums + argmax copied back to the host.

N > 1 (launched by torch.distributed.run, one rank per GPU): utterances shard by rank, models
replicated, no data-path collective; per-GPU work is fixed ("weak").  torch is
used only for the gloo barrier / max-over-ranks of the elapsed time -- never for device work.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 16000
MFCC_KW = dict(win_length_ms=25, win_shift_ms=10)      # cfg-0 framing; FFT 2048 / 50 filters / 13 ceps defaults
ND = 2                                                  # 13 -> 39 dims
N_MODELS, N_MIX, DIM = 100, 64, 39
N_UTT, FRAMES_PER_UTT = 1000, 1000
MODEL_SEED, AUDIO_SEED = 7, 2000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector peak == fp32 dense MFMA peak


def build_workload(rank, n_utt, frames_per_utt):
    """Synthetic speaker-like audio: one unique 10 s clip per speaker, repeated across the rank's
    utterances with a per-utterance gain so that no two clips are byte-identical."""
    from speaker_recognition_amd import synth
    L, shift = int(MFCC_KW["win_length_ms"] / 1000 * FS), int(MFCC_KW["win_shift_ms"] / 1000 * FS)
    n_samples = (frames_per_utt + ND - 1) * shift + L
    base = {}
    clips = []
    rng = np.random.default_rng(AUDIO_SEED + 7919 * rank)
    for u in range(n_utt):
        s = u % N_MODELS
        if s not in base:
            base[s] = synth.synth_speech(s, n_samples / FS + 0.01, FS, seed=AUDIO_SEED + 1000 * rank + s)[:n_samples]
        gain = 0.5 + 0.5 * rng.random()
        clips.append(np.round(base[s] * gain).astype(np.int16))
    models = [synth.synth_gmm(N_MIX, DIM, MODEL_SEED + s) for s in range(N_MODELS)]
    return clips, models


def cpu_baseline_leg(n_utt_sample, seconds):
    spec = dict(fs=FS, mfcc_kw=MFCC_KW, nd=ND, n_utt=n_utt_sample, seconds=seconds, n_models=N_MODELS,
                n_mix=N_MIX, dim=DIM, seed=AUDIO_SEED + 500000, model_seed=MODEL_SEED)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), json.dumps(spec)],
                         capture_output=True, text=True, timeout=1200)
    if out.returncode != 0:
        return None, "cpu baseline failed: " + out.stderr[-400:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=N_UTT, help="utterances per GPU (default: the cfg-1 size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-utts", type=int, default=12)
    ap.add_argument("--device-override", type=int, default=-1,
                    help="testing only: put every rank on this device (N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")          # host-side barrier / reduction only

    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM

    _lib.set_device(local_rank if args.device_override < 0 else args.device_override)
    clips, models = build_workload(rank, args.utts, FRAMES_PER_UTT)
    pcm = Batch.from_pcm(clips)                           # resident in HBM before the timed region
    ex = MfccExtractor(FS, **MFCC_KW)
    ms = ModelSet([GMM.from_arrays(*m) for m in models])
    n_frames = sum(max(0, ex.num_frames(len(c)) - ND) for c in clips)

    def barrier():
        _lib.synchronize()
        if dist is not None:
            dist.barrier()

    _lib.profile_enable(True)      # HIP-event kernel timers (pre-warms the runtime's event pool once)
    for _ in range(args.warmup):
        sums, arg = ex.predict_batch(ms, pcm, nd=ND)
    _lib.profile_reset()           # timed region starts with zeroed timers
    barrier()
    t0 = time.perf_counter()
    marks = []
    for _ in range(args.steps):
        sums, arg = ex.predict_batch(ms, pcm, nd=ND)
        marks.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("SR_BENCH_DEBUG"):
        print("step ends (ms since t0):", [round((m - t0) * 1e3, 2) for m in marks], "total", round(elapsed * 1e3, 2), file=sys.stderr)
    _lib.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank != 0:
        if dist is not None:
            dist.barrier()
        return

    ms_score, n_score = _lib.profile_get(_lib.T_SCORE)
    ms_mfcc, n_mfcc = _lib.profile_get(_lib.T_MFCC)
    ms_cmvn, n_cmvn = _lib.profile_get(_lib.T_CMVN)
    ms_fin, n_fin = _lib.profile_get(_lib.T_FINALIZE)
    avg_score_s = (ms_score / max(1, n_score)) * 1e-3
    flops_per_launch = float(n_frames) * N_MODELS * N_MIX * (4 * DIM + 6)
    bytes_per_launch = float(n_frames) * 4 * DIM                               # fp32 frame read once
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("gmm_score_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    achieved_tf = flops_per_launch / avg_score_s / 1e12 if avg_score_s > 0 else 0.0
    kname = _lib.last_score_kernel()
    intensity = flops_per_launch / bytes_per_launch
    if "bf16x3" in kname:
        # split-bf16 engine: every fp32 product is six bf16 part products on the bf16 matrix cores
        ks = int(kname.split("<")[1].split(",")[0])
        executed = float(n_frames) * N_MODELS * ((N_MIX + 31) // 32) * (6 * ks) * (2 * 32 * 32 * 16) / 32.0
        ratio = executed / flops_per_launch
        peak = BF16_PEAK_TFLOPS / ratio
        roof = {
            "kernel": kname + "; auto-selected engine",
            "bound": "mfma",
            "note": "compute-bound (%.0f flop/B vs machine balance ~20).  achieved = the algorithmic S*K*(4D+6) flops per "
                    "frame / launch time.  The kernel evaluates each fp32 product as six bf16 part products "
                    "(3-way exact split of both operands, fp32 accumulate) on the bf16 matrix cores, so it executes "
                    "%.2fx the algorithmic flops (6 products, contraction padded to 16*%d); peak = dense bf16 MFMA "
                    "%.0f TFLOP/s / %.2f.  For scale: the fp32 MFMA / fp32 vector peak is %.1f TFLOP/s."
                    % (intensity, ratio, ks, BF16_PEAK_TFLOPS, ratio, FP32_PEAK_TFLOPS),
            "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tf / peak,
            "executed_bf16_tflops": executed / avg_score_s / 1e12 if avg_score_s > 0 else 0.0,
            "bf16_dense_peak": BF16_PEAK_TFLOPS, "fp32_peak": FP32_PEAK_TFLOPS,
            "achieved_over_fp32_peak": achieved_tf / FP32_PEAK_TFLOPS,
        }
        dtype = "f32 (operands split exactly into 3 bf16 parts, 6 part products on the bf16 matrix cores, fp32 accumulate)"
    else:
        roof = {
            "kernel": kname + "; auto-selected engine",
            "bound": "mfma",
            "note": "compute-bound: arithmetic intensity S*K*(4D+6)/(4D) = %.0f flop/B vs machine balance ~20, so "
                    "HBM cannot be the bound; peak = dense fp32 MFMA = fp32 vector peak = 157.3 TFLOP/s; flops are "
                    "the algorithmic S*K*(4D+6) per frame" % intensity,
            "achieved": achieved_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved_tf / FP32_PEAK_TFLOPS,
        }
        dtype = "f32"
    roof.update({
        "traffic": traffic,
        "avg_launch_ms": 1e3 * avg_score_s, "launches": n_score,
        "hbm": {"achieved_GBps": bytes_per_launch / avg_score_s / 1e9 if avg_score_s > 0 else 0.0,
                "peak_GBps": HBM_PEAK_GBS,
                "frac": (bytes_per_launch / avg_score_s / 1e9 / HBM_PEAK_GBS) if avg_score_s > 0 else 0.0},
    })
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
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: 16 kHz synthetic PCM -> 13 MFCC (25/10 ms, FFT 2048, "
                               "50 filters) + CMVN + delta + delta-delta = 39 dims; 100 speaker GMMs x 64 "
                               "diagonal mixtures; %d utterances x %d frames per GPU" % (args.utts, FRAMES_PER_UTT),
                   "frames_per_gpu": n_frames, "speakers": N_MODELS, "mixtures": N_MIX, "dim": DIM,
                   "sharding": "utterances/%d ranks, models replicated, no collective" % world},
        "roofline": roof,
        "kernel_ms_per_step": {"mfcc_frames": ms_mfcc / max(1, args.steps), "cmvn_delta": ms_cmvn / max(1, args.steps),
                               "gmm_score": ms_score / max(1, args.steps), "finalize": ms_fin / max(1, args.steps)},
        "device": _lib.device_name(),
    }
    if world == 1:
        # the other engines on the same batch, outside the timed region (HIP-event time per launch)
        alt = {}
        _lib.profile_enable(True)
        for name, eng in (("fp32_mfma", 2), ("vector_alu", 1)):
            _lib.set_option("score_engine", eng)
            ex.predict_batch(ms, pcm, nd=ND)
            _lib.profile_reset()
            for _ in range(3):
                ex.predict_batch(ms, pcm, nd=ND)
            t_alt, n_alt = _lib.profile_get(_lib.T_SCORE)
            alt[name] = {"kernel": _lib.last_score_kernel(), "ms_per_launch": t_alt / max(1, n_alt),
                         "frac_of_fp32_peak": flops_per_launch / (t_alt / max(1, n_alt) * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
        _lib.set_option("score_engine", 0)
        _lib.profile_enable(False)
        result["roofline"]["other_engines"] = alt
    if not args.no_cpu_baseline and world == 1:
        cb, spec = cpu_baseline_leg(args.cpu_sample_utts, 10.04)
        if cb is None:
            result["cpu_baseline"] = {"error": spec}
        else:
            # parity of the device path on exactly the CPU sample
            from speaker_recognition_amd import synth
            sample = [synth.synth_speech(u % N_MODELS, 10.04, FS, seed=spec["seed"] + u) for u in range(spec["n_utt"])]
            dsums, darg = ex.predict_batch(ms, Batch.from_pcm(sample), nd=ND)
            csums = np.array(cb["sums"])
            result["cpu_baseline"] = {
                "value": cb["frames_per_s"], "unit": "frames/s", "cores": cb["cores"], "kind": cb["kind"],
                "sample": "%d utterances x 10.04 s (%d frames) of the same workload, all %d speaker models; GMM "
                          "scoring by the reference's compiled C++ score_batch (concurrency=cores), MFCC by the "
                          "float64 numpy restatement of MFCC.py (1 core)" % (spec["n_utt"], cb["n_frames"], N_MODELS),
                "mfcc_frames_per_s": cb["mfcc_frames_per_s"], "gmm_frames_per_s": cb["gmm_frames_per_s"],
            }
            result["parity"] = {
                "argmax_mismatches": int(np.sum(darg != np.array(cb["argmax"]))),
                "max_rel_sum_diff": float(np.max(np.abs(dsums - csums) / np.maximum(1.0, np.abs(csums)))),
                "utterances": spec["n_utt"],
            }
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()


if __name__ == "__main__":
    main()
