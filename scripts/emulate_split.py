#!/usr/bin/env python3
"""CPU emulation (numpy) of split-operand matrix-core arithmetic for the expanded quadratic form
  log2 density_k(x) = sum_d (A2_kd x'_d^2 + A1_kd x'_d) + C_k
against the float64 log-sum-exp, to size the error of a scheme BEFORE writing its kernel:
  bf16x3/6 : three bf16 parts per operand, six part products            (the round-1 engine)
  fp16x2/3 : two fp16 parts per operand, three part products a0b0 + a0b1 + a1b0
  fp16x2/4 : ... plus a1b1
Products are exact in fp32 for both; accumulation is emulated as an fp32 sum.
Usage: emulate_split.py [K] [D] [S] [N] [shared]"""
import sys
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_recognition_amd import synth  # noqa: E402

LOG2E = 1.0 / np.log(2.0)


def coeffs(model, center, escale):
    """A2, A1, C (float64) of one model for x' = (x - center) / 2^escale."""
    w, mu, sg = model
    mu = mu - center
    sc = 2.0 ** escale
    A2 = -LOG2E / (2 * sg ** 2) * sc ** 2
    A1 = LOG2E * mu / sg ** 2 * sc
    C = LOG2E * (np.log(w) - np.sum(np.log(np.sqrt(2 * np.pi) * sg), axis=1) - np.sum(mu ** 2 / (2 * sg ** 2), axis=1))
    return A2, A1, C


def split_bf16(v, parts):
    v = v.astype(np.float32)
    out = []
    r = v.copy()
    for _ in range(parts):
        u = r.view(np.uint32)
        hi = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)
        out.append(hi)
        r = (r - hi).astype(np.float32)
    return out


def split_fp16(v, parts, ftz=False):
    v = v.astype(np.float32)
    out = []
    r = v.copy()
    for _ in range(parts):
        hi = r.astype(np.float16)
        if ftz:
            hi = np.where(np.abs(hi) < np.float16(6.104e-5), np.float16(0), hi)
        hi = hi.astype(np.float32)
        out.append(hi)
        r = (r - hi).astype(np.float32)
    return out


def lse2(z):
    m = z.max(axis=1, keepdims=True)
    return (m[:, 0] + np.log2(np.sum(np.exp2(z - m), axis=1)))


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 39
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    shared = len(sys.argv) > 5 and sys.argv[5] == "shared"
    if shared:
        ubm = synth.synth_gmm(K, D, 7)
        models = [ubm] + [synth.synth_map_speaker(ubm, 100 + s) for s in range(S - 1)]
    else:
        models = [synth.synth_gmm(K, D, 7 + s) for s in range(S)]
    X = np.concatenate([synth.draw_frames(models[u % S], N // 4, 42 + u, outlier_frac=0.0) for u in range(4)]).astype(np.float64)
    center = np.mean(np.concatenate([m[1] for m in models]), axis=0)
    # per-dimension power-of-two scale: median sigma
    sig_all = np.concatenate([m[2] for m in models])
    escale = np.round(np.log2(np.exp(np.mean(np.log(sig_all), axis=0))))
    for scaled in (False, True):
        es = escale if scaled else np.zeros(D)
        xp = ((X - center) / 2.0 ** es).astype(np.float32).astype(np.float64)
        B = np.concatenate([xp ** 2, xp, np.ones((len(xp), 1))], axis=1)           # [N, 2D+1]
        res = {}
        amp = 0.0
        for name in ("f64", "f32", "bf16x3/6", "fp16x2/3", "fp16x2/3ftz", "fp16x2/4", "bf16x3/3"):
            worst, rms, cnt = 0.0, 0.0, 0
            for m in models:
                A2, A1, C = coeffs(m, center, es)
                amp = max(amp, float(np.max(np.sum(((m[1] - center) / m[2]) ** 2, axis=1))))
                A = np.concatenate([A2, A1, C[:, None]], axis=1)                   # [K, 2D+1]
                ref = lse2(B @ A.T)
                if name == "f64":
                    z = B @ A.T
                elif name == "f32":
                    z = (B.astype(np.float32) @ A.T.astype(np.float32)).astype(np.float64)
                else:
                    if name.startswith("bf16"):
                        a = split_bf16(A, 3)
                        b = split_bf16(B, 3)
                    else:
                        ftz = name.endswith("ftz")
                        a = split_fp16(A, 2, ftz)
                        b = split_fp16(B, 2, ftz)
                    pairs = {"bf16x3/6": [(2, 0), (1, 0), (1, 1), (0, 1), (0, 2), (0, 0)],
                             "bf16x3/3": [(1, 0), (0, 1), (0, 0)],
                             "fp16x2/3": [(1, 0), (0, 1), (0, 0)], "fp16x2/3ftz": [(1, 0), (0, 1), (0, 0)],
                             "fp16x2/4": [(1, 1), (1, 0), (0, 1), (0, 0)]}[name]
                    z = np.zeros((len(B), K), np.float32)
                    for (i, j) in pairs:
                        z = (z + b[j] @ a[i].T).astype(np.float32)
                    z = z.astype(np.float64)
                got = lse2(z)
                ll_ref = ref * np.log(2.0)
                rel = np.abs(got - ref) * np.log(2.0) / np.maximum(1.0, np.abs(ll_ref))
                worst = max(worst, float(rel.max()))
                rms += float(np.sum(rel ** 2))
                cnt += len(rel)
            res[name] = (worst, np.sqrt(rms / cnt))
        print("K=%d D=%d S=%d N=%d shared=%s scaled=%s amp=%.0f" % (K, D, S, len(X), shared, scaled, amp))
        for k, (w, r) in res.items():
            print("   %-12s max_rel %.3e  rms_rel %.3e" % (k, w, r))


if __name__ == "__main__":
    main()
