"""oracle/parity_check.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY.

Checker leg of bench.py: reads a pickle {name: {"models": [(w, mean, sigma), ...], "X": float64[n, D],
"offsets": int[U + 1], "device_sums": float64[U, S]}}, scores every utterance under every model
with the C restatement of the reference's arithmetic (oracle/gmm_oracle.c, mode 0 = what its C ABI
computes), and prints one JSON line {name: {"max_rel_sum_diff_vs_oracle": ..., ...}}.  Runs as a
subprocess: the benchmark process itself never imports oracle/."""
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import gmm_oracle as go
    if not os.path.exists(go.ORACLE_SO):
        go.build(ref=False)
    req = pickle.load(open(sys.argv[1], "rb"))
    out = {}
    for name, r in req.items():
        X = np.ascontiguousarray(r["X"], dtype=np.float64)
        off = np.asarray(r["offsets"])
        dev = np.asarray(r["device_sums"], dtype=np.float64)
        worst, am = 0.0, 0
        want = np.zeros_like(dev)
        for s, m in enumerate(r["models"]):
            ll = go.score_batch(go.GMMParams(*[np.asarray(a, dtype=np.float64) for a in m]), X)
            for u in range(len(off) - 1):
                want[u, s] = ll[off[u]:off[u + 1]].sum()
        worst = float(np.max(np.abs(dev - want) / np.maximum(1.0, np.abs(want))))
        if dev.shape[1] > 1:
            am = int(np.sum(np.argmax(dev, axis=1) != np.argmax(want, axis=1)))
        out[name] = {"max_rel_sum_diff_vs_oracle": worst, "argmax_mismatches": am,
                     "utterances": int(dev.shape[0]), "models": int(dev.shape[1]), "frames": int(len(X))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
