"""GPU: the N > 1 PROCESS path on the device (the reference's analogue is the Pool over utterances,
src/test/test-gmm.py:128-133).  tests/test_shard_gloo.py covers partition + gather on CPU with the per-shard compute stood in by
the oracle; here every rank runs the REAL fused device step (PCM -> MFCC -> CMVN/deltas -> every model -> sums + argmax) on its
own utterances through ``shard.predict_sharded`` -- ranks stacked on device 0, the one GPU of the test box -- and the gathered
result must equal the single-process result on the same utterances bit for bit.  And bench.py's N > 1 path is launched exactly
as the driver launches it (its own spawner, and torch.distributed.run), stacked on device 0, at a reduced size."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_UTT, N_SPK, K_MIX = 29, 14, 64          # ragged utterances; UBM + 14 speakers sharing sigma / weights (the matrix-core shared-sigma engine)
MFCC_KW = dict(win_length_ms=25, win_shift_ms=10)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _workload():
    """the same on every rank and in the parent: utterances of 0.5-2.2 s, a UBM + MAP-style speakers"""
    from speaker_recognition_amd import synth
    rng = np.random.default_rng(17)
    secs = rng.uniform(0.5, 2.2, N_UTT)
    pcm = [synth.synth_speech(u % N_SPK, float(secs[u]), 16000, seed=300 + u) for u in range(N_UTT)]
    ubm = synth.synth_gmm(K_MIX, 39, 99)
    raw = [ubm] + [synth.synth_map_speaker(ubm, 500 + s) for s in range(N_SPK)]
    return pcm, raw


def _device_step(pcm, raw, idx):
    from speaker_recognition_amd.core import Batch, MfccExtractor, ModelSet
    from speaker_recognition_amd.pygmm import GMM
    ex = MfccExtractor(16000, **MFCC_KW)
    ms = ModelSet([GMM.from_arrays(*m) for m in raw])
    if len(idx) == 0:
        return np.zeros((0, len(raw))), np.zeros(0, np.int32), ""
    sums, arg = ex.predict_batch(ms, Batch.from_pcm([pcm[i] for i in idx]), nd=2)
    from speaker_recognition_amd import _lib
    return sums, arg, _lib.last_score_kernel()


def _rank(rank, world, port, backend, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), SR_RENDEZVOUS=backend, SR_RDZV_NONCE="gpu-test-%d" % port)
    sys.path.insert(0, ROOT)
    if backend == "socket":
        sys.modules["torch"] = None           # the socket carrier must not need torch
    from speaker_recognition_amd import _lib
    from speaker_recognition_amd.shard import predict_sharded
    _lib.set_device(0)                        # ranks stacked on the one device of the test box
    pcm, raw = _workload()
    seen = {}

    def compute(idx):
        sums, arg, kernel = _device_step(pcm, raw, idx)
        seen["kernel"], seen["n"] = kernel, len(idx)
        return sums, arg

    arg, sums = predict_sharded(N_UTT, [len(p) for p in pcm], compute, len(raw), want_sums=True, backend=backend)
    q.put((rank, seen["n"], seen["kernel"], arg.tobytes(), sums.tobytes()))


@pytest.mark.parametrize("backend,world", [("socket", 2), ("socket", 3), ("gloo", 2)])
def test_ranks_on_the_device_equal_one_process(built_lib, backend, world):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")             # never fork a process that may hold a GPU runtime
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    pcm, raw = _workload()
    sums1, arg1, kernel1 = _device_step(pcm, raw, list(range(N_UTT)))       # ONE process, all utterances in one batch
    assert "h2" in kernel1                                                    # the shared-sigma matrix-core engine did the work
    assert sum(r[1] for r in res) == N_UTT and all(r[1] > 0 for r in res)     # every rank had a share and ran the device step
    assert all("h2" in r[2] for r in res)
    for r in res:                                                             # every rank holds the full, identical answer
        assert np.array_equal(np.frombuffer(r[3], np.int32), arg1)
        assert np.array_equal(np.frombuffer(r[4], np.float64).reshape(N_UTT, len(raw)), sums1)   # bit for bit
    assert np.all(arg1 >= 0)


BENCH_ARGS = ["--gpus", "2", "--device-override", "0", "--utts", "200", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
              "--cfg3-total-frames", "400000"]


def _check_line(out, tmp_blocks):
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    assert out.returncode == 0, out.stderr[-2000:]
    assert lines, out.stderr[-2000:]
    assert len(lines[-1]) <= 1500
    line = json.loads(lines[-1])                                              # what the driver parses: stdout's LAST line
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and abs(line["value"] - 2 * line["frames_per_s_per_gpu"]) <= 1e-6 * line["value"]
    assert 0.0 < line["scaling_efficiency_vs_rank0_alone"] < 1.5             # stacked on one device: about 0.5
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["achieved"] > 0 and rf["peak"] == 2500.0 and 0 < rf["frac"] < 1
    assert line["config"]["frames_per_gpu"] == 200 * 1000 and "workload" in line["config"]
    full = json.load(open(tmp_blocks))
    assert len(full["ranks"]) == 2 and [r["device"] for r in full["ranks"]] == [0, 0]
    assert all(r["setup_s"] > 0 and r["peak_rss_mb"] > 0 for r in full["ranks"])
    strong = full["configs[3]_strong_scaling"]
    assert "error" not in strong, strong
    assert strong["frames_total"] == 400000 and all(p["own_speaker_wins"] for p in strong["per_rank"])
    one = full["one_process_all_devices"]
    assert "error" not in one, one
    assert one["slots_2"]["all_utterances_decided"] and one["slots_2"]["slot_devices"] == [0, 0]
    return line


def test_bench_two_ranks_own_spawner(built_lib, tmp_path):
    blocks = str(tmp_path / "blocks.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_ARGS + ["--blocks-out", blocks],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT)
    _check_line(out, blocks)


def test_bench_two_ranks_as_the_driver_launches_it(built_lib, tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`"""
    pytest.importorskip("torch")
    blocks = str(tmp_path / "blocks.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + BENCH_ARGS + ["--blocks-out", blocks]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    _check_line(out, blocks)
