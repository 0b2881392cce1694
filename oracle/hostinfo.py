"""oracle/hostinfo.py -- TEST / MEASUREMENT INFRASTRUCTURE ONLY: how many host cores a checker / baseline pool can really use.

os.cpu_count() reports the machine (256 hardware threads on the GPU boxes); the container's CPU quota (cgroup cpu.max -- 16
cores' worth on the same boxes, round 4) and the affinity mask are what a multiprocessing.Pool gets.  A pool of 256 busy
processes on a 16-core quota runs slower than a pool of 16 (measured: speed-up 8.1 against 9.9 at 32) and hides what was
actually used; the pools of cpu_baseline.py / parity_check.py are sized by effective_cores() and report all three figures."""
import os


def cgroup_quota_cores():
    """CPU quota of this container in cores (float), None when unlimited / unknown."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()          # cgroup v2: "<quota|max> <period>"
        if txt[0] != "max":
            return float(txt[0]) / float(txt[1])
        return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())   # cgroup v1
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 else None
    except (OSError, ValueError):
        return None


def effective_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cgroup_quota_cores()
    if q is not None:
        n = min(n, max(1, int(round(q))))
    return max(1, n)


def describe() -> dict:
    return {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_quota_cores": cgroup_quota_cores(), "effective_cores": effective_cores()}


def single_thread_blas():
    """Before numpy is imported in a process whose children each do their own BLAS calls: one thread per process (a Pool of N
    processes x an N-thread BLAS each is N^2 threads on N cores)."""
    for k in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(k, "1")
