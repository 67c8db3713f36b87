"""Host-side fits under a container's CPU quota.

The reference's setup-time fits that stay on the host (``gmm_fit``'s float32 covariances + Cholesky ladder, sklearn /
SciPy fits when ``config.device_fit`` is off) run on torch's / BLAS's intra-op thread pools, which size themselves by the
machine (``os.cpu_count()``).  In a container with a CPU quota - the GPU boxes of this project: 256 hardware threads
visible, ``cpu.max`` = 16 CPUs - that oversubscribes the quota and the kernel throttles the process: ``gmm_fit`` on
50 000 x 512 rows took 716 ms with torch's default 128 threads and 60 ms with 16 (``tools/debug/gmm_fit_threads.py``).

``usable_cpus()``: CPUs this process may actually use = min(affinity mask, cgroup quota).
``host_compute()``: context manager that caps torch's intra-op pool (and, where ``threadpoolctl`` is importable, the
BLAS / OpenMP pools NumPy and SciPy use) at that number for the duration; a process already within the quota is left alone.
"""
import contextlib
import math
import os
import threading

__all__ = ["usable_cpus", "host_compute"]


def _cgroup_quota():
    """CPUs granted by the cgroup CPU controller (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``) or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, math.ceil(int(quota) / int(period)))
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return max(1, math.ceil(quota / period))
    except (OSError, ValueError):
        pass
    return None


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = _cgroup_quota()
    return max(1, min(n, quota) if quota else n)


_LOCK = threading.RLock()
_DEPTH = 0          # nested / concurrent host_compute() contexts in this process
_BEFORE = None      # torch's intra-op thread count seen by the FIRST context to enter
_LIMITER = None     # threadpoolctl limiter set by the first context


@contextlib.contextmanager
def host_compute():
    """Cap the process-global pools for the duration.  The effect is GLOBAL (torch's intra-op pool and the BLAS / OpenMP pools
    belong to the process, not to a thread): the cap is set by the first context to enter and lifted by the last one to leave
    (depth counter under a lock), so overlapping contexts on several threads cannot restore each other's 'before' value out of
    order.  Other CPU torch work of the process sees the cap while any context is open.  Keep device calls outside of it."""
    import torch

    global _DEPTH, _BEFORE, _LIMITER
    cap = usable_cpus()
    with _LOCK:
        _DEPTH += 1
        if _DEPTH == 1:
            _BEFORE = torch.get_num_threads()
            if _BEFORE > cap:
                torch.set_num_threads(cap)
            try:
                from threadpoolctl import threadpool_info, threadpool_limits

                # only the pools that are larger than the quota (a pool the user already made smaller stays as it is)
                over = {lib["user_api"]: cap for lib in threadpool_info() if lib.get("num_threads", 0) > cap}
                _LIMITER = threadpool_limits(limits=over) if over else None
            except Exception:  # threadpoolctl absent or a pool it cannot drive: torch's pool is capped anyway
                _LIMITER = None
    try:
        yield cap
    finally:
        with _LOCK:
            _DEPTH -= 1
            if _DEPTH == 0:
                if _LIMITER is not None:
                    _LIMITER.restore_original_limits()
                    _LIMITER = None
                if _BEFORE is not None and torch.get_num_threads() != _BEFORE:
                    torch.set_num_threads(_BEFORE)
                _BEFORE = None
