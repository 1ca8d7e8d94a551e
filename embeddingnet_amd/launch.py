"""One process per GPU without an external launcher: start the ranks, pin them, watch them.

The reference's multi-GPU switch (`GENERAL.gpu_ids`, /root/reference/tools/train.py:121-140) runs inside one process; here the
data-parallel world is one process per GPU (torch.distributed over RCCL), started by `bench.py --gpus N` or by
`tools/train.py` for a config that lists several GPUs.  This module is the part both share:

  * spawn(): N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started BEFORE anything in the parent touches the
    GPU (a process that has initialised HIP must never re-execute itself); the parent then POLLS all of them — when one
    exits non-zero the others are terminated and its code is returned at once (a rank that dies in start-up would otherwise
    leave its siblings waiting in the rendezvous or in their first collective until the process-group timeout); a
    parent-side time limit; SIGTERM / SIGINT forwarded to the children; a lost race for the rendezvous port (exit code
    EXIT_PORT_IN_USE from init_distributed) restarts the world on a fresh port.
  * pin_to_gpu_numa(): called by each rank before its first GPU call — the rank's threads stay on the cores of the NUMA node
    its GPU hangs off (sysfs: KFD topology -> DRM render node -> local_cpulist); no numactl / taskset hop, which would be an
    exec after the environment was set up.

No torch import here: the parent of a spawned world never needs it.
"""
import glob
import os
import signal
import socket
import subprocess
import sys
import time

EXIT_PORT_IN_USE = 98          # a rank could not bind / reach the rendezvous port (errno EADDRINUSE)
EXIT_TIMEOUT = 124             # the parent's time limit expired (as coreutils' timeout)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _terminate(kids, grace_s=10.0):
    for k in kids:
        if k.poll() is None:
            try:
                k.terminate()
            except OSError:
                pass
    t0 = time.time()
    while any(k.poll() is None for k in kids) and time.time() - t0 < grace_s:
        time.sleep(0.05)
    for k in kids:
        if k.poll() is None:
            try:
                k.kill()
            except OSError:
                pass
    for k in kids:
        try:
            k.wait(timeout=5)
        except (subprocess.TimeoutExpired, OSError):
            pass


def spawn(n, argv, timeout_s=None, retries=2, env=None, log=None):
    """Run `argv` as ranks 0..n-1 of one world on this node; returns the exit code of the first rank that failed (0 when all
    succeeded, EXIT_TIMEOUT when the time limit expired).  Rank 0 inherits stdout (it prints the result line); every rank
    inherits stderr.  timeout_s: None / 0 = no limit (EMBNET_SPAWN_TIMEOUT_S overrides)."""
    log = log or (lambda *a: print(*a, file=sys.stderr, flush=True))
    timeout_s = float(os.environ.get("EMBNET_SPAWN_TIMEOUT_S", timeout_s or 0)) or None
    base = dict(os.environ if env is None else env)
    for attempt in range(retries + 1):
        port = free_port()
        kids = []
        for r in range(n):
            # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver supports only dmabuf IPC; RCCL's intra-node transport and
            # CUDA-tensor sharing fail with `hipIpcGetMemHandle: invalid argument` without it (the image exports it; kept for
            # children started from a scrubbed environment)
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EMBNET_SPAWNED="1")
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            kids.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
        prev = {}

        def forward(sig, _frame):
            _terminate(kids)
            raise SystemExit(128 + sig)

        for sig in (signal.SIGTERM, signal.SIGINT):
            try:
                prev[sig] = signal.signal(sig, forward)
            except ValueError:                 # not the main thread (a test harness): no forwarding
                pass
        t0, rc = time.time(), 0
        try:
            while True:
                codes = [k.poll() for k in kids]
                bad = next(((r, c) for r, c in enumerate(codes) if c not in (None, 0)), None)
                if bad is not None:
                    rc = bad[1] if bad[1] > 0 else 128 - bad[1]          # a signal's negative code -> 128 + signal
                    alive = [r for r, c in enumerate(codes) if c is None]
                    log(f"[launch] rank {bad[0]} exited with code {bad[1]} after {time.time() - t0:.1f} s; terminating ranks {alive}")
                    _terminate(kids)
                    break
                if all(c == 0 for c in codes):
                    break
                if timeout_s and time.time() - t0 > timeout_s:
                    log(f"[launch] time limit of {timeout_s:.0f} s expired; terminating all ranks")
                    _terminate(kids)
                    rc = EXIT_TIMEOUT
                    break
                time.sleep(0.1)
        finally:
            for sig, h in prev.items():
                signal.signal(sig, h)
        if rc == EXIT_PORT_IN_USE and attempt < retries:
            log(f"[launch] rendezvous port {port} was taken before rank 0 bound it; restarting the world on a new port")
            continue
        return rc
    return rc


# ---- core affinity by GPU ------------------------------------------------------------------------------------------------
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_cpu_sets(kfd_root="/sys/class/kfd/kfd/topology/nodes", drm_root="/sys/class/drm"):
    """-> [set of CPUs local to HIP device i] (None where sysfs does not say).  HIP enumerates the KFD topology's GPU nodes
    in node order; each names its DRM render minor, whose PCI device lists the CPUs of its NUMA node."""
    out = []
    nodes = sorted(glob.glob(os.path.join(kfd_root, "*")), key=lambda p: int(os.path.basename(p)) if os.path.basename(p).isdigit() else 1 << 30)
    for node in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(node, "properties")).read().splitlines() if " " in l)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:          # a CPU node
            continue
        cpus = None
        try:
            dev = os.path.join(drm_root, f"renderD{int(props['drm_render_minor'])}", "device")
            numa = int(open(os.path.join(dev, "numa_node")).read())
            if numa >= 0:
                cpus = _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read()) or None
        except (OSError, KeyError, ValueError):
            cpus = None
        out.append(cpus)
    return out


def visible_device(local_rank):
    """The physical HIP device index behind `local_rank` under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (integers only)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            ids = [s.strip() for s in v.split(",") if s.strip()]
            if local_rank < len(ids) and ids[local_rank].isdigit():
                return int(ids[local_rank])
            return None
    return local_rank


def pin_to_gpu_numa(local_rank, cpu_sets=None):
    """Restrict this process to the cores of its GPU's NUMA node (intersected with the cores it may already use).  Returns a
    short description for the log; never raises — an unknown topology leaves the affinity alone.  EMBNET_PIN=0 turns it off."""
    if os.environ.get("EMBNET_PIN", "1") == "0":
        return "affinity unchanged (EMBNET_PIN=0)"
    try:
        have = os.sched_getaffinity(0)
        sets = gpu_cpu_sets() if cpu_sets is None else cpu_sets
        dev = visible_device(local_rank)
        local = sets[dev] if (dev is not None and dev < len(sets)) else None
        if not local:
            return f"affinity unchanged: {len(have)} cores (no NUMA information for device {dev})"
        want = have & local
        if not want:
            return f"affinity unchanged: {len(have)} cores (none of them local to device {dev})"
        os.sched_setaffinity(0, want)
        return f"pinned to {len(want)} cores local to device {dev} ({min(want)}-{max(want)})"
    except (OSError, AttributeError, ValueError) as e:
        return f"affinity unchanged ({e})"
