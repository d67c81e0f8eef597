"""One process per GPU on a single node, started from a parent that has not touched the GPU.

The launch convention is the one the reference's training script expects from its launcher
(train.py:424-436: `init_process_group(backend="nccl", init_method="env://")`): every rank finds RANK,
LOCAL_RANK, WORLD_SIZE, MASTER_ADDR and MASTER_PORT in its environment.  `spawn_ranks` is that launcher for
programs (bench.py) that are asked for N GPUs without one: it starts N fresh interpreters -- the parent never
initialises HIP, and no process that has is ever replaced by another -- relays rank 0's standard output and
returns the first non-zero exit status.
"""
import os
import socket
import subprocess
import sys
import threading
import time


def in_launched_rank(environ=None):
    """True inside a process that a launcher (this one or torch.distributed.run) started as one rank."""
    env = os.environ if environ is None else environ
    return "RANK" in env and "WORLD_SIZE" in env


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank, world_size, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this driver
    return env


def spawn_ranks(world_size, command, timeout=None, env=None, stdout=None, stderr=None):
    """Runs `command` (argv list) as `world_size` ranks and waits for all of them.

    Rank 0's stdout goes to `stdout` (default: this process's stdout) so that the one line it prints is the
    line of the whole job; the other ranks' stdout is forwarded to `stderr`, prefixed with the rank.  If a
    rank exits non-zero the remaining ones are terminated (by their own PIDs) and that status is returned;
    on timeout every rank is killed and 124 is returned."""
    if world_size < 1:
        raise ValueError("world_size must be positive, got %d" % world_size)
    out = sys.stdout if stdout is None else stdout
    err = sys.stderr if stderr is None else stderr
    port = free_port()
    procs = []
    for rank in range(world_size):
        procs.append(subprocess.Popen(list(command), env=rank_environment(rank, world_size, port, env),
                                      stdout=subprocess.PIPE, stderr=None if stderr is None else subprocess.PIPE,
                                      text=True, bufsize=1))

    def pump(stream, sink, prefix):
        for line in stream:
            sink.write(prefix + line)
            sink.flush()

    threads = []
    for rank, p in enumerate(procs):
        threads.append(threading.Thread(target=pump, args=(p.stdout, out if rank == 0 else err,
                                                           "" if rank == 0 else "[rank %d] " % rank), daemon=True))
        if p.stderr is not None:
            threads.append(threading.Thread(target=pump, args=(p.stderr, err, "[rank %d] " % rank), daemon=True))
    for t in threads:
        t.start()

    status = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    pending = set(range(world_size))
    while pending:
        for rank in sorted(pending):
            rc = procs[rank].poll()
            if rc is None:
                continue
            pending.discard(rank)
            if rc != 0 and status == 0:
                status = rc
        if status != 0 or (deadline is not None and time.monotonic() > deadline):
            if status == 0:
                status = 124
            for rank in pending:
                procs[rank].terminate()
            for rank in pending:
                try:
                    procs[rank].wait(timeout=10)
                except subprocess.TimeoutExpired:
                    procs[rank].kill()
                    procs[rank].wait()
            pending.clear()
            break
        if pending:
            time.sleep(0.02)
    for t in threads:
        t.join(timeout=5)
    return status
