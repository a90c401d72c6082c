"""One process per GPU, started by the parent itself.

`bench.py --gpus N` (and any other N-rank job) must be startable without torchrun: the parent -- which never touches
the GPU, so nothing that has initialised HIP is ever re-executed -- starts N children with the usual rendezvous
environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT), relays rank 0's stdout, and fails if
any child fails.  No torch import here.
"""
import os
import socket
import subprocess
import sys
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this stack
    return env


def launch_ranks(world, argv, timeout=None, poll_s=0.05):
    """start `argv` once per rank; -> (exit code, rank 0's stdout as text).  Ranks other than 0 have their stdout sent to
    stderr (the contract is ONE line on stdout, rank 0's).  The first child that fails ends the job: the others are
    terminated (by PID) and its exit code is returned; a timeout counts as failure (code 124)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    try:
        for r in range(world):
            out = subprocess.PIPE if r == 0 else sys.stderr
            procs.append(subprocess.Popen(argv, env=rank_env(r, world, port), stdout=out))
        deadline = None if timeout is None else time.monotonic() + timeout
        rc = 0
        # rank 0's pipe is drained by communicate() in a helper thread so a chatty child cannot block on a full pipe
        import threading
        captured = {}

        def drain():
            captured["out"] = procs[0].stdout.read()
        t = threading.Thread(target=drain, daemon=True)
        t.start()
        pending = set(range(world))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    print("[launch] rank %d of %d exited with code %d" % (r, world, code), file=sys.stderr, flush=True)
                    break
            if pending and rc == 0:
                if deadline is not None and time.monotonic() > deadline:
                    print("[launch] timeout: ranks %s still running" % sorted(pending), file=sys.stderr, flush=True)
                    rc = 124
                    break
                time.sleep(poll_s)
        if rc != 0:
            for r in pending:
                if procs[r].poll() is None:
                    procs[r].terminate()
            t_end = time.monotonic() + 10
            for r in pending:
                try:
                    procs[r].wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
        t.join(timeout=10)
        return rc, (captured.get("out") or b"").decode(errors="replace")
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def relay_one_json_line(text):
    """the last line of `text` that is a JSON object (rank 0 prints exactly one); None if there is none"""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return line
    return None
