"""Parent that spawns one rank process per GPU -- the multi-GPU counterpart of the reference's parent process that
creates its worker pool (FunscriptFlow.pyw:1190-1191, Pool(processes=threads)).

    spawn_ranks(script, argv, n)   start n fresh children of `script` with the torch.distributed environment
                                   (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT),
                                   relay rank 0's stdout to this process's stdout (the one JSON line of bench.py),
                                   send every other rank's stdout to stderr, return the first non-zero exit status
                                   (the surviving ranks are terminated: never a retry, never a re-exec).

The parent never imports torch and never touches HIP, so it does not count against a box's limit of processes that may
have the card open, and forking / spawning from it is safe.  Nothing here is GPU code; `bench.py --gpus N` (N > 1,
started from a plain shell) is its caller.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    """A TCP port that is free on 127.0.0.1 right now (the rendezvous store of rank 0 binds it a moment later)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("OMP_NUM_THREADS", "2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the rank processes here
    return env


def _status(rc):
    return rc if rc >= 0 else 128 - rc  # killed by signal s -> 128 + s, like a shell


def spawn_ranks(script, argv, n, out=None, err=None, grace_s=10.0, poll_s=0.05):
    """Run `python script *argv` as n rank processes; returns the exit status to leave with (0 = every rank succeeded)."""
    out = out or sys.stdout
    err = err or sys.stderr
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=rank_env(r, n, port),
                                      stdout=subprocess.PIPE, stderr=None, bufsize=0))

    def relay(p, dst):
        for line in iter(p.stdout.readline, b""):
            if dst is None:
                continue                      # our own reader went away: keep draining so that the rank never blocks on its pipe
            try:
                dst.write(line.decode("utf-8", "replace"))
                dst.flush()
            except (BrokenPipeError, ValueError):
                dst = None

    threads = [threading.Thread(target=relay, args=(p, out if r == 0 else err), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except ProcessLookupError:
                    pass

    def on_signal(signum, _frame):
        stop_all(signal.SIGTERM)
        raise SystemExit(128 + signum)

    old = {}
    if threading.current_thread() is threading.main_thread():
        for s in (signal.SIGTERM, signal.SIGINT):
            old[s] = signal.signal(s, on_signal)
    rc = 0
    try:
        live = set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = _status(code)
                    err.write(f"launch: rank {r}/{n} exited with status {rc}; stopping the other ranks\n")
                    err.flush()
                    break
            if live and rc == 0:
                time.sleep(poll_s)
        if rc != 0:
            stop_all(signal.SIGTERM)
            t_end = time.monotonic() + grace_s
            for p in procs:
                try:
                    p.wait(timeout=max(0.0, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
    finally:
        stop_all(signal.SIGKILL)
        for s, h in old.items():
            signal.signal(s, h)
    for t in threads:
        t.join(timeout=5)
    return rc
