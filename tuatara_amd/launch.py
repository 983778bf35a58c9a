"""One process per GPU, started from ONE command: the fan-out of the reference's recogniser threads (/root/reference/tuatara.cpp:461-475:
six `std::thread`s over the chunk queue, joined, outputs sorted) one level up - here the workers are whole ranks, each with its own GPU and
engine, and what is joined is the ranks' exit status plus rank 0's one JSON line.

`run_ranks(argv, world, ...)` starts `world` child processes of the same script (never exec: the parent stays, touches no GPU and owns the
deadline), gives each its rank through the environment torch.distributed.run would set (RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE /
MASTER_ADDR / MASTER_PORT), relays rank 0's stdout, and returns non-zero when any rank fails or the deadline passes - after ending the
others by PID.  `StageWatchdog` is the ranks' side of it: a rank that sits in one named stage (communicator set-up, the first gather)
longer than its allowance prints the stage and leaves with status 3, so that a hang on an 8-GPU box reads as a diagnosis, not as a timeout.

stdlib only: importable without numpy / torch / a GPU (tests/test_launch_cpu.py drives it with a stub rank body at world size 2)."""
from __future__ import annotations

import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence, Tuple

CHILD_ENV = "TUATARA_RANK_CHILD"      # set in every rank this launcher starts (a rank never launches again)


def free_ports(addr: str = "127.0.0.1", count: int = 1) -> List[int]:
    """`count` free TCP ports, all DIFFERENT: the sockets are held open together while the kernel hands the numbers out (two back-to-back probes of one
    socket each could return the same number twice - MASTER_PORT and the ranks' own rendezvous port then collided)."""
    socks = []
    try:
        for _ in range(count):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.bind((addr, 0))
            socks.append(s)
        return [s.getsockname()[1] for s in socks]
    finally:
        for s in socks:
            s.close()


def free_port(addr: str = "127.0.0.1") -> int:
    return free_ports(addr, 1)[0]


def wants_launch(gpus: int) -> bool:
    """True in the process a user (or the driver) started as `python3 bench.py --gpus N` with N > 1 and no launcher around it."""
    return gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get(CHILD_ENV)


def _end(procs: Sequence[subprocess.Popen], grace: float = 5.0) -> None:
    """terminate, then kill, exactly the processes this launcher started"""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.time() + grace
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def run_ranks(script: str, argv: Sequence[str], world: int, deadline_s: float = 1500.0, addr: str = "127.0.0.1", port: Optional[int] = None,
              extra_env: Optional[dict] = None, out=None, err=None) -> Tuple[int, Optional[str]]:
    """Start `world` ranks of `script argv...`; returns (exit status, rank 0's last JSON line or None).  Status 0 only when every rank
    left with 0 and rank 0 printed a JSON line; 124 when the deadline passed; otherwise the first failing rank's status."""
    out = out or sys.stdout
    err = err or sys.stderr
    # the ranks' own rendezvous (comm.cpp: RCCL unique id / the framed-TCP transport) listens on TUATARA_COMM_PORT; its default MASTER_PORT + 1 was
    # never probed, so another listener there (a concurrent launch) left rank 0 spinning in bind().  Both ports come from one probe that holds two
    # sockets at once, so they differ from each other; a caller-chosen TUATARA_COMM_PORT equal to MASTER_PORT is replaced
    probed = free_ports(addr, 2)
    port = port or probed[0]
    comm_port = os.environ.get("TUATARA_COMM_PORT") or str(probed[1] if probed[1] != port else probed[0])
    if int(comm_port) == int(port):
        comm_port = str(next(p for p in free_ports(addr, 3) if p != port))
    procs: List[subprocess.Popen] = []
    lines0: List[str] = []

    def pump(stream, rank):            # rank 0's stdout is collected (the JSON line), everything else goes to stderr with its rank
        for raw in iter(stream.readline, ""):
            line = raw.rstrip("\n")
            if rank == 0:
                lines0.append(line)
            else:
                print(f"[rank {rank}] {line}", file=err, flush=True)
        stream.close()

    threads = []
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": addr,
                        "MASTER_PORT": str(port), "TUATARA_COMM_PORT": comm_port, CHILD_ENV: "1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
            if extra_env:
                env.update(extra_env)
            p = subprocess.Popen([sys.executable, script, *argv], env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1)
            procs.append(p)
            t = threading.Thread(target=pump, args=(p.stdout, r), daemon=True)
            t.start()
            threads.append(t)
        t_end = time.time() + deadline_s
        status = 0
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"[launch] rank {r} left with status {c}: ending the other ranks", file=err, flush=True)
                status = c if c > 0 else 128 - c
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > t_end:
                alive = [r for r, c in enumerate(codes) if c is None]
                print(f"[launch] deadline of {deadline_s:.0f} s passed with ranks {alive} still running: ending them", file=err, flush=True)
                status = 124
                break
            time.sleep(0.05)
    finally:
        _end(procs)
    for t in threads:
        t.join(timeout=2.0)
    js = None
    for line in lines0:
        s = line.strip()
        if s.startswith("{") and s.endswith("}"):
            try:
                json.loads(s)
                js = s
            except ValueError:
                pass
        elif s:
            print(f"[rank 0] {line}", file=err, flush=True)
    if status == 0 and js is None:
        print("[launch] every rank left with status 0 but rank 0 printed no JSON line", file=err, flush=True)
        status = 1
    if js is not None and status == 0:
        print(js, file=out, flush=True)
    return status, js


class StageWatchdog:
    """`with wd.stage("communicator set-up", 120): ...` - if the block is still running after its allowance the process prints
    {"error": ..., "rank": r, "stage": name} on stderr and leaves with status 3 (os._exit: the main thread may be inside a collective
    that will never return).  One daemon thread, armed / disarmed by the context manager."""

    def __init__(self, rank: int = 0, exit_code: int = 3):
        self.rank, self.exit_code = rank, exit_code
        self._lock = threading.Lock()
        self._deadline: Optional[float] = None
        self._name = ""
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        while True:
            time.sleep(0.2)
            with self._lock:
                dl, name = self._deadline, self._name
            if dl is not None and time.time() > dl:
                print(json.dumps({"error": "stage exceeded its deadline", "rank": self.rank, "stage": name}), file=sys.stderr, flush=True)
                os._exit(self.exit_code)

    def stage(self, name: str, seconds: float):
        wd = self

        class _Ctx:
            def __enter__(self_inner):
                with wd._lock:
                    wd._name, wd._deadline = name, time.time() + seconds

            def __exit__(self_inner, *exc):
                with wd._lock:
                    wd._deadline = None
                return False

        return _Ctx()
