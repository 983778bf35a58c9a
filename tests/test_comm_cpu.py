"""CPU suite: the host-only pieces of the C++ multi-GPU path (no GPU, no RCCL): the TCP rendezvous that hands rank 0's NCCL ids to the
other ranks (ttr_comm_create_tcp), driven here with three processes, and the library's comm symbols.  The framing logic
(ttr_gather_layout) is covered in tests/test_dist_cpu.py; the RCCL calls themselves in tests/test_gpu_dist.py (world 1) and the
driver's scaling bench (N = 2, 4, 8)."""
import ctypes as C
import multiprocessing as mp
import socket


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, q, delay):
    import time
    from tuatara_amd.build import build_lib
    from tuatara_amd.engine import load
    build_lib()
    lib = load()
    time.sleep(delay)                                   # the listener may come up after the peers start knocking
    buf = C.create_string_buffer(bytes(range(256)) if rank == 0 else b"\0" * 256, 256)
    rc = lib.ttr_dbg_tcp_share(rank, world, b"127.0.0.1", port, buf, 256)
    q.put((rank, rc, buf.raw == bytes(range(256)), lib.ttr_last_error().decode() if rc else ""))


def test_tcp_rendezvous_three_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 3, port, q, 1.0 if r == 0 else 0.0)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert got == [(0, 0, True, ""), (1, 0, True, ""), (2, 0, True, "")], got


def test_rendezvous_reports_a_dead_address():
    from tuatara_amd.build import build_lib
    from tuatara_amd.engine import load
    build_lib()
    lib = load()
    buf = C.create_string_buffer(16)
    assert lib.ttr_dbg_tcp_share(1, 2, b"no.such.host.invalid", 1, buf, 16) == -1
    assert "rendezvous" in lib.ttr_last_error().decode()
    assert lib.ttr_dbg_tcp_share(0, 1, b"127.0.0.1", 1, buf, 16) == 0          # world 1: nothing to share
    assert lib.ttr_dbg_tcp_share(5, 2, b"127.0.0.1", 1, buf, 16) == -1         # rank out of range


def _rank_cfg(rank, world, port, q, delay, timeout_s):
    import os
    import time
    os.environ["TUATARA_COMM_TIMEOUT"] = str(timeout_s)
    from tuatara_amd.build import build_lib
    from tuatara_amd.engine import load
    build_lib()
    lib = load()
    time.sleep(delay)
    buf = C.create_string_buffer(b"\x07" * 16 if rank == 0 else b"\0" * 16, 16)
    rc = lib.ttr_dbg_tcp_share(rank, world, b"127.0.0.1", port, buf, 16)
    q.put((rank, world, rc, lib.ttr_last_error().decode() if rc else ""))


def test_rendezvous_tells_a_refused_rank_why():
    """rank 0 (world 2) is joined by a rank started with world 3, then by the real rank 1, then - the meeting over - by a second rank 1: the odd ones are told
    the reason before their socket closes (they used to see a bare 'recv' failure) or find nobody listening, the good one gets the bytes.  The three joiners are
    started one after the other's verdict, so the order does not depend on how fast a process comes up."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_cfg, args=(0, 2, port, q, 0.0, 120))]
    procs[0].start()
    got = []
    procs.append(ctx.Process(target=_rank_cfg, args=(1, 3, port, q, 0.5, 60)))          # wrong world size: turned away while rank 0 keeps listening
    procs[-1].start()
    got.append(q.get(timeout=120))
    assert got[0][:2] == (1, 3) and got[0][2] != 0 and "another world size" in got[0][3], got
    procs.append(ctx.Process(target=_rank_cfg, args=(1, 2, port, q, 0.0, 60)))          # the real rank 1
    procs[-1].start()
    got += [q.get(timeout=120), q.get(timeout=120)]                                     # rank 0 and rank 1, in either order
    assert sorted(g[:3] for g in got[1:]) == [(0, 2, 0), (1, 2, 0)], got
    procs.append(ctx.Process(target=_rank_cfg, args=(1, 2, port, q, 0.0, 3)))           # a duplicate after the meeting: nobody listens any more
    procs[-1].start()
    late = q.get(timeout=120)
    assert late[:2] == (1, 2) and late[2] != 0 and "rendezvous" in late[3], late
    for p in procs:
        p.join(timeout=30)
