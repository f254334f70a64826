"""CPU multi-process tests of pb3d/rendezvous.py::ControlPlane -- the barrier / broadcast / max-reduction / abort plane that
bench.py runs on under torch.distributed.run (no PyTorch inside; no GPU needed)."""
import multiprocessing as mp
import os
import socket
import struct
import sys
import time

import pytest

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _setup():
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)


def _collectives(rank, world, port, q):
    try:
        _setup()
        from pb3d.rendezvous import ControlPlane
        cp = ControlPlane(rank, world, port=port, timeout=30.0)
        uid = cp.broadcast(bytes(range(128)) if rank == 0 else None)          # the RCCL unique id travels as bytes
        ok = uid == bytes(range(128))
        ok = ok and cp.broadcast("text" if rank == 0 else None) == "text"
        ok = ok and cp.allreduce_max(0.25 + rank) == 0.25 + world - 1
        ok = ok and cp.allreduce_max(-rank) == 0
        g = cp.gather([rank, float(rank) / 2])
        ok = ok and (g == [[r, r / 2] for r in range(world)] if rank == 0 else g is None)
        # a barrier really waits for the slowest rank
        if rank == world - 1:
            time.sleep(0.5)
        t0 = time.time()
        cp.barrier()
        waited = time.time() - t0
        ok = ok and (waited < 0.4 if rank == world - 1 else waited > 0.3)
        for _ in range(50):
            cp.barrier()
        if rank == 0:                                                           # arbitrary objects are refused at the sender, before
            try:                                                                # anything reaches the wire: the next collective still lines up
                cp.broadcast(object())
                ok = False
            except TypeError:
                pass
            cp.broadcast("resync")
        else:
            ok = ok and cp.broadcast(None) == "resync"
        cp.close()
        q.put((rank, ok))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _run(target, world, *extra, timeout=60):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(timeout=20)
    return res


@pytest.mark.timeout(120)
def test_control_plane_collectives_world3():
    res = _run(_collectives, 3)
    assert res == {0: True, 1: True, 2: True}, res


def _aborting(rank, world, port, q):
    try:
        _setup()
        from pb3d.rendezvous import ControlPlane, ControlPlaneAbort
        cp = ControlPlane(rank, world, port=port, timeout=30.0)
        cp.barrier()
        got = None
        try:
            if rank == 2:
                cp.abort("rank 2 failed in its phase")                         # a rank that raised tells the others to leave
                raise ControlPlaneAbort("rank 2 failed in its phase")
            if rank == 1:
                time.sleep(1.0)                                                 # still busy when the abort arrives at rank 0
            cp.allreduce_max(rank)
        except ControlPlaneAbort as e:
            got = str(e)
        cp.close()
        q.put((rank, got))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(120)
def test_control_plane_abort_reaches_every_rank():
    res = _run(_aborting, 3)
    assert res == {r: "rank 2 failed in its phase" for r in range(3)}, res


def _out_of_step(rank, world, port, q):
    try:
        _setup()
        from pb3d.rendezvous import ControlPlane, ControlPlaneError
        cp = ControlPlane(rank, world, port=port, timeout=5.0)
        try:
            if rank == 0:
                cp.allreduce_max(1.0)                                           # rank 1 is in a barrier: kinds differ
            else:
                cp.barrier()
            q.put((rank, "mixed silently"))
        except ControlPlaneError as e:
            q.put((rank, "out of step" in str(e) or "no '" in str(e) or "closed" in str(e)))
        cp.close()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(120)
def test_control_plane_detects_ranks_out_of_step():
    res = _run(_out_of_step, 2)
    assert res[0] is True and res[1] is True, res


def test_control_plane_rank0_unreachable_times_out():
    _setup()
    from pb3d.rendezvous import ControlPlane, ControlPlaneError
    t0 = time.time()
    with pytest.raises(ControlPlaneError, match="rank 0 not reachable"):
        ControlPlane(1, 2, port=_free_port(), timeout=1.0)
    assert time.time() - t0 < 10


def test_control_plane_missing_rank_times_out_on_rank0():
    _setup()
    from pb3d.rendezvous import ControlPlane, ControlPlaneError
    with pytest.raises(ControlPlaneError, match="only 1 of 2 ranks"):
        ControlPlane(0, 2, port=_free_port(), timeout=1.0)


def _rank0_with_intruder(rank, world, port, q):
    try:
        _setup()
        from pb3d.rendezvous import ControlPlane
        cp = ControlPlane(rank, world, port=port, timeout=30.0)
        cp.barrier()
        q.put((rank, cp.broadcast("fine" if rank == 0 else None)))
        cp.close()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.timeout(120)
def test_control_plane_survives_bad_handshakes_and_binds_loopback():
    """Connections that do not know the secret (garbage bytes, a wrong HMAC, a pickle) are dropped; rank 0 keeps accepting and
    the real rank still gets in.  The listener is on 127.0.0.1 only."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_rank0_with_intruder, args=(0, 2, port, q))
    p0.start()
    conn = None
    for _ in range(100):                                                        # find rank 0's listener
        for off in range(1, 33):
            try:
                conn = socket.create_connection(("127.0.0.1", port + off), timeout=0.5)
                break
            except OSError:
                continue
        if conn:
            break
        time.sleep(0.1)
    assert conn is not None
    peer = conn.getpeername()
    assert peer[0] == "127.0.0.1"
    # 1) garbage instead of an answer
    conn.recv(4096)
    conn.sendall(b"\x80\x04\x95\x10\x00\x00\x00cos\nsystem\n")                  # the start of a pickle: must never be interpreted
    conn.close()
    # 2) well-formed frame, wrong HMAC
    c2 = socket.create_connection(peer, timeout=2.0)
    c2.recv(4096)
    bad = b'{"t":"auth","r":1,"v":"' + b"0" * 64 + b'"}'
    c2.sendall(struct.pack("!I", len(bad)) + bad)
    time.sleep(0.2)
    c2.close()
    # 3) the real rank 1
    p1 = ctx.Process(target=_rank0_with_intruder, args=(1, 2, port, q))
    p1.start()
    res = dict(q.get(timeout=60) for _ in range(2))
    p0.join(timeout=20); p1.join(timeout=20)
    assert res == {0: "fine", 1: "fine"}, res


def test_control_plane_secret_file_is_private(tmp_path, monkeypatch):
    _setup()
    from pb3d import rendezvous as rdv
    monkeypatch.setenv("XDG_RUNTIME_DIR", str(tmp_path))
    monkeypatch.delenv("PB3D_RDV_SECRET", raising=False)
    secret, path = rdv._publish_secret(12345)
    assert len(secret) == 32 and rdv._read_secret(12345) == secret
    assert (os.stat(path).st_mode & 0o777) == 0o600 and (os.stat(os.path.dirname(path)).st_mode & 0o777) == 0o700
    monkeypatch.setenv("PB3D_RDV_SECRET", "ab" * 32)                            # a launcher-distributed secret wins, no file is written
    assert rdv._publish_secret(12346) == (bytes.fromhex("ab" * 32), None) and rdv._read_secret(12346) == bytes.fromhex("ab" * 32)
