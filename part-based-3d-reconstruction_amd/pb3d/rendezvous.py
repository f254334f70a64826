"""Minimal control plane for one-process-per-GPU runs on a single node (no PyTorch needed).

The data path between GPUs is RCCL (pb3d.dist.allgather); all the launcher-side runtime needs besides is a way to hand the
RCCL unique id to every rank, a barrier, a max-reduction of a timing and an abort signal.  It is a socket star: rank 0 listens
on the LOOPBACK interface at a port derived from MASTER_PORT (torchrun's own store owns MASTER_PORT itself), the other ranks
connect to it.

Wire format: length-prefixed JSON frames of a fixed shape, {"t": kind, "s": sequence number, "v": value} -- nothing is ever
unpickled.  `v` is a number, a string, a list of those, or {"hex": ...} for bytes.
Authentication: HMAC-SHA256 challenge/response on a random secret.  The secret comes from PB3D_RDV_SECRET (hex) when the
launcher distributes one; otherwise rank 0 creates it with os.urandom and publishes it in a file only this user can read
(mode 0600 in a 0700 directory), which works because every rank of a single-node run shares the filesystem and the uid.  A
peer that fails the handshake is dropped and rank 0 keeps accepting; accept() and every receive have a deadline.
Every collective carries its kind and a sequence number, so ranks that have fallen out of step fail loudly instead of mixing
values; abort(reason) makes every rank's next (or current) collective raise ControlPlaneAbort.
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import tempfile
import time

_PORT_OFFSETS = range(1, 33)
_MAX_FRAME = 1 << 20


class ControlPlaneError(RuntimeError):
    pass


class ControlPlaneAbort(ControlPlaneError):
    """Some rank called abort(); carries its reason."""


def _enc(v):
    if isinstance(v, (bytes, bytearray)):
        return {"hex": bytes(v).hex()}
    if isinstance(v, (list, tuple)):
        return [_enc(x) for x in v]
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    raise TypeError(f"pb3d rendezvous carries numbers, strings, bytes and lists of them, not {type(v).__name__}")


def _dec(v):
    if isinstance(v, dict):
        return bytes.fromhex(v["hex"])
    if isinstance(v, list):
        return [_dec(x) for x in v]
    return v


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ControlPlaneError("pb3d rendezvous: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_frame(sock, obj):
    data = json.dumps(obj, separators=(",", ":")).encode()
    if len(data) > _MAX_FRAME:
        raise ControlPlaneError("pb3d rendezvous: frame too large")
    sock.sendall(struct.pack("!I", len(data)) + data)


def _recv_frame(sock):
    (n,) = struct.unpack("!I", _recv_exact(sock, 4))
    if n > _MAX_FRAME:
        raise ControlPlaneError("pb3d rendezvous: oversized frame")
    obj = json.loads(_recv_exact(sock, n).decode())
    if not (isinstance(obj, dict) and isinstance(obj.get("t"), str)):
        raise ControlPlaneError("pb3d rendezvous: malformed frame")
    return obj


def _secret_path(base_port):
    root = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
    d = os.path.join(root, f"pb3d-rdv-{os.getuid()}")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise ControlPlaneError(f"pb3d rendezvous: {d} is not private to this user")
    return os.path.join(d, f"port-{base_port}.key")


def _publish_secret(base_port):
    env = os.environ.get("PB3D_RDV_SECRET", "")
    if env:
        return bytes.fromhex(env), None
    secret = os.urandom(32)
    path = _secret_path(base_port)
    tmp = f"{path}.{os.getpid()}"
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    with os.fdopen(fd, "wb") as f:
        f.write(secret)
    os.replace(tmp, path)                       # readers see the old key or the new one, never a torn file
    return secret, path


def _read_secret(base_port):
    env = os.environ.get("PB3D_RDV_SECRET", "")
    if env:
        return bytes.fromhex(env)
    try:
        with open(_secret_path(base_port), "rb") as f:
            s = f.read()
        return s if len(s) == 32 else None
    except OSError:
        return None


class ControlPlane:
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=180.0):
        self.rank = int(os.environ.get("RANK", 0) if rank is None else rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1) if world is None else world)
        self.timeout = float(timeout)
        base = int(os.environ.get("MASTER_PORT", 29500) if port is None else port)
        # single node by contract: the listener never binds a routable interface, whatever MASTER_ADDR says
        addr = addr or "127.0.0.1"
        self.peers = []
        self.conn = None
        self.seq = 0
        self._secret_file = None
        if self.world == 1:
            return
        deadline = time.time() + self.timeout
        if self.rank == 0:
            secret, self._secret_file = _publish_secret(base)
            listener = None
            for off in _PORT_OFFSETS:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, base + off))
                    s.listen(self.world + 8)
                    listener = s
                    break
                except OSError:
                    s.close()
            if listener is None:
                raise ControlPlaneError("pb3d rendezvous: no free control port")
            slots = {}
            try:
                while len(slots) < self.world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise ControlPlaneError(f"pb3d rendezvous: only {len(slots) + 1} of {self.world} ranks arrived within {self.timeout:.0f} s")
                    listener.settimeout(left)
                    try:
                        c, _ = listener.accept()
                    except socket.timeout:
                        continue
                    r = self._handshake_server(c, secret)
                    if r is None or r in slots or not (1 <= r < self.world):
                        c.close()               # failed or duplicate handshake: drop the peer, keep accepting
                        continue
                    c.settimeout(self.timeout)
                    slots[r] = c
            finally:
                listener.close()
            self.peers = [slots[r] for r in range(1, self.world)]
        else:
            while self.conn is None:
                secret = _read_secret(base)
                if secret is not None:
                    for off in _PORT_OFFSETS:
                        try:
                            c = socket.create_connection((addr, base + off), timeout=2.0)
                        except OSError:
                            continue
                        if self._handshake_client(c, secret):
                            c.settimeout(self.timeout)
                            self.conn = c
                            break
                        c.close()
                if self.conn is None:
                    if time.time() > deadline:
                        raise ControlPlaneError("pb3d rendezvous: rank 0 not reachable")
                    time.sleep(0.1)

    # ---- handshake: server sends a nonce, client answers HMAC(secret, nonce || rank), server confirms with HMAC(secret, answer)
    def _handshake_server(self, c, secret):
        try:
            c.settimeout(5.0)
            nonce = os.urandom(32)
            _send_frame(c, {"t": "hello", "v": nonce.hex()})
            msg = _recv_frame(c)
            if msg.get("t") != "auth" or not isinstance(msg.get("r"), int) or not isinstance(msg.get("v"), str):
                return None
            want = hmac.new(secret, nonce + struct.pack("!I", msg["r"] & 0xffffffff), hashlib.sha256).hexdigest()
            if not hmac.compare_digest(want, msg["v"]):
                return None
            _send_frame(c, {"t": "ok", "v": hmac.new(secret, bytes.fromhex(want), hashlib.sha256).hexdigest()})
            return msg["r"]
        except (OSError, ValueError, TypeError, AttributeError, ControlPlaneError):
            return None

    def _handshake_client(self, c, secret):
        try:
            c.settimeout(5.0)
            msg = _recv_frame(c)
            if msg.get("t") != "hello" or not isinstance(msg.get("v"), str):
                return False
            nonce = bytes.fromhex(msg["v"])
            mine = hmac.new(secret, nonce + struct.pack("!I", self.rank), hashlib.sha256).hexdigest()
            _send_frame(c, {"t": "auth", "r": self.rank, "v": mine})
            ok = _recv_frame(c)
            if ok.get("t") != "ok" or not isinstance(ok.get("v"), str):
                return False
            return hmac.compare_digest(ok["v"], hmac.new(secret, bytes.fromhex(mine), hashlib.sha256).hexdigest())
        except (OSError, ValueError, TypeError, AttributeError, ControlPlaneError):
            return False

    # ---- typed, sequence-numbered frames ---------------------------------------------------------------------------------
    def _send(self, sock, kind, value=None):
        _send_frame(sock, {"t": kind, "s": self.seq, "v": _enc(value)})

    def _recv(self, sock, kind):
        try:
            msg = _recv_frame(sock)
        except socket.timeout:
            raise ControlPlaneError(f"pb3d rendezvous: no '{kind}' message within {self.timeout:.0f} s") from None
        if msg["t"] == "abort":
            raise ControlPlaneAbort(str(_dec(msg.get("v"))))
        if msg["t"] != kind or msg.get("s") != self.seq:
            raise ControlPlaneError(f"pb3d rendezvous: ranks out of step (expected {kind}#{self.seq}, got {msg['t']}#{msg.get('s')})")
        return _dec(msg.get("v"))

    def _collect(self, kind, value):
        """rank 0: every peer's value for this collective, taken in arrival order (select), so that one peer's abort is
        seen -- and relayed to all -- even while another peer is still busy."""
        import select
        vals = {0: value}
        pending = {c: r for r, c in enumerate(self.peers, start=1)}
        deadline = time.time() + self.timeout
        try:
            while pending:
                left = deadline - time.time()
                if left <= 0:
                    raise ControlPlaneError(f"pb3d rendezvous: ranks {sorted(pending.values())} sent no '{kind}' within {self.timeout:.0f} s")
                ready, _, _ = select.select(list(pending), [], [], left)
                for c in ready:
                    vals[pending.pop(c)] = self._recv(c, kind)
        except ControlPlaneAbort as e:
            self._relay_abort(str(e))
            raise
        return [vals[r] for r in range(self.world)]

    def _relay_abort(self, reason):
        for c in self.peers:
            try:
                _send_frame(c, {"t": "abort", "s": self.seq, "v": reason})
            except OSError:
                pass

    def broadcast(self, obj=None):
        """rank 0's object to every rank."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            _enc(obj)                      # refuse what the wire does not carry BEFORE anything is sent
        self.seq += 1
        if self.rank == 0:
            for c in self.peers:
                self._send(c, "bc", obj)
            return obj
        return self._recv(self.conn, "bc")

    def allreduce_max(self, value):
        if self.world == 1:
            return value
        self.seq += 1
        if self.rank == 0:
            m = max(self._collect("max", value))
            for c in self.peers:
                self._send(c, "maxr", m)
            return m
        self._send(self.conn, "max", value)
        return self._recv(self.conn, "maxr")

    def gather(self, obj):
        """list of every rank's object on rank 0 (None elsewhere)."""
        if self.world == 1:
            return [obj]
        self.seq += 1
        if self.rank == 0:
            return self._collect("ga", obj)
        self._send(self.conn, "ga", obj)
        return None

    def barrier(self):
        if self.world == 1:
            return
        self.seq += 1
        if self.rank == 0:
            self._collect("bar", 0)
            for c in self.peers:
                self._send(c, "barr", 0)
        else:
            self._send(self.conn, "bar", 0)
            self._recv(self.conn, "barr")

    def abort(self, reason):
        """Tell every rank to leave: their current or next collective raises ControlPlaneAbort(reason)."""
        if self.world == 1:
            return
        try:
            if self.rank == 0:
                self._relay_abort(str(reason))
            else:
                _send_frame(self.conn, {"t": "abort", "s": self.seq, "v": str(reason)})
        except OSError:
            pass

    def close(self):
        for c in self.peers:
            try:
                c.close()
            except OSError:
                pass
        if self.conn is not None:
            try:
                self.conn.close()
            except OSError:
                pass
        if self._secret_file:
            try:
                os.unlink(self._secret_file)
            except OSError:
                pass
            self._secret_file = None
