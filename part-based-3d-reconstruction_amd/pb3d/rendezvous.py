"""Minimal control plane for one-process-per-GPU runs on a single node (no PyTorch needed).

The data path between GPUs is RCCL (pb3d.dist.allgather); all the launcher-side runtime needs
besides is a way to hand the RCCL unique id to every rank, a barrier and a max-reduction of a
timing.  Rank 0 listens on MASTER_ADDR at a port derived from MASTER_PORT (torchrun's own store
owns MASTER_PORT itself); the other ranks connect with an authenticated handshake.
"""
import os
import time
from multiprocessing.connection import Client, Listener

_PORT_OFFSETS = range(1, 33)


class ControlPlane:
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=180.0):
        self.rank = int(os.environ.get("RANK", 0) if rank is None else rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1) if world is None else world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", 29500) if port is None else port)
        key = f"pb3d-{base}-{self.world}".encode()
        self.peers = []
        self.conn = None
        if self.world == 1:
            return
        if self.rank == 0:
            listener = None
            for off in _PORT_OFFSETS:
                try:
                    listener = Listener((addr, base + off), authkey=key)
                    break
                except OSError:
                    continue
            if listener is None:
                raise RuntimeError("pb3d rendezvous: no free control port")
            slots = {}
            while len(slots) < self.world - 1:
                c = listener.accept()
                slots[c.recv()] = c
            self.peers = [slots[r] for r in range(1, self.world)]
            listener.close()
        else:
            deadline = time.time() + timeout
            while self.conn is None:
                for off in _PORT_OFFSETS:
                    try:
                        self.conn = Client((addr, base + off), authkey=key)
                        break
                    except Exception:
                        continue
                if self.conn is None:
                    if time.time() > deadline:
                        raise RuntimeError("pb3d rendezvous: rank 0 not reachable")
                    time.sleep(0.2)
            self.conn.send(self.rank)

    def broadcast(self, obj=None):
        """rank 0's object to every rank."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            for c in self.peers:
                c.send(obj)
            return obj
        return self.conn.recv()

    def allreduce_max(self, value):
        if self.world == 1:
            return value
        if self.rank == 0:
            m = max([value] + [c.recv() for c in self.peers])
            for c in self.peers:
                c.send(m)
            return m
        self.conn.send(value)
        return self.conn.recv()

    def gather(self, obj):
        """list of every rank's object on rank 0 (None elsewhere)."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [c.recv() for c in self.peers]
        self.conn.send(obj)
        return None

    def barrier(self):
        self.allreduce_max(0)

    def close(self):
        for c in self.peers:
            c.close()
        if self.conn is not None:
            self.conn.close()
