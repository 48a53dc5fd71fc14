"""Host-side rendezvous of the one-process-per-GPU ranks of ONE node: a tiny TCP all-gather.

Why not torch.distributed: the product library links the system ROCm runtime; importing torch
would map a second HIP runtime into every rank only to broadcast the 128-byte RCCL unique id and
to run a few barriers.  Everything the ranks need from each other on the host side -- broadcast,
barrier, max over ranks -- is an all-gather of a small JSON value, so that is all this is.  Every
halo byte moves through RCCL inside the C++ library; nothing here is on the data path.

Rank 0 listens on an ephemeral port of 127.0.0.1 and publishes it in a file that every rank can
name without talking to anybody: ``$TMPDIR/sfl_rdzv_<key>`` with ``key`` = ``SFL_RDZV_KEY`` from
the environment (set by bench.py's own launcher) or ``<parent pid>_<MASTER_PORT>`` (under
``python -m torch.distributed.run`` every worker has the same parent, the elastic agent, and the
agent itself occupies MASTER_PORT).  A stale file of a crashed earlier run points at a dead port
or answers with another nonce; clients simply keep polling until the handshake succeeds.
"""
from __future__ import annotations

import base64
import json
import os
import socket
import struct
import tempfile
import time


def _send(sock: socket.socket, obj) -> None:
    raw = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(raw)) + raw)


def _recv(sock: socket.socket):
    def exactly(n):
        buf = b""
        while len(buf) < n:
            part = sock.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += part
        return buf
    (n,) = struct.unpack("<I", exactly(4))
    return json.loads(exactly(n).decode())


def rendezvous_file(key: str | None = None) -> str:
    """bench.py's launcher hands its ranks a private directory (SFL_RDZV_DIR, mode 0700, removed when it exits);
    under other launchers the file lives in the temp directory under a per-user name and is created exclusively."""
    key = key or os.environ.get("SFL_RDZV_KEY") or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
    private = os.environ.get("SFL_RDZV_DIR")
    if private and os.path.isdir(private):
        return os.path.join(private, f"sfl_rdzv_{key}")
    return os.path.join(tempfile.gettempdir(), f"sfl_rdzv_{os.getuid()}_{key}")


class Rendezvous:
    """all_gather / broadcast / barrier / max over the ranks of one node (world == 1: no sockets)."""

    def __init__(self, rank: int, world: int, key: str | None = None, timeout_s: float = 300.0):
        self.rank, self.world = rank, world
        self._peers: list[socket.socket] = []   # rank 0: sockets of ranks 1 .. world-1 (by rank)
        self._root: socket.socket | None = None
        self._path = None
        if world == 1:
            return
        self._path = rendezvous_file(key)
        deadline = time.monotonic() + timeout_s
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            nonce = base64.b16encode(os.urandom(8)).decode()
            tmp = f"{self._path}.{os.getpid()}.{nonce}"
            # O_EXCL | O_NOFOLLOW: never written through a link somebody else planted under a guessable name
            fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW | os.O_WRONLY, 0o600)
            with os.fdopen(fd, "w") as f:
                json.dump({"port": srv.getsockname()[1], "nonce": nonce, "world": world}, f)
            os.replace(tmp, self._path)     # atomic: readers see the old file or the new one
            by_rank: dict[int, socket.socket] = {}
            srv.settimeout(1.0)
            while len(by_rank) < world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: only {len(by_rank) + 1} of {world} ranks arrived")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(timeout_s)
                try:
                    hello = _recv(conn)
                except (ConnectionError, OSError, ValueError):
                    conn.close()
                    continue
                ok = hello.get("nonce") == nonce and 0 < hello.get("rank", -1) < world \
                    and hello["rank"] not in by_rank
                _send(conn, {"ok": ok})
                if ok:
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    by_rank[hello["rank"]] = conn
                else:
                    conn.close()
            srv.close()
            self._peers = [by_rank[r] for r in range(1, world)]
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {rank} found no rank 0 behind {self._path}")
                try:
                    with open(self._path) as f:
                        info = json.load(f)
                    if info.get("world") != world:
                        raise ValueError("stale rendezvous file")
                    s = socket.create_connection(("127.0.0.1", info["port"]), timeout=5.0)
                    s.settimeout(timeout_s)
                    _send(s, {"rank": rank, "nonce": info["nonce"]})
                    if _recv(s).get("ok"):
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        self._root = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass
                time.sleep(0.05)

    def all_gather(self, value):
        """Every rank contributes a JSON-serialisable value; every rank receives the list by rank."""
        if self.world == 1:
            return [value]
        if self.rank == 0:
            values = [value] + [_recv(p) for p in self._peers]
            for p in self._peers:
                _send(p, values)
            return values
        _send(self._root, value)
        return _recv(self._root)

    def barrier(self) -> None:
        self.all_gather(None)

    def broadcast_bytes(self, data: bytes | None) -> bytes:
        """Rank 0's bytes on every rank."""
        got = self.all_gather(base64.b64encode(data).decode() if self.rank == 0 else None)
        return base64.b64decode(got[0])

    def max(self, values):
        """Element-wise maximum of a list of numbers over all ranks."""
        rows = self.all_gather(list(values))
        return [max(col) for col in zip(*rows)]

    def close(self) -> None:
        for s in self._peers + ([self._root] if self._root else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._root = [], None
        if self.rank == 0 and self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
