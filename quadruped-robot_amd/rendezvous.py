"""Launcher plumbing of the multi-GPU bench without any GPU array library: rendezvous, barrier, max-reduce and byte broadcast / all-gather of
the ranks of ONE node over a localhost TCP socket (one process per GPU; the only collective of the path itself is the RCCL
all-gather behind the C ABI -- this module only hands rank 0's 128-byte communicator id around and keeps the ranks' clocks honest).

Ranks find each other through the environment every launcher sets (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  MASTER_PORT itself is
usually taken (torch.distributed.run's agent keeps its own store there), so rank 0 listens on an ephemeral port and publishes it in a file
under the temporary directory, named after MASTER_ADDR / MASTER_PORT and written atomically; the other ranks poll for the file, connect and
say who they are.  A stale file of an earlier run (same port, rank 0 died before it could remove it) names a port nobody listens on -- or
somebody else's -- and the hello fails: the rank goes back to polling until the deadline.  Every operation has a deadline: a rank that dies
takes the others out of their waits with an exception instead of hanging the node."""
import os
import pickle
import socket
import struct
import tempfile
import time


class RendezvousError(RuntimeError):
    pass


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        hdr += chunk
    (n,) = struct.unpack("<Q", hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise RendezvousError("peer closed the connection")
        buf += chunk
    return pickle.loads(bytes(buf))


def port_file(addr=None, port=None):
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = port or os.environ.get("MASTER_PORT", "29500")
    return os.path.join(tempfile.gettempdir(), "qrgpu_rdv_%s_%s.port" % (addr.replace(":", "_").replace("/", "_"), port))


class Group:
    """One rank's end of the group.  Every collective is "send my part to rank 0, get everybody's parts back": world sizes are a handful of
    ranks of one node, the payloads a few hundred bytes (48 KB per rank in the one-GPU rehearsal of the gather)."""

    MAGIC = "qrgpu-rendezvous-1"

    def __init__(self, rank=None, world=None, timeout=120.0, op_timeout=600.0):
        self.rank = int(os.environ["RANK"]) if rank is None else int(rank)
        self.world = int(os.environ["WORLD_SIZE"]) if world is None else int(world)
        self.op_timeout = float(op_timeout)
        self._seq = 0
        self._file = port_file()
        self._server = None
        self._peers = {}
        self._sock = None
        if self.world == 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            self._serve(deadline)
        else:
            self._connect(deadline)

    # ---- rank 0 -----------------------------------------------------------------------------------------------------------
    def _serve(self, deadline):
        try:
            os.unlink(self._file)                          # (a stale file of an earlier run with this port)
        except OSError:
            pass
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind(("127.0.0.1", 0))
        srv.listen(self.world + 4)
        self._server = srv
        nonce = "%d-%f" % (os.getpid(), time.time())
        tmp = self._file + ".%d.tmp" % os.getpid()
        with open(tmp, "w") as f:
            f.write("%d %s\n" % (srv.getsockname()[1], nonce))
        os.replace(tmp, self._file)
        while len(self._peers) < self.world - 1:
            srv.settimeout(max(0.05, deadline - time.time()))
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                raise RendezvousError("rank 0: %d of %d ranks arrived within the deadline" % (len(self._peers) + 1, self.world))
            conn.settimeout(10.0)
            try:
                hello = _recv(conn)
                ok = isinstance(hello, dict) and hello.get("magic") == self.MAGIC and hello.get("world") == self.world and hello.get("nonce") == nonce \
                    and isinstance(hello.get("rank"), int) and 0 < hello["rank"] < self.world and hello["rank"] not in self._peers
                _send(conn, {"ok": bool(ok)})
            except Exception:
                conn.close()
                continue
            if not ok:
                conn.close()
                continue
            conn.settimeout(None)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._peers[hello["rank"]] = conn
        try:
            os.unlink(self._file)                          # everybody is in: the file has done its job
        except OSError:
            pass

    # ---- the other ranks ----------------------------------------------------------------------------------------------------
    def _connect(self, deadline):
        last = "no port file %s" % self._file
        while time.time() < deadline:
            try:
                with open(self._file) as f:
                    port_s, nonce = f.read().split()
                s = socket.create_connection(("127.0.0.1", int(port_s)), timeout=2.0)
                s.settimeout(10.0)
                _send(s, {"magic": self.MAGIC, "rank": self.rank, "world": self.world, "nonce": nonce})
                if _recv(s).get("ok"):
                    s.settimeout(None)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s
                    return
                s.close()
                last = "rank 0 refused the hello (another run's file?)"
            except (OSError, ValueError, RendezvousError, pickle.UnpicklingError, EOFError) as e:
                last = repr(e)
            time.sleep(0.05)
        raise RendezvousError("rank %d: no rendezvous with rank 0 within the deadline (%s)" % (self.rank, last))

    # ---- collectives ----------------------------------------------------------------------------------------------------------
    def _exchange(self, part):
        """-> list of every rank's part, in rank order."""
        if self.world == 1:
            return [part]
        self._seq += 1
        if self.rank == 0:
            parts = {0: part}
            for r, conn in self._peers.items():
                conn.settimeout(self.op_timeout)
                try:
                    seq, p = _recv(conn)
                except (socket.timeout, OSError) as e:
                    raise RendezvousError("rank 0: rank %d did not reach collective %d (%r)" % (r, self._seq, e))
                if seq != self._seq:
                    raise RendezvousError("rank 0: rank %d is at collective %d, rank 0 at %d" % (r, seq, self._seq))
                parts[r] = p
            out = [parts[r] for r in range(self.world)]
            for conn in self._peers.values():
                _send(conn, out)
            return out
        self._sock.settimeout(self.op_timeout)
        try:
            _send(self._sock, (self._seq, part))
            return _recv(self._sock)
        except (socket.timeout, OSError) as e:
            raise RendezvousError("rank %d: collective %d did not complete (%r)" % (self.rank, self._seq, e))

    def barrier(self):
        self._exchange(None)

    def allreduce_max(self, value):
        return max(float(v) for v in self._exchange(float(value)))

    def broadcast_bytes(self, blob, src=0):
        parts = self._exchange(bytes(blob) if self.rank == src else None)
        return parts[src]

    def allgather_bytes(self, blob):
        return self._exchange(bytes(blob))

    def close(self):
        for conn in self._peers.values():
            try:
                conn.close()
            except OSError:
                pass
        self._peers = {}
        for s in (self._sock, self._server):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._sock = self._server = None


def exchange_comm_id(group, make_id):
    """Rank 0 calls make_id() (qrgpu.comm_unique_id: the 128-byte ncclUniqueId blob) and every rank returns that blob."""
    blob = group.broadcast_bytes(make_id() if group.rank == 0 else b"", src=0)
    if len(blob) != 128:
        raise ValueError("communicator id must be 128 bytes, got %d" % len(blob))
    return blob
