"""Ghost-row reduction between ranks (one process per GPU).

The product path is inside the library: init_comm(iga) + iga.reduce_ghost_rows(A, b) / iga.refresh_ghosts(v)
(include/petiga_amd.h: IGXCommInitRCCL, IGXReduceGhostRows, IGXRefreshGhosts).  GhostExchange / GhostRefresh below drive the
same pack / unpack kernels from Python with torch.distributed as the transport; the tests use them as an independent check.

Each rank packs the rows of nodes it holds but does not own, one message per upper neighbour
(IGXPackGhostRows), the messages travel point-to-point (RCCL over xGMI: every neighbour pair of a
2x2x2 grid has its own link, so the <=7 messages of a rank move concurrently), and the owner adds them to
its rows (IGXUnpackGhostRows).  This is the only collective step of the assembly path; it replaces the
PETSc stash traffic of MatAssemblyBegin/End + VecAssemblyBegin/End (src/petigaksp.c:197-200).
"""
import torch
import torch.distributed as dist


def plan(iga, with_mat=True, with_vec=True):
    """[(peer, doubles)] for the send list and for the receive list."""
    size = lambda m, v: (m if with_mat else 0) + (v if with_vec else 0)
    return ([(r, size(m, v)) for r, m, v in iga.neighbors(True)], [(r, size(m, v)) for r, m, v in iga.neighbors(False)])


def p2p_exchange(send_bufs, send_peers, recv_bufs, recv_peers):
    """One grouped batch of isend/irecv (ncclGroupStart/End under RCCL); a peer gets at most one message."""
    if not send_bufs and not recv_bufs:
        return
    staged = dist.get_backend() == "gloo" and any(b.is_cuda for b in list(send_bufs) + list(recv_bufs))
    if staged:   # test transport (no RCCL): gloo moves host memory only, so stage through the host
        dev_recv = recv_bufs
        send_bufs = [b.cpu() for b in send_bufs]
        recv_bufs = [torch.empty(b.shape, dtype=b.dtype) for b in dev_recv]
    ops = [dist.P2POp(dist.irecv, b, p) for b, p in zip(recv_bufs, recv_peers)]
    ops += [dist.P2POp(dist.isend, b, p) for b, p in zip(send_bufs, send_peers)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if staged:
        for d, h in zip(dev_recv, recv_bufs):
            d.copy_(h)


def _device_view(ptr, count):
    """float64 torch view of `count` doubles of device memory at `ptr` (no copy)."""
    class _A:
        pass
    a = _A()
    a.__cuda_array_interface__ = dict(shape=(int(count),), typestr="<f8", data=(int(ptr), False), version=3)
    return torch.as_tensor(a, device="cuda")


def init_comm(iga, transport=None):
    """Binds the library's own exchange (IGXReduceGhostRows / IGXRefreshGhosts) for this rank of the default process
    group.  transport "rccl": the library's grouped ncclSend / ncclRecv (the product path; the unique id is broadcast
    through torch.distributed).  transport "host": torch.distributed moves the packed device buffers (test transport:
    gloo stages through the host, so ranks may share one GPU)."""
    transport = transport or ("rccl" if dist.get_backend() == "nccl" else "host")
    if transport == "rccl":
        # Every rank must end up on the same transport, and ncclCommInitRank is a blocking collective: a rank that cannot bind
        # librccl.so must be known BEFORE anybody enters it (the others would wait in it for ever).  So: (1) every rank binds the
        # library only (IGXCommGetUniqueId into a scratch id: dlopen + one local call) and the outcomes are gathered;
        # (2) only when all of them succeeded does rank 0's id travel and the communicator get created; (3) the outcome of
        # that is gathered once more.
        err, uid = None, None
        try:
            uid = iga.comm_unique_id()
        except Exception as e:          # librccl.so could not be bound on this rank
            err = e
        flags = [None] * dist.get_world_size()
        dist.all_gather_object(flags, uid is not None)
        ok = all(flags)
        if ok:
            ids = [uid if dist.get_rank() == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                iga.comm_init_rccl(ids[0])
            except Exception as e:
                ok, err = False, e
            flags = [None] * dist.get_world_size()
            dist.all_gather_object(flags, bool(ok))
        if not all(flags):
            import sys
            if dist.get_rank() == 0:
                print("petiga_amd.exchange: the library's RCCL binding failed on rank(s) %s (%s); falling back to torch.distributed point-to-point"
                      % ([i for i, f in enumerate(flags) if not f], err), file=sys.stderr)
            if ok:
                iga.comm_destroy()
            transport = "host"
    if transport != "rccl":
        def move(send, recv):
            p2p_exchange([_device_view(p, n) for _, p, n in send], [r for r, _, _ in send],
                         [_device_view(p, n) for _, p, n in recv], [r for r, _, _ in recv])
            torch.cuda.synchronize()
        iga.comm_init_transport(move)
    return transport


class GhostExchange:
    """Pre-allocated device buffers + the reduce step for one (matrix, vector) pair."""

    def __init__(self, iga, A, b, device="cuda"):
        self.iga, self.A, self.b = iga, A, b
        s, r = plan(iga, A is not None, b is not None)
        self.send_peers = [p for p, _ in s]
        self.recv_peers = [p for p, _ in r]
        self.send_bufs = [torch.empty(max(n, 1), dtype=torch.float64, device=device) for _, n in s]
        self.recv_bufs = [torch.empty(max(n, 1), dtype=torch.float64, device=device) for _, n in r]
        self.bytes_sent = 8 * sum(n for _, n in s)

    def reduce(self):
        for k, buf in enumerate(self.send_bufs):
            self.iga.pack_ghost_rows(self.A, self.b, k, buf.data_ptr())
        self.iga.synchronize()          # packs ran on the engine's stream; the transport uses torch's
        p2p_exchange(self.send_bufs, self.send_peers, self.recv_bufs, self.recv_peers)
        torch.cuda.synchronize()
        for k, buf in enumerate(self.recv_bufs):
            self.iga.unpack_ghost_rows(self.A, self.b, k, buf.data_ptr())


class GhostRefresh:
    """Owner -> ghost copies of a state vector before a nonlinear assembly (the reverse of GhostExchange.reduce):
    replaces DMGlobalToLocal of IGAGetLocalVecArray (src/petigavec.c:256-269)."""

    def __init__(self, iga, device="cuda"):
        self.iga = iga
        s, r = plan(iga, with_mat=False, with_vec=True)
        # messages flow against the reduction: I send to my lower neighbours (receive list), receive from the upper ones
        self.send_peers = [p for p, _ in r]
        self.recv_peers = [p for p, _ in s]
        self.send_bufs = [torch.empty(max(n, 1), dtype=torch.float64, device=device) for _, n in r]
        self.recv_bufs = [torch.empty(max(n, 1), dtype=torch.float64, device=device) for _, n in s]

    def refresh(self, vec):
        for k, buf in enumerate(self.send_bufs):
            self.iga.pack_owner_values(vec, k, buf.data_ptr())
        self.iga.synchronize()
        p2p_exchange(self.send_bufs, self.send_peers, self.recv_bufs, self.recv_peers)
        torch.cuda.synchronize()
        for k, buf in enumerate(self.recv_bufs):
            self.iga.unpack_ghost_values(vec, k, buf.data_ptr())
