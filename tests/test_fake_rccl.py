"""tests/fake_rccl/libfake_rccl.so -- the test double of librccl.so that lets the library's RCCL transport (comm.hpp kind == 1) run
with N processes on ONE GPU.  A double is only worth something if it behaves like the real thing where a schedule can go wrong,
so its own semantics are pinned here, called directly through ctypes (no petiga_amd in the way):

  * it exports every symbol comm.hpp binds (CPU test);
  * a send reads its buffer in STREAM ORDER (what the stream wrote before it is what arrives), and work enqueued after a group sees
    the received data -- with the host never waiting in between;
  * the operations of a group progress concurrently (two ranks that each receive from and send to the other in ONE group finish),
    and the same exchange split into a receive group followed by a send group HANGS and is reported (exit code 86);
  * messages of a pair match in issue order; a count mismatch is an error."""
import os
import re
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FAKE = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


def test_double_exports_what_the_library_binds():
    import ctypes
    if not os.path.exists(FAKE):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as E
        E.build_fake_rccl()
    src = open(os.path.join(ROOT, "petiga_amd", "csrc", "comm.hpp")).read()
    wanted = sorted(set(re.findall(r'sym\("(nccl\w+)"\)', src)))
    assert len(wanted) >= 9 and "ncclSend" in wanted and "ncclRecv" in wanted and "ncclCommCount" in wanted
    lib = ctypes.CDLL(FAKE, mode=ctypes.RTLD_LOCAL)
    for name in wanted:
        assert hasattr(lib, name), name
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    assert lib.ncclGetErrorString(5) == b"invalid usage"
    uid = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(uid) == 0 and uid.raw.startswith(b"/fake_rccl_")


WORKER = textwrap.dedent('''
    import ctypes as C, os, sys, time
    rank, world, uidfile, mode, fake = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    import torch
    torch.cuda.set_device(0)
    L = C.CDLL(fake, mode=C.RTLD_LOCAL)
    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    for f in (L.ncclSend, L.ncclRecv):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    uid = Uid()
    if rank == 0:
        assert L.ncclGetUniqueId(C.byref(uid)) == 0
        open(uidfile + ".tmp", "wb").write(bytes(uid)); os.rename(uidfile + ".tmp", uidfile)
    else:
        while not os.path.exists(uidfile): time.sleep(0.01)
        C.memmove(C.byref(uid), open(uidfile, "rb").read(), 128)
    comm = C.c_void_p()
    assert L.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    n = C.c_int(); assert L.ncclCommCount(comm, C.byref(n)) == 0 and n.value == world
    st = torch.cuda.Stream()
    F64 = 8
    other = (rank + 1) % world
    N = 1 << 16
    send = torch.zeros(N, dtype=torch.float64, device="cuda")
    recv = torch.full((N,), -1.0, dtype=torch.float64, device="cuda")
    out = torch.zeros(N, dtype=torch.float64, device="cuda")
    def group(ops):
        assert L.ncclGroupStart() == 0
        for kind, buf, cnt, peer in ops:
            assert (L.ncclSend if kind == "s" else L.ncclRecv)(buf.data_ptr(), cnt, F64, peer, comm, st.cuda_stream) == 0
        assert L.ncclGroupEnd() == 0
    with torch.cuda.stream(st):
        if mode == "order":
            # the stream is held back (a spinning kernel), THEN writes the payload, THEN sends: stream order decides what arrives;
            # the work after the group (out = 2 * recv) is enqueued at once and must see the data
            torch.arange(4, device="cuda").sum().item()      # (kernels loaded before the clock matters)
            torch.cuda._sleep(200_000_000)
            send.copy_(torch.arange(N, dtype=torch.float64, device="cuda") + 1000.0 * rank)
            group([("r", recv, N, other), ("s", send, N, other)])
            out.copy_(recv * 2.0)
            # a second exchange of the pair in the other sizes: matched in issue order
            tmp = send[:7] + 0.5
            group([("r", recv[:7], 7, other), ("s", tmp, 7, other)])
            assert not st.query()      # everything above was only ENQUEUED: the host was never held by the exchange
            st.synchronize()
            want = torch.arange(N, dtype=torch.float64, device="cuda") + 1000.0 * other
            assert torch.equal(out, 2.0 * want), (rank, out[:4], want[:4])
            assert torch.equal(recv[:7], want[:7] + 0.5)
        elif mode == "split":
            # the SAME exchange as two groups on one stream, receives first: each rank waits for a send that the other rank's
            # stream can only reach after ITS receive -- a deadlock with real RCCL, and the double must say so
            send.fill_(float(rank))
            group([("r", recv, N, other)])
            group([("s", send, N, other)])
            st.synchronize()
        elif mode == "mismatch":
            group([("r", recv, 8 if rank == 0 else N, other), ("s", send, N, other)])
            st.synchronize()
    torch.cuda.synchronize()
    assert L.ncclCommDestroy(comm) == 0
    print("rank", rank, "ok")
''')


def _run(mode, tmp_path, timeout_s):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FAKE_RCCL_TIMEOUT_S=str(timeout_s), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path / "uid"), mode, FAKE], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    return [p.returncode for p in procs], outs


@pytest.mark.gpu
def test_stream_order_and_concurrent_progress_inside_a_group(tmp_path):
    rcs, outs = _run("order", tmp_path, 60)
    assert rcs == [0, 0], outs


@pytest.mark.gpu
def test_receive_group_before_send_group_deadlocks_and_is_reported(tmp_path):
    rcs, outs = _run("split", tmp_path, 5)
    assert all(rc == 86 for rc in rcs), (rcs, outs)
    assert any("DEADLOCK" in o for o in outs) and all("ok" not in o for o in outs), outs
    # (a rank that leaves through the watchdog unlinks the shared-memory names it created or opened: nothing to clean here, and
    #  nobody may clean /dev/shm by prefix -- another test process may be using the double at the same time)


@pytest.mark.gpu
def test_count_mismatch_is_an_error(tmp_path):
    rcs, outs = _run("mismatch", tmp_path, 20)
    # the side that issues second sees the other's size at ncclGroupEnd (ncclInvalidUsage); the side that waits is told and leaves
    assert all(rc != 0 for rc in rcs) and 86 in rcs and any("count mismatch" in o for o in outs), (rcs, outs)
