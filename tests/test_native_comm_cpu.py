"""CPU: the bookkeeping around the library's RCCL communicator (torchlsq.distributed.native_comm) that needs no GPU -- what a
look-up may and may not decide, and that a cached communicator is re-validated against the world it is asked for."""
import os
import socket

import pytest
import torch
import torch.distributed as dist


@pytest.fixture()
def world_of_one():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
    from torchlsq import distributed as D
    D._COMMS.clear()
    yield D
    D._COMMS.clear()
    dist.destroy_process_group()


def test_a_lookup_never_decides_for_later_calls(world_of_one, monkeypatch):
    """join() -- a look-up, create=False -- before the first sharded backward must not switch the native route off for the rest
    of the process (round-5 advisor): it caches nothing, and the next creating call still gets as far as asking for the backend"""
    D = world_of_one
    dev = torch.device("cuda", 0)
    asked = []
    monkeypatch.setattr(dist, "get_backend", lambda group=None: asked.append(1) or "gloo")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    D.join(None, dev)
    assert D.native_comm(None, dev, create=False) is None
    assert not D._COMMS and not asked, "a look-up cached a decision"
    assert D.native_comm(None, dev) is None and asked == [1]           # the creating call decides (gloo: torch.distributed) ...
    assert list(D._COMMS) == [(0, 0)] and D._COMMS[(0, 0)][0] is None   # ... and only that is cached, tied to this process group
    assert D.native_comm(None, dev) is None and asked == [1]           # (a hit: nothing is asked again)


def test_a_cached_communicator_of_another_world_is_dropped(world_of_one, monkeypatch):
    """dist.destroy_process_group() + a new init (or a recycled id(group)) must not be handed the old world's communicator"""
    D = world_of_one
    dev = torch.device("cuda", 0)

    class Comm:
        handle = 1
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    D._COMMS[(0, 0)] = (Comm(), object())                  # a communicator of a process group that is gone
    assert D.native_comm(None, dev, create=False) is None and (0, 0) not in D._COMMS
    mine = Comm()
    D._COMMS[(0, 0)] = (mine, D._process_group(None))
    assert D.native_comm(None, dev, create=False) is mine
    mine.handle = None                      # destroyed behind the cache's back
    assert D.native_comm(None, dev, create=False) is None and (0, 0) not in D._COMMS


def test_transposition_detection_is_pure_stride_logic():
    """torchlsq._hip_host._transposition (what decides whether a grad in another dense order goes through lsq_hip_relayout): two
    dense orders that differ by ONE swap of adjacent dimension groups give (A, B, C) with g's memory [A][B][C] and x's [A][C][B];
    anything else -- the same order, size-1 dims that make two formats one memory, a non-dense view, a double permutation -- None"""
    from torchlsq import extension as E
    g = torch.empty(6, 96, 7, 5)
    x = torch.empty(6, 96, 7, 5).contiguous(memory_format=torch.channels_last)
    assert E._transposition(g, x) == (6, 96, 35) and E._transposition(x, g) == (6, 35, 96)
    assert E._transposition(g, g.clone()) is None
    one = torch.empty(4, 8, 1, 1)
    assert E._transposition(one, one.contiguous(memory_format=torch.channels_last)) is None        # H*W == 1: one and the same memory
    btf = torch.empty(3, 50, 20)
    bft = torch.empty(3, 20, 50).permute(0, 2, 1)                                                   # shape [3,50,20], memory [B][F][T]
    assert E._transposition(btf, bft) == (3, 50, 20)
    assert E._transposition(btf[:, ::2], bft[:, ::2]) is None                                       # not dense
    d5 = torch.empty(2, 3, 4, 5, 6)
    assert E._transposition(d5, d5.permute(0, 3, 4, 1, 2).contiguous().permute(0, 3, 4, 1, 2)) == (2, 12, 30)   # [A][(1,2)][(3,4)] -> [A][(3,4)][(1,2)]
    assert E._transposition(d5, d5.permute(0, 4, 3, 2, 1).contiguous().permute(0, 4, 3, 2, 1)) is None          # a full reversal is no single swap
    # whole-tensor transposition (no common prefix): A == 1
    m = torch.empty(33, 129)
    assert E._transposition(m, torch.empty(129, 33).t()) == (1, 33, 129)
