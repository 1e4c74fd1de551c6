"""CPU, world_size 2 over gloo: the N>1 path of the decode (utterance sharding + the single
packed-weight broadcast + max-over-ranks timing).  On the GPU box the same code runs over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from shallow_wavenet_amd import config as C, dist as D
from shallow_wavenet_amd.runtime import pack_state_dict
from shallow_wavenet_amd.synth import synth_state_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    cfg = C.tiny("laplace", 2, 4)
    packed = pack_state_dict(cfg, synth_state_dict(cfg, seed=9)) if r == 0 else None
    buf = D.broadcast_packed(cfg, packed, "cpu")
    utts = [f"utt{i:02d}" for i in range(7)]
    mine = D.shard_utterances(utts, w)[r]
    tmax = D.max_over_ranks(1.0 + r, "cpu")
    tsum = D.sum_over_ranks(float(len(mine)), "cpu")
    D.barrier()
    q.put((r, float(buf.double().sum()), int(buf.numel()), mine, tmax, tsum))
    torch.distributed.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = C.tiny("laplace", 2, 4)
    ref = pack_state_dict(cfg, synth_state_dict(cfg, seed=9))
    assert res[0][2] == res[1][2] == ref.numel() == D.packed_size(cfg)
    assert res[0][1] == res[1][1] == float(ref.double().sum())          # both ranks hold rank 0's weights
    assert res[0][3] == ["utt00", "utt01", "utt02", "utt03"] and res[1][3] == ["utt04", "utt05", "utt06"]
    assert res[0][4] == res[1][4] == 2.0 and res[0][5] == res[1][5] == 7.0


def test_shard_matches_numpy_array_split():
    items = list(range(23))
    for n in (1, 2, 3, 8):
        want = [a.tolist() for a in np.array_split(items, n)]
        assert D.shard_utterances(items, n) == want


def _worker_sparse(rank, world, port, q):
    """more ranks than utterances: the spare rank joins the broadcast and the barrier with an empty shard."""
    from shallow_wavenet_amd import decode_driver as DD
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    cfg = C.tiny("laplace", 1, 0)
    packed = pack_state_dict(cfg, synth_state_dict(cfg, seed=9)) if r == 0 else None
    buf = D.broadcast_packed(cfg, packed, "cpu")
    mine = D.shard_utterances(["utt0.npy", "utt1.npy"], w)[r]
    n_batches = len(DD.plan_batches(mine, [5] * len(mine), 4))
    D.barrier()
    q.put((r, int(buf.numel()), mine, n_batches))
    torch.distributed.destroy_process_group()


def test_more_ranks_than_utterances_world3():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sparse, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[2] for r in res] == [["utt0.npy"], ["utt1.npy"], []]
    assert [r[3] for r in res] == [1, 1, 0]
