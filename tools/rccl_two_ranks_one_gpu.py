#!/usr/bin/env python3
"""Probe: can two ranks that share ONE GPU form an RCCL communicator (zigz_rccl_comm) on this stack?  RCCL normally refuses
duplicate devices; if it does here too, the > 1-rank RCCL path stays unexercised until a multi-GPU node is available.
    python tools/rccl_two_ranks_one_gpu.py
"""
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rank_main(rank, world, q_id, q_out):
    sys.path.insert(0, ROOT)
    try:
        from zigz_amd.shard import RcclComm
        if rank == 0:
            uid = RcclComm.unique_id()
            for _ in range(world - 1):
                q_id.put(uid)
        else:
            uid = q_id.get(timeout=60)
        c = RcclComm(0, uid, rank, world)
        got = c.all_gather(bytes([rank + 1]) * 16)
        red = c.all_reduce_u64([rank + 1, 10 * (rank + 1)])
        c.close()
        q_out.put((rank, "ok", [g[0] for g in got], [int(x) for x in red]))
    except Exception as e:  # noqa: BLE001
        q_out.put((rank, "error", repr(e)[:300], None))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    world = 2
    q_id, q_out = mp.Queue(), mp.Queue()
    ps = [mp.Process(target=rank_main, args=(r, world, q_id, q_out)) for r in range(world)]
    for p in ps:
        p.start()
    res = []
    for _ in ps:
        try:
            res.append(q_out.get(timeout=120))
        except Exception:
            res.append(("?", "timeout", None, None))
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    for r in sorted(res, key=str):
        print(r)
