"""N > 1 path on CPU: two processes (gloo), each renders its interleaved pixel tiles with the product's shard map and
per-slot functions (host simulation), then one reduce(sum) of the float3 framebuffer -- the same steps bench.py performs
with RCCL.  The reduced image must be bit-identical to the single-process oracle (one owner per pixel, sum with zeros)."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out):
    import torch
    sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    art = ge.load_package()
    import conv
    import hostsim
    import orc
    cs = orc.CornellScene()
    sd = conv.desc_from_oracle(art, cs)
    W, H = 80, 48
    hostsim.set_shard(art, rank, world, 16)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=5)
    acc, rays = hostsim.render(art, sd, p, W, H)
    owned = int(np.count_nonzero(acc.any(-1)))
    t = torch.from_numpy(acc.reshape(-1))
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    r = torch.tensor([float(rays)], dtype=torch.float64)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    if rank == 0:
        ref, _, cnt = orc.render(cs.scene, orc.make_params(W, H, orc.PT_MIS, True, 8, 1, seed=5))
        ok = np.array_equal(t.numpy().view(np.uint32), ref.reshape(-1).view(np.uint32)) and int(r.item()) == cnt.rays
        open(out, "w").write("OK %d" % owned if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_shards_reduce_to_the_oracle_image(tmp_path):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, 29517 + os.getpid() % 1000, out), nprocs=2, join=True)
    res = open(out).read()
    assert res.startswith("OK"), res
    assert 0 < int(res.split()[1]) < 80 * 48          # rank 0 owned only part of the frame
