"""Worker for tests/test_multi_gpu_cpu.py: one rank of the row-sharded render + frame reduce, on CPU.

The product has no CPU renderer, so the per-rank shard is produced by the oracle (allowed inside
tests/); what is under test is the multi-GPU DECOMPOSITION that bench.py uses on the GPUs: rows
interleaved over ranks (row y -> rank y % world), full-frame accumulators with zeros elsewhere, one
exchange per frame -- the owned rows gathered on rank 0 (bench.gather_rows_to_root, bench.py's
default) or a reduce(sum) (bench.reduce_to_root) -- and that the result is bit-identical to a single
rank rendering the whole frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc          # noqa: E402
import bench        # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    exchange = sys.argv[2] if len(sys.argv) > 2 else "reduce"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = orc.load_golden_scene("cornell_mirror").with_resolution(96, 54)
    iters, depth = 3, 6
    cfg = orc.default_config(depth, row_offset=rank, row_stride=world)
    part, live = orc.render(scene, cfg, 1, iters, nthreads=2)
    rows = np.arange(scene.H) % world == rank
    assert not part[~rows].any()
    acc = torch.from_numpy(part.reshape(-1).copy())
    if exchange == "gather":
        bench.gather_rows_to_root(acc, scene.H, scene.W, 0)      # owned rows only (bench.py's default)
    else:
        bench.reduce_to_root(acc, 0)
    counts = torch.from_numpy(live.astype(np.int64))
    dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        full, full_live = orc.render(scene, orc.default_config(depth), 1, iters, nthreads=2)
        ok = np.array_equal(acc.numpy().reshape(full.shape), full) and np.array_equal(counts.numpy(), full_live.astype(np.int64))
        open(out, "w").write("OK" if ok else "MISMATCH")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
