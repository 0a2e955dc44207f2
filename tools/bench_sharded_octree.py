#!/usr/bin/env python3
"""Octree step sharded over N GPUs (ShardedOctree: tree rebuilt on every rank, force + leapfrog on the rank's bodies,
one all-gather of positions).  Not the headline bench (that is bench.py, all-pairs); same launch convention:

    python tools/bench_sharded_octree.py --gpus 1 [--bodies 1000000 --steps 50 --warmup 5 --theta 0.5 --force-dist]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_sharded_octree.py --gpus N
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--bodies", type=int, default=1000000)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--theta", type=float, default=0.5)
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL path even with one rank")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    nb = load_package()
    hs = nb.build_model(nb.F64, 3, "galaxy", args.bodies)
    sim = nb.parallel.ShardedOctree(hs, rank, world, theta=args.theta, torch_device=dev, force_exchange=use_dist)
    for _ in range(args.warmup):
        sim.step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        size, mass = sim.tree.info(sim._stream())
        print(json.dumps({"metric": "body-steps/s, 3D double octree theta=%.2f galaxy" % args.theta, "value": hs.n * args.steps / el,
                          "n_gpus": world, "steps": args.steps, "ms_per_step": el / args.steps * 1e3, "scaling": "strong",
                          "config": {"workload": "octree, n=%d" % hs.n, "split": sim.describe(), "tree_size": int(size)}}))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
