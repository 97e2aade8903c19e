"""N>1 path on CPU: two gloo ranks each take their cost-balanced ZMW shard; the concatenated result equals
the single-process result (ZMWs are independent: no collective on the data path).  The
compute stand-in on CPU is the oracle (this is a test); on the GPU box bench.py runs the HIP path per rank."""
import os
import subprocess
import sys

import numpy as np

from ccs_amd import api, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from ccs_amd import api, shard
import oracle_lib as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
batch = api.synth(7, (3, 9), (200, 1200), seed=77)
mine = shard.shard(batch, rank, world)
res = api.Results.allocate(mine)
O.consensus_batch(api.default_model(), api.default_opts(), mine, res)
payload = [(res.sequence(z).tobytes(), float(res.rq[z]), int(res.status[z])) for z in range(mine.n_zmw)]
gathered = [None] * world
dist.all_gather_object(gathered, payload)      # host-side aggregation of results only
dist.barrier()
if rank == 0:
    pickle.dump([x for part in gathered for x in part], open(sys.argv[2], "wb"))
dist.destroy_process_group()
'''


def test_shard_bounds_balanced(built):
    b = api.synth(40, (3, 30), (300, 3000), seed=5)
    for world in (1, 2, 3, 8):
        bd = shard.shard_bounds(b, world)
        assert bd[0] == 0 and bd[-1] == 40 and (np.diff(bd) >= 0).all()
        cost = shard.zmw_cost(b)
        per = [cost[bd[i]:bd[i + 1]].sum() for i in range(world)]
        assert max(per) <= cost.sum() / world + cost.max()


def test_two_rank_gloo_matches_single_process(built, tmp_path):
    import pickle
    import oracle_lib as O
    from conftest import free_port
    out = tmp_path / "gathered.pkl"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script), ROOT, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    got = pickle.load(open(out, "rb"))
    batch = api.synth(7, (3, 9), (200, 1200), seed=77)
    ref = api.Results.allocate(batch)
    O.consensus_batch(api.default_model(), api.default_opts(), batch, ref)
    assert len(got) == 7
    for z in range(7):
        assert got[z][0] == ref.sequence(z).tobytes() and got[z][1] == float(ref.rq[z]) and got[z][2] == int(ref.status[z])


def test_bench_refuses_gpus_it_does_not_have(built):
    """`bench.py --gpus N` never reports a number for GPUs that are not there, and a launcher whose world size differs from
    --gpus is an error, not a one-GPU line labelled N (CPU box: zero devices; a GPU box with one device answers the same for --gpus 2)"""
    import torch
    ndev = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CCSX_BENCH_DEVICES")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--steps", "1", "--warmup", "1", "--zmws", "8", "--no-cpu-baseline", "--extra", ""],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "needs devices" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--zmws", "8", "--no-cpu-baseline", "--extra", ""],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
