"""World-size-2 CPU test (gloo) of the N>1 path: shard by batch -> per-rank forward -> ONE
all_gather of boxes -> same result as the single-process run (SPEC.md §10).  The per-rank forward
here is the CPU oracle (allowed in tests/); on the GPU box bench.py runs the HIP detector through
the same sad_amd.dist functions over RCCL."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SAD_ROOT"])
import oracle, sad_amd
from sad_amd import config, synth
from sad_amd.dist import run_sharded, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = config.TINY
w = synth.make_weights(cfg, 0)
pts = torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points))
calls = []
def fwd(local):
    calls.append(local.shape[0])
    return torch.from_numpy(oracle.detector_forward(local.numpy(), cfg, w))
out = run_sharded(fwd, pts)
assert calls == [4 // world], calls
assert shard_range(4, rank, world) == (rank * 4 // world, (rank + 1) * 4 // world)
if rank == 0:
    np.save(os.environ["SAD_OUT"], out.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_and_all_gather_world2(tmp_path, orc, sad):
    from sad_amd import config, synth
    out = tmp_path / "boxes.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SAD_ROOT=ROOT, SAD_OUT=str(out), OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                           "29533", str(script)], env=env, timeout=600)
    got = np.load(out)
    cfg = config.TINY
    want = orc.detector_forward(synth.make_tiny_batch(0, 4, cfg.n_points), cfg, synth.make_weights(cfg, 0))
    assert got.shape == (4, cfg.n_cand, 9)
    np.testing.assert_array_equal(got, want)


def test_shard_range_errors(sad):
    import pytest
    from sad_amd.dist import shard_range
    assert shard_range(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
