"""GPU side of the N>1 path on ONE device (-m gpu): two ranks (gloo rendezvous, both on cuda:0) run
the HIP detector on their shard and gather the boxes through dist.AsyncBoxGather, the hook bench.py
uses; the result must equal the single-process run bit for bit.  (RCCL itself needs one GPU per
rank — that run is the driver's.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SAD_ROOT"])
import sad_amd
from sad_amd import config, synth
from sad_amd.detector import SADDetector
from sad_amd.dist import AsyncBoxGather, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
cfg = config.TINY
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
pts = torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points)).to(dev)
lo, hi = shard_range(4, rank, world)
gather = AsyncBoxGather(dev)
outs = []
consumer = torch.cuda.Stream(device=dev)
copies = []
for step in range(3):                      # several steps in flight, as in bench.py
    out, ev = det.submit(pts[lo:hi].contiguous(), post=gather)
    assert gather.event is not None and ev is gather.event    # the returned event covers the collective
    # a consumer that follows the submit() docstring: waits on ev ONLY, then reads `out` elsewhere
    with torch.cuda.stream(consumer):
        consumer.wait_event(ev)
        out.record_stream(consumer)
        copies.append(out.clone())
    outs.append(out)
consumer.synchronize()
assert copies[-1].shape[0] == 4
gather.wait()
torch.cuda.synchronize()
assert all(torch.equal(o, outs[0]) for o in outs)
assert all(torch.equal(c, outs[0]) for c in copies), "out read after ev.wait() differs: event does not cover the gather"
if rank == 0:
    np.save(os.environ["SAD_OUT"], outs[-1].cpu().numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_one_device_async_gather(tmp_path, sad, dev):
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    out = tmp_path / "boxes.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SAD_ROOT=ROOT, SAD_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)], env=env, timeout=600)
    cfg = config.TINY
    det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
    want = det(torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points)).to(dev)).cpu().numpy()
    got = np.load(out)
    assert got.shape == (4, cfg.n_cand, 9)
    np.testing.assert_array_equal(got, want)
