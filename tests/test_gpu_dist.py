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


RCCL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SAD_ROOT"])
import sad_amd
from sad_amd import config, synth
from sad_amd.detector import SADDetector
from sad_amd.dist import AsyncBoxGather, all_gather_boxes
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)          # "nccl" is RCCL on ROCm: a one-rank communicator
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
cfg = config.TINY
det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
pts = torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points)).to(dev)
gather = AsyncBoxGather(dev)
outs, locals_ = [], []
for step in range(4):
    out, ev = det.submit(pts, post=gather)
    assert gather.event is not None and ev is gather.event, "the forced collective did not run"
    outs.append(out)
ev.synchronize()
gather.wait()
torch.cuda.synchronize()
want = det(pts)
torch.cuda.synchronize()
assert all(o.data_ptr() != want.data_ptr() for o in outs)
assert all(torch.equal(o, want) for o in outs), "boxes changed on their way through the RCCL all_gather"
assert torch.equal(all_gather_boxes(want), want)
t = torch.tensor([3.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # the reductions bench.py uses
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 3.0
np.save(os.environ["SAD_OUT"], outs[-1].cpu().numpy())
dist.destroy_process_group()
'''


def test_one_rank_rccl_collective(tmp_path, sad, dev):
    """The step's collective on the REAL backend: a one-rank RCCL communicator on cuda:0 runs ``all_gather_into_tensor`` on
    the communication stream behind every submit (``SAD_DIST_FORCE_COLLECTIVE``: a one-rank group otherwise skips it).  What
    a one-GPU box can show of the N > 1 path: the backend loads and initialises on this hardware, the stream hand-off of
    ``AsyncBoxGather`` is right for RCCL (not only for gloo), and the boxes come out unchanged."""
    import torch
    from sad_amd import config, synth
    from sad_amd.detector import SADDetector
    out = tmp_path / "boxes.npy"
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, SAD_ROOT=ROOT, SAD_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0", SAD_DIST_FORCE_COLLECTIVE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    subprocess.check_call([sys.executable, str(script)], env=env, timeout=600)
    cfg = config.TINY
    det = SADDetector(cfg, synth.make_weights(cfg, 0), dev)
    want = det(torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points)).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(np.load(out), want)


def test_placed_streams_and_extra_streams(sad, dev):
    """A detector makes its streams through ``_runtime.placed_streams`` (main streams alone on their dispatch pipes) and hands
    the caller the extra ones: distinct streams, the gather and the ingest pipeline take them, results unchanged."""
    import torch
    from sad_amd import _runtime, config, synth
    from sad_amd.detector import SADDetector
    from sad_amd.dist import AsyncBoxGather
    from sad_amd.pipeline import IngestPipeline
    sides, mains, extras = _runtime.placed_streams(dev, 3, 2, 2)
    hs = [s.cuda_stream for s in sides + mains + extras]
    assert len(sides) == 3 and len(mains) == 2 and len(extras) == 2 and len(set(hs)) == 7 and all(h != 0 for h in hs)
    cfg = config.TINY
    wts = synth.make_weights(cfg, 0)
    det = SADDetector(cfg, wts, dev, n_fps_streams=4)
    assert len(det.extra_streams) == 2 and len(det._sides) == 4 and len(det._mains) == 2
    pts = torch.from_numpy(synth.make_tiny_batch(0, 4, cfg.n_points)).to(dev)
    want = det(pts)
    g = AsyncBoxGather(dev, stream=det.extra_streams[0])
    assert g.stream is det.extra_streams[0]
    outs = [det.submit(pts, post=g)[0] for _ in range(5)]
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs)
    pipe = IngestPipeline(det, 4, cols=pts.shape[2], max_points_per_scene=cfg.n_points + 8, in_slots=1, out_slots=2)
    assert pipe.ingest is det.extra_streams[1]
    det2 = SADDetector(cfg, wts, dev, streams=(det._sides, det._mains, det.extra_streams))
    assert det2.extra_streams[0] is det.extra_streams[0]
    assert torch.equal(det2(pts), want)
