"""Measurement build only (bash tools/probe/build_stamps.sh rowst mlp_bf16_rows.hip -DSAD_ROWS_STAMPS -> build/libsad_rowst.so):
per-phase s_memtime sums over the chunks of the first 1024 waves of the row-streaming bf16 layer (queued loop).
usage: rows_stamps.py ROWS K COUT [f32|bf16]     e.g. 8192 1536 512 (cluster.agg), 16384 768 256 (sa3.agg), 32768 1536 512 (nuScenes cluster.agg)"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
os.environ["SAD_AMD_LIB"] = os.path.join(root, "build", "libsad_rowst.so")
import numpy as np, torch
import sad_amd
from sad_amd import ops, synth
dev = torch.device("cuda:0")
R, K, CO = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
xdt = torch.bfloat16 if (len(sys.argv) < 5 or sys.argv[4] == "bf16") else torch.float32
rng = np.random.default_rng(0)
mlp = ops.PackedMLPBf16(synth.make_mlp_weights([K, CO], rng), False, dev)
x = torch.randn((R, K), device=dev).to(xdt)
out = torch.empty((R, CO), device=dev, dtype=torch.bfloat16)
from sad_amd import _lib
_lib.set_option("mlp_rows_form", 0 if os.environ.get("SAD_ROWS_FORM2") else 1)
for _ in range(3): mlp.rows(x, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); mlp.rows(x, out=out); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(os.environ["SAD_AMD_LIB"])
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert L.sad_debug_read_rows_stamps(buf) == 0
s = np.array(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
s = s[s[:, 4] > 0]
ch = s[:, 4].sum()
print(f"rows {R} x {K} -> {CO} ({xdt}): launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build); waves sampled {len(s)}, chunks per wave {s[:, 4].mean():.1f}")
if os.environ.get("SAD_ROWS_FORM2"):
    print(f"  second form, per chunk: DMA issue {s[:, 0].sum() / ch:.1f}, B fragments (LDS) {s[:, 1].sum() / ch:.1f}, weight reads + MFMAs {s[:, 2].sum() / ch:.1f}, "
          f"vmcnt wait {s[:, 3].sum() / ch:.1f}, barrier {s[:, 7].sum() / ch:.1f}; wave life {s[:, 5].mean():.0f} cycles = {s[:, 6].mean() / 100.0:.1f} us")
    sys.exit(0)
print(f"  per chunk (s_memtime cycles; 16 MFMAs of a full chunk with four channel tiles = 512): wait rows {s[:, 0].sum() / ch:.1f}, LDS reads + MFMAs {s[:, 1].sum() / ch:.1f}, "
      f"wait weights + LDS store {s[:, 2].sum() / ch:.1f}, barrier {s[:, 3].sum() / ch:.1f}; wave life {s[:, 5].mean():.0f} cycles = {s[:, 6].mean() / 100.0:.1f} us ({s[:, 5].sum() / s[:, 6].sum() * 100:.0f} s_memtime ticks per us)")
