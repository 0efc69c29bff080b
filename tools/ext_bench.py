"""Times the §8(f) operators (NMS, F-FPS, backward kernels) at detector-like sizes, B = 32."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sad_amd
from sad_amd import ops, autograd as ag
dev = torch.device("cuda:0")
B = 32
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
# NMS: 256 boxes per scene
rng = np.random.default_rng(0)
bx = np.zeros((B, 256, 9), np.float32)
bx[..., 0] = rng.uniform(0, 70, (B, 256)); bx[..., 1] = rng.uniform(-40, 40, (B, 256))
bx[..., 3:6] = rng.uniform(1.5, 4.5, (B, 256, 3)); bx[..., 6] = rng.uniform(-3, 3, (B, 256)); bx[..., 7] = rng.uniform(0, 1, (B, 256))
bt = torch.from_numpy(bx).to(dev)
for sk in (True, False):
    t = timeit(lambda: ops.nms_bev(bt, 0.1, 0.1, single_kernel=sk))
    print(f"nms_bev B={B} K=256 single_kernel={sk}: {t*1e3:.1f} us per batch ({B*256*255/2/(t*1e-3)/1e9:.2f} G IoU pairs/s)")
# F-FPS: SA2-like 4096 -> 512 with 64 / 128 channels
for N, C, M in ((4096, 64, 1024), (4096, 128, 512), (1024, 256, 256)):
    xyz = torch.rand(B, N, 3, device=dev) * 50
    feat = torch.randn(B, N, C, device=dev)
    ws = torch.empty((B * N * N,), device=dev)
    from sad_amd._lib import lib, check
    tp = timeit(lambda: check(lib().sad_pairdist_f32(xyz.data_ptr(), feat.data_ptr(), C, B, N, C, 1.0, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), "pd"))
    tf = timeit(lambda: ops.ffps(xyz, feat, M))
    ops_ = 3.0 * B * N * N * (C + 3)
    print(f"ffps N={N} C={C} M={M}: pairdist {tp:.3f} ms = {ops_/tp/1e9:.1f} TFLOP/s VALU ({B*N*N*4/tp/1e6:.0f} GB/s written), total {tf:.3f} ms, chain {(tf-tp)*1e3/M:.2f} us/step")
# backward: SA2 branch shape
C, N, M, S = 64, 4096, 1024, 32
g = torch.randn(B, C, M, S, device=dev)
idx = torch.randint(0, N, (B, M, S), device=dev, dtype=torch.int32)
for name, kw in (("channel-major scatter", dict(via_point_major=False)), ("point-major atomics + transpose", dict()), ("point-major output", dict(point_major=True))):
    t = timeit(lambda: ag.group_points_grad(g, idx, N, **kw))
    print(f"group_points_grad [{name}] C={C} N={N} M={M} S={S}: {t:.3f} ms = {B*C*M*S*4/t/1e6:.0f} GB/s of added bytes")
x = torch.randn(B, C, M, S, device=dev)
t = timeit(lambda: ag.max_pool_s_with_arg(x))
print(f"max_pool_s: {t:.3f} ms = {B*C*M*S*4/t/1e6:.0f} GB/s")
