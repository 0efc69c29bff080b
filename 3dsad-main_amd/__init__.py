"""MI355X-native set-abstraction + size-adaptive-clustering hot path (import as ``sad_amd``).

Operator surface named by BASELINE.json ``north_star`` (the upstream reference,
``/root/reference/README.md:1-2``, ships no code): ``fps``, ``ball_query``, ``knn_query``,
``group_points``, ``gather_points``, ``sa_module`` — Python host code on PyTorch-ROCm calling
hand-written gfx950 HIP kernels through the C-ABI library declared in ``include/sad_amd.h``.

``config`` and ``synth`` are numpy-only; everything that touches the GPU is imported lazily so the
data modules stay importable without torch.  There is no CPU fallback: operators raise if the HIP
library is missing or a tensor is not on a GPU.
"""
__version__ = "0.1.0"

# the package owns its runtime preconditions (GPU_MAX_HW_QUEUES before HIP initialises): _runtime.py
from . import _runtime
HW_QUEUES_STATE = _runtime.ensure_hw_queues()

_LAZY = {
    "fps": "ops", "ball_query": "ops", "ball_query_multi": "ops", "knn_query": "ops",
    "group_points": "ops", "gather_points": "ops", "gather_xyz": "ops",
    "mlp_chain": "ops", "PackedMLP": "ops", "nms_bev": "ops",
    "SAModuleMSG": "sa_module", "SAModule": "sa_module", "sa_module": "sa_module",
    "SADDetector": "detector", "IngestPipeline": "pipeline",
    "shard_range": "dist", "all_gather_boxes": "dist", "run_sharded": "dist",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")
        return getattr(mod, name)
    raise AttributeError(name)
