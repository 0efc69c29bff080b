"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/sad_amd.h declares; the Python surface refuses CPU tensors (no silent fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sad_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sad_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(sad):
    from sad_amd import _lib
    _lib.build()
    handle = ctypes.CDLL(_lib.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/sad_amd.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert _lib.lib().sad_version() == _lib.ABI_VERSION == 4


def test_no_oracle_in_product_path():
    """The product package must never import or call the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "3dsad-main_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libsad_oracle" not in src and "orc_" not in src, f


def test_host_side_argument_errors(sad):
    """Error paths that need no GPU: bad arguments come back as negative codes + a message."""
    from sad_amd import _lib
    L = _lib.lib()
    assert L.sad_fps_f32(None, 1, 8, 4, None, None, None) == -1
    assert b"NULL" in L.sad_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.sad_fps_f32(p, 1, 8, 9, p, None, None) == -1            # M > N
    assert L.sad_knn_f32(p, p, 1, 8, 2, 65, p, None) == -1          # K > 64
    assert L.sad_ball_query_f32(p, p, 0.5, None, 1, 8, 2, 65, p, None) == -1   # nsample > 64
    assert L.sad_group_points(p, p, 1, 1, 8, 2, 2, 3, p, None) == -1  # elem_size 3
    assert L.sad_set_option(b"no_such_option", 1) == -1
    assert L.sad_fps_workspace_bytes(2, 1024) == 0
    assert L.sad_fps_workspace_bytes(2, 16384) == 2 * 16384 * 4 + 2 * 65536 * 16   # Z-order permutation + sorted records
    assert L.sad_fps_workspace_bytes(2, 65536) == 2 * 65536 * 4 + 2 * 65536 * 16   # permutation + sorted float4 records
    assert L.sad_fps_workspace_bytes(2, 100000) == 2 * 100000 * 4             # plain min-distance workspace
    assert L.sad_ffps_workspace_bytes(2, 1000) == 2 * 1000 * 1000 * 4
    assert L.sad_ffps_f32(p, p, 4, 1, 8, 4, 9, 1.0, p, p, None) == -1        # M > N
    dims = (ctypes.c_int * 4)(259, 256, 512, 1024)
    n = L.sad_mlp_packed_floats(3, dims, 1)
    kp = [264, 256, 512]
    npad = [256, 512, 1024]
    raw = sum(-(-ci * co // 4) * 4 + -(-co // 4) * 4 for ci, co in zip((259, 256, 512), (256, 512, 1024)))
    assert n == sum(a + a * k for a, k in zip(npad, kp)) + raw   # MFMA fragments + plain k-major copy


def test_struct_size_mismatch_is_refused(sad):
    """a caller built against another header (shorter / longer sad_mlp_args) is refused with SAD_EINVAL
    before any field behind the common prefix is read; the ctypes mirrors have the C sizes."""
    from sad_amd import _lib
    L = _lib.lib()
    for cls, fn in ((_lib.MlpArgs, L.sad_mlp_chain_f32), (_lib.MlpBf16Args, L.sad_mlp_chain_bf16)):
        a = cls()
        a.struct_size = ctypes.sizeof(cls) - 8          # e.g. the round-1 struct without its trailing fields
        assert fn(ctypes.byref(a), None) == -1
        assert b"struct_size" in L.sad_last_error()
        a.struct_size = ctypes.sizeof(cls)              # right size, NULL everything: the next check fails instead
        assert fn(ctypes.byref(a), None) == -1
        assert b"struct_size" not in L.sad_last_error()


def test_layer_streamed_chain_refuses_rows_beyond_4gib(sad):
    """mlp_layer_kernel forms row byte offsets in 32 bits; a plain-row call whose rows x stride x 4
    reaches 4 GiB is refused with SAD_EUNSUPPORTED on the host (nothing is launched, nothing wraps)."""
    from sad_amd import _lib
    L = _lib.lib()
    a = _lib.MlpArgs()
    a.struct_size = ctypes.sizeof(_lib.MlpArgs)
    a.feat, a.packed, a.out = 0x10000, 0x20000, 0x30000      # never dereferenced: the call fails on the host
    a.L, a.S, a.B, a.relu_mask, a.geometry = 1, 1, 1, 1, 3
    a.dims[0], a.dims[1] = 512, 128
    a.C, a.ld_feat, a.ld_out = 512, 512, 128
    a.M = 2 * 1024 * 1024 + 128                               # x 512 floats x 4 bytes = 4 GiB + 256 KiB
    assert L.sad_mlp_chain_f32(ctypes.byref(a), None) == -2
    assert b"4 GiB" in L.sad_last_error()
    # grouped form: 64 nuScenes-sized scenes dense (VERDICT's example): hidden activations 2.1 M rows x 2 KiB
    g = _lib.MlpArgs()
    g.struct_size = ctypes.sizeof(_lib.MlpArgs)
    for f in ("xyz", "new_xyz", "idx", "cnt", "workspace", "feat", "packed", "out", "scratch"):
        setattr(g, f, 0x10000)
    g.L, g.B, g.N, g.M, g.S, g.C, g.ld_feat = 3, 64, 2048, 1024, 32, 256, 256
    for i, d in enumerate((259, 256, 512, 1024)):
        g.dims[i] = d
    g.relu_mask, g.geometry, g.ld_out = 7, 3, 1024
    dims = (ctypes.c_int * 4)(259, 256, 512, 1024)
    g.scratch_bytes = L.sad_mlp_scratch_bytes(64, 1024, 32, 3, dims)
    assert L.sad_mlp_chain_f32(ctypes.byref(g), None) == -2
    assert b"4 GiB" in L.sad_last_error()


def test_split_pooling_boundary_checks(sad):
    """ABI 4 (split pooling): sizes and refusals that need no GPU.  A table grows by one int per group, a continuation buffer has one row per
    32-row tile + the zero row, and the layer that reads split-pooled rows is ONE plain layer of bf16 rows on the row-streaming kernel."""
    from sad_amd import _lib
    L = _lib.lib()
    assert L.sad_mlp_cont_bytes(2, 100, 32, 64) == ((2 * 100 * 32 + 31) // 32 + 1) * 64 * 2
    assert L.sad_mlp_cont_bytes(0, 100, 32, 64) == 0
    ng, S = 2 * 100, 32
    before = 4 + (ng + 1) + (ng * S // 32 + 2) + (ng // 1024 + 2) + 2 * ng * S
    assert L.sad_mlp_workspace_bytes(2, 100, S) == 4 * (((before + 3) & ~3) + ng + 1) + 64
    assert L.sad_mlp_rowscan_split(1, None, None, None, 1, 8, 4, None, None, None, None) == -1
    a = _lib.MlpBf16Args()
    a.struct_size = ctypes.sizeof(_lib.MlpBf16Args)
    a.feat, a.packed, a.out = 0x10000, 0x20000, 0x30000          # never dereferenced: every call below fails on the host
    a.B, a.M, a.S, a.relu_mask = 1, 256, 1, 1
    a.n_pool = 1
    a.pool_ws[0], a.pool_cont[0], a.pool_S[0], a.pool_cols[0] = 0x40000, 0x50000, 32, 128
    # two layers: refused (only a single plain layer reads split-pooled rows)
    a.L, a.C, a.ld_feat, a.feat_bf16, a.ld_out = 2, 128, 128, 1, 32
    a.dims[0], a.dims[1], a.dims[2] = 128, 64, 32
    assert L.sad_mlp_chain_bf16(ctypes.byref(a), None) == -2 and b"split-pooled" in L.sad_last_error()
    # f32 rows: refused
    a.L, a.feat_bf16 = 1, 0
    assert L.sad_mlp_chain_bf16(ctypes.byref(a), None) == -2
    # column widths that do not add up to C / are not multiples of 16
    a.feat_bf16 = 1
    a.pool_cols[0] = 64
    assert L.sad_mlp_chain_bf16(ctypes.byref(a), None) == -1 and b"add up" in L.sad_last_error()
    a.pool_cols[0], a.C, a.ld_feat, a.dims[0] = 24, 24, 24, 24
    assert L.sad_mlp_chain_bf16(ctypes.byref(a), None) == -1
    # a grouped call with bf16 output but no continuation buffer / another kernel than the register-resident chain
    g = _lib.MlpBf16Args()
    g.struct_size = ctypes.sizeof(_lib.MlpBf16Args)
    for f in ("xyz", "new_xyz", "idx", "feat", "packed", "out", "cnt", "workspace"):
        setattr(g, f, 0x10000)
    g.B, g.N, g.M, g.S, g.C, g.L, g.ld_feat, g.feat_bf16, g.relu_mask, g.ld_out = 1, 64, 16, 32, 64, 3, 64, 1, 7, 128
    for i, d in enumerate((67, 64, 64, 128)):
        g.dims[i] = d
    g.out_bf16, g.geometry = 1, 2
    assert L.sad_mlp_chain_bf16(ctypes.byref(g), None) == -1 and b"continuation" in L.sad_last_error()
    g.cont, g.geometry = 0x60000, 128
    assert L.sad_mlp_chain_bf16(ctypes.byref(g), None) == -1


def test_ops_refuse_cpu_tensors(sad):
    import torch
    from sad_amd import ops
    x = torch.zeros(1, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.fps(x, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.ball_query(0.2, 4, x, x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.group_points(torch.zeros(1, 2, 16), torch.zeros(1, 2, 2, dtype=torch.int32))


def test_config_work_accounting(sad):
    """SURVEY.md §8(d) figures: the roofline numbers in bench.py come from these."""
    from sad_amd import config
    w = config.work_per_scene(config.KITTI)
    assert w["fps_updates"] == 71827456 and w["fps_steps"] == 5632
    assert w["pair_tests"] == 215744512
    assert abs(w["ball_query_bytes"] / 1e6 - 3.86) < 0.01
    assert abs(w["group_points_bytes"] / 1e6 - 91.6) < 0.1
    assert abs(w["mlp_flops"] / 1e9 - 31.0) < 0.1
    names = [n for n, _ in config.mlp_layers(config.KITTI)]
    assert names[0] == "sa1.b0" and names[-1] == "head" and "cluster.b1" in names


def test_synth_is_deterministic(sad):
    import numpy as np
    from sad_amd import synth
    a, b = synth.make_scene(3), synth.make_scene(3)
    assert a.shape == (16384, 4) and a.dtype == np.float32
    np.testing.assert_array_equal(a, b)
    assert not np.array_equal(a, synth.make_scene(4))
    assert a[:, 0].min() >= 0 and a[:, 0].max() <= 70.4 and abs(a[:, 1]).max() <= 40


def test_boundary_is_reentrant_across_threads(sad):
    """SURVEY.md §8(b) "re-entrant, no global mutable state": sad_last_error is thread-local, the
    tuning knobs are atomics and the argument-error paths share nothing — four threads hammering
    different failing calls (ctypes releases the GIL around each call) always read back their OWN
    message, while a fifth flips an option the whole time."""
    import threading
    from sad_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    cases = [
        (lambda: L.sad_set_option(b"bogus_alpha", 1), b"bogus_alpha"),
        (lambda: L.sad_set_option(b"bogus_beta", 1), b"bogus_beta"),
        (lambda: L.sad_fps_f32(None, 1, 8, 4, None, None, None), b"sad_fps_f32"),
        (lambda: L.sad_knn_f32(p, p, 1, 8, 2, 65, p, None), b"sad_knn_f32"),
    ]
    errors = []
    stop = threading.Event()

    def flipper():
        v = 0
        while not stop.is_set():
            L.sad_set_option(b"mlp_noxcd", v & 1)
            v += 1
        L.sad_set_option(b"mlp_noxcd", 0)

    def worker(call, needle):
        for _ in range(3000):
            if call() != -1:
                errors.append((needle, "return code"))
                return
            msg = L.sad_last_error()
            if needle not in msg:
                errors.append((needle, msg))
                return

    f = threading.Thread(target=flipper)
    f.start()
    ts = [threading.Thread(target=worker, args=c) for c in cases]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    f.join()
    assert not errors, errors[:3]
    # a thread that never failed sees an empty message, whatever the others did
    seen = []
    t = threading.Thread(target=lambda: seen.append(L.sad_last_error()))
    t.start()
    t.join()
    assert seen == [b""]


def test_package_owns_hw_queue_precondition(monkeypatch):
    """The package (not only bench.py) exports GPU_MAX_HW_QUEUES before HIP initialises, leaves a user's value alone, and
    warns when a pipeline creates more streams than there are hardware queues.  Pure env logic."""
    import warnings
    from sad_amd import _runtime
    env = {}
    assert _runtime.ensure_hw_queues(env, initialised=False) == "set" and env["GPU_MAX_HW_QUEUES"] == "24"
    assert _runtime.ensure_hw_queues(env, initialised=False) == "user"          # second import: already there
    env = {"GPU_MAX_HW_QUEUES": "8"}
    assert _runtime.ensure_hw_queues(env, initialised=False) == "user" and env["GPU_MAX_HW_QUEUES"] == "8"
    env = {}
    assert _runtime.ensure_hw_queues(env, initialised=True) == "late" and "GPU_MAX_HW_QUEUES" not in env
    # torch imported, its CUDA flag still False: is_available() / a profiler preload may have brought HIP up unseen
    env = {}
    assert _runtime.ensure_hw_queues(env, initialised=None) == "unknown" and env["GPU_MAX_HW_QUEUES"] == "24"
    assert _runtime.hw_queues({}) == 4 and _runtime.hw_queues({"GPU_MAX_HW_QUEUES": "16"}) == 16
    assert _runtime.hw_queues({"GPU_MAX_HW_QUEUES": "x"}) == 4
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert _runtime.check_stream_budget(6, "set", {"GPU_MAX_HW_QUEUES": "16"})        # default detector: 2 + 3 + 1
        assert _runtime.check_stream_budget(4, "late", {})                                # fits HIP's default 4
        assert _runtime.check_stream_budget(4, "late", {})                                # ... every time (returned False on the second call)
        assert _runtime.check_stream_budget(4, "unknown", {"GPU_MAX_HW_QUEUES": "16"})
    with pytest.warns(RuntimeWarning, match="after torch was imported"):                                     # more than 4 streams on an export nobody may have read
        assert _runtime.check_stream_budget(6, "unknown", {"GPU_MAX_HW_QUEUES": "16"})
    _runtime._warned.discard(None)
    assert _runtime.check_stream_budget(3, "late") and _runtime.check_stream_budget(3, "late")   # process-wide path (environ=None), twice
    with pytest.warns(RuntimeWarning, match="hardware queues"):
        assert not _runtime.check_stream_budget(6, "late", {})
    with pytest.warns(RuntimeWarning, match="share a"):
        assert not _runtime.check_stream_budget(9, "user", {"GPU_MAX_HW_QUEUES": "8"})
    # the import itself did it for this process (conftest imports nothing that touches the GPU first)
    import sad_amd
    assert sad_amd.HW_QUEUES_STATE in ("set", "user", "unknown")
    import os
    assert os.environ.get("GPU_MAX_HW_QUEUES")


def test_stream_placement_order(monkeypatch):
    """Order of first use of a pipeline's streams (``_runtime.placement_order``): hardware queues are numbered in that order and
    numbers four apart share a dispatch pipe, so each of the (at most two) main streams must be the ONLY live stream of its
    residue class — sampling / extra streams on the even places, idle dummies on the later odd places.  Pure logic."""
    from sad_amd import _runtime
    monkeypatch.delenv("SAD_NO_STREAM_PLACEMENT", raising=False)
    for n_side, n_main, n_extra in ((3, 2, 1), (6, 2, 2), (3, 1, 0), (1, 1, 0), (0, 1, 0), (8, 2, 1), (2, 2, 0), (0, 2, 1)):
        order = _runtime.placement_order(n_side, n_main, n_extra)
        kinds = [k for k, _ in order]
        assert sorted(i for k, i in order if k == "side") == list(range(n_side))
        assert sorted(i for k, i in order if k == "main") == list(range(n_main))
        assert sorted(i for k, i in order if k == "extra") == list(range(n_extra))
        assert kinds[0] != "dummy" and kinds[-1] != "dummy"
        for pos, (k, _) in enumerate(order):
            if k == "main":
                same_pipe = [kinds[q] for q in range(len(order)) if q != pos and q % 4 == pos % 4]
                assert all(x == "dummy" for x in same_pipe), (n_side, n_main, n_extra, order)
    # three main streams: each on a pipe of its own, everything else on the fourth
    order = _runtime.placement_order(4, 3, 1)
    kinds = [k for k, _ in order]
    assert kinds[:4] == ["side", "main", "main", "main"] and kinds.count("side") == 4 and kinds.count("extra") == 1
    assert all(k == "dummy" for q, k in enumerate(kinds) if q > 3 and q % 4 != 0) and all(k != "dummy" for q, k in enumerate(kinds) if q % 4 == 0)
    # four main streams: no pipe to spare, plain order, no dummies; and the A/B switch
    assert [k for k, _ in _runtime.placement_order(6, 4, 1)] == ["side"] * 6 + ["main"] * 4 + ["extra"]
    monkeypatch.setenv("SAD_NO_STREAM_PLACEMENT", "1")
    assert [k for k, _ in _runtime.placement_order(3, 2, 1)] == ["side"] * 3 + ["main"] * 2 + ["extra"]
