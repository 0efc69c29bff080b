"""Host logic of the step plans (3dsad-main_amd/plan.py) that needs no GPU: what the recorder keeps of a launch, how the input
pointer is found and patched, what it refuses, and that a failing launch surfaces on replay.  (Replay against the eager path on
the device: tests/test_gpu_plan.py.)"""
import ctypes

import pytest


class _FakeLib:
    """Stands in for the ctypes library: launches append what they were called with, size queries return numbers."""

    def __init__(self):
        self.calls = []
        self.fail = False

    def sad_fps_f32(self, *a):
        self.calls.append(("fps", a))
        return -3 if self.fail else 0

    def sad_copy_rows_u32(self, *a):
        self.calls.append(("copy", a))
        return 0

    def sad_mlp_chain_f32(self, *a):
        self.calls.append(("mlp", a))
        return 0

    def sad_fps_workspace_bytes(self, *a):
        return 128

    def sad_last_error(self):
        return b"fake failure"


class _T:   # the two things the recorder asks of an input tensor
    def __init__(self, ptr, nbytes):
        self._p, self._n = ptr, nbytes

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def element_size(self):
        return 1


def _plan(monkeypatch):
    import sad_amd  # noqa: F401
    from sad_amd import _lib, plan
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    return fake, plan, plan.Recorder(real_lib=fake)


def test_launches_are_recorded_and_size_queries_are_not(monkeypatch):
    fake, plan, rec = _plan(monkeypatch)
    assert rec.lib.sad_fps_workspace_bytes(2, 4096) == 128          # passed through, not recorded
    assert rec.lib.sad_fps_f32(0x9000, 2, 4096, 512, 0xA000, 0, 7) == 0
    assert len(rec.ops) == 1 and rec.ops[0][0] == plan.OP_CALL and rec.ops[0][2] == (0x9000, 2, 4096, 512, 0xA000, 0, 7)
    fake.fail = True
    assert rec.lib.sad_fps_f32(0x9000, 2, 4096, 512, 0xA000, 0, 7) == -3
    assert len(rec.ops) == 1                                          # a refused launch is not part of the plan


def test_input_pointer_is_found_by_range_and_patched(monkeypatch):
    fake, plan, rec = _plan(monkeypatch)
    rec.mark_input(_T(0x10000, 4096))
    rec.lib.sad_copy_rows_u32(0x10000, 4, 0x50000, 3, 100, 3, 11)             # reads the input at +0
    rec.lib.sad_copy_rows_u32(0x10000 + 12, 4, 0x60000, 1, 100, 1, 11)        # ... and at +12 (the feature column)
    rec.lib.sad_fps_f32(0x50000, 2, 50, 10, 0x70000, 0, 11)                   # reads a buffer the plan owns: not patched
    p = rec.finish(out="boxes", stream=None)
    assert sorted(p.patches) == [(0, 0, 0), (1, 0, 12)] and p.n_calls == 3
    fake.calls.clear()
    assert p.replay(0x88000) == "boxes"
    assert [c[1][0] for c in fake.calls] == [0x88000, 0x88000 + 12, 0x50000]
    assert p.replay(0x99000) == "boxes" and fake.calls[3][1][0] == 0x99000     # patched again on every replay
    fake.fail = True
    with pytest.raises(RuntimeError, match="fake failure"):
        p.replay(0x88000)


def test_recorder_refuses_what_it_cannot_replay(monkeypatch):
    fake, plan, rec = _plan(monkeypatch)
    with pytest.raises(plan.PlanUnsupported, match="exactly one input"):
        rec.finish(out=None, stream=None)
    rec.mark_input(_T(0x10000, 4096))
    rec.lib.sad_fps_f32(0x50000, 2, 50, 10, 0x70000, 0, 11)
    with pytest.raises(plan.PlanUnsupported, match="no launch reads the input"):
        rec.finish(out=None, stream=None)
    # a pointer INSIDE an argument block that aliases the input cannot be patched: refused
    from sad_amd._lib import MlpArgs
    a = MlpArgs()
    a.feat = 0x10000 + 12
    rec.lib.sad_copy_rows_u32(0x10000, 4, 0x50000, 3, 100, 3, 11)
    rec.lib.sad_mlp_chain_f32(ctypes.byref(a), 11)
    with pytest.raises(plan.PlanUnsupported, match="feat"):
        rec.finish(out=None, stream=None)


def test_unrecordable_only_raises_while_recording(monkeypatch):
    import sad_amd  # noqa: F401
    from sad_amd import _lib, plan
    plan.unrecordable("fill")                                        # nothing while no plan records
    _lib.set_recorder(object())
    try:
        with pytest.raises(plan.PlanUnsupported):
            plan.unrecordable("fill")
    finally:
        _lib.set_recorder(None)
    assert _lib.is_launch("sad_mlp_chain_f32") and _lib.is_launch("sad_copy_rows_u32")
    assert not _lib.is_launch("sad_mlp_workspace_bytes") and not _lib.is_launch("sad_mlp_padded_dims") and not _lib.is_launch("sad_version")


def test_stage_level_tuner_prefers_table_kernels():
    """The decision that flipped a coin in round 5: the small cluster branch is faster alone on the tiled kernel, the mixed dispatch
    within 2 % of the uniform one — the uniform table assignment must win unless the mixed pick is more than 10 % faster."""
    import sad_amd  # noqa: F401
    from sad_amd.ops import choose_stage_assignment as choose
    # the measured case: picked [tiled, 3] 0.775 ms against all-3 0.782 ms (and all-2 / all-4 refused for these shapes)
    assert choose([24831, 3], 0.775, {2: None, 3: 0.782, 4: None}) == ([3, 3], 0.782)
    assert choose([24831, 3], 0.790, {2: None, 3: 0.782, 4: None}) == ([3, 3], 0.782)
    # a mixed pick that really is far faster is kept
    assert choose([24831, 3], 0.600, {2: None, 3: 0.782, 4: None}) == ([24831, 3], 0.600)
    # picks that are all table kernels: replaced only by a uniform assignment that is 2 % faster
    assert choose([2, 4, 4], 0.630, {2: 0.650, 3: 0.700, 4: 0.625}) == ([2, 4, 4], 0.630)
    assert choose([2, 4, 4], 0.630, {2: 0.650, 3: 0.700, 4: 0.600}) == ([4, 4, 4], 0.600)
    # among uniform assignments an earlier code keeps a tie within 2 %
    assert choose([0, 0], 1.0, {2: 0.700, 3: 0.695, 4: 0.690})[0] == [2, 2]
    # nothing uniform fits: the picks stand; the mixed dispatch was refused: the uniform one is taken
    assert choose([831, 832], 0.5, {2: None, 3: None, 4: None}) == ([831, 832], 0.5)
    assert choose([831, 3], None, {2: None, 3: 0.9, 4: None}) == ([3, 3], 0.9)
