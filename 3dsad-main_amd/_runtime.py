"""Runtime preconditions the package owns (no torch import here: this runs at ``import sad_amd``).

HIP multiplexes a process's streams onto ``GPU_MAX_HW_QUEUES`` hardware queues (default 4).  The detector's
pipeline uses two main streams, three to eight sampling streams, a gather and an ingest stream (plus idle placeholders: below); when a 3 - 16 ms FPS kernel
shares a hardware queue with MLP launches (or with another FPS chain) the queue serialises them — measured:
9.5 - 13.2 k scenes/s instead of 14 k on the f32 benchmark.  The variable is read when the HIP
runtime initialises, so it has to be in the environment before the first HIP call of the process.  There is no
reference behaviour to mirror (``/root/reference/README.md:1-2`` is the whole upstream repository).
"""
import os
import sys
import warnings

HW_QUEUES_ENV = "GPU_MAX_HW_QUEUES"
HW_QUEUES_DEFAULT = 4          # HIP's default when the variable is unset
HW_QUEUES_WANTED = 24          # every stream of the largest pipeline on its own queue, placeholders included (placed_streams below: 2 main + 8 sampling + 2 extra + 7 idle = 19)


def _hip_initialised():
    """Has this process made a HIP call already?  False: torch is not imported (certainly not).  True: torch's CUDA
    state is initialised.  None = UNKNOWN: torch is imported but its flag is still False — ``torch.cuda.is_available()``
    / ``device_count()`` call hipGetDeviceCount, which brings the HIP runtime up (and makes it read the variable) without
    setting that flag, and a profiler preload (rocprofv3 --pmc) does the same before Python starts."""
    torch = sys.modules.get("torch")
    if torch is None:
        return False
    try:
        return True if torch.cuda.is_initialized() else None
    except Exception:
        return None


_ASK = object()


def ensure_hw_queues(environ=None, initialised=_ASK) -> str:
    """Called at ``import sad_amd``.  Leaves a value the user exported alone; otherwise exports
    ``GPU_MAX_HW_QUEUES=24`` when the HIP runtime cannot have initialised yet.  Returns what happened:
    "user" (already set), "set" (exported here: torch was not imported, no HIP call can have been made through it),
    "unknown" (exported here, but torch was imported already and ANY ``torch.cuda`` call — ``is_available()`` included —
    or a profiler preload may have initialised HIP with the default of 4 queues: the export may be read by nobody),
    "late" (unset, and torch's CUDA state is up: nothing done).

    The variable must be in the environment before the first HIP call of the process, ``torch.cuda.is_available()``
    included: export it in the shell (the documented way), or ``import sad_amd`` before ``import torch`` does anything
    with the GPU.  Note that the export changes ``os.environ`` for every other HIP user of the process."""
    env = os.environ if environ is None else environ
    if env.get(HW_QUEUES_ENV):
        return "user"
    up = _hip_initialised() if initialised is _ASK else initialised       # True / False / None (= unknown)
    if up:
        return "late"
    env[HW_QUEUES_ENV] = str(HW_QUEUES_WANTED)
    return "set" if up is False else "unknown"


def hw_queues(environ=None) -> int:
    """Hardware queues the runtime was (or will be) told to use."""
    env = os.environ if environ is None else environ
    try:
        return max(1, int(env.get(HW_QUEUES_ENV, "") or HW_QUEUES_DEFAULT))
    except ValueError:
        return HW_QUEUES_DEFAULT


_warned = set()


def check_stream_budget(n_streams: int, state: str, environ=None) -> bool:
    """Warn (once per kind and process) when a pipeline is about to create more streams than there are hardware queues.
    ``state`` is what ``ensure_hw_queues`` returned at import.  Returns True when the budget is fine ("unknown": the
    variable was exported after torch was imported — more than 4 streams get one warning that the export may be unread)."""
    q = hw_queues(environ)
    if n_streams <= q and state not in ("late", "unknown"):
        return True
    if state == "unknown" and n_streams > HW_QUEUES_DEFAULT and n_streams <= q:
        if environ is None:
            if "unknown" in _warned:
                return True
            _warned.add("unknown")
        warnings.warn(f"sad_amd: {HW_QUEUES_ENV}={q} was exported at `import sad_amd`, after torch was imported: if any torch.cuda call "
                      f"(is_available() included) or a profiler preload initialised HIP before that, the runtime kept its default of "
                      f"{HW_QUEUES_DEFAULT} hardware queues and {n_streams} streams will share them.  Export the variable in the shell.",
                      RuntimeWarning, stacklevel=3)
        return True
    kind = "late" if (state == "late" and n_streams > HW_QUEUES_DEFAULT) else ("over" if n_streams > q else None)
    if kind is None:             # ("late" with at most 4 streams: the default queues are enough)
        return True
    if environ is None:          # (explicit environments are the unit tests: always warn there)
        if kind in _warned:
            return False
        _warned.add(kind)
    if state == "late" and n_streams > HW_QUEUES_DEFAULT:
        warnings.warn(f"sad_amd: {n_streams} HIP streams are about to share {HW_QUEUES_DEFAULT} hardware queues: "
                      f"{HW_QUEUES_ENV} was unset when the HIP runtime initialised (import sad_amd, or export "
                      f"{HW_QUEUES_ENV}={HW_QUEUES_WANTED}, before the first HIP call).  Sampling chains will serialise "
                      "with MLP launches (measured: -10 to -30 % throughput).", RuntimeWarning, stacklevel=3)
        return False
    if n_streams > q:
        warnings.warn(f"sad_amd: {n_streams} HIP streams on {HW_QUEUES_ENV}={q} hardware queues: streams that share a "
                      f"queue serialise (use {HW_QUEUES_WANTED}, or fewer sampling streams).", RuntimeWarning, stacklevel=3)
        return False
    return True


# ---- stream placement --------------------------------------------------------------------------------------------------
# A HIP stream takes its hardware queue when it is first used, queues are numbered in that order, and queues whose numbers
# differ by a multiple of four are served by the same dispatch pipe of the command processor (measured on MI355X, round 5:
# DESIGN.md §5 "stream placement", profiles/r05_pipe_rule.txt).  A main stream — ~40 launches per step, many of them a few microseconds long — that shares
# its pipe with a sampling stream, with the gather stream (whose head packet is a barrier waiting for the step's end) or with
# the other main stream loses 4 - 7 % of the pipelined step: its short kernels wait for the pipe.  So the streams of a
# pipeline are created AND touched here in an order that leaves each main stream alone on its pipe: sampling and extra
# streams take the even places, the (at most two) main streams the first two odd places, idle dummy streams the later odd
# places.  Whatever took queues before (the null stream) is idle while the pipeline runs.
_PLACEHOLDERS = []          # the idle streams (torch takes streams from a pool and never destroys them; kept for the record)
_SETS = {}                  # (device, n_side, n_main, n_extra) -> the set made for it: see placed_streams


def placement_order(n_side: int, n_main: int, n_extra: int):
    """The order of first use: a list of ("side", i) / ("main", i) / ("extra", i) / ("dummy", None).  With more than three
    main streams there is no pipe to spare: plain order."""
    if n_main > 3 or os.environ.get("SAD_NO_STREAM_PLACEMENT"):
        return ([("side", i) for i in range(n_side)] + [("main", i) for i in range(n_main)] +
                [("extra", i) for i in range(n_extra)])
    if n_main == 3:
        # three main streams take three pipes (places 1, 2, 3); everything else shares the fourth (places 0, 4, 8, ...)
        rest = [("side", i) for i in range(n_side)] + [("extra", i) for i in range(n_extra)]
        order, mains, p = [], [("main", i) for i in range(3)], 0
        while rest or mains:
            if p % 4 == 0:
                order.append(rest.pop(0) if rest else ("dummy", None))
            elif p in (1, 2, 3):
                order.append(mains.pop(0))
            else:
                order.append(("dummy", None))
            p += 1
        while order and order[-1][0] == "dummy":
            order.pop()
        return order
    even = [("side", i) for i in range(n_side)] + [("extra", i) for i in range(n_extra)]
    odd = [("main", i) for i in range(n_main)]
    order = []
    while even or odd:
        if even:
            order.append(even.pop(0))
        elif odd:
            order.append(("dummy", None))
        if odd:
            order.append(odd.pop(0))
        elif even:
            order.append(("dummy", None))
    while order and order[-1][0] == "dummy":
        order.pop()
    while order and order[0][0] == "dummy":
        order.pop(0)
    return order


def placed_streams(device, n_side: int, n_main: int, n_extra: int = 0):
    """(sampling streams, main streams, extra streams) of one pipeline on ``device``, each touched once (an event record and
    a wait: the stream has its hardware queue afterwards) in ``placement_order``.
    ONE set per device and shape: a second call with the same counts returns the same streams.  Queue numbers only mean
    something for the first streams a process maps (torch hands out 32 pool streams round-robin, the device serves 24 queues
    at full speed), so a second set could be neither placed nor given queues of its own; detectors that share a set are
    ordered against each other on it, which is what ``SADDetector(streams=...)`` asks a process with several detectors to do
    anyway.  ``SAD_NEW_STREAMS_PER_DETECTOR=1``: a fresh set per call (the behaviour before; for experiments)."""
    import torch
    key = (str(torch.device(device)), n_side, n_main, n_extra)
    if key in _SETS and not os.environ.get("SAD_NEW_STREAMS_PER_DETECTOR"):
        side, main, extra = _SETS[key]
        return list(side), list(main), list(extra)
    out = {"side": [None] * n_side, "main": [None] * n_main, "extra": [None] * n_extra}
    for kind, i in placement_order(n_side, n_main, n_extra):
        s = torch.cuda.Stream(device=device)
        if kind == "dummy":
            _PLACEHOLDERS.append(s)
        else:
            out[kind][i] = s
        e = torch.cuda.Event()
        e.record(s)
        s.synchronize()
    _SETS[key] = (out["side"], out["main"], out["extra"])
    return list(out["side"]), list(out["main"]), list(out["extra"])
