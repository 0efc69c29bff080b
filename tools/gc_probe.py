"""Does Python's cyclic garbage collector stall the pipelined loop?  Runs bench.py in this process with gc.callbacks logging every
collection (generation, duration, time), optionally with the collector disabled, and prints a summary to stderr after bench.py's JSON line.
usage: python tools/gc_probe.py [--gc-off] -- <bench.py arguments>"""
import gc, os, runpy, sys, time
args = sys.argv[1:]
gc_off = "--gc-off" in args
if gc_off:
    args.remove("--gc-off")
if "--" in args:
    args = args[args.index("--") + 1:]
events = []                      # (t_start, generation, seconds, collected)
_t = {}
def cb(phase, info):
    if phase == "start":
        _t["s"] = time.perf_counter()
    else:
        events.append((_t["s"], info["generation"], time.perf_counter() - _t["s"], info["collected"]))
gc.callbacks.append(cb)
if gc_off:
    gc.disable()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, "bench.py")] + args
t0 = time.perf_counter()
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
by = {0: [], 1: [], 2: []}
for t, g, dt, n in events:
    by[g].append(dt)
print(f"[gc_probe] automatic collection {'OFF' if gc_off else 'on'} (thresholds {gc.get_threshold()}), process ran {time.perf_counter() - t0:.1f} s", file=sys.stderr)
for g in (0, 1, 2):
    v = sorted(by[g])
    if v:
        print(f"[gc_probe] generation {g}: {len(v)} collections, total {sum(v) * 1e3:.1f} ms, median {v[len(v) // 2] * 1e3:.2f} ms, max {v[-1] * 1e3:.2f} ms", file=sys.stderr)
    else:
        print(f"[gc_probe] generation {g}: none", file=sys.stderr)
long = [(t - t0, g, dt) for t, g, dt, n in events if dt > 2e-3]
print("[gc_probe] collections over 2 ms (s since start, generation, ms): " + ", ".join(f"({t:.2f}, g{g}, {dt * 1e3:.1f})" for t, g, dt in long[-12:]), file=sys.stderr)
