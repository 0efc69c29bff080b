"""Summary table of bench.py JSON lines collected under gpurun_out/<dir>/*.json (one gpurun call = one box), written for profiles/.
usage: python tools/ab_summary.py PATH[:note] ...   PATH = a directory (every *.json in it) or one file - name the file when the directory
was reused by earlier calls (gpurun merges into existing directories).  A line without bf16_leg / configs4_leg is a standalone run."""
import glob, json, os, sys
def row(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    name = os.path.basename(path)[:-5]
    if "bf16_leg" in d:
        b, c = d["bf16_leg"], d["configs4_leg"]
        return (f"  {name:10s} f32 {d['value']:8.0f} ({d['ms_per_step']:.3f} ms)  | bf16_leg {b['value']:8.0f} ({b['ms_per_step']:.3f} ms, p50 {b['step_ms_p50']:.3f}, "
                f"MLP {b['mlp_ms_per_step']:.3f}, frac {b['roofline']['frac']:.3f})  | configs4_leg {c['value']:7.0f} ({c['ms_per_step']:.3f} ms, p50 {c['step_ms_p50']:.3f}, "
                f"MLP {c['mlp_ms_per_step']:.3f}, frac {c['roofline']['frac']:.3f})")
    wl = d["config"]["workload"].split(":")[0]
    rf = d.get("roofline")          # (absent with --no-launch-timing)
    tail = f"MLP {rf['ms_per_step']:.3f}, frac {rf['frac']:.3f}" if rf else "no per-launch timing"
    fps = [k for k in d.get("kernels", []) if "fps" in k.get("kernel", "")]
    if fps:
        tail += f", FPS {fps[0].get('cu_ms_per_step')} CU.ms"
    opts = d["config"].get("opts") or []
    return (f"  {name:10s} standalone {wl} {d['dtype']}{' ' + ','.join(opts) if opts else ''}: {d['value']:7.0f} scenes/s ({d['ms_per_step']:.3f} ms, p50 {d['step_ms']['p50']:.3f}, "
            f"max {d['step_ms']['max']:.3f}, steps over 2 x p50: {d['step_ms']['over_2x_p50']}; {tail})")
for arg in sys.argv[1:]:
    d, _, note = arg.partition(":")
    print(f"{d}  {note}")
    for f in ([d] if os.path.isfile(d) else sorted(glob.glob(os.path.join(d, "*.json")))):
        if os.path.basename(f).startswith("."):
            continue
        print(row(f))
