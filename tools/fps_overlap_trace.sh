# kernel trace of tools/fps_overlap_probe.py restricted to one stage: bash tools/fps_overlap_trace.sh sa3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fot; rm -rf $out; mkdir -p $out
SAD_PROBE_STAGES=$1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/fps_overlap_probe.py > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
grep -E "alone" $out/run.log
python3 - $out <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def nm(n):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", n); return m.group(1) if m else n[:30]
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("mlp_", "fps_cell"))]
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  q{r.get('Queue_Id','?'):>3s}  {nm(r['Kernel_Name'])}")
PY
