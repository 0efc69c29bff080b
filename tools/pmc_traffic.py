"""HBM traffic of the MLP dispatches of one step from the rocprofv3 --pmc passes of tools/pmc.sh
(FETCH_SIZE and WRITE_SIZE collected in separate passes).  usage: python tools/pmc_traffic.py gpurun_out/pmc profiles/r02_pmc_traffic.json
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB... (see `note`)."""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
def is_mlp(n):      # every MLP kernel of either arithmetic + the row-packing scans (not the one-off weight packing)
    return ("mlp_" in n or "bf16_rows_kernel" in n or "bf16_rows2_kernel" in n or "rowscan_" in n) and "pack_kernel" not in n
def total(passname, counter):
    tot, steps = 0.0, 0
    for f in glob.glob(os.path.join(root, passname, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(root, "pmc_" + passname, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            if "fps_cell" in n:     # fps_cell / fps_cell2 / fps_cellg / fps_cellg2: one per forward pass
                steps += 1
            if is_mlp(n):
                tot += float(r["Counter_Value"])
    return tot, steps
fetch, s1 = total("fetch", "FETCH_SIZE")
write, s2 = total("write", "WRITE_SIZE")
# FETCH_SIZE / WRITE_SIZE are reported in kilobytes; on gfx950 FETCH_SIZE counts 32-byte requests as 64-byte units
# halved, i.e. the raw value is doubled (MI355X_MICROARCH.md, HBM section) — same correction as in round 1
res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_r04.sh, parts pmc / bf16 / nus: bench.py --geometry-file "
                 "<geometry of the run> --main-streams 1 --steps 3 --warmup 1), summed over the MLP dispatches (every mlp_* / bf16_rows kernel + the row-packing scans) and divided by the forward passes of the run; FETCH_SIZE doubled "
                 "per the gfx950 correction of MI355X_MICROARCH.md (HBM section); gather-width reads uncalibrated",
       "forward_passes": s1,
       "fetch_size_raw_kb_per_step": fetch / max(1, s1),
       "fetch_bytes_per_step": int(2 * fetch * 1024 / max(1, s1)),
       "write_bytes_per_step": int(write * 1024 / max(1, s2)),
       "build": sys.argv[3] if len(sys.argv) > 3 else "r02"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
