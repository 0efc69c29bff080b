"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch over the timed
dispatches).  Usage: python tools/pmc_summary.py gpurun_out/pmc > profiles/<name>.txt"""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        key = (short, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for key, cs in agg.items():
    rows.append((key, {k: sum(v) / len(v) for k, v in cs.items()}, len(next(iter(cs.values())))))
rows.sort(key=lambda t: -t[1].get("SQ_BUSY_CYCLES", t[1].get("FETCH_SIZE", 0)))
for key, c, n in rows:
    if not any(k in key[0] for k in ("mlp_chain", "mlp_multi", "mlp_reg", "mlp_coop", "mlp_layer", "mlp_bf16", "bf16_rows", "rowscan", "ball_query", "grid_query", "fps_", "group")):
        continue
    print(f"{key[0]} grid={key[1]} lds={key[2]} dispatches={n}")
    for k in sorted(c):
        print(f"    {k:34s} {c[k]:.4g}")
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 256 CUs x 4 SIMDs
        dur = c["GRBM_GUI_ACTIVE"] / 8.0
        print(f"    {'derived: MFMA pipe busy fraction':34s} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * dur):.3f}")
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        print(f"    {'derived: HBM MB (2*FETCH + WRITE)':34s} {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) / 1024.0:.1f}")
