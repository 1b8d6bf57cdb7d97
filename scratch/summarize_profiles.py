"""Turn a gpurun_out/<tag>/ visit (scratch/gpu_round.sh) into the committed summaries under profiles/."""
import glob, json, os, shutil, sys
import pandas as pd

tag = sys.argv[1]; out_tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
ks = max(glob.glob(os.path.join(src, "prof", "*", "*_kernel_stats.csv")), key=os.path.getmtime)       # a tag may have been visited twice
shutil.copy(ks, "profiles/%s_rocprofv3_kernel_stats.csv" % out_tag)
stats = pd.read_csv(ks)
rows = {}
for nm in ("fetch", "write", "sq"):
    f = glob.glob(os.path.join(src, "pmc_%s" % nm, "*", "*_counter_collection.csv"))
    if not f: continue
    d = pd.read_csv(max(f, key=os.path.getmtime))
    d = d[d["Kernel_Name"].str.contains("qrgpu::")]
    g = d.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].mean()
    for (k, c), v in g.items():
        if "qr_mpc_kernel<4, true, 0>" in k or "qr_mpc_kernelILi4ELb1ELi0" in k: key = "qr_mpc_kernel"          # the main pass, not the rescue launch
        elif "qr_wbc_kernel" in k: key = "qr_wbc_kernel"
        else: continue
        rows.setdefault(key, {})[c] = float(v)
traffic = {}
lines = ["# rocprofv3 summary (%s): `python3 bench.py` = 1024 A1 robots, h=10, full tick, MI355X" % out_tag, "",
         "Kernel durations (`--kernel-trace --stats`, file %s_rocprofv3_kernel_stats.csv):" % out_tag, "",
         "| kernel | calls | avg ns | min ns | max ns |", "|---|---:|---:|---:|---:|"]
for _, r in stats.iterrows():
    if "qrgpu::" in r["Name"]:
        nm = r["Name"]
        base = [k for k in ("qr_mpc_kernel<4, true, 0>", "qr_mpc_kernel<4, true, 1>", "qr_mpc_kernel<9, true, 0>", "qr_mpc_kernel<9, false, 0>", "qr_mpc_kernel", "qr_wbc_kernel",
                            "qr_frontend_kernel", "qr_vmc_kernel", "qr_lpt_order_kernel", "qr_selftest_kernel") if k in nm]
        label = base[0] if base else nm[:40]
        if label == "qr_mpc_kernel<4, true, 1>": label += " (rescue launch, carries the longest-first sort)"
        lines.append("| %s | %d | %.0f | %d | %d |" % (label, r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
lines += ["", "PMC (separate passes, mean per launch).  FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md §HBM the read side",
          "is doubled (gfx950 tallies 128-B requests at 64 B; exact only for wide coalesced streams, an upper bound here):", "",
          "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes/launch (2*F+W) | algorithmic bytes/launch | ratio |", "|---|---:|---:|---:|---:|---:|"]
alg = {"qr_mpc_kernel": 1024 * (28 + 160 + 12 + 24 + 1) * 4, "qr_wbc_kernel": 1024 * (37 + 67 + 3 + 3 + 12 + 1) * 4}
for k, v in rows.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        traffic[k] = b
        lines.append("| %s | %.1f | %.1f | %.3g | %d | %.1f |" % (k, v["FETCH_SIZE"], v["WRITE_SIZE"], b, alg[k], b / alg[k]))
lines += ["", "SQ counters (mean per launch):", "", "| kernel | " + " | ".join(sorted({c for v in rows.values() for c in v if c.startswith("SQ_")})) + " |"]
cols = sorted({c for v in rows.values() for c in v if c.startswith("SQ_")})
lines.append("|---|" + "---:|" * len(cols))
for k, v in rows.items():
    lines.append("| %s | " % k + " | ".join("%.3g" % v.get(c, float("nan")) for c in cols) + " |")
open("profiles/%s_rocprofv3_summary.md" % out_tag, "w").write("\n".join(lines) + "\n")
json.dump(traffic, open("profiles/traffic_latest.json", "w"))
for f in ("bench.json", "pytest_gpu.log", "smoke.log"):
    p = os.path.join(src, f)
    if os.path.exists(p): shutil.copy(p, "profiles/%s_%s" % (out_tag, f))
print("\n".join(lines))
