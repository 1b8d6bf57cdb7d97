"""Turn a gpurun_out/<tag>/ visit (scratch/gpu_round4.sh) into the committed summaries under profiles/ (prefix r04_):
r04_rocprofv3_kernel_stats.csv, r04_pmc_summary.json (per kernel and launch: every counter's mean; hbm_bytes_per_launch = (2 x FETCH_SIZE +
WRITE_SIZE) KiB x 1024 -- FETCH_SIZE counts 64 B per 128-B request on gfx950, MI355X_MICROARCH.md; the commit the passes were taken at; the
recipe), r04_pmc_<first counter>_sample.csv."""
import glob, json, os, shutil, subprocess, sys
import pandas as pd
tag = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else "r04"
src = os.path.join("gpurun_out", tag)
ks = max(glob.glob(os.path.join(src, "prof", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(ks, "profiles/%s_rocprofv3_kernel_stats.csv" % pre)
NAMES = {"mpc_main": "qr_mpc_kernel", "wbc": "qr_wbc_kernel"}
def key(k):
    # (template arguments beyond the fourth -- MINW, H16 -- are at their defaults in these three)
    if "qr_mpc_kernel<2, false, false, 512, 0" in k or "qr_mpc_kernel<2, false, false, 512>" in k: return "mpc_main"
    if "qr_mpc_kernel<4, true, true, 256" in k: return "mpc_list"
    if "qr_mpc_kernel<2, true, false, 512, 0" in k or "qr_mpc_kernel<2, true, false, 512>" in k: return "mpc_planned"
    if "qr_wbc_kernel" in k: return "wbc"
    return None
summ = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    d = pd.read_csv(f)
    d["k"] = d["Kernel_Name"].map(key)
    d = d[d["k"].notna()]
    g = d.groupby(["k", "Counter_Name", "Dispatch_Id"])["Counter_Value"].sum().groupby(level=[0, 1]).mean()
    for (k, c), v in g.items(): summ.setdefault(k, {})[c] = float(v)
    c0 = d["Counter_Name"].iloc[0]
    d[d["k"] == "mpc_main"].head(40).drop(columns=["k"]).to_csv("profiles/%s_pmc_%s_sample.csv" % (pre, c0), index=False)
commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"]).decode().strip()
out = {"commit": commit,
       "recipe": "scratch/gpu_round4.sh: rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-side, one run per "
                 "counter group, serial tick (QRGPU_TICK_PIPELINE=0) with the planned launch forked by an event (QRGPU_LAB=1 QRGPU_PLANNED_FORK=1: counter collection runs one kernel at a time); 1024 robots per launch; mean over the launches of a run",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "counters": summ, "kernels": {}}
for k, v in summ.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e = {"hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, "read_bytes": 2 * v["FETCH_SIZE"] * 1024.0, "written_bytes": v["WRITE_SIZE"] * 1024.0}
        if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_INSTS_LDS", 0) > 0: e["lds_bank_conflict_cycles_per_lds_instruction"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_INSTS_LDS"]
        if "SQ_LDS_BANK_CONFLICT" in v and "SQ_LDS_IDX_ACTIVE" in v: e["lds_bank_conflict_share_of_lds_active_cycles"] = v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])
        out["kernels"][NAMES.get(k, k)] = e
json.dump(out, open("profiles/%s_pmc_summary.json" % pre, "w"), indent=1)
p = os.path.join(src, "prof_bench.json")
if os.path.exists(p): shutil.copy(p, "profiles/%s_bench_under_rocprofv3.json" % pre)
st = pd.read_csv(ks)
print(st[st["Name"].str.contains("qrgpu")][["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_string())
print(json.dumps(out["kernels"], indent=1))
m = summ.get("mpc_main", {})
if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
    print("mpc_main VALU wave-instr %.1f M, per SIMD-cycle %.3f; WAIT_ANY %.0f M of %.0f M wave quad-cycles" % (m["SQ_INSTS_VALU"] / 1e6, m["SQ_INSTS_VALU"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), m.get("SQ_WAIT_ANY", 0) / 1e6, m.get("SQ_WAVE_CYCLES", 0) / 1e6))

# the bench lines, logs and text outputs of the visit
import re
for f in sorted(glob.glob(os.path.join(src, "*.json"))):
    name = os.path.basename(f)
    if name == "prof_bench.json": continue
    txt = open(f).read().strip()
    if txt: open("profiles/%s_%s" % (pre, name), "w").write(txt.splitlines()[-1] + "\n")
for name in ("pytest_gpu.log", "smoke.log", "overlap_ab.txt", "overlap_timeline.txt"):
    f = os.path.join(src, name)
    if os.path.exists(f): shutil.copy(f, "profiles/%s_%s" % (pre, name))
for u, dst in (("ubench_cumask.txt", "cumask.txt"), ("ubench_pipes.txt", "pipes.txt")):
    f = os.path.join(src, u)
    if os.path.exists(f):
        txt = open(f).read()
        if u == "ubench_cumask.txt" and os.path.exists(os.path.join(src, "ubench_cumask2.txt")): txt += "\n--- cumask2: occupancy under CU-masked streams ---\n" + open(os.path.join(src, "ubench_cumask2.txt")).read()
        open("profiles/%s_%s" % (pre, dst), "w").write(txt)
