"""Turn a gpurun_out/<tag>/ visit (scratch/gpu_round2.sh) into the committed summaries under profiles/ (prefix r02_)."""
import glob, json, os, shutil, sys
import pandas as pd
tag = sys.argv[1]; pre = sys.argv[2] if len(sys.argv) > 2 else "r02"
src = os.path.join("gpurun_out", tag)
ks = max(glob.glob(os.path.join(src, "prof", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(ks, "profiles/%s_rocprofv3_kernel_stats.csv" % pre)
def key(k):
    if "qr_mpc_kernel<2, false, false, 512>" in k: return "mpc_main"
    if "qr_mpc_kernel<4, true, true, 256>" in k: return "mpc_list"
    if "qr_wbc_kernel" in k: return "wbc"
    return None
summ = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    d = pd.read_csv(f)
    d["k"] = d["Kernel_Name"].map(key)
    d = d[d["k"].notna()]
    # one row per (dispatch, counter); sum over the dimensions a counter is split into, then mean per launch
    g = d.groupby(["k", "Counter_Name", "Dispatch_Id"])["Counter_Value"].sum().groupby(level=[0, 1]).mean()
    for (k, c), v in g.items(): summ.setdefault(k, {})[c] = float(v)
    c0 = d["Counter_Name"].iloc[0]
    d[d["k"] == "mpc_main"].head(40).drop(columns=["k"]).to_csv("profiles/%s_pmc_%s_sample.csv" % (pre, c0), index=False)
json.dump(summ, open("profiles/%s_pmc_summary.json" % pre, "w"), indent=1)
for a, b in (("bench.json", "bench.json"), ("bench_config4_f32.json", "bench_config4_f32.json"), ("bench_config4_bf16x3.json", "bench_config4_bf16x3.json"),
             ("bench_8192.json", "bench_8192_robots.json"), ("bench_config1.json", "bench_config1_256_mpc_only.json"), ("prof_bench.json", "bench_under_rocprofv3.json"), ("pytest_gpu.log", "pytest_gpu.log"), ("smoke.log", "smoke.log")):
    p = os.path.join(src, a)
    if os.path.exists(p): shutil.copy(p, "profiles/%s_%s" % (pre, b))
st = pd.read_csv(ks)
print(st[st["Name"].str.contains("qrgpu")][["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_string())
m = summ.get("mpc_main", {}); w = summ.get("wbc", {})
for nm, v in (("mpc_main", m), ("wbc", w)):
    if "FETCH_SIZE" in v: print(nm, "HBM KiB per launch: read 2*FETCH = %.0f, written %.0f" % (2 * v["FETCH_SIZE"], v["WRITE_SIZE"]))
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        print(nm, "VALU wave-instr %.1f M, per SIMD-cycle %.3f (1024 SIMDs, %d cycles)" % (v["SQ_INSTS_VALU"] / 1e6, v["SQ_INSTS_VALU"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), v["GRBM_GUI_ACTIVE"] / 8))   # (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
print(json.dumps(summ, indent=1)[:3500])
