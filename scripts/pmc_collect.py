"""Collect the per-pass counter CSVs of scripts/pmc_fine_net.sh into gpurun_out/pmc_<tag>.json (developer tool, GPU box).

Per pass: the LAST mlp_forward* dispatch of the run (the timed repetitions of scripts/quick_time.py are identical launches;
the first one includes cold caches).  Derived figures follow MI355X_MICROARCH.md: FETCH_SIZE is reported in KiB and reads
one half of the bytes of wide coalesced reads on gfx950 (x2), WRITE_SIZE in KiB is exact for 16-byte stores; the SQ_* wave
counters count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs, GRBM_GUI_ACTIVE is
the sum over the 8 XCDs."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "fine_net"
out = {"tag": tag, "passes": {}}
for name in ("fetch", "write", "l2", "mfma", "sqA", "sqB"):
    files = glob.glob(f"gpurun_out/pmc_{tag}_{name}/*/*counter_collection.csv")
    if not files:
        out["passes"][name] = None
        continue
    per_dispatch = {}
    kernel = None
    for r in csv.DictReader(open(files[0])):
        if "mlp_forward" not in r["Kernel_Name"]:
            continue
        kernel = r["Kernel_Name"]
        per_dispatch.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if not per_dispatch:
        out["passes"][name] = None
        continue
    last = per_dispatch[max(per_dispatch)]
    out["kernel"] = kernel
    out["passes"][name] = last
    # kernel duration of the same dispatch from the kernel trace of this pass
    tr = glob.glob(f"gpurun_out/pmc_{tag}_{name}/*/*kernel_trace.csv")
    if tr:
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in csv.DictReader(open(tr[0]))
                if "mlp_forward" in r["Kernel_Name"]]
        if durs:
            last["kernel_ms"] = durs[-1]

p = out["passes"]
d = out["derived"] = {}
if p.get("fetch") and p.get("write"):
    fetch_b = p["fetch"]["FETCH_SIZE"] * 1024.0
    write_b = p["write"]["WRITE_SIZE"] * 1024.0
    d["hbm_read_bytes"] = 2.0 * fetch_b
    d["hbm_write_bytes"] = write_b
    d["hbm_bytes_per_launch"] = 2.0 * fetch_b + write_b
    points = int(os.environ.get("PMC_POINTS", 160000 * 192))
    d["algorithmic_bytes_per_launch"] = points * 20 + 160000 * 44 if "PMC_POINTS" not in os.environ else points * 20
if p.get("l2"):
    h, m = p["l2"]["TCC_HIT_sum"], p["l2"]["TCC_MISS_sum"]
    d["l2_hit_rate"] = h / (h + m)
if p.get("mfma"):
    busy, gui = p["mfma"]["SQ_VALU_MFMA_BUSY_CYCLES"], p["mfma"]["GRBM_GUI_ACTIVE"]
    cycles = gui / 8.0
    d["gpu_cycles"] = cycles
    d["matrix_pipe_busy_frac"] = busy / (cycles * 1024.0)
    if "kernel_ms" in p["mfma"]:
        d["effective_clock_ghz"] = cycles / (p["mfma"]["kernel_ms"] * 1e-3) / 1e9
if p.get("sqA"):
    a = p["sqA"]
    wc = a["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS",
              "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_MISC"):
        if k in a:
            d[k + "/WAVE_CYCLES"] = a[k] / wc
if p.get("sqB") and p.get("sqA"):
    b = p["sqB"]
    wc = p["sqA"]["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA"):
        if k in b:
            d[k + "/WAVE_CYCLES"] = b[k] / wc
# which kernel sources the record belongs to (bench.py refuses a record collected on other sources)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out["source_sha16"] = bench.kernel_source_sha16()
    prec = "fp16" if ", 2, 8, 16u" in out.get("kernel", "") else "bf16"
    out["isa_sha16"] = bench.kernel_isa_sha16(prec)   # the kernel binary's identity (None without the LLVM tools)
except Exception as exc:  # noqa: BLE001
    out["source_sha16"] = f"unavailable: {exc}"
json.dump(out, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
