"""Collect scripts/pmc_train_kernels.sh's passes into gpurun_out/pmc_train_<mode>.json: per training kernel, the counters averaged
over its fine-network launches (the dispatches within 30 % of that kernel's longest launch in the same pass's kernel trace)."""
import csv, glob, json, sys
from collections import defaultdict

mode = sys.argv[1]
import os
KERNELS = tuple(os.environ["PMC_KERNELS"].split(";")) if os.environ.get("PMC_KERNELS") else \
    ("mlp_forward_kernel", "mlp_backward_kernel", "mlp_forward48_kernel", "mlp_backward48_kernel", "weight_grad_batch_kernel")
out = {"mode": mode, "kernels": {}}
for name in ("fetch", "write", "mfma", "sqA", "sqB"):
    cc = glob.glob(f"gpurun_out/pmc_train_{mode}_{name}/*/*counter_collection.csv")
    tr = glob.glob(f"gpurun_out/pmc_train_{mode}_{name}/*/*kernel_trace.csv")
    if not cc or not tr:
        continue
    dur = {}
    for r in csv.DictReader(open(tr[0])):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    per = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(cc[0])):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                per[k][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, disp in per.items():
        longest = max(dur.get(d, 0.0) for d in disp)
        fine = [d for d in disp if dur.get(d, 0.0) > 0.7 * longest]
        rec = out["kernels"].setdefault(k, {})
        for c in next(iter(disp.values())):
            rec[c] = sum(disp[d].get(c, 0.0) for d in fine) / len(fine)
        rec[f"kernel_ms_{name}"] = sum(dur[d] for d in fine) / len(fine)
        rec["launches_averaged"] = len(fine)
for k, rec in out["kernels"].items():
    d = rec["derived"] = {}
    if "GRBM_GUI_ACTIVE" in rec:
        cycles = rec["GRBM_GUI_ACTIVE"] / 8.0
        d["gpu_cycles"] = cycles
        d["matrix_pipe_busy_frac"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)
    if "SQ_WAVE_CYCLES" in rec:
        w = rec["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            d[c + "/wave_cycles"] = rec[c] / w
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        # MI355X_MICROARCH.md: FETCH_SIZE in KiB, one half of the bytes of wide coalesced reads on gfx950 (x 2); WRITE_SIZE in KiB
        d["hbm_read_bytes"] = 2.0 * rec["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes"] = rec["WRITE_SIZE"] * 1024.0
        ms = rec.get("kernel_ms_mfma") or rec.get("kernel_ms_fetch")
        d["hbm_tb_per_s"] = (d["hbm_read_bytes"] + d["hbm_write_bytes"]) / (ms * 1e-3) / 1e12
        d["hbm_frac_of_8tb"] = d["hbm_tb_per_s"] / 8.0
        d["hbm_frac_of_measured_copy_6p29tb"] = d["hbm_tb_per_s"] / 6.29
    if "SQ_LDS_IDX_ACTIVE" in rec and rec["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict/lds_idx_active"] = rec["SQ_LDS_BANK_CONFLICT"] / rec["SQ_LDS_IDX_ACTIVE"]
json.dump(out, open(f"gpurun_out/pmc_train_{mode}.json", "w"), indent=1)
for k, rec in out["kernels"].items():
    print(k, json.dumps(rec["derived"]), {c: rec[c] for c in rec if c.startswith("kernel_ms")})
