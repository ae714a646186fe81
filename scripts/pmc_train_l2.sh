#!/bin/bash
# Developer tool (run on the GPU box): L2 hit rate and HBM traffic of the training kernels, default (non-temporal) stores vs
# plain stores (exp_libs/libPLAINST.so built with -DDN_STORE_POLICY_ID=0).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in nt plain; do
  if [ $tag = plain ]; then export DEXNERF_HIP_LIB=exp_libs/libPLAINST.so; else unset DEXNERF_HIP_LIB; fi
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_train_${tag}_l2 -- python3 scripts/train_kernels_time.py > gpurun_out/pmc_train_${tag}_l2.log 2>&1
  # FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit one pass (4 slots): one run each
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_train_${tag}_fetch -- python3 scripts/train_kernels_time.py > gpurun_out/pmc_train_${tag}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_train_${tag}_write -- python3 scripts/train_kernels_time.py > gpurun_out/pmc_train_${tag}_write.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("nt", "plain"):
    for st in ("l2", "fetch", "write"):
        f = glob.glob(f"gpurun_out/pmc_train_{tag}_{st}/*/*counter_collection.csv")[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "forward_save" if "mlp_forward" in k else ("backward" if "mlp_backward" in k else ("weight_grad" if "weight_grad" in k else None))
            if name:
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (name, ctr), vals in agg.items():
            out.setdefault(tag, {}).setdefault(name, {})[ctr] = max(vals)   # the 786 k-point launch
for tag, d in out.items():
    for name, c in d.items():
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_train_l2.json", "w"), indent=1)
PY
