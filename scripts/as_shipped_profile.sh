cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_asshipped -- python3 dex-nerf_amd/train_dexnerf.py --iters 600 --size 64 --views 8 --num-random-rays 1024 --layers 4 --width 128 --num-fine 64 --validate-every 0 --quiet --precision bf16 --no-hip-graph > gpurun_out/r04_asshipped.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/prof_r04_asshipped/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=0
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v2=v[len(v)//4:]   # skip warm-up
    print(f"{k:70s} n={len(v):5d} avg={sum(v2)/len(v2)/1000:8.2f} us  per-iter={sum(v2)/len(v2)*len(v)/600/1000:8.2f} us")
    tot+=sum(v2)/len(v2)*len(v)/600/1000
print("sum per iteration", tot)
PY
