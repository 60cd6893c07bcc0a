#!/bin/bash
# LDS bank-conflict share of the dominant kernels: SQ_LDS_BANK_CONFLICT (extra cycles) / SQ_LDS_IDX_ACTIVE (all LDS-array cycles), separate --pmc runs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for W in ${PMC_LDS_WORKLOADS:-c2c_2p20_b512 c2c_1024_b65536 c2c_2p15_b16384 c2c_2p13_b65536 r2c_2p12_b262144}; do
  for C in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/lds_${C}_$W -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/lds_${C}_$W.log 2>&1
  done
  python3 - $R/gpurun_out $W <<'PY'
import sys, glob, csv, collections
root, w = sys.argv[1], sys.argv[2]
tot = {}
for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    acc = collections.defaultdict(float)
    for f in glob.glob(f"{root}/lds_{c}_{w}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"][:60]] += float(r["Counter_Value"])
    tot[c] = acc
for k in tot["SQ_LDS_IDX_ACTIVE"]:
    a, b = tot["SQ_LDS_IDX_ACTIVE"][k], tot["SQ_LDS_BANK_CONFLICT"].get(k, 0.0)
    if a > 0 and "fill_random" not in k:
        print(f"{w}: {k}: conflict cycles / LDS-active cycles = {b:.3g} / {a:.3g} = {b / a:.3f}")
PY
done
