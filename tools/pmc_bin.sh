#!/bin/bash
# PMC passes (WRITE_SIZE, FETCH_SIZE; separate runs, no tracing) of a standalone microbenchmark binary:
#   tools/pmc_bin.sh <binary path relative to the repo> <tag>      -> gpurun_out/<tag>_{plain.log,pmc.txt}
# FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md "HBM"); unit KiB.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/$1; T=$2
mkdir -p $R/gpurun_out
$B > $R/gpurun_out/${T}_plain.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_write -- $B > $R/gpurun_out/${T}_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_pmc_fetch -- $B > $R/gpurun_out/${T}_pmc_fetch.log 2>&1
python3 - "$R/gpurun_out/${T}_pmc_write" "$R/gpurun_out/${T}_pmc_fetch" > $R/gpurun_out/${T}_pmc.txt <<'PY'
import csv, glob, sys
def load(d):
    rows = []
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:60], float(r["Counter_Value"])))
    return sorted(rows)
w, f = load(sys.argv[1]), load(sys.argv[2])
print("# dispatch kernel WRITE_SIZE[MiB] FETCH_SIZE_x2[MiB]")
for (i, k, wv), (_, _, fv) in zip(w, f):
    print(f"{i:3d} {k:60s} {wv / 1024:10.1f} {2 * fv / 1024:10.1f}")
PY
cat $R/gpurun_out/${T}_pmc.txt
