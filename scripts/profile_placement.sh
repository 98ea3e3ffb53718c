#!/bin/bash
# Counters of the same streaming-write sweep over a fast and a slow virtual range (scripts/exp/place_pmc.hip).
# Usage (GPU box, repository root): bash scripts/profile_placement.sh <tag>   -> gpurun_out/<tag>_place_pmc.txt
tag=${1:-r02}
out=gpurun_out/${tag}_place_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
BIN=$R/scripts/exp/place_pmc
$BIN > $R/$out/plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $R/$out/trace -o trace -- $BIN > $R/$out/trace.log 2>&1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" \
           "GRBM_EA_BUSY GRBM_TC_BUSY" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/$out/pmc$i -o pmc -- $BIN > $R/$out/pmc$i.log 2>&1
done
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "wr_fast_range" in k or "wr_slow_range" in k:
            acc[row["Counter_Name"]]["fast" if "fast" in k else "slow"].append(float(row["Counter_Value"]))
with open(out + ".txt", "w") as fh:
    fh.write(open(out + "/plain.txt").read())
    for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
        fh.write(open(f).read())
    fh.write("\ncounter: mean per dispatch over the fast range | over the slow range | slow/fast\n")
    for name in sorted(acc):
        a, b = acc[name]["fast"], acc[name]["slow"]
        if a and b:
            ma, mb = sum(a) / len(a), sum(b) / len(b)
            fh.write(f"{name:48s} {ma:16.1f} {mb:16.1f} {mb / ma if ma else float('nan'):8.3f}\n")
print(open(out + ".txt").read())
PY
