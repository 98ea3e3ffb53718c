export TMPDIR=/tmp
mkdir -p gpurun_out/assign6
for Z in 1 0; do
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/assign6/ea$Z -o p -- scripts/exp/assign_sorted_probe 108 $Z > /dev/null 2>&1
python3 - $Z <<'PY'
import csv, glob, collections, sys
z = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/assign6/ea{z}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "apply_" in r["Kernel_Name"] and r["Grid_Size"] == str(256*16*256):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    n = sum(v["TCC_EA0_RDREQ_sum"]) / len(v["TCC_EA0_RDREQ_sum"]); n32 = sum(v["TCC_EA0_RDREQ_32B_sum"]) / len(v["TCC_EA0_RDREQ_32B_sum"])
    print("numbering", "z-fastest" if z == "1" else "x-fastest", k, "EA read requests", round(n / 1e6, 2), "M, of them 32 B", round(n32 / 1e6, 2), "M ->", round((n32 * 32 + (n - n32) * 64) / 1e6, 1), "MB")
PY
rm -rf gpurun_out/assign6/ea$Z
done
