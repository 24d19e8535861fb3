"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch of each kernel
over the last N dispatches.  usage: pmc_summary.py <dir> [N]"""
import csv, glob, sys, collections
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
    per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(per):
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    print(k, {c: round(sum(v[-N:]) / len(v[-N:]) / 1e6, 3) for c, v in sorted(per[k].items())}, "(millions)")
