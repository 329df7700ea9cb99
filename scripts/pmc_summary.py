"""Mean counter values per dispatch of one kernel from rocprofv3 --pmc CSVs (scripts/pmc_passes.sh)."""
import collections, csv, glob, sys
root, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{root}/pmc_*/*counter_collection.csv") + glob.glob(f"{root}/pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and r["Grid_Size"] == sys.argv[3]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:36s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
