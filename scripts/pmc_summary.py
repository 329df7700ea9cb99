"""Mean counter values per dispatch of the kernels matching a name, from rocprofv3 --pmc CSVs
(scripts/pmc_kernel.sh / pmc_passes.sh):  python scripts/pmc_summary.py <dir> <kernel substring> [grid size]"""
import collections, csv, glob, sys
root, kern = sys.argv[1], sys.argv[2]
grid = sys.argv[3] if len(sys.argv) > 3 else None
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f"{root}/pmc_*/*counter_collection.csv") + glob.glob(f"{root}/pmc_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:36s} mean {sum(v)/len(v):16.1f}  (n={len(v)})")
