"""Sum rocprofv3 PMC counters per kernel name from a counter_collection CSV: python scripts/pmc_table.py <csv> <name substring>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for r in rows:
    if sub in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:36s} {acc[k] / n[k]:16.0f}   (avg of {n[k]} dispatches)")
