import csv, sys, collections, glob
# usage: pmc_mean.py <dir> <counter>
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == sys.argv[2]:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if k.startswith("k_rep"):
        print(sys.argv[2], k, len(v), sum(v) / len(v))
