import csv, sys, collections, glob, re
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
pat = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
        if m and pat in m.group(1):
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 2) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
