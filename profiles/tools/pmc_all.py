import csv, sys, collections, glob, re
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+(<[0-9, ]+>)?)", row["Kernel_Name"])
        if m and "conv" in m.group(1):
            acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[m.group(1)]["_dur_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(d["_dur_ns"]))
