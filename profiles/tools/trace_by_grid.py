"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid size): calls, total and mean duration."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[key][0] += 1; acc[key][1] += d
tot = sum(v[1] for v in acc.values())
print("total kernel time %.1f ms (/%g = %.1f ms)" % (tot / 1e3, div, tot / 1e3 / div))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 70]:
    print("%-44s wg=%6d y=%3d z=%3d  n=%5.0f  mean %8.1f us  total %8.2f ms  %5.2f%%" % (k[0][:44], k[1], k[2], k[3], v[0] / div, v[1] / v[0], v[1] / 1e3 / div, 100 * v[1] / tot))
