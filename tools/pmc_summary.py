"""Per-kernel summary of a rocprofv3 --pmc CSV directory (first dispatch of each kernel/grid)."""
import collections
import csv
import glob
import sys


def main(d, pat):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"], r["Grid_Size"], r["Dispatch_Id"])
        a = agg.setdefault(key, {"_t": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    seen = set()
    for (k, g, _), c in agg.items():
        if (k, g) in seen or pat not in k:
            continue
        seen.add((k, g))
        name = k[k.find("::", 5) + 2:][:70]
        print("%-70s grid %9s  %8.1f us  " % (name, g, c["_t"]) + "  ".join("%s=%.4g" % (n, v) for n, v in c.items() if n != "_t"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
