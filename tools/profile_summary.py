"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a small text table (for profiles/)."""
import csv
import glob
import sys


def main(d, steps):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s  (kernel time per bench step: %.1f us over %d steps)" % (f, tot / 1e3 / steps, steps))
    print("%-100s %7s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows:
        print("%-100s %7s %12.1f %10.2f %6.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                    float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
