"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv into a short
markdown table (kept under profiles/)."""
import csv
import sys


def main(path, steps, title):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {title}\n")
    print(f"source: `{path}` — {steps} steps (warm-up included), total GPU kernel time "
          f"{tot / 1e6:.2f} ms = {tot / 1e6 / steps:.3f} ms/step\n")
    print("| kernel | calls | calls/step | avg µs | total ms | % |")
    print("|---|---:|---:|---:|---:|---:|")
    for r in rows[:25]:
        name = r["Name"].replace("|", "/")
        name = name if len(name) <= 110 else name[:107] + "..."
        t = float(r["TotalDurationNs"])
        print(f"| `{name}` | {r['Calls']} | {int(r['Calls']) / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | "
              f"{t / 1e6:.2f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
