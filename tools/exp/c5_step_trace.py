"""One training step of bench.py --config c5 as a launch list: kernel, grid, duration, gap to the previous launch.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config c5 --steps 4 --warmup 2 --no-cpu-baseline
    python3 tools/exp/c5_step_trace.py DIR  > step.txt

The step is cut out of the trace as the launches between the last two `features_kernel` dispatches (first kernel of
arreau_train_forward behind the graph kernels)."""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "prep_kernel" in r["Kernel_Name"]]
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if "features_kernel" in r["Kernel_Name"]]
which = sys.argv[2] if len(sys.argv) > 2 else "fb"   # fb: a forward + backward step of the device-only loop; full: a step with its optimizer part
a, b = (marks[-3], marks[-2]) if which == "fb" else (marks[4], marks[5])
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
tot = 0.0
print("launches %d, wall %.1f us" % (len(step), (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
agg = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|arreau_sgemm_detail::|void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    grid = "%sx%sx%s/%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]),
                            int(r["Grid_Size_Z"]) // int(r["Workgroup_Size_Z"]), r["Workgroup_Size_X"])
    print("%9.1f  %-60s %-16s %8.1f us  gap %6.1f" % ((s - t0) / 1e3, name, grid, (e - s) / 1e3, (s - prev_end) / 1e3))
    tot += (e - s) / 1e3
    k = agg.setdefault(name, [0, 0.0])
    k[0] += 1
    k[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
print("kernel time %.1f us" % tot)
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %4d %9.1f us" % (name, n, t))
