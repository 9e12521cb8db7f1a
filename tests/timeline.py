"""One step's kernels with start / end / length (us) and HIP queue from a `rocprofv3 --kernel-trace` CSV
(`*_kernel_trace.csv`): who runs beside whom on the library's two streams, where a stream waits.

  python3 tests/timeline.py gpurun_out/prof_train/train_kernel_trace.csv copy_segs_kernel 2     # training step (two markers per step)
  python3 tests/timeline.py gpurun_out/prof/bench_kernel_trace.csv corrector_update_kernel       # one PC step

The step printed is the one between the third- and second-last occurrence of the marker kernel (x stride)."""
import csv
import sys

path, marker = sys.argv[1], sys.argv[2]
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]][::stride]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print(f"step: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[a:b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:70]}")
