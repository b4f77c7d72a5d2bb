#!/bin/bash
# start / end of the upload pieces and the inflate kernels of one device-front load (rocprofv3 kernel + memory-copy trace); GPU box
export TMPDIR=/tmp
[ -f /tmp/bench.bam ] || python tools/bench_bam.py --reads 100000 --write-only /tmp/bench.bam
out=gpurun_out/timeline
mkdir -p $out
timeout -k 10 180 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o tl -- tests/cpp/bam_test /tmp/bench.bam 0 2147483647 --device --time --threads 16 > $out/stdout.txt 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/timeline/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob("gpurun_out/timeline/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
t0 = rows[0][0]
for s, e, n in rows:
    if e - s > 200000:
        print("%9.3f %9.3f  %8.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
