#!/bin/bash
# kernel timeline of a short sampling run with two stream groups (timestamps of every launch)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 $R/bench.py --steps 1 --warmup 0 --T 120 --no-cpu-baseline --no-policy-compare --no-eval --profile-stride 0 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
ls -la $O
python3 - <<PY
import csv, glob
f = glob.glob('$O/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Queue_Id', r.get('Stream_Id',''))) for r in rows)
# keep the last 40 % (steady part)
ev = ev[int(len(ev) * 0.5):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n, q in ev: pts += [(s, 1), (e, -1)]
pts.sort()
busy = {0: 0, 1: 0, 2: 0}
lvl, last = 0, pts[0][0]
for t, d in pts:
  busy[min(lvl, 2)] += t - last; last = t; lvl += d
tot = t1 - t0
print('wall %.2f ms; no kernel %.1f %%, one kernel %.1f %%, two or more %.1f %%' % (tot / 1e6, 100 * busy[0] / tot, 100 * busy[1] / tot, 100 * busy[2] / tot))
with open('$O/tail.txt', 'w') as o:
  for s, e, n, q in ev[:400]: o.write('%10.1f %8.1f %s q=%s\n' % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
PY
rm -f $O/*kernel_trace.csv
