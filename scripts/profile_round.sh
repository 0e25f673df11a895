#!/bin/bash
# Round profile: kernel trace + stats of the default bench command, then HBM traffic counters
# (separate --pmc passes, no trace domains mixed in).  Summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r02}
DATA=${2:-gaussian}
mkdir -p gpurun_out/prof_$R
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R/trace -o $R --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --data $DATA > gpurun_out/prof_$R/bench_under_rocprof.log 2>&1
echo "trace rc=$?"; tail -1 gpurun_out/prof_$R/bench_under_rocprof.log | cut -c1-400
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/prof_$R/pmc_$c -o $R --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bf16-leg --data $DATA > gpurun_out/prof_$R/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof_$R/pmc_TCC -o $R --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bf16-leg --data $DATA > gpurun_out/prof_$R/pmc_TCC.log 2>&1
echo "pmc TCC rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d gpurun_out/prof_$R/pmc_SQ -o $R --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bf16-leg --data $DATA > gpurun_out/prof_$R/pmc_SQ.log 2>&1
echo "pmc SQ rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d gpurun_out/prof_$R/pmc_SQ2 -o $R --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-bf16-leg --data $DATA > gpurun_out/prof_$R/pmc_SQ2.log 2>&1
echo "pmc SQ2 rc=$?"
python - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/prof_$R/pmc_*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name'].split('(')[0][:60]
        if 'mmf' in kn:
            acc[kn][r['Counter_Name']] += float(r['Counter_Value'])
    for kn, d in acc.items():
        out.setdefault(kn, {}).update(d)
print(json.dumps(out, indent=1)[:3000])
json.dump(out, open('gpurun_out/prof_$R/pmc_summary.json', 'w'), indent=1)
# HBM-side bytes per scan launch (gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; both counters are in KiB)
for kn, d in out.items():
    if 'scan_b16x' in kn and 'FETCH_SIZE' in d:
        fetch, write = 2.0 * d['FETCH_SIZE'] * 1024.0, d.get('WRITE_SIZE', 0.0) * 1024.0
        json.dump({"kernel": kn, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write, "round": "$R", "data": "$DATA",
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 1 launch each; FETCH_SIZE x 1024 x 2 per the gfx950 correction"},
                  open('gpurun_out/prof_$R/traffic_entry.json', 'w'), indent=1)
PY
head -8 gpurun_out/prof_$R/trace/${R}_kernel_stats.csv | cut -c1-200
