#!/bin/bash
# timing-only ablations of the scan kernel + PMC counters (results of debug modes are wrong by design)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dbg in 0 1 2 3 4 5 7; do
  echo -n "debug=$dbg " ; MMF_SCAN_DEBUG=$dbg timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scan_ms=%.2f TF=%.0f fallback=%d' % (j['roofline']['kernel_ms'], j['roofline']['achieved'], j['config']['fallback_rows']))"
done
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -c . gpurun_out/counters_list.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp -d gpurun_out/pmc_$tag -o pmc --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_*/*counter_collection.csv')):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'scan_b16' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value'])
    print(f, dict(acc))
PY
