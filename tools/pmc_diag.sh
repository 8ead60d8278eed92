# Where do the waves of the less-than-peak kernels spend their cycles?  Two PMC passes (8 SQ counters each) over
# (a) the training iteration and (b) one SDS step; per kernel: WAVE_CYCLES split into WAIT_ANY (parked on s_waitcnt /
# barrier), WAIT_INST_ANY (issue stall), ACTIVE_INST_ANY, plus LDS / VMEM instruction mix.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for W in ${PMC_DIAG_WORKLOADS:-train sds}; do
  if [ $W = train ]; then PROG="tools/train_speed.py"; else PROG="tools/sds_profile_steps.py"; fi
  for P in A B; do
    if [ $P = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES";
    else C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; fi
    D=gpurun_out/pmc_diag_${W}_$P; mkdir -p $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 $PROG > $D/out.txt 2> $D/err.log
    find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} $D/summary.json > $D/summary_top.txt
    find $D -name '*.csv' -delete; find $D -name '*.db' -delete
  done
done
python3 - <<'PY'
import json
import os
for W in os.environ.get('PMC_DIAG_WORKLOADS', 'train sds').split():
    a = {e['kernel']: e for e in json.load(open(f'gpurun_out/pmc_diag_{W}_A/summary.json'))}
    b = {e['kernel']: e for e in json.load(open(f'gpurun_out/pmc_diag_{W}_B/summary.json'))}
    print('==', W)
    for k, e in sorted(a.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:9]:
        wc = e.get('SQ_WAVE_CYCLES', 1) or 1
        f = b.get(k, {})
        print(k[:60].ljust(60), 'ms', round(e['total_ms'], 1), 'wait_any', round(e.get('SQ_WAIT_ANY', 0) / wc, 2), 'wait_inst', round(e.get('SQ_WAIT_INST_ANY', 0) / wc, 2),
              'active', round(e.get('SQ_ACTIVE_INST_ANY', 0) / wc, 2), 'wait_lds', round(e.get('SQ_WAIT_INST_LDS', 0) / wc, 3),
              '| per MFMA: valu', round(f.get('SQ_INSTS_VALU', 0) / max(f.get('SQ_INSTS_MFMA', 1), 1), 2), 'lds', round(f.get('SQ_INSTS_LDS', 0) / max(f.get('SQ_INSTS_MFMA', 1), 1), 2),
              'vmem', round((f.get('SQ_INSTS_VMEM_RD', 0) + f.get('SQ_INSTS_VMEM_WR', 0)) / max(f.get('SQ_INSTS_MFMA', 1), 1), 3), 'salu', round(f.get('SQ_INSTS_SALU', 0) / max(f.get('SQ_INSTS_MFMA', 1), 1), 2),
              'bank_conf/lds', round(f.get('SQ_LDS_BANK_CONFLICT', 0) / max(f.get('SQ_INSTS_LDS', 1), 1), 2))
PY
