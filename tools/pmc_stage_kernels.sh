# Where do the waves of the HBM-bound stage kernels spend their cycles, and how many instructions do they issue per ray?
# Two PMC passes (8 SQ counters each) over tools/micro_bench.py, once per setting of an A/B switch of the sample_pdf_merge kernel:
#   default: MVIP_SAMPLE_PAIR = 2 (two step-interleaved rays per wave, what ships) vs 0 (the one-ray-per-wave kernel of round 5)
#            -> gpurun_out/r6_pmc_stage_kernels_rays_per_wave.json
#   MVIP_STAGE_VAR=MVIP_SAMPLE_COUNTING MVIP_STAGE_OUT=r6_pmc_stage_kernels.json: the switch of the round-6 EXPERIMENT commit
#            dbca558 (counting merge + rays-per-wave prefetch, measured no faster and reverted; profiles/r6_pmc_stage_kernels.json
#            was taken there).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
VAR=${MVIP_STAGE_VAR:-MVIP_SAMPLE_PAIR}; SET=${MVIP_STAGE_SETTINGS:-2 0}; export MVIP_STAGE_VAR=$VAR MVIP_STAGE_SETTINGS="$SET" MVIP_STAGE_OUT=${MVIP_STAGE_OUT:-r6_pmc_stage_kernels_rays_per_wave.json}
for CNT in $SET; do
  for P in A B; do
    if [ $P = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES";
    else C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; fi
    D=gpurun_out/pmc_stage_${CNT}_$P; mkdir -p $D
    export $VAR=$CNT
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 tools/micro_bench.py > $D/out.txt 2> $D/err.log
    find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} $D/summary.json > $D/summary_top.txt
    find $D -name '*.csv' -delete; find $D -name '*.db' -delete
  done
done
python3 - <<'PY'
import json, os
out = {'what': 'PMC view of the stage kernels over tools/micro_bench.py (21 timed + 1 warm-up launches per line of that script; counters summed over all launches of a kernel name): fractions of SQ_WAVE_CYCLES and instructions per wave', 'settings': {}}
for cnt in os.environ['MVIP_STAGE_SETTINGS'].split():
    a = {e['kernel']: e for e in json.load(open(f'gpurun_out/pmc_stage_{cnt}_A/summary.json'))}
    b = {e['kernel']: e for e in json.load(open(f'gpurun_out/pmc_stage_{cnt}_B/summary.json'))}
    rows = []
    for k, e in a.items():
        if not any(s in k for s in ('sample_pdf', 'composite', 'stratified', 'posenc')):
            continue
        wc = e.get('SQ_WAVE_CYCLES', 0) or 1
        f = b.get(k, {})
        waves = e.get('SQ_WAVES', 0) or 1
        rows.append({'kernel': k, 'dispatches': e['dispatches'], 'total_ms': round(e['total_ms'], 3),
                     'wait_any': round(e.get('SQ_WAIT_ANY', 0) / wc, 3), 'wait_inst_any': round(e.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
                     'active_inst_any': round(e.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3), 'wait_inst_lds': round(e.get('SQ_WAIT_INST_LDS', 0) / wc, 3),
                     'wave_cycles_per_wave': round(wc / waves, 1),
                     'valu_per_wave': round(f.get('SQ_INSTS_VALU', 0) / waves, 1), 'lds_per_wave': round(f.get('SQ_INSTS_LDS', 0) / waves, 1),
                     'salu_per_wave': round(f.get('SQ_INSTS_SALU', 0) / waves, 1),
                     'vmem_per_wave': round((f.get('SQ_INSTS_VMEM_RD', 0) + f.get('SQ_INSTS_VMEM_WR', 0)) / waves, 2),
                     'lds_bank_conflict_cycles_per_lds_inst': round(f.get('SQ_LDS_BANK_CONFLICT', 0) / max(f.get('SQ_INSTS_LDS', 1), 1), 2)})
    out['settings'][f"{os.environ['MVIP_STAGE_VAR']}={cnt}"] = rows
    for r in rows:
        print(cnt, json.dumps(r))
out['head'] = os.environ.get('MVIP_HEAD', 'unknown')
json.dump(out, open('gpurun_out/' + os.environ['MVIP_STAGE_OUT'], 'w'), indent=1)
PY
