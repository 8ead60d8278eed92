# Round 4 diagnostics: (1) timing-only modes of the 3x3 convolution in its TWO-product form (experiment build in
# lib_experiment/), (2) PMC instruction mix / wait split of one SDS step per kernel, (3) PMC instruction counts of the stage kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
MVIP_LIB_PATH=$R/mvip_nerf_amd/lib_experiment/libmvipnerf.so MVIP_ALLOW_EXPERIMENT_BUILD=1 MVIP_CONV_SUSTAIN=30 timeout 900 python tools/conv_experiment.py > gpurun_out/r4_conv_experiment.txt 2>&1; cat gpurun_out/r4_conv_experiment.txt
PMC_DIAG_WORKLOADS=sds timeout 900 bash tools/pmc_diag.sh > gpurun_out/r4_pmc_diag_sds.txt 2>&1; tail -12 gpurun_out/r4_pmc_diag_sds.txt
D=gpurun_out/pmc_micro; mkdir -p $D
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $D -o run -- python3 tools/micro_bench.py > $D/out.txt 2> $D/err.log
find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} $D/summary.json > $D/summary_top.txt
find $D -name '*.csv' -delete; find $D -name '*.db' -delete
python3 - <<'P'
import json
for e in json.load(open('gpurun_out/pmc_micro/summary.json')):
    if 'sample_pdf' in e['kernel'] or 'composite' in e['kernel']:
        w = e.get('SQ_WAVES', 1) or 1
        print(e['kernel'][:50], 'dispatches', e['dispatches'], 'per wave: valu', round(e.get('SQ_INSTS_VALU', 0) / w), 'salu', round(e.get('SQ_INSTS_SALU', 0) / w), 'lds', round(e.get('SQ_INSTS_LDS', 0) / w),
              'wave_cycles', round(e.get('SQ_WAVE_CYCLES', 0) / w), 'wait_inst', round(e.get('SQ_WAIT_INST_ANY', 0) / w), 'active_valu', round(e.get('SQ_ACTIVE_INST_VALU', 0) / w))
P
