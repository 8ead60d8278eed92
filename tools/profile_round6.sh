# Round-6 evidence run on the GPU box -- run as the LAST step of the round, after the last kernel change:
#     gpurun --timeout 2400 -- "MVIP_HEAD=$(git rev-parse HEAD) bash tools/profile_round6.sh"
# (the snapshot carries no .git: the commit the files belong to comes in through MVIP_HEAD and is written INTO every JSON the
# script produces, and into r6_profile_head.json for the CSV / text files).  Every file lands in gpurun_out/ under the name it is
# committed as in profiles/:
#  (1) rocprofv3 --kernel-trace --stats of the DEFAULT bench command        -> r6_bench_kernel_stats.csv, r6_bench_line_under_rocprof.json
#  (2) matrix-pipe utilisation per kernel (PMC) of a short bench            -> r6_pmc_mfma_util.json        (render, training, SDS kernels)
#  (3) HBM traffic of the dominant render launch (FETCH_SIZE / WRITE_SIZE)  -> r6_pmc_mlp_forward.json
#  (4) HBM traffic of STEADY-STATE SDS steps per kernel                     -> r6_pmc_sds_traffic.json
#  (5) one SDS step per kernel + hipGraph replay, fp32 networks and --fp16  -> r6_sds_step_f32.json, r6_sds_step_fp16.json
#  (6) isolated HBM-bound stage kernels                                      -> r6_micro_hbm_kernels.jsonl
#  (7) configs[2] / configs[3] iterations per kernel                         -> r6_config2_step_kernels.json, r6_config3_step_kernels.json
#  (8) the round's A/B on the final build: LayerNorm statistics from the GEMM epilogue x GroupNorm moments from the
#      unsplit convolution's epilogue, each on / off                        -> r6_sds_fusion_ab.json
#      (the sample_pdf_merge experiment's A/B and PMC view were taken on commit dbca558, which holds the experiment's code:
#       profiles/r6_sample_merge_ab.jsonl, profiles/r6_pmc_stage_kernels.json)
# PMC runs are their own processes with --kernel-trace only (never combined with --stats / sys-trace), with the SDS steps launched
# kernel by kernel (MVIP_SDS_GRAPHS=0) and under their own timeouts.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
HEAD=${MVIP_HEAD:-unknown}
mkdir -p gpurun_out/prof_r6 gpurun_out/pmc_r6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r6 -o run -- python3 bench.py > gpurun_out/r6_bench_line_under_rocprof.json 2> gpurun_out/prof_r6/err.log
find gpurun_out/prof_r6 -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/r6_bench_kernel_stats.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:41]:
    r[0]=r[0][:110]; w.writerow(r)
"
find gpurun_out/prof_r6 -name '*kernel_trace.csv' -delete; find gpurun_out/prof_r6 -name '*.db' -delete
MVIP_HEAD=$HEAD bash tools/pmc_mfma_util.sh > gpurun_out/pmc_r6/summary_top.txt 2>&1     # three bounded passes (see the script)
cd /tmp; cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=gpurun_out/pmc_r6_$c; mkdir -p $D
  MVIP_SDS_GRAPHS=0 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid > $D/line.json 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} --longest mlp_forward16_kernel > $D/dominant.json
  find $D -name '*.csv' -delete; find $D -name '*.db' -delete
  D=gpurun_out/pmc_r6_sds_$c; mkdir -p $D
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 tools/sds_profile_steps.py 5 > $D/out.txt 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $D/cc.csv
  find $D -name '*.db' -delete
done
python3 - <<'P'
import json
f = json.load(open('gpurun_out/pmc_r6_fetch/dominant.json')); w = json.load(open('gpurun_out/pmc_r6_write/dominant.json'))
pts = 190512 * 128
json.dump({'kernel': 'mvip::f16p::mlp_forward16_kernel<true,false,0> (fine pass of one 378x504 frame: 190,512 rays x 128 samples; the longest dispatch)',
           'fetch_size_KB': f['value_KB'], 'write_size_KB': w['value_KB'], 'hbm_bytes_per_launch': (2 * f['value_KB'] + w['value_KB']) * 1024,
           'launch_ms': max(f['ms'], w['ms']), 'algorithmic_bytes': pts * 20 + 190512 * 44,
           'command': 'rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid; longest mlp_forward16_kernel dispatch (tools/pmc_summary.py --longest); FETCH doubled per the gfx950 note'},
          open('gpurun_out/r6_pmc_mlp_forward.json', 'w'), indent=1)
P
python3 tools/pmc_sds_traffic.py gpurun_out/pmc_r6_sds_fetch/cc.csv gpurun_out/pmc_r6_sds_write/cc.csv 5 gpurun_out/r6_pmc_sds_traffic.json 2 > gpurun_out/r6_pmc_sds_traffic.txt 2>&1
find gpurun_out/pmc_r6_sds_fetch gpurun_out/pmc_r6_sds_write -name '*.csv' -delete
python3 tools/sds_step_profile.py --graphs --out=r6_sds_step_f32.json > gpurun_out/r6_sds_step_f32.txt 2>&1
python3 tools/sds_step_profile.py --fp16 --graphs --out=r6_sds_step_fp16.json > gpurun_out/r6_sds_step_fp16.txt 2>&1
python3 tools/micro_bench.py 2>/dev/null | grep '^{' > gpurun_out/r6_micro_hbm_kernels.jsonl
python3 tools/config_step_profile.py 2 > gpurun_out/r6_config2.txt 2>&1
python3 tools/config_step_profile.py 3 > gpurun_out/r6_config3.txt 2>&1
python3 tools/sds_fusion_ab.py > gpurun_out/r6_sds_fusion_ab.txt 2>&1
# ---- stamp: the commit goes INTO every JSON (a list becomes {"head", "rows"}), and next to the CSV / text files ----
MVIP_HEAD=$HEAD python3 - <<'P'
import glob, json, os
head = os.environ.get('MVIP_HEAD', 'unknown')
stamped = []
for p in sorted(glob.glob('gpurun_out/r6_*.json')):
    if p.endswith('r6_profile_head.json'):
        continue
    try:
        d = json.load(open(p))
    except Exception:
        continue
    if isinstance(d, list):
        d = {'head': head, 'rows': d}
    else:
        d['head'] = head
    json.dump(d, open(p, 'w'), indent=1)
    stamped.append(os.path.basename(p))
others = sorted(os.path.basename(p) for p in glob.glob('gpurun_out/r6_*') if not p.endswith('.json'))
json.dump({'head': head, 'what': 'every gpurun_out/r6_* file below was produced by ONE run of tools/profile_round6.sh on a snapshot of this commit',
           'json_files_stamped': stamped, 'other_files': others}, open('gpurun_out/r6_profile_head.json', 'w'), indent=1)
P
tail -c 400 gpurun_out/r6_bench_line_under_rocprof.json; echo; head -c 1500 gpurun_out/pmc_r6/summary_top.txt; cat gpurun_out/r6_pmc_mlp_forward.json; head -18 gpurun_out/r6_pmc_sds_traffic.txt
