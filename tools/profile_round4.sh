# Round-4 evidence run on the GPU box.  Every file lands in gpurun_out/ under the name it is committed as in profiles/:
#  (1) rocprofv3 --kernel-trace --stats of the DEFAULT bench command        -> r4_bench_kernel_stats.csv, r4_bench_line_under_rocprof.json
#  (2) matrix-pipe utilisation per kernel (PMC) of a short bench            -> r4_pmc_mfma_util.json
#  (3) HBM traffic of the dominant render launch (FETCH_SIZE / WRITE_SIZE)  -> r4_pmc_mlp_forward.json
#  (4) HBM traffic of STEADY-STATE SDS steps per kernel (5 eager steps, the first two cut off as one-time work)
#                                                                            -> r4_pmc_sds_traffic.json
#  (5) one SDS step per kernel + hipGraph replay, fp32 networks (two products) and --fp16 mode -> r4_sds_step_f32.json, r4_sds_step_fp16.json
#  (6) isolated HBM-bound stage kernels                                      -> r4_micro_hbm_kernels.jsonl
# PMC runs are their own processes with --kernel-trace only (never combined with --stats / sys-trace), with the SDS steps launched
# kernel by kernel (MVIP_SDS_GRAPHS=0) and under their own timeouts: per-dispatch counter collection next to replayed hipGraphs of the
# multi-view step did not finish in 49 minutes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r4 gpurun_out/pmc_r4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4 -o run -- python3 bench.py > gpurun_out/r4_bench_line_under_rocprof.json 2> gpurun_out/prof_r4/err.log
find gpurun_out/prof_r4 -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/r4_bench_kernel_stats.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:41]:
    r[0]=r[0][:110]; w.writerow(r)
"
find gpurun_out/prof_r4 -name '*kernel_trace.csv' -delete; find gpurun_out/prof_r4 -name '*.db' -delete
MVIP_SDS_GRAPHS=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r4 -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 1 --sds-steps 1 --no-cpu-baseline --no-hashgrid > gpurun_out/pmc_r4/line.json 2> gpurun_out/pmc_r4/err.log
find gpurun_out/pmc_r4 -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} gpurun_out/r4_pmc_mfma_util.json > gpurun_out/pmc_r4/summary_top.txt
find gpurun_out/pmc_r4 -name '*.csv' -delete; find gpurun_out/pmc_r4 -name '*.db' -delete
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=gpurun_out/pmc_r4_$c; mkdir -p $D
  MVIP_SDS_GRAPHS=0 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid > $D/line.json 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} --longest mlp_forward16_kernel > $D/dominant.json
  find $D -name '*.csv' -delete; find $D -name '*.db' -delete
  D=gpurun_out/pmc_r4_sds_$c; mkdir -p $D
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 tools/sds_profile_steps.py 5 > $D/out.txt 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $D/cc.csv
  find $D -name '*.db' -delete
done
python3 - <<'P'
import json
f = json.load(open('gpurun_out/pmc_r4_fetch/dominant.json')); w = json.load(open('gpurun_out/pmc_r4_write/dominant.json'))
pts = 190512 * 128
json.dump({'kernel': 'mvip::f16p::mlp_forward16_kernel<true,false,0> (fine pass of one 378x504 frame: 190,512 rays x 128 samples; the longest dispatch)',
           'fetch_size_KB': f['value_KB'], 'write_size_KB': w['value_KB'], 'hbm_bytes_per_launch': (2 * f['value_KB'] + w['value_KB']) * 1024,
           'launch_ms': max(f['ms'], w['ms']), 'algorithmic_bytes': pts * 20 + 190512 * 44,
           'command': 'rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid; longest mlp_forward16_kernel dispatch (tools/pmc_summary.py --longest); FETCH doubled per the gfx950 note'},
          open('gpurun_out/r4_pmc_mlp_forward.json', 'w'), indent=1)
P
python3 tools/pmc_sds_traffic.py gpurun_out/pmc_r4_sds_fetch/cc.csv gpurun_out/pmc_r4_sds_write/cc.csv 5 gpurun_out/r4_pmc_sds_traffic.json 2 > gpurun_out/r4_pmc_sds_traffic.txt 2>&1
find gpurun_out/pmc_r4_sds_fetch gpurun_out/pmc_r4_sds_write -name '*.csv' -delete
python3 tools/sds_step_profile.py --graphs --out=r4_sds_step_f32.json > gpurun_out/r4_sds_step_f32.txt 2>&1
python3 tools/sds_step_profile.py --fp16 --graphs --out=r4_sds_step_fp16.json > gpurun_out/r4_sds_step_fp16.txt 2>&1
python3 tools/micro_bench.py 2>/dev/null | grep '^{' > gpurun_out/r4_micro_hbm_kernels.jsonl
tail -c 400 gpurun_out/r4_bench_line_under_rocprof.json; echo; head -c 1200 gpurun_out/pmc_r4/summary_top.txt; cat gpurun_out/r4_pmc_mlp_forward.json; head -18 gpurun_out/r4_pmc_sds_traffic.txt
