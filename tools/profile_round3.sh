# Round-3 evidence run on the GPU box: (1) rocprofv3 --kernel-trace --stats of the DEFAULT bench command, (2) matrix-pipe
# utilisation per kernel (PMC) of a short bench, (3) HBM traffic of the dominant render launch and of ONE SDS step per kernel
# (FETCH_SIZE / WRITE_SIZE in separate passes), (4) per-kernel torch-profiler views.  Summaries land in gpurun_out/ and are
# copied into profiles/ as r3_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_r3 gpurun_out/pmc_r3 gpurun_out/pmc_r3_fetch gpurun_out/pmc_r3_write gpurun_out/pmc_r3_sds_fetch gpurun_out/pmc_r3_sds_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3 -o run -- python3 bench.py > gpurun_out/prof_r3/bench_line.json 2> gpurun_out/prof_r3/err.log
find gpurun_out/prof_r3 -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/prof_r3/top40.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:41]:
    r[0]=r[0][:110]; w.writerow(r)
"
find gpurun_out/prof_r3 -name '*kernel_trace.csv' -delete; find gpurun_out/prof_r3 -name '*.db' -delete
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r3 -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 1 --sds-steps 1 --no-cpu-baseline --no-hashgrid > gpurun_out/pmc_r3/line.json 2> gpurun_out/pmc_r3/err.log
find gpurun_out/pmc_r3 -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} gpurun_out/pmc_r3/summary.json > gpurun_out/pmc_r3/summary_top.txt
find gpurun_out/pmc_r3 -name '*.csv' -delete; find gpurun_out/pmc_r3 -name '*.db' -delete
for C in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/pmc_r3_$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid > $D/line.json 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} --longest mlp_forward16_kernel > $D/dominant.json
  find $D -name '*.csv' -delete; find $D -name '*.db' -delete
  D=gpurun_out/pmc_r3_sds_$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 tools/sds_profile_steps.py 3 > $D/out.txt 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $D/cc.csv
  find $D -name '*.db' -delete
done
python3 tools/pmc_sds_traffic.py gpurun_out/pmc_r3_sds_fetch/cc.csv gpurun_out/pmc_r3_sds_write/cc.csv 3 gpurun_out/r3_pmc_sds_traffic.json > gpurun_out/r3_pmc_sds_traffic.txt 2>&1
find gpurun_out/pmc_r3_sds_fetch gpurun_out/pmc_r3_sds_write -name '*.csv' -delete
python3 tools/sds_step_profile.py --graphs --out=r3_sds_step_f32.json > gpurun_out/r3_sds_step_f32.txt 2>&1
python3 tools/sds_step_profile.py --fp16 --graphs --out=r3_sds_step_fp16.json > gpurun_out/r3_sds_step_fp16.txt 2>&1
python3 tools/conv_kernel_times.py > gpurun_out/r3_conv_kernel_times.txt 2>&1
tail -c 300 gpurun_out/prof_r3/bench_line.json; echo; head -c 1500 gpurun_out/pmc_r3/summary_top.txt; cat gpurun_out/pmc_r3_fetch/dominant.json gpurun_out/pmc_r3_write/dominant.json; head -16 gpurun_out/r3_pmc_sds_traffic.txt
