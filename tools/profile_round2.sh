# Round-2 evidence run on the GPU box: (1) rocprofv3 --kernel-trace --stats of the DEFAULT bench command,
# (2) matrix-pipe utilisation per kernel (PMC) of a short bench, (3) HBM traffic of the dominant launch (two PMC passes),
# (4) one steady SDS step per kernel (torch profiler).  Summaries land in gpurun_out/ and are copied into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2 $R/gpurun_out/pmc_r2 $R/gpurun_out/pmc_r2_fetch $R/gpurun_out/pmc_r2_write
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -o run -- python3 bench.py > gpurun_out/prof_r2/bench_line.json 2> gpurun_out/prof_r2/err.log
find gpurun_out/prof_r2 -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv,sys
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/prof_r2/top40.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:41]:
    r[0]=r[0][:110]; w.writerow(r)
"
find gpurun_out/prof_r2 -name '*kernel_trace.csv' -delete
find gpurun_out/prof_r2 -name '*.db' -delete
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r2 -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 1 --sds-steps 1 --no-cpu-baseline --no-hashgrid > gpurun_out/pmc_r2/line.json 2> gpurun_out/pmc_r2/err.log
find gpurun_out/pmc_r2 -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} gpurun_out/pmc_r2/summary.json > gpurun_out/pmc_r2/summary_top.txt
find gpurun_out/pmc_r2 -name '*.csv' -delete; find gpurun_out/pmc_r2 -name '*.db' -delete
for C in FETCH_SIZE WRITE_SIZE; do
  D=gpurun_out/pmc_r2_$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid > $D/line.json 2> $D/err.log
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} --longest mlp_forward16_kernel > $D/dominant.json
  find $D -name '*.csv' -delete; find $D -name '*.db' -delete
done
python3 tools/sds_step_profile.py > gpurun_out/sds_step_profile.txt 2>&1
python3 tools/unet_profile.py hip > gpurun_out/unet_profile.txt 2>&1
tail -c 400 gpurun_out/prof_r2/bench_line.json; echo; head -c 1800 gpurun_out/pmc_r2/summary_top.txt; cat gpurun_out/pmc_r2_fetch/dominant.json gpurun_out/pmc_r2_write/dominant.json; head -5 gpurun_out/sds_step_profile.txt
