cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r2 $R/gpurun_out/pmc_r2
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2 -o run -- python3 bench.py > gpurun_out/prof_r2/bench_line.json 2> gpurun_out/prof_r2/err.log
find gpurun_out/prof_r2 -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv,sys
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/prof_r2/top30.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:31]:
    r[0]=r[0][:110]; w.writerow(r)
"
find gpurun_out/prof_r2 -name '*kernel_trace.csv' -delete
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_r2 -o run -- python3 bench.py --steps 1 --warmup 0 --train-steps 1 --sds-steps 1 --no-cpu-baseline --no-hashgrid > gpurun_out/pmc_r2/line.json 2> gpurun_out/pmc_r2/err.log
find gpurun_out/pmc_r2 -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} gpurun_out/pmc_r2/summary.json > gpurun_out/pmc_r2/summary_top.txt
find gpurun_out/pmc_r2 -name '*.csv' -delete
tail -c 600 gpurun_out/prof_r2/bench_line.json; head -c 2500 gpurun_out/pmc_r2/summary_top.txt
