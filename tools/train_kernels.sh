# per-kernel time of the second-stage training iteration (tools/train_speed.py) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/train_kernels
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_kernels -o run -- python3 tools/train_speed.py > gpurun_out/train_kernels/line.json 2> gpurun_out/train_kernels/err.log
find gpurun_out/train_kernels -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/train_kernels/top30.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:31]:
    r[0]=r[0][:100]; w.writerow(r)
"
find gpurun_out/train_kernels -name '*kernel_trace.csv' -delete
find gpurun_out/train_kernels -name '*.db' -delete
cat gpurun_out/train_kernels/line.json; cut -c1-200 gpurun_out/train_kernels/top30.csv | head -24
