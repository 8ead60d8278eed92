# per-kernel time of the hash-grid model's training iteration (tools/hashgrid_train_profile.py) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hg_train_kernels
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hg_train_kernels -o run -- python3 tools/hashgrid_train_profile.py 10 > gpurun_out/hg_train_kernels/line.txt 2> gpurun_out/hg_train_kernels/err.log
find gpurun_out/hg_train_kernels -name '*kernel_stats.csv' | head -1 | xargs -I{} python3 -c "
import csv
rows=list(csv.reader(open('{}')))
w=csv.writer(open('gpurun_out/hg_train_kernels/top20.csv','w'),quoting=csv.QUOTE_ALL)
w.writerow(rows[0])
for r in rows[1:21]:
    r[0]=r[0][:100]; w.writerow(r)
"
find gpurun_out/hg_train_kernels -name '*kernel_trace.csv' -delete
find gpurun_out/hg_train_kernels -name '*.db' -delete
tail -1 gpurun_out/hg_train_kernels/line.txt; cut -c1-170 gpurun_out/hg_train_kernels/top20.csv | head -16
