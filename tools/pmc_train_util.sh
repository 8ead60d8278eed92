# matrix-pipe utilisation per kernel of the training iteration (tools/train_speed.py): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
D=gpurun_out/pmc_train_util; mkdir -p $D
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o run -- python3 tools/train_speed.py > $D/out.txt 2> $D/err.log
find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} $D/summary.json > $D/summary_top.txt
find $D -name '*.csv' -delete; find $D -name '*.db' -delete
head -c 3000 $D/summary_top.txt
