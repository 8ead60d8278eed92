cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_r6_timeline; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -o run -- python3 tools/sds_replay_timeline.py --run 8 > $D/out.txt 2> $D/err.log
F=$(find $D -name '*kernel_trace.csv' | head -1)
echo trace $F
python3 tools/sds_replay_timeline.py $F gpurun_out/r6_sds_replay_timeline.json
find $D -name '*.csv' -delete; find $D -name '*.db' -delete
