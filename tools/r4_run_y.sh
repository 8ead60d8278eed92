# Refresh of the SDS-step evidence on the final build; every step under its own timeout (a PMC pass over the whole bench hung for
# 49 minutes in the previous attempt -- rocprofv3 counter collection next to the graphed multi-view step -- so the PMC passes here
# profile eager steps only, as round 3 / 4 did).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for C in FETCH_SIZE WRITE_SIZE; do
  c=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=gpurun_out/pmc_r4_sds_$c; rm -rf $D; mkdir -p $D
  timeout 420 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o run -- python3 tools/sds_profile_steps.py 5 > $D/out.txt 2> $D/err.log
  echo "pmc $C rc=$?"
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} $D/cc.csv
  find $D -name '*.db' -delete
done
timeout 120 python3 tools/pmc_sds_traffic.py gpurun_out/pmc_r4_sds_fetch/cc.csv gpurun_out/pmc_r4_sds_write/cc.csv 5 gpurun_out/r4_pmc_sds_traffic.json 2 > gpurun_out/r4_pmc_sds_traffic.txt 2>&1
find gpurun_out/pmc_r4_sds_fetch gpurun_out/pmc_r4_sds_write -name '*.csv' -delete
head -12 gpurun_out/r4_pmc_sds_traffic.txt
timeout 300 python3 tools/sds_step_profile.py --graphs --out=r4_sds_step_f32.json > gpurun_out/r4_sds_step_f32.txt 2>&1
timeout 300 python3 tools/sds_step_profile.py --fp16 --graphs --out=r4_sds_step_fp16.json > gpurun_out/r4_sds_step_fp16.txt 2>&1
grep -E "median wall|hipGraph" gpurun_out/r4_sds_step_f32.txt gpurun_out/r4_sds_step_fp16.txt
