# Round 4, first GPU run: the -m gpu suite, the SDS step per kernel with two and with three products, the graph-vs-eager
# draw diagnostic, the render-path A/B and the default bench line.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4a_pytest.txt
timeout 600 python tools/sds_step_profile.py --graphs --out=r4a_sds_step_f32.json --sequence=r4a_sds_seq.json > gpurun_out/r4a_sds_step_f32.txt 2>&1; head -30 gpurun_out/r4a_sds_step_f32.txt; tail -2 gpurun_out/r4a_sds_step_f32.txt
timeout 600 python tools/sds_step_profile.py --graphs --three --out=r4a_sds_step_f32_three.json > gpurun_out/r4a_sds_step_f32_three.txt 2>&1; head -8 gpurun_out/r4a_sds_step_f32_three.txt; tail -2 gpurun_out/r4a_sds_step_f32_three.txt
timeout 600 python tools/sds_step_profile.py --fp16 --graphs --out=r4a_sds_step_fp16.json > gpurun_out/r4a_sds_step_fp16.txt 2>&1; head -3 gpurun_out/r4a_sds_step_fp16.txt; tail -1 gpurun_out/r4a_sds_step_fp16.txt
timeout 600 python tools/graph_vs_eager_draws.py > gpurun_out/r4a_draws.txt 2>&1; tail -22 gpurun_out/r4a_draws.txt
timeout 600 python tools/fused_render_ab.py > gpurun_out/r4a_fused_ab.txt 2>&1; python - <<'P'
import json
d=json.load(open('gpurun_out/r4_fused_render_ab.json'))
print({k:(round(v['frame_378x504_ms'],2), round(v['render_rays_1024_ms'],3), v['frame_378x504_launches']) for k,v in d.items()})
P
timeout 1200 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"; python - <<'P'
import json
d=json.loads(open('gpurun_out/r4a_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['sds']['ms_per_step'], d['sds'].get('ms_per_step_fp16_hipgraph'), d.get('config2_rgb_normal_sds'), d.get('config3_rgb_normal_colla_sds'), d['train'], d.get('train_with_sds'))
P
