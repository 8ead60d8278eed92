# Matrix-pipe utilisation per kernel, SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128), in THREE bounded counter passes --
# the render legs of bench.py, the training iteration (tools/train_speed.py) and eager SDS steps (tools/sds_profile_steps.py) --
# merged into gpurun_out/${MVIP_ROUND:-r6}_pmc_mfma_util.json.  Round 5: the one-pass form (the whole bench under --pmc) took rocprofv3 down
# twice (a SIGSEGV inside the tool, then "AQL packet is malformed" followed by a hang until the timeout), so each leg is its
# own process with a hard KILL timeout; the three commands are recorded in the output.
#   gpurun -- "MVIP_HEAD=$(git rev-parse HEAD) bash tools/pmc_mfma_util.sh"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HEAD=${MVIP_HEAD:-unknown}
PMC="--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv"
pass() {   # name, then the program
  D=gpurun_out/pmc_util_$1; shift; mkdir -p $D
  MVIP_SDS_GRAPHS=0 MVIP_SDS_TWO_STREAMS=0 MVIP_SDS_TERM_STREAMS=0 timeout -s KILL ${PASS_LIMIT:-300} rocprofv3 $PMC -d $D -o run -- "$@" > $D/out.txt 2> $D/err.log
  echo "$D rc=$?"
  find $D -name '*counter_collection.csv' | head -1 | xargs -I{} python3 tools/pmc_summary.py {} $D/summary.json > $D/summary_top.txt
  find $D -name '*.csv' -delete; find $D -name '*.db' -delete
}
pass render python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid
pass train python3 tools/train_speed.py
pass sds python3 tools/sds_profile_steps.py 3
MVIP_HEAD=$HEAD python3 - <<'P'
import json, os
out = {'head': os.environ['MVIP_HEAD'],
       'what': 'mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) per kernel name, three counter passes '
               '(tools/pmc_mfma_util.sh); MVIP_SDS_GRAPHS=0 MVIP_SDS_TWO_STREAMS=0 MVIP_SDS_TERM_STREAMS=0 (one stream, eager)',
       'passes': {}}
cmds = {'render': 'python3 bench.py --steps 1 --warmup 0 --train-steps 0 --sds-steps 0 --no-cpu-baseline --no-hashgrid',
        'train': 'python3 tools/train_speed.py', 'sds': 'python3 tools/sds_profile_steps.py 3'}
for name, cmd in cmds.items():
    p = f'gpurun_out/pmc_util_{name}/summary.json'
    rows = json.load(open(p)) if os.path.exists(p) else None
    if rows is not None:
        rows = sorted(rows, key=lambda r: -r['total_ms'])[:40]
    out['passes'][name] = {'command': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- ' + cmd,
                           'rows': rows}
json.dump(out, open('gpurun_out/' + os.environ.get('MVIP_ROUND', 'r6') + '_pmc_mfma_util.json', 'w'), indent=1)
for name, p in out['passes'].items():
    print(name, 'MISSING' if p['rows'] is None else '')
    for r in (p['rows'] or [])[:8]:
        print('  %-70s n=%-5d %9.2f ms  util=%s' % (r['kernel'][:70], r['dispatches'], r['total_ms'], r.get('mfma_pipe_util')))
P
