# A/B of the fused render tail: the normal build vs a timing-only build without the tail (results missing there).
# The experiment build goes to mvip_nerf_amd/lib_experiment/ (csrc/build.py) and is opened through MVIP_LIB_PATH +
# MVIP_ALLOW_EXPERIMENT_BUILD=1: the product library in lib/ is never overwritten.
set -e
cd $GRAFT_REPO_ROOT
python tools/fused_render_ab.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('normal build:', {k:(round(v['frame_378x504_ms'],1), round(v['render_rays_1024_ms'],3)) for k,v in d.items()})"
MVIP_EXTRA_FLAGS=-DMVIP_EXPERIMENT_NO_FUSE_TAIL python -m mvip_nerf_amd.csrc.build > /dev/null 2>&1
MVIP_LIB_PATH=$GRAFT_REPO_ROOT/mvip_nerf_amd/lib_experiment/libmvipnerf.so MVIP_ALLOW_EXPERIMENT_BUILD=1 \
python tools/fused_render_ab.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('no-tail build:', {k:(round(v['frame_378x504_ms'],1), round(v['render_rays_1024_ms'],3)) for k,v in d.items()})"
