#!/bin/bash
# Counter pass over a config-2 render (refract_cow_cbox 1024x1024, 16 spp) for the shipped library and for tuning variants
# (tools/build_variant.sh), one rocprofv3 run each:  bash tools/profile_experiments.sh VARIANT [VARIANT ...]
# Leaves gpurun_out/prof_exp/<name>/; tools/summarise_experiments.py reads them together with the A/B logs.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_exp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in shipped "$@"; do
	if [ "$v" = shipped ]; then unset PTX_AMD_LIB; else export PTX_AMD_LIB=$R/pathtracer_amd/lib/variants/libptx_$v.so; fi
	for leg in refract_cow_cbox instanced_cows; do
		rm -rf $OUT/${v}_$leg
		timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/${v}_$leg -o p -- python3 $R/tools/prof_render.py 16 $leg > $OUT/${v}_$leg.log 2>&1
	done
	echo "[experiments] $v done"
done
python3 $R/tools/reduce_profiles.py $OUT
