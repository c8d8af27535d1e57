#!/bin/bash
# Profiling recipe, run on the GPU box through gpurun:  bash tools/profile_gpu.sh [section ...]
# Sections: bench closest other closest10m render icache   (default: all but icache; a gpurun call is limited to 20 minutes, so the
# sections are usually run in two or three calls).
# Leaves rocprofv3 databases under gpurun_out/prof/; tools/summarise_profiles.py turns them into the summaries
# kept under profiles/.  Counter passes are runs of their own (kernel trace and counters are never combined),
# one pass per counter group, each under its own time limit; the program stands directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/pathtracer_amd/buildinfo.py > $OUT/source_hash.txt
SECTIONS="${@:-bench closest other closest10m render}"

# pmc <dir> <counters...> -- <program...>: one counter pass
pmc() {
	local d=$1; shift
	local counters=()
	while [ "$1" != "--" ]; do counters+=("$1"); shift; done
	shift
	rm -rf $OUT/$d
	timeout -s KILL 240 rocprofv3 --pmc "${counters[@]}" -d $OUT/$d -o p -- "$@" > $OUT/$d.log 2>&1
}

SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
SQ2="SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
MEM="TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"
SQR="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"

for section in $SECTIONS; do
case $section in
bench)
	# kernel trace + stats of the default bench command (without the 4K leg, which adds 8.5 G paths of the same kernels)
	rm -rf $OUT/trace
	timeout -k 10 700 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-4k > $OUT/bench_under_rocprof.log 2>&1
	echo "[profile] bench trace done"
	;;
closest)
	# culled closest-hit kernel on the 1M-triangle soup: fabric-side traffic, L2 hit rate, issue counters, vector-memory path
	P="python3 $R/tools/prof_closest.py 16777216 2 c soup1m"
	pmc c_fetch FETCH_SIZE -- $P
	pmc c_write WRITE_SIZE -- $P
	pmc c_l2 TCC_HIT_sum TCC_MISS_sum -- $P
	pmc c_sq1 $SQ1 -- $P
	pmc c_sq2 $SQ2 -- $P
	pmc c_mem $MEM -- $P
	echo "[profile] closest-hit counters done"
	;;
other)
	# the other two ray queries of the bench (closest hit in the reference order, occlusion in the culled order)
	P="python3 $R/tools/prof_closest.py 16777216 2 ra soup1m"
	pmc o_fetch FETCH_SIZE -- $P
	pmc o_l2 TCC_HIT_sum TCC_MISS_sum -- $P
	pmc o_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -- $P
	pmc o_mem TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -- $P
	echo "[profile] other ray queries' counters done"
	;;
closest10m)
	# the same queries on the two 10 M-triangle meshes (trees that live in HBM): culled closest hit and occlusion
	for mesh in cows10m soup10m; do
		P="python3 $R/tools/prof_closest.py 16777216 2 ca $mesh"
		pmc m_${mesh}_fetch FETCH_SIZE -- $P
		pmc m_${mesh}_write WRITE_SIZE -- $P
		pmc m_${mesh}_l2 TCC_HIT_sum TCC_MISS_sum -- $P
		pmc m_${mesh}_sq1 $SQ1 -- $P
		pmc m_${mesh}_sq2 $SQ2 -- $P
		pmc m_${mesh}_mem $MEM -- $P
		echo "[profile] $mesh counters done"
	done
	;;
render)
	# every kernel of a render, per benchmark leg: kernel trace, then the counter groups
	rm -rf $OUT/r_trace
	timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/r_trace -o t -- python3 $R/tools/prof_render.py 256 refract_cow_cbox > $OUT/r_trace.log 2>&1
	for leg in cbox refract_cow_cbox instanced_cows flat_cows; do
		P="python3 $R/tools/prof_render.py 16 $leg"
		pmc r_${leg}_sq1 $SQR -- $P
		pmc r_${leg}_sq2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE -- $P
		pmc r_${leg}_mem TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE -- $P
		pmc r_${leg}_fetch FETCH_SIZE -- $P
		pmc r_${leg}_write WRITE_SIZE -- $P
		pmc r_${leg}_l2 TCC_HIT_sum TCC_MISS_sum -- $P
		echo "[profile] render leg $leg counters done"
	done
	;;
icache)
	# instruction-cache behaviour of the render kernels (their code is 16 KB and 46 KB)
	for leg in refract_cow_cbox instanced_cows; do
		pmc i_${leg} SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 $R/tools/prof_render.py 16 $leg
	done
	echo "[profile] instruction cache counters done"
	;;
*)
	echo "unknown section $section"; exit 2
	;;
esac
done
python3 $R/tools/reduce_profiles.py $OUT
du -sh $OUT
